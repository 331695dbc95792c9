"""Is the training forward's loss bit-stable from step to step (no weight update in between)?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth, training
from openvision_amd.model import create_model
from openvision_amd.loss import ClipLoss
cfg = preset(os.environ.get("MODEL", "vit-large-patch14-224")); B = int(os.environ.get("BATCH", "256"))
m = create_model(cfg, device="cuda", state_dict=synth.make_state_dict(cfg))
S = cfg["vision_cfg"]["image_size"]
img = synth.make_images(B, S, seed=1).to("cuda"); tok = synth.make_captions(B, seed=1).to("cuda")
loss_fn = ClipLoss()
vals, gn = [], []
for i in range(6):
    m.zero_grad(set_to_none=True)
    fi, ft, sc = training.clip_forward(m, img, tok)
    loss = loss_fn(fi, ft, sc); loss.backward()
    vals.append(loss.detach().double().item())
    gn.append(sum(float(p.grad.double().pow(2).sum()) for p in m.parameters() if p.grad is not None))
print("loss per step:", ["%.9f" % v for v in vals])
print("grad sumsq per step:", ["%.9e" % v for v in gn])
with torch.no_grad():
    fi, ft = m.encode_image(img.to(torch.bfloat16), True), m.encode_text(tok, True)
    print("inference-path loss: %.9f" % loss_fn(fi, ft, m.logit_scale.exp()).double().item())
