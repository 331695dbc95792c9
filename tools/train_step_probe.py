"""Time of one training step (forward with saved block inputs + InfoNCE + backward of both towers) on ONE GPU, L/14@224 by default.
MODEL / BATCH environment variables select another preset / batch.  Not the headline metric (that is the forward step of bench.py)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth, training
from openvision_amd.model import create_model
from openvision_amd.loss import ClipLoss

name = os.environ.get("MODEL", "vit-large-patch14-224")
B = int(os.environ.get("BATCH", "256"))
cfg = preset(name)
m = create_model(cfg, device="cuda", state_dict=synth.make_state_dict(cfg))
S = cfg["vision_cfg"]["image_size"]
img = synth.make_images(B, S, seed=1).to("cuda")
tok = synth.make_captions(B, seed=1).to("cuda")
loss_fn = ClipLoss()

def step():
    m.zero_grad(set_to_none=True)
    fi, ft, sc = training.clip_forward(m, img, tok)
    loss = loss_fn(fi, ft, sc)
    loss.backward()
    return loss

def fwd_only():
    with torch.no_grad():
        fi, ft = m.encode_image(img.to(torch.bfloat16), True), m.encode_text(tok, True)
        return loss_fn(fi, ft, m.logit_scale.exp())

opt = None

def full_step():
    """forward keeping activations + InfoNCE + backward + FusedAdamW (global-norm clipping on), gradients in the optimiser's flat buffers"""
    opt.zero_grad()
    fi, ft, sc = training.clip_forward(m, img, tok)
    loss = loss_fn(fi, ft, sc)
    loss.backward()
    opt.step(grad_scale=opt.all_reduce_gradients(1))
    return loss

for fn, label in ((fwd_only, "inference forward + loss"), (step, "training step (forward-saving + loss + backward)"),
                  (full_step, "complete training step (+ FusedAdamW update, clipping on; weights re-packed every step)")):
    if fn is full_step:
        opt = training.FusedAdamW(m, lr=1e-6, clip_norm=1.0)
    fn(); torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name} B={B} {label}: {ms:.1f} ms = {B / ms * 1e3:.0f} img/s; loss {float(out.detach()):.4f}; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
