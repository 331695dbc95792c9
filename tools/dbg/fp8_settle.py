import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth
from openvision_amd.model import create_model
DEV = "cuda:0"
cfg = preset("vit-large-patch14-224")
m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, 0, "sharp"))
imgs = [synth.make_structured_images(256, 224, seed=60 + k).to(DEV).to(torch.bfloat16) for k in range(2)]
m.set_precision("fp8")
for k in range(2): m.encode_image(imgs[k])
m.visual.transformer.freeze_fp8_scales()
tw = m.visual.transformer.tower()
nl = tw.layers
prev = None
for it in range(5):
    outs = [m.encode_image(imgs[k], normalize=True) for k in range(2)]
    cur = tw.h_amax[:2 * nl].clone()
    print("iter", it, "scales changed vs previous iter:", None if prev is None else int((cur != prev).sum()),
          "same output as previous iter:", None if it == 0 else bool(torch.equal(outs[0], last[0]) and torch.equal(outs[1], last[1])))
    prev, last = cur, outs
part = m.encode_image(imgs[0][:100], normalize=True)
full = m.encode_image(imgs[0], normalize=True)
d = 1 - torch.nn.functional.cosine_similarity(part, full[:100])
print("part vs full[:100]: max 1-cos", float(d.max()), "bitwise equal rows:", int((part == full[:100]).all(dim=1).sum()), "of 100")
part2 = m.encode_image(imgs[0][:255], normalize=True)
print("B=255 vs full[:255] bitwise equal rows:", int((part2 == full[:255]).all(dim=1).sum()))
