"""Diagnostic: bitwise repeatability of the L/14 towers at the bench batch (run under different OVHIP_* toggles)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth
from openvision_amd.model import create_model
cfg = preset("vit-large-patch14-224")
m = create_model(cfg, device="cuda:0", state_dict=synth.make_state_dict(cfg))
B = int(os.environ.get("B", "256"))
img = synth.make_images(B, 224, seed=5).to("cuda:0").to(torch.bfloat16)
tok = synth.make_captions(B, seed=5).to("cuda:0")
ref_i, ref_t = m.encode_image(img).clone(), m.encode_text(tok).clone()
bad = 0
for it in range(6):
    a, b = m.encode_image(img), m.encode_text(tok)
    di, dt = (a - ref_i).abs().max().item(), (b - ref_t).abs().max().item()
    rows = ((a != ref_i).any(dim=1)).nonzero().flatten().tolist()
    if di or dt:
        bad += 1
        print(f"  iter {it}: image maxdiff {di:.3e} rows {rows[:8]}{'...' if len(rows)>8 else ''} ({len(rows)} rows)  text maxdiff {dt:.3e}")
print({k: v for k, v in os.environ.items() if k.startswith("OVHIP")}, "B", B, "-> nondeterministic iterations:", bad)
