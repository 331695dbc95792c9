"""ov_adamw_step / ov_sumsq launch time and achieved HBM rate over the L/14 model's 414 M parameters (24 B / element for the update)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth, training
from openvision_amd.model import create_model
cfg = preset("vit-large-patch14-224")
m = create_model(cfg, device="cuda", state_dict=synth.make_state_dict(cfg))
for clip in (None, 1.0):
    opt = training.FusedAdamW(m, lr=1e-4, clip_norm=clip)
    n = sum(g["flat"].numel() for g in opt.groups)
    for g in opt.groups:
        g["grad"].normal_()
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        opt.step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byts = n * (24 + (4 if clip else 0))
    print(f"FusedAdamW.step over {n / 1e6:.1f} M parameters, clipping {'on' if clip else 'off'}: {ms:.3f} ms = {byts / ms / 1e6:.0f} GB/s "
          f"({byts / ms / 1e6 / 8000:.2f} of the 8 TB/s HBM peak)")
