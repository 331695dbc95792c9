"""A/B two packing/env configurations of the step in ONE process, interleaved rounds (cdna guide rule 24).
usage: python tools/ab_bench.py KEY=VAL[,KEY=VAL] KEY=VAL[,...]   (env applied while each model is built / first run)"""
import os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth
from openvision_amd.model import create_model
from openvision_amd.loss import ClipLoss
cfg = preset("vit-large-patch14-224")
sd = synth.make_state_dict(cfg)
B = 256
img = synth.make_images(B, 224, seed=1000).to("cuda:0").to(torch.bfloat16)
tok = synth.make_captions(B, seed=1000).to("cuda:0")
arms = []
for spec in sys.argv[1:3]:
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    for k, v in env.items():
        os.environ[k] = v
    m = create_model(cfg, device="cuda:0", state_dict=sd)
    lf = ClipLoss()
    ni, nt, s = m(img, tok); lf(ni, nt, s); torch.cuda.synchronize()
    for k in env:
        os.environ.pop(k)
    arms.append((spec, m, lf, []))
for rnd in range(6):
    for spec, m, lf, ts in arms:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ni, nt, s = m(img, tok); l = lf(ni, nt, s)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e3)
for spec, m, lf, ts in arms:
    print(f"{spec or 'default':40s} median {statistics.median(ts):.3f} ms  min {min(ts):.3f}  all {[round(t,2) for t in ts]}")
