"""Where the skinny kernel stops paying: time per GEMM for the four L/14 block shapes at small M, default dispatch with the skinny
kernel off (OVHIP_GEMM_SKINNY_TILES=0) against the skinny kernel forced (OVHIP_GEMM_VARIANT=4); each in a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, torch
sys.path.insert(0, os.environ["OV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["OV_ROOT"], "tests"))
import hipops as H
for M in (257, 514, 1028, 2056, 4112, 8224, 16448):
    row = []
    for (N, K, epi) in ((3072, 1024, 0), (1024, 1024, 3), (4096, 1024, 1), (1024, 4096, 3)):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(5): H.gemm(a, w, b, epi=epi, resid=r if epi == 3 else None, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): H.gemm(a, w, b, epi=epi, resid=r if epi == 3 else None, out=out)
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(M, " ".join(f"{v:8.1f}" for v in row), flush=True)
'''
for name, env in (("default kernels (skinny off)", {"OVHIP_GEMM_SKINNY_TILES": "0"}), ("skinny forced", {"OVHIP_GEMM_VARIANT": "4"})):
    print(name, "-- us per launch: M | qkv out fc proj", flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OV_ROOT=ROOT, **env), check=True)
