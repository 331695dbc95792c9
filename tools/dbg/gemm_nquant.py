"""Probe: what the 256-wide n-tile costs at widths that are not multiples of 256 (S/8: D = 384 -> N = 384 / 1152 / 1536)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H
torch.manual_seed(0)
M = 295040
def t(N, K, epi):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 3 else None
    for _ in range(3): H.gemm(a, w, bias, epi=epi, resid=r)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): H.gemm(a, w, bias, epi=epi, resid=r)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"M={M} N={N} K={K} epi={epi}: {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s  (tiles_n {(N + 255) // 256}, fill {N / ((N + 255) // 256 * 256):.2f})", flush=True)
for N, K, epi in [(384, 1536, 3), (512, 1536, 3), (256, 1536, 3), (384, 384, 3), (512, 384, 3), (256, 384, 3), (1152, 384, 0), (1280, 384, 0), (1024, 384, 0),
                  (1536, 384, 1)]:
    t(N, K, epi)
