"""Half last n-tile of the persistent GEMM: every epilogue against fp32 torch at S/8-like shapes, then timings."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H
torch.manual_seed(0)
def check(M, N, K, epi, fold=False):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 3 else None
    if fold:
        out = H.gemm_ln(a, w, bias, w.float().sum(1), H.rowstats(a), epi=epi)
        ref = torch.nn.functional.layer_norm(a.float(), (K,), None, None, 1e-6) @ w.float().T + bias
    else:
        out = H.gemm(a, w, bias, epi=epi, resid=r)
        ref = a.float() @ w.float().T + bias
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    if epi == 3: ref = ref + r.float()
    err = (out.float() - ref).abs()
    bad = err > 0.06 + 0.02 * ref.abs()
    print(f"M={M} N={N} K={K} epi={epi} fold={fold}: max err {err.max().item():.4f} bad {int(bad.sum())}", flush=True)
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("   rows%256", sorted(set((rows % 256).tolist()))[:24], "cols", sorted(set(cols.tolist()))[:24], "n rows", len(rows))
for args in [(70001, 384, 384, 0), (70001, 384, 384, 1), (70001, 384, 384, 3), (66000, 1152, 192, 0), (66100, 632, 256, 3), (66100, 632, 256, 1)]:
    check(*args)
for args in [(70001, 384, 384, 0), (70001, 384, 384, 1), (66000, 1152, 384, 0)]:
    check(*args, fold=True)
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
    print(f"M={M} N={N} K={K} epi={epi}: {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
for N, K, epi in [(384, 1536, 3), (512, 1536, 3), (384, 384, 3), (512, 384, 3), (1152, 384, 0), (1280, 384, 0)]:
    t(N, K, epi)
