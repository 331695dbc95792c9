"""Diagnostic: ov_gemm_fp8 against fp32 matmul of the dequantised operands (exact products, fp32 accumulation), and its rate."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H
torch.manual_seed(0)
def run(M, N, K, epi, time_it=False):
    a = torch.randn(M, K, device="cuda") * (torch.rand(M, 1, device="cuda") * 3 + 0.2)
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    aq, rs = H.quantize_rows_e4m3(a); wq, cs = H.quantize_rows_e4m3(w)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 3 else None
    out = H.gemm_fp8(aq, wq, rs, cs, bias, epi, res)
    ad = aq.view(torch.float8_e4m3fn).float() * rs[:, None]; wd = wq.view(torch.float8_e4m3fn).float() * cs[:, None]
    ref = ad @ wd.T + bias
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    if epi == 3: ref = ref + res.float()
    err = (out.float() - ref).abs()
    tol = 0.02 + 0.01 * ref.abs()
    bad = int((err > tol).sum())
    msg = f"M={M} N={N} K={K} epi={epi}: max err {err.max().item():.4f} bad {bad}"
    if bad:
        rows = (err > tol).any(1).nonzero().flatten(); cols = (err > tol).any(0).nonzero().flatten()
        msg += f" rows {len(rows)} [{rows.min().item()}..{rows.max().item()}] cols {len(cols)} [{cols.min().item()}..{cols.max().item()}]"
    if time_it:
        for _ in range(3): H.gemm_fp8(aq, wq, rs, cs, bias, epi, res)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): H.gemm_fp8(aq, wq, rs, cs, bias, epi, res)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        msg += f"  {ms * 1e3:.1f} us = {2.0 * M * N * K / (ms * 1e-3) / 1e12:.0f} TFLOP/s"
    print(msg, flush=True)
for args in [(256, 256, 384, 0), (512, 768, 1024, 0), (300, 520, 512, 0), (4096, 4096, 1024, 1), (4096, 1024, 4096, 3)]:
    run(*args)
for args in [(65536, 3072, 1024, 0), (65536, 4096, 1024, 1), (65536, 1024, 4096, 3), (65536, 1024, 1024, 3), (65535, 4096, 1024, 1)]:
    run(*args, time_it=True)
