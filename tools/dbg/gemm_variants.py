"""Launch time of ov_gemm at the L/14 shapes for one OVHIP_GEMM_VARIANT (run once per variant: the switch is read once)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H
print("OVHIP_GEMM_VARIANT =", os.environ.get("OVHIP_GEMM_VARIANT", "0"))
for (M, N, K, epi) in [(65536, 3072, 1024, 0), (65536, 4096, 1024, 1), (65536, 1024, 4096, 3), (65536, 1024, 1024, 3)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    if os.environ.get("ZEROS") == "1":            # power check: zero operands draw less, the chip clocks higher
        a.zero_(); w.zero_()
    bias = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 3 else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ref = (a[:512].float() @ w.float().T + bias)
    H.gemm(a, w, bias, epi=epi, resid=r, out=out)
    if epi == 0:
        print("   max abs err vs fp32 (first 512 rows):", float((out[:512].float() - ref).abs().max()))
    for _ in range(3): H.gemm(a, w, bias, epi=epi, resid=r, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): H.gemm(a, w, bias, epi=epi, resid=r, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"   M={M} N={N} K={K} epi={epi}: {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s", flush=True)
