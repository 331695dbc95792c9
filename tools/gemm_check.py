"""Diagnostic: persistent GEMM (all epilogues) against torch fp32 at tower shapes; prints where mismatches sit."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H
torch.manual_seed(0)
def check(M, N, K, epi, fold=False):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 3 else None
    if fold:
        g_ = torch.rand(K, device="cuda") + 0.5; b_ = torch.randn(K, device="cuda") * 0.1
        wg = (w.float() * g_[None, :]).to(torch.bfloat16)
        out = H.gemm_ln(a, wg, w.float() @ b_ + bias, wg.float().sum(1), H.rowstats(a), epi=epi)
        x = torch.nn.functional.layer_norm(a.float(), (K,), None, None, 1e-6)
        ref = x @ wg.float().T + (w.float() @ b_ + bias)
    else:
        out = H.gemm(a, w, bias, epi=epi, resid=r)
        ref = a.float() @ w.float().T + bias
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    if epi == 2: ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if epi == 3: ref = ref + r.float()
    err = (out.float() - ref).abs()
    bad = err > 0.06 + 0.02 * ref.abs()
    msg = f"M={M} N={N} K={K} epi={epi} fold={fold}: max err {err.max().item():.4f}, bad {int(bad.sum())}"
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        msg += f" rows[{rows.min().item()}..{rows.max().item()}] n={len(rows)} rows%256 {sorted(set((rows % 256).tolist()))[:20]} cols n={len(cols)} cols%256 {sorted(set((cols % 256).tolist()))[:20]} tiles_m {sorted(set((rows // 256).tolist()))[:12]}"
    print(msg, flush=True)
    if bad.any():
        Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
        bp = torch.zeros(Mp, Np, device="cuda"); bp[:M, :N] = bad.float()
        tb = bp.view(Mp // 256, 256, Np // 256, 256).sum((1, 3)) / 65536
        print("   tiles:", tb.numel(), "clean", int((tb == 0).sum()), "all-bad(>0.9)", int((tb > 0.9).sum()), "partial", int(((tb > 0) & (tb <= 0.9)).sum()))
        print("   first 3 tile rows bad fraction:\n", (tb[:3] * 100).round().int().tolist())
        ti = (tb > 0).nonzero()[0]
        blk = bad[ti[0] * 256:(ti[0] + 1) * 256, ti[1] * 256:(ti[1] + 1) * 256].float()
        print("   first bad tile", ti.tolist(), "bad per 16-row group:", blk.view(16, 16, 256).sum((1, 2)).int().tolist(), "per 64-col group:", blk.view(256, 4, 64).sum((0, 2)).int().tolist())
for args in [(65535, 1024, 1024, 0), (65535, 3072, 1024, 0), (65535, 1024, 1024, 3), (65535, 4096, 1024, 1), (65535, 1024, 4096, 3),
             (65536, 1024, 128, 0), (20000, 3072, 768, 2)]:
    check(*args)
for args in [(65535, 3072, 1024, 0), (65535, 4096, 1024, 1), (20000, 3072, 768, 2)]:
    check(*args, fold=True)
