"""Probe: does running c_fc -> c_proj over row chunks whose hidden activations fit the 256 MB Infinity Cache beat one pass over all rows?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H
torch.manual_seed(0)
def run(M, D, F, chunks, iters=6):
    torch.manual_seed(1)
    x = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    w1 = (torch.randn(F, D, device="cuda") / D ** 0.5).to(torch.bfloat16); b1 = torch.randn(F, device="cuda") * 0.1
    w2 = (torch.randn(D, F, device="cuda") / F ** 0.5).to(torch.bfloat16); b2 = torch.randn(D, device="cuda") * 0.1
    cs = w1.float().sum(1); st = H.rowstats(x)
    hid = torch.empty(M, F, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    bounds = [(M * i // chunks // 256 * 256 if i < chunks else M) for i in range(chunks + 1)]
    def step():
        for i in range(chunks):
            lo, hi = bounds[i], bounds[i + 1]
            H.gemm_ln(x[lo:hi], w1, b1, cs, st[lo:hi], epi=1, out=hid[lo:hi])
            H.gemm(hid[lo:hi], w2, b2, epi=3, resid=x[lo:hi], out=out[lo:hi])
    for _ in range(2): step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"M={M} D={D} F={F} chunks={chunks}: {ms:.3f} ms per MLP  ({4.0 * M * D * F / ms / 1e9:.0f} TFLOP/s)  hidden per chunk {M * F * 2 / chunks / 1e6:.0f} MB", flush=True)
    return out
for M, D, F in ((295040, 384, 1536), (65792, 1024, 4096)):
    ref = None
    for c in (1, 2, 4, 8):
        o = run(M, D, F, c)
        if ref is None: ref = o
        else: assert torch.equal(o, ref), "chunked result differs"
