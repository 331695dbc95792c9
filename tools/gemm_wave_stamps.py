"""Diagnostic: per-wave timeline of the persistent GEMM's tile hand-over and epilogue (s_memtime, shader cycles)."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
# the s_memtime stamps are compiled out of the product library: this tool needs the variant build
#   python tools/build_variant.py stamps gemm.hip -DOVHIP_STAMPS=1        (on the build host; the .so travels with gpurun)
if "OVHIP_LIB" not in os.environ:
    if not os.path.exists(os.path.join(ROOT, "openvision_amd", "libovhip_stamps.so")):
        sys.exit("build the stamps variant first: python tools/build_variant.py stamps gemm.hip -DOVHIP_STAMPS=1")
    os.environ["OVHIP_LIB"] = "libovhip_stamps.so"
import hipops as H
from openvision_amd import _lib
lib = _lib.load()
M = 65535
cases = {"qkv": (3072, 1024, 0), "out": (1024, 1024, 3), "fc": (4096, 1024, 1), "proj": (1024, 4096, 3)}
slots = 6
for name in sys.argv[1:] or list(cases):
    N, K, epi = cases[name]
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    x = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out = x.clone() if epi == 3 else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(10):
        H.gemm(a, w, bias, epi=epi, resid=out if epi == 3 else None, out=out)
    buf = torch.zeros(256 * slots * 8, dtype=torch.int64, device="cuda")
    wbuf = torch.zeros(256 * slots * 8 * 8, dtype=torch.int64, device="cuda")
    lib.ov_debug_gemm_stamps(_lib.ptr(buf), slots); lib.ov_debug_gemm_wave_stamps(_lib.ptr(wbuf))
    H.gemm(a, w, bias, epi=epi, resid=out if epi == 3 else None, out=out)
    torch.cuda.synchronize()
    lib.ov_debug_gemm_stamps(None, 0); lib.ov_debug_gemm_wave_stamps(None)
    st = buf.cpu().numpy().reshape(256, slots, 8).astype(np.float64)
    ws = wbuf.cpu().numpy().reshape(256, slots, 8, 8).astype(np.float64)
    nt_ = 3 if (ws[:, 1:4, :, 6] > 0).all(axis=(1, 2)).any() else 2      # tiles 1..3 (>= 5 tiles per workgroup) or 1..2 (4 tiles)
    ok = (ws[:, 1:1 + nt_, :, 6] > 0).all(axis=(1, 2))
    ws = ws[ok][:, 1:1 + nt_]
    t0 = ws[:, :, :, 0].min(axis=2, keepdims=True)    # first wave out of the main loop
    names = ["main loop end", "epilogue_stream entry", "parameters read", "pass 1 done", "pass 4 done", "last store issued", "past tile barrier"]
    print(f"{name}: N={N} K={K} epi={epi}; cycles after the first wave left the main loop, mean over {ws.shape[0]} workgroups x {nt_} tiles")
    for k, nm in enumerate(names):
        d = ws[:, :, :, k] - t0
        print(f"   {nm:<24} per wave 0..7: " + " ".join(f"{v:7.0f}" for v in d.mean(axis=(0, 1))) + f"   | max over waves {d.max(axis=2).mean():7.0f}")
    tile = st[ok][:, 2, 0] - st[ok][:, 1, 0]
    print(f"   tile period (tile 1 start -> tile 2 start): {tile.mean():.0f} cycles; main loop {np.mean(st[ok][:, 1, 1] - st[ok][:, 1, 0]):.0f}")
