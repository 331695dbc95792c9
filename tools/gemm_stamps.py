"""Diagnostic: per-tile timeline of the persistent GEMM from s_memtime stamps.  A tick is one SHADER cycle (MI355X_MICROARCH.md), so
the numbers below are in units of 100 cycles, and ticks / wall time of the stamped launch gives the clock the kernel actually ran at."""
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
for name, (N, K, epi) in cases.items():
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    x = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out = x.clone() if epi == 3 else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        H.gemm(a, w, bias, epi=epi, resid=out if epi == 3 else None, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        H.gemm(a, w, bias, epi=epi, resid=out if epi == 3 else None, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: avg of 20 launches {e0.elapsed_time(e1) * 50:.1f} us = {2 * M * N * K / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12:.0f} TFLOP/s")
    if os.environ.get("STAMPS", "1") == "0":
        continue
    slots = int(os.environ.get('SLOTS', '20'))
    SW = int(os.environ.get('STAMP_W', '8'))
    buf = torch.zeros(256 * slots * SW, dtype=torch.int64, device="cuda")
    lib.ov_debug_gemm_stamps(_lib.ptr(buf), slots)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); H.gemm(a, w, bias, epi=epi, resid=out if epi == 3 else None, out=out); e1.record()
    torch.cuda.synchronize()
    lib.ov_debug_gemm_stamps(None, 0)
    st = buf.cpu().numpy().reshape(256, slots, SW).astype(np.float64)
    ntile = int((st[:, :, 0] > 0).sum(1).max())
    st = st[:, :ntile]
    t0 = st[:, 0, 0].min()
    us = lambda v: v / 100.0          # s_memtime ticks are shader cycles: unit = 100 cycles
    main = us(st[:, :, 1] - st[:, :, 0]); align = us(st[:, :, 2] - st[:, :, 1]); epi_t = us(st[:, :, 3] - st[:, :, 2])
    gap = us(st[:, 1:, 0] - st[:, :-1, 3]) if ntile > 1 else np.zeros((256, 1))
    ok = st[:, 0, 0] > 0                                           # the XCDs' counters are not synchronised: span per workgroup
    span = float(np.median(st[ok][:, :, 3].max(axis=1) - st[ok][:, 0, 0]))   # cycles from its first tile start to its last epilogue end
    print(f"{name}: shader clock during the launch ~ {span / (e0.elapsed_time(e1) * 1e-3) / 1e9:.2f} GHz ({span:.0f} cycles in {e0.elapsed_time(e1)*1e3:.1f} us; "
          f"K-tile floor = 2048 MFMA cycles per SIMD)")
    print(f"{name}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, tiles/CU {ntile}, K-tiles {K//64} [unit: 100 cycles]: main {main.mean():.2f} "
          f"(min {main.min():.2f} max {main.max():.2f}) = {main.mean()/(K//64):.3f} us/K-tile; align {align.mean():.2f}; "
          f"epilogue {epi_t.mean():.2f} (min {epi_t.min():.2f} max {epi_t.max():.2f}); restart gap {gap.mean():.2f}; "
          f"first start spread {us(st[:,0,0].max()-t0):.2f}, last end spread {us(st[:,-1,3].max()-st[:,-1,3].min()):.2f}")
    if SW < 8:
        continue
    k0 = us(st[:, :, 4] - st[:, :, 0]); k1 = us(st[:, :, 5] - st[:, :, 4]); k23 = us(st[:, :, 6] - st[:, :, 5]) / 2
    k47 = us(st[:, :, 7] - st[:, :, 6]) / 4; rest = us(st[:, :, 1] - st[:, :, 7]) / max(K // 64 - 8, 1)
    print(f"   K-tile us: t0 {k0[:,1:].mean():.3f} (first tile {k0[:,0].mean():.3f}) t1 {k1[:,1:].mean():.3f} t2-3 {k23[:,1:].mean():.3f} "
          f"t4-7 {k47[:,1:].mean():.3f} rest {rest[:,1:].mean():.3f}")
    print("   per-tile main (CU 0):", np.round(main[0], 2), " epilogue (CU 0):", np.round(epi_t[0], 2))
