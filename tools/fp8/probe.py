"""Diagnostic: lane map and rate of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3, unit e8m0 scales) with exact integer data.
Build here:  hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o tools/fp8/libfp8probe.so tools/fp8/probe.hip"""
import ctypes, os, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libfp8probe.so"))
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ENC = {0: 0x00, 1: 0x38, 2: 0x40, 3: 0x44, 4: 0x48, -1: 0xB8, -2: 0xC0, -3: 0xC4, -4: 0xC8}     # OCP e4m3fn
rng = np.random.default_rng(0)
A = rng.integers(-4, 5, (16, 128)); B = rng.integers(-4, 5, (128, 16))
REF = (A @ B).astype(np.float32)
enc = np.vectorize(ENC.get)
hyp = {
    "k = 32*(l>>4) + j": lambda q, j: 32 * q + j,
    "k = 16*(l>>4) + (j&15) + 64*(j>>4)": lambda q, j: 16 * q + (j & 15) + 64 * (j >> 4),
    "k = 8*(l>>4) + (j&7) + 32*(j>>3)": lambda q, j: 8 * q + (j & 7) + 32 * (j >> 3),
    "k = 4*(l>>4) + (j&3) + 16*(j>>2)": lambda q, j: 4 * q + (j & 3) + 16 * (j >> 2),
}
for name, f in hyp.items():
    a = np.zeros((64, 32), np.uint8); b = np.zeros((64, 32), np.uint8)
    for l in range(64):
        for j in range(32):
            k = f(l >> 4, j)
            a[l, j] = enc(A[l & 15, k]); b[l, j] = enc(B[k, l & 15])
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.zeros(64, 4, device="cuda")
    rc = lib.fp8_probe(P(ta), P(tb), P(out), 127, 127, S()); torch.cuda.synchronize()
    o = out.cpu().numpy()
    got = np.zeros((16, 16), np.float32)
    for l in range(64):
        for r in range(4):
            got[(l >> 4) * 4 + r, l & 15] = o[l, r]                 # C/D: col = lane & 15, row = (lane >> 4) * 4 + reg
    print(f"{name:42s} rc={rc} match={np.array_equal(got, REF)} max|diff|={np.abs(got - REF).max():.0f}")
# scale semantics: 2^(e-127) per operand
for sa, sb in ((128, 127), (127, 129), (126, 126)):
    out = torch.zeros(64, 4, device="cuda")
    lib.fp8_probe(P(ta), P(tb), P(out), sa, sb, S()); torch.cuda.synchronize()
    print("scale_a", sa, "scale_b", sb, "ratio to unit-scale result (elementwise median):", float(np.median(out.cpu().numpy()[o != 0] / o[o != 0])))
# rate
blocks, iters = 256 * 8, 4000
buf = torch.zeros(blocks * 256, device="cuda")
lib.fp8_rate(P(buf), blocks, 10, S()); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); lib.fp8_rate(P(buf), blocks, iters, S()); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
flop = blocks * 4 * iters * 4 * 2.0 * 16 * 16 * 128
print(f"bare MX-fp8 16x16x128 loop: {flop / (ms * 1e-3) / 1e12:.0f} TFLOP/s ({ms:.2f} ms)")
