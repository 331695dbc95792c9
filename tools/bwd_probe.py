"""Launch times of the operator-level backward kernels at the L/14 block shapes (M = 256 * 257 rows)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H

M = int(os.environ.get("M", 256 * 257))
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("c_fc", 4096, 1024), ("c_proj", 1024, 4096)):
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    fl = 2.0 * M * N * K
    for what in (("dx",), ("dw",), ("db",)):
        ms = timeit(lambda: H.linear_backward(dy, x, w, want=what))
        extra = f" = {fl / ms / 1e9:.0f} TFLOP/s incl. transposes" if what[0] != "db" else f" = {M * N * 2 / ms / 1e6:.0f} GB/s"
        print(f"linear_backward {name} [{M} x {N} x {K}] {what[0]}: {ms:.3f} ms{extra}", flush=True)
    del dy, x, w
x = torch.randn(M, 1024, device="cuda").to(torch.bfloat16); dy = torch.randn(M, 1024, device="cuda").to(torch.bfloat16)
g = torch.ones(1024, device="cuda")
ms = timeit(lambda: H.layernorm_backward(x, g, dy))
print(f"layernorm_backward [{M} x 1024]: {ms:.3f} ms = {3 * M * 1024 * 2 / ms / 1e6:.0f} GB/s (x, dy read + dx written)")
a = torch.randn(M, 4096, device="cuda").to(torch.bfloat16); dh = torch.randn(M, 4096, device="cuda").to(torch.bfloat16)
for tanh in (False, True):
    ms = timeit(lambda: H.gelu_backward(a, dh, tanh))
    print(f"gelu_backward tanh={tanh} [{M} x 4096]: {ms:.3f} ms = {3 * M * 4096 * 2 / ms / 1e6:.0f} GB/s")
B, L, Hh = 256, 257, 16
qkv = torch.randn(B * L, 3 * Hh * 64, device="cuda").to(torch.bfloat16); dout = torch.randn(B * L, Hh * 64, device="cuda").to(torch.bfloat16)
out = H.attention(qkv, B, L, Hh)
ms = timeit(lambda: H.attention_backward(qkv, out, dout, B, L, Hh))
print(f"attention_backward B {B} L {L} H {Hh}: {ms:.3f} ms = {10.0 * B * Hh * L * L * 64 / ms / 1e9:.0f} TFLOP/s (minimal 5 products; forward: 4 L^2 64 per head)")
