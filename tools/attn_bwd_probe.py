"""attention backward launch time (B, L, H from the environment; default the L/14 shape)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H
B, L, Hh = int(os.environ.get("B", 256)), int(os.environ.get("L", 257)), int(os.environ.get("H", 16))
qkv = torch.randn(B * L, 3 * Hh * 64, device="cuda").to(torch.bfloat16); dout = torch.randn(B * L, Hh * 64, device="cuda").to(torch.bfloat16)
out = H.attention(qkv, B, L, Hh)
for _ in range(3): H.attention_backward(qkv, out, dout, B, L, Hh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): H.attention_backward(qkv, out, dout, B, L, Hh)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"attention_backward B {B} L {L} H {Hh}: {ms:.3f} ms = {10.0 * B * Hh * L * L * 64 / ms / 1e9:.0f} TFLOP/s (minimal 5 products)")
if L <= 288:
    out, lse = H.attention_lse(qkv, B, L, Hh)
    for _ in range(3): H.attention_backward_saved(qkv, out, dout, lse, B, L, Hh)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): H.attention_backward_saved(qkv, out, dout, lse, B, L, Hh)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"attention_backward_saved (row lse kept by the forward): {ms:.3f} ms = {10.0 * B * Hh * L * L * 64 / ms / 1e9:.0f} TFLOP/s")
