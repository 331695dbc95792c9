import os, sys, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H
B, L, Hh = int(os.environ.get("B", "256")), int(os.environ.get("L", "257")), int(os.environ.get("H", "16"))
qkv = torch.randn(B * L, 3 * Hh * 64, device="cuda").to(torch.bfloat16)
for _ in range(3):
    H.attention(qkv, B, L, Hh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    H.attention(qkv, B, L, Hh)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("mode", os.environ.get("OVHIP_ATTN_MODE", "0"), "B", B, "L", L, "H", Hh, "ms/launch", ms, "TFLOP/s", 4.0 * B * Hh * L * L * 64 / (ms * 1e-3) / 1e12)
