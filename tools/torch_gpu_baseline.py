"""Informational: the reference's op sequence executed by PyTorch-ROCm's own kernels (hipBLASLt / SDPA / aten) on the same
MI355X, bf16 weights+activations — i.e. what a user of the reference gets by calling model.to('cuda').bfloat16().
Uses the oracle restatement (tests-only module) as the op sequence; NOT part of the product or of bench.py."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth
from oracle import clip_ref as R
import torch.nn.functional as F

def mha_sdpa(x, in_w, in_b, out_w, out_b, heads):
    B, L, D = x.shape
    hd = D // heads
    q, k, v = F.linear(x, in_w, in_b).split(D, dim=-1)
    sp = lambda t: t.reshape(B, L, heads, hd).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
    return F.linear(o.transpose(1, 2).reshape(B, L, D), out_w, out_b)
R.mha = mha_sdpa          # what nn.MultiheadAttention dispatches to on GPU (fused SDPA)

cfg = preset("vit-large-patch14-224")
dev = "cuda:0"
sd = {k: (v.to(dev).to(torch.bfloat16) if v.dim() > 0 else v.to(dev)) for k, v in synth.make_state_dict(cfg).items()}
for k in list(sd):
    if "ln_" in k or k.endswith("logit_scale"):
        sd[k] = sd[k].float()
B = int(os.environ.get("B", "256"))
img = synth.make_images(B, 224, seed=1000).to(dev).to(torch.bfloat16)
tok = synth.make_captions(B, seed=1000).to(dev)
def step():
    with torch.no_grad():
        ni, nt, s = R.clip_forward(img, tok, sd, cfg)
        li = s * ni.float() @ nt.float().T
        lab = torch.arange(B, device=dev)
        return (F.cross_entropy(li, lab) + F.cross_entropy(li.T, lab)) / 2
for _ in range(3): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"torch-ROCm eager bf16 (hipBLASLt + SDPA), B={B}: {B*n/dt:.1f} img/s, {dt/n*1e3:.2f} ms/step, loss {float(l):.4f}")
