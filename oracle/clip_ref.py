"""ORACLE — CPU restatement of the reference's encode-and-contrast path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / reported CPU baseline; nothing under ``openvision_amd/`` imports it
and the product path has no CPU fallback.

What it restates (plain ``torch`` CPU functional ops in fp32 — the same aten kernels the reference's
``nn.Module`` tree dispatches to), each function citing the reference lines it follows
(paths relative to ``/root/reference/src/convert_upload/open_clip/``):

  layer_norm            transformer.py:15-30     (eps 1e-6: transformer.py:458,690)
  mha                   transformer.py:225,239-252 -> torch nn.MultiheadAttention arithmetic
  resblock              transformer.py:254-265
  vision_forward        transformer.py:609-651   (final_ln_after_pool branch :638-640, pool :599-601)
  encode_text           model.py:269-284, transformer.py:654-666 ('last' / 'first')
  encode_image/forward  model.py:265-267, 295-315
  clip_loss             loss.py:89-131 (+ gather_features loss.py:19-63 as `all_*` arguments)

PARITY PINNING: pinned.  The reference is Python and importable in the build container, so
``tests/golden/make_golden.py`` runs the reference's own ``open_clip.model.CLIP`` / ``open_clip.loss.ClipLoss``
on formula weights (``openvision_amd/synth.py``) and commits inputs + outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against those vectors (fp32, atol 2e-5).
The reference's own test-suite holds no fixture for this path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

LN_EPS = 1e-6


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = LN_EPS) -> torch.Tensor:
    """transformer.py:24-30 (LayerNorm) / :15-21 (LayerNormFp32: stats in fp32, cast back)."""
    orig = x.dtype
    y = F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)
    return y.to(orig)


def gelu(x: torch.Tensor, tanh: bool) -> torch.Tensor:
    """nn.GELU() for the vision tower (erf; src/models/vit.py:202), nn.GELU(approximate='tanh')
    for the text tower (model.py:196-197, transfer_jax2hf.py:595)."""
    return F.gelu(x, approximate="tanh" if tanh else "none")


def layer_norm_backward(x, w, dy, eps: float = LN_EPS):
    """What autograd returns for transformer.py:24-30 (F.layer_norm, biased variance), in closed form, fp32:
    dx = rstd (q - mean(q) - xhat mean(q xhat)) with q = dy * w;  dw = sum_rows dy * xhat;  db = sum_rows dy."""
    x, w, dy = x.float(), w.float(), dy.float()
    mean = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((x - mean) ** 2).mean(-1, keepdim=True) + eps)
    xh = (x - mean) * rstd
    q = dy * w
    dx = rstd * (q - q.mean(-1, keepdim=True) - xh * (q * xh).mean(-1, keepdim=True))
    flat = lambda t: t.reshape(-1, t.shape[-1])
    return dx, (flat(dy) * flat(xh)).sum(0), flat(dy).sum(0)


def linear_backward(dy, x, w):
    """nn.Linear y = x w^T + b (transformer.py:232-236 c_fc / c_proj; nn.MultiheadAttention's in/out projections):
    dx = dy w, dw = dy^T x, db = sum_rows dy."""
    dy, x, w = dy.float(), x.float(), w.float()
    dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
    return dy @ w, dy2.T @ x2, dy2.sum(0)


def gelu_backward(a, dh, tanh: bool):
    """d gelu(a) / d a * dh.  erf form: Phi(a) + a phi(a); tanh form: 0.5 (1 + t) + 0.5 a (1 - t^2) u', t = tanh(u),
    u = sqrt(2/pi) (a + 0.044715 a^3)."""
    a, dh = a.double(), dh.double()
    if tanh:
        c = math.sqrt(2.0 / math.pi)
        t = torch.tanh(c * (a + 0.044715 * a ** 3))
        g = 0.5 * (1 + t) + 0.5 * a * (1 - t * t) * c * (1 + 3 * 0.044715 * a * a)
    else:
        g = 0.5 * (1 + torch.erf(a / math.sqrt(2.0))) + a * torch.exp(-0.5 * a * a) / math.sqrt(2.0 * math.pi)
    return (dh * g).float()


def attention_backward(q, k, v, do, scale: float):
    """Backward of softmax(scale q k^T) v (the core of nn.MultiheadAttention, transformer.py:225,239-252) in closed form, fp32:
    dv = p^T do, dp = do v^T, ds = p * (dp - rowsum(p * dp)), dq = scale ds k, dk = scale ds^T q.  Shapes [..., L, d]."""
    q, k, v, do = q.float(), k.float(), v.float(), do.float()
    p = torch.softmax(scale * q @ k.transpose(-1, -2), dim=-1)
    dv = p.transpose(-1, -2) @ do
    dp = do @ v.transpose(-1, -2)
    ds = p * (dp - (p * dp).sum(-1, keepdim=True))
    return scale * ds @ k, scale * ds.transpose(-1, -2) @ q, dv


def mha(x: torch.Tensor, in_w, in_b, out_w, out_b, heads: int) -> torch.Tensor:
    """Self-attention of nn.MultiheadAttention(batch_first=True), no mask, no dropout
    (transformer.py:225,239-252): packed qkv projection (order q,k,v), q scaled by hd^-0.5,
    softmax over keys, heads merged, out_proj."""
    B, L, D = x.shape
    hd = D // heads
    qkv = F.linear(x, in_w, in_b)                       # [B,L,3D]
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(B, L, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.reshape(B, L, heads, hd).transpose(1, 2)
    v = v.reshape(B, L, heads, hd).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, L, D)
    return F.linear(o, out_w, out_b)


def resblock(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str, heads: int, tanh_gelu: bool,
             eps: float = LN_EPS) -> torch.Tensor:
    """ResidualAttentionBlock.forward, transformer.py:254-265 (ls_1/ls_2 = Identity)."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
    x = x + mha(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
    h = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
    h = gelu(h, tanh_gelu)
    h = F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    return x + h


def resblock_backward(x, dy, sd, p: str, heads: int, tanh_gelu: bool, eps: float = LN_EPS):
    """Backward of ``resblock`` (what autograd computes through transformer.py:254-265), composed from the closed-form operator
    backwards above, fp32.  Returns (dx, {reference parameter name (without prefix): gradient})."""
    f = lambda k: sd[p + k].float()
    x, dy = x.float(), dy.float()
    B, L, D = x.shape
    hd = D // heads
    split = lambda t: t.reshape(B, L, heads, hd).transpose(1, 2)
    merge = lambda t: t.transpose(1, 2).reshape(B, L, D)
    # forward intermediates
    n1 = layer_norm(x, f("ln_1.weight"), f("ln_1.bias"), eps)
    qkv = F.linear(n1, f("attn.in_proj_weight"), f("attn.in_proj_bias"))
    q, k, v = [split(t) for t in qkv.split(D, dim=-1)]
    o = merge(torch.softmax((q * hd ** -0.5) @ k.transpose(-1, -2), dim=-1) @ v)
    x1 = x + F.linear(o, f("attn.out_proj.weight"), f("attn.out_proj.bias"))
    n2 = layer_norm(x1, f("ln_2.weight"), f("ln_2.bias"), eps)
    a = F.linear(n2, f("mlp.c_fc.weight"), f("mlp.c_fc.bias"))
    h = gelu(a, tanh_gelu)
    g = {}
    # MLP branch
    dh, g["mlp.c_proj.weight"], g["mlp.c_proj.bias"] = linear_backward(dy, h, f("mlp.c_proj.weight"))
    da = gelu_backward(a, dh, tanh_gelu)
    dn2, g["mlp.c_fc.weight"], g["mlp.c_fc.bias"] = linear_backward(da, n2, f("mlp.c_fc.weight"))
    dl, g["ln_2.weight"], g["ln_2.bias"] = layer_norm_backward(x1, f("ln_2.weight"), dn2, eps)
    dx1 = dy + dl
    # attention branch
    do, g["attn.out_proj.weight"], g["attn.out_proj.bias"] = linear_backward(dx1, o, f("attn.out_proj.weight"))
    dq, dk, dv = attention_backward(q, k, v, split(do), hd ** -0.5)
    dqkv = torch.cat([merge(dq), merge(dk), merge(dv)], dim=-1)
    dn1, g["attn.in_proj_weight"], g["attn.in_proj_bias"] = linear_backward(dqkv, n1, f("attn.in_proj_weight"))
    dl, g["ln_1.weight"], g["ln_1.bias"] = layer_norm_backward(x, f("ln_1.weight"), dn1, eps)
    return dx1 + dl, g


def block_stack(x, sd, prefix: str, layers: int, heads: int, tanh_gelu: bool, eps: float = LN_EPS):
    """Transformer.forward, transformer.py:355-366."""
    for i in range(layers):
        x = resblock(x, sd, f"{prefix}resblocks.{i}.", heads, tanh_gelu, eps)
    return x


def patch_embed(image: torch.Tensor, sd, patch: int) -> torch.Tensor:
    """conv1 (stride = kernel = P, no bias) + reshape/permute + cls concat + pos-emb add,
    transformer.py:610-617.  Returns [B, g*g+1, D]."""
    x = F.conv2d(image, sd["visual.conv1.weight"], None, stride=patch)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd["visual.class_embedding"].view(1, 1, -1).expand(x.shape[0], -1, -1).to(x.dtype)
    x = torch.cat([cls, x], dim=1)
    return x + sd["visual.positional_embedding"].to(x.dtype)


def vision_forward(image, sd, vcfg: dict, tanh_gelu: bool = False, eps: float = LN_EPS,
                   return_tokens: bool = False):
    """VisionTransformer.forward for OpenVision configs (ln_pre = Identity, pool -> ln_post -> proj),
    transformer.py:609-651."""
    heads = vcfg["width"] // vcfg["head_width"]
    x = patch_embed(image, sd, vcfg["patch_size"])
    x = block_stack(x, sd, "visual.transformer.", vcfg["layers"], heads, tanh_gelu, eps)
    pool = vcfg.get("pool_type", "avg")
    lnw, lnb = sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]
    if vcfg.get("final_ln_after_pool", True):
        pooled = x[:, 1:].mean(dim=1) if pool == "avg" else x[:, 0]       # :599-603
        pooled = layer_norm(pooled, lnw, lnb, eps)                        # :638-640
    else:
        xx = layer_norm(x, lnw, lnb, eps)
        pooled = xx[:, 1:].mean(dim=1) if pool == "avg" else xx[:, 0]
    out = pooled @ sd["visual.proj"]                                       # :645-646
    return (out, x) if return_tokens else out


def text_forward(tokens: torch.Tensor, sd, tcfg: dict, eps: float = LN_EPS, return_tokens: bool = False):
    """CLIP.encode_text without the final normalize, model.py:269-282."""
    tanh_gelu = bool(tcfg.get("act_kwargs")) and tcfg["act_kwargs"].get("approximate") == "tanh"
    x = F.embedding(tokens, sd["token_embedding.weight"])
    x = x + sd["positional_embedding"]
    x = block_stack(x, sd, "transformer.", tcfg["layers"], tcfg["heads"], tanh_gelu, eps)
    xx = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"], eps)
    pool = tcfg.get("pool_type", "last")
    pooled = xx[:, -1] if pool == "last" else xx[:, 0]                     # transformer.py:655-658
    out = pooled @ sd["text_projection"]
    return (out, x) if return_tokens else out


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """F.normalize(x, dim=-1): x / max(||x||_2, 1e-12), model.py:267,284."""
    return F.normalize(x, dim=-1)


def encode_image(image, sd, model_cfg: dict, normalize: bool = False):
    v = model_cfg["vision_cfg"]
    tanh = bool(v.get("act_kwargs")) and v["act_kwargs"].get("approximate") == "tanh"
    f = vision_forward(image, sd, v, tanh)
    return l2_normalize(f) if normalize else f


def encode_text(tokens, sd, model_cfg: dict, normalize: bool = False):
    f = text_forward(tokens, sd, model_cfg["text_cfg"])
    return l2_normalize(f) if normalize else f


def clip_forward(image, tokens, sd, model_cfg: dict):
    """CLIP.forward, model.py:295-315: normalised features + exp(logit_scale)."""
    return (encode_image(image, sd, model_cfg, True), encode_text(tokens, sd, model_cfg, True),
            sd["logit_scale"].exp())


def clip_loss(img: torch.Tensor, txt: torch.Tensor, logit_scale, all_img: Optional[torch.Tensor] = None,
              all_txt: Optional[torch.Tensor] = None, rank: int = 0) -> torch.Tensor:
    """ClipLoss.forward (loss.py:120-131) with local_loss=True semantics when ``all_*`` are given:
    logits [b,N] both ways (loss.py:108-110), labels arange(b) + b*rank (:93-94), mean CE / 2.
    With ``all_* = None`` it is the single-process loss (loss.py:114-116)."""
    if all_img is None:
        all_img, all_txt = img, txt
    b = img.shape[0]
    li = logit_scale * img.float() @ all_txt.float().T
    lt = logit_scale * txt.float() @ all_img.float().T
    labels = torch.arange(b, dtype=torch.long) + b * rank
    return (F.cross_entropy(li, labels) + F.cross_entropy(lt, labels)) / 2


def clip_loss_grads(img, txt, logit_scale, all_img=None, all_txt=None, rank: int = 0):
    """Closed-form gradient of ``clip_loss`` (what autograd gives the reference through loss.py:102-131), with every argument
    treated as an independent leaf:  P = (softmax(logits) - onehot) / (2 b) per direction, then
        d img = s P_i all_txt,  d txt = s P_t all_img,  d all_txt = s P_i^T img,  d all_img = s P_t^T txt,
        d s   = sum P_i * (img all_txt^T) + sum P_t * (txt all_img^T).
    Returns (d_img, d_txt, d_all_img, d_all_txt, d_scale); the caller adds / routes the gathered-side terms the way
    gather_features (loss.py:19-63) makes them flow."""
    if all_img is None:
        all_img, all_txt = img, txt
    img, txt, all_img, all_txt = img.float(), txt.float(), all_img.float(), all_txt.float()
    s = float(logit_scale)
    b = img.shape[0]
    di, dt = img @ all_txt.T, txt @ all_img.T
    idx = torch.arange(b) + b * rank
    pi, pt = torch.softmax(s * di, 1), torch.softmax(s * dt, 1)
    pi[torch.arange(b), idx] -= 1.0
    pt[torch.arange(b), idx] -= 1.0
    pi, pt = pi / (2 * b), pt / (2 * b)
    return (s * pi @ all_txt, s * pt @ all_img, s * pt.T @ txt, s * pi.T @ img, (pi * di).sum() + (pt * dt).sum())


def clip_loss_backward_ranks(img_all, txt_all, logit_scale, ws: int, local_loss: bool, gather_with_grad: bool):
    """Per-rank (d image_features, d text_features, d logit_scale) of ClipLoss at world_size ``ws`` from the full feature
    sets, routing the gathered-side gradient as gather_features does (loss.py:39-61): detached gather -> nothing flows back
    except through the own chunk put back when ``not local_loss`` (:57-59); ``gather_with_grad`` -> the backward of
    torch.distributed.nn.all_gather sums every rank's gathered-side gradient and hands each rank its own rows."""
    n, e = img_all.shape
    b = n // ws
    per = []
    for r in range(ws):
        sl = slice(r * b, (r + 1) * b)
        if local_loss:
            per.append(clip_loss_grads(img_all[sl], txt_all[sl], logit_scale, img_all, txt_all, r))
        else:
            per.append(clip_loss_grads(img_all, txt_all, logit_scale, img_all, txt_all, 0))
    out = []
    for r in range(ws):
        sl = slice(r * b, (r + 1) * b)
        d_img, d_txt, d_ai, d_at, d_s = per[r]
        if local_loss:
            gi, gt = d_img.clone(), d_txt.clone()
            if gather_with_grad:
                gi += sum(p[2] for p in per)[sl]
                gt += sum(p[3] for p in per)[sl]
        else:
            if gather_with_grad:
                gi = sum(p[0] + p[2] for p in per)[sl]
                gt = sum(p[1] + p[3] for p in per)[sl]
            else:
                gi, gt = (d_img + d_ai)[sl], (d_txt + d_at)[sl]
        out.append((gi, gt, d_s))
    return out


def clip_loss_terms(img, txt, logit_scale, all_img=None, all_txt=None, rank: int = 0):
    """Per-row LSE and diagonal logits of both directions (what the fused HIP kernel emits)."""
    if all_img is None:
        all_img, all_txt = img, txt
    b = img.shape[0]
    li = logit_scale * img.float() @ all_txt.float().T
    lt = logit_scale * txt.float() @ all_img.float().T
    idx = torch.arange(b) + b * rank
    return (torch.logsumexp(li, 1), li[torch.arange(b), idx],
            torch.logsumexp(lt, 1), lt[torch.arange(b), idx])


def zero_shot_table(img_feat: torch.Tensor, txt_feat: torch.Tensor, logit_scale: torch.Tensor):
    """ov-zero-shot-test.py:176-183,192: cosine = img @ txt.T on normalised features,
    probs = softmax(exp(logit_scale) * cosine), argsort descending."""
    i = img_feat / img_feat.norm(dim=-1, keepdim=True)
    t = txt_feat / txt_feat.norm(dim=-1, keepdim=True)
    cos = i @ t.T
    probs = (logit_scale.exp() * cos).softmax(dim=-1)
    return cos, probs, cos.argsort(dim=-1, descending=True)
