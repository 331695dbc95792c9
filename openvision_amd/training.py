"""Training-side forward with gradients: the block stacks and the InfoNCE loss are autograd nodes whose
forward AND backward are the HIP kernels (``ov_tower_forward_saving`` / ``ov_tower_backward``, ``ov_clip_loss`` /
``ov_clip_loss_backward``).  The light ends around the towers — patch projection, class / positional embeddings, pooling,
``ln_post`` / ``ln_final``, the output projections and the L2 normalisation (0.2 % of a step's FLOPs) — are ordinary torch
operations on the device, differentiated by torch autograd; they are plumbing between the two HIP nodes, exactly the modules the
reference differentiates the same way (open_clip/transformer.py:609-651, model.py:265-315).

Opt-in: the inference entry points of ``openvision_amd.model`` never build a graph; a training loop calls

    img_f, txt_f, scale = training.clip_forward(model, images, tokens)       # features carry grad
    loss = ClipLoss(...)(img_f, txt_f, scale)                               # openvision_amd.loss.ClipLoss
    loss.backward()                                                         # .grad on every parameter, as with the reference

Activation memory: the tower keeps, per layer and token, the block input, the packed qkv, the attention output and the mid-block
residual (6 D bf16: 19 GB for L/14 at B=256, sized for the 288 GB of an MI355X); the LayerNorm outputs and the c_fc
pre-activation are recomputed during the backward.  Every preset is covered: head dims 72 / 80 (So400m, H/14) take the streaming
attention backward with d zero-padded to 96, an MLP width that is not a multiple of 64 (So400m) is zero-padded.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check, ptr, stream_ptr

_NAMES = ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc_w", "fc_b", "proj_w", "proj_b")


def _block_tensors(blk) -> List[torch.Tensor]:
    """The module's own parameters in ov_block_weights order (reference names: transformer.py:210-236)."""
    return [blk.ln_1.weight, blk.ln_1.bias, blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.attn.out_proj.weight,
            blk.attn.out_proj.bias, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias,
            blk.mlp.c_proj.weight, blk.mlp.c_proj.bias]


def _device_copy(p: torch.Tensor) -> torch.Tensor:
    """Kernel layout of one parameter: weight matrices bf16 [out, in], vectors fp32."""
    return (p.detach().to(torch.bfloat16) if p.dim() == 2 else p.detach().float()).contiguous()


def _pad_mlp(ts: List[torch.Tensor], mlp: int, mlp_pad: int) -> List[torch.Tensor]:
    """Zero-pad c_fc rows / bias and c_proj columns to the kernels' hidden pitch (a multiple of 64; So400m: 4304 -> 4352)."""
    if mlp_pad == mlp:
        return ts
    pad = mlp_pad - mlp
    ts = list(ts)
    ts[8] = torch.cat([ts[8], ts[8].new_zeros(pad, ts[8].shape[1])]).contiguous()
    ts[9] = torch.cat([ts[9], ts[9].new_zeros(pad)]).contiguous()
    ts[10] = torch.cat([ts[10], ts[10].new_zeros(ts[10].shape[0], pad)], dim=1).contiguous()
    return ts


class _TowerFn(torch.autograd.Function):
    """Transformer.forward (transformer.py:355-366) as one autograd node over all its blocks."""

    @staticmethod
    def forward(ctx, transformer, x, *params):
        lib = _lib.load()
        blocks = list(transformer.resblocks)
        b0 = blocks[0]
        d, heads, mlp, mlp_pad = b0.attn.embed_dim, b0.attn.num_heads, b0.mlp_dim, b0.mlp_pad
        if d % 64 or d % heads or (d // heads) % 8 or d // heads > 96:
            raise _lib.OvhipError("training path: width % 64 == 0 and head_dim % 8 == 0, <= 96 are required")
        bsz, seq, _ = x.shape
        cfg = _lib.TowerCfg(d, len(blocks), heads, mlp, mlp_pad, int(b0.gelu_tanh), float(b0.ln_1.eps))
        handle = lib.ov_tower_create(C.byref(cfg))
        if not handle:
            raise _lib.OvhipError("ov_tower_create failed")
        try:
            keep = []
            for i in range(len(blocks)):
                ts = _pad_mlp([_device_copy(p) for p in params[12 * i:12 * i + 12]], mlp, mlp_pad)
                keep.append(ts)
                bw = _lib.BlockWeights(*[C.c_void_p(t.data_ptr()) for t in ts], None, None)
                check(lib.ov_tower_set_block(handle, i, C.byref(bw)), "ov_tower_set_block")
            xb = x.detach().to(torch.bfloat16).contiguous().clone()
            saved = torch.empty(lib.ov_tower_saved_bytes(handle, bsz, seq) // 2, dtype=torch.bfloat16, device=x.device)
            nbytes = lib.ov_tower_workspace_bytes(handle, bsz, seq)
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=x.device)
            check(lib.ov_tower_forward_saving(handle, ptr(xb), ptr(saved), bsz, seq, ptr(ws), nbytes, stream_ptr()),
                  "ov_tower_forward_saving")
        finally:
            lib.ov_tower_destroy(handle)
        ctx.cfg, ctx.keep, ctx.saved, ctx.shape, ctx.mlp = cfg, keep, saved, (bsz, seq, d), mlp
        ctx.x_dtype, ctx.p_dtypes = x.dtype, [p.dtype for p in params]
        return xb.to(x.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        bsz, seq, d = ctx.shape
        layers = len(ctx.keep)
        handle = lib.ov_tower_create(C.byref(ctx.cfg))
        try:
            for i, ts in enumerate(ctx.keep):
                bw = _lib.BlockWeights(*[C.c_void_p(t.data_ptr()) for t in ts], None, None)
                check(lib.ov_tower_set_block(handle, i, C.byref(bw)), "ov_tower_set_block")
            grads = [[torch.empty_like(t) for t in ts] for ts in ctx.keep]
            garr = (_lib.BlockGrads * layers)(*[_lib.BlockGrads(*[C.c_void_p(t.data_ptr()) for t in g]) for g in grads])
            dx = grad_out.detach().to(torch.bfloat16).contiguous().clone()
            nbytes = lib.ov_tower_backward_workspace_bytes(handle, bsz, seq)
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dx.device)
            check(lib.ov_tower_backward(handle, ptr(ctx.saved), ptr(dx), garr, bsz, seq, ptr(ws), nbytes, stream_ptr()),
                  "ov_tower_backward")
        finally:
            lib.ov_tower_destroy(handle)
        mlp = ctx.mlp                                             # drop the (exactly zero) gradients of the MLP padding
        for gs in grads:
            gs[8], gs[9], gs[10] = gs[8][:mlp], gs[9][:mlp], gs[10][:, :mlp]
        flat = [g.to(ctx.p_dtypes[12 * i + j]) for i, gs in enumerate(grads) for j, g in enumerate(gs)]
        return (None, dx.view(bsz, seq, d).to(ctx.x_dtype), *flat)


def tower_forward(transformer, x: torch.Tensor) -> torch.Tensor:
    """``transformer(x)`` with gradients: x [B, L, D] on the device -> same shape; d x and every block parameter receive grad."""
    if not x.is_cuda:
        raise _lib.OvhipError("training path: tensors must live on an MI355X device (no CPU fallback)")
    params = [p for blk in transformer.resblocks for p in _block_tensors(blk)]
    return _TowerFn.apply(transformer, x, *params)


def encode_image(model, image: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """CLIP.encode_image (model.py:265-267) with gradients.  VisionTransformer.forward, transformer.py:609-651, for the
    OpenVision configuration (no ln_pre, pool -> ln_post -> proj)."""
    v = model.visual
    if not isinstance(v.ln_pre, torch.nn.Identity):
        raise _lib.OvhipError("training path: ln_pre is Identity for OpenVision towers")
    p = v.patch_size[0]
    w = v.conv1.weight
    # conv1 (:610-612, stride = kernel, no bias) as patch rows times W^T: the same sums as F.conv2d, and a plain GEMM for autograd
    # (MIOpen's fp32 convolution path costs tens of ms per step at this shape)
    bsz, _, hh, ww = image.shape
    gh, gw = hh // p, ww // p
    patches = image.float().reshape(bsz, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(bsz, gh * gw, 3 * p * p)
    x = patches @ w.float().reshape(w.shape[0], -1).t()
    cls = v.class_embedding.float().expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + v.positional_embedding.float()                            # :615-617
    x = tower_forward(v.transformer, x)
    if v.final_ln_after_pool:
        pooled = x[:, 1:].mean(dim=1) if v.pool_type == "avg" else x[:, 0]
        pooled = F.layer_norm(pooled, (pooled.shape[-1],), v.ln_post.weight.float(), v.ln_post.bias.float(), v.ln_post.eps)
    else:
        xx = F.layer_norm(x, (x.shape[-1],), v.ln_post.weight.float(), v.ln_post.bias.float(), v.ln_post.eps)
        pooled = xx[:, 1:].mean(dim=1) if v.pool_type == "avg" else xx[:, 0]
    out = pooled @ v.proj.float()
    return F.normalize(out, dim=-1) if normalize else out


def encode_text(model, text: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """CLIP.encode_text (model.py:269-284) with gradients: no mask, ln_final on all tokens, pool per text_pool_type."""
    x = F.embedding(text, model.token_embedding.weight.float()) + model.positional_embedding.float()
    x = tower_forward(model.transformer, x)
    x = F.layer_norm(x, (x.shape[-1],), model.ln_final.weight.float(), model.ln_final.bias.float(), model.ln_final.eps)
    pool = getattr(model, "text_pool_type", "last")
    if pool == "last":
        pooled = x[:, -1]
    elif pool == "first":
        pooled = x[:, 0]
    elif pool == "argmax":
        pooled = x[torch.arange(x.shape[0], device=x.device), text.argmax(dim=-1)]
    else:
        raise _lib.OvhipError(f"training path: text pool type {pool!r} is not supported")
    out = pooled @ model.text_projection.float()
    return F.normalize(out, dim=-1) if normalize else out


def clip_forward(model, image: torch.Tensor, text: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CLIP.forward (model.py:295-315) with gradients: (image_features, text_features, logit_scale.exp())."""
    return encode_image(model, image, True), encode_text(model, text, True), model.logit_scale.exp()
