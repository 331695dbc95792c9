"""Training-side forward with gradients: the block stacks and the InfoNCE loss are autograd nodes whose
forward AND backward are the HIP kernels (``ov_tower_forward_saving`` / ``ov_tower_backward``, ``ov_clip_loss`` /
``ov_clip_loss_backward``).  The light ends around the towers run on the HIP operators too: the patch projection and the output
projections are autograd nodes over ``ov_gemm`` / ``ov_linear_backward``, ``ln_post`` / ``ln_final`` over ``ov_layernorm`` /
``ov_layernorm_backward`` (open_clip/transformer.py:609-651, model.py:265-315); what is left to torch autograd is data movement and
element-wise plumbing (class / positional embedding adds, token gather, mean pooling, L2 normalisation) -- no aten GEMM.

Opt-in: the inference entry points of ``openvision_amd.model`` never build a graph; a training loop calls

    img_f, txt_f, scale = training.clip_forward(model, images, tokens)       # features carry grad
    loss = ClipLoss(...)(img_f, txt_f, scale)                               # openvision_amd.loss.ClipLoss
    loss.backward()                                                         # .grad on every parameter, as with the reference

Activation memory: the tower keeps, per layer and token, the block input, the packed qkv, the attention output, the mid-block
residual, both LayerNorm outputs and the c_fc pre-activation and activation (8 D + 2 mlp bf16: 52 GB for L/14 at B=256, sized for
the 288 GB of an MI355X); the backward runs nothing of the forward again.  Every preset is covered: head dims 72 / 80 (So400m, H/14) take the streaming
attention backward with d zero-padded to 96, an MLP width that is not a multiple of 64 (So400m) is zero-padded.
"""
from __future__ import annotations

import os
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


class _Pool:
    """Grow-only pool of device buffers, one free list per purpose ("saved": the activations a forward keeps for its backward, 19 GB at
    L/14, B = 256; "ws": kernel workspaces): a step takes what it needs in forward and hands it back at the end of backward, so
    steady-state training allocates nothing per step.  Retention sizes itself: per purpose the pool keeps as many free buffers as
    were ever outstanding at once (one saved buffer per chunk of `set_backward_chunk_layers`, one workspace), and a request is served
    by the SMALLEST free buffer of its own purpose that fits, so a workspace never takes a saved-activation buffer."""

    def __init__(self):
        self.lists = {}                    # purpose -> free buffers, ascending size
        self.out = {}                      # purpose -> buffers currently handed out
        self.peak = {}                     # purpose -> high-water mark of `out`

    @property
    def free(self) -> List[torch.Tensor]:
        return [t for lst in self.lists.values() for t in lst]

    def take(self, nbytes: int, device, kind: str = "ws") -> torch.Tensor:
        lst = self.lists.setdefault(kind, [])
        self.out[kind] = self.out.get(kind, 0) + 1
        self.peak[kind] = max(self.peak.get(kind, 0), self.out[kind])
        for i, t in enumerate(lst):
            if t.device == device and t.numel() >= nbytes:
                t = lst.pop(i)
                t._ovhip_gen += 1              # whoever still holds it from an earlier step can tell it was recycled
                return t
        t = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        t._ovhip_gen, t._ovhip_kind = 0, kind
        return t

    def give(self, t: torch.Tensor) -> None:
        kind = getattr(t, "_ovhip_kind", "ws")
        lst = self.lists.setdefault(kind, [])
        if any(u is t for u in lst):
            return
        self.out[kind] = max(0, self.out.get(kind, 0) - 1)
        lst.append(t)
        lst.sort(key=lambda x: x.numel())
        del lst[:-max(1, self.peak.get(kind, 1))]        # too many: the smallest go (a grown request made them useless)


def _train_state(transformer):
    st = getattr(transformer, "_ovhip_train_state", None)
    if st is None:
        st = {"chunks": {}, "pool": _Pool()}            # chunks: (first layer, last layer + 1) -> {"sig", "keep"}
        object.__setattr__(transformer, "_ovhip_train_state", st)
    return st


def _packed_blocks(transformer, params, mlp: int, mlp_pad: int, span=(0, -1)):
    """Kernel-layout copies of every block's parameters, rebuilt only when a parameter changed: keyed, as the inference path's
    cache is, by (data_ptr, _version) of the sources (an optimiser step bumps _version; frozen weights -- gradient ascent on the
    inputs, ov-gradient-ascent.py -- hit the cache every step).  `model.invalidate_packed()` drops it (writes through .data)."""
    from .model import _PACK_EPOCH
    st = _train_state(transformer)["chunks"].setdefault(tuple(span), {"sig": None, "keep": None})
    sig = (_PACK_EPOCH[0], mlp_pad) + tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)
    if st["sig"] != sig:
        st["keep"] = [_pad_mlp([_device_copy(p) for p in params[12 * i:12 * i + 12]], mlp, mlp_pad) for i in range(len(params) // 12)]
        st["sig"] = sig
    return st["keep"]


class _TowerFn(torch.autograd.Function):
    """Transformer.forward (transformer.py:355-366) as one autograd node over the blocks [lo, hi) (all of them by default; several
    consecutive nodes when ``tower_forward`` is asked for chunks, so that parameter gradients become available chunk by chunk)."""

    @staticmethod
    def forward(ctx, transformer, lo, hi, x, *params):
        lib = _lib.load()
        blocks = list(transformer.resblocks)[lo:hi]
        b0 = blocks[0]
        d, heads, mlp, mlp_pad = b0.attn.embed_dim, b0.attn.num_heads, b0.mlp_dim, b0.mlp_pad
        if d % 64 or d % heads or (d // heads) % 8 or d // heads > 96:
            raise _lib.OvhipError("training path: width % 64 == 0 and head_dim % 8 == 0, <= 96 are required")
        bsz, seq, _ = x.shape
        cfg = _lib.TowerCfg(d, len(blocks), heads, mlp, mlp_pad, int(b0.gelu_tanh), float(b0.ln_1.eps))
        handle = lib.ov_tower_create(C.byref(cfg))
        if not handle:
            raise _lib.OvhipError("ov_tower_create failed")
        pool = _train_state(transformer)["pool"]
        try:
            keep = _packed_blocks(transformer, params, mlp, mlp_pad, (lo, hi))
            for i, ts in enumerate(keep):
                bw = _lib.BlockWeights(*[C.c_void_p(t.data_ptr()) for t in ts], None, None)
                check(lib.ov_tower_set_block(handle, i, C.byref(bw)), "ov_tower_set_block")
            xb = x.detach().to(torch.bfloat16).contiguous().clone()
            saved = pool.take(lib.ov_tower_saved_bytes(handle, bsz, seq), x.device, "saved")
            nbytes = lib.ov_tower_workspace_bytes(handle, bsz, seq)
            ws = pool.take(nbytes, x.device)
            check(lib.ov_tower_forward_saving(handle, ptr(xb), ptr(saved), bsz, seq, ptr(ws), nbytes, stream_ptr()),
                  "ov_tower_forward_saving")
            pool.give(ws)
        finally:
            lib.ov_tower_destroy(handle)
        ctx.cfg, ctx.keep, ctx.saved, ctx.shape, ctx.mlp, ctx.pool = cfg, keep, saved, (bsz, seq, d), mlp, pool
        ctx.saved_gen = saved._ovhip_gen
        ctx.x_dtype, ctx.p_dtypes = x.dtype, [p.dtype for p in params]
        return xb.to(x.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        bsz, seq, d = ctx.shape
        layers = len(ctx.keep)
        if ctx.saved._ovhip_gen != ctx.saved_gen:
            raise _lib.OvhipError("training path: the saved activations of this graph were recycled by a later forward; a second "
                                  "backward over the same graph must come before the next forward of this tower")
        handle = lib.ov_tower_create(C.byref(ctx.cfg))
        try:
            for i, ts in enumerate(ctx.keep):
                bw = _lib.BlockWeights(*[C.c_void_p(t.data_ptr()) for t in ts], None, None)
                check(lib.ov_tower_set_block(handle, i, C.byref(bw)), "ov_tower_set_block")
            grads = [[torch.empty_like(t) for t in ts] for ts in ctx.keep]
            garr = (_lib.BlockGrads * layers)(*[_lib.BlockGrads(*[C.c_void_p(t.data_ptr()) for t in g]) for g in grads])
            dx = grad_out.detach().to(torch.bfloat16).contiguous().clone()
            nbytes = lib.ov_tower_backward_workspace_bytes(handle, bsz, seq)
            ws = ctx.pool.take(nbytes, dx.device)
            check(lib.ov_tower_backward(handle, ptr(ctx.saved), ptr(dx), garr, bsz, seq, ptr(ws), nbytes, stream_ptr()),
                  "ov_tower_backward")
            ctx.pool.give(ws)
            ctx.pool.give(ctx.saved)         # ordered on the stream: the next forward's writes come after this backward's reads
        finally:
            lib.ov_tower_destroy(handle)
        mlp = ctx.mlp                                             # drop the (exactly zero) gradients of the MLP padding
        for gs in grads:
            gs[8], gs[9], gs[10] = gs[8][:mlp], gs[9][:mlp], gs[10][:, :mlp]
        flat = [g.to(ctx.p_dtypes[12 * i + j]) for i, gs in enumerate(grads) for j, g in enumerate(gs)]
        return (None, None, None, dx.view(bsz, seq, d).to(ctx.x_dtype), *flat)


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class _LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) on ov_gemm, backward on ov_linear_backward: the light-end projections of the towers (patch projection
    transformer.py:610-612 as a GEMM over patch rows, `@ proj` :645-646, `@ text_projection` model.py:282) -- no aten / hipBLASLt GEMM
    on the training path.  x [M, K] any float dtype, w [N, K]; K and N are zero-padded to the kernels' 64 granule."""

    @staticmethod
    def forward(ctx, x, w, bias):
        lib = _lib.load()
        m, k = x.shape
        n = w.shape[0]
        kp, npad = _round_up(k, 64), _round_up(n, 64)
        xb = torch.zeros(m, kp, dtype=torch.bfloat16, device=x.device)
        xb[:, :k] = x.detach()
        wb = torch.zeros(npad, kp, dtype=torch.bfloat16, device=x.device)
        wb[:n, :k] = w.detach()
        bb = None
        if bias is not None:
            bb = torch.zeros(npad, dtype=torch.float32, device=x.device)
            bb[:n] = bias.detach().float()
        out = torch.empty(m, npad, dtype=torch.bfloat16, device=x.device)
        check(lib.ov_gemm(ptr(xb), kp, ptr(wb), kp, ptr(bb), ptr(out), npad, m, npad, kp, _lib.EPI_BIAS, None, 0, 0, 0, 0, stream_ptr()),
              "ov_gemm")
        ctx.save_for_backward(xb, wb)
        ctx.dims = (m, n, k, kp, npad, bias is not None, x.dtype, w.dtype, bias.dtype if bias is not None else None)
        return out[:, :n].float()

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xb, wb = ctx.saved_tensors
        m, n, k, kp, npad, has_b, xd, wd, bd = ctx.dims
        dyb = torch.zeros(m, npad, dtype=torch.bfloat16, device=dy.device)
        dyb[:, :n] = dy.detach()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty(m, kp, dtype=torch.bfloat16, device=dy.device) if need_x else None
        dw = torch.empty(npad, kp, dtype=torch.bfloat16, device=dy.device) if need_w else None
        db = torch.empty(npad, dtype=torch.float32, device=dy.device) if has_b else None
        nb = lib.ov_linear_backward_workspace_bytes(m, npad, kp)
        ws = torch.empty(nb + 256, dtype=torch.uint8, device=dy.device)
        check(lib.ov_linear_backward(ptr(dyb), npad, ptr(xb), kp, ptr(wb), kp, m, npad, kp, ptr(dx), kp, ptr(dw), kp, ptr(db), ptr(ws), nb,
                                     stream_ptr()), "ov_linear_backward")
        return (dx[:, :k].to(xd) if need_x else None, dw[:n, :k].to(wd) if need_w else None, db[:n].to(bd) if has_b else None)


class _LayerNormFn(torch.autograd.Function):
    """LayerNorm (transformer.py:15-30: fp32 statistics) on ov_layernorm / ov_layernorm_backward: ln_post on the pooled rows, ln_final
    on the text tokens."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        lib = _lib.load()
        d = x.shape[-1]
        xb = x.detach().to(torch.bfloat16).contiguous().view(-1, d)
        g, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y = torch.empty(xb.shape, dtype=torch.float32, device=x.device)
        check(lib.ov_layernorm(ptr(xb), _lib.OV_BF16, d, ptr(g), ptr(b), ptr(y), _lib.OV_F32, d, xb.shape[0], d, float(eps), stream_ptr()),
              "ov_layernorm")
        ctx.save_for_backward(xb, g)
        ctx.meta = (x.shape, float(eps), x.dtype, weight.dtype, bias.dtype)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xb, g = ctx.saved_tensors
        shape, eps, xd, wd, bd = ctx.meta
        rows, d = xb.shape
        dyb = dy.detach().to(torch.bfloat16).contiguous().view(rows, d)
        dx = torch.empty_like(xb)
        dg = torch.empty(d, dtype=torch.float32, device=dy.device)
        db = torch.empty(d, dtype=torch.float32, device=dy.device)
        nb = lib.ov_layernorm_backward_workspace_bytes(rows, d)
        ws = torch.empty(nb + 256, dtype=torch.uint8, device=dy.device)
        check(lib.ov_layernorm_backward(ptr(xb), d, ptr(g), ptr(dyb), d, None, 0, ptr(dx), d, ptr(dg), ptr(db), rows, d, eps, ptr(ws), nb,
                                        stream_ptr()), "ov_layernorm_backward")
        return dx.view(shape).to(xd), dg.to(wd), db.to(bd), None


CHUNK_LAYERS = [max(0, int(os.environ.get("OVHIP_TRAIN_CHUNK_LAYERS", "0") or 0))]   # > 0: towers run as consecutive autograd nodes of that many blocks


def set_backward_chunk_layers(n: int) -> None:
    """Run every tower as consecutive autograd nodes of ``n`` blocks each (0 = one node per tower, the default).  The arithmetic
    is unchanged; what changes is WHEN parameter gradients exist: after each chunk's backward instead of after the whole tower's,
    which is what lets ``FusedAdamW.overlap_gradient_exchange`` start a bucket's all-reduce while earlier blocks are still in
    their backward (the reference leaves this to DistributedDataParallel's bucket hooks)."""
    CHUNK_LAYERS[0] = max(0, int(n))


def tower_forward(transformer, x: torch.Tensor) -> torch.Tensor:
    """``transformer(x)`` with gradients: x [B, L, D] on the device -> same shape; d x and every block parameter receive grad."""
    if not x.is_cuda:
        raise _lib.OvhipError("training path: tensors must live on an MI355X device (no CPU fallback)")
    blocks = list(transformer.resblocks)
    step = CHUNK_LAYERS[0] if CHUNK_LAYERS[0] > 0 else len(blocks)
    for lo in range(0, len(blocks), step):
        hi = min(len(blocks), lo + step)
        params = [p for blk in blocks[lo:hi] for p in _block_tensors(blk)]
        x = _TowerFn.apply(transformer, lo, hi, x, *params)
    return x


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
    patches = image.reshape(bsz, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(bsz * gh * gw, 3 * p * p)
    x = _LinearFn.apply(patches, w.reshape(w.shape[0], -1), None).view(bsz, gh * gw, -1)
    cls = v.class_embedding.float().expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + v.positional_embedding.float()                            # :615-617
    x = tower_forward(v.transformer, x)
    if v.final_ln_after_pool:
        pooled = x[:, 1:].mean(dim=1) if v.pool_type == "avg" else x[:, 0]
        pooled = _LayerNormFn.apply(pooled, v.ln_post.weight, v.ln_post.bias, v.ln_post.eps)
    else:
        xx = _LayerNormFn.apply(x, v.ln_post.weight, v.ln_post.bias, v.ln_post.eps)
        pooled = xx[:, 1:].mean(dim=1) if v.pool_type == "avg" else xx[:, 0]
    out = _LinearFn.apply(pooled, v.proj.t(), None)
    return F.normalize(out, dim=-1) if normalize else out


def encode_text_embeddings(model, x: torch.Tensor, normalize: bool = True, pool_index=None) -> torch.Tensor:
    """The text tower from token EMBEDDINGS x [B, T, width] (before the positional embedding), with gradients to x: the entry of
    ov-gradient-ascent.py:102-127, which feeds `soft_one_hot @ token_embedding.weight` instead of token ids.  `pool_index` (int64 [B])
    selects the pooled position for text_pool_type 'argmax' (ids are not available here)."""
    x = x.float() + model.positional_embedding.float()[: x.shape[1]]
    x = tower_forward(model.transformer, x)
    x = _LayerNormFn.apply(x, model.ln_final.weight, model.ln_final.bias, model.ln_final.eps)
    pool = getattr(model, "text_pool_type", "last")
    if pool == "last":
        pooled = x[:, -1]
    elif pool == "first":
        pooled = x[:, 0]
    elif pool == "argmax":
        if pool_index is None:
            raise _lib.OvhipError("training path: text pool type 'argmax' needs pool_index when embeddings are given")
        pooled = x[torch.arange(x.shape[0], device=x.device), pool_index]
    else:
        raise _lib.OvhipError(f"training path: text pool type {pool!r} is not supported")
    out = _LinearFn.apply(pooled, model.text_projection.t(), None)
    return F.normalize(out, dim=-1) if normalize else out


def encode_text(model, text: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """CLIP.encode_text (model.py:269-284) with gradients: no mask, ln_final on all tokens, pool per text_pool_type.  `text`: int64
    token ids [B, T], or a float [B, T, vocab] matrix of (soft) one-hot rows (ov-gradient-ascent.py:105: `text @ token_embedding.weight`,
    gradients flow to the rows)."""
    if text.is_floating_point():
        if text.dim() != 3 or text.shape[-1] != model.token_embedding.weight.shape[0]:
            raise ValueError("soft tokens must be [B, T, vocab_size]")
        return encode_text_embeddings(model, text.float() @ model.token_embedding.weight.float(), normalize,
                                      text.argmax(dim=-1).argmax(dim=-1))
    x = F.embedding(text, model.token_embedding.weight.float())
    return encode_text_embeddings(model, x, normalize, text.argmax(dim=-1))


def clip_forward(model, image: torch.Tensor, text: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CLIP.forward (model.py:295-315) with gradients: (image_features, text_features, logit_scale.exp())."""
    return encode_image(model, image, True), encode_text(model, text, True), model.logit_scale.exp()


# --------------------------------------------------------------------------------------------------------------------------
# parameter update + data-parallel gradient exchange: what closes the training step (src/main_clip.py:480-483)
# --------------------------------------------------------------------------------------------------------------------------
def default_decay_filter(name: str, p: torch.Tensor) -> bool:
    """The reference decays '.*/kernel$' only (build_optax.py:259: Dense / Conv kernels): here the 2-D (and conv) weight matrices;
    biases, LayerNorm parameters, class / positional / token embeddings and the logit scale are not decayed."""
    if name.endswith("token_embedding.weight") or "positional_embedding" in name or name.endswith("class_embedding"):
        return False
    return p.dim() >= 2


class FusedAdamW:
    """The reference trainer's update (optax chain of src/optim/build_optax.py:272-278 with config.optax = scale_by_adam(b1=0.9,
    b2=0.95, mu_dtype=bfloat16), decoupled weight decay scaled by the learning rate, optional global-norm clipping) as ONE HIP launch
    per parameter group over flat buffers (``ov_adamw_step``), plus the data-parallel gradient exchange.

    Parameters are re-homed into two flat fp32 buffers (decayed / not decayed) and their ``.grad`` into matching flat gradient
    buffers, so the all-reduce works on a few large contiguous buckets (xGMI rings are per-link bound: few, large messages) and the
    update is two launches.  fp32 parameters only.  After ``step()`` the packed bf16 copies of the model are invalidated (the
    kernel writes parameters without touching ``_version``)."""

    def __init__(self, model, lr: float, b1: float = 0.9, b2: float = 0.95, eps: float = 1e-8, wd: float = 0.2,
                 clip_norm: float = None, decay_filter=default_decay_filter, bucket_bytes: int = 256 << 20):
        self.model, self.lr, self.b1, self.b2, self.eps, self.wd, self.clip_norm = model, lr, b1, b2, eps, wd, clip_norm
        self.bucket_bytes = int(bucket_bytes)
        self.t = 0
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        dev = named[0][1].device          # CPU tensors: buffers and the gradient exchange work (gloo tests); step() refuses
        self.groups = []
        for decayed in (True, False):
            ps = [(n, p) for n, p in named if bool(decay_filter(n, p)) == decayed]
            if not ps:
                continue
            if any(p.dtype != torch.float32 for _, p in ps):
                raise _lib.OvhipError("FusedAdamW: fp32 master parameters only")
            offs, total = [], 0
            for _, p in ps:
                offs.append(total)
                total += (p.numel() + 3) // 4 * 4                       # every tensor starts 16-byte aligned
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            grad = torch.zeros(total, dtype=torch.float32, device=dev)
            with torch.no_grad():
                for (_, p), o in zip(ps, offs):
                    flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
                    p.data = flat[o:o + p.numel()].view(p.shape)
                    p.grad = grad[o:o + p.numel()].view(p.shape)
            self.groups.append(dict(params=ps, offs=offs, flat=flat, grad=grad, wd=wd if decayed else 0.0,
                                    mu=torch.zeros(total, dtype=torch.bfloat16, device=dev),
                                    nu=torch.zeros(total, dtype=torch.float32, device=dev)))
        self._gn = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = torch.empty(4096 + 256, dtype=torch.uint8, device=dev)            # >= ov_sumsq_workspace_bytes()
        from .model import invalidate_packed
        invalidate_packed()

    def state_dict(self) -> dict:
        """Step count and the flat moments per group, with the names, shapes and offsets that say what lies where (the reference
        trainer checkpoints its optax state the same way: src/main_clip.py, the `opt` entry of the checkpoint tree)."""
        return dict(t=self.t, hyper=dict(b1=self.b1, b2=self.b2, eps=self.eps),
                    groups=[dict(names=[n for n, _ in g["params"]], shapes=[tuple(p.shape) for _, p in g["params"]], offs=list(g["offs"]),
                                 wd=g["wd"], mu=g["mu"].detach().clone(), nu=g["nu"].detach().clone()) for g in self.groups])

    def load_state_dict(self, sd: dict) -> None:
        """Restore what `state_dict` returned; the parameter layout (names, shapes, offsets per group) must be the one of this model."""
        if len(sd["groups"]) != len(self.groups):
            raise ValueError(f"optimizer state has {len(sd['groups'])} groups, this optimizer {len(self.groups)}")
        for g, s in zip(self.groups, sd["groups"]):
            mine = ([n for n, _ in g["params"]], [tuple(p.shape) for _, p in g["params"]], list(g["offs"]))
            if (list(s["names"]), [tuple(x) for x in s["shapes"]], list(s["offs"])) != mine:
                raise ValueError("optimizer state was saved for a different parameter layout")
        for g, s in zip(self.groups, sd["groups"]):
            g["mu"].copy_(s["mu"].to(g["mu"].device, torch.bfloat16))
            g["nu"].copy_(s["nu"].to(g["nu"].device, torch.float32))
        self.t = int(sd["t"])

    def zero_grad(self) -> None:
        """Zero the flat gradient buffers; ``.grad`` stays a view of them (set_to_none would detach the views)."""
        for g in self.groups:
            g["grad"].zero_()
            for (_, p), o in zip(g["params"], g["offs"]):
                want = g["grad"][o:o + p.numel()]
                if p.grad is None or p.grad.data_ptr() != want.data_ptr():
                    p.grad = want.view(p.shape)
        self._rearm()

    def _collect(self) -> None:
        """A ``.grad`` that autograd replaced instead of accumulating in place is copied back into its flat slot; the slot of a
        parameter whose ``.grad`` is None (set_to_none by someone else's zero_grad) is zeroed, so no stale gradient is applied."""
        for g in self.groups:
            for (_, p), o in zip(g["params"], g["offs"]):
                want = g["grad"][o:o + p.numel()]
                if p.grad is None:                                   # model.zero_grad(set_to_none=True): no gradient this step,
                    want.zero_()                                     # not last step's values
                    p.grad = want.view(p.shape)
                elif p.grad.data_ptr() != want.data_ptr():
                    want.copy_(p.grad.detach().reshape(-1).float())
                    p.grad = want.view(p.shape)

    def buckets(self):
        """The flat gradient buffers cut into contiguous chunks of at most ``bucket_bytes``."""
        per = max(1, self.bucket_bytes // 4)
        for g in self.groups:
            n = g["grad"].numel()
            for s in range(0, n, per):
                yield g["grad"][s:min(n, s + per)]

    def overlap_gradient_exchange(self, world_size: int, group=None, always_collective: bool = False) -> None:
        """Start each bucket's SUM all-reduce from autograd, as soon as the last parameter gradient that lies in the bucket has been
        accumulated (post-accumulate-grad hooks), instead of after the whole backward -- DistributedDataParallel's bucket overlap
        (the reference trainer: src/main_clip.py wraps the model the same way) on the flat buffers of this optimiser.  With
        ``set_backward_chunk_layers(n)`` the towers' gradients arrive chunk by chunk, so the exchange of the later blocks' buckets
        runs under the backward of the earlier ones.  ``all_reduce_gradients`` then only launches what is left and waits.  Call
        ``zero_grad()`` (this class's) before every backward: it re-arms the buckets.  ``always_collective`` issues the collectives
        in a world of one rank as well (exercises the RCCL calls on a single GPU)."""
        for h in getattr(self, "_ov_hooks", []):                # a second call (new bucket size, new group) replaces the hooks:
            h.remove()                                          # two hooks per parameter would arm the buckets at half the gradients
        self._ov_hooks = []
        self._ov = dict(world=int(world_size), group=group, works=[], pending=[], launched=[], force=bool(always_collective))
        self._ov_buckets = []                                   # (group index, start, stop)
        per = max(1, self.bucket_bytes // 4)
        for gi, g in enumerate(self.groups):
            n = g["grad"].numel()
            self._ov_buckets += [(gi, s, min(n, s + per)) for s in range(0, n, per)]
        self._ov_members = [[] for _ in self._ov_buckets]       # parameters overlapping each bucket
        for gi, g in enumerate(self.groups):
            for pi, ((_, p), o) in enumerate(zip(g["params"], g["offs"])):
                mine = [bi for bi, (bg, s, e) in enumerate(self._ov_buckets) if bg == gi and s < o + p.numel() and o < e]
                for bi in mine:
                    self._ov_members[bi].append((gi, pi))
                self._ov_hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(gi, pi, o, mine)))
        self._rearm()

    def _rearm(self) -> None:
        ov = getattr(self, "_ov", None)
        if ov is not None:
            ov["works"], ov["launched"] = [], [False] * len(self._ov_buckets)
            ov["pending"] = [len(m) for m in self._ov_members]

    def _launch_bucket(self, bi: int) -> None:
        import torch.distributed as dist
        ov = self._ov
        gi, s, e = self._ov_buckets[bi]
        ov["launched"][bi] = True
        if ov["world"] > 1 or ov["force"]:             # force: issue the collective in a world of one rank too (RCCL path test)
            ov["works"].append(dist.all_reduce(self.groups[gi]["grad"][s:e], op=dist.ReduceOp.SUM, group=ov["group"], async_op=True))

    def _make_hook(self, gi: int, pi: int, off: int, buckets):
        def hook(p):
            want = self.groups[gi]["grad"][off:off + p.numel()]
            if p.grad is not None and p.grad.data_ptr() != want.data_ptr():      # autograd replaced the view: fold it back first
                want.copy_(p.grad.detach().reshape(-1).float())
                p.grad = want.view(p.shape)
            ov = self._ov
            for bi in buckets:
                ov["pending"][bi] -= 1
                if ov["pending"][bi] == 0 and not ov["launched"][bi]:
                    self._launch_bucket(bi)
        return hook

    def all_reduce_gradients(self, world_size: int, group=None) -> float:
        """SUM all-reduce of every bucket (RCCL when the backend is 'nccl'), all issued before any is waited for; returns the factor
        (1 / world_size) that ``step(grad_scale=...)`` folds into the update instead of a separate averaging pass.  After
        ``overlap_gradient_exchange`` most buckets are already in flight: the rest (parameters that received no gradient) is
        launched here, then everything is waited for."""
        import torch.distributed as dist
        ov = getattr(self, "_ov", None)
        if ov is not None:
            self._collect()
            for bi in range(len(self._ov_buckets)):
                if not ov["launched"][bi]:
                    self._launch_bucket(bi)
            for w in ov["works"]:
                w.wait()
            ov["works"] = []
            return 1.0 / ov["world"]
        self._collect()
        if world_size > 1:
            works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True) for b in self.buckets()]
            for w in works:
                w.wait()
        return 1.0 / world_size

    def step(self, lr: float = None, grad_scale: float = 1.0) -> None:
        """One update at learning rate ``lr`` (default: the constructor's; the caller evaluates its schedule on the host)."""
        if self._gn.device.type != "cuda":
            raise _lib.OvhipError("FusedAdamW.step: parameters must live on an MI355X device (no CPU fallback)")
        lib = _lib.load()
        if self._ws.numel() < lib.ov_sumsq_workspace_bytes():
            self._ws = torch.empty(lib.ov_sumsq_workspace_bytes() + 256, dtype=torch.uint8, device=self._gn.device)
        self._collect()
        self.t += 1
        lr = self.lr if lr is None else lr
        gn = None
        if self.clip_norm is not None:
            for i, g in enumerate(self.groups):
                check(lib.ov_sumsq(ptr(g["grad"]), g["grad"].numel(), ptr(self._gn), int(i > 0), ptr(self._ws), self._ws.numel(),
                                   stream_ptr()), "ov_sumsq")
            gn = self._gn
        for g in self.groups:
            check(lib.ov_adamw_step(ptr(g["flat"]), ptr(g["grad"]), ptr(g["mu"]), ptr(g["nu"]), g["flat"].numel(), float(lr), self.b1,
                                    self.b2, self.eps, float(g["wd"]), self.t, float(grad_scale), ptr(gn) if gn is not None else None,
                                    float(self.clip_norm or 0.0), stream_ptr()), "ov_adamw_step")
        from .model import invalidate_packed
        invalidate_packed()
