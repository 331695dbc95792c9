"""MI355X-native counterpart of the reference's ``open_clip.model.CLIP`` for OpenVision configs.

Same constructor, attribute tree, state-dict keys and methods as the reference
(``src/convert_upload/open_clip/model.py:220-315``, ``transformer.py:210-265,319-366,434-651``), so that
``ov-zero-shot-test.py``-style callers (which walk ``model.visual.conv1 / .ln_pre / .transformer / .ln_post /
.proj`` and ``model.token_embedding / .transformer / .ln_final / .text_projection``) run unchanged — but every
module's ``forward`` enqueues hand-written gfx950 kernels through the C ABI of ``libovhip.so``.

Precision: the HIP path computes like the reference's 'bf16' mode (``factory.py:259,275-296``): bf16 weights and
activations with fp32 accumulation, LayerNorm statistics / affine, biases and the returned embeddings in fp32.
Parameters stay ordinary ``nn.Parameter``s (fp32 by default) so ``load_state_dict`` / ``.to()`` / ``.float()``
behave as usual; bf16 device copies in the kernels' packed layout are rebuilt lazily when a parameter changes.

These entry points are the inference path: they never build an autograd graph and their outputs have ``requires_grad=False``;
the training step with gradients is ``openvision_amd.training``.
There is NO CPU / eager fallback: calling any forward with CPU tensors, or without libovhip.so, raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from typing import Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import OV_BF16, OV_F32, EPI_BIAS, EPI_GELU_ERF, EPI_GELU_TANH, EPI_RESIDUAL, ptr, stream_ptr, check
from .config import (CLIPVisionCfg, CLIPTextCfg, vision_cfg_from, text_cfg_from, gelu_is_tanh, ln_eps,
                     mlp_width)

MAX_MICRO_BATCH = 1024          # encode_* split larger batches so the workspace stays bounded


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.OvhipError(f"{what}: expected a tensor on an MI355X device, got {t.device}. "
                              f"openvision_amd has no CPU fallback (the reference CPU path lives in oracle/ for tests).")


def _dtype_flag(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return OV_F32
    if t.dtype == torch.bfloat16:
        return OV_BF16
    raise TypeError(f"unsupported dtype {t.dtype} (float32 or bfloat16)")


class _Workspace:
    """Grow-only device scratch buffer, one per module tree and device."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.gen = 0            # bumped on every reallocation: captured hipGraphs hold the OLD buffer's address (_GraphCache)

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.device != device or self.buf.numel() < nbytes:
            self.buf = None
            self.buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
            self.gen += 1
        return self.buf


_PACK_EPOCH = [0]        # bumped by invalidate_packed(): every _Packed signature carries it


class _Packed:
    """Cache of packed device tensors keyed by the source parameters' (data_ptr, version).  Writes that bypass autograd's
    version counter (``p.data.copy_()``, ``p.data.mul_()``, EMA / weight surgery through ``.data``) are NOT seen: call
    ``invalidate_packed()`` (or ``CLIP.invalidate_packed()``) after them.  ``load_state_dict`` does it by itself."""

    def __init__(self):
        self.sig = None
        self.val = None

    def get(self, params, builder, extra=None):
        sig = (extra, _PACK_EPOCH[0]) + tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)
        if sig != self.sig:
            with torch.no_grad():
                self.val = builder()
            self.sig = sig
        return self.val


def invalidate_packed() -> None:
    """Drop every packed / folded / quantised device copy of every model in this process; they are rebuilt on next use."""
    _PACK_EPOCH[0] += 1


class _GraphCache:
    """hipGraph replay of a launch-bound encode call (batch-1 zero-shot path: ~170 kernel launches per image whose host-side
    launch cost exceeds their device time).  The launch sequence is captured ONCE per (shape, dtype, flags) on a capture stream
    (every kernel of libovhip goes to the current torch stream, so torch.cuda.graph records them as graph kernel nodes), then
    replayed: copy the input into the captured call's static input, one hipGraphLaunch, clone the static output.  Weights AND the
    grow-only workspaces are baked in as pointers: `sig_fn()` covers both (pack epoch, parameter versions, workspace generations);
    when it changes -- a parameter update, or a larger call (graphed or not) that made a workspace reallocate, after which the
    old buffer may belong to somebody else -- every graph is dropped and re-captured on next use."""

    def __init__(self):
        self.graphs = {}
        self.sig = None

    def run(self, key, sig_fn, x: torch.Tensor, fn):
        if self.sig != sig_fn():
            self.graphs.clear()
        ent = self.graphs.get(key)
        if ent is None:
            static_in = x.clone()
            fn(static_in)                                   # warm-up outside capture: packs weights, sizes workspaces, sets attributes
            torch.cuda.current_stream(x.device).synchronize()
            if self.sig != sig_fn():                        # the warm-up itself grew a workspace: the other graphs hold stale pointers
                self.graphs.clear()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = fn(static_in)
            ent = self.graphs[key] = (g, static_in, static_out)
            self.sig = sig_fn()
        g, static_in, static_out = ent
        static_in.copy_(x)
        g.replay()
        return static_out.clone()


def _pack_matrix(w: torch.Tensor, n_pad: int, k_pad: int) -> torch.Tensor:
    """[N,K] any float dtype -> zero-padded bf16 [n_pad, k_pad], contiguous."""
    n, k = w.shape
    out = torch.zeros(n_pad, k_pad, dtype=torch.bfloat16, device=w.device)
    out[:n, :k] = w.detach().to(torch.bfloat16)
    return out


def _pack_vec(v: Optional[torch.Tensor], n_pad: int, device) -> torch.Tensor:
    out = torch.zeros(n_pad, dtype=torch.float32, device=device)
    if v is not None:
        out[: v.numel()] = v.detach().float()
    return out


# ------------------------------------------------------------------------------------------------------
# leaf modules
# ------------------------------------------------------------------------------------------------------
class LayerNorm(nn.LayerNorm):
    """transformer.py:24-30 / :15-21 — fp32 statistics, output cast back to the input dtype."""

    def __init__(self, normalized_shape, eps: float = 1e-6):
        super().__init__(normalized_shape, eps=eps)
        self._pk = _Packed()

    def packed(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._pk.get((self.weight, self.bias),
                            lambda: (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_cuda(x, "LayerNorm")
        d = self.normalized_shape[0]
        if x.shape[-1] != d:
            raise ValueError(f"LayerNorm: last dim {x.shape[-1]} != {d}")
        lib = _lib.load()
        xin = x if x.dtype in (torch.float32, torch.bfloat16) else x.float()
        xin = xin.contiguous().view(-1, d)
        y = torch.empty_like(xin)
        g, b = self.packed()
        check(lib.ov_layernorm(ptr(xin), _dtype_flag(xin), d, ptr(g), ptr(b), ptr(y), _dtype_flag(y), d,
                               xin.shape[0], d, float(self.eps), stream_ptr()), "ov_layernorm")
        return y.view(x.shape).to(x.dtype)


class Linear(nn.Module):
    """Parameter holder with nn.Linear's attribute names; forward = ov_gemm (+bias)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.normal_(self.weight, std=in_features ** -0.5)
        if bias:
            nn.init.zeros_(self.bias)
        self._pk = _Packed()

    def packed(self, n_pad: Optional[int] = None, k_pad: Optional[int] = None):
        n_pad = n_pad or _round_up(self.out_features, 8)
        k_pad = k_pad or _round_up(self.in_features, 64)
        ps = (self.weight,) if self.bias is None else (self.weight, self.bias)
        return self._pk.get(ps, lambda: (_pack_matrix(self.weight, n_pad, k_pad),
                                         _pack_vec(self.bias, n_pad, self.weight.device)), extra=(n_pad, k_pad))

    def forward(self, x: torch.Tensor, epilogue: int = EPI_BIAS) -> torch.Tensor:
        _require_cuda(x, "Linear")
        k = self.in_features
        w, b = self.packed()
        n_pad, k_pad = w.shape
        a = x.reshape(-1, k).to(torch.bfloat16)
        if k_pad != k:
            a = torch.nn.functional.pad(a, (0, k_pad - k))
        a = a.contiguous()
        out = torch.empty(a.shape[0], n_pad, dtype=torch.bfloat16, device=x.device)
        check(_lib.load().ov_gemm(ptr(a), k_pad, ptr(w), k_pad, ptr(b), ptr(out), n_pad, a.shape[0], n_pad, k_pad,
                                  epilogue, None, 0, 0, 0, 0, stream_ptr()), "ov_gemm")
        return out[:, : self.out_features].reshape(*x.shape[:-1], self.out_features).to(x.dtype)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class GELU(nn.GELU):
    """Marker module (``mlp.gelu``): inside a block the activation is fused into the c_fc GEMM epilogue."""

    def forward(self, x):  # pragma: no cover - never on the product path
        raise _lib.OvhipError("GELU is fused into the c_fc GEMM epilogue on the HIP path; call the block or the mlp's "
                              "parent, not mlp.gelu directly")


class MultiheadAttention(nn.Module):
    """Parameter layout of nn.MultiheadAttention(batch_first=True) (packed in_proj, out_proj) — transformer.py:225."""

    def __init__(self, embed_dim: int, num_heads: int):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.batch_first = True
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = Linear(embed_dim, embed_dim, bias=True)
        nn.init.normal_(self.in_proj_weight, std=embed_dim ** -0.5)
        self._pk = _Packed()

    def packed_in(self):
        d = self.embed_dim
        return self._pk.get((self.in_proj_weight, self.in_proj_bias),
                            lambda: (_pack_matrix(self.in_proj_weight, 3 * d, _round_up(d, 64)),
                                     _pack_vec(self.in_proj_bias, 3 * d, self.in_proj_weight.device)))


class PatchConv(nn.Module):
    """conv1: Conv2d(3, width, kernel = stride = P, bias=False) (transformer.py:469) as im2col + MFMA GEMM."""

    def __init__(self, width: int, patch: int):
        super().__init__()
        self.in_channels, self.out_channels = 3, width
        self.kernel_size = self.stride = (patch, patch)
        self.weight = nn.Parameter(torch.empty(width, 3, patch, patch))
        nn.init.normal_(self.weight, std=(3 * patch * patch) ** -0.5)
        self.bias = None
        self._pk = _Packed()

    @property
    def kpad(self) -> int:
        return _round_up(3 * self.kernel_size[0] ** 2, 64)

    def packed(self) -> torch.Tensor:
        return self._pk.get((self.weight,), lambda: _pack_matrix(self.weight.reshape(self.out_channels, -1),
                                                                 self.out_channels, self.kpad))

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        """[B,3,S,S] -> [B, width, g, g] (what nn.Conv2d returns; ov-zero-shot-test.py:105)."""
        _require_cuda(image, "conv1")
        lib = _lib.load()
        p = self.kernel_size[0]
        bsz, _, s, _ = image.shape
        g = s // p
        img = image.contiguous() if image.dtype in (torch.float32, torch.bfloat16) else image.float().contiguous()
        cols = torch.empty(bsz * g * g, self.kpad, dtype=torch.bfloat16, device=image.device)
        check(lib.ov_im2col_patches(ptr(img), _dtype_flag(img), ptr(cols), bsz, s, p, self.kpad, stream_ptr()), "ov_im2col")
        w = self.packed()
        out = torch.empty(bsz * g * g, self.out_channels, dtype=torch.bfloat16, device=image.device)
        check(lib.ov_gemm(ptr(cols), self.kpad, ptr(w), self.kpad, None, ptr(out), self.out_channels, bsz * g * g,
                          self.out_channels, self.kpad, EPI_BIAS, None, 0, 0, 0, 0, stream_ptr()), "ov_gemm(conv1)")
        return out.view(bsz, g, g, self.out_channels).permute(0, 3, 1, 2).to(image.dtype)


class Embedding(nn.Embedding):
    """token_embedding.  As a stand-alone call it is a pure row gather (index_select, no arithmetic);
    ``CLIP.encode_text`` uses the fused HIP gather + pos-emb add instead."""

    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__(num_embeddings, embedding_dim)
        self._pk = _Packed()

    def packed(self) -> torch.Tensor:
        return self._pk.get((self.weight,), lambda: self.weight.detach().to(torch.bfloat16).contiguous())

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        _require_cuda(tokens, "token_embedding")
        return self.weight.detach().index_select(0, tokens.reshape(-1)).view(*tokens.shape, -1)


# ------------------------------------------------------------------------------------------------------
# transformer
# ------------------------------------------------------------------------------------------------------
class ResidualAttentionBlock(nn.Module):
    """transformer.py:210-265 with ls_1 = ls_2 = Identity and no cross-attention."""

    def __init__(self, d_model: int, n_head: int, mlp_ratio: float = 4.0, act_kwargs: Optional[dict] = None,
                 eps: float = 1e-6):
        super().__init__()
        self.ln_1 = LayerNorm(d_model, eps=eps)
        self.attn = MultiheadAttention(d_model, n_head)
        self.ls_1 = nn.Identity()
        self.ln_2 = LayerNorm(d_model, eps=eps)
        mlp = mlp_width(d_model, mlp_ratio)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", Linear(d_model, mlp)),
            ("gelu", GELU(**(act_kwargs or {}))),
            ("c_proj", Linear(mlp, d_model)),
        ]))
        self.ls_2 = nn.Identity()
        self.gelu_tanh = gelu_is_tanh(act_kwargs)
        self.mlp_dim, self.mlp_pad = mlp, _round_up(mlp, 64)
        self._fold_pk = _Packed()
        self._fp8_pk = _Packed()

    def packed_block(self, fold_ln: bool = True):
        """Device weights of this block in the kernels' layout.  With ``fold_ln`` (default; ``OVHIP_NO_LN_FOLD=1`` disables)
        ln_1 / ln_2 are folded into the QKV / c_fc GEMMs: W' = bf16(gamma * W), colsum = sum_k W', cvec = beta @ W^T + b
        (see ov_gemm_ln in include/ovhip.h); the LayerNorm output is never materialised."""
        d = self.attn.embed_dim
        dev = self.attn.in_proj_weight.device
        g1, b1 = self.ln_1.packed()
        g2, b2 = self.ln_2.packed()
        wo, bo = self.attn.out_proj.packed(d, _round_up(d, 64))
        wp, bp = self.mlp.c_proj.packed(d, self.mlp_pad)
        if not fold_ln:
            wq, bq = self.attn.packed_in()
            wf, bf = self.mlp.c_fc.packed(self.mlp_pad, _round_up(d, 64))
            keep = (g1, b1, wq, bq, wo, bo, g2, b2, wf, bf, wp, bp)
            return _lib.BlockWeights(*[C.c_void_p(t.data_ptr()) for t in keep], None, None), keep

        def fold(w, b, gamma, beta, n_pad):
            w32 = w.detach().float()
            wg = _pack_matrix(w32 * gamma[None, :], n_pad, _round_up(d, 64))          # bf16(gamma * W), zero padded
            colsum = wg.float().sum(dim=1).contiguous()                               # sums of the ROUNDED weights
            cvec = _pack_vec((w32.to(torch.bfloat16).float() @ beta) + (b.detach().float() if b is not None else 0.0),
                             n_pad, dev)
            return wg, cvec, colsum

        key = (self.ln_1.weight, self.ln_1.bias, self.ln_2.weight, self.ln_2.bias, self.attn.in_proj_weight,
               self.attn.in_proj_bias, self.mlp.c_fc.weight, self.mlp.c_fc.bias)
        wq, cq, sq, wf, cf, sf = self._fold_pk.get(key, lambda: fold(self.attn.in_proj_weight, self.attn.in_proj_bias, g1, b1, 3 * d)
                                                    + fold(self.mlp.c_fc.weight, self.mlp.c_fc.bias, g2, b2, self.mlp_pad))
        keep = (g1, b1, wq, cq, wo, bo, g2, b2, wf, cf, wp, bp, sq, sf)
        return _lib.BlockWeights(*[C.c_void_p(t.data_ptr()) for t in keep]), keep

    def packed_block_fp8(self):
        """fp8 (OCP e4m3) copies of the four weight matrices for the fp8 path (BASELINE.json config #5): per-output-row absmax
        scaling (scale = max|w_row| / 448), zero padding as in the bf16 layout; biases as the module holds them."""
        d = self.attn.embed_dim
        dev = self.attn.in_proj_weight.device

        def q8(w, n_pad, k_pad):
            w32 = torch.zeros(n_pad, k_pad, dtype=torch.float32, device=dev)
            w32[: w.shape[0], : w.shape[1]] = w.detach().float()
            scale = (w32.abs().amax(dim=1) / 448.0).clamp_min(1e-30)
            return (w32 / scale[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous(), scale.contiguous()

        def build():
            wq, sq = q8(self.attn.in_proj_weight, 3 * d, d)
            wo, so = q8(self.attn.out_proj.weight, d, d)
            wf, sf = q8(self.mlp.c_fc.weight, self.mlp_pad, d)
            wp, sp = q8(self.mlp.c_proj.weight, d, self.mlp_pad)
            bq = _pack_vec(self.attn.in_proj_bias, 3 * d, dev)
            bf = _pack_vec(self.mlp.c_fc.bias, self.mlp_pad, dev)
            return (wq, sq, bq, wo, so, wf, sf, bf, wp, sp)

        key = (self.attn.in_proj_weight, self.attn.in_proj_bias, self.attn.out_proj.weight, self.mlp.c_fc.weight,
               self.mlp.c_fc.bias, self.mlp.c_proj.weight)
        keep = self._fp8_pk.get(key, build)
        return _lib.BlockFp8(*[C.c_void_p(t.data_ptr()) for t in keep]), keep

    def forward(self, q_x: torch.Tensor, k_x=None, v_x=None, attn_mask=None) -> torch.Tensor:
        if k_x is not None or v_x is not None or attn_mask is not None:
            raise NotImplementedError("OpenVision blocks are unmasked self-attention (no k_x/v_x/attn_mask)")
        return _run_blocks([self], q_x)


class _TowerHandle:
    """ov_tower handle + the packed tensors it borrows (kept alive here)."""

    def __init__(self, blocks, fp8: bool = False, mask=None):
        lib = _lib.load()
        b0 = blocks[0]
        d = b0.attn.embed_dim
        if d % 64:
            raise _lib.OvhipError(f"width {d} must be a multiple of 64 for the gfx950 kernels")
        cfg = _lib.TowerCfg(d, len(blocks), b0.attn.num_heads, b0.mlp_dim, b0.mlp_pad, int(b0.gelu_tanh), float(b0.ln_1.eps))
        self.handle = lib.ov_tower_create(C.byref(cfg))
        if not self.handle:
            raise _lib.OvhipError("ov_tower_create failed (invalid tower configuration)")
        self.keep = []
        for i, blk in enumerate(blocks):
            bw, keep = blk.packed_block(fold_ln=os.environ.get("OVHIP_NO_LN_FOLD", "0") != "1")
            self.keep.append(keep)
            check(lib.ov_tower_set_block(self.handle, i, C.byref(bw)), "ov_tower_set_block")
            if fp8:
                b8, keep8 = blk.packed_block_fp8()
                self.keep.append(keep8)
                check(lib.ov_tower_set_block_fp8(self.handle, i, C.byref(b8)), "ov_tower_set_block_fp8")
        self.width, self.layers, self.fp8 = d, len(blocks), fp8
        self.h_amax = None
        self.mask = [FP8_ALL] * len(blocks) if mask is None else list(mask)
        if fp8 and mask is not None:
            m8 = (C.c_ubyte * len(blocks))(*self.mask)
            check(lib.ov_tower_set_fp8_mask(self.handle, m8, len(blocks)), "ov_tower_set_fp8_mask")
        if fp8:
            # per-layer running maxima of the MLP hidden and of the attention output: recorded while they are quantised row by row
            # (mode 1), then the static scales of the fused c_fc -> c_proj and attention -> out_proj hand-overs (mode 2)
            # [hidden | attention out] maxima behind the scales, then the same two sets as recorded by the running forward
            self.h_amax = torch.zeros(4 * len(blocks), dtype=torch.float32, device=b0.attn.in_proj_weight.device)
            check(lib.ov_tower_set_fp8_hidden_scale(self.handle, ptr(self.h_amax), 1), "ov_tower_set_fp8_hidden_scale")

    def freeze_fp8_scales(self, delayed: bool = True) -> None:
        if not self.fp8:
            raise _lib.OvhipError("freeze_fp8_scales: the tower is not in fp8 precision")
        need = [(m & FP8_FC) and (m & FP8_PROJ) for m in self.mask] + [bool(m & FP8_OUT) for m in self.mask]   # scales the mask uses
        seen = (self.h_amax[: 2 * self.layers] > 0).tolist()
        if any(n and not s_ for n, s_ in zip(need, seen)):
            raise _lib.OvhipError("freeze_fp8_scales: run at least one forward in fp8 precision first (calibration)")
        check(_lib.load().ov_tower_set_fp8_hidden_scale(self.handle, ptr(self.h_amax), 2 if delayed else 3), "ov_tower_set_fp8_hidden_scale")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().ov_tower_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _block_params(blocks):
    ps = []
    for b in blocks:
        ps.extend(p for p in b.parameters())
    return ps


FP8_QKV, FP8_OUT, FP8_FC, FP8_PROJ, FP8_ALL = 1, 2, 4, 8, 15      # ovhip.h OV_FP8_*: which GEMMs of a block take e4m3 operands
PRECISIONS = ("bf16", "fp8", "fp8-mixed")


def fp8_mixed_mask(layers: int):
    """The "fp8-mixed" recipe (BASELINE.json config #5 inside a stated tolerance; DESIGN.md section 7, profiles/r03_fp8_ablation.md):
    e4m3 operands where the per-GEMM ablation showed them cheap in accuracy and rich in time -- the two MLP products (c_fc -> c_proj, the
    hidden handed over in e4m3 with a static scale) -- and bf16 for the attention projections (QKV, out_proj), whose quantisation is
    what moves the embeddings on ill-conditioned ('sharp') weights.  OVHIP_FP8_MIXED_MASK (an OV_FP8_* bit set) overrides it."""
    m = int(os.environ.get("OVHIP_FP8_MIXED_MASK", str(FP8_FC | FP8_PROJ)))
    return [m & FP8_ALL] * layers


def default_precision() -> str:
    """GEMM operand precision of the block stacks: "bf16" (default), "fp8" (OVHIP_PRECISION=fp8; e4m3 weights and activations on the
    MX-scaled MFMA for all four GEMMs of a block, BASELINE.json config #5) or "fp8-mixed" (fp8_mixed_mask)."""
    p = os.environ.get("OVHIP_PRECISION", "bf16").lower()
    if p not in PRECISIONS:
        raise ValueError(f"OVHIP_PRECISION={p!r}: expected one of {PRECISIONS}")
    return p


class _TowerCache:
    def __init__(self):
        self._pk = _Packed()
        self.precision = default_precision()
        self.mask = None             # explicit per-layer OV_FP8_* masks (set_fp8_mask); None = what the precision name implies

    def get(self, blocks) -> _TowerHandle:
        fp8 = self.precision != "bf16"
        mask = None
        if fp8:
            mask = self.mask if self.mask is not None else (fp8_mixed_mask(len(blocks)) if self.precision == "fp8-mixed" else None)
            if mask is not None and len(mask) != len(blocks):
                mask = (list(mask) + [mask[-1]] * len(blocks))[: len(blocks)]        # sub-stacks (exploded forward): layer-wise prefix
        key = (fp8, None if mask is None else tuple(mask))
        return self._pk.get(_block_params(blocks), lambda: _TowerHandle(blocks, fp8=fp8, mask=mask), extra=key)


def _run_blocks(blocks, x: torch.Tensor, cache: Optional[_TowerCache] = None, ws: Optional[_Workspace] = None):
    """x [B, L, D] (fp32 or bf16, cuda) -> same shape/dtype after the given blocks (ov_tower_forward)."""
    _require_cuda(x, "Transformer")
    if x.dim() != 3:
        raise ValueError("expected [batch, tokens, width]")
    lib = _lib.load()
    tower = (cache or _TowerCache()).get(blocks)
    bsz, seq, d = x.shape
    if d != tower.width:
        raise ValueError(f"width {d} != {tower.width}")
    xb = x.detach().to(torch.bfloat16).contiguous().clone() if x.dtype == torch.bfloat16 else x.detach().to(torch.bfloat16).contiguous()
    nbytes = lib.ov_tower_workspace_bytes(tower.handle, bsz, seq)
    wsb = (ws or _Workspace()).get(nbytes, x.device)
    check(lib.ov_tower_forward(tower.handle, ptr(xb), bsz, seq, ptr(wsb), nbytes, stream_ptr()), "ov_tower_forward")
    return xb.to(x.dtype)


class Transformer(nn.Module):
    """transformer.py:319-366."""

    def __init__(self, width: int, layers: int, heads: int, mlp_ratio: float = 4.0, act_kwargs: Optional[dict] = None,
                 eps: float = 1e-6, batch_first: bool = True):
        super().__init__()
        self.width, self.layers, self.batch_first = width, layers, batch_first
        self.grad_checkpointing = False
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, mlp_ratio, act_kwargs, eps)
                                        for _ in range(layers)])
        self._cache = _TowerCache()
        self._ws = _Workspace()

    def get_cast_dtype(self) -> torch.dtype:
        return self.resblocks[0].mlp.c_fc.weight.dtype          # transformer.py:350-353

    def tower(self) -> _TowerHandle:
        return self._cache.get(list(self.resblocks))

    def set_precision(self, precision: str, mask=None) -> None:
        """"bf16", "fp8" (all four GEMMs of every block in e4m3) or "fp8-mixed" (fp8_mixed_mask); `mask`: an OV_FP8_* bit set for
        every layer, or one per layer, instead of what the name implies.  The tower is re-packed on next use."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        if mask is not None and precision == "bf16":
            raise ValueError("an fp8 mask needs an fp8 precision")
        self._cache.precision = precision
        self._cache.mask = None if mask is None else ([int(mask)] * self.layers if isinstance(mask, int) else [int(v) for v in mask])

    def freeze_fp8_scales(self, delayed: bool = True) -> None:
        """fp8 precision: after at least one forward (which records the per-layer maximum of the MLP hidden), switch c_fc -> c_proj to
        the fused hand-over with a static scale (2 x the recorded maximum / 448) -- no bf16 round trip of the hidden.  delayed=True
        (training-style): every forward first rolls the maxima recorded by the previous one into the scales (they only grow, and a
        changed scale re-rounds every e4m3 value downstream: results are then NOT bitwise repeatable from call to call);
        delayed=False (serving): the scales stay as they are: repeatable and independent of the batch composition."""
        self.tower().freeze_fp8_scales(delayed)

    def forward(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if attn_mask is not None:
            raise NotImplementedError("attn_mask is None on the OpenVision path (no_causal_mask=True)")
        return _run_blocks(list(self.resblocks), x, self._cache, self._ws)


class VisionTransformer(nn.Module):
    """transformer.py:434-651 restricted to what OpenVision instantiates (no attentional pool, patch_dropout = 0)."""

    def __init__(self, image_size: int, patch_size: int, width: int, layers: int, heads: int, mlp_ratio: float,
                 output_dim: int, pool_type: str = "avg", final_ln_after_pool: bool = True, no_ln_pre: bool = True,
                 act_kwargs: Optional[dict] = None, eps: float = 1e-6, output_tokens: bool = False):
        super().__init__()
        self.output_tokens = output_tokens
        self.image_size = (image_size, image_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.final_ln_after_pool = final_ln_after_pool
        self.output_dim = output_dim
        self.conv1 = PatchConv(width, patch_size)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.patch_dropout = nn.Identity()
        self.ln_pre = nn.Identity() if no_ln_pre else LayerNorm(width, eps=eps)
        self.transformer = Transformer(width, layers, heads, mlp_ratio, act_kwargs, eps)
        self.attn_pool = None
        self.pool_type = pool_type
        self.ln_post = LayerNorm(width, eps=eps)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._pk = _Packed()
        self._ws = _Workspace()

    # -- packed head -----------------------------------------------------------------------------------
    def _head(self):
        def build():
            dev = self.proj.device
            pos32 = self.positional_embedding.detach().float().contiguous()
            keep = dict(conv=self.conv1.packed(), cls=self.class_embedding.detach().float().contiguous(),
                        pos=pos32.to(torch.bfloat16).contiguous(), pos32=pos32, ln=self.ln_post.packed(),
                        proj_t=self.proj.detach().t().to(torch.bfloat16).contiguous())
            e = self.output_dim
            if e % 8:
                raise _lib.OvhipError("embed_dim must be a multiple of 8")
            h = _lib.VisionHead(self.image_size[0], self.patch_size[0], self.conv1.kpad, int(self.pool_type == "avg"),
                                int(self.final_ln_after_pool), e, e, ptr(keep["conv"]), ptr(keep["cls"]), ptr(keep["pos"]),
                                ptr(keep["pos32"]), ptr(keep["ln"][0]), ptr(keep["ln"][1]), ptr(keep["proj_t"]))
            return h, keep
        return self._pk.get((self.conv1.weight, self.class_embedding, self.positional_embedding, self.ln_post.weight,
                             self.ln_post.bias, self.proj), build)

    def _check_image(self, x: torch.Tensor):
        _require_cuda(x, "VisionTransformer")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.image_size[0] or x.shape[3] != self.image_size[1]:
            raise ValueError(f"expected [B,3,{self.image_size[0]},{self.image_size[1]}], got {tuple(x.shape)}")

    def _encode(self, x: torch.Tensor, normalize: bool) -> torch.Tensor:
        self._check_image(x)
        lib = _lib.load()
        head, _keep = self._head()
        tower = self.transformer.tower()
        img = x.detach()
        img = img.contiguous() if img.dtype in (torch.float32, torch.bfloat16) else img.float().contiguous()
        bsz = img.shape[0]
        out = torch.empty(bsz, self.output_dim, dtype=torch.float32, device=x.device)
        st = stream_ptr()
        for b0 in range(0, bsz, MAX_MICRO_BATCH):
            nb = min(MAX_MICRO_BATCH, bsz - b0)
            nbytes = lib.ov_vision_workspace_bytes(tower.handle, C.byref(head), nb)
            ws = self._ws.get(nbytes, x.device)
            if isinstance(self.ln_pre, nn.Identity):
                check(lib.ov_encode_image(tower.handle, C.byref(head), ptr(img[b0:]), _dtype_flag(img), nb, ptr(out[b0:]),
                                          int(normalize), ptr(ws), nbytes, st), "ov_encode_image")
            else:
                tok = self._embed_tokens(img[b0:b0 + nb])
                tok = self.transformer(self.ln_pre(tok))
                out[b0:b0 + nb] = self._head_forward(tok, normalize)
        return out

    def _embed_tokens(self, img: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        head, _ = self._head()
        tower = self.transformer.tower()
        bsz = img.shape[0]
        seq = self.grid_size[0] * self.grid_size[1] + 1
        tok = torch.empty(bsz, seq, tower.width, dtype=torch.bfloat16, device=img.device)
        nbytes = bsz * (seq - 1) * self.conv1.kpad * 2
        ws = self._ws.get(nbytes, img.device)
        check(lib.ov_vision_embed(tower.handle, C.byref(head), ptr(img), _dtype_flag(img), bsz, ptr(tok), ptr(ws), nbytes,
                                  stream_ptr()), "ov_vision_embed")
        return tok

    def _head_forward(self, tok: torch.Tensor, normalize: bool) -> torch.Tensor:
        lib = _lib.load()
        head, _ = self._head()
        tower = self.transformer.tower()
        bsz = tok.shape[0]
        tokb = tok.detach().to(torch.bfloat16).contiguous()
        out = torch.empty(bsz, self.output_dim, dtype=torch.float32, device=tok.device)
        nbytes = bsz * tower.width * 8 + bsz * self.output_dim * 2 + 4096
        ws = self._ws.get(nbytes, tok.device)
        check(lib.ov_vision_head_forward(tower.handle, C.byref(head), ptr(tokb), bsz, ptr(out), int(normalize), ptr(ws),
                                         nbytes, stream_ptr()), "ov_vision_head_forward")
        return out

    def forward(self, x: torch.Tensor):
        if self.output_tokens:
            self._check_image(x)
            img = x.detach().contiguous()
            tok = self._embed_tokens(img if img.dtype in (torch.float32, torch.bfloat16) else img.float())
            tok = self.transformer(self.ln_pre(tok))
            pooled = self._head_forward(tok, False)
            tokens = tok[:, 1:]
            if not self.final_ln_after_pool:        # transformer.py:641-645: ln_post runs on all tokens BEFORE pooling
                tokens = self.ln_post(tokens.contiguous())
            return pooled, tokens.to(x.dtype)
        return self._encode(x, False)


# ------------------------------------------------------------------------------------------------------
# CLIP
# ------------------------------------------------------------------------------------------------------
class CLIP(nn.Module):
    """Drop-in for ``open_clip.model.CLIP`` (model.py:220-315) on OpenVision configs."""
    output_dict: torch.jit.Final[bool]

    def __init__(self, embed_dim: int, vision_cfg: Union[dict, CLIPVisionCfg], text_cfg: Union[dict, CLIPTextCfg],
                 quick_gelu: bool = False, init_logit_scale: float = np.log(1 / 0.07),
                 init_logit_bias: Optional[float] = None, cast_dtype: Optional[torch.dtype] = None,
                 output_dict: bool = False):
        super().__init__()
        if quick_gelu:
            raise NotImplementedError("quick_gelu is not used by OpenVision configs")
        if init_logit_bias is not None:
            raise NotImplementedError("logit_bias (SigLIP) is outside the InfoNCE path")
        self.output_dict = output_dict
        v = vision_cfg_from(vision_cfg)
        t = text_cfg_from(text_cfg)
        self.vision_cfg, self.text_cfg, self.embed_dim = v, t, embed_dim
        self.visual = VisionTransformer(
            image_size=v.image_size, patch_size=v.patch_size, width=v.width, layers=v.layers,
            heads=v.width // v.head_width, mlp_ratio=v.mlp_ratio, output_dim=embed_dim, pool_type=v.pool_type,
            final_ln_after_pool=v.final_ln_after_pool, no_ln_pre=v.no_ln_pre, act_kwargs=v.act_kwargs,
            eps=ln_eps(v.norm_kwargs), output_tokens=v.output_tokens)
        # text tower parts are adopted at the top level, as the reference does (model.py:239-248)
        self.transformer = Transformer(t.width, t.layers, t.heads, t.mlp_ratio, t.act_kwargs, ln_eps(t.norm_kwargs))
        self.context_length = t.context_length
        self.vocab_size = t.vocab_size
        self.token_embedding = Embedding(t.vocab_size, t.width)
        self.positional_embedding = nn.Parameter(torch.empty(t.context_length, t.width))
        nn.init.normal_(self.positional_embedding, std=0.01)
        self.ln_final = LayerNorm(t.width, eps=ln_eps(t.norm_kwargs))
        self.text_projection = nn.Parameter(torch.empty(t.width, embed_dim))
        nn.init.normal_(self.text_projection, std=t.width ** -0.5)
        self.text_pool_type = t.pool_type
        self.register_buffer("attn_mask", None, persistent=False)     # no_causal_mask -> None (transformer.py:722-725)
        self.logit_scale = nn.Parameter(torch.ones([]) * float(init_logit_scale))
        self.logit_bias = None
        self._pk = _Packed()
        self._ws = _Workspace()
        self._err = None
        self._side_stream = None
        self._graphs = _GraphCache()
        self.graph_max_batch = int(os.environ.get("OVHIP_GRAPH_MAX_BATCH", "0"))    # 0 = off; see use_graphs()
        self.overlap_towers = os.environ.get("OVHIP_OVERLAP_TOWERS", "0") == "1"   # measured gain < 1 %: opt-in
        if cast_dtype is not None and cast_dtype not in (torch.float32, torch.bfloat16):
            raise NotImplementedError("cast_dtype must be float32 or bfloat16")

    # -- text head -------------------------------------------------------------------------------------
    def _text_head(self):
        def build():
            keep = dict(tok=self.token_embedding.packed(),
                        pos=self.positional_embedding.detach().to(torch.bfloat16).contiguous(), ln=self.ln_final.packed(),
                        proj_t=self.text_projection.detach().t().to(torch.bfloat16).contiguous())
            if self.embed_dim % 8:
                raise _lib.OvhipError("embed_dim must be a multiple of 8")
            h = _lib.TextHead(self.context_length, self.vocab_size, int(self.text_pool_type == "last"), self.embed_dim,
                              ptr(keep["tok"]), ptr(keep["pos"]), ptr(keep["ln"][0]), ptr(keep["ln"][1]), ptr(keep["proj_t"]))
            return h, keep
        return self._pk.get((self.token_embedding.weight, self.positional_embedding, self.ln_final.weight,
                             self.ln_final.bias, self.text_projection), build)

    # -- reference API -----------------------------------------------------------------------------------
    def set_precision(self, precision: str, mask=None) -> None:
        """GEMM operand precision of both block stacks: "bf16" (default), "fp8" (e4m3 weights and activations on the MX-scaled MFMA
        for all four GEMMs of every block) or "fp8-mixed" (e4m3 for the MLP products only: model.fp8_mixed_mask); `mask` = an explicit
        OV_FP8_* bit set instead.  Embedding, heads, attention, LayerNorm statistics and the loss stay as they are."""
        self.visual.transformer.set_precision(precision, mask)
        self.transformer.set_precision(precision, mask)

    def freeze_fp8_scales(self, delayed: bool = True) -> None:
        """fp8 precision, after a calibration forward of both towers: static scales for the MLP hidden (see Transformer)."""
        self.visual.transformer.freeze_fp8_scales(delayed)
        self.transformer.freeze_fp8_scales(delayed)

    def use_graphs(self, max_batch: int = 8) -> None:
        """Replay encode_image / encode_text calls of at most `max_batch` rows as captured hipGraphs (0 turns it off).  For the
        launch-bound small-batch path (ov-zero-shot-test.py:167-195 encodes one image at a time); results are bitwise those of the
        plain launches."""
        self.graph_max_batch = int(max_batch)

    def _weights_sig(self):
        ws = (self._ws, self.visual._ws, self.visual.transformer._ws, self.transformer._ws)
        return (_PACK_EPOCH[0],) + tuple(w.gen for w in ws) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def encode_image(self, image: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """model.py:265-267.  Returns fp32 [B, embed_dim]."""
        if 0 < image.shape[0] <= self.graph_max_batch and image.is_cuda and not torch.cuda.is_current_stream_capturing():
            self.visual._check_image(image)
            x = image.detach()
            x = x.contiguous() if x.dtype in (torch.float32, torch.bfloat16) else x.float().contiguous()
            return self._graphs.run(("img", tuple(x.shape), x.dtype, bool(normalize), self.visual.transformer._cache.precision),
                                    self._weights_sig, x, lambda t: self.visual._encode(t, normalize))
        return self.visual._encode(image, normalize)

    def encode_text(self, text: torch.Tensor, normalize: bool = False, _graphed: bool = False) -> torch.Tensor:
        """model.py:269-284.  ``text`` int64 [B, context_length] (the whole pos-emb is added: model.py:274)."""
        _require_cuda(text, "encode_text")
        if text.dim() != 2 or text.shape[1] != self.context_length:
            raise ValueError(f"expected tokens [B,{self.context_length}], got {tuple(text.shape)}")
        if 0 < text.shape[0] <= self.graph_max_batch and not torch.cuda.is_current_stream_capturing() and not _graphed:
            t = text.detach().to(torch.int64).contiguous()
            return self._graphs.run(("txt", tuple(t.shape), bool(normalize), self.transformer._cache.precision), self._weights_sig,
                                    t, lambda z: self.encode_text(z, normalize, _graphed=True))
        lib = _lib.load()
        head, _keep = self._text_head()
        tower = self.transformer.tower()
        tok = text.detach().to(torch.int64).contiguous()
        bsz = tok.shape[0]
        out = torch.empty(bsz, self.embed_dim, dtype=torch.float32, device=text.device)
        if self._err is None or self._err.device != text.device:
            self._err = torch.zeros(1, dtype=torch.int32, device=text.device)
        st = stream_ptr()
        mb = MAX_MICRO_BATCH * 4
        for b0 in range(0, bsz, mb):
            nb = min(mb, bsz - b0)
            nbytes = lib.ov_text_workspace_bytes(tower.handle, C.byref(head), nb)
            ws = self._ws.get(nbytes, text.device)
            check(lib.ov_encode_text(tower.handle, C.byref(head), ptr(tok[b0:]), nb, ptr(out[b0:]), int(normalize),
                                     ptr(self._err), ptr(ws), nbytes, st), "ov_encode_text")
        return out

    def check_token_range(self) -> None:
        """Synchronising check of the device-side flag set when a token id was outside [0, vocab_size)
        (torch's nn.Embedding raises IndexError for that: model.py:272)."""
        if self._err is not None and int(self._err.item()) != 0:
            self._err.zero_()
            raise IndexError("token id out of range in encode_text")

    def get_logits(self, image, text):
        """model.py:286-293."""
        i = self.encode_image(image, normalize=True)
        t = self.encode_text(text, normalize=True)
        li = logits(i, t, self.logit_scale.detach().exp())       # the scale stays on the device (model.py:288)
        return li, li.T

    def forward(self, image: Optional[torch.Tensor] = None, text: Optional[torch.Tensor] = None):
        """model.py:295-315."""
        if image is not None and text is not None and self.overlap_towers and text.is_cuda:
            # The towers are independent until the loss: the text tower runs on a side stream so that its small,
            # launch/latency-bound kernels (and both towers' HBM-bound LayerNorms) fill CUs the other tower leaves idle.
            cur = torch.cuda.current_stream(text.device)
            if self._side_stream is None or self._side_stream.device != text.device:
                self._side_stream = torch.cuda.Stream(device=text.device)
            side = self._side_stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                text_features = self.encode_text(text, normalize=True)
            image_features = self.encode_image(image, normalize=True)
            cur.wait_stream(side)
            text_features.record_stream(cur)
        else:
            image_features = self.encode_image(image, normalize=True) if image is not None else None
            text_features = self.encode_text(text, normalize=True) if text is not None else None
        scale = self.logit_scale.detach().exp()
        if self.output_dict:
            return {"image_features": image_features, "text_features": text_features, "logit_scale": scale}
        return image_features, text_features, scale

    def invalidate_packed(self) -> None:
        """Forget the bf16 / fp8 packed copies, folded LayerNorm weights and tower handles (see ``_Packed``): required after
        in-place parameter writes through ``.data``, which leave ``_version`` unchanged."""
        invalidate_packed()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        invalidate_packed()          # copy_ under no_grad bumps _version today; do not depend on it
        return out

    def lock_image_tower(self, *a, **k):
        raise NotImplementedError("not provided: the training path (openvision_amd.training) differentiates every parameter")

    def set_grad_checkpointing(self, enable=True):
        if enable:
            raise NotImplementedError("not needed: openvision_amd.training keeps per-layer activations and recomputes the rest itself")


def logits(a: torch.Tensor, b: torch.Tensor, scale=1.0) -> torch.Tensor:
    """scale * a @ b.T for fp32 embeddings on device (exact-fp32 MFMA kernel).  `scale`: a float, or a 0-d / 1-element tensor that is
    read on the device (no host synchronisation)."""
    _require_cuda(a, "logits")
    sdev = None
    if isinstance(scale, torch.Tensor):
        sdev = scale.detach().float().reshape(1).to(a.device)
        scale = 1.0
    a32, b32 = a.detach().float().contiguous(), b.detach().float().contiguous()
    out = torch.empty(a32.shape[0], b32.shape[0], dtype=torch.float32, device=a.device)
    check(_lib.load().ov_logits(ptr(a32), ptr(b32), ptr(out), out.shape[1], a32.shape[0], b32.shape[0], a32.shape[1],
                                float(scale), ptr(sdev), stream_ptr()), "ov_logits")
    return out


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """F.normalize(x, dim=-1) on device -> fp32."""
    _require_cuda(x, "l2_normalize")
    xin = x.detach()
    xin = xin.contiguous() if xin.dtype in (torch.float32, torch.bfloat16) else xin.float().contiguous()
    x2 = xin.view(-1, xin.shape[-1])
    out = torch.empty(x2.shape, dtype=torch.float32, device=x.device)
    check(_lib.load().ov_l2norm(ptr(x2), _dtype_flag(x2), x2.shape[1], ptr(out), x2.shape[1], x2.shape[0], x2.shape[1],
                                stream_ptr()), "ov_l2norm")
    return out.view(x.shape)


def create_model(model_cfg: dict, device=None, state_dict: Optional[dict] = None, **kw) -> CLIP:
    """``CLIP(**model_cfg)`` + optional strict ``load_state_dict`` (ov-zero-shot-test.py:52-56)."""
    m = CLIP(embed_dim=model_cfg["embed_dim"], vision_cfg=model_cfg["vision_cfg"], text_cfg=model_cfg["text_cfg"],
             **{k: v for k, v in model_cfg.items() if k not in ("embed_dim", "vision_cfg", "text_cfg")}, **kw)
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=True)
    if device is not None:
        m = m.to(device)
    return m.eval()
