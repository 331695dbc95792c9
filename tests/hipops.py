"""Thin test-side wrappers: torch tensors -> C ABI calls of libovhip.so (granular operators)."""
import ctypes as C

import torch

from openvision_amd import _lib
from openvision_amd._lib import ptr, stream_ptr, check, OV_BF16, OV_F32


def _flag(t):
    return OV_F32 if t.dtype == torch.float32 else OV_BF16


def layernorm(x, g, b, eps=1e-6, out_dtype=None):
    lib = _lib.load()
    x2 = x.contiguous().view(-1, x.shape[-1])
    y = torch.empty(x2.shape, dtype=out_dtype or x.dtype, device=x.device)
    check(lib.ov_layernorm(ptr(x2), _flag(x2), x2.shape[1], ptr(g), ptr(b), ptr(y), _flag(y), x2.shape[1], x2.shape[0],
                           x2.shape[1], eps, stream_ptr()))
    return y.view(x.shape)


def gemm(a, w, bias=None, epi=0, resid=None, out=None, out_group=0, resid_mod=0, resid_off=0, out_rows=None):
    lib = _lib.load()
    m, k = a.shape
    n = w.shape[0]
    if out is None:
        out = torch.zeros(out_rows or m, n, dtype=torch.bfloat16, device=a.device)
    check(lib.ov_gemm(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), m, n, k, epi,
                      ptr(resid), resid.stride(0) if resid is not None else 0, out_group, resid_mod, resid_off, stream_ptr()))
    return out


def gemm_keep(a, w, bias, epi, ldc=None, ldc2=None):
    """C = gelu(a w^T + bias), C2 = a w^T + bias (both bf16) from one launch: ov_gemm_keep."""
    lib = _lib.load()
    m, k = a.shape
    n = w.shape[0]
    out = torch.zeros(m, ldc or n, dtype=torch.bfloat16, device=a.device)
    pre = torch.zeros(m, ldc2 or n, dtype=torch.bfloat16, device=a.device)
    check(lib.ov_gemm_keep(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), ptr(pre), pre.stride(0),
                           m, n, k, epi, stream_ptr()))
    return out, pre


def attention(qkv, B, L, H, hd=64):
    lib = _lib.load()
    out = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device=qkv.device)
    check(lib.ov_attention(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), B, L, H, hd, hd ** -0.5, stream_ptr()))
    return out


def clip_loss(img, txt, all_img, all_txt, scale, label_offset):
    lib = _lib.load()
    b, e = img.shape
    n = all_img.shape[0]
    nb = lib.ov_clip_loss_workspace_bytes(b, n)
    ws = torch.empty(nb + 16, dtype=torch.uint8, device=img.device)
    out = torch.empty(1, dtype=torch.float32, device=img.device)
    terms = torch.empty(4, b, dtype=torch.float32, device=img.device)
    sc = (scale.detach().float().reshape(1) if isinstance(scale, torch.Tensor) and scale.is_cuda
          else torch.full((1,), float(scale), dtype=torch.float32, device=img.device))
    check(lib.ov_clip_loss(ptr(img), ptr(txt), ptr(all_img), ptr(all_txt), b, n, e, ptr(sc), label_offset, ptr(out),
                           ptr(terms), ptr(ws), nb, stream_ptr()))
    return out[0], terms


def rowstats(x, eps=1e-6):
    lib = _lib.load()
    st = torch.empty(x.shape[0], 2, dtype=torch.float32, device=x.device)
    check(lib.ov_rowstats(ptr(x), x.stride(0), ptr(st), x.shape[0], x.shape[1], eps, stream_ptr()))
    return st


def rowparts(x):
    """Partial sums of the row statistics: [rows, D / 32, 2] fp32 (ov_rowparts)."""
    lib = _lib.load()
    parts = torch.empty(x.shape[0], x.shape[1] // 32, 2, dtype=torch.float32, device=x.device)
    check(lib.ov_rowparts(ptr(x), x.stride(0), ptr(parts), x.shape[0], x.shape[1], stream_ptr()))
    return parts


def rowstats_finalize(parts, eps=1e-6):
    lib = _lib.load()
    st = torch.empty(parts.shape[0], 2, dtype=torch.float32, device=parts.device)
    check(lib.ov_rowstats_finalize(ptr(parts), ptr(st), parts.shape[0], parts.shape[1] * 32, eps, stream_ptr()))
    return st


def gemm_rowparts(a, w, bias, resid, out=None):
    """Residual GEMM that also leaves the partial sums of its output rows (ov_gemm_rowparts)."""
    lib = _lib.load()
    m, k = a.shape
    n = w.shape[0]
    if out is None:
        out = torch.zeros(m, n, dtype=torch.bfloat16, device=a.device)
    parts = torch.full((m, n // 32, 2), float("nan"), dtype=torch.float32, device=a.device)
    check(lib.ov_gemm_rowparts(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), m, n, k,
                               ptr(resid), resid.stride(0), ptr(parts), stream_ptr()))
    return out, parts


def gemm_ln(x, wg, cvec, colsum, stats, epi=0, out=None):
    lib = _lib.load()
    m, k = x.shape
    n = wg.shape[0]
    if out is None:
        out = torch.empty(m, n, dtype=torch.bfloat16, device=x.device)
    check(lib.ov_gemm_ln(ptr(x), x.stride(0), ptr(wg), wg.stride(0), ptr(cvec), ptr(colsum), ptr(stats), ptr(out), out.stride(0),
                         m, n, k, epi, stream_ptr()))
    return out


def quantize_rows_e4m3(x):
    """Per-row absmax scaling to OCP e4m3fn (max 448): returns (uint8 view of the fp8 tensor, fp32 scales)."""
    amax = x.float().abs().amax(dim=1).clamp_min(1e-12)
    scale = amax / 448.0
    q = (x.float() / scale[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale.contiguous()


def gemm_fp8(aq, wq, rowscale, colscale, bias=None, epi=0, resid=None):
    lib = _lib.load()
    m, k = aq.shape
    n = wq.shape[0]
    out = torch.empty(m, n, dtype=torch.bfloat16, device=aq.device)
    check(lib.ov_gemm_fp8(ptr(aq), aq.stride(0), ptr(wq), wq.stride(0), ptr(rowscale), ptr(colscale),
                          ptr(bias) if bias is not None else None, ptr(out), out.stride(0), m, n, k, epi,
                          ptr(resid) if resid is not None else None, resid.stride(0) if resid is not None else 0, stream_ptr()))
    return out


def quant_rows_fp8(x, gamma=None, beta=None, eps=1e-6, amax=None):
    """bf16 [rows, D] -> (uint8 e4m3 [rows, D], fp32 scales); with gamma/beta: LayerNorm first."""
    lib = _lib.load()
    rows, d = x.shape
    q = torch.empty(rows, d, dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    if gamma is None:
        check(lib.ov_quant_rows_fp8(ptr(x), x.stride(0), ptr(q), q.stride(0), ptr(sc), rows, d,
                                    ptr(amax) if amax is not None else None, stream_ptr()))
    else:
        check(lib.ov_layernorm_quant_fp8(ptr(x), x.stride(0), ptr(gamma), ptr(beta), ptr(q), q.stride(0), ptr(sc), rows, d, eps,
                                         stream_ptr()))
    return q, sc


def gemm_fp8_static(aq, wq, colscale, bias, epi, rowscale=None, in_amax=None, out_amax=None, resid=None, amax_next=None):
    """out_amax: returns e4m3 bytes [m, n]; in_amax: returns bf16."""
    lib = _lib.load()
    m, k = aq.shape
    n = wq.shape[0]
    out = torch.empty(m, n, dtype=torch.uint8 if out_amax is not None else torch.bfloat16, device=aq.device)
    check(lib.ov_gemm_fp8_static(ptr(aq), aq.stride(0), ptr(wq), wq.stride(0), ptr(rowscale) if rowscale is not None else None,
                                 ptr(in_amax) if in_amax is not None else None, ptr(colscale), ptr(bias) if bias is not None else None,
                                 ptr(out), out.stride(0), ptr(out_amax) if out_amax is not None else None,
                                 ptr(amax_next) if amax_next is not None else None, m, n, k, epi,
                                 ptr(resid) if resid is not None else None, resid.stride(0) if resid is not None else 0, stream_ptr()))
    return out


def attention_fp8out(qkv, B, L, Hh, amax, amax_next=None):
    """head_dim 64; returns e4m3 bytes [B*L, Hh*64] under the static scale 2 * amax / 448."""
    lib = _lib.load()
    d = Hh * 64
    out = torch.empty(B * L, d, dtype=torch.uint8, device=qkv.device)
    check(lib.ov_attention_fp8out(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), B, L, Hh, 64, 0.125, ptr(amax),
                                  ptr(amax_next) if amax_next is not None else None, stream_ptr()))
    return out


def clip_loss_backward(img, txt, all_img, all_txt, scale, label_offset, terms, grad=1.0, gathered=True):
    """ov_clip_loss_backward: returns (d_img, d_txt, d_all_img | None, d_all_txt | None, d_scale)."""
    lib = _lib.load()
    b, e = img.shape
    n = all_img.shape[0]
    d_img, d_txt = torch.empty_like(img), torch.empty_like(txt)
    d_ai = torch.empty_like(all_img) if gathered else None
    d_at = torch.empty_like(all_txt) if gathered else None
    d_s = torch.empty(1, dtype=torch.float32, device=img.device)
    nb = lib.ov_clip_loss_backward_workspace_bytes(b, n)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=img.device)
    sc = torch.full((1,), float(scale), dtype=torch.float32, device=img.device)
    gr = torch.full((1,), float(grad), dtype=torch.float32, device=img.device)
    check(lib.ov_clip_loss_backward(ptr(img), ptr(txt), ptr(all_img), ptr(all_txt), b, n, e, ptr(sc), label_offset, ptr(terms),
                                    ptr(gr), ptr(d_img), ptr(d_txt), ptr(d_ai) if gathered else None,
                                    ptr(d_at) if gathered else None, ptr(d_s), ptr(ws), nb, stream_ptr()), "ov_clip_loss_backward")
    return d_img, d_txt, d_ai, d_at, d_s


def linear_backward(dy, x, w, want=("dx", "dw", "db")):
    """ov_linear_backward for y = x w^T + b: returns (dX bf16 [M,K] | None, dW bf16 [N,K] | None, db fp32 [N] | None)."""
    lib = _lib.load()
    M, N = dy.shape
    K = x.shape[1]
    dx = torch.empty(M, K, dtype=torch.bfloat16, device=dy.device) if "dx" in want else None
    dw = torch.empty(N, K, dtype=torch.bfloat16, device=dy.device) if "dw" in want else None
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if "db" in want else None
    nb = lib.ov_linear_backward_workspace_bytes(M, N, K)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=dy.device)
    check(lib.ov_linear_backward(ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(w), w.stride(0), M, N, K,
                                 ptr(dx) if dx is not None else None, K, ptr(dw) if dw is not None else None, K,
                                 ptr(db) if db is not None else None, ptr(ws), nb, stream_ptr()), "ov_linear_backward")
    return dx, dw, db


def transpose(x):
    lib = _lib.load()
    R, C = x.shape
    rp = (R + 63) // 64 * 64
    out = torch.full((C, rp), 7.0, dtype=torch.bfloat16, device=x.device)
    check(lib.ov_transpose_bf16(ptr(x), x.stride(0), R, C, ptr(out), rp, stream_ptr()), "ov_transpose_bf16")
    return out


def layernorm_backward(x, gamma, dy, eps=1e-6, dres=None):
    lib = _lib.load()
    rows, D = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(D, dtype=torch.float32, device=x.device)
    db = torch.empty(D, dtype=torch.float32, device=x.device)
    nb = lib.ov_layernorm_backward_workspace_bytes(rows, D)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=x.device)
    check(lib.ov_layernorm_backward(ptr(x), x.stride(0), ptr(gamma), ptr(dy), dy.stride(0), ptr(dres) if dres is not None else None,
                                    dres.stride(0) if dres is not None else 0, ptr(dx), dx.stride(0), ptr(dg), ptr(db),
                                    rows, D, eps, ptr(ws), nb, stream_ptr()), "ov_layernorm_backward")
    return dx, dg, db


def gelu_backward(a, dh, tanh, with_h=False):
    lib = _lib.load()
    rows, N = a.shape
    da = torch.empty_like(a)
    h = torch.empty_like(a) if with_h else None
    check(lib.ov_gelu_backward(ptr(a), a.stride(0), ptr(dh), dh.stride(0), ptr(da), da.stride(0), ptr(h) if with_h else None,
                               h.stride(0) if with_h else 0, rows, N, int(tanh), stream_ptr()), "ov_gelu_backward")
    return (da, h) if with_h else da


def block_backward(cfg, weights, x, dy, B, L):
    """ov_block_backward.  weights: dict of the module's own tensors (ln1_w, ln1_b fp32; qkv_w bf16 [3D, D]; qkv_b fp32; ...).
    Returns (dx bf16, grads dict)."""
    import ctypes as C
    lib = _lib.load()
    names = ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc_w", "fc_b", "proj_w", "proj_b")
    wst = _lib.BlockWeights(*[C.c_void_p(weights[n].data_ptr()) for n in names], None, None)
    grads = {n: torch.empty_like(weights[n]) for n in names}
    gst = _lib.BlockGrads(*[C.c_void_p(grads[n].data_ptr()) for n in names])
    dx = torch.empty_like(x)
    nb = lib.ov_block_backward_workspace_bytes(C.byref(cfg), B, L)
    assert nb > 0
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=x.device)
    check(lib.ov_block_backward(C.byref(cfg), C.byref(wst), ptr(x), None, ptr(dy), ptr(dx), C.byref(gst), B, L, ptr(ws), nb, stream_ptr()),
          "ov_block_backward")
    return dx, grads


def tower1_forward_backward(cfg, weights, x, dy, B, L):
    """A one-layer tower through ov_tower_forward_saving + ov_tower_backward (every forward intermediate kept, fused backward
    epilogues): returns (y bf16, dx bf16, grads dict)."""
    import ctypes as C
    lib = _lib.load()
    names = ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc_w", "fc_b", "proj_w", "proj_b")
    handle = lib.ov_tower_create(C.byref(cfg))
    assert handle
    try:
        wst = _lib.BlockWeights(*[C.c_void_p(weights[n].data_ptr()) for n in names], None, None)
        check(lib.ov_tower_set_block(handle, 0, C.byref(wst)), "ov_tower_set_block")
        y = x.clone()
        saved = torch.empty(lib.ov_tower_saved_bytes(handle, B, L) + 256, dtype=torch.uint8, device=x.device)
        nb = lib.ov_tower_workspace_bytes(handle, B, L)
        ws = torch.empty(nb + 256, dtype=torch.uint8, device=x.device)
        check(lib.ov_tower_forward_saving(handle, ptr(y), ptr(saved), B, L, ptr(ws), nb, stream_ptr()), "ov_tower_forward_saving")
        grads = {n: torch.empty_like(weights[n]) for n in names}
        garr = (_lib.BlockGrads * 1)(_lib.BlockGrads(*[C.c_void_p(grads[n].data_ptr()) for n in names]))
        dx = dy.clone()
        nb2 = lib.ov_tower_backward_workspace_bytes(handle, B, L)
        ws2 = torch.empty(nb2 + 256, dtype=torch.uint8, device=x.device)
        check(lib.ov_tower_backward(handle, ptr(saved), ptr(dx), garr, B, L, ptr(ws2), nb2, stream_ptr()), "ov_tower_backward")
        torch.cuda.synchronize()
    finally:
        lib.ov_tower_destroy(handle)
    return y, dx, grads


def attention_backward(qkv, out, dout, B, L, H, hd=64):
    lib = _lib.load()
    dqkv = torch.empty_like(qkv)
    nb = lib.ov_attention_backward_workspace_bytes(B, L, H, hd)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=qkv.device)
    check(lib.ov_attention_backward(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(dout), dout.stride(0), ptr(dqkv), dqkv.stride(0),
                                    B, L, H, hd, hd ** -0.5, ptr(ws), nb, stream_ptr()), "ov_attention_backward")
    return dqkv


def attention_lse(qkv, B, L, H, hd=64):
    """ov_attention_lse: (out, lse [B*H, L rounded up to 32] fp32 in log2 units)."""
    lib = _lib.load()
    out = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.full((B * H, (L + 31) // 32 * 32), float("nan"), dtype=torch.float32, device=qkv.device)
    check(lib.ov_attention_lse(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(lse), B, L, H, hd, hd ** -0.5, stream_ptr()),
          "ov_attention_lse")
    return out, lse


def attention_backward_saved(qkv, out, dout, lse, B, L, H, hd=64):
    lib = _lib.load()
    dqkv = torch.empty_like(qkv)
    nb = lib.ov_attention_backward_workspace_bytes(B, L, H, hd)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=qkv.device)
    check(lib.ov_attention_backward_saved(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(dout), dout.stride(0), ptr(dqkv),
                                          dqkv.stride(0), ptr(lse), B, L, H, hd, hd ** -0.5, ptr(ws), nb, stream_ptr()),
          "ov_attention_backward_saved")
    return dqkv


def gemm_tn_batched(p, q, chunk, sums=False):
    """ov_gemm_tn_batched: partials [batch, NI, NJ] bf16 of P^T Q over row ranges of `chunk` contraction rows
    (sums=True: and the fp32 [batch, NI] column sums of P over the same ranges)."""
    lib = _lib.load()
    mc, ni = p.shape
    nj = q.shape[1]
    batch = (mc + chunk - 1) // chunk
    out = torch.empty(batch, ni, nj, dtype=torch.bfloat16, device=p.device)
    ps = torch.full((batch, ni), float("nan"), dtype=torch.float32, device=p.device) if sums else None
    check(lib.ov_gemm_tn_batched(ptr(p), p.stride(0), ptr(q), q.stride(0), ptr(out), nj, ni * nj, mc, ni, nj, chunk, batch, ptr(ps),
                                 stream_ptr()), "ov_gemm_tn_batched")
    return (out, ps) if sums else out
