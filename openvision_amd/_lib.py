"""ctypes binding of libovhip.so (the C ABI declared in include/ovhip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.  The Python
layer above never computes on its own — torch is used for device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libovhip.so")
if os.environ.get("OVHIP_LIB"):          # diagnostics only: A/B builds with compile-time switches (tools/build_variant.py), same ABI
    LIB_PATH = os.path.join(_PKG, os.environ["OVHIP_LIB"]) if os.sep not in os.environ["OVHIP_LIB"] else os.environ["OVHIP_LIB"]

OV_F32, OV_BF16 = 0, 1
EPI_BIAS, EPI_GELU_ERF, EPI_GELU_TANH, EPI_RESIDUAL = 0, 1, 2, 3

c_void_p, c_int, c_int64, c_float, c_size_t = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class TowerCfg(C.Structure):
    _fields_ = [("width", c_int), ("layers", c_int), ("heads", c_int), ("mlp", c_int), ("mlp_pad", c_int),
                ("gelu_tanh", c_int), ("ln_eps", c_float)]


class BlockWeights(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b",
                                        "fc_w", "fc_b", "proj_w", "proj_b", "qkv_colsum", "fc_colsum")]


class BlockGrads(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc_w", "fc_b",
                                        "proj_w", "proj_b")]


class BlockSaved(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("qkv", "attn_out", "x1", "fc_pre", "ln1_out", "ln2_out", "fc_act", "attn_lse")]


class BlockFp8(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("qkv_w8", "qkv_s", "qkv_b", "out_w8", "out_s", "fc_w8", "fc_s", "fc_b", "proj_w8", "proj_s")]


class VisionHead(C.Structure):
    _fields_ = [("image_size", c_int), ("patch_size", c_int), ("kpad", c_int), ("pool_avg", c_int),
                ("final_ln_after_pool", c_int), ("embed_dim", c_int), ("embed_pad", c_int),
                ("conv_w", c_void_p), ("cls", c_void_p), ("pos", c_void_p), ("pos_f32", c_void_p),
                ("ln_post_w", c_void_p), ("ln_post_b", c_void_p), ("proj_t", c_void_p)]


class TextHead(C.Structure):
    _fields_ = [("context_length", c_int), ("vocab_size", c_int), ("pool_last", c_int), ("embed_dim", c_int),
                ("token_embedding", c_void_p), ("pos", c_void_p), ("ln_final_w", c_void_p), ("ln_final_b", c_void_p),
                ("proj_t", c_void_p)]


# name -> (restype, argtypes); must list EVERY symbol include/ovhip.h declares (tests/test_cabi.py checks)
SIGNATURES = {
    "ov_abi_version": (c_int, []),
    "ov_error_string": (C.c_char_p, [c_int]),
    "ov_device_check": (c_int, []),
    "ov_layernorm": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int,
                             c_float, c_void_p]),
    "ov_gemm": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                        c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "ov_gemm_fp8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                            c_int, c_void_p, c_int64, c_void_p]),
    "ov_quant_rows_fp8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "ov_attention_fp8out": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "ov_amax_roll": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "ov_gemm_fp8_static": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                   c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "ov_tower_set_fp8_hidden_scale": (c_int, [c_void_p, c_void_p, c_int]),
    "ov_tower_set_fp8_mask": (c_int, [c_void_p, c_void_p, c_int]),
    "ov_layernorm_quant_fp8": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_float,
                                       c_void_p]),
    "ov_gemm_ln": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                           c_int, c_int, c_void_p]),
    "ov_gemm_keep": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                             c_int, c_void_p]),
    "ov_rowstats": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "ov_rowparts": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "ov_rowstats_finalize": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "ov_gemm_rowparts": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64,
                                 c_void_p, c_void_p]),
    "ov_attention": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ov_im2col_patches": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ov_cls_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ov_mean_pool": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ov_text_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                              c_void_p]),
    "ov_gather_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "ov_convert": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p]),
    "ov_l2norm": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "ov_logits": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ov_preprocess_image": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ov_class_mean_normalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ov_topk": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ov_clip_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "ov_clip_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                             c_void_p, c_void_p, c_size_t, c_void_p]),
    "ov_clip_loss_backward_workspace_bytes": (c_size_t, [c_int, c_int]),
    "ov_gemm_batched": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int,
                                c_int, c_void_p]),
    "ov_gemm_tn_batched": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int64, c_int,
                                   c_void_p, c_void_p]),
    "ov_attention_lse": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ov_attention_backward_saved": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int,
                                            c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "ov_attention_backward_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ov_attention_backward": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                      c_float, c_void_p, c_size_t, c_void_p]),
    "ov_transpose_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "ov_linear_backward_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "ov_linear_backward": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64,
                                   c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ov_layernorm_backward_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "ov_layernorm_backward": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_int64, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "ov_gelu_backward": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                 c_void_p]),
    "ov_tower_forward_saving": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ov_tower_backward_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "ov_tower_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ov_block_backward_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "ov_block_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t,
                                  c_void_p]),
    "ov_tower_saved_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "ov_clip_loss_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ov_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float,
                              c_void_p, c_float, c_void_p]),
    "ov_sumsq_workspace_bytes": (c_size_t, []),
    "ov_sumsq": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "ov_profile_enable": (c_int, [C.c_uint, c_int]),
    "ov_profile_read": (c_int, [c_int, C.POINTER(C.c_double), C.POINTER(c_int), C.POINTER(C.c_double)]),
    "ov_debug_gemm_stamps": (c_int, [c_void_p, c_int]),
    "ov_debug_gemm_wave_stamps": (c_int, [c_void_p]),
    "ov_tower_create": (c_void_p, [C.POINTER(TowerCfg)]),
    "ov_tower_destroy": (None, [c_void_p]),
    "ov_tower_set_block": (c_int, [c_void_p, c_int, C.POINTER(BlockWeights)]),
    "ov_tower_set_block_fp8": (c_int, [c_void_p, c_int, c_void_p]),
    "ov_tower_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "ov_tower_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ov_vision_workspace_bytes": (c_size_t, [c_void_p, C.POINTER(VisionHead), c_int]),
    "ov_vision_embed": (c_int, [c_void_p, C.POINTER(VisionHead), c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                c_void_p]),
    "ov_vision_head_forward": (c_int, [c_void_p, C.POINTER(VisionHead), c_void_p, c_int, c_void_p, c_int, c_void_p,
                                       c_size_t, c_void_p]),
    "ov_encode_image": (c_int, [c_void_p, C.POINTER(VisionHead), c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                c_size_t, c_void_p]),
    "ov_text_workspace_bytes": (c_size_t, [c_void_p, C.POINTER(TextHead), c_int]),
    "ov_encode_text": (c_int, [c_void_p, C.POINTER(TextHead), c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                               c_size_t, c_void_p]),
}

ABI_VERSION = 2      # include/ovhip.h OV_ABI_VERSION
_lib = None
_lock = threading.Lock()


class OvhipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen libovhip.so and bind every entry point.  Raises if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch bundles its own ROCm runtime (libamdhip64); it must be the first HIP runtime mapped into the
        # process so that libovhip.so binds to the SAME runtime instance (two runtimes = "No HIP GPUs").
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise OvhipError(f"{LIB_PATH} is missing: build it with `python -m openvision_amd.build` "
                             f"(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError -> ABI mismatch, fail loudly
            fn.restype, fn.argtypes = res, args
        if lib.ov_abi_version() != ABI_VERSION:
            raise OvhipError("libovhip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ov_error_string(rc)
        raise OvhipError(f"libovhip {what} failed: {msg.decode() if msg else rc} (status {rc})")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
