"""CPU: the C-ABI library loads and exports every symbol include/ovhip.h declares (no compute calls)."""
import ctypes
import os
import re

from openvision_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ovhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ov_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ovhip.h but not exported by libovhip.so"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in openvision_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms


def test_library_loads_and_reports_missing_device_cleanly():
    lib = _lib.load()
    assert lib.ov_abi_version() == 2
    assert lib.ov_error_string(0) == b"ok"
    assert lib.ov_error_string(-2).startswith(b"shape")
    import torch
    if not torch.cuda.is_available():
        rc = lib.ov_device_check()
        assert rc < 0 and lib.ov_error_string(rc)          # an error code, not a crash


def test_argument_validation_without_gpu():
    lib = _lib.load()
    # null pointers / bad sizes are rejected before any HIP call
    assert lib.ov_gemm(None, 64, None, 64, None, None, 64, 1, 8, 64, 0, None, 0, 0, 0, 0, None) == -1
    assert lib.ov_layernorm(None, 1, 8, None, None, None, 1, 8, 1, 8, 1e-6, None) == -1
    assert lib.ov_clip_loss_workspace_bytes(256, 2048) > 0
    assert lib.ov_tower_create(None) is None
    cfg = _lib.TowerCfg(192, 2, 3, 768, 768, 0, 1e-6)
    t = lib.ov_tower_create(ctypes.byref(cfg))
    assert t
    assert lib.ov_tower_workspace_bytes(t, 4, 101) >= 4 * 101 * (192 + 768) * 2
    assert lib.ov_tower_forward(t, None, 4, 101, None, 0, None) == -1
    lib.ov_tower_destroy(t)
    bad = _lib.TowerCfg(100, 2, 3, 400, 400, 0, 1e-6)        # width not a multiple of 64
    assert lib.ov_tower_create(ctypes.byref(bad)) is None


def test_argument_validation_of_the_widening_entry_points():
    """Entry points added for the §8f rows and the fp8 path: invalid arguments are status codes, never HIP calls or crashes."""
    lib = _lib.load()
    assert lib.ov_gemm_fp8(None, 1024, None, 1024, None, None, None, None, 1024, 256, 256, 1024, 0, None, 0, None) == -1
    assert lib.ov_gemm_fp8_static(None, 1024, None, 1024, None, None, None, None, None, 1024, None, None, 256, 256, 1024, 1, None, 0, None) == -1
    assert lib.ov_amax_roll(None, None, 4, None) == -1
    # loss backward: null pointers are rejected; the workspace holds one partial per 32-row tile and direction
    assert lib.ov_clip_loss_backward(None, None, None, None, 16, 16, 192, None, 0, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.ov_clip_loss_backward_workspace_bytes(256, 2048) >= 2 * 8 * 4
    # operator-level backward: nulls and unsupported shapes are status codes
    assert lib.ov_transpose_bf16(None, 8, 8, 8, None, 64, None) == -1
    assert lib.ov_linear_backward(None, 64, None, 64, None, 64, 8, 64, 64, None, 64, None, 64, None, None, 0, None) == -1
    assert lib.ov_layernorm_backward(None, 8, None, None, 8, None, 0, None, 8, None, None, 1, 8, 1e-6, None, 0, None) == -1
    assert lib.ov_gelu_backward(None, 8, None, 8, None, 8, None, 0, 1, 8, 0, None) == -1
    assert lib.ov_block_backward(None, None, None, None, None, None, None, 1, 8, None, 0, None) == -1
    assert lib.ov_tower_saved_bytes(None, 1, 8) == 0
    # parameter update: nulls / bad hyper-parameters are status codes
    assert lib.ov_adamw_step(None, None, None, None, 16, 1e-3, 0.9, 0.95, 1e-8, 0.0, 1, 1.0, None, 0.0, None) == -1
    assert lib.ov_sumsq(None, 16, None, 0, None, 0, None) == -1 and lib.ov_sumsq_workspace_bytes() >= 1024
    assert lib.ov_block_backward_workspace_bytes(None, 1, 8) == 0
    assert lib.ov_attention_backward(None, 192, None, 64, None, 64, None, 192, 1, 8, 1, 64, 0.125, None, 0, None) == -1
    assert lib.ov_attention_backward_workspace_bytes(2, 257, 16, 64) == 0 and lib.ov_attention_backward_workspace_bytes(2, 2305, 6, 64) >= 2 * 2 * 6 * 2305 * 4
    assert lib.ov_attention_backward_workspace_bytes(2, 257, 16, 80) > 0
    assert lib.ov_gemm_batched(None, 64, 64, None, 64, 64, None, 64, 64, 64, 64, 64, 2, None) == -1
    assert lib.ov_linear_backward_workspace_bytes(65792, 4096, 1024) >= 2 * 65792 * (4096 + 1024)
    assert lib.ov_layernorm_backward_workspace_bytes(65792, 1024) > 0
    assert lib.ov_quant_rows_fp8(None, 1024, None, 1024, None, 4, 1024, None, None) == -1
    assert lib.ov_layernorm_quant_fp8(None, 1024, None, None, None, 1024, None, 4, 1024, 1e-6, None) == -1
    assert lib.ov_attention_fp8out(None, 192, None, 64, 1, 32, 1, 64, 0.125, None, None, None) == -1
    assert lib.ov_topk(None, 8, 1, 8, 1, 1, None, None, None) == -1
    assert lib.ov_class_mean_normalize(None, None, 1, 1, 8, None) == -1
    assert lib.ov_preprocess_image(None, 8, 8, None, None, 1, 8, None, None, 1, 8, 0, 8, None, 0, 0, 8, 8, None, None, None, 0, None) == -1
    # fp8 copies need K-tiles of 128 elements, at least three of them: a 192-wide tower is refused, a 384-wide one accepted
    small = _lib.TowerCfg(192, 1, 3, 768, 768, 0, 1e-6)
    t = lib.ov_tower_create(ctypes.byref(small))
    dummy = (ctypes.c_char * 64)()
    addr = (ctypes.addressof(dummy) + 15) & ~15
    q = _lib.BlockFp8(*([addr] * 10))
    assert lib.ov_tower_set_block_fp8(t, 0, ctypes.byref(q)) == -2
    assert lib.ov_tower_set_fp8_hidden_scale(t, None, 1) == -1
    lib.ov_tower_destroy(t)
    ok = _lib.TowerCfg(384, 1, 6, 1536, 1536, 0, 1e-6)
    t = lib.ov_tower_create(ctypes.byref(ok))
    base = lib.ov_tower_workspace_bytes(t, 2, 64)
    assert lib.ov_tower_set_block_fp8(t, 0, ctypes.byref(q)) == 0
    assert lib.ov_tower_workspace_bytes(t, 2, 64) > base            # room for the e4m3 activations and their scales
    assert lib.ov_tower_set_block_fp8(t, 0, None) == 0              # cleared: back to the bf16 footprint
    assert lib.ov_tower_workspace_bytes(t, 2, 64) == base
    lib.ov_tower_destroy(t)
