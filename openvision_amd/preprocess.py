"""Device-side image front-end: the reference's per-image CPU transform, run by HIP kernels.

Reference: ``transforms.Resize((S, S)) -> convert("RGB") -> ToTensor -> Normalize`` (``ov-zero-shot-test.py:72-77``) and
``open_clip.transform.image_transform`` eval branch (``src/convert_upload/open_clip/transform.py:355-392``: 'squash' =
``Resize((S, S))``; 'shortest' = ``Resize(S)`` on the short edge + ``CenterCrop(S)``).  torchvision (absent here) runs
``PIL.Image.resize`` for PIL inputs; Pillow (12.2.0 in this image) implements it in ``src/libImaging/Resample.c`` as a two-pass
fixed-point convolution.  ``resize_plan`` below restates its ``precompute_coeffs`` / ``normalize_coeffs_8bpc`` in the same double
arithmetic, the kernels (``csrc/preprocess.hip``) do the integer convolution, so the uint8 image equals Pillow's bit for bit.

The tokenizer half of that row is ``openvision_amd.tokenizer``.
"""
from __future__ import annotations

import ctypes
import math
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ptr, stream_ptr, check
from .config import DEFAULT_PREPROCESS

PRECISION_BITS = 32 - 8 - 2            # Resample.c


def _bilinear(x: float) -> float:      # Resample.c bilinear_filter, support 1.0
    if x < 0.0:
        x = -x
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x: float) -> float:       # Resample.c bicubic_filter (a = -0.5), support 2.0
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


@lru_cache(maxsize=64)
def resize_plan(in_size: int, out_size: int, interpolation: str = "bilinear") -> Tuple[np.ndarray, np.ndarray, int]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size): bounds int32 [out, 2] = (first, count),
    taps int32 [out, ksize] (22-bit fixed point)."""
    filt, fsupport = FILTERS[interpolation]
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def output_geometry(h: int, w: int, size: int, resize_mode: str) -> Tuple[int, int, int, int]:
    """(Hr, Wr, crop_y, crop_x): 'squash' = Resize((S, S)); 'shortest' = Resize(S) (torchvision: long edge int(S * long / short))
    + CenterCrop(S) (torchvision: int(round((H - S) / 2.0)))."""
    if resize_mode == "squash":
        return size, size, 0, 0
    if resize_mode != "shortest":
        raise NotImplementedError(f"resize_mode {resize_mode!r} (only 'squash' and 'shortest')")
    if w <= h:
        wr, hr = size, int(size * h / w)
    else:
        hr, wr = size, int(size * w / h)
    return hr, wr, int(round((hr - size) / 2.0)), int(round((wr - size) / 2.0))


_plan_cache = {}


def _device_plan(in_size, out_size, interpolation, device):
    key = (in_size, out_size, interpolation, str(device))
    if key not in _plan_cache:
        b, k, ks = resize_plan(in_size, out_size, interpolation)
        _plan_cache[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks, b)
    return _plan_cache[key]


def preprocess(images: Sequence[torch.Tensor], size: int, mean=None, std=None, resize_mode: str = "squash",
               interpolation: str = "bilinear", dtype: torch.dtype = torch.float32, device="cuda:0") -> torch.Tensor:
    """uint8 RGB images [H, W, 3] (any sizes; torch tensors or numpy arrays) -> normalised [B, 3, size, size] on the device."""
    mean = list(mean if mean is not None else DEFAULT_PREPROCESS["mean"])
    std = list(std if std is not None else DEFAULT_PREPROCESS["std"])
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.OvhipError("preprocess: needs an MI355X device (no CPU fallback)")
    out = torch.empty(len(images), 3, size, size, dtype=dtype, device=dev)
    cm = (ctypes.c_float * 3)(*mean)
    cs = (ctypes.c_float * 3)(*std)
    code = {torch.float32: 0, torch.bfloat16: 1}[dtype]
    for i, im in enumerate(images):
        im = torch.as_tensor(np.ascontiguousarray(im) if isinstance(im, np.ndarray) else im)
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
            raise ValueError("preprocess: expected uint8 [H, W, 3] RGB images")
        im = im.to(dev).contiguous()
        h, w = int(im.shape[0]), int(im.shape[1])
        hr, wr, cy, cx = output_geometry(h, w, size, resize_mode)
        bx, kx, ksx, _ = _device_plan(w, wr, interpolation, dev)
        by, ky, ksy, by_host = _device_plan(h, hr, interpolation, dev)
        row0 = int(by_host[cy, 0])                                  # Pillow's ybox, restricted to the cropped rows
        row1 = int(by_host[cy + size - 1, 0] + by_host[cy + size - 1, 1])
        tmp = torch.empty(row1 - row0, wr, 3, dtype=torch.uint8, device=dev)
        check(lib.ov_preprocess_image(ptr(im), h, w, ptr(bx), ptr(kx), ksx, wr, ptr(by), ptr(ky), ksy, hr, row0, row1 - row0,
                                      ptr(tmp), cx, cy, size, size, cm, cs, ptr(out[i]), code, stream_ptr()), "ov_preprocess_image")
    return out
