"""Checkpoint ingestion: an HF-format OpenVision directory -> the MI355X CLIP.

What the reference does: read ``open_clip_config.json``, build ``CLIP(vision_cfg, text_cfg, **clip_args)``, ``torch.load``
``open_clip_pytorch_model.bin`` and ``load_state_dict`` strictly (``ov-zero-shot-test.py:37-56``); the converter defines the
file names (``src/convert_upload/transfer_jax2hf.py:71-72,637``: ``open_clip_pytorch_model.bin``, and the never-written
``open_clip_model.safetensors``).  Here both are accepted; ``.bin`` is read with ``weights_only=True`` (no code execution).
Reading an orbax checkpoint needs jax/orbax and is out of scope; the converter's JAX -> open_clip key map on a flat
name -> array dictionary is in ``openvision_amd.convert_jax``.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

import torch

from . import config as ovcfg
from .model import CLIP, create_model

BIN_NAME = "open_clip_pytorch_model.bin"
SAFETENSORS_NAMES = ("open_clip_model.safetensors", "model.safetensors")


def _unwrap(sd):
    """factory.py:134-147 (load_state_dict): a training checkpoint keeps the weights under 'state_dict'; keys saved from a
    DistributedDataParallel wrapper carry a 'module.' prefix (decided, as the reference does, by the FIRST key)."""
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    if sd and next(iter(sd)).startswith("module"):
        sd = {k[7:]: v for k, v in sd.items()}
    return sd


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    for n in SAFETENSORS_NAMES:
        f = os.path.join(path, n)
        if os.path.exists(f):
            from safetensors.torch import load_file
            return _unwrap(load_file(f, device="cpu"))
    f = os.path.join(path, BIN_NAME)
    if os.path.exists(f):
        return _unwrap(torch.load(f, map_location="cpu", weights_only=True))
    raise FileNotFoundError(f"no {BIN_NAME} / {SAFETENSORS_NAMES[0]} under {path}")


def from_pretrained(path: str, device: Optional[str] = "cuda:0") -> Tuple[CLIP, dict]:
    """Returns (model.eval() on ``device``, preprocess_cfg).  Strict key/shape match, as the reference's loader."""
    model_cfg, pp = ovcfg.load_config_dir(path)
    sd = read_state_dict(path)
    core = {k: v for k, v in model_cfg.items() if k in ("embed_dim", "vision_cfg", "text_cfg")}
    extra = {k: v for k, v in model_cfg.items() if k not in core and k not in ("quick_gelu",) or (k == "quick_gelu" and v)}
    extra = {k: v for k, v in extra.items() if k in ("quick_gelu", "init_logit_scale", "init_logit_bias", "output_dict")}
    model = create_model({**core, **extra}, device=device, state_dict=sd)
    return model, pp


def save_pretrained(model: CLIP, model_cfg: dict, path: str, preprocess_cfg: Optional[dict] = None, safetensors: bool = True,
                    torch_bin: bool = True) -> None:
    """Write an HF-format directory (config JSON + weights) that ``from_pretrained`` and the reference's loader accept."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "open_clip_config.json"), "w") as f:
        json.dump({"model_cfg": model_cfg, "preprocess_cfg": preprocess_cfg or dict(ovcfg.DEFAULT_PREPROCESS)}, f, indent=2)
    sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    if torch_bin:
        torch.save(sd, os.path.join(path, BIN_NAME))
    if safetensors:
        from safetensors.torch import save_file
        save_file(sd, os.path.join(path, SAFETENSORS_NAMES[0]))
