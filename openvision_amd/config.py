"""Model configuration: the ``open_clip_config.json`` schema and the OpenVision size table.

Mirrors the dataclasses the reference builds its towers from
(``src/convert_upload/open_clip/model.py:27-84`` ``CLIPVisionCfg`` / ``CLIPTextCfg``) and the
size tables of the JAX->HF converter (``src/convert_upload/transfer_jax2hf.py:76-92``).
Only the fields the OpenVision encode-and-contrast path reads are kept; anything that would
select a different architecture (attentional pooling, timm/HF towers, quick_gelu, layer-scale,
patch dropout > 0) is rejected loudly instead of being silently ignored.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field, asdict
from typing import Any, Dict, Optional, Tuple, Union


@dataclass
class CLIPVisionCfg:
    # reference: open_clip/model.py:27-55
    layers: int = 12
    width: int = 768
    head_width: int = 64
    mlp_ratio: float = 4.0
    patch_size: int = 16
    image_size: int = 224
    ls_init_value: Optional[float] = None
    patch_dropout: float = 0.0
    attentional_pool: bool = False
    no_ln_pre: bool = False
    pos_embed_type: str = "learnable"
    final_ln_after_pool: bool = False
    pool_type: str = "tok"
    output_tokens: bool = False
    act_kwargs: Optional[dict] = None
    norm_kwargs: Optional[dict] = None
    timm_model_name: Optional[str] = None


@dataclass
class CLIPTextCfg:
    # reference: open_clip/model.py:58-84
    context_length: int = 77
    vocab_size: int = 49408
    hf_tokenizer_name: Optional[str] = None
    tokenizer_kwargs: Optional[dict] = None
    width: int = 512
    heads: int = 8
    layers: int = 12
    mlp_ratio: float = 4.0
    ls_init_value: Optional[float] = None
    embed_cls: bool = False
    pad_id: int = 0
    no_causal_mask: bool = False
    final_ln_after_pool: bool = False
    pool_type: str = "argmax"
    proj_bias: bool = False
    output_tokens: bool = False
    act_kwargs: Optional[dict] = None
    norm_kwargs: Optional[dict] = None
    hf_model_name: Optional[str] = None


def _from_dict(cls, d: Union[dict, Any]):
    if isinstance(d, cls):
        return d
    known = {f for f in cls.__dataclass_fields__}
    extra = {k: v for k, v in d.items() if k not in known}
    # Unknown keys that only matter for other model families are tolerated when falsy.
    for k, v in extra.items():
        if v not in (None, False, 0, 0.0, "", [], {}):
            raise ValueError(f"{cls.__name__}: unsupported config key {k!r}={v!r} "
                             f"(outside the OpenVision encode-and-contrast path)")
    return cls(**{k: v for k, v in d.items() if k in known})


def vision_cfg_from(d) -> CLIPVisionCfg:
    c = _from_dict(CLIPVisionCfg, d)
    if c.timm_model_name or c.attentional_pool or c.ls_init_value is not None or c.patch_dropout:
        raise ValueError("vision_cfg selects a tower outside the OpenVision ViT path "
                         "(timm / attentional pool / layer-scale / patch-dropout)")
    if c.pool_type not in ("avg", "tok"):
        raise ValueError(f"vision pool_type {c.pool_type!r} not supported (OpenVision uses 'avg')")
    if c.width % c.head_width:
        raise ValueError("vision width must be a multiple of head_width")
    if c.image_size % c.patch_size:
        raise ValueError("image_size must be a multiple of patch_size")
    return c


def text_cfg_from(d) -> CLIPTextCfg:
    c = _from_dict(CLIPTextCfg, d)
    if c.hf_model_name or c.ls_init_value is not None or c.embed_cls or c.proj_bias:
        raise ValueError("text_cfg selects a tower outside the OpenVision text path")
    if not c.no_causal_mask:
        raise ValueError("OpenVision text towers are unmasked (no_causal_mask=true); "
                         "causal text attention is not on this path")
    if c.pool_type not in ("last", "first"):
        raise ValueError(f"text pool_type {c.pool_type!r} not supported (OpenVision uses 'last')")
    if c.width % c.heads:
        raise ValueError("text width must be a multiple of heads")
    return c


def gelu_is_tanh(act_kwargs: Optional[dict]) -> bool:
    """nn.GELU(**act_kwargs): 'none' (erf) unless approximate == 'tanh' (model.py:196-197)."""
    if not act_kwargs:
        return False
    a = act_kwargs.get("approximate", "none")
    if a not in ("none", "tanh"):
        raise ValueError(f"GELU approximate={a!r}")
    return a == "tanh"


def ln_eps(norm_kwargs: Optional[dict]) -> float:
    """The vendored LayerNorm defaults to eps=1e-6 (transformer.py:458,690)."""
    if norm_kwargs and "eps" in norm_kwargs:
        return float(norm_kwargs["eps"])
    return 1e-6


# OpenVision size table: (vision width, layers, mlp_ratio), text (width, layers, heads), embed_dim.
# transfer_jax2hf.py:76-92, configs/openvision.py:257-263.
_SIZES = {
    "Ti":     dict(vw=192,  vl=12, vmr=4.0,    tw=192,  tl=12, th=3,  e=192,  hw=64),
    "S":      dict(vw=384,  vl=12, vmr=4.0,    tw=384,  tl=12, th=6,  e=384,  hw=64),
    "B":      dict(vw=768,  vl=12, vmr=4.0,    tw=512,  tl=12, th=8,  e=512,  hw=64),
    "L":      dict(vw=1024, vl=24, vmr=4.0,    tw=768,  tl=12, th=12, e=768,  hw=64),
    "So400m": dict(vw=1152, vl=27, vmr=3.7362, tw=1152, tl=27, th=16, e=1152, hw=72, tmr=3.7362),
    "H":      dict(vw=1280, vl=32, vmr=4.0,    tw=1024, tl=24, th=16, e=1024, hw=80),
}


def openvision_model_cfg(size: str, patch: int, image: int, *, context_length: int = 80,
                         vocab_size: int = 32000) -> Dict[str, Any]:
    """``model_cfg`` block of an OpenVision ``open_clip_config.json``."""
    s = _SIZES[size]
    return {
        "embed_dim": s["e"],
        "vision_cfg": {
            "image_size": image, "patch_size": patch, "layers": s["vl"], "width": s["vw"],
            "head_width": s["hw"], "mlp_ratio": s["vmr"], "pool_type": "avg",
            "final_ln_after_pool": True, "no_ln_pre": True,
        },
        "text_cfg": {
            "context_length": context_length, "vocab_size": vocab_size, "layers": s["tl"],
            "width": s["tw"], "heads": s["th"], "mlp_ratio": s.get("tmr", 4.0),
            "pool_type": "last", "no_causal_mask": True,
            "act_kwargs": {"approximate": "tanh"},
        },
    }


PRESETS = {
    "vit-tiny-patch16-160": ("Ti", 16, 160),
    "vit-small-patch8-384": ("S", 8, 384),
    "vit-base-patch16-224": ("B", 16, 224),
    "vit-large-patch14-224": ("L", 14, 224),
    "vit-so400m-patch14-224": ("So400m", 14, 224),
    "vit-huge-patch14-224": ("H", 14, 224),
}


def preset(name: str) -> Dict[str, Any]:
    return openvision_model_cfg(*PRESETS[name])


DEFAULT_PREPROCESS = {"mean": [0.48145466, 0.4578275, 0.40821073],
                      "std": [0.26862954, 0.26130258, 0.27577711]}


def load_config_dir(path: str) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """Read ``<dir>/open_clip_config.json`` the way ov-zero-shot-test.py:38-49 does.

    Returns (model_cfg, preprocess_cfg)."""
    with open(os.path.join(path, "open_clip_config.json"), "r") as f:
        cfg = json.load(f)
    return cfg["model_cfg"], cfg.get("preprocess_cfg", dict(DEFAULT_PREPROCESS))


def mlp_width(width: int, mlp_ratio: float) -> int:
    return int(width * mlp_ratio)   # transformer.py:231
