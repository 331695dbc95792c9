"""Formula-generated weights and synthetic batches.

No OpenVision checkpoint exists offline, so every test, fixture and benchmark uses weights drawn
from a counter-based generator keyed by the *state-dict key name*: ``Philox(key=[crc32(name), seed])``.
The same function therefore rebuilds byte-identical weights in this container (where the golden
vectors are made with the reference as oracle) and on the GPU box (where the reference is absent).

State-dict key set = the strict key set of the reference's ``CLIP`` for OpenVision configs
(``open_clip/model.py:223-254``, verified against the vendored class).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Any

import numpy as np
import torch

from .config import vision_cfg_from, text_cfg_from, mlp_width


def _gen(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()), seed & 0xFFFFFFFF]))


def _normal(name: str, shape, std: float, seed: int, mean: float = 0.0) -> torch.Tensor:
    a = _gen(name, seed).standard_normal(size=shape, dtype=np.float32)
    a *= np.float32(std)
    if mean:
        a += np.float32(mean)
    return torch.from_numpy(a)


def posemb_sincos_2d(h: int, w: int, width: int, temperature: float = 10000.0) -> torch.Tensor:
    """MoCo-v3 2-D sincos table with a zero cls row, as the converter stores it in the checkpoint
    (``transfer_jax2hf.py:96-112,132-133``): [sin x, cos x, sin y, cos y], omega_k = T^(-k/(W/4-1))."""
    assert width % 4 == 0
    y, x = np.mgrid[:h, :w]
    omega = np.arange(width // 4, dtype=np.float64) / (width // 4 - 1)
    omega = 1.0 / (temperature ** omega)
    yy = np.einsum("m,d->md", y.flatten().astype(np.float64), omega)
    xx = np.einsum("m,d->md", x.flatten().astype(np.float64), omega)
    pe = np.concatenate([np.sin(xx), np.cos(xx), np.sin(yy), np.cos(yy)], axis=1)
    pe = np.concatenate([np.zeros([1, width]), pe], axis=0)
    return torch.from_numpy(pe.astype(np.float32))


def _block(sd: Dict[str, torch.Tensor], prefix: str, d: int, mlp: int, seed: int) -> None:
    n = lambda k, shape, std, mean=0.0: sd.__setitem__(prefix + k, _normal(prefix + k, shape, std, seed, mean))
    n("ln_1.weight", (d,), 0.1, 1.0)
    n("ln_1.bias", (d,), 0.1)
    n("attn.in_proj_weight", (3 * d, d), 1.0 / math.sqrt(d))
    n("attn.in_proj_bias", (3 * d,), 0.1)
    n("attn.out_proj.weight", (d, d), 0.5 / math.sqrt(d))
    n("attn.out_proj.bias", (d,), 0.1)
    n("ln_2.weight", (d,), 0.1, 1.0)
    n("ln_2.bias", (d,), 0.1)
    n("mlp.c_fc.weight", (mlp, d), 1.0 / math.sqrt(d))
    n("mlp.c_fc.bias", (mlp,), 0.1)
    n("mlp.c_proj.weight", (d, mlp), 0.5 / math.sqrt(mlp))
    n("mlp.c_proj.bias", (d,), 0.1)


def make_state_dict(model_cfg: Dict[str, Any], seed: int = 0, variant: str = "v1") -> Dict[str, torch.Tensor]:
    """fp32 state dict for ``CLIP(**model_cfg)`` (reference key names and shapes).

    ``variant="v1"``: plain random weights — the encoder they define is nearly input-independent (different images land
    within cos 0.99 of each other), good for throughput runs and operator-level parity only.
    ``variant="sharp"``: the same draws reshaped so that distinct inputs give separated embeddings (peaked attention,
    small biases, mean-free MLP output projections, a decaying spectrum shared by both output projections): the weight
    set of the discriminating parity fixtures (``tests/golden/*_sharp.npz``)."""
    sd = _make_v1(model_cfg, seed)
    if variant == "sharp":
        _sharpen(sd, int(model_cfg["embed_dim"]))
    elif variant != "v1":
        raise ValueError(f"unknown weight variant {variant!r}")
    return sd


SHARP = {"qk_vision": 2.5, "qk_text": 1.5, "bias": 0.2, "token_embedding": 2.0, "spectrum": 4.0}


def _sharpen(sd: Dict[str, torch.Tensor], e: int) -> None:
    prof = 1.0 / (1.0 + torch.arange(e, dtype=torch.float32) / SHARP["spectrum"])
    prof = prof / prof.norm() * math.sqrt(e)
    for k in list(sd):
        if k.endswith(".bias") or k.endswith("in_proj_bias"):
            sd[k] = sd[k] * SHARP["bias"]
        elif k.endswith("c_proj.weight"):
            sd[k] = sd[k] - sd[k].mean(dim=1, keepdim=True)
        elif k.endswith("in_proj_weight"):
            d = sd[k].shape[1]
            sd[k][:2 * d] *= SHARP["qk_vision"] if k.startswith("visual.") else SHARP["qk_text"]
    sd["token_embedding.weight"] = sd["token_embedding.weight"] * SHARP["token_embedding"]
    sd["visual.proj"] = sd["visual.proj"] * prof
    sd["text_projection"] = sd["text_projection"] * prof


def _make_v1(model_cfg: Dict[str, Any], seed: int = 0) -> Dict[str, torch.Tensor]:
    v = vision_cfg_from(model_cfg["vision_cfg"])
    t = text_cfg_from(model_cfg["text_cfg"])
    e = int(model_cfg["embed_dim"])
    sd: Dict[str, torch.Tensor] = {}
    g = v.image_size // v.patch_size
    d = v.width
    sd["visual.class_embedding"] = _normal("visual.class_embedding", (d,), 0.5, seed)
    sd["visual.positional_embedding"] = posemb_sincos_2d(g, g, d)
    sd["visual.proj"] = _normal("visual.proj", (d, e), 1.0 / math.sqrt(d), seed)
    sd["visual.conv1.weight"] = _normal("visual.conv1.weight", (d, 3, v.patch_size, v.patch_size),
                                        1.0 / math.sqrt(3 * v.patch_size ** 2), seed)
    for i in range(v.layers):
        _block(sd, f"visual.transformer.resblocks.{i}.", d, mlp_width(d, v.mlp_ratio), seed)
    sd["visual.ln_post.weight"] = _normal("visual.ln_post.weight", (d,), 0.1, seed, 1.0)
    sd["visual.ln_post.bias"] = _normal("visual.ln_post.bias", (d,), 0.1, seed)

    dt = t.width
    sd["token_embedding.weight"] = _normal("token_embedding.weight", (t.vocab_size, dt), 0.5, seed)
    sd["positional_embedding"] = _normal("positional_embedding", (t.context_length, dt), 0.25, seed)
    for i in range(t.layers):
        _block(sd, f"transformer.resblocks.{i}.", dt, mlp_width(dt, t.mlp_ratio), seed)
    sd["ln_final.weight"] = _normal("ln_final.weight", (dt,), 0.1, seed, 1.0)
    sd["ln_final.bias"] = _normal("ln_final.bias", (dt,), 0.1, seed)
    sd["text_projection"] = _normal("text_projection", (dt, e), 1.0 / math.sqrt(dt), seed)
    sd["logit_scale"] = torch.tensor(math.log(1.0 / 0.07), dtype=torch.float32)   # model.py:229
    return sd


def make_images(batch: int, image_size: int, seed: int = 0) -> torch.Tensor:
    """[B,3,S,S] ~ N(0,1): the distribution after Normalize (ov-zero-shot-test.py:76)."""
    return _normal("synthetic.images", (batch, 3, image_size, image_size), 1.0, seed)


def make_structured_images(batch: int, image_size: int, seed: int = 0) -> torch.Tensor:
    """[B,3,S,S]: 0.6 * N(0,1) pixel noise + a smooth per-image colour field (a 4x4x3 N(0,1) grid, bilinearly upsampled).
    Unlike ``make_images`` the images differ in their global statistics, which is what survives the tower's mean pooling:
    the inputs of the discriminating fixtures (committed there as fp16, so the interpolation need not be bit-stable)."""
    noise = _normal("synthetic.structured.noise", (batch, 3, image_size, image_size), 0.6, seed)
    low = _normal("synthetic.structured.field", (batch, 3, 4, 4), 1.0, seed)
    low = torch.nn.functional.interpolate(low, size=(image_size, image_size), mode="bilinear", align_corners=False)
    return noise + low


def make_captions(batch: int, context_length: int = 80, vocab_size: int = 32000, seed: int = 0,
                  cls_id: int = 101) -> torch.Tensor:
    """int64 [B,T] in the training token format (``src/transforms/bert_ops.py:496-507``):
    ``[1=bos, w_1..w_k, 2=eos, 0=pad.., 101=cls at T-1]``, k ~ U[4,60], w ~ U[1000, V)."""
    g = _gen("synthetic.captions", seed)
    out = np.zeros((batch, context_length), dtype=np.int64)
    hi = min(60, context_length - 3)
    k = g.integers(min(4, hi), hi + 1, size=batch)
    w = g.integers(min(1000, vocab_size - 1), vocab_size, size=(batch, context_length))
    for b in range(batch):
        out[b, 0] = 1
        out[b, 1:1 + k[b]] = w[b, :k[b]]
        out[b, 1 + k[b]] = 2
        out[b, context_length - 1] = cls_id
    return torch.from_numpy(out)


# --- algorithmic work (SURVEY.md §8d / BASELINE.md §4); FLOP = 2 * MAC -----------------------------
def tower_flops(width: int, layers: int, mlp: int, seq: int, heads: int) -> float:
    hd = width // heads
    per_tok = 2 * width * 3 * width + 2 * width * width + 2 * 2 * width * mlp
    attn = 2 * 2 * seq * seq * hd * heads
    return layers * (seq * per_tok + attn)


def model_flops(model_cfg: Dict[str, Any]) -> Dict[str, float]:
    v = vision_cfg_from(model_cfg["vision_cfg"])
    t = text_cfg_from(model_cfg["text_cfg"])
    e = int(model_cfg["embed_dim"])
    g = v.image_size // v.patch_size
    L = g * g + 1
    img = tower_flops(v.width, v.layers, mlp_width(v.width, v.mlp_ratio), L, v.width // v.head_width)
    img += 2 * g * g * 3 * v.patch_size ** 2 * v.width + 2 * v.width * e
    txt = tower_flops(t.width, t.layers, mlp_width(t.width, t.mlp_ratio), t.context_length, t.heads)
    txt += 2 * t.width * e
    return {"image": float(img), "text": float(txt), "pair": float(img + txt)}
