"""JAX parameter names -> open_clip state dict (the key map of the reference's converter
``src/convert_upload/transfer_jax2hf.py:115-453``, ``txt_model='vit'`` branch — the one OpenVision uses).

Input is the FLAT parameter dictionary of the JAX trainer (``tree_flatten_with_names``: '/'-joined names -> numpy arrays), e.g.
exported with ``numpy.savez``; reading an orbax checkpoint directory itself needs jax/orbax and is not done here.  Pure host-side
re-layout (transposes, reshapes, q|k|v packing) in numpy; every tensor comes out fp32 like the converter's (``DTYPE``).

    img/cls [1,1,D] -> visual.class_embedding            img/embedding/kernel [P,P,3,D] -> visual.conv1.weight [D,3,P,P]
    img/encoder_norm/* -> visual.ln_post.*               img/head/kernel [D,E] -> visual.proj
    img/Transformer/encoderblock_i/LayerNorm_{0,1}/{scale,bias} -> visual.transformer.resblocks.i.ln_{1,2}.{weight,bias}
    .../MlpBlock_0/Dense_{0,1}/kernel [in,out] -> mlp.{c_fc,c_proj}.weight [out,in]
    .../MultiHeadDotProductAttention_0/{query,key,value}/kernel [D,H,hd] -> attn.in_proj_weight [3D,D] (q | k | v), bias [H,hd] -> [3D]
    .../MultiHeadDotProductAttention_0/out/kernel [H,hd,D] -> attn.out_proj.weight [D,D]
    txt/pos_embedding [1,T,Dt] -> positional_embedding   txt/Embed_0/embedding -> token_embedding.weight
    txt/encoder_norm/* -> ln_final.*                     txt/head/kernel -> text_projection     t [1] -> logit_scale
The vision positional embedding is either learned (``img/pos_embedding``) or the fixed MoCo-v3 2-D sincos table the converter
regenerates (``transfer_jax2hf.py:98-112``; OpenVision: sincos2d, class row zeros).

The reference converter cannot be executed here (it imports jax / orbax / flax at module level), so this restatement is checked by
its own inverse and by the strict key set of the model (``tests/test_convert_jax.py``): **parity unpinned** against a run of the
reference converter.
"""
from __future__ import annotations

import re
from typing import Dict, Optional, Tuple

import numpy as np
import torch

_BLOCK = re.compile(r"^(img|txt)/Transformer/encoderblock_(\d+)/(.+)$")


def posemb_sincos_2d(h: int, w: int, width: int, temperature: float = 10_000.0, cls_token: bool = True) -> np.ndarray:
    """transfer_jax2hf.py:98-112 (MoCo v3): [sin x, cos x, sin y, cos y] with omega_k = T^(-k / (width/4 - 1)); class row zeros."""
    if width % 4:
        raise ValueError("width must be a multiple of 4 for the sincos positional embedding")
    y, x = np.mgrid[:h, :w]
    omega = np.arange(width // 4, dtype=np.float32) / np.float32(width // 4 - 1)
    omega = (1.0 / (np.float32(temperature) ** omega)).astype(np.float32)
    yy = np.einsum("m,d->md", y.flatten().astype(np.float32), omega)
    xx = np.einsum("m,d->md", x.flatten().astype(np.float32), omega)
    pe = np.concatenate([np.sin(xx), np.cos(xx), np.sin(yy), np.cos(yy)], axis=1).astype(np.float32)
    if cls_token:
        pe = np.concatenate([np.zeros((1, width), np.float32), pe], axis=0)
    return pe


def jax_to_open_clip(flat: Dict[str, np.ndarray], grid: Optional[Tuple[int, int]] = None, pos_embed: str = "sincos2d",
                     use_dense_general: bool = True) -> Dict[str, torch.Tensor]:
    """Flat JAX parameters -> open_clip state dict (fp32).  ``grid`` = (rows, cols) of patches, needed for ``sincos2d``."""
    f = {k: np.asarray(v, dtype=np.float32) for k, v in flat.items()}
    sd: Dict[str, np.ndarray] = {}
    if pos_embed == "learn":
        sd["visual.positional_embedding"] = np.squeeze(f.pop("img/pos_embedding"))
    elif pos_embed == "sincos2d":
        if grid is None:
            raise ValueError("sincos2d needs the patch grid")
        width = f["img/cls"].shape[-1]
        sd["visual.positional_embedding"] = posemb_sincos_2d(grid[0], grid[1], width, cls_token=True)
        f.pop("img/pos_embedding", None)
    else:
        raise ValueError(f"unknown pos_embed {pos_embed!r}")

    qkv: Dict[Tuple[str, int], Dict[str, np.ndarray]] = {}
    simple = {"img/encoder_norm/scale": "visual.ln_post.weight", "img/encoder_norm/bias": "visual.ln_post.bias",
              "img/head/kernel": "visual.proj", "img/embedding/bias": "visual.conv1.bias", "img/head/bias": "visual.proj_bias",
              "txt/Embed_0/embedding": "token_embedding.weight", "txt/encoder_norm/scale": "ln_final.weight",
              "txt/encoder_norm/bias": "ln_final.bias", "txt/head/kernel": "text_projection"}
    for k, v in f.items():
        if k in simple:
            sd[simple[k]] = v
        elif k == "img/cls":
            sd["visual.class_embedding"] = v[0, 0, :]
        elif k == "img/embedding/kernel":
            sd["visual.conv1.weight"] = v.transpose(3, 2, 0, 1)
        elif k == "txt/pos_embedding":
            sd["positional_embedding"] = v[0]
        elif k == "t":
            sd["logit_scale"] = np.asarray(v).reshape(-1)[0]
        else:
            m = _BLOCK.match(k)
            if not m:
                raise ValueError(f"unexpected parameter {k!r}")
            tower, i, rest = m.group(1), int(m.group(2)), m.group(3)
            p = ("visual.transformer" if tower == "img" else "transformer") + f".resblocks.{i}."
            parts = rest.split("/")
            if parts[0].startswith("LayerNorm_"):
                n = int(parts[0].split("_")[1]) + 1
                sd[p + f"ln_{n}." + {"scale": "weight", "bias": "bias"}[parts[1]]] = v
            elif parts[0] == "MlpBlock_0":
                name = {"Dense_0": "mlp.c_fc", "Dense_1": "mlp.c_proj"}[parts[1]]
                sd[p + name + (".weight" if parts[2] == "kernel" else ".bias")] = v.transpose(1, 0) if parts[2] == "kernel" else v
            elif parts[0] == "MultiHeadDotProductAttention_0":
                if parts[1] == "out":
                    if parts[2] == "bias":
                        sd[p + "attn.out_proj.bias"] = v
                    else:
                        w = v.reshape(v.shape[0] * v.shape[1], v.shape[2]) if use_dense_general else v
                        sd[p + "attn.out_proj.weight"] = w.transpose(1, 0)
                elif parts[1] in ("query", "key", "value"):
                    qkv.setdefault((p, i), {})[parts[1] + "/" + parts[2]] = v
                else:
                    raise ValueError(f"unexpected parameter {k!r}")
            else:
                raise ValueError(f"unexpected parameter {k!r}")
    for (p, _), d in qkv.items():
        ws, bs = [], []
        for name in ("query", "key", "value"):
            w, b = d[name + "/kernel"], d[name + "/bias"]
            if use_dense_general:
                w, b = w.reshape(w.shape[0], w.shape[1] * w.shape[2]), b.reshape(-1)
            ws.append(w.transpose(1, 0))
            bs.append(b)
        sd[p + "attn.in_proj_weight"] = np.concatenate(ws, axis=0)
        sd[p + "attn.in_proj_bias"] = np.concatenate(bs, axis=0)
    return {k: torch.tensor(np.ascontiguousarray(v) if np.ndim(v) else float(v), dtype=torch.float32) for k, v in sd.items()}


def open_clip_to_jax(sd: Dict[str, torch.Tensor], heads_vision: int, heads_text: int, patch: int) -> Dict[str, np.ndarray]:
    """The inverse re-layout (for export and for testing the map): open_clip state dict -> flat JAX parameter names."""
    out: Dict[str, np.ndarray] = {}
    g = lambda k: sd[k].detach().float().cpu().numpy()
    d = g("visual.class_embedding").shape[0]
    out["img/cls"] = g("visual.class_embedding").reshape(1, 1, d)
    out["img/embedding/kernel"] = g("visual.conv1.weight").transpose(2, 3, 1, 0)
    out["img/pos_embedding"] = g("visual.positional_embedding")[None]
    out["img/encoder_norm/scale"], out["img/encoder_norm/bias"] = g("visual.ln_post.weight"), g("visual.ln_post.bias")
    out["img/head/kernel"] = g("visual.proj")
    out["txt/pos_embedding"] = g("positional_embedding")[None]
    out["txt/Embed_0/embedding"] = g("token_embedding.weight")
    out["txt/encoder_norm/scale"], out["txt/encoder_norm/bias"] = g("ln_final.weight"), g("ln_final.bias")
    out["txt/head/kernel"] = g("text_projection")
    out["t"] = g("logit_scale").reshape(1)
    for tower, prefix, heads in (("img", "visual.transformer", heads_vision), ("txt", "transformer", heads_text)):
        i = 0
        while f"{prefix}.resblocks.{i}.ln_1.weight" in sd:
            p, q = f"{prefix}.resblocks.{i}.", f"{tower}/Transformer/encoderblock_{i}/"
            for n in (1, 2):
                out[q + f"LayerNorm_{n - 1}/scale"], out[q + f"LayerNorm_{n - 1}/bias"] = g(p + f"ln_{n}.weight"), g(p + f"ln_{n}.bias")
            for a, b in (("Dense_0", "mlp.c_fc"), ("Dense_1", "mlp.c_proj")):
                out[q + f"MlpBlock_0/{a}/kernel"], out[q + f"MlpBlock_0/{a}/bias"] = g(p + b + ".weight").T, g(p + b + ".bias")
            w, bb = g(p + "attn.in_proj_weight"), g(p + "attn.in_proj_bias")
            dm = w.shape[1]
            hd = dm // heads
            for j, name in enumerate(("query", "key", "value")):
                out[q + f"MultiHeadDotProductAttention_0/{name}/kernel"] = w[j * dm:(j + 1) * dm].T.reshape(dm, heads, hd)
                out[q + f"MultiHeadDotProductAttention_0/{name}/bias"] = bb[j * dm:(j + 1) * dm].reshape(heads, hd)
            out[q + "MultiHeadDotProductAttention_0/out/kernel"] = g(p + "attn.out_proj.weight").T.reshape(heads, hd, dm)
            out[q + "MultiHeadDotProductAttention_0/out/bias"] = g(p + "attn.out_proj.bias")
            i += 1
    return out
