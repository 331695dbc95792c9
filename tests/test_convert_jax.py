"""The JAX -> open_clip key map (openvision_amd.convert_jax, restating transfer_jax2hf.py:115-453): round trip through its inverse,
strict key set of the model, and the MoCo-v3 sincos table against the formula weights' own positional embedding."""
import numpy as np
import pytest
import torch

from openvision_amd import preset, synth
from openvision_amd.convert_jax import jax_to_open_clip, open_clip_to_jax, posemb_sincos_2d


@pytest.mark.parametrize("name", ["vit-tiny-patch16-160", "vit-so400m-patch14-224"])
def test_round_trip_and_key_set(name):
    cfg = preset(name)
    v, t = cfg["vision_cfg"], cfg["text_cfg"]
    if name.startswith("vit-so400m"):                    # keep the big preset's SHAPES (head_dim 72, mlp 4304) but two layers
        v, t = dict(v, layers=2), dict(t, layers=2)
        cfg = dict(cfg, vision_cfg=v, text_cfg=t)
    sd = synth.make_state_dict(cfg)
    hv, ht = v["width"] // v["head_width"], t["heads"]
    flat = open_clip_to_jax(sd, hv, ht, v["patch_size"])
    d = v["width"]
    assert flat["img/Transformer/encoderblock_0/MultiHeadDotProductAttention_0/query/kernel"].shape == (d, hv, d // hv)
    assert flat["img/Transformer/encoderblock_1/MultiHeadDotProductAttention_0/out/kernel"].shape == (hv, d // hv, d)
    assert flat["img/embedding/kernel"].shape == (v["patch_size"], v["patch_size"], 3, d)
    g = v["image_size"] // v["patch_size"]
    back = jax_to_open_clip(flat, grid=(g, g), pos_embed="learn")
    assert set(back) == set(sd)                           # exactly the keys the strict load_state_dict expects
    for k in sd:
        assert back[k].dtype == torch.float32 and torch.equal(back[k], sd[k].float()), k
    again = jax_to_open_clip(flat, grid=(g, g), pos_embed="sincos2d")
    assert torch.allclose(again["visual.positional_embedding"], sd["visual.positional_embedding"].float(), atol=2e-6)


def test_sincos_table_and_errors():
    pe = posemb_sincos_2d(3, 2, 8)
    assert pe.shape == (7, 8) and not pe[0].any()
    # patch (row 1, col 1) = flat index 3: x = 1, y = 1; omega = [1, 1e-4]
    np.testing.assert_allclose(pe[1 + 3], [np.sin(1), np.sin(1e-4), np.cos(1), np.cos(1e-4)] * 2, rtol=1e-6)
    with pytest.raises(ValueError):
        jax_to_open_clip({"img/cls": np.zeros((1, 1, 8)), "weird/name": np.zeros(3)}, grid=(2, 2))
