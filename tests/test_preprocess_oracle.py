"""CPU: the image-transform oracle (oracle/preprocess_ref.py) against Pillow's committed outputs, and against Pillow itself."""
import numpy as np
import pytest

from conftest import golden
from oracle import preprocess_ref as P
from openvision_amd.config import DEFAULT_PREPROCESS as PP
from openvision_amd.preprocess import resize_plan, output_geometry


def test_oracle_matches_committed_pillow_outputs():
    g = golden("preprocess.npz")
    for i in range(5):
        assert np.array_equal(P.pil_resize_u8(g[f"cat{i}_in"], 160, 160, "bilinear"), g[f"cat{i}_u8"])      # bit-exact uint8
    for i in range(2):
        out = P.transform(g[f"cat{i}_in"], 160, PP["mean"], PP["std"], "squash", "bilinear")
        assert np.array_equal(out, g[f"cat{i}_out"])                                                         # bit-exact fp32
    for k in ("noise", "grad", "up"):
        img = g[f"{k}_in"]
        assert np.array_equal(P.pil_resize_u8(img, 160, 160, "bilinear"), g[f"{k}_u8_squash_bilinear"])
        hr, wr, cy, cx = output_geometry(img.shape[0], img.shape[1], 224, "shortest")
        r = P.pil_resize_u8(img, hr, wr, "bicubic")[cy:cy + 224, cx:cx + 224]
        assert np.array_equal(r, g[f"{k}_u8_shortest_bicubic"])


def test_oracle_matches_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    g = np.random.default_rng(3)
    for (h, w, oh, ow) in [(37, 53, 160, 160), (224, 224, 160, 160), (64, 64, 64, 200), (5, 7, 3, 2), (130, 90, 224, 224)]:
        img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for name, flt in (("bilinear", Image.BILINEAR), ("bicubic", Image.BICUBIC)):
            assert np.array_equal(P.pil_resize_u8(img, oh, ow, name), np.asarray(Image.fromarray(img).resize((ow, oh), flt)))


def test_plan_shapes_and_unit_gain():
    for (i, o, f) in [(224, 160, "bilinear"), (40, 224, "bicubic"), (1000, 384, "bicubic")]:
        b, k, ks = resize_plan(i, o, f)
        assert b.shape == (o, 2) and k.shape == (o, ks)
        assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= i).all()
        assert np.abs(k.sum(1) - (1 << 22)).max() <= ks            # taps sum to 1.0 in 22-bit fixed point (rounding)
