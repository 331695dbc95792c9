"""CPU: the oracle (oracle/clip_ref.py) against the golden vectors the REFERENCE produced
(tests/golden/make_golden.py).  This is what pins the oracle; fp32, tight tolerances."""
import os

import numpy as np
import pytest
import torch

from openvision_amd import config as ovcfg, synth
from oracle import clip_ref as R
from conftest import golden

T = torch.from_numpy


def close(a, b, atol, rtol=1e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def test_ops_layernorm_gelu_block():
    g = golden("ops.npz")
    assert float(g["ln_eps"]) == 1e-6
    close(R.layer_norm(T(g["ln_x"]), T(g["ln_w"]), T(g["ln_b"])), g["ln_y"], 1e-6)
    close(R.gelu(T(g["gelu_x"]), False), g["gelu_erf"], 1e-7)
    close(R.gelu(T(g["gelu_x"]), True), g["gelu_tanh"], 1e-7)
    sd = synth.make_state_dict(ovcfg.preset("vit-tiny-patch16-160"), 0)
    y = R.resblock(T(g["blk_x"]), sd, "visual.transformer.resblocks.0.", 3, False)
    close(y, g["blk_y"], 2e-5)


@pytest.fixture(scope="module")
def tiny():
    cfg = ovcfg.preset("vit-tiny-patch16-160")
    return cfg, synth.make_state_dict(cfg, 0)


def test_tiny_tower_activations(tiny):
    cfg, sd = tiny
    g = golden("tiny16_160.npz")
    img = T(g["images"])
    x = R.patch_embed(img, sd, 16)
    conv = torch.nn.functional.conv2d(img, sd["visual.conv1.weight"], None, stride=16)
    close(conv, g["conv1"], 1e-5)
    x = R.resblock(x, sd, "visual.transformer.resblocks.0.", 3, False)
    close(x, g["block0"], 2e-5)
    feat, tokens = R.vision_forward(img, sd, cfg["vision_cfg"], return_tokens=True)
    close(tokens, g["block11"], 2e-4, 1e-4)
    close(feat, g["image_features"], 1e-4, 1e-4)


def test_tiny_text_and_forward(tiny):
    cfg, sd = tiny
    g = golden("tiny16_160.npz")
    img, tok = T(g["images"]), T(g["tokens"])
    close(R.encode_text(tok, sd, cfg), g["text_features"], 1e-4, 1e-4)
    ni, nt, s = R.clip_forward(img, tok, sd, cfg)
    close(ni, g["image_norm"], 2e-6)
    close(nt, g["text_norm"], 2e-6)
    close(s, g["logit_scale_exp"], 1e-5)
    close(s * ni @ nt.T, g["logits_per_image"], 1e-4)
    close(s * nt @ ni.T, g["logits_per_text"], 1e-4)
    close(R.clip_loss(ni, nt, s), g["loss"], 1e-5)


def test_tiny_testcat_zero_shot_table():
    cfg = ovcfg.preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg, 0, "sharp")
    g = golden("tiny16_160_testcat.npz")
    assert str(g["variant"]) == "sharp"
    img, tok = T(g["images"].astype(np.float32)), T(g["tokens"])
    cos, probs, order = R.zero_shot_table(R.encode_image(img, sd, cfg), R.encode_text(tok, sd, cfg),
                                          sd["logit_scale"])
    close(cos, g["cosine"], 2e-6)
    close(probs, g["probs"], 2e-5)
    assert np.array_equal(order.numpy(), g["argsort"])            # top-k indices bit-exact
    assert np.array_equal(probs.argmax(-1).numpy(), g["best"])


def test_tiny_testcat_cli_table_from_the_png_files():
    """The image-sensitive zero-shot fixture (make_golden.gen_testcat_cli): the oracle on the committed PNG files (ToTensor + Normalize,
    the script's Resize((160, 160)) being the identity on them), the fixture's two replaced projections and its token rows reproduces
    the reference's cosine table, every decided rank and both of the script's decisions (best text per image, best image per text);
    the CLI's own tokenizer gives the fixture's token rows for the nine prompts."""
    from PIL import Image
    from openvision_amd.tokenizer import WordPieceTokenizer
    cfg = ovcfg.preset("vit-tiny-patch16-160")
    g = golden("tiny16_160_testcat_cli.npz")
    sd = synth.make_state_dict(cfg, 0, "sharp")
    sd["visual.proj"], sd["text_projection"] = T(g["visual_proj"]), T(g["text_projection"])
    pp = ovcfg.DEFAULT_PREPROCESS
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "testcat_160")
    assert sorted(os.listdir(here)) == [str(n) for n in g["names"]]
    arr = [(np.asarray(Image.open(os.path.join(here, str(n))).convert("RGB"), dtype=np.float32) / 255.0 - np.asarray(pp["mean"], np.float32))
           / np.asarray(pp["std"], np.float32) for n in g["names"]]
    img = T(np.stack(arr).transpose(0, 3, 1, 2).copy())
    tok = WordPieceTokenizer(None, context_length=80)([str(t) for t in g["prompts"]])
    assert np.array_equal(tok.numpy(), g["tokens"])
    cos, probs, order = R.zero_shot_table(R.encode_image(img, sd, cfg), R.encode_text(tok, sd, cfg), sd["logit_scale"])
    close(cos, g["cosine"], 2e-5)                                  # the projections amplify fp32 summation-order noise ~10x
    gap = float(g["gap"])
    for r in range(cos.shape[0]):
        k = 0
        srt = np.sort(g["cosine"][r])[::-1]
        while k + 1 < len(srt) and srt[k] - srt[k + 1] > gap:
            k += 1
        assert k >= 3 and np.array_equal(order[r, :k].numpy(), g["argsort"][r][:k])
    assert np.array_equal(probs.argmax(-1).numpy(), g["best_text_per_image"])
    assert np.array_equal(probs.argmax(0).numpy(), g["best_image_per_text"])


@pytest.mark.timeout(600)
def test_large_features():
    cfg = ovcfg.preset("vit-large-patch14-224")
    sd = synth.make_state_dict(cfg, 0)
    g = golden("large14_224.npz")
    img, tok = T(g["images"].astype(np.float32)), T(g["tokens"])
    feat, tokens = R.vision_forward(img, sd, cfg["vision_cfg"], return_tokens=True)
    close(tokens[:, :4], g["block23_head"], 1e-3, 1e-4)
    close(tokens[:, -2:], g["block23_tail"], 1e-3, 1e-4)
    close(feat, g["image_features"], 2e-4, 1e-4)
    tf = R.encode_text(tok, sd, cfg)
    close(tf, g["text_features"], 2e-4, 1e-4)
    ni, nt = R.l2_normalize(feat), R.l2_normalize(tf)
    close(R.clip_loss(ni, nt, sd["logit_scale"].exp()), g["loss"], 1e-5)
    # the reference's own bf16 mode sits this far from its fp32 mode (documents the bf16 budget)
    cb = torch.nn.functional.cosine_similarity(T(g["image_features_refbf16"]), T(g["image_features"]))
    assert float((1 - cb).max()) < 1e-3


@pytest.mark.timeout(600)
def test_small8_384_features():
    cfg = ovcfg.preset("vit-small-patch8-384")
    sd = synth.make_state_dict(cfg, 0)
    g = golden("small8_384.npz")
    feat, tokens = R.vision_forward(T(g["images"].astype(np.float32)), sd, cfg["vision_cfg"], return_tokens=True)
    close(tokens[:, :4], g["block11_head"], 5e-4, 1e-4)
    close(feat, g["image_features"], 2e-4, 1e-4)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,preset,last", [("tiny16_160_sharp.npz", "vit-tiny-patch16-160", 11),
                                               ("large14_224_sharp.npz", "vit-large-patch14-224", 23),
                                               ("small8_384_sharp.npz", "vit-small-patch8-384", 11)])
def test_sharp_fixtures_discriminate_and_pin_the_oracle(name, preset, last):
    """The discriminating fixtures ('sharp' weights, structured images): the reference's outputs for DIFFERENT inputs are far
    apart (so a wrong row / input-independent encoder cannot pass the cosine gate), and the oracle reproduces them."""
    cfg = ovcfg.preset(preset)
    sd = synth.make_state_dict(cfg, 0, "sharp")
    g = golden(name)
    n = g["images"].shape[0]
    off = g["image_image_cos"][~np.eye(n, dtype=bool)]
    assert off.max() < 0.9                                  # different images: 1 - cos > 0.1 = 100x the 1e-3 gate
    img = T(g["images"].astype(np.float32))
    feat, tokens = R.vision_forward(img, sd, cfg["vision_cfg"], return_tokens=True)
    close(tokens[:, :4], g[f"block{last}_head"], 2e-3, 2e-4)
    close(tokens[:, -2:], g[f"block{last}_tail"], 2e-3, 2e-4)
    cos = torch.nn.functional.cosine_similarity(feat, T(g["image_features"]))
    assert float((1 - cos).max()) < 1e-6
    x = R.resblock(R.patch_embed(img, sd, cfg["vision_cfg"]["patch_size"]), sd, "visual.transformer.resblocks.0.",
                   cfg["vision_cfg"]["width"] // cfg["vision_cfg"]["head_width"], False)
    close(x[:, :4], g["block0_head"], 1e-4, 1e-4)
    if "text_features" in g.files:
        tf = R.encode_text(T(g["tokens"]), sd, cfg)
        assert float((1 - torch.nn.functional.cosine_similarity(tf, T(g["text_features"]))).max()) < 1e-6
        offt = g["text_text_cos"][~np.eye(tf.shape[0], dtype=bool)]
        assert offt.max() < 0.95
        c = R.l2_normalize(feat) @ R.l2_normalize(tf).T
        close(c, g["cosine"], 2e-5)


def test_cliploss_local_losses_match_reference_ranks():
    g = golden("cliploss_ws.npz")
    img, txt, s = T(g["img"]), T(g["txt"]), T(g["scale"])
    close(R.clip_loss(img, txt, s), g["loss_ws1"], 1e-6)
    for ws in (2, 8):
        b = img.shape[0] // ws
        mine = [float(R.clip_loss(img[r * b:(r + 1) * b], txt[r * b:(r + 1) * b], s, img, txt, r)) for r in range(ws)]
        np.testing.assert_allclose(mine, g[f"local_losses_ws{ws}"], atol=1e-6)
        # mean over ranks of the local losses == the global single-process loss
        assert abs(np.mean(mine) - float(g["loss_ws1"])) < 1e-6


def test_oracle_loss_gradients_match_reference_autograd():
    """oracle.clip_loss_grads / clip_loss_backward_ranks (closed form) vs autograd through the reference ClipLoss
    (tests/golden/cliploss_grad.npz: world_size 1, and per rank at world_size 2 for local_loss x gather_with_grad)."""
    g = golden("cliploss_grad.npz")
    img, txt, s = torch.from_numpy(g["img"]), torch.from_numpy(g["txt"]), torch.from_numpy(g["scale"])
    d_img, d_txt, d_ai, d_at, d_s = R.clip_loss_grads(img, txt, s)
    assert (d_img + d_ai - torch.from_numpy(g["dimg_ws1"])).abs().max() < 2e-6
    assert (d_txt + d_at - torch.from_numpy(g["dtxt_ws1"])).abs().max() < 2e-6
    assert abs(float(d_s) - float(g["dscale_ws1"])) < 2e-6
    for local_loss in (True, False):
        for gwg in (False, True):
            tag = f"ws2_local{int(local_loss)}_gwg{int(gwg)}"
            got = R.clip_loss_backward_ranks(img, txt, s, 2, local_loss, gwg)
            for r in range(2):
                assert (got[r][0] - torch.from_numpy(g[tag + "_dimg"][r])).abs().max() < 2e-6, tag
                assert (got[r][1] - torch.from_numpy(g[tag + "_dtxt"][r])).abs().max() < 2e-6, tag
                assert abs(float(got[r][2]) - float(g[tag + "_dscale"][r])) < 2e-6, tag


def test_oracle_operator_backward_matches_reference_autograd():
    """Closed-form LayerNorm / Linear / GELU backward of the oracle vs autograd through the reference's modules (opgrad.npz)."""
    g = {k: torch.from_numpy(v) for k, v in golden("opgrad.npz").items()}
    dx, dw, db = R.layer_norm_backward(g["ln_x"], g["ln_w"], g["ln_dy"], 1e-6)
    assert (dx - g["ln_dx"]).abs().max() < 5e-6 and (dw - g["ln_dw"]).abs().max() < 2e-5 and (db - g["ln_db"]).abs().max() < 2e-5
    dx, dw, db = R.linear_backward(g["lin_dy"], g["lin_x"], g["lin_w"])
    assert (dx - g["lin_dx"]).abs().max() < 2e-5 and (dw - g["lin_dw"]).abs().max() < 5e-5 and (db - g["lin_db"]).abs().max() < 2e-5
    for name, tanh in (("erf", False), ("tanh", True)):
        da = R.gelu_backward(g[f"gelu_{name}_a"], g[f"gelu_{name}_dh"], tanh)
        assert (da - g[f"gelu_{name}_da"]).abs().max() < 2e-6, name


def test_oracle_attention_backward_matches_reference_autograd():
    """oracle.attention_backward vs autograd through nn.MultiheadAttention as the reference block builds it (opgrad.npz, identity
    out-projection): d x = dq Wq + dk Wk + dv Wv and d W_in = [dq; dk; dv]^T x."""
    g = {k: torch.from_numpy(v) for k, v in golden("opgrad.npz").items()}
    x, w, dy = g["mha_x"], g["mha_w"], g["mha_dy"]
    B, L, D = x.shape
    H = 2
    qkv = x @ w.T
    q, k, v = [t.view(B, L, H, 64).transpose(1, 2) for t in qkv.split(D, dim=-1)]
    do = dy.view(B, L, H, 64).transpose(1, 2)
    dq, dk, dv = R.attention_backward(q, k, v, do, 64 ** -0.5)
    dqkv = torch.cat([t.transpose(1, 2).reshape(B, L, D) for t in (dq, dk, dv)], dim=-1)
    assert (dqkv @ w - g["mha_dx"]).abs().max() < 2e-5
    assert (dqkv.reshape(-1, 3 * D).T @ x.reshape(-1, D) - g["mha_dw"]).abs().max() < 5e-5


def _check_block_grads(g, grads, dx, rtol_w, atol):
    assert (dx - torch.from_numpy(g["dx"])).abs().max() < atol
    for name, gr in grads.items():
        if gr.dim() == 2:
            head, norm = torch.from_numpy(g["g." + name + ".head"]), float(g["g." + name + ".norm"])
            assert (gr[:6].float() - head).abs().max() < atol + rtol_w * float(head.abs().max()), name
            assert abs(float(gr.double().norm()) - norm) < rtol_w * norm + atol, name
        else:
            want = torch.from_numpy(g["g." + name])
            assert (gr.float() - want).abs().max() < atol + rtol_w * float(want.abs().max()), name


def test_oracle_block_backward_matches_reference_autograd():
    """oracle.resblock_backward (closed-form composition) vs autograd through the reference ResidualAttentionBlock (blockgrad.npz)."""
    g = golden("blockgrad.npz")
    cfg = ovcfg.preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg, 0)
    dx, grads = R.resblock_backward(torch.from_numpy(g["x"]), torch.from_numpy(g["dy"]), sd, "visual.transformer.resblocks.0.", 3, False, 1e-6)
    _check_block_grads(g, grads, dx, 1e-4, 5e-5)
