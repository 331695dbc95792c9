"""GPU parity: the drop-in CLIP / ClipLoss surface against the golden vectors produced by the reference.

north_star tolerance: embeddings within 1e-3 cosine of the PyTorch (fp32) reference, top-k indices equal.
The HIP path computes in bf16 (fp32 accumulate); the reference's own bf16 mode sits ~2e-5 (1-cos) from its
fp32 mode on these weights (tests/test_oracle_golden.py::test_large_features)."""
import os

import numpy as np
import pytest
import torch

from openvision_amd import preset, synth
from openvision_amd.model import create_model, logits
from openvision_amd.loss import ClipLoss
from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
COS_TOL = 1e-3


def one_minus_cos(a, b):
    return (1 - torch.nn.functional.cosine_similarity(a.float().cpu(), torch.as_tensor(b).float(), dim=-1)).max().item()


@pytest.fixture(scope="module")
def tiny():
    cfg = preset("vit-tiny-patch16-160")
    return create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg))


def test_tiny_encode_image_text_and_loss(tiny):
    g = golden("tiny16_160.npz")
    img, tok = torch.from_numpy(g["images"]).to(DEV), torch.from_numpy(g["tokens"]).to(DEV)
    fi, ft = tiny.encode_image(img), tiny.encode_text(tok)
    assert fi.dtype == torch.float32 and fi.shape == (4, 192) and not fi.requires_grad
    assert one_minus_cos(fi, g["image_features"]) < COS_TOL
    assert one_minus_cos(ft, g["text_features"]) < COS_TOL
    ni, nt, s = tiny(img, tok)
    assert abs(float(s) - float(g["logit_scale_exp"])) < 1e-4
    np.testing.assert_allclose(ni.norm(dim=-1).cpu().numpy(), 1.0, atol=1e-5)
    li, lt = tiny.get_logits(img, tok)
    # logits = 14.29 * cosine: bf16-path error budget 14.29 * ~5e-3
    np.testing.assert_allclose(li.cpu().numpy(), g["logits_per_image"], atol=0.15)
    np.testing.assert_allclose(lt.cpu().numpy(), g["logits_per_text"], atol=0.15)
    loss = ClipLoss()(ni, nt, s)
    assert abs(float(loss) - float(g["loss"])) < 0.05
    tiny.check_token_range()


def test_tiny_exploded_forward_like_ov_zero_shot_test(tiny):
    """ov-zero-shot-test.py:103-155 walks the sub-modules; the exploded path must agree with encode_*."""
    g = golden("tiny16_160.npz")
    m = tiny
    image, text = torch.from_numpy(g["images"]).to(DEV), torch.from_numpy(g["tokens"]).to(DEV)
    x = m.visual.conv1(image)
    np.testing.assert_allclose(x.cpu().numpy(), g["conv1"], atol=3e-2, rtol=2e-2)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = m.visual.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype, device=x.device)
    x = torch.cat([cls, x], dim=1)
    x = x + m.visual.positional_embedding.to(x.dtype)
    x = m.visual.patch_dropout(x)
    x = m.visual.ln_pre(x)
    x = m.visual.transformer(x)
    assert np.abs(x.cpu().numpy() - g["block11"]).max() < 0.25     # bf16 residual stream over 12 blocks
    pooled = x[:, 1:].mean(dim=1)
    pooled = m.visual.ln_post(pooled)
    feats = pooled @ m.visual.proj
    assert one_minus_cos(feats, g["image_features"]) < COS_TOL
    cast = m.transformer.get_cast_dtype()
    t = m.token_embedding(text).to(cast) + m.positional_embedding[: text.shape[1]].to(cast)
    t = m.transformer(t, attn_mask=m.attn_mask)
    t = m.ln_final(t)[:, -1] @ m.text_projection
    assert one_minus_cos(t, g["text_features"]) < COS_TOL


TOPK_GAP = 5e-2     # tests/golden/make_golden.py builds the table so that the reference's neighbouring ranks are this far apart


def _leading_ranks(row, gap):
    srt = np.sort(row)[::-1]
    k = 0
    while k + 1 < len(srt) and srt[k] - srt[k + 1] > gap:
        k += 1
    return k


@pytest.fixture(scope="module")
def tiny_sharp():
    cfg = preset("vit-tiny-patch16-160")
    return create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, 0, "sharp"))


def test_tiny_testcat_table_topk(tiny_sharp):
    """ov-zero-shot-test.py:176-192 on the 5 testcat images x 9 caption rows ('sharp' weights): the whole per-image ranking and each
    image's best caption must equal the reference's (north_star: top-k indices bit-exact).  Not vacuous in the ranks -- the number
    the reference decides with margin is asserted first (all 8 per row) -- but NOT image-sensitive: on these weights the five
    pictures (one photograph, different captions drawn on it) rank the captions alike.  The image-dependent table, the script's
    second table (best image per text) and the checkpoint-directory loading are test_zero_shot_cli_from_checkpoint_directory."""
    m = tiny_sharp
    g = golden("tiny16_160_testcat.npz")
    assert str(g["variant"]) == "sharp"
    img = torch.from_numpy(g["images"].astype(np.float32)).to(DEV)
    tok = torch.from_numpy(g["tokens"]).to(DEV)
    tf = m.encode_text(tok, normalize=True)
    cos = []
    for i in range(img.shape[0]):                               # batch 1 per image, as the script does
        cos.append(logits(m.encode_image(img[i:i + 1], normalize=True), tf)[0])
    cos = torch.stack(cos).cpu()
    ref = g["cosine"]
    # cosine budget of the bf16 path: two embeddings each within 1e-3 cosine (angle 0.045 rad) of the reference
    assert np.abs(cos.numpy() - ref).max() < TOPK_GAP / 2
    order = cos.argsort(dim=-1, descending=True).numpy()
    for r in range(ref.shape[0]):
        k = _leading_ranks(ref[r], TOPK_GAP)
        assert k >= 3, f"fixture row {r} decides only {k} ranks with margin"
        assert np.array_equal(order[r][:k], g["argsort"][r][:k]), f"row {r}: top-{k} order differs"
        if k == ref.shape[1] - 1:
            assert np.array_equal(order[r], g["argsort"][r])
    probs = (float(m.logit_scale.exp()) * cos).softmax(dim=-1)
    assert np.array_equal(probs.argmax(dim=-1).numpy(), g["best"])
    np.testing.assert_allclose(probs.numpy(), g["probs"], atol=0.05)


def test_row_statistics_from_the_residual_epilogues_opt_in():
    """OVHIP_ROWPARTS=1 (read once per process: a child process): the LayerNorm statistics in front of the folded QKV / c_fc GEMMs come
    from the residual GEMMs' epilogues (ov_gemm_rowparts -> ov_rowstats_finalize; one-pass variance) instead of ov_rowstats' two passes
    over the residual stream.  Same parity bar as the default path (L/14 golden, fp32 reference, 1e-3 cosine on both weight sets'
    image embeddings) and rows stay bitwise batch-invariant across the three producers of the sums (skinny kernel at batch 1,
    stand-alone pass behind the non-persistent kernel at batch 40, persistent kernel's epilogue at batch 256)."""
    import subprocess
    import sys
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["OV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["OV_ROOT"], "tests"))
from openvision_amd import preset, synth
from openvision_amd.model import create_model
cfg = preset("vit-large-patch14-224")
for variant, gname in (("v1", "large14_224.npz"), ("sharp", "large14_224_sharp.npz")):
    g = np.load(os.path.join(os.environ["OV_ROOT"], "tests", "golden", gname))
    m = create_model(cfg, device="cuda:0", state_dict=synth.make_state_dict(cfg, 0, variant))
    img = torch.from_numpy(g["images"].astype(np.float32)).cuda()
    f = m.encode_image(img).cpu()
    c = (1 - torch.nn.functional.cosine_similarity(f, torch.from_numpy(g["image_features"].astype(np.float32)))).max().item()
    assert c < 1e-3, (variant, c)
    if variant == "sharp":
        big = synth.make_structured_images(256, 224, seed=9).cuda().to(torch.bfloat16)
        fb = m.encode_image(big)
        assert torch.equal(fb[7:8], m.encode_image(big[7:8])), "batch 1 vs 256"
        assert torch.equal(fb[:40], m.encode_image(big[:40])), "batch 40 vs 256"
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OVHIP_ROWPARTS="1", OV_ROOT=root), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-3000:]


def test_zero_shot_cli_from_checkpoint_directory(tmp_path):
    """SURVEY 8f row 1 + row a8, end to end and image-sensitive: weights -> checkpoint.save_pretrained (open_clip_config.json +
    open_clip_pytorch_model.bin, what ov-zero-shot-test.py:37-56 loads) -> `python -m openvision_amd.zero_shot` on the five testcat PNG
    files with nine real prompts (tokenised by the CLI's own tokenizer) -> its PRINTED tables against the reference's
    (tests/golden/tiny16_160_testcat_cli.npz, make_golden.gen_testcat_cli: the reference model on the same files, prompts tokenised by
    the `tokenizers` library): every cosine within gap / 2, each image's decided leading ranks (>= 3 per row, all five rows
    different), and the script's second table, best image per text (ov-zero-shot-test.py:198-208), whose nine winners cover all five
    images.  A swapped, repeated or input-independent image embedding fails both tables."""
    import subprocess
    import sys
    from openvision_amd import checkpoint
    g = golden("tiny16_160_testcat_cli.npz")
    gap = float(g["gap"])
    ref, names, prompts = g["cosine"], [str(n) for n in g["names"]], [str(t) for t in g["prompts"]]
    # the fixture's own properties first (a test that cannot fail proves nothing)
    ks = [_leading_ranks(r, gap) for r in ref]
    assert min(ks) >= 3 and len({tuple(g["argsort"][r][:3].tolist()) for r in range(5)}) >= 3
    assert len(set(g["best_image_per_text"].tolist())) >= 4
    col = np.sort(ref, axis=0)
    assert (col[-1] - col[-2]).min() >= gap and float(g["cos_err_ref_bf16"]) < gap / 2
    cfg = preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg, 0, "sharp")
    sd["visual.proj"] = torch.from_numpy(g["visual_proj"])
    sd["text_projection"] = torch.from_numpy(g["text_projection"])
    model = create_model(cfg, device=DEV, state_dict=sd)
    ckpt = str(tmp_path / "ckpt")
    checkpoint.save_pretrained(model, cfg, ckpt, safetensors=False, torch_bin=True)
    assert sorted(os.listdir(ckpt)) == ["open_clip_config.json", "open_clip_pytorch_model.bin"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "openvision_amd.zero_shot", "--use_model", ckpt, "--image_dir",
                        os.path.join(root, "tests", "golden", "testcat_160"), "--prompts", "|".join(prompts)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = p.stdout
    assert "Visual Config Used:" in out and "Pool type:             avg" in out
    table, best, cur, sect = {}, {}, None, None
    for line in out.splitlines():
        if line.startswith("=== "):
            sect = line
        elif line.startswith("--- ") and line.endswith(" ---"):
            cur = line[4:-4]
            table[cur] = []
        elif "cosine:" in line and cur is not None and "Cosine" in (sect or ""):
            lab, rest = line.split("cosine:")
            table[cur].append((lab.strip(), float(rest.split("prob:")[0])))
        elif " <- " in line and "Best Image" in (sect or ""):
            lab, rest = line.split(" <- ")
            best[lab.strip()] = rest.split("(")[0].strip()
    assert list(table) == names                                      # sorted directory listing, as the script
    for r, n in enumerate(names):
        assert len(table[n]) == len(prompts)
        got = {lab: c for lab, c in table[n]}
        for j, t in enumerate(prompts):
            assert abs(got[t] - ref[r, j]) < gap / 2, (n, t, got[t], ref[r, j])
        order = [prompts.index(lab) for lab, _ in table[n]]          # printed in descending cosine
        assert order[:ks[r]] == g["argsort"][r][:ks[r]].tolist(), (n, order, g["argsort"][r])
    assert {t: names[int(g["best_image_per_text"][j])] for j, t in enumerate(prompts)} == best


def _block_prefix_tokens(m, img, upto):
    """Residual stream of the HIP vision tower after block `upto` (the same launches encode_image makes, stopped early)."""
    from openvision_amd.model import _run_blocks
    tok = m.visual._embed_tokens(img.contiguous())
    return _run_blocks(list(m.visual.transformer.resblocks[: upto + 1]), tok).float().cpu().numpy()


# Budget of the bf16 residual stream against the reference's fp32 stream: the distance of the REFERENCE'S OWN bf16 mode
# (factory.py:275-296, run by make_golden.py on the same inputs and stored beside the fp32 slices) from its fp32 run.  On the sharp
# weights the per-token stream is ill-conditioned (peaked attention: rms error 0.29 of a stream rms 2.75 after 24 blocks in the
# reference's bf16 mode) while the pooled embedding is not.  The HIP path must be no further away than 1.3x that (+ 0.5 % of the
# stream rms), as rms over the slice, and within 2x of its worst element.
def _check_stream(got, want, ref_bf16, what):
    rms = float(np.sqrt((want.astype(np.float64) ** 2).mean()))
    e_hip, e_ref = np.abs(got - want), np.abs(ref_bf16 - want)
    r_hip, r_ref = float(np.sqrt((e_hip ** 2).mean())), float(np.sqrt((e_ref ** 2).mean()))
    assert r_hip < 1.3 * r_ref + 0.005 * rms and e_hip.max() < 2.0 * e_ref.max() + 0.02 * rms, \
        f"{what}: rms err {r_hip:.5f} (reference bf16 mode {r_ref:.5f}), max {e_hip.max():.4f} ({e_ref.max():.4f}), stream rms {rms:.3f}"
    return r_hip, r_ref


def _sharp_case(name, pname, blocks, text=True):
    cfg = preset(pname)
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, 0, "sharp"))
    g = golden(name)
    img = torch.from_numpy(g["images"].astype(np.float32)).to(DEV)
    n = img.shape[0]
    fi = m.encode_image(img)
    oc = one_minus_cos(fi, g["image_features"])
    assert oc < COS_TOL, f"{name}: image embeddings 1-cos {oc:.2e}"
    # the gate discriminates: every OTHER image's reference embedding is > 100x the tolerance away
    ni = torch.nn.functional.normalize(fi.cpu(), dim=-1)
    nr = torch.nn.functional.normalize(torch.from_numpy(g["image_features"]), dim=-1)
    cross = (ni @ nr.T).numpy()
    assert (1 - cross[~np.eye(n, dtype=bool)]).min() > 0.1
    assert np.array_equal(cross.argmax(axis=1), np.arange(n))
    for b in blocks:
        tokens = _block_prefix_tokens(m, img, b)
        for part, sl in (("head", slice(0, 4)), ("mid", slice(100, 102)), ("tail", slice(-2, None))):
            r = _check_stream(tokens[:, sl], g[f"block{b}_{part}"], g[f"block{b}_{part}_refbf16"], f"{name} block {b} {part}")
            print(f"{name} block {b} {part}: rms err HIP {r[0]:.5f} / reference bf16 mode {r[1]:.5f}")
    if text:
        tok = torch.from_numpy(g["tokens"]).to(DEV)
        ft = m.encode_text(tok)
        assert one_minus_cos(ft, g["text_features"]) < COS_TOL
        c = logits(m.encode_image(img, normalize=True), m.encode_text(tok, normalize=True)).cpu().numpy()
        assert np.abs(c - g["cosine"]).max() < TOPK_GAP / 2
        for r in range(n):
            k = _leading_ranks(g["cosine"][r], TOPK_GAP)
            assert np.array_equal(np.argsort(-c[r], kind="stable")[:k], g["argsort"][r][:k])
        if "loss" in g.files:
            a, b_, s_ = m(img, tok)
            assert abs(float(ClipLoss()(a, b_, s_)) - float(g["loss"])) < 0.05
    return m, g


def test_tiny_sharp_blocks_and_features():
    _sharp_case("tiny16_160_sharp.npz", "vit-tiny-patch16-160", [0, 5, 11])


@pytest.mark.timeout(900)
def test_large14_224_sharp_blocks_and_features():
    """Headline size on discriminating inputs: pooled embeddings AND the residual stream after blocks 0 / 11 / 23 against the
    reference's fp32 activations (the reference's own bf16 mode sits up to 8.6e-4 (1 - cos) from its fp32 mode on these weights)."""
    m, g = _sharp_case("large14_224_sharp.npz", "vit-large-patch14-224", [0, 11, 23])
    img = torch.from_numpy(g["images"].astype(np.float32)).to(DEV)
    assert one_minus_cos(m.encode_image(img), g["image_features_refbf16"]) < 2 * COS_TOL


@pytest.mark.timeout(900)
def test_small8_384_sharp_blocks_and_features():
    _sharp_case("small8_384_sharp.npz", "vit-small-patch8-384", [0, 11], text=False)


def test_batch_invariance_and_determinism(tiny):
    img = synth.make_images(9, 160, seed=5).to(DEV)
    a = tiny.encode_image(img)
    b = torch.cat([tiny.encode_image(img[:4]), tiny.encode_image(img[4:])])
    assert torch.equal(a, b)                                     # row results do not depend on batch composition
    assert torch.equal(a, tiny.encode_image(img))                # bitwise repeatable
    # batch 1 (the script's operating point, ov-zero-shot-test.py:167-181: M = 101 rows -> the skinny small-M GEMM kernel) against the
    # same image inside a batch of 9 (M = 909 rows -> the persistent 256 x 256 kernels): the kernels share the order of accumulation
    assert torch.equal(tiny.encode_image(img[3:4]), a[3:4])
    tok = synth.make_captions(9, seed=5).to(DEV)
    t9 = tiny.encode_text(tok)
    assert torch.equal(tiny.encode_text(tok[2:3]), t9[2:3]) and torch.equal(tiny.encode_text(tok[:4]), t9[:4])


@pytest.mark.timeout(900)
def test_large14_224_features():
    cfg = preset("vit-large-patch14-224")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg))
    g = golden("large14_224.npz")
    img = torch.from_numpy(g["images"].astype(np.float32)).to(DEV)
    tok = torch.from_numpy(g["tokens"]).to(DEV)
    fi, ft = m.encode_image(img), m.encode_text(tok)
    assert one_minus_cos(fi, g["image_features"]) < COS_TOL
    assert one_minus_cos(ft, g["text_features"]) < COS_TOL
    # same-precision comparison: against the reference's own bf16 mode
    assert one_minus_cos(fi, g["image_features_refbf16"]) < COS_TOL
    ni, nt, s = m(img, tok)
    assert abs(float(ClipLoss()(ni, nt, s)) - float(g["loss"])) < 0.05
    # full-size property checks (config #2 shape, B=256 would need the whole-batch oracle): batch invariance at B=32
    big = synth.make_images(48, 224, seed=3).to(DEV).to(torch.bfloat16)
    fb = m.encode_image(big)
    assert torch.equal(fb[:7], m.encode_image(big[:7]))
    assert torch.equal(fb[5:6], m.encode_image(big[5:6]))         # batch 1: M = 257 rows, the skinny kernel
    assert torch.isfinite(fb).all()
    # bitwise repeatability with several heads per persistent attention workgroup (48*16 heads > 256 CUs) and
    # persistent GEMM tiles: guards the LDS-DMA double-buffer hand-offs (a real race was caught this way)
    tb = synth.make_captions(48, seed=3).to(DEV)
    ft0 = m.encode_text(tb)
    for _ in range(4):
        assert torch.equal(fb, m.encode_image(big))
        assert torch.equal(ft0, m.encode_text(tb))


@pytest.mark.timeout(900)
def test_small8_384_long_sequence():
    cfg = preset("vit-small-patch8-384")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg))
    g = golden("small8_384.npz")
    fi = m.encode_image(torch.from_numpy(g["images"].astype(np.float32)).to(DEV))
    assert one_minus_cos(fi, g["image_features"]) < COS_TOL


@pytest.mark.timeout(900)
def test_full_size_batch256_properties():
    """BASELINE config #2/#3 size (ViT-L/14@224, per-GPU batch 256): size-independent properties instead of a CPU oracle
    run (the fp32 CPU path needs minutes per batch): batch-composition invariance across the persistent/tail-split
    scheduling, loss symmetry, and 'mean of the 8 ranks' local strip losses == the global loss' (SURVEY.md §3.3)."""
    cfg = preset("vit-large-patch14-224")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg))
    img = synth.make_images(256, 224, seed=9).to(DEV).to(torch.bfloat16)
    tok = synth.make_captions(256, seed=9).to(DEV)
    ni, nt, s = m(img, tok)
    assert torch.isfinite(ni).all() and torch.isfinite(nt).all()
    np.testing.assert_allclose(ni.norm(dim=-1).cpu().numpy(), 1.0, atol=1e-5)
    parts = torch.cat([m.encode_image(img[:100], normalize=True), m.encode_image(img[100:], normalize=True)])
    assert torch.equal(ni, parts)                        # tail split / persistent scheduling do not change any row
    assert torch.equal(nt, torch.cat([m.encode_text(tok[:37], normalize=True), m.encode_text(tok[37:], normalize=True)]))
    full = float(ClipLoss()(ni, nt, s))
    assert abs(full - float(ClipLoss()(nt, ni, s))) < 1e-6                     # symmetric in the two modalities
    from hipops import clip_loss
    b = 32
    local = [float(clip_loss(ni[r * b:(r + 1) * b].contiguous(), nt[r * b:(r + 1) * b].contiguous(), ni, nt, float(s), r * b)[0])
             for r in range(8)]
    assert abs(np.mean(local) - full) < 2e-5
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(0)).to(DEV)
    assert abs(float(ClipLoss()(ni[perm].contiguous(), nt[perm].contiguous(), s)) - full) < 2e-5   # pair order is irrelevant


@pytest.mark.parametrize("variant", ["v1", "sharp"])
def test_base16_224_against_oracle(variant):
    """A size the golden files do not hold (B/16@224: 197 tokens, 12 heads, E=512, text width 512): checked against the
    oracle run here on the CPU in fp32, on the benign and on the discriminating weight set."""
    from oracle import clip_ref as R
    cfg = preset("vit-base-patch16-224")
    sd = synth.make_state_dict(cfg, 0, variant)
    m = create_model(cfg, device=DEV, state_dict=sd)
    img = synth.make_images(3, 224, seed=2) if variant == "v1" else synth.make_structured_images(3, 224, seed=2)
    tok = synth.make_captions(3, seed=2)
    ni, nt, s = m(img.to(DEV), tok.to(DEV))
    ri, rt, rs = R.clip_forward(img, tok, sd, cfg)
    assert one_minus_cos(ni, ri) < COS_TOL and one_minus_cos(nt, rt) < COS_TOL
    assert abs(float(ClipLoss()(ni, nt, s)) - float(R.clip_loss(ri, rt, rs))) < 0.05


@pytest.mark.parametrize("variant", ["v1", "sharp"])
@pytest.mark.parametrize("width,head_width", [(320, 80), (576, 72)])
def test_head_dims_of_h14_and_so400m(width, head_width, variant):
    """head_dim 80 (OpenVision H/14) and 72 (So400m, with its 3.7362 MLP ratio -> padded hidden width) against the oracle."""
    from oracle import clip_ref as R
    cfg = preset("vit-tiny-patch16-160")
    cfg["vision_cfg"] = dict(cfg["vision_cfg"], width=width, head_width=head_width, layers=2, mlp_ratio=3.7362)
    cfg["text_cfg"] = dict(cfg["text_cfg"], width=width, heads=width // head_width, layers=2, mlp_ratio=3.7362)
    sd = synth.make_state_dict(cfg, 0, variant)
    m = create_model(cfg, device=DEV, state_dict=sd)
    img = synth.make_images(3, 160, seed=4) if variant == "v1" else synth.make_structured_images(3, 160, seed=4)
    tok = synth.make_captions(3, seed=4)
    ni, nt, s = m(img.to(DEV), tok.to(DEV))
    ri, rt, rs = R.clip_forward(img, tok, sd, cfg)
    assert one_minus_cos(ni, ri) < COS_TOL and one_minus_cos(nt, rt) < COS_TOL


def test_zero_shot_and_retrieval_evaluators(tiny):
    """SURVEY.md §8f row 3 on device: classifier weights, top-k (bit-exact indices on the SAME logits) and recall@k
    against the numpy oracle (oracle/eval_ref.py)."""
    from oracle import eval_ref as E
    from openvision_amd import evaluate as ev
    from openvision_amd.model import logits
    C_, T_ = 7, 3
    ctok = synth.make_captions(C_ * T_, seed=12).view(C_, T_, 80).to(DEV)
    w = ev.build_zero_shot_classifier(tiny, ctok, num_classes_per_batch=4)          # [E, C]
    emb = tiny.encode_text(ctok.view(-1, 80), normalize=True).cpu().numpy()
    np.testing.assert_allclose(w.cpu().numpy(), E.zero_shot_classifier(emb, C_, T_), atol=2e-6)
    img = synth.make_images(12, 160, seed=12).to(DEV)
    f = tiny.encode_image(img, normalize=True)
    lg = logits(f, w.T.contiguous(), 100.0)
    val, idx = ev.topk(lg, 5)
    ref_idx = np.argsort(-lg.cpu().numpy(), axis=1, kind="stable")[:, :5]
    assert np.array_equal(idx.cpu().numpy(), ref_idx)                                 # bit-exact index work
    labels = torch.from_numpy(ref_idx[:, 0].copy())
    labels[::3] = (labels[::3] + 1) % C_                                              # break every third label
    assert ev.count_correct(idx[:, 0].cpu(), labels) == E.count_correct(f.cpu().numpy(), w.T.cpu().numpy(), labels.numpy(),
                                                                         np.ones(12, bool))
    # retrieval: 12 images, 24 texts (2 per image)
    ttok = synth.make_captions(24, seed=13).to(DEV)
    te = tiny.encode_text(ttok, normalize=True)
    corr = [i // 2 for i in range(24)]
    got = ev.retrieval_recall(f, te, corr)
    sim = logits(f, te, 1.0).cpu().numpy()
    i2t, t2i = E.image_to_text_retrieval_eval(-sim, corr), E.text_to_image_retrieval_eval(-sim, corr)
    for k in (1, 5, 10):
        assert abs(got[f"img2txt/Recall@{k}"] - i2t[f"Recall@{k}"]) < 1e-7
        assert abs(got[f"txt2img/Recall@{k}"] - t2i[f"Recall@{k}"]) < 1e-7
    # ties: equal values rank by index; smallest-k order on distances
    x = torch.tensor([[1.0, 3.0, 3.0, 2.0, 3.0], [0.0, 0.0, 0.0, 0.0, 0.0]], device=DEV)
    assert ev.topk(x, 3)[1].tolist() == [[1, 2, 4], [0, 1, 2]]
    assert ev.topk(x, 2, largest=False)[1].tolist() == [[0, 3], [0, 1]]


def test_evaluators_against_reference_fixture():
    """The device evaluators (ov_logits + ov_topk + ov_class_mean_normalize) on the committed embeddings of tests/golden/eval.npz
    against the outputs of the REFERENCE's image_text_retrieval.py:24-87 / zero_shot_classifier.py:21-68 (make_golden.py gen_eval)."""
    from openvision_amd import evaluate as ev
    from openvision_amd import _lib
    from openvision_amd._lib import ptr, stream_ptr, check
    g = golden("eval.npz")
    zi, zt = torch.from_numpy(g["zimg"]).to(DEV), torch.from_numpy(g["ztxt"]).to(DEV)
    got = ev.retrieval_recall(zi, zt, list(g["corr"]))
    for n, k in enumerate(g["thresholds"]):
        assert abs(got[f"txt2img/Recall@{k}"] - g["t2i"][n]) < 1e-7, k
        assert abs(got[f"img2txt/Recall@{k}"] - g["i2t"][n]) < 1e-7, k
    emb = torch.from_numpy(g["zs_text_norm"]).to(DEV)
    C_, T_ = int(g["zs_classes"]), int(g["zs_templates"])
    out = torch.empty(C_, emb.shape[1], dtype=torch.float32, device=DEV)
    check(_lib.load().ov_class_mean_normalize(ptr(emb), ptr(out), C_, T_, emb.shape[1], stream_ptr()), "ov_class_mean_normalize")
    np.testing.assert_allclose(out.T.cpu().numpy(), g["zs_weights"], atol=2e-6)


def test_device_preprocess_bit_exact():
    """SURVEY.md §8f row 2 (image half): HIP resize + ToTensor + Normalize against Pillow's committed outputs (uint8 stage via the
    float output: x / 255 and (x - mean) / std are exact IEEE ops) and against the numpy oracle on odd geometries."""
    from oracle import preprocess_ref as P
    from openvision_amd import preprocess as pp
    from openvision_amd.config import DEFAULT_PREPROCESS as PPC
    g = golden("preprocess.npz")
    mean, std = PPC["mean"], PPC["std"]
    cats = [g[f"cat{i}_in"] for i in range(5)]
    out = pp.preprocess(cats, 160, mean, std, "squash", "bilinear", device=DEV).cpu().numpy()
    for i in range(2):
        assert np.array_equal(out[i], g[f"cat{i}_out"])                               # the script's transform, bit for bit
    for i in range(5):
        u8 = g[f"cat{i}_u8"].astype(np.float32) / np.float32(255.0)
        ref = ((u8 - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)).transpose(2, 0, 1)
        assert np.array_equal(out[i], ref)
    imgs = [g["noise_in"], g["grad_in"], g["up_in"]]
    got = pp.preprocess(imgs, 224, mean, std, "shortest", "bicubic", device=DEV).cpu().numpy()
    for k, o in zip(("noise", "grad", "up"), got):
        ref = ((g[f"{k}_u8_shortest_bicubic"].astype(np.float32) / np.float32(255.0) - np.asarray(mean, np.float32))
               / np.asarray(std, np.float32)).transpose(2, 0, 1)
        assert np.array_equal(o, ref)
    rng = np.random.default_rng(9)
    odd = [rng.integers(0, 256, s, dtype=np.uint8) for s in [(5, 7, 3), (333, 77, 3), (160, 160, 3), (1, 1, 3), (640, 481, 3)]]
    for mode, interp, size in (("squash", "bicubic", 160), ("shortest", "bilinear", 64)):
        got = pp.preprocess(odd, size, mean, std, mode, interp, device=DEV).cpu().numpy()
        for im, o in zip(odd, got):
            assert np.array_equal(o, P.transform(im, size, mean, std, mode, interp))
    b16 = pp.preprocess(cats[:1], 160, mean, std, dtype=torch.bfloat16, device=DEV)
    assert torch.equal(b16.float().cpu(), torch.from_numpy(out[:1]).to(torch.bfloat16).float())


FP8_COS_TOL = 5e-3     # ACHIEVED tolerance of the e4m3 path against the fp32 reference (1 - cos: image 1.2e-3, text 2.4e-3 on the v1
                        # weights).  It does NOT meet north_star's 1e-3 bar of the bf16 path; config #5 is reported at this tolerance.


@pytest.mark.timeout(900)
def test_fp8_e4m3_within_5e3_cosine_of_fp32_reference_large14():
    """BASELINE.json config #5 (1 GPU, small batch): fp8 e4m3 weights/activations in the four GEMMs of every block.  The reference
    has no fp8 mode; the check is against its fp32 outputs (golden) at the tolerance this e4m3 path ACHIEVES, stated in the test
    name: 1 - cos < 5e-3 per embedding (measured 1.2e-3 image / 2.4e-3 text) -- outside the 1e-3 bar the bf16 path meets.  The fp8
    run must also differ from the bf16 run (i.e. the fp8 kernels really ran)."""
    cfg = preset("vit-large-patch14-224")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg))
    g = golden("large14_224.npz")
    img = torch.from_numpy(g["images"].astype(np.float32)).to(DEV)
    tok = torch.from_numpy(g["tokens"]).to(DEV)
    f16, t16 = m.encode_image(img, normalize=True), m.encode_text(tok, normalize=True)
    m.set_precision("fp8")
    try:
        f8, t8 = m.encode_image(img, normalize=True), m.encode_text(tok, normalize=True)
    finally:
        m.set_precision("bf16")
    ref_i = torch.from_numpy(g["image_features"].astype(np.float32))
    ref_t = torch.from_numpy(g["text_features"].astype(np.float32))
    ci = torch.nn.functional.cosine_similarity(f8.cpu(), ref_i).min().item()
    ct = torch.nn.functional.cosine_similarity(t8.cpu(), ref_t).min().item()
    print(f"fp8 vs fp32 reference: min cosine image {ci:.5f} text {ct:.5f}")
    assert 1 - ci < FP8_COS_TOL and 1 - ct < FP8_COS_TOL
    assert not torch.equal(f8, f16) and not torch.equal(t8, t16)
    # static scales for the MLP hidden (fused c_fc -> c_proj hand-over), calibrated by the forwards above
    m.set_precision("fp8")
    try:
        m.encode_image(img); m.encode_text(tok)                           # calibration pass of the re-packed towers
        m.freeze_fp8_scales()
        f8s, t8s = m.encode_image(img, normalize=True), m.encode_text(tok, normalize=True)
        # delayed scaling: the producers keep recording maxima, the next forward rolls them into the scales
        tw = m.visual.transformer.tower()
        nl = tw.layers
        cur0 = tw.h_amax[: 2 * nl].clone()
        assert bool((tw.h_amax[2 * nl:] > 0).all())
        m.encode_image(img * 3.0)                                          # larger activations ...
        m.encode_image(img)                                                # ... are rolled in at the top of the following forward
        assert bool((tw.h_amax[: 2 * nl] >= cur0).all()) and bool((tw.h_amax[: 2 * nl] > cur0).any())
    finally:
        m.set_precision("bf16")
    cis = torch.nn.functional.cosine_similarity(f8s.cpu(), ref_i).min().item()
    cts = torch.nn.functional.cosine_similarity(t8s.cpu(), ref_t).min().item()
    print(f"fp8 static hidden scales vs fp32 reference: min cosine image {cis:.5f} text {cts:.5f}")
    assert 1 - cis < FP8_COS_TOL and 1 - cts < FP8_COS_TOL
    assert torch.equal(m.encode_image(img, normalize=True), f16)          # back on the bf16 path, bit for bit


@pytest.mark.timeout(900)
def test_fp8_config5_per_gpu_shape_b256_microbatches():
    """Config #5 at its per-GPU shape through the bench's own path: ViT-L/14@224, fp8 static (delayed) scales calibrated on warm-up
    steps, micro-batches of 256 pairs, ONE InfoNCE over all of them (bench.py --precision fp8 --micro-batches k; 16 micro-batches
    = the 4 096 pairs per GPU of train.sh:18; 2 here to bound the test's time).  Checked against the bf16 path on the SAME inputs
    ('sharp' weights + structured images, so embeddings of different pairs are far apart): the distribution of 1 - cos over the 512
    embeddings (bounds = what this build achieves, stated below), nearest neighbours preserved, the fp8 loss within 2 % of the bf16
    loss, batch-composition invariance under static scales."""
    cfg = preset("vit-large-patch14-224")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, 0, "sharp"))
    b, mb = 256, 2
    imgs = [synth.make_structured_images(b, 224, seed=60 + k).to(DEV).to(torch.bfloat16) for k in range(mb)]
    toks = [synth.make_captions(b, seed=60 + k).to(DEV) for k in range(mb)]

    def step():
        fi, ft = [], []
        for k in range(mb):
            ni, nt, s = m(imgs[k], toks[k])
            fi.append(ni)
            ft.append(nt)
        return torch.cat(fi), torch.cat(ft), s
    i16, t16, s = step()
    l16 = float(ClipLoss()(i16, t16, s))
    m.set_precision("fp8")
    try:
        id8, td8, _ = step()                                     # row-wise dynamic scales; records the maxima (calibration)
        m.freeze_fp8_scales()
        step()                                                   # delayed scaling settles
        i8, t8, s8 = step()
        l8 = float(ClipLoss()(i8, t8, s8))
        # delayed scales keep creeping (a changed scale re-rounds everything downstream, which moves later maxima): row results are
        # repeatable and batch-invariant only once the scales are FROZEN (serving mode)
        m.freeze_fp8_scales(delayed=False)
        full = m.encode_image(imgs[0], normalize=True)
        part = m.encode_image(imgs[0][:100], normalize=True)
        full_again = m.encode_image(imgs[0], normalize=True)
    finally:
        m.set_precision("bf16")
    stats = {}
    for name, a8, a16 in (("image static", i8, i16), ("text static", t8, t16), ("image dynamic", id8, i16), ("text dynamic", td8, t16)):
        d = (1 - torch.nn.functional.cosine_similarity(a8, a16)).float().cpu().numpy()
        stats[name] = (float(np.median(d)), float(np.quantile(d, 0.99)), float(d.max()))
        print(f"fp8 {name} scales vs bf16 at B=2x256 (sharp weights): 1-cos median {stats[name][0]:.2e} p99 {stats[name][1]:.2e} max {stats[name][2]:.2e}")
    print(f"loss fp8 {l8:.4f} bf16 {l16:.4f}")
    # The ALL-FOUR-GEMMS e4m3 path has no accuracy claim: it is the throughput end of config #5 (the recipe WITH a stated tolerance is
    # "fp8-mixed": test_fp8_mixed_recipe_within_its_stated_tolerance).  The numbers below are regression guards at 2x what the per-GEMM
    # ablation (profiles/r03_fp8_ablation.md, frozen scales, same weights) attributes to this mask: image median 1.5e-2 / max 5.8e-2,
    # text 2.3e-3 / 8.4e-3 against the bf16 path, of which the QKV product ALONE gives 1.2e-2 / 5.6e-2 -- the sharp weights scale q / k
    # by 2.5 for peaked attention, so the e4m3 rounding of the logits' operands moves the softmax; the MLP pair alone gives 3.8e-3 /
    # 1.6e-2.  (Round 2's comment here once said "median 4e-3" beside a measured 1.5e-2: 4e-3 is the MLP-only figure, 1.5e-2 the
    # all-four one.)  Row-wise dynamic scales give the same numbers: it is the 3-bit mantissa, not the scaling scheme.
    assert stats["image static"][0] < 3e-2 and stats["image static"][2] < 0.12
    assert stats["text static"][0] < 5e-3 and stats["text static"][2] < 0.02
    assert np.isfinite(l8) and abs(l8 - l16) < 0.02 * l16 + 0.02
    # nearest-neighbour structure survives: each fp8 image embedding is closest to its own bf16 embedding
    nn = (i8 @ i16.T).argmax(dim=1).cpu()
    assert torch.equal(nn, torch.arange(mb * b))
    assert torch.equal(part, full[:100]) and torch.equal(full, full_again)        # frozen scales: rows independent, bitwise repeatable
    assert (1 - torch.nn.functional.cosine_similarity(full, i16[:b])).max().item() < 0.12


# Tolerance of the "fp8-mixed" recipe, CHOSEN BEFORE the test ran (round 3; the per-GEMM ablation it was derived from is committed as
# profiles/r03_fp8_ablation.md): 1 - cos against the REFERENCE's fp32 outputs on the committed golden inputs,
#   * well-conditioned (v1) weights: <= 2e-3 for every embedding of both towers (2x north_star's bf16 bar);
#   * ill-conditioned ('sharp') weights: <= 2e-3 for the text tower, <= 1e-2 for the image tower.  The image figure is NOT the 2e-3 the
#     round-2 review asked for: on these weights the reference's own bf16 mode already sits 8.6e-4 from its fp32 mode, the e4m3 MLP
#     pair measures 5.8e-3, and the ablation prices every block's MLP pair at ~2e-4, so 2e-3 would leave e4m3 in ~5 of 24 blocks
#     (+4 % instead of +17 %).  1e-2 = an order of magnitude above the reference's bf16 mode is what this recipe is held to.
FP8_MIXED_TOL = {"v1": (2e-3, 2e-3), "sharp": (1e-2, 2e-3)}          # (image, text)


@pytest.mark.parametrize("variant,gname", [("v1", "large14_224.npz"), ("sharp", "large14_224_sharp.npz")])
def test_fp8_mixed_recipe_within_its_stated_tolerance(variant, gname):
    """BASELINE.json config #5 inside a stated tolerance: CLIP.set_precision("fp8-mixed") = e4m3 operands on c_fc and c_proj (the hidden
    handed over in e4m3 with a static scale), bf16 on QKV / out_proj (model.fp8_mixed_mask; +17 % img/s over bf16 at batch 256,
    profiles/r03_fp8_ablation.md).  With FROZEN scales (serving mode: what a tolerance can be stated for), against the reference's fp32
    outputs: FP8_MIXED_TOL; bitwise repeatable and batch-composition invariant; the e4m3 kernels really ran; back to bf16 bit for bit."""
    cfg = preset("vit-large-patch14-224")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, 0, variant))
    g = golden(gname)
    img = torch.from_numpy(g["images"].astype(np.float32)).to(DEV)
    tok = torch.from_numpy(g["tokens"]).to(DEV)
    ref_i = torch.from_numpy(g["image_features"].astype(np.float32))
    ref_t = torch.from_numpy(g["text_features"].astype(np.float32))
    f16, t16 = m.encode_image(img), m.encode_text(tok)
    cal_i = (synth.make_structured_images(64, 224, seed=5) if variant == "sharp" else synth.make_images(64, 224, seed=5)).to(DEV)
    cal_t = synth.make_captions(64, seed=5).to(DEV)
    m.set_precision("fp8-mixed")
    try:
        from openvision_amd.model import FP8_FC, FP8_PROJ
        assert m.visual.transformer.tower().mask == [FP8_FC | FP8_PROJ] * 24
        m.encode_image(cal_i); m.encode_text(cal_t)                          # calibration: maxima of the MLP hidden, per layer
        m.encode_image(img); m.encode_text(tok)
        m.freeze_fp8_scales(delayed=False)
        f8, t8 = m.encode_image(img), m.encode_text(tok)
        again = m.encode_image(img)
        both = m.encode_image(torch.cat([img, cal_i[:5].to(img.dtype)]))     # another batch composition
    finally:
        m.set_precision("bf16")
    ci = (1 - torch.nn.functional.cosine_similarity(f8.cpu(), ref_i)).max().item()
    ct = (1 - torch.nn.functional.cosine_similarity(t8.cpu(), ref_t)).max().item()
    print(f"fp8-mixed ({variant}) vs fp32 reference: worst 1-cos image {ci:.2e} text {ct:.2e}")
    tol_i, tol_t = FP8_MIXED_TOL[variant]
    assert ci <= tol_i and ct <= tol_t
    assert torch.equal(again, f8) and torch.equal(both[: img.shape[0]], f8)
    assert not torch.equal(f8, f16) and not torch.equal(t8, t16)
    assert torch.equal(m.encode_image(img), f16)


def test_checkpoint_dir_to_device(tiny, tmp_path):
    from openvision_amd import checkpoint as ck
    cfg = preset("vit-tiny-patch16-160")
    ck.save_pretrained(tiny, cfg, str(tmp_path), torch_bin=False)
    m2, _ = ck.from_pretrained(str(tmp_path), device=DEV)
    img = synth.make_images(2, 160, seed=1).to(DEV)
    assert torch.equal(m2.encode_image(img), tiny.encode_image(img))


def test_no_cpu_fallback(tiny):
    from openvision_amd._lib import OvhipError
    with pytest.raises(OvhipError):
        tiny.encode_image(torch.zeros(1, 3, 160, 160))
    with pytest.raises(OvhipError):
        ClipLoss()(torch.zeros(2, 192), torch.zeros(2, 192), 1.0)


def test_training_step_gradients_tiny(tiny):
    """One training step on the Tiny model: openvision_amd.training.clip_forward + ClipLoss + backward (HIP tower and loss nodes, torch
    autograd for the light ends) against torch autograd through the oracle's fp32 restatement of the reference on the CPU.  The tower
    works in bf16 (12 blocks, every intermediate rounded), so gradients are compared by direction and size: cosine >= 0.99 and norm
    within 5 % for every parameter tensor whose gradient is not negligible."""
    from oracle import clip_ref as R
    from openvision_amd import training
    from openvision_amd.loss import ClipLoss
    cfg = preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg)
    img, tok = synth.make_images(6, 160, seed=21), synth.make_captions(6, seed=21)
    # oracle: autograd through the restated forward
    sdg = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    fi, ft, sc = R.clip_forward(img, tok, sdg, cfg)
    ref_loss = R.clip_loss(fi, ft, sc)
    ref_loss.backward()
    # product
    tiny.zero_grad(set_to_none=True)
    for p in tiny.parameters():
        p.requires_grad_(True)
    gi, gt, gs = training.clip_forward(tiny, img.to(DEV), tok.to(DEV))
    loss = ClipLoss()(gi, gt, gs)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 2e-2
    loss.backward()
    got = dict(tiny.named_parameters())
    ref_scale = max(float(v.grad.norm()) for v in sdg.values() if v.grad is not None)
    checked = 0
    for name, ref in sdg.items():
        if ref.grad is None or name not in got:
            continue
        g, r = got[name].grad, ref.grad
        assert g is not None, name
        g = g.float().cpu()
        rn = float(r.norm())
        if rn < 1e-3 * ref_scale:
            continue
        cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-30))
        assert cos > 0.99, (name, cos)
        assert abs(float(g.norm()) - rn) < 0.05 * rn, (name, float(g.norm()), rn)
        checked += 1
    assert checked > 100                          # every block's weights, biases and LayerNorms of both towers + the ends
    # the inference path builds no graph and is unchanged by the training call
    with torch.no_grad():
        f = tiny.encode_image(img.to(DEV), normalize=True)
    assert not f.requires_grad and one_minus_cos(f.cpu(), fi.detach()) < 1e-3


def test_training_soft_token_entry_gradient(tiny):
    """ov-gradient-ascent.py:102-127,241-259: the text tower fed with `soft_one_hot @ token_embedding.weight`, differentiated w.r.t. the
    soft rows (weights frozen: the packed-weight cache of the training path is hit on the second call).  Reference = torch autograd
    through the oracle's block stack on the CPU."""
    from oracle import clip_ref as R
    from openvision_amd import training
    cfg = preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg)
    tcfg = cfg["text_cfg"]
    g = torch.Generator().manual_seed(3)
    V, T = tcfg["vocab_size"], tcfg["context_length"]
    ids = synth.make_captions(3, seed=5)
    soft0 = torch.nn.functional.one_hot(ids, V).float() * 0.9 + torch.rand(3, T, V, generator=g) * (0.1 / V)
    target = torch.nn.functional.normalize(torch.randn(3, cfg["embed_dim"], generator=g), dim=-1)
    # oracle
    sr = soft0.clone().requires_grad_(True)
    x = sr @ sd["token_embedding.weight"] + sd["positional_embedding"]
    x = R.block_stack(x, sd, "transformer.", tcfg["layers"], tcfg["heads"], True)
    f = R.layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])[:, -1] @ sd["text_projection"]
    ref_loss = -(torch.nn.functional.normalize(f, dim=-1) * target).sum(-1).mean()
    ref_loss.backward()
    # product (two calls: the second one must reuse the packed weights)
    for p in tiny.parameters():
        p.requires_grad_(False)
    try:
        for _ in range(2):
            sp = soft0.clone().to(DEV).requires_grad_(True)
            tf = training.encode_text(tiny, sp, normalize=True)
            loss = -(tf * target.to(DEV)).sum(-1).mean()
            loss.backward()
        (st,) = tiny.transformer._ovhip_train_state["chunks"].values()      # one autograd node per tower: one packed chunk
        sig0 = st["sig"]
        training.encode_text(tiny, sp.detach(), normalize=True)
        assert st["sig"] is sig0                                   # cache hit: nothing re-packed
    finally:
        for p in tiny.parameters():
            p.requires_grad_(True)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 2e-2
    gp, gr = sp.grad.float().cpu(), sr.grad
    cos = float((gp * gr).sum() / (gp.norm() * gr.norm()))
    assert cos > 0.98 and abs(float(gp.norm()) - float(gr.norm())) < 0.08 * float(gr.norm()), (cos, float(gp.norm()), float(gr.norm()))


def test_training_loss_decreases_tiny(tiny):
    """A few SGD steps on one fixed batch through the HIP forward/backward: the loss must go down steadily."""
    from openvision_amd import training
    from openvision_amd.loss import ClipLoss
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg))
    img, tok = synth.make_images(8, 160, seed=31).to(DEV), synth.make_captions(8, seed=31).to(DEV)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        loss = ClipLoss()(*training.clip_forward(m, img, tok))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.05, losses
    assert sum(b < a for a, b in zip(losses, losses[1:])) >= 6, losses


@pytest.mark.parametrize("name,size,B", [("vit-tiny-patch16-160", 160, 8), ("vit-large-patch14-224", 224, 12)])
def test_training_step_is_bitwise_repeatable(name, size, B):
    """No weight update between them: four training steps (forward keeping every intermediate incl. the attention's row lse, loss,
    backward) must give the SAME loss and the same gradient of every parameter, bit for bit.  Every kernel on the path is
    deterministic by construction (fixed-order reductions, no atomics); what this catches is a hand-counted wait that no longer
    covers a load -- the attention forward once produced run-to-run differences of 15 % in the gradient norm that way (a runtime
    branch between its stores and the wait on the prefetched Q rows)."""
    from openvision_amd import training
    from openvision_amd.loss import ClipLoss
    cfg = preset(name)
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, variant="sharp"))
    img, tok = synth.make_structured_images(B, size, seed=7).to(DEV), synth.make_captions(B, seed=7).to(DEV)
    first = None
    for it in range(4):
        m.zero_grad(set_to_none=True)
        loss = ClipLoss()(*training.clip_forward(m, img, tok))
        loss.backward()
        cur = [loss.detach().clone()] + [p.grad.detach().clone() for p in m.parameters()]
        if first is None:
            first = cur
        else:
            bad = [i for i, (a, b) in enumerate(zip(first, cur)) if not torch.equal(a, b)]
            assert not bad, (it, len(bad), bad[:5])


def test_chunked_tower_backward_is_bitwise_the_single_node_backward():
    """training.set_backward_chunk_layers(n): the towers as consecutive autograd nodes of n blocks (gradients become available
    chunk by chunk for the overlapped exchange) -- the same kernels on the same data: loss and every gradient bit for bit."""
    from openvision_amd import training
    from openvision_amd.loss import ClipLoss
    cfg = preset("vit-tiny-patch16-160")
    img, tok = synth.make_structured_images(8, 160, seed=9).to(DEV), synth.make_captions(8, seed=9).to(DEV)
    outs = []
    try:
        for chunk in (0, 5, 1):
            training.set_backward_chunk_layers(chunk)
            m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, variant="sharp"))
            loss = ClipLoss()(*training.clip_forward(m, img, tok))
            loss.backward()
            outs.append([loss.detach().clone()] + [p.grad.detach().clone() for p in m.parameters()])
    finally:
        training.set_backward_chunk_layers(0)
    for other in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], other))


def _ddp_rank(rank, ws, store, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from openvision_amd import training
    from openvision_amd.loss import ClipLoss
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=ws)
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg))
    img, tok = synth.make_images(8, 160, seed=41), synth.make_captions(8, seed=41)
    b = 8 // ws
    li, lt = img[rank * b:(rank + 1) * b].to(DEV), tok[rank * b:(rank + 1) * b].to(DEV)
    loss = ClipLoss(local_loss=True, gather_with_grad=True, rank=rank, world_size=ws)(*training.clip_forward(m, li, lt))
    loss.backward()
    out = {}
    for name, p in m.named_parameters():                      # what DistributedDataParallel does: average the ranks' gradients
        g = p.grad.detach().float().cpu()
        dist.all_reduce(g)
        out[name] = g / ws
    lt_ = loss.detach().float().cpu()
    dist.all_reduce(lt_)
    q.put((rank, float(lt_ / ws), {k: v.numpy() for k, v in out.items()} if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_matches_single_process():
    """The reference's training configuration (--local-loss --gather-with-grad, scripts/project/openvision/train.sh) on two ranks,
    gradients averaged as DDP does, against one process on the whole batch: same loss and the same gradients up to bf16 noise."""
    import tempfile
    import torch.multiprocessing as mp
    from openvision_amd import training
    from openvision_amd.loss import ClipLoss
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as d:
        q = ctx.Queue()
        ps = [ctx.Process(target=_ddp_rank, args=(r, 2, os.path.join(d, "store"), q)) for r in range(2)]
        [p.start() for p in ps]
        res = [q.get(timeout=600) for _ in range(2)]
        [p.join(60) for p in ps]
    dp_loss = res[0][1]
    dp_grads = next(r[2] for r in res if r[2] is not None)
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg))
    img, tok = synth.make_images(8, 160, seed=41).to(DEV), synth.make_captions(8, seed=41).to(DEV)
    loss = ClipLoss()(*training.clip_forward(m, img, tok))
    loss.backward()
    assert abs(float(loss.detach()) - dp_loss) < 2e-3
    scale = max(float(p.grad.norm()) for p in m.parameters())
    checked = 0
    for name, p in m.named_parameters():
        g, d_ = p.grad.float().cpu(), torch.from_numpy(dp_grads[name])
        if float(g.norm()) < 1e-3 * scale:
            continue
        cos = float((g * d_).sum() / (g.norm() * d_.norm() + 1e-30))
        assert cos > 0.995, (name, cos)
        assert abs(float(d_.norm()) - float(g.norm())) < 0.03 * float(g.norm()), name
        checked += 1
    assert checked > 100


def test_tokenizer_feeds_encode_text(tiny):
    """Captions -> openvision_amd.tokenizer -> encode_text: ids in range, embeddings equal to the oracle's on the same ids."""
    from oracle import clip_ref as R
    from openvision_amd.tokenizer import WordPieceTokenizer
    cfg = preset("vit-tiny-patch16-160")
    tok = WordPieceTokenizer(context_length=cfg["text_cfg"]["context_length"])
    ids = tok(["a photo of a cat", "a photo of a dog", "two cats sleeping on a pink couch next to remote controls", ""])
    assert int(ids.max()) < cfg["text_cfg"]["vocab_size"] and ids.shape == (4, 80)
    f = tiny.encode_text(ids.to(DEV), normalize=True).cpu()
    ref = R.encode_text(ids, synth.make_state_dict(cfg), cfg, True)
    assert one_minus_cos(f, ref) < 1e-3


def test_zero_shot_cli_with_text_prompts(tiny, tmp_path, capsys):
    """python -m openvision_amd.zero_shot with --prompts: config directory, images from disk, prompts tokenised here."""
    from PIL import Image
    from openvision_amd import checkpoint, zero_shot
    cfg = preset("vit-tiny-patch16-160")
    mdir, idir = tmp_path / "model", tmp_path / "images"
    checkpoint.save_pretrained(tiny, cfg, str(mdir))
    idir.mkdir()
    rng = np.random.default_rng(3)
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, size=(90 + 10 * i, 120, 3), dtype=np.uint8)).save(idir / f"img{i}.png")
    rc = zero_shot.main(["--use_model", str(mdir), "--image_dir", str(idir), "--prompts", "a photo of a cat|a photo of a dog|a remote control"])
    out = capsys.readouterr().out
    assert rc == 0 and "Best Image Per Text" in out and "a photo of a dog" in out and out.count("cosine:") == 9


def test_step_enqueues_without_host_synchronisation(tiny):
    """ABI 2: the logit scale reaches ov_clip_loss / ov_logits as a device scalar, so forward + InfoNCE enqueue without draining
    the stream (the reference never leaves the device either: loss.py:120-131).  torch raises on any synchronising call in
    sync-debug mode 'error'; the library itself holds no hipStreamSynchronize / blocking copy on this path."""
    img = synth.make_images(8, 160, seed=6).to(DEV)
    tok = synth.make_captions(8, seed=6).to(DEV)
    loss_fn = ClipLoss()
    want = float(loss_fn(*tiny(img, tok)))                     # warm-up: packs weights, sizes workspaces
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        ni, nt, s = tiny(img, tok)
        loss = loss_fn(ni, nt, s)
        li, lt = tiny.get_logits(img, tok)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert abs(float(loss) - want) < 1e-6
    assert torch.isfinite(li).all()


def _nccl_ws1_rank(store, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from openvision_amd.loss import ClipLoss, gather_features, _sum_over_ranks_own_chunk
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{store}", rank=0, world_size=1, device_id=torch.device(DEV))
    g = torch.Generator().manual_seed(5)
    img = torch.nn.functional.normalize(torch.randn(24, 192, generator=g), dim=-1).to(DEV)
    txt = torch.nn.functional.normalize(img.cpu() * 0.5 + torch.randn(24, 192, generator=g) * 0.08, dim=-1).to(DEV)
    s = torch.tensor(1 / 0.07, device=DEV)
    out = {}
    ai, at = gather_features(img, txt, True, True, 0, 1, force=True)          # RCCL all_gather_into_tensor, world 1
    out["gather_ok"] = bool(torch.equal(ai, img) and torch.equal(at, txt))
    full = torch.randn(24, 384, generator=g).to(DEV)
    out["rs_ok"] = bool(torch.equal(_sum_over_ranks_own_chunk(full.clone(), 24, 0), full))   # RCCL reduce_scatter_tensor, world 1
    for name, kw in (("plain", {}), ("coll_local", dict(local_loss=True, gather_with_grad=True)),
                     ("coll_global", dict(local_loss=False, gather_with_grad=True))):
        a, b, c = img.clone().requires_grad_(True), txt.clone().requires_grad_(True), s.clone().requires_grad_(True)
        fn = ClipLoss(rank=0, world_size=1, **kw)
        fn.always_collective = bool(kw)                                         # take the world_size > 1 code path at world 1
        loss = fn(a, b, c)
        loss.backward()
        out[name] = (float(loss.detach()), a.grad.cpu().numpy(), b.grad.cpu().numpy(), float(c.grad))
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _nccl_overlap_rank(store, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from openvision_amd import training
    from openvision_amd.loss import ClipLoss
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{store}", rank=0, world_size=1, device_id=torch.device(DEV))
    cfg = preset("vit-tiny-patch16-160")
    img, tok = synth.make_structured_images(8, 160, seed=11).to(DEV), synth.make_captions(8, seed=11).to(DEV)
    res = {}
    for mode in ("plain", "overlap"):
        m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, variant="sharp"))
        opt = training.FusedAdamW(m, lr=1e-3, bucket_bytes=4 << 20)
        if mode == "overlap":
            training.set_backward_chunk_layers(4)
            opt.overlap_gradient_exchange(1, always_collective=True)
        opt.zero_grad()
        loss = ClipLoss()(*training.clip_forward(m, img, tok))
        loss.backward()
        launched = sum(opt._ov["launched"]) if mode == "overlap" else 0
        nb = len(opt._ov_buckets) if mode == "overlap" else 0
        scale = opt.all_reduce_gradients(1)
        opt.step(grad_scale=scale)
        torch.cuda.synchronize()
        res[mode] = (float(loss.detach()), launched, nb, {n: p.detach().float().cpu().numpy() for n, p in m.named_parameters()})
        training.set_backward_chunk_layers(0)
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gradient_exchange_on_rccl_world_size_1():
    """FusedAdamW.overlap_gradient_exchange with the collectives forced in a world of one rank on the 'nccl' backend: the RCCL
    all-reduces are issued from autograd's hooks while the chunked backward is still running, on the one GPU there is.  Loss and the
    UPDATED parameters (one AdamW step) must equal those of the plain sequence bit for bit (a sum over one rank is the identity)."""
    import tempfile
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as d:
        q = ctx.Queue()
        p = ctx.Process(target=_nccl_overlap_rank, args=(os.path.join(d, "store"), q))
        p.start()
        out = q.get(timeout=600)
        p.join(120)
    l0, _, _, p0 = out["plain"]
    l1, launched, nb, p1 = out["overlap"]
    assert l0 == l1 and nb >= 2 and launched == nb
    for n in p0:
        np.testing.assert_array_equal(p0[n], p1[n], err_msg=n)


def test_rccl_branch_of_the_loss_at_world_size_1():
    """The RCCL ('nccl' backend) code path of the data-parallel loss -- all_gather_into_tensor in gather_features and
    reduce_scatter_tensor in the gathered-side gradient routing -- executed on the one GPU there is: a world of one rank, with
    ClipLoss forced through its world_size > 1 branch.  Loss and every gradient must equal the plain single-process ClipLoss.
    (N > 1 scaling is unmeasured: no multi-GPU node was available to this build.)"""
    import tempfile
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as d:
        q = ctx.Queue()
        p = ctx.Process(target=_nccl_ws1_rank, args=(os.path.join(d, "store"), q))
        p.start()
        out = q.get(timeout=600)
        p.join(120)
    assert out["gather_ok"] and out["rs_ok"]
    l0, gi0, gt0, gs0 = out["plain"]
    for name in ("coll_local", "coll_global"):
        l, gi, gt, gs = out[name]
        assert abs(l - l0) < 1e-6, name
        np.testing.assert_allclose(gi, gi0, atol=2e-7, err_msg=name)
        np.testing.assert_allclose(gt, gt0, atol=2e-7, err_msg=name)
        assert abs(gs - gs0) < 1e-6, name


def test_hipgraph_replay_of_small_batch_encode_is_bitwise(tiny):
    """CLIP.use_graphs: the batch-1 path of ov-zero-shot-test.py:167-195 as a captured hipGraph -- results bit for bit those of the
    plain launches, for new inputs (the graph is replayed, not re-captured) and again after a weight change (graphs dropped)."""
    img = synth.make_images(3, 160, seed=71).to(DEV)
    tok = synth.make_captions(5, seed=71).to(DEV)
    want_i = [tiny.encode_image(img[i:i + 1], normalize=True) for i in range(3)]
    want_t = tiny.encode_text(tok, normalize=True)
    tiny.use_graphs(8)
    try:
        got_i = [tiny.encode_image(img[i:i + 1], normalize=True) for i in range(3)]
        assert len(tiny._graphs.graphs) == 1                       # one capture, three replays
        got_t = tiny.encode_text(tok, normalize=True)
        for a, b in zip(want_i, got_i):
            assert torch.equal(a, b)
        assert torch.equal(want_t, got_t)
        assert torch.equal(tiny.encode_text(tok[:2], normalize=True), want_t[:2])     # another shape: another graph
        with torch.no_grad():
            tiny.visual.ln_post.bias.add_(0.25)
        changed = tiny.encode_image(img[0:1], normalize=True)      # weights changed: re-captured, not stale
        tiny.use_graphs(0)
        assert torch.equal(changed, tiny.encode_image(img[0:1], normalize=True)) and not torch.equal(changed, want_i[0])
    finally:
        tiny.use_graphs(0)
        with torch.no_grad():
            tiny.visual.ln_post.bias.sub_(0.25)


def test_hipgraph_survives_workspace_growth(tiny):
    """A captured graph bakes in the addresses of the grow-only workspaces; a later, larger call (graphed or not) reallocates them and
    the old buffer goes back to the allocator.  The graph cache carries the workspaces' generations in its signature: capture B = 1,
    run B = 8 (graphed) and B = 64 (plain), let somebody else take the freed memory, replay B = 1 -- bit for bit the plain result."""
    import gc
    img = synth.make_images(64, 160, seed=91).to(DEV)
    want1 = tiny.encode_image(img[0:1], normalize=True)
    want8 = tiny.encode_image(img[0:8], normalize=True)
    for m in (tiny, tiny.visual, tiny.visual.transformer, tiny.transformer):       # start from cold workspaces: B = 1 sizes them
        m._ws.buf = None
    gc.collect(); torch.cuda.empty_cache()
    tiny.use_graphs(8)
    try:
        assert torch.equal(tiny.encode_image(img[0:1], normalize=True), want1)
        gens = tiny._weights_sig()[1:5]
        assert torch.equal(tiny.encode_image(img[0:8], normalize=True), want8)          # a second graph; its warm-up grows the workspaces
        big = tiny.encode_image(img, normalize=True)                                    # above graph_max_batch: plain launches, grows again
        assert tiny._weights_sig()[1:5] != gens
        squatters = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]    # whoever gets the freed workspace memory
        assert torch.equal(tiny.encode_image(img[0:1], normalize=True), want1)
        assert torch.equal(tiny.encode_image(img[0:8], normalize=True), want8)
        assert all(bool(torch.isnan(t).all()) for t in squatters)                       # and nobody wrote through a stale pointer
        assert torch.equal(big[0:8], want8)
    finally:
        tiny.use_graphs(0)


def test_fused_adamw_matches_the_optax_restatement(tiny):
    """training.FusedAdamW (ov_adamw_step / ov_sumsq) over three steps with clipping and a 1/world_size gradient scale against
    oracle/optim_ref.py (numpy restatement of the reference trainer's optax chain, build_optax.py:272-278: parity unpinned vs optax
    itself, which is not installed).  Also: parameters stay the model's parameters (views of the flat buffers), the packed bf16
    copies follow the update, weight decay only touches the weight matrices."""
    from oracle import optim_ref as O
    from openvision_amd import training
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg))
    before = m.encode_image(synth.make_images(2, 160, seed=3).to(DEV))
    opt = training.FusedAdamW(m, lr=1e-3, wd=0.2, clip_norm=1.0)
    assert len(opt.groups) == 2 and opt.groups[0]["wd"] == 0.2 and opt.groups[1]["wd"] == 0.0
    names = [n for n, _ in opt.groups[0]["params"]]
    assert "visual.proj" in names and "visual.conv1.weight" in names and "visual.ln_post.weight" not in names and "logit_scale" not in names
    ref = [dict(p=g["flat"].cpu().numpy().copy(), mu=np.zeros(g["flat"].numel(), np.float32), nu=np.zeros(g["flat"].numel(), np.float32),
                wd=g["wd"]) for g in opt.groups]
    gen = torch.Generator().manual_seed(11)
    for step in range(1, 4):
        opt.zero_grad()
        grads = []
        for g in opt.groups:
            gr = torch.randn(g["grad"].numel(), generator=gen) * (0.05 * step)
            g["grad"].copy_(gr.to(DEV))
            grads.append(gr.numpy())
        scale = 0.5                                                  # as after a SUM all-reduce over two ranks
        gnorm = np.sqrt(sum((gr.astype(np.float64) ** 2).sum() for gr in grads)) * scale
        opt.step(lr=1e-3 * step, grad_scale=scale)
        for r, gr in zip(ref, grads):
            r["p"], r["mu"], r["nu"] = O.adamw_step(r["p"], gr, r["mu"], r["nu"], step, 1e-3 * step, 0.9, 0.95, 1e-8, r["wd"], scale, 1.0, gnorm)
    for g, r in zip(opt.groups, ref):
        # the clip factor comes from an fp32 device reduction (the oracle sums in fp64): it differs in the last bit, which flips the
        # bf16 rounding of the first moment for ~1e-4 of the elements (one bf16 ulp of mu = 2^-8 of that element's update)
        got_p = g["flat"].cpu().numpy()
        bad = ~np.isclose(got_p, r["p"], rtol=2e-5, atol=2e-6)
        assert bad.mean() < 1e-3 and np.abs(got_p - r["p"]).max() < 3e-3 * 2 ** -6, (bad.mean(), np.abs(got_p - r["p"]).max())
        np.testing.assert_allclose(g["nu"].cpu().numpy(), r["nu"], rtol=2e-5, atol=1e-12)
        mu = g["mu"].float().cpu().numpy()
        assert np.mean(mu != r["mu"]) < 1e-3 and np.abs(mu - r["mu"]).max() <= np.abs(r["mu"]).max() * 2 ** -7   # bf16: ulp flips only
    p = dict(m.named_parameters())["visual.proj"]
    assert p.data_ptr() >= opt.groups[0]["flat"].data_ptr() and p.is_cuda              # still the module's parameter, re-homed
    after = m.encode_image(synth.make_images(2, 160, seed=3).to(DEV))
    assert not torch.equal(before, after)                                              # packed copies were invalidated by step()
    # end to end: two optimiser steps on a real loss go down
    from openvision_amd.loss import ClipLoss
    img, tok = synth.make_images(8, 160, seed=31).to(DEV), synth.make_captions(8, seed=31).to(DEV)
    opt2 = training.FusedAdamW(m, lr=2e-4, wd=0.0)
    losses = []
    for _ in range(4):
        opt2.zero_grad()
        loss = ClipLoss()(*training.clip_forward(m, img, tok))
        loss.backward()
        opt2.step(grad_scale=opt2.all_reduce_gradients(1))
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
