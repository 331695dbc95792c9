"""CPU: host-side logic of the drop-in surface (config, state-dict contract, packing, error behaviour)."""
import json
import math

import numpy as np
import pytest
import torch

from openvision_amd import config as ovcfg, synth, preset
from openvision_amd.model import CLIP, create_model, _pack_matrix
from openvision_amd.loss import ClipLoss
from openvision_amd._lib import OvhipError


def test_presets_match_reference_size_table():
    c = preset("vit-large-patch14-224")
    assert c["embed_dim"] == 768 and c["vision_cfg"]["width"] == 1024 and c["vision_cfg"]["layers"] == 24
    assert c["text_cfg"]["width"] == 768 and c["text_cfg"]["heads"] == 12 and c["text_cfg"]["context_length"] == 80
    t = preset("vit-tiny-patch16-160")
    assert (t["vision_cfg"]["image_size"], t["vision_cfg"]["patch_size"], t["embed_dim"]) == (160, 16, 192)
    assert ovcfg.mlp_width(1152, 3.7362) == 4304                       # So400m (transfer_jax2hf.py:81)


def test_flop_model_matches_survey():
    f = synth.model_flops(preset("vit-large-patch14-224"))
    assert abs(f["image"] / 1e9 - 162.03) < 0.02 and abs(f["text"] / 1e9 - 13.83) < 0.01
    assert abs(f["pair"] / 1e9 - 175.85) < 0.02
    s = synth.model_flops(preset("vit-small-patch8-384"))
    assert abs(s["image"] / 1e9 - 196.16) < 0.05


def test_unsupported_configs_fail_loudly():
    c = preset("vit-tiny-patch16-160")
    with pytest.raises(ValueError):
        ovcfg.vision_cfg_from({**c["vision_cfg"], "attentional_pool": True})
    with pytest.raises(ValueError):
        ovcfg.text_cfg_from({**c["text_cfg"], "no_causal_mask": False})
    with pytest.raises(ValueError):
        ovcfg.vision_cfg_from({**c["vision_cfg"], "timm_model_name": "vit_base"})
    with pytest.raises(NotImplementedError):
        CLIP(192, c["vision_cfg"], c["text_cfg"], quick_gelu=True)


def test_state_dict_contract_and_attribute_tree():
    cfg = preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg)
    m = create_model(cfg, state_dict=sd)                      # strict load, as ov-zero-shot-test.py:54
    msd = m.state_dict()
    assert set(msd) == set(sd) and len(sd) == 300
    for k in sd:
        assert tuple(msd[k].shape) == tuple(sd[k].shape), k
        assert torch.equal(msd[k], sd[k])
    assert "visual.ln_pre.weight" not in msd and "visual.conv1.bias" not in msd and "attn_mask" not in msd
    v = m.visual                                               # attributes the scripts print / walk (:59-65, :105-153)
    assert v.pool_type == "avg" and v.final_ln_after_pool and v.attn_pool is None
    assert v.proj.shape == (192, 192) and v.positional_embedding.shape == (101, 192) and v.class_embedding.shape == (192,)
    assert isinstance(v.ln_pre, torch.nn.Identity) and isinstance(v.patch_dropout, torch.nn.Identity)
    assert v.image_size == (160, 160) and v.grid_size == (10, 10)
    assert m.transformer.get_cast_dtype() == torch.float32
    assert m.visual.transformer.resblocks[0].mlp.c_proj.in_features == 768      # ov-feature-visualization.py:151-155
    assert m.attn_mask is None and m.context_length == 80 and m.vocab_size == 32000 and m.text_pool_type == "last"
    assert abs(float(m.logit_scale) - math.log(1 / 0.07)) < 1e-6
    assert m.visual.transformer.resblocks[0].mlp.gelu.approximate == "none"
    assert m.transformer.resblocks[0].mlp.gelu.approximate == "tanh"
    m2 = m.float().eval()
    assert m2 is m
    with pytest.raises(RuntimeError):
        m.load_state_dict({**sd, "visual.ln_pre.weight": torch.ones(192)}, strict=True)


def test_no_cpu_fallback_anywhere():
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg)
    with pytest.raises(OvhipError):
        m.encode_image(torch.zeros(1, 3, 160, 160))
    with pytest.raises(OvhipError):
        m.encode_text(torch.zeros(1, 80, dtype=torch.long))
    with pytest.raises(OvhipError):
        m.visual.transformer(torch.zeros(1, 101, 192))
    with pytest.raises(OvhipError):
        m.visual.ln_post(torch.zeros(2, 192))
    with pytest.raises(OvhipError):
        m.visual.conv1(torch.zeros(1, 3, 160, 160))
    with pytest.raises(OvhipError):
        ClipLoss()(torch.zeros(2, 8), torch.zeros(2, 8), torch.tensor(1.0))
    with pytest.raises(NotImplementedError):
        ClipLoss(use_horovod=True)


def test_packing_pads_with_zeros():
    w = torch.arange(12, dtype=torch.float32).view(3, 4)
    p = _pack_matrix(w, 8, 64)
    assert p.shape == (8, 64) and p.dtype == torch.bfloat16
    assert torch.equal(p[:3, :4].float(), w) and float(p[3:].abs().sum()) == 0 and float(p[:, 4:].abs().sum()) == 0


def test_synth_is_deterministic_and_caption_format():
    a = synth.make_state_dict(preset("vit-tiny-patch16-160"), seed=0)["visual.conv1.weight"]
    b = synth.make_state_dict(preset("vit-tiny-patch16-160"), seed=0)["visual.conv1.weight"]
    assert torch.equal(a, b)
    t = synth.make_captions(16, 80, 32000, seed=3)
    assert t.dtype == torch.int64 and t.shape == (16, 80)
    assert (t[:, 0] == 1).all() and (t[:, 79] == 101).all()           # [bos ... eos pad.. cls@79] (bert_ops.py:496-507)
    for row in t:
        eos = (row == 2).nonzero()[0, 0].item()
        assert 5 <= eos <= 61 and (row[eos + 1:79] == 0).all() and (row[1:eos] >= 1000).all()
    pe = synth.posemb_sincos_2d(10, 10, 192)
    assert pe.shape == (101, 192) and float(pe[0].abs().sum()) == 0    # cls row zeros (transfer_jax2hf.py:108-109)


def test_config_dir_roundtrip(tmp_path):
    cfg = preset("vit-tiny-patch16-160")
    (tmp_path / "open_clip_config.json").write_text(json.dumps({"model_cfg": cfg, "preprocess_cfg": ovcfg.DEFAULT_PREPROCESS}))
    mc, pp = ovcfg.load_config_dir(str(tmp_path))
    assert mc == cfg and pp["mean"][0] == pytest.approx(0.48145466)


def test_invalidate_packed_after_dot_data_writes():
    """In-place writes through .data leave _version and data_ptr unchanged, so the packed-weight cache cannot see them:
    invalidate_packed() (module-level or CLIP.invalidate_packed) must force a rebuild; load_state_dict does it by itself."""
    from openvision_amd import model as M
    lin = M.Linear(64, 8)
    w0, _ = lin.packed()
    assert lin.packed()[0] is w0                               # cached
    lin.weight.data.mul_(2.0)                                   # invisible to (data_ptr, _version)
    assert lin.packed()[0] is w0                               # stale: this is the documented hazard
    M.invalidate_packed()
    w1, _ = lin.packed()
    assert w1 is not w0 and torch.equal(w1.float(), (w0.float() * 2).to(torch.bfloat16).float())
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg, state_dict=synth.make_state_dict(cfg))
    ln = m.visual.ln_post.packed()
    m.load_state_dict(synth.make_state_dict(cfg, seed=1))
    assert m.visual.ln_post.packed() is not ln
    m.invalidate_packed()


def test_optimizer_oracle_self_consistency_and_decay_filter():
    """oracle/optim_ref.py on a hand case (first step of Adam with bias correction: the direction is sign(g) / (1 + eps-ish)), bf16
    rounding of the first moment, clipping; and the decay mask of training.default_decay_filter on the real parameter names."""
    from oracle import optim_ref as O
    from openvision_amd import training
    p, g = np.array([1.0, -2.0, 0.5], np.float32), np.array([0.1, -0.2, 0.0], np.float32)
    p1, mu, nu = O.adamw_step(p, g, np.zeros(3, np.float32), np.zeros(3, np.float32), 1, 0.1, wd=0.0)
    np.testing.assert_allclose(p1, p - 0.1 * np.sign(g), atol=5e-3)          # step 1: mu_hat / sqrt(nu_hat) = sign(g) up to bf16(mu)
    assert np.all(O._bf16(mu) == mu)
    p2, _, _ = O.adamw_step(p, g, np.zeros(3, np.float32), np.zeros(3, np.float32), 1, 0.1, wd=0.5)
    np.testing.assert_allclose(p2 - p1, -0.1 * 0.5 * p, atol=1e-6)           # decoupled decay, scaled by lr
    _, mu_c, _ = O.adamw_step(p, g * 100, np.zeros(3, np.float32), np.zeros(3, np.float32), 1, 0.1, clip_norm=1.0)
    assert abs(np.linalg.norm(mu_c) / 0.1 - 1.0) < 1e-2                      # clipped to norm 1 before the moments
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg, state_dict=synth.make_state_dict(cfg))
    dec = {n for n, q in m.named_parameters() if training.default_decay_filter(n, q)}
    assert "visual.conv1.weight" in dec and "text_projection" in dec and "transformer.resblocks.0.mlp.c_fc.weight" in dec
    assert not ({"logit_scale", "token_embedding.weight", "positional_embedding", "visual.positional_embedding",
                 "visual.class_embedding", "ln_final.weight", "transformer.resblocks.0.mlp.c_fc.bias"} & dec)


def test_optimizer_update_is_formed_from_the_unrounded_first_moment():
    """optax.scale_by_adam forms the update from the fp32 moment and casts mu to mu_dtype only for storage: a gradient whose first
    moment is NOT representable in bf16 must move the parameter by the unrounded amount (oracle/optim_ref.py; the HIP kernel is
    held to this oracle in tests/test_gpu_model.py)."""
    from oracle import optim_ref as O
    g = np.array([1.2345678], np.float32)                                   # (1 - b1) * g has more than 8 significant bits
    z = np.zeros(1, np.float32)
    p1, mu, nu = O.adamw_step(z, g, z, z, 1, 1.0, b1=0.9, b2=0.95, eps=0.0)
    m32 = np.float32(1.0 - 0.9) * g
    assert O._bf16(m32)[0] != m32[0] and mu[0] == O._bf16(m32)[0]           # stored rounded ...
    want = -(m32 / np.float32(1.0 - 0.9)) / np.sqrt((np.float32(1.0 - 0.95) * (g * g)) / np.float32(1.0 - 0.95))
    assert p1[0] == np.float32(want[0])                                      # ... used unrounded
    assert nu[0] == (np.float32(1.0 - 0.95) * (g * g))[0]                    # (1 - b2) * (g * g), optax's grouping


def test_training_pool_keeps_what_a_step_needs_and_serves_by_purpose():
    """training._Pool: one free list per purpose, smallest fit, and as many retained buffers as were ever outstanding together (one
    saved-activation buffer per backward chunk), so chunked training re-uses all its buffers every step."""
    from openvision_amd.training import _Pool
    pool, dev = _Pool(), torch.device("cpu")
    saved = [pool.take(1000 * (i + 1), dev, "saved") for i in range(6)]      # six chunks alive at once
    ws = pool.take(100, dev)
    pool.give(ws)
    for t in saved:
        pool.give(t)
    assert len(pool.lists["saved"]) == 6 and len(pool.lists["ws"]) == 1
    ptrs = {t.data_ptr() for t in saved}
    again = [pool.take(1000 * (i + 1), dev, "saved") for i in range(6)]
    assert {t.data_ptr() for t in again} == ptrs                            # nothing reallocated
    assert [t.numel() for t in again] == [t.numel() for t in saved]         # smallest fit: each request got its own size back
    assert all(t._ovhip_gen == 1 for t in again)
    w2 = pool.take(50, dev)
    assert w2.data_ptr() == ws.data_ptr()                                   # a workspace request never takes a saved buffer
    pool.give(w2)
    pool.give(w2)                                                           # handing back twice is harmless
    assert len(pool.lists["ws"]) == 1


def test_fused_adamw_state_dict_hooks_and_none_gradients():
    """training.FusedAdamW host logic on CPU tensors: state_dict/load_state_dict round-trips step count and moments and refuses another
    layout; a second overlap_gradient_exchange replaces the hooks instead of doubling them; a parameter whose .grad was set to None
    contributes zeros, not last step's gradient."""
    from openvision_amd import training
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg, state_dict=synth.make_state_dict(cfg))
    opt = training.FusedAdamW(m, lr=1e-3, bucket_bytes=1 << 20)
    opt.t = 7
    for g in opt.groups:
        g["mu"].copy_(torch.randn(g["mu"].numel()).bfloat16())
        g["nu"].copy_(torch.rand(g["nu"].numel()))
    sd = opt.state_dict()
    m2 = create_model(cfg, state_dict=synth.make_state_dict(cfg))
    opt2 = training.FusedAdamW(m2, lr=1e-3)
    opt2.load_state_dict(sd)
    assert opt2.t == 7 and all(torch.equal(a["mu"], b["mu"]) and torch.equal(a["nu"], b["nu"]) for a, b in zip(opt.groups, opt2.groups))
    sd["groups"][0]["mu"].zero_()
    assert opt.groups[0]["mu"].abs().sum() > 0                               # the state dict holds copies
    bad = dict(sd, groups=[dict(sd["groups"][0], offs=[o + 4 for o in sd["groups"][0]["offs"]]), sd["groups"][1]])
    with pytest.raises(ValueError):
        opt2.load_state_dict(bad)
    # hooks: the second registration replaces the first
    opt.overlap_gradient_exchange(1)
    opt.overlap_gradient_exchange(1)
    n_params = sum(len(g["params"]) for g in opt.groups)
    assert len(opt._ov_hooks) == n_params
    opt.zero_grad()
    loss = sum((p * 2.0).sum() for p in m.parameters())
    loss.backward()
    assert all(x == 0 for x in opt._ov["pending"]) and all(opt._ov["launched"])   # each bucket armed exactly once per gradient
    assert opt.all_reduce_gradients(1) == 1.0
    assert all(bool((g["grad"][:16] == 2.0).all()) for g in opt.groups)
    # a .grad dropped behind the optimiser's back is a zero gradient
    name, p = opt.groups[0]["params"][0]
    p.grad = None
    opt._collect()
    assert p.grad is not None and float(p.grad.abs().sum()) == 0.0 and p.grad.data_ptr() == opt.groups[0]["grad"].data_ptr()
