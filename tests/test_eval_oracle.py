"""CPU: the evaluator oracle on hand-checkable cases + checkpoint directory round trip (host logic only)."""
import numpy as np
import torch

from oracle import eval_ref as E
from openvision_amd import preset, synth
from openvision_amd import checkpoint as ck
from openvision_amd.model import create_model


def test_retrieval_recall_hand_case():
    # 2 images, 4 texts (2 per image); distances chosen so image 0 ranks its texts 1st/3rd, image 1 ranks its texts 1st/2nd
    d = np.array([[0.1, 0.7, 0.5, 0.9],
                  [0.8, 0.6, 0.2, 0.3]])
    corr = [0, 0, 1, 1]
    r = E.image_to_text_retrieval_eval(d, corr)
    assert r["Recall@1"] == 1.0
    t = E.text_to_image_retrieval_eval(d, corr)
    # text 1 (belongs to image 0) is closer to image 1 (0.6 < 0.7) -> miss at k=1; the other three hit
    assert t["Recall@1"] == 0.75 and t["Recall@5"] == 1.0


def test_count_correct_multilabel_any():
    zimg = np.eye(3, dtype=np.float32)
    ztxt = np.eye(3, dtype=np.float32)[[2, 0, 1]]          # class j embedding
    labels = np.array([[1, -1], [2, 0], [1, -1]])          # predictions are classes 1, 2, 0: hit, hit (any-of), miss
    assert E.count_correct(zimg, ztxt, labels, np.array([True, True, True])) == 2
    assert E.count_correct(zimg, ztxt, labels, np.array([True, False, True])) == 1


def test_zero_shot_classifier_shapes_and_norm():
    g = np.random.default_rng(0)
    e = g.standard_normal((6 * 4, 16)).astype(np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    w = E.zero_shot_classifier(e, 6, 4)
    assert w.shape == (16, 6)
    np.testing.assert_allclose(np.linalg.norm(w, axis=0), 1.0, atol=1e-6)


def test_retrieval_recall_pinned_to_reference():
    """oracle.eval_ref vs tests/golden/eval.npz = outputs of the reference's image_text_retrieval.py:24-87 on the committed matrix."""
    from conftest import golden
    g = golden("eval.npz")
    t = E.text_to_image_retrieval_eval(g["dist"], list(g["corr"]))
    i = E.image_to_text_retrieval_eval(g["dist"], list(g["corr"]))
    for n, k in enumerate(g["thresholds"]):
        assert t[f"Recall@{k}"] == g["t2i"][n] and i[f"Recall@{k}"] == g["i2t"][n]
    assert 0.2 < g["t2i"][0] < 0.6 < g["t2i"][2] < 1.0          # the fixture discriminates (not all-hit / all-miss)


def test_zero_shot_classifier_pinned_to_reference():
    """oracle.eval_ref.zero_shot_classifier vs the reference's build_zero_shot_classifier (zero_shot_classifier.py:21-68)."""
    from conftest import golden
    g = golden("eval.npz")
    w = E.zero_shot_classifier(g["zs_text_norm"], int(g["zs_classes"]), int(g["zs_templates"]))
    np.testing.assert_allclose(w, g["zs_weights"], atol=1e-6)


def test_checkpoint_module_prefix_and_state_dict_wrapper(tmp_path):
    """factory.py:134-147: {'state_dict': ...} wrappers and DDP 'module.' prefixes are removed before the strict load."""
    cfg = preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg)
    m = create_model(cfg, state_dict=sd)
    ck.save_pretrained(m, cfg, str(tmp_path), safetensors=False)
    torch.save({"epoch": 3, "state_dict": {"module." + k: v for k, v in sd.items()}}, tmp_path / ck.BIN_NAME)
    m2, _ = ck.from_pretrained(str(tmp_path), device=None)
    for k, v in sd.items():
        assert torch.equal(m2.state_dict()[k], v), k


def test_checkpoint_dir_roundtrip(tmp_path):
    cfg = preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg)
    m = create_model(cfg, state_dict=sd)
    ck.save_pretrained(m, cfg, str(tmp_path))
    for fmt in ("safetensors", "bin"):
        d = tmp_path / fmt
        d.mkdir()
        (d / "open_clip_config.json").write_text((tmp_path / "open_clip_config.json").read_text())
        src = ck.SAFETENSORS_NAMES[0] if fmt == "safetensors" else ck.BIN_NAME
        (d / src).write_bytes((tmp_path / src).read_bytes())
        m2, pp = ck.from_pretrained(str(d), device=None)
        assert pp["mean"][0] == 0.48145466
        for k, v in sd.items():
            assert torch.equal(m2.state_dict()[k], v), k
    import pytest
    bad = dict(sd)
    bad.pop("visual.proj")
    torch.save(bad, tmp_path / "bin" / ck.BIN_NAME)
    with pytest.raises(RuntimeError):
        ck.from_pretrained(str(tmp_path / "bin"), device=None)      # strict load, like the reference
