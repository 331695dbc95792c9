"""CPU: bench.py's bookkeeping (no GPU): per-class algorithmic work adds up to the model FLOP count, host core detection."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["vit-large-patch14-224", "vit-small-patch8-384", "vit-tiny-patch16-160"])
def test_class_work_matches_model_flops(bench, name):
    from openvision_amd import preset, synth
    cfg = preset(name)
    b = 8
    work = bench.class_work(cfg, b)
    assert set(work) == set(bench.PROF_CLASSES)
    mfma = sum(v for k, (bound, v) in work.items() if bound == "mfma")
    total = synth.model_flops(cfg)["pair"] * b
    # the block stacks carry all but the patch embedding, the two projection heads and the logits (< 1 %)
    assert 0.98 * total < mfma <= total      # the rest: patch embedding, projection heads, logits (1.4 % on the tiny model)
    assert work["ln"][0] == "hbm" and work["ln"][1] > 0


def test_host_cores_is_sane(bench):
    n = bench.host_cores()
    assert 1 <= n <= 16
