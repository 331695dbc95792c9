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


def test_bench_launches_its_own_ranks_gloo_dry_run():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself, relay exactly one JSON line and report the
    world it really ran in (rehearsed on the CPU: gloo, no kernels).  The same launcher code path starts the RCCL ranks."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-gloo", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_ranks_seen"] == 2 and out["gather_rank_ordered"] is True and out["dry_run"] is True and out["steps"] == 3
    # what the first N > 1 run on hardware is diagnosed with: the exchange step's own time and every rank's step time
    comm = out["comm"]
    assert comm["calls_timed"] == 3 and comm["payload_bytes"] == 8 * 2 * 16 * 4 and comm["received_bytes_per_rank"] == 2 * comm["payload_bytes"]
    assert 0 < comm["allgather_ms"] <= comm["allgather_ms_max_rank"] <= comm["allgather_ms_worst_call"] * 1.0001
    rk = comm["rank_ms_per_step"]
    assert len(rk["per_rank"]) == 2 and rk["min"] == min(rk["per_rank"]) and rk["max"] == max(rk["per_rank"]) and rk["min"] > 0


def test_bench_launcher_propagates_rank_failure():
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # two real ranks in a container without a GPU: every rank fails at device selection -> the parent must exit non-zero, no JSON
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: the run succeeds")
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
