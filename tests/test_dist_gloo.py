"""CPU, world_size 2 over gloo: the N>1 host path — rank-ordered all-gather of packed [b, 2E] embeddings
(openvision_amd.loss.gather_features), label offsets b*rank, and the data-parallel batch sharding bench.py uses.
The loss arithmetic itself is HIP-only (no CPU fallback), so the oracle stands in as the checker here."""
import os
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden


def _worker(rank, ws, store, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=ws)
    from openvision_amd.loss import gather_features, ClipLoss
    from oracle import clip_ref as R
    g = golden("cliploss_ws.npz")
    img, txt, s = torch.from_numpy(g["img"]), torch.from_numpy(g["txt"]), torch.from_numpy(g["scale"])
    b = img.shape[0] // ws
    li, lt = img[rank * b:(rank + 1) * b].contiguous(), txt[rank * b:(rank + 1) * b].contiguous()
    ai, at = gather_features(li, lt, local_loss=True, rank=rank, world_size=ws)
    ok_order = bool(torch.equal(ai, img) and torch.equal(at, txt))          # rank order == torch.cat(gathered)
    loss = float(R.clip_loss(li, lt, s, ai, at, rank))                        # labels i + b*rank
    try:                                                                      # product loss on CPU tensors must refuse
        ClipLoss(local_loss=True, rank=rank, world_size=ws)(li, lt, s)
        refused = False
    except RuntimeError:
        refused = True
    q.put((rank, ok_order, loss, refused))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_and_local_loss_world_size_2():
    ws = 2
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as d:
        q = ctx.Queue()
        ps = [ctx.Process(target=_worker, args=(r, ws, os.path.join(d, "store"), q)) for r in range(ws)]
        [p.start() for p in ps]
        res = sorted(q.get(timeout=300) for _ in range(ws))
        [p.join(60) for p in ps]
    g = golden("cliploss_ws.npz")
    for rank, ok_order, loss, refused in res:
        assert ok_order and refused
        assert abs(loss - float(g["local_losses_ws2"][rank])) < 1e-6       # == the reference's per-rank ClipLoss
    assert abs(np.mean([r[2] for r in res]) - float(g["loss_ws1"])) < 1e-6


def test_gather_requires_process_group():
    from openvision_amd.loss import gather_features
    import pytest
    with pytest.raises(RuntimeError):
        gather_features(torch.zeros(2, 8), torch.zeros(2, 8), world_size=2)
    a, b = gather_features(torch.ones(2, 8), torch.zeros(2, 8), world_size=1)
    assert a.shape == (2, 8)


def _grad_worker(rank, ws, store, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=ws)
    from openvision_amd import preset, synth, training
    from openvision_amd.model import create_model
    from openvision_amd._lib import OvhipError
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg, state_dict=synth.make_state_dict(cfg))
    opt = training.FusedAdamW(m, lr=1e-3, bucket_bytes=1 << 20)             # many buckets: 1 MiB each
    nb = sum(1 for _ in opt.buckets())
    g = torch.Generator().manual_seed(100 + rank)
    for p in m.parameters():
        p.grad.copy_(torch.randn(p.shape, generator=g))                     # in place: .grad stays a view of the flat buffer
    m.logit_scale.grad = torch.full((), 3.0 + rank)                         # replaced, not accumulated: _collect must copy it back
    scale = opt.all_reduce_gradients(ws)
    gsum = {n: p.grad.clone() for n, p in m.named_parameters()}
    try:
        opt.step()
        refused = False
    except OvhipError:
        refused = True
    q.put((rank, nb, scale, refused, {k: v.numpy() for k, v in gsum.items()} if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_all_reduce_world_size_2():
    """training.FusedAdamW.all_reduce_gradients over gloo at world_size 2: every bucket of the flat gradient buffers is summed across
    ranks (the reference averages with the 1/world_size factor, returned for the update kernel), a replaced .grad is folded back in,
    and the HIP update refuses CPU tensors."""
    ws = 2
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as d:
        q = ctx.Queue()
        ps = [ctx.Process(target=_grad_worker, args=(r, ws, os.path.join(d, "store"), q)) for r in range(ws)]
        [p.start() for p in ps]
        res = sorted((q.get(timeout=300) for _ in range(ws)), key=lambda r: r[0])
        [p.join(60) for p in ps]
    assert all(r[1] > 5 and r[2] == 0.5 and r[3] for r in res)
    got = res[0][4]
    from openvision_amd import preset, synth
    from openvision_amd.model import create_model
    m = create_model(preset("vit-tiny-patch16-160"))
    want = {}
    for rank in range(ws):
        g = torch.Generator().manual_seed(100 + rank)
        for n, p in m.named_parameters():
            want[n] = want.get(n, 0) + torch.randn(p.shape, generator=g)
    for n in want:
        ref = want[n].numpy() if n != "logit_scale" else np.float32(3.0 + 4.0)
        np.testing.assert_allclose(got[n], ref, rtol=1e-6, atol=1e-6, err_msg=n)


def _overlap_worker(rank, ws, store, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=ws)
    from openvision_amd import preset, synth, training
    from openvision_amd.model import create_model
    cfg = preset("vit-tiny-patch16-160")
    m = create_model(cfg, state_dict=synth.make_state_dict(cfg))
    opt = training.FusedAdamW(m, lr=1e-3, bucket_bytes=1 << 20)
    opt.overlap_gradient_exchange(ws)
    nb = len(opt._ov_buckets)
    res = []
    for it in range(2):                                                       # two backward passes: the buckets re-arm in zero_grad
        opt.zero_grad()
        g = torch.Generator().manual_seed(1000 * it + 100 + rank)
        coef = {n: torch.randn(p.shape, generator=g) for n, p in m.named_parameters()}
        loss = sum((p * coef[n]).sum() for n, p in m.named_parameters())      # d loss / d p = coef: every hook fires during backward
        loss.backward()
        launched = sum(opt._ov["launched"])                                   # buckets already in flight when backward returns
        scale = opt.all_reduce_gradients(ws)
        res.append((launched, scale, {n: p.grad.clone().numpy() for n, p in m.named_parameters()} if rank == 0 else None))
    q.put((rank, nb, res))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gradient_exchange_world_size_2():
    """FusedAdamW.overlap_gradient_exchange over gloo at world_size 2: every bucket's all-reduce is started from autograd's
    post-accumulate hooks (all of them are in flight when backward returns), the sums are those of the plain exchange, and the
    buckets re-arm for the next backward."""
    ws = 2
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as d:
        q = ctx.Queue()
        ps = [ctx.Process(target=_overlap_worker, args=(r, ws, os.path.join(d, "store"), q)) for r in range(ws)]
        [p.start() for p in ps]
        out = sorted((q.get(timeout=300) for _ in range(ws)), key=lambda r: r[0])
        [p.join(60) for p in ps]
    from openvision_amd import preset
    from openvision_amd.model import create_model
    m = create_model(preset("vit-tiny-patch16-160"))
    for rank, nb, res in out:
        assert nb > 5
        for launched, scale, _ in res:
            assert launched == nb and scale == 0.5
    for it in range(2):
        want = {}
        for rank in range(ws):
            g = torch.Generator().manual_seed(1000 * it + 100 + rank)
            for n, p in m.named_parameters():
                want[n] = want.get(n, 0) + torch.randn(p.shape, generator=g)
        got = out[0][2][it][2]
        for n in want:
            np.testing.assert_allclose(got[n], want[n].numpy(), rtol=1e-6, atol=1e-6, err_msg=f"{it} {n}")
