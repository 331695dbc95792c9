#!/usr/bin/env python3
"""bench.py — images/sec of the ViT-L/14@224 encode-and-contrast step on N x MI355X (BASELINE.json metric).

One step = image tower + text tower (bf16 compute, fp32 accumulate) on a per-GPU batch of synthetic
image/caption pairs already resident in HBM, the RCCL all-gather of the L2-normalised embeddings when
N > 1, and the fused InfoNCE on the local [b, N*b] logit strips.  Pure data parallelism, weak scaling
(per-GPU batch fixed).  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # launches its own N ranks (one process per GPU) and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # the same, under an external launcher
    python bench.py --gpus 2 --dry-run-gloo                # CPU rehearsal of the N-rank plumbing (gloo; no kernels, no throughput)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; must be set before the HIP runtime starts

PEAK_BF16_TFLOPS = 2500.0         # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0         # dense MX-scaled fp8 MFMA, same guide
PEAK_HBM_GBS = 8000.0             # HBM3E, same guide
PROF_CLASSES = {"ln": 0, "gemm_qkv": 1, "attention": 2, "gemm_out": 3, "gemm_fc": 4, "gemm_proj": 5, "gemm_fc_text": 6}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (images = captions)")
    ap.add_argument("--model", default="vit-large-patch14-224")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--breakdown", action="store_true", help="extra untimed step timing every kernel class")
    ap.add_argument("--micro-batches", type=int, default=1,
                    help="encode this many micro-batches of --batch pairs per step, one InfoNCE over all of them "
                         "(config #5: 16 x 256 per GPU = 32k pairs on 8 GPUs)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp8", "fp8-mixed"], help="GEMM operand precision of the block stacks")
    ap.add_argument("--fp8-mask", type=int, default=-1, help="with an fp8 precision: explicit OV_FP8_* bit set (1 QKV, 2 out_proj, 4 c_fc, 8 c_proj)")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (launcher, rendezvous, rank-ordered all-gather, barrier-fenced "
                         "timing, max over ranks) on the gloo backend: runs no kernel and reports no throughput")
    return ap.parse_args()


def launch_ranks(n: int) -> int:
    """`--gpus N` without a launcher around us: start N fresh ranks (one process per GPU) through torch.distributed.run and relay
    rank 0's single JSON line.  Runs BEFORE torch is imported: the parent never touches a device.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        sys.stderr.write(p.stdout)
        sys.stderr.write(f"bench.py: the {n}-rank run failed (exit code {p.returncode})\n")
        return p.returncode or 1
    print(lines[-1], flush=True)
    return 0


def dry_run_gloo(a, result_out) -> None:
    """The N-rank host path without a GPU: what a rank does around the kernels (rendezvous from the launcher's environment,
    contiguous batch shard, the packed rank-ordered all-gather of openvision_amd.loss.gather_features, barrier-fenced timing, max
    over ranks, one JSON line on rank 0).  No throughput is reported: nothing is computed."""
    import torch
    import torch.distributed as dist
    from openvision_amd.loss import gather_features, record_comm
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
    b, e = 8, 16
    g = torch.Generator().manual_seed(0)
    full_i, full_t = torch.randn(world * b, e, generator=g), torch.randn(world * b, e, generator=g)
    li, lt = full_i[rank * b:(rank + 1) * b].contiguous(), full_t[rank * b:(rank + 1) * b].contiguous()

    def fence():
        if world > 1:
            dist.barrier()
    ok = True
    for _ in range(a.warmup):
        gather_features(li, lt, True, False, rank, world)
    comm_log = []
    record_comm(comm_log if world > 1 else None)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ai, at = gather_features(li, lt, True, False, rank, world)
        ok = ok and bool(torch.equal(ai, full_i) and torch.equal(at, full_t))
    fence()
    record_comm(None)
    own_dt = time.perf_counter() - t0
    dt = torch.tensor([own_dt, 0.0 if ok else 1.0], dtype=torch.float64)
    comm = comm_summary(comm_log, own_dt, a.steps, li.shape[0] * 2 * e * 4, world, None)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        result_out.write(json.dumps({"metric": "dry run (gloo, CPU): multi-rank plumbing only", "value": None, "unit": None,
                                     "n_gpus": 0, "n_ranks_seen": dist.get_world_size() if world > 1 else 1,
                                     "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(float(dt[0]) / a.steps * 1e3, 3),
                                     "gather_rank_ordered": bool(dt[1] == 0), "scaling": "weak", "data": "synthetic",
                                     "comm": comm, "dry_run": True}) + "\n")
        result_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if dt[1] != 0:
        raise SystemExit("gathered rows are not in rank order")


def fc_kernel_name(rows: int, n: int) -> str:
    """The instantiation the library launches for the LN-folded erf-GELU c_fc GEMM (csrc/gemm.hip, launch<EPI>): outputs of at least
    OVHIP_GEMM_NT_MIN_MB (192) MB are streamed and take the LDS-transposed epilogue, smaller ones the direct one."""
    env = os.environ.get("OVHIP_GEMM_GELU_LDS")
    big = rows * n * 2 >= int(os.environ.get("OVHIP_GEMM_NT_MIN_MB", "192")) << 20
    lds = (env[0] == "1") if env else big
    return ("gemm_bf16_persist<1, true, false, false, false, false> (EPI erf-GELU, LN fold, LDS-transposed whole-line stores)" if lds
            else "gemm_bf16_persist<1, true, true, false, false, false> (EPI erf-GELU, LN fold, direct stores)")


def comm_summary(comm_log, own_dt, steps, payload_bytes, world, dev):
    """What a first multi-GPU run needs to be diagnosed (every rank calls this: it holds two small collectives): the exchange step's
    own time inside the timed region (one all-gather of the packed [b, 2E] fp32 embeddings per step: event pairs on the launch
    stream, recorded by gather_features) and every rank's step time before the max over ranks.  None in a world of one."""
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if comm_log and not isinstance(comm_log[0][0], float):
        ag = [e0.elapsed_time(e1) for e0, e1 in comm_log]                 # ms; the caller has synchronised
    else:
        ag = [(t1 - t0) * 1e3 for t0, t1 in comm_log]
    mine = torch.tensor([own_dt / steps * 1e3, sum(ag) / max(len(ag), 1), max(ag) if ag else 0.0], dtype=torch.float64,
                        device=dev if dev is not None else "cpu")
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    rows = torch.stack(allr).cpu()
    return {"collective": "all_gather_into_tensor(packed [b, 2E] fp32), one per step", "payload_bytes": int(payload_bytes),
            "received_bytes_per_rank": int(payload_bytes) * world, "calls_timed": len(ag),
            "allgather_ms": round(float(rows[:, 1].mean()), 4), "allgather_ms_max_rank": round(float(rows[:, 1].max()), 4),
            "allgather_ms_worst_call": round(float(rows[:, 2].max()), 4),
            "rank_ms_per_step": {"min": round(float(rows[:, 0].min()), 3), "max": round(float(rows[:, 0].max()), 3),
                                 "per_rank": [round(float(v), 3) for v in rows[:, 0]]},
            "note": "the all-gather time includes waiting for the slowest rank's towers (it is the step's only synchronisation point)"}


def cpu_model() -> str:
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def class_work(cfg, b):
    """Algorithmic work of one step per profiler class: FLOP for the MFMA-bound classes, HBM bytes for the
    row-statistics launches.  Rows of the tail image (side stream, overlapped) are left out, as in the class times."""
    v, t = cfg["vision_cfg"], cfg["text_cfg"]
    g = v["image_size"] // v["patch_size"]
    Lv, Dv, Dt, T = g * g + 1, v["width"], t["width"], t["context_length"]
    Fv, Ft = int(Dv * v["mlp_ratio"]), int(Dt * t.get("mlp_ratio", 4.0))
    vl, tl = v["layers"], t["layers"]
    hdv = v.get("head_width", 64)
    hv, ht = Dv // hdv, t["heads"]
    Mv, Mt = b * Lv, b * T
    return {
        "ln": ("hbm", 2.0 * (2 * vl * Mv * Dv + 2 * tl * Mt * Dt)),                       # each launch reads the bf16 stream once
        "gemm_qkv": ("mfma", 2.0 * (vl * Mv * 3 * Dv * Dv + tl * Mt * 3 * Dt * Dt)),
        "attention": ("mfma", 4.0 * (vl * b * hv * Lv * Lv * hdv + tl * b * ht * T * T * (Dt // ht))),
        "gemm_out": ("mfma", 2.0 * (vl * Mv * Dv * Dv + tl * Mt * Dt * Dt)),
        "gemm_fc": ("mfma", 2.0 * vl * Mv * Dv * Fv),
        "gemm_proj": ("mfma", 2.0 * (vl * Mv * Fv * Dv + tl * Mt * Ft * Dt)),
        "gemm_fc_text": ("mfma", 2.0 * tl * Mt * Dt * Ft),
    }


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota (the GPU box grants a
    1-GPU job a 16-CPU share of a 256-thread host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(cfg, sd, budget_s: float):
    """The oracle (CPU restatement of the reference path, plain torch fp32 aten kernels) on a bounded sample of the
    SAME workload (SURVEY.md section 8d protocol): image+text towers + InfoNCE at batch 8 (64 for the tiny model), 3 warm-up and
    >= 5 timed iterations, about budget_s seconds of work, all granted host cores."""
    import torch
    from openvision_amd import synth
    from oracle import clip_ref as R
    cores = host_cores()
    torch.set_num_threads(cores)
    b = 64 if cfg["vision_cfg"]["width"] < 256 else 8
    img = synth.make_images(b, cfg["vision_cfg"]["image_size"], seed=123)
    tok = synth.make_captions(b, cfg["text_cfg"]["context_length"], cfg["text_cfg"]["vocab_size"], seed=123)
    sdf = {k: v.float() for k, v in sd.items()}

    def step():
        with torch.no_grad():
            ni, nt, s = R.clip_forward(img, tok, sdf, cfg)
            return R.clip_loss(ni, nt, s)
    for _ in range(3):
        step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if (dt >= budget_s and n >= 5) or n >= 50:
            break
    return {"value": round(n * b / dt, 3), "unit": "images/sec", "cores": cores, "cpu": cpu_model(), "kind": "port",
            "sample": f"{n} iterations of batch {b} (image+text towers + InfoNCE, fp32 torch CPU ops via oracle/clip_ref.py), "
                      f"{dt:.1f} s after 3 warm-ups"}


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))           # parent of the ranks: never imports torch, never touches a device
    # stdout carries exactly ONE line, the JSON result: libraries that print to fd 1 (RCCL's banner when a communicator is
    # created, ...) are sent to stderr for the rest of the run; the result goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dry_run_gloo:
        return dry_run_gloo(a, result_out)
    import torch
    import torch.distributed as dist
    a.gpus = world                                        # under a launcher the world it made is authoritative
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)       # "nccl" IS RCCL on ROCm

    from openvision_amd import preset, synth, _lib
    from openvision_amd.model import create_model
    from openvision_amd.loss import ClipLoss, record_comm
    lib = _lib.load()
    _lib.check(lib.ov_device_check(), "ov_device_check")

    cfg = preset(a.model)
    sd = synth.make_state_dict(cfg, seed=0)
    model = create_model(cfg, device=dev, state_dict=sd)
    if a.precision != "bf16":
        model.set_precision(a.precision, a.fp8_mask if a.fp8_mask >= 0 else None)
    loss_fn = ClipLoss(local_loss=True, rank=rank, world_size=world)
    b = a.batch
    S = cfg["vision_cfg"]["image_size"]
    T, V = cfg["text_cfg"]["context_length"], cfg["text_cfg"]["vocab_size"]
    images = synth.make_images(b, S, seed=1000 + rank).to(dev).to(torch.bfloat16)      # resident in HBM
    tokens = synth.make_captions(b, T, V, seed=1000 + rank).to(dev)

    mb = max(1, a.micro_batches)

    def step():
        if mb == 1:
            ni, nt, s = model(images, tokens)
            return loss_fn(ni, nt, s)
        fi, ft = [], []
        for _ in range(mb):                    # the same resident micro-batch each time: synthetic data, full arithmetic
            ni, nt, s = model(images, tokens)
            fi.append(ni)
            ft.append(nt)
        return loss_fn(torch.cat(fi), torch.cat(ft), s)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = step()
    if a.precision != "bf16" and a.warmup > 0 and os.environ.get("OVHIP_FP8_DYNAMIC", "0") != "1":
        model.freeze_fp8_scales()              # static scales of the MLP hidden, calibrated on the warm-up steps
        loss = step()
    # in-situ timing of the dominant kernel (the vision MLP c_fc GEMM) during the timed region
    nrec = a.steps * max(1, a.micro_batches) * cfg["vision_cfg"]["layers"] + 8
    _lib.check(lib.ov_profile_enable(1 << PROF_CLASSES["gemm_fc"], nrec), "ov_profile_enable")
    comm_log = []
    record_comm(comm_log if world > 1 else None)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    record_comm(None)
    comm = comm_summary(comm_log, dt, a.steps, b * mb * 2 * cfg["embed_dim"] * 4, world, dev)
    tot, cnt, rows = C.c_double(0), C.c_int(0), C.c_double(0)
    _lib.check(lib.ov_profile_read(PROF_CLASSES["gemm_fc"], C.byref(tot), C.byref(cnt), C.byref(rows)), "ov_profile_read")
    _lib.check(lib.ov_profile_enable(0, 0), "ov_profile_enable")
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    model.check_token_range()
    loss_val = float(loss)

    # per-class breakdown (diagnostic, untimed)
    breakdown = None
    if a.breakdown and rank == 0:
        _lib.check(lib.ov_profile_enable(0x7f, 8 * (cfg["vision_cfg"]["layers"] + cfg["text_cfg"]["layers"]) + 8), "prof")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record()
        torch.cuda.synchronize()
        breakdown = {"step_ms": round(e0.elapsed_time(e1), 3)}
        work = class_work(cfg, b)
        for name, cid in PROF_CLASSES.items():
            t_, c_ = C.c_double(0), C.c_int(0)
            _lib.check(lib.ov_profile_read(cid, C.byref(t_), C.byref(c_), None), "prof read")
            ent = {"ms": round(t_.value, 3), "launches": c_.value}
            if t_.value > 0:
                bound, amount = work[name]
                if bound == "mfma":
                    ent.update(bound="mfma", achieved=round(amount / (t_.value * 1e-3) / 1e12, 1), unit="TFLOP/s",
                               frac=round(amount / (t_.value * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4))
                else:
                    ent.update(bound="hbm", achieved=round(amount / (t_.value * 1e-3) / 1e9, 1), unit="GB/s",
                               frac=round(amount / (t_.value * 1e-3) / 1e9 / PEAK_HBM_GBS, 4))
            breakdown[name] = ent
        _lib.check(lib.ov_profile_enable(0, 0), "prof off")

    if rank == 0:
        flops = synth.model_flops(cfg)
        vl, tl = cfg["vision_cfg"]["layers"], cfg["text_cfg"]["layers"]
        # dominant kernel: gemm_bf16_256x256<GELU> (c_fc).  Algorithmic FLOP per launch = 2*M*N*K.
        g = S // cfg["vision_cfg"]["patch_size"]
        Lv, Dv, Dt = g * g + 1, cfg["vision_cfg"]["width"], cfg["text_cfg"]["width"]
        launches = max(cnt.value, 1)
        # M of the timed launches comes from the library (the tower may peel tail images onto a side stream)
        flop_per_launch = 2.0 * (rows.value / launches) * Dv * int(Dv * cfg["vision_cfg"]["mlp_ratio"])
        avg_ms = tot.value / launches
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                if tj.get("model", "vit-large-patch14-224") == a.model and tj.get("batch", 256) == b:
                    traffic = tj.get("gemm_fc_bytes_per_launch")
            except Exception:
                traffic = None
        from openvision_amd.model import fp8_mixed_mask
        eff_mask = 0 if a.precision == "bf16" else (a.fp8_mask if a.fp8_mask >= 0 else (15 if a.precision == "fp8" else fp8_mixed_mask(1)[0]))
        fc_fp8 = bool(eff_mask & 4)                      # the roofline kernel (vision c_fc) runs on the fp8 MFMA
        peak_tf = PEAK_FP8_TFLOPS if fc_fp8 else PEAK_BF16_TFLOPS
        out = {
            "metric": "images/sec (node) ViT-L/14@224 fwd+InfoNCE" if a.model == "vit-large-patch14-224"
                      else f"images/sec (node) {a.model} fwd+InfoNCE",
            "value": round(world * b * mb * a.steps / dt, 2),
            "unit": "images/sec",
            "n_gpus": world, "n_ranks_seen": dist.get_world_size() if world > 1 else 1, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.model}: image tower + text tower (T={T}) + InfoNCE, per-GPU batch {b}, "
                                   f"formula weights, {'bf16' if a.precision == 'bf16' else 'fp8 e4m3 (MX-scaled) on GEMMs mask ' + str(eff_mask) + ' (1 QKV 2 out 4 c_fc 8 c_proj), bf16 elsewhere;'} MFMA / fp32 accumulate",
                       "fp8_mask": eff_mask,
                       "global_batch": world * b * mb, "micro_batches": mb, "parallelism": f"dp{world}",
                       "gflop_per_pair": round(flops["pair"] / 1e9, 2),
                       "model_tflops_per_gpu": round(flops["pair"] * b * mb * a.steps / dt / 1e12, 1)},
            "loss": round(loss_val, 5),
            "roofline": {"bound": "mfma", "kernel": (fc_kernel_name(int(rows.value / launches), int(Dv * cfg['vision_cfg']['mlp_ratio'])) if not fc_fp8
                                                     else "gemm_fp8_persist<1> (dequantise + bias + erf-GELU)")
                                   + f" = vision mlp.c_fc, N={int(Dv * cfg['vision_cfg']['mlp_ratio'])} K={Dv}, "
                                   f"M={int(rows.value / launches)} rows per launch",
                         "achieved": round(achieved, 1), "peak": peak_tf, "unit": "TFLOP/s",
                         "frac": round(achieved / peak_tf, 4), "traffic": traffic if a.precision == "bf16" else None,
                         "traffic_source": ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this "
                                            "command, NOT measured in this run") if (traffic is not None and a.precision == "bf16") else None,
                         "avg_launch_ms": round(avg_ms, 4), "launches": cnt.value,
                         "flop_per_launch": flop_per_launch},
        }
        out["comm"] = comm
        if breakdown:
            out["breakdown"] = breakdown
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
