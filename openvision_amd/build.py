"""Build libovhip.so (hipcc, gfx950 only) in-tree: ``python -m openvision_amd.build``.

hipcc cross-compiles without a GPU.  Objects go to ``openvision_amd/csrc/build/``; the library lands at
``openvision_amd/libovhip.so`` (git-ignored, but it travels with the gpurun snapshot)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "libovhip.so")
SOURCES = ["gemm.hip", "gemm_fp8.hip", "quant.hip", "layernorm.hip", "attention.hip", "attention_bwd.hip", "embed.hip", "loss.hip", "eval.hip", "preprocess.hip", "backward.hip", "optim.hip", "tower.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    deps = [os.path.join(CSRC, "common.h"), os.path.join(PKG, "..", "include", "ovhip.h")]
    cc = hipcc()

    def compile_one(src):
        sp = os.path.join(CSRC, src)
        obj = os.path.join(bdir, src + ".o")
        stamp = obj + ".sha"
        dg = _digest([sp] + deps)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg:
            return obj
        cmd = [cc] + FLAGS + ["-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stamp, "w") as f:
            f.write(dg)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
