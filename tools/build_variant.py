"""Diagnostics: build libovhip_<TAG>.so with extra compile-time switches on some sources, for A/B runs of compile-time variants in
ONE gpurun session (select with OVHIP_LIB=libovhip_<TAG>.so).  The other objects are the default build's.

    python tools/build_variant.py TAG gemm.hip[,attention.hip] -DOVHIP_ST_LDS=2 ...
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvision_amd import build as B  # noqa: E402

tag, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build(verbose=False)
vdir = os.path.join(B.CSRC, "build", "var_" + tag)
os.makedirs(vdir, exist_ok=True)
objs = []
procs = []
for src in B.SOURCES:
    if src in srcs:
        obj = os.path.join(vdir, src + ".o")
        procs.append(subprocess.Popen([B.hipcc()] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj]))
    else:
        obj = os.path.join(B.CSRC, "build", src + ".o")
    objs.append(obj)
for p in procs:
    if p.wait():
        sys.exit(1)
out = os.path.join(B.PKG, f"libovhip_{tag}.so")
subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
