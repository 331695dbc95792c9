"""Per-kernel register / scratch / LDS use of one csrc file, from hipcc's device asm (cross-compiles, no GPU needed).

    python tools/kernel_resources.py gemm.hip [regex] [-D...]      # prints one line per kernel whose demangled-ish name matches
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvision_amd import build as B  # noqa: E402


def resources(src, extra=()):
    out = subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S", "-o", "-",
                          *extra, os.path.join(B.CSRC, src)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    res = {}
    for blk in out.split("  - .agpr_count:")[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        res[g("name")] = dict(agpr=blk.split("\n")[0].strip(), vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                              lds=g("group_segment_fixed_size"), spill=g("vgpr_spill_count"))
    return res, out


if __name__ == "__main__":
    src = sys.argv[1]
    pat = next((a for a in sys.argv[2:] if not a.startswith("-")), ".")
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    res, _ = resources(src, extra)
    for name, r in sorted(res.items()):
        if re.search(pat, name):
            print(f"{name[:110]:<110} vgpr {r['vgpr']:>3} agpr {r['agpr']:>3} sgpr {r['sgpr']:>3} scratch {r['scratch']:>4} lds {r['lds']:>6}")
