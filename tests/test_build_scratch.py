"""CPU (cross-compile only): kernels that fetch operands with inline-asm global loads / LDS-DMA and hand-counted `s_waitcnt vmcnt(N)`
must not use scratch.  Once the register allocator spills, a spill store can copy an asm load's destination to scratch before the
data has landed and a spill reload adds memory operations the hand counts do not know (found the hard way in
the streaming attention experiments).  hipcc --cuda-device-only -S writes the code-object metadata per kernel; the test fails if
`.private_segment_fixed_size` (bytes of scratch per lane) is non-zero for any such kernel."""
import os
import re
import subprocess

import pytest

from openvision_amd import build as B

# source file -> regex of the (mangled) kernel names that rely on asm loads + counted waits
GUARDED = {
    "gemm.hip": r"gemm_bf16_persist|gemm_bf16_pp",
    "attention.hip": r"attn_fwd_hd64_persist|attn_fwd_hd64_stream",
}


def kernel_scratch(src):
    out = subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S",
                          "-o", "-", os.path.join(B.CSRC, src)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                         text=True).stdout
    res = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)", out):
        res[m.group(1)] = int(m.group(2))
    return res


@pytest.mark.timeout(900)
@pytest.mark.parametrize("src", sorted(GUARDED))
def test_asm_load_kernels_use_no_scratch(src):
    scratch = kernel_scratch(src)
    guarded = {k: v for k, v in scratch.items() if re.search(GUARDED[src], k)}
    assert guarded, f"no guarded kernel found in {src}: {sorted(scratch)}"
    bad = {k: v for k, v in guarded.items() if v != 0}
    assert not bad, f"scratch in kernels with inline-asm loads: {bad}"
