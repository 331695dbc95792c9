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
    # round 3: the fp8 GEMM (LDS-DMA + counted vmcnt like the bf16 kernel; it carried 120-140 B of scratch, reloaded INSIDE the K loop
    # behind vmcnt(0)) and the resident attention backward (no counted waits, but at its 168-register limit: kept at zero so that a
    # later asm load cannot land in a spilling kernel unnoticed)
    "gemm_fp8.hip": r"gemm_fp8_persist",
    "attention_bwd.hip": r"attn_bwd_hd64",
}


_ASM = {}


def device_asm(src):
    if src not in _ASM:
        _ASM[src] = subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S",
                                    "-o", "-", os.path.join(B.CSRC, src)], check=True, stdout=subprocess.PIPE,
                                   stderr=subprocess.DEVNULL, text=True).stdout
    return _ASM[src]


def kernel_scratch(src):
    out = device_asm(src)
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


# ---- second guard: an MFMA result read by INLINE ASM ---------------------------------------------------------------------------------
# hipcc's hazard recogniser puts the wait states an MFMA result needs before a VALU read (passes + 3: 11 for the 8-pass
# v_mfma_f32_32x32x16_bf16, 7 for the 4-pass 16x16x32) in front of its own instructions, not in front of inline asm.  The attention
# kernels take the row maximum of the score tile with an inline-asm v_max3_f32; where nothing else sits between the score MFMAs and
# that read, the maximum was taken from stale registers (round 2: last-bit differences from run to run).  Straight-line scan of the
# generated ISA: every VGPR an inline-asm VALU instruction reads must be at least that many issue slots behind the MFMA that wrote it.
_PASSES = {"32x32x16": 8, "16x16x32": 4, "32x32x8": 16, "16x16x16": 8, "16x16x128": 8, "32x32x64": 16}


def _regs(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def inline_asm_mfma_hazards(asm_text, kernel_re):
    bad = []
    for fm in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", asm_text, re.S | re.M):      # label (hipcc appends "; @name") .. s_endpgm
        name, body = fm.group(1), fm.group(2)
        if not re.search(kernel_re, name):
            continue
        pending = {}                 # vgpr -> wait states still required before a VALU may read it
        in_asm = False
        for line in body.split("\n"):
            t = line.split(";")[0].strip() if not line.strip().startswith(";;#") else line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(".") or t.startswith(";"):
                continue
            if t.endswith(":"):       # label: a join point, distances unknown -> keep the requirements (conservative)
                continue
            op, _, rest = t.partition(" ")
            toks = [x.strip() for x in rest.split(",")] if rest else []
            if in_asm and op.startswith("v_") and not op.startswith("v_mfma"):
                for tok in toks[1:]:
                    for r in _regs(tok):
                        if pending.get(r, 0) > 0:
                            bad.append((name, t, r, pending[r]))
            slots = 1
            if op == "s_nop":
                slots = int(toks[0]) + 1 if toks and toks[0].isdigit() else 1
            for r in list(pending):
                pending[r] -= slots
                if pending[r] <= 0:
                    del pending[r]
            if op.startswith("v_mfma") or op.startswith("v_smfma"):
                m = re.search(r"(\d+x\d+x\d+)", op)
                need = _PASSES.get(m.group(1) if m else "", 16) + 3
                for r in _regs(toks[0]) if toks else ():
                    pending[r] = need
    return bad


@pytest.mark.timeout(900)
def test_inline_asm_never_reads_an_mfma_result_too_early():
    asm = device_asm("attention.hip")
    assert len(re.findall(r"^_Z\w*attn_fwd\w*:", asm, re.M)) >= 6              # the scanner does see the kernels
    bad = inline_asm_mfma_hazards(asm, r"attn_fwd")
    assert not bad, f"inline-asm VALU reads of MFMA results inside the hazard window: {bad[:5]} ({len(bad)} in all)"


def test_the_hazard_scanner_sees_a_planted_hazard():
    planted = """_Zplanted:
	v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]
	v_add_f32_e32 v40, v41, v42
	;;#ASMSTART
	v_max3_f32 v50, v0, v1, v2
	;;#ASMEND
	s_nop 15
	;;#ASMSTART
	v_max3_f32 v51, v3, v4, v5
	;;#ASMEND
	s_endpgm
"""
    bad = inline_asm_mfma_hazards(planted, r"planted")
    assert len(bad) == 3 and all(b[1].startswith("v_max3_f32 v50") for b in bad), bad
