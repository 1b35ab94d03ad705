"""Build checks on the generated gfx950 code (no GPU needed: hipcc cross-compiles)."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(__file__), "..", "representationlearning_amd", "csrc")


def _isa_scan():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import isa_scan
    return isa_scan


def test_crossed_operand_detector():
    """What tools/isa_scan.py calls crossed: a source whose op_sel bit is 1 and op_sel_hi bit 0 (omitted: op_sel 0, op_sel_hi 1)."""
    crossed = _isa_scan().crossed
    assert crossed("v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]")          # lesson 23 (gate_weights_bwd2_kernel)
    assert crossed("v_pk_add_f32 v[92:93], v[92:93], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]")    # lesson 59 (conv_wgrad_pw_kernel)
    assert crossed("v_pk_mul_f32 v[26:27], v[20:21], v[26:27] op_sel:[1,0] op_sel_hi:[0,1]")    # crossed first source
    assert crossed("v_pk_fma_f32 v[12:13], v[14:15], v[16:17], v[12:13] op_sel:[0,0,1] op_sel_hi:[1,1,0]")
    assert not crossed("v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0]")                   # broadcasts of one half: not crossed
    assert not crossed("v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0]")
    assert not crossed("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9]")
    assert not crossed("v_pk_add_u16 v2, v4, v6 op_sel:[0,1] op_sel_hi:[1,0]")
    # what the library must not contain: the second source's op_sel bit set (crossed or broadcast) - wrong low results beside MFMAs of
    # another wave (tools/pk_crossed_repro.hip) - and any crossed source
    unsafe = _isa_scan().unsafe
    assert unsafe("v_pk_add_f32 v[92:93], v[92:93], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]")
    assert unsafe("v_pk_mul_f32 v[76:77], v[76:77], v[164:165] op_sel:[0,1]")                   # (the fp32 attention kernels' form before round 6)
    assert unsafe("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel:[0,1,0]")
    assert unsafe("v_pk_mul_f32 v[26:27], v[20:21], v[26:27] op_sel:[1,0] op_sel_hi:[0,1]")     # crossed first source: clean in the probe, banned anyway
    assert not unsafe("v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0]")                       # first source's high half to both: measured clean
    assert not unsafe("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel_hi:[1,1,0]")
    assert not unsafe("v_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")   # the hand-written subtraction of win_attn_fwd.hip


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("unit", ["gate", "upsample", "win_attn_fwd"])
def test_units_built_without_slp_have_no_unsafe_packed_arithmetic(tmp_path, unit):
    """`v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (the SLP vectoriser's pairing of the 7x7 taps in gate_weights_bwd2_kernel)
    returned 0 in its low half for lanes 48..63 of a wave once other streams kept the CUs busy: one tap of the transposed
    convolution missing in a 16-pixel row of dpooled, in 5-15 % of the replays of a captured step (DESIGN.md lesson 23;
    tools/replay_race.py).  Round 6 found the second instance: `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` in conv_wgrad_pw_kernel's
    bias sums (the vectoriser's pairing of bf16 element 1 of one word with element 0 of the next), again the low half, 3-5 % of the
    replays at the test geometry, a fifth of the eager launches at the benchmark's (lesson 59; tools/replay_param_noise.py, tools/pw_race.py; the
    instruction alone beside MFMAs of another wave: tools/pk_crossed_repro.hip).  The instruction alone (tools/pk_crossed_repro.hip): every packed fp32 form whose SECOND
    source has its op_sel bit set - crossed or broadcast, in place or not - fails beside MFMAs of another wave of the SIMD; that includes the
    `v_pk_mul_f32 acc, acc, x op_sel:[0,1]` the vectoriser put into the fp32 attention kernels.  The units where the vectoriser
    produces such forms are built without it: this test compiles them with the Makefile's own command line and looks at the ISA."""
    unsafe = _isa_scan().unsafe
    out = subprocess.run(["make", "-n", "-B", "build/%s.o" % unit], cwd=CSRC, capture_output=True, text=True, check=True).stdout
    cmd = [l for l in out.splitlines() if unit + ".hip" in l and "hipcc" in l.split()[0]]
    assert cmd, out
    args = cmd[0].split()
    assert "-fno-slp-vectorize" in args, cmd[0]
    i = args.index("-c")
    asm = str(tmp_path / (unit + ".s"))
    args = args[:i] + ["-S", "--cuda-device-only", unit + ".hip", "-o", asm]
    subprocess.run(args, cwd=CSRC, check=True, capture_output=True)
    text = open(asm).read()
    assert {"gate": "gate_weights_bwd2_kernel", "upsample": "bilinear_fwd_kernel", "win_attn_fwd": "winattn_fwd_kernel"}[unit] in text
    bad = [l for l in text.splitlines() if unsafe(l)]
    assert not bad, bad[:3]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="no ROCm LLVM tools")
def test_built_library_has_no_unsafe_packed_arithmetic():
    """The same instruction forms anywhere in the built librssf.so (every embedded gfx950 code object is disassembled:
    tools/isa_scan.py): a kernel that grows one through a compiler choice is caught here, not by a drifting training run."""
    isa_scan = _isa_scan()
    if not os.path.exists(isa_scan.LIB):
        pytest.skip("librssf.so not built")
    hits, n = isa_scan.scan()
    assert n >= 15, n            # one code object per translation unit with kernels
    assert not hits, hits[:3]
