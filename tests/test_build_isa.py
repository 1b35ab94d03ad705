"""Build checks on the generated gfx950 code (no GPU needed: hipcc cross-compiles)."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(__file__), "..", "representationlearning_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_gate_kernels_have_no_cross_lane_packed_multiply(tmp_path):
    """`v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (the SLP vectoriser's pairing of the 7x7 taps in gate_weights_bwd2_kernel)
    returned 0 in its low half for lanes 48..63 of a wave once other streams kept the CUs busy: one tap of the transposed
    convolution missing in a 16-pixel row of dpooled, in 5-15 % of the replays of a captured step (DESIGN.md lesson 23;
    tools/replay_race.py).  gate.hip is therefore built without SLP vectorisation: this test compiles it with the Makefile's own
    command line and looks at the ISA."""
    out = subprocess.run(["make", "-n", "-B", "build/gate.o"], cwd=CSRC, capture_output=True, text=True, check=True).stdout
    cmd = [l for l in out.splitlines() if "gate.hip" in l and "hipcc" in l.split()[0]]
    assert cmd, out
    args = cmd[0].split()
    assert "-fno-slp-vectorize" in args, cmd[0]
    i = args.index("-c")
    asm = str(tmp_path / "gate.s")
    args = args[:i] + ["-S", "--cuda-device-only", "gate.hip", "-o", asm]
    subprocess.run(args, cwd=CSRC, check=True, capture_output=True)
    text = open(asm).read()
    assert "gate_weights_bwd2_kernel" in text
    bad = [l for l in text.splitlines() if re.search(r"v_pk_mul_f32.*op_sel:\[0,1\].*op_sel_hi:\[1,0\]", l)]
    assert not bad, bad[:3]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="no ROCm LLVM tools")
def test_built_library_has_no_cross_lane_packed_multiply():
    """The same instruction form anywhere in the built librssf.so (every embedded gfx950 code object is disassembled:
    tools/isa_scan.py): a kernel that grows it through a compiler choice is caught here, not by a drifting training run."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import isa_scan
    if not os.path.exists(isa_scan.LIB):
        pytest.skip("librssf.so not built")
    hits, n = isa_scan.scan()
    assert n >= 15, n            # one code object per translation unit with kernels
    assert not hits, hits[:3]
