"""The hardware behaviour the library's build rule rests on (DESIGN.md lesson 59), checked on the box the tests run on."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
PROBE = os.path.join(os.path.dirname(__file__), "..", "tools", "bin", "pk_crossed_repro")


@pytest.mark.skipif(not os.path.exists(PROBE), reason="tools/bin/pk_crossed_repro not built (__graft_entry__.build())")
def test_packed_fp32_forms_the_library_ships_are_right_beside_mfmas():
    """tools/pk_crossed_repro.hip: 2 048 workgroups (two waves per SIMD) run packed fp32 instructions on integer-valued operands while other
    waves of the SIMD issue MFMAs, every thread checks its results against scalar arithmetic.  The forms librssf.so contains - natural
    selects, op_sel_hi-only broadcasts, the op_sel bit on the FIRST source - must count 0 wrong threads.  The forms it must NOT contain
    (tests/test_build_isa.py) - the SECOND source's op_sel bit, crossed or broadcast - are expected to fail on gfx950 (7-19 % of the threads
    on the round's boxes); a box where they pass is reported, not failed: the rule is then merely conservative there."""
    out = subprocess.run([PROBE, "20", "2048"], capture_output=True, text=True, timeout=300, check=True).stdout
    rows = [(l[:98].strip(), int(m.group(1))) for l in out.splitlines() for m in [re.search(r"\s(\d+) threads wrong of", l)] if m]
    assert len(rows) >= 30, out
    unsafe_words = ("second source crossed", "second source: high half", "op_sel:[0,1]", "op_sel:[0,1,0]")
    safe = [(w, n) for w, n in rows if not any(u in w for u in unsafe_words)]
    unsafe = [(w, n) for w, n in rows if any(u in w for u in unsafe_words)]
    assert len(safe) >= 12 and len(unsafe) >= 10, rows
    bad = [(w, n) for w, n in safe if n != 0]
    assert not bad, bad
    hit = [(w, n) for w, n in unsafe if n > 0]
    print("forms with the second source's op_sel bit set that went wrong on this box: %d of %d rows" % (len(hit), len(unsafe)))
    for w, n in hit[:4]:
        print("   %9d  %s" % (n, w))
