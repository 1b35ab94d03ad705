"""Disassembles every gfx950 code object inside librssf.so and counts instruction patterns (default: the packed fp32 operand
selects of DESIGN.md lessons 23 and 59 - see `unsafe`).  No GPU needed.
  python tools/isa_scan.py [regex]"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "representationlearning_amd", "lib", "librssf.so")


def _selects(line):
    lo = re.search(r"op_sel:\[([01,]+)\]", line)
    hi = re.search(r"op_sel_hi:\[([01,]+)\]", line)
    lo = [int(c) for c in lo.group(1).split(",")] if lo else [0, 0, 0]
    hi = [int(c) for c in hi.group(1).split(",")] if hi else [1, 1, 1]
    return lo, hi


def crossed(line):
    """True for a packed fp32 instruction with a CROSSED operand: for some source i, op_sel[i] = 1 and op_sel_hi[i] = 0 - the low
    result reads the pair's HIGH register and the high result its LOW register (omitted fields default to op_sel 0, op_sel_hi 1).
    Broadcasts (op_sel[i] = op_sel_hi[i]) are not crossed."""
    if not re.search(r"\bv_pk_(mul|add|fma)_f32\b", line):
        return False
    lo, hi = _selects(line)
    return any(a == 1 and b == 0 for a, b in zip(lo, hi))


def unsafe(line):
    """True for the packed fp32 forms this library does not ship: the SECOND source's op_sel bit set (the low result reads src1's high
    register - crossed or broadcast: wrong low results beside MFMAs of another wave on gfx950, tools/pk_crossed_repro.hip), and - for
    good measure - any crossed source (the first / third source crossed measured clean, but only the vectoriser ever produces them)."""
    if not re.search(r"\bv_pk_(mul|add|fma)_f32\b", line):
        return False
    lo, _ = _selects(line)
    return (len(lo) > 1 and lo[1] == 1) or crossed(line)


def code_objects(lib=LIB):
    """Yields the disassembly text of each embedded gfx950 code object (one offload bundle per translation unit)."""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for i, s in enumerate(starts):
            part = os.path.join(d, "b%d.bin" % i)
            open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, "b%d.co" % i)
            r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            yield subprocess.run([LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def scan(pattern=None, lib=LIB):
    """[(kernel symbol, matching line)] over the whole library, and the number of code objects looked at.  pattern: a regular
    expression, or None for the packed fp32 forms the library does not ship (`unsafe`)."""
    match = unsafe if pattern is None else re.compile(pattern).search
    hits, n = [], 0
    for text in code_objects(lib):
        n += 1
        sym = "?"
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                sym = m.group(1)
            elif match(line):
                hits.append((sym, line.strip()))
    return hits, n


if __name__ == "__main__":
    hits, n = scan(sys.argv[1] if len(sys.argv) > 1 else None)
    print("%d code objects, %d matching instructions" % (n, len(hits)))
    per = {}
    for s, _ in hits:
        per[s] = per.get(s, 0) + 1
    for s, c in sorted(per.items(), key=lambda kv: -kv[1])[:20]:
        print("%6d  %s" % (c, s[:140]))
