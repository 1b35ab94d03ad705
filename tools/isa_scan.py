"""Disassembles every gfx950 code object inside librssf.so and counts instruction patterns (default: the packed multiply with
crossed operand selects of DESIGN.md lesson 23).  No GPU needed.
  python tools/isa_scan.py [regex]"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "representationlearning_amd", "lib", "librssf.so")
CROSSED_PK_MUL = r"v_pk_mul_f32.*op_sel:\[0,1\].*op_sel_hi:\[1,0\]"


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


def scan(pattern=CROSSED_PK_MUL, lib=LIB):
    """[(kernel symbol, matching line)] over the whole library, and the number of code objects looked at."""
    rx, hits, n = re.compile(pattern), [], 0
    for text in code_objects(lib):
        n += 1
        sym = "?"
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                sym = m.group(1)
            elif rx.search(line):
                hits.append((sym, line.strip()))
    return hits, n


if __name__ == "__main__":
    hits, n = scan(sys.argv[1] if len(sys.argv) > 1 else CROSSED_PK_MUL)
    print("%d code objects, %d matching instructions" % (n, len(hits)))
    per = {}
    for s, _ in hits:
        per[s] = per.get(s, 0) + 1
    for s, c in sorted(per.items(), key=lambda kv: -kv[1])[:20]:
        print("%6d  %s" % (c, s[:140]))
