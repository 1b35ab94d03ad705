"""Instruction mix of one kernel from a gfx950 assembly listing (hipcc -S --cuda-device-only):
   python tools/isa_mix.py file.s <substring of the mangled kernel name> [<second substring> ...]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
keys = sys.argv[2:]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and l.split(":")[0].count("") and all(k in l.split(":")[0] for k in keys))
end = next(i for i in range(start, len(lines)) if lines[i].strip() == "s_endpgm")
body = [l.strip() for l in lines[start + 1:end + 1] if l.strip() and not l.strip().startswith((".", ";", "//"))]
c = collections.Counter()
for l in body:
    op = l.split()[0]
    k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "waitcnt" if op.startswith("s_waitcnt") else
         "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu" if op.startswith("s_") else "label" if op.endswith(":") else "other")
    c[k] += 1
print(lines[start].split(":")[0][:100])
print(dict(c), "total", len(body))
vc = collections.Counter(l.split()[0] for l in body if l.startswith("v_") and not l.startswith("v_mfma"))
print("VALU:", vc.most_common(22))
mc = collections.Counter(l.split()[0] for l in body if l.startswith(("v_mfma", "ds_", "global_", "buffer_", "scratch_")))
print("MFMA/LDS/VMEM:", mc.most_common(16))
