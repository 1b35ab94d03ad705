import sys, torch, ctypes
sys.path.insert(0, ".")   # run from the repository root: python tools/<script>.py
from representationlearning_amd import _lib as L
lib = L.load()
def run(addr):
    a = torch.tensor(addr, dtype=torch.int32, device="cuda")
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    L.check(lib.rssf_debug_trread(L.ptr(a), L.ptr(out), L.stream()), "tr")
    torch.cuda.synchronize()
    return out.cpu().view(64, 4).tolist()
# pattern A: lane l -> elements 4l (contiguous 8-byte slots)
rA = run([4 * l for l in range(64)])
# pattern B: row-major [K=16 rows][16 cols] tile, row stride 16 elements: lane (l15, g) -> row 4g? addr = l15*16 + g*4  (row = l15, 4 elems at col 4g)
rB = run([(l & 15) * 16 + (l >> 4) * 4 for l in range(64)])
# pattern C: addr = (l>>4)*64 + (l&15)*4 : group g owns a 4x16 block at g*64; lane i within group -> row i/4?, ...
rC = run([(l >> 4) * 64 + (l & 15) * 4 for l in range(64)])
for name, r in (("A", rA), ("B", rB), ("C", rC)):
    print(name)
    for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 32, 48, 63):
        print("  lane", l, r[l])
