"""BatchNorm passes at C = 32 / 262 144 rows with WARM operands (the same buffers every launch: they stay in the 256 MB Infinity
Cache) against COLD ones (a ring of buffer sets > 256 MB, what a training step sees), plus a plain bf16 copy as the streaming
yardstick.  Run on the GPU box from the repo root."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from representationlearning_amd import _lib as L
lib = L.load()
dev = "cuda"
rows, C = 262144, int(os.environ.get("C", 32))
NSET = int(os.environ.get("NSET", 12))            # 12 x (3 x 16.8 MB) = 600 MB

def bench(fn, n=48):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

sets = []
for _ in range(NSET):
    sets.append(dict(dy=torch.randn(rows, C, device=dev).bfloat16(), raw=torch.randn(rows, C, device=dev).bfloat16(),
                     out=torch.empty(rows, C, device=dev, dtype=torch.bfloat16)))
ss = torch.randn(2, C, device=dev); mi = torch.rand(2, C, device=dev) + 0.5
sums = torch.zeros(8, 2, C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
st = L.stream()
nb = rows * C * 2
for name, nset in (("warm", 1), ("cold", NSET)):
    S = sets[:nset]
    t_r = bench(lambda i: lib.rssf_bn_bwd_reduce(L.ptr(S[i % nset]["dy"]), L.ptr(S[i % nset]["raw"]), L.ptr(ss), None, L.ptr(sums), rows, C, 1, None, 1, st))
    t_a = bench(lambda i: lib.rssf_bn_bwd_apply(L.ptr(S[i % nset]["dy"]), L.ptr(S[i % nset]["raw"]), L.ptr(ss), L.ptr(mi), L.ptr(sums), None,
                                                L.ptr(S[i % nset]["out"]), None, L.ptr(dg), L.ptr(db), rows, C, 1, float(rows), 1, 1.0, 1, st))
    t_f = bench(lambda i: lib.rssf_bn_apply(L.ptr(S[i % nset]["raw"]), L.ptr(ss), None, None, L.ptr(S[i % nset]["out"]), rows, C, 1, 1, st))
    t_c = bench(lambda i: S[i % nset]["out"].copy_(S[i % nset]["raw"]))
    print("%s: reduce %.1f us (%.2f TB/s)  bwd_apply %.1f us (%.2f TB/s)  apply %.1f us (%.2f TB/s)  torch copy %.1f us (%.2f TB/s)"
          % (name, t_r, 2 * nb / t_r / 1e6, t_a, 3 * nb / t_a / 1e6, t_f, 2 * nb / t_f / 1e6, t_c, 2 * nb / t_c / 1e6), flush=True)
