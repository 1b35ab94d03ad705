"""Micro-benchmark of the BatchNorm backward kernels (run on the GPU box)."""
import torch, sys
sys.path.insert(0, ".")   # run from the repository root: python tools/<script>.py
from representationlearning_amd import _lib as L
lib = L.load()
dev = 'cuda'
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, C, res in [(262144, 32, True), (262144, 32, False), (262144, 64, False), (262144, 256, True), (65536, 64, True), (16384, 128, True), (4096, 256, True)]:
    dy = torch.randn(rows, C, device=dev).bfloat16(); raw = torch.randn(rows, C, device=dev).bfloat16()
    rp = torch.randn(rows, C, device=dev).bfloat16() if res else None
    ss = torch.randn(2, C, device=dev); mi = torch.rand(2, C, device=dev) + 0.5
    sums = torch.zeros(8, 2, C, device=dev)
    draw = torch.empty_like(raw); dres = torch.empty_like(raw) if res else None
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    y = torch.empty_like(raw)
    st = L.stream()
    t_r = bench(lambda: lib.rssf_bn_bwd_reduce(L.ptr(dy), L.ptr(raw), L.ptr(ss), L.ptr(rp), L.ptr(sums), rows, C, 1, None, 1, st))
    t_a = bench(lambda: lib.rssf_bn_bwd_apply(L.ptr(dy), L.ptr(raw), L.ptr(ss), L.ptr(mi), L.ptr(sums), L.ptr(rp), L.ptr(draw), L.ptr(dres), L.ptr(dg), L.ptr(db), rows, C, 1, float(rows), 1, 1.0, 1, st))
    t_f = bench(lambda: lib.rssf_bn_apply(L.ptr(raw), L.ptr(ss), L.ptr(rp), None, L.ptr(y), rows, C, 1, 1, st))
    nb = rows * C * 2
    nr = 3 if res else 2
    print(f"rows={rows} C={C} res={res}: reduce {t_r:.1f} us ({nr*nb/t_r/1e6:.2f} TB/s)  bwd_apply {t_a:.1f} us ({(2*nr-1)*nb/t_a/1e6:.2f} TB/s)  apply {t_f:.1f} us ({nr*nb/t_f/1e6:.2f} TB/s)")
