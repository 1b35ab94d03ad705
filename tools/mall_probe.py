"""Does the 256 MB memory-side cache absorb a write -> read-back of a buffer that is REUSED?  x.add_(1) (read + write of the buffer) on the
same S-MB tensor 40 times against 40 different tensors of that size; GB/s by the bytes the kernel asks for.
   python tools/mall_probe.py"""
import torch
def run(tensors, reps):
    for t in tensors[:2]:
        t.add_(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        tensors[r % len(tensors)].add_(1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3
for mb in (8, 16, 32, 64, 128, 192, 256, 512):
    n = mb * (1 << 20) // 4
    many = [torch.zeros(n, device="cuda") for _ in range(max(2, 4096 // mb))]
    one = [many[0]]
    reps = 40
    t_one, t_many = run(one, reps), run(many, reps)
    gb = 2 * mb / 1024 * reps
    print("%4d MB buffer: reused %7.0f GB/s   streamed over %3d buffers %7.0f GB/s" % (mb, gb / t_one, len(many), gb / t_many))
    del many, one
