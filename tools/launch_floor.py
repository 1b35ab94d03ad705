import torch, sys, os
sys.path.insert(0, "."); 
from representationlearning_amd import ops
x = torch.zeros(1024, device="cuda")
def ev(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("zero_(1024 floats) back-to-back: %.2f us/launch" % ev(lambda: ops.zero_(x)))
