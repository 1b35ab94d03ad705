"""Does a captured hipGraph run independent branches concurrently?  4 chains of 3x3 convolutions (HRNet branch shapes) captured
on one stream vs forked onto 4 streams (run from the repository root on the GPU box)."""
import sys, torch
sys.path.insert(0, ".")   # run from the repository root: python tools/<script>.py
from representationlearning_amd import nnf
dev = "cuda"
shapes = [(32, 128), (64, 64), (128, 32), (256, 16)]
convs = [torch.nn.Conv2d(c, c, 3, padding=1, bias=False).to(dev) for c, _ in shapes]
specs = [nnf.spec_of([c]) for c in convs]
xs = [torch.randn(16, s, s, c, device=dev).bfloat16() for c, s in shapes]
def chain(i, n=12):
    x = xs[i]
    for _ in range(n):
        x = nnf._conv_forward(specs[i], x, [convs[i].weight], None, None)
    return x
def run(par):
    cur = torch.cuda.current_stream()
    if not par:
        return [chain(i) for i in range(4)]
    ss = [torch.cuda.Stream() for _ in range(3)]
    outs = [None] * 4
    for i in range(1, 4):
        ss[i - 1].wait_stream(cur)
        with torch.cuda.stream(ss[i - 1]):
            outs[i] = chain(i)
    outs[0] = chain(0)
    for s in ss:
        cur.wait_stream(s)
    return outs
for par in (False, True):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(par)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = run(par)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("parallel" if par else "serial  ", "%.1f us per replay (48 conv launches + 48 packs)" % (e0.elapsed_time(e1) / 20 * 1e3))
