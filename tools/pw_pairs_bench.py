"""The point-wise stream kernel against the gather kernel (RSSF_CONV_GENERIC) at the 1 x 1 convolutions of the fuse paths of a Base step
(B = 16; the map of a branch with C channels is 4096 / C on a side): forward with fused statistics and the data-gradient direction.
   python tools/pw_pairs_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from representationlearning_amd import nnf
dev = "cuda"
def t_us(fn, n=40):
    """us per call inside a replayed hipGraph of n calls (eager calls from Python are host-bound at ~15 us)"""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
pairs = [(256, 64, 128), (64, 256, 128), (64, 32), (128, 64), (256, 128), (128, 32), (256, 64), (256, 32), (32, 64), (64, 128), (128, 256), (32, 128), (64, 256), (32, 256), (64, 64)]
for pr in pairs:
    cin, cout = pr[0], pr[1]
    side = 4096 // max(cin, 32) if cin >= cout else 4096 // max(cout, 32)      # the LOW-resolution map of the pair (the larger channel count)
    if len(pr) > 2: side = pr[2]                                               # (layer1: 256 / 64 channels on the 128 x 128 map)
    B, H, W = 16, side, side
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev)
    spec = nnf.spec_of([conv])
    x = torch.randn(B, H, W, cin, device=dev).bfloat16()
    st = torch.zeros(nnf.BN_SLOTS * 2 * cout, device=dev)
    w = [conv.weight.detach()]
    row = []
    for generic in (True, False):
        row.append(t_us(lambda: nnf._conv_forward(spec, x, w, None, st, generic=generic)))
    print("%3d -> %3d at 16 x %3d x %3d: gather %6.1f us   stream %6.1f us   (%.1f MB)" % (cin, cout, H, W, row[0], row[1], B * H * W * (cin + cout) * 2 / 1e6), flush=True)
