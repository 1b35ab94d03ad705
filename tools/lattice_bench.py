"""MlpDWBN's fused 17-tap convolution at the benchmark geometry (16 x 128 x 128 x 128 bf16), forward (bias + fused statistics) and data
gradient (fused BatchNorm-backward statistics): lattice kernel against the gather kernel, HIP events over back-to-back launches.
    python tools/lattice_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from representationlearning_amd import nnf

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C, H, W, dev = 128, 128, 128, "cuda"
torch.manual_seed(0)
convs = [nn.Conv2d(C, C, 1, 1).to(dev), nn.Conv2d(C, C, 3, 1, padding=6, dilation=6).to(dev), nn.Conv2d(C, C, 3, 1, padding=12, dilation=12).to(dev)]
spec = nnf.spec_of(convs)
weights = [c.weight.detach() for c in convs]
bias = sum(c.bias.detach() for c in convs).float().contiguous()
xs = [torch.randn(B, H, W, C, device=dev).bfloat16() for _ in range(6)]      # a ring of inputs: 6 x 67 MB (a step's operands miss the L2)
raw = torch.randn(B, H, W, C, device=dev).bfloat16()
link = nnf.BnBwdLink(); link.raw, link.rp, link.act, link.C = raw, None, 2, C
link.ss = torch.stack([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3]).contiguous()
flops = 2.0 * B * H * W * C * C * 17


def timed(fn, n=30):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for on in ("0", "1"):
    os.environ["RSSF_LATTICE"] = on
    st = torch.zeros(nnf.BN_SLOTS * 2 * C, device=dev)
    sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * C, device=dev)
    nnf._conv_forward(spec, xs[0], weights, bias, st)          # packs once outside the timing? (the pack is per call here: time it apart)
    tf = timed(lambda i: nnf._conv_forward(spec, xs[i % 6], weights, bias, st))
    td = timed(lambda i: nnf._conv_dgrad(spec, xs[i % 6], weights, (B, H, W, C), None, bn=(link, sm)))
    tp = timed(lambda i: nnf._pack(spec, weights, False, torch.bfloat16, dev))
    print("RSSF_LATTICE=%s  forward %.1f us  dgrad+bnbwd %.1f us  (each incl. the weight pack of %.1f us)  -> %.0f / %.0f TFLOP/s executed"
          % (on, tf, td, tp, flops / (tf - tp) / 1e6, flops / (td - tp) / 1e6), flush=True)
