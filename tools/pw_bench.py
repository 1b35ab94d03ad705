"""MlpDWBN's fc1 forward (32 -> 128, bias + fused statistics) and fc2 data gradient (32 -> 128 with the fused BatchNorm-backward
statistics, GELU) at 16 x 128 x 128: the stream kernel (csrc/conv_pw.hip) against the gather kernel.  python tools/pw_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from representationlearning_amd import nnf
dev, B, H, W = "cuda", 16, 128, 128
torch.manual_seed(0)
fc1, fc2 = nn.Conv2d(32, 128, 1).to(dev), nn.Conv2d(128, 32, 1).to(dev)
s1, s2 = nnf.spec_of([fc1]), nnf.spec_of([fc2])
xs = [torch.randn(B, H, W, 32, device=dev).bfloat16() for _ in range(8)]
link = nnf.BnBwdLink(); link.raw, link.rp, link.act, link.C = torch.randn(B, H, W, 128, device=dev).bfloat16(), None, 2, 128
link.ss = torch.stack([torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.3]).contiguous()
b1 = fc1.bias.detach().float().contiguous()
def timed(fn, n=40):
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for on in ("0", "1"):
    os.environ["RSSF_PW"] = on
    st = torch.zeros(nnf.BN_SLOTS * 256, device=dev); sm = torch.zeros(nnf.BN_BWD_SLOTS * 256, device=dev)
    tp = timed(lambda i: nnf._pack(s1, [fc1.weight.detach()], False, torch.bfloat16, dev))
    tf = timed(lambda i: nnf._conv_forward(s1, xs[i % 8], [fc1.weight.detach()], b1, st))
    td = timed(lambda i: nnf._conv_dgrad(s2, xs[i % 8], [fc2.weight.detach()], (B, H, W, 128), None, bn=(link, sm)))
    ys = [torch.randn(B, H, W, 128, device=dev).bfloat16() for _ in range(4)]
    b2 = fc2.bias.detach().float().contiguous()
    st2 = torch.zeros(nnf.BN_SLOTS * 64, device=dev)
    tf2 = timed(lambda i: nnf._conv_forward(s2, ys[i % 4], [fc2.weight.detach()], b2, st2))
    td1 = timed(lambda i: nnf._conv_dgrad(s1, ys[i % 4], [fc1.weight.detach()], (B, H, W, 32), None))
    print("RSSF_PW=%s  fc1 forward %.1f us  fc2 dgrad+bnbwd %.1f us  fc2 forward %.1f us  fc1 dgrad %.1f us  (each incl. a weight pack of %.1f us)" % (on, tf, td, tf2, td1, tp), flush=True)
