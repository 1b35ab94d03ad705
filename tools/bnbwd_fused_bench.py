"""Data gradient of a 3x3 convolution followed by the BatchNorm-backward statistics of the previous layer: two launches
(rssf_conv_gather_add + rssf_bn_bwd_reduce) against one (rssf_conv_gather_bnbwd), and the difference of the two statistics.
C, H, B, ACT, RES via the environment; run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from representationlearning_amd import _lib as L, nnf
import bench
C, H, B = int(os.environ.get("C", 32)), int(os.environ.get("H", 128)), int(os.environ.get("B", 16))
K, CO = int(os.environ.get("K", 3)), int(os.environ.get("CO", 0)) or C      # kernel size; channels of the convolution OUTPUT (dout)
act, res = int(os.environ.get("ACT", 1)), os.environ.get("RES", "0") == "1"
lib = L.load()
torch.manual_seed(0)
conv = torch.nn.Conv2d(C, CO, K, padding=K // 2, bias=False).cuda()
spec = nnf.spec_of([conv])
dout = torch.randn(B, H, H, CO, device="cuda").bfloat16()
raw = torch.randn(B, H, H, C, device="cuda").bfloat16()
rp = torch.randn(B, H, H, C, device="cuda").bfloat16() if res else None
ss = torch.cat([torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3]).contiguous()
wpk = nnf._pack(spec, [conv.weight], True, dout.dtype, dout.device)
dx1, dx2 = torch.empty_like(raw), torch.empty_like(raw)
s1 = torch.zeros(nnf.BN_BWD_SLOTS * 2 * C, device="cuda"); s2 = torch.zeros_like(s1)
rows = B * H * H

def two():
    L.check(lib.rssf_conv_gather_add(L.ptr(dout), L.ptr(wpk), L.ptr(dx1), None, None, None, None, B, H, H, CO, H, H, C, 1, 1, spec.ntaps,
                                     spec.c_ndy, spec.c_ndx, L.dtype_code(dout), L.stream()), "dgrad")
    L.check(lib.rssf_bn_bwd_reduce(L.ptr(dx1), L.ptr(raw), L.ptr(ss), L.ptr(rp), L.ptr(s1), rows, C, act, None, L.dtype_code(raw), L.stream()), "reduce")

def one():
    L.check(lib.rssf_conv_gather_bnbwd(L.ptr(dout), L.ptr(wpk), L.ptr(dx2), None, L.ptr(raw), L.ptr(rp), L.ptr(ss), act, L.ptr(s2), B, H, H, CO, H, H, C,
                                       1, 1, spec.ntaps, spec.c_ndy, spec.c_ndx, L.dtype_code(dout), L.stream()), "dgrad+stats")

two(); one(); torch.cuda.synchronize()
a, b = s1.view(-1, 2, C).sum(0), s2.view(-1, 2, C).sum(0)
print("dx identical:", bool((dx1 == dx2).all()), " stats rel err: %.2e" % float((a - b).abs().max() / a.abs().max()))
t2, t1 = bench._time_us(two, 40), bench._time_us(one, 40)
print("K=%d CO=%d " % (K, CO), end="")
print("C=%d %dx%d B=%d act=%d res=%d: dgrad + reduce %.2f us, fused %.2f us" % (C, H, H, B, act, res, t2, t1), flush=True)
