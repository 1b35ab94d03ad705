"""1x1 convolutions of the MlpDWBN at the bench geometry (fc1 32 -> 128, fc2 128 -> 32 and their data gradients), stand-alone:
us per launch and effective GB/s (read in + write out).  Run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from representationlearning_amd import _lib as L, nnf
import bench

B, H, W = int(os.environ.get("B", 16)), 128, 128
lib = L.load()
for cin, cout, stats in ((32, 128, True), (32, 128, False), (128, 32, True), (128, 32, False), (64, 64, True), (128, 128, True), (256, 256, True), (480, 480, True), (64, 256, True), (256, 64, True)):
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    spec = nnf.spec_of([conv])
    x = torch.randn(B, H, W, cin, device="cuda").bfloat16()
    wpk = nnf._pack(spec, [conv.weight], False, x.dtype, x.device)
    out = torch.empty(B, H, W, cout, device="cuda", dtype=x.dtype)
    st = torch.zeros(nnf.BN_SLOTS * 2 * cout, device="cuda") if stats else None

    def launch():
        L.check(lib.rssf_conv_gather_add(L.ptr(x), L.ptr(wpk), L.ptr(out), None, L.ptr(st), None, None, B, H, W, cin, H, W, cout, 1, 1, spec.ntaps,
                                         spec.c_dy, spec.c_dx, L.dtype_code(x), L.stream()), "rssf_conv_gather_add")
    us = bench._time_us(launch, 30)
    mb = B * H * W * (cin + cout) * 2 / 1e6
    print("1x1 %3d -> %3d  stats=%d  %7.2f us  %6.0f GB/s  (%.1f MB)" % (cin, cout, stats, us, mb * 1e-3 / (us * 1e-6), mb), flush=True)
