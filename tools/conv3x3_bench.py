"""3x3 / stride-1 convolution forward at the four HRNet branch shapes of a Base step (B = 16): launch time and TFLOP/s.
RSSF_HALO_MAXC=<c> sends layers wider than c to the generic gather kernel instead of the halo kernel (A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from representationlearning_amd import _lib as L, nnf
import bench
lib = L.load()
B = int(os.environ.get("B", 16))
for C, H in ((32, 128), (64, 64), (128, 32), (256, 16)):
    conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=False).cuda()
    spec = nnf.spec_of([conv])
    x = torch.randn(B, H, H, C, device="cuda").bfloat16()
    wpk = nnf._pack(spec, [conv.weight], False, x.dtype, x.device)
    out = torch.empty_like(x)
    st = torch.zeros(nnf.BN_SLOTS * 2 * C, device="cuda")

    def launch():
        L.check(lib.rssf_conv_gather_add(L.ptr(x), L.ptr(wpk), L.ptr(out), None, L.ptr(st), None, None, B, H, H, C, H, H, C, 1, 1, spec.ntaps,
                                         spec.c_dy, spec.c_dx, L.dtype_code(x), L.stream()), "conv")
    us = bench._time_us(launch, 60)
    print("C=%3d %3dx%-3d B=%d  %6.2f us  %6.0f TFLOP/s  %5.0f GB/s" % (C, H, H, B, us, 2.0 * B * H * H * C * C * 9 / us / 1e6, 2 * x.numel() * 2 / us / 1e3), flush=True)
