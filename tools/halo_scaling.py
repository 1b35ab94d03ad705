"""3x3 / stride-1 convolution (halo kernel) launch time against the batch size at a fixed map: separates the fixed cost of a launch
from the cost per round of resident blocks.  C, H via the environment; run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from representationlearning_amd import _lib as L, nnf
import bench
C, H = int(os.environ.get("C", 32)), int(os.environ.get("H", 128))
lib = L.load()
conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=False).cuda()
spec = nnf.spec_of([conv])
for B in (1, 2, 4, 8, 12, 16, 24, 32, 64):
    x = torch.randn(B, H, H, C, device="cuda").bfloat16()
    wpk = nnf._pack(spec, [conv.weight], False, x.dtype, x.device)
    out = torch.empty_like(x)
    st = torch.zeros(nnf.BN_SLOTS * 2 * C, device="cuda")
    for stats in (None, st):
        def launch():
            L.check(lib.rssf_conv_gather_add(L.ptr(x), L.ptr(wpk), L.ptr(out), None, L.ptr(stats), None, None, B, H, H, C, H, H, C, 1, 1, spec.ntaps,
                                             spec.c_dy, spec.c_dx, L.dtype_code(x), L.stream()), "conv")
        us = bench._time_us(launch, 40)
        print("C=%d %dx%d B=%2d stats=%d  blocks %5d  %6.2f us  %5.0f GB/s" % (C, H, H, B, stats is not None, B * (H // 8) * (H // 16) * max(1, C // 64 if C >= 64 else 1), us,
              2 * x.numel() * 2 / us / 1e3), flush=True)
