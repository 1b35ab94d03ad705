"""Micro-benchmark of conv kernels on the MlpDWBN 19-tap shape and HRNet 3x3 shapes (run on the GPU box)."""
import sys, torch
sys.path.insert(0, ".")   # run from the repository root: python tools/<script>.py
from representationlearning_amd import nnf, _lib as L
which = sys.argv[1] if len(sys.argv) > 1 else "mlp"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda"
torch.manual_seed(0)
def ev(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
if which == "mlp":
    C, B, H, W = 128, 16, 128, 128
    convs = [torch.nn.Conv2d(C, C, 1), torch.nn.Conv2d(C, C, 3, padding=6, dilation=6), torch.nn.Conv2d(C, C, 3, padding=12, dilation=12)]
else:
    C, B, H, W = {"c32": (32, 16, 128, 128), "c64": (64, 16, 64, 64), "c128": (128, 16, 32, 32), "c256": (256, 16, 16, 16)}[which]
    convs = [torch.nn.Conv2d(C, C, 3, padding=1, bias=False)]
convs = [c.to(dev) for c in convs]
spec = nnf.spec_of(convs)
x = torch.randn(B, H, W, C, device=dev).bfloat16()
dout = torch.randn(B, H, W, C, device=dev).bfloat16()
ws = [c.weight for c in convs]
gw = [torch.zeros_like(w) for w in ws]
flops = 2.0 * B * H * W * C * C * spec.ntaps
t = ev(lambda: nnf._conv_wgrad(spec, dout, x, gw, None), iters)
print(f"{which}: wgrad {t:.1f} us  {flops / t / 1e6:.0f} TFLOP/s")
t = ev(lambda: nnf._conv_forward(spec, x, ws, None, None), iters)
print(f"{which}: fwd   {t:.1f} us  {flops / t / 1e6:.0f} TFLOP/s")
t = ev(lambda: nnf._conv_dgrad(spec, dout, ws, x.shape), iters)
print(f"{which}: dgrad {t:.1f} us  {flops / t / 1e6:.0f} TFLOP/s")
