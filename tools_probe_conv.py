"""GPU probe (not part of the product): MIOpen conv throughput NCHW vs channels_last, bf16, for the HRNet shapes."""
import time, torch, torch.nn.functional as F
dev = "cuda"
print(torch.cuda.get_device_name(0), torch.version.hip)
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n
B = 16
shapes = [(32, 32, 128, 3, 1, 1), (64, 64, 64, 3, 1, 1), (128, 128, 32, 3, 1, 1), (256, 256, 16, 3, 1, 1),
          (128, 128, 128, 3, 6, 6), (128, 128, 128, 3, 12, 12), (128, 128, 128, 1, 0, 1), (32, 128, 128, 1, 0, 1),
          (480, 480, 128, 1, 0, 1), (64, 256, 128, 1, 0, 1)]
for (ci, co, hw, k, pad, dil) in shapes:
    for fmt in ("nchw", "nhwc"):
        x = torch.randn(B, ci, hw, hw, device=dev, dtype=torch.bfloat16)
        w = torch.randn(co, ci, k, k, device=dev, dtype=torch.bfloat16)
        if fmt == "nhwc":
            x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(); w.requires_grad_()
        flops = 2 * B * hw * hw * ci * co * k * k
        try:
            tf = bench(lambda: F.conv2d(x, w, None, 1, pad, dil))
            y = F.conv2d(x, w, None, 1, pad, dil); gy = torch.randn_like(y)
            tb = bench(lambda: torch.autograd.grad(F.conv2d(x, w, None, 1, pad, dil), (x, w), gy))
            print(f"conv {ci}->{co} @{hw} k{k} d{dil} {fmt}: fwd {tf*1e3:.3f} ms ({flops/tf/1e12:.1f} TF)  fwd+bwd {tb*1e3:.3f} ms ({3*flops/tb/1e12:.1f} TF)", flush=True)
        except Exception as e:
            print("conv", ci, co, hw, k, fmt, "ERR", repr(e)[:200], flush=True)
# batchnorm
for fmt in ("nchw", "nhwc"):
    x = torch.randn(B, 32, 128, 128, device=dev, dtype=torch.bfloat16)
    if fmt == "nhwc": x = x.contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(32).to(dev)
    t = bench(lambda: bn(x)); print("bn 32@128", fmt, f"{t*1e6:.1f} us", flush=True)
