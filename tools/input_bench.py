"""Throughput of the device input pipeline (rssf_input_pipeline): 16 x 512^2 crops out of 64 resident 1024^2 tiles."""
import sys, torch
sys.path.insert(0, ".")
from representationlearning_amd.data import DeviceAugment
dev = "cuda"
img = torch.randint(0, 256, (64, 1024, 1024, 3), dtype=torch.uint8, device=dev)
msk = torch.randint(0, 8, (64, 1024, 1024), dtype=torch.uint8, device=dev)
for dtype in (torch.bfloat16, torch.float32):
    aug = DeviceAugment(img, msk, crop=512, dtype=dtype, seed=0)
    p = aug.draw(16)
    for _ in range(3): aug.apply(p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): aug.apply(p)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    px = 16 * 512 * 512
    byts = px * (4 + 3 * torch.empty(0, dtype=dtype).element_size() + 8)
    print("%s: %.1f us per batch of 16 (incl. the 256-byte parameter upload) = %.0f images/s, %.2f TB/s of algorithmic bytes" % (dtype, us, 16 / us * 1e6, byts / us / 1e6))
