"""Window-attention forward at the bench geometry: us per launch for the library named by RSSF_LIB_OVERRIDE (default: in-tree),
plus a checksum of the output against the in-tree library's (run on the GPU box from the repo root)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from representationlearning_amd import ops, _lib

B, S, C = int(os.environ.get("B", 16)), int(os.environ.get("S", 512)), int(os.environ.get("C", 32))
H = W = S // 4
torch.manual_seed(0)
x = torch.randn(B, H * W, C, device="cuda").bfloat16(); y = torch.randn(B, H * W, C, device="cuda").bfloat16()
g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
_, sx = ops.layernorm_fwd(x, g, b, want_y=False); _, sy = ops.layernorm_fwd(y, g, b, want_y=False)
omega = torch.rand(B, 2, H * W, device="cuda")
w = {}
for n in "qkvo":
    w["w" + n] = (torch.randn(C, C, device="cuda") / C ** 0.5).contiguous(); w["b" + n] = torch.randn(C, device="cuda") * 0.1
for _ in range(5):
    out = ops.winattn_fwd(x, y, sx, sy, omega, g, b, w, H, W, 2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50):
    out = ops.winattn_fwd(x, y, sx, sy, omega, g, b, w, H, W, 2)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print("%s  %.2f us/launch  %.1f GB/s algorithmic  checksum %.6f abs %.6f" % (os.path.basename(_lib.LIB_PATH), us, 3 * B * H * W * C * 2 / us / 1e3,
      float(out.float().sum()), float(out.float().abs().mean())), flush=True)
