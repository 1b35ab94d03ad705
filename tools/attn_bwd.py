"""Times rssf_winattn_fwd / rssf_winattn_bwd at the benchmark geometry (run on the GPU box)."""
import sys, torch
sys.path.insert(0, ".")   # run from the repository root: python tools/<script>.py
from representationlearning_amd import ops
B, H, W, C = 16, 128, 128, 32
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, H * W, C, device=dev).bfloat16(); y = torch.randn(B, H * W, C, device=dev).bfloat16()
dout = torch.randn(B, H * W, C, device=dev).bfloat16()
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
_, sx = ops.layernorm_fwd(x, g, b, want_y=False); _, sy = ops.layernorm_fwd(y, g, b, want_y=False)
omega = torch.full((B, 2, H * W), 0.5, device=dev)
w = {}
for n in ("q", "k", "v", "o"):
    w["w" + n] = (torch.randn(C, C, device=dev) / C ** 0.5).contiguous(); w["b" + n] = torch.zeros(C, device=dev)
gw = {k: torch.zeros_like(v) for k, v in w.items()}
def ev(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("fwd %.1f us" % ev(lambda: ops.winattn_fwd(x, y, sx, sy, omega, g, b, w, H, W, 2), iters))
print("bwd %.1f us" % ev(lambda: ops.winattn_bwd(dout, x, y, sx, sy, omega, g, b, w, gw, H, W, 2), iters))
