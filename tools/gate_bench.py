"""Times the LayerNorm / gate kernels at the benchmark geometry (run on the GPU box)."""
import sys, torch
sys.path.insert(0, ".")   # run from the repository root: python tools/<script>.py
from representationlearning_amd import ops
B, H, W, C = 16, 128, 128, 32
N = H * W
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, N, C, device=dev).bfloat16(); y = torch.randn(B, N, C, device=dev).bfloat16(); dy = torch.randn(B, N, C, device=dev).bfloat16()
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
def ev(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
_, sx = ops.layernorm_fwd(x, g, b, want_y=False); _, sy = ops.layernorm_fwd(y, g, b, want_y=False)
print("ln_fwd(stats only) %.1f us" % ev(lambda: ops.layernorm_fwd(x, g, b, want_y=False)))
print("ln_fwd(y)          %.1f us" % ev(lambda: ops.layernorm_fwd(x, g, b, want_y=True)))
print("ln_bwd             %.1f us" % ev(lambda: ops.layernorm_bwd(dy, x, sx, g, dg, db)))
print("ln_bwd(+add)       %.1f us" % ev(lambda: ops.layernorm_bwd(dy, x, sx, g, dg, db, dx_add=y)))
pooled, argmax = ops.gate_pool_fwd(x, y, sx, sy, g, b)
print("gate_pool_fwd      %.1f us" % ev(lambda: ops.gate_pool_fwd(x, y, sx, sy, g, b)))
print("ln statistics x 2 + gate pooling as ONE launch (rssf_ln_gate_pool_fwd)  %.1f us" % ev(lambda: ops.ln_gate_pool_fwd(x, y, g, b)))
k = torch.randn(2, 2, 7, 7, device=dev) * 0.1; wl = torch.randn(2, 2, device=dev); bl = torch.zeros(2, device=dev)
gsig, omega, _ = ops.gate_weights_fwd(pooled, k, wl, bl, H, W)
print("gate_weights_fwd   %.1f us" % ev(lambda: ops.gate_weights_fwd(pooled, k, wl, bl, H, W)))
domega = torch.randn(B, 2, N, device=dev)
dk, dwl, dbl = torch.zeros_like(k), torch.zeros_like(wl), torch.zeros_like(bl)
dpooled = ops.gate_weights_bwd(domega, pooled, gsig, omega, k, wl, dk, dwl, dbl, H, W)
print("gate_weights_bwd   %.1f us" % ev(lambda: ops.gate_weights_bwd(domega, pooled, gsig, omega, k, wl, dk, dwl, dbl, H, W)))
dxh, dyh = x.clone(), y.clone()
print("gate_pool_bwd      %.1f us" % ev(lambda: ops.gate_pool_bwd_(dpooled, argmax, dxh, dyh)))
print("gate pool backward + ln_bwd(+add) + ln_bwd as ONE launch (rssf_gate_pool_ln_bwd)  %.1f us"
      % ev(lambda: ops.gate_pool_ln_bwd(dpooled, argmax, dxh, dyh, x, y, sx, sy, g, dg, db, dx_add=dy)))
