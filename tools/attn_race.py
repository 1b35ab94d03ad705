"""Bitwise repeatability of rssf_winattn_bwd's data gradients while another stream keeps the GPU busy (debug aid: an
intra-kernel race shows up once the waves of a workgroup no longer start together)."""
import sys, torch
sys.path.insert(0, ".")
from representationlearning_amd import ops
B, H, W, C = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 32, 32, 32)))
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 300
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, H * W, C, device=dev).bfloat16(); y = torch.randn(B, H * W, C, device=dev).bfloat16()
dout = torch.randn(B, H * W, C, device=dev).bfloat16()
g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
_, sx = ops.layernorm_fwd(x, g, b, want_y=False); _, sy = ops.layernorm_fwd(y, g, b, want_y=False)
omega = torch.rand(B, 2, H * W, device=dev)
w = {}
for n in ("q", "k", "v", "o"):
    w["w" + n] = (torch.randn(C, C, device=dev) / C ** 0.5).contiguous(); w["b" + n] = torch.randn(C, device=dev) * 0.1
def run():
    gw = {k: torch.zeros_like(v) for k, v in w.items()}
    return ops.winattn_bwd(dout, x, y, sx, sy, omega, g, b, w, gw, H, W, 2) + (gw,)
ref = run(); torch.cuda.synchronize()
side = torch.cuda.Stream()
junk = torch.randn(64, 1 << 16, device=dev)
for mode in ("idle", "busy"):
    bad = [0, 0, 0]
    for r in range(reps):
        if mode == "busy":
            with torch.cuda.stream(side):
                for _ in range(3 + r % 5):
                    junk.mul_(1.0001).add_(0.001)
        out = run()
        torch.cuda.synchronize()
        for i in range(3):
            if not torch.equal(out[i], ref[i]):
                bad[i] += 1
                if bad[i] <= 2:
                    d = (out[i].float() - ref[i].float()).abs()
                    print(mode, "rep", r, ("dxhat", "dyhat", "domega")[i], "differs: max %.3g of %.3g, n=%d" % (d.max().item(), ref[i].float().abs().max().item(), int((d > 0).sum())), flush=True)
    print(mode, "mismatching launches of", reps, "(dxhat, dyhat, domega):", bad, flush=True)
