"""Bitwise repeatability of rssf_gate_weights_bwd (dpooled) with another stream keeping the GPU busy, eager and inside a
replayed hipGraph with a parallel branch (debug aid)."""
import sys, torch
sys.path.insert(0, ".")
from representationlearning_amd import ops
B, H, W = 2, 32, 32
N = H * W
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = "cuda"
torch.manual_seed(0)
domega = torch.randn(B, 2, N, device=dev) * 1e-3
pooled = torch.randn(B, 4, N, device=dev)
kk = torch.randn(2, 2, 7, 7, device=dev) * 0.1
wl = torch.randn(2, 2, device=dev)
bl = torch.randn(2, device=dev)
gsig, omega, _ = ops.gate_weights_fwd(pooled, kk, wl, bl, H, W)
def run():
    dk, dwl, dbl = torch.zeros_like(kk), torch.zeros_like(wl), torch.zeros(2, device=dev)
    return ops.gate_weights_bwd(domega, pooled, gsig, omega, kk, wl, dk, dwl, dbl, H, W)
ref = run().clone(); torch.cuda.synchronize()
side = torch.cuda.Stream()
a = torch.randn(2, 64, 64, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
conv = torch.nn.Conv2d(64, 64, 3, padding=1).to(dev).bfloat16().to(memory_format=torch.channels_last)
junk = torch.randn(64, 1 << 14, device=dev)
def noise(k):
    for _ in range(k):
        junk.mul_(1.0001).add_(0.001)
        with torch.no_grad():
            conv(a)
for mode in ("idle", "busy"):
    bad = 0
    for r in range(reps):
        if mode == "busy":
            with torch.cuda.stream(side):
                noise(2 + r % 4)
        out = run()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            if bad <= 3:
                nz = (out != ref).nonzero()
                print(mode, "rep", r, "n", len(nz), nz[:4].tolist(), flush=True)
    print("eager", mode, "mismatching launches:", bad, "of", reps, flush=True)
# graph with a parallel branch
main = torch.cuda.Stream()
outs = []
with torch.cuda.stream(main):
    noise(1); run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            noise(6)
        for _ in range(4):
            outs.append(run())
        main.wait_stream(side)
    bad = 0
    for r in range(reps):
        g.replay(); torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                if bad <= 3:
                    nz = (o != ref).nonzero()
                    print("graph rep", r, "n", len(nz), nz[:4].tolist(), flush=True)
    print("graph mismatching results:", bad, "of", reps * len(outs), flush=True)
