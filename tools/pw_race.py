"""Stand-alone form of DESIGN.md lesson 59 (no trainer, no model): the ONE launch that is MlpDWBN's fc1 backward (rssf_conv_wgrad_bnapply_dgrad,
128 <- 32 channels: weight gradient + BatchNorm-backward apply + data gradient; its bias sums held the crossed packed add) repeated from
unchanged operands - alone, with another stream keeping the GPU busy (eager), and as a node of a replayed hipGraph with a parallel branch.
Counts the launches whose bias gradient differs from the first by more than TOL = 1e-3 x max |dbias| (fp32 atomics order: 1e-4 of it) and prints which elements.
   python tools/pw_race.py [reps=300] [B=2] [H=32] [W=32]         RSSF_LIB_OVERRIDE=<another build> to test it;  NOISE=bn|conv|ew|all"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from representationlearning_amd import nnf, _lib as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, H, W = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 32, 32)
NOISE = os.environ.get("NOISE", "all")
dev = "cuda"
lib = L.load()
torch.manual_seed(41)
cin, cout, act = 32, 128, 2
conv = torch.nn.Conv2d(cin, cout, 1).to(dev)
spec = nnf.spec_of([conv])
x = torch.randn(B, H, W, cin, device=dev).bfloat16()
dy = torch.randn(B, H, W, cout, device=dev).bfloat16()
raw = (torch.randn(B, H, W, cout, device=dev) * 1.3 + 0.2).bfloat16()
mean, var = raw.float().mean((0, 1, 2)), raw.float().var((0, 1, 2), unbiased=False)
istd = torch.rsqrt(var + 1e-5)
gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2
ss = torch.stack([gamma * istd, beta - mean * gamma * istd]).contiguous()
mi = torch.stack([mean, istd]).contiguous()
rows, n = B * H * W, float(B * H * W)
sums = torch.zeros(nnf.BN_BWD_SLOTS * 2 * cout, device=dev)
L.check(lib.rssf_bn_bwd_reduce(L.ptr(dy), L.ptr(raw), L.ptr(ss), None, L.ptr(sums), rows, cout, act, None, L.dtype_code(raw), L.stream()), "reduce")
w = conv.weight.detach().contiguous()
draw, dx = torch.empty_like(raw), torch.empty(B, H, W, cin, device=dev).bfloat16()
dg, dbt = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
dw, db = torch.zeros_like(w), torch.zeros(cout, device=dev)
bn = (dy, raw, ss, mi, sums, None, None, dg, dbt, act, n, True, 1.0)

def run():
    db.zero_(); dw.zero_(); dg.zero_(); dbt.zero_()
    nnf._conv_wgrad(spec, draw, x, [dw], db, bn=bn, dgrad=(w, dx))

run(); torch.cuda.synchronize()
ref_db, ref_dw, ref_dx = db.clone(), dw.clone(), dx.clone()
print("fc1 backward at %d x %d x %d: |dbias| max %.3g" % (B, H, W, float(ref_db.abs().max())))

side = torch.cuda.Stream()
a = torch.randn(2, 64, 64, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
cv = torch.nn.Conv2d(64, 64, 3, padding=1).to(dev).bfloat16().to(memory_format=torch.channels_last)
junk = torch.randn(64, 1 << 14, device=dev)
n_dy = torch.randn(B, 2 * H, 2 * W, 64, device=dev).bfloat16()
n_raw = torch.randn(B, 2 * H, 2 * W, 64, device=dev).bfloat16()
n_ss = torch.stack([torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)]).contiguous()
n_sums = torch.zeros(nnf.BN_BWD_SLOTS * 2 * 64, device=dev)
def noise(k):
    for _ in range(k):
        if NOISE in ("ew", "all"):
            junk.mul_(1.0001).add_(0.001)
        if NOISE in ("conv", "all"):
            with torch.no_grad():
                cv(a)
        if NOISE in ("bn", "all"):
            L.check(lib.rssf_bn_bwd_reduce(L.ptr(n_dy), L.ptr(n_raw), L.ptr(n_ss), None, L.ptr(n_sums), n_dy.numel() // 64, 64, 1, None, L.dtype_code(n_raw), L.stream()), "reduce")

TOL = 1e-3 * float(ref_db.abs().max())
def check(tag, r, bad):
    d = (db - ref_db).abs()
    if float(d.max()) > TOL or not torch.equal(dx, ref_dx) or float((dw - ref_dw).abs().max()) > TOL:
        bad += 1
        if bad <= 3:
            print("  %s rep %d: dbias elements off %s (max %.3g); dx equal %s; dw max diff %.3g" % (tag, r, (d > TOL).nonzero().flatten().tolist()[:20], float(d.max()), torch.equal(dx, ref_dx), float((dw - ref_dw).abs().max())), flush=True)
    return bad

for mode in ("idle", "busy"):
    bad = 0
    for r in range(reps):
        if mode == "busy":
            with torch.cuda.stream(side):
                noise(2 + r % 4)
        run(); torch.cuda.synchronize()
        bad = check("eager " + mode, r, bad)
    print("eager %-5s launches with a wrong result: %d of %d" % (mode, bad, reps), flush=True)

main = torch.cuda.Stream()
with torch.cuda.stream(main):
    noise(1); run(); torch.cuda.synchronize()
    for branch in (False, True):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            if branch:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    noise(int(os.environ.get("NOISE_K", "3")))
            run()
            if branch:
                main.wait_stream(side)
        bad = 0
        for r in range(reps):
            g.replay(); torch.cuda.synchronize()
            bad = check("graph", r, bad)
        print("graph %s: replays with a wrong result: %d of %d" % ("with a parallel branch" if branch else "single queue", bad, reps), flush=True)
