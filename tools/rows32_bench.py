"""(The weight-gradient cases time the halo kernel in both columns: the row-stream weight gradient was measured and dropped, DESIGN lesson 55.)
The 32 -> 32 channel 3x3 convolution at branch 0's geometry (B x 128 x 128 x 32): the row-stream kernel (csrc/conv_rows32.hip)
against the halo kernel (RSSF_CONV_GENERIC in the call), forward / pre-activation forward / data gradient with the fused
BatchNorm-backward statistics and a skip gradient, launch times by HIP events - back to back on one operand set (cache-resident, what
a step's producer -> consumer chain sees) and cycling through a ring of operand sets larger than the 256 MB Infinity Cache.
   B=16 python tools/rows32_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from representationlearning_amd import _lib as L, nnf
import bench
lib = L.load()
B = int(os.environ.get("B", 16)); H = W = int(os.environ.get("HW", 128)); C = 32
RING = int(os.environ.get("RING", 12))
dev = "cuda"
torch.manual_seed(0)
conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=False).to(dev)
spec = nnf.spec_of([conv])
wf = nnf._pack(spec, [conv.weight], False, torch.bfloat16, torch.device(dev))
wt = nnf._pack(spec, [conv.weight], True, torch.bfloat16, torch.device(dev))
mk = lambda: torch.randn(B, H, W, C, device=dev).bfloat16()
sets = [dict(x=mk(), raw=mk(), res=mk(), add=mk(), out=torch.empty(B, H, W, C, device=dev, dtype=torch.bfloat16)) for _ in range(RING)]
st = torch.zeros(nnf.BN_SLOTS * 2 * C, device=dev)
sums = torch.zeros(nnf.BN_BWD_SLOTS * 2 * C, device=dev)
ss = torch.stack([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3]).contiguous()
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
mi, pss = torch.empty(2 * C, device=dev), torch.empty(2 * C, device=dev)
pst = torch.zeros(nnf.BN_SLOTS * 2 * C, device=dev); pst[:C] = 0.1 * B * H * W; pst[C:2 * C] = 1.2 * B * H * W
n = float(B * H * W)
state = {"i": 0, "ring": False}


def cur():
    if state["ring"]:
        state["i"] = (state["i"] + 1) % RING
    return sets[state["i"]]


def fwd(flag):
    s = cur()
    L.check(lib.rssf_conv_gather_add(L.ptr(s["x"]), L.ptr(wf), L.ptr(s["out"]), None, L.ptr(st), None, None, B, H, W, C, H, W, C, 1, 1, 9, spec.c_dy, spec.c_dx,
                                     L.RSSF_BF16 | flag, L.stream()), "fwd")


def pre(flag):
    s = cur()
    L.check(lib.rssf_conv_gather_preact(L.ptr(s["x"]), L.ptr(pst), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(mi), L.ptr(pss), n, 0.1, 1e-5, 1, 1,
                                        L.ptr(wf), L.ptr(s["out"]), None, L.ptr(st), None, B, H, W, C, H, W, C, 1, 1, 9, spec.c_dy, spec.c_dx,
                                        L.RSSF_BF16 | flag, L.stream()), "pre")


def dgrad(flag, res=True, add=True, bn=True):
    s = cur()
    if bn:
        L.check(lib.rssf_conv_gather_bnbwd(L.ptr(s["x"]), L.ptr(wt), L.ptr(s["out"]), L.ptr(s["add"]) if add else None, L.ptr(s["raw"]),
                                           L.ptr(s["res"]) if res else None, L.ptr(ss), 1, L.ptr(sums), B, H, W, C, H, W, C, 1, 1, 9, spec.c_ndy, spec.c_ndx,
                                           L.RSSF_BF16 | flag, L.stream()), "dgrad")
    else:
        L.check(lib.rssf_conv_gather_add(L.ptr(s["x"]), L.ptr(wt), L.ptr(s["out"]), None, None, L.ptr(s["add"]) if add else None, None, B, H, W, C, H, W, C, 1, 1, 9,
                                         spec.c_ndy, spec.c_ndx, L.RSSF_BF16 | flag, L.stream()), "dgrad")


nws = lib.rssf_conv_wgrad_workspace_elems(B, H, W, C, C, 9)
wsb = torch.empty(nws, device=dev)
dwb = torch.zeros(C, C, 3, 3, device=dev)
dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
mi2 = torch.cat([torch.randn(C, device=dev) * 0.2, torch.rand(C, device=dev) + 0.5]).contiguous()
xss = torch.cat([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3]).contiguous()
sums2 = torch.randn(nnf.BN_BWD_SLOTS * 2 * C, device=dev)


import ctypes
job = L.WgradReduceJob()
DEFER = os.environ.get("DEFER", "1") != "0"      # first stage only (the step defers every reduction to one batched launch)


def wgrad(flag, fuse=True, res=True, xpre=True):
    s = cur()
    tail = (L.ptr(dwb), None, None, spec.c_ksizes, 1, spec.c_src, spec.c_kpos, spec.c_alias, None, L.ptr(wsb), B, H, W, C, H, W, C, 1, 9, spec.c_dy, spec.c_dx,
            ctypes.byref(job) if DEFER else None, L.RSSF_BF16 | flag, L.stream())
    if fuse:
        L.check(lib.rssf_conv_wgrad_bnapply(L.ptr(s["x"]), L.ptr(s["raw"]), L.ptr(ss), L.ptr(mi2), L.ptr(sums2), L.ptr(s["res"]) if res else None, L.ptr(s["out"]),
                                            L.ptr(s["add"]) if res else None, L.ptr(dgm), L.ptr(dbt), 1, n, 1, 1.0, L.ptr(s["raw"]), L.ptr(xss) if xpre else None, 1,
                                            *tail), "wgrad fused")
    else:
        L.check(lib.rssf_conv_wgrad(L.ptr(s["x"]), L.ptr(s["raw"]), *tail), "wgrad")


cases = [("weight gradient plain", lambda f: wgrad(f, False, False, False), 2),
         ("weight gradient + BN-backward apply", lambda f: wgrad(f, True, False, False), 4),
         ("weight gradient + apply + residual + pre-activation input", lambda f: wgrad(f, True, True, True), 6),
         ("forward + statistics", fwd, 2), ("pre-activation forward + statistics", pre, 2),
         ("data gradient plain", lambda f: dgrad(f, False, False, False), 2),
         ("data gradient + BN-backward statistics", lambda f: dgrad(f, False, False, True), 3),
         ("data gradient + statistics + residual + skip gradient", lambda f: dgrad(f, True, True, True), 5)]
tb = B * H * W * C * 2 / 1e6
print("B=%d %dx%d C=%d: one tensor = %.1f MB, 4.8 GFLOP at B = 16" % (B, H, W, C, tb))
ONLY = os.environ.get("CASES")
for name, fn, ntens in cases:
    if ONLY and not any(k in name for k in ONLY.split(",")):
        continue
    row = []
    for ring in (False, True):
        state["ring"] = ring
        for flag in (L.CONV_GENERIC, 0):
            row.append(bench._time_us(lambda: fn(flag), 60 if not ring else 5 * RING))
    print("%-56s hot: halo %6.2f us  rows32 %6.2f us (%4.2f TB/s)   cold ring: halo %6.2f us  rows32 %6.2f us (%4.2f TB/s)"
          % (name, row[0], row[1], ntens * tb / row[1], row[2], row[3], ntens * tb / row[3]), flush=True)
