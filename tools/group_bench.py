"""Grouped launches (rssf.h "Grouped launches") against the one-problem launches they replace, at the shapes of a Base step's
stage-4 HighResolutionModule (B = 16: 128^2 x 32, 64^2 x 64, 32^2 x 128, 16^2 x 256), bf16.  HIP-event time per call over a ring of
buffer sets larger than the 256 MB Infinity Cache (what a step sees).
  python tools/group_bench.py [branches=4] [iters=30]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from representationlearning_amd import _lib as L  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_gpu_group import _fill, _pack  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
RING = 3
DEV, BF = "cuda", torch.bfloat16
lib = L.load()
SHAPES = [(16, 128, 128, 32), (16, 64, 64, 64), (16, 32, 32, 128), (16, 16, 16, 256)][:NB]
if os.environ.get("GB_CH"):          # e.g. GB_CH=32,32,32,32: the channel counts of the problems (map side = 4096 / C)
    SHAPES = [(16, 4096 // int(c), 4096 // int(c), int(c)) for c in os.environ["GB_CH"].split(",")]
    NB = len(SHAPES)
ONLY = os.environ.get("GB_ONLY", "")          # substring filter on the phase names


def timeit(fn, sets):
    for k in range(3):
        fn(sets[k % len(sets)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(ITERS):
        fn(sets[k % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / ITERS


def mk(B, H, W, C):
    return torch.randn(B, H, W, C, device=DEV).to(BF)


def conv_sets(mirrored):
    sets = []
    for r in range(RING):
        arr = (L.Conv3x3Item * NB)()
        keep = []
        for i, (B, H, W, C) in enumerate(SHAPES):
            x, out = mk(B, H, W, C), mk(B, H, W, C)
            w = torch.randn(C, C, 3, 3, device=DEV) / (3 * C ** 0.5)
            wp = _pack(w, bool(mirrored))
            kw = dict(in_=x, wpk=wp, out=out, B=B, H=H, W=W, Cin=C, Cout=C)
            if mirrored:
                raw, add = mk(B, H, W, C), mk(B, H, W, C)
                ss = torch.rand(2, C, device=DEV)
                sm = torch.zeros(8 * 2 * C, device=DEV)
                kw.update(addend=add, bn_raw=raw, bn_ss=ss, bn_sums=sm, bn_act=1)
                keep += [raw, add, ss, sm]
            else:
                st = torch.zeros(16 * 2 * C, device=DEV)
                kw.update(stats=st)
                keep.append(st)
            _fill(arr[i], **kw)
            keep += [x, out, wp]
        sets.append((arr, keep))
    return sets


def wgrad_sets(res):
    sets = []
    for r in range(RING):
        arr = (L.Wgrad3x3Item * NB)()
        keep = []
        for i, (B, H, W, C) in enumerate(SHAPES):
            x, dy, raw, draw = mk(B, H, W, C), mk(B, H, W, C), mk(B, H, W, C), mk(B, H, W, C)
            rp, dres = (mk(B, H, W, C), mk(B, H, W, C)) if res else (None, None)
            ss, mi = torch.rand(2, C, device=DEV), torch.rand(2, C, device=DEV)
            sums = torch.randn(8, 2, C, device=DEV)
            dg, db, dw = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, C, 3, 3, device=DEV)
            ws = torch.zeros(lib.rssf_conv_wgrad_workspace_elems(B, H, W, C, C, 9), device=DEV)
            job = L.WgradReduceJob()
            _fill(arr[i], in_=x, dw=dw, workspace=ws, defer_reduce=ctypes.addressof(job), bn_dy=dy, bn_raw=raw, bn_ss=ss, bn_mi=mi, bn_sums=sums,
                  bn_res=rp, draw=draw, dres=dres, dgamma=dg, dbeta=db, bn_n=float(B * H * W), pscale=1.0, bn_act=1, bn_training=1, B=B, H=H, W=W,
                  Cin=C, Cout=C)
            keep += [x, dy, raw, draw, rp, dres, ss, mi, sums, dg, db, dw, ws, job]
        sets.append((arr, keep))
    return sets


def bn_sets():
    sets = []
    for r in range(RING):
        fa, ra = (L.BnApplyItem * NB)(), (L.BnReduceItem * NB)()
        keep = []
        for i, (B, H, W, C) in enumerate(SHAPES):
            rows = B * H * W
            raw, rp, y, dy = mk(B, H, W, C), mk(B, H, W, C), mk(B, H, W, C), mk(B, H, W, C)
            st = torch.rand(16, 2, C, device=DEV) * rows / 16
            gamma, beta, rm, rv = torch.rand(C, device=DEV), torch.rand(C, device=DEV), torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            mi, ss = torch.zeros(2, C, device=DEV), torch.rand(2, C, device=DEV)
            sums = torch.zeros(8 * 2 * C, device=DEV)
            _fill(fa[i], raw=raw, stats=st, gamma=gamma, beta=beta, running_mean=rm, running_var=rv, mean_invstd=mi, scale_shift=ss, res_pre=rp,
                  res_post=None, y=y, rows=rows, n=float(rows), momentum=0.1, eps=1e-5, C=C, act=1, training=1)
            _fill(ra[i], dy=dy, raw=raw, scale_shift=ss, res_pre=rp, sums=sums, rows=rows, C=C, act=1)
            keep += [raw, rp, y, dy, st, gamma, beta, rm, rv, mi, ss, sums]
        sets.append((fa, ra, keep))
    return sets


def report(name, tg, ts):
    print("%-34s grouped %7.1f us   one by one %7.1f us   x%.2f" % (name, tg, ts, ts / tg), flush=True)


st = L.stream
for mirrored, name in ((0, "3x3 forward + statistics"), (1, "3x3 data gradient + bn-bwd stats")):
    if ONLY and ONLY not in name:
        continue
    sets = conv_sets(mirrored)
    g = timeit(lambda s: L.check(lib.rssf_conv3x3_group(ctypes.cast(s[0], ctypes.c_void_p), NB, mirrored, L.RSSF_BF16, st()), "g"), sets)

    def single(s):
        for i in range(NB):
            L.check(lib.rssf_conv3x3_group(ctypes.addressof(s[0]) + i * ctypes.sizeof(L.Conv3x3Item), 1, mirrored, L.RSSF_BF16, st()), "s")
    report(name, g, timeit(single, sets))
    del sets
for res in (False, True):
    if ONLY and ONLY not in "weight gradient":
        continue
    sets = wgrad_sets(res)
    g = timeit(lambda s: L.check(lib.rssf_conv3x3_wgrad_group(ctypes.cast(s[0], ctypes.c_void_p), NB, L.RSSF_BF16, st()), "g"), sets)

    def single(s):
        for i in range(NB):
            L.check(lib.rssf_conv3x3_wgrad_group(ctypes.addressof(s[0]) + i * ctypes.sizeof(L.Wgrad3x3Item), 1, L.RSSF_BF16, st()), "s")
    report("3x3 weight gradient + bn apply" + (" + res" if res else ""), g, timeit(single, sets))
    del sets
if ONLY and ONLY not in "bn":
    sys.exit(0)
sets = bn_sets()
g = timeit(lambda s: L.check(lib.rssf_bn_finalize_apply_group(ctypes.cast(s[0], ctypes.c_void_p), NB, L.RSSF_BF16, st()), "g"), sets)


def single_fa(s):
    for i in range(NB):
        L.check(lib.rssf_bn_finalize_apply_group(ctypes.addressof(s[0]) + i * ctypes.sizeof(L.BnApplyItem), 1, L.RSSF_BF16, st()), "s")


report("bn finalize + apply (+ res)", g, timeit(single_fa, sets))
g = timeit(lambda s: L.check(lib.rssf_bn_bwd_reduce_group(ctypes.cast(s[1], ctypes.c_void_p), NB, L.RSSF_BF16, st()), "g"), sets)


def single_ra(s):
    for i in range(NB):
        L.check(lib.rssf_bn_bwd_reduce_group(ctypes.addressof(s[1]) + i * ctypes.sizeof(L.BnReduceItem), 1, L.RSSF_BF16, st()), "s")


report("bn backward statistics", g, timeit(single_ra, sets))
