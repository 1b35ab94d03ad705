#!/usr/bin/env python3
"""bench.py — RSSFormer-Base training step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = forward + CGFL loss + backward + gradient all-reduce (N>1) + global-norm clip + SGD on a fixed synthetic
minibatch already resident in HBM: RSSFormer-Base (HRNetV2-W32 + 8 transformer blocks, 32.14 M params), 6 classes,
per-GPU batch 16 x 3 x 512 x 512 (BASELINE configs[1] / [2]), bf16 activations with fp32 master weights.  Prints ONE
JSON line on rank 0.  `roofline` times the fused window cross-attention forward kernel (the kernel BASELINE's
north_star names; HBM-bound) with HIP events on the launch stream; `cpu_baseline` times the CPU
restatement (oracle/) of the same step on the host cores, rank 0, N=1 only.  `roofline_mfma` is the same measurement
for the step's largest GEMM (the MlpDWBN 19-tap fused convolution), which is MFMA-bound.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FLOP_PER_IMG = 528.07e9        # fwd+bwd matmul/conv FLOPs per 512x512 image, Base
# fwd+bwd FLOPs per image at the size SURVEY §6 counted them on (they scale with the pixel count), branch-0 width, heads
VARIANTS = {"tiny": (48.17e9, 256, 18), "base": (FLOP_PER_IMG, 512, 32), "large": (4548.4e9, 1024, 48)}
MFMA_BF16_PEAK = 2.5e15


def _source_sha16(*rel):
    import hashlib
    h = hashlib.sha256()
    for r in rel:
        with open(os.path.join(ROOT, r), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


ATTN_SOURCES = ("representationlearning_amd/csrc/win_attn_fwd.hip", "representationlearning_amd/csrc/win_attn.hip.h",
                "representationlearning_amd/csrc/common.hip.h")
MLP_SOURCES = ("representationlearning_amd/csrc/conv_taps128.hip", "representationlearning_amd/csrc/conv.hip.h",
               "representationlearning_amd/csrc/common.hip.h")


def _profiled_traffic(kernel, sources=ATTN_SOURCES, stamp="source_sha16"):
    """HBM-side bytes per launch of `kernel` from the committed counter pass (the newest profiles/rNN_hbm_traffic.json whose source stamp matches, written by
    tools/hbm_traffic.sh + tools/hbm_traffic_json.py: rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum in a
    run of its own, reads doubled as MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read inside this process.
    The file is stamped with the hash of the kernel's sources: a number measured on OTHER code is reported as null, not stale."""
    import glob
    sha = _source_sha16(*sources)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):      # newest round first
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get(stamp) == sha:
                e = d[kernel]
                return int(e["read_bytes"] + e["write_bytes"])
        except (OSError, KeyError, ValueError):
            continue
    return None


def _time_us(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def measure_dominant_kernels(B, S, iters=30, C=32):
    """In-run HIP-event timings of the kernels that dominate the step BY TIME (profiles/r02_step_census.txt), at the benchmark
    geometry of branch 0 (C = 32, (S/4)^2 map), each against the roofline that bounds it: the BatchNorm passes and the 3x3 halo
    convolution / its weight gradient (HBM: arithmetic intensity 144 FLOP/B at C = 32), the attention backward (HBM)."""
    from representationlearning_amd import _lib as L, nnf, ops
    lib = L.load()
    H = W = S // 4
    dev = "cuda"
    rows = B * H * W
    torch.manual_seed(1)
    raw = torch.randn(B, H, W, C, device=dev).bfloat16()
    dy = torch.randn(B, H, W, C, device=dev).bfloat16()
    y = torch.empty_like(raw)
    draw = torch.empty_like(raw)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mi, ss = torch.empty(2, C, device=dev), torch.empty(2, C, device=dev)
    stats = torch.zeros(nnf.BN_SLOTS * 2 * C, device=dev)
    stats[:C] = raw.float().sum((0, 1, 2)); stats[C:2 * C] = raw.float().square().sum((0, 1, 2))
    sums = torch.zeros(nnf.BN_BWD_SLOTS * 2 * C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    code, st = L.RSSF_BF16, L.stream
    tensor_bytes = rows * C * 2
    out = []

    def entry(kernel, us, alg_bytes, what, flops=None):
        gbs = alg_bytes / (us * 1e-6) / 1e9
        d = dict(kernel=kernel, bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                 us_per_launch=round(us, 2), algorithmic_bytes=alg_bytes, what=what)
        if flops:
            d["tflops"] = round(flops / (us * 1e-6) / 1e12, 1)
        out.append(d)

    us = _time_us(lambda: L.check(lib.rssf_bn_finalize_apply(L.ptr(raw), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(mi),
                                                             L.ptr(ss), None, None, L.ptr(y), rows, C, 1, float(rows), 0.1, 1e-5, 1, code, st()), "bn"), iters)
    entry("bn_finapply_kernel<bf16,8>", us, 2 * tensor_bytes, "BatchNorm finalize+apply+ReLU, C=%d: read raw, write y" % C)
    us = _time_us(lambda: (sums.zero_(), L.check(lib.rssf_bn_bwd_reduce(L.ptr(dy), L.ptr(raw), L.ptr(ss), None, L.ptr(sums), rows, C, 1, None, code, st()), "bn")), iters)
    entry("bn_bwd_reduce_kernel<bf16,8> (+ 2 KB memset)", us, 2 * tensor_bytes, "BatchNorm backward statistics, C=%d: read dy, raw" % C)
    us = _time_us(lambda: L.check(lib.rssf_bn_bwd_apply(L.ptr(dy), L.ptr(raw), L.ptr(ss), L.ptr(mi), L.ptr(sums), None, L.ptr(draw), None, L.ptr(dg),
                                                        L.ptr(db), rows, C, 1, float(rows), 1, 1.0, code, st()), "bn"), iters)
    entry("bn_bwd_apply_kernel<bf16,8>", us, 3 * tensor_bytes, "BatchNorm backward apply, C=%d: read dy, raw, write draw" % C)
    conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=False).to(dev)
    spec = nnf.spec_of([conv])
    x = torch.relu(torch.randn(B, H, W, C, device=dev)).bfloat16()
    cstats = torch.zeros(nnf.BN_SLOTS * 2 * C, device=dev)
    flops = 2.0 * rows * C * C * 9
    # the launches alone, as in the step: the weights are packed once per step for all layers (nnf.PackPlan), the split-K second stage of
    # every weight gradient is the step's ONE batched reduction
    import ctypes
    wpk = nnf._pack(spec, [conv.weight], False, x.dtype, x.device)
    wpk_t = nnf._pack(spec, [conv.weight], True, x.dtype, x.device)
    cout = torch.empty_like(x)
    dims = (B, H, W, C, H, W, C, 1, 1, 9)
    us = _time_us(lambda: L.check(lib.rssf_conv_gather_add(L.ptr(x), L.ptr(wpk), L.ptr(cout), None, L.ptr(cstats), None, None, *dims, spec.c_dy, spec.c_dx,
                                                           code, st()), "conv"), iters)
    entry("conv3x3_rows32_kernel<forward>", us, 2 * tensor_bytes, "3x3 conv %d->%d with fused BN statistics: read in, write out" % (C, C), flops)
    us = _time_us(lambda: L.check(lib.rssf_conv_gather_add(L.ptr(x), L.ptr(wpk), L.ptr(cout), None, L.ptr(cstats), None, None, *dims, spec.c_dy, spec.c_dx,
                                                           code | L.CONV_GENERIC, st()), "conv"), iters)
    entry("conv3x3_halo_kernel<8,%d> (the generic kernel on the same call: RSSF_CONV_GENERIC)" % C, us, 2 * tensor_bytes,
          "3x3 conv %d->%d with fused BN statistics: read in, write out" % (C, C), flops)
    us = _time_us(lambda: L.check(lib.rssf_conv_gather_bnbwd(L.ptr(dy), L.ptr(wpk_t), L.ptr(cout), L.ptr(draw), L.ptr(raw), L.ptr(x), L.ptr(ss), 1, L.ptr(sums),
                                                             *dims, spec.c_ndy, spec.c_ndx, code, st()), "dgrad"), iters)
    entry("conv3x3_rows32_kernel<data gradient, BN-backward statistics, residual, skip gradient>", us, 5 * tensor_bytes,
          "3x3 data gradient %d->%d: read dout, raw, residual, skip gradient; write dx" % (C, C), flops)
    dw = torch.zeros_like(conv.weight)
    nws = lib.rssf_conv_wgrad_workspace_elems(B, H, W, C, C, 9)
    wws = torch.empty(nws, device=dev)
    wjob = L.WgradReduceJob()
    us = _time_us(lambda: L.check(lib.rssf_conv_wgrad(L.ptr(dy), L.ptr(x), L.ptr(dw), None, None, spec.c_ksizes, 1, spec.c_src, spec.c_kpos, spec.c_alias, None,
                                                      L.ptr(wws), B, H, W, C, H, W, C, 1, 9, spec.c_dy, spec.c_dx, ctypes.byref(wjob), code, st()), "wgrad"), iters)
    entry("conv3x3_wgrad_halo_kernel (first stage; %d partial planes)" % int(wjob.ksplit), us, 2 * tensor_bytes,
          "3x3 weight gradient %dx%d: read dout, in" % (C, C), flops)
    # MlpDWBN's fc1 backward at the same map: 4C <- C point-wise weight gradient with the BatchNorm-backward apply fused in
    # (conv_wgrad_pw.hip): reads dy, raw [4C] and the input [C], writes draw [4C]
    C4 = 4 * C
    fc1 = torch.nn.Conv2d(C, C4, 1, bias=False).to(dev)
    spec1 = nnf.spec_of([fc1])
    x1 = torch.randn(B, H, W, C, device=dev).bfloat16()
    dy4, raw4, draw4 = (torch.randn(B, H, W, C4, device=dev).bfloat16() for _ in range(3))
    ss4 = torch.stack([torch.ones(C4, device=dev), torch.zeros(C4, device=dev)]).contiguous()
    mi4 = torch.stack([torch.zeros(C4, device=dev), torch.ones(C4, device=dev)]).contiguous()
    sums4 = torch.zeros(nnf.BN_BWD_SLOTS * 2 * C4, device=dev)
    dg4, db4 = torch.zeros(C4, device=dev), torch.zeros(C4, device=dev)
    dw1 = torch.zeros_like(fc1.weight, dtype=torch.float32)
    bn4 = (dy4, raw4, ss4, mi4, sums4, None, None, dg4, db4, 2, float(rows), True, 1.0)
    us = _time_us(lambda: nnf._conv_wgrad(spec1, draw4, x1, [dw1], None, bn=bn4), iters)
    entry("conv_wgrad_pw_kernel<fused apply> + wgrad_reduce_kernel", us, 3 * rows * C4 * 2 + tensor_bytes,
          "point-wise weight gradient %d<-%d + BatchNorm backward apply (GELU): read dy, raw, in; write draw" % (C4, C), 2.0 * rows * C * C4)
    # attention backward (the kernel furthest below its roofline in round 1)
    N = H * W
    xa = torch.randn(B, N, C, device=dev).bfloat16(); ya = torch.randn(B, N, C, device=dev).bfloat16(); da = torch.randn(B, N, C, device=dev).bfloat16()
    g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    _, sx = ops.layernorm_fwd(xa, g1, b1, want_y=False); _, sy = ops.layernorm_fwd(ya, g1, b1, want_y=False)
    omega = torch.full((B, 2, N), 0.5, device=dev)
    w = {}
    for n in "qkvo":
        w["w" + n] = (torch.randn(C, C, device=dev) / C ** 0.5).contiguous(); w["b" + n] = torch.zeros(C, device=dev)
    gw = {k: torch.zeros_like(v) for k, v in w.items()}
    us = _time_us(lambda: ops.winattn_bwd(da, xa, ya, sx, sy, omega, g1, b1, w, gw, H, W, 2), max(5, iters // 3))
    entry("winattn_bwd_kernel<bf16,Dims<%d,2>> + domega_reduce_kernel" % C, us, 5 * B * N * C * 2, "read x, y, dout; write dxhat, dyhat")
    return out


def measure_window_attention(B, S, iters=30, C=32):
    """Average duration of one rssf_winattn_fwd launch at the benchmark geometry (branch 0: C=32, (S/4)^2 tokens)."""
    from representationlearning_amd import ops
    H = W = S // 4
    dev = "cuda"
    x = torch.randn(B, H * W, C, device=dev).bfloat16()
    y = torch.randn(B, H * W, C, device=dev).bfloat16()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    _, sx = ops.layernorm_fwd(x, g, b, want_y=False)
    _, sy = ops.layernorm_fwd(y, g, b, want_y=False)
    omega = torch.full((B, 2, H * W), 0.5, device=dev)
    w = {}
    for n in ("q", "k", "v", "o"):
        w["w" + n] = (torch.randn(C, C, device=dev) / C ** 0.5).contiguous()
        w["b" + n] = torch.zeros(C, device=dev)
    for _ in range(5):
        ops.winattn_fwd(x, y, sx, sy, omega, g, b, w, H, W, 2)
    ms = _time_us(lambda: ops.winattn_fwd(x, y, sx, sy, omega, g, b, w, H, W, 2), iters, warm=0) / 1e3
    alg_bytes = 3 * B * H * W * C * 2                # read low, read high, write out (bf16)
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return dict(bound="hbm", kernel="winattn_fwd_kernel<bf16,Dims<%d,2>,contiguous>" % C, achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(gbs / HBM_PEAK_GBS, 4), traffic=_profiled_traffic("winattn_fwd_kernel") if (B, S, C) == (16, 512, 32) else None,
                us_per_launch=round(ms * 1e3, 2), algorithmic_bytes=alg_bytes)


def measure_mlp_conv(B, S, iters=20, C=128):
    """Average duration of the MlpDWBN fused {1x1 + 3x3 dil 6 + 3x3 dil 12} convolution (ONE implicit-GEMM launch, 17 distinct taps,
    128 -> 128 channels on the (S/4)^2 map): the largest single GEMM of the step, MFMA-bound."""
    from representationlearning_amd import _lib as L, nnf
    H = W = S // 4
    dev = "cuda"
    torch.manual_seed(0)
    convs = [torch.nn.Conv2d(C, C, 1), torch.nn.Conv2d(C, C, 3, padding=6, dilation=6), torch.nn.Conv2d(C, C, 3, padding=12, dilation=12)]
    convs = [c.to(dev) for c in convs]
    spec = nnf.spec_of(convs)
    x = torch.nn.functional.gelu(torch.randn(B, H, W, C, device=dev)).bfloat16()      # the distribution fc1 + GELU feeds it
    # as in the training step: the weights are packed once per step for ALL convolutions (nnf.PackPlan, one launch), the
    # convolution itself is the rssf_conv_gather_add launch timed here
    wpk = nnf._pack(spec, [c.weight for c in convs], False, x.dtype, x.device)
    out = torch.empty(B, H, W, C, device=dev, dtype=x.dtype)
    lib = L.load()

    def launch():
        L.check(lib.rssf_conv_gather_add(L.ptr(x), L.ptr(wpk), L.ptr(out), None, None, None, None, B, H, W, C, H, W, C, 1, 1, spec.ntaps,
                                         spec.c_dy, spec.c_dx, L.dtype_code(x), L.stream()), "rssf_conv_gather_add")

    ms = _time_us(launch, iters) / 1e3
    flops = 2.0 * B * H * W * C * C * 19             # the reference's three convolutions: 1 + 9 + 9 kernel positions
    tf = flops / (ms * 1e-3) / 1e12

    # the other two GEMMs of the same layer (one number per direction): the data gradient - the same kernel on the mirrored
    # taps and the transposed pack - and the weight gradient's first stage from the transposed input copy (conv_wgrad_planes.hip; the
    # split-K second stage is the step's ONE batched reduction, not part of this launch)
    import ctypes
    dout = torch.randn(B, H, W, C, device=dev).bfloat16()
    wpk_t = nnf._pack(spec, [c.weight for c in convs], True, x.dtype, x.device)
    dx = torch.empty_like(x)

    def launch_dgrad():
        L.check(lib.rssf_conv_gather_add(L.ptr(dout), L.ptr(wpk_t), L.ptr(dx), None, None, None, None, B, H, W, C, H, W, C, 1, 1, spec.ntaps,
                                         spec.c_ndy, spec.c_ndx, L.dtype_code(x), L.stream()), "rssf_conv_gather_add")

    us_d = _time_us(launch_dgrad, iters)
    family = [dict(direction="data gradient", kernel="conv_taps128_kernel", us_per_launch=round(us_d, 2),
                   frac=round(flops / (us_d * 1e-6) / MFMA_BF16_PEAK, 4))]
    P = nnf.PLANES_PAD
    if lib.rssf_conv_wgrad_planes_supported(B, H, W, C, C, 1, spec.ntaps, spec.c_dy, spec.c_dx, P, L.dtype_code(x)) == 1:
        planes = torch.zeros(C, B, H + 2 * P, W + 2 * P, device=dev, dtype=x.dtype)
        planes[:, :, P:P + H, P:P + W] = x.permute(3, 0, 1, 2)
        dws = [torch.zeros_like(c.weight, dtype=torch.float32) for c in convs]
        ws = torch.empty(lib.rssf_conv_wgrad_workspace_elems(B, H, W, C, C, spec.ntaps), device=dev, dtype=torch.float32)
        job = L.WgradReduceJob()

        def launch_wgrad():
            L.check(lib.rssf_conv_wgrad_planes(L.ptr(dout), L.ptr(planes), P, L.ptr(dws[0]), L.ptr(dws[1]), L.ptr(dws[2]), spec.c_ksizes, 3, spec.c_src,
                                               spec.c_kpos, spec.c_alias, None, L.ptr(ws), B, H, W, C, spec.ntaps, spec.c_dy, spec.c_dx,
                                               ctypes.byref(job), L.dtype_code(x), L.stream()), "rssf_conv_wgrad_planes")

        us_w = _time_us(launch_wgrad, iters)
        family.append(dict(direction="weight gradient (first stage)", kernel="conv_wgrad_planes_kernel", us_per_launch=round(us_w, 2),
                           frac=round(flops / (us_w * 1e-6) / MFMA_BF16_PEAK, 4), partial_planes=int(job.ksplit)))
    return dict(bound="mfma", kernel="conv_taps128_kernel (MlpDWBN fused {1x1 + 3x3 dil 6 + 3x3 dil 12}, %d->%d ch)" % (C, C),
                achieved=round(tf, 1), peak=MFMA_BF16_PEAK / 1e12, unit="TFLOP/s", frac=round(tf * 1e12 / MFMA_BF16_PEAK, 4),
                traffic=_profiled_traffic("conv_taps128_kernel", MLP_SOURCES, "mlp_source_sha16") if (B, S, C) == (16, 512, 128) else None,
                us_per_launch=round(ms * 1e3, 2), algorithmic_flops=flops,
                executed_flops=2.0 * B * H * W * C * C * spec.ntaps,       # the three centre taps share one pixel: 17 taps run
                same_layer=family)


def _cpu_baseline_child(threads, batch, max_steps):
    """Runs in a subprocess: prints one line per finished step so the parent can stop it at its deadline."""
    torch.set_num_threads(threads)
    from oracle import rssformer_cpu as O
    from representationlearning_amd.configs import synthetic_batch
    P = O.default_init_({k: v.clone() for k, v in O.model_template("base").items()})
    for v in P.values():
        if v.is_floating_point():
            v.requires_grad_()
    for k in P:
        if "running" in k:
            P[k].requires_grad_(False)
    x, y = synthetic_batch(batch, 512, device="cpu")
    params = [v for v in P.values() if v.requires_grad]
    opt = torch.optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=1e-4)
    for i in range(max_steps + 1):
        t = time.time()
        opt.zero_grad(set_to_none=True)
        O.model_forward(x, P, True, y).backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 35.0)
        opt.step()
        print("CPU_STEP %d %.4f" % (i, time.time() - t), flush=True)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=150.0, batch=2, warm=3, timed=5):
    """CPU restatement of the same training step (oracle/, plain PyTorch fp32) on the host cores: Base, 512x512, B = 2, 3 warm-up +
    5 timed steps (SURVEY §8d / BASELINE.md §3).  Bounded: a subprocess is given `budget_s` seconds; the timed steps that finished
    by then are averaged (the sample string says how many)."""
    max_steps = warm + timed - 1
    import subprocess
    threads = min(os.cpu_count() or 1, 32)      # oneDNN does not scale past a few dozen threads at this size
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(threads), str(batch), str(max_steps)]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
    times, deadline = [], time.time() + budget_s
    import select
    while time.time() < deadline:
        r, _, _ = select.select([proc.stdout], [], [], max(0.0, min(1.0, deadline - time.time())))
        if r:
            line = proc.stdout.readline()
            if not line:
                break
            if line.startswith("CPU_STEP"):
                _, i, dt = line.split()
                if int(i) >= warm:
                    times.append(float(dt))
        elif proc.poll() is not None:
            break
    if proc.poll() is None:
        proc.kill()
    cpu = "%s, %d logical cores" % (_cpu_model(), os.cpu_count() or 0)
    if not times:
        return dict(value=None, unit="images/s", cores=threads, kind="port", cpu=cpu,
                    sample="no timed step of the CPU oracle (fp32, B=%d, 3x512x512) finished within %.0f s" % (batch, budget_s))
    dt = sum(times) / len(times)
    return dict(value=round(batch / dt, 4), unit="images/s", cores=threads, kind="port", cpu=cpu,
                sample="%d warm-up + %d timed steps of the CPU oracle (fp32, B=%d, 3x512x512, fwd+loss+bwd+clip+SGD), %d threads"
                       % (warm, len(times), batch, threads))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-child":
        _cpu_baseline_child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        return
    # stdout carries the ONE JSON line and nothing else: RCCL prints a five-line version banner to the C stdout of every rank at its
    # first communicator (flushed at exit), so file descriptor 1 is pointed at stderr for the run and the line goes to a private
    # copy of the real stdout
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (BASELINE: 16)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--variant", default="base")
    ap.add_argument("--fp32", action="store_true", help="fp32 activations (parity mode) instead of bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sync-bn", action="store_true")
    args = ap.parse_args()

    from representationlearning_amd import _lib
    from representationlearning_amd.configs import rssformer_config, synthetic_batch
    from representationlearning_amd.core import registry
    from representationlearning_amd.trainer import Trainer, init_distributed
    _lib.load()                                   # fail loudly if the HIP library is missing
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); the CPU oracle is only the baseline leg")
    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    registry.register_all()
    torch.manual_seed(2333)
    model = registry.MODEL["RSSFormer"](rssformer_config(args.variant)).cuda()
    trainer = Trainer(model, bf16=not args.fp32, sync_bn=not args.no_sync_bn)
    img, lab = synthetic_batch(args.batch, args.size, seed=2333 + rank)
    target = dict(cls=lab)

    # building the step: the trainer runs `graph_warmup` eager steps and captures the whole step into a hipGraph on the next one.
    # That is set-up (like compiling), done before the W warm-up steps so that a small --warmup does not put the capture inside the
    # timed region; the count is reported in config.graph_init_steps.
    init_steps = trainer.graph_warmup + 1 if trainer.use_graph else 0
    for _ in range(init_steps):
        trainer.step(img, target)
    for _ in range(args.warmup):
        loss = trainer.step(img, target)
    nch = (1 + len(trainer.side_comms)) if trainer.p2p is not None else 0
    for k in range(nch):                              # (the exchange kernels count what they wait for their peers: start the count here)
        trainer.p2p.wait_us(k, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(img, target)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss)
    # per rank and exchange channel (the step's stream, then the side streams): exchanges per step and the ms per step they spent waiting
    # for their peers' words - rank skew + xGMI write / poll latency, the part of a data-parallel step that is not this rank's own work
    exch = None
    if nch:
        mine = [trainer.p2p.wait_us(k) for k in range(nch)]
        t = torch.tensor([[us / 1e3 / args.steps, n / args.steps] for us, n in mine], device="cuda", dtype=torch.float64)
        allr = [torch.zeros_like(t) for _ in range(world)] if world > 1 else [t]
        if world > 1:
            dist.all_gather(allr, t)
        exch = [{"rank": r, "wait_ms_per_step": [round(float(v[0]), 3) for v in a], "exchanges_per_step": [int(round(float(v[1]))) for v in a]}
                for r, a in enumerate(allr)]

    if rank == 0:
        f0, s0, c0 = VARIANTS[args.variant]
        flop_img = f0 * (args.size / s0) ** 2
        ms = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        line = {
            "metric": "train images/sec RSSFormer-%s %dx%d bf16" % (args.variant.capitalize(), args.size, args.size), "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
            "config": {"workload": "RSSFormer-%s (HRNetV2 + 8 window-attention transformer blocks), %dx3x%dx%d per GPU, 6 classes, "
                                   "fwd+CGFL loss+bwd+clip+SGD, random-init weights" % (args.variant, args.batch, args.size, args.size),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                       "sync_bn": (not args.no_sync_bn) and world > 1,
                       "step_launch": "hipGraph replay" if trainer.graph is not None else "eager", "graph_init_steps": init_steps,
                       "rccl_nranks": (trainer.grad_comm.nranks() if hasattr(trainer.grad_comm, "nranks") else None) if world > 1 else None,
                       "p2p_self_test": None if world == 1 else ("passed" if trainer.p2p is not None else "not run / failed: RCCL exchanges"),
                       "p2p_exchange_wait": exch,
                       "collectives": None if world == 1 else (
                           "SyncBN: %s, one channel / communicator per stream (main + %d side); gradients: %s" % (
                               "peer-to-peer kernel over hipIpc windows" if trainer.p2p is not None else "RCCL all-reduce", len(trainer.side_comms),
                               "RCCL all-reduce of 6 flat buckets beside backward" if getattr(trainer.grad_comm, "direct", False)
                               else "torch.distributed (bucketed, hook-driven)")
                           if getattr(trainer.comm, "direct", False) else "torch.distributed (bucketed, hook-driven)")},
            "final_loss": round(final_loss, 5),
            "whole_step_mfma_frac": round(value * flop_img / (world * MFMA_BF16_PEAK), 5),
        }
        # the kernel rooflines are per-GPU figures: rank 0 measures them at every N (the other ranks wait at the barrier below);
        # the CPU baseline is an N = 1 leg only
        line["roofline"] = measure_window_attention(args.batch, args.size, C=c0)
        line["roofline_mfma"] = measure_mlp_conv(args.batch, args.size, C=4 * c0)
        if world == 1:
            line["roofline_kernels"] = measure_dominant_kernels(args.batch, args.size, C=c0)
            line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline()
        result_out.write(json.dumps(line) + "\n")
        result_out.flush()
    if world > 1:
        dist.barrier()
        trainer.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
