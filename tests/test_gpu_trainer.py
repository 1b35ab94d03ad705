"""GPU: the training step end to end (flat buffers, direct gradient accumulation, fused clip+SGD) and the data-parallel
plumbing on a real RCCL communicator (1-rank group, collectives forced on): same losses as the plain path."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _mk(seed=0):
    from representationlearning_amd.configs import rssformer_config
    from representationlearning_amd.core import registry
    registry.register_all()
    torch.manual_seed(seed)
    return registry.MODEL["RSSFormer"](rssformer_config("base")).cuda()


def _run(trainer, steps=3):
    from representationlearning_amd.configs import synthetic_batch
    img, lab = synthetic_batch(2, 128, seed=5)
    return [float(trainer.step(img, dict(cls=lab))) for _ in range(steps)]


def test_sgd_kernel_matches_torch_sgd():
    """rssf_grad_sqnorm + rssf_sgd_step == clip_grad_norm_(35) + torch.optim.SGD(momentum .9, wd 1e-4), 3 steps."""
    from representationlearning_amd import ops
    torch.manual_seed(0)
    n = 100003
    p = torch.randn(n, device="cuda"); g = [torch.randn(n, device="cuda") * s for s in (0.05, 3.0, 0.5)]
    ref = p.clone().requires_grad_()
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.9, weight_decay=1e-4)
    mom = torch.zeros_like(p); sq = torch.zeros(ops.SQNORM_ELEMS, device="cuda")
    for i, gi in enumerate(g):
        ref.grad = gi.clone() * 0.5
        torch.nn.utils.clip_grad_norm_([ref], 35.0)
        opt.step()
        ops.grad_sqnorm(gi, sq)
        ops.sgd_step_(p, gi, mom, sq, 0.5, 35.0, 0.01, 0.9, 1e-4, i == 0)
    # momentum buffer starts at zero, so the `first_step` special case of torch.optim.SGD equals the general formula
    p2 = torch.randn(n, device="cuda"); p3 = p2.clone(); m2 = torch.zeros_like(p2); m3 = torch.zeros_like(p2)
    lr_dev = torch.tensor([0.01], device="cuda")
    ops.grad_sqnorm(g[0], sq)
    ops.sgd_step_(p2, g[0], m2, sq, 1.0, 35.0, 0.01, 0.9, 1e-4, True)
    ops.sgd_step_(p3, g[0], m3, sq, 1.0, 35.0, 123.0, 0.9, 1e-4, False, lr_dev=lr_dev)
    assert torch.equal(p2, p3)
    assert float((p - ref.detach()).abs().max()) < 1e-5


def test_training_steps_decrease_loss_fp32_and_bf16():
    from representationlearning_amd.trainer import Trainer
    for bf16 in (False, True):
        # small LR: at this tiny size (B=2, 128x128) lr 0.01 makes the first steps noisy
        losses = _run(Trainer(_mk(), bf16=bf16, base_lr=0.002), steps=8)
        assert all(l == l for l in losses) and min(losses[-3:]) < losses[0], losses


def test_direct_gradients_equal_autograd_gradients():
    """The trainer's kernels accumulate parameter gradients straight into the flat `.grad` buffer (nnf.grad_target: no
    AccumulateGrad adds; the attention node hands the two 7x7 gate kernels, the mixing weight and bias to
    rssf_gate_weights_bwd as four destinations).  Same numbers as the plain autograd route of the same model, where every
    node returns its gradients and autograd accumulates them.  (Deterministic statistics on both sides: this tiny
    configuration amplifies the last-bit noise of atomically summed BatchNorm statistics to per cents.)"""
    from representationlearning_amd.trainer import Trainer
    from representationlearning_amd.configs import synthetic_batch
    from representationlearning_amd import nnf
    from tests.helpers import rel_err
    img, lab = synthetic_batch(2, 128, seed=5)
    m = _mk(4)
    rt = nnf.Runtime()
    rt.deterministic = True
    with nnf.use(rt):
        out = m(img, dict(cls=lab))
        sum(v for k, v in out.items() if k.endswith("loss")).backward()
    plain = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
    t = Trainer(_mk(4), bf16=False, base_lr=0.0, weight_decay=0.0, use_graph=False, deterministic=True)
    t.step(img, dict(cls=lab))
    names = {id(p): n for n, p in t.model.named_parameters()}
    direct = {names[id(p)]: t.flat.grad[o:o + p.numel()].view_as(p) for p, o in zip(t.flat.params, t.flat.offsets)}
    keys = sorted(n for n in plain if n in direct)
    gate = [n for n in keys if "atrous_block" in n or "weight_levels" in n]
    assert len(keys) > 1000 and len(gate) == 8 * 4, (len(keys), len(gate))
    cat = lambda d, ks: torch.cat([d[n].flatten() for n in ks]).cpu()
    assert rel_err(cat(direct, keys), cat(plain, keys)) < 1e-6
    assert rel_err(cat(direct, gate), cat(plain, gate)) < 1e-5


@pytest.mark.parametrize("bf16", [False, True])
def test_deterministic_mode_is_bit_identical(bf16):
    """RSSF_DETERMINISTIC / Trainer(deterministic=True): fixed-order BatchNorm statistics (conv epilogue partials + ordered
    fold, backward reduce likewise) and a one-block-per-sample loss reduction.  Two independent runs give the SAME bits for the
    loss and every BatchNorm running statistic; parameter gradients agree to fp32 rounding (a few of their reductions still
    use float atomics: LayerNorm / attention / gate parameter sums - they do not feed back into the step)."""
    from representationlearning_amd.trainer import Trainer
    from tests.helpers import rel_err
    runs = []
    for _ in range(2):
        t = Trainer(_mk(3), bf16=bf16, base_lr=0.0, weight_decay=0.0, use_graph=False, deterministic=True)
        losses = _run(t, steps=2)
        bufs = torch.cat([b.detach().flatten().float() for b in t.model.buffers() if b.is_floating_point()])
        runs.append((losses, t.flat.grad.clone(), bufs))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][2], runs[1][2])
    assert rel_err(runs[0][1].cpu(), runs[1][1].cpu()) < 1e-6


@pytest.mark.parametrize("backend", ["torch", "rccl", "rccl+graph"])
def test_dp_plumbing_on_one_rank_matches_plain_path(backend):
    """SyncBN exchanges + bucketed gradient all-reduce over RCCL (world 1) must reproduce the plain single-GPU step, through
    torch.distributed (hook-driven async buckets), through the direct rssf_comm_* communicators (buckets on a side stream),
    and with that step captured into a hipGraph.  Deterministic statistics, lr = 0: the loss must agree to fp32 rounding and
    the all-reduced flat gradient to 1e-5."""
    from representationlearning_amd.trainer import Trainer
    from tests.helpers import rel_err
    t0 = Trainer(_mk(1), bf16=False, base_lr=0.0, weight_decay=0.0, use_graph=False, deterministic=True)
    plain = _run(t0, steps=2)
    g_plain = t0.flat.grad.clone()
    graph = backend.endswith("+graph")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RSSF_FORCE_DP="1", RSSF_DP_BACKEND=backend.split("+")[0],
                      RSSF_GRAPH="1" if graph else "0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    tr = None
    try:
        tr = Trainer(_mk(1), bf16=False, sync_bn=True, base_lr=0.0, weight_decay=0.0, deterministic=True)
        assert tr.buckets is not None and len(tr.buckets.bounds) >= 4
        if backend == "torch":
            assert not tr.comm.direct
        else:
            assert tr.comm.direct and tr.grad_comm.direct and tr.grad_comm is not tr.comm and tr.buckets.side is not None
        dp = _run(tr, steps=6 if graph else 2)
        if not graph:
            assert all(tr.buckets.launched)
        if graph:
            assert tr.graph is not None and tr._replayed >= 2
        g_dp = tr.flat.grad.clone()
    finally:
        if tr is not None:
            tr.close()
        dist.destroy_process_group()
        for k in ("RSSF_FORCE_DP", "RSSF_DP_BACKEND", "RSSF_GRAPH"):
            os.environ.pop(k)
    assert max(abs(a - plain[0]) for a in dp) < 1e-6 * abs(plain[0]), (plain, dp)
    assert plain[0] == plain[1]                                     # lr = 0, deterministic: the step is a fixed point
    assert rel_err(g_dp.cpu(), g_plain.cpu()) < 1e-5, rel_err(g_dp.cpu(), g_plain.cpu())


def _failed_capture_worker():
    import contextlib
    import io
    from representationlearning_amd.trainer import Trainer
    t0 = Trainer(_mk(1), bf16=False, base_lr=0.0, weight_decay=0.0, use_graph=False, deterministic=True)
    plain = _run(t0, steps=2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RSSF_FORCE_DP="1", RSSF_DP_BACKEND="rccl", RSSF_GRAPH="1",
                      RSSF_TEST_FAIL_CAPTURE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    out = io.StringIO()
    tr = Trainer(_mk(1), bf16=False, sync_bn=True, base_lr=0.0, weight_decay=0.0, deterministic=True)
    assert tr.use_graph and tr.rt.branch_streams
    with contextlib.redirect_stdout(out):
        dp = _run(tr, steps=7)                     # 3 warm-up, the failed capture + its eager step, 3 more eager steps
    assert tr.graph is None and not tr.use_graph and not tr.rt.branch_streams
    torch.cuda.synchronize()
    x = torch.empty(1 << 28, device="cuda")        # a fresh allocation and a synchronising call still work (no stream is stuck capturing)
    del x
    assert not torch.cuda.is_current_stream_capturing()
    assert "hipGraph capture failed" in out.getvalue(), out.getvalue()
    assert max(abs(a - plain[0]) for a in dp) < 1e-6 * abs(plain[0]), (plain, dp)
    tr.close()
    dist.destroy_process_group()


def test_failed_capture_falls_back_to_working_eager_steps():
    """An exception in the middle of the captured step (injected inside a gradient bucket's launch: side streams forked, nothing
    joined) must end the capture cleanly: the trainer says so, drops the graph and keeps training with eager launches on healthy
    streams - same losses as a trainer that never tried to capture.  In a process of its own: a graph that was captured with
    parallel branches and thrown away unlaunched is one more way into the hip::Graph::UpdateStreams crash of a LATER graph launch
    (DESIGN lesson 27) - seen when this ran inside the full suite; a trainer whose capture failed launches no graph again."""
    import torch.multiprocessing as mp
    p = mp.get_context("spawn").Process(target=_failed_capture_worker)
    p.start()
    p.join(600)
    assert p.exitcode == 0, "worker failed (exit code %s)" % p.exitcode


def test_graph_replay_matches_eager():
    """The captured whole-step hipGraph reproduces the eager step (lr = 0 so every step sees the same parameters;
    deterministic statistics so the comparison is not blurred by atomics order)."""
    from representationlearning_amd.trainer import Trainer, flush_bn_counters
    from representationlearning_amd.configs import synthetic_batch
    from tests.helpers import rel_err
    img, lab = synthetic_batch(2, 128, seed=5)
    te = Trainer(_mk(2), bf16=False, base_lr=0.0, weight_decay=0.0, use_graph=False, deterministic=True)
    le = [float(te.step(img, dict(cls=lab))) for _ in range(2)]
    ge = te.flat.grad.clone()
    tg = Trainer(_mk(2), bf16=False, base_lr=0.0, weight_decay=0.0, use_graph=True, deterministic=True)
    lg = [tg.step(img, dict(cls=lab)) for _ in range(6)]       # 3 eager warm-up + capture + 2 replays
    assert tg.graph is not None and tg._replayed >= 2
    assert lg[-1].data_ptr() != lg[-2].data_ptr()              # replays hand out copies, not the static loss tensor
    lg = [float(l) for l in lg]
    assert abs(lg[-1] - le[0]) < 1e-6 * abs(le[0]), (lg, le)
    assert rel_err(tg.flat.grad.cpu(), ge.cpu()) < 1e-5
    # num_batches_tracked counts executed steps only: the Python pass that records the graph does not count
    flush_bn_counters(tg)
    assert int(tg.model.backbone.hrnet.bn1.num_batches_tracked) == 6
    # a different batch through the replayed graph gives a different loss (static input buffers are refreshed)
    img2, lab2 = synthetic_batch(2, 128, seed=6)
    l2 = float(tg.step(img2, dict(cls=lab2)))
    assert abs(l2 - lg[-1]) > 1e-4
    # a batch of another shape cannot go through the captured step: it runs eagerly instead of broadcasting into the buffers
    img3, lab3 = synthetic_batch(1, 128, seed=7)
    l3 = float(tg.step(img3, dict(cls=lab3)))
    assert l3 == l3 and tg._replayed == 4


def test_sgd_skips_parameters_without_gradient():
    """torch.optim.SGD leaves a parameter whose .grad is None untouched (no weight decay, no momentum): `headaux` never
    receives a gradient (the loss uses it under no_grad, module/CGFL.py:75-97), so it must keep its initial values."""
    from representationlearning_amd.trainer import Trainer
    m = _mk(4)
    w0 = m.headaux[0].weight.detach().clone()
    c0 = m.backbone.hrnet.conv1.weight.detach().clone()
    t = Trainer(m, bf16=False, base_lr=0.01, weight_decay=0.1, use_graph=False)
    _run(t, steps=2)
    assert torch.equal(m.headaux[0].weight.detach(), w0)
    assert not torch.equal(m.backbone.hrnet.conv1.weight.detach(), c0)
    covered = sum(e - s for s, e in t.sgd_ranges)
    assert covered == t.flat.numel - (t.flat.offsets[-1] - t.flat.offsets[-3])       # everything but headaux.0.{weight,bias}


def test_replayed_step_repeats_under_its_own_concurrency():
    """One captured step (branches and forked fuse outputs as parallel graph branches), replayed 120 times from an unchanged state
    (lr = 0, deterministic statistics): the loss has the same bits every time and EVERY ELEMENT of the flat gradient moves by fp32
    summation noise only (a few parameter sums use float atomics: 1e-8 at most).  Round 2 found a kernel that dropped one term of a
    7x7 convolution in 5-15 % of such replays - and in no eager or single-stream run (DESIGN.md lesson 23; tools/replay_race.py is the
    diagnostic form of this test); round 6 found the second one - MlpDWBN's fc1 bias gradient (conv_wgrad_pw_kernel) 1e-4 off in
    3-5 % of the replays, the same crossed-operand packed instruction (lesson 59; tools/replay_param_noise.py) - which the norm-wise
    comparison this test used to make could not see: 16 elements of 40 million."""
    from representationlearning_amd.trainer import Trainer
    from representationlearning_amd.configs import synthetic_batch
    from tests.helpers import rel_err
    img, lab = synthetic_batch(2, 128, seed=5)
    t = Trainer(_mk(6), bf16=True, base_lr=0.0, weight_decay=0.0, use_graph=True, deterministic=True)
    while t.graph is None or t._replayed < 2:
        l0 = t.step(img, dict(cls=lab)).clone()
    torch.cuda.synchronize()
    g0 = t.flat.grad.clone()
    worst, worst_abs = 0.0, 0.0
    for r in range(120):
        l = t.step(img, dict(cls=lab))
        torch.cuda.synchronize()
        assert torch.equal(l, l0), (r, float(l), float(l0))
        if r % 4 == 0:
            worst = max(worst, rel_err(t.flat.grad.cpu(), g0.cpu()))
        d = float((t.flat.grad - g0).abs().max())
        if d > 1e-6:
            i = int((t.flat.grad - g0).abs().argmax())
            k = max(j for j, o in enumerate(t.flat.offsets[:-1]) if o <= i)
            name = [n for n, p in t.model.named_parameters() if p is t.flat.params[k]]
            raise AssertionError("replay %d: gradient element %d (%s) moved by %.3g" % (r, i, name, d))
        worst_abs = max(worst_abs, d)
    assert worst < 1e-5, worst


@pytest.mark.parametrize("branch_streams", ["1", "0"])
def test_graph_replay_survives_device_sync_and_foreign_work(branch_streams):
    """A replayed step must not depend on anything outside the graph: a device synchronisation and unrelated allocations / kernels
    between replays leave the loss sequence where eager launches put it.  (Round 1's step kept a hipBLASLt GEMM - the aux head's
    Linear - inside the capture; after a torch.cuda.synchronize() its replays returned garbage: the loss jumped from 1.75 to 0.99,
    with the label check of this round to NaN.  Single-stream captures - the data-parallel configuration - hit it every time.)"""
    from representationlearning_amd.trainer import Trainer
    from representationlearning_amd.configs import synthetic_batch
    img, lab = synthetic_batch(2, 128, seed=5)

    def attempt():
        os.environ["RSSF_BRANCH_STREAMS"] = branch_streams
        try:
            te = Trainer(_mk(6), bf16=True, base_lr=0.002, use_graph=False, deterministic=True)
            le = [float(te.step(img, dict(cls=lab))) for _ in range(9)]
            tg = Trainer(_mk(6), bf16=True, base_lr=0.002, use_graph=True, deterministic=True)
            lg = []
            for i in range(9):
                if i >= 4:
                    torch.cuda.synchronize()
                    junk = [torch.randn(1 << 20, device="cuda").square().sum() for _ in range(4)]       # foreign allocations + kernels
                    del junk
                lg.append(float(tg.step(img, dict(cls=lab))))
        finally:
            os.environ.pop("RSSF_BRANCH_STREAMS")
        assert tg.graph is not None and tg._replayed >= 5
        assert all(l == l for l in lg), lg
        return max(abs(a - b) / abs(b) for a, b in zip(lg, le)), lg, le

    # A dependence on state outside the graph fails EVERY time (round 1's did).  Once in ~6 runs of the whole suite (round 5; twice in a
    # row in one run of round 6) a replay came back 1e-4 off and the training steps amplified it to 2.5 %: root-caused in round 6 - not
    # the foreign work but MlpDWBN's fc1 bias gradient, wrong in 3-5 % of ALL multi-queue replays (DESIGN.md lesson 59; fixed, and
    # test_replayed_step_repeats_under_its_own_concurrency now looks at every gradient element).  With the fix the replayed trajectory
    # leaves the eager one where two eager runs leave each other (step 22-54 of 60, the fp32-atomics floor; tools/replay_poison.py);
    # the second attempt stays as a fence.
    worst, lg, le = attempt()
    if worst >= 2e-3:
        print("replay vs eager %.3g on the first attempt; running once more" % worst)
        worst, lg, le = attempt()
    assert worst < 2e-3, (lg, le)


@pytest.mark.parametrize("bf16,deterministic,batch,size", [(True, True, 2, 128), (False, True, 2, 128), (True, False, 2, 256), (True, False, 1, 512)])
def test_step_does_not_read_uninitialised_lds(bf16, deterministic, batch, size):
    """Every kernel of the step with the LDS of all CUs poisoned (rssf_debug_poison_lds: NaN bit patterns, then zeros) right before
    the step: loss and gradients of a deterministic eager step must come out bit-identical to the unpoisoned run.  A kernel whose result
    depends on LDS it never wrote (a pad column that enters an MFMA, a fold buffer read past what was written) would pass every parity
    test until a different kernel ran before it on the same CU - inside a replayed step or after foreign work.  Deterministic mode:
    identical losses; the benchmark's kernel set (statistics links, the MLP's register-operand / planes / point-wise stream kernels at
    64^2 and 128^2 maps): finite and within the run-to-run floor of two clean runs."""
    from representationlearning_amd import _lib as L
    from representationlearning_amd.trainer import Trainer
    from representationlearning_amd.configs import synthetic_batch
    from tests.helpers import rel_err
    lib = L.load()
    img, lab = synthetic_batch(batch, size, seed=5)
    scratch = torch.zeros(1, device="cuda", dtype=torch.int32)
    res = []
    for pattern in (None, None, 0x7fc07fc0, 0x00000000, 0xffffffff):
        t = Trainer(_mk(6), bf16=bf16, base_lr=0.0, use_graph=False, deterministic=deterministic)
        losses = []
        for _ in range(2):
            if pattern is not None:
                L.check(lib.rssf_debug_poison_lds(pattern, L.ptr(scratch), L.stream()), "rssf_debug_poison_lds")
            losses.append(float(t.step(img, dict(cls=lab))))
        torch.cuda.synchronize()
        res.append((losses, t.flat.grad.clone()))
    # (deterministic mode fixes the statistics and the loss; a few gradient sums still go through atomics: two clean runs give the floor)
    floor = rel_err(res[1][1].cpu(), res[0][1].cpu())
    if deterministic:
        assert res[1][0] == res[0][0] and floor < 1e-5, (res[1][0], res[0][0], floor)
    for (l, g) in res[2:]:
        if deterministic:
            assert l == res[0][0], (l, res[0][0])
        else:       # (a random-init network amplifies the atomics' rounding noise: the four clean losses give the spread)
            clean = res[0][0] + res[1][0]
            spread = (max(clean) - min(clean)) / min(clean)
            assert all(x == x and abs(x - clean[0]) <= max(5 * spread, 3e-2) * abs(clean[0]) for x in l), (l, clean)
        err = rel_err(g.cpu(), res[0][1].cpu())
        assert bool(torch.isfinite(g).all()) and err <= max(3 * floor, 1e-6), (err, floor)
