"""GPU parity: gate + fused window cross-attention (HIP, through the C ABI) vs the CPU oracle and the
golden vectors of the reference.  Runs on the MI355X only (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import rssformer_cpu as O
from oracle.procedural import proc_input
from tests.helpers import golden, proc_params, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"

F32_TOL = 2e-4     # fp32-I/O mode (exact f32 MFMA): rel. Frobenius error vs CPU fp32
BF16_TOL = 2.5e-2  # bf16-I/O mode: storage rounding of activations + bf16 MFMA operands


def _attn_params(C):
    t = {k[len("attn."):]: v for k, v in O.block_template(C).items() if k.startswith("attn.")}
    P = proc_params(t, requires_grad=False)
    ln = proc_params({"norm1.weight": torch.empty(C), "norm1.bias": torch.empty(C)}, requires_grad=False)
    return P, ln


def _dev_weights(P):
    w = {}
    for n in ("q", "k", "v", "out"):
        key = {"q": "wq", "k": "wk", "v": "wv", "out": "wo"}[n]
        w[key] = P[f"attn.{n}_proj.weight"].to(DEV).contiguous()
        w["b" + key[1]] = P[f"attn.{n}_proj.bias"].to(DEV).contiguous()
    return w


def _gate_k(P):
    return torch.stack([P["atrous_block1.conv1.weight"][0], P["atrous_block2.conv1.weight"][0]]).to(DEV).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mma_helpers(dtype):
    from representationlearning_amd import ops
    torch.manual_seed(0)
    K = 32
    a = torch.randn(16, K).to(dtype)
    b = torch.randn(16, K).to(dtype)       # asymmetric, non-square-symmetric operands
    d = ops.debug_mma(a.to(DEV), b.to(DEV)).cpu()
    af, bf = a.float(), b.float()
    E = af @ bf.T
    assert rel_err(d[0], E) < 1e-5
    Ek = E.to(dtype).float()
    F = bf[:, :16] @ Ek
    G = Ek.T @ Ek
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(d[1], F) < tol
    assert rel_err(d[2], G) < tol


@pytest.mark.parametrize("rows,C", [(1000, 32), (777, 18), (64, 48), (4096, 128), (3001, 48), (515, 24), (1030, 96)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm(rows, C, dtype):
    from representationlearning_amd import ops
    torch.manual_seed(1)
    x = torch.randn(rows, C)
    g, b = torch.randn(C), torch.randn(C)
    dy = torch.randn(rows, C)
    xd = x.to(dtype)
    xr = xd.float().requires_grad_()
    gr, br = g.clone().requires_grad_(), b.clone().requires_grad_()
    yr = torch.nn.functional.layer_norm(xr, (C,), gr, br, 1e-6)
    yr.backward(dy.to(dtype).float())
    y, st = ops.layernorm_fwd(xd.to(DEV), g.to(DEV), b.to(DEV))
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel_err(y.float().cpu(), yr.detach()) < tol
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.layernorm_bwd(dy.to(dtype).to(DEV), xd.to(DEV), st, g.to(DEV), dg, db)
    assert rel_err(dx.float().cpu(), xr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    assert rel_err(dg.cpu(), gr.grad) < 1e-4
    assert rel_err(db.cpu(), br.grad) < 1e-4


CASES = [(1, 32, 7, 7), (2, 32, 10, 10), (1, 32, 14, 14), (1, 32, 20, 12), (1, 18, 9, 11), (1, 48, 8, 8)]


def _run_forward(B, C, H, W, dtype):
    """Returns (out_hip [B,N,C] fp32 cpu, logits_hip, oracle pieces)."""
    from representationlearning_amd import ops
    P, ln = _attn_params(C)
    x = proc_input((B, H * W, C), 0.2)
    y = proc_input((B, H * W, C), 0.8)
    xd, yd = x.to(dtype).to(DEV), y.to(dtype).to(DEV)
    g, b = ln["norm1.weight"].to(DEV), ln["norm1.bias"].to(DEV)
    _, sx = ops.layernorm_fwd(xd, g, b, want_y=False)
    _, sy = ops.layernorm_fwd(yd, g, b, want_y=False)
    pooled, amax = ops.gate_pool_fwd(xd, yd, sx, sy, g, b)
    gsig, omega, logits = ops.gate_weights_fwd(pooled, _gate_k(P), P["weight_levels.weight"].reshape(2, 2).to(DEV).contiguous(),
                                               P["weight_levels.bias"].to(DEV), H, W, want_logits=True)
    out = ops.winattn_fwd(xd, yd, sx, sy, omega, g, b, _dev_weights(P), H, W, 2)
    torch.cuda.synchronize()
    # oracle on the same (dtype-rounded) inputs
    xr, yr = x.to(dtype).float(), y.to(dtype).float()
    lnf = lambda t: torch.nn.functional.layer_norm(t, (C,), ln["norm1.weight"], ln["norm1.bias"], 1e-6)
    _, _, _, lv = O.gate(lnf(xr), lnf(yr), P, "", H, W)
    ref = xr + O.interlaced_attention(lnf(xr), lnf(yr), P, "", H, W)
    return out.float().cpu(), logits.cpu().reshape(B, 2, H, W), ref, lv, (xr, P, ln)


@pytest.mark.parametrize("B,C,H,W", CASES)
def test_winattn_fwd_fp32_vs_oracle_and_golden(B, C, H, W):
    out, logits, ref, lv, (xr, P, ln) = _run_forward(B, C, H, W, torch.float32)
    assert rel_err(logits, lv) < 1e-5
    assert rel_err(out, ref) < F32_TOL
    # golden (reference itself): LN is applied inside our kernel, the golden case fed raw tokens -> compare the
    # attention term on LN-free inputs by choosing identity LN?  The golden used x,y directly as LN outputs, so run
    # the gate/attention with gamma=1,beta=0 stats {0,1}:
    from representationlearning_amd import ops
    g = golden(f"attn_B{B}_C{C}_H{H}_W{W}")
    x = proc_input((B, H * W, C), 0.2).to(DEV)
    y = proc_input((B, H * W, C), 0.8).to(DEV)
    one, zero = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    st = torch.tensor([0.0, 1.0], device=DEV).repeat(B * H * W, 1).contiguous()
    pooled, _ = ops.gate_pool_fwd(x, y, st, st, one, zero)
    _, omega, lg = ops.gate_weights_fwd(pooled, _gate_k(P), P["weight_levels.weight"].reshape(2, 2).to(DEV).contiguous(),
                                        P["weight_levels.bias"].to(DEV), H, W, want_logits=True)
    o2 = ops.winattn_fwd(x, y, st, st, omega, one, zero, _dev_weights(P), H, W, 2)
    assert rel_err(lg.cpu().reshape(B, 2, H, W), g["gate_logits"]) < 1e-5
    assert rel_err((o2 - x).cpu(), g["out"]) < F32_TOL


@pytest.mark.parametrize("B,C,H,W", CASES)
def test_winattn_fwd_bf16(B, C, H, W):
    out, logits, ref, lv, _ = _run_forward(B, C, H, W, torch.bfloat16)
    assert rel_err(logits, lv) < 1e-4          # gate maps are fp32 end to end
    assert rel_err(out, ref) < BF16_TOL


def test_winattn_base_shape_properties():
    """BASELINE config-2 geometry (B=16, C=32, 128x128): window independence + finite outputs at full size."""
    from representationlearning_amd import ops
    torch.manual_seed(3)
    B, C, H, W = 16, 32, 128, 128
    P, ln = _attn_params(C)
    x = torch.randn(B, H * W, C, device=DEV, dtype=torch.bfloat16)
    y = torch.randn(B, H * W, C, device=DEV, dtype=torch.bfloat16)
    g, b = ln["norm1.weight"].to(DEV), ln["norm1.bias"].to(DEV)
    _, sx = ops.layernorm_fwd(x, g, b, want_y=False)
    _, sy = ops.layernorm_fwd(y, g, b, want_y=False)
    omega = torch.full((B, 2, H * W), 0.5, device=DEV)
    w = _dev_weights(P)
    out = ops.winattn_fwd(x, y, sx, sy, omega, g, b, w, H, W, 2)
    assert torch.isfinite(out.float()).all()
    # perturbing one token of `high` changes only the outputs of its own 7x7 window (pad 2 before)
    y2 = y.clone()
    u, v = 40, 77
    y2[3, u * W + v] += 1.0
    _, sy2 = ops.layernorm_fwd(y2, g, b, want_y=False)
    out2 = ops.winattn_fwd(x, y2, sx, sy2, omega, g, b, w, H, W, 2)
    diff = (out2.float() - out.float()).abs().reshape(B, H, W, C).amax(-1)
    assert diff[[0, 1, 2] + list(range(4, 16))].max() == 0
    qh, qw = (u + 2) // 7, (v + 2) // 7
    mask = torch.zeros(H, W, dtype=torch.bool, device=DEV)
    mask[max(0, qh * 7 - 2): qh * 7 + 5, max(0, qw * 7 - 2): qw * 7 + 5] = True
    assert diff[3][~mask].max() == 0 and diff[3][mask].max() > 0
    # batch-permutation equivariance
    perm = torch.randperm(B, device=DEV)
    outp = ops.winattn_fwd(x[perm].contiguous(), y[perm].contiguous(), sx.view(B, -1, 2)[perm].reshape(-1, 2).contiguous(),
                           sy.view(B, -1, 2)[perm].reshape(-1, 2).contiguous(), omega, g, b, w, H, W, 2)
    assert torch.equal(outp, out[perm])


@pytest.mark.parametrize("C,H,W,dtype", [(32, 16, 16, torch.bfloat16), (48, 14, 21, torch.bfloat16), (18, 7, 7, torch.float32)])
def test_winattn_bwd_workspace_route_matches_atomic_route(C, H, W, dtype):
    """rssf_winattn_bwd with prod_ws (products + reduce launch, domega overwritten) and without (atomics into a zeroed
    domega) are the same sums in a different order (fp32 mode; in bf16 mode the scratch products are bf16): every other output is
    bit-identical, domega agrees to the rounding of its summands."""
    from representationlearning_amd import ops
    B, N = 2, H * W
    P, ln = _attn_params(C)
    x = proc_input((B, N, C), 0.2).to(dtype).to(DEV)
    y = proc_input((B, N, C), 0.8).to(dtype).to(DEV)
    dout = proc_input((B, N, C), 1.7).to(dtype).to(DEV)
    g, b = ln["norm1.weight"].to(DEV), ln["norm1.bias"].to(DEV)
    _, sx = ops.layernorm_fwd(x, g, b, want_y=False)
    _, sy = ops.layernorm_fwd(y, g, b, want_y=False)
    omega = (proc_input((B, 2, N), 0.5).abs() + 0.25).to(DEV).contiguous()
    w = _dev_weights(P)
    res = []
    for workspace in (True, False):
        gw = {k: torch.zeros_like(v) for k, v in w.items()}
        dxh, dyh, dom = ops.winattn_bwd(dout, x, y, sx, sy, omega, g, b, w, gw, H, W, 2, workspace=workspace)
        torch.cuda.synchronize()
        res.append((dxh, dyh, dom, gw))
    (dx0, dy0, dom0, gw0), (dx1, dy1, dom1, gw1) = res
    assert torch.equal(dx0, dx1) and torch.equal(dy0, dy1)
    assert torch.isfinite(dom0).all() and dom0.abs().max() > 0
    # the workspace route keeps the products in the activation dtype: bf16 rounding of each of the C summands in bf16 mode
    assert rel_err(dom0.cpu(), dom1.cpu()) < (1e-5 if dtype == torch.float32 else 5e-3)
    for k in gw0:
        assert rel_err(gw0[k].cpu(), gw1[k].cpu()) < 1e-5, k


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 19, 23), (3, 7, 7)])
def test_gate_weights_backward_vs_autograd_reference(B, H, W):
    """rssf_gate_weights_bwd (omega = softmax_2(Wl [sigmoid(conv7x7(pooled_0; k_0)); sigmoid(conv7x7(pooled_1; k_1))] + bl),
    multihead_isa_pool_attention.py:30-37) against autograd through the same formula in fp32: dpooled, dk, dwl, dbl.  The two 7x7
    kernels' gradients go to one [2,2,7,7] buffer or to two separate ones (`dk_stream1`): same numbers."""
    from representationlearning_amd import ops
    g = torch.Generator().manual_seed(B * 100 + H)
    N = H * W
    pooled = torch.randn(B, 4, N, generator=g).to(DEV)
    k = (torch.randn(2, 2, 7, 7, generator=g) * 0.2).to(DEV)
    wl = torch.randn(2, 2, generator=g).to(DEV)
    bl = torch.randn(2, generator=g).to(DEV)
    domega = torch.randn(B, 2, N, generator=g).to(DEV)
    gsig, omega, _ = ops.gate_weights_fwd(pooled, k, wl, bl, H, W)
    # reference
    pr, kr, wr, br = (t.clone().requires_grad_(True) for t in (pooled, k, wl, bl))
    maps = pr.view(B, 2, 2, H, W)
    gs = torch.stack([torch.sigmoid(torch.nn.functional.conv2d(maps[:, s], kr[s:s + 1], padding=3))[:, 0] for s in range(2)], 1)   # [B,2,H,W]
    logits = torch.einsum("os,bshw->bohw", wr, gs) + br.view(1, 2, 1, 1)
    om = torch.softmax(logits, 1).reshape(B, 2, N)
    assert rel_err(omega.cpu(), om.detach().cpu()) < 1e-5
    (om * domega).sum().backward()
    dk, dwl, dbl = torch.zeros_like(k), torch.zeros_like(wl), torch.zeros_like(bl)
    dpooled = ops.gate_weights_bwd(domega, pooled, gsig, omega, k, wl, dk, dwl, dbl, H, W).clone()
    torch.cuda.synchronize()
    assert rel_err(dpooled.cpu(), pr.grad.cpu()) < 1e-4
    assert rel_err(dk.cpu(), kr.grad.cpu()) < 1e-4
    assert rel_err(dwl.cpu(), wr.grad.cpu()) < 1e-4 and rel_err(dbl.cpu(), br.grad.cpu()) < 1e-4
    # two destinations
    dk0, dk1 = torch.zeros(2, 7, 7, device=DEV), torch.zeros(2, 7, 7, device=DEV)
    dwl2, dbl2 = torch.zeros_like(wl), torch.zeros_like(bl)
    dp2 = ops.gate_weights_bwd(domega, pooled, gsig, omega, k, wl, dk0, dwl2, dbl2, H, W, dk_stream1=dk1).clone()
    assert torch.equal(dp2, dpooled)
    assert rel_err(torch.stack([dk0, dk1]).cpu(), dk.cpu()) < 1e-6 and rel_err(dwl2.cpu(), dwl.cpu()) < 1e-6


def _gate_bwd_under_load_worker(B, H, W):
    from representationlearning_amd import ops
    g = torch.Generator().manual_seed(B * 100 + H)
    N = H * W
    pooled = torch.randn(B, 4, N, generator=g).to(DEV)
    k = (torch.randn(2, 2, 7, 7, generator=g) * 0.2).to(DEV)
    wl = torch.randn(2, 2, generator=g).to(DEV)
    bl = torch.randn(2, generator=g).to(DEV)
    domega = torch.randn(B, 2, N, generator=g).to(DEV)
    gsig, omega, _ = ops.gate_weights_fwd(pooled, k, wl, bl, H, W)
    dpooled = ops.gate_weights_bwd(domega, pooled, gsig, omega, k, wl, torch.zeros_like(k), torch.zeros_like(wl), torch.zeros_like(bl), H, W).clone()
    side = torch.cuda.Stream()
    junk = torch.randn(64, 1 << 16, device=DEV)
    for r in range(10):
        with torch.cuda.stream(side):
            for _ in range(4):
                junk.mul_(1.0001)
        d = ops.gate_weights_bwd(domega, pooled, gsig, omega, k, wl, torch.zeros_like(k), torch.zeros_like(wl), torch.zeros_like(bl), H, W)
        assert torch.equal(d, dpooled), r
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 19, 23)])
def test_gate_weights_backward_repeats_under_load(B, H, W):
    """Ten launches in a row with a second stream keeping the CUs busy give the same bits (DESIGN.md lesson 23).  In a process of
    its own: on this ROCm a process that has run kernels on the default stream BESIDE a user stream later crashes inside
    hip::Graph::UpdateStreams when it replays a captured step with parallel branches (`pytest tests/test_gpu_attention.py
    tests/test_gpu_trainer.py` died in test_graph_replay_matches_eager with this loop in-process, round 3's code included; without
    it 46 + 14 tests pass - DESIGN.md lesson 27)."""
    import torch.multiprocessing as mp
    p = mp.get_context("spawn").Process(target=_gate_bwd_under_load_worker, args=(B, H, W))
    p.start()
    p.join(300)
    assert p.exitcode == 0, "worker failed (exit code %s)" % p.exitcode


@pytest.mark.parametrize("B,H,W,C,dtype", [(2, 16, 16, 32, torch.bfloat16), (1, 24, 32, 32, torch.float32), (3, 8, 8, 16, torch.bfloat16), (16, 128, 128, 32, torch.bfloat16),
                                            (1, 8, 8, 64, torch.bfloat16)])
def test_fused_ln_statistics_and_gate_pooling_equal_the_three_launches(B, H, W, C, dtype):
    """rssf_ln_gate_pool_fwd (norm1's statistics of both token streams formed inside the gate's pooling pass: one read of x and y
    instead of two) against rssf_layernorm_fwd x 2 + rssf_gate_pool_fwd: the statistics and the pooled maps agree to the last bit or
    one ulp, the arg-max indices agree; shapes the fused launch does not take say so."""
    from representationlearning_amd import ops, _lib as L
    torch.manual_seed(5)
    N = H * W
    x = (torch.randn(B, N, C, device=DEV) * 1.3 + 0.2).to(dtype)
    y = (torch.randn(B, N, C, device=DEV) * 0.7 - 0.1).to(dtype)
    g, b = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.2
    fused = ops.ln_gate_pool_fwd(x, y, g, b)
    assert fused is not None
    _, sx = ops.layernorm_fwd(x, g, b, want_y=False)
    _, sy = ops.layernorm_fwd(y, g, b, want_y=False)
    pooled, argmax = ops.gate_pool_fwd(x, y, sx, sy, g, b)
    # the same arithmetic in the same order; the two translation units differ in the last bit here and there (1 ulp in ~8 % of the
    # statistics: fused multiply-add choices), an arg-max may only move between view-channels that tie to that precision
    for got, want, name in zip(fused[:3], (sx, sy, pooled), ("stats_x", "stats_y", "pooled")):
        assert torch.allclose(got, want, rtol=3e-7, atol=1e-6), (name, float((got - want).abs().max()))
    assert float((fused[3] == argmax).float().mean()) > 0.999
    lib = L.load()
    assert lib.rssf_ln_gate_pool_fwd_supported(2, 100, 48, L.RSSF_BF16) == 0          # N % C != 0
    assert lib.rssf_ln_gate_pool_fwd_supported(2, 48 * 16, 48, L.RSSF_BF16) == 0      # six vectors per row: not a DPP group
    assert lib.rssf_ln_gate_pool_fwd_supported(2, 18 * 8, 18, L.RSSF_F32) == 0


@pytest.mark.parametrize("B,H,W,C,dtype", [(2, 16, 16, 32, torch.bfloat16), (1, 24, 32, 32, torch.float32), (3, 8, 8, 16, torch.bfloat16), (16, 128, 128, 32, torch.bfloat16),
                                            (1, 8, 8, 64, torch.bfloat16)])
@pytest.mark.parametrize("add", [True, False])
def test_fused_gate_pool_and_layernorm_backward_equal_the_three_launches(B, H, W, C, dtype, add):
    """rssf_gate_pool_ln_bwd (the gate-path gradient merged into d(LN1 output) and norm1's backward on both token streams in one
    walk) against rssf_gate_pool_bwd + rssf_layernorm_bwd x 2: dx, dy to the output rounding, dgamma / dbeta to the summation order."""
    from representationlearning_amd import ops
    torch.manual_seed(9)
    N = H * W
    x = (torch.randn(B, N, C, device=DEV) * 1.3 + 0.2).to(dtype)
    y = (torch.randn(B, N, C, device=DEV) * 0.7 - 0.1).to(dtype)
    dxh, dyh = torch.randn(B, N, C, device=DEV).to(dtype), torch.randn(B, N, C, device=DEV).to(dtype)
    dout = torch.randn(B, N, C, device=DEV).to(dtype) if add else None
    g, b = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.2
    _, sx = ops.layernorm_fwd(x, g, b, want_y=False)
    _, sy = ops.layernorm_fwd(y, g, b, want_y=False)
    _, argmax = ops.gate_pool_fwd(x, y, sx, sy, g, b)
    dpooled = torch.randn(B, 4, N, device=DEV)
    dg1, db1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    fused = ops.gate_pool_ln_bwd(dpooled, argmax, dxh, dyh, x, y, sx, sy, g, dg1, db1, dx_add=dout)
    assert fused is not None
    a, c = dxh.clone(), dyh.clone()
    ops.gate_pool_bwd_(dpooled, argmax, a, c)
    dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.layernorm_bwd(a, x, sx, g, dg2, db2, dx_add=dout)
    dy = ops.layernorm_bwd(c, y, sy, g, dg2, db2)
    tol = 1e-5 if dtype == torch.float32 else 4e-3            # (bf16: one output rounding)
    assert rel_err(fused[0].float().cpu(), dx.float().cpu()) < tol and rel_err(fused[1].float().cpu(), dy.float().cpu()) < tol
    assert rel_err(dg1.cpu(), dg2.cpu()) < 2e-5 and rel_err(db1.cpu(), db2.cpu()) < 2e-5
