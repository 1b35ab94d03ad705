"""Full-size geometry pinned to the ORACLE (not to the HIP path itself): the whole RSSFormer step in fp32-I/O mode at the
benchmark's resolution against oracle.rssformer_cpu.model_forward run IN the test on the host cores (the oracle is pinned to the
reference by tests/test_oracle_golden.py), and GPU consumers of the reference goldens that so far only the CPU oracle read
(MlpDWBN: modules/ffn_block.py:237-270; neck + head: module/baseline/hrnet_aux.py:51-68, 78-81).
Reference of the step: module/baseline/hrnet_aux.py:89-110."""
import numpy as np
import pytest
import torch

from oracle import rssformer_cpu as O
from oracle.procedural import proc_input, proc_labels, procedural_state, seeded_input, seeded_state
from tests.helpers import golden, rel_err, seeded_params

pytestmark = pytest.mark.gpu
DEV = "cuda"

# parameters whose gradient passes through an argmax / max INSIDE their own module (the gate's channel max and alpha's max(M):
# multihead_isa_pool_attention.py:101-115, DAL.py:873-1020): a near-tie that resolves differently moves them discontinuously
ROUTED = (".attn.atrous_block", ".attn.weight_levels", ".attn.attn.q_proj", ".attn.attn.k_proj", ".transformer.norm1.")


def _build(variant):
    from tests.test_gpu_model import build
    return build(variant)


def _oracle_step(variant, x, y, dtype=torch.float32, backward=True):
    P = seeded_params(O.model_template(variant))
    if dtype != torch.float32:
        P = {k: (v.detach().to(dtype).requires_grad_(v.requires_grad) if v.is_floating_point() else v) for k, v in P.items()}
    taps = {}
    loss = O.model_forward(x.to(dtype), P, True, y, taps)
    if backward:
        loss.backward()
    return loss.detach(), taps, P


def _gpu_step(variant, x, y, backward=True, deterministic=False):
    """deterministic: fixed-order BatchNorm statistics and loss sums (nnf.Runtime.deterministic) - the comparison with the
    oracle then carries no run-to-run noise of the statistics' atomics."""
    from representationlearning_amd import nnf
    rt = nnf.current()
    was, rt.deterministic = rt.deterministic, bool(deterministic)
    try:
        return _gpu_step_inner(variant, x, y, backward)
    finally:
        rt.deterministic = was


def _gpu_step_inner(variant, x, y, backward):
    m = _build(variant).train()
    hr = m.backbone.hrnet
    taps, handles = {}, []
    handles.append(hr.layer1.register_forward_hook(lambda md, i, o: taps.__setitem__("layer1", o.detach())))
    for st in (2, 3, 4):
        handles.append(getattr(hr, "stage%d" % st).register_forward_hook(lambda md, i, o, st=st: taps.__setitem__("stage%d" % st, o[0].detach())))
    loss = m(x.to(DEV), dict(cls=y.to(DEV)))["fc_loss"]
    if backward:
        loss.backward()
    taps["logits"] = m._last_logits.detach()
    for h in handles:
        h.remove()
    return loss.detach(), taps, m


def test_base_full_step_2x512_fp32_vs_oracle():
    """Base, 2 x 3 x 512 x 512 (the bench's resolution: 128^2 / 64^2 / 32^2 / 16^2 maps, 5 776 attention windows per block, the
    256 x 128 gather tile of the MlpDWBN convolutions, the neck at 480 channels x 128^2), forward + loss + backward in fp32-I/O mode
    against the CPU oracle on the same seeded weights and inputs: loss and logits to north_star's 1e-3, every stage output, the
    parameter gradients in distribution (yardstick: the oracle's own fp64 run) and ELEMENT-WISE for every parameter whose own
    module does not route its gradient through a max."""
    B, S = 2, 512
    x, y = seeded_input((B, 3, S, S), 7), proc_labels(B, S, S, 6, 8)
    loss_g, taps_g, m = _gpu_step("base", x, y, deterministic=True)       # (no atomics noise in the oracle comparison)
    loss_o, taps_o, P = _oracle_step("base", x, y)
    assert abs(float(loss_g) - float(loss_o)) < 1e-3 * abs(float(loss_o)), (float(loss_g), float(loss_o))
    assert rel_err(taps_g["logits"][:, :, ::16, ::16].cpu(), taps_o["logits"][:, :, ::16, ::16].detach()) < 1e-3
    for k in ("layer1", "stage2", "stage3", "stage4"):
        assert rel_err(taps_g[k].float().cpu(), taps_o[k].detach()) < 1e-3, k
        a, b = float(taps_g[k].double().sum()), float(taps_o[k].detach().double().sum())
        assert abs(a - b) < 1e-3 * float(taps_o[k].detach().double().abs().sum()), k          # per-stage checksum
    # gradients: the oracle's own precision variants as the yardstick for the max-routed tail
    _, _, P64 = _oracle_step("base", x, y, torch.float64)
    got = {k: p.grad for k, p in m.named_parameters()}
    names = [k for k, v in P.items() if v.requires_grad and v.grad is not None]
    ref = {k: float(P[k].grad.double().norm()) for k in names}
    ref64 = {k: float(P64[k].grad.norm()) for k in names}
    gn = {k: (0.0 if got[k] is None else float(got[k].double().norm())) for k in names}
    floor = 2e-3 * float(np.median(list(ref.values())))
    live = [k for k in names if ref[k] > 10 * floor]
    dev = np.array([abs(gn[k] - ref[k]) / ref[k] for k in live])
    self_dev = np.array([abs(ref64[k] - ref[k]) / ref[k] for k in live])
    assert np.median(dev) < max(3e-3, 3 * np.median(self_dev)), (np.median(dev), np.median(self_dev))
    assert np.percentile(dev, 95) < max(3e-2, 3 * np.percentile(self_dev, 95)), (np.percentile(dev, 95), np.percentile(self_dev, 95))
    # element-wise (norm of the difference over the norm), parameter by parameter, outside the max-routed modules
    # (bar: 2e-2; a parameter whose gradient is a cancelling sum - a BatchNorm bias deep in the net - gets 4x the distance between
    # the oracle's OWN fp32 and fp64 gradients when that is larger: the reference cannot pin it tighter than it pins itself)
    plain = [k for k in live if not any(s in k for s in ROUTED)]
    assert len(plain) > 0.8 * len(live)
    errs = {k: rel_err(got[k].cpu(), P[k].grad) for k in plain}
    spread = {k: rel_err(P64[k].grad.float(), P[k].grad) for k in plain}
    # (round 5 had loosened this to 3e-2 / 5 x for the run-to-run noise of the statistics' atomics on a cancelling sum - one run in
    # three saw a BatchNorm weight of stage4's fuse layers at 2.6e-2; round 6 runs the step in deterministic-statistics mode instead
    # and holds the round-4 bar)
    bad = [(k, errs[k], spread[k]) for k in plain if errs[k] > max(2e-2, 4 * spread[k])]
    assert not bad, bad[:5]
    # in distribution the HIP path sits as close to the fp32 oracle as the oracle's fp64 run does (measured, MI355X: median element-wise
    # distance 3.0e-2 vs 2.1e-2, p95 3.8e-2 vs 2.7e-2 - a 300-layer BatchNorm network at random initialisation amplifies rounding differences of ANY two
    # implementations to a few per cent of the gradient, cf. tests/test_gpu_model.py::_full_size_modes)
    e, sp = np.array([errs[k] for k in plain]), np.array([spread[k] for k in plain])
    print("element-wise gradient distance: HIP vs fp32 oracle median %.2e p95 %.2e max %.2e | fp64 vs fp32 oracle median %.2e p95 %.2e max %.2e"
          % (np.median(e), np.percentile(e, 95), e.max(), np.median(sp), np.percentile(sp, 95), sp.max()))
    assert np.median(e) <= max(2e-2, 1.5 * np.median(sp)) and np.percentile(e, 95) <= max(2e-2, 1.5 * np.percentile(sp, 95))
    tail = [k for k in plain if k.startswith(("neck.", "head."))]              # downstream of every transformer block
    assert tail and all(errs[k] <= 2e-2 for k in tail), {k: errs[k] for k in tail}
    # ... and the routed ones against the oracle's own fp32 / fp64 spread
    # (a gate / q / k parameter takes its gradient through an argmax: one near-tie that resolves differently moves it by tens of per
    # cent - in the oracle's own fp32 / fp64 pair as well - so these are held in distribution, and no single one may be off by half)
    routed = [k for k in live if any(s in k for s in ROUTED)]
    er = np.array([rel_err(got[k].cpu(), P[k].grad) for k in routed])
    sr = np.array([rel_err(P64[k].grad.float(), P[k].grad) for k in routed])
    print("max-routed parameters: HIP vs fp32 oracle median %.2e max %.2e | fp64 vs fp32 oracle median %.2e max %.2e" % (np.median(er), er.max(), np.median(sr), sr.max()))
    assert np.median(er) <= max(5e-2, 2 * np.median(sr)) and er.max() < 0.5, (routed[int(er.argmax())], er.max())
    assert got["headaux.0.weight"] is None or float(got["headaux.0.weight"].abs().max()) == 0.0


def test_large_forward_1x1024_fp32_vs_oracle():
    """Large (w48), 1 x 3 x 1024 x 1024 - BASELINE config 4's tile: training-mode forward + loss (C = 48 window attention over
    37 x 37 windows of the 256^2 map, 192-wide MlpDWBN) against the CPU oracle."""
    B, S = 1, 1024
    x, y = seeded_input((B, 3, S, S), 7), proc_labels(B, S, S, 6, 8)
    with torch.no_grad():
        loss_g, taps_g, _ = _gpu_step("large", x, y, backward=False)
        loss_o, taps_o, _ = _oracle_step("large", x, y, backward=False)
    assert abs(float(loss_g) - float(loss_o)) < 1e-3 * abs(float(loss_o)), (float(loss_g), float(loss_o))
    assert rel_err(taps_g["logits"][:, :, ::32, ::32].cpu(), taps_o["logits"][:, :, ::32, ::32]) < 1e-3
    for k in ("layer1", "stage2", "stage3", "stage4"):
        assert rel_err(taps_g[k].float().cpu(), taps_o[k]) < 1e-3, k


@pytest.mark.parametrize("B,C,H,W", [(2, 32, 16, 13), (1, 18, 30, 30)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mlp_dwbn_vs_reference_golden(B, C, H, W, dtype):
    """MlpDWBN forward + backward against the golden of the reference's own module (ffn_block.py:237-270)."""
    import torch.nn as nn
    from representationlearning_amd.module.baseline.base_hrnet.modules.ffn_block import MlpDWBN
    g = golden(f"mlp_B{B}_C{C}_H{H}_W{W}")
    m = MlpDWBN(C, 4 * C, C, nn.GELU, nn.GELU, 0.0)
    m.load_state_dict(seeded_state(m.state_dict()))
    m = m.to(DEV).train()
    z = seeded_input((B, H * W, C), 21).to(DEV).to(dtype).requires_grad_()
    out = m(z, H, W)
    (out.float() * seeded_input(out.shape, 22).to(DEV)).sum().backward()
    tol_o, tol_g, tol_p = (2e-4, 5e-4, 1e-3) if dtype == torch.float32 else (3e-2, 8e-2, 8e-2)
    assert rel_err(out.detach().float().cpu(), g["out"]) < tol_o
    if B > 1:
        # (the B = 1 fixture's GRADIENTS are an artefact of the host library, not of the reference's mathematics: with one sample the
        # token-major output gradient reaches F.batch_norm's CPU backward as a [1, C, H, W] view whose size-1 batch stride makes it pass
        # for both memory formats, and that backward then reads it in the wrong one - `batch_norm(x).backward(g)` differs by 141 %
        # from `.backward(g.contiguous())` at B = 1 and by 2e-8 at B = 2 (tools/README.md, round 4).  Forward and running statistics
        # of that fixture are unaffected and are checked; gradients are checked on the B = 2 fixture.)
        assert rel_err(z.grad.float().cpu(), g["gz"]) < tol_g
        for k, p in m.named_parameters():
            ref = g["g_" + k.replace(".", "_")]
            if k.endswith(".bias") and not k.startswith("norm"):
                # a convolution bias in front of a BatchNorm receives NO gradient (the mean is subtracted): the fixture holds rounding noise
                wref = g["g_" + k[:-4].replace(".", "_") + "weight"]
                assert float(np.linalg.norm(ref)) < 1e-4 * float(np.linalg.norm(wref)) and float(p.grad.norm()) < 1e-2 * float(np.linalg.norm(wref)), k
                continue
            assert rel_err(p.grad.cpu(), ref) < tol_p, k
    if dtype == torch.float32:
        for k, v in m.named_buffers():
            if "running" in k:
                assert rel_err(v.cpu(), g["b_" + k.replace(".", "_")]) < 2e-4, k


def test_neck_head_vs_reference_golden():
    """SimpleFusion8 (bilinear resize of four maps written into one 480-channel buffer + 1x1 conv + BN + ReLU), the 1x1 head with
    x4 bilinear up-sampling and the auxiliary head against the golden of the reference's own modules (hrnet_aux.py:51-68, 78-87)."""
    from representationlearning_amd import nnf
    g = golden("neck_head")
    m = _build("base")
    m.load_state_dict(procedural_state(m.state_dict()))
    m = m.to(DEV).train()
    feats = [proc_input((2, c, 12 // s, 8 // s), 0.3 * i) for i, (c, s) in enumerate(((32, 1), (64, 2), (128, 4), (256, 4)))]
    feats[3] = proc_input((2, 256, 2, 1), 0.9)
    feats = [f.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_() for f in feats]
    fused, f0 = m.neck(feats)
    aux = nnf.aux_head(f0, m.headaux[0])
    lg = nnf.conv_bias(fused, m.head[0])
    lg = nnf.upsample_bilinear(lg, (lg.shape[2] * 4, lg.shape[3] * 4))
    (lg * proc_input(lg.shape, 0.7).to(DEV)).sum().backward()
    assert rel_err(lg.detach().cpu(), g["logits"]) < 2e-4
    assert rel_err(aux.cpu(), g["aux"]) < 2e-4
    for i, f in enumerate(feats):
        assert rel_err(f.grad.cpu(), g[f"gf{i}"]) < 5e-4, i
    assert rel_err(m.head[0].weight.grad.cpu(), g["g_head_w"]) < 5e-4
    assert rel_err(m.neck.fuse_conv[0].weight.grad.sum((2, 3)).cpu(), g["g_neck_w_sum"]) < 5e-4
