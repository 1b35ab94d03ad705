"""The lattice kernel of MlpDWBN's fused {1x1 + 3x3 dil 6 + 3x3 dil 12} convolution (csrc/conv_lattice.hip; reference
ffn_block.py:226-228, 250-257) against the generic gather kernel on the same operands and against torch fp32 convolutions:
forward with bias + fused BatchNorm statistics, data gradient with addend and with the producer's BatchNorm-backward statistics,
on ragged maps (lattice classes of different sizes, several tiles per class) and at the benchmark geometry."""
import os

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _convs(C, seed):
    torch.manual_seed(seed)
    return [nn.Conv2d(C, C, 1, 1).to(DEV), nn.Conv2d(C, C, 3, 1, padding=6, dilation=6).to(DEV),
            nn.Conv2d(C, C, 3, 1, padding=12, dilation=12).to(DEV)]


class _Switch:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("RSSF_LATTICE")
        os.environ["RSSF_LATTICE"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("RSSF_LATTICE", None)
        else:
            os.environ["RSSF_LATTICE"] = self.old


SHAPES = [(1, 30, 26), (2, 67, 70), (1, 133, 140), (3, 6, 5), (2, 128, 128)]


@pytest.fixture(params=["whole", "quarter-tail"], autouse=True)
def _tail_mode(request):
    """Every case twice: as whole workgroups, and with a pretended 16-CU chip so that the remainder past the last full round takes the
    quarter-workgroup path (32 output channels per workgroup) that the benchmark geometry uses for its last 64 of 576 tiles."""
    old = os.environ.get("RSSF_LATTICE_CUS")
    if request.param == "quarter-tail":
        os.environ["RSSF_LATTICE_CUS"] = "16"
    else:
        os.environ.pop("RSSF_LATTICE_CUS", None)
    yield
    if old is None:
        os.environ.pop("RSSF_LATTICE_CUS", None)
    else:
        os.environ["RSSF_LATTICE_CUS"] = old


@pytest.mark.parametrize("B,H,W", SHAPES)
def test_lattice_forward_matches_gather_and_torch(B, H, W):
    from representationlearning_amd import nnf
    from representationlearning_amd import _lib as L
    C = 128
    convs = _convs(C, 3)
    spec = nnf.spec_of(convs)
    weights = [c.weight.detach() for c in convs]
    bias = sum(c.bias.detach() for c in convs).float().contiguous()
    x = torch.randn(B, H, W, C, device=DEV).bfloat16()
    nslots = nnf.BN_SLOTS
    outs, stats = [], []
    for on in (False, True):
        with _Switch(on):
            st = torch.zeros(nslots * 2 * C, device=DEV)
            outs.append(nnf._conv_forward(spec, x, weights, bias, st))
            stats.append(st.view(nslots, 2, C).sum(0))
    torch.cuda.synchronize()
    ref = sum(F.conv2d(x.permute(0, 3, 1, 2).float(), c.weight.detach().bfloat16().float(), None, 1, c.padding, c.dilation) for c in convs)
    ref = (ref + bias.view(1, C, 1, 1)).permute(0, 2, 3, 1)
    assert rel_err(outs[1].float().cpu(), ref.cpu()) < 6e-3            # bf16 output rounding (2^-9 relative, rms ~ 2e-3)
    assert rel_err(outs[1].float().cpu(), outs[0].float().cpu()) < 3e-3   # same fp32 sums in another order, then the same rounding
    assert rel_err(stats[1].cpu(), stats[0].cpu()) < 1e-4
    n = B * H * W
    assert rel_err(stats[1][0].cpu() / n, ref.reshape(-1, C).mean(0).cpu()) < 2e-2 or float(ref.mean().abs()) < 1e-3
    assert rel_err(stats[1][1].cpu() / n, (ref.reshape(-1, C) ** 2).mean(0).cpu()) < 1e-3


@pytest.mark.parametrize("B,H,W", SHAPES)
@pytest.mark.parametrize("act", [0, 2])
def test_lattice_dgrad_matches_gather_and_torch(B, H, W, act):
    """Data gradient (mirrored taps, transposed slabs) with an addend and the fused BatchNorm-backward statistics of the producer."""
    from representationlearning_amd import nnf
    from representationlearning_amd import _lib as L
    C = 128
    convs = _convs(C, 4)
    spec = nnf.spec_of(convs)
    weights = [c.weight.detach() for c in convs]
    dout = torch.randn(B, H, W, C, device=DEV).bfloat16()
    addend = torch.randn(B, H, W, C, device=DEV).bfloat16()
    raw = torch.randn(B, H, W, C, device=DEV).bfloat16()
    link = nnf.BnBwdLink()
    link.raw, link.rp, link.act, link.C = raw, None, act, C
    link.ss = torch.stack([torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3]).contiguous()
    res, sums = [], []
    for on in (False, True):
        with _Switch(on):
            sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * C, device=DEV)
            res.append(nnf._conv_dgrad(spec, dout, weights, (B, H, W, C), addend, bn=(link, sm)).clone())
            sums.append(sm.view(nnf.BN_BWD_SLOTS, 2, C).sum(0))
            res.append(nnf._conv_dgrad(spec, dout, weights, (B, H, W, C), None).clone())
    torch.cuda.synchronize()
    xr = torch.zeros(B, C, H, W, device=DEV, requires_grad=True)
    y = sum(F.conv2d(xr, c.weight.detach().bfloat16().float(), None, 1, c.padding, c.dilation) for c in convs)
    y.backward(dout.permute(0, 3, 1, 2).float())
    ref = xr.grad.permute(0, 2, 3, 1)
    assert rel_err(res[3].float().cpu(), ref.cpu()) < 6e-3
    assert rel_err(res[3].float().cpu(), res[1].float().cpu()) < 3e-3
    assert rel_err(res[2].float().cpu(), (ref + addend.float()).cpu()) < 6e-3
    assert rel_err(res[2].float().cpu(), res[0].float().cpu()) < 3e-3
    assert rel_err(sums[1].cpu(), sums[0].cpu()) < 2e-3                 # statistics of bf16 values that differ in the last bit here and there


# ---- the point-wise 32 -> 128 stream kernel (csrc/conv_pw.hip) against the gather kernel --------------------------------------------
class _PwSwitch(_Switch):
    def __enter__(self):
        self.old = os.environ.get("RSSF_PW")
        os.environ["RSSF_PW"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("RSSF_PW", None)
        else:
            os.environ["RSSF_PW"] = self.old


@pytest.mark.parametrize("B,H,W", [(1, 7, 9), (2, 32, 32), (1, 128, 128), (3, 5, 5)])
def test_pointwise_forward_matches_gather_and_torch(B, H, W):
    """MlpDWBN's fc1 (ffn_block.py:219-221): Conv2d(32, 128, 1) with bias and the fused BatchNorm statistics; pixel counts that are
    and are not multiples of the 16-pixel tile."""
    from representationlearning_amd import nnf
    torch.manual_seed(7)
    conv = nn.Conv2d(32, 128, 1).to(DEV)
    spec = nnf.spec_of([conv])
    x = torch.randn(B, H, W, 32, device=DEV).bfloat16()
    outs, stats = [], []
    for on in (False, True):
        with _PwSwitch(on):
            st = torch.zeros(nnf.BN_SLOTS * 2 * 128, device=DEV)
            outs.append(nnf._conv_forward(spec, x, [conv.weight.detach()], conv.bias.detach().float().contiguous(), st))
            stats.append(st.view(nnf.BN_SLOTS, 2, 128).sum(0))
    torch.cuda.synchronize()
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), conv.weight.detach().bfloat16().float(), conv.bias.detach().float()).permute(0, 2, 3, 1)
    assert rel_err(outs[1].float().cpu(), ref.cpu()) < 4e-3
    # (the bias is the MFMA's accumulator input here and an fp32 add after it there: the last bf16 bit may differ)
    assert rel_err(outs[1].float().cpu(), outs[0].float().cpu()) < 2e-3
    assert rel_err(stats[1].cpu(), stats[0].cpu()) < 1e-5


@pytest.mark.parametrize("B,H,W", [(1, 7, 9), (2, 32, 32), (1, 128, 128)])
@pytest.mark.parametrize("act,res", [(0, False), (2, False), (1, True)])
def test_pointwise_dgrad_matches_gather(B, H, W, act, res):
    """fc2's data gradient (the transpose of Conv2d(128, 32, 1), ffn_block.py:246-249) with the fused BatchNorm-backward statistics
    of the layer before it."""
    from representationlearning_amd import nnf
    torch.manual_seed(8)
    conv = nn.Conv2d(128, 32, 1).to(DEV)
    spec = nnf.spec_of([conv])
    dout = torch.randn(B, H, W, 32, device=DEV).bfloat16()
    link = nnf.BnBwdLink()
    link.raw, link.act, link.C = torch.randn(B, H, W, 128, device=DEV).bfloat16(), act, 128
    link.rp = torch.randn(B, H, W, 128, device=DEV).bfloat16() if res else None
    link.ss = torch.stack([torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.3]).contiguous()
    outs, sums, plain = [], [], []
    for on in (False, True):
        with _PwSwitch(on):
            sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * 128, device=DEV)
            outs.append(nnf._conv_dgrad(spec, dout, [conv.weight.detach()], (B, H, W, 128), None, bn=(link, sm)).clone())
            sums.append(sm.view(nnf.BN_BWD_SLOTS, 2, 128).sum(0))
            plain.append(nnf._conv_dgrad(spec, dout, [conv.weight.detach()], (B, H, W, 128), None).clone())
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(dout.permute(0, 3, 1, 2).float(), conv.weight.detach().bfloat16().float()).permute(0, 2, 3, 1)
    assert rel_err(outs[1].float().cpu(), ref.cpu()) < 4e-3
    assert torch.equal(outs[1], outs[0]) and torch.equal(plain[1], plain[0])
    assert rel_err(sums[1].cpu(), sums[0].cpu()) < 1e-4


@pytest.mark.parametrize("B,H,W", [(1, 7, 9), (2, 32, 32), (1, 128, 128)])
def test_pointwise_128_to_32_forward_and_dgrad(B, H, W):
    """The other direction of the stream kernel: fc2 forward (Conv2d(128, 32, 1) with bias and fused statistics) and fc1's data
    gradient (the transpose of Conv2d(32, 128, 1), plain and with fused BatchNorm-backward statistics)."""
    from representationlearning_amd import nnf
    torch.manual_seed(9)
    fc2, fc1 = nn.Conv2d(128, 32, 1).to(DEV), nn.Conv2d(32, 128, 1).to(DEV)
    s2, s1 = nnf.spec_of([fc2]), nnf.spec_of([fc1])
    x = torch.randn(B, H, W, 128, device=DEV).bfloat16()
    link = nnf.BnBwdLink()
    link.raw, link.rp, link.act, link.C = torch.randn(B, H, W, 32, device=DEV).bfloat16(), None, 1, 32
    link.ss = torch.stack([torch.rand(32, device=DEV) + 0.5, torch.randn(32, device=DEV) * 0.3]).contiguous()
    res = []
    for on in (False, True):
        with _PwSwitch(on):
            st = torch.zeros(nnf.BN_SLOTS * 2 * 32, device=DEV)
            sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * 32, device=DEV)
            y = nnf._conv_forward(s2, x, [fc2.weight.detach()], fc2.bias.detach().float().contiguous(), st)
            d0 = nnf._conv_dgrad(s1, x, [fc1.weight.detach()], (B, H, W, 32), None).clone()
            d1 = nnf._conv_dgrad(s1, x, [fc1.weight.detach()], (B, H, W, 32), None, bn=(link, sm)).clone()
            res.append((y, st.view(nnf.BN_SLOTS, 2, 32).sum(0), d0, d1, sm.view(nnf.BN_BWD_SLOTS, 2, 32).sum(0)))
    torch.cuda.synchronize()
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), fc2.weight.detach().bfloat16().float(), fc2.bias.detach().float()).permute(0, 2, 3, 1)
    refd = F.conv_transpose2d(x.permute(0, 3, 1, 2).float(), fc1.weight.detach().bfloat16().float()).permute(0, 2, 3, 1)
    (y0, st0, a0, b0, sm0), (y1, st1, a1, b1, sm1) = res
    assert rel_err(y1.float().cpu(), ref.cpu()) < 4e-3 and rel_err(y1.float().cpu(), y0.float().cpu()) < 2e-3
    assert rel_err(st1.cpu(), st0.cpu()) < 1e-4
    assert rel_err(a1.float().cpu(), refd.cpu()) < 4e-3 and rel_err(a1.float().cpu(), a0.float().cpu()) < 2e-3
    assert torch.equal(a1, b1)
    assert rel_err(sm1.cpu(), sm0.cpu()) < 2e-3


@pytest.mark.parametrize("B,H,W", [(1, 7, 9), (2, 64, 64)])
def test_pointwise_64_to_256_forward_and_dgrad(B, H, W):
    """layer1's Bottleneck expansions (_hrnet_rssformer.py:249-287): Conv2d(64, 256, 1) forward with fused statistics and the data
    gradient of Conv2d(256, 64, 1) with the fused BatchNorm-backward statistics (ReLU, residual) - two 128-channel slices per pixel tile."""
    from representationlearning_amd import nnf
    torch.manual_seed(10)
    up, down = nn.Conv2d(64, 256, 1, bias=False).to(DEV), nn.Conv2d(256, 64, 1, bias=False).to(DEV)
    su, sd = nnf.spec_of([up]), nnf.spec_of([down])
    x = torch.randn(B, H, W, 64, device=DEV).bfloat16()
    link = nnf.BnBwdLink()
    link.raw, link.rp, link.act, link.C = torch.randn(B, H, W, 256, device=DEV).bfloat16(), torch.randn(B, H, W, 256, device=DEV).bfloat16(), 1, 256
    link.ss = torch.stack([torch.rand(256, device=DEV) + 0.5, torch.randn(256, device=DEV) * 0.3]).contiguous()
    res = []
    for on in (False, True):
        with _PwSwitch(on):
            st = torch.zeros(nnf.BN_SLOTS * 2 * 256, device=DEV)
            sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * 256, device=DEV)
            y = nnf._conv_forward(su, x, [up.weight.detach()], None, st)
            d = nnf._conv_dgrad(sd, x, [down.weight.detach()], (B, H, W, 256), None, bn=(link, sm)).clone()
            res.append((y, st.view(nnf.BN_SLOTS, 2, 256).sum(0), d, sm.view(nnf.BN_BWD_SLOTS, 2, 256).sum(0)))
    torch.cuda.synchronize()
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), up.weight.detach().bfloat16().float()).permute(0, 2, 3, 1)
    refd = F.conv_transpose2d(x.permute(0, 3, 1, 2).float(), down.weight.detach().bfloat16().float()).permute(0, 2, 3, 1)
    (y0, st0, d0, sm0), (y1, st1, d1, sm1) = res
    assert rel_err(y1.float().cpu(), ref.cpu()) < 4e-3 and rel_err(y1.float().cpu(), y0.float().cpu()) < 2e-3
    assert rel_err(st1.cpu(), st0.cpu()) < 1e-4
    assert rel_err(d1.float().cpu(), refd.cpu()) < 4e-3 and rel_err(d1.float().cpu(), d0.float().cpu()) < 2e-3
    assert rel_err(sm1.cpu(), sm0.cpu()) < 2e-3
