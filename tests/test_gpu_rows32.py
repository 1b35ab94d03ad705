"""csrc/conv_rows32.hip - the 32 -> 32 channel 3x3 convolution as a row stream (weights in registers, no LDS staging; the BasicBlock
convolutions of HRNet's full-resolution branch, reference _hrnet_rssformer.py:216-246) - against the halo kernel on the same operands
(rssf.h RSSF_CONV_GENERIC, an argument of the call) and against torch fp32 convolutions: forward with and without bias / fused
BatchNorm statistics, the pre-activation input (the producer's BatchNorm finalize + ReLU on load), the mirrored data gradient with
addend (also in place) and the producer's BatchNorm-backward statistics, ragged maps (W % 16 != 0, H % 8 != 0), one-row and
one-column maps, the benchmark geometry, and a grouped launch with a 32-channel member."""
import ctypes

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (B, H, W): ragged strips / bands, single strips, many blocks, the bench's branch-0 map
SHAPES = [(1, 7, 9), (2, 16, 16), (3, 5, 40), (1, 33, 130), (1, 1, 50), (2, 50, 1), (1, 8, 16), (2, 128, 128), (16, 128, 128)]


def _conv(seed, bias=False):
    torch.manual_seed(seed)
    return nn.Conv2d(32, 32, 3, 1, 1, bias=bias).to(DEV)


@pytest.mark.parametrize("B,H,W", SHAPES)
def test_rows32_forward_matches_halo_and_torch(B, H, W):
    from representationlearning_amd import nnf
    conv = _conv(3, bias=True)
    spec = nnf.spec_of([conv])
    w = [conv.weight.detach()]
    bias = conv.bias.detach().float().contiguous()
    x = torch.randn(B, H, W, 32, device=DEV).bfloat16()
    outs, stats = [], []
    for on in (False, True):
        st = torch.zeros(nnf.BN_SLOTS * 2 * 32, device=DEV)
        outs.append(nnf._conv_forward(spec, x, w, None, st, generic=not on))
        stats.append(st.view(nnf.BN_SLOTS, 2, 32).sum(0))
        outs.append(nnf._conv_forward(spec, x, w, bias, None, generic=not on))
    torch.cuda.synchronize()
    # one K-step per tap and the taps of an output row in the halo kernel's order: the same fp32 sums, the same bits
    assert torch.equal(outs[2], outs[0])
    assert rel_err(outs[3].float().cpu(), outs[1].float().cpu()) < 3e-3          # (bias: accumulator seed here, added last there)
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), conv.weight.detach().bfloat16().float(), None, 1, 1).permute(0, 2, 3, 1)
    assert rel_err(outs[2].float().cpu(), ref.cpu()) < 6e-3
    assert rel_err(outs[3].float().cpu(), (ref + bias.view(1, 1, 1, 32)).cpu()) < 6e-3
    assert rel_err(stats[1].cpu(), stats[0].cpu()) < 1e-5                        # the same values in another summation order
    n = B * H * W
    assert rel_err(stats[1][1].cpu() / n, (ref.reshape(-1, 32) ** 2).mean(0).cpu()) < 1e-3


@pytest.mark.parametrize("B,H,W", SHAPES)
@pytest.mark.parametrize("act", [0, 1])
def test_rows32_preact_forward_matches_halo(B, H, W, act):
    """rssf_conv_gather_preact: the producer's BatchNorm finalized in the launch (mean / invstd / scale / shift published, running
    statistics updated) and act(raw * scale + shift) formed on the operand registers - zero where the ACTIVATION is padding."""
    from representationlearning_amd import _lib as L, nnf
    lib = L.load()
    conv = _conv(5)
    spec = nnf.spec_of([conv])
    wpk = nnf._pack(spec, [conv.weight], False, torch.bfloat16, torch.device(DEV))
    raw_in = (torch.randn(B, H, W, 32, device=DEV) * 1.5 + 0.2).bfloat16()
    n = float(B * H * W)
    x32 = raw_in.float().reshape(-1, 32)
    stats_in = torch.zeros(nnf.BN_SLOTS, 2, 32, device=DEV)
    stats_in[1, 0], stats_in[1, 1] = x32.sum(0) * 0.25, (x32 * x32).sum(0) * 0.5
    stats_in[7, 0], stats_in[7, 1] = x32.sum(0) * 0.75, (x32 * x32).sum(0) * 0.5
    gamma, beta = torch.rand(32, device=DEV) + 0.5, torch.randn(32, device=DEV) * 0.5 + 0.3
    res = []
    for flag in (L.CONV_GENERIC, 0):
        rm, rv = torch.zeros(32, device=DEV), torch.ones(32, device=DEV)
        mi, ss = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
        o = torch.empty(B, H, W, 32, device=DEV, dtype=torch.bfloat16)
        st = torch.zeros(nnf.BN_SLOTS * 64, device=DEV)
        L.check(lib.rssf_conv_gather_preact(L.ptr(raw_in), L.ptr(stats_in), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(mi), L.ptr(ss), n, 0.1, 1e-5, 1,
                                            act, L.ptr(wpk), L.ptr(o), None, L.ptr(st), None, B, H, W, 32, H, W, 32, 1, 1, 9, spec.c_dy, spec.c_dx,
                                            L.RSSF_BF16 | flag, L.stream()), "preact conv")
        res.append((o, rm, rv, mi, ss, st.view(nnf.BN_SLOTS, 2, 32).sum(0)))
    torch.cuda.synchronize()
    for k in range(5):
        assert torch.equal(res[0][k], res[1][k]), k
    assert rel_err(res[1][5].cpu(), res[0][5].cpu()) < 1e-5
    # and against torch: act(bn(raw)) rounded to bf16, zero-padded, convolved
    mean, var = x32.mean(0), x32.var(0, unbiased=False)
    sc = gamma / torch.sqrt(var + 1e-5)
    z = raw_in.float() * sc + (beta - mean * sc)
    y = (z.clamp_min(0) if act == 1 else z).bfloat16().float()
    ref = F.conv2d(y.permute(0, 3, 1, 2), conv.weight.detach().bfloat16().float(), None, 1, 1).permute(0, 2, 3, 1)
    assert rel_err(res[1][0].float().cpu(), ref.cpu()) < 2e-2


@pytest.mark.parametrize("B,H,W", SHAPES)
@pytest.mark.parametrize("act,res", [(0, False), (1, False), (1, True)])
def test_rows32_dgrad_matches_halo_and_torch(B, H, W, act, res):
    """Data gradient (mirrored taps, transposed slabs) with an addend and the fused BatchNorm-backward statistics of the producer."""
    from representationlearning_amd import nnf
    conv = _conv(4)
    spec = nnf.spec_of([conv])
    w = [conv.weight.detach()]
    dout = torch.randn(B, H, W, 32, device=DEV).bfloat16()
    addend = torch.randn(B, H, W, 32, device=DEV).bfloat16()
    link = nnf.BnBwdLink()
    link.raw, link.act, link.C = torch.randn(B, H, W, 32, device=DEV).bfloat16(), act, 32
    link.rp = torch.randn(B, H, W, 32, device=DEV).bfloat16() if res else None
    link.ss = torch.stack([torch.rand(32, device=DEV) + 0.5, torch.randn(32, device=DEV) * 0.3]).contiguous()
    got, sums = [], []
    for on in (False, True):
        sm = torch.zeros(nnf.BN_BWD_SLOTS * 64, device=DEV)
        got.append(nnf._conv_dgrad(spec, dout, w, (B, H, W, 32), addend, bn=(link, sm), generic=not on).clone())
        sums.append(sm.view(nnf.BN_BWD_SLOTS, 2, 32).sum(0))
        got.append(nnf._conv_dgrad(spec, dout, w, (B, H, W, 32), None, generic=not on).clone())
        sm2 = torch.zeros(nnf.BN_BWD_SLOTS * 64, device=DEV)
        got.append(nnf._conv_dgrad(spec, dout, w, (B, H, W, 32), None, bn=(link, sm2), generic=not on).clone())
        sums.append(sm2.view(nnf.BN_BWD_SLOTS, 2, 32).sum(0))
    torch.cuda.synchronize()
    xr = torch.zeros(B, 32, H, W, device=DEV, requires_grad=True)
    F.conv2d(xr, conv.weight.detach().bfloat16().float(), None, 1, 1).backward(dout.permute(0, 3, 1, 2).float())
    ref = xr.grad.permute(0, 2, 3, 1)
    g_add, g_plain, g_bn = got[3], got[4], got[5]
    assert rel_err(g_plain.float().cpu(), ref.cpu()) < 6e-3
    assert rel_err(g_plain.float().cpu(), got[1].float().cpu()) < 3e-3         # the same sums in another order (rows ascend here)
    assert torch.equal(g_bn, g_plain)
    assert rel_err(g_add.float().cpu(), (ref + addend.float()).cpu()) < 6e-3
    assert rel_err(g_add.float().cpu(), got[0].float().cpu()) < 4e-3           # (added before rounding here, after it there)
    # the statistics are those of the bf16 values each kernel stored: hold them against a host evaluation on THIS kernel's output
    for g, sm in ((g_add, sums[2]), (g_bn, sums[3])):
        x = link.raw.float()
        z = x * link.ss[0] + link.ss[1] + (link.rp.float() if res else 0.0)
        dz = g.float() * ((z > 0).float() if act == 1 else 1.0)
        want = torch.stack([dz.reshape(-1, 32).sum(0), (dz * x).reshape(-1, 32).sum(0)])
        scale = float(dz.abs().reshape(-1, 32).sum(0).max()) + 1e-6
        assert float((sm - want).abs().max()) / scale < 2e-5
    assert rel_err(sums[3].cpu(), sums[1].cpu()) < 2e-2                           # (last-bit differences of the stored gradients)


def test_rows32_in_place_accumulation():
    """nnf.GradAccum's use: the data gradient accumulates into its own addend (out == addend)."""
    from representationlearning_amd import _lib as L, nnf
    lib = L.load()
    conv = _conv(8)
    spec = nnf.spec_of([conv])
    B, H, W = 2, 21, 37
    wpk = nnf._pack(spec, [conv.weight], True, torch.bfloat16, torch.device(DEV))
    dout = torch.randn(B, H, W, 32, device=DEV).bfloat16()
    acc0 = torch.randn(B, H, W, 32, device=DEV).bfloat16()
    outs = []
    for flag in (L.CONV_GENERIC, 0):
        buf = acc0.clone()
        L.check(lib.rssf_conv_gather_add(L.ptr(dout), L.ptr(wpk), L.ptr(buf), None, None, L.ptr(buf), None, B, H, W, 32, H, W, 32, 1, 1, 9, spec.c_ndy,
                                         spec.c_ndx, L.RSSF_BF16 | flag, L.stream()), "rssf_conv_gather_add")
        outs.append(buf)
    torch.cuda.synchronize()
    xr = torch.zeros(B, 32, H, W, device=DEV, requires_grad=True)
    F.conv2d(xr, conv.weight.detach().bfloat16().float(), None, 1, 1).backward(dout.permute(0, 3, 1, 2).float())
    ref = xr.grad.permute(0, 2, 3, 1) + acc0.float()
    assert rel_err(outs[1].float().cpu(), ref.cpu()) < 6e-3
    assert rel_err(outs[1].float().cpu(), outs[0].float().cpu()) < 4e-3


def test_rows32_member_of_a_grouped_launch():
    """rssf_conv3x3_group with a 32-channel member (the lock-step walk of a HighResolutionModule on one stream): that member runs on
    the row-stream kernel, the others as the group - every output equals the single-problem launches'."""
    from representationlearning_amd import nnf
    torch.manual_seed(2)
    chans, sizes = [32, 64, 128], [32, 16, 8]
    convs = [nn.Conv2d(c, c, 3, 1, 1, bias=False).to(DEV) for c in chans]
    xs = [torch.randn(2, s, s, c, device=DEV).bfloat16() for c, s in zip(chans, sizes)]
    singles, st1 = [], []
    for conv, x in zip(convs, xs):
        st = torch.zeros(nnf.BN_SLOTS * 2 * conv.out_channels, device=DEV)
        singles.append(nnf._conv_forward(nnf.spec_of([conv]), x, [conv.weight.detach()], None, st))
        st1.append(st)
    entries, outs, st2 = [], [], []
    for conv, x in zip(convs, xs):
        spec = nnf.spec_of([conv])
        wpk = nnf._pack(spec, [conv.weight.detach()], False, x.dtype, x.device)
        o = torch.empty_like(x)
        st = torch.zeros(nnf.BN_SLOTS * 2 * conv.out_channels, device=DEV)
        B, H, W, C = x.shape
        entries.append(dict(in_=x, wpk=wpk, out=o, stats=st, B=B, H=H, W=W, Cin=C, Cout=C))
        outs.append(o); st2.append(st)
    from representationlearning_amd import _lib as L
    nnf._group_conv3x3(entries, False, L.RSSF_BF16)
    torch.cuda.synchronize()
    for a, b, s, t in zip(singles, outs, st1, st2):
        assert torch.equal(a, b)
        C = a.shape[-1]
        assert rel_err(t.view(-1, 2, C).sum(0).cpu(), s.view(-1, 2, C).sum(0).cpu()) < 1e-5

