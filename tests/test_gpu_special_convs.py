"""The shape-specialised convolution kernels against the generic gather kernel on the same operands (rssf.h RSSF_CONV_GENERIC,
an argument of the call) and against torch fp32 convolutions:
  * csrc/conv_taps128.hip - MlpDWBN's fused {1x1 + 3x3 dil 6 + 3x3 dil 12} sum at 128 channels (reference ffn_block.py:226-228, 250-257):
    forward with bias + fused BatchNorm statistics, data gradient with addend and with the producer's BatchNorm-backward statistics;
    maps so low that most taps leave the image, several workgroups, the benchmark geometry;
  * csrc/conv_pw.hip - the point-wise stream kernel (fc1 / fc2 of MlpDWBN, layer1's Bottleneck expansions)."""
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


# (B, H, W): pixels a multiple of 512 and rows a multiple of 64 take the specialised kernel; (1, 30, 26) does not (both runs generic)
SHAPES = [(1, 8, 64), (1, 64, 64), (3, 16, 128), (2, 128, 128), (1, 4, 256), (1, 30, 26)]


@pytest.mark.parametrize("B,H,W", SHAPES)
def test_taps128_forward_matches_gather_and_torch(B, H, W):
    from representationlearning_amd import nnf
    C = 128
    convs = _convs(C, 3)
    spec = nnf.spec_of(convs)
    weights = [c.weight.detach() for c in convs]
    bias = sum(c.bias.detach() for c in convs).float().contiguous()
    x = torch.randn(B, H, W, C, device=DEV).bfloat16()
    nslots = nnf.BN_SLOTS
    outs, stats = [], []
    for on in (False, True):
        st = torch.zeros(nslots * 2 * C, device=DEV)
        outs.append(nnf._conv_forward(spec, x, weights, bias, st, generic=not on))
        stats.append(st.view(nslots, 2, C).sum(0))
        outs.append(nnf._conv_forward(spec, x, weights, None, None, generic=not on))          # no bias, no statistics
    torch.cuda.synchronize()
    ref = sum(F.conv2d(x.permute(0, 3, 1, 2).float(), c.weight.detach().bfloat16().float(), None, 1, c.padding, c.dilation) for c in convs)
    refb = (ref + bias.view(1, C, 1, 1)).permute(0, 2, 3, 1)
    assert rel_err(outs[2].float().cpu(), refb.cpu()) < 6e-3            # bf16 output rounding (2^-9 relative, rms ~ 2e-3)
    assert rel_err(outs[2].float().cpu(), outs[0].float().cpu()) < 3e-3   # the same fp32 sums in another order, then the same rounding
    assert rel_err(outs[3].float().cpu(), ref.permute(0, 2, 3, 1).cpu()) < 6e-3
    assert rel_err(outs[3].float().cpu(), outs[1].float().cpu()) < 3e-3
    # statistics of the bf16 values stored (specialised) / of the fp32 values before rounding (generic)
    assert rel_err(stats[1].cpu(), stats[0].cpu()) < 3e-4
    n = B * H * W
    assert rel_err(stats[1][0].cpu() / n, refb.reshape(-1, C).mean(0).cpu()) < 2e-2 or float(refb.mean().abs()) < 1e-3
    assert rel_err(stats[1][1].cpu() / n, (refb.reshape(-1, C) ** 2).mean(0).cpu()) < 1e-3


@pytest.mark.parametrize("B,H,W", SHAPES)
@pytest.mark.parametrize("act,res", [(0, False), (2, False), (1, True)])
def test_taps128_dgrad_matches_gather_and_torch(B, H, W, act, res):
    """Data gradient (mirrored taps, transposed slabs) with an addend and the fused BatchNorm-backward statistics of the producer."""
    from representationlearning_amd import nnf
    C = 128
    convs = _convs(C, 4)
    spec = nnf.spec_of(convs)
    weights = [c.weight.detach() for c in convs]
    dout = torch.randn(B, H, W, C, device=DEV).bfloat16()
    addend = torch.randn(B, H, W, C, device=DEV).bfloat16()
    link = nnf.BnBwdLink()
    link.raw, link.act, link.C = torch.randn(B, H, W, C, device=DEV).bfloat16(), act, C
    link.rp = torch.randn(B, H, W, C, device=DEV).bfloat16() if res else None
    link.ss = torch.stack([torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3]).contiguous()
    got, sums = [], []
    for on in (False, True):
        sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * C, device=DEV)
        got.append(nnf._conv_dgrad(spec, dout, weights, (B, H, W, C), addend, bn=(link, sm), generic=not on).clone())
        sums.append(sm.view(nnf.BN_BWD_SLOTS, 2, C).sum(0))
        got.append(nnf._conv_dgrad(spec, dout, weights, (B, H, W, C), None, generic=not on).clone())
    torch.cuda.synchronize()
    xr = torch.zeros(B, C, H, W, device=DEV, requires_grad=True)
    y = sum(F.conv2d(xr, c.weight.detach().bfloat16().float(), None, 1, c.padding, c.dilation) for c in convs)
    y.backward(dout.permute(0, 3, 1, 2).float())
    ref = xr.grad.permute(0, 2, 3, 1)
    assert rel_err(got[3].float().cpu(), ref.cpu()) < 6e-3
    assert rel_err(got[3].float().cpu(), got[1].float().cpu()) < 3e-3
    assert rel_err(got[2].float().cpu(), (ref + addend.float()).cpu()) < 6e-3
    assert rel_err(got[2].float().cpu(), got[0].float().cpu()) < 3e-3
    assert rel_err(sums[1].cpu(), sums[0].cpu()) < 2e-3                 # statistics of bf16 values that differ in the last bit here and there


def test_taps128_in_place_accumulation_and_other_tap_sets():
    """The data gradient accumulating into its own addend (nnf.GradAccum's use: out == addend) and a tap set that is not MlpDWBN's
    (a 3 x 5 window with gaps): any (dy, dx) list of 8..19 taps takes the kernel."""
    import ctypes
    from representationlearning_amd import _lib as L
    lib = L.load()
    C, B, H, W = 128, 1, 16, 64
    torch.manual_seed(11)
    taps = [(dy, dx) for dy in (-3, 0, 2) for dx in (-7, -1, 0, 1, 5)][:13]
    nt = len(taps)
    wpk = (torch.randn(nt, C, C, device=DEV) * 0.05).bfloat16()
    x = torch.randn(B, H, W, C, device=DEV).bfloat16()
    acc0 = torch.randn(B, H, W, C, device=DEV).bfloat16()
    dy = (ctypes.c_int * nt)(*[t[0] for t in taps]); dx = (ctypes.c_int * nt)(*[t[1] for t in taps])
    outs = []
    for flag in (L.CONV_GENERIC, 0):
        buf = acc0.clone()
        L.check(lib.rssf_conv_gather_add(L.ptr(x), L.ptr(wpk), L.ptr(buf), None, None, L.ptr(buf), None, B, H, W, C, H, W, C, 1, 1, nt, dy, dx,
                                         L.RSSF_BF16 | flag, L.stream()), "rssf_conv_gather_add")
        outs.append(buf)
    torch.cuda.synchronize()
    xf = torch.nn.functional.pad(x.float(), (0, 0, 7, 7, 3, 3))
    ref = acc0.float().clone()
    for t, (ddy, ddx) in enumerate(taps):
        ref += xf[:, 3 + ddy:3 + ddy + H, 7 + ddx:7 + ddx + W, :] @ wpk[t].float().t()
    assert rel_err(outs[1].float().cpu(), ref.cpu()) < 8e-3
    assert rel_err(outs[1].float().cpu(), outs[0].float().cpu()) < 4e-3


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
        st = torch.zeros(nnf.BN_SLOTS * 2 * 128, device=DEV)
        outs.append(nnf._conv_forward(spec, x, [conv.weight.detach()], conv.bias.detach().float().contiguous(), st, generic=not on))
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
        sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * 128, device=DEV)
        outs.append(nnf._conv_dgrad(spec, dout, [conv.weight.detach()], (B, H, W, 128), None, bn=(link, sm), generic=not on).clone())
        sums.append(sm.view(nnf.BN_BWD_SLOTS, 2, 128).sum(0))
        plain.append(nnf._conv_dgrad(spec, dout, [conv.weight.detach()], (B, H, W, 128), None, generic=not on).clone())
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
        st = torch.zeros(nnf.BN_SLOTS * 2 * 32, device=DEV)
        sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * 32, device=DEV)
        y = nnf._conv_forward(s2, x, [fc2.weight.detach()], fc2.bias.detach().float().contiguous(), st, generic=not on)
        d0 = nnf._conv_dgrad(s1, x, [fc1.weight.detach()], (B, H, W, 32), None, generic=not on).clone()
        d1 = nnf._conv_dgrad(s1, x, [fc1.weight.detach()], (B, H, W, 32), None, bn=(link, sm), generic=not on).clone()
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
        st = torch.zeros(nnf.BN_SLOTS * 2 * 256, device=DEV)
        sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * 256, device=DEV)
        y = nnf._conv_forward(su, x, [up.weight.detach()], None, st, generic=not on)
        d = nnf._conv_dgrad(sd, x, [down.weight.detach()], (B, H, W, 256), None, bn=(link, sm), generic=not on).clone()
        res.append((y, st.view(nnf.BN_SLOTS, 2, 256).sum(0), d, sm.view(nnf.BN_BWD_SLOTS, 2, 256).sum(0)))
    torch.cuda.synchronize()
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), up.weight.detach().bfloat16().float()).permute(0, 2, 3, 1)
    refd = F.conv_transpose2d(x.permute(0, 3, 1, 2).float(), down.weight.detach().bfloat16().float()).permute(0, 2, 3, 1)
    (y0, st0, d0, sm0), (y1, st1, d1, sm1) = res
    assert rel_err(y1.float().cpu(), ref.cpu()) < 4e-3 and rel_err(y1.float().cpu(), y0.float().cpu()) < 2e-3
    assert rel_err(st1.cpu(), st0.cpu()) < 1e-4
    assert rel_err(d1.float().cpu(), refd.cpu()) < 4e-3 and rel_err(d1.float().cpu(), d0.float().cpu()) < 2e-3
    assert rel_err(sm1.cpu(), sm0.cpu()) < 2e-3


@pytest.mark.parametrize("B,H,W", [(1, 7, 9), (2, 64, 64), (1, 128, 128)])
@pytest.mark.parametrize("mode", ["plain", "bn", "in_place"])
def test_pointwise_64_to_256_dgrad_with_skip_gradient(B, H, W, mode):
    """layer1's conv1 data gradients (the transpose of Conv2d(256, 64, 1), _hrnet_rssformer.py:249-287) ADD the residual path's
    gradient: the stream kernel carries the addend (16-byte pieces, one tile ahead, added before rounding) - against the generic
    kernel (which rounds the convolution first) and fp32; with the fused BatchNorm-backward statistics; accumulating in place."""
    from representationlearning_amd import nnf
    torch.manual_seed(11)
    down = nn.Conv2d(256, 64, 1, bias=False).to(DEV)
    sd = nnf.spec_of([down])
    dout = torch.randn(B, H, W, 64, device=DEV).bfloat16()
    skip = torch.randn(B, H, W, 256, device=DEV).bfloat16()
    link = nnf.BnBwdLink()
    link.raw, link.rp, link.act, link.C = torch.randn(B, H, W, 256, device=DEV).bfloat16(), torch.randn(B, H, W, 256, device=DEV).bfloat16(), 1, 256
    link.ss = torch.stack([torch.rand(256, device=DEV) + 0.5, torch.randn(256, device=DEV) * 0.3]).contiguous()
    outs, sums = [], []
    for on in (False, True):
        sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * 256, device=DEV)
        if mode == "in_place":
            buf = skip.clone()
            d = nnf._conv_dgrad(sd, dout, [down.weight.detach()], (B, H, W, 256), buf, out=buf, generic=not on)
            assert d.data_ptr() == buf.data_ptr()
        else:
            d = nnf._conv_dgrad(sd, dout, [down.weight.detach()], (B, H, W, 256), skip, bn=(link, sm) if mode == "bn" else None, generic=not on)
        outs.append(d.clone())
        sums.append(sm.view(nnf.BN_BWD_SLOTS, 2, 256).sum(0))
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(dout.permute(0, 3, 1, 2).float(), down.weight.detach().bfloat16().float()).permute(0, 2, 3, 1) + skip.float()
    assert rel_err(outs[1].float().cpu(), ref.cpu()) < 4e-3
    assert rel_err(outs[1].float().cpu(), outs[0].float().cpu()) < 4e-3
    # (one rounding instead of two: not further from fp32 than the generic kernel)
    assert rel_err(outs[1].float().cpu(), ref.cpu()) <= rel_err(outs[0].float().cpu(), ref.cpu()) * 1.05
    if mode == "bn":
        assert rel_err(sums[1].cpu(), sums[0].cpu()) < 3e-3


@pytest.mark.parametrize("B,H,W,Cin,Cout,taps", [
    (2, 16, 64, 480, 480, [(0, 0)]),                                   # the neck's point-wise convolution (hrnet_aux.py:45-49), small map
    (1, 8, 64, 160, 224, [(0, 0), (-1, 2), (3, -5)]),                 # a ragged last input chunk (32 of 128) and output tile (96 of 128)
    (1, 8, 128, 128, 256, [(dy, dx) for dy in (-1, 0, 1) for dx in (-2, 0, 2)][:5]),
])
@pytest.mark.parametrize("mode", ["forward", "dgrad_bn"])
def test_taps128_general_channels_match_gather_and_torch(B, H, W, Cin, Cout, taps, mode):
    """The general form of csrc/conv_taps128.hip (K-steps = (tap, 128-channel chunk) pairs, one 128-channel output tile per workgroup):
    bias + statistics epilogue / addend + BatchNorm-backward statistics epilogue, against the generic kernel (RSSF_CONV_GENERIC) and
    an fp32 torch contraction of the same packed slabs."""
    import ctypes
    from representationlearning_amd import _lib as L, nnf
    lib = L.load()
    torch.manual_seed(21)
    nt = len(taps)
    rows_p, cols_p = lib.rssf_conv_packed_rows(Cout), lib.rssf_conv_packed_cols(Cin, L.RSSF_BF16)
    wpk = torch.zeros(nt, rows_p, cols_p, device=DEV)
    wpk[:, :Cout, :Cin] = torch.randn(nt, Cout, Cin, device=DEV) * (0.5 / (Cin * nt) ** 0.5)
    wpk = wpk.bfloat16()
    x = torch.randn(B, H, W, Cin, device=DEV).bfloat16()
    dy = (ctypes.c_int * nt)(*[t[0] for t in taps]); dx = (ctypes.c_int * nt)(*[t[1] for t in taps])
    bias = torch.randn(Cout, device=DEV)
    addend = torch.randn(B, H, W, Cout, device=DEV).bfloat16()
    raw = torch.randn(B, H, W, Cout, device=DEV).bfloat16()
    ss = torch.stack([torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV) * 0.3]).contiguous()
    outs, sts = [], []
    for flag in (L.CONV_GENERIC, 0):
        out = torch.empty(B, H, W, Cout, device=DEV, dtype=torch.bfloat16)
        if mode == "forward":
            st = torch.zeros(nnf.BN_SLOTS * 2 * Cout, device=DEV)
            L.check(lib.rssf_conv_gather_add(L.ptr(x), L.ptr(wpk), L.ptr(out), L.ptr(bias), L.ptr(st), None, None, B, H, W, Cin, H, W, Cout, 1, 1,
                                             nt, dy, dx, L.RSSF_BF16 | flag, L.stream()), "rssf_conv_gather_add")
            sts.append(st.view(nnf.BN_SLOTS, 2, Cout).sum(0))
        else:
            st = torch.zeros(nnf.BN_BWD_SLOTS * 2 * Cout, device=DEV)
            L.check(lib.rssf_conv_gather_bnbwd(L.ptr(x), L.ptr(wpk), L.ptr(out), L.ptr(addend), L.ptr(raw), None, L.ptr(ss), 2, L.ptr(st), B, H, W,
                                               Cin, H, W, Cout, 1, 1, nt, dy, dx, L.RSSF_BF16 | flag, L.stream()), "rssf_conv_gather_bnbwd")
            sts.append(st.view(nnf.BN_BWD_SLOTS, 2, Cout).sum(0))
        outs.append(out)
    torch.cuda.synchronize()
    py, px = max(abs(t[0]) for t in taps), max(abs(t[1]) for t in taps)
    xf = F.pad(x.float(), (0, 0, px, px, py, py))
    ref = torch.zeros(B, H, W, Cout, device=DEV)
    for t, (ddy, ddx) in enumerate(taps):
        ref += xf[:, py + ddy:py + ddy + H, px + ddx:px + ddx + W, :] @ wpk[t, :Cout, :Cin].float().t()
    ref = ref + bias if mode == "forward" else ref + addend.float()
    assert rel_err(outs[1].float().cpu(), ref.cpu()) < 8e-3
    assert rel_err(outs[1].float().cpu(), outs[0].float().cpu()) < 4e-3
    assert rel_err(sts[1].cpu(), sts[0].cpu()) < (3e-4 if mode == "forward" else 3e-3)


# ---- csrc/conv_wgrad_planes.hip: the tap sum's weight gradient from a transposed, zero-bordered copy of its input ----
def _planes_of(x, rt, key):
    from representationlearning_amd import nnf
    B, H, W, C = x.shape
    P = nnf.PLANES_PAD
    buf, k, gen = rt.planes_buffer(key, (C, B, H + 2 * P, W + 2 * P), x.dtype, x.device)
    buf.zero_()
    buf[:, :, P:P + H, P:P + W] = x.permute(3, 0, 1, 2)
    return (buf, P, k, gen)


@pytest.mark.parametrize("B,H,W", [(1, 8, 128), (2, 128, 128), (1, 16, 256), (3, 4, 128), (1, 30, 128)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_wgrad_planes_matches_generic_and_torch(B, H, W, with_bias):
    from representationlearning_amd import nnf, _lib as L
    C = 128
    convs = _convs(C, 11)
    spec = nnf.spec_of(convs)
    x = torch.randn(B, H, W, C, device=DEV).bfloat16()
    dout = (torch.randn(B, H, W, C, device=DEV) * 0.5).bfloat16()
    rt = nnf.current()
    planes = _planes_of(x, rt, ("test", B, H, W))
    assert L.load().rssf_conv_wgrad_planes_supported(B, H, W, C, C, 1, spec.ntaps, spec.c_dy, spec.c_dx, planes[1], L.dtype_code(x)) == 1
    got = []
    for pl in (None, planes):
        dws = [torch.zeros_like(c.weight, dtype=torch.float32) + 0.25 for c in convs]        # the kernels ACCUMULATE
        db = torch.full((C,), 0.5, device=DEV) if with_bias else None
        nnf._conv_wgrad(spec, dout, x, dws, db, rt, planes=pl)
        got.append((dws, db))
    torch.cuda.synchronize()
    xr = x.permute(0, 3, 1, 2).float()
    ws = [c.weight.detach().clone().float().requires_grad_(True) for c in convs]
    y = sum(F.conv2d(xr, w, None, 1, c.padding, c.dilation) for w, c in zip(ws, convs))
    y.backward(dout.permute(0, 3, 1, 2).float())
    for i in range(3):
        ref = ws[i].grad + 0.25
        assert rel_err(got[1][0][i].cpu(), ref.cpu()) < 2e-5, i           # bf16 operands are exact in fp32: only the summation order differs
        assert rel_err(got[1][0][i].cpu(), got[0][0][i].cpu()) < 2e-5, i
    if with_bias:
        ref = dout.float().sum((0, 1, 2)) + 0.5
        assert rel_err(got[1][1].cpu(), ref.cpu()) < 2e-5
        assert rel_err(got[1][1].cpu(), got[0][1].cpu()) < 2e-5


def test_wgrad_planes_stale_copy_takes_the_ordinary_path():
    """A transposed copy that was overwritten since the forward pass that made it (the layer ran twice) must not be used."""
    from representationlearning_amd import nnf
    C, B, H, W = 128, 1, 8, 128
    convs = _convs(C, 12)
    spec = nnf.spec_of(convs)
    x = torch.randn(B, H, W, C, device=DEV).bfloat16()
    dout = torch.randn(B, H, W, C, device=DEV).bfloat16()
    rt = nnf.current()
    old = _planes_of(x, rt, ("stale",))
    new = _planes_of(torch.zeros_like(x), rt, ("stale",))          # the same buffer, rewritten (zeros): `old` is stale now
    assert old[0].data_ptr() == new[0].data_ptr() and not rt.planes_current(old[2], old[3]) and rt.planes_current(new[2], new[3])
    a = [torch.zeros_like(c.weight, dtype=torch.float32) for c in convs]
    b = [torch.zeros_like(c.weight, dtype=torch.float32) for c in convs]
    nnf._conv_wgrad(spec, dout, x, a, None, rt, planes=old)
    nnf._conv_wgrad(spec, dout, x, b, None, rt, planes=None)
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert float(v.abs().max()) > 0 and torch.equal(u, v)


@pytest.mark.parametrize("B,H,W", [(1, 8, 128), (2, 128, 128), (1, 2, 256)])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("training", [True, False])
def test_bn_finalize_apply_planes_matches_apply(B, H, W, act, training):
    """rssf_bn_finalize_apply_planes: y, the BatchNorm results and the running statistics of rssf_bn_finalize_apply, plus the transposed
    copy (interior = y, border untouched)."""
    from representationlearning_amd import nnf, _lib as L
    C, P = 128, nnf.PLANES_PAD
    lib = L.load()
    raw = (torch.randn(B, H, W, C, device=DEV) * 1.5 + 0.3).bfloat16()
    r = raw.float().reshape(-1, C)
    stats = torch.zeros(nnf.BN_SLOTS, 2, C, device=DEV)
    stats[0, 0], stats[0, 1] = r.sum(0), (r * r).sum(0)
    stats[3] = 0.0
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.2
    n = float(B * H * W)
    res = []
    for planes_on in (False, True):
        rm, rv = torch.full((C,), 0.1, device=DEV), torch.full((C,), 0.9, device=DEV)
        mi, ss = torch.empty(2, C, device=DEV), torch.empty(2, C, device=DEV)
        y = torch.empty_like(raw)
        st = stats.clone().view(-1)
        if planes_on:
            assert lib.rssf_bn_finalize_apply_planes_supported(B, H, W, C, P, L.dtype_code(raw)) == 1
            pl = torch.full((C, B, H + 2 * P, W + 2 * P), 7.0, device=DEV, dtype=raw.dtype)
            L.check(lib.rssf_bn_finalize_apply_planes(L.ptr(raw), L.ptr(st), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(mi), L.ptr(ss),
                                                      L.ptr(y), L.ptr(pl), B, H, W, C, P, act, n, 0.1, 1e-5, int(training), L.dtype_code(raw),
                                                      L.stream()), "planes")
        else:
            pl = None
            L.check(lib.rssf_bn_finalize_apply(L.ptr(raw), L.ptr(st), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(mi), L.ptr(ss), None, None,
                                               L.ptr(y), B * H * W, C, act, n, 0.1, 1e-5, int(training), L.dtype_code(raw), L.stream()), "apply")
        res.append((y, mi, ss, rm, rv, pl))
    torch.cuda.synchronize()
    a, b = res
    for u, v in zip(a[1:5], b[1:5]):                    # the same fp32 expressions; the compiler contracts them into FMAs kernel by kernel
        assert rel_err(v.cpu(), u.cpu()) < 1e-6
    assert rel_err(b[0].float().cpu(), a[0].float().cpu()) < 1e-4           # (a last-bit scale difference moves a few bf16 roundings)
    pl = b[5]
    assert torch.equal(pl[:, :, P:P + H, P:P + W], b[0].permute(3, 0, 1, 2))
    border = pl.clone()
    border[:, :, P:P + H, P:P + W] = 7.0
    assert bool((border == 7.0).all())                 # only the interior is written


# a convolution bias in front of a training-mode BatchNorm has a gradient of exactly zero (the mean is subtracted): what the kernels
# return is the rounding residue of sum(draw), different in every run
ZERO_GRAD_BIASES = ("fc1.bias", "dw.bias", "dw6.bias", "dw12.bias", "fc2.bias")


def test_mlp_block_gradients_with_and_without_planes():
    """MlpDWBN forward + backward with the transposed-copy weight gradient on and off: the same outputs, the same gradients up to the
    summation order of the tap sum's weight gradient (and the bf16 rounding noise named below)."""
    from representationlearning_amd import nnf
    from representationlearning_amd.module.baseline.base_hrnet.modules.ffn_block import MlpDWBN
    torch.manual_seed(5)
    m = MlpDWBN(128, 128, 128).to(DEV).train()
    x = torch.randn(2, 128, 128, 128, device=DEV).bfloat16()            # [B, H, W, C]
    dy = torch.randn(2, 128, 128, 128, device=DEV).bfloat16()
    runs = []
    keep = nnf._WGRAD_PLANES
    try:
        for on in (False, False, True):
            nnf._WGRAD_PLANES = on
            for p in m.parameters():
                p.grad = None
            xi = x.clone().permute(0, 3, 1, 2).requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m.forward_nhwc(xi)
            y.backward(dy.permute(0, 3, 1, 2))
            torch.cuda.synchronize()
            runs.append((y.detach().float(), xi.grad.float(), {k: p.grad.float().clone() for k, p in m.named_parameters()}))
    finally:
        nnf._WGRAD_PLANES = keep
    assert any(k[0] == m.norm1.weight.data_ptr() for k in nnf.current().planes)       # the path under test ran
    # two runs of ONE configuration already differ (the BatchNorm statistics are sums of atomics; bf16 roundings move with their last
    # bit): the run with the transposed copy must sit within a small multiple of that floor
    def floor(a, b):
        return max(3.0 * rel_err(a.cpu(), b.cpu()), 5e-3)
    assert rel_err(runs[2][0].cpu(), runs[0][0].cpu()) < floor(runs[1][0], runs[0][0])
    assert rel_err(runs[2][1].cpu(), runs[0][1].cpu()) < floor(runs[1][1], runs[0][1])
    for k in runs[0][2]:
        if k in ZERO_GRAD_BIASES:
            continue
        assert rel_err(runs[2][2][k].cpu(), runs[0][2][k].cpu()) < floor(runs[1][2][k], runs[0][2][k]), k


# ---- csrc/conv_wgrad_pw.hip: the narrow point-wise weight gradients, one block per pixel range ----
PW_WGRAD_SHAPES = [(128, 32), (32, 128), (256, 64), (64, 256), (32, 64), (64, 32), (64, 128), (128, 64), (64, 64)]       # (Cout, Cin)


@pytest.mark.parametrize("cout,cin", PW_WGRAD_SHAPES)
@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 128, 128), (3, 24, 32)])
def test_pointwise_wgrad_matches_generic_and_torch(cout, cin, B, H, W):
    from representationlearning_amd import nnf
    torch.manual_seed(21)
    conv = nn.Conv2d(cin, cout, 1).to(DEV)
    spec = nnf.spec_of([conv])
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    dout = (torch.randn(B, H, W, cout, device=DEV) * 0.5).bfloat16()
    got = []
    for generic in (True, False):
        dw = torch.full_like(conv.weight, 0.25, dtype=torch.float32)           # the kernels ACCUMULATE
        db = torch.full((cout,), 0.5, device=DEV)
        nnf._conv_wgrad(spec, dout, x, [dw], db, generic=generic)
        got.append((dw, db))
    torch.cuda.synchronize()
    ref = torch.einsum("bhwo,bhwi->oi", dout.float(), x.float()).view(cout, cin, 1, 1) + 0.25
    assert rel_err(got[1][0].cpu(), ref.cpu()) < 2e-5               # exact products, fp32 sums in another order
    assert rel_err(got[1][0].cpu(), got[0][0].cpu()) < 2e-5
    refb = dout.float().sum((0, 1, 2)) + 0.5
    assert rel_err(got[1][1].cpu(), refb.cpu()) < 2e-5 and rel_err(got[1][1].cpu(), got[0][1].cpu()) < 2e-5


@pytest.mark.parametrize("cout,cin", [(128, 32), (256, 64), (32, 128), (64, 64)])
@pytest.mark.parametrize("act,res,training", [(2, False, True), (1, True, True), (0, False, True), (1, False, False)])
def test_pointwise_wgrad_with_fused_bn_apply_matches_two_launches(cout, cin, act, res, training):
    """rssf_conv_wgrad_bnapply on the narrow point-wise kernel: draw / dz / dgamma / dbeta of rssf_bn_bwd_apply (bit-identical: one shared
    definition of the arithmetic) and the weight gradient of the separate launches."""
    from representationlearning_amd import nnf, _lib as L
    lib = L.load()
    torch.manual_seed(22)
    B, H, W = 2, 64, 64
    conv = nn.Conv2d(cin, cout, 1, bias=False).to(DEV)
    spec = nnf.spec_of([conv])
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    dy = torch.randn(B, H, W, cout, device=DEV).bfloat16()
    raw = (torch.randn(B, H, W, cout, device=DEV) * 1.3 + 0.2).bfloat16()
    rp = torch.randn(B, H, W, cout, device=DEV).bfloat16() if res else None
    mean, var = raw.float().mean((0, 1, 2)), raw.float().var((0, 1, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    gamma, beta = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.2
    ss = torch.stack([gamma * istd, beta - mean * gamma * istd]).contiguous()
    mi = torch.stack([mean, istd]).contiguous()
    rows, n = B * H * W, float(B * H * W)
    sums = torch.zeros(nnf.BN_BWD_SLOTS * 2 * cout, device=DEV)
    L.check(lib.rssf_bn_bwd_reduce(L.ptr(dy), L.ptr(raw), L.ptr(ss), L.ptr(rp), L.ptr(sums), rows, cout, act, None, L.dtype_code(raw), L.stream()), "reduce")
    outs = []
    for generic in (True, False):
        draw, dres = torch.empty_like(raw), (torch.empty_like(raw) if res else None)
        dg, dbt = torch.full((cout,), 0.125, device=DEV), torch.full((cout,), 0.25, device=DEV)
        dw = torch.zeros_like(conv.weight, dtype=torch.float32)
        bn = (dy, raw, ss, mi, sums, rp, dres, dg, dbt, act, n, training, 0.5)
        nnf._conv_wgrad(spec, draw, x, [dw], None, bn=bn, generic=generic)
        outs.append((draw, dres, dg, dbt, dw))
    torch.cuda.synchronize()
    a, b = outs
    assert torch.equal(a[0], b[0]) and (not res or torch.equal(a[1], b[1]))
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert float(b[4].abs().max()) > 0 and rel_err(b[4].cpu(), a[4].cpu()) < 2e-5


def test_mlp_block_with_norm2_applied_on_load_matches_materialised():
    """MlpDWBN with GELU(norm2(sum)) applied by fc2's kernels on load (rssf_conv_gather_preact / rssf_conv_wgrad_preact on the point-wise
    stream kernels) against the form that writes the activation: outputs, input gradient, every parameter gradient and the running
    statistics, within a small multiple of the run-to-run floor of one configuration."""
    from representationlearning_amd import nnf
    from representationlearning_amd.module.baseline.base_hrnet.modules.ffn_block import MlpDWBN
    torch.manual_seed(7)
    x = torch.randn(2, 64, 64, 32, device=DEV).bfloat16()               # [B, H, W, C]: fc1 32 -> 128, fc2 128 -> 32
    dy = torch.randn(2, 64, 64, 32, device=DEV).bfloat16()
    res = torch.randn(2, 64, 64, 32, device=DEV).bfloat16()
    keep = nnf._DEFER_BN_APPLY
    runs = []
    try:
        for on in (False, False, True):
            nnf._DEFER_BN_APPLY = on
            nnf._DEFER_CACHE.clear()
            torch.manual_seed(8)
            m = MlpDWBN(32, 128, 32).to(DEV).train()
            xi = x.clone().permute(0, 3, 1, 2).requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m.forward_nhwc(xi, res.permute(0, 3, 1, 2), post_relu=True)
            y.backward(dy.permute(0, 3, 1, 2))
            torch.cuda.synchronize()
            deferred = nnf.can_defer_apply(xi, [m.dw, m.dw6, m.dw12], m.fc2)
            assert deferred == on, "the path under test must be the one that ran"
            runs.append((y.detach().float(), xi.grad.float(), {k: p.grad.float().clone() for k, p in m.named_parameters()},
                         {k: b.float().clone() for k, b in m.named_buffers() if "running" in k}))
    finally:
        nnf._DEFER_BN_APPLY = keep
        nnf._DEFER_CACHE.clear()

    def floor(a, b):            # (a wrong channel map or a missing activation is an O(1) error: the bar only has to clear the bf16 noise)
        return max(3.0 * rel_err(a.cpu(), b.cpu()), 2e-2)
    assert rel_err(runs[2][0].cpu(), runs[0][0].cpu()) < floor(runs[1][0], runs[0][0])
    assert rel_err(runs[2][1].cpu(), runs[0][1].cpu()) < floor(runs[1][1], runs[0][1])
    for k in runs[0][2]:
        if k in ZERO_GRAD_BIASES:
            continue
        assert rel_err(runs[2][2][k].cpu(), runs[0][2][k].cpu()) < floor(runs[1][2][k], runs[0][2][k]), k
    for k in runs[0][3]:
        assert rel_err(runs[2][3][k].cpu(), runs[0][3][k].cpu()) < 1e-5, k


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 128, 128), (3, 16, 64)])
@pytest.mark.parametrize("act,training", [(2, True), (1, True), (0, False)])
def test_pointwise_backward_in_one_launch_matches_three(B, H, W, act, training):
    """rssf_conv_wgrad_bnapply_dgrad (MlpDWBN's fc1, 128 <- 32): draw / dgamma / dbeta / dW / dbias of rssf_conv_wgrad_bnapply and the data
    gradient of rssf_conv_gather on the transposed pack, from one launch."""
    from representationlearning_amd import nnf, _lib as L
    lib = L.load()
    torch.manual_seed(41)
    cin, cout = 32, 128
    conv = nn.Conv2d(cin, cout, 1).to(DEV)
    spec = nnf.spec_of([conv])
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    dy = torch.randn(B, H, W, cout, device=DEV).bfloat16()
    raw = (torch.randn(B, H, W, cout, device=DEV) * 1.3 + 0.2).bfloat16()
    mean, var = raw.float().mean((0, 1, 2)), raw.float().var((0, 1, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    gamma, beta = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.2
    ss = torch.stack([gamma * istd, beta - mean * gamma * istd]).contiguous()
    mi = torch.stack([mean, istd]).contiguous()
    rows, n = B * H * W, float(B * H * W)
    assert lib.rssf_conv_wgrad_bnapply_dgrad_supported(B, H, W, cin, H, W, cout, 1, 1, spec.c_dy, spec.c_dx, 0, L.dtype_code(raw)) == 1
    sums = torch.zeros(nnf.BN_BWD_SLOTS * 2 * cout, device=DEV)
    L.check(lib.rssf_bn_bwd_reduce(L.ptr(dy), L.ptr(raw), L.ptr(ss), None, L.ptr(sums), rows, cout, act, None, L.dtype_code(raw), L.stream()), "reduce")
    w = conv.weight.detach().contiguous()
    outs = []
    for fused in (False, True):
        draw = torch.empty_like(raw)
        dg, dbt = torch.full((cout,), 0.125, device=DEV), torch.full((cout,), 0.25, device=DEV)
        dw, db = torch.zeros_like(w), torch.zeros(cout, device=DEV)
        bn = (dy, raw, ss, mi, sums, None, None, dg, dbt, act, n, training, 0.5)
        if fused:
            dx = torch.full((B, H, W, cin), 7.0, device=DEV).bfloat16()
            nnf._conv_wgrad(spec, draw, x, [dw], db, bn=bn, dgrad=(w, dx))
        else:
            nnf._conv_wgrad(spec, draw, x, [dw], db, bn=bn)
            dx = nnf._conv_dgrad(spec, draw, [w], (B, H, W, cin), None).clone()
        outs.append((draw, dg, dbt, dw, db, dx))
    torch.cuda.synchronize()
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert float(b[3].abs().max()) > 0 and rel_err(b[3].cpu(), a[3].cpu()) < 2e-5 and rel_err(b[4].cpu(), a[4].cpu()) < 2e-4
    ref = torch.einsum("bhwo,oi->bhwi", a[0].float(), w.bfloat16().float().view(cout, cin))
    assert rel_err(b[5].float().cpu(), ref.cpu()) < 4e-3 and rel_err(b[5].float().cpu(), a[5].float().cpu()) < 2e-3


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 128, 128), (3, 16, 64)])
@pytest.mark.parametrize("act", [2, 1, 0])
def test_pointwise_backward_behind_a_preactivation_input_in_one_launch(B, H, W, act):
    """rssf_conv_wgrad_preact_dgrad (MlpDWBN's fc2, 32 <- 128, input = the raw tap sum): dW / dbias of rssf_conv_wgrad_preact, dx and the
    producer's BatchNorm-backward statistics of rssf_conv_gather_bnbwd - from one pass over the raw input."""
    from representationlearning_amd import nnf, _lib as L
    lib = L.load()
    torch.manual_seed(43)
    cin, cout = 128, 32
    conv = nn.Conv2d(cin, cout, 1).to(DEV)
    spec = nnf.spec_of([conv])
    raw_in = (torch.randn(B, H, W, cin, device=DEV) * 1.2 + 0.1).bfloat16()
    dout = torch.randn(B, H, W, cout, device=DEV).bfloat16()
    ss = torch.stack([torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.3]).contiguous()
    assert lib.rssf_conv_wgrad_preact_dgrad_supported(B, H, W, cin, cout, L.dtype_code(raw_in)) == 1
    w = conv.weight.detach().contiguous()
    link = nnf.BnBwdLink()
    link.raw, link.rp, link.act, link.C, link.ss = raw_in, None, act, cin, ss
    outs = []
    for fused in (False, True):
        dw, db = torch.zeros_like(w), torch.zeros(cout, device=DEV)
        sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * cin, device=DEV)
        if fused:
            dx = torch.full((B, H, W, cin), 7.0, device=DEV).bfloat16()
            nnf._conv_wgrad(spec, dout, raw_in, [dw], db, xpre=(ss, act), dgrad=(w, dx, sm))
        else:
            nnf._conv_wgrad(spec, dout, raw_in, [dw], db, xpre=(ss, act))
            dx = nnf._conv_dgrad(spec, dout, [w], (B, H, W, cin), None, bn=(link, sm)).clone()
        outs.append((dw, db, dx, sm.view(nnf.BN_BWD_SLOTS, 2, cin).sum(0)))
    torch.cuda.synchronize()
    a, b = outs
    z = raw_in.float() * ss[0] + ss[1]
    y = torch.relu(z) if act == 1 else (F.gelu(z) if act == 2 else z)
    ref_dw = torch.einsum("bhwo,bhwi->oi", dout.float(), y.bfloat16().float()).view(cout, cin, 1, 1)
    assert rel_err(b[0].cpu(), ref_dw.cpu()) < 3e-3 and rel_err(b[0].cpu(), a[0].cpu()) < 2e-5
    assert rel_err(b[1].cpu(), a[1].cpu()) < 2e-5
    ref_dx = torch.einsum("bhwo,oi->bhwi", dout.float(), w.bfloat16().float().view(cout, cin))
    assert rel_err(b[2].float().cpu(), ref_dx.cpu()) < 4e-3 and rel_err(b[2].float().cpu(), a[2].float().cpu()) < 2e-3
    assert rel_err(b[3].cpu(), a[3].cpu()) < 2e-3


@pytest.mark.parametrize("B,H,W", [(2, 128, 128), (1, 256, 128), (3, 64, 192)])
@pytest.mark.parametrize("training", [True, False])
def test_stem_backward_apply_and_weight_gradient_in_one_pass(B, H, W, training):
    """csrc/conv_wgrad_stem.hip (the stem's Conv2d(3, 64, 3, stride 2, padding 1) behind BatchNorm + ReLU, _hrnet_rssformer.py:407-413): draw /
    dgamma / dbeta of rssf_bn_bwd_apply bit for bit, the weight gradient of the generic kernel and of torch; with RSSF_WGRAD_NO_DRAW the
    same gradients and `draw` left alone."""
    from representationlearning_amd import nnf, _lib as L
    lib = L.load()
    torch.manual_seed(51)
    conv = nn.Conv2d(3, 64, 3, 2, 1, bias=False).to(DEV)
    spec = nnf.spec_of([conv])
    OH, OW, C = H // 2, W // 2, 64
    x = torch.randn(B, H, W, 3, device=DEV).bfloat16()
    dy = torch.randn(B, OH, OW, C, device=DEV).bfloat16()
    raw = (torch.randn(B, OH, OW, C, device=DEV) * 1.3 + 0.2).bfloat16()
    mean, var = raw.float().mean((0, 1, 2)), raw.float().var((0, 1, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.2
    ss = torch.stack([gamma * istd, beta - mean * gamma * istd]).contiguous()
    mi = torch.stack([mean, istd]).contiguous()
    rows, n = B * OH * OW, float(B * OH * OW)
    sums = torch.zeros(nnf.BN_BWD_SLOTS * 2 * C, device=DEV)
    L.check(lib.rssf_bn_bwd_reduce(L.ptr(dy), L.ptr(raw), L.ptr(ss), None, L.ptr(sums), rows, C, 1, None, L.dtype_code(raw), L.stream()), "reduce")
    outs = []
    for generic, no_draw in ((True, False), (False, False), (False, True)):
        draw = torch.full_like(raw, 7.0)
        dg, dbt = torch.full((C,), 0.125, device=DEV), torch.full((C,), 0.25, device=DEV)
        dw = torch.zeros_like(conv.weight, dtype=torch.float32)
        bn = (dy, raw, ss, mi, sums, None, None, dg, dbt, 1, n, training, 0.5)
        nnf._conv_wgrad(spec, draw, x, [dw], None, bn=bn, generic=generic, no_draw=no_draw)
        outs.append((draw, dg, dbt, dw))
    torch.cuda.synchronize()
    a, b, c = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert torch.equal(b[1], c[1]) and torch.equal(b[2], c[2]) and bool((c[0].float() == 7.0).all())        # draw untouched on request
    xr = x.permute(0, 3, 1, 2).float()
    w = conv.weight.detach().clone().float().requires_grad_(True)
    F.conv2d(xr, w, None, 2, 1).backward(a[0].permute(0, 3, 1, 2).float())
    assert float(b[3].abs().max()) > 0
    assert rel_err(b[3].cpu(), w.grad.cpu()) < 2e-5 and rel_err(b[3].cpu(), a[3].cpu()) < 2e-5 and rel_err(c[3].cpu(), b[3].cpu()) < 1e-6


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 128, 64), (3, 16, 48), (1, 1, 16)])        # (H, W) of the OUTPUT gradient; dx is twice that
@pytest.mark.parametrize("act,res", [(None, False), (1, False), (1, True), (2, False)])
def test_stride2_dgrad_stream_matches_gather_and_torch(B, H, W, act, res):
    """csrc/conv_dgrad_s2.hip (the stem's Conv2d(64, 64, 3, stride 2, padding 1), _hrnet_rssformer.py:409-413): the data gradient by output
    parity against the generic gather kernel (RSSF_CONV_GENERIC) and torch, plain and with the producer's BatchNorm-backward statistics."""
    from representationlearning_amd import nnf
    torch.manual_seed(61)
    conv = nn.Conv2d(64, 64, 3, 2, 1, bias=False).to(DEV)
    spec = nnf.spec_of([conv])
    dout = torch.randn(B, H, W, 64, device=DEV).bfloat16()
    shape = (B, 2 * H, 2 * W, 64)
    link = None
    if act is not None:
        link = nnf.BnBwdLink()
        link.raw, link.act, link.C = torch.randn(*shape, device=DEV).bfloat16(), act, 64
        link.rp = torch.randn(*shape, device=DEV).bfloat16() if res else None
        link.ss = torch.stack([torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.3]).contiguous()
    got = []
    for on in (False, True):
        sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * 64, device=DEV)
        d = nnf._conv_dgrad(spec, dout, [conv.weight.detach()], shape, None, bn=(link, sm) if link is not None else None, generic=not on).clone()
        got.append((d, sm.view(nnf.BN_BWD_SLOTS, 2, 64).sum(0)))
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(dout.permute(0, 3, 1, 2).float(), conv.weight.detach().bfloat16().float(), None, 2, 1, output_padding=1).permute(0, 2, 3, 1)
    assert rel_err(got[1][0].float().cpu(), ref.cpu()) < 4e-3 and rel_err(got[1][0].float().cpu(), got[0][0].float().cpu()) < 2e-3
    if link is not None:
        assert rel_err(got[1][1].cpu(), got[0][1].cpu()) < 2e-3


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 128, 96), (3, 32, 32), (1, 2, 32)])
def test_stem_forward_stream_matches_gather_and_torch(B, H, W):
    """csrc/conv_stem_fwd.hip (Conv2d(3, 64, 3, stride 2, padding 1) on the channel-padded image, _hrnet_rssformer.py:407, 441): against the
    generic gather kernel and torch, with the fused BatchNorm statistics."""
    from representationlearning_amd import nnf
    torch.manual_seed(71)
    conv = nn.Conv2d(3, 64, 3, 2, 1, bias=False).to(DEV)
    spec = nnf.spec_of([conv])
    x = torch.randn(B, H, W, 3, device=DEV).bfloat16()
    xp = nnf._pad_channels(x)
    outs, stats = [], []
    for on in (False, True):
        st = torch.zeros(nnf.BN_SLOTS * 2 * 64, device=DEV)
        outs.append(nnf._conv_forward(spec, xp, [conv.weight.detach()], None, st, generic=not on))
        stats.append(st.view(nnf.BN_SLOTS, 2, 64).sum(0))
        outs.append(nnf._conv_forward(spec, xp, [conv.weight.detach()], None, None, generic=not on))
    torch.cuda.synchronize()
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), conv.weight.detach().bfloat16().float(), None, 2, 1).permute(0, 2, 3, 1)
    for o in (outs[2], outs[3]):
        assert tuple(o.shape) == (B, H // 2, W // 2, 64)
        assert rel_err(o.float().cpu(), ref.cpu()) < 6e-3
    assert rel_err(outs[2].float().cpu(), outs[0].float().cpu()) < 3e-3 and torch.equal(outs[2], outs[3])
    assert rel_err(stats[1].cpu(), stats[0].cpu()) < 3e-4
    n = B * (H // 2) * (W // 2)
    assert rel_err(stats[1][1].cpu() / n, (ref.reshape(-1, 64) ** 2).mean(0).cpu()) < 1e-3


@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 64), (32, 32), (32, 128), (64, 32)])        # the convolution: dx has cin channels, dout cout
@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 16, 64), (3, 8, 16)])
@pytest.mark.parametrize("mode", ["plain", "addend", "in_place", "bn"])
def test_stride2_dgrad_stream_shapes_and_accumulation(cin, cout, B, H, W, mode):
    """The down-sampling fuse convolutions of the HighResolutionModules (_hrnet_rssformer.py:380-405) on csrc/conv_dgrad_s2.hip: every
    instantiated channel pair, with a skip-gradient addend, accumulating in place (nnf.GradAccum) and with the producer's statistics."""
    from representationlearning_amd import nnf
    torch.manual_seed(63)
    conv = nn.Conv2d(cin, cout, 3, 2, 1, bias=False).to(DEV)
    spec = nnf.spec_of([conv])
    dout = torch.randn(B, H, W, cout, device=DEV).bfloat16()
    shape = (B, 2 * H, 2 * W, cin)
    base = torch.randn(*shape, device=DEV).bfloat16()
    link = None
    if mode == "bn":
        link = nnf.BnBwdLink()
        link.raw, link.act, link.C, link.rp = torch.randn(*shape, device=DEV).bfloat16(), 1, cin, None
        link.ss = torch.stack([torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.3]).contiguous()
    got = []
    for on in (False, True):
        sm = torch.zeros(nnf.BN_BWD_SLOTS * 2 * cin, device=DEV)
        if mode == "in_place":
            buf = base.clone()
            d = nnf._conv_dgrad(spec, dout, [conv.weight.detach()], shape, buf, out=buf, generic=not on).clone()
        else:
            d = nnf._conv_dgrad(spec, dout, [conv.weight.detach()], shape, base if mode == "addend" else None,
                                bn=(link, sm) if link is not None else None, generic=not on).clone()
        got.append((d, sm.view(nnf.BN_BWD_SLOTS, 2, cin).sum(0)))
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(dout.permute(0, 3, 1, 2).float(), conv.weight.detach().bfloat16().float(), None, 2, 1, output_padding=1).permute(0, 2, 3, 1)
    if mode in ("addend", "in_place"):
        ref = ref + base.float()
    # (two bf16 roundings of sums formed in different orders: each within 2^-9 of the exact value, up to ~2.5e-3 of each other)
    assert rel_err(got[1][0].float().cpu(), ref.cpu()) < 4e-3 and rel_err(got[1][0].float().cpu(), got[0][0].float().cpu()) < 3e-3
    if link is not None:
        assert rel_err(got[1][1].cpu(), got[0][1].cpu()) < 2e-3


@pytest.mark.parametrize("cout", [32, 64])
@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 32, 64), (3, 16, 32)])
@pytest.mark.parametrize("fused", [False, True])
def test_stride2_weight_gradient_block_owns_all_taps(cout, B, H, W, fused):
    """The 32-input-channel instances of csrc/conv_wgrad_stem.hip (the down-sampling fuse convolutions, _hrnet_rssformer.py:380-405): weight
    gradient against the generic one-tap-per-block kernel and torch, plain and with the BatchNorm-backward apply in the launch."""
    from representationlearning_amd import nnf, _lib as L
    lib = L.load()
    torch.manual_seed(81)
    conv = nn.Conv2d(32, cout, 3, 2, 1, bias=False).to(DEV)
    spec = nnf.spec_of([conv])
    OH, OW = H // 2, W // 2
    x = torch.randn(B, H, W, 32, device=DEV).bfloat16()
    dy = torch.randn(B, OH, OW, cout, device=DEV).bfloat16()
    raw = (torch.randn(B, OH, OW, cout, device=DEV) * 1.3 + 0.2).bfloat16()
    mean, var = raw.float().mean((0, 1, 2)), raw.float().var((0, 1, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    gamma, beta = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.2
    ss = torch.stack([gamma * istd, beta - mean * gamma * istd]).contiguous()
    mi = torch.stack([mean, istd]).contiguous()
    rows, n = B * OH * OW, float(B * OH * OW)
    sums = torch.zeros(nnf.BN_BWD_SLOTS * 2 * cout, device=DEV)
    L.check(lib.rssf_bn_bwd_reduce(L.ptr(dy), L.ptr(raw), L.ptr(ss), None, L.ptr(sums), rows, cout, 1, None, L.dtype_code(raw), L.stream()), "reduce")
    outs = []
    for generic in (True, False):
        dw = torch.zeros_like(conv.weight, dtype=torch.float32)
        if fused:
            draw = torch.empty_like(raw)
            dg, dbt = torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV)
            nnf._conv_wgrad(spec, draw, x, [dw], None, bn=(dy, raw, ss, mi, sums, None, None, dg, dbt, 1, n, True, 1.0), generic=generic)
            outs.append((dw, draw, dg, dbt))
        else:
            nnf._conv_wgrad(spec, dy, x, [dw], None, generic=generic)
            outs.append((dw, dy, None, None))
    torch.cuda.synchronize()
    a, b = outs
    if fused:
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    xr = x.permute(0, 3, 1, 2).float()
    w = conv.weight.detach().clone().float().requires_grad_(True)
    F.conv2d(xr, w, None, 2, 1).backward(a[1].permute(0, 3, 1, 2).float())
    assert float(b[0].abs().max()) > 0
    assert rel_err(b[0].cpu(), w.grad.cpu()) < 2e-5 and rel_err(b[0].cpu(), a[0].cpu()) < 2e-5
