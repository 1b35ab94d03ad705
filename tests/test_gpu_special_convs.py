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
