"""GPU parity of the implicit-GEMM convolution + fused BatchNorm/activation kernels (through the C ABI) against the
CPU restatement (plain torch fp32 conv2d / batch_norm / gelu as used by oracle/rssformer_cpu.py)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk_conv(ci, co, k, s=1, p=0, d=1, bias=False, seed=0):
    torch.manual_seed(seed)
    return nn.Conv2d(ci, co, k, s, p, d, bias=bias)


def _mk_bn(c, seed=0, sync=False):
    torch.manual_seed(seed + 100)
    bn = (nn.SyncBatchNorm if sync else nn.BatchNorm2d)(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.8, 1.3)
    return bn


CONVS = [  # ci, co, k, stride, pad, dil, bias, B, H, W
    (32, 32, 3, 1, 1, 1, False, 2, 17, 13),
    (64, 64, 3, 1, 1, 1, False, 1, 16, 16),
    (32, 64, 3, 2, 1, 1, False, 2, 16, 14),
    (64, 128, 3, 2, 1, 1, False, 1, 9, 11),
    (3, 64, 3, 2, 1, 1, False, 2, 32, 32),
    (64, 256, 1, 1, 0, 1, False, 1, 12, 12),
    (256, 64, 1, 1, 0, 1, False, 1, 12, 12),
    (18, 18, 3, 1, 1, 1, False, 1, 10, 10),
    (36, 18, 1, 1, 0, 1, False, 1, 8, 8),
    (128, 128, 3, 1, 6, 6, True, 1, 20, 20),
    (480, 480, 1, 1, 0, 1, True, 1, 8, 8),
    (256, 256, 3, 1, 1, 1, False, 2, 4, 4),
    (48, 96, 3, 1, 1, 1, True, 3, 21, 37),          # halo kernel: ragged tiles, partial channel chunk, bias
    (64, 128, 3, 1, 1, 1, False, 4, 64, 128),       # halo kernel: 64-wide channel tile
    (480, 6, 1, 1, 0, 1, True, 1, 16, 16),          # the head's shape: 6 output channels (gradient padded to 8)
    (128, 128, 3, 1, 1, 1, False, 16, 32, 32),      # halo kernel at branch 2 of a Base step (8-row tiles, 4 channel chunks)
    (160, 160, 3, 1, 1, 1, True, 2, 12, 20),        # ... five channel chunks, 4-row tiles, bias
]


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("act,mode", [(1, "train"), (2, "train"), (0, "eval")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_bn_act(cfg, act, mode, dtype):
    from representationlearning_amd import nnf
    ci, co, k, s, p, d, bias, B, H, W = cfg
    conv, bn = _mk_conv(ci, co, k, s, p, d, bias), _mk_bn(co)
    bn.train(mode == "train")
    torch.manual_seed(5)
    x = torch.randn(B, ci, H, W).to(dtype).float()
    # CPU reference (fp32)
    conv_r, bn_r = _mk_conv(ci, co, k, s, p, d, bias), _mk_bn(co)
    bn_r.train(mode == "train")
    xr = x.clone().requires_grad_()
    z = bn_r(conv_r(xr))
    res = torch.randn_like(z).to(dtype).float()
    resr = res.clone().requires_grad_()
    yr = {0: lambda t: t, 1: F.relu, 2: F.gelu}[act](z + resr)
    gy = torch.randn_like(yr).to(dtype).float()
    yr.backward(gy)
    # HIP
    conv, bn = conv.to(DEV), bn.to(DEV)
    xd = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    rd = res.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = nnf.conv_bn_act(xd, conv, bn, act, res_pre=rd)
    y.backward(gy.to(DEV).to(dtype))
    f32 = dtype == torch.float32
    assert rel_err(y.detach().float().cpu(), yr.detach()) < (2e-5 if f32 else 1.5e-2)
    assert rel_err(xd.grad.float().cpu(), xr.grad) < (2e-4 if f32 else 7e-2)
    # bf16: ReLU/GELU masks flip where z ~ 0 differs between bf16 and fp32 arithmetic -> a few full-size element errors
    assert rel_err(rd.grad.float().cpu(), resr.grad) < (2e-5 if f32 else 8e-2)
    assert rel_err(conv.weight.grad.cpu(), conv_r.weight.grad) < (2e-4 if f32 else 7e-2)
    assert rel_err(bn.weight.grad.cpu(), bn_r.weight.grad) < (2e-4 if f32 else 7e-2)
    assert rel_err(bn.bias.grad.cpu(), bn_r.bias.grad) < (2e-4 if f32 else 7e-2)
    if mode == "train":
        assert rel_err(bn.running_mean.cpu(), bn_r.running_mean) < (1e-5 if f32 else 5e-3)
        assert rel_err(bn.running_var.cpu(), bn_r.running_var) < (1e-5 if f32 else 5e-3)
    if bias and mode == "eval":
        assert rel_err(conv.bias.grad.cpu(), conv_r.bias.grad) < (2e-4 if f32 else 7e-2)


@pytest.mark.parametrize("C,B,H,W", [(128, 1, 30, 26), (72, 2, 14, 14)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_mlp_conv(C, B, H, W, dtype):
    """The 19-tap fused {1x1 + 3x3 dil 6 + 3x3 dil 12} convolution + SyncBN + GELU + post-activation residual."""
    from representationlearning_amd import nnf

    def mk():
        torch.manual_seed(3)
        return [nn.Conv2d(C, C, 1, 1), nn.Conv2d(C, C, 3, 1, padding=6, dilation=6), nn.Conv2d(C, C, 3, 1, padding=12, dilation=12)]
    cr, bnr = mk(), _mk_bn(C, sync=False).train()
    ch, bnh = [c.to(DEV) for c in mk()], _mk_bn(C, sync=True).to(DEV).train()
    torch.manual_seed(9)
    x = torch.randn(B, C, H, W).to(dtype).float()
    post = torch.randn(B, C, H, W).to(dtype).float()
    xr = x.clone().requires_grad_()
    yr = F.gelu(bnr(cr[0](xr) + cr[1](xr) + cr[2](xr))) + post
    gy = torch.randn_like(yr).to(dtype).float()
    yr.backward(gy)
    xd = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    pd = post.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = nnf.conv_bn_act(xd, ch, bnh, 2, res_post=pd)
    y.backward(gy.to(DEV).to(dtype))
    f32 = dtype == torch.float32
    assert rel_err(y.detach().float().cpu(), yr.detach()) < (2e-5 if f32 else 1.5e-2)
    assert rel_err(xd.grad.float().cpu(), xr.grad) < (3e-4 if f32 else 7e-2)
    assert rel_err(pd.grad.float().cpu(), gy) < 1e-6
    for a, b in zip(ch, cr):
        assert rel_err(a.weight.grad.cpu(), b.weight.grad) < (3e-4 if f32 else 7e-2)


def test_fused_mlp_conv_wide_tile_vs_torch_fp32():
    """B=8 x 128 ch x 128^2 (>= 512 tiles of 256 pixels): the MLP sum takes the 256 x 128 tile of the gather kernel, which
    the small cases never reach.  Reference: torch fp32 convolutions on the same device over the bf16-rounded operands."""
    from representationlearning_amd import nnf
    C, B, H, W = 128, 8, 128, 128
    torch.manual_seed(5)
    convs = [nn.Conv2d(C, C, 1, 1).to(DEV), nn.Conv2d(C, C, 3, 1, padding=6, dilation=6).to(DEV),
             nn.Conv2d(C, C, 3, 1, padding=12, dilation=12).to(DEV)]
    bn = _mk_bn(C, sync=True).to(DEV).train()
    bnr = _mk_bn(C, sync=False).to(DEV).train()
    x = torch.randn(B, C, H, W, device=DEV).bfloat16()
    gy = torch.randn(B, C, H, W, device=DEV).bfloat16()
    xr = x.float().requires_grad_()
    wr = [c.weight.detach().clone().requires_grad_() for c in convs]
    yr = F.gelu(bnr(sum(F.conv2d(xr, w, c.bias.detach(), 1, c.padding, c.dilation) for w, c in zip(wr, convs))))
    yr.backward(gy.float())
    xd = x.contiguous(memory_format=torch.channels_last).requires_grad_()
    y = nnf.conv_bn_act(xd, convs, bn, 2)
    y.backward(gy)
    assert rel_err(y.detach().float().cpu(), yr.detach().cpu()) < 1.5e-2
    assert rel_err(xd.grad.float().cpu(), xr.grad.cpu()) < 7e-2
    for a, b in zip(convs, wr):
        assert rel_err(a.weight.grad.cpu(), b.grad.cpu()) < 7e-2


def test_conv_bias_head():
    from representationlearning_amd import nnf
    conv_r = _mk_conv(480, 6, 1, bias=True)
    conv = _mk_conv(480, 6, 1, bias=True).to(DEV)
    x = torch.randn(2, 480, 9, 7)
    xr = x.clone().requires_grad_()
    yr = conv_r(xr)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = nnf.conv_bias(xd, conv)
    y.backward(gy.to(DEV))
    assert rel_err(y.detach().cpu(), yr.detach()) < 2e-5
    assert rel_err(xd.grad.cpu(), xr.grad) < 2e-4
    assert rel_err(conv.weight.grad.cpu(), conv_r.weight.grad) < 2e-4
    assert rel_err(conv.bias.grad.cpu(), conv_r.bias.grad) < 2e-4


@pytest.mark.parametrize("order", [(0, 1, 2), (2, 0, 1)])
def test_grad_accum_matches_autograd_sum(order):
    """nnf.fanout / GradAccum: a tensor feeding three convolutions (3x3 stride 1, 3x3 stride 2, 1x1) and one non-convolution consumer
    gets the same gradient as through autograd's own accumulation, whatever order the consumers were built in; the convolutions'
    parameter gradients are untouched by the in-place accumulation (reference: the fuse layer fan-out, _hrnet_rssformer.py:361-435)."""
    from representationlearning_amd import nnf
    torch.manual_seed(3)
    convs = [_mk_conv(32, 32, 3, 1, 1, seed=1).to(DEV), _mk_conv(32, 64, 3, 2, 1, seed=2).to(DEV), _mk_conv(32, 16, 1, seed=3).to(DEV)]
    bns = [_mk_bn(32, 1).to(DEV).train(), _mk_bn(64, 2).to(DEV).train(), _mk_bn(16, 3).to(DEV).train()]
    x0 = torch.randn(2, 32, 24, 20, device=DEV).contiguous(memory_format=torch.channels_last)
    gs = [torch.randn(2, 32, 24, 20, device=DEV), torch.randn(2, 64, 12, 10, device=DEV), torch.randn(2, 16, 24, 20, device=DEV)]
    res = {}
    for fused in (False, True):
        for c in convs:
            c.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        src = x * 1.0                                        # a non-leaf, as in the model
        xa, acc = nnf.fanout(src, 3) if fused else (src, None)
        assert (acc is not None) == fused
        loss = (xa * xa).sum() * 0.01                          # the non-convolution consumer
        for k in order:
            y = nnf.conv_bn_act(xa, convs[k], bns[k], nnf.ACT_RELU, grad_accum=acc)
            loss = loss + (y * gs[k]).sum()
        loss.backward()
        assert acc is None or acc.buf is None                  # handed over and released
        res[fused] = (x.grad.clone(), [c.weight.grad.clone() for c in convs])
    assert rel_err(res[True][0].cpu(), res[False][0].cpu()) < 1e-5
    for a_, b_ in zip(res[True][1], res[False][1]):
        assert rel_err(a_.cpu(), b_.cpu()) < 1e-6


@pytest.mark.parametrize("k,stride,cin,cout,H,W,act,res,dtype", [
    (3, 1, 32, 32, 24, 20, 1, False, torch.bfloat16),      # halo kernel
    (3, 1, 64, 64, 17, 33, 1, True, torch.bfloat16),       # halo kernel, ragged edges, residual before the activation
    (3, 1, 128, 128, 16, 16, 2, False, torch.bfloat16),    # halo kernel, GELU
    (3, 1, 256, 256, 16, 16, 1, True, torch.bfloat16),     # halo kernel, 8 channel chunks, residual before the activation
    (1, 1, 128, 32, 20, 24, 2, False, torch.bfloat16),     # 1x1 (MlpDWBN fc2's data gradient: 128-wide statistics)
    (1, 1, 32, 128, 20, 24, 0, True, torch.bfloat16),      # 1x1, no activation, residual
    (3, 2, 64, 64, 24, 24, 1, False, torch.bfloat16),      # stride 2 (the stem's conv2): strided data gradient
    (3, 1, 18, 18, 12, 12, 1, False, torch.bfloat16),      # channels not a multiple of 8: convolution + separate pass inside
    (3, 1, 32, 32, 12, 16, 1, True, torch.float32),        # fp32 mode: convolution + separate pass inside
])
def test_conv_gather_bnbwd_equals_separate_reduce(k, stride, cin, cout, H, W, act, res, dtype):
    """rssf_conv_gather_bnbwd (data gradient + the BatchNorm-backward statistics of the previous layer in one launch) against
    rssf_conv_gather_add followed by rssf_bn_bwd_reduce: identical gradient, statistics equal up to summation order.
    `cin` = channels of the convolution INPUT (= of the gradient produced = of the BatchNorm whose statistics are formed)."""
    from representationlearning_amd import _lib as L, nnf
    lib = L.load()
    torch.manual_seed(5)
    B = 3
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False).to(DEV)
    spec = nnf.spec_of([conv])
    OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    dout = torch.randn(B, OH, OW, cout, device=DEV).to(dtype)
    raw = torch.randn(B, H, W, cin, device=DEV).to(dtype)
    rp = torch.randn(B, H, W, cin, device=DEV).to(dtype) if res else None
    ss = torch.cat([torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.3]).contiguous()
    dp = nnf._pad_channels(dout)
    wpk = nnf._pack(spec, [conv.weight], True, dout.dtype, dout.device)
    dx1, dx2 = torch.empty_like(raw), torch.empty_like(raw)
    s1 = torch.zeros(nnf.BN_BWD_SLOTS * 2 * cin, device=DEV)
    s2 = torch.zeros_like(s1)
    common = (B, OH, OW, dp.shape[3], H, W, cin, 1, spec.stride, spec.ntaps, spec.c_ndy, spec.c_ndx, L.dtype_code(dout), L.stream())
    L.check(lib.rssf_conv_gather_add(L.ptr(dp), L.ptr(wpk), L.ptr(dx1), None, None, None, None, *common), "dgrad")
    L.check(lib.rssf_bn_bwd_reduce(L.ptr(dx1), L.ptr(raw), L.ptr(ss), L.ptr(rp), L.ptr(s1), B * H * W, cin, act, None, L.dtype_code(raw), L.stream()),
            "reduce")
    L.check(lib.rssf_conv_gather_bnbwd(L.ptr(dp), L.ptr(wpk), L.ptr(dx2), None, L.ptr(raw), L.ptr(rp), L.ptr(ss), act, L.ptr(s2), *common), "fused")
    assert torch.equal(dx1, dx2)
    a_, b_ = s1.view(-1, 2, cin).sum(0), s2.view(-1, 2, cin).sum(0)
    assert rel_err(b_.cpu(), a_.cpu()) < 2e-6
    # with the skip gradient as addend (the BasicBlock case): the statistics see the SUM
    add = torch.randn_like(raw)
    s3, s4 = torch.zeros_like(s1), torch.zeros_like(s1)
    dx3, dx4 = torch.empty_like(raw), torch.empty_like(raw)
    L.check(lib.rssf_conv_gather_add(L.ptr(dp), L.ptr(wpk), L.ptr(dx3), None, None, L.ptr(add), None, *common), "dgrad+add")
    L.check(lib.rssf_bn_bwd_reduce(L.ptr(dx3), L.ptr(raw), L.ptr(ss), L.ptr(rp), L.ptr(s3), B * H * W, cin, act, None, L.dtype_code(raw), L.stream()),
            "reduce")
    L.check(lib.rssf_conv_gather_bnbwd(L.ptr(dp), L.ptr(wpk), L.ptr(dx4), L.ptr(add), L.ptr(raw), L.ptr(rp), L.ptr(ss), act, L.ptr(s4), *common), "fused+add")
    assert torch.equal(dx3, dx4)
    assert rel_err(s4.view(-1, 2, cin).sum(0).cpu(), s3.view(-1, 2, cin).sum(0).cpu()) < 2e-6


@pytest.mark.parametrize("k,cin,cout,H,W,act,res,training,dtype", [
    (3, 32, 32, 24, 20, 1, False, 1, torch.bfloat16),      # halo weight-gradient kernel
    (3, 64, 64, 17, 33, 1, True, 1, torch.bfloat16),       # 2 x 2 channel tiles, ragged edges, residual (dres written)
    (3, 32, 128, 16, 16, 2, False, 1, torch.bfloat16),     # GELU, more output than input tiles
    (3, 32, 32, 16, 16, 1, True, 0, torch.bfloat16),       # eval-mode BatchNorm (draw = scale * dz)
    (1, 32, 64, 20, 24, 1, False, 1, torch.bfloat16),      # 1x1: apply + plain weight gradient inside the entry point
    (3, 32, 32, 12, 16, 1, True, 1, torch.float32),        # fp32: likewise
])
def test_conv_wgrad_bnapply_equals_apply_then_wgrad(k, cin, cout, H, W, act, res, training, dtype):
    """rssf_conv_wgrad_bnapply (BatchNorm-backward apply inside the weight-gradient launch) against rssf_bn_bwd_apply followed by
    rssf_conv_wgrad: draw, dres, dgamma / dbeta and the weight gradient are bit-identical (same operation order)."""
    from representationlearning_amd import _lib as L, nnf
    lib = L.load()
    torch.manual_seed(7)
    B = 3
    conv = nn.Conv2d(cin, cout, k, 1, k // 2, bias=False).to(DEV)
    spec = nnf.spec_of([conv])
    x = torch.randn(B, H, W, cin, device=DEV).to(dtype)
    dy = torch.randn(B, H, W, cout, device=DEV).to(dtype)
    raw = torch.randn(B, H, W, cout, device=DEV).to(dtype)
    rp = torch.randn(B, H, W, cout, device=DEV).to(dtype) if res else None
    ss = torch.cat([torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.3]).contiguous()
    mi = torch.cat([torch.randn(cout, device=DEV) * 0.2, torch.rand(cout, device=DEV) + 0.5]).contiguous()
    sums = torch.randn(nnf.BN_BWD_SLOTS * 2 * cout, device=DEV)
    n = float(B * H * W)
    nws = lib.rssf_conv_wgrad_workspace_elems(B, H, W, cin, cout, spec.ntaps)
    out = {}
    for fused in (False, True):
        draw = torch.empty_like(raw)
        dres = torch.empty_like(raw) if res else None
        dg, db = torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV)
        dw = torch.zeros_like(conv.weight, dtype=torch.float32)
        ws = torch.empty(nws, device=DEV)
        tail = (L.ptr(dw), None, None, spec.c_ksizes, 1, spec.c_src, spec.c_kpos, spec.c_alias, None, L.ptr(ws), B, H, W, cin, H, W, cout, 1,
                spec.ntaps, spec.c_dy, spec.c_dx, None, L.dtype_code(x), L.stream())
        if fused:
            L.check(lib.rssf_conv_wgrad_bnapply(L.ptr(dy), L.ptr(raw), L.ptr(ss), L.ptr(mi), L.ptr(sums), L.ptr(rp), L.ptr(draw), L.ptr(dres), L.ptr(dg),
                                                L.ptr(db), act, n, training, 0.5, L.ptr(x), None, 0, *tail), "fused")
        else:
            L.check(lib.rssf_bn_bwd_apply(L.ptr(dy), L.ptr(raw), L.ptr(ss), L.ptr(mi), L.ptr(sums), L.ptr(rp), L.ptr(draw), L.ptr(dres), L.ptr(dg),
                                          L.ptr(db), B * H * W, cout, act, n, training, 0.5, L.dtype_code(raw), L.stream()), "apply")
            L.check(lib.rssf_conv_wgrad(L.ptr(draw), L.ptr(x), *tail), "wgrad")
        out[fused] = (draw, dres, dg, db, dw)
    for a_, b_, name in zip(out[True], out[False], ("draw", "dres", "dgamma", "dbeta", "dw")):
        if a_ is not None:
            assert torch.equal(a_, b_), name


@pytest.mark.parametrize("cin,cout,H,W,act", [(32, 32, 24, 20, 1), (64, 64, 17, 33, 1), (32, 64, 16, 16, 2), (128, 128, 9, 16, 0), (256, 256, 16, 16, 1), (160, 32, 11, 9, 2)])
def test_conv_preact_input_equals_apply_then_conv(cin, cout, H, W, act):
    """rssf_conv_gather_preact / rssf_conv_wgrad_bnapply(in_scale_shift): the producer's BatchNorm + activation applied while the
    consumer stages the RAW tensor, against rssf_bn_apply followed by the plain launches - bit-identical output, fused forward
    statistics equal up to atomics order, bit-identical weight gradient; the zero padding is that of the ACTIVATION."""
    from representationlearning_amd import _lib as L, nnf
    lib = L.load()
    torch.manual_seed(11)
    B = 3
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=False).to(DEV)
    spec = nnf.spec_of([conv])
    code = L.RSSF_BF16
    assert lib.rssf_conv_gather_preact_supported(B, H, W, cin, H, W, cout, 1, 1, spec.ntaps, spec.c_dy, spec.c_dx, code) == 1
    assert lib.rssf_conv_wgrad_preact_supported(B, H, W, cin, H, W, cout, 1, spec.ntaps, 1, spec.c_dy, spec.c_dx, 0, code) == 1
    assert lib.rssf_conv_gather_preact_supported(B, H, W, cin, H, W, cout, 1, 1, spec.ntaps, spec.c_dy, spec.c_dx, L.RSSF_F32) == 0
    raw_in = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    # the producer's BatchNorm: statistics slots as its convolution epilogue leaves them, beta != 0 so that act(shift) != 0
    n = float(B * H * W)
    x32 = raw_in.float().reshape(-1, cin)
    stats_in = torch.zeros(nnf.BN_SLOTS, 2, cin, device=DEV)
    stats_in[3, 0], stats_in[3, 1] = x32.sum(0), (x32 * x32).sum(0)
    gamma, beta = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.5 + 0.3
    y = torch.empty_like(raw_in)
    rm1, rv1, mi1, ss1 = torch.zeros(cin, device=DEV), torch.ones(cin, device=DEV), torch.empty(2 * cin, device=DEV), torch.empty(2 * cin, device=DEV)
    L.check(lib.rssf_bn_finalize_apply(L.ptr(raw_in), L.ptr(stats_in), L.ptr(gamma), L.ptr(beta), L.ptr(rm1), L.ptr(rv1), L.ptr(mi1), L.ptr(ss1), None, None,
                                       L.ptr(y), B * H * W, cin, act, n, 0.1, 1e-5, 1, code, L.stream()), "finalize+apply")
    rm2, rv2, mi2, ss_in = torch.zeros(cin, device=DEV), torch.ones(cin, device=DEV), torch.empty(2 * cin, device=DEV), torch.empty(2 * cin, device=DEV)
    wpk = nnf._pack(spec, [conv.weight], False, y.dtype, y.device)
    o1, o2 = torch.empty(B, H, W, cout, device=DEV, dtype=y.dtype), torch.empty(B, H, W, cout, device=DEV, dtype=y.dtype)
    st1 = torch.zeros(nnf.BN_SLOTS * 2 * cout, device=DEV)
    st2 = torch.zeros_like(st1)
    dims = (B, H, W, cin, H, W, cout, 1, 1, spec.ntaps, spec.c_dy, spec.c_dx, code, L.stream())
    L.check(lib.rssf_conv_gather_add(L.ptr(y), L.ptr(wpk), L.ptr(o1), None, L.ptr(st1), None, None, *dims), "conv")
    L.check(lib.rssf_conv_gather_preact(L.ptr(raw_in), L.ptr(stats_in), L.ptr(gamma), L.ptr(beta), L.ptr(rm2), L.ptr(rv2), L.ptr(mi2), L.ptr(ss_in), n, 0.1,
                                        1e-5, 1, act, L.ptr(wpk), L.ptr(o2), None, L.ptr(st2), None, *dims), "preact conv")
    assert torch.equal(o1, o2)
    for a_, b_ in ((rm1, rm2), (rv1, rv2), (mi1, mi2), (ss1, ss_in)):            # the producer's BatchNorm was finalized by the launch
        assert torch.equal(a_, b_)
    assert rel_err(st2.view(-1, 2, cout).sum(0).cpu(), st1.view(-1, 2, cout).sum(0).cpu()) < 2e-6
    # weight gradient with the same raw operand (and the fused backward apply of THIS layer)
    dy = torch.randn(B, H, W, cout, device=DEV).bfloat16()
    raw = torch.randn(B, H, W, cout, device=DEV).bfloat16()
    ss = torch.cat([torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.3]).contiguous()
    mi = torch.cat([torch.randn(cout, device=DEV) * 0.2, torch.rand(cout, device=DEV) + 0.5]).contiguous()
    sums = torch.randn(nnf.BN_BWD_SLOTS * 2 * cout, device=DEV)
    nws = lib.rssf_conv_wgrad_workspace_elems(B, H, W, cin, cout, spec.ntaps)
    res = {}
    for fused in (False, True):
        draw = torch.empty_like(raw)
        dg, db = torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV)
        dw = torch.zeros_like(conv.weight, dtype=torch.float32)
        ws = torch.empty(nws, device=DEV)
        tail = (L.ptr(dw), None, None, spec.c_ksizes, 1, spec.c_src, spec.c_kpos, spec.c_alias, None, L.ptr(ws), B, H, W, cin, H, W, cout, 1,
                spec.ntaps, spec.c_dy, spec.c_dx, None, code, L.stream())
        head = (L.ptr(dy), L.ptr(raw), L.ptr(ss), L.ptr(mi), L.ptr(sums), None, L.ptr(draw), None, L.ptr(dg), L.ptr(db), 1, float(B * H * W), 1, 1.0)
        if fused:
            L.check(lib.rssf_conv_wgrad_bnapply(*head, L.ptr(raw_in), L.ptr(ss_in), act, *tail), "fused")
        else:
            L.check(lib.rssf_conv_wgrad_bnapply(*head, L.ptr(y), None, 0, *tail), "plain")
        res[fused] = (draw, dw)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


def test_conv_base_shape_linearity_bf16():
    """BASELINE config-2 size (B=16, 128->128 @128x128, 19 taps): linearity + finite (size-independent property)."""
    from representationlearning_amd import nnf
    torch.manual_seed(0)
    C = 128
    convs = [nn.Conv2d(C, C, 1, 1).to(DEV), nn.Conv2d(C, C, 3, 1, padding=6, dilation=6).to(DEV),
             nn.Conv2d(C, C, 3, 1, padding=12, dilation=12).to(DEV)]
    spec = nnf.spec_of(convs)
    cl = torch.channels_last
    a = torch.randn(16, C, 128, 128, device=DEV).bfloat16().contiguous(memory_format=cl)
    b = torch.randn(16, C, 128, 128, device=DEV).bfloat16().contiguous(memory_format=cl)
    w = [c.weight for c in convs]
    f = lambda t: nnf._conv_forward(spec, nnf._nhwc(t), w, None, None).float()
    ya, yb, yab = f(a), f(b), f((a.float() + b.float()).bfloat16())
    assert torch.isfinite(yab).all()
    assert rel_err((ya + yb).cpu(), yab.cpu()) < 2e-2


@pytest.mark.parametrize("B,C,IH,IW,OH,OW", [(2, 64, 6, 5, 12, 10), (1, 256, 2, 1, 16, 8), (2, 6, 8, 8, 32, 32), (1, 18, 5, 7, 20, 28),
                                                    (2, 2, 5, 5, 17, 13), (1, 7, 4, 6, 9, 11), (1, 6, 64, 64, 256, 256), (3, 16, 3, 3, 3, 3),
                                                    (1, 256, 4, 4, 32, 32), (2, 128, 8, 6, 32, 24), (1, 64, 5, 5, 40, 40),      # factors >= 4: the row-parallel backward
                                                    (2, 6, 128, 128, 512, 512), (1, 6, 37, 53, 148, 212)])      # the head's logits: the tiled few-channel backward (full and ragged tiles)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_bilinear(B, C, IH, IW, OH, OW, dtype):
    from representationlearning_amd import nnf
    torch.manual_seed(2)
    x = torch.randn(B, C, IH, IW).to(dtype).float()
    xr = x.clone().requires_grad_()
    yr = F.interpolate(xr, size=(OH, OW), mode="bilinear", align_corners=True)
    gy = torch.randn_like(yr).to(dtype).float()
    yr.backward(gy)
    xd = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = nnf.upsample_bilinear(xd, (OH, OW))
    y.backward(gy.to(DEV).to(dtype))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(y.detach().float().cpu(), yr.detach()) < tol
    assert rel_err(xd.grad.float().cpu(), xr.grad) < tol


@pytest.mark.parametrize("B,C,H,W", [(2, 32, 12, 8), (16, 32, 128, 128), (3, 24, 7, 5), (2, 20, 9, 9), (1, 64, 33, 31)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_aux_head_matches_torch(B, C, H, W, dtype):
    """nnf.aux_head = Linear(AdaptiveAvgPool2d(1)(f).flatten(1)) (hrnet_aux.py:86-87, 99-100): the vector pooling kernel (C a multiple of
    the 16-byte vector) and the scalar one, pixel counts that are not multiples of the 32 chunks."""
    from representationlearning_amd import nnf
    torch.manual_seed(5)
    lin = nn.Linear(C, 7).to(DEV)
    f = torch.randn(B, C, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    out = nnf.aux_head(f, lin)
    ref = F.linear(f.float().mean((2, 3)), lin.weight.detach().float(), lin.bias.detach().float())
    assert rel_err(out.cpu(), ref.cpu()) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_bilinear_concat(dtype):
    """SimpleFusion8's resize + cat (hrnet_aux.py:61-64) written slice by slice into one buffer, gradient read in place."""
    from representationlearning_amd import nnf
    torch.manual_seed(2)
    shapes = [(2, 32, 12, 10), (2, 64, 6, 5), (2, 128, 3, 3), (2, 24, 2, 1)]
    xs = [torch.randn(*sh).to(dtype).float() for sh in shapes]
    xr = [x.clone().requires_grad_() for x in xs]
    yr = torch.cat([xr[0]] + [F.interpolate(x, size=(12, 10), mode="bilinear", align_corners=True) for x in xr[1:]], dim=1)
    gy = torch.randn_like(yr).to(dtype).float()
    yr.backward(gy)
    xd = [x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_() for x in xs]
    y = nnf.upsample_bilinear_concat(xd, (12, 10))
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(DEV).to(dtype))
    f32 = dtype == torch.float32
    assert rel_err(y.detach().float().cpu(), yr.detach()) < (1e-6 if f32 else 4e-3)
    assert torch.equal(y.detach()[:, :32].float().cpu(), xs[0])                     # the un-resized branch is an exact copy
    for a, b in zip(xd, xr):
        assert rel_err(a.grad.float().cpu(), b.grad) < (1e-5 if f32 else 6e-3)


@pytest.mark.parametrize("B,C,H,W,s", [(2, 32, 5, 7, 2), (1, 36, 3, 3, 4), (1, 64, 2, 2, 8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_nearest_add(B, C, H, W, s, dtype):
    from representationlearning_amd import nnf
    torch.manual_seed(4)
    x = torch.randn(B, C, H, W).to(dtype).float()
    acc = torch.randn(B, C, H * s, W * s).to(dtype).float()
    xr, ar = x.clone().requires_grad_(), acc.clone().requires_grad_()
    yr = ar + F.interpolate(xr, scale_factor=s, mode="nearest")
    gy = torch.randn_like(yr).to(dtype).float()
    yr.backward(gy)
    cl = torch.channels_last
    xd = x.to(DEV).to(dtype).contiguous(memory_format=cl).requires_grad_()
    ad = acc.to(DEV).to(dtype).contiguous(memory_format=cl).requires_grad_()
    y = nnf.upsample_nearest_add(ad, xd, s)
    y.backward(gy.to(DEV).to(dtype))
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert rel_err(y.detach().float().cpu(), yr.detach()) < tol
    assert rel_err(xd.grad.float().cpu(), xr.grad) < tol
    assert rel_err(ad.grad.float().cpu(), ar.grad) < 1e-6
    y2 = nnf.upsample_nearest_add(None, xd, s)
    assert rel_err(y2.detach().float().cpu(), F.interpolate(x, scale_factor=s, mode="nearest")) < tol


def test_lds_transpose_read():
    """Semantics of ds_read_b64_tr_b16 that conv_wgrad.hip relies on: in each 16-lane group, lane i receives element
    (i % 4) of the 8-byte chunks addressed by lanes 4j + i/4 (j = 0..3)."""
    from representationlearning_amd import _lib as L
    addr = torch.tensor([(l >> 4) * 256 + ((l & 15) >> 2) * 40 + (l & 3) * 4 for l in range(64)], dtype=torch.int32, device=DEV)
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    L.check(L.load().rssf_debug_trread(L.ptr(addr), L.ptr(out), L.stream()), "rssf_debug_trread")
    got = out.cpu().view(64, 4)
    a = addr.cpu().tolist()
    for l in range(64):
        g, i = l >> 4, l & 15
        want = [a[g * 16 + 4 * j + i // 4] + i % 4 for j in range(4)]      # lds[k] == k
        assert got[l].tolist() == want, (l, got[l].tolist(), want)


@pytest.mark.parametrize("tag", ["mixed", "one_all_ignore", "no_fg", "all_fg"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cgfl_loss_vs_golden(tag, dtype):
    """HIP loss kernels vs the golden vectors of the reference's SegmentationLossaux (incl. edge cases)."""
    from oracle.procedural import proc_input
    from tests.helpers import golden
    from representationlearning_amd import nnf
    g = golden(f"loss_{tag}")
    y = torch.from_numpy(g["y"]).to(DEV)
    lg = (proc_input((3, 6, 12, 10), 0.4) * 2.0).to(dtype)
    aux = proc_input((3, 7), 1.3).to(DEV).requires_grad_()
    ld = lg.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    loss = nnf.cgfl_loss(ld, y, aux)
    loss.backward()
    f32 = dtype == torch.float32
    assert abs(float(loss.detach()) - float(g["loss"])) < (2e-6 if f32 else 5e-3) * max(1.0, abs(float(g["loss"])))
    assert rel_err(ld.grad.float().cpu(), g["glogits"]) < (1e-5 if f32 else 1e-2)
    assert aux.grad is None


@pytest.mark.parametrize("K", [2, 5, 6, 7, 9])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cgfl_loss_class_counts_vs_oracle(K, dtype):
    """The loss kernels at other class counts than the goldens' six: the compile-time instantiations (6, 7 = LoveDA) and the run-time
    form, against the CPU oracle on the values the kernel reads."""
    from oracle import rssformer_cpu as O
    from representationlearning_amd import nnf
    torch.manual_seed(K)
    lg = (torch.randn(2, K, 9, 13) * 2.0).to(dtype)
    y = torch.randint(-1, K, (2, 9, 13))
    aux = torch.randn(2, 7)
    lr = lg.float().clone().requires_grad_()
    want = O.cgfl_loss(lr, y, aux)
    want.backward()
    ld = lg.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    loss = nnf.cgfl_loss(ld, y.to(DEV), aux.to(DEV))
    loss.backward()
    f32 = dtype == torch.float32
    assert abs(float(loss.detach()) - float(want.detach())) < (2e-6 if f32 else 5e-3) * max(1.0, abs(float(want.detach())))
    assert rel_err(ld.grad.float().cpu(), lr.grad) < (1e-5 if f32 else 1e-2)


def test_softmax_focalloss_takes_the_reference_gamma_vector():
    """`softmax_focalloss(y_pred, y_true, gamma=l1)` - the reference's own call (CGFL.py:221) with the per-sample l1 vector of
    MCTransAuxLoss - equals the `aux=` form whose finalize launch derives l1 itself, and the oracle."""
    from oracle import rssformer_cpu as O
    from representationlearning_amd.module.CGFL import softmax_focalloss
    torch.manual_seed(3)
    B, K = 3, 6
    lg = torch.randn(B, K, 10, 11) * 2.0
    y = torch.randint(-1, K, (B, 10, 11))
    aux = torch.randn(B, 7)
    fg = (y > 0) & (y != -1)
    lab = torch.zeros_like(aux)
    lab[:, 0] = (~fg).flatten(1).any(1).float()
    lab[:, 1] = fg.flatten(1).any(1).float()
    l1 = (1.0 / (1.0 + torch.exp((aux - lab).abs()))).sum(1) / (2 * B)
    want = O.cgfl_loss(lg.clone(), y, aux)
    a = lg.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    b = lg.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    la = softmax_focalloss(a, y.to(DEV), gamma=l1.to(DEV))
    lb = softmax_focalloss(b, y.to(DEV), aux=aux.to(DEV))
    la.backward(); lb.backward()
    assert abs(float(la) - float(want)) < 2e-6 * max(1.0, abs(float(want))) and abs(float(la) - float(lb)) < 1e-6
    assert rel_err(a.grad.float().cpu(), b.grad.float().cpu()) < 1e-6
    with pytest.raises(TypeError):
        softmax_focalloss(a, y.to(DEV), gamma=2)


def test_cgfl_loss_full_size_properties():
    """BASELINE config-2 size (16 x 6 x 512 x 512): finite, gradient sums to zero over classes, zero on ignored pixels."""
    from representationlearning_amd import nnf
    torch.manual_seed(0)
    lg = torch.randn(16, 6, 512, 512, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    y = torch.randint(-1, 6, (16, 512, 512), device=DEV)
    aux = torch.randn(16, 7, device=DEV)
    loss = nnf.cgfl_loss(lg, y, aux)
    loss.backward()
    assert torch.isfinite(loss) and float(loss) > 0
    gsum = lg.grad.float().sum(1)
    assert float(gsum.abs().max()) < 1e-6
    assert float(lg.grad.float().abs().sum(1)[y == -1].max()) == 0.0


def test_lane_reductions_without_lds():
    """The DPP / v_permlane16_swap / v_permlane32_swap reductions of common.hip.h (softmax row max / sum of the window attention
    without ds_bpermute) - lane semantics MEASURED, like the transpose read above."""
    from representationlearning_amd import _lib as L
    torch.manual_seed(3)
    v = torch.randn(64, device=DEV)
    out = torch.empty(10, 64, device=DEV)
    L.check(L.load().rssf_debug_lane_reduce(L.ptr(v), L.ptr(out), L.stream()), "rssf_debug_lane_reduce")
    v, out = v.cpu().double(), out.cpu().double()
    lanes = torch.arange(64)
    assert torch.allclose(out[0], v + v[lanes ^ 16], atol=1e-6), (out[0], v + v[lanes ^ 16])
    assert torch.allclose(out[1], v + v[lanes ^ 32], atol=1e-6)
    rows = v.view(4, 16)
    assert torch.allclose(out[2], rows.sum(0).repeat(4), atol=1e-6)
    assert torch.equal(out[3], rows.max(0).values.repeat(4))
    assert torch.allclose(out[4], v.sum().expand(64), atol=1e-5)
    assert torch.equal(out[5], v.max().expand(64))
    # raw semantics of the two swaps on a = 1000 + lane, b = 2000 + lane (what common.hip.h documents)
    a, b = 1000 + lanes, 2000 + lanes
    ar, br = a.view(4, 16), b.view(4, 16)
    assert out[6].long().tolist() == torch.cat([ar[0], br[0], ar[2], br[2]]).tolist()
    assert out[7].long().tolist() == torch.cat([ar[1], br[1], ar[3], br[3]]).tolist()
    assert out[8].long().tolist() == torch.cat([a[:32], b[:32]]).tolist()
    assert out[9].long().tolist() == torch.cat([a[32:], b[32:]]).tolist()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [1, 2])
def test_conv_bn_act_post_relu_and_add(dtype, act):
    """relu(act(bn(conv(x))) + res_post) as ONE BatchNorm pass (RSSF_ACT_POST_RELU: HighResolutionModule's `relu(transformer(..))`,
    _hrnet_rssformer.py:435, inside MlpDWBN's last layer) and its backward (rssf_bn_bwd_reduce_post / _apply_post: dy masked where
    the output is <= 0, d(res_post) handed back), and rssf_add, against plain torch on the same values."""
    from representationlearning_amd import nnf
    conv, bn = _mk_conv(64, 32, 1, bias=True), _mk_bn(32)
    conv_r, bn_r = _mk_conv(64, 32, 1, bias=True), _mk_bn(32)
    conv, bn = conv.to(DEV), bn.to(DEV).train()
    bn_r.train()
    torch.manual_seed(3)
    x = torch.randn(2, 64, 12, 10).to(dtype).float()
    r = torch.randn(2, 32, 12, 10).to(dtype).float()
    w = torch.randn(2, 32, 12, 10)
    xr, rr = x.clone().requires_grad_(), r.clone().requires_grad_()
    z = bn_r(conv_r(xr))
    yr = F.relu((F.relu(z) if act == 1 else F.gelu(z)) + rr)
    (yr * w).sum().backward()
    xg = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    rg = r.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    yg = nnf.conv_bn_act(xg, conv, bn, act, res_post=rg, post_relu=True)
    (yg.float() * w.to(DEV)).sum().backward()
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel_err(yg.detach().float().cpu(), yr.detach()) < tol
    assert float((yg.detach() < 0).sum()) == 0
    assert rel_err(xg.grad.float().cpu(), xr.grad) < 3 * tol
    assert rel_err(rg.grad.float().cpu(), rr.grad) < tol
    assert rel_err(conv.weight.grad.cpu(), conv_r.weight.grad) < 3 * tol
    assert rel_err(bn.weight.grad.cpu(), bn_r.weight.grad) < 3 * tol and rel_err(bn.bias.grad.cpu(), bn_r.bias.grad) < 3 * tol
    a, b = xg.detach(), torch.randn_like(xg)
    s = nnf.add(a, b)
    assert torch.equal(s, (a.float() + b.float()).to(dtype)) and s.stride() == a.stride()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_batched_pack_equals_single_pack(dtype):
    """rssf_conv_pack_batch (one launch for every convolution, a block per (co, ci) tile with all taps) writes exactly the images
    rssf_conv_pack writes convolution by convolution: 3x3, 1x1, strided, the 3-channel stem and the 6-class head (pad rows / columns),
    wide layers, and MlpDWBN's 17-tap sum with its summed centre taps - forward and transposed layouts."""
    from representationlearning_amd import nnf
    torch.manual_seed(11)
    layers = [[nn.Conv2d(32, 32, 3, 1, 1, bias=False)], [nn.Conv2d(3, 64, 3, 2, 1, bias=False)], [nn.Conv2d(480, 6, 1)], [nn.Conv2d(64, 128, 3, 2, 1, bias=False)],
              [nn.Conv2d(256, 256, 3, 1, 1, bias=False)], [nn.Conv2d(18, 36, 3, 1, 1, bias=False)], [nn.Conv2d(32, 128, 1)], [nn.Conv2d(480, 480, 1, bias=False)],
              [nn.Conv2d(128, 128, 1, 1), nn.Conv2d(128, 128, 3, 1, padding=6, dilation=6), nn.Conv2d(128, 128, 3, 1, padding=12, dilation=12)],
              [nn.Conv2d(72, 72, 1, 1), nn.Conv2d(72, 72, 3, 1, padding=6, dilation=6), nn.Conv2d(72, 72, 3, 1, padding=12, dilation=12)]]
    layers = [[c.to(DEV) for c in cs] for cs in layers]
    plan = nnf.PackPlan()
    plan.recording = True
    keys = []
    for cs in layers:
        spec = nnf.spec_of(cs)
        for tr in (False, True):
            key = (id(spec), tr, dtype)
            plan.record(key, spec, [c.weight.detach() for c in cs], tr, dtype)
            keys.append((key, spec, cs, tr))
    plan.recording = False
    plan.build()
    plan.refresh()
    torch.cuda.synchronize()
    for key, spec, cs, tr in keys:
        single = nnf._pack(spec, [c.weight.detach() for c in cs], tr, dtype, torch.device(DEV), rt=nnf.Runtime())
        got = plan.lookup(key)
        assert got is not None and got.shape == single.shape
        assert torch.equal(got, single), (tuple(cs[0].weight.shape), len(cs), tr)
