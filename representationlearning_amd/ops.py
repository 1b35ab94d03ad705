"""Thin Python wrappers over the librssf C ABI (one function per entry point, no arithmetic here).

All activations are channels-last token tensors [B, N, C] (== NHWC) in fp32 or bf16 on the GPU;
parameters / statistics / parameter gradients are fp32.  Autograd glue lives in `autograd.py`.
"""
import ctypes

import torch

from . import _lib as L

LN_EPS = 1e-6


def _f32(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), "parameters/statistics must be contiguous fp32"
    return t


def _tok(t):
    if not t.is_contiguous():
        raise RuntimeError("librssf expects contiguous channels-last tokens [B,N,C]")
    return t


def _same(*ts):
    """All activation tensors of one call must share dtype/shape/device (the C ABI carries ONE dtype code)."""
    a = ts[0]
    for t in ts[1:]:
        if t is not None and (t.dtype != a.dtype or t.shape != a.shape or t.device != a.device):
            raise RuntimeError(f"librssf: mismatched activation tensors: {a.dtype}{tuple(a.shape)} vs {t.dtype}{tuple(t.shape)}")


def layernorm_fwd(x, gamma, beta, want_y=True, eps=LN_EPS):
    """x [..., C] -> (y or None, stats [rows, 2] = {mean, rstd})."""
    L.require_gpu(x)
    _tok(x)
    C = x.shape[-1]
    rows = x.numel() // C
    stats = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
    y = torch.empty_like(x) if want_y else None
    L.check(L.load().rssf_layernorm_fwd(L.ptr(x), L.ptr(_f32(gamma)), L.ptr(_f32(beta)), L.ptr(y), L.ptr(stats),
                                        rows, C, eps, L.dtype_code(x), L.stream()), "rssf_layernorm_fwd")
    return y, stats


def layernorm_bwd(dy, x, stats, gamma, dgamma, dbeta, dx_add=None):
    """Returns dx (+ dx_add); accumulates into dgamma / dbeta."""
    L.require_gpu(dy, x)
    _tok(dy); _tok(x); _same(x, dy, dx_add)
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    if dx_add is not None:
        _tok(dx_add)
    L.check(L.load().rssf_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(_f32(gamma)), L.ptr(dx_add), L.ptr(dx),
                                        L.ptr(_f32(dgamma)), L.ptr(_f32(dbeta)), rows, C, L.dtype_code(x), L.stream()),
            "rssf_layernorm_bwd")
    return dx


def gate_pool_fwd(x, y, stats_x, stats_y, gamma, beta):
    _same(x, y)
    B, N, C = x.shape
    pooled = torch.empty(B, 4, N, device=x.device, dtype=torch.float32)
    argmax = torch.empty(B, 2, N, device=x.device, dtype=torch.int32)
    L.check(L.load().rssf_gate_pool_fwd(L.ptr(_tok(x)), L.ptr(_tok(y)), L.ptr(stats_x), L.ptr(stats_y), L.ptr(_f32(gamma)),
                                        L.ptr(_f32(beta)), L.ptr(pooled), L.ptr(argmax), B, N, C, L.dtype_code(x),
                                        L.stream()), "rssf_gate_pool_fwd")
    return pooled, argmax


def ln_gate_pool_fwd(x, y, gamma, beta, eps=LN_EPS):
    """(stats_x, stats_y, pooled, argmax): the LayerNorm statistics of both token streams and the gate's pooled maps in ONE pass over x
    and y (rssf_ln_gate_pool_fwd), or None where the fused launch does not take the shape (the caller then runs the three launches)."""
    _same(x, y)
    B, N, C = x.shape
    lib = L.load()
    if lib.rssf_ln_gate_pool_fwd_supported(B, N, C, L.dtype_code(x)) != 1:
        return None
    sx = torch.empty(B * N, 2, device=x.device, dtype=torch.float32)
    sy = torch.empty_like(sx)
    pooled = torch.empty(B, 4, N, device=x.device, dtype=torch.float32)
    argmax = torch.empty(B, 2, N, device=x.device, dtype=torch.int32)
    L.check(lib.rssf_ln_gate_pool_fwd(L.ptr(_tok(x)), L.ptr(_tok(y)), L.ptr(_f32(gamma)), L.ptr(_f32(beta)), eps, L.ptr(sx), L.ptr(sy), L.ptr(pooled),
                                      L.ptr(argmax), B, N, C, L.dtype_code(x), L.stream()), "rssf_ln_gate_pool_fwd")
    return sx, sy, pooled, argmax


def gate_weights_fwd(pooled, k, wl, bl, H, W, want_logits=False):
    """k: [2,2,7,7] (stream, {mean,max}, 7, 7); wl [2,2]; bl [2] -> gsig, omega (, logits) each [B,2,N]."""
    B = pooled.shape[0]
    N = H * W
    gsig = torch.empty(B, 2, N, device=pooled.device, dtype=torch.float32)
    omega = torch.empty_like(gsig)
    logits = torch.empty_like(gsig) if want_logits else None
    L.check(L.load().rssf_gate_weights_fwd(L.ptr(pooled), L.ptr(_f32(k)), L.ptr(_f32(wl)), L.ptr(_f32(bl)), L.ptr(gsig),
                                           L.ptr(omega), L.ptr(logits), B, H, W, L.stream()), "rssf_gate_weights_fwd")
    return gsig, omega, logits


def gate_weights_bwd(domega, pooled, gsig, omega, k, wl, dk, dwl, dbl, H, W, dk_stream1=None):
    """Returns dpooled [B,4,N]; accumulates dk [2,2,7,7], dwl [2,2], dbl [2] - or, with dk_stream1 (98 values), dk [2,7,7] takes
    the first 7x7 kernel's gradient and dk_stream1 the second's (the two .grad buffers of the module's two parameters)."""
    B = pooled.shape[0]
    N = H * W
    buf = torch.empty(B * 6 * N + 16 * 202, device=pooled.device, dtype=torch.float32)   # [B][4][N] dpooled + scratch (rssf.h)
    L.check(L.load().rssf_gate_weights_bwd(L.ptr(domega), L.ptr(pooled), L.ptr(gsig), L.ptr(omega), L.ptr(_f32(k)),
                                           L.ptr(_f32(wl)), L.ptr(buf), L.ptr(_f32(dk)), L.ptr(None if dk_stream1 is None else _f32(dk_stream1)), L.ptr(_f32(dwl)), L.ptr(_f32(dbl)),
                                           B, H, W, L.stream()), "rssf_gate_weights_bwd")
    return buf[: B * 4 * N].view(B, 4, N)


def gate_pool_bwd_(dpooled, argmax, dxhat, dyhat):
    _same(dxhat, dyhat)
    B, N, C = dxhat.shape
    L.check(L.load().rssf_gate_pool_bwd(L.ptr(dpooled), L.ptr(argmax), L.ptr(_tok(dxhat)), L.ptr(_tok(dyhat)), B, N, C,
                                        L.dtype_code(dxhat), L.stream()), "rssf_gate_pool_bwd")


def gate_pool_ln_bwd(dpooled, argmax, dxhat, dyhat, x, y, stats_x, stats_y, gamma, dgamma, dbeta, dx_add=None):
    """(dx, dy): the gate-path gradient merged into d(LN1 output) and the LayerNorm backward of both token streams in ONE pass
    (rssf_gate_pool_ln_bwd; dgamma / dbeta accumulate), or None where the fused launch does not take the shape."""
    _same(dxhat, dyhat); _same(x, y); _same(x, dxhat, dx_add)
    B, N, C = x.shape
    lib = L.load()
    if lib.rssf_gate_pool_ln_bwd_supported(B, N, C, L.dtype_code(x)) != 1:
        return None
    dx, dy = torch.empty_like(x), torch.empty_like(y)
    if dx_add is not None:
        _tok(dx_add)
    L.check(lib.rssf_gate_pool_ln_bwd(L.ptr(dpooled), L.ptr(argmax), L.ptr(_tok(dxhat)), L.ptr(_tok(dyhat)), L.ptr(_tok(x)), L.ptr(_tok(y)),
                                      L.ptr(stats_x), L.ptr(stats_y), L.ptr(_f32(gamma)), L.ptr(dx_add), L.ptr(dx), L.ptr(dy), L.ptr(_f32(dgamma)),
                                      L.ptr(_f32(dbeta)), B, N, C, L.dtype_code(x), L.stream()), "rssf_gate_pool_ln_bwd")
    return dx, dy


def _winattn_params(x, y, stats_x, stats_y, omega, ln_g, ln_b, w, H, W, heads, out):
    B, N, C = x.shape
    p = L.WinAttnFwdParams()
    p.x, p.y = x.data_ptr(), y.data_ptr()
    p.stats_x, p.stats_y, p.omega = stats_x.data_ptr(), stats_y.data_ptr(), omega.data_ptr()
    p.ln_gamma, p.ln_beta = _f32(ln_g).data_ptr(), _f32(ln_b).data_ptr()
    for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo"):
        setattr(p, n, _f32(w[n]).data_ptr())
    p.out = 0 if out is None else out.data_ptr()
    p.B, p.H, p.W, p.C, p.heads, p.window, p.dtype = B, H, W, C, heads, 7, L.dtype_code(x)
    return p


def winattn_fwd(x, y, stats_x, stats_y, omega, ln_g, ln_b, w, H, W, heads=2):
    """out = x + WindowCrossAttention(LN1(x)*omega0, LN1(y)*omega1).  w: dict wq,bq,wk,bk,wv,bv,wo,bo (fp32)."""
    L.require_gpu(x, y)
    _tok(x); _tok(y); _same(x, y)
    out = torch.empty_like(x)
    p = _winattn_params(x, y, stats_x, stats_y, omega, ln_g, ln_b, w, H, W, heads, out)
    L.check(L.load().rssf_winattn_fwd(ctypes.byref(p), L.stream()), "rssf_winattn_fwd")
    return out


def winattn_bwd(dout, x, y, stats_x, stats_y, omega, ln_g, ln_b, w, gw, H, W, heads=2, workspace=True):
    """Backward of the attention term.  gw: dict of fp32 grads (dwq..dbo) accumulated in place.
    Returns dxhat, dyhat (grad w.r.t. LN1 outputs through the attention path) and domega [B,2,N].
    workspace=False takes the ABI's no-scratch route (domega accumulated with atomics into a zeroed buffer)."""
    L.require_gpu(dout, x, y)
    _tok(dout); _same(x, y, dout)
    B, N, C = x.shape
    dxhat = torch.empty_like(x)
    dyhat = torch.empty_like(y)
    if workspace:
        domega = torch.empty(B, 2, N, device=x.device, dtype=torch.float32)
        prod_ws = torch.empty(L.load().rssf_winattn_bwd_workspace_elems(B, H, W, C), device=x.device, dtype=x.dtype)   # gate-gradient products (rssf.h)
    else:
        domega = torch.zeros(B, 2, N, device=x.device, dtype=torch.float32)
        prod_ws = None
    bp = L.WinAttnBwdParams()
    bp.f = _winattn_params(x, y, stats_x, stats_y, omega, ln_g, ln_b, w, H, W, heads, None)
    bp.dout, bp.dxhat, bp.dyhat, bp.domega = dout.data_ptr(), dxhat.data_ptr(), dyhat.data_ptr(), domega.data_ptr()
    bp.prod_ws = prod_ws.data_ptr() if workspace else None
    for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo"):
        setattr(bp, "d" + n, _f32(gw[n]).data_ptr())
    L.check(L.load().rssf_winattn_bwd(ctypes.byref(bp), L.stream()), "rssf_winattn_bwd")
    return dxhat, dyhat, domega


def argmax_confusion(scores, labels=None, cm=None, ignore_index=-1, want_pred=True):
    """scores: logical [B,K,H,W] channels-last (or [npix,K]) class scores; labels [B,H,W] int64.  Returns pred [B,H,W] int32
    (if want_pred) and accumulates cm [K,K] int64 += confusion of the non-ignored pixels (eval.py:66-71 of the reference)."""
    L.require_gpu(scores)
    if scores.dim() == 4:
        sh = scores.permute(0, 2, 3, 1)
        sh = sh if sh.is_contiguous() else sh.contiguous()
        out_shape = sh.shape[:3]
    else:
        sh, out_shape = scores.contiguous(), scores.shape[:1]
    K = sh.shape[-1]
    npix = sh.numel() // K
    pred = torch.empty(out_shape, device=scores.device, dtype=torch.int32) if want_pred else None
    if labels is not None:
        labels = labels.to(device=scores.device, dtype=torch.int64).contiguous()
        if labels.numel() != npix or cm is None or cm.dtype != torch.int64 or tuple(cm.shape) != (K, K) or not cm.is_cuda:
            raise ValueError("argmax_confusion: labels must match the pixels and cm must be a [K,K] int64 device tensor")
    L.check(L.load().rssf_argmax_confusion(L.ptr(sh), L.ptr(labels), L.ptr(pred), L.ptr(cm), npix, K, ignore_index, L.dtype_code(sh),
                                           L.stream()), "rssf_argmax_confusion")
    return pred


def zero_(t):
    """t.zero_() for a contiguous fp32 device tensor as a librssf kernel launch (never a memset node: see rssf_zero_f32)."""
    if t.numel() == 0:
        return t
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        return t.zero_()
    L.check(L.load().rssf_zero_f32(L.ptr(t), t.numel(), L.stream()), "rssf_zero_f32")
    return t


SQNORM_ELEMS = 1 + 2048          # 1 + RSSF_SQNORM_BLOCKS


def grad_sqnorm(flat_grad, out):
    """out[0] = ||flat_grad||^2 (fp32, on the current stream); out: fp32 [SQNORM_ELEMS] (result + per-block partials)."""
    L.require_gpu(flat_grad)
    if out.numel() < SQNORM_ELEMS:
        raise ValueError("grad_sqnorm: `out` must hold %d floats (include/rssf.h)" % SQNORM_ELEMS)
    L.check(L.load().rssf_grad_sqnorm(L.ptr(_f32(flat_grad)), flat_grad.numel(), L.ptr(out), L.stream()), "rssf_grad_sqnorm")
    return out


def sgd_step_(flat_p, flat_g, flat_m, sqnorm, grad_scale, max_norm, lr, momentum, weight_decay, first_step, lr_dev=None):
    """lr_dev: optional 1-element fp32 device tensor holding the LR (read at execution time: hipGraph-replay safe)."""
    L.require_gpu(flat_p, flat_g, flat_m)
    L.check(L.load().rssf_sgd_step(L.ptr(_f32(flat_p)), L.ptr(_f32(flat_g)), L.ptr(_f32(flat_m)), flat_p.numel(), L.ptr(sqnorm),
                                   float(grad_scale), float(max_norm), L.ptr(lr_dev), float(lr), float(momentum),
                                   float(weight_decay), int(bool(first_step)), L.stream()), "rssf_sgd_step")


def debug_mma(a, b):
    K = a.shape[1]
    d = torch.empty(3, 16, 16, device=a.device, dtype=torch.float32)
    L.check(L.load().rssf_debug_mma(L.ptr(a.contiguous()), L.ptr(b.contiguous()), L.ptr(d), K, L.dtype_code(a), L.stream()),
            "rssf_debug_mma")
    return d


# ---- Mix-Transformer / CAM inference operators (csrc/mit.hip; reference SCD-AAAI2023/network/mix_transformer.py, utils/camutils.py) ----
def mha_fwd(q, kv, heads, scale, want_logits=False):
    """q [B, N, C], kv [B, M, 2C] (k | v, heads-major) -> (softmax(q k^T * scale) v [B, N, C], raw q k^T logits [B, heads, N, M] fp32 or
    None).  Attention.forward of the reference (mix_transformer.py:93-131), no dropout."""
    L.require_gpu(q, kv)
    _tok(q); _tok(kv)
    B, N, C = q.shape
    M = kv.shape[1]
    if kv.shape != (B, M, 2 * C) or kv.dtype != q.dtype or C % heads:
        raise RuntimeError(f"mha_fwd: q {tuple(q.shape)} {q.dtype} / kv {tuple(kv.shape)} {kv.dtype} / heads {heads}")
    out = torch.empty_like(q)
    logits = torch.empty(B, heads, N, M, device=q.device, dtype=torch.float32) if want_logits else None
    L.check(L.load().rssf_mha_fwd(L.ptr(q), L.ptr(kv), L.ptr(out), L.ptr(logits), B, N, M, heads, C // heads, float(scale),
                                  L.dtype_code(q), L.stream()), "rssf_mha_fwd")
    return out, logits


def dwconv3x3(x, weight, bias, act=0):
    """x [B, H, W, C] channels-last -> act(depthwise 3x3 (padding 1) + bias); weight [C, 1, 3, 3]; act 0 none / 2 GELU."""
    L.require_gpu(x)
    _tok(x)
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    L.check(L.load().rssf_dwconv3x3(L.ptr(x), L.ptr(_f32(weight)), L.ptr(None if bias is None else _f32(bias)), L.ptr(y), B, H, W, C, int(act),
                                    L.dtype_code(x), L.stream()), "rssf_dwconv3x3")
    return y


def attn_proj_sigmoid(a0, a1, weight, bias):
    """sigmoid(Conv2d(2 * heads, 1, 1)(cat([a0, a1], 1)))[:, 0] for two logit tensors [B, heads, N, M] (TSCD_model.py:73-75)."""
    L.require_gpu(a0, a1)
    B, heads = a0.shape[:2]
    if a0.shape != a1.shape or a0.dtype != torch.float32 or a1.dtype != torch.float32 or weight.numel() != 2 * heads:
        raise RuntimeError("attn_proj_sigmoid: two fp32 [B, heads, N, M] tensors and a [1, 2 * heads, 1, 1] weight expected")
    a0, a1 = a0.contiguous(), a1.contiguous()
    out = torch.empty((B,) + tuple(a0.shape[2:]), device=a0.device, dtype=torch.float32)
    L.check(L.load().rssf_attn_proj_sigmoid(L.ptr(a0), L.ptr(a1), L.ptr(_f32(weight).reshape(-1)), L.ptr(None if bias is None else _f32(bias)),
                                            L.ptr(out), B, heads, out[0].numel(), L.stream()), "rssf_attn_proj_sigmoid")
    return out


def attn_pred(q0, kv0, q1, kv1, weight, bias, heads):
    """sigmoid(Conv2d(2 * heads, 1, 1)(cat([q0 k0^T, q1 k1^T], 1)))[:, 0] from the projections themselves (q_s [B, N, C], kv_s
    [B, M, 2C]): the attention prediction of TSCD.forward (TSCD_model.py:73-75) without materialising the logits."""
    L.require_gpu(q0, kv0, q1, kv1)
    for t in (q0, kv0, q1, kv1):
        _tok(t)
    B, N, C = q0.shape
    M = kv0.shape[1]
    if q1.shape != q0.shape or kv0.shape != (B, M, 2 * C) or kv1.shape != kv0.shape or weight.numel() != 2 * heads or C % heads \
            or len({q0.dtype, q1.dtype, kv0.dtype, kv1.dtype}) != 1:
        raise RuntimeError("attn_pred: q [B, N, C] / kv [B, M, 2C] of two blocks in one dtype and a [1, 2 * heads, 1, 1] weight expected")
    out = torch.empty(B, N, M, device=q0.device, dtype=torch.float32)
    L.check(L.load().rssf_attn_pred(L.ptr(q0), L.ptr(kv0), L.ptr(q1), L.ptr(kv1), L.ptr(_f32(weight).reshape(-1)),
                                    L.ptr(None if bias is None else _f32(bias)), L.ptr(out), B, N, M, heads, C // heads, L.dtype_code(q0), L.stream()),
            "rssf_attn_pred")
    return out


def resize_bilinear_planar(x, size):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=False) for a contiguous NCHW tensor (every plane on its own)."""
    L.require_gpu(x)
    x = x.contiguous()
    B, C, H, W = x.shape
    out = torch.empty(B, C, size[0], size[1], device=x.device, dtype=x.dtype)
    L.check(L.load().rssf_resize_bilinear(L.ptr(x), L.ptr(out), B * C, H, W, size[0], size[1], 1, L.dtype_code(x), L.stream()), "rssf_resize_bilinear")
    return out


def resize_bilinear_nhwc(xh, size):
    """F.interpolate(size=size, mode='bilinear', align_corners=False) of a channels-last activation [B, H, W, C]."""
    L.require_gpu(xh)
    _tok(xh)
    B, H, W, C = xh.shape
    out = torch.empty(B, size[0], size[1], C, device=xh.device, dtype=xh.dtype)
    L.check(L.load().rssf_resize_bilinear(L.ptr(xh), L.ptr(out), B, H, W, size[0], size[1], C, L.dtype_code(xh), L.stream()), "rssf_resize_bilinear")
    return out


def cam_merge_(acc, cam, accumulate):
    """acc [b, K, H, W] fp32 (+)= relu(max(up(cam[:b]), up(cam[b:]).flip(-1))), cam [2b, hc, wc, K] channels-last (camutils.py:93-96)."""
    L.require_gpu(acc, cam)
    _tok(cam)
    b, K, H, W = acc.shape
    if cam.shape[0] != 2 * b or cam.shape[3] != K or acc.dtype != torch.float32 or not acc.is_contiguous():
        raise RuntimeError(f"cam_merge: acc {tuple(acc.shape)} {acc.dtype} / cam {tuple(cam.shape)}")
    L.check(L.load().rssf_cam_merge(L.ptr(cam), L.ptr(acc), b, K, cam.shape[1], cam.shape[2], H, W, int(bool(accumulate)), L.dtype_code(cam),
                                    L.stream()), "rssf_cam_merge")
    return acc


def cam_normalize_(acc):
    """per (image, class) plane: (x - min) / (max - min + 1e-5), in place (camutils.py:111-112)."""
    L.require_gpu(acc)
    if acc.dtype != torch.float32 or not acc.is_contiguous():
        raise RuntimeError("cam_normalize: contiguous fp32 tensor expected")
    b, K, H, W = acc.shape
    L.check(L.load().rssf_cam_normalize(L.ptr(acc), b * K, H * W, L.stream()), "rssf_cam_normalize")
    return acc
