"""torch.autograd.Function shims: each forward/backward is a short sequence of librssf C-ABI calls.

No arithmetic happens in Python here beyond allocating outputs and zeroing gradient accumulators; torch is
the allocator / stream / autograd-graph plumbing.
"""
import torch

import os

from . import nnf, ops

_FUSED_LN_POOL = os.environ.get("RSSF_FUSED_LN_POOL", "1") != "0"      # A/B switch (tools/round.sh ab): the three-launch form


def _gate_kernels(k1, k2):
    """[2 (stream), 2 (mean, max), 7, 7] view of the two SpatialAttention kernels (multihead_isa_pool_attention.py:30-31).  Under the
    trainer both parameters are neighbours in the flat fp32 buffer: a strided view, no torch.stack copy per block and step."""
    a, b = k1[0], k2[0]
    if (a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and b.data_ptr() == a.data_ptr() + a.numel() * 4):
        return torch.as_strided(a.detach(), (2,) + tuple(a.shape), (a.numel(),) + tuple(a.stride()), a.storage_offset())
    return torch.stack([a, b]).contiguous()


class GatedWindowCrossAttention(torch.autograd.Function):
    """out = x + Attn(LN1(x), LN1(y)) for one GeneralTransformerBlock (modules/MTFM.py:107), i.e.
    norm1 on both streams -> InterlacedPoolAttention2 (saliency gate, 7x7 windows, Mhca) -> residual.

    x, y: channels-last tokens [B, N, C].  Saved for backward: x, y and the small fp32 maps only
    (LN stats, pooled maps, gate sigmoids, omega); everything else is recomputed in the backward kernel.
    """

    @staticmethod
    def forward(ctx, x, y, ln_g, ln_b, k1, k2, wl, bl, wq, bq, wk, bk, wv, bv, wo, bo, H, W, heads):
        x = x.contiguous()
        y = y.contiguous()
        fused = ops.ln_gate_pool_fwd(x, y, ln_g, ln_b) if _FUSED_LN_POOL else None      # norm1's statistics of both streams + the gate's pooling: one pass
        if fused is not None:
            sx, sy, pooled, argmax = fused
        else:
            _, sx = ops.layernorm_fwd(x, ln_g, ln_b, want_y=False)
            _, sy = ops.layernorm_fwd(y, ln_g, ln_b, want_y=False)
            pooled, argmax = ops.gate_pool_fwd(x, y, sx, sy, ln_g, ln_b)
        kk = _gate_kernels(k1, k2)                                   # [2(stream), 2(mean,max), 7, 7]
        wl2 = wl.reshape(2, 2).contiguous()
        gsig, omega, _ = ops.gate_weights_fwd(pooled, kk, wl2, bl, H, W)
        w = dict(wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, wo=wo, bo=bo)
        out = ops.winattn_fwd(x, y, sx, sy, omega, ln_g, ln_b, w, H, W, heads)
        ctx.save_for_backward(x, y, sx, sy, pooled, argmax, gsig, omega, kk, wl2, ln_g, ln_b, wq, bq, wk, bk, wv, bv, wo, bo)
        ctx.dims = (H, W, heads)
        ctx.params = dict(ln_g=ln_g, ln_b=ln_b, wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, wo=wo, bo=bo)
        ctx.gate_params = (k1, k2, wl, bl)
        ctx.rt = nnf.current()
        return out

    @staticmethod
    def backward(ctx, dout):
        (x, y, sx, sy, pooled, argmax, gsig, omega, kk, wl2, ln_g, ln_b, wq, bq, wk, bk, wv, bv, wo, bo) = ctx.saved_tensors
        H, W, heads = ctx.dims
        dout = dout.contiguous()
        w = dict(wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, wo=wo, bo=bo)
        P, rt = ctx.params, ctx.rt
        tg = {n: nnf.grad_target(P[n], rt) for n in P}            # kernels accumulate straight into .grad when possible
        gw = {n: tg[n][0] for n in w}
        dxhat, dyhat, domega = ops.winattn_bwd(dout, x, y, sx, sy, omega, ln_g, ln_b, w, gw, H, W, heads)
        # the four gate parameters (two 7x7 kernels, the 1x1 mixing weight and bias): straight into their .grad buffers when the
        # trainer owns them (the fold launch of rssf_gate_weights_bwd accumulates) - no staging slice, copy or AccumulateGrad add
        gate_direct = rt.direct and all(p.requires_grad and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous()
                                        for p in ctx.gate_params)
        nk = kk.numel()
        if gate_direct:
            gt = [nnf.grad_target(p, rt) for p in ctx.gate_params]
            dpooled = ops.gate_weights_bwd(domega, pooled, gsig, omega, kk, wl2, gt[0][0].view(-1), gt[2][0].view(-1), gt[3][0], H, W,
                                           dk_stream1=gt[1][0].view(-1))
            ggate = tuple(nnf.grad_result(p, t, d, rt) for p, (t, d) in zip(ctx.gate_params, gt))
        else:
            zb = nnf._zeros(nk + 8, x.device, rt)            # one slice of the step's pre-zeroed pool instead of three fills
            dk, dwl, dbl = zb[:nk].view_as(kk), zb[nk:nk + 4].view(2, 2), zb[nk + 4:nk + 6]
            dpooled = ops.gate_weights_bwd(domega, pooled, gsig, omega, kk, wl2, dk, dwl, dbl, H, W)
            gp = zb[:nk + 6].clone()                           # the pool slice is recycled next step: hand autograd a copy
            h = nk // 2
            ggate = (gp[:h].view_as(kk[0:1]), gp[h:nk].view_as(kk[1:2]), gp[nk:nk + 4].view(2, 2, 1, 1), gp[nk + 4:nk + 6])
        dg, db = tg["ln_g"][0], tg["ln_b"][0]
        # gate-path gradient merged into d(LN1 output) + norm1's backward on both streams (+ the residual path on x): one pass
        fused = ops.gate_pool_ln_bwd(dpooled, argmax, dxhat, dyhat, x, y, sx, sy, ln_g, dg, db, dx_add=dout) if _FUSED_LN_POOL else None
        if fused is not None:
            dx, dy = fused
        else:
            ops.gate_pool_bwd_(dpooled, argmax, dxhat, dyhat)
            dx = ops.layernorm_bwd(dxhat, x, sx, ln_g, dg, db, dx_add=dout)      # + residual path
            dy = ops.layernorm_bwd(dyhat, y, sy, ln_g, dg, db)
        r = {n: nnf.grad_result(P[n], tg[n][0], tg[n][1], rt) for n in P}
        return (dx, dy, r["ln_g"], r["ln_b"], *ggate,
                r["wq"], r["bq"], r["wk"], r["bk"], r["wv"], r["bv"], r["wo"], r["bo"], None, None, None)


class LayerNormTokens(torch.autograd.Function):
    """nn.LayerNorm(C, eps=1e-6) over channels-last tokens (modules/MTFM.py:81, norm2)."""

    @staticmethod
    def forward(ctx, x, g, b):
        x = x.contiguous()
        y, st = ops.layernorm_fwd(x, g, b)
        ctx.save_for_backward(x, st, g)
        ctx.params = (g, b)
        ctx.rt = nnf.current()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st, g = ctx.saved_tensors
        pg, pb = ctx.params
        rt = ctx.rt
        (dg, gd), (db, bd) = nnf.grad_target(pg, rt), nnf.grad_target(pb, rt)
        dx = ops.layernorm_bwd(dy.contiguous(), x, st, g, dg, db)
        return dx, nnf.grad_result(pg, dg, gd, rt), nnf.grad_result(pb, db, bd, rt)


class LayerNormTokensRes(torch.autograd.Function):
    """(LN(x), x): norm2 of GeneralTransformerBlock together with the skip that goes around the MLP (MTFM.py:109: x + mlp(norm2(x))).
    The gradient of x is `ln_bwd(dz) + dskip`; as two autograd edges the sum was a separate element-wise pass over the token
    tensor per block, here it is the addend of the LayerNorm-backward launch."""

    @staticmethod
    def forward(ctx, x, g, b):
        x = x.contiguous()
        y, st = ops.layernorm_fwd(x, g, b)
        ctx.save_for_backward(x, st, g)
        ctx.params = (g, b)
        ctx.rt = nnf.current()
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, st, g = ctx.saved_tensors
        pg, pb = ctx.params
        rt = ctx.rt
        if dy is None:
            return dskip, None, None
        (dg, gd), (db, bd) = nnf.grad_target(pg, rt), nnf.grad_target(pb, rt)
        add = None if dskip is None else dskip.contiguous()
        if add is not None and add.dtype != x.dtype:
            add = add.to(x.dtype)
        dx = ops.layernorm_bwd(dy.contiguous(), x, st, g, dg, db, dx_add=add)
        return dx, nnf.grad_result(pg, dg, gd, rt), nnf.grad_result(pb, db, bd, rt)


def _identity_ln(x):
    B, N, C = x.shape
    st = torch.tensor([0.0, 1.0], device=x.device, dtype=torch.float32).repeat(B * N, 1).contiguous()
    return st, torch.ones(C, device=x.device), torch.zeros(C, device=x.device)


class PrenormedGatedWindowCrossAttention(torch.autograd.Function):
    """InterlacedPoolAttention2.forward on its own (multihead_isa_pool_attention.py:148-188): inputs are the
    already-normalised token streams; returns the attention term only (no residual)."""

    @staticmethod
    def forward(ctx, x, y, k1, k2, wl, bl, wq, bq, wk, bk, wv, bv, wo, bo, H, W, heads):
        x = x.contiguous()
        y = y.contiguous()
        st, one, zero = _identity_ln(x)
        pooled, argmax = ops.gate_pool_fwd(x, y, st, st, one, zero)
        kk = torch.stack([k1[0], k2[0]]).contiguous()
        wl2 = wl.reshape(2, 2).contiguous()
        gsig, omega, _ = ops.gate_weights_fwd(pooled, kk, wl2, bl, H, W)
        w = dict(wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, wo=wo, bo=bo)
        out = ops.winattn_fwd(x, y, st, st, omega, one, zero, w, H, W, heads)
        ctx.save_for_backward(x, y, st, pooled, argmax, gsig, omega, kk, wl2, one, zero, wq, bq, wk, bk, wv, bv, wo, bo)
        ctx.dims = (H, W, heads)
        return out - x

    @staticmethod
    def backward(ctx, dout):
        (x, y, st, pooled, argmax, gsig, omega, kk, wl2, one, zero, wq, bq, wk, bk, wv, bv, wo, bo) = ctx.saved_tensors
        H, W, heads = ctx.dims
        w = dict(wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, wo=wo, bo=bo)
        gw = {n: torch.zeros_like(t) for n, t in w.items()}
        dxhat, dyhat, domega = ops.winattn_bwd(dout.contiguous(), x, y, st, st, omega, one, zero, w, gw, H, W, heads)
        dk = torch.zeros_like(kk)
        dwl = torch.zeros_like(wl2)
        dbl = torch.zeros(2, device=x.device, dtype=torch.float32)
        dpooled = ops.gate_weights_bwd(domega, pooled, gsig, omega, kk, wl2, dk, dwl, dbl, H, W)
        ops.gate_pool_bwd_(dpooled, argmax, dxhat, dyhat)
        return (dxhat, dyhat, dk[0:1].clone(), dk[1:2].clone(), dwl.reshape(2, 2, 1, 1), dbl,
                gw["wq"], gw["bq"], gw["wk"], gw["bk"], gw["wv"], gw["bv"], gw["wo"], gw["bo"], None, None, None)


class PlainWindowCrossAttention(torch.autograd.Function):
    """Mhca on pre-grouped windows (modules/DAL.py:785-1030): x, y are [nWin, 49, C]; each window is treated as
    one 7x7 image, with identity LN and unit gate, and the residual removed."""

    @staticmethod
    def forward(ctx, x, y, wq, bq, wk, bk, wv, bv, wo, bo, H, W, heads):
        st, one, zero = _identity_ln(x)
        omega = torch.ones(x.shape[0], 2, H * W, device=x.device, dtype=torch.float32)
        w = dict(wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, wo=wo, bo=bo)
        out = ops.winattn_fwd(x, y, st, st, omega, one, zero, w, H, W, heads)
        ctx.save_for_backward(x, y, st, omega, one, zero, wq, bq, wk, bk, wv, bv, wo, bo)
        ctx.dims = (H, W, heads)
        return out - x

    @staticmethod
    def backward(ctx, dout):
        (x, y, st, omega, one, zero, wq, bq, wk, bk, wv, bv, wo, bo) = ctx.saved_tensors
        H, W, heads = ctx.dims
        w = dict(wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, wo=wo, bo=bo)
        gw = {n: torch.zeros_like(t) for n, t in w.items()}
        dxhat, dyhat, _ = ops.winattn_bwd(dout.contiguous(), x, y, st, st, omega, one, zero, w, gw, H, W, heads)
        return (dxhat, dyhat, gw["wq"], gw["bq"], gw["wk"], gw["bk"], gw["wv"], gw["bv"], gw["wo"], gw["bo"], None, None, None)
