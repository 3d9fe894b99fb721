"""Differentiable wrappers of the HIP kernels (training step).

torch.autograd is used as the graph/ordering/accumulation engine only: every Function's forward AND backward is
libemip_hip.so kernels on channels-last activations; parameter gradients come out as f32 tensors in the reference's
parameter shapes (so `clip_gradient` + AdamW semantics and DDP-style reducers apply unchanged).  Weight gradients are
TN contractions (`emip_gemm_tn`, `emip_conv2d_wgrad`), input gradients reuse the forward GEMM / conv kernels on
transposed or flipped weight packs, attention backward is the unfused softmax-backward formulation (recompute P,
batched GEMMs) built from the same kernels.
"""
import torch
from torch.autograd import Function

from . import ops


def colsum_f32(dy):
    """bias gradient: sum over rows of a channels-last tensor -> f32 [C]"""
    return ops.colsum(dy)


def unpack_conv_grad(dw_packed, cout, cin, k, cin_pad=None, perm=None):
    """[Cout, k*k*Cin_pad] (ci fastest) -> [Cout, Cin, k, k] in the reference's parameter layout"""
    cp = cin_pad or cin
    g = dw_packed.view(cout, k, k, cp)[..., :cin].permute(0, 3, 1, 2)
    if perm is not None:
        out = torch.empty_like(g)
        out[:, perm] = g
        g = out
    return g.contiguous()


class LinearFn(Function):
    """y = x W^T + b (+ res).  wp: W packed [N,K] in the activation dtype, wpt: W^T packed [K,N]."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, wp, wpt):
        y = ops.gemm(x, wp, bias=bias, res=res)
        ctx.save_for_backward(x, weight)
        ctx.wpt, ctx.has_bias, ctx.has_res = wpt, bias is not None, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.gemm(dy, ctx.wpt) if ctx.needs_input_grad[0] else None
        dw = ops.gemm_tn(dy, x).view_as(weight) if ctx.needs_input_grad[1] else None
        db = colsum_f32(dy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None, None


class ConvFn(Function):
    """NHWC conv.  wp: forward pack; wdg: pack for the input gradient (flipped + transposed, or W^T for patch convs)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wp, wdg, k, s, p, cin_pad):
        y = ops.conv2d(x, wp, k, k, s, p, bias=bias)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (wdg, k, s, p, cin_pad, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        wdg, k, s, p, cin_pad, has_bias = ctx.cfg
        dy = dy.contiguous()
        B, H, W, Cx = x.shape
        cout, cin = weight.shape[0], weight.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if s == 1:
                dx = ops.conv2d(dy, wdg, k, k, 1, k - 1 - p)
            elif k == s and p == 0:          # non-overlapping patches: one GEMM + un-patchify
                pm = ops.gemm(dy, wdg)       # [B,Ho,Wo,k*k*Cin]
                dx = ops.depatchify(pm.view(-1, k * k * Cx), B, dy.shape[1], dy.shape[2], k, Cx)
            else:                            # transposed conv = stride-1 conv of the zero-inserted gradient
                z = ops.zero_insert(dy, H, W, s)
                dx = ops.conv2d(z, wdg, k, k, 1, k - 1 - p)
        if ctx.needs_input_grad[1]:
            dw = unpack_conv_grad(ops.conv2d_wgrad(dy, x, k, k, s, p), cout, cin, k, cin_pad)
        if has_bias and ctx.needs_input_grad[2]:
            db = colsum_f32(dy)
        return dx, dw, db, None, None, None, None, None, None


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y = ops.layernorm(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dg = torch.zeros_like(gamma)
        db = torch.zeros_like(gamma)
        dx = ops.layernorm_bwd(x, dy.contiguous(), gamma, ctx.eps, dg, db)
        return dx, dg, db, None


class DwConvFn(Function):
    """depthwise 3x3 (+ exact GELU).  wt: [9][C] f32 pack, wt_flip: taps reversed (input gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wt, wt_flip, gelu):
        z = ops.dwconv3x3(x, wt, bias)
        y = ops.dwconv3x3(x, wt, bias, act=ops.ACT_GELU) if gelu else z
        ctx.save_for_backward(x, z if gelu else x, weight)
        ctx.cfg = (wt_flip, gelu, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, weight = ctx.saved_tensors
        wt_flip, gelu, has_bias = ctx.cfg
        dy = dy.contiguous()
        dz = ops.gelu_bwd(z, dy) if gelu else dy
        dx = ops.dwconv3x3(dz, wt_flip) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            C = x.shape[-1]
            dwt = torch.zeros((9, C), dtype=torch.float32, device=x.device)
            dbt = torch.zeros(C, dtype=torch.float32, device=x.device) if has_bias else None
            ops.dwconv3x3_wgrad(x, dz, dwt, dbt)
            dw = dwt.t().reshape(weight.shape).contiguous()
            db = dbt
        return dx, dw, db, None, None, None


class SraAttentionFn(Function):
    """softmax(q k^T scale) v with head_dim 64 and <= 128 keys (PVT spatial-reduction attention).
    q [B,N,C], kv [B,Lk,2C] (k | v, head h at columns 64h).  Backward recomputes P per head."""

    @staticmethod
    def forward(ctx, q, kv, heads, scale):
        B, N, C = q.shape
        Lk = kv.shape[1]
        out = torch.empty_like(q)
        ops.attention(q, kv, kv[..., C:], out, batch=B, heads=heads, nwin=1, Lq=N, Lk=Lk, D=64, DV=64, q_bs=N * C,
                      k_bs=Lk * 2 * C, v_bs=Lk * 2 * C, o_bs=N * C, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C, q_hs=64,
                      k_hs=64, v_hs=64, o_hs=64, scale=scale)
        ctx.save_for_backward(q, kv)
        ctx.cfg = (heads, scale)
        return out

    @staticmethod
    def backward(ctx, do):
        q, kv = ctx.saved_tensors
        heads, scale = ctx.cfg
        do = do.contiguous()
        B, N, C = q.shape
        Lk = kv.shape[1]
        Lp = 128
        assert Lk <= Lp
        dt, dev = q.dtype, q.device
        dq = torch.empty_like(q)
        dkv_pad = torch.zeros((B, Lp, 2 * C), dtype=dt, device=dev)
        S = torch.empty((B, N, Lp), dtype=dt, device=dev)
        dP = torch.empty((B, N, Lp), dtype=dt, device=dev)
        for h in range(heads):
            qh, kh, vh, doh = q[..., 64 * h:], kv[..., 64 * h:], kv[..., C + 64 * h:], do[..., 64 * h:]
            ops.gemm_batched(qh, kh, S, B, N, Lk, 64, C, 2 * C, Lp, N * C, Lk * 2 * C, N * Lp)
            P = ops.softmax_rows(S.view(B * N, Lp), Lk, scale, out=S.view(B * N, Lp)).view(B, N, Lp)
            dV = ops.gemm_tn_batched(P, doh, B, N, Lp, 64, Lp, C, N * Lp, N * C)              # f32 [B,Lp,64]
            ops.gemm_batched(doh, vh, dP, B, N, Lk, 64, C, 2 * C, Lp, N * C, Lk * 2 * C, N * Lp)
            dS = ops.softmax_bwd_rows(P.view(B * N, Lp), dP.view(B * N, Lp), Lk, scale, out=dP.view(B * N, Lp))
            dS = dS.view(B, N, Lp)
            khT = ops.transpose_pad(kv[:, :, 64 * h:64 * h + 64], Lp)                        # [B,64,Lp]
            ops.gemm_batched(dS, khT, dq[..., 64 * h:], B, N, 64, Lp, Lp, Lp, C, N * Lp, 64 * Lp, N * C)
            dK = ops.gemm_tn_batched(dS, qh, B, N, Lp, 64, Lp, C, N * Lp, N * C)             # f32 [B,Lp,64]
            ops.copy_cols(dK.view(B * Lp, 64), 0, 64, dkv_pad.view(B * Lp, 2 * C), 64 * h)
            ops.copy_cols(dV.view(B * Lp, 64), 0, 64, dkv_pad.view(B * Lp, 2 * C), C + 64 * h)
        return dq, dkv_pad[:, :Lk].contiguous(), None, None
