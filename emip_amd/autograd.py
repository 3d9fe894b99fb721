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
    def forward(ctx, x, weight, bias, res, wp, wpt, bpack=None):
        # wp may carry zero-padded output rows (N padded to a multiple of 8: the 1-channel mask head); bpack is the
        # matching padded bias
        y = ops.gemm(x, wp, bias=bias if bpack is None else bpack, res=res)
        ctx.save_for_backward(x, weight)
        ctx.wpt, ctx.has_bias, ctx.has_res = wpt, bias is not None, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.gemm(dy, ctx.wpt) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            n = weight.shape[0]
            kw = weight.numel() // n                      # x may carry zero-padded columns beyond the weight's K
            dw = ops.gemm_tn(dy, x)[:n, :kw].contiguous().view_as(weight)
        db = colsum_f32(dy)[:weight.shape[0]].contiguous() if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None, None, None


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


class BNReluFn(Function):
    """train-mode BatchNorm2d (+ReLU) on a conv output: batch statistics, running buffers updated by the caller."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu):
        sums = ops.chan_stats(x, 1)
        y = ops.chan_norm_apply(x, sums, 1, eps, relu_inner=relu, gamma=gamma, beta=beta)
        ctx.save_for_backward(x, y, gamma, sums)
        ctx.cfg = (eps, relu)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, dy, _dsums):
        x, y, gamma, sums = ctx.saved_tensors
        eps, relu = ctx.cfg
        dg = torch.zeros_like(gamma)
        db = torch.zeros_like(gamma)
        dx = ops.bn_train_bwd(x, dy.contiguous(), y if relu else None, sums, gamma, dg, db, eps)
        return dx, dg, db, None, None


class BilinearFn(Function):
    """channels-last bilinear resize (nn.Upsample x2, align_corners=True)"""

    @staticmethod
    def forward(ctx, x, Ho, Wo, align):
        ctx.cfg = (x.shape[1], x.shape[2], align, x.dtype)
        return ops.bilinear(x, Ho, Wo, align)

    @staticmethod
    def backward(ctx, dy):
        H, W, align, dt = ctx.cfg
        return ops.bilinear_bwd(dy.contiguous(), H, W, align).to(dt), None, None, None


class BilinearPlanarFn(Function):
    """channels xc..xc+C of a channels-last tensor -> planar f32 [B,C,Ho,Wo] (x8 mask logits / train-mode flow)"""

    @staticmethod
    def forward(ctx, x, xc, C, Ho, Wo, align, mul):
        ctx.cfg = (x.shape, xc, C, align, mul, x.dtype)
        return ops.bilinear_planar(x, xc, C, Ho, Wo, align, mul)

    @staticmethod
    def backward(ctx, dy):
        shape, xc, C, align, mul, dt = ctx.cfg
        g = ops.bilinear_planar_bwd(dy.contiguous(), shape[1], shape[2], align, mul)       # f32 [B,H,W,C]
        if shape[-1] == C and xc == 0:
            return g.to(dt), None, None, None, None, None, None
        dx = torch.zeros(shape, dtype=dt, device=dy.device)
        ops.copy_cols(g, 0, C, dx, xc)
        return dx, None, None, None, None, None, None


class MulFn(Function):
    """a * b (* c) elementwise"""

    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b, c)
        return ops.eltwise(a, b, 0) if c is None else ops.eltwise(a, b, 1, c3=c)

    @staticmethod
    def backward(ctx, dy):
        a, b, c = ctx.saved_tensors
        dy = dy.contiguous()
        if c is None:
            return ops.eltwise(dy, b, 0), ops.eltwise(dy, a, 0), None
        return ops.eltwise(dy, b, 1, c3=c), ops.eltwise(dy, a, 1, c3=c), ops.eltwise(dy, a, 1, c3=b)


class ConcatFn(Function):
    """channel concatenation of channels-last tensors (optionally zero-padded to cpad channels)"""

    @staticmethod
    def forward(ctx, cpad, *parts):
        widths = [p.shape[-1] for p in parts]
        total = sum(widths)
        cpad = cpad or total
        out = torch.empty(parts[0].shape[:-1] + (cpad,), dtype=parts[0].dtype, device=parts[0].device)
        off = 0
        for i, p in enumerate(parts):
            last = i == len(parts) - 1
            ops.copy_cols(p, 0, widths[i], out, off, (cpad - off) if last else widths[i])
            off += widths[i]
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, dy):
        grads, off = [], 0
        for w in ctx.widths:
            g = torch.empty(dy.shape[:-1] + (w,), dtype=dy.dtype, device=dy.device)
            ops.copy_cols(dy, off, w, g, 0)
            grads.append(g)
            off += w
        return (None,) + tuple(grads)


class GateFn(Function):
    """y = gelu(z[:, :Ch]) * z[:, Ch:], zero-padded to cpad channels"""

    @staticmethod
    def forward(ctx, z, ch, cpad):
        ctx.save_for_backward(z)
        ctx.ch = ch
        return ops.gate_fwd(z, ch, cpad)

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        return ops.gate_bwd(z, dy.contiguous(), ctx.ch), None, None


class MdtaFn(Function):
    """channel attention of the MDTA block: out[p, c1] = sum_c2 softmax(qhat^T khat * tau)[c1, c2] v[p, c2]
    q [B,h,w,128], kv [B,h,w,256] (k | v), temperature [heads,1,1]"""

    @staticmethod
    def forward(ctx, q, kv, temperature):
        B, h, w, C = q.shape
        P, heads = h * w, C // 64
        temp = temperature.detach().reshape(-1).contiguous()
        ws, attn = ops.mdta_attn_ws(q.view(B, P, C), kv.view(B, P, 2 * C)[..., :C], temp, B, heads, P)
        o = torch.empty_like(q)
        for hd in range(heads):
            ops.gemm_batched(kv.view(B, P, 2 * C)[..., C + 64 * hd:], attn[:, hd], o.view(B, P, C)[..., 64 * hd:],
                             batch=B, M=P, N=64, K=64, lda=2 * C, ldw=64, ldc=C, bsA=P * 2 * C, bsW=heads * 4096,
                             bsC=P * C)
        ctx.save_for_backward(q, kv, temperature, ws, attn)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, temperature, ws, attn = ctx.saved_tensors
        do = do.contiguous()
        B, h, w, C = q.shape
        P, heads = h * w, C // 64
        q3, kv3, do3 = q.view(B, P, C), kv.view(B, P, 2 * C), do.view(B, P, C)
        temp = temperature.detach().reshape(-1).contiguous()
        dA = torch.empty((B, heads, 64, 64), dtype=torch.float32, device=q.device)
        for hd in range(heads):       # dA[c1,c2] = sum_p do[p,c1] v[p,c2]
            dA[:, hd] = ops.gemm_tn_batched(do3[..., 64 * hd:], kv3[..., C + 64 * hd:], B, P, 64, 64, C, 2 * C, P * C,
                                            P * 2 * C)
        dG, dGT, sq, sk, dtau = ops.mdta_bwd_small(ws, temp, attn, dA, B, heads)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        dq3, dkv3 = dq.view(B, P, C), dkv.view(B, P, 2 * C)
        for hd in range(heads):
            qh, kh = q3[..., 64 * hd:64 * hd + 64], kv3[..., 64 * hd:64 * hd + 64]
            attnT = ops.transpose_pad(attn[:, hd], 64)                                          # [B,64(c2),64(c1)]
            # dV[p,c2] = sum_c1 do[p,c1] A[c1,c2]
            ops.gemm_batched(do3[..., 64 * hd:], attnT, dkv3[..., C + 64 * hd:], B, P, 64, 64, C, 64, 2 * C, P * C,
                             4096, P * 2 * C)
            # dQ[p,c1] = sum_c2 dG[c1,c2] K[p,c2] + sq[c1] Q[p,c1]
            ops.gemm_batched(kh, dG[:, hd], dq3[..., 64 * hd:], B, P, 64, 64, 2 * C, 64, C, P * 2 * C, heads * 4096,
                             P * C)
            ops.colscale_add(dq3[..., 64 * hd:64 * hd + 64], qh, sq[:, hd], heads * 64, P,
                             dq3[..., 64 * hd:64 * hd + 64])
            # dK[p,c2] = sum_c1 dG[c1,c2] Q[p,c1] + sk[c2] K[p,c2]
            ops.gemm_batched(qh, dGT[:, hd], dkv3[..., 64 * hd:], B, P, 64, 64, C, 64, 2 * C, P * C, heads * 4096,
                             P * 2 * C)
            ops.colscale_add(dkv3[..., 64 * hd:64 * hd + 64], kh, sk[:, hd], heads * 64, P,
                             dkv3[..., 64 * hd:64 * hd + 64])
        return dq, dkv, dtau.view_as(temperature)


class HybridELossFn(Function):
    """hybrid_e_loss (loss/loss_pred.py:4-22) -> f32 [1]; gradient w.r.t. the logits only."""

    @staticmethod
    def forward(ctx, pred, mask):
        pred, mask = pred.contiguous(), mask.contiguous()
        out, ws = ops.hybrid_e_loss_fwd_ws(pred, mask)
        ctx.save_for_backward(pred, mask, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        pred, mask, ws = ctx.saved_tensors
        return ops.hybrid_e_loss_bwd(pred, mask, ws, g.contiguous().float()), None


class UnflowPairLossFn(Function):
    """0.5 * (photometric(im1, warp(im2, fw), m1) + photometric(im2, warp(im1, bw), m2)) (loss_flow.py:96-131) -> f32 [1];
    gradient w.r.t. the two flows through the warps (the occlusion masks are thresholded, hence constant)."""

    @staticmethod
    def forward(ctx, fw, bw, im1, im2, m1, m2):
        r1, r2 = ops.flow_warp(im2, fw), ops.flow_warp(im1, bw)
        out = torch.empty(1, dtype=torch.float32, device=fw.device)
        ws1 = ops.photometric_loss_ws(im1, r1, m1, out, 0.5, False)
        ws2 = ops.photometric_loss_ws(im2, r2, m2, out, 0.5, True)
        ctx.save_for_backward(fw, bw, im1, im2, m1, m2, r1, r2, ws1, ws2)
        return out

    @staticmethod
    def backward(ctx, g):
        fw, bw, im1, im2, m1, m2, r1, r2, ws1, ws2 = ctx.saved_tensors
        g = g.contiguous().float()
        d1 = ops.photometric_loss_bwd(im1, r1, m1, ws1, g, 0.5)
        d2 = ops.photometric_loss_bwd(im2, r2, m2, ws2, g, 0.5)
        return ops.flow_warp_bwd(im2, fw, d1), ops.flow_warp_bwd(im1, bw, d2), None, None, None, None
