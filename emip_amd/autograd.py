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


def conv_weight_grad(dy, x, weight, k, s, p, cin_pad=None, bias=None):
    """(weight gradient of a k x k convolution in the parameter's [Cout, Cin, k, k] layout, bias gradient or None); inside a
    training step the contraction joins the step's grouped weight-gradient launch where it is eligible (ops.conv2d_wgrad),
    which then also sums the columns of dy.  bias: the bias PARAMETER when its gradient is wanted (else None)."""
    cout, cin = weight.shape[0], weight.shape[1]
    cp = cin_pad or cin
    lazy = lambda t: t.view(cout, k, k, cp)[..., :cin].permute(0, 3, 1, 2)        # strided view: fixup() adds it in one pass
    dwp, g, db = ops.conv2d_wgrad(dy, x, k, k, s, p, defer_to=(weight if weight.is_leaf else None, lazy),
                                  want_db=bias is not None, bias_param=bias if (bias is not None and bias.is_leaf) else None)
    if g is None:
        g = unpack_conv_grad(dwp, cout, cin, k, cin_pad)
    if bias is not None and db is None:
        db = colsum_f32(dy)
    return g, (db[:] if db is not None else None)          # a fresh view object (AccumulateGrad steals only unshared tensors)


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
    def forward(ctx, x, weight, bias, res, wp, wpt, bpack=None, drop=None):
        # wp may carry zero-padded output rows (N padded to a multiple of 8: the 1-channel mask head); bpack is the
        # matching padded bias.  drop = (scale f32 [B], scale [B, C], scale - 1 [B, C], rows per sample): stochastic depth,
        # y = res + scale[sample] * (x W^T + b), applied in the GEMM's epilogue where the 8-wave body takes the launch
        b = bias if bpack is None else bpack
        ctx.drop = None
        if drop is not None:
            sb, s_bc, sm1, rps = drop
            ctx.drop = (sm1, rps)
            y = ops.gemm_rowscale(x, wp, b, res, sb, rps)
            if y is None:
                br = ops.gemm(x, wp, bias=b)
                y = torch.empty_like(res)
                ops.colscale_add(res, br, s_bc, s_bc.shape[1], rps, y)
        else:
            y = ops.gemm(x, wp, bias=b, res=res)
        ctx.save_for_backward(x, weight)
        ctx.wpt, ctx.has_bias, ctx.has_res = wpt, bias is not None, res is not None
        ctx.bias_param = bias if (bias is not None and bias.is_leaf) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        if ctx.drop is not None:                  # the branch sees scale * dy; the skip path the plain dy
            sm1, rps = ctx.drop
            dyb = torch.empty_like(dy)
            ops.colscale_add(dy, dy, sm1, sm1.shape[1], rps, dyb)        # dy + (s - 1) dy = s dy
            dy = dyb
        dx = ops.gemm(dy, ctx.wpt) if ctx.needs_input_grad[0] else None
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            n = weight.shape[0]
            kw = weight.numel() // n                      # x may carry zero-padded columns beyond the weight's K
            # The grouped launch fills dwf / dbf at flush_wgrads(): only a result autograd receives UNTOUCHED may wait for
            # it.  With padded rows (the 1-channel mask head) or padded K columns (the 340 -> 344 GDFN hidden width) the
            # slice below is a copy, so those launch now.
            exact = dy.shape[-1] == n and x.shape[-1] == kw
            own = (weight if weight.is_leaf else None, ctx.bias_param)
            if want_db:                                    # bias gradient from the dY tiles the weight-gradient GEMM stages
                dwf, dbf = ops.gemm_tn(dy, x, with_colsum=True, defer=exact, owners=own)
                db = dbf[:n] if exact else dbf[:n].contiguous()       # (a fresh view object: see below)
            else:
                dwf = ops.gemm_tn(dy, x, defer=exact, owners=own)
            # autograd keeps a returned gradient AS the parameter's .grad only if nothing else holds the tensor object; the
            # deferral queue holds dwf / dbf themselves, so fresh views go out (a held object would be cloned -- of zeros)
            dw = dwf.view_as(weight) if exact else dwf[:n, :kw].contiguous().view_as(weight)
        elif want_db:
            db = colsum_f32(dy)[:weight.shape[0]].contiguous()
        return dx, dw, db, dres, None, None, None, None


class FanOutFn(Function):
    """(x, x W_1^T, ..., x W_k^T) for FROZEN bias-free weights: the token stream of a GMFlow transformer block feeds five
    projections and the skip connection (transformer.py:160-190, 348-401), and autograd would sum the six gradients that come
    back with five add launches (46 per step over the 6 blocks).  Here they meet in the residual epilogues of the
    input-gradient GEMMs: dx = dskip + sum_i dy_i W_i, one GEMM per projection, each adding the running sum.
    packs: (W_i packed [N, K], W_i^T packed [K, N]) per projection."""

    @staticmethod
    def forward(ctx, x, *packs):
        ctx.wts = [wt for _, wt in packs]
        ctx.set_materialize_grads(False)
        return (x,) + tuple(ops.gemm(x, wp) for wp, _ in packs)

    @staticmethod
    def backward(ctx, dskip, *dys):
        dx, own = dskip, False
        for dy, wt in zip(dys, ctx.wts):
            if dy is None:
                continue
            dy = dy.contiguous()
            if dx is None:
                dx, own = ops.gemm(dy, wt), True
            elif own:
                dx = ops.gemm(dy, wt, res=dx, out=dx)           # the running sum is this node's own tensor: in place
            else:
                dx, own = ops.gemm(dy, wt, res=dx.contiguous()), True     # dskip may be shared with other nodes: out of place
        return (dx,) + (None,) * len(ctx.wts)


class Linear2Fn(Function):
    """y = [a | b] W^T for a FROZEN bias-free weight, without the concatenated operand: the forward GEMM walks K over the two
    sources (emip_gemm's a2), the backward makes the two input gradients from the two row blocks of W^T (GMFlow's FFN input
    torch.cat([source, message]), transformer.py:201-202: a concat copy forward and a split copy backward per block otherwise).
    wp: W packed [N, Ka + Kb]; wpt: W^T packed [Ka + Kb, N]."""

    @staticmethod
    def forward(ctx, a, b, wp, wpt):
        ctx.wpt, ctx.ka = wpt, a.shape[-1]
        return ops.gemm(a, wp, a2=b)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        wpt, ka = ctx.wpt, ctx.ka
        da = ops.gemm(dy, wpt[:ka]) if ctx.needs_input_grad[0] else None
        db = ops.gemm(dy, wpt[ka:]) if ctx.needs_input_grad[1] else None
        return da, db, None, None


class ConvFn(Function):
    """NHWC conv.  wp: forward pack; wdg: pack for the input gradient (flipped + transposed, or W^T for patch convs)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wp, wdg, k, s, p, cin_pad):
        y = ops.conv2d(x, wp, k, k, s, p, bias=bias)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (wdg, k, s, p, cin_pad, bias is not None)
        ctx.bias_param = bias if (bias is not None and bias.is_leaf) else None
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
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            # the bias gradient rides the same (grouped) launch when the bias is a leaf parameter
            dw, db = conv_weight_grad(dy, x, weight, k, s, p, cin_pad, bias=ctx.bias_param if want_db else None)
        if want_db and db is None:
            db = colsum_f32(dy)
        return dx, dw, db, None, None, None, None, None, None


class ConvCorr0Fn(Function):
    """conv_corr.0 (model/EMIP_short/model.py:59,96: Conv2d(44*44, 968, 3, 1, 1) on the raw correlation volume of
    matching.py:16-20) from the volume's rank-C factors -- see CoUpdater.run_conv_corr_factored for the algebra.
        tokens [2B, n, C] (frame 1 | frame 2), weight [Cout, n, 3, 3], bias [Cout]
        wr  [Cout * 9, n]: weight / sqrt(C), rows (co, tap);  wrt [n, Cout * 9]: its transpose (activation-dtype packs)
        -> y [B, h, w, Cout] = bias + sum_tap sum_d F0[p + tap, d] G[b][co, tap, d],   G[b] = wr F1[b]
    The backward is four per-image GEMMs, one stacked TN contraction and a gather: dP = dY G2 (G2 = G as [Cout, 9 C]), dF0 =
    col2im(dP), dG2 = dY^T P, dF1 = wr^T dG, dwr = sum_b dG[b] F1[b]^T.  Neither the volume nor its gradient exists."""

    @staticmethod
    def forward(ctx, tokens, weight, bias, wr, wrt, h, w):
        B2, n, C = tokens.shape
        B, cout, dt = B2 // 2, weight.shape[0], tokens.dtype
        f0, f1 = tokens[:B], tokens[B:]
        f1t = ops.transpose_pad(f1, n)                                           # [B, C, n]
        g = torch.empty((B, cout * 9, C), dtype=dt, device=tokens.device)
        ops.gemm_batched(wr, f1t, g, B, cout * 9, C, n, n, n, C, 0, C * n, cout * 9 * C)
        pm = ops.im2col3x3(f0.reshape(B, h, w, C))                               # [B, n, 9 C]
        y = torch.empty((B, h, w, cout), dtype=dt, device=tokens.device)
        ops.gemm_batched_bias(pm, g, y, B, n, cout, 9 * C, 9 * C, 9 * C, cout, n * 9 * C, cout * 9 * C, n * cout,
                              bias=bias.detach().float() if bias is not None else None)
        ctx.save_for_backward(f1t, g, pm, weight)
        ctx.cfg = (wrt, h, w, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        f1t, g, pm, weight = ctx.saved_tensors
        wrt, h, w, has_bias = ctx.cfg
        B, C, n = f1t.shape
        cout, dt = weight.shape[0], f1t.dtype
        dy = dy.contiguous().view(B, n, cout)
        dtok = dw = db = None
        if has_bias and ctx.needs_input_grad[2]:
            db = colsum_f32(dy)
        dg = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dg2 = ops.gemm_tn_batched(dy, pm, B, n, cout, 9 * C, cout, 9 * C, n * cout, n * 9 * C)     # f32 [B, Cout, 9 C]
            dg = dg2.to(dt).view(B, cout * 9, C)
            dgt = ops.transpose_pad(dg, cout * 9)                                # [B, C, Cout * 9]
        if ctx.needs_input_grad[0]:
            g2t = ops.transpose_pad(g.view(B, cout, 9 * C), cout)                # [B, 9 C, Cout]
            dp = torch.empty((B, n, 9 * C), dtype=dt, device=dy.device)
            ops.gemm_batched(dy, g2t, dp, B, n, 9 * C, cout, cout, cout, 9 * C, n * cout, 9 * C * cout, n * 9 * C)
            dtok = torch.empty((2 * B, n, C), dtype=dt, device=dy.device)
            ops.col2im3x3(dp, B, h, w, C, out=dtok[:B])
            ops.gemm_batched(wrt, dgt, dtok[B:], B, n, C, cout * 9, cout * 9, cout * 9, C, 0, C * cout * 9, n * C)
        if ctx.needs_input_grad[1]:
            dwr = ops.gemm_tn(dgt.view(B * C, cout * 9), f1t.view(B * C, n))     # f32 [Cout * 9, n]
            dw = (dwr.view(cout, 3, 3, n).permute(0, 3, 1, 2) * C ** -0.5).contiguous()
        return dtok, dw, db, None, None, None, None


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
        dx, dg, db = ops.layernorm_bwd_fresh(x, dy.contiguous(), gamma, ctx.eps)
        return dx, dg, db, None


class LayerNormSkipFn(Function):
    """(LayerNorm(x), x): the pre-norm residual blocks read x twice, for the norm and for the skip connection.  Handing the
    skip operand out of THIS node makes both gradients arrive here, and the LayerNorm backward kernel adds the skip path's
    in its store (emip_layernorm_bwd_res) -- otherwise autograd sums them with one add launch per residual branch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y = ops.layernorm(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        return y, x                  # an input returned as an output: autograd re-wraps it as an output of this node

    @staticmethod
    def backward(ctx, dy, dskip):
        x, gamma = ctx.saved_tensors
        if dy is None:
            return dskip, None, None, None
        dx, dg, db = ops.layernorm_bwd_fresh(x, dy.contiguous(), gamma, ctx.eps,
                                             dres=None if dskip is None else dskip.contiguous())
        return dx, dg, db, None


class QSrFn(Function):
    """q = h Wq^T + bq and s = sr_conv(h) (k x k, stride k: non-overlapping patches) of the SAME normed tokens h
    (lib/pvt_v2.py:105-110): one node, so that the two input gradients meet in the residual epilogue of the q-gradient
    GEMM instead of in an autograd add launch."""

    @staticmethod
    def forward(ctx, h, wq, bq, wsr, bsr, wq_p, wq_t, wsr_p, wsr_d, k):
        q = ops.gemm(h, wq_p, bias=bq)
        s = ops.conv2d(h, wsr_p, k, k, k, 0, bias=bsr)
        ctx.save_for_backward(h, wq, wsr)
        ctx.cfg = (wq_t, wsr_d, k, bq is not None, bsr is not None)
        ctx.bq_param = bq if (bq is not None and bq.is_leaf) else None
        ctx.bsr_param = bsr if (bsr is not None and bsr.is_leaf) else None
        return q, s

    @staticmethod
    def backward(ctx, dq, ds):
        h, wq, wsr = ctx.saved_tensors
        wq_t, wsr_d, k, has_bq, has_bsr = ctx.cfg
        dq, ds = dq.contiguous(), ds.contiguous()
        B, H, W, C = h.shape
        dh = dwq = dbq = dwsr = dbsr = None
        if ctx.needs_input_grad[0]:
            pm = ops.gemm(ds, wsr_d)                                   # [B,Ho,Wo,k*k*C]
            dh = ops.depatchify(pm.view(-1, k * k * C), B, ds.shape[1], ds.shape[2], k, C)
            dh = ops.gemm(dq, wq_t, res=dh, out=dh)                    # + dq Wq, added in the epilogue
        if ctx.needs_input_grad[1]:
            own = (wq if wq.is_leaf else None, ctx.bq_param)
            if has_bq and ctx.needs_input_grad[2]:
                dwq, dbq = ops.gemm_tn(dq, h, with_colsum=True, defer=True, owners=own)
            else:
                dwq = ops.gemm_tn(dq, h, defer=True, owners=own)
            dwq = dwq.view_as(wq)
            dbq = dbq[:] if dbq is not None else None          # fresh view objects, as in LinearFn.backward
        elif has_bq and ctx.needs_input_grad[2]:
            dbq = colsum_f32(dq)
        want_dbsr = has_bsr and ctx.needs_input_grad[4]
        if ctx.needs_input_grad[3]:
            dwsr, dbsr = conv_weight_grad(ds, h, wsr, k, k, 0, bias=ctx.bsr_param if want_dbsr else None)
        if want_dbsr and dbsr is None:
            dbsr = colsum_f32(ds)
        return dh, dwq, dbq, dwsr, dbsr, None, None, None, None, None


DW_BWD_FUSED = True         # emip_dwconv3x3_bwd_fused instead of gelu_bwd + dwconv3x3(flipped) + dwconv3x3_wgrad


class DwConvFn(Function):
    """depthwise 3x3 (+ exact GELU).  wt: [9][C] f32 pack, wt_flip: taps reversed (input gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wt, wt_flip, gelu):
        if gelu:
            y, z = ops.dwconv3x3_dual(x, wt, bias, ops.ACT_GELU)      # one pass: GELU output + pre-activation for backward
        else:
            y = z = ops.dwconv3x3(x, wt, bias)
        ctx.save_for_backward(x, z if gelu else x, weight)
        ctx.cfg = (wt_flip, gelu, bias is not None)
        ctx.wt = wt
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, weight = ctx.saved_tensors
        wt_flip, gelu, has_bias = ctx.cfg
        dy = dy.contiguous()
        C = x.shape[-1]
        if (DW_BWD_FUSED and x.dtype == torch.bfloat16 and C % 8 == 0 and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
                and x.is_contiguous() and z.is_contiguous()):
            # one pass over the hidden gradient: GELU backward, input gradient and weight / bias gradient (dw_bwd.hip);
            # the weight gradient comes out in the parameter's own order, so autograd receives arena views as they are
            acc = ops.grad_zeros((10 * C,), x.device)
            dwt, dbt = acc[:9 * C], (acc[9 * C:] if has_bias else None)
            dx = ops.dwconv3x3_bwd_fused(x, z, dy, ctx.wt, dwt, dbt, gelu)
            return dx, dwt.view(weight.shape), dbt, None, None, None
        dz = ops.gelu_bwd(z, dy) if gelu else dy
        dx = ops.dwconv3x3(dz, wt_flip) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            C = x.shape[-1]
            acc = ops.grad_zeros((10, C), x.device)                              # 9 taps + bias
            dwt, dbt = acc[:9], (acc[9] if has_bias else None)
            ops.dwconv3x3_wgrad(x, dz, dwt, dbt)
            dw = dwt.t().reshape(weight.shape).contiguous()
            db = dbt
        return dx, dw, db, None, None, None


class SraAttentionFn(Function):
    """softmax(q k^T scale) v with head_dim 64 and <= 128 keys (PVT spatial-reduction attention).
    q [B,N,C], kv [B,Lk,2C] (k | v, head h at columns 64h).  bf16: one launch forward (emip_sra_attention_lse, which also
    leaves the log-sum-exp of every query) and ONE launch backward (emip_sra_attention_bwd: P recomputed from L, dQ / dK / dV
    in a single pass over the queries).  f32 parity mode: the generic attention kernel forward, the unfused chain backward."""

    @staticmethod
    def forward(ctx, q, kv, heads, scale):
        B, N, C = q.shape
        Lk = kv.shape[1]
        out = torch.empty_like(q)
        fused = q.dtype == torch.bfloat16 and Lk <= 128 and q.is_contiguous() and kv.is_contiguous()
        if fused:
            L = ops.sra_attention_lse(q, kv, out, B, heads, N, Lk, scale)
            ctx.save_for_backward(q, kv, out, L)
        else:
            ops.attention(q, kv, kv[..., C:], out, batch=B, heads=heads, nwin=1, Lq=N, Lk=Lk, D=64, DV=64, q_bs=N * C,
                          k_bs=Lk * 2 * C, v_bs=Lk * 2 * C, o_bs=N * C, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C, q_hs=64,
                          k_hs=64, v_hs=64, o_hs=64, scale=scale)
            ctx.save_for_backward(q, kv)
        ctx.cfg = (heads, scale, fused)
        return out

    @staticmethod
    def backward(ctx, do):
        heads, scale, fused = ctx.cfg
        if fused:
            q, kv, out, L = ctx.saved_tensors
            B, N, C = q.shape
            Lk = kv.shape[1]
            dq, dkv32 = ops.sra_attention_bwd(q, kv, out, do.contiguous(), L, B, heads, N, Lk, scale)
            if dkv32.dtype == q.dtype:
                return dq, dkv32, None, None
            dkv = torch.empty((B, Lk, 2 * C), dtype=q.dtype, device=q.device)
            ops.copy_cols(dkv32.view(B * Lk, 2 * C), 0, 2 * C, dkv.view(B * Lk, 2 * C), 0)
            return dq, dkv, None, None
        return _sra_backward_unfused(ctx, do)




def _sra_backward_unfused(ctx, do):
    """all heads in one launch per step (two-level batch: image, head); dK / dV are accumulated straight into the f32
    gradient of the kv projection output"""
    q, kv = ctx.saved_tensors
    heads, scale, _ = ctx.cfg
    do = do.contiguous()
    B, N, C = q.shape
    Lk = kv.shape[1]
    Lp = 128
    assert Lk <= Lp
    dt, dev = q.dtype, q.device
    Z = B * heads
    k_, v_ = kv, kv[..., C:]                                  # head h: columns 64 h .. of k, C + 64 h .. of v
    S = torch.empty((B, heads, N, Lp), dtype=dt, device=dev)
    ops.gemm_heads(q, k_, S, Z, heads, N, Lk, 64, C, 2 * C, Lp, N * C, 64, Lk * 2 * C, 64, heads * N * Lp, N * Lp)
    P = ops.softmax_rows(S.view(Z * N, Lp), Lk, scale, out=S.view(Z * N, Lp))
    dkv32 = torch.zeros((B, Lp, 2 * C), dtype=torch.float32, device=dev)
    ops.gemm_tn_heads(P, do, dkv32[..., C:], Z, heads, N, Lp, 64, Lp, C, 2 * C, heads * N * Lp, N * Lp, N * C, 64,
                      Lp * 2 * C, 64)                                                         # dV
    dP = torch.empty((B, heads, N, Lp), dtype=dt, device=dev)
    ops.gemm_heads(do, v_, dP, Z, heads, N, Lk, 64, C, 2 * C, Lp, N * C, 64, Lk * 2 * C, 64, heads * N * Lp, N * Lp)
    dS = ops.softmax_bwd_rows(P, dP.view(Z * N, Lp), Lk, scale, out=dP.view(Z * N, Lp))
    kT = ops.transpose_pad_heads(kv, Z, heads, Lk, 64, Lp, 2 * C, Lk * 2 * C, 64)               # [Z,64,Lp]
    dq = torch.empty_like(q)
    ops.gemm_heads(dS, kT, dq, Z, heads, N, 64, Lp, Lp, Lp, C, heads * N * Lp, N * Lp, heads * 64 * Lp, 64 * Lp,
                   N * C, 64)
    ops.gemm_tn_heads(dS, q, dkv32, Z, heads, N, Lp, 64, Lp, C, 2 * C, heads * N * Lp, N * Lp, N * C, 64, Lp * 2 * C,
                      64)                                                                      # dK
    dkv = torch.empty((B, Lp, 2 * C), dtype=dt, device=dev)
    ops.copy_cols(dkv32.view(B * Lp, 2 * C), 0, 2 * C, dkv.view(B * Lp, 2 * C), 0)
    return dq, dkv[:, :Lk].contiguous(), None, None


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
        dg = ops.grad_zeros(gamma.shape, gamma.device)
        db = ops.grad_zeros(gamma.shape, gamma.device)
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


# ---------------------------------------------------------------------------------------------------------------------
# GMFlow stream (frozen weights: input gradients only).  transformer.py / matching.py / gmflow.py of the reference.

def _round_up(x, m):
    return (x + m - 1) // m * m


def _to_act(x32, dt):
    """f32 rows -> activation dtype (a HIP copy kernel, not a torch cast)"""
    if dt == torch.float32:
        return x32
    out = torch.empty(x32.shape, dtype=dt, device=x32.device)
    C = x32.shape[-1]
    ops.copy_cols(x32.view(-1, C), 0, C, out.view(-1, C), 0)
    return out


def dense_attention_bwd(q, k, v, do, L, scale, gq=None, gk=None, nwin=1, dscore=None, need_dv=True):
    """Backward of softmax(q k^T * scale [+ shift mask]) v on dense batches.
    q, k [Z, Lp, D], v [Zv, Lp, DV] (Zv = 1: shared), do [Z, Lp, DV], activation dtype, contiguous, rows >= L zero.
    dscore: optional upstream gradient w.r.t. the SCALED scores [Z, Lp, Lp].
    Returns dq (activation dtype), dk f32, dv f32 | None (all [Z, Lp, *])."""
    Z, Lp, D = q.shape
    DV = v.shape[-1]
    dt, dev = q.dtype, q.device
    S = torch.empty((Z, Lp, Lp), dtype=dt, device=dev)
    ops.gemm_batched(q, k, S, Z, Lp, Lp, D, D, D, Lp, Lp * D, Lp * D, Lp * Lp)
    P = ops.softmax_rows(S.view(Z * Lp, Lp), L, scale, gq, gk, period=Lp, nwin=nwin, out=S.view(Z * Lp, Lp))
    dV = ops.gemm_tn_batched(P, do, Z, Lp, Lp, DV, Lp, DV, Lp * Lp, Lp * DV) if need_dv else None
    dP = torch.empty((Z, Lp, Lp), dtype=dt, device=dev)
    ops.gemm_batched(do, v, dP, Z, Lp, Lp, DV, DV, DV, Lp, Lp * DV, Lp * DV if v.shape[0] == Z else 0, Lp * Lp)
    dS = ops.softmax_bwd_rows(P, dP.view(Z * Lp, Lp), L, scale, out=dP.view(Z * Lp, Lp))
    if dscore is not None:
        assert L == Lp
        ops.axpby(dS, dscore.reshape(Z * Lp, Lp), 1.0, scale, out=dS)
    kT = ops.transpose_pad(k, Lp)                                                       # [Z, D, Lp]
    dq = torch.empty_like(q)
    ops.gemm_batched(dS, kT, dq, Z, Lp, D, Lp, Lp, Lp, D, Lp * Lp, D * Lp, Lp * D)
    dK = ops.gemm_tn_batched(dS, q, Z, Lp, Lp, D, Lp, D, Lp * Lp, Lp * D)
    return dq, dK, dV


WATTN_BWD_FUSED = True      # emip_window_attention (+ log-sum-exp) / emip_window_attention_bwd in the bf16 training step


class WindowAttentionFn(Function):
    """Swin split-window single-head attention (transformer.py:46-105).  k / v of frame b are those of frame (b + kv_rot) mod 2B
    (cross attention between the frames of a pair: no swapped copies).  bf16 at 128 channels: the dedicated forward kernel,
    which also leaves every query's log-sum-exp, and the three-launch fused backward (wattn_bwd.hip).  Otherwise: the generic
    fused forward through the index tables; backward gathers each window into a dense batch and runs the unfused formulation."""

    @staticmethod
    def forward(ctx, q, k, v, h, w, shift, splits, kv_rot=0):
        from .model.EMIP_short.motion.gmflow.tables import window_tables
        B2, n, C = q.shape
        rows_t, gid = window_tables(h, w, splits, shift, q.device)
        L = (h // splits) * (w // splits)
        out = torch.empty((B2, n, C), dtype=q.dtype, device=q.device)
        fused = WATTN_BWD_FUSED and q.dtype == torch.bfloat16 and C == 128 and 64 <= L <= 512
        if fused:
            lse = torch.empty((B2, n), dtype=torch.float32, device=q.device)
            ops.window_attention(q, k, v, out, rows_t, gid if shift else None, n, C ** -0.5, kv_rot, lse=lse)
            ctx.save_for_backward(q, k, v, out, lse)
        else:
            if kv_rot:
                k, v = torch.roll(k, -kv_rot, 0), torch.roll(v, -kv_rot, 0)
            ops.attention(q, k, v, out, batch=B2, heads=1, nwin=splits * splits, Lq=L, Lk=L, D=C, DV=C, q_bs=n * C,
                          k_bs=n * C, v_bs=n * C, o_bs=n * C, ldq=C, ldk=C, ldv=C, ldo=C, q_rows=rows_t, k_rows=rows_t,
                          q_gid=gid if shift else None, k_gid=gid if shift else None, scale=C ** -0.5)
            ctx.save_for_backward(q, k, v)
        ctx.cfg = (h, w, shift, splits, kv_rot, fused)
        return out

    @staticmethod
    def backward(ctx, do):
        from .model.EMIP_short.motion.gmflow.tables import window_tables
        h, w, shift, splits, kv_rot, fused = ctx.cfg
        nwin = splits * splits
        rows_t, gid = window_tables(h, w, splits, shift, do.device)
        if fused:
            q, k, v, out, lse = ctx.saved_tensors
            n, C = q.shape[1], q.shape[2]
            dq, dk, dv = ops.window_attention_bwd(q, k, v, out, do.contiguous(), lse, rows_t, gid if shift else None, n,
                                                  C ** -0.5, kv_rot)
            return dq, dk, dv, None, None, None, None, None
        q, k, v = ctx.saved_tensors
        B2, n, C = q.shape
        L = n // nwin
        Lp = _round_up(L, 8)
        gq = None
        if shift:
            gq = torch.zeros((nwin, Lp), dtype=torch.int32, device=q.device)
            gq[:, :L] = gid
        g = lambda t: ops.window_rows(t.contiguous(), rows_t, B2, nwin, L, Lp, n, C)
        dq, dK, dV = dense_attention_bwd(g(q), g(k), g(v), g(do), L, C ** -0.5, gq, gid if shift else None, nwin)
        s = lambda t: ops.window_rows(t, rows_t, B2, nwin, L, Lp, n, C, scatter=True)
        dk, dv = s(_to_act(dK, q.dtype)), s(_to_act(dV, q.dtype))
        if kv_rot:
            dk, dv = torch.roll(dk, kv_rot, 0), torch.roll(dv, kv_rot, 0)
        return s(dq), dk, dv, None, None, None, None, None


MATCH_BWD_FUSED = True      # emip_match (+ log-sum-exp) / emip_match_bwd in the bf16 training step


def _match_fused(t):
    return MATCH_BWD_FUSED and ops.match_eligible(t)


class GlobalMatchFn(Function):
    """Global correlation softmax in both directions (matching.py:8-41).  tokens [2B, n, C] (frame 0 | frame 1) ->
    (expected positions f32 with (x, y) in columns 0..1: [2B, n, 2] on the fused path, [2B, n, 32] otherwise; scaled correlation
    [B, n(src), n(tgt)]).  bf16 at 128 channels: emip_match forward (one launch, leaves the log-sum-exp) and emip_match_bwd."""

    @staticmethod
    def forward(ctx, c0, grid, W, want_corr=True):
        B2, n, C = c0.shape
        B = B2 // 2
        # want_corr=False (fused path only): nobody convolves the volume (ConvCorr0Fn works from the tokens), it is not written
        corr = torch.empty((B, n, n), dtype=c0.dtype, device=c0.device) if (want_corr or not _match_fused(c0)) else None
        if _match_fused(c0):
            lse = torch.empty((B2, n), dtype=torch.float32, device=c0.device)
            o = ops.match(c0, c0, W, C ** -0.5, scores=corr, kv_rot=B, sub_grid=False, lse=lse)
            ctx.save_for_backward(c0, o, lse)
            ctx.cfg = (W, True)
            return o, corr
        o = torch.empty((B2, n, 32), dtype=torch.float32, device=c0.device)
        common = dict(batch=B, heads=1, nwin=1, Lq=n, Lk=n, D=C, DV=32, q_bs=n * C, k_bs=n * C, v_bs=0, o_bs=n * 32,
                      ldq=C, ldk=C, ldv=32, ldo=32, scale=C ** -0.5)
        ops.attention(c0[:B], c0[B:], grid, o[:B], scores=corr, s_bs=n * n, lds=n, **common)
        ops.attention(c0[B:], c0[:B], grid, o[B:], **common)
        ctx.save_for_backward(c0, grid)
        ctx.cfg = (W, False)
        return o, corr

    @staticmethod
    def backward(ctx, do, dcorr):
        W, fused = ctx.cfg
        if fused:
            c0, o, lse = ctx.saved_tensors
            B2, n, C = c0.shape
            dtok, _ = ops.match_bwd(c0, c0, W, C ** -0.5, o, do.contiguous(), lse,
                                    dscores=dcorr.contiguous() if dcorr is not None else None, kv_rot=B2 // 2,
                                    sub_grid=False, accum=True)
            return dtok, None, None, None
        c0, grid = ctx.saved_tensors
        B2, n, C = c0.shape
        B = B2 // 2
        assert n % 8 == 0
        dt = c0.dtype
        doT = _to_act(do.contiguous(), dt)
        f0, f1 = c0[:B], c0[B:]
        scale = C ** -0.5
        dq_f, dk_f, _ = dense_attention_bwd(f0, f1, grid[None], doT[:B], n, scale,
                                            dscore=dcorr.contiguous() if dcorr is not None else None, need_dv=False)
        dq_b, dk_b, _ = dense_attention_bwd(f1, f0, grid[None], doT[B:], n, scale, need_dv=False)
        dc0 = torch.empty_like(c0)
        ops.axpby(dq_f, _to_act(dk_b, dt), 1.0, 1.0, out=dc0[:B])
        ops.axpby(dq_b, _to_act(dk_f, dt), 1.0, 1.0, out=dc0[B:])
        return dc0, None, None, None


class FlowPropFn(Function):
    """Flow propagation by feature self-similarity (transformer.py:485-533): softmax(q k^T / sqrt(C)) flow; the flow
    operand is detached in the reference (gmflow.py:139).  Output f32 [N, n, 2] on the fused path, [N, n, 32] otherwise."""

    @staticmethod
    def forward(ctx, q, k, flow):
        N, n, C = q.shape
        if _match_fused(q) and _match_fused(k):
            v = flow.reshape(N, n, 2).contiguous()
            lse = torch.empty((N, n), dtype=torch.float32, device=q.device)
            o = ops.match(q, k, 1, C ** -0.5, v=v, sub_grid=False, lse=lse)
            ctx.save_for_backward(q, k, v, o, lse)
            ctx.fused = True
            return o
        v = torch.empty((N, n, 32), dtype=q.dtype, device=q.device)
        ops.copy_cols(flow.view(N * n, 2), 0, 2, v.view(N * n, 32), 0, 32)
        o = torch.empty((N, n, 32), dtype=torch.float32, device=q.device)
        ops.attention(q, k, v, o, batch=N, heads=1, nwin=1, Lq=n, Lk=n, D=C, DV=32, q_bs=n * C, k_bs=n * C,
                      v_bs=n * 32, o_bs=n * 32, ldq=C, ldk=C, ldv=32, ldo=32, scale=C ** -0.5)
        ctx.save_for_backward(q, k, v)
        ctx.fused = False
        return o

    @staticmethod
    def backward(ctx, do):
        if ctx.fused:
            q, k, v, o, lse = ctx.saved_tensors
            dq, dk = ops.match_bwd(q, k, 1, q.shape[2] ** -0.5, o, do.contiguous(), lse, v=v, sub_grid=False)
            return dq, dk, None
        q, k, v = ctx.saved_tensors
        N, n, C = q.shape
        dq, dk, _ = dense_attention_bwd(q, k, v, _to_act(do.contiguous(), q.dtype), n, C ** -0.5, need_dv=False)
        return dq, _to_act(dk, q.dtype), None


class CorrespToFlowFn(Function):
    """columns 0..1 of the expectation (minus the pixel grid when sub_grid) -> flow f32 [N,h,w,2]"""

    @staticmethod
    def forward(ctx, o, N, h, w, sub_grid):
        ctx.cols = o.shape[-1]
        return ops.corresp_to_flow(o, N, h, w, sub_grid)

    @staticmethod
    def backward(ctx, dflow):
        N, h, w, _ = dflow.shape
        if ctx.cols == 2:                                   # the fused matching path hands over [N, n, 2]: the gradient as it is
            return dflow.contiguous().view(N, h * w, 2), None, None, None, None
        do = torch.empty((N, h * w, 32), dtype=torch.float32, device=dflow.device)
        ops.copy_cols(dflow.contiguous().view(-1, 2), 0, 2, do.view(-1, 32), 0, 32)
        return do, None, None, None, None


class FlowToActFn(Function):
    """flow f32 [N,h,w,2] -> activation dtype [N,h,w,8] (zero padded): the flow channels of the upsampler input"""

    @staticmethod
    def forward(ctx, flow, dt):
        out = torch.empty(flow.shape[:-1] + (8,), dtype=dt, device=flow.device)
        ops.copy_cols(flow, 0, 2, out, 0, 8)
        return out

    @staticmethod
    def backward(ctx, dy):
        g = torch.empty(dy.shape[:-1] + (2,), dtype=torch.float32, device=dy.device)
        ops.copy_cols(dy.contiguous(), 0, 2, g, 0)
        return g, None


class ConvexUpsampleFn(Function):
    """gmflow.py:64-77"""

    @staticmethod
    def forward(ctx, logits, flow):
        ctx.save_for_backward(logits, flow)
        return ops.convex_upsample(logits, flow)

    @staticmethod
    def backward(ctx, dy):
        logits, flow = ctx.saved_tensors
        dl, df = ops.convex_upsample_bwd(logits, flow, dy.contiguous())
        return dl, df


class ActFn(Function):
    """standalone ReLU / exact GELU"""

    @staticmethod
    def forward(ctx, x, act):
        y = ops.act_fwd(x, act)
        ctx.save_for_backward(x if act == ops.ACT_GELU else y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (t,) = ctx.saved_tensors
        dy = dy.contiguous()
        return (ops.gelu_bwd(t, dy) if ctx.act == ops.ACT_GELU else ops.relu_bwd(t, dy)), None


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.eltwise(a, b, 2)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class BcastAddFn(Function):
    """a[r] + table[r % period] (the window position encoding)"""

    @staticmethod
    def forward(ctx, a, table, period):
        return ops.eltwise(a, table, 3, period=period)

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None


class DropPathAddFn(Function):
    """residual + scale[b] * branch: stochastic depth on the residual branch (timm DropPath as used by pvt_v2.py:
    per-sample Bernoulli(keep) / keep).  scale: f32 [B, C] (the per-sample value repeated over channels), scale_m1 = scale - 1."""

    @staticmethod
    def forward(ctx, residual, branch, scale, scale_m1, rows_per_sample):
        out = torch.empty_like(residual)
        ops.colscale_add(residual, branch, scale, scale.shape[1], rows_per_sample, out)
        ctx.save_for_backward(scale_m1)
        ctx.rps = rows_per_sample
        return out

    @staticmethod
    def backward(ctx, dy):
        (scale_m1,) = ctx.saved_tensors
        dy = dy.contiguous()
        db = torch.empty_like(dy)
        ops.colscale_add(dy, dy, scale_m1, scale_m1.shape[1], ctx.rps, db)        # dy + (s - 1) dy = s dy
        return dy, db, None, None, None


class MemoryReadFn(Function):
    """EMIP-long memory read (LTM.py:49-68): softmax over the T*n stored positions of K_mem^T k_q / sqrt(C), times V_mem.
    q [S,n,C], keys / values [S,T*n,C] -> [S,n,C].  Backward: unfused formulation on the rectangular score matrix; only
    the rows of keys / values that belong to the current frame carry gradient in train_long.py (the older frames were
    detached), but all of them are returned -- autograd drops what nobody asked for."""

    @staticmethod
    def forward(ctx, q, keys, values):
        S, n, C = q.shape
        Lk = keys.shape[1]
        out = torch.empty((S, n, C), dtype=q.dtype, device=q.device)
        ops.attention(q, keys, values, out, batch=S, heads=1, nwin=1, Lq=n, Lk=Lk, D=C, DV=C, q_bs=n * C, k_bs=Lk * C,
                      v_bs=Lk * C, o_bs=n * C, ldq=C, ldk=C, ldv=C, ldo=C, scale=C ** -0.5)
        ctx.save_for_backward(q, keys, values)
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        do = do.contiguous()
        S, n, C = q.shape
        Lk = k.shape[1]
        Lp = _round_up(Lk, 8)
        dt, dev = q.dtype, q.device
        scale = C ** -0.5
        Sm = torch.empty((S, n, Lp), dtype=dt, device=dev)
        ops.gemm_batched(q, k, Sm, S, n, Lk, C, C, C, Lp, n * C, Lk * C, n * Lp)
        P = ops.softmax_rows(Sm.view(S * n, Lp), Lk, scale, out=Sm.view(S * n, Lp))
        dV = ops.gemm_tn_batched(P, do, S, n, Lp, C, Lp, C, n * Lp, n * C)                 # f32 [S,Lp,C]
        dP = torch.empty((S, n, Lp), dtype=dt, device=dev)
        ops.gemm_batched(do, v, dP, S, n, Lk, C, C, C, Lp, n * C, Lk * C, n * Lp)
        dS = ops.softmax_bwd_rows(P, dP.view(S * n, Lp), Lk, scale, out=dP.view(S * n, Lp))
        kT = ops.transpose_pad(k, Lp)                                                      # [S,C,Lp]
        dq = torch.empty_like(q)
        ops.gemm_batched(dS, kT, dq, S, n, C, Lp, Lp, Lp, C, n * Lp, C * Lp, n * C)
        dK = ops.gemm_tn_batched(dS, q, S, n, Lp, C, Lp, C, n * Lp, n * C)                 # f32 [S,Lp,C]
        return dq, _to_act(dK[:, :Lk].contiguous(), dt), _to_act(dV[:, :Lk].contiguous(), dt)
