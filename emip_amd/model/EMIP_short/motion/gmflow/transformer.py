"""GMFlow feature transformer and flow-propagation attention on MI355X kernels.

State-dict layout follows /root/reference/model/EMIP_short/motion/gmflow/transformer.py:
layers.{i}.{self_attn,cross_attn_ffn}.{q_proj,k_proj,v_proj,merge,norm1[,mlp.0,mlp.2,norm2],adaptor_fc1,adaptor_fc2}
and feature_flow_attn.{q_proj,k_proj}.  Design differences from the reference's op stream:
  * both frames run as one [2B, 1936, 128] token batch; q/k/v of the self-attention are ONE GEMM (N = 384);
  * the cross-attention target is "the other half of the batch at block start": its k/v projection is written
    with the halves swapped, so no concatenated/rolled copies of the features are ever made;
  * window split, cyclic shift and the -100 shift mask are index tables of the fused attention kernel;
  * torch.cat([source, message]) feeding the FFN is a two-source K loop inside the GEMM.
"""
import torch
import torch.nn as nn

from ..... import ops
from .....autograd import (ActFn, AddFn, ConcatFn, CorrespToFlowFn, FanOutFn, FlowPropFn, LayerNormFn, Linear2Fn, LinearFn,
                           WindowAttentionFn)
from .....nn_base import EmipModule, f32, pack_linear
from .tables import window_tables


class TransformerLayer(EmipModule):
    """Parameter holder + packing for one attention layer (transformer.py:108-211)."""

    def __init__(self, d_model=256, nhead=1, attention_type='swin', no_ffn=False, ffn_dim_expansion=4,
                 with_shift=False, **kwargs):
        super().__init__()
        self.dim, self.nhead, self.no_ffn, self.with_shift = d_model, nhead, no_ffn, with_shift
        self.attention_type = attention_type
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.norm1 = nn.LayerNorm(d_model)
        if not no_ffn:
            c = d_model * 2
            self.mlp = nn.Sequential(nn.Linear(c, c * ffn_dim_expansion, bias=False), nn.GELU(),
                                     nn.Linear(c * ffn_dim_expansion, d_model, bias=False))
            self.norm2 = nn.LayerNorm(d_model)
        # the reference's unused adaptor (transformer.py:148-151): state_dict entries only
        self.adaptor_fc1 = nn.Linear(128, 32)
        self.adaptor_fc2 = nn.Linear(32, 128)

    def weights(self):
        dt = self.cdtype

        def build(q, k, v, m, g, b):
            return dict(qkv=torch.cat([q, k, v], 0).detach().to(dt).contiguous(), q=pack_linear(q, dt),
                        kv=torch.cat([k, v], 0).detach().to(dt).contiguous(), merge=pack_linear(m, dt),
                        n1=(f32(g), f32(b)),
                        mfrag=(ops.wattn_merge_pack(m, dt) if (dt == torch.bfloat16 and tuple(m.shape) == (128, 128)) else None),
                        qfrag=(ops.wattn_q_pack(q, dt) if (dt == torch.bfloat16 and tuple(q.shape) == (128, 128)) else None))
        w = self.packed("attn", (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.merge.weight,
                                 self.norm1.weight, self.norm1.bias), build)
        if not self.no_ffn:
            f = self.packed("ffn", (self.mlp[0].weight, self.mlp[2].weight, self.norm2.weight, self.norm2.bias),
                            lambda a, b, g, be: dict(m0=pack_linear(a, dt), m2=pack_linear(b, dt), n2=(f32(g), f32(be)),
                                                     ffn=(ops.ffn_block_packs(a, b, dt) if (dt == torch.bfloat16 and
                                                          tuple(a.shape) == (1024, 256)) else None)))
            w = dict(w, **f)
        return w


    def tpacks(self, name):
        """(W packed [N, K], W^T packed [K, N]) of the bias-free Linear `name` in the activation dtype"""
        mod = self.mlp[int(name[3:])] if name.startswith("mlp") else getattr(self, name)
        dt = self.cdtype
        return self.packed("t_" + name, (mod.weight,), lambda a: (pack_linear(a, dt), a.detach().t().to(dt).contiguous()))

    def frozen(self, *names):
        return not any((self.mlp[int(n[3:])] if n.startswith("mlp") else getattr(self, n)).weight.requires_grad for n in names)

    def lin(self, name, x, res=None):
        """differentiable bias-free Linear through the frozen-or-not weight `name`"""
        mod = self.mlp[int(name[3:])] if name.startswith("mlp") else getattr(self, name)
        dt = self.cdtype
        wp, wpt = self.packed("t_" + name, (mod.weight,), lambda a: (pack_linear(a, dt),
                                                                     a.detach().t().to(dt).contiguous()))
        return LinearFn.apply(x, mod.weight, mod.bias, res, wp, wpt)




class TransformerBlock(EmipModule):
    """self attention, then cross attention + FFN (transformer.py:348-401)."""

    def fused_in(self):
        """[q | k | v of the self attention | k | v of the cross attention] as ONE [5C, C] weight: all five projections read
        the block's input tokens (the cross attention's source is the other frame AS IT IS AT BLOCK START), so they are one
        GEMM over the whole batch instead of three launches"""
        dt = self.cdtype
        sa, ca = self.self_attn, self.cross_attn_ffn
        return self.packed("in5", (sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight, ca.k_proj.weight, ca.v_proj.weight),
                           lambda *ws_: torch.cat([t.detach() for t in ws_], 0).to(dt).contiguous())

    def fused_in4(self):
        """[k | v of the self attention | k | v of the cross attention] as one [4C, C] weight: the form used when both q projections
        run inside their attention launches (emip_window_attention_merge with wq_pack)"""
        dt = self.cdtype
        sa, ca = self.self_attn, self.cross_attn_ffn
        return self.packed("in4", (sa.k_proj.weight, sa.v_proj.weight, ca.k_proj.weight, ca.v_proj.weight),
                           lambda *ws_: torch.cat([t.detach() for t in ws_], 0).to(dt).contiguous())

    def __init__(self, d_model=256, nhead=1, attention_type='swin', ffn_dim_expansion=4, with_shift=False, **kw):
        super().__init__()
        self.self_attn = TransformerLayer(d_model, nhead, attention_type, no_ffn=True,
                                          ffn_dim_expansion=ffn_dim_expansion, with_shift=with_shift)
        self.cross_attn_ffn = TransformerLayer(d_model, nhead, attention_type, ffn_dim_expansion=ffn_dim_expansion,
                                               with_shift=with_shift)


def _window_attention(q, k, v, B2, h, w, C, ldq, ldk, ldv, shift, splits, kv_rot=0):
    """q/k/v: views with row strides ld*; returns message [B2, h*w, C].  kv_rot: keys / values of batch element b come from
    element (b + kv_rot) mod B2 (cross attention: the other frame of the pair)"""
    rows, gid = window_tables(h, w, splits, shift, q.device)
    L = (h // splits) * (w // splits)
    out = torch.empty((B2, h * w, C), dtype=q.dtype, device=q.device)
    n = h * w
    if q.dtype == torch.bfloat16 and C == 128 and 64 <= L <= 512:
        return ops.window_attention(q[..., :C], k[..., :C], v[..., :C], out, rows, gid if shift else None, n, C ** -0.5, kv_rot)
    ops.attention(q, k, v, out, batch=B2, heads=1, nwin=splits * splits, Lq=L, Lk=L, D=C, DV=C, q_bs=n * ldq,
                  k_bs=n * ldk, v_bs=n * ldv, o_bs=n * C, ldq=ldq, ldk=ldk, ldv=ldv, ldo=C, q_rows=rows, k_rows=rows,
                  q_gid=gid if shift else None, k_gid=gid if shift else None, scale=C ** -0.5, kv_rot=kv_rot)
    return out


TRAIN_FANOUT = True     # training path: FanOutFn / Linear2Fn for the frozen projections (46 autograd adds and 24 concat / split copies per step)
WATTN_QPROJ = True      # ... and the q projection in its prologue (the fused input GEMM then makes k | v | k' | v' only)
WATTN_MERGE = True      # emip_window_attention_merge: attention + merge + norm1 (+ residual) in one launch (bf16 inference)
FFN_BLOCK = True        # emip_ffn_block: mlp[0] + GELU + mlp[2] + norm2 + residual in one launch (bf16 inference)


class FeatureTransformer(EmipModule):
    """transformer.py:404-482"""

    def __init__(self, num_layers=6, d_model=128, nhead=1, attention_type='swin', ffn_dim_expansion=4, **kwargs):
        super().__init__()
        assert attention_type == 'swin' and nhead == 1 and d_model == 128
        self.attention_type, self.d_model, self.nhead = attention_type, d_model, nhead
        self.layers = nn.ModuleList([
            TransformerBlock(d_model=d_model, nhead=nhead, attention_type=attention_type,
                             ffn_dim_expansion=ffn_dim_expansion, with_shift=(i % 2 == 1))
            for i in range(num_layers)])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def run_train(self, c0, h, w, attn_num_splits=2):
        """differentiable variant of run(): same op stream, out of place, every step a Function over HIP kernels"""
        B2, n, C = c0.shape
        B = B2 // 2
        for blk in self.layers:
            sa, ca = blk.self_attn, blk.cross_attn_ffn
            shift = sa.with_shift
            # keys / values of the cross attention: the OTHER frame's tokens as they are at block start, read in place from the
            # other half of the batch (kv_rot = B), like the inference path
            fused = TRAIN_FANOUT and sa.frozen("q_proj", "k_proj", "v_proj") and ca.frozen("q_proj", "k_proj", "v_proj", "mlp0")
            if fused:
                # frozen weights (train.py:340-342: the case the training step runs): the block input's five projections and its
                # skip connection leave ONE node, whose backward adds the six gradients in GEMM epilogues (autograd.FanOutFn)
                c0, q, k, v, kx, vx = FanOutFn.apply(c0, sa.tpacks("q_proj"), sa.tpacks("k_proj"), sa.tpacks("v_proj"),
                                                     ca.tpacks("k_proj"), ca.tpacks("v_proj"))
            else:
                kx, vx = ca.lin("k_proj", c0), ca.lin("v_proj", c0)
                q, k, v = sa.lin("q_proj", c0), sa.lin("k_proj", c0), sa.lin("v_proj", c0)
            msg = WindowAttentionFn.apply(q, k, v, h, w, shift, attn_num_splits)
            msg = LayerNormFn.apply(sa.lin("merge", msg), sa.norm1.weight, sa.norm1.bias, sa.norm1.eps)
            c1 = AddFn.apply(c0, msg)
            if fused:
                c1, q1 = FanOutFn.apply(c1, ca.tpacks("q_proj"))
            else:
                q1 = ca.lin("q_proj", c1)
            msg = WindowAttentionFn.apply(q1, kx, vx, h, w, shift, attn_num_splits, B)
            msg = LayerNormFn.apply(ca.lin("merge", msg), ca.norm1.weight, ca.norm1.bias, ca.norm1.eps)
            if fused:
                hid = ActFn.apply(Linear2Fn.apply(c1, msg, *ca.tpacks("mlp0")), ops.ACT_GELU)      # no concatenated copy
            else:
                hid = ActFn.apply(ca.lin("mlp0", ConcatFn.apply(None, c1, msg)), ops.ACT_GELU)
            msg = LayerNormFn.apply(ca.lin("mlp2", hid), ca.norm2.weight, ca.norm2.bias, ca.norm2.eps)
            c0 = AddFn.apply(c1, msg)
        return c0

    def run(self, c0, h, w, attn_num_splits=2):
        """c0: [2B, h*w, C] tokens, frame-0 features in the first half, frame-1 in the second.  In place."""
        if torch.is_grad_enabled() and c0.requires_grad:
            return self.run_train(c0, h, w, attn_num_splits)
        B2, n, C = c0.shape
        B = B2 // 2
        assert n == h * w and C == self.d_model
        dt = c0.dtype
        for blk in self.layers:
            ws, wc = blk.self_attn.weights(), blk.cross_attn_ffn.weights()
            shift = blk.self_attn.with_shift
            # one GEMM: q | k | v of the self attention and k | v of the cross attention, whose source is the OTHER frame as it
            # is at block start -- read in place from the other half of the batch (kv_rot = B)
            fused = WATTN_MERGE and ws.get("mfrag") is not None and 64 <= (h // attn_num_splits) * (w // attn_num_splits) <= 512
            qin = fused and WATTN_QPROJ and ws.get("qfrag") is not None and wc.get("qfrag") is not None and wc.get("mfrag") is not None \
                and (h // attn_num_splits) * (w // attn_num_splits) > 128
            if qin:
                # both q projections run in the prologue of their attention launch: k | v | k' | v' only
                big = ops.gemm(c0, blk.fused_in4())                              # [2B, n, 4C]
                rows_t, gid_t = window_tables(h, w, attn_num_splits, shift, c0.device)
                gm = gid_t if shift else None
                ops.window_attention_merge(c0, big[..., :C], big[..., C:2 * C], c0, rows_t, gm, n, C ** -0.5, ws["mfrag"],
                                           ws["n1"][0], ws["n1"][1], blk.self_attn.norm1.eps, res=c0, wq_pack=ws["qfrag"])
                msg = torch.empty((B2, n, C), dtype=dt, device=c0.device)
                ops.window_attention_merge(c0, big[..., 2 * C:3 * C], big[..., 3 * C:], msg, rows_t, gm, n, C ** -0.5, wc["mfrag"],
                                           wc["n1"][0], wc["n1"][1], blk.cross_attn_ffn.norm1.eps, kv_rot=B, wq_pack=wc["qfrag"])
                if FFN_BLOCK and wc.get("ffn") is not None:
                    ops.ffn_block(c0, msg, wc["ffn"][0], wc["ffn"][1], wc["n2"][0], wc["n2"][1], blk.cross_attn_ffn.norm2.eps,
                                  res=c0, out=c0)
                else:
                    ops.gemm_ln_out(ops.gemm(c0, wc["m0"], a2=msg, act=ops.ACT_GELU), wc["m2"], wc["n2"][0], wc["n2"][1],
                                    blk.cross_attn_ffn.norm2.eps, res=c0, out=c0)
                continue
            big = ops.gemm(c0, blk.fused_in())                                   # [2B, n, 5C]
            ck, cv, ldc5, rot = big[..., 3 * C:], big[..., 4 * C:], 5 * C, B
            ldb = big.shape[-1]
            # ---- self attention (no FFN): c0 += LN(merge(attn))
            if fused:      # attention + merge + norm1 + residual in one launch: c0 += LN(merge(attn))
                rows_t, gid_t = window_tables(h, w, attn_num_splits, shift, c0.device)
                ops.window_attention_merge(big[..., :C], big[..., C:2 * C], big[..., 2 * C:3 * C], c0, rows_t,
                                           gid_t if shift else None, n, C ** -0.5, ws["mfrag"], ws["n1"][0], ws["n1"][1],
                                           blk.self_attn.norm1.eps, res=c0)
            else:
                msg = _window_attention(big, big[..., C:], big[..., 2 * C:], B2, h, w, C, ldb, ldb, ldb, shift,
                                        attn_num_splits)
                ops.gemm_ln_out(msg, ws["merge"], ws["n1"][0], ws["n1"][1], blk.self_attn.norm1.eps, res=c0, out=c0)  # c0 += LN(merge)
            # ---- cross attention + FFN
            q = ops.gemm(c0, wc["q"])
            if fused and wc.get("mfrag") is not None:      # msg = LN(merge(cross attention)), one launch
                msg = torch.empty((B2, n, C), dtype=dt, device=c0.device)
                ops.window_attention_merge(q, ck[..., :C], cv[..., :C], msg, rows_t, gid_t if shift else None, n, C ** -0.5,
                                           wc["mfrag"], wc["n1"][0], wc["n1"][1], blk.cross_attn_ffn.norm1.eps, kv_rot=rot)
            else:
                msg = _window_attention(q, ck, cv, B2, h, w, C, C, ldc5, ldc5, shift, attn_num_splits, kv_rot=rot)
                msg = ops.gemm_ln_out(msg, wc["merge"], wc["n1"][0], wc["n1"][1], blk.cross_attn_ffn.norm1.eps)
            if FFN_BLOCK and wc.get("ffn") is not None:      # the whole FFN in one launch: the 1024-wide hidden tensor stays on the CU
                ops.ffn_block(c0, msg, wc["ffn"][0], wc["ffn"][1], wc["n2"][0], wc["n2"][1], blk.cross_attn_ffn.norm2.eps,
                              res=c0, out=c0)
            else:
                hid = ops.gemm(c0, wc["m0"], a2=msg, act=ops.ACT_GELU)
                ops.gemm_ln_out(hid, wc["m2"], wc["n2"][0], wc["n2"][1], blk.cross_attn_ffn.norm2.eps, res=c0, out=c0)  # c0 += LN(mlp)
        return c0


class FeatureFlowAttention(EmipModule):
    """Flow propagation (transformer.py:485-533); note key = k_proj(q_proj(feature)) as in the reference."""

    def __init__(self, in_channels, **kwargs):
        super().__init__()
        self.q_proj = nn.Linear(in_channels, in_channels)
        self.k_proj = nn.Linear(in_channels, in_channels)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def run(self, feat, flow, h, w):
        """feat [N, h*w, C] tokens, flow f32 [N, h, w, 2] -> propagated flow f32 [N, h, w, 2]"""
        dt = feat.dtype
        N, n, C = feat.shape
        if torch.is_grad_enabled() and feat.requires_grad:
            packs = self.packed("wt", (self.q_proj.weight, self.k_proj.weight), lambda a, c: (
                pack_linear(a, dt), a.detach().t().to(dt).contiguous(), pack_linear(c, dt),
                c.detach().t().to(dt).contiguous()))
            q = LinearFn.apply(feat, self.q_proj.weight, self.q_proj.bias, None, packs[0], packs[1])
            k = LinearFn.apply(q, self.k_proj.weight, self.k_proj.bias, None, packs[2], packs[3])
            return CorrespToFlowFn.apply(FlowPropFn.apply(q, k, flow.detach()), N, h, w, False)
        wq, bq, wk, bk = self.packed("w", (self.q_proj.weight, self.q_proj.bias, self.k_proj.weight,
                                           self.k_proj.bias),
                                     lambda a, b, c, d: (pack_linear(a, dt), f32(b), pack_linear(c, dt), f32(d)))
        q = ops.gemm(feat, wq, bias=bq)
        k = ops.gemm(q, wk, bias=bk)
        if dt == torch.bfloat16 and C == 128:       # the flow itself is the value: one launch, no padded value / output buffers
            return ops.match(q, k, w, C ** -0.5, v=flow.reshape(N, n, 2), sub_grid=False).view(N, h, w, 2)
        v = torch.empty((N, n, 32), dtype=dt, device=feat.device)
        ops.copy_cols(flow.view(N * n, 2), 0, 2, v.view(N * n, 32), 0, 32)
        o = torch.empty((N, n, 32), dtype=torch.float32, device=feat.device)
        ops.attention(q, k, v, o, batch=N, heads=1, nwin=1, Lq=n, Lk=n, D=C, DV=32, q_bs=n * C, k_bs=n * C,
                      v_bs=n * 32, o_bs=n * 32, ldq=C, ldk=C, ldv=32, ldo=32, scale=C ** -0.5)
        return ops.corresp_to_flow(o, N, h, w, False)
