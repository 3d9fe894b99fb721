"""GMFlow motion stream (frozen in EMIP) on MI355X kernels.

Constructor and state_dict follow /root/reference/model/EMIP_short/motion/gmflow/gmflow.py
(backbone.*, transformer.*, feature_flow_attn.*, upsampler.{0,2}.*).  The forward keeps both flow
directions in one 2B batch.  Global matching (matching.py:8-41) is two launches of the fused attention
kernel with V = the pixel grid: forward direction (which also writes the raw correlation volume that
CoUpdater.conv_corr consumes, in its native [B][src][tgt] layout) and backward direction (roles of the
two feature maps swapped = the transposed correlation, never materialised).
"""
import torch
import torch.nn as nn

from ..... import ops
from .....autograd import (ActFn, BcastAddFn, BilinearPlanarFn, ConcatFn, ConvFn, ConvexUpsampleFn, CorrespToFlowFn,
                           FlowToActFn, GlobalMatchFn, LinearFn)
from .....nn_base import EmipModule, f32, pack_conv, pack_linear, to_cl
from .backbone import CNNEncoder
from .tables import grid_values, position_table
from .transformer import FeatureFlowAttention, FeatureTransformer


class GMFlow(EmipModule):
    def __init__(self, num_scales=1, upsample_factor=8, feature_channels=128, attention_type='swin',
                 num_transformer_layers=6, ffn_dim_expansion=4, num_head=1, args=None, **kwargs):
        super().__init__()
        assert num_scales == 1 and upsample_factor == 8
        self.num_scales, self.feature_channels, self.upsample_factor = num_scales, feature_channels, upsample_factor
        self.attention_type, self.num_transformer_layers = attention_type, num_transformer_layers
        self.backbone = CNNEncoder(output_dim=feature_channels, num_output_scales=num_scales)
        self.transformer = FeatureTransformer(num_layers=num_transformer_layers, d_model=feature_channels,
                                              nhead=num_head, attention_type=attention_type,
                                              ffn_dim_expansion=ffn_dim_expansion)
        self.feature_flow_attn = FeatureFlowAttention(in_channels=feature_channels)
        self.upsampler = nn.Sequential(nn.Conv2d(2 + feature_channels, 256, 3, 1, 1), nn.ReLU(inplace=True),
                                       nn.Conv2d(256, upsample_factor ** 2 * 9, 1, 1, 0))
        g = args['GMFlow']
        self.attn_splits_list, self.corr_radius_list = g['attn_splits_list'], g['corr_radius_list']
        self.prop_radius_list, self.pred_bidir_flow = g['prop_radius_list'], g['pred_bidir_flow']
        assert self.corr_radius_list == [-1] and self.prop_radius_list == [-1] and self.pred_bidir_flow, \
            "only the shipped configuration (global matching, bidirectional) is built"

    def run_train(self, ab, corr=True):
        """differentiable variant of run(): ab = prompted features of both frames [2B,h,w,C] (frame 1 | frame 2).
        The GMFlow weights are frozen (model.py:61-63), so only input gradients flow.  corr=False: the correlation volume is
        not written (the caller convolves it through its factors, self.last["tokens"]); None is returned for it."""
        dt = self.cdtype
        B2, h, w, C = ab.shape
        n = h * w
        splits = self.attn_splits_list[0]
        pos = position_table(h, w, C, splits, dt, ab.device)
        c0 = BcastAddFn.apply(ab.view(B2, n, C), pos, n)
        c0 = self.transformer.run_train(c0, h, w, splits)
        o, corr = GlobalMatchFn.apply(c0, grid_values(h, w, dt, ab.device), w, corr)
        flow = CorrespToFlowFn.apply(o, B2, h, w, True)
        preds = []
        if self.training:
            preds.append(BilinearPlanarFn.apply(flow, 0, 2, 8 * h, 8 * w, True, 8.0))
        flow = self.feature_flow_attn.run(c0, flow, h, w)
        cin = C + 8
        perm = list(range(2, 2 + C)) + [0, 1]
        up0, up2 = self.upsampler[0], self.upsampler[2]

        def build(p, r):
            wpad = torch.cat([p.detach()[:, perm], p.new_zeros(p.shape[0], cin - p.shape[1], 3, 3)], 1)
            r2 = r.detach().reshape(r.shape[0], -1)
            return (pack_conv(wpad, dt), pack_conv(wpad.flip(2, 3).permute(1, 0, 2, 3), dt), pack_linear(r2, dt),
                    r2.t().to(dt).contiguous())
        w0, w0d, w2, w2t = self.packed("up_t", (up0.weight, up2.weight), build)
        u = ConcatFn.apply(None, c0.view(B2, h, w, C), FlowToActFn.apply(flow, dt))
        u = ActFn.apply(ConvFn.apply(u, up0.weight, up0.bias, w0, w0d, 3, 1, 1, cin), ops.ACT_RELU)
        logits = LinearFn.apply(u, up2.weight, up2.bias, None, w2, w2t)
        preds.append(ConvexUpsampleFn.apply(logits, flow))
        self.last = dict(tokens=c0, flow_prop=flow)
        return preds, corr

    def run(self, a, b, flows=True, corr=True):
        """a, b: channels-last prompted features [B,h,w,C] of frame 1 / frame 2.
        Returns (flow predictions: list of planar f32 [2B,2,8h,8w], corr [B, h*w(src), h*w(tgt)]).
        corr=False: the raw correlation volume is not written (the caller works from self.last["tokens"]); returns None for it.
        flows=False (EMIP-long's short-term part: only the correlation volume is read): no backward-direction matching, no
        flow propagation, no upsampling; returns ([], corr)."""
        dt = self.cdtype
        B, h, w, C = a.shape
        n = h * w
        splits = self.attn_splits_list[0]
        c0 = torch.empty((2 * B, n, C), dtype=dt, device=a.device)
        pos = position_table(h, w, C, splits, dt, a.device)
        ops.eltwise(a.view(B, n, C), pos, 3, period=n, out=c0[:B])
        ops.eltwise(b.view(B, n, C), pos, 3, period=n, out=c0[B:])
        c0 = self.transformer.run(c0, h, w, splits)

        # ---- global correlation + softmax -> correspondence (both directions)
        want_corr = corr
        corr = torch.empty((B, n, n), dtype=dt, device=a.device) if want_corr else None
        if ops.match_eligible(c0):
            # one launch for both directions (emip_match): batch z < B = frame 1 against frame 2 with the raw correlation written
            # out, z >= B the reverse; the flow leaves the kernel as f32 [2B, h, w, 2]
            if not flows:
                if want_corr:
                    ops.match(c0[:B], c0[B:], w, C ** -0.5, scores=corr)
                self.last = dict(tokens=c0)
                return [], corr
            flow = ops.match(c0, c0, w, C ** -0.5, scores=corr, kv_rot=B).view(2 * B, h, w, 2)
            return self._finish(c0, flow, corr, B, h, w, C)
        grid = grid_values(h, w, dt, a.device)
        o = torch.empty((2 * B, n, 32), dtype=torch.float32, device=a.device)
        common = dict(batch=B, heads=1, nwin=1, Lq=n, Lk=n, D=C, DV=32, q_bs=n * C, k_bs=n * C, v_bs=0, o_bs=n * 32,
                      ldq=C, ldk=C, ldv=32, ldo=32, scale=C ** -0.5)
        if not flows:
            if want_corr:
                ops.attention(c0[:B], c0[B:], grid, o[:B], scores=corr, s_bs=n * n, lds=n, **common)
            self.last = dict(tokens=c0)
            return [], corr
        if want_corr:
            ops.attention(c0[:B], c0[B:], grid, o[:B], scores=corr, s_bs=n * n, lds=n, **common)
        else:
            ops.attention(c0[:B], c0[B:], grid, o[:B], **common)
        ops.attention(c0[B:], c0[:B], grid, o[B:], **common)
        flow = ops.corresp_to_flow(o, 2 * B, h, w, True)      # f32 [2B,h,w,2]
        return self._finish(c0, flow, corr, B, h, w, C)

    def _finish(self, c0, flow, corr, B, h, w, C):
        """flow propagation + convex upsampling behind the matching (gmflow.py:130-155)"""
        dt = self.cdtype
        preds = []
        if self.training:
            preds.append(ops.bilinear_planar(flow, 0, 2, 8 * h, 8 * w, True, mul=8.0))

        # ---- flow propagation + convex upsampling
        flow = self.feature_flow_attn.run(c0, flow, h, w)
        cin = C + 8                                           # [feature(128) | flow(2) | zero pad(6)]
        perm = list(range(2, 2 + C)) + [0, 1]                 # the reference concatenates (flow, feature)
        w0, b0, w2, b2 = self.packed(
            "up", (self.upsampler[0].weight, self.upsampler[0].bias, self.upsampler[2].weight,
                   self.upsampler[2].bias),
            lambda p, q, r, s: (pack_conv(p, dt, cin_pad=cin, perm=perm), f32(q),
                                pack_linear(r.reshape(r.shape[0], -1), dt), f32(s)))
        u = torch.empty((2 * B, h, w, cin), dtype=dt, device=c0.device)
        ops.copy_cols(c0, 0, C, u, 0)
        ops.copy_cols(flow, 0, 2, u, C, 8)
        u = ops.conv2d(u, w0, 3, 3, 1, 1, bias=b0, act=ops.ACT_RELU)
        logits = ops.gemm(u, w2, bias=b2)
        preds.append(ops.convex_upsample(logits, flow))
        self.last = dict(tokens=c0, flow_prop=flow)
        return preds, corr

    def forward(self, feature0_list, feature1_list):
        """Reference signature (gmflow.py:81-162): planar features in, (flow_fw list, flow_bw list, corr) out,
        corr viewed [B, tgt, h, w(src)] like matching.py:18-20."""
        a, b = feature0_list[0], feature1_list[0]
        B, C, h, w = a.shape
        dt = self.cdtype
        preds, corr = self.run(to_cl(a, dt), to_cl(b, dt))
        fw = [p[:B] for p in preds]
        bw = [p[B:] for p in preds]
        return fw, bw, corr.float().view(B, h, w, h * w).permute(0, 3, 1, 2)
