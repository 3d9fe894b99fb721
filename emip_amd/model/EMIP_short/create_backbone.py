"""Backbone wrapper, ConvBR, DimensionalReduction and the neighbour-connection decoder.

Module names follow /root/reference/model/EMIP_short/create_backbone.py so that state_dict keys match
(backbone.feat_net.pvtv2_en.*, backbone.decoder.NCD.*, decoder.*, dr{1,2,3}.reduce.{0,1}.{conv,bn}.*).
Forward passes run on channels-last tensors: a ConvBR is ONE implicit-GEMM launch (BatchNorm folded
into the packed weights in eval mode, ReLU in the epilogue); torch.cat is replaced by writing each
producer straight into its channel slice of the concat buffer.
"""
import torch
import torch.nn as nn

from ... import ops
from ...autograd import BilinearFn, BilinearPlanarFn, BNReluFn, ConcatFn, ConvFn, LinearFn, MulFn
from ...nn_base import EmipModule, f32, conv_dgrad_pack, fold_bn, pack_conv, to_cl, to_planar


# eval-mode ConvBR launches with few output tiles and a long K walk (11 x 11 x 512 -> 32: 31 tiles x 72 K tiles, 73 us) with K
# split inside the launch (emip_conv2d_ksplit; tests/test_ops_gpu.py compares both forms).  OFF: 1552 against 1562 pairs/s with
# three steps in flight (tools/flag_ab.py), see lib/pvt_v2.py SR_KSPLIT
KSPLIT = False


class ConvBR(EmipModule):
    """conv3x3 (no bias) + BatchNorm2d + ReLU (create_backbone.py:22-42)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__()
        assert dilation == 1
        self.k, self.s, self.p = kernel_size, stride, padding
        self.conv = nn.Conv2d(in_channel, out_channel, kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channel)
        nn.init.kaiming_normal_(self.conv.weight, a=1)

    def run(self, x, out=None):
        dt = self.cdtype
        if torch.is_grad_enabled():
            y = conv_bn_relu_autograd(self, self.conv, self.bn, x, self.k, self.s, self.p)
            if out is not None:
                raise RuntimeError("in-place concat outputs are an inference-only optimisation")
            return y
        if self.training:
            return conv_bn_train(self, self.conv, self.bn, x, self.k, self.s, self.p, out=out)
        w, b = self.packed("w", (self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean,
                                 self.bn.running_var),
                           lambda cw, *_: (lambda wb: (pack_conv(wb[0], dt), wb[1]))(fold_bn(cw, None, self.bn)))
        if KSPLIT and out is None and x.is_cuda:
            B, H, W, Cin = x.shape
            Ho, Wo = (H + 2 * self.p - self.k) // self.s + 1, (W + 2 * self.p - self.k) // self.s + 1
            ks = ops.ksplit_for(B * Ho * Wo, w.shape[0], self.k * self.k * Cin, dt)
            if ks:      # few output tiles, long K walk (the reductions of the 11 x 11 / 22 x 22 stages to 32 channels)
                return ops.conv2d_ksplit(x, w, self.k, self.k, self.s, self.p, ks, bias=b, act=ops.ACT_RELU)
        return ops.conv2d(x, w, self.k, self.k, self.s, self.p, bias=b, act=ops.ACT_RELU, out=out)

    def forward(self, x):
        return to_planar(self.run(to_cl(x, self.cdtype)))


def conv_bn_train(owner, conv, bn, x, k, s, p, out=None, relu=True):
    """Train-mode conv + BatchNorm with per-replica batch statistics (the reference keeps plain
    BatchNorm2d under DDP, train.py:279) + ReLU.  Running buffers are updated like nn.BatchNorm2d."""
    dt = owner.cdtype
    w, b = owner.packed("wt", (conv.weight,) + ((conv.bias,) if conv.bias is not None else ()),
                        lambda cw, *cb: (pack_conv(cw, dt), f32(cb[0]) if cb else None))
    g, be = owner.packed("bn", (bn.weight, bn.bias), lambda a, c: (f32(a), f32(c)))
    y = ops.conv2d(x, w, k, k, s, p, bias=b)
    sums = ops.chan_stats(y, 1)
    _update_running_stats(bn, sums, y.shape[0] * y.shape[1] * y.shape[2])
    return ops.chan_norm_apply(y, sums, 1, bn.eps, relu_inner=relu, gamma=g, beta=be, out=out if out is not None else y)


def _update_running_stats(bn, sums, n):
    """nn.BatchNorm2d's train-mode bookkeeping (momentum update with the unbiased variance, num_batches_tracked += 1) in one
    launch from the f64 sums the normalisation used"""
    ops.bn_running_update(sums, bn.running_mean, bn.running_var, bn.num_batches_tracked, n, bn.momentum)


def conv_bn_relu_autograd(owner, conv, bn, x, k, s, p, relu=True):
    """differentiable conv + BatchNorm (batch statistics in train mode, running statistics in eval mode) + ReLU"""
    dt = owner.cdtype
    wp, wdg = owner.packed("wag", (conv.weight,), lambda cw: (
        pack_conv(cw, dt), conv_dgrad_pack(cw, dt, k, s, p)))
    y = ConvFn.apply(x, conv.weight, conv.bias, wp, wdg, k, s, p, None)
    if not owner.training:
        raise NotImplementedError("gradients through eval-mode BatchNorm are not built (train.py always calls .train())")
    y, sums = BNReluFn.apply(y, bn.weight, bn.bias, bn.eps, relu)
    _update_running_stats(bn, sums, y.shape[0] * y.shape[1] * y.shape[2])
    return y


class DimensionalReduction(EmipModule):
    """Two ConvBR (create_backbone.py:199-208)."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.reduce = nn.Sequential(ConvBR(in_channel, out_channel, 3, padding=1),
                                    ConvBR(out_channel, out_channel, 3, padding=1))

    def run(self, x):
        return self.reduce[1].run(self.reduce[0].run(x))

    def forward(self, x):
        return to_planar(self.run(to_cl(x, self.cdtype)))


class NeighborConnectionDecoder(EmipModule):
    """NCD (create_backbone.py:46-76): x2 bilinear upsampling (align_corners=True), ConvBR, elementwise
    products, two concats, 1x1 head, x8 bilinear (align_corners=False) to the mask logits."""

    def __init__(self, channel):
        super().__init__()
        c = self.channel = channel
        self.conv_upsample1 = ConvBR(c, c, 3, padding=1)
        self.conv_upsample2 = ConvBR(c, c, 3, padding=1)
        self.conv_upsample3 = ConvBR(c, c, 3, padding=1)
        self.conv_upsample4 = ConvBR(c, c, 3, padding=1)
        self.conv_upsample5 = ConvBR(2 * c, 2 * c, 3, padding=1)
        self.conv_concat2 = ConvBR(2 * c, 2 * c, 3, padding=1)
        self.conv_concat3 = ConvBR(3 * c, 3 * c, 3, padding=1)
        self.conv4 = ConvBR(3 * c, 3 * c, 3, padding=1)
        self.conv5 = nn.Conv2d(3 * c, 1, 1)

    @staticmethod
    def _up2(x):
        return ops.bilinear(x, 2 * x.shape[1], 2 * x.shape[2], True)

    def run_train(self, zt5, zt4, zt3):
        dt = self.cdtype
        up = lambda t: BilinearFn.apply(t, 2 * t.shape[1], 2 * t.shape[2], True)
        up5, up4 = up(zt5), up(zt4)
        zt4_1 = MulFn.apply(self.conv_upsample1.run(up5), zt4, None)
        zt3_1 = MulFn.apply(self.conv_upsample2.run(up(zt4_1)), self.conv_upsample3.run(up4), zt3)
        zt4_2 = self.conv_concat2.run(ConcatFn.apply(None, zt4_1, self.conv_upsample4.run(up5)))
        zt3_2 = self.conv_concat3.run(ConcatFn.apply(None, zt3_1, self.conv_upsample5.run(up(zt4_2))))
        x = self.conv4.run(zt3_2)
        w5 = self.conv5.weight
        # the 1x1 head has N = 1: pad it to 8 output channels so that the backward GEMMs keep 16-byte rows
        def build(a, b):
            w8 = torch.zeros(8, a.shape[1], device=a.device)
            w8[0] = a.detach().reshape(-1)
            b8 = torch.zeros(8, device=a.device)
            b8[0] = b.detach()[0]
            return w8.to(dt).contiguous(), w8.t().to(dt).contiguous(), b8
        wp, wpt, b8 = self.packed("c5t", (w5, self.conv5.bias), build)
        pc = LinearFn.apply(x, w5, self.conv5.bias, None, wp, wpt, b8)          # [B,44,44,8], channel 0 = logits
        # detached: a module attribute that carries a grad_fn keeps the step's whole autograd graph alive until the next
        # forward, and with it every parameter's AccumulateGrad node and the stream that node was created under
        self.last_pc = pc.detach()
        return BilinearPlanarFn.apply(pc, 0, 1, 8 * pc.shape[1], 8 * pc.shape[2], False, 1.0)

    def run(self, zt5, zt4, zt3):
        """channels-last inputs [B,11,11,c], [B,22,22,c], [B,44,44,c] -> planar f32 logits [B,1,352,352]"""
        if torch.is_grad_enabled():
            return self.run_train(zt5, zt4, zt3)
        dt, c = self.cdtype, self.channel
        B = zt5.shape[0]
        up5, up4 = self._up2(zt5), self._up2(zt4)
        cat2 = torch.empty((B,) + tuple(zt4.shape[1:3]) + (2 * c,), dtype=dt, device=zt5.device)
        cat3 = torch.empty((B,) + tuple(zt3.shape[1:3]) + (3 * c,), dtype=dt, device=zt5.device)
        zt4_1 = ops.eltwise(self.conv_upsample1.run(up5), zt4, 0, out=cat2[..., :c])
        self.conv_upsample4.run(up5, out=cat2[..., c:])
        ops.eltwise(self.conv_upsample2.run(self._up2(zt4_1)), self.conv_upsample3.run(up4), 1, c3=zt3,
                    out=cat3[..., :c])
        zt4_2 = self.conv_concat2.run(cat2)
        self.conv_upsample5.run(self._up2(zt4_2), out=cat3[..., c:])
        x = self.conv4.run(self.conv_concat3.run(cat3))
        w5, b5 = self.packed("c5", (self.conv5.weight, self.conv5.bias),
                             lambda a, b: (a.detach().reshape(1, -1).to(dt).contiguous(), f32(b)))
        pc = ops.gemm(x, w5, bias=b5)                       # [B,44,44,1]
        self.last_pc = pc
        return ops.bilinear_planar(pc, 0, 1, 8 * pc.shape[1], 8 * pc.shape[2], False)

    def forward(self, zt5, zt4, zt3):
        dt = self.cdtype
        return self.run(to_cl(zt5, dt), to_cl(zt4, dt), to_cl(zt3, dt))


class FeatureExtraction(EmipModule):
    """Backbone selector (create_backbone.py:78-163).  Only the shipped configuration, pvt_v2_b5, is built
    (configs/configs.yaml:33); the reference's other branches are unused and partly unimportable."""

    def __init__(self, channel=32, pretrained=None, backbone_name='pvt_v2_b5', input_shape=None):
        super().__init__()
        self.backbone_name = backbone_name
        if backbone_name != 'pvt_v2_b5':
            raise Exception("Invalid Architecture Symbol: {}".format(backbone_name))
        from ...lib.pvt_v2 import pvt_v2_b5
        self.pvtv2_en = pvt_v2_b5(pretrained=None, in_channel_list=input_shape)

    def run(self, img_cl, deep=None, fork=None):
        """stage outputs 2..4; deep = (lo, hi): stages 3 and 4 for the images lo .. hi - 1 only; fork: the stream they run on
        (pvt_v2.run)"""
        return self.pvtv2_en.run(img_cl, deep=deep, fork=fork)[1:]

    def forward(self, x):
        return tuple(to_planar(o) for o in self.run(to_cl(x, self.cdtype, 8)))


class Decoder(EmipModule):
    def __init__(self, channel=32):
        super().__init__()
        self.NCD = NeighborConnectionDecoder(channel)

    def forward(self, x):
        return self.NCD(x[2], x[1], x[0])


class Network(EmipModule):
    """feat_net + (unused by EMIP) decoder, create_backbone.py:182-196."""

    def __init__(self, channel=32, pretrained=None, backbone_name='pvt_v2_b5', input_shape=None):
        super().__init__()
        self.channel = channel
        self.feat_net = FeatureExtraction(channel=channel, pretrained=pretrained, backbone_name=backbone_name,
                                          input_shape=input_shape)
        self.decoder = Decoder(self.channel)

    def forward(self, x):
        return self.decoder(self.feat_net(x))
