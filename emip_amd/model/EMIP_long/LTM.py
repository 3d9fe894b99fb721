"""Long-term memory (space-time-memory style) of EMIP-long on MI355X kernels.

State dict as /root/reference/model/EMIP_long/LTM.py: KV_M_r4.{Key,Value}, KV_Q_r4.{Key,Value},
fusion.{conv1_m,conv1_fusion}.*, plus Decoder / dr1 / dr2 / dr3 which the reference registers but never calls.

memorize (LTM.py:103-111):  r4 = conv3x3(512->128)(relu(bn(conv3x3(128->512)(fea + corr)))), k = Key(r4), v = Value(r4)
segment  (LTM.py:122-132, Memory.forward :49-68):  p = softmax over the T*1936 memory positions of
         K_mem^T k_q / sqrt(128); mem = V_mem p; out = cat(mem, v_q).
The memory read is one launch of the fused attention kernel (queries = the 1936 pixels of the current frame,
keys/values = the T <= 5 stored frames, channels-last [S, T*1936, 128]), written straight into the first half
of the concat buffer.
"""
import torch
import torch.nn as nn

from ... import ops
from ...autograd import ConcatFn, ConvFn, MemoryReadFn
from ...nn_base import EmipModule, f32, conv_dgrad_pack, fold_bn, pack_conv
from ..EMIP_short.create_backbone import DimensionalReduction, NeighborConnectionDecoder


class fusion(EmipModule):
    def __init__(self):
        super().__init__()
        self.conv1_m = nn.Sequential(nn.Conv2d(1, 64, 3, 1, 1), nn.LayerNorm(64), nn.ReLU(inplace=True),
                                     nn.Conv2d(64, 128, 3, 1, 1))          # never used by the reference either
        self.conv1_fusion = nn.Sequential(nn.Conv2d(128, 512, 3, 1, 1), nn.BatchNorm2d(512), nn.ReLU(inplace=True),
                                          nn.Conv2d(512, 128, 3, 1, 1))

    def run(self, fea, corr):
        dt = self.cdtype
        c0, bn, c3 = self.conv1_fusion[0], self.conv1_fusion[1], self.conv1_fusion[3]
        w0, b0 = self.packed("c0", (c0.weight, c0.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var),
                             lambda cw, cb, *_: (lambda wb: (pack_conv(wb[0], dt), wb[1]))(fold_bn(cw, cb, bn)))
        w3, b3 = self.packed("c3", (c3.weight, c3.bias), lambda a, b: (pack_conv(a, dt), f32(b)))
        x = ops.eltwise(fea, corr, 2)
        if torch.is_grad_enabled():
            # training step (train_long.py:37-58): batch-statistics BatchNorm, every op an autograd Function; fea and corr
            # come out of the frozen short-term part and are constants
            from ..EMIP_short.create_backbone import conv_bn_relu_autograd
            y = conv_bn_relu_autograd(self, c0, bn, x, 3, 1, 1)
            w3p, w3d = self.packed("c3t", (c3.weight,), lambda a: (
                pack_conv(a, dt), conv_dgrad_pack(a, dt, 3, 1, 1)))
            return ConvFn.apply(y, c3.weight, c3.bias, w3p, w3d, 3, 1, 1, None)
        if self.training:
            from ..EMIP_short.create_backbone import conv_bn_train
            x = conv_bn_train(self, c0, bn, x, 3, 1, 1)
            return ops.conv2d(x, w3, 3, 3, 1, 1, bias=b3)
        x = ops.conv2d(x, w0, 3, 3, 1, 1, bias=b0, act=ops.ACT_RELU)
        return ops.conv2d(x, w3, 3, 3, 1, 1, bias=b3)


class KeyValue(EmipModule):
    def __init__(self, indim, keydim, valdim):
        super().__init__()
        self.Key = nn.Conv2d(indim, keydim, kernel_size=(3, 3), padding=(1, 1), stride=1)
        self.Value = nn.Conv2d(indim, valdim, kernel_size=(3, 3), padding=(1, 1), stride=1)

    def run(self, x, out_k=None, out_v=None):
        dt = self.cdtype
        if torch.is_grad_enabled():
            pk, pkd, pv, pvd = self.packed("kvt", (self.Key.weight, self.Value.weight), lambda a, c: (
                pack_conv(a, dt), conv_dgrad_pack(a, dt, 3, 1, 1),
                pack_conv(c, dt), conv_dgrad_pack(c, dt, 3, 1, 1)))
            return (ConvFn.apply(x, self.Key.weight, self.Key.bias, pk, pkd, 3, 1, 1, None),
                    ConvFn.apply(x, self.Value.weight, self.Value.bias, pv, pvd, 3, 1, 1, None))
        wk, bk, wv, bv = self.packed("kv", (self.Key.weight, self.Key.bias, self.Value.weight, self.Value.bias),
                                     lambda a, b, c, d: (pack_conv(a, dt), f32(b), pack_conv(c, dt), f32(d)))
        return (ops.conv2d(x, wk, 3, 3, 1, 1, bias=bk, out=out_k), ops.conv2d(x, wv, 3, 3, 1, 1, bias=bv, out=out_v))


class Memory(nn.Module):
    """No parameters (LTM.py:44-68); the read itself happens in LTM.segment_cl."""


class LTM(EmipModule):
    def __init__(self):
        super().__init__()
        self.KV_M_r4 = KeyValue(128, keydim=128, valdim=128)
        self.KV_Q_r4 = KeyValue(128, keydim=128, valdim=128)
        self.Memory = Memory()
        self.fusion = fusion()
        self.Decoder = NeighborConnectionDecoder(32)
        self.dr1 = DimensionalReduction(256, 32)
        self.dr2 = DimensionalReduction(320, 32)
        self.dr3 = DimensionalReduction(512, 32)

    def memorize_cl(self, fea0, corr):
        """fea0, corr: channels-last [S,h,w,128] -> key, value channels-last [S,h,w,128]"""
        return self.KV_M_r4.run(self.fusion.run(fea0, corr))

    def segment_cl(self, fea0, keys, values):
        """fea0 [S,h,w,128]; keys/values [S,T,h*w,128] -> cat(mem, v_q) channels-last [S,h,w,256]"""
        S, h, w, C = fea0.shape
        T, n = keys.shape[1], h * w
        if torch.is_grad_enabled():
            kq, vq = self.KV_Q_r4.run(fea0)
            mem = MemoryReadFn.apply(kq.view(S, n, C), keys.reshape(S, T * n, C), values.reshape(S, T * n, C))
            return ConcatFn.apply(None, mem.view(S, h, w, C), vq)
        out = torch.empty((S, h, w, 2 * C), dtype=fea0.dtype, device=fea0.device)
        kq, _ = self.KV_Q_r4.run(fea0, out_v=out[..., C:])
        ops.attention(kq, keys, values, out, batch=S, heads=1, nwin=1, Lq=n, Lk=T * n, D=C, DV=C, q_bs=n * C,
                      k_bs=T * n * C, v_bs=T * n * C, o_bs=n * 2 * C, ldq=C, ldk=C, ldv=C, ldo=2 * C,
                      scale=C ** -0.5)
        return out
