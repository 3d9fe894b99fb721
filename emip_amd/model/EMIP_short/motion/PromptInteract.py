"""Prompt injectors (camouflage feeder / motion collector): one MDTA transformer block.

State dict as /root/reference/model/EMIP_short/motion/PromptInteract.py:436-464
(transformer.{norm1,norm2,norm3}.body.{weight,bias}, transformer.attn.{temperature,q,q_dwconv,kv,kv_dwconv,
project_out}.weight, transformer.ffn.{project_in,dwconv,project_out}.weight).  Only the Injector of that file
is on EMIP's path; its other classes are dead code in the shipped configuration and are not rebuilt.

Kernel plan for x + MDTA(LN(x), LN(y)) then x + GDFN(LN(x)) on channels-last [B,44,44,128]:
LN rows -> 1x1 conv as GEMM -> depthwise 3x3 -> per-(image, head) 64x64 Gram over the 1936 pixels with the
L2 normalisation folded in as a rank-1 rescale -> softmax -> attn @ v as a 64-wide batched GEMM ->
project_out GEMM with the residual (the "prompt injection" add) in its epilogue; the gated FFN fuses
dwconv + GELU gate in one kernel and its output projection again carries the residual add.
"""
import torch
import torch.nn as nn

from .... import ops
from ....autograd import DwConvFn, GateFn, LayerNormFn, LinearFn, MdtaFn
from ....nn_base import EmipModule, f32, lin_packs, lin_packs_kpad, pack_dw, pack_linear, to_cl, to_planar


class WithBias_LayerNorm(nn.Module):
    def __init__(self, normalized_shape):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))


class LayerNorm(nn.Module):
    """holder named like the reference: .body.{weight,bias} (PromptInteract.py:352-362)"""

    def __init__(self, dim, LayerNorm_type='WithBias'):
        super().__init__()
        assert LayerNorm_type == 'WithBias'
        self.body = WithBias_LayerNorm(dim)


class FeedForward(nn.Module):
    def __init__(self, dim, ffn_expansion_factor, bias):
        super().__init__()
        hidden = int(dim * ffn_expansion_factor)
        self.hidden = hidden
        self.project_in = nn.Conv2d(dim, hidden * 2, kernel_size=1, bias=bias)
        self.dwconv = nn.Conv2d(hidden * 2, hidden * 2, kernel_size=3, stride=1, padding=1, groups=hidden * 2,
                                bias=bias)
        self.project_out = nn.Conv2d(hidden, dim, kernel_size=1, bias=bias)


class Attention_MDTA(nn.Module):
    def __init__(self, dim, num_heads, bias):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.q = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)
        self.q_dwconv = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim, bias=bias)
        self.kv = nn.Conv2d(dim, dim * 2, kernel_size=1, bias=bias)
        self.kv_dwconv = nn.Conv2d(dim * 2, dim * 2, kernel_size=3, stride=1, padding=1, groups=dim * 2, bias=bias)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)


class TransformerBlock_MDTA(EmipModule):
    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__()
        assert not bias and dim // num_heads == 64
        self.dim, self.heads = dim, num_heads
        self.norm1 = LayerNorm(dim, LayerNorm_type)
        self.attn = Attention_MDTA(dim, num_heads, bias)
        self.norm2 = LayerNorm(dim, LayerNorm_type)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)
        self.norm3 = LayerNorm(dim, LayerNorm_type)

    def _weights(self):
        dt = self.cdtype
        a, f = self.attn, self.ffn
        hid = f.hidden
        hid_pad = (hid + 7) // 8 * 8

        def build(n1w, n1b, n2w, n2b, n3w, n3b, temp, q, qd, kv, kvd, po, pin, dw, pout):
            wout = pout.detach().float().reshape(pout.shape[0], hid)
            wout = torch.cat([wout, wout.new_zeros(wout.shape[0], hid_pad - hid)], 1)
            return dict(n1=(f32(n1w), f32(n1b)), n2=(f32(n2w), f32(n2b)), n3=(f32(n3w), f32(n3b)),
                        temp=f32(temp).reshape(-1), q=pack_linear(q.reshape(q.shape[0], -1), dt), qd=pack_dw(qd),
                        kv=pack_linear(kv.reshape(kv.shape[0], -1), dt), kvd=pack_dw(kvd),
                        po=pack_linear(po.reshape(po.shape[0], -1), dt),
                        pin=pack_linear(pin.reshape(pin.shape[0], -1), dt), dw=pack_dw(dw),
                        pout=wout.to(dt).contiguous(), hid_pad=hid_pad)
        return self.packed("w", (self.norm1.body.weight, self.norm1.body.bias, self.norm2.body.weight,
                                 self.norm2.body.bias, self.norm3.body.weight, self.norm3.body.bias, a.temperature,
                                 a.q.weight, a.q_dwconv.weight, a.kv.weight, a.kv_dwconv.weight,
                                 a.project_out.weight, f.project_in.weight, f.dwconv.weight, f.project_out.weight),
                           build)

    def run_train(self, x, y):
        dt = self.cdtype
        a, f = self.attn, self.ffn
        hid = f.hidden
        hid_pad = (hid + 7) // 8 * 8

        def lin(wt):
            return lin_packs(wt.detach().reshape(wt.shape[0], -1), dt)

        def build(q, qd, kv, kvd, po, pin, dw, pout):
            return dict(q=lin(q), kv=lin(kv), po=lin(po), pin=lin(pin),
                        pout=lin_packs_kpad(pout.detach().reshape(pout.shape[0], hid), dt, hid_pad),
                        qd=(pack_dw(qd), pack_dw(qd, flip=True)),
                        kvd=(pack_dw(kvd), pack_dw(kvd, flip=True)),
                        dw=(pack_dw(dw), pack_dw(dw, flip=True)))
        w = self.packed("wt", (a.q.weight, a.q_dwconv.weight, a.kv.weight, a.kv_dwconv.weight, a.project_out.weight,
                               f.project_in.weight, f.dwconv.weight, f.project_out.weight), build)
        n1, n2, n3 = self.norm1.body, self.norm2.body, self.norm3.body
        xn = LayerNormFn.apply(x, n1.weight, n1.bias, 1e-5)
        yn = LayerNormFn.apply(y, n2.weight, n2.bias, 1e-5)
        q = LinearFn.apply(xn, a.q.weight, None, None, *w["q"])
        q = DwConvFn.apply(q, a.q_dwconv.weight, None, w["qd"][0], w["qd"][1], False)
        kv = LinearFn.apply(yn, a.kv.weight, None, None, *w["kv"])
        kv = DwConvFn.apply(kv, a.kv_dwconv.weight, None, w["kvd"][0], w["kvd"][1], False)
        o = MdtaFn.apply(q, kv, a.temperature)
        x1 = LinearFn.apply(o, a.project_out.weight, None, x, *w["po"])
        t = LinearFn.apply(LayerNormFn.apply(x1, n3.weight, n3.bias, 1e-5), f.project_in.weight, None, None, *w["pin"])
        t = DwConvFn.apply(t, f.dwconv.weight, None, w["dw"][0], w["dw"][1], False)
        t = GateFn.apply(t, hid, hid_pad)
        return LinearFn.apply(t, f.project_out.weight, None, x1, *w["pout"])

    def run(self, x, y):
        """x, y: channels-last [B,h,w,C]; returns a new tensor x + attn(x, y) + ffn(...)"""
        if torch.is_grad_enabled():
            return self.run_train(x, y)
        w = self._weights()
        B, h, wd, C = x.shape
        P, heads = h * wd, self.heads
        xn = ops.layernorm(x, w["n1"][0], w["n1"][1], 1e-5)
        yn = ops.layernorm(y, w["n2"][0], w["n2"][1], 1e-5)
        q = ops.dwconv3x3(ops.gemm(xn, w["q"]), w["qd"])
        kv = ops.dwconv3x3(ops.gemm(yn, w["kv"]), w["kvd"])           # [B,h,w,2C]: k | v
        attn = ops.mdta_attn(q.view(B, P, C), kv.view(B, P, 2 * C)[..., :C], w["temp"], B, heads, P)
        o = torch.empty((B, h, wd, C), dtype=x.dtype, device=x.device)
        for hd in range(heads):                                        # out[p, c1] = sum_c2 attn[c1, c2] v[p, c2]
            ops.gemm_batched(kv.view(B, P, 2 * C)[..., C + 64 * hd:], attn[:, hd], o.view(B, P, C)[..., 64 * hd:],
                             batch=B, M=P, N=64, K=64, lda=2 * C, ldw=64, ldc=C, bsA=P * 2 * C, bsW=heads * 4096,
                             bsC=P * C)
        x1 = ops.gemm(o, w["po"], res=x)                               # x + project_out(attn @ v)
        t = ops.gemm(ops.layernorm(x1, w["n3"][0], w["n3"][1], 1e-5), w["pin"])
        t = ops.dwconv3x3_gated(t, w["dw"], w["hid_pad"])
        return ops.gemm(t, w["pout"], res=x1, out=x1)


class Injector(EmipModule):
    def __init__(self, args=None):
        super().__init__()
        self.transformer = TransformerBlock_MDTA(dim=128, num_heads=2, ffn_expansion_factor=2.66, bias=False,
                                                 LayerNorm_type='WithBias')

    def run(self, x, y):
        return self.transformer.run(x, y)

    def forward(self, image_embeddings, flow):
        dt = self.cdtype
        return to_planar(self.run(to_cl(image_embeddings, dt), to_cl(flow, dt)))
