"""GMFlow CNN encoder (1/8 resolution features) on MI355X kernels.

Keys match /root/reference/model/EMIP_short/motion/gmflow/backbone.py (conv1, layer{1,2,3}.{0,1}.{conv1,conv2,
downsample.0}, conv2, plus the never-called adaptor parameters dwconv64/96/128, dwconv_pre, dwconv, dwconv_post
which exist only as state_dict entries).  Each 3x3 conv is an implicit GEMM; InstanceNorm2d (affine=False, no
running statistics) is a per-(image, channel) statistics pass plus one fused normalise/ReLU/residual pass.
"""
import torch
import torch.nn as nn

from ..... import ops
from .....nn_base import EmipModule, f32, pack_conv, pack_linear, to_cl, to_planar


def _conv_inorm(x, w, k, stride, pad, relu, bias=None, res=None, relu_outer=False):
    """conv + InstanceNorm2d (+ReLU, + residual): the conv's first workgroup clears the statistics scratch, so the pair of
    normalisation kernels needs no zero-fill launch of its own"""
    B, cout = x.shape[0], w.shape[0]
    sums = torch.empty((B, cout, 2), dtype=torch.float64, device=x.device)
    y = ops.conv2d(x, w, k, k, stride, pad, bias=bias, zero=sums)
    ops.chan_stats(y, B, sums=sums)
    return ops.chan_norm_apply(y, sums, B, 1e-5, relu_inner=relu, relu_outer=relu_outer, res=res, out=y)


# the stride-1 3 x 3 convolutions of the residual blocks (10 of the encoder's 15 convolutions) as direct convolutions on an LDS
# halo tile with norm1 + ReLU applied while conv2 stages its input and the InstanceNorm sums taken in the conv epilogues
# (emip_conv3x3_halo, conv_halo.hip): a stride-1 block is 3 launches instead of 6 and 4 tensor passes instead of 9.  bf16 and
# the shapes that kernel takes (352 x 352 inputs: all three levels); anything else takes conv + statistics + normalise launches
CNN_HALO = True
# widest block that takes it.  Measured (MI355X, 16 pairs, tools/flag_ab.py; pairs/s with 4 steps in flight | one step at a time):
# none 2223 | 1344, 64: 2327 | 1400, 96: 2405 | 1432; the 128-channel level on the final build 2417 -> 2433 | 1515 -> 1526 (its
# first version, which rebuilt the normalisation table from global memory for every tile, lost 16 % in flight)
CNN_HALO_MAXC = 128


# a block input that is itself relu(InstanceNorm(raw conv output)) -- the stem's -- or a residual that is InstanceNorm(raw) -- the
# downsample branch's -- is never stored: the consumers normalise the raw tensor from its sums (the halo conv while it stages,
# emip_chan_norm_apply_res for the skip connection, rounding as the stored tensor would have been: the same bits, one pass less)
CNN_RAW_RES = True
# the 7 x 7 stride-2 stem as a direct convolution with its sums in the epilogue (emip_conv_stem): alone 132 -> 64 us at 32 images
CNN_STEM = True


def _halo_block(x, w1p, w2p, ws, x_sums=None):
    """backbone.py:39-69 for in_planes == planes, stride 1: relu(x + norm2(conv2(relu(norm1(conv1(x)))))); x_sums: x is a raw
    conv output whose relu(InstanceNorm(.)) is the block's real input"""
    B, C = x.shape[0], x.shape[-1]
    s1 = torch.empty((B, C, 2), dtype=torch.float64, device=x.device)
    s2 = torch.empty((B, C, 2), dtype=torch.float64, device=x.device)
    y1 = ops.conv3x3_halo(x, w1p, in_sums=x_sums, in_eps=1e-5, out_sums=s1, ws=ws)
    y2 = ops.conv3x3_halo(y1, w2p, in_sums=s1, in_eps=1e-5, out_sums=s2, ws=ws, out=torch.empty_like(y1))
    return ops.chan_norm_apply(y2, s2, B, 1e-5, relu_inner=True, relu_outer=True, res=x, out=y2, res_sums=x_sums, res_relu=True)


class ResidualBlock(EmipModule):
    """backbone.py:39-69"""

    def __init__(self, in_planes, planes, norm_layer=nn.InstanceNorm2d, stride=1, dilation=1):
        super().__init__()
        assert dilation == 1
        self.stride = stride
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride, bias=False)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1, bias=False)
        if stride == 1 and in_planes == planes:
            self.downsample = None
        else:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride),
                                            norm_layer(planes))

    def run(self, x, halo_ws=None, x_sums=None):
        dt = self.cdtype
        planes = self.conv2.weight.shape[0]
        B, Ho, Wo = x.shape[0], (x.shape[1] - 1) // self.stride + 1, (x.shape[2] - 1) // self.stride + 1
        if halo_ws is not None and not (dt == torch.bfloat16 and planes <= CNN_HALO_MAXC and ops.conv3x3_halo_eligible(B, Ho, Wo, planes, planes)
                                        and halo_ws.numel() >= ops.conv3x3_halo_ws_bytes(B, Ho, Wo, planes)):
            halo_ws = None
        if halo_ws is not None and self.downsample is None and self.stride == 1:
            w1p, w2p = self.packed("wh", (self.conv1.weight, self.conv2.weight),
                                   lambda a, b: (ops.conv3x3_halo_pack(pack_conv(a, dt)), ops.conv3x3_halo_pack(pack_conv(b, dt))))
            return _halo_block(x, w1p, w2p, halo_ws, x_sums)
        assert x_sums is None
        if halo_ws is not None:
            # the strided first convolution and the 1 x 1 downsample stay implicit GEMMs; conv2 normalises conv1's raw output
            # while it stages it and takes its own sums
            w1, w2p, wd, bd = self.packed("wh2", (self.conv1.weight, self.conv2.weight, self.downsample[0].weight, self.downsample[0].bias),
                                          lambda a, b, c, d: (pack_conv(a, dt), ops.conv3x3_halo_pack(pack_conv(b, dt)), pack_conv(c, dt), f32(d)))
            s1 = torch.empty((B, planes, 2), dtype=torch.float64, device=x.device)
            s2 = torch.empty((B, planes, 2), dtype=torch.float64, device=x.device)
            y1 = ops.conv2d(x, w1, 3, 3, self.stride, 1, zero=s1)
            ops.chan_stats(y1, B, sums=s1)
            y2 = ops.conv3x3_halo(y1, w2p, in_sums=s1, in_eps=1e-5, out_sums=s2, ws=halo_ws, out=torch.empty_like(y1))
            if CNN_RAW_RES:
                sd = torch.empty((B, planes, 2), dtype=torch.float64, device=x.device)
                yd = ops.conv2d(x, wd, 1, 1, self.stride, 0, bias=bd, zero=sd)
                ops.chan_stats(yd, B, sums=sd)
                return ops.chan_norm_apply(y2, s2, B, 1e-5, relu_inner=True, relu_outer=True, res=yd, out=y2, res_sums=sd, res_relu=False)
            res = _conv_inorm(x, wd, 1, self.stride, 0, relu=False, bias=bd)
            return ops.chan_norm_apply(y2, s2, B, 1e-5, relu_inner=True, relu_outer=True, res=res, out=y2)
        w1, w2 = self.packed("w", (self.conv1.weight, self.conv2.weight),
                             lambda a, b: (pack_conv(a, dt), pack_conv(b, dt)))
        y = _conv_inorm(x, w1, 3, self.stride, 1, relu=True)
        if self.downsample is not None:
            wd, bd = self.packed("d", (self.downsample[0].weight, self.downsample[0].bias),
                                 lambda a, b: (pack_conv(a, dt), f32(b)))
            x = _conv_inorm(x, wd, 1, self.stride, 0, relu=False, bias=bd)
        return _conv_inorm(y, w2, 3, 1, 1, relu=True, res=x, relu_outer=True)


class CNNEncoder(EmipModule):
    """backbone.py:72-192 with num_output_scales == 1 (strides 2, 1, 2, 2 -> 1/8)."""

    def __init__(self, output_dim=128, norm_layer=nn.InstanceNorm2d, num_output_scales=1, **kwargs):
        super().__init__()
        assert num_output_scales == 1, "EMIP uses a single 1/8 scale (gmflow.py:13-21)"
        dims = [64, 96, 128]
        self.conv1 = nn.Conv2d(3, dims[0], kernel_size=7, stride=2, padding=3, bias=False)
        self.in_planes = dims[0]
        self.layer1 = self._make_layer(dims[0], 1, norm_layer)
        self.layer2 = self._make_layer(dims[1], 2, norm_layer)
        self.layer3 = self._make_layer(dims[2], 2, norm_layer)
        self.conv2 = nn.Conv2d(dims[2], output_dim, 1, 1, 0)
        # adaptor parameters of the reference (backbone.py:99-111): registered, never used in forward
        self.dwconv64 = nn.Conv2d(64, 64, 3, 1, 1, bias=True, groups=64)
        self.dwconv96 = nn.Conv2d(96, 96, 3, 1, 1, bias=True, groups=96)
        self.dwconv128 = nn.Conv2d(128, 128, 3, 1, 1, bias=True, groups=128)
        self.dwconv_pre = nn.Conv2d(64, 16, kernel_size=3, stride=1, padding=1, bias=False)
        self.dwconv = nn.Conv2d(16, 16, 3, 1, 1, bias=True, groups=16)
        self.dwconv_post = nn.Conv2d(16, 64, kernel_size=3, stride=1, padding=1, bias=False)

    def _make_layer(self, dim, stride, norm_layer):
        layers = (ResidualBlock(self.in_planes, dim, norm_layer, stride=stride), ResidualBlock(dim, dim, norm_layer))
        self.in_planes = dim
        return nn.Sequential(*layers)

    def run(self, img_cl):
        dt = self.cdtype
        cin = img_cl.shape[-1]
        w1, w2, b2 = self.packed("w", (self.conv1.weight, self.conv2.weight, self.conv2.bias),
                                 lambda a, b, c: (pack_conv(a, dt, cin_pad=cin), pack_linear(b.reshape(b.shape[0], -1), dt),
                                                  f32(c)))
        B, H, W = img_cl.shape[0], (img_cl.shape[1] + 1) // 2, (img_cl.shape[2] + 1) // 2
        if CNN_HALO and dt == torch.bfloat16 and ops.conv3x3_halo_eligible(B, H, W, 64, 64):
            # the stem as before; its statistics scratch and the ticket block of the halo launches' workspace are one allocation,
            # cleared by the stem conv's first workgroup (the halo launches leave the tickets at zero, a graph replay starts from
            # zero again); the workspace is sized for layer1, which needs the most, and serves all three levels in turn
            nb = B * 64 * 2 * 8
            wsb = ops.conv3x3_halo_ws_bytes(B, H, W, 64)
            buf = torch.empty(nb + wsb, dtype=torch.uint8, device=img_cl.device)
            sums, ws = buf[:nb].view(torch.float64).view(B, 64, 2), buf[nb:]
            if CNN_STEM and cin == 8 and ops.conv_stem_eligible(B, img_cl.shape[1], img_cl.shape[2], 8, 64):
                # the 7 x 7 stem as a direct convolution too, its InstanceNorm sums from the epilogue; the ticket block is cleared
                # by a small fill (the stem no longer runs through the GEMM whose first workgroup used to clear it)
                buf[nb:nb + (4 * B + 63) // 64 * 64].zero_()
                w1s = self.packed("ws", (self.conv1.weight,), lambda a: (ops.conv_stem_pack(pack_conv(a, dt, cin_pad=8)),))[0]
                y = ops.conv_stem(img_cl, w1s, out_sums=sums, ws=ws)
            else:
                y = ops.conv2d(img_cl, w1, 7, 7, 2, 3, zero=buf[:nb + (4 * B + 63) // 64 * 64])
                ops.chan_stats(y, B, sums=sums)
            if CNN_RAW_RES and CNN_HALO_MAXC >= 64:
                x = self.layer1[0].run(y, ws, x_sums=sums)        # the stem's norm + relu happen inside layer1.0
            else:
                x = self.layer1[0].run(ops.chan_norm_apply(y, sums, B, 1e-5, relu_inner=True, out=y), ws)
            x = self.layer1[1].run(x, ws)
        else:
            ws = None
            x = _conv_inorm(img_cl, w1, 7, 2, 3, relu=True)
            x = self.layer1[1].run(self.layer1[0].run(x))
        for layer in (self.layer2, self.layer3):
            x = layer[1].run(layer[0].run(x, ws), ws)
        return ops.gemm(x, w2, bias=b2)

    def forward(self, x):
        return [to_planar(self.run(to_cl(x, self.cdtype, 8)))]
