"""EMIP-short surface: CoUpdater(args).forward(image1, image2) -> (mask, flow_fw, flow_bw).

Drop-in for /root/reference/model/EMIP_short/model.py: same constructor dictionary, same forward signature and
return structure, same state_dict keys (including the reference's never-called modules dr2_new, dr3_new,
downscaling1, upscaling3/4 and backbone.decoder, which exist as parameters only).  Under torch.no_grad() the forward is a
fixed sequence of libemip_hip.so launches on the current stream; with grad enabled (train mode) every step is a
torch.autograd.Function whose forward and backward are the same library's kernels (emip_amd/autograd.py,
emip_amd/train.py).  Both frames go through the PVTv2 and GMFlow-CNN streams as ONE batch of 2B images, activations
stay channels-last in HBM between kernels, and only the two boundary conversions (planar images in, planar
mask/flows out) touch the reference's NCHW layout.
"""
from typing import Tuple

import torch
import torch.nn as nn

from ... import ops
from ...nn_base import EmipModule, _record, f32, conv_dgrad_pack, fold_bn, pack_conv, to_cl
from .create_backbone import DimensionalReduction, NeighborConnectionDecoder, Network, conv_bn_train
from .motion.common import LayerNorm2d
from .motion.gmflow.gmflow import GMFlow
from .motion.PromptInteract import Injector


# Order of the two independent encoders at the head of the forward.  emip_amd.graph captures every second sub-batch graph
# with the GMFlow CNN first: the concurrent streams then do not walk the same phases in lockstep -- one is in the
# bandwidth-bound CNN while the other is in the PVT stages (+1.4 %).  STAGGER = False captures every graph in
# the same order (tests/test_timed_config_gpu.py runs both orders against each other).
CNN_FIRST = False
STAGGER = True
# eval mode: conv_corr.0 computed from the rank-128 factors of the correlation volume (run_conv_corr_factored: 8.6 instead of
# 65 GFLOP per pair, no 7.5-MB volume per pair); False = the reference's literal order, the 3 x 3 conv over 1936 channels
CONV_CORR_FACTORED = True
# the two per-image GEMMs of the factored conv_corr.0 on the 8-wave body (emip_gemm8_batched) instead of the 4-wave strided-batched one
CONV_CORR_GEMM8 = True
# The reference runs the whole PVT backbone on BOTH frames (model.py:87-88) and then reads fea_2[0] only (:92; fea_1[1], fea_1[2]
# feed the decoder, :99-100): stages 3 and 4 of the second frame -- 43 of the 52 blocks -- produce values nothing reads, in the
# forward and (zero gradient) in the backward.  True: those stages run on the frame whose deep features are read (frame 1;
# frame 2 for EMIP-long's steps, model_long.py:71,89-90,113-116).  Outputs and gradients are the reference's; False: literal order.
PVT_DEEP_ONE_FRAME = True
# PVT stages 3-4 on a forked stream beside the GMFlow half of the forward (inference).  One step at a time: 10.59 -> 9.56 ms;
# with four steps in flight the two branches only compete: 2421 -> 2391 pairs/s (tools/flag_ab.py).  So the module default is
# off and graph.PipelinedShort captures its LATENCY graph (replay_alone) with it on
FORK_DEEP = False
FORK_PRIORITY = 0       # priority of the forked stream (-1 = high: no effect measured)
FORK_CNN = False        # with FORK_DEEP: the GMFlow CNN on a third branch beside PVT stages 1-2
# The training step (one step at a time by nature): PVT stages 3-4 on ops.fork_stream in the forward; autograd runs a node's
# backward on the stream of its forward, so the backward of those 43 blocks runs beside the backward of the GMFlow half as well.
# The deferred weight-gradient queue (ops.WgradQueue.flush) and the end of the step (train.train_step) order the two streams.
FORK_DEEP_TRAIN = True
FORK_TRAIN_PRIORITY = 0     # HIP stream priority of that branch (-1 = high)
# Two more branches were measured inside the captured step and bought nothing (61.55 ms with this fork alone; 61.51 with the frozen
# GMFlow CNN on a third stream beside PVT stages 1-2, 61.87 with unFlowLoss forward + backward on a side stream, 61.53 with both;
# tools/train_graph_ab.py, round 4): they are not in the code.

def _detach_tree(v):
    """the same nest of dicts / lists / tuples with every tensor detached"""
    if torch.is_tensor(v):
        return v.detach()
    if isinstance(v, dict):
        return {k: _detach_tree(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(_detach_tree(x) for x in v)
    return v


class CoUpdater(EmipModule):
    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.channel = args['channel']
        self.test_mode = args['test_mode']
        self.corr_levels, self.corr_radius = args['corr_levels'], args['corr_radius']
        self.hidden_dim, self.context_dim = args['hidden_dim'], args['context_dim']
        self.iters, self.inp_size = args['iters'], args['inp_size']

        self.backbone = Network(channel=self.channel, pretrained=None, backbone_name=args['backbone_name'],
                                input_shape=args['in_channel_list'])
        self.decoder = NeighborConnectionDecoder(self.channel)
        self.GMFlow = GMFlow(feature_channels=args['GMFlow']['feature_channels'], args=args)
        self.dr1 = DimensionalReduction(128, self.channel)
        self.dr2 = DimensionalReduction(320, self.channel)
        self.dr3 = DimensionalReduction(512, self.channel)
        self.conv_corr = nn.Sequential(nn.Conv2d(44 * 44, 968, 3, 1, 1), nn.BatchNorm2d(968), nn.ReLU(inplace=True),
                                       nn.Conv2d(968, 128, 3, 1, 1))
        self.injector = Injector()
        self.injector1 = Injector()
        # ---- parameters the reference registers but never uses in forward (model.py:53-58,66-84)
        self.dr2_new = nn.Conv2d(128, 32, kernel_size=3, stride=2, padding=1)
        self.dr3_new = nn.Sequential(nn.Conv2d(128, 64, 3, 2, 1), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                                     nn.Conv2d(64, 32, 3, 2, 1), nn.BatchNorm2d(32), nn.ReLU(inplace=True))
        self.downscaling1 = nn.Sequential(nn.Conv2d(64, 128, kernel_size=2, stride=2), LayerNorm2d(128), nn.GELU())
        self.upscaling4 = nn.Sequential(nn.ConvTranspose2d(512, 256, kernel_size=2, stride=2), LayerNorm2d(256),
                                        nn.GELU(), nn.ConvTranspose2d(256, 128, kernel_size=2, stride=2), nn.GELU())
        self.upscaling3 = nn.Sequential(nn.ConvTranspose2d(320, 128, kernel_size=2, stride=2), LayerNorm2d(128),
                                        nn.GELU())

    # ------------------------------------------------------------------------------------------
    def run_conv_corr(self, corr):
        """corr: [B, src, tgt] = channels-last [B,44,44,1936] (the reference's permuted view, matching.py:18-20)."""
        dt = self.cdtype
        B, n, _ = corr.shape
        h = w = int(round(n ** 0.5))
        x = corr.view(B, h, w, n)
        c0, bn, c3 = self.conv_corr[0], self.conv_corr[1], self.conv_corr[3]
        w3, b3 = self.packed("cc3", (c3.weight, c3.bias), lambda a, b: (pack_conv(a, dt), f32(b)))
        if torch.is_grad_enabled():
            from ...autograd import ConvFn
            from .create_backbone import conv_bn_relu_autograd
            y = conv_bn_relu_autograd(self, c0, bn, x, 3, 1, 1)
            w3p, w3d = self.packed("cc3t", (c3.weight,), lambda a: (
                pack_conv(a, dt), conv_dgrad_pack(a, dt, 3, 1, 1)))
            return ConvFn.apply(y, c3.weight, c3.bias, w3p, w3d, 3, 1, 1, None)
        if self.training:
            y = conv_bn_train(self, c0, bn, x, 3, 1, 1)
        else:
            w0, b0 = self.packed("cc0", (c0.weight, c0.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var),
                                 lambda cw, cb, *_: (lambda wb: (pack_conv(wb[0], dt), wb[1]))(fold_bn(cw, cb, bn)))
            y = ops.conv2d(x, w0, 3, 3, 1, 1, bias=b0, act=ops.ACT_RELU)
        return ops.conv2d(y, w3, 3, 3, 1, 1, bias=b3)

    def run_conv_corr_factored(self, tokens, B, h, w):
        """conv_corr (model.py:59-62,96) of the correlation volume WITHOUT the volume (eval mode).  The reference convolves
        corr[b, q, p] = <F0[b, p], F1[b, q]> / sqrt(C) (matching.py:16-20: raw global correlation, target position q as the
        channel) with conv_corr.0 (1936 -> 968 channels, 3 x 3): 65 GFLOP per pair.  corr is a rank-C product (C = 128), so
            out[b, p, co] = sum_tap sum_q W[co, q, tap] corr[b, q, p + tap]
                          = sum_tap sum_d F0[b, p + tap, d] * G[b, co, tap, d],   G[b] = (W' / sqrt(C)) F1[b]
        i.e. a per-image GEMM that turns the weights into a 128-channel 3 x 3 kernel G[b] ([968 * 9, 1936] x [1936, 128]) and
        a per-image 3 x 3 convolution of F0 with it (patch matrix [1936, 1152] x G[b]^T): 8.6 GFLOP per pair, and the 7.5 MB
        volume per pair is neither written nor read.  Zero padding of corr along p is zero padding of F0.  BatchNorm (eval) is
        folded into W' / the bias, ReLU in the epilogue; the sums are the reference's, re-associated (f32 mode: 1e-6).
        tokens: GMFlow's final features [2B, h*w, C] (frame 1 | frame 2)."""
        dt = self.cdtype
        n, C = h * w, tokens.shape[-1]
        c0, bn, c3 = self.conv_corr[0], self.conv_corr[1], self.conv_corr[3]
        cout = c0.weight.shape[0]

        g8 = CONV_CORR_GEMM8 and dt == torch.bfloat16 and (9 * C) % 64 == 0 and C % 8 == 0 and cout % 8 == 0
        npad = (n + 63) // 64 * 64 if g8 else n      # the 8-wave body walks K in tiles of 64: the 1936 target pixels padded to 1984

        def build(cw, cb, *_):
            wf, bf = fold_bn(cw, cb, bn)                                         # [968, 1936, 3, 3] f32
            wr = (wf * C ** -0.5).permute(0, 2, 3, 1).reshape(cout * 9, n)       # rows (co, tap), columns q
            if npad != n:
                wr = torch.nn.functional.pad(wr, (0, npad - n))
            return wr.to(dt).contiguous(), bf
        wr, b0 = self.packed("ccf%d" % npad, (c0.weight, c0.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var), build)
        w3, b3 = self.packed("cc3", (c3.weight, c3.bias), lambda a, b: (pack_conv(a, dt), f32(b)))
        f0, f1 = tokens[:B], tokens[B:]
        f1t = ops.transpose_pad(f1, npad)                                        # [B, C, n (padded with zeros)]
        g = torch.empty((B, cout * 9, C), dtype=dt, device=tokens.device)        # = per-image conv weights [968][(tap, d)]
        pm = ops.im2col3x3(f0.reshape(B, h, w, C))                               # [B, n, 9 C]
        y = torch.empty((B, h, w, cout), dtype=dt, device=tokens.device)
        if g8:
            ops.gemm8_batched(wr, f1t, g, B, cout * 9, C, npad, npad, npad, C, 0, C * npad, cout * 9 * C)
            ops.gemm8_batched(pm, g, y, B, n, cout, 9 * C, 9 * C, 9 * C, cout, n * 9 * C, cout * 9 * C, n * cout,
                              bias=b0, act=ops.ACT_RELU)
        else:
            ops.gemm_batched(wr, f1t, g, B, cout * 9, C, n, n, n, C, 0, C * n, cout * 9 * C)
            ops.gemm_batched_bias(pm, g, y, B, n, cout, 9 * C, 9 * C, 9 * C, cout, n * 9 * C, cout * 9 * C, n * cout,
                                  bias=b0, act=ops.ACT_RELU)
        return ops.conv2d(y, w3, 3, 3, 1, 1, bias=b3)

    def run_conv_corr_factored_train(self, tokens, h, w):
        """the training form of run_conv_corr_factored: conv_corr.0 through autograd.ConvCorr0Fn, then BatchNorm on batch
        statistics + ReLU and conv_corr.3 as in run_conv_corr"""
        from ...autograd import BNReluFn, ConvCorr0Fn, ConvFn
        from .create_backbone import _update_running_stats
        dt = self.cdtype
        n, C = tokens.shape[1], tokens.shape[2]
        c0, bn, c3 = self.conv_corr[0], self.conv_corr[1], self.conv_corr[3]
        cout = c0.weight.shape[0]

        def build(cw):
            sc = C ** -0.5
            wr = (cw.detach().float() * sc).permute(0, 2, 3, 1).reshape(cout * 9, n).to(dt).contiguous()
            wrt = wr.t().contiguous()
            # both are strided copies of the parameter: kept current by the one-launch refresh after the optimizer step
            _record(cw, wr, (1, cout, 9, n), (0, n * 9, 1, 9), scale=sc)
            _record(cw, wrt, (1, n, cout, 9), (0, 9, n * 9, 1), scale=sc)
            return wr, wrt
        wr, wrt = self.packed("ccft", (c0.weight,), build)
        y = ConvCorr0Fn.apply(tokens, c0.weight, c0.bias, wr, wrt, h, w)
        y, sums = BNReluFn.apply(y, bn.weight, bn.bias, bn.eps, True)
        _update_running_stats(bn, sums, y.shape[0] * y.shape[1] * y.shape[2])
        w3p, w3d = self.packed("cc3t", (c3.weight,), lambda a: (pack_conv(a, dt), conv_dgrad_pack(a, dt, 3, 1, 1)))
        return ConvFn.apply(y, c3.weight, c3.bias, w3p, w3d, 3, 1, 1, None)

    def last_corr(self):
        """the raw correlation volume [B, src, tgt] of the last run() (computed on demand when the factored conv_corr ran)"""
        L = self.last
        if L.get("corr") is None:
            c0 = self.GMFlow.last["tokens"]
            B = c0.shape[0] // 2
            n, C = c0.shape[1], c0.shape[2]
            wdt = int(round(n ** 0.5))
            corr = torch.empty((B, n, n), dtype=c0.dtype, device=c0.device)
            if ops.match_eligible(c0):
                ops.match(c0[:B], c0[B:], wdt, C ** -0.5, scores=corr)
            else:       # f32 mode, or a frame size emip_match does not take: F0 F1^T per image on the batched GEMM, then the scale
                ops.gemm_batched(c0[:B], c0[B:], corr, B, n, n, C, C, C, n, n * C, n * C, n * n)
                corr.mul_(C ** -0.5)
            L["corr"] = corr
        return L["corr"]

    def run(self, image1, image2, tail=True):
        """Planar images [B,3,H,W] -> (mask planar f32, flow predictions [2B,2,H,W] list, intermediates).
        tail=False: stop behind the motion collector (`conv_corr`) -- EMIP-long reads the backbone features and the
        correlation features of its short-term part and decodes on its own (model_long.py:68-117), so the short-term
        prompt injection / reductions / decoder and the flow predictions would be computed for nothing; returns (None, [])."""
        dt = self.cdtype
        B = image1.shape[0]
        if torch.is_grad_enabled() or image1.shape != image2.shape or image1.dtype != torch.float32:
            imgs = to_cl(torch.cat((image1, image2), 0), dt, 8)        # [2B,H,W,8]
        else:       # both frames straight into their halves of the channels-last batch (no concatenated f32 copy)
            imgs = torch.empty((2 * B,) + tuple(image1.shape[2:]) + (8,), dtype=dt, device=image1.device)
            ops.planar_to_cl(image1.contiguous(), dt, 8, out=imgs[:B])
            ops.planar_to_cl(image2.contiguous(), dt, 8, out=imgs[B:])
        # stage 2 of both frames; stages 3, 4 of the frame whose deep features are read (tail=False: EMIP-long reads frame 2's)
        deep = ((0, B) if tail else (B, 2 * B)) if PVT_DEEP_ONE_FRAME else None
        # PVT stages 3-4 (40 + 3 blocks of small launches, 40 % of a step's serial time) feed only the reductions in front of the
        # decoder: on a forked stream they run beside the GMFlow half (inside a captured graph: a fork / join of the graph)
        fork = None
        if FORK_DEEP_TRAIN and tail and torch.is_grad_enabled() and self.training and imgs.is_cuda:
            # training step: the same branch on ops.fork_stream (its backward then runs there as well); train_step joins it
            fork = ops.fork_stream(imgs.device, priority=FORK_TRAIN_PRIORITY, name="deep%d" % FORK_TRAIN_PRIORITY)
        elif FORK_DEEP and tail and not torch.is_grad_enabled() and imgs.is_cuda:
            fork = getattr(self, "_fork", None)
            if fork is None or fork.device != imgs.device:
                fork = torch.cuda.Stream(device=imgs.device, priority=FORK_PRIORITY)
                object.__setattr__(self, "_fork", fork)
        if fork is not None and FORK_CNN and not torch.is_grad_enabled():
            # ... and the GMFlow CNN on a third branch from the start, beside PVT stages 1-2
            cur = torch.cuda.current_stream()
            fork2 = getattr(self, "_fork2", None)
            if fork2 is None or fork2.device != imgs.device:
                fork2 = torch.cuda.Stream(device=imgs.device)
                object.__setattr__(self, "_fork2", fork2)
            fork2.wait_stream(cur)
            imgs.record_stream(fork2)
            with torch.cuda.stream(fork2):
                gm = self.GMFlow.backbone.run(imgs)
            gm.record_stream(cur)
            fea = self.backbone.feat_net.run(imgs, deep=deep, fork=fork)
            cur.wait_stream(fork2)
        elif CNN_FIRST:
            gm = self.GMFlow.backbone.run(imgs)                        # [2B,44,44,128]
            fea = self.backbone.feat_net.run(imgs, deep=deep, fork=fork)
        else:
            fea = self.backbone.feat_net.run(imgs, deep=deep, fork=fork)
            gm = self.GMFlow.backbone.run(imgs)                        # [2B,44,44,128]
        ab = self.injector.run(gm, fea[0])                             # camouflage feeder (shared weights)
        if torch.is_grad_enabled() and ab.requires_grad:
            if CONV_CORR_FACTORED and self.training:
                preds, corr = self.GMFlow.run_train(ab, corr=False)
                cc = self.run_conv_corr_factored_train(self.GMFlow.last["tokens"], ab.shape[1], ab.shape[2])
            else:
                preds, corr = self.GMFlow.run_train(ab)
                cc = self.run_conv_corr(corr)                          # motion collector, part 1
        elif CONV_CORR_FACTORED and not self.training and not torch.is_grad_enabled():
            # eval: conv_corr.0 through the rank-128 factors of the volume; the volume itself is never materialised
            preds, corr = self.GMFlow.run(ab[:B], ab[B:], flows=tail, corr=False)
            cc = self.run_conv_corr_factored(self.GMFlow.last["tokens"], B, ab.shape[1], ab.shape[2])
        else:
            preds, corr = self.GMFlow.run(ab[:B], ab[B:], flows=tail)
            cc = self.run_conv_corr(corr)
        if not tail:
            self.last = dict(fea=fea, gm=gm, ab=ab, corr=corr, conv_corr=cc)
            self._release_graph()
            return None, preds
        fea_new = self.injector1.run(fea[0][:B], cc)
        f1 = self.dr1.run(fea_new)
        if fork is not None:
            torch.cuda.current_stream().wait_stream(fork)              # the deep features
        f2 = self.dr2.run(fea[1][:B])
        f3 = self.dr3.run(fea[2][:B])
        mask = self.decoder.run(f3, f2, f1)
        self.last = dict(fea=fea, gm=gm, ab=ab, corr=corr, conv_corr=cc, inj1=fea_new, dr=(f1, f2, f3))
        self._release_graph()
        return mask, preds

    def _release_graph(self):
        """the kept intermediates (`.last`: values for parity checks) must not hold the autograd graph of a training forward:
        a graph that outlives its step keeps every parameter's AccumulateGrad node, and those remember the stream they were
        created under -- the next step on another stream (a capture stream, train.GraphedTrainStep) would be ordered against it"""
        if torch.is_grad_enabled():
            self.last = _detach_tree(self.last)
            self.GMFlow.last = _detach_tree(self.GMFlow.last)

    def forward(self, image1, image2):
        B = image1.shape[0]
        if torch.is_grad_enabled() and not self.training and any(p.requires_grad for p in self.parameters()):
            # eval mode with autograd on (the reference allows the call; none of its drivers does it, test.py:21 runs under
            # no_grad): the values are the reference's, computed by the inference kernels; the backward kernels exist for the
            # training configuration only (batch-statistics BatchNorm, train.py:38), so the outputs carry no graph and a
            # .backward() on them fails where the divergence would begin, not here
            if not getattr(self, "_warned_eval_grad", False):
                import warnings
                warnings.warn("emip_amd CoUpdater.eval() called with autograd enabled: outputs are computed without a graph "
                              "(use .train() for gradients, torch.no_grad() for inference)")
                object.__setattr__(self, "_warned_eval_grad", True)
            with torch.no_grad():
                mask, preds = self.run(image1, image2)
        else:
            mask, preds = self.run(image1, image2)
        flow_fw = [p[:B] for p in preds]
        flow_bw = [p[B:] for p in preds]
        return mask, flow_fw, flow_bw

    def postprocess_masks(self, masks: torch.Tensor, input_size: Tuple[int, ...],
                          original_size: Tuple[int, ...]) -> torch.Tensor:
        """model.py:105-134 (unused by the drivers): resize to input_size, crop, resize to original_size."""
        import torch.nn.functional as F
        masks = F.interpolate(masks, (input_size, input_size), mode="bilinear", align_corners=False)
        masks = masks[..., :input_size, :input_size]
        return F.interpolate(masks, original_size, mode="bilinear", align_corners=False)
