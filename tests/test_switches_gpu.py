"""Every module-level switch of the product path, flipped one at a time: the bf16 forward (and, for the training switches, the
gradients of one training step) must agree with the default setting inside the bf16 mode's own repeatability band.  The defaults are
the measured winners; the other value of each switch stays a working code path (smaller shapes, other dtypes and
the A/B tools use them), so it is tested."""
import importlib

import pytest
import torch

from emip_amd.filler import synthetic_gt, synthetic_pair

pytestmark = pytest.mark.gpu

INFERENCE = [("emip_amd.lib.pvt_v2", "SRA_FUSED", False), ("emip_amd.lib.pvt_v2", "SRA_BLOCK_MAXC", 320),
             ("emip_amd.lib.pvt_v2", "SRA_BLOCK_WIDE_ROWS", 10 ** 9), ("emip_amd.lib.pvt_v2", "SR_WIDE_TILE", 192), ("emip_amd.lib.pvt_v2", "SR_WIDE_TILE", True),
             ("emip_amd.model.EMIP_short.motion.gmflow.backbone", "CNN_HALO", False),
             ("emip_amd.model.EMIP_short.motion.gmflow.backbone", "CNN_HALO_MAXC", 64),
             ("emip_amd.model.EMIP_short.motion.gmflow.backbone", "CNN_RAW_RES", False),
             ("emip_amd.model.EMIP_short.motion.gmflow.backbone", "CNN_STEM", False),
             ("emip_amd.model.EMIP_short.model", "CONV_CORR_GEMM8", False), ("emip_amd.lib.pvt_v2", "STATS_IN_LAUNCH", False),
             ("emip_amd.lib.pvt_v2", "FC1DW_BAND_MIN_ROWS", 0), ("emip_amd.lib.pvt_v2", "FC1DW_BAND_MIN_ROWS", 9),
             ("emip_amd.lib.pvt_v2", "MLP_BLOCK", True), ("emip_amd.lib.pvt_v2", "SR_KSPLIT", True),
             ("emip_amd.lib.pvt_v2", "FUSED_LN", False),
             ("emip_amd.model.EMIP_short.create_backbone", "KSPLIT", True),
             ("emip_amd.model.EMIP_short.model", "CNN_FIRST", True),
             ("emip_amd.model.EMIP_short.model", "CONV_CORR_FACTORED", False),
             ("emip_amd.model.EMIP_short.model", "PVT_DEEP_ONE_FRAME", False),
             ("emip_amd.model.EMIP_short.motion.gmflow.transformer", "WATTN_QPROJ", False),
             ("emip_amd.model.EMIP_short.motion.gmflow.transformer", "WATTN_MERGE", False),
             ("emip_amd.model.EMIP_short.motion.gmflow.transformer", "FFN_BLOCK", False)]
TRAINING = [("emip_amd.autograd", "DW_BWD_FUSED", False), ("emip_amd.autograd", "WATTN_BWD_FUSED", False),
            ("emip_amd.autograd", "MATCH_BWD_FUSED", False), ("emip_amd.ops", "WIDE_WGRAD", False),
            ("emip_amd.model.EMIP_short.model", "CONV_CORR_FACTORED", False),
            ("emip_amd.model.EMIP_short.model", "PVT_DEEP_ONE_FRAME", False)]


def _iou(a, b):
    a, b = a > 0, b > 0
    return float((a & b).sum().item() + 1e-9) / float((a | b).sum().item() + 1e-9)


@pytest.fixture(scope="module")
def bf16_net(model_args, short_sd):
    from emip_amd import nn_base
    from emip_amd.model.EMIP_short.model import CoUpdater
    nn_base.set_default_dtype(torch.bfloat16)
    try:
        net = CoUpdater(model_args)
        net.load_state_dict(short_sd)
        net = net.to("cuda:0").eval()
        im1, im2 = synthetic_pair(4, seed=321)
        im1, im2 = im1.cuda(), im2.cuda()
        with torch.no_grad():
            base = net(im1, im2)[0].float()
            again = net(im1, im2)[0].float()
        yield net, im1, im2, base, (again - base).abs().max().item(), _iou(again, base)
    finally:
        nn_base.set_default_dtype(torch.float32)


@pytest.mark.parametrize("modname,attr,value", INFERENCE)
def test_inference_switch(bf16_net, modname, attr, value):
    from emip_amd import nn_base
    net, im1, im2, base, jit, iou_jit = bf16_net
    mod = importlib.import_module(modname)
    old = getattr(mod, attr)
    nn_base.set_default_dtype(torch.bfloat16)
    try:
        setattr(mod, attr, value)
        with torch.no_grad():
            out = net(im1, im2)[0].float()
    finally:
        setattr(mod, attr, old)
        nn_base.set_default_dtype(torch.float32)
    d, iou = (out - base).abs().max().item(), _iou(out, base)
    print(f"  {modname.split('.')[-1]}.{attr} = {value!r}: max |dlogit| {d:.4f} (two default runs: {jit:.4f}), IoU {iou:.5f} ({iou_jit:.5f})")
    assert torch.isfinite(out).all()
    # round 4: two runs of ONE path are bit-identical (jit = 0); another path rounds at other places: bf16 noise, bounded by a
    # share of the logit range (measured 0.27 .. 0.49 on logits spanning 8)
    span = (base.max() - base.min()).item()
    # (mask IoU of the bf16 mode against the f32 mode is 0.977 on these pairs: two bf16 paths may sit that far apart twice over)
    assert d <= max(2.5 * jit + 0.15, 0.08 * span) and iou >= iou_jit - 0.045


@pytest.mark.parametrize("modname,attr,value", TRAINING)
def test_training_switch(model_args, short_sd, modname, attr, value):
    """one bf16 training step (batch 2) with the switch flipped: loss and a sample of gradients against the default"""
    from emip_amd import nn_base
    from emip_amd.model.EMIP_short.model import CoUpdater
    from emip_amd.train import build_optimizer, freeze_like_reference, train_step
    mod = importlib.import_module(modname)
    old = getattr(mod, attr)
    # well-conditioned gradients only: the injector's (flow loss through the random-weight GMFlow) correlate with the f32 mode at
    # cosine 0.2-0.4 in EVERY bf16 variant (tools/dbg_wattn_grad.py), so they bound nothing; the attention-backward kernels
    # themselves are pinned against torch autograd in test_wattn_gpu.py / test_match_gpu.py / test_dw_bwd_gpu.py
    names = ["backbone.feat_net.pvtv2_en.block3.5.mlp.dwconv.dwconv.weight", "backbone.feat_net.pvtv2_en.block2.1.attn.q.weight",
             "backbone.feat_net.pvtv2_en.block1.0.mlp.fc1.weight", "dr1.reduce.0.conv.weight"]

    def grads(flag):
        nn_base.set_default_dtype(torch.bfloat16)
        try:
            setattr(mod, attr, flag)
            torch.manual_seed(0)
            net = CoUpdater(model_args)
            net.load_state_dict(short_sd)
            net = freeze_like_reference(net.to("cuda:0").train())
            for m in net.modules():
                if hasattr(m, "drop_path_rate"):
                    m.drop_path_rate = 0.0
            opt = build_optimizer(net, lr=0.0)
            im1, im2 = synthetic_pair(2, seed=11)
            gt = synthetic_gt(2, seed=11)
            loss = train_step(net, opt, None, im1.cuda(), im2.cuda(), gt.cuda())
            ps = dict(net.named_parameters())
            return [float(x) for x in loss], {n: ps[n].grad.detach().float().clone() for n in names if n in ps and ps[n].grad is not None}
        finally:
            setattr(mod, attr, old)
            nn_base.set_default_dtype(torch.float32)

    l0, g0 = grads(old)
    l0b, g0b = grads(old)
    l1, g1 = grads(value)
    assert len(g0) >= 3
    for n in g0:
        ref = g0[n].abs().max().item() + 1e-12
        jit = (g0b[n] - g0[n]).abs().max().item() / ref
        d = (g1[n] - g0[n]).abs().max().item() / ref
        cos = torch.nn.functional.cosine_similarity(g1[n].flatten(), g0[n].flatten(), dim=0).item()
        print(f"  {attr} = {value!r} {n}: rel {d:.4f} (two default runs: {jit:.4f}), cosine {cos:.4f}")
        # the bf16 step is bimodal (tests/test_repack_gpu.py): two default runs can coincide to 1e-7 while a third lands 10-20 %
        # away, so the band has a floor; the direction of the gradient is the sharper check against a wrong backward
        assert d <= 3.0 * jit + 0.3 and cos >= 0.95, (n, d, jit, cos)
    assert abs(l1[0] - l0[0]) <= 3.0 * abs(l0b[0] - l0[0]) + 0.05 * abs(l0[0])
