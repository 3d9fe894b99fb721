"""BASELINE.json configs at their full size (bf16): the batch-32 training step equals the accumulation of the same pairs in
chunks; the packed-weight caches follow the fused optimizer (step 2 uses the weights step 1 wrote); 8 bf16 video streams with
a full 5-frame window agree with the f32 path and are independent of each other."""
import numpy as np
import pytest
import torch

from emip_amd.filler import synthetic_gt, synthetic_pair

pytestmark = pytest.mark.gpu
NAMES = ["decoder.conv5.weight", "backbone.feat_net.pvtv2_en.block3.20.mlp.fc2.weight", "conv_corr.3.weight",
         "injector1.transformer.ffn.project_out.weight", "dr2.reduce.0.conv.weight"]


def _net(model_args, short_sd, dtype, train=True, drop=False):
    from emip_amd import nn_base
    from emip_amd.model.EMIP_short.model import CoUpdater
    from emip_amd.train import freeze_like_reference
    nn_base.set_default_dtype(dtype)
    net = CoUpdater(model_args)
    net.load_state_dict(short_sd)
    net = freeze_like_reference(net.to("cuda:0"))
    net = net.train() if train else net.eval()
    if not drop:
        for m in net.modules():
            if hasattr(m, "drop_path_rate"):
                m.drop_path_rate = 0.0
    return net


def _loss(net, im1, im2, gt, scale=1.0):
    from emip_amd.loss.loss_pred import hybrid_e_loss
    mask, fw, bw = net(im1, im2)
    w = torch.linspace(-1, 1, 352 * 352, device=im1.device).view(1, 1, 352, 352) / (352 * 352)
    return (hybrid_e_loss(mask, gt) + sum((f * w).mean(0).sum() + (b_ * w.flip(-1)).mean(0).sum() for f, b_ in zip(fw, bw))) * scale


def test_batch32_bf16_training_step_matches_the_f32_parity_mode(model_args, short_sd):
    """configs[2] at full size: forward + both losses + backward over 32 pairs in bf16 against the SAME step in the f32 parity
    mode (same batch, hence the same BatchNorm batch statistics -- a chunked accumulation would change them): losses, and the
    gradients of five parameters spread over decoder / PVT stage 3 / conv_corr / injector1 / dr2.  bf16 gradients repeat to
    ~2-3 % run to run (atomics order on bf16 activations), so the bound is a bf16 one."""
    from emip_amd import nn_base
    from emip_amd.loss.loss_flow import unFlowLoss
    from emip_amd.loss.loss_pred import hybrid_e_loss
    try:
        B = 32
        im1, im2 = synthetic_pair(B, seed=4242)
        gt = synthetic_gt(B, seed=4242)
        im1, im2, gt = im1.cuda(), im2.cuda(), gt.cuda()
        res = {}
        for dt in (torch.float32, torch.bfloat16):
            net = _net(model_args, short_sd, dt)
            torch.cuda.reset_peak_memory_stats()
            with torch.enable_grad():
                mask, fw, bw = net(im1, im2)
                lp = hybrid_e_loss(mask, gt)
                lf = unFlowLoss().compute_loss([torch.cat((fw[i], bw[i]), 1) for i in range(len(fw))], torch.cat((im1, im2), 1))[0]
                # gradients: hybrid_e_loss + a smooth functional of the flows (the photometric loss is piecewise -- bilinear
                # cell, |.|, SSIM clamp -- and its gradient moves by ~10 % between two f32 runs already)
                w = torch.linspace(-1, 1, 352 * 352, device=im1.device).view(1, 1, 352, 352) / (352 * 352)
                (lp + sum((f * w).mean(0).sum() + (b_ * w.flip(-1)).mean(0).sum() for f, b_ in zip(fw, bw))).backward()
            p = dict(net.named_parameters())
            res[dt] = (lp.item(), lf.item(), {n: p[n].grad.detach().float().clone() for n in NAMES},
                       torch.cuda.max_memory_allocated() / 2 ** 30)
            del net, p, mask, fw, bw
            torch.cuda.empty_cache()
        (lp32, lf32, g32, mem32), (lp16, lf16, g16, mem16) = res[torch.float32], res[torch.bfloat16]
        print(f"  losses f32 {lp32:.4f} + {lf32:.4f}, bf16 {lp16:.4f} + {lf16:.4f}; peak memory {mem32:.1f} / {mem16:.1f} GiB")
        assert abs(lp16 - lp32) < 2e-2 * max(1.0, abs(lp32)) and abs(lf16 - lf32) < 5e-2 * max(1.0, abs(lf32))
        for n in NAMES:
            rel = (g16[n] - g32[n]).norm().item() / max(g32[n].norm().item(), 1e-12)
            print(f"  {n}: bf16 vs f32 rel L2 {rel:.3e}  (|g| {g32[n].norm().item():.3e})")
            # bf16 residual stream + bf16 activations: the decoder-side gradient agrees to < 1 %, 20 residual blocks deep to ~20 %
            assert rel < (0.03 if n.startswith("decoder.") else 0.3), (n, rel)
        assert mem16 < 80
    finally:
        nn_base.set_default_dtype(torch.float32)


def test_second_training_step_uses_the_weights_the_fused_optimizer_wrote(model_args, short_sd):
    """the fused clamp+AdamW writes parameters through raw pointers: the packed-weight caches (bf16 copies, conv / dgrad
    packs, folded norms) must follow.  After one step, the SAME module's forward == a fresh module loaded with its
    state_dict; and a high learning rate makes the loss move between steps."""
    from emip_amd import nn_base
    from emip_amd.train import build_optimizer, train_step
    try:
        im1, im2 = synthetic_pair(2, seed=7)
        gt = synthetic_gt(2, seed=7)
        im1, im2, gt = im1.cuda(), im2.cuda(), gt.cuda()
        net = _net(model_args, short_sd, torch.bfloat16)
        opt = build_optimizer(net, lr=2e-3, weight_decay=1e-7, clip=0.5)
        with torch.no_grad():
            m_before = net(im1, im2)[0].float()
        l0 = train_step(net, opt, None, im1, im2, gt)[1].item()
        sd1 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        with torch.no_grad():
            m_same = net(im1, im2)[0].float()              # train mode, batch statistics: same module, after the step
        fresh = _net(model_args, sd1, torch.bfloat16)
        with torch.no_grad():
            m_fresh = fresh(im1, im2)[0].float()
        moved = (m_same - m_before).abs().max().item()
        diff = (m_same - m_fresh).abs().max().item()
        print(f"  one step at lr 2e-3 moves the mask logits by {moved:.2f}; same module vs fresh module on its state_dict: {diff:.3f}")
        # a stale pack would leave `moved` between the two; what is left is the bf16 forward's run-to-run jitter (~0.4: f32
        # atomics in the statistics feeding bf16 roundings)
        assert moved > 3.0 and diff < 0.08 * moved, (moved, diff)
        # and the eval-mode packs (BatchNorm folded into the convs) follow too
        with torch.no_grad():
            e_same, e_fresh = net.eval()(im1, im2)[0].float(), fresh.eval()(im1, im2)[0].float()
        assert (e_same - e_fresh).abs().max().item() < 0.08 * moved
        net.train()
        losses = [l0] + [train_step(net, opt, None, im1, im2, gt)[1].item() for _ in range(6)]
        print("hybrid_e_loss over 7 steps at lr 2e-3:", [round(x, 4) for x in losses])
        assert losses[-1] < losses[0] - 0.05, losses
    finally:
        nn_base.set_default_dtype(torch.float32)


def test_eight_bf16_streams_full_window_match_f32_and_are_independent(model_args, long_sd):
    """configs[3]: 8 streams, 5-frame window full.  bf16 masks vs the f32 path (IoU, logits), and stream 3 of the batch of 8
    equals the same video run alone."""
    from emip_amd import nn_base
    from emip_amd.model.EMIP_long.model_long import Model_long

    def build(dtype):
        nn_base.set_default_dtype(dtype)
        n = Model_long(model_args)
        n.load_state_dict(long_sd)
        return n.to("cuda:0").eval()

    def run(net, frames, upto):
        k = v = None
        m = None
        with torch.no_grad():
            for i in range(upto):
                m, k, v = net.forward_streams(frames[i], frames[i + 1], i, k, v)
        return m, k

    try:
        S, T = 8, 8
        frames = [torch.cat([synthetic_pair(1, seed=900 + s, shift=(t - 3, 2 - t))[1] for s in range(S)], 0).cuda()
                  for t in range(T + 1)]
        n32 = build(torch.float32)
        m32, k32 = run(n32, frames, T)
        del n32
        torch.cuda.empty_cache()
        n16 = build(torch.bfloat16)
        m16, k16 = run(n16, frames, T)
        assert k16.shape[3] == 5 and k32.shape[3] == 5                       # window saturated
        a, b = m16.float() > 0, m32 > 0
        iou = (a & b).sum().item() / max((a | b).sum().item(), 1)
        dlog = (m16.float() - m32).abs().max().item()
        print(f"  8-stream bf16 vs f32: IoU {iou:.4f}, max |dlogit| {dlog:.3f} on logits up to {m32.abs().max().item():.1f}")
        # 8 recurrent frames in bf16 (memory fed back): masks agree on 96 % of the union, logits to ~3 % of their range
        assert iou > 0.95 and dlog < 0.08 * max(1.0, m32.abs().max().item())
        one = [f[3:4] for f in frames]
        m1, _ = run(n16, one, T)
        # Stream 3 alone against stream 3 inside the batch of 8: the same arithmetic, but the batch size changes tile choices
        # and the key-split count of the memory read (ops.KV_SPLIT_TARGET workgroups), i.e. the f32 summation order under
        # bf16 storage -- amplified over 8 recurrent frames: ~2.6 % of the logit range at most, masks as close as bf16 is to f32
        d1 = (m1.float() - m16[3:4].float()).abs().max().item()
        a1, b1 = m1.float() > 0, m16[3:4].float() > 0
        iou1 = (a1 & b1).sum().item() / max((a1 | b1).sum().item(), 1)
        print(f"  stream 3 alone vs in the batch: max |dlogit| {d1:.3f}, IoU {iou1:.4f}")
        assert d1 < 5e-2 * max(1.0, m32.abs().max().item()) and iou1 > 0.95
    finally:
        nn_base.set_default_dtype(torch.float32)
