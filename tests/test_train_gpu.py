"""Training path (autograd over HIP kernels): parameter and input gradients against torch autograd through the
CPU oracle on the same weights and inputs (f32 mode)."""
import pytest
import torch

from emip_amd.filler import synthetic_pair

pytestmark = pytest.mark.gpu
PVT = "backbone.feat_net.pvtv2_en."


def _relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def test_pvt_backbone_gradients_vs_oracle(short_sd):
    from emip_amd import nn_base
    from emip_amd.lib.pvt_v2 import pvt_v2_b5
    from emip_amd.nn_base import to_cl
    from oracle import emip_oracle as O
    nn_base.set_default_dtype(torch.float32)
    sd = {k[len(PVT):]: v for k, v in short_sd.items() if k.startswith(PVT)}
    net = pvt_v2_b5()
    net.load_state_dict(sd)
    net = net.to("cuda:0").train()
    for m in net.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    img, _ = synthetic_pair(1, seed=321)
    gs = [torch.randn(1, c, s, s, generator=torch.Generator().manual_seed(10 + i))
          for i, (c, s) in enumerate(((64, 88), (128, 44), (320, 22), (512, 11)))]
    # ---- HIP
    with torch.enable_grad():
        outs = net.run(to_cl(img.cuda(), torch.float32, 8))
        loss = sum((o * g.permute(0, 2, 3, 1).cuda()).sum() for o, g in zip(outs, gs))
        loss.backward()
    # ---- oracle
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    routs = O.pvt_forward(img, {"p." + k: v for k, v in ref.items()}, "p")
    rloss = sum((o * g).sum() for o, g in zip(routs, gs))
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 1e-2 * max(1.0, abs(rloss.item()))
    names = ["patch_embed1.proj.weight", "patch_embed1.norm.bias", "block1.0.attn.sr.weight", "block1.0.attn.q.weight",
             "block1.2.mlp.dwconv.dwconv.weight", "block2.0.attn.kv.weight", "block2.3.mlp.fc1.bias",
             "patch_embed3.proj.weight", "block3.0.norm1.weight", "block3.17.attn.proj.weight",
             "block3.39.mlp.fc2.weight", "block3.39.attn.norm.weight", "patch_embed4.proj.bias",
             "block4.1.attn.kv.bias", "block4.2.mlp.dwconv.dwconv.bias", "norm4.weight", "norm2.bias"]
    p = dict(net.named_parameters())
    worst = 0.0
    for n in names:
        assert p[n].grad is not None, n
        e = _relerr(p[n].grad, ref[n].grad)
        worst = max(worst, e)
        assert e < 5e-3, f"{n}: relative gradient error {e}"
    missing = [n for n, q in p.items() if q.grad is None]
    assert not missing, missing[:5]
    print("worst relative gradient error", worst)


def _grads_close(net_params, ref, names, tol):
    worst = 0.0
    for n in names:
        g = net_params[n].grad
        assert g is not None, n
        r = ref[n].grad
        # a conv bias in front of a train-mode BatchNorm has an exactly-zero true gradient: floor the scale
        e = ((g.detach().float().cpu() - r).abs().max() / max(r.abs().max().item(), 1e-3)).item()
        worst = max(worst, e)
        assert e < tol, f"{n}: relative gradient error {e}"
    return worst


def test_injector_gradients_vs_oracle(short_sd):
    from emip_amd import nn_base
    from emip_amd.model.EMIP_short.motion.PromptInteract import Injector
    from emip_amd.nn_base import to_cl
    from oracle import emip_oracle as O
    nn_base.set_default_dtype(torch.float32)
    sd = {k[len("injector."):]: v for k, v in short_sd.items() if k.startswith("injector.")}
    net = Injector()
    net.load_state_dict(sd)
    net = net.to("cuda:0").train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 128, 44, 44, generator=g)
    y = torch.randn(2, 128, 44, 44, generator=g) * 2
    go = torch.randn(2, 128, 44, 44, generator=g)
    xc = to_cl(x.cuda(), torch.float32).requires_grad_(True)
    yc = to_cl(y.cuda(), torch.float32).requires_grad_(True)
    with torch.enable_grad():
        out = net.run(xc, yc)
        (out * go.permute(0, 2, 3, 1).cuda()).sum().backward()
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    ro = O.injector_forward(xr, yr, {"i." + k: v for k, v in ref.items()}, "i")
    (ro * go).sum().backward()
    assert _relerr(out.permute(0, 3, 1, 2), ro) < 1e-4
    assert _relerr(xc.grad.permute(0, 3, 1, 2), xr.grad) < 2e-3 and _relerr(yc.grad.permute(0, 3, 1, 2), yr.grad) < 2e-3
    worst = _grads_close(dict(net.named_parameters()), ref, list(sd.keys()), 5e-3)
    print("injector worst relative gradient error", worst)


def test_decoder_convcorr_gradients_vs_oracle(short_sd, model_args):
    """dr1/dr2/dr3 + NCD decoder + conv_corr in train mode (BatchNorm batch statistics) + hybrid-free scalar loss"""
    from emip_amd import nn_base
    from emip_amd.model.EMIP_short.model import CoUpdater
    from emip_amd.nn_base import to_cl
    from oracle import emip_oracle as O
    nn_base.set_default_dtype(torch.float32)
    net = CoUpdater(model_args)
    net.load_state_dict(short_sd)
    net = net.to("cuda:0").train()
    g = torch.Generator().manual_seed(6)
    B = 2
    f1 = torch.randn(B, 128, 44, 44, generator=g)
    f2 = torch.randn(B, 320, 22, 22, generator=g)
    f3 = torch.randn(B, 512, 11, 11, generator=g)
    corr = torch.randn(B, 1936, 44, 44, generator=g) * 3       # [B, tgt, h, w(src)] like the reference
    gm = torch.randn(B, 1, 352, 352, generator=g)
    gc = torch.randn(B, 128, 44, 44, generator=g)
    ins = [to_cl(t.cuda(), torch.float32).requires_grad_(True) for t in (f1, f2, f3)]
    corr_cl = corr.permute(0, 2, 3, 1).reshape(B, 1936, 1936).contiguous().cuda().requires_grad_(True)   # [B, src, tgt]
    with torch.enable_grad():
        mask = net.decoder.run(net.dr3.run(ins[2]), net.dr2.run(ins[1]), net.dr1.run(ins[0]))
        cc = net.run_conv_corr(corr_cl)
        loss = (mask * gm.cuda()).sum() + (cc * gc.permute(0, 2, 3, 1).cuda()).sum()
        loss.backward()
    keys = [k for k in short_sd if k.split(".")[0] in ("dr1", "dr2", "dr3", "decoder", "conv_corr") and
            "running" not in k and "num_batches" not in k]
    ref = {k: short_sd[k].clone().requires_grad_(True) for k in keys}
    full = dict(short_sd)
    full.update(ref)
    rin = [t.clone().requires_grad_(True) for t in (f1, f2, f3)]
    rcorr = corr.clone().requires_grad_(True)
    rmask = O.ncd_forward(O.dim_reduction(rin[2], full, "dr3", True), O.dim_reduction(rin[1], full, "dr2", True),
                          O.dim_reduction(rin[0], full, "dr1", True), full, "decoder", True)
    rcc = O.conv_corr_forward(rcorr, full, "conv_corr", True)
    rloss = (rmask * gm).sum() + (rcc * gc).sum()
    rloss.backward()
    assert _relerr(mask, rmask) < 1e-4 and _relerr(cc.permute(0, 3, 1, 2), rcc) < 1e-4
    # ReLU masks of near-zero pre-activations may flip between the two implementations: bounds are looser than
    # for the ReLU-free PVT / injector chains
    for a, b in zip(ins, rin):
        e = _relerr(a.grad.permute(0, 3, 1, 2), b.grad)
        print("input grad rel err", e)
        assert e < 2e-2
    dc = corr_cl.grad.view(B, 44, 44, 1936).permute(0, 3, 1, 2)
    print("corr grad rel err", _relerr(dc, rcorr.grad))
    assert _relerr(dc, rcorr.grad) < 2e-2
    worst = _grads_close(dict(net.named_parameters()), ref, keys, 2e-2)
    print("decoder/conv_corr worst relative gradient error", worst)


def test_gmflow_input_gradients_vs_oracle(short_sd):
    """frozen GMFlow stream: gradient w.r.t. the prompted features through both flow predictions and the correlation"""
    import json
    import os
    from emip_amd import nn_base
    from emip_amd.model.EMIP_short.motion.gmflow.gmflow import GMFlow
    from oracle import emip_oracle as O
    nn_base.set_default_dtype(torch.float32)
    args = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_args.json")))
    sd = {k[len("GMFlow."):]: v for k, v in short_sd.items() if k.startswith("GMFlow.")}
    net = GMFlow(feature_channels=128, args=args)
    net.load_state_dict(sd)
    net = net.to("cuda:0").train()
    for q in net.parameters():
        q.requires_grad_(False)
    B, h, w, C = 1, 44, 44, 128
    g = torch.Generator().manual_seed(77)
    ab = torch.randn(2 * B, C, h, w, generator=g) * 0.5
    w_lr = torch.randn(2 * B, 2, 8 * h, 8 * w, generator=g) / (8 * h * 8 * w)
    w_up = torch.randn(2 * B, 2, 8 * h, 8 * w, generator=g) / (8 * h * 8 * w)
    w_corr = torch.randn(B, h * w, h * w, generator=g) / (h * w)
    # ---- HIP
    x = ab.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    with torch.enable_grad():
        preds, corr = net.run_train(x)
        loss = (preds[0] * w_lr.cuda()).sum() + (preds[1] * w_up.cuda()).sum() + (corr * w_corr.cuda()).sum()
        loss.backward()
    # ---- oracle
    xr = ab.clone().requires_grad_(True)
    rsd = {"GMFlow." + k: v for k, v in sd.items()}
    fw, bw, rcorr, _ = O.gmflow_forward(xr[:B], xr[B:], rsd, "GMFlow", training=True)
    # oracle corr: [B, tgt, h, w(src)] view of [B, src, tgt]
    rc = rcorr.permute(0, 2, 3, 1).reshape(B, h * w, h * w)
    rloss = ((torch.cat((fw[0], bw[0])) * w_lr).sum() + (torch.cat((fw[1], bw[1])) * w_up).sum() + (rc * w_corr).sum())
    rloss.backward()
    print("loss", loss.item(), rloss.item())
    assert abs(loss.item() - rloss.item()) < 1e-3 * max(1.0, abs(rloss.item()))
    e = _relerr(x.grad.permute(0, 3, 1, 2), xr.grad)
    print("gmflow input-gradient relative error", e)
    assert e < 5e-3


def _short_net(model_args, short_sd):
    from emip_amd.model.EMIP_short.model import CoUpdater
    from emip_amd.train import freeze_like_reference
    net = CoUpdater(model_args)
    net.load_state_dict(short_sd)
    return freeze_like_reference(net.to("cuda:0").train())


def test_full_training_step_gradients_vs_oracle(model_args, short_sd):
    """forward + hybrid_e_loss + unFlowLoss + backward of the whole EMIP-short model (train mode, DropPath masks forced
    to the oracle's) against torch autograd through the CPU oracle"""
    from emip_amd import nn_base
    from emip_amd.filler import synthetic_gt
    from emip_amd.loss.loss_flow import unFlowLoss
    from emip_amd.loss.loss_pred import hybrid_e_loss
    from oracle import emip_oracle as O
    nn_base.set_default_dtype(torch.float32)
    net = _short_net(model_args, short_sd)
    B = 1
    im1, im2 = synthetic_pair(B, seed=99)
    gt = synthetic_gt(B, seed=99)
    # DropPath factors: one draw per (stage, block, branch) and per image of the 2B batch
    g = torch.Generator().manual_seed(3)
    blocks = [(i, j) for i in range(4) for j in range(len(getattr(net.backbone.feat_net.pvtv2_en, f"block{i + 1}")))]
    dm = {}
    for (i, j) in blocks:
        blk = getattr(net.backbone.feat_net.pvtv2_en, f"block{i + 1}")[j]
        keep = 1.0 - blk.drop_path_rate
        forced = {}
        for tag in ("attn", "mlp"):
            s = torch.floor(keep + torch.rand(2 * B, generator=g)) / keep
            forced[tag] = s
            dm[(i, j, tag)] = s
        blk.forced_drop = forced
    names = ["backbone.feat_net.pvtv2_en.patch_embed1.proj.weight", "backbone.feat_net.pvtv2_en.block1.0.attn.q.weight",
             "backbone.feat_net.pvtv2_en.block2.3.mlp.fc1.weight", "backbone.feat_net.pvtv2_en.block3.20.attn.kv.weight",
             "backbone.feat_net.pvtv2_en.block4.2.mlp.fc2.weight", "backbone.feat_net.pvtv2_en.norm4.weight",
             "injector.transformer.attn.temperature", "injector.transformer.attn.q.weight",
             "injector.transformer.ffn.project_out.weight", "injector1.transformer.attn.kv.weight",
             "conv_corr.0.weight", "conv_corr.3.weight", "conv_corr.1.weight", "dr1.reduce.0.conv.weight",
             "dr3.reduce.1.bn.weight", "decoder.conv_upsample5.conv.weight", "decoder.conv5.weight",
             "decoder.conv5.bias"]
    gw = torch.Generator().manual_seed(8)
    wfl = [torch.randn(2 * B, 2, 352, 352, generator=gw) / (352 * 352) for _ in range(2)]
    p = dict(net.named_parameters())

    def collect(params, get):
        out = {n: get(params[n]).detach().float().cpu().clone() for n in names}
        for q in params.values():
            if torch.is_tensor(q) and q.grad is not None:
                q.grad = None
        return out

    # ---- HIP: one forward, two backward passes (A: hybrid + a smooth linear functional of the flows, B: the real loss)
    with torch.enable_grad():
        mask, fw, bw = net(im1.cuda(), im2.cuda())
        lp = hybrid_e_loss(mask, gt.cuda())
        lf = unFlowLoss().compute_loss([torch.cat((fw[i], bw[i]), 1) for i in range(len(fw))],
                                       torch.cat((im1, im2), 1).cuda())[0]
        lin = sum((torch.cat((fw[i], bw[i]), 0) * wfl[i].cuda()).sum() for i in range(2))
        (lp + lin).backward(retain_graph=True)
        frozen_with_grad = [n for n, q in p.items() if not q.requires_grad and q.grad is not None]
        missing = {n for n, q in p.items() if q.requires_grad and q.grad is None}
        gA = collect(p, lambda q: q.grad)
        (lp + lf).backward()
        gB = collect(p, lambda q: q.grad)
    # ---- oracle (the two frames go through the backbone separately there: split the masks)
    ref = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in short_sd.items()}
    dm1 = {k: v[:B] for k, v in dm.items()}
    dm2 = {k: v[B:] for k, v in dm.items()}
    rmask, rfw, rbw = O.short_forward(im1, im2, ref, training=True, drop_masks=(dm1, dm2))
    rlp = O.hybrid_e_loss(rmask, gt)
    rlf = O.unflow_loss([torch.cat((rfw[i], rbw[i]), 1) for i in range(len(rfw))], torch.cat((im1, im2), 1))
    rlin = sum((torch.cat((rfw[i], rbw[i]), 0) * wfl[i]).sum() for i in range(2))
    (rlp + rlin).backward(retain_graph=True)
    ref_missing = {n for n, q in ref.items() if q.is_floating_point() and q.requires_grad and q.grad is None}
    rA = collect(ref, lambda q: q.grad)
    (rlp + rlf).backward()
    rB = collect(ref, lambda q: q.grad)
    print("losses", lp.item(), rlp.item(), lf.item(), rlf.item(), lin.item(), rlin.item())
    assert abs(lp.item() - rlp.item()) < 1e-3 and abs(lf.item() - rlf.item()) < 1e-3
    assert abs(lin.item() - rlin.item()) < 1e-3 * max(1.0, abs(rlin.item()))
    print("max |flow_hip - flow_oracle| (px)", (fw[1].detach().cpu() - rfw[1].detach()).abs().max().item())

    def worst_err(g, r, tag):
        worst = 0.0
        for n in names:
            e = ((g[n] - r[n]).abs().max() / max(r[n].abs().max().item(), 1e-6)).item()
            print(f"  [{tag}] {n}: {e:.3e} (scale {r[n].abs().max().item():.3e})")
            worst = max(worst, e)
        return worst
    wA = worst_err(gA, rA, "smooth")
    wB = worst_err(gB, rB, "real")
    print("full-step worst relative gradient error: smooth flow functional", wA, " real unFlowLoss", wB)
    # A: every kernel on the backward path is exercised with a smooth objective; what is left is ReLU / BatchNorm
    # mask flips behind conv_corr.0 (1-3 % of a 1.6e-3 gradient, varies run to run with the f32 atomics)
    assert wA < 5e-2
    # B: the photometric loss is piecewise (bilinear cell of the warp, |.|, SSIM clamp): 1e-4 px of forward rounding
    # difference moves a few pixels across a kink, which perturbs the small flow-loss gradients reaching the injector
    assert wB < 0.3
    assert not frozen_with_grad
    # parameters the reference never reaches (dead modules, GMFlow adaptors) stay without gradient on both sides
    assert missing == {n for n in ref_missing if n in p and p[n].requires_grad}, (sorted(missing ^ ref_missing))[:6]


def test_train_step_runs_and_updates(model_args, short_sd):
    """emip_amd.train.train_step: loss finite, trainable parameters move, frozen ones do not (bf16 compute)"""
    from emip_amd import nn_base
    from emip_amd.filler import synthetic_gt
    from emip_amd.train import build_optimizer, train_step
    nn_base.set_default_dtype(torch.bfloat16)
    try:
        net = _short_net(model_args, short_sd)
        opt = build_optimizer(net, lr=1e-5, weight_decay=1e-7, clip=0.5)
        im1, im2 = synthetic_pair(2, seed=5)
        gt = synthetic_gt(2, seed=5)
        w_tr = net.decoder.conv5.weight.detach().clone()
        w_fr = net.GMFlow.upsampler[0].weight.detach().clone()
        losses = [train_step(net, opt, None, im1.cuda(), im2.cuda(), gt.cuda())[0].item() for _ in range(2)]
        print("bf16 train losses", losses)
        assert all(l == l and abs(l) < 1e4 for l in losses)
        assert (net.decoder.conv5.weight.detach() - w_tr).abs().max().item() > 0
        assert torch.equal(net.GMFlow.upsampler[0].weight.detach(), w_fr)
    finally:
        nn_base.set_default_dtype(torch.float32)


def test_training_step_gradients_vs_reference_golden(model_args, short_sd, golden):
    """the HIP training step against gradients the REFERENCE itself produced (tests/golden/short_train_grads.npz,
    oracle/make_golden_short_train.py: train mode, DropPath off, hybrid_e_loss + unFlowLoss)"""
    import numpy as np
    from emip_amd import nn_base
    from emip_amd.filler import synthetic_gt
    from emip_amd.loss.loss_flow import unFlowLoss
    from emip_amd.loss.loss_pred import hybrid_e_loss
    nn_base.set_default_dtype(torch.float32)
    g = golden("short_train_grads.npz")
    net = _short_net(model_args, short_sd)
    for m in net.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    im1, im2 = synthetic_pair(1, seed=99)
    gt = synthetic_gt(1, seed=99)
    with torch.enable_grad():
        mask, fw, bw = net(im1.cuda(), im2.cuda())
        lp = hybrid_e_loss(mask, gt.cuda())
        lf = unFlowLoss().compute_loss([torch.cat((fw[i], bw[i]), 1) for i in range(len(fw))],
                                       torch.cat((im1, im2), 1).cuda())[0]
        (lp + lf).backward()
    assert abs(lp.item() - float(g["loss_pred"])) < 1e-3 and abs(lf.item() - float(g["loss_flow"])) < 1e-3
    assert (mask.detach().cpu()[:, :, ::4, ::4] - torch.from_numpy(g["mask"])).abs().max().item() < 2e-3
    p = dict(net.named_parameters())

    def tol(n):
        # the photometric loss is piecewise: parameters that only see it through the flows repeat to ~10 % between two
        # f32 forwards that differ by 0.03 px (see test_full_training_step_gradients_vs_oracle); the rest is tight
        if n.startswith("injector."):
            return 0.3
        if n == "conv_corr.0.weight" or "block1." in n or "block2." in n or "patch_embed1" in n:
            return 0.06
        return 1e-2
    for i, n in enumerate(str(x) for x in g["names"]):
        gr = p[n].grad
        assert gr is not None, n
        ref_stats, ref_head = g["g%d_stats" % i], g["g%d_head" % i]
        err = np.abs(gr.detach().reshape(-1)[:64].cpu().numpy() - ref_head).max() / max(ref_stats[2], 1e-12)
        l2 = abs(gr.double().pow(2).sum().sqrt().item() - ref_stats[1]) / max(ref_stats[1], 1e-12)
        print(f"  {n}: head {err:.2e} l2 {l2:.2e}")
        assert err < tol(n) and l2 < tol(n), (n, err, l2)
