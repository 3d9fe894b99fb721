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
