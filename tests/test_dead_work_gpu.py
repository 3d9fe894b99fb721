"""The two places where the product path does less arithmetic than the reference's forward writes down, held against the literal
order in the f32 mode (no rounding noise to hide behind):
  * PVT_DEEP_ONE_FRAME: model.py:87-92 runs the backbone on both frames and reads fea_2[0] only -- stages 3 and 4 of the second
    frame are not computed; outputs AND parameter gradients must be those of the run that computes them;
  * CONV_CORR_FACTORED: conv_corr.0 of the rank-128 correlation volume (model.py:59,96; matching.py:16-20) through its factors."""
import importlib

import pytest
import torch

from emip_amd.filler import synthetic_gt, synthetic_pair

pytestmark = pytest.mark.gpu
MODEL = "emip_amd.model.EMIP_short.model"


def _f32_net(model_args, short_sd, train):
    from emip_amd.model.EMIP_short.model import CoUpdater
    net = CoUpdater(model_args)
    net.load_state_dict(short_sd)
    net = net.to("cuda:0")
    return net.train() if train else net.eval()


@pytest.mark.parametrize("attr", ["PVT_DEEP_ONE_FRAME", "CONV_CORR_FACTORED"])
def test_f32_outputs_equal_the_literal_order(model_args, short_sd, attr):
    from emip_amd import nn_base
    mod = importlib.import_module(MODEL)
    nn_base.set_default_dtype(torch.float32)
    net = _f32_net(model_args, short_sd, False)
    im1, im2 = synthetic_pair(2, seed=77)
    im1, im2 = im1.cuda(), im2.cuda()
    outs = {}
    for v in (True, False):
        old = getattr(mod, attr)
        setattr(mod, attr, v)
        try:
            with torch.no_grad():
                mask, fw, bw = net(im1, im2)
            outs[v] = (mask.clone(), fw[-1].clone(), bw[-1].clone(), net.last["conv_corr"].float().clone())
        finally:
            setattr(mod, attr, old)
    for a, b, name, tol in zip(outs[True], outs[False], ("mask", "flow_fw", "flow_bw", "conv_corr"), (2e-4, 2e-3, 2e-3, 2e-4)):
        d = (a - b).abs().max().item()
        print(f"  {attr}: {name} max |difference| {d:.3e} (range {b.abs().max().item():.3g})")
        assert d <= tol * max(1.0, b.abs().max().item()), (attr, name, d)


def test_f32_gradients_equal_the_literal_order(model_args, short_sd):
    """one training backward (both losses, DropPath off) in the f32 mode with the second frame's deep stages dropped and with
    conv_corr.0 through the factors: every parameter gradient equals that of the literal order"""
    from emip_amd import nn_base
    from emip_amd.loss.loss_flow import unFlowLoss
    from emip_amd.loss.loss_pred import hybrid_e_loss
    from emip_amd.train import freeze_like_reference
    mod = importlib.import_module(MODEL)
    nn_base.set_default_dtype(torch.float32)
    im1, im2 = synthetic_pair(2, seed=5)
    gt = synthetic_gt(2, seed=5).cuda()
    im1, im2 = im1.cuda(), im2.cuda()
    grads = {}
    for run, v in enumerate((True, False, False)):       # the literal order twice: its own run-to-run band is the yardstick
        olds = (mod.PVT_DEEP_ONE_FRAME, mod.CONV_CORR_FACTORED)
        mod.PVT_DEEP_ONE_FRAME = mod.CONV_CORR_FACTORED = v
        try:
            net = freeze_like_reference(_f32_net(model_args, short_sd, True))
            for m in net.modules():
                if hasattr(m, "drop_path_rate"):
                    m.drop_path_rate = 0.0
            preds = net(im1, im2)
            pair = [torch.cat((preds[1][i], preds[2][i]), 1) for i in range(len(preds[1]))]
            loss = hybrid_e_loss(preds[0], gt) + unFlowLoss().compute_loss(pair, torch.cat((im1, im2), 1))[0]
            loss.backward()
            grads[run] = ({n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None}, loss.item())
            del net
        finally:
            mod.PVT_DEEP_ONE_FRAME, mod.CONV_CORR_FACTORED = olds
    (ga, la), (gb, lb), (gc, _) = grads[0], grads[1], grads[2]
    assert set(ga) == set(gb)
    assert abs(la - lb) <= 1e-4 * abs(lb), (la, lb)
    rel = {n: ((ga[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-20)).item() for n in gb}
    noise = {n: ((gc[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-20)).item() for n in gb}
    # conv_corr.0.bias sits in front of a BatchNorm: its true gradient is zero, what is left is rounding
    rel.pop("conv_corr.0.bias")
    worst = sorted(((v, noise[n], n) for n, v in rel.items()), reverse=True)[:6]
    print("  largest relative gradient differences (difference, the literal order against itself, name):", worst)
    # The photometric loss is piecewise (bilinear cell, |.|, SSIM clamp) and the forward's f32 atomics move its inputs in the
    # last bit: two runs of the SAME order differ by 5-15 % on the parameters that see the flow loss only (`injector.*`), by a
    # few per cent on the backbone, by 1e-3 on the mask path.  The bound is that band.
    for n, v in rel.items():
        if n.startswith("injector."):
            a, b = ga[n].norm().item(), gb[n].norm().item()
            assert torch.isfinite(ga[n]).all() and 0.5 * b <= a <= 2.0 * b and v <= 0.5, (n, a, b, v)
        else:
            assert v <= 4.0 * noise[n] + (5e-2 if n.startswith("conv_corr.") else 2e-2), (n, v, noise[n])
    tight = sorted(((v, n) for n, v in rel.items() if n.startswith(("decoder.", "dr", "injector1."))), reverse=True)
    print("  mask path (decoder, reductions, injector1): largest", tight[:3], "median %.1e" % tight[len(tight) // 2][0])
    assert tight[0][0] <= 1e-2 and tight[len(tight) // 2][0] <= 1e-4, tight[:3]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 1e-2)])
def test_conv_corr0_fn_matches_the_literal_volume_convolution(dtype, tol):
    """autograd.ConvCorr0Fn (output, token gradient, weight and bias gradient) against torch autograd through the literal
    formulation: corr = F0 F1^T / sqrt(C) viewed [B, n(tgt), h, w(src)] (matching.py:16-20) -> conv2d (model.py:59)"""
    from emip_amd import nn_base
    from emip_amd.autograd import ConvCorr0Fn
    B, h, w, C, cout = 2, 44, 44, 128, 968
    n = h * w
    g = torch.Generator(device="cuda").manual_seed(3)
    tok = (torch.randn(2 * B, n, C, device="cuda", generator=g) * 0.5).to(dtype).requires_grad_(True)
    weight = (torch.randn(cout, n, 3, 3, device="cuda", generator=g) * 0.01).requires_grad_(True)
    bias = torch.randn(cout, device="cuda", generator=g).requires_grad_(True)
    dy = torch.randn(B, h, w, cout, device="cuda", generator=g).to(dtype)
    wr = (weight.detach() * C ** -0.5).permute(0, 2, 3, 1).reshape(cout * 9, n).to(dtype).contiguous()
    nn_base.set_default_dtype(dtype)
    try:
        y = ConvCorr0Fn.apply(tok, weight, bias, wr, wr.t().contiguous(), h, w)
        y.backward(dy)
    finally:
        nn_base.set_default_dtype(torch.float32)
    got = (y.detach().float(), tok.grad.float().clone(), weight.grad.clone(), bias.grad.clone())
    tok.grad = weight.grad = bias.grad = None
    t32 = tok.detach().float().requires_grad_(True)
    corr = torch.matmul(t32[:B], t32[B:].transpose(1, 2)) * C ** -0.5          # [B, src, tgt]
    x = corr.view(B, h, w, n).permute(0, 3, 1, 2)                               # [B, tgt, h, w]
    yr = torch.nn.functional.conv2d(x, weight.to(dtype).float() if dtype != torch.float32 else weight, bias, padding=1)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    ref = (yr.detach().permute(0, 2, 3, 1), t32.grad, weight.grad, bias.grad)
    for a, b, name in zip(got, ref, ("y", "d tokens", "d weight", "d bias")):
        d = ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()
        print(f"  {dtype}: {name} relative max difference {d:.2e}")
        assert d <= tol, (name, d)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C", [(2, 44, 44, 128), (3, 7, 5, 16), (1, 1, 9, 8)])
def test_patch_matrix_and_its_adjoint(dtype, B, H, W, C):
    """emip_im2col3x3 against torch unfold (tap-major columns, zero padding) and emip_col2im3x3 against its autograd adjoint"""
    from emip_amd import ops
    torch.manual_seed(H * W + C)
    x = torch.randn(B, H, W, C, device="cuda").to(dtype)
    pm = ops.im2col3x3(x)                                                   # [B, H W, 9 C]
    unf = torch.nn.functional.unfold(x.float().permute(0, 3, 1, 2), 3, padding=1)          # [B, C * 9, H W], channel-major
    ref = unf.view(B, C, 9, H * W).permute(0, 3, 2, 1).reshape(B, H * W, 9 * C)            # -> tap-major columns
    assert torch.equal(pm.float(), ref.to(dtype).float())
    dy = torch.randn(B, H * W, 9 * C, device="cuda").to(dtype)
    dx = ops.col2im3x3(dy, B, H, W, C)
    xr = x.float().requires_grad_(True)
    u = torch.nn.functional.unfold(xr.permute(0, 3, 1, 2), 3, padding=1).view(B, C, 9, H * W).permute(0, 3, 2, 1).reshape(B, H * W, 9 * C)
    u.backward(dy.float())
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert (dx.float().view(B, H, W, C) - xr.grad).abs().max() <= tol * max(1.0, xr.grad.abs().max().item())
