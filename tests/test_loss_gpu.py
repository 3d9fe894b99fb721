"""Loss-side forward kernels (A17-A20) against the reference's own outputs (tests/golden)."""
import pytest
import torch

from emip_amd.filler import synthetic_gt, synthetic_pair

pytestmark = pytest.mark.gpu


def test_hybrid_and_unflow_micro(golden):
    from emip_amd.loss.loss_flow import unFlowLoss
    from emip_amd.loss.loss_pred import hybrid_e_loss
    g = golden("loss_micro.npz")
    dev = "cuda:0"
    x, y, flow = (torch.from_numpy(g[k]).to(dev) for k in ("x", "y", "flow"))
    h = hybrid_e_loss(torch.from_numpy(g["pred"]).to(dev), torch.from_numpy(g["gt"]).to(dev))
    assert abs(h.item() - float(g["hybrid"])) < 2e-5
    flows4 = [torch.cat([flow, -flow * 0.5], 1), torch.cat([flow * 0.9, -flow * 0.4], 1)]
    u = unFlowLoss().compute_loss(flows4, torch.cat((x, y), 1))[0]
    assert abs(u.item() - float(g["unflow"])) < 2e-4


def test_train_forward_losses_vs_reference(model_args, short_sd, golden):
    """train-mode HIP forward + HIP losses against the reference's loss values on the same inputs"""
    from emip_amd import nn_base
    from emip_amd.loss.loss_flow import unFlowLoss
    from emip_amd.loss.loss_pred import hybrid_e_loss
    from emip_amd.model.EMIP_short.model import CoUpdater
    nn_base.set_default_dtype(torch.float32)
    g = golden("short_train_b2.npz")
    net = CoUpdater(model_args)
    net.load_state_dict(short_sd)
    net = net.to("cuda:0").train()
    for m in net.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    im1, im2 = synthetic_pair(2, seed=77)
    gt = synthetic_gt(2, seed=99).cuda()
    im1, im2 = im1.cuda(), im2.cuda()
    with torch.no_grad():
        mask, fw, bw = net(im1, im2)
        lp = hybrid_e_loss(mask, gt)
        lf = unFlowLoss().compute_loss([torch.cat([fw[i], bw[i]], 1) for i in range(2)], torch.cat((im1, im2), 1))[0]
    assert abs(lp.item() - float(g["loss_pred"])) < 2e-3
    assert abs(lf.item() - float(g["loss_flow"])) < 5e-3      # flow under random weights is ill-conditioned


def _oracle():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "emip_oracle", os.path.join(os.path.dirname(__file__), "..", "oracle", "emip_oracle.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.gpu
def test_hybrid_e_loss_backward_matches_oracle_autograd():
    import torch
    from emip_amd.loss.loss_pred import hybrid_e_loss
    O = _oracle()
    g = torch.Generator().manual_seed(5)
    pred = (torch.randn(3, 1, 96, 96, generator=g) * 2).requires_grad_(True)
    mask = (torch.rand(3, 1, 96, 96, generator=g) > 0.6).float()
    lo = O.hybrid_e_loss(pred.double(), mask.double())
    (lo * 1.7).backward()
    pd = pred.detach().cuda().requires_grad_(True)
    l = hybrid_e_loss(pd, mask.cuda())
    (l * 1.7).backward()
    assert abs(l.item() - lo.item()) < 1e-5
    err = (pd.grad.cpu() - pred.grad).abs().max().item()
    scale = pred.grad.abs().max().item()
    print(f"hybrid backward err {err:.3e} scale {scale:.3e}")
    assert err <= 1e-4 * scale + 1e-9


@pytest.mark.gpu
def test_unflow_loss_backward_matches_oracle_autograd():
    import torch
    from emip_amd.loss.loss_flow import unFlowLoss
    O = _oracle()
    g = torch.Generator().manual_seed(6)
    B, H, W = 2, 64, 80
    # smooth images so that the photometric gradient is well conditioned
    base = torch.rand(B, 6, H // 8, W // 8, generator=g)
    images = torch.nn.functional.interpolate(base, (H, W), mode="bilinear", align_corners=True).contiguous()
    flow = (torch.randn(B, 4, H, W, generator=g) * 1.5).requires_grad_(True)
    lo = O.unflow_loss([flow.double()], images.double())
    lo.backward()
    fd = flow.detach().cuda().requires_grad_(True)
    out = unFlowLoss().compute_loss([fd], images.cuda())
    out[0].backward()
    assert abs(out[0].item() - lo.item()) < 1e-4 * max(1.0, abs(lo.item()))
    gref = flow.grad.float()
    err = (fd.grad.cpu() - gref).abs()
    scale = gref.abs().max().item()
    # positions where a warp coordinate sits within float rounding of an integer may pick the other bilinear cell
    bad = (err > 1e-3 * scale + 1e-9).float().mean().item()
    print(f"unflow backward max err {err.max().item():.3e} scale {scale:.3e} outlier fraction {bad:.2e}")
    assert bad < 1e-3
