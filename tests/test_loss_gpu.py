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
