"""unFlowLoss on MI355X: /root/reference/loss/loss_flow.py:60-138.

total = sum over the flow pyramid (w_scales = 1) of 0.5 * (photometric(im1, warp(im2, fw), 1-occ1) +
photometric(im2, warp(im1, bw), 1-occ2)); occlusion masks come from the backward flows of scale 0 and are
reused for the other scales; the reference computes and discards the smoothness term (:134-137).
At 352x352 the 'area' resize of the images and the 'nearest' resize of the masks are identities."""
import torch

from .. import ops
from ..autograd import UnflowPairLossFn


class unFlowLoss(torch.nn.Module):
    def compute_loss(self, output, target):
        flows = output
        im1, im2 = target[:, :3].contiguous(), target[:, 3:].contiguous()
        total = torch.zeros(1, dtype=torch.float32, device=target.device)
        m1 = m2 = None
        for i, flow in enumerate(flows):
            assert flow.shape[-2:] == im1.shape[-2:], "only the full-resolution pyramid of EMIP is built"
            fw, bw = flow[:, :2].contiguous(), flow[:, 2:].contiguous()
            if i == 0:
                m1 = ops.occ_mask_backward(bw.detach(), complement=True)       # 1 - occlusion(backward flow)
                m2 = ops.occ_mask_backward(fw.detach(), complement=True)
            if torch.is_grad_enabled() and flow.requires_grad:
                total = total + UnflowPairLossFn.apply(fw, bw, im1, im2, m1, m2)
                continue
            r1 = ops.flow_warp(im2, fw)
            r2 = ops.flow_warp(im1, bw)
            ops.photometric_loss(im1, r1, m1, total, weight=0.5, accumulate=True)
            ops.photometric_loss(im2, r2, m2, total, weight=0.5, accumulate=True)
        return total[0], total[0], 0.0, flows[0].abs().mean()
