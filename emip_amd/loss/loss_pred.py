"""hybrid_e_loss on MI355X: /root/reference/loss/loss_pred.py:4-22 (forward and, under autograd, backward kernels)."""
import torch

from .. import ops
from ..autograd import HybridELossFn


def hybrid_e_loss(pred, mask):
    if torch.is_grad_enabled() and pred.requires_grad:
        return HybridELossFn.apply(pred, mask)[0]
    return ops.hybrid_e_loss(pred.contiguous(), mask.contiguous())[0]
