"""`utils.utils` of the reference's drivers (train.py:26,61): the element-wise gradient clamp.

`FusedClampAdamW` (emip_amd.optim) fuses this clamp into the optimizer launch; this stand-alone form exists so that a
driver that keeps `torch.optim.AdamW` + `clip_gradient(optimizer, clip)` (/root/reference/train.py:61-62, utils/utils.py:1-11)
runs unchanged.  One foreach launch over all gradients instead of one clamp kernel per parameter."""
import torch


def clip_gradient(optimizer, grad_clip):
    """clamp every gradient of the optimizer's parameters to [-grad_clip, grad_clip], element-wise and in place"""
    grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    if grads:
        with torch.no_grad():
            torch._foreach_clamp_min_(grads, -grad_clip)
            torch._foreach_clamp_max_(grads, grad_clip)
