"""One EMIP-short training step on MI355X (/root/reference/train.py:36-66,340-342,380).

    model = CoUpdater(args).cuda().train(); freeze_like_reference(model)
    opt = build_optimizer(model, lr, weight_decay, clip)       # fused clamp + AdamW (one launch)
    red = GradReducer(trainable(model))                        # bucketed all-reduce over RCCL (world > 1)
    losses = train_step(model, opt, red, image1, image2, gts)

Forward, both losses, backward, gradient clamp and the AdamW update are libemip_hip.so kernels; torch.autograd orders
the backward and accumulates parameter gradients, torch.distributed carries the gradient buckets.
"""
import torch

from . import ops
from .dp import GradReducer
from .loss.loss_flow import unFlowLoss
from .loss.loss_pred import hybrid_e_loss
from .optim import FusedClampAdamW


def freeze_like_reference(model):
    """train.py:340-342: every GMFlow parameter except the (never-called) dwconv / adaptor ones is frozen"""
    for name, para in model.named_parameters():
        if "GMFlow" in name and 'dwconv' not in name and 'adaptor' not in name:
            para.requires_grad = False
    return model


def trainable(model):
    return [p for p in model.parameters() if p.requires_grad]


def build_optimizer(model, lr=1e-5, weight_decay=1e-7, clip=0.5):
    """train.py:380 (AdamW over the parameters that require grad) fused with clip_gradient (train.py:61)"""
    return FusedClampAdamW(trainable(model), lr=lr, weight_decay=weight_decay, clip=clip)


_flow_loss = unFlowLoss()


def train_step(model, optimizer, reducer, image1, image2, gts):
    """train.py:43-62.  Returns (loss, loss_pred, loss_flow) as 0-dim device tensors (no host sync here)."""
    optimizer.zero_grad(set_to_none=True)
    ops.ARENA.begin(image1.device)          # one fill clears every gradient accumulator of this step (ops.GradArena)
    failed = True
    try:
        with torch.enable_grad():
            preds = model(image1, image2)
            loss_pred = hybrid_e_loss(preds[0], gts)
            image_pair = torch.cat((image1, image2), dim=1)
            flow_pair = [torch.cat((preds[1][i], preds[2][i]), dim=1) for i in range(len(preds[1]))]
            loss_flow = _flow_loss.compute_loss(flow_pair, image_pair)[0]
            loss = loss_pred + loss_flow
            loss.backward()
            ops.flush_wgrads()                  # the Linear weight gradients backward deferred, as one grouped launch
        failed = False
    finally:
        ops.ARENA.end(failed)               # on an exception: the deferred queue is dropped, the original error propagates
    if reducer is not None:
        reducer.finish()
    optimizer.step()
    return loss.detach(), loss_pred.detach(), loss_flow.detach()


def freeze_short_term(model_long):
    """train_long.py:404-406: everything under `short_term` is frozen"""
    for name, para in model_long.named_parameters():
        if "short_term" in name:
            para.requires_grad_(False)
    return model_long


def train_long_video(model_long, optimizer, reducer, frames, masks):
    """train_long.py:41-61: one optimizer step PER FRAME of a video clip, the memory fed back detached.
    frames [N,3,H,W], masks [N,1,H,W] on the device.  Returns the summed loss of the clip (0-dim device tensor)."""
    memory_k = memory_v = None
    loss_iter = None
    for index in range(1, len(frames)):
        optimizer.zero_grad(set_to_none=True)
        with torch.enable_grad():
            preds, memory_k, memory_v = model_long(frames[index - 1], frames[index], index, memory_k, memory_v)
            memory_k, memory_v = memory_k.detach(), memory_v.detach()
            loss = hybrid_e_loss(preds, masks[index].unsqueeze(dim=0))
            loss.backward()
        if reducer is not None:
            reducer.finish()
        optimizer.step()
        loss_iter = loss.detach() if loss_iter is None else loss_iter + loss.detach()
    return loss_iter


__all__ = ["freeze_like_reference", "freeze_short_term", "trainable", "build_optimizer", "train_step",
           "train_long_video", "GradReducer"]
