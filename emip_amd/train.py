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


def forward_backward(model, image1, image2, gts):
    """train.py:43-60 up to the optimizer: forward, both losses, backward, the deferred weight gradients.  Leaves the gradients
    in `.grad`; returns (loss, loss_pred, loss_flow) as 0-dim device tensors (no host sync here).  Every launch is stream-ordered
    device work with fixed shapes, so the whole call can be captured into a hipGraph (GraphedTrainStep)."""
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
            ops.flush_wgrads()                  # the weight gradients backward deferred: one grouped launch per kind and stream
            ops.join_forks(image1.device)       # the forked branch (model.FORK_DEEP_TRAIN) before anything reads gradients
        failed = False
    finally:
        if failed and image1.is_cuda:
            # whatever a forked branch still runs writes arena slices the next begin() clears: wait for it here as well
            try:
                ops.join_forks(image1.device)
            except Exception:                   # noqa: BLE001 -- the original error is the one to report
                pass
        ops.ARENA.end(failed)               # on an exception: the deferred queue is dropped, the original error propagates
    return loss.detach(), loss_pred.detach(), loss_flow.detach()


def train_step(model, optimizer, reducer, image1, image2, gts):
    """train.py:43-62.  Returns (loss, loss_pred, loss_flow) as 0-dim device tensors (no host sync here)."""
    optimizer.zero_grad(set_to_none=True)
    losses = forward_backward(model, image1, image2, gts)
    if reducer is not None:
        reducer.finish()
    optimizer.step()
    return losses


class GraphedTrainStep:
    """The training step with forward + losses + backward + weight gradients replayed as ONE hipGraph.

    An eager step costs the host ~62 ms of Python / autograd / launch path (2 600 C-ABI calls, 2 200 autograd nodes) against
    ~67 ms of kernels at batch 32: the GPU runs barely behind the host, and a forked branch (model.FORK_DEEP_TRAIN) finds
    nothing queued to run beside.  Captured, the host issues one graph launch + the optimizer's two launches per step, and the
    fork / join of PVT stages 3-4 (forward and backward) are branches of the graph.

        gs = GraphedTrainStep(model, optimizer, image1, image2, gts)     # shapes, dtype and device are fixed from here on
        loss, loss_pred, loss_flow = gs.step(image1, image2, gts)        # copies the batch in, replays, steps the optimizer

    What is captured is exactly forward_backward() -- the same Functions and kernels, stochastic depth included (the graph
    advances the generator's Philox offset on every replay).  Outside the graph: the fused clamp + AdamW launch (its step count
    and learning rate are launch arguments) and the one-launch refresh of the weight packs.  The gradients live in the graph's
    memory pool: `.grad` tensors keep their addresses from replay to replay, so nothing is re-uploaded.  Single-process only
    (a GradReducer's collectives are not captured); the eager train_step() stays the reference for parity and for N > 1."""

    def __init__(self, model, optimizer, image1, image2, gts, warmup=2):
        dev = image1.device
        assert dev.type == "cuda", "hipGraph capture needs the device"
        self.model, self.optimizer = model, optimizer
        self.image1, self.image2, self.gts = image1.clone(), image2.clone(), gts.clone()
        # The modules keep the intermediates of their last forward (`.last`, for parity checks); after a training forward those
        # tensors hold the autograd graph, and with it the parameters' AccumulateGrad nodes, whose stream is the one they were
        # first created under.  A node that survives from an eager step on the default stream makes the engine order the
        # default stream against the capture -- which invalidates it.  Dropped here, they are created again by the warm-up
        # passes below, on the capture stream (and the forked one), and stay with it.
        import gc
        for m in model.modules():
            if isinstance(getattr(m, "last", None), dict):
                m.last = {}
        gc.collect()
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(max(1, warmup)):           # eager passes on the capture stream: weight packs, arena size, LDS attributes;
                optimizer.zero_grad(set_to_none=True)     # parameters and optimizer state are not touched
                forward_backward(model, self.image1, self.image2, self.gts)
            optimizer.zero_grad(set_to_none=True)     # the captured backward allocates .grad from the graph's pool
        torch.cuda.synchronize(dev)
        # pinned staging buffers for the record tables of the captured weight-gradient flushes (no host allocation, event or
        # synchronisation may happen inside a capture); the graph's copy nodes read them on every replay
        self._hosts = ops.WGRADS.capture_buffers(16)
        ops.WGRADS.capture_pool = list(self._hosts)
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.losses = forward_backward(model, self.image1, self.image2, self.gts)
        finally:
            ops.WGRADS.capture_pool = None
        self.grads = [(p, p.grad) for p in model.parameters() if p.grad is not None]

    def replay(self):
        """forward + backward of the batch in the static input buffers; gradients in `.grad`, losses in self.losses"""
        for p, g in self.grads:                       # someone may have dropped or replaced them (zero_grad)
            if p.grad is not g:
                p.grad = g
        self.graph.replay()
        return self.losses

    def step(self, image1=None, image2=None, gts=None):
        if image1 is not None:
            for got, buf in ((image1, self.image1), (image2, self.image2), (gts, self.gts)):
                if got.shape != buf.shape or got.dtype != buf.dtype:
                    raise ValueError("GraphedTrainStep was captured for %s %s, got %s %s: build a new one for another batch shape"
                                     % (tuple(buf.shape), buf.dtype, tuple(got.shape), got.dtype))
            self.image1.copy_(image1, non_blocking=True)
            self.image2.copy_(image2, non_blocking=True)
            self.gts.copy_(gts, non_blocking=True)
        self.replay()
        self.optimizer.step()
        return self.losses


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


__all__ = ["freeze_like_reference", "freeze_short_term", "trainable", "build_optimizer", "train_step", "forward_backward",
           "GraphedTrainStep", "train_long_video", "GradReducer"]
