"""Data-parallel gradient reduction for the EMIP training step: bucketed all-reduce(mean) launched as
gradients become ready, overlapped with the rest of backward (one process per GPU, torch.distributed; the
"nccl" backend is RCCL over xGMI on MI355X).

Why not DDP: the reference wraps the model in DDP(find_unused_parameters=True) (train.py:279) because 108 of its
tensors never receive gradients (dead modules, GMFlow adaptor parameters) and GMFlow is frozen after wrapping.
Here the reducer is simply built over the parameters that DO train, so there is no unused-parameter bitmap
all-reduce and no per-iteration graph traversal; BatchNorm statistics stay per replica exactly as in the reference
(no SyncBN).  xGMI is point-to-point (7 links per GPU): few, large buckets keep every link busy, so the default
bucket is 64 MB and `conv_corr.0.weight` (67 MB f32, ready EARLY in backward) travels alone and first.

Usage:
    reducer = GradReducer([p for p in model.parameters() if p.requires_grad])
    loss.backward()            # hooks fire per parameter; full buckets start their all-reduce immediately
    reducer.finish()           # waits, divides by world size, writes the averaged gradients back
    optimizer.step()
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "offsets", "numel", "flat", "pending", "work", "launched")

    def __init__(self, params):
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        self.flat = None
        self.pending = len(params)
        self.work = None
        self.launched = False


class GradReducer:
    def __init__(self, params, bucket_bytes=64 << 20, group=None, comm_dtype=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.comm_dtype = comm_dtype
        params = [p for p in params if p.requires_grad]
        # gradients become ready roughly in reverse registration order
        order = list(reversed(params))
        self.buckets, cur, cur_bytes = [], [], 0
        for p in order:
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            if nbytes >= bucket_bytes:            # a tensor as large as a bucket travels alone
                self.buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(_Bucket(cur))
        self._where = {}
        self._hooks = []
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        if self.world == 1:
            return
        b, _ = self._where[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        ref = b.params[0]
        dtype = self.comm_dtype or torch.float32
        if b.flat is None or b.flat.device != ref.device or b.flat.dtype != dtype:
            b.flat = torch.empty(b.numel, dtype=dtype, device=ref.device)
        for p, off in zip(b.params, b.offsets):
            seg = b.flat[off:off + p.numel()]
            if p.grad is None:
                seg.zero_()                       # parameter did not take part in this step
            else:
                seg.copy_(p.grad.reshape(-1))
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        b.launched = True

    def finish(self):
        """Complete the step: reduce buckets whose gradients never all arrived, wait, average, write back."""
        if self.world > 1:
            for b in self.buckets:
                if not b.launched:
                    self._launch(b)
            inv = 1.0 / self.world
            for b in self.buckets:
                b.work.wait()
                for p, off in zip(b.params, b.offsets):
                    if p.grad is not None:
                        p.grad.copy_(b.flat[off:off + p.numel()].view_as(p.grad).to(p.grad.dtype) * inv)
        for b in self.buckets:
            b.pending, b.work, b.launched = len(b.params), None, False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module, src=0, group=None):
    """One-time parameter/buffer broadcast at start-up (what DDP does in its constructor)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)
