"""Data-parallel gradient reduction for the EMIP training step: bucketed all-reduce(mean) launched as gradients become
ready, overlapped with the rest of backward (one process per GPU, torch.distributed; the "nccl" backend is RCCL over
xGMI on MI355X).

Why not DDP: the reference wraps the model in DDP(find_unused_parameters=True) (train.py:279) because 108 of its
tensors never receive gradients (dead modules, GMFlow adaptor parameters) and GMFlow is frozen after wrapping.  Here

  * the FIRST step is a calibration step: the reducer logs the order in which gradients actually become ready and which
    parameters never get one; rank 0's log is broadcast, and from the second step on the buckets hold only parameters that
    train, in gradient-ready order (so bucket k is complete -- and its exchange in flight -- while backward still works on
    the layers of bucket k+1; `conv_corr.0.weight`, 67 MB and ready early, travels alone and first).  Parameters that
    never received a gradient are not exchanged at all (no unused-parameter bitmap, no zero-filled segments); should one
    of them get a gradient later it is reduced in a trailing bucket, identically on every rank;
  * xGMI is point-to-point (7 links per GPU): few, large buckets keep every link busy -- default 64 MB;
  * algo="direct": reduce-scatter + all-gather written as all-to-all + all-gather on the full mesh (every peer link
    carries 1/world of the bucket at once, instead of a ring's one-link-at-a-time), with `comm_dtype=torch.bfloat16`
    transport and float32 ACCUMULATION of the received shards (a bf16 all-reduce would accumulate in bf16);
    algo="allreduce": one RCCL all-reduce per bucket in `comm_dtype` (default float32);
  * the exchange runs on a side stream; BatchNorm statistics stay per replica exactly as in the reference (no SyncBN).

Usage:
    reducer = GradReducer([p for p in model.parameters() if p.requires_grad])
    loss.backward()            # hooks fire per parameter; full buckets start their exchange immediately
    reducer.finish()           # waits, writes the averaged gradients back
    optimizer.step()
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "offsets", "numel", "padded", "flat", "pending", "work", "result", "launched", "t_launch")

    def __init__(self, params, world):
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        self.padded = (n + world - 1) // world * world
        self.flat = None
        self.pending = len(params)
        self.work = None
        self.result = None
        self.launched = False
        self.t_launch = None


def _make_buckets(order, bucket_bytes, world):
    buckets, cur, cur_bytes = [], [], 0
    for p in order:
        nbytes = p.numel() * 4
        if cur and cur_bytes + nbytes > bucket_bytes:
            buckets.append(_Bucket(cur, world))
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nbytes
        if nbytes >= bucket_bytes:            # a tensor as large as a bucket travels alone
            buckets.append(_Bucket(cur, world))
            cur, cur_bytes = [], 0
    if cur:
        buckets.append(_Bucket(cur, world))
    return buckets


class GradReducer:
    def __init__(self, params, bucket_bytes=64 << 20, group=None, comm_dtype=None, algo="allreduce", record_events=False):
        assert algo in ("allreduce", "direct")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.comm_dtype = comm_dtype
        self.algo = algo
        self.bucket_bytes = bucket_bytes
        self.record_events = record_events
        self.params = [p for p in params if p.requires_grad]
        self._index = {p: i for i, p in enumerate(self.params)}
        # before the calibration step: reverse registration order, every parameter
        self.buckets = _make_buckets(list(reversed(self.params)), bucket_bytes, self.world)
        self.dead = []                      # parameters that received no gradient in the calibration step
        self.calibrated = False
        self.ready_order = []               # parameter indices in the order their gradients became ready (calibration)
        self.launch_log = []                # bucket indices in launch order of the last step (tests / diagnostics)
        self._side = None
        self._rebind()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _rebind(self):
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for p in b.params:
                self._where[p] = bi

    # ------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        if not self.calibrated:
            self.ready_order.append(self._index[p])
        if self.world == 1:
            return
        bi = self._where.get(p)
        if bi is None:                      # a parameter the calibration step saw without gradient: trailing bucket in finish()
            return
        b = self.buckets[bi]
        b.pending -= 1
        if b.pending == 0:
            self._launch(bi)

    def _stream(self, dev):
        if dev.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _launch(self, bi):
        b = self.buckets[bi]
        ref = b.params[0]
        if ref.is_cuda:
            from . import ops
            ops.flush_wgrads()              # weight gradients the step has deferred must be on the compute stream first
            ops.WGRADS.fixup(b.params)      # ... and be what these parameters' .grad holds
        side = self._stream(ref.device)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(ref.device))
            ctx = torch.cuda.stream(side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            if b.flat is None or b.flat.device != ref.device:
                b.flat = torch.zeros(b.padded, dtype=torch.float32, device=ref.device)
            for p, off in zip(b.params, b.offsets):
                seg = b.flat[off:off + p.numel()]
                if p.grad is None:
                    seg.zero_()                       # parameter did not take part in this step
                else:
                    seg.copy_(p.grad.reshape(-1))
            if self.record_events and side is not None:
                b.t_launch = torch.cuda.Event(enable_timing=True)
                b.t_launch.record(side)
            cd = self.comm_dtype or torch.float32
            if self.algo == "allreduce":
                buf = b.flat if cd == torch.float32 else b.flat.to(cd)
                b.work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                b.result = buf
            else:
                # reduce-scatter as all-to-all (shard j of every rank lands on rank j), f32 sum, all-gather of the result
                w, chunk = self.world, b.padded // self.world
                send = b.flat.view(w, chunk).to(cd)
                recv = torch.empty_like(send)
                dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group)
                shard = recv.float().sum(0).to(cd)
                out = torch.empty(w * chunk, dtype=cd, device=ref.device)
                b.work = dist.all_gather_into_tensor(out, shard, group=self.group, async_op=True)
                b.result = out
        b.launched = True
        self.launch_log.append(bi)

    def finish(self):
        """Complete the step: exchange what is still pending, wait, average, write the gradients back."""
        if self.world > 1:
            for bi, b in enumerate(self.buckets):
                if not b.launched:
                    self._launch(bi)
            late = [p for p in self.dead if p.grad is not None]
            if late:                        # same graph on every rank, so every rank takes this branch together
                tb = _Bucket(late, self.world)
                self.buckets.append(tb)
                self._launch(len(self.buckets) - 1)
            inv = 1.0 / self.world
            dev = self.buckets[0].params[0].device
            side = self._stream(dev)
            for b in self.buckets:
                b.work.wait()
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
            for b in self.buckets:
                for p, off in zip(b.params, b.offsets):
                    if p.grad is not None:
                        p.grad.copy_(b.result[off:off + p.numel()].view_as(p.grad).to(p.grad.dtype) * inv)
            if late:
                self.buckets.pop()
        if not self.calibrated:
            self._calibrate()
        for b in self.buckets:
            b.pending, b.work, b.result, b.launched = len(b.params), None, None, False

    def _calibrate(self):
        """after the first step: buckets = the parameters that received a gradient, in the order they became ready (rank 0's
        order is broadcast so that every rank builds the same buckets)"""
        order = list(self.ready_order)
        if self.world > 1:
            box = [order]
            dist.broadcast_object_list(box, src=0, group=self.group)
            order = box[0]
        seen = set(order)
        self.dead = [p for i, p in enumerate(self.params) if i not in seen]
        self.buckets = _make_buckets([self.params[i] for i in order], self.bucket_bytes, self.world)
        self._rebind()
        self.calibrated = True
        self.ready_order = order

    def begin_step(self):
        """optional: clear the per-step launch log (tests read it after finish())"""
        self.launch_log = []

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
