"""Data-parallel gradient reduction for the EMIP training step: bucketed all-reduce(mean) launched as gradients become
ready, overlapped with the rest of backward (one process per GPU, torch.distributed; the "nccl" backend is RCCL over
xGMI on MI355X).

Why not DDP: the reference wraps the model in DDP(find_unused_parameters=True) (train.py:279) because 108 of its
tensors never receive gradients (dead modules, GMFlow adaptor parameters) and GMFlow is frozen after wrapping.  Here

  * the FIRST step is a calibration step: the reducer logs the order in which gradients actually become ready and which
    parameters never get one; rank 0's log is broadcast, and from the second step on the buckets hold only parameters that
    train, in gradient-ready order (so bucket k is complete -- and its exchange in flight -- while backward still works on
    the layers of bucket k+1; `conv_corr.0.weight`, 67 MB and ready early, travels alone and first).  Parameters that
    never received a gradient are not exchanged; whether any of them got one later is agreed on by a one-word MAX
    all-reduce per step ON THE HOST (a gloo control group beside the RCCL one: the local answer is host knowledge --
    `p.grad is not None` -- so nothing waits for the device), and then ALL of them travel in a trailing bucket (zeros
    where a rank has none).  Regular buckets leave strictly in index order -- bucket k only once bucket k-1 has left,
    the rest from finish() -- and a parameter of a regular bucket without a gradient on some rank takes part with
    zeros there, so every rank issues the same collectives in the same order and receives the same mean whatever its
    local graph looked like;
  * the buckets are slices of ONE flat transport buffer.  A bucket costs one launch on the way out (emip_grad_pack: a
    table-driven gather of its ~150 gradient tensors, f32 or bf16 on the wire) and the whole step ONE launch on the way
    back (emip_grad_unpack: every `.grad` <- flat slice / world).  The tables are static after calibration; the gradient
    addresses repeat from step to step (ops.GradArena), so the pointer array is re-uploaded only when one moved;
  * xGMI is point-to-point (7 links per GPU): few, large buckets keep every link busy -- default 64 MB;
  * algo="direct": reduce-scatter + all-gather written as all-to-all + all-gather on the full mesh (every peer link
    carries 1/world of the bucket at once, instead of a ring's one-link-at-a-time), with `comm_dtype=torch.bfloat16`
    transport and float32 ACCUMULATION of the received shards (emip_shard_sum; a bf16 all-reduce would accumulate in
    bf16); algo="allreduce": one RCCL all-reduce per bucket in `comm_dtype` (default float32);
  * the exchange runs on a side stream; BatchNorm statistics stay per replica exactly as in the reference (no SyncBN).

CPU tensors (the gloo rehearsals of tests/test_dist_cpu.py) take the same code with torch copies in place of the two
table kernels: that branch is bookkeeping under test, not a compute fallback -- the model has no CPU path to feed it.

Usage:
    reducer = GradReducer([p for p in model.parameters() if p.requires_grad])
    loss.backward()            # hooks fire per parameter; full buckets start their exchange immediately
    reducer.finish()           # waits, writes the averaged gradients back
    optimizer.step()
"""
import struct

import torch
import torch.distributed as dist

ALIGN = 64          # bucket starts, in elements (256 B of f32)


class _Bucket:
    __slots__ = ("params", "offsets", "numel", "padded", "lo", "rec0", "blk0", "nblk", "pending", "work", "launched",
                 "t_launch", "ptrs")

    def __init__(self, params, world):
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 3) & ~3         # every tensor starts on a 16-byte boundary of the flat buffer (vector accesses)
        self.numel = n
        unit = world * ALIGN
        self.padded = (n + unit - 1) // unit * unit
        self.lo = self.rec0 = self.blk0 = self.nblk = 0
        self.pending = len(params)
        self.work = None
        self.launched = False
        self.t_launch = None
        self.ptrs = None                  # gradient addresses this bucket was last packed from


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
    def __init__(self, params, bucket_bytes=64 << 20, group=None, comm_dtype=None, algo="allreduce", record_events=False,
                 single_rank_collectives=False):
        """single_rank_collectives: with a process group of ONE rank, still run every pack / collective / unpack (a mean
        over one rank: the gradients come back unchanged) -- the hardware rehearsal of the RCCL calls on a one-GPU box
        (tests/test_dp_gpu.py); without it a world of one skips the exchange altogether."""
        assert algo in ("allreduce", "direct")
        assert comm_dtype in (None, torch.float32, torch.bfloat16)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.exchange = self.world > 1 or (single_rank_collectives and dist.is_available() and dist.is_initialized())
        # control group for host-side agreement (the late-bucket decision): gloo, so that a one-word MAX over the ranks
        # costs an inter-process round trip and no device synchronisation.  A gloo default group serves as it is.
        self._ctl = group
        if self.exchange and dist.get_backend(group) != "gloo":
            self._ctl = dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None,
                                       backend="gloo")
        self.comm_dtype = comm_dtype or torch.float32
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
        self._seen = set()
        self.launch_log = []                # bucket indices in launch order of the last step (tests / diagnostics)
        self.kernel_launches = 0            # libemip_hip.so launches this reducer issued in the last step (tests)
        self.host_ms = 0.0                  # record_events: host time of the step spent in pack / unpack (not in collectives)
        self._kev = []                      # record_events: (start, end) event pairs around the pack / unpack launches
        self._side = None
        self._late = None                   # the trailing bucket of the calibration step's dead parameters (built on demand)
        self._next = 0                      # index of the next regular bucket to leave (buckets leave in index order)
        self._rebind()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    # ---- layout: every bucket a slice of one flat buffer, static device tables ---------------------------------------------
    def _rebind(self):
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for p in b.params:
                self._where[p] = bi
        self._flat = self._out = self._tables = None

    def _all_buckets(self):
        return self.buckets + ([self._late] if self._late is not None else [])

    def _build(self, dev):
        """flat buffer(s) + (cuda) the record / block tables of every bucket, the late bucket included"""
        bl = self._all_buckets()
        lo = rec = blk = 0
        recs, bmap = bytearray(), []
        chunk = 2048
        if dev.type == "cuda":
            from . import _lib
            chunk = _lib.load().emip_adamw_chunk()
        for b in bl:
            b.lo, b.rec0, b.blk0 = lo, rec, blk
            for i, (p, off) in enumerate(zip(b.params, b.offsets)):
                recs += struct.pack("<qq", lo + off, p.numel())
                bmap += [(rec + i, c) for c in range((p.numel() + chunk - 1) // chunk)]
            rec += len(b.params)
            b.nblk = len(bmap) - blk
            blk = len(bmap)
            lo += b.padded
            b.ptrs = None
        self._flat = torch.zeros(lo, dtype=self.comm_dtype, device=dev)          # pads stay zero for the life of the layout
        self._out = torch.empty(lo, dtype=self.comm_dtype, device=dev) if self.algo == "direct" else self._flat
        if dev.type == "cuda":
            self._tables = dict(
                recs=torch.frombuffer(recs, dtype=torch.uint8).clone().to(dev),
                bmap=torch.tensor(bmap, dtype=torch.int32).to(dev), nrec=rec,
                gdev=torch.zeros(rec, dtype=torch.int64, device=dev),
                # pinned staging of the gradient addresses, two copies used in turn per bucket: the upload that last read a
                # copy is two uploads old when it is rewritten, so waiting for its event never stalls the host
                ghost=[torch.zeros(rec, dtype=torch.int64).pin_memory() for _ in range(2)], gev={}, turn={})

    def _ensure(self, dev):
        if self._flat is None or self._flat.device != dev:
            self._build(dev)

    # ------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        if not self.calibrated:
            i = self._index[p]
            if i not in self._seen:         # a second backward before finish() (gradient accumulation) logs nothing new
                self._seen.add(i)
                self.ready_order.append(i)
        if not self.exchange:
            return
        bi = self._where.get(p)
        if bi is None:                      # a parameter the calibration step saw without gradient: trailing bucket in finish()
            return
        b = self.buckets[bi]
        if b.launched:                      # gradient accumulation: a later backward re-arms nothing; finish() sends what is there
            return
        b.pending -= 1
        # strictly in index order (as DDP does): a rank whose local graph completes bucket k before bucket k-1 -- or
        # lacks one of k-1's gradients altogether, which finish() then supplies as zeros -- must not enqueue k's
        # collective ahead of k-1's on the shared communicator
        while self._next < len(self.buckets) and self.buckets[self._next].pending <= 0:
            self._launch(self.buckets[self._next], self._next)
            self._next += 1

    def _stream(self, dev):
        if dev.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _pack(self, b, dev):
        """gradients of bucket b -> its slice of the flat buffer (runs on the side stream)"""
        if self.calibrated:
            for p in b.params:
                if p.grad is None:
                    # no gradient on THIS rank (a branch its local graph did not take): it sends zeros and receives the mean
                    # like every other rank (what DDP writes back); left at None the other ranks would step the parameter
                    # alone.  (The calibration step leaves None alone: those parameters are about to be declared dead.)
                    p.grad = torch.zeros(p.shape, dtype=torch.float32 if dev.type == "cuda" else p.dtype, device=dev)
        if dev.type != "cuda":
            for p, off in zip(b.params, b.offsets):
                seg = self._flat[b.lo + off:b.lo + off + p.numel()]
                if p.grad is None:
                    seg.zero_()
                else:
                    seg.copy_(p.grad.reshape(-1))
            return
        from . import _lib
        t = self._tables
        ptrs = [0 if p.grad is None else p.grad.data_ptr() for p in b.params]
        if ptrs != b.ptrs:                  # steady state: the gradient arena hands out the same addresses every step
            assert all(p.grad is None or (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) for p in b.params)
            k = t["turn"][b.rec0] = t["turn"].get(b.rec0, 0) ^ 1
            ev = t["gev"].get((b.rec0, k))
            if ev is not None:
                ev.synchronize()            # the upload that read this copy two uploads ago (long done)
            host = t["ghost"][k][b.rec0:b.rec0 + len(ptrs)]
            host.copy_(torch.tensor(ptrs, dtype=torch.int64))
            t["gdev"][b.rec0:b.rec0 + len(ptrs)].copy_(host, non_blocking=True)
            ev = t["gev"][(b.rec0, k)] = torch.cuda.Event()
            ev.record()
            b.ptrs = ptrs
        if b.nblk:
            self._kcall("emip_grad_pack", t["recs"].data_ptr(), t["bmap"].data_ptr() + 8 * b.blk0, t["gdev"].data_ptr(), b.nblk,
                        self._flat.data_ptr(), int(self.comm_dtype == torch.bfloat16),
                        torch.cuda.current_stream(dev).cuda_stream)

    def _launch(self, b, bi):
        ref = b.params[0]
        dev = ref.device
        self._ensure(dev)
        from . import ops
        if ref.is_cuda:
            ops.flush_wgrads()              # weight gradients the step has deferred must be on their compute streams first
            # ... each queue was launched on its own stream (ops.WgradQueue): fixup() adds convolution gradients on THIS one
            cur = torch.cuda.current_stream(dev)
            for st in ops.step_streams(dev):
                if st.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(st)
            ops.WGRADS.fixup(b.params)      # ... and be what these parameters' .grad holds
        side = self._stream(dev)
        if side is not None:
            # gradients of this bucket may come from the step's main stream or from a forked branch (model.FORK_DEEP_TRAIN);
            # the hook that fills the bucket runs on only one of them
            for st in ops.step_streams(dev):
                side.wait_stream(st)
            ctx = torch.cuda.stream(side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            self._timed(self._pack, b, dev)
            if self.record_events and side is not None:
                b.t_launch = torch.cuda.Event(enable_timing=True)
                b.t_launch.record(side)
            seg = self._flat[b.lo:b.lo + b.padded]
            if self.algo == "allreduce":
                b.work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                # reduce-scatter as all-to-all (shard j of every rank lands on rank j), f32 sum, all-gather of the result
                w, chunk = self.world, b.padded // self.world
                recv = torch.empty(b.padded, dtype=self.comm_dtype, device=dev)
                dist.all_to_all_single(recv, seg, group=self.group)
                shard = torch.empty(chunk, dtype=self.comm_dtype, device=dev)
                if dev.type == "cuda":
                    self._kcall("emip_shard_sum", recv.data_ptr(), shard.data_ptr(), w, chunk,
                                int(self.comm_dtype == torch.bfloat16), torch.cuda.current_stream(dev).cuda_stream)
                else:
                    shard.copy_(recv.view(w, chunk).float().sum(0))
                b.work = dist.all_gather_into_tensor(self._out[b.lo:b.lo + b.padded], shard, group=self.group, async_op=True)
        b.launched = True
        self.launch_log.append(bi)

    def _timed(self, fn, *a):
        """fn(*a); with record_events also its host time (tests: the reducer's own overhead, apart from the collectives)"""
        if not self.record_events:
            return fn(*a)
        import time
        t0 = time.perf_counter()
        fn(*a)
        self.host_ms += (time.perf_counter() - t0) * 1e3

    def _kcall(self, name, *args):
        """one libemip_hip.so launch; with record_events a HIP-event pair tightly around it"""
        from . import _lib
        self.kernel_launches += 1
        if not self.record_events:
            return _lib.call(name, *args)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        _lib.call(name, *args)
        ev[1].record()
        self._kev.append(ev)

    def kernel_ms(self):
        """record_events: device time of this step's pack / unpack launches (call after a synchronize)"""
        return sum(a.elapsed_time(b) for a, b in self._kev)

    def _unpack(self, dev, late):
        """every .grad <- reduced flat slice / world: ONE launch for the whole step"""
        inv = 1.0 / self.world
        bl = self.buckets + ([self._late] if late else [])
        if dev.type != "cuda":
            for b in bl:
                for p, off in zip(b.params, b.offsets):
                    if p.grad is not None:
                        p.grad.copy_(self._out[b.lo + off:b.lo + off + p.numel()].view_as(p.grad).to(p.grad.dtype) * inv)
            return
        from . import _lib
        t = self._tables
        nblk = sum(b.nblk for b in bl)
        if nblk:
            self._kcall("emip_grad_unpack", t["recs"].data_ptr(), t["bmap"].data_ptr(), t["gdev"].data_ptr(), nblk,
                        self._out.data_ptr(), int(self.comm_dtype == torch.bfloat16), float(inv),
                        torch.cuda.current_stream(dev).cuda_stream)

    def finish(self):
        """Complete the step: exchange what is still pending, wait, average, write the gradients back."""
        if self.exchange and self.buckets:
            dev = self.buckets[0].params[0].device
            self._ensure(dev)
            for bi in range(self._next, len(self.buckets)):      # what backward did not complete, still in index order
                self._launch(self.buckets[bi], bi)
            self._next = len(self.buckets)
            late = False
            if self._late is not None:
                # did ANY rank give a calibration-dead parameter a gradient this step?  One word, MAX over the ranks: the
                # decision is the same everywhere, whatever the local graphs looked like.  The local answer is host
                # knowledge and the word travels over the gloo control group: the device is not waited for (round 3 read a
                # device tensor back here, a host <-> device synchronisation at the end of every backward)
                flag = torch.tensor([float(any(p.grad is not None for p in self.dead))])
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._ctl)
                late = bool(flag[0] > 0)
            if late:
                for p in self._late.params:              # ranks without a gradient for it take part with zeros
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._launch(self._late, len(self.buckets))
            for b in self.buckets + ([self._late] if late else []):
                b.work.wait()
            side = self._stream(dev)
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
            self._timed(self._unpack, dev, late)
        if not self.calibrated:
            self._calibrate()
        self._reset()

    def _reset(self):
        for b in self._all_buckets():
            b.pending, b.work, b.launched = len(b.params), None, False
        self._next = 0

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
        # the calibration step's dead parameters keep a (5 MB) trailing slice of the flat buffer: exchanged only in a step in
        # which some rank produced a gradient for one of them
        self._late = _Bucket(list(self.dead), self.world) if self.dead else None
        self._rebind()
        self.calibrated = True
        self.ready_order = order

    def begin_step(self):
        """optional: clear the per-step launch log and launch counter (tests read them after finish())"""
        self.launch_log = []
        self.kernel_launches = 0
        self.host_ms, self._kev = 0.0, []

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
