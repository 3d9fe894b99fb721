"""hipGraph replay of the EMIP-short forward, with the batch split over concurrent HIP streams.

The forward is ~1.6k kernel launches with fixed shapes; replaying it as a hipGraph removes the host launch path
from the step (MI355X: ~3-4 us of host time per eager launch).  Most of those kernels are short (10-60 us) and
bound by their own load -> compute -> store phases rather than by chip throughput, so the batch is additionally
cut into `splits` sub-batches, each captured into its own graph and replayed on its own stream: kernels of
different sub-batches overlap and fill the phase bubbles (measured on MI355X at 16 pairs: 1 stream 734 pairs/s,
2 streams 811, 4 streams 866, 8 streams 634).  Capture goes through torch.cuda.CUDAGraph only for the stream-capture
bookkeeping and its private memory pool; every node is a libemip_hip.so kernel (or a memset it issues)."""
import torch


class _Part:
    def __init__(self, net, batch, size, device, warmup, cnn_first=False, fork_deep=None):
        """fork_deep: capture with PVT stages 3-4 on a forked branch of the graph (model.FORK_DEEP); None = the module's setting"""
        from .model.EMIP_short import model as _m
        prev, _m.CNN_FIRST = _m.CNN_FIRST, bool(cnn_first and _m.STAGGER)
        prev_f = _m.FORK_DEEP
        if fork_deep is not None:
            _m.FORK_DEEP = bool(fork_deep)
        try:
            self._build(net, batch, size, device, warmup)
        finally:
            _m.CNN_FIRST = prev
            _m.FORK_DEEP = prev_f

    def _build(self, net, batch, size, device, warmup):
        self.batch = batch
        self.im1 = torch.zeros(batch, 3, size, size, device=device)
        self.im2 = torch.zeros(batch, 3, size, size, device=device)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # packs weights, builds tables, sets LDS attributes
                net.run(self.im1, self.im2)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.mask, self.preds = net.run(self.im1, self.im2)
        self.last = dict(net.last)        # captured intermediates (debugging / parity checks)


class GraphedShort:
    """Static-shape inference replay of CoUpdater: call(image1, image2) -> (mask, flow_fw, flow_bw)."""

    def __init__(self, net, batch, size=352, device="cuda:0", warmup=2, splits=4):
        while splits > 1 and batch % splits:
            splits -= 1
        self.net, self.batch, self.splits = net, batch, splits
        # every second sub-batch runs the GMFlow CNN before the PVT backbone (the two are independent): the streams then do
        # not walk the same phases of the forward in lockstep
        self.parts = [_Part(net, batch // splits, size, device, warmup, cnn_first=i % 2 == 1) for i in range(splits)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(splits)]
        for p in self.parts:          # prime: the first launch of a graph does one-time runtime work; do it serially
            p.graph.replay()
            torch.cuda.synchronize()

    def replay(self):
        """Launch every sub-batch graph on its own stream; the caller's stream waits for all of them."""
        cur = torch.cuda.current_stream()
        for p, s in zip(self.parts, self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                p.graph.replay()
        for s in self.streams:
            cur.wait_stream(s)

    def replay_free(self):
        """Throughput mode: enqueue one replay per stream WITHOUT joining; successive calls let the streams run
        ahead of each other (the caller synchronises the device when it needs the results)."""
        for p, s in zip(self.parts, self.streams):
            with torch.cuda.stream(s):
                p.graph.replay()

    def load(self, image1, image2):
        n = self.batch // self.splits
        for i, p in enumerate(self.parts):
            p.im1.copy_(image1[i * n:(i + 1) * n])
            p.im2.copy_(image2[i * n:(i + 1) * n])

    def outputs(self):
        mask = torch.cat([p.mask for p in self.parts], 0)
        npred = len(self.parts[0].preds)
        fw = [torch.cat([p.preds[k][:p.batch] for p in self.parts], 0) for k in range(npred)]
        bw = [torch.cat([p.preds[k][p.batch:] for p in self.parts], 0) for k in range(npred)]
        return mask, fw, bw

    def __call__(self, image1, image2):
        self.load(image1, image2)
        self.replay()
        return self.outputs()


class PipelinedShort:
    """Throughput replay for a stream of independent 16-pair requests: every step is ONE whole-batch forward (a single graph,
    no sub-batch split), and up to `inflight` consecutive steps run concurrently, each on its own HIP stream with its own
    graph and static buffers -- what a serving loop with several requests in flight does.  Kernels of different steps fill each
    other's launch ramps and tails (measured on MI355X, 16 pairs, bf16: one step at a time 1137 pairs/s, two 8-pair halves of
    one step 1260, two steps in flight 1441, three 1521).  Odd-numbered graphs are captured with the GMFlow CNN ahead of the
    PVT backbone so that neighbouring steps do not walk the same phases in lockstep."""

    def __init__(self, net, batch, inflight=3, size=352, device="cuda:0", warmup=2):
        self.net, self.batch, self.inflight = net, batch, inflight
        self.parts = [_Part(net, batch, size, device, warmup, cnn_first=i % 2 == 1) for i in range(inflight)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(inflight)]
        self.done = [None] * inflight    # per slot: event behind its last replay
        self.turn = 0
        for p in self.parts:          # prime: the first launch of a graph does one-time runtime work; do it serially
            p.graph.replay()
            torch.cuda.synchronize()

    @property
    def splits(self):
        return 1

    def load(self, image1, image2, slot=None):
        """inputs of the next step (slot None: the same batch into every in-flight slot, as the benchmark does).  The copies run
        on the SLOT's stream, behind whatever produced the images on the caller's stream: they cannot overtake the slot's
        previous replay (which still reads im1 / im2) and the next replay cannot start before they have landed."""
        cur = torch.cuda.current_stream()
        for i, p in enumerate(self.parts):
            if slot is None or slot == i:
                s = self.streams[i]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    p.im1.copy_(image1, non_blocking=True)
                    p.im2.copy_(image2, non_blocking=True)
                image1.record_stream(s)
                image2.record_stream(s)

    def replay_free(self):
        """enqueue ONE step on the next slot's stream without joining; returns the slot"""
        i = self.turn
        self.turn = (i + 1) % self.inflight
        s = self.streams[i]
        with torch.cuda.stream(s):
            self.parts[i].graph.replay()
            self.done[i] = torch.cuda.Event()
            self.done[i].record(s)
        return i

    def replay_alone(self):
        """ONE step on an otherwise idle device, from the LATENCY graph: the same forward captured with PVT stages 3-4 on a
        forked branch beside the GMFlow half (model.FORK_DEEP: 10.6 -> 9.6 ms alone; with four steps in flight the branches
        only compete, -1.2 %, so the in-flight graphs stay linear).  What a serving loop replays while its queue is empty.
        Built on first use from slot 0's inputs; returns (mask, flow_fw, flow_bw) after a device synchronisation."""
        if getattr(self, "alone", None) is None:
            torch.cuda.synchronize()
            p0 = self.parts[0]
            self.alone = _Part(self.net, self.batch, p0.im1.shape[-1], p0.im1.device, 1, fork_deep=True)
            self.alone.im1.copy_(p0.im1)
            self.alone.im2.copy_(p0.im2)
            self.alone.graph.replay()
            torch.cuda.synchronize()
        p = self.alone
        p.graph.replay()
        torch.cuda.synchronize()
        return p.mask, [q[:p.batch] for q in p.preds], [q[p.batch:] for q in p.preds]

    def outputs(self, slot=0):
        """the static output buffers of a slot; the caller's stream waits for the slot's last replay first (the buffers are
        rewritten by the slot's NEXT replay: consume them, or copy them on this stream, before enqueueing it)"""
        p = self.parts[slot]
        if self.done[slot] is not None:
            torch.cuda.current_stream().wait_event(self.done[slot])
        return p.mask, [q[:p.batch] for q in p.preds], [q[p.batch:] for q in p.preds]


class _LongPart:
    def __init__(self, net, streams, size, device, warmup, cnn_first=False):
        from .model.EMIP_short import model as _m
        prev, _m.CNN_FIRST = _m.CNN_FIRST, bool(cnn_first and _m.STAGGER)
        try:
            self._build(net, streams, size, device, warmup)
        finally:
            _m.CNN_FIRST = prev

    def _build(self, net, streams, size, device, warmup):
        T, n, C = net.WINDOW, (size // 8) ** 2, 128
        self.f0 = torch.zeros(streams, 3, size, size, device=device)
        self.f1 = torch.zeros(streams, 3, size, size, device=device)
        self.mem_k = torch.zeros(streams, T, n, C, dtype=net.cdtype, device=device)
        self.mem_v = torch.zeros(streams, T, n, C, dtype=net.cdtype, device=device)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                net.step_cl(self.f0, self.f1, self.mem_k, self.mem_v)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.mask, k, v = net.step_cl(self.f0, self.f1, self.mem_k, self.mem_v)
            self.mem_k.copy_(k)            # the window slides inside the graph: the next replay reads the new memory
            self.mem_v.copy_(v)


class GraphedLong:
    """Steady-state EMIP-long step (memory window full, model_long.py:105-107) for `streams` independent video streams,
    replayed as hipGraphs with the streams split over concurrent HIP streams like GraphedShort.  The memory lives in
    static channels-last buffers that each replay updates in place; `seed_memory` fills them from the reference-layout
    tensors `Model_long.forward` returns."""

    def __init__(self, net, streams, size=352, device="cuda:0", warmup=2, splits=4):
        while splits > 1 and streams % splits:
            splits -= 1
        self.net, self.nstreams, self.splits = net, streams, splits
        self.parts = [_LongPart(net, streams // splits, size, device, warmup, cnn_first=i % 2 == 1) for i in range(splits)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(splits)]
        for p in self.parts:
            p.graph.replay()
            torch.cuda.synchronize()

    def seed_memory(self, memory_k, memory_v):
        """memory_k / memory_v: [S,1,128,5,44,44] (the reference layout) -> the static channels-last buffers"""
        mk, mv = self.net._mem_from_ref(memory_k), self.net._mem_from_ref(memory_v)
        n = self.nstreams // self.splits
        for i, p in enumerate(self.parts):
            p.mem_k.copy_(mk[i * n:(i + 1) * n])
            p.mem_v.copy_(mv[i * n:(i + 1) * n])

    def load(self, frames0, frames1):
        n = self.nstreams // self.splits
        for i, p in enumerate(self.parts):
            p.f0.copy_(frames0[i * n:(i + 1) * n])
            p.f1.copy_(frames1[i * n:(i + 1) * n])

    def replay(self):
        cur = torch.cuda.current_stream()
        for p, s in zip(self.parts, self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                p.graph.replay()
        for s in self.streams:
            cur.wait_stream(s)

    def replay_free(self):
        for p, s in zip(self.parts, self.streams):
            with torch.cuda.stream(s):
                p.graph.replay()

    def masks(self):
        return torch.cat([p.mask for p in self.parts], 0)

    def memory(self):
        """current memory in the reference layout [S,1,128,5,44,44] (f32)"""
        k = torch.cat([p.mem_k for p in self.parts], 0)
        v = torch.cat([p.mem_v for p in self.parts], 0)
        h = w = int(round(k.shape[2] ** 0.5))
        return self.net._mem_to_ref(k, h, w), self.net._mem_to_ref(v, h, w)

    def __call__(self, frames0, frames1):
        self.load(frames0, frames1)
        self.replay()
        return self.masks()


class _LongSlot:
    """one in-flight GROUP of `group` consecutive time steps of PipelinedLong: graph A (the memory-independent part of all of them
    as one batch of group x streams pairs) and one graph B per time step (memory read + long decoder) over static buffers;
    `keys[g]` / `values[g]` are B's window of step g, gathered from the ring right before that B is replayed"""

    def __init__(self, net, streams, size, device, warmup, group=1):
        T, n, C = net.WINDOW, (size // 8) ** 2, 128
        S, G = streams, group
        self.f0 = torch.zeros(G * S, 3, size, size, device=device)
        self.f1 = torch.zeros(G * S, 3, size, size, device=device)
        self.keys = [torch.zeros(S, T, n, C, dtype=net.cdtype, device=device) for _ in range(G)]
        self.values = [torch.zeros(S, T, n, C, dtype=net.cdtype, device=device) for _ in range(G)]
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                a = net.step_a(self.f0, self.f1)
                net.step_b(a[0][:S], a[1][:S], a[2][:S], self.keys[0], self.values[0])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph_a):
            self.f0s, self.f2_2, self.f2_3, self.pk, self.pv = net.step_a(self.f0, self.f1)
        self.graph_b, self.mask = [], []
        for g in range(G):
            gb = torch.cuda.CUDAGraph()
            sl = slice(g * S, (g + 1) * S)
            with torch.no_grad(), torch.cuda.graph(gb, pool=self.graph_a.pool()):
                m = net.step_b(self.f0s[sl], self.f2_2[sl], self.f2_3[sl], self.keys[g], self.values[g])
            self.graph_b.append(gb)
            self.mask.append(m)


class PipelinedLong:
    """EMIP-long at steady state with consecutive TIME STEPS in flight.  The memory of model_long.py:97-107 is a sliding window
    of the (key, value) pairs the last five frames contributed, and a frame's pair depends only on that frame's short-term
    features -- nothing the memory read produces is fed back.  So a step splits into A (short-term encoders, side reductions,
    LTM.memorize: ~90 % of the work, independent of every other step) and B (LTM.segment over the window + long decoder), and the
    A parts of successive steps overlap like the independent requests of PipelinedShort; B(t) waits for A(t-4) ... A(t).
    `group` consecutive time steps share ONE A graph (a batch of group x streams pairs: the kernels run at the shapes of the
    EMIP-short benchmark), `inflight` groups are in flight on as many HIP streams; the pairs live in a ring of
    inflight * group + 5 entries, B's window is gathered from it (2 launches), the entries of a group are written behind its A."""

    def __init__(self, net, streams, inflight=3, size=352, device="cuda:0", warmup=2, group=1):
        self.net, self.nstreams, self.inflight, self.group = net, streams, inflight, group
        self.T = net.WINDOW
        self.R = inflight * group + self.T
        n, C = (size // 8) ** 2, 128
        self.ring_k = torch.zeros(streams, self.R, n, C, dtype=net.cdtype, device=device)
        self.ring_v = torch.zeros_like(self.ring_k)
        self.slots = [_LongSlot(net, streams, size, device, warmup, group) for _ in range(inflight)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(inflight)]
        self.splits = 1
        # window of step t = ring entries (t - 4 .. t) mod R, one index vector per residue of t
        self.idx = [torch.tensor([(r - self.T + 1 + j) % self.R for j in range(self.T)], device=device) for r in range(self.R)]
        self.t = 0
        self.ev_a = {}                 # step -> event "ring entry of the step is written"
        self.ev_b = {}                 # step -> event "B of the step has read its window"
        self.last = None

    def seed_memory(self, memory_k, memory_v):
        """memory_k / memory_v: [S,1,128,5,44,44] (the reference layout): the pairs of steps -5 .. -1"""
        mk, mv = self.net._mem_from_ref(memory_k), self.net._mem_from_ref(memory_v)
        torch.cuda.synchronize()
        self.t, self.ev_a, self.ev_b = 0, {}, {}
        for j in range(self.T):
            e = (j - self.T) % self.R
            self.ring_k[:, e].copy_(mk[:, j])
            self.ring_v[:, e].copy_(mv[:, j])
        torch.cuda.synchronize()

    def load(self, frames0, frames1, slot=None, sub=None):
        """frames of one time step ([S,3,H,W] each) into every (slot, sub-step) -- the benchmark -- or into one"""
        S = self.nstreams
        for i, p in enumerate(self.slots):
            if slot is not None and slot != i:
                continue
            for g in range(self.group):
                if sub is None or sub == g:
                    p.f0[g * S:(g + 1) * S].copy_(frames0)
                    p.f1[g * S:(g + 1) * S].copy_(frames1)

    def replay_free(self):
        """enqueue time step t; returns (slot, sub-step) whose `mask` will hold its result.  The first step of a group replays
        the group's A graph (its frames must all be loaded by then: a look-ahead of group - 1 frames)."""
        t = self.t
        S, G = self.nstreams, self.group
        grp, g = divmod(t, G)
        i = grp % self.inflight
        p, s = self.slots[i], self.streams[i]
        with torch.cuda.stream(s):
            if g == 0:
                # EVERY reader of the ring entries about to be overwritten: the entry of step u last held the pair of step
                # u - R, which the B parts of steps u - R .. u - R + T - 1 gather (index_select on their own streams).  Waiting
                # for the numerically last of them only (round 3) left the other four ordered by queueing distance, not by an
                # event
                waited = set()
                for u in range(t, t + G):
                    for r in range(u - self.R, u - self.R + self.T):
                        old = self.ev_b.get(r)
                        if old is not None and r not in waited:
                            waited.add(r)
                            s.wait_event(old)
                for u in range(t, t + G):                    # step u - R is the oldest reader of anything still to be rewritten
                    self.ev_b.pop(u - self.R, None)
                p.graph_a.replay()
                for u in range(G):
                    e = (t + u) % self.R
                    self.ring_k[:, e].copy_(p.pk[u * S:(u + 1) * S, 0])
                    self.ring_v[:, e].copy_(p.pv[u * S:(u + 1) * S, 0])
                ea = torch.cuda.Event()
                ea.record(s)
                for u in range(G):
                    self.ev_a[t + u] = ea
            for u in range(t - self.T + 1, t):
                if u in self.ev_a and u // G != grp:         # (the group's own entries are ordered by the stream)
                    s.wait_event(self.ev_a[u])
            e = t % self.R
            torch.index_select(self.ring_k, 1, self.idx[e], out=p.keys[g])
            torch.index_select(self.ring_v, 1, self.idx[e], out=p.values[g])
            p.graph_b[g].replay()
            eb = torch.cuda.Event()
            eb.record(s)
            self.ev_b[t] = eb
        self.ev_a.pop(t - self.T - G, None)
        self.t = t + 1
        self.last = (i, g)
        return i if G == 1 else (i, g)

    def masks(self, slot=None, sub=0):
        if slot is None:
            slot, sub = self.last
        elif isinstance(slot, tuple):
            slot, sub = slot
        return self.slots[slot].mask[sub]

    def memory(self):
        """the window after the last enqueued step, reference layout [S,1,128,5,44,44] (f32), oldest frame first"""
        torch.cuda.synchronize()
        e = (self.t - 1) % self.R
        k = torch.index_select(self.ring_k, 1, self.idx[e])
        v = torch.index_select(self.ring_v, 1, self.idx[e])
        h = w = int(round(k.shape[2] ** 0.5))
        return self.net._mem_to_ref(k, h, w), self.net._mem_to_ref(v, h, w)
