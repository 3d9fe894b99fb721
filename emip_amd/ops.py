"""Tensor-level wrappers over the C ABI (include/emip_hip.h).

PyTorch is used for device memory and streams only: every function here checks
layouts on the host, allocates the output with torch.empty and hands raw device
pointers to libemip_hip.so on the current stream.  Activations are channels-last
([..., C] with unit channel stride; a channel slice of a wider buffer is fine)."""
import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, EMIP_BF16, EMIP_F32  # noqa: F401


def dt_code(dtype):
    if dtype == torch.float32:
        return EMIP_F32
    if dtype == torch.bfloat16:
        return EMIP_BF16
    raise TypeError(f"emip_amd supports float32 and bfloat16 activations, got {dtype}")


def _stream():
    """raw handle of the current HIP stream of the current device (the C accessors: torch.cuda.current_stream() costs 8 us
    of Python per call, and there is one call per launch)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _p(t):
    return None if t is None else t.data_ptr()


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _dev(t):
    if not t.is_cuda:
        raise _lib.EmipLibraryError("emip_amd ops need device tensors (there is no CPU path)")


def rows(t):
    """(M, C, ld) of a channels-last tensor whose leading dims collapse to rows of stride ld."""
    _dev(t)
    assert t.dim() >= 2 and t.stride(-1) == 1, (t.shape, t.stride())
    ld = t.stride(-2)
    m = t.shape[-2]
    for i in range(t.dim() - 3, -1, -1):
        assert t.shape[i] == 1 or t.stride(i) == t.stride(i + 1) * t.shape[i + 1], (t.shape, t.stride())
        m *= t.shape[i]
    return m, t.shape[-1], ld


class GradArena:
    """The zero-initialised f32 reduction targets of one backward pass (weight / bias / norm gradients: what the kernels
    accumulate into with atomics and what autograd then hands to the optimizer as .grad) carved from ONE buffer that one
    fill clears at the start of the step, instead of one clearing launch per tensor (~830 per EMIP-short step).  The
    buffer is sized by the demand of the previous step; whatever does not fit (first step, changed shapes) falls back
    to torch.zeros.  Views stay valid until the next begin(): train_step drops the old gradients first."""

    ALIGN = 64            # floats (256 B)

    def __init__(self):
        self.buf, self.cursor, self.limit, self.need, self.active = None, 0, 0, 0, False

    def begin(self, device):
        if self.need and (self.buf is None or self.buf.numel() < self.need or self.buf.device != torch.device(device)):
            self.buf = torch.empty(self.need + (self.need >> 4), dtype=torch.float32, device=device)
        self.limit = min(self.need, self.buf.numel()) if self.buf is not None else 0
        if torch.device(device).type == "cuda":
            STEP_STREAM[_dev_index(device)] = torch.cuda.current_stream(torch.device(device))
        if self.limit:
            self.buf[:self.limit].zero_()
        self.cursor, self.need, self.active = 0, 0, True

    def end(self, failed=False):
        """failed: the step raised somewhere between begin() and here.  Deferred results then have no complete set of
        `.grad` tensors to live in, and their arena slices are re-carved by the next begin(): everything queued is DROPPED
        (WgradQueue.reset) instead of being flushed into gradients the caller will not use -- a fixup() that raises here
        would mask the original exception and leave stale entries for the next step to add to fresh gradients."""
        self.active = False
        if failed:
            WGRADS.reset()
            return
        WGRADS.flush()                       # nothing deferred outlives the step
        WGRADS.fixup()                       # ... and every deferred result is what its parameter's .grad holds

    def zeros(self, shape, device):
        n = 1
        for d in shape:
            n *= d
        if not self.active:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        pad = (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.need += pad
        if self.cursor + pad > self.limit or self.buf.device != torch.device(device):
            return torch.zeros(shape, dtype=torch.float32, device=device)
        out = self.buf[self.cursor:self.cursor + n].view(shape)
        self.cursor += pad
        return out


ARENA = GradArena()


class WgradQueue:
    """Deferred Linear weight gradients of a training step.  While the gradient arena is active, gemm_tn does not launch:
    it takes the (pre-cleared) output slices from the arena, remembers the operands and returns.  flush() hands everything
    queued to ONE grouped launch (emip_gemm_tn8_group: thousands of work items, no launch short of tiles, m splits only to
    bound the item length).  The operands (dY, X) stay alive until then -- a few GB at batch 32 on a 288-GB device.
    Flush points: the end of backward (train_step), before a gradient bucket leaves for its all-reduce (GradReducer), and
    whenever MAX problems are waiting.

    One queue PER STREAM: an item waits in the queue of the stream its backward node ran on and is launched there.  With PVT
    stages 3-4 on a forked stream (model.FORK_DEEP_TRAIN) the grouped launch of that branch's ~250 weight gradients then
    depends on that branch only -- in the captured step it runs beside the backward of stages 1-2 instead of behind it --
    and no launch reads operands another stream produced."""

    MAX = 160

    def __init__(self):
        self.queues, self.enabled = {}, True    # raw stream handle -> [torch stream or None (host tensors), items]
        self.capture_pool = None        # pinned staging buffers handed in for a hipGraph capture (train.GraphedTrainStep)
        self.stage = {}
        self.owners = []                # (arena slice, parameter) of every deferred result of this step
        self.post = []                  # (packed result, parameter, unpack): convolution gradients, added to .grad by fixup()
        self.fixed = 0                  # results fixup() had to add by hand (diagnostics / tests)

    @property
    def items(self):
        """everything waiting, whatever its stream"""
        return [it for _, q in self.queues.values() for it in q]

    def _queue(self, t):
        if not t.is_cuda:
            return self.queues.setdefault(0, [None, []])
        h = _stream()
        q = self.queues.get(h)
        if q is None:
            q = self.queues[h] = [torch.cuda.current_stream(t.device), []]
        return q

    def add(self, a, b, c, db, M, N, K, lda, ldb, owners=()):
        # wide (256 x 320) tiles where the output shape fills them (gemm_tn16.hip), 128 x 128 otherwise (gemm_tn8.hip)
        kind = 16 if WIDE_WGRAD and _lib.load().emip_gemm_tn16_eligible(M, N, K, lda, ldb) else 8
        q = self._queue(a)
        q[1].append((a, b, c, db, M, N, K, lda, ldb, kind, None))
        self.owners += [(t, p) for t, p in owners if t is not None and p is not None]
        if len(q[1]) >= self.MAX:
            self._flush_queue(q)

    def add_conv(self, dy, x, dw, db, cv, param, unpack, bias_param=None):
        B, H, W, Cin, ldx, Cout, lddy, kh, kw, stride, pad = cv
        Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        q = self._queue(dy)
        q[1].append((dy, x, dw, db, B * Ho * Wo, Cout, kh * kw * Cin, lddy, ldx, 16, cv))
        self.post.append((dw, param, unpack))
        if db is not None and bias_param is not None:
            self.owners.append((db, bias_param))
        if len(q[1]) >= self.MAX:
            self._flush_queue(q)

    def reset(self):
        """drop everything queued or registered (the exception path of a training step): no launch, no gradient touched"""
        self.queues, self.owners, self.post = {}, [], []

    def fixup(self, params=None):
        """A deferred result is written into its arena slice AFTER autograd has taken that (still zero) slice as the
        parameter's gradient.  That is only the gradient if AccumulateGrad kept the slice itself as `.grad` -- it does for
        a parameter used once per backward, but a second use of the same weight, a tensor hook or create_graph make autograd
        sum or clone the zeros out of place, and the result would land in an orphaned slice.  Call after flush(): every
        result whose parameter's `.grad` is not the slice is added to that `.grad`.  params: restrict to these parameters
        (a gradient bucket about to leave); owners whose gradient is not assigned yet stay registered."""
        if not self.owners and not self.post:
            return
        assert not self.items, "fixup() runs behind flush()"
        only = None if params is None else {id(p) for p in params}
        # the lists are taken over first: whatever happens below (a final fixup that finds a parameter without .grad
        # raises), no entry of this step survives into the next one, whose arena slices hold other tensors
        post, owners, self.post, self.owners = self.post, self.owners, [], []
        later, keep, orphan = [], [], None
        for t, p, unpack in post:                  # convolution gradients: packed [Cout][ky][kx][ci] -> the parameter's layout
            if (only is not None and id(p) not in only) or (p.grad is None and params is not None):
                later.append((t, p, unpack))
            elif p.grad is None:
                orphan = p
            else:
                p.grad.add_(unpack(t).to(p.grad.dtype))
        for t, p in owners:
            g = p.grad
            if (only is not None and id(p) not in only) or (g is None and params is not None):
                keep.append((t, p))
            elif g is None:
                orphan = p
            elif g.data_ptr() != t.data_ptr() or g.numel() != t.numel():
                g.add_(t.view_as(g).to(g.dtype))
                self.fixed += 1
        if params is not None:                     # a partial fixup (one gradient bucket): the rest stays registered
            self.post, self.owners = later, keep
        if orphan is not None:
            raise _lib.EmipLibraryError("a deferred weight gradient has no parameter .grad to live in")

    def flush(self):
        """every queue, each on its own stream"""
        queues, self.queues = self.queues, {}
        for q in queues.values():
            self._flush_queue(q)

    def _flush_queue(self, q):
        st, items = q
        if not items:
            return
        q[1] = []
        ctx = torch.cuda.stream(st) if st is not None else _nullcontext()
        with ctx:
            for kind in (8, 16):
                sel = [it for it in items if it[9] == kind]
                if sel:
                    self._launch(sel, kind)

    def capture_buffers(self, n):
        """n pinned buffers, each large enough for the record table of one flush of either kind"""
        lib = _lib.load()
        rs = max(lib.emip_gemm_tn8_group_recsize(), lib.emip_gemm_tn16_recsize())
        return [torch.empty(rs * self.MAX, dtype=torch.uint8).pin_memory() for _ in range(n)]

    @staticmethod
    def _plan(lib, items, kind, host, rs):
        """the work-item records of one grouped launch, written into the pinned buffer `host`; returns the item count"""
        total = 0
        for i, (a, b, c, db, M, N, K, lda, ldb, _, cv) in enumerate(items):
            rec = host.data_ptr() + i * rs
            if kind == 8:
                n = lib.emip_gemm_tn8_group_plan(rec, _p(a), _p(b), _p(c), _p(db), M, N, K, lda, ldb, K, total)
            elif cv is None:
                n = lib.emip_gemm_tn16_plan(rec, _p(a), _p(b), _p(c), _p(db), M, N, K, lda, ldb, K, total, 0)
            else:
                n = lib.emip_conv_wgrad16_plan(rec, _p(a), _p(b), _p(c), _p(db), *cv, total, 0)
            if n <= 0:
                raise _lib.EmipLibraryError("weight-gradient plan (%d) failed for %r" % (kind, (M, N, K, lda, ldb, cv)))
            total += n
        return total

    def _launch(self, items, kind):
        lib = _lib.load()
        rs = lib.emip_gemm_tn8_group_recsize() if kind == 8 else lib.emip_gemm_tn16_recsize()
        dev = items[0][0].device
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture: the table goes through a pinned buffer that belongs to the graph's owner (the copy node
            # reads it on every replay; the operand addresses in it are the graph pool's and do not change), no events
            if not self.capture_pool:
                raise _lib.EmipLibraryError("weight-gradient flush inside a graph capture without a staging buffer "
                                            "(WgradQueue.capture_pool: see train.GraphedTrainStep)")
            host = self.capture_pool.pop()
            total = self._plan(lib, items, kind, host, rs)
            table = torch.empty(rs * len(items), dtype=torch.uint8, device=dev)
            table.copy_(host[:rs * len(items)], non_blocking=True)
            _lib.call("emip_gemm_tn8_group" if kind == 8 else "emip_gemm_tn16_group", _p(table), len(items), total, _stream())
            return
        # The records go to the device through one of two pinned staging buffers and a non-blocking copy: a pageable .to()
        # is stream-ordered AND blocking, i.e. it stalls the host until the GPU has caught up (5 ms per flush, measured).
        st = self.stage.get(kind)
        if st is None or st[0][0].numel() < rs * self.MAX:
            st = self.stage[kind] = [[torch.empty(rs * self.MAX, dtype=torch.uint8).pin_memory() for _ in range(2)], [None, None], 0]
        k = st[2] = st[2] ^ 1
        if st[1][k] is not None:
            st[1][k].synchronize()                 # the copy that read this buffer two flushes ago (long done)
        host = st[0][k]
        total = self._plan(lib, items, kind, host, rs)
        with torch.cuda.device(dev):
            table = torch.empty(rs * len(items), dtype=torch.uint8, device=dev)
            table.copy_(host[:rs * len(items)], non_blocking=True)
            st[1][k] = torch.cuda.Event()
            st[1][k].record()
            _lib.call("emip_gemm_tn8_group" if kind == 8 else "emip_gemm_tn16_group", _p(table), len(items), total, _stream())


WIDE_WGRAD = True      # deferred weight gradients on the 256 x 320 tiles of gemm_tn16.hip where the shape fills them (else all on gemm_tn8)

WGRADS = WgradQueue()


def _dev_index(device):
    dev = torch.device(device)
    return dev.index if dev.index is not None else torch.cuda.current_device()


FORK_STREAMS = {}       # (device index, name) -> side streams the model enqueues whole branches on (CoUpdater.run: PVT stages 3-4, ...)
FORK_USED = set()       # keys of FORK_STREAMS that took work since the last join_forks()


def fork_stream(device, priority=0, name="deep"):
    """the side stream `name` of `device` for a forked branch (created on first use); the caller is about to enqueue on it, so
    the next join_forks() waits for it"""
    key = (_dev_index(device), name)
    st = FORK_STREAMS.get(key)
    if st is None:
        st = FORK_STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", key[0]), priority=priority)
    FORK_USED.add(key)
    return st


def step_streams(device):
    """every stream a training step may have enqueued backward kernels on: the caller's current one and the forked branches'"""
    di = _dev_index(device)
    out = [torch.cuda.current_stream(torch.device("cuda", di))]
    for (idx, _), st in FORK_STREAMS.items():
        if idx == di:
            out.append(st)
    main = STEP_STREAM.get(di)
    if main is not None and all(main.cuda_stream != o.cuda_stream for o in out):
        out.append(main)
    return out


STEP_STREAM = {}        # device index -> the stream train_step runs on (GradArena.begin notes it)


def join_forks(device=None):
    """the current stream waits for everything the forked branches hold so far (end of backward, before the optimizer reads the
    gradients / the arena is cleared again / a gradient bucket leaves)"""
    for key in sorted(FORK_USED):
        # only streams that took work in this step: inside a hipGraph capture a wait on a stream that is not part of the
        # capture would tie the graph to work outside it
        if device is None or _dev_index(device) == key[0]:
            st = FORK_STREAMS[key]
            torch.cuda.current_stream(st.device).wait_stream(st)
            FORK_USED.discard(key)


def flush_wgrads():
    """launch the weight gradients the training step has deferred so far (no-op when none are waiting)"""
    WGRADS.flush()


def grad_zeros(shape, device):
    """f32 zeros for a reduction target that becomes a gradient (see GradArena)"""
    return ARENA.zeros(tuple(shape), device)


def _nbytes(t):
    return t.numel() * t.element_size() if t is not None else 0


def gemm_stats_ws_bytes(M, N):
    return int(_lib.load().emip_gemm_stats_ws_bytes(int(M), int(N)))


def gemm(a, w, bias=None, res=None, act=ACT_NONE, out=None, a2=None, ln_stats=None, ln_eps=0.0, out_stats=None,
         zero=None, colsum=None, stats_ws=None):
    """out[m, n] = act(a[m, :] . w[n, :] (+ a2 . w[n, K1:]) + bias[n]) + res[m, n]
    ln_stats f32 [M,2]: normalise the rows of a on the fly ((x - mean) * rstd; gamma / beta folded into w / bias);
    colsum f32 [N] (= w.float().sum(1)) with ln_stats: the same LayerNorm applied on the output side (emip_gemm_lne);
    out_stats f32 [M,2]: accumulate (sum, sum of squares) of the stored rows; zero: scratch tensor this launch clears;
    stats_ws: workspace of emip_gemm_ln_ws (out_stats of rows spanning more than two column tiles without a second pass)."""
    M, K1, lda = rows(a)
    N, K = w.shape
    lda2 = 0
    if a2 is not None:
        M2, K2, lda2 = rows(a2)
        assert M2 == M and K1 + K2 == K
    else:
        assert K1 == K, (K1, K)
    if out is None:
        out = torch.empty(a.shape[:-1] + (N,), dtype=a.dtype, device=a.device)
    Mo, No, ldc = rows(out)
    assert Mo == M and No == N and w.dtype == a.dtype and out.dtype == a.dtype and w.is_contiguous()
    ldr = 0
    if res is not None:
        Mr, Nr, ldr = rows(res)
        assert Mr == M and Nr == N and res.dtype == a.dtype
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    if ln_stats is None and out_stats is None and zero is None:
        _lib.call("emip_gemm", _p(a), _p(a2), _p(w), _p(out), _p(bias), _p(res), M, N, K, K1, lda, lda2, K, ldc, ldr,
                  act, 1, 0, 0, 0, 0, dt_code(a.dtype), _stream())
        return out
    for t in (ln_stats, out_stats):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == 2 * M)
    if colsum is not None and ln_stats is not None:
        assert a2 is None and colsum.dtype == torch.float32 and colsum.numel() == N and colsum.is_contiguous()
        _lib.call("emip_gemm_lne", _p(a), _p(w), _p(out), _p(bias), _p(res), M, N, K, lda, K, ldc, ldr, act, _p(ln_stats),
                  _p(colsum), float(ln_eps), _p(out_stats), _p(zero), _nbytes(zero), dt_code(a.dtype), _stream())
        return out
    if stats_ws is not None and out_stats is not None:
        # uint8 workspace (gemm_stats_ws_bytes, ticket block zero): row statistics combined inside the launch
        assert stats_ws.dtype == torch.uint8 and stats_ws.is_contiguous()
        _lib.call("emip_gemm_ln_ws", _p(a), _p(a2), _p(w), _p(out), _p(bias), _p(res), M, N, K, K1, lda, lda2, K, ldc, ldr, act,
                  1, 0, 0, 0, 0, _p(ln_stats), K, float(ln_eps), _p(out_stats), _p(zero), _nbytes(zero), dt_code(a.dtype),
                  _p(stats_ws), stats_ws.numel(), _stream())
        return out
    _lib.call("emip_gemm_ln", _p(a), _p(a2), _p(w), _p(out), _p(bias), _p(res), M, N, K, K1, lda, lda2, K, ldc, ldr, act,
              1, 0, 0, 0, 0, _p(ln_stats), K, float(ln_eps), _p(out_stats), _p(zero), _nbytes(zero), dt_code(a.dtype),
              _stream())
    return out


def gemm_rowscale(a, w, bias, res, rowscale, rs_rows):
    """res + rowscale[m // rs_rows] * (a w^T + bias): the branch of a residual block scaled per sample in the GEMM's epilogue
    (stochastic depth).  Returns None when the launch is not one the 8-wave body takes (the caller then scales separately)."""
    M, K, lda = rows(a)
    N = w.shape[0]
    if not (a.dtype == torch.bfloat16 and _lib.load().emip_gemm8_dispatch(M, N, K, lda, K, K, 0, 0)):
        return None
    out = torch.empty(a.shape[:-1] + (N,), dtype=a.dtype, device=a.device)
    _, _, ldr = rows(res)
    assert rowscale.dtype == torch.float32 and rowscale.is_contiguous() and rowscale.numel() * rs_rows >= M
    _lib.call("emip_gemm8_rs", _p(a), None, _p(w), _p(out), _p(bias), _p(res), M, N, K, K, lda, 0, K, N, ldr, ACT_NONE, None,
              None, 0.0, None, None, 0, _p(rowscale), int(rs_rows), 0, _stream())
    return out


def gemm8(a, w, bias=None, res=None, act=ACT_NONE, out=None, a2=None, ln_stats=None, ln_eps=0.0, colsum=None,
          out_stats=None, zero=None, cfg=0):
    """emip_gemm8: the 8-wave bf16 body (same arithmetic and hooks as `gemm`; the LayerNorm hook is the output-side form)"""
    M, K1, lda = rows(a)
    N, K = w.shape
    lda2 = 0
    if a2 is not None:
        M2, K2, lda2 = rows(a2)
        assert M2 == M and K1 + K2 == K
    else:
        assert K1 == K, (K1, K)
    assert a.dtype == torch.bfloat16 and w.dtype == a.dtype and w.is_contiguous()
    if out is None:
        out = torch.empty(a.shape[:-1] + (N,), dtype=a.dtype, device=a.device)
    Mo, No, ldc = rows(out)
    assert Mo == M and No == N and out.dtype == a.dtype
    ldr = 0
    if res is not None:
        Mr, Nr, ldr = rows(res)
        assert Mr == M and Nr == N and res.dtype == a.dtype
    for t in (ln_stats, out_stats):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == 2 * M)
    assert (ln_stats is None) == (colsum is None)
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == N)
    _lib.call("emip_gemm8", _p(a), _p(a2), _p(w), _p(out), _p(bias), _p(res), M, N, K, K1, lda, lda2, K, ldc, ldr, act,
              _p(ln_stats), _p(colsum), float(ln_eps), _p(out_stats), _p(zero), _nbytes(zero), int(cfg), _stream())
    return out


def conv8(x, w, kh, kw, stride=1, pad=0, bias=None, res=None, act=ACT_NONE, out=None, zero=None, out_stats=None, cfg=0,
          ln_stats=None, tapsum=None, ln_eps=0.0, stats_ws=None):
    """emip_conv8: implicit-GEMM conv on the 8-wave bf16 body; x [B,H,W,Cin] channels-last, w packed [Cout, kh*kw*Cin];
    ln_stats f32 [B*H*W, 2] + tapsum f32 [kh*kw, Cout]: LayerNorm of the input pixels applied on the output side, per tap"""
    _dev(x)
    B, H, W, Cin = x.shape
    assert x.stride(-1) == 1 and x.stride(1) == W * x.stride(2) and (B == 1 or x.stride(0) == H * x.stride(1))
    ldx = x.stride(2)
    Cout = w.shape[0]
    assert w.shape[1] == kh * kw * Cin and w.is_contiguous() and w.dtype == x.dtype == torch.bfloat16
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    Mo, No, ldy = rows(out)
    assert Mo == B * Ho * Wo and No == Cout
    ldr = 0
    if res is not None:
        Mr, Nr, ldr = rows(res)
        assert Mr == Mo and Nr == Cout
    if ln_stats is not None:
        assert ln_stats.dtype == torch.float32 and ln_stats.numel() == 2 * B * H * W and ln_stats.is_contiguous()
        assert tapsum.dtype == torch.float32 and tuple(tapsum.shape) == (kh * kw, Cout) and tapsum.is_contiguous()
    if stats_ws is not None and out_stats is not None:      # row statistics of wide rows combined inside the launch (see gemm)
        _lib.call("emip_conv8_ws", _p(x), _p(w), _p(out), _p(bias), _p(res), B, H, W, Cin, ldx, Cout, kh, kw, stride, pad, ldy,
                  ldr, act, _p(ln_stats), _p(tapsum), float(ln_eps), _p(out_stats), _p(zero), _nbytes(zero), int(cfg),
                  _p(stats_ws), stats_ws.numel(), _stream())
        return out
    _lib.call("emip_conv8", _p(x), _p(w), _p(out), _p(bias), _p(res), B, H, W, Cin, ldx, Cout, kh, kw, stride, pad, ldy,
              ldr, act, _p(ln_stats), _p(tapsum), float(ln_eps), _p(out_stats), _p(zero), _nbytes(zero), int(cfg), _stream())
    return out


def gemm_batched(a, w, out, batch, M, N, K, lda, ldw, ldc, bsA, bsW, bsC):
    """Strided-batched gemm on raw views (a, w, out are tensors whose data_ptr is the first operand)."""
    _dev(a)
    _lib.call("emip_gemm", _p(a), None, _p(w), _p(out), None, None, M, N, K, K, lda, 0, ldw, ldc, 0, ACT_NONE, batch,
              bsA, bsW, bsC, 0, dt_code(a.dtype), _stream())
    return out


def gemm_batched_bias(a, w, out, batch, M, N, K, lda, ldw, ldc, bsA, bsW, bsC, bias=None, act=ACT_NONE):
    """gemm_batched with the bias / activation epilogue (bias f32 [N], shared by the batch)"""
    _dev(a)
    _lib.call("emip_gemm", _p(a), None, _p(w), _p(out), _p(bias), None, M, N, K, K, lda, 0, ldw, ldc, 0, act, batch,
              bsA, bsW, bsC, 0, dt_code(a.dtype), _stream())
    return out


def gemm8_batched(a, w, out, batch, M, N, K, lda, ldw, ldc, bsA, bsW, bsC, bias=None, act=ACT_NONE, cfg=0):
    """gemm_batched_bias on the 8-wave bf16 body (K % 64 == 0, strides multiples of 8)"""
    _dev(a)
    assert a.dtype == torch.bfloat16 and K % 64 == 0
    _lib.call("emip_gemm8_batched", _p(a), _p(w), _p(out), _p(bias), M, N, K, lda, ldw, ldc, act, batch, bsA, bsW, bsC, int(cfg),
              _stream())
    return out


def im2col3x3(x):
    """x [B,H,W,C] channels-last -> the 3 x 3 patch matrix [B, H*W, 9*C] (zero padding, tap-major columns)"""
    _dev(x)
    B, H, W, C = x.shape
    _, _, ldx = rows(x)
    y = torch.empty((B, H * W, 9 * C), dtype=x.dtype, device=x.device)
    _lib.call("emip_im2col3x3", _p(x), ldx, _p(y), 9 * C, B, H, W, C, dt_code(x.dtype), _stream())
    return y


def col2im3x3(dy, B, H, W, C, out=None):
    """adjoint of im2col3x3: dy [B, H*W, 9*C] -> [B, H*W, C] (out: a contiguous tensor of that shape to write into)"""
    _dev(dy)
    dx = out if out is not None else torch.empty((B, H * W, C), dtype=dy.dtype, device=dy.device)
    assert dx.is_contiguous() and dx.numel() == B * H * W * C and dy.is_contiguous()
    _lib.call("emip_col2im3x3", _p(dy), 9 * C, _p(dx), C, B, H, W, C, dt_code(dy.dtype), _stream())
    return dx


def conv2d_splitk(x, w, kh, kw, stride, pad, bias, acc, ksplit, ln_stats=None, ln_eps=0.0, zero=None):
    """split-K conv: partial sums (bias included once) are ADDED into acc f32 [B*Ho*Wo, Cout] (zero beforehand)"""
    _dev(x)
    B, H, W, Cin = x.shape
    ldx = x.stride(2)
    Cout = w.shape[0]
    assert acc.dtype == torch.float32 and acc.is_contiguous() and acc.shape[-1] == Cout and ksplit > 1
    _lib.call("emip_conv2d_splitk", _p(x), _p(w), None, _p(bias), None, B, H, W, Cin, ldx, Cout, kh, kw, stride, pad,
              Cout, 0, ACT_NONE, _p(zero), _nbytes(zero), _p(ln_stats), float(ln_eps), None, _p(acc), Cout, int(ksplit),
              dt_code(x.dtype), _stream())
    return acc


KSPLIT_MAX = 2       # two partial tiles per output tile: their f32 atomic adds commute (reproducible); 16 before round 4


def ksplit_for(M, Cout, K, dtype):
    """split count of emip_conv2d_ksplit for an [M, Cout] output over K: 0 = enough tiles / too short a walk to split"""
    tiles = ((M + 63) // 64) * ((Cout + 63) // 64)
    nk = K // (64 if dtype == torch.bfloat16 else 32)
    if tiles >= 192 or nk < 16:
        return 0
    return max(2, min(nk // 4, (256 + tiles - 1) // tiles, KSPLIT_MAX))


def conv2d_ksplit(x, w, kh, kw, stride, pad, ksplit, bias=None, act=ACT_NONE, ln_stats=None, ln_eps=0.0, out_stats=None, out=None):
    """conv2d / conv2d_ln with the K walk split over `ksplit` workgroups per tile and reduced inside the launch; the zeroed
    accumulator + tickets are one fresh allocation (inside a graph capture it is private to that graph, so concurrent replays
    do not share it; the launch leaves it zero)"""
    _dev(x)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    assert w.shape[1] == kh * kw * Cin and w.is_contiguous() and w.dtype == x.dtype
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    M = B * Ho * Wo
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    Mo, No, ldy = rows(out)
    assert Mo == M and No == Cout
    ntick = ((M + 63) // 64) * ((Cout + 63) // 64)
    scratch = torch.zeros(M * Cout + ntick, dtype=torch.float32, device=x.device)
    _lib.call("emip_conv2d_ksplit", _p(x), _p(w), _p(out), _p(bias), B, H, W, Cin, x.stride(2), Cout, kh, kw, stride, pad, ldy,
              act, _p(ln_stats), float(ln_eps), _p(out_stats), _p(scratch), scratch.data_ptr() + 4 * M * Cout, int(ksplit),
              dt_code(x.dtype), _stream())
    return out


def rows_finalize(acc, dtype, shape, out_stats=None):
    """complete f32 rows [M, C] -> tensor of `shape` (last dim C) in `dtype`, + (sum, sum of squares) of the stored rows"""
    M, C = acc.shape
    y = torch.empty(shape, dtype=dtype, device=acc.device)
    _lib.call("emip_rows_finalize", _p(acc), C, _p(y), C, _p(out_stats), M, C, dt_code(dtype), _stream())
    return y


import ctypes as _ct


class ConvDesc(_ct.Structure):
    """emip_conv_desc of include/emip_hip.h"""
    _fields_ = [("X", _ct.c_void_p), ("W", _ct.c_void_p), ("Y", _ct.c_void_p), ("bias", _ct.c_void_p), ("R", _ct.c_void_p),
                ("B", _ct.c_int), ("H", _ct.c_int), ("Wd", _ct.c_int), ("Cin", _ct.c_int), ("ldx", _ct.c_long),
                ("Cout", _ct.c_int), ("KH", _ct.c_int), ("KW", _ct.c_int), ("stride", _ct.c_int), ("pad", _ct.c_int),
                ("ldy", _ct.c_long), ("ldr", _ct.c_long), ("act", _ct.c_int), ("ln_stats", _ct.c_void_p),
                ("ln_eps", _ct.c_float), ("out_stats", _ct.c_void_p), ("acc", _ct.c_void_p), ("ticket", _ct.c_void_p),
                ("ksplit", _ct.c_int), ("colsum", _ct.c_void_p)]


def conv_desc(x, w, k, stride, pad, bias, out, ln_stats, ln_eps, out_stats=None, res=None, act=ACT_NONE, acc=None,
              ticket=None, ksplit=1, colsum=None):
    """acc / ticket / ksplit: fused split-K (second problem of a pair): f32 [rows, Cout] accumulator and one u32 counter per
    64x64 output tile, zero before the first launch (every launch leaves them zero)"""
    B, H, W, Cin = x.shape
    assert w.shape[1] == k * k * Cin and w.is_contiguous() and w.dtype == x.dtype and ln_stats.dtype == torch.float32
    if ksplit > 1:
        assert acc.dtype == torch.float32 and acc.numel() >= rows(out)[0] * w.shape[0] and ticket.dtype == torch.int32
        assert ticket.numel() >= ((rows(out)[0] + 63) // 64) * ((w.shape[0] + 63) // 64)
    return ConvDesc(_p(x), _p(w), _p(out), _p(bias), _p(res), B, H, W, Cin, x.stride(2), w.shape[0], k, k, stride, pad,
                    rows(out)[2], rows(res)[2] if res is not None else 0, act, _p(ln_stats), float(ln_eps), _p(out_stats),
                    _p(acc) if ksplit > 1 else None, _p(ticket) if ksplit > 1 else None, int(ksplit), _p(colsum))


def conv2d_pair(da, db, dtype):
    """two convs (ConvDesc) with the normalising loader in one launch"""
    _lib.call("emip_conv2d_pair", _ct.addressof(da), _ct.addressof(db), dt_code(dtype), _stream())


def conv2d(x, w, kh, kw, stride=1, pad=0, bias=None, res=None, act=ACT_NONE, out=None, zero=None, ln_stats=None,
           ln_eps=0.0, out_stats=None):
    """x [B,H,W,Cin] channels-last, w packed [Cout, kh*kw*Cin] -> [B,Ho,Wo,Cout]."""
    _dev(x)
    B, H, W, Cin = x.shape
    assert x.stride(-1) == 1 and x.stride(1) == W * x.stride(2) and (B == 1 or x.stride(0) == H * x.stride(1))
    ldx = x.stride(2)
    Cout = w.shape[0]
    assert w.shape[1] == kh * kw * Cin and w.is_contiguous() and w.dtype == x.dtype
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    Mo, No, ldy = rows(out)
    assert Mo == B * Ho * Wo and No == Cout
    ldr = 0
    if res is not None:
        Mr, Nr, ldr = rows(res)
        assert Mr == Mo and Nr == Cout
    # zero: a scratch tensor the kernel's first workgroup clears (the statistics buffer of the normalisation that follows)
    if ln_stats is None and out_stats is None:
        _lib.call("emip_conv2d", _p(x), _p(w), _p(out), _p(bias), _p(res), B, H, W, Cin, ldx, Cout, kh, kw, stride, pad,
                  ldy, ldr, act, _p(zero), _nbytes(zero), dt_code(x.dtype), _stream())
        return out
    assert ln_stats is None or (ln_stats.dtype == torch.float32 and ln_stats.numel() == 2 * B * H * W)
    assert out_stats is None or (out_stats.dtype == torch.float32 and out_stats.numel() == 2 * Mo)
    _lib.call("emip_conv2d_ln", _p(x), _p(w), _p(out), _p(bias), _p(res), B, H, W, Cin, ldx, Cout, kh, kw, stride, pad,
              ldy, ldr, act, _p(zero), _nbytes(zero), _p(ln_stats), float(ln_eps), _p(out_stats), dt_code(x.dtype),
              _stream())
    return out


def layernorm(x, gamma, beta, eps, out=None, res=None, out_stats=None):
    """LN(x) [+ res]; out may alias x or res.  out_stats f32 [M,2]: (sum, sum of squares) of the stored rows"""
    M, C, ldx = rows(x)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    Mo, Co, ldy = rows(out)
    assert Mo == M and Co == C
    ldr = rows(res)[2] if res is not None else 0
    _lib.call("emip_layernorm", _p(x), ldx, _p(out), ldy, _p(gamma), _p(beta), _p(res), ldr, _p(out_stats), M, C,
              float(eps), dt_code(x.dtype), _stream())
    return out


def gemm_ln_out(a, w, gamma, beta, eps, bias=None, res=None, out=None):
    """res + LayerNorm(a w^T + bias) * gamma + beta over the N <= 128 output columns -- a post-norm Linear: the norm runs in
    the GEMM epilogue (emip_gemm8_lno) where the shape allows, else GEMM + emip_layernorm.  out may alias res."""
    M, K, lda = rows(a)
    N = w.shape[0]
    if out is None:
        out = torch.empty(a.shape[:-1] + (N,), dtype=a.dtype, device=a.device)
    if (a.dtype == torch.bfloat16 and N <= 128 and N % 8 == 0 and K % 64 == 0 and M >= 1024
            and w.is_contiguous() and lda % 8 == 0):
        _, _, ldc = rows(out)
        ldr = rows(res)[2] if res is not None else 0
        if ldc % 8 == 0 and ldr % 8 == 0:
            _lib.call("emip_gemm8_lno", _p(a), _p(w), _p(out), _p(bias), _p(res), _p(gamma), _p(beta), float(eps), M, N, K, lda,
                      w.stride(0), ldc, ldr, _stream())
            return out
    msg = gemm(a, w, bias=bias)
    return layernorm(msg, gamma, beta, eps, out=out, res=res)


def ffn_block_packs(w0, w2, dtype=torch.bfloat16):
    """mlp[0].weight [1024, 256] and mlp[2].weight [128, 1024] in the fragment order emip_ffn_block streams: one 1-KB piece = the
    MFMA A operand of all 64 lanes (lane = row r + 32 h holds 8 values).  W0: [chunk c][k-step i][h][r][j] = W0[32 c + r][16 i + 8 h
    + j]; W2: [chunk c][row tile d][k-step sp][h][r][j] = W2[32 d + r][32 c + 16 sp + 8 (j >> 2) + 4 h + (j & 3)] -- the hidden
    channel the accumulator register 8 sp + j of lane half h holds after H^T = W0 X^T."""
    assert tuple(w0.shape) == (1024, 256) and tuple(w2.shape) == (128, 1024)
    p0 = w0.detach().to(dtype).view(32, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
    sp, h, j = torch.meshgrid(torch.arange(2), torch.arange(2), torch.arange(8), indexing="ij")
    hid = (16 * sp + 8 * (j >> 2) + 4 * h + (j & 3)).to(w2.device)                     # [sp][h][j]
    w2v = w2.detach().to(dtype).view(4, 32, 32, 32)                                    # [d][r][c][hidden in chunk]
    p2 = w2v[:, :, :, hid]                                                             # [d][r][c][sp][h][j]
    p2 = p2.permute(2, 0, 3, 4, 1, 5).contiguous()                                     # [c][d][sp][h][r][j]
    return p0.view(-1), p2.view(-1)


def ffn_block(x1, x2, w0p, w2p, gamma, beta, eps, res=None, out=None):
    """emip_ffn_block: out = res + LayerNorm(GELU([x1 | x2] W0^T) W2^T) * gamma + beta on [.., 128] bf16 tokens"""
    _dev(x1)
    M, C, ld1 = rows(x1)
    _, _, ld2 = rows(x2)
    assert C == 128 and x2.shape[-1] == 128 and x1.dtype == x2.dtype == torch.bfloat16
    if out is None:
        out = torch.empty(x1.shape, dtype=x1.dtype, device=x1.device)
    ldo = rows(out)[2]
    ldr = rows(res)[2] if res is not None else 0
    _lib.call("emip_ffn_block", _p(x1), ld1, _p(x2), ld2, _p(w0p), _p(w2p), _p(gamma), _p(beta), float(eps), _p(res), ldr,
              _p(out), ldo, M, _stream())
    return out


# emip_attention_splitkv for >= KV_SPLIT_MIN_KEYS keys on grids of < KV_SPLIT_TARGET workgroups (two per CU)
KV_SPLIT_MIN_KEYS = 1024
KV_SPLIT_TARGET = 512


def attention(q, k, v, out, *, batch, heads, nwin, Lq, Lk, D, DV, q_bs, k_bs, v_bs, o_bs, ldq, ldk, ldv, ldo,
              q_hs=0, k_hs=0, v_hs=0, o_hs=0, q_rows=None, k_rows=None, q_gid=None, k_gid=None, scale=1.0,
              scores=None, s_bs=0, lds=0, ksplit=None, kv_rot=0):
    """... kv_rot: keys / values of batch element b are read from element (b + kv_rot) mod batch"""
    _dev(q)
    o_f32 = 1 if (out.dtype == torch.float32 and q.dtype != torch.float32) else 0
    # long key sets on small grids: split the keys over workgroups (>= 4 key tiles each).  The kernel runs one wave per SIMD
    # and workgroup, so it wants TWO workgroups per CU to hide its load -> MFMA -> softmax chain: 16 pairs of global matching
    # (256 workgroups unsplit) take 118 us unsplit and 80 us split in two (tools/corr_bench.py), hence the 512 target
    bkv = 64 if q.dtype == torch.bfloat16 else 32
    ntile = (Lk + bkv - 1) // bkv
    wgs = ((Lq + 127) // 128) * heads * batch * nwin
    if ksplit is not None:
        pass
    elif not (Lk >= KV_SPLIT_MIN_KEYS and wgs < KV_SPLIT_TARGET):
        ksplit = 1
    else:
        ksplit = max(1, min(ntile // 4, (KV_SPLIT_TARGET + wgs - 1) // wgs, 32))
    if ksplit > 1 or kv_rot:
        ws = (torch.empty(batch * nwin * heads * ksplit * Lq * (DV + 2), dtype=torch.float32, device=q.device)
              if ksplit > 1 else None)
        _lib.call("emip_attention_rot", _p(q), _p(k), _p(v), _p(out), _p(scores), batch, heads, nwin, Lq, Lk, D, DV,
                  q_bs, k_bs, v_bs, o_bs, s_bs, ldq, ldk, ldv, ldo, lds, q_hs, k_hs, v_hs, o_hs, _p(q_rows), _p(k_rows),
                  _p(q_gid), _p(k_gid), float(scale), o_f32, ksplit, _p(ws), int(kv_rot), dt_code(q.dtype), _stream())
        return out
    _lib.call("emip_attention", _p(q), _p(k), _p(v), _p(out), _p(scores), batch, heads, nwin, Lq, Lk, D, DV, q_bs,
              k_bs, v_bs, o_bs, s_bs, ldq, ldk, ldv, ldo, lds, q_hs, k_hs, v_hs, o_hs, _p(q_rows), _p(k_rows),
              _p(q_gid), _p(k_gid), float(scale), o_f32, dt_code(q.dtype), _stream())
    return out


def window_attention(q, k, v, out, rows, gid, tokens, scale, kv_rot=0, lse=None):
    """emip_window_attention: q / k / v / out [B, tokens, >=128] bf16 views (unit channel stride), rows / gid int32 [nwin, L];
    lse: f32 [B, tokens] receives the log2-sum-exp of every query (what window_attention_bwd needs)"""
    _dev(q)
    B = q.shape[0]
    nwin, L = rows.shape
    assert q.dtype == k.dtype == v.dtype == out.dtype == torch.bfloat16 and rows.dtype == torch.int32 and rows.is_contiguous()
    assert gid is None or (gid.dtype == torch.int32 and gid.shape == rows.shape and gid.is_contiguous())
    for t in (q, k, v, out):
        assert t.dim() == 3 and t.shape[0] == B and t.shape[1] == tokens and t.stride(2) == 1
    _lib.call("emip_window_attention", _p(q), _p(k), _p(v), _p(out), B, nwin, L, q.stride(1), k.stride(1), v.stride(1),
              out.stride(1), q.stride(0), k.stride(0), v.stride(0), out.stride(0), _p(rows), _p(gid), int(tokens), int(kv_rot),
              float(scale), _p(lse), _stream())
    return out


def wattn_merge_pack(wm, dtype=torch.bfloat16):
    """merge.weight [128, 128] in the fragment order of emip_window_attention_merge: [row tile dd][k-step ks = 2 d + sp][h][r][j] =
    Wm[32 dd + r][32 d + 16 sp + 8 (j >> 2) + 4 h + (j & 3)] (the channel the attention accumulator register 8 sp + j of tile d holds)"""
    assert tuple(wm.shape) == (128, 128)
    d, sp, h, j = torch.meshgrid(torch.arange(4), torch.arange(2), torch.arange(2), torch.arange(8), indexing="ij")
    ch = (32 * d + 16 * sp + 8 * (j >> 2) + 4 * h + (j & 3)).to(wm.device)               # [d][sp][h][j]
    w = wm.detach().to(dtype).view(4, 32, 128)[:, :, ch]                                  # [dd][r][d][sp][h][j]
    return w.permute(0, 2, 3, 4, 1, 5).contiguous().view(-1)                              # [dd][d][sp][h][r][j]


def wattn_q_pack(wq, dtype=torch.bfloat16):
    """q_proj.weight [128, 128] for the prologue of emip_window_attention_merge: [row tile dd][k-step ks][h][rho][j] =
    Wq[32 dd + swap23(rho)][16 ks + 8 h + j]; with bits 2 and 3 of the row index swapped inside every 16 the accumulator registers
    of Q^T = Wq X^T come out in the channel order an MFMA B fragment wants"""
    assert tuple(wq.shape) == (128, 128)
    rho = torch.arange(32)
    src = ((rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1)).to(wq.device)
    w = wq.detach().to(dtype).view(4, 32, 8, 2, 8)[:, src]                               # [dd][rho][ks][h][j]
    return w.permute(0, 2, 3, 1, 4).contiguous().view(-1)                                # [dd][ks][h][rho][j]


def window_attention_merge(q, k, v, out, rows, gid, tokens, scale, wm_pack, gamma, beta, eps, res=None, kv_rot=0, wq_pack=None):
    """emip_window_attention_merge: out = res + LayerNorm(merge(window attention)) * gamma + beta, one launch; wq_pack: q is the
    token matrix and the q projection runs in the launch too"""
    _dev(q)
    B = q.shape[0]
    nwin, L = rows.shape
    assert q.dtype == k.dtype == v.dtype == out.dtype == torch.bfloat16 and rows.dtype == torch.int32 and rows.is_contiguous()
    for t in (q, k, v, out) + ((res,) if res is not None else ()):
        assert t.dim() == 3 and t.shape[0] == B and t.shape[1] == tokens and t.stride(2) == 1
    _lib.call("emip_window_attention_merge", _p(q), _p(k), _p(v), _p(out), B, nwin, L, q.stride(1), k.stride(1), v.stride(1),
              out.stride(1), q.stride(0), k.stride(0), v.stride(0), out.stride(0), _p(rows), _p(gid), int(tokens), int(kv_rot),
              float(scale), _p(wm_pack), _p(gamma), _p(beta), float(eps), _p(res), res.stride(1) if res is not None else 0,
              res.stride(0) if res is not None else 0, _p(wq_pack), _stream())
    return out


def window_attention_bwd(q, k, v, out, dout, lse, rows, gid, tokens, scale, kv_rot=0):
    """emip_window_attention_bwd -> (dq, dk, dv) bf16 [B, tokens, 128]; out / dout contiguous, q / k / v may be column slices"""
    _dev(q)
    B = q.shape[0]
    nwin, L = rows.shape
    assert out.is_contiguous() and dout.is_contiguous() and out.shape == (B, tokens, 128) and dout.shape == out.shape
    assert lse.dtype == torch.float32 and lse.is_contiguous() and lse.numel() == B * tokens
    for t in (q, k, v):
        assert t.dtype == torch.bfloat16 and t.dim() == 3 and t.shape[0] == B and t.shape[1] == tokens and t.stride(2) == 1
    g = torch.empty((3, B, tokens, 128), dtype=torch.bfloat16, device=q.device)
    delta = torch.empty((B, tokens), dtype=torch.float32, device=q.device)
    _lib.call("emip_window_attention_bwd", _p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), _p(delta), _p(g[0]), _p(g[1]), _p(g[2]),
              B, nwin, L, q.stride(1), k.stride(1), v.stride(1), q.stride(0), k.stride(0), v.stride(0), _p(rows), _p(gid),
              int(tokens), int(kv_rot), float(scale), _stream())
    return g[0], g[1], g[2]


def match_eligible(t):
    """emip_match takes bf16 tokens [Z, n, 128] with unit channel stride, 128 <= n <= 2048, n % 8 == 0 (352 x 352 frames:
    n = 1936; 384 x 384 would be 2304: callers fall back to the generic attention kernel / batched GEMM there)"""
    n, C = t.shape[1], t.shape[2]
    return t.dtype == torch.bfloat16 and C == 128 and 128 <= n <= 2048 and n % 8 == 0 and t.stride(2) == 1


def match(q, k, W, scale, v=None, scores=None, kv_rot=0, sub_grid=True, lse=None):
    """emip_match: q, k bf16 [Z, n, 128] (unit channel stride), v f32 [Z, n, 2] or None (= the pixel grid of width W),
    scores bf16 [Zs, n, n] or None -> f32 [Z, n, 2]; keys / values of batch z come from batch (z + kv_rot) mod Z"""
    _dev(q)
    Z, n, C = q.shape
    assert C == 128 and k.shape == q.shape and q.dtype == k.dtype == torch.bfloat16
    assert q.stride(2) == 1 and k.stride(2) == 1
    out = torch.empty((Z, n, 2), dtype=torch.float32, device=q.device)
    Zs = 0
    if scores is not None:
        Zs = scores.shape[0]
        assert scores.shape == (Zs, n, n) and scores.dtype == torch.bfloat16 and scores.is_contiguous()
    if v is not None:
        assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == Z * n * 2
    _lib.call("emip_match", _p(q), _p(k), _p(v), _p(scores), _p(out), Z, Zs, n, int(W), q.stride(1), k.stride(1), q.stride(0),
              k.stride(0), int(kv_rot), float(scale), int(sub_grid), _p(lse), _stream())
    return out


def match_bwd(q, k, W, scale, out, dout, lse, v=None, dscores=None, kv_rot=0, sub_grid=True, accum=False):
    """emip_match_bwd -> (dq, dk) bf16 [Z, n, 128]; accum: dk is added into dq's buffer (q and k are the same tokens) and the one
    tensor is returned twice"""
    _dev(q)
    Z, n, C = q.shape
    assert C == 128 and k.shape == q.shape and q.dtype == k.dtype == torch.bfloat16 and q.stride(2) == 1 and k.stride(2) == 1
    assert out.dtype == dout.dtype == lse.dtype == torch.float32 and out.is_contiguous() and dout.is_contiguous()
    assert out.numel() == Z * n * 2 and dout.numel() == Z * n * 2 and lse.numel() == Z * n
    Zs = 0
    if dscores is not None:
        Zs = dscores.shape[0]
        assert dscores.shape == (Zs, n, n) and dscores.dtype == torch.bfloat16 and dscores.is_contiguous()
    if v is not None:
        assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == Z * n * 2
    dq = torch.empty((Z, n, 128), dtype=torch.bfloat16, device=q.device)
    dk = dq if accum else torch.empty_like(dq)
    stat = torch.empty((Z, n, 4), dtype=torch.float32, device=q.device)
    _lib.call("emip_match_bwd", _p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), _p(dscores), _p(stat), _p(dq), _p(dk), Z, Zs, n,
              int(W), q.stride(1), k.stride(1), q.stride(0), k.stride(0), int(kv_rot), float(scale), int(sub_grid), int(accum),
              _stream())
    return dq, dk


def sra_attention(q, kv, out, batch, heads, Lq, Lk, scale):
    """emip_sra_attention (bf16): q [B,Lq,C], kv [B,Lk,2C] (k | v), out [B,Lq,C], C = heads * 64, Lk <= 128"""
    _dev(q)
    C = heads * 64
    assert q.dtype == kv.dtype == out.dtype == torch.bfloat16 and q.is_contiguous() and kv.is_contiguous() and out.is_contiguous()
    assert q.numel() == batch * Lq * C and kv.numel() == batch * Lk * 2 * C and out.numel() == q.numel()
    _lib.call("emip_sra_attention", _p(q), _p(kv), _p(out), batch, heads, Lq, Lk, C, float(scale), _stream())
    return out


def sra_attention_lse(q, kv, out, batch, heads, Lq, Lk, scale):
    """sra_attention that also returns L f32 [batch, heads, Lq] (log2-sum-exp of the scaled scores) for sra_attention_bwd"""
    _dev(q)
    C = heads * 64
    assert q.dtype == kv.dtype == out.dtype == torch.bfloat16 and q.is_contiguous() and kv.is_contiguous() and out.is_contiguous()
    L = torch.empty((batch, heads, Lq), dtype=torch.float32, device=q.device)
    _lib.call("emip_sra_attention_lse", _p(q), _p(kv), _p(out), _p(L), batch, heads, Lq, Lk, C, float(scale), _stream())
    return L


def sra_attention_bwd(q, kv, out, dout, L, batch, heads, Lq, Lk, scale):
    """-> (dq bf16 [B,Lq,C], dkv [B,Lk,2C]: dK at columns 64 h, dV at C + 64 h; bf16 when batch * heads >= 256, else f32)"""
    C = heads * 64
    assert q.is_contiguous() and kv.is_contiguous() and out.is_contiguous() and dout.is_contiguous()
    dq = torch.empty_like(q)
    if batch * heads >= 256:          # one workgroup per (image, head) fills the chip: final bf16 values, stored
        dkv = torch.empty((batch, Lk, 2 * C), dtype=torch.bfloat16, device=q.device)
        _lib.call("emip_sra_attention_bwd_bf16", _p(q), _p(kv), _p(out), _p(dout), _p(L), _p(dq), _p(dkv), batch, heads, Lq, Lk,
                  C, float(scale), _stream())
        return dq, dkv
    dkv = grad_zeros((batch, Lk, 2 * C), q.device)          # arena slice inside a training step: no fill launch of its own
    _lib.call("emip_sra_attention_bwd", _p(q), _p(kv), _p(out), _p(dout), _p(L), _p(dq), _p(dkv), batch, heads, Lq, Lk, C,
              float(scale), _stream())
    return dq, dkv


def sra_block_eligible(C, Lk):
    return bool(_lib.load().emip_sra_block_eligible(C, Lk))


def swap23(n, device=None):
    """index permutation of the emip_sra_block weight packs: bits 2 and 3 swapped inside every 16 (an involution)"""
    i = torch.arange(n, device=device)
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)


def sra_block(x, stats, eps, wq, bq, colsum_q, kv, wp, bp, heads, scale, out_stats=None):
    """x <- x + proj(softmax((LN(x) wq^T) k^T scale) v), in place, one launch (bf16).  x [B,H,W,C] raw tokens with their row
    statistics stats [B*H*W, 2]; wq = (W_q gamma)[swap23] with bq / colsum_q in channel order, kv [B,Lk,2C], wp =
    W_proj[swap23][:, swap23]; out_stats [B*H*W, 2] receives the statistics of the new rows"""
    B, H, W, C = x.shape
    M, _, ldx = rows(x)
    Lk = kv.shape[1]
    assert x.dtype == wq.dtype == wp.dtype == kv.dtype == torch.bfloat16 and C == heads * 64
    assert wq.is_contiguous() and wp.is_contiguous() and kv.is_contiguous() and kv.shape == (B, Lk, 2 * C)
    assert wq.shape == wp.shape == (C, C)
    _lib.call("emip_sra_block", _p(x), ldx, _p(stats), float(eps), _p(wq), _p(bq), _p(colsum_q), _p(kv), _p(wp), _p(bp),
              _p(x), ldx, _p(out_stats), B, H * W, Lk, C, float(scale), _stream())
    return x


def sra_qattn(x, stats, eps, wq, bq, colsum_q, kv, heads, scale):
    """softmax((LN(x) wq^T) k^T scale) v -> [B,H,W,C] bf16 with the q projection computed inside the attention launch (one head
    per workgroup; C = 320).  wq = (W_q gamma)[swap23]"""
    B, H, W, C = x.shape
    M, _, ldx = rows(x)
    Lk = kv.shape[1]
    assert x.dtype == wq.dtype == kv.dtype == torch.bfloat16 and C == heads * 64 and wq.shape == (C, C)
    assert wq.is_contiguous() and kv.is_contiguous() and kv.shape == (B, Lk, 2 * C)
    out = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    _lib.call("emip_sra_qattn", _p(x), ldx, _p(stats), float(eps), _p(wq), _p(bq), _p(colsum_q), _p(kv), _p(out), C, B, H * W, Lk,
              C, float(scale), _stream())
    return out


def mdta_attn(q, kv_k, temperature, B, heads, P):
    """q [B,P,ldq] view, kv_k [B,P,ldk] view (the k half) -> attn [B,heads,64,64]."""
    _dev(q)
    ws = torch.empty(_lib.load().emip_mdta_ws_floats(B, heads), dtype=torch.float32, device=q.device)
    attn = torch.empty((B, heads, 64, 64), dtype=q.dtype, device=q.device)
    _lib.call("emip_mdta_attn", _p(q), q.stride(-2), q.stride(0), _p(kv_k), kv_k.stride(-2), kv_k.stride(0),
              _p(temperature), _p(ws), _p(attn), B, heads, P, dt_code(q.dtype), _stream())
    return attn


def dwconv3x3(x, wt, bias=None, act=ACT_NONE, out=None):
    B, H, W, C = x.shape
    M, _, ldx = rows(x)
    if out is None:
        out = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    _, _, ldy = rows(out)
    _lib.call("emip_dwconv3x3", _p(x), ldx, _p(out), ldy, _p(wt), _p(bias), B, H, W, C, act, dt_code(x.dtype),
              _stream())
    return out


def mlp_fc1dw_eligible(B, H, W, K, N):
    return bool(_lib.load().emip_mlp_fc1dw_eligible(B, H, W, K, N))


def mlp_fc1dw_band_rows(B, H, W, K, N):
    """output rows per band of the banded form (maps of more than 512 tokens); 0 = not eligible"""
    return int(_lib.load().emip_mlp_fc1dw_band_rows(B, H, W, K, N))


def mlp_fc1dw(x, w1, b1, colsum, ln_stats, eps, wd, bd):
    """GELU(dwconv3x3(LN(x) w1^T + b1) + bd) for images of <= 512 tokens (bf16): x [B,H,W,K] raw tokens with their row
    statistics ln_stats [B*H*W, 2], w1 [N,K] with the LayerNorm scale folded in, colsum [N] its row sums -> [B,H,W,N]"""
    B, H, W, K = x.shape
    M, _, ldx = rows(x)
    N = w1.shape[0]
    assert x.dtype == w1.dtype == torch.bfloat16 and w1.is_contiguous() and w1.shape[1] == K and wd.shape == (9, N)
    out = torch.empty((B, H, W, N), dtype=x.dtype, device=x.device)
    _lib.call("emip_mlp_fc1dw", _p(x), ldx, _p(w1), _p(b1), _p(colsum), _p(ln_stats), float(eps), _p(wd), _p(bd), _p(out), N,
              B, H, W, K, N, _stream())
    return out


def mlp_block_eligible(B, H, W, C, N):
    return bool(_lib.load().emip_mlp_block_eligible(B, H, W, C, N))


def mlp_block_consts(wd, bd, b1, colsum):
    """the per-chunk constant blocks of emip_mlp_block: f32 [N / 64][12][64] = 9 depthwise taps | bd | b1 | colsum(W1)
    from wd f32 [9, N] and the three f32 [N] vectors"""
    N = wd.shape[1]
    rows = torch.cat((wd.float(), bd.float().view(1, N), b1.float().view(1, N), colsum.float().view(1, N)), 0)     # [12, N]
    return rows.view(12, N // 64, 64).permute(1, 0, 2).contiguous()


def mlp_block(x, w1, w2, cst, b2, ln_stats, eps, out, out_stats=None):
    """out = x + fc2(GELU(dwconv3x3(LN(x) w1^T + b1) + bd)) + b2, one launch (bf16, C = 320, N = 1280); out must not alias x"""
    B, H, W, C = x.shape
    M, _, ldx = rows(x)
    _, _, ldo = rows(out)
    N = w1.shape[0]
    assert x.dtype == w1.dtype == w2.dtype == out.dtype == torch.bfloat16 and w1.is_contiguous() and w2.is_contiguous()
    assert w1.shape == (N, C) and w2.shape == (C, N) and cst.shape == (N // 64, 12, 64) and cst.is_contiguous()
    assert out.shape == x.shape and out.data_ptr() != x.data_ptr()
    _lib.call("emip_mlp_block", _p(x), ldx, _p(w1), _p(w2), _p(cst), _p(b2), _p(ln_stats), float(eps), _p(out), ldo,
              _p(out_stats), B, H, W, C, N, _stream())
    return out


def mlp_band_eligible(B, H, W, C, N):
    return bool(_lib.load().emip_mlp_band_eligible(B, H, W, C, N))


def mlp_band_packs(w1, b1, colsum, w2, wd, bd):
    """the weight stream of emip_mlp_band (mlp_band.hip).  w1 bf16 [1280, 320] (LayerNorm scale folded in), b1 / colsum f32
    [1280] (fc1 bias + W1 beta; row sums of the packed w1), w2 bf16 [320, 1280], wd f32 [9, 1280], bd f32 [1280] ->
    (stages uint8 [42, 41984], taps f32 [40, 10, 32]).  Stage t = [W1 chunk t | W2 chunk t - 2 | b1, colsum of chunk t]: the
    matrices in MFMA-fragment order -- one 1-KB piece is the A operand (32 rows x 16 k) of all 64 lanes, lane l holding row
    l & 31, k = 8 (l >> 5) .. + 7 -- so that a chunk is contiguous memory for the LDS-DMA ring and every fragment read is a
    ds_read_b128 at lane * 16."""
    N, C = w1.shape
    assert (N, C) == (1280, 320) and w2.shape == (C, N) and w1.dtype == w2.dtype == torch.bfloat16
    nch, dev = N // 32, w1.device
    # W1p[j, i, hh, row, e] = w1[32 j + row, 16 i + 8 hh + e]
    w1p = w1.view(nch, 32, C // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(nch, -1)             # [40, 10240] bf16
    # W2p[j, d, sp, hh, row, e] = w2[32 d + row, 32 j + 16 sp + 8 hh + e]
    w2p = w2.view(C // 32, 32, nch, 2, 2, 8).permute(2, 0, 3, 4, 1, 5).contiguous().view(nch, -1)         # [40, 10240] bf16
    nst = nch + 2
    sb = _lib.load().emip_mlp_band_stage_bytes() // nst
    st = torch.zeros((nst, sb), dtype=torch.uint8, device=dev)
    st[:nch, :20480] = w1p.view(torch.uint8).view(nch, 20480)
    st[2:, 20480:40960] = w2p.view(torch.uint8).view(nch, 20480)
    cst = torch.cat((b1.float().view(nch, 32), colsum.float().view(nch, 32)), 1).contiguous()            # [40, 64] f32
    st[:nch, 40960:40960 + 256] = cst.view(torch.uint8).view(nch, 256)
    taps = torch.cat((wd.float(), bd.float().view(1, N)), 0).view(10, nch, 32).permute(1, 0, 2).contiguous()
    return st, taps


MLP_BAND_BANDS = 0      # workgroups per image of emip_mlp_band: 4, 8, or 0 = by batch (the output bits do not depend on it)


def mlp_band(x, stages, taps, b2, ln_stats, eps, out, out_stats=None, bands=None):
    """out = x + fc2(GELU(dwconv3x3(LN(x) w1^T + b1) + bd)) + b2, one launch, a quarter or an eighth of an image per workgroup
    (bf16, 22 x 22 tokens, C = 320, N = 1280); out must not alias x"""
    B, H, W, C = x.shape
    M, _, ldx = rows(x)
    _, _, ldo = rows(out)
    assert x.dtype == out.dtype == torch.bfloat16 and stages.dtype == torch.uint8 and stages.is_contiguous()
    assert taps.shape == (40, 10, 32) and taps.is_contiguous() and taps.dtype == torch.float32
    assert out.shape == x.shape and out.data_ptr() != x.data_ptr()
    _lib.call("emip_mlp_band", _p(x), ldx, _p(stages), _p(taps), _p(b2), _p(ln_stats), float(eps), _p(out), ldo,
              _p(out_stats), B, H, W, C, 1280, MLP_BAND_BANDS if bands is None else bands, _stream())
    return out


def conv3x3_halo_eligible(B, H, W, Cin, Cout):
    return bool(_lib.load().emip_conv3x3_halo_eligible(B, H, W, Cin, Cout))


def conv3x3_halo_pack(w):
    """pack_conv's [C, 3 * 3 * C] bf16 (C = 64 / 96 / 128) -> the MFMA-fragment order of emip_conv3x3_halo:
    [tap][d][ks][lane][e] = w[32 d + lane % 32][tap][16 ks + 8 (lane // 32) + e]"""
    C = w.shape[0]
    assert w.dtype == torch.bfloat16 and w.shape == (C, 9 * C) and C in (64, 96, 128)
    v = w.view(C // 32, 32, 9, C // 16, 2, 8)         # [d][row][tap][ks][half][e]
    return v.permute(2, 0, 3, 4, 1, 5).contiguous().view(-1)     # [tap][d][ks][half][row][e]: lane = 32 half + row


def conv3x3_halo_ws_bytes(B, H, W, C=64):
    return int(_lib.load().emip_conv3x3_halo_ws_bytes(B, H, W, C))


def conv3x3_halo_ws(B, H, W, device, C=64):
    """zeroed statistics workspace of emip_conv3x3_halo (allocate OUTSIDE the timed path: launches leave the tickets zero)"""
    return torch.zeros(conv3x3_halo_ws_bytes(B, H, W, C), dtype=torch.uint8, device=device)


def conv3x3_halo(x, wp, out=None, in_sums=None, in_eps=1e-5, out_sums=None, ws=None):
    """y = conv3x3(relu(instance_norm(x)) if in_sums is given else x), stride 1, pad 1, C -> C, bf16 channels-last;
    out_sums f64 [B, C, 2] receives (sum, sum of squares) of y per image and channel (needs ws)"""
    B, H, W, C = x.shape
    assert x.dtype == torch.bfloat16 and x.stride(-1) == 1 and x.stride(1) == W * x.stride(2) and (B == 1 or x.stride(0) == H * x.stride(1))
    if out is None:
        out = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    for t in (in_sums, out_sums):
        assert t is None or (t.dtype == torch.float64 and t.is_contiguous() and t.numel() == B * C * 2)
    _lib.call("emip_conv3x3_halo", _p(x), x.stride(2), _p(wp), _p(out), out.stride(2), B, H, W, C, C, _p(in_sums), float(in_eps),
              _p(out_sums), _p(ws), _nbytes(ws), _stream())
    return out


def conv_stem_eligible(B, H, W, Cin, Cout):
    return bool(_lib.load().emip_conv_stem_eligible(B, H, W, Cin, Cout))


def conv_stem_pack(w):
    """pack_conv's [64, 7 * 7 * 8] bf16 -> the fragment order of emip_conv_stem: [d][ks][lane][e] = w[32 d + lane % 32][tap = 2 ks +
    lane // 32][e], zero for tap 49"""
    assert w.dtype == torch.bfloat16 and w.shape == (64, 392)
    wz = torch.zeros((64, 50, 8), dtype=w.dtype, device=w.device)
    wz[:, :49] = w.view(64, 49, 8)
    v = wz.view(2, 32, 25, 2, 8)                      # [d][row][ks][half][e]
    return v.permute(0, 2, 3, 1, 4).contiguous().view(-1)        # [d][ks][half][row][e]: lane = 32 half + row


def conv_stem(x, wp, out_sums=None, ws=None, out=None):
    """y [B, H / 2, W / 2, 64] = conv7x7(x [B, H, W, 8], stride 2, pad 3), bf16; out_sums f64 [B, 64, 2] of y (needs ws)"""
    B, H, W, C = x.shape
    assert x.dtype == torch.bfloat16 and C == 8 and x.stride(-1) == 1 and x.stride(1) == W * x.stride(2)
    if out is None:
        out = torch.empty((B, H // 2, W // 2, 64), dtype=x.dtype, device=x.device)
    assert out_sums is None or (out_sums.dtype == torch.float64 and out_sums.is_contiguous() and out_sums.numel() == B * 128)
    _lib.call("emip_conv_stem", _p(x), x.stride(2), _p(wp), _p(out), out.stride(2), B, H, W, 8, 64, _p(out_sums), _p(ws), _nbytes(ws),
              _stream())
    return out


def dwconv3x3_dual(x, wt, bias, act):
    """-> (act(dwconv(x)), dwconv(x)): activation output and pre-activation values from one pass"""
    B, H, W, C = x.shape
    M, _, ldx = rows(x)
    y = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    z = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    _lib.call("emip_dwconv3x3_dual", _p(x), ldx, _p(y), C, _p(z), C, _p(wt), _p(bias), B, H, W, C, act,
              dt_code(x.dtype), _stream())
    return y, z


def dwconv3x3_gated(x, wt, cout_pad, bias=None):
    B, H, W, C2 = x.shape
    _, _, ldx = rows(x)
    out = torch.empty((B, H, W, cout_pad), dtype=x.dtype, device=x.device)
    _lib.call("emip_dwconv3x3_gated", _p(x), ldx, _p(out), cout_pad, _p(wt), _p(bias), B, H, W, C2, cout_pad,
              dt_code(x.dtype), _stream())
    return out


def chan_stats(x, groups, sums=None):
    """sums f64 [groups, C, 2]; pass `sums` when the producing conv2d(..., zero=sums) already cleared it"""
    M, C, ldx = rows(x)
    assert M % groups == 0
    pre = sums is not None
    if sums is None:
        sums = torch.empty((groups, C, 2), dtype=torch.float64, device=x.device)
    assert sums.shape == (groups, C, 2) and sums.dtype == torch.float64 and sums.is_contiguous()
    _lib.call("emip_chan_stats", _p(x), ldx, _p(sums), groups, M // groups, C, int(pre), dt_code(x.dtype), _stream())
    return sums


def chan_norm_apply(x, sums, groups, eps, relu_inner, relu_outer=False, res=None, gamma=None, beta=None, out=None, res_sums=None,
                    res_relu=False):
    """res_sums: the residual is a raw conv output, normalised on the fly from its own sums (+ ReLU with res_relu)"""
    M, C, ldx = rows(x)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _, _, ldy = rows(out)
    ldr = rows(res)[2] if res is not None else 0
    if res_sums is not None:
        assert res is not None and res_sums.dtype == torch.float64 and res_sums.is_contiguous() and res_sums.numel() == groups * C * 2
        _lib.call("emip_chan_norm_apply_res", _p(x), ldx, _p(out), ldy, _p(res), ldr, _p(sums), _p(gamma), _p(beta), groups,
                  M // groups, C, float(eps), int(relu_inner), int(relu_outer), _p(res_sums), int(res_relu), dt_code(x.dtype), _stream())
        return out
    _lib.call("emip_chan_norm_apply", _p(x), ldx, _p(out), ldy, _p(res), ldr, _p(sums), _p(gamma), _p(beta), groups,
              M // groups, C, float(eps), int(relu_inner), int(relu_outer), dt_code(x.dtype), _stream())
    return out


def bilinear(x, Ho, Wo, align_corners, mul=1.0, out=None):
    B, H, W, C = x.shape
    _, _, ldx = rows(x)
    if out is None:
        out = torch.empty((B, Ho, Wo, C), dtype=x.dtype, device=x.device)
    _, _, ldy = rows(out)
    _lib.call("emip_bilinear", _p(x), ldx, _p(out), ldy, B, H, W, C, Ho, Wo, int(align_corners), float(mul),
              dt_code(x.dtype), _stream())
    return out


def bilinear_planar(x, xc, C, Ho, Wo, align_corners, mul=1.0):
    B, H, W, _ = x.shape
    _, _, ldx = rows(x)
    out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
    _lib.call("emip_bilinear_planar", _p(x), ldx, xc, _p(out), B, H, W, C, Ho, Wo, int(align_corners), float(mul),
              dt_code(x.dtype), _stream())
    return out


def eltwise(a, b, mode, c3=None, period=0, out=None):
    M, C, lda = rows(a)
    _, _, ldb = rows(b)
    ldc3 = rows(c3)[2] if c3 is not None else 0
    if out is None:
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    _, _, ldy = rows(out)
    _lib.call("emip_eltwise", _p(a), lda, _p(b), ldb, _p(c3), ldc3, _p(out), ldy, M, C, mode, period,
              dt_code(a.dtype), _stream())
    return out


def planar_to_cl(x, dtype, cpad=None, out=None):
    """planar f32 [B,C,H,W] -> channels-last [B,H,W,cpad] (out: a contiguous [B,H,W,cpad] view to fill, e.g. half of a batch)"""
    _dev(x)
    B, C, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    cpad = cpad or C
    if out is None:
        out = torch.empty((B, H, W, cpad), dtype=dtype, device=x.device)
    assert out.shape == (B, H, W, cpad) and out.dtype == dtype and out.is_contiguous()
    _lib.call("emip_planar_to_cl", _p(x), _p(out), cpad, B, C, H * W, cpad, dt_code(dtype), _stream())
    return out


def cl_to_planar(x, xc=0, C=None):
    """channels-last [B,H,W,*] (channels xc..xc+C) -> planar f32 [B,C,H,W]"""
    B, H, W, Ct = x.shape
    C = C or Ct
    _, _, ldx = rows(x)
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    _lib.call("emip_cl_to_planar", _p(x), ldx, xc, _p(out), B, C, H * W, dt_code(x.dtype), _stream())
    return out


def copy_cols(x, xc, C, out, yc, cpad=None):
    M, _, ldx = rows(x)
    Mo, _, ldy = rows(out)
    assert M == Mo
    _lib.call("emip_copy_cols", _p(x), ldx, xc, dt_code(x.dtype), _p(out), ldy, yc, dt_code(out.dtype), M, C,
              cpad or C, _stream())
    return out


def convex_upsample(logits, flow):
    N, H, W, _ = logits.shape
    assert flow.dtype == torch.float32 and flow.is_contiguous() and flow.shape == (N, H, W, 2)
    out = torch.empty((N, 2, 8 * H, 8 * W), dtype=torch.float32, device=logits.device)
    _lib.call("emip_convex_upsample", _p(logits), rows(logits)[2], _p(flow), _p(out), N, H, W, dt_code(logits.dtype),
              _stream())
    return out


def corresp_to_flow(o, N, H, W, sub_grid):
    assert o.dtype == torch.float32
    flow = torch.empty((N, H, W, 2), dtype=torch.float32, device=o.device)
    _lib.call("emip_corresp_to_flow", _p(o), o.stride(-2), _p(flow), N, H, W, int(sub_grid), _stream())
    return flow


def flow_warp(x, flow):
    _dev(x)
    B, C, H, W = x.shape
    assert x.dtype == torch.float32 and flow.dtype == torch.float32 and x.is_contiguous() and flow.is_contiguous()
    out = torch.empty_like(x)
    _lib.call("emip_flow_warp", _p(x), _p(flow), _p(out), B, C, H, W, _stream())
    return out


def occ_corners(flow):
    _dev(flow)
    B, _, H, W = flow.shape
    assert flow.dtype == torch.float32 and flow.is_contiguous()
    idx = torch.empty((B, 4 * H * W), dtype=torch.int64, device=flow.device)
    wts = torch.empty((B, 4 * H * W), dtype=torch.float32, device=flow.device)
    _lib.call("emip_occ_corners", _p(flow), _p(idx), _p(wts), B, H, W, _stream())
    return idx, wts


def occ_mask_backward(flow, th=0.2, complement=False):
    _dev(flow)
    B, _, H, W = flow.shape
    assert flow.dtype == torch.float32 and flow.is_contiguous()
    ws = torch.empty((B, H * W), dtype=torch.float32, device=flow.device)
    occ = torch.empty((B, 1, H, W), dtype=torch.float32, device=flow.device)
    _lib.call("emip_occ_mask_backward", _p(flow), _p(ws), _p(occ), B, H, W, float(th), int(complement), _stream())
    return occ


def hybrid_e_loss(pred, mask):
    """pred (logits), mask: planar f32 [B,1,H,W] -> scalar tensor f32 [1]"""
    _dev(pred)
    B, _, H, W = pred.shape
    assert pred.dtype == torch.float32 and mask.dtype == torch.float32 and pred.is_contiguous() and mask.is_contiguous()
    ws = torch.empty((B, 8), dtype=torch.float64, device=pred.device)
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    _lib.call("emip_hybrid_e_loss", _p(pred), _p(mask), _p(ws), _p(out), B, H, W, _stream())
    return out


def photometric_loss(im, rec, mask, out, weight=1.0, accumulate=False):
    _dev(im)
    B, C, H, W = im.shape
    for t in (im, rec, mask):
        assert t.dtype == torch.float32 and t.is_contiguous()
    ws = torch.empty(4, dtype=torch.float64, device=im.device)
    _lib.call("emip_photometric_loss", _p(im), _p(rec), _p(mask), _p(ws), _p(out), B, C, H, W, float(weight),
              int(accumulate), _stream())
    return out


def gemm_tn(a, b, with_colsum=False, defer=False, owners=(None, None)):
    """c[n, k] = sum_m a[m, n] * b[m, k]  (weight gradient: a = dY, b = X) -> f32 [N, K]
    with_colsum: also return sum_m a[m, n] (the bias gradient) from the same launch
    defer: the caller hands the returned tensors to autograd AS THEY ARE (no slice, copy or arithmetic on them), so the
    launch may wait for flush_wgrads(); owners = (weight parameter, bias parameter) whose .grad they become (WgradQueue.fixup)"""
    M, N, lda = rows(a)
    Mb, K, ldb = rows(b)
    assert M == Mb and a.dtype == b.dtype
    if ARENA.active:                   # training step: accumulate into slices of the step's pre-cleared gradient arena
        c = grad_zeros((N, K), a.device)
        db = grad_zeros((N,), a.device) if with_colsum else None
        if (defer and WGRADS.enabled and a.dtype == torch.bfloat16 and _lib.load().emip_gemm_tn8_eligible(M, N, K, lda, ldb)
                and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
            # launched with the other layers' by flush_wgrads()
            WGRADS.add(a, b, c, db, M, N, K, lda, ldb, owners=((c, owners[0]), (db, owners[1])))
        else:
            _lib.call("emip_gemm_tn_into", _p(a), _p(b), _p(c), _p(db), M, N, K, lda, ldb, K, dt_code(a.dtype), _stream())
        return (c, db) if with_colsum else c
    if not with_colsum:
        c = torch.empty((N, K), dtype=torch.float32, device=a.device)
        _lib.call("emip_gemm_tn", _p(a), _p(b), _p(c), M, N, K, lda, ldb, K, 1, 0, 0, 0, dt_code(a.dtype), _stream())
        return c
    buf = torch.empty(N * K + N, dtype=torch.float32, device=a.device)     # db right behind c: one clearing launch for both
    c, db = buf[:N * K].view(N, K), buf[N * K:]
    _lib.call("emip_gemm_tn_bias", _p(a), _p(b), _p(c), _p(db), M, N, K, lda, ldb, K, 1, 0, 0, 0, dt_code(a.dtype),
              _stream())
    return c, db


def gemm_tn_batched(a, b, batch, M, N, K, lda, ldb, bsA, bsB):
    """per batch z: c[z][n, k] = sum_m a[z][m, n] * b[z][m, k] -> f32 [batch, N, K]"""
    _dev(a)
    c = torch.empty((batch, N, K), dtype=torch.float32, device=a.device)
    _lib.call("emip_gemm_tn", _p(a), _p(b), _p(c), M, N, K, lda, ldb, K, batch, bsA, bsB, N * K, dt_code(a.dtype),
              _stream())
    return c


def conv2d_wgrad(dy, x, kh, kw, stride, pad, defer_to=None, want_db=False, bias_param=None):
    """dy [B,Ho,Wo,Cout], x [B,H,W,Cin] channels-last -> dW f32 [Cout, kh*kw*Cin] (packed like the forward weights).
    defer_to = (parameter, unpack): the caller wants the gradient in the parameter's layout, unpack(dW) -> that layout.  Inside a
    training step the contraction may then wait for flush_wgrads() (grouped launch); the return value is (None, zeros of the
    parameter's shape from the arena, db) and WgradQueue.fixup() adds unpack(dW) to the parameter's .grad; with want_db the
    bias gradient (column sums of dy) comes out of the same launch into an arena slice the caller hands to autograd as it is.
    Otherwise (dW, None, None): the caller computes the bias gradient itself."""
    _dev(x)
    B, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    _, _, ldx = rows(x)
    _, _, lddy = rows(dy)
    if ARENA.active:
        dw = grad_zeros((Cout, kh * kw * Cin), x.device)
        if (defer_to is not None and defer_to[0] is not None and WGRADS.enabled and WIDE_WGRAD and x.dtype == torch.bfloat16
                and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
                and _lib.load().emip_conv_wgrad16_eligible(B, H, W, Cin, ldx, Cout, lddy, kh, kw, stride, pad)):
            param, unpack = defer_to
            g = grad_zeros(tuple(param.shape), x.device)
            db = grad_zeros((Cout,), x.device) if want_db else None
            WGRADS.add_conv(dy, x, dw, db, (B, H, W, Cin, ldx, Cout, lddy, kh, kw, stride, pad), param, unpack, bias_param)
            return None, g, db
        _lib.call("emip_conv2d_wgrad_into", _p(dy), _p(x), _p(dw), B, H, W, Cin, ldx, Cout, lddy, kh, kw, stride, pad,
                  dt_code(x.dtype), _stream())
        return dw, None, None
    dw = torch.empty((Cout, kh * kw * Cin), dtype=torch.float32, device=x.device)
    _lib.call("emip_conv2d_wgrad", _p(dy), _p(x), _p(dw), B, H, W, Cin, ldx, Cout, lddy, kh, kw, stride, pad,
              dt_code(x.dtype), _stream())
    return dw, None, None


def layernorm_bwd(x, dy, gamma, eps, dgamma, dbeta):
    """dx; dgamma / dbeta (f32 [C]) are accumulated into"""
    M, C, ldx = rows(x)
    _, _, lddy = rows(dy)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _lib.call("emip_layernorm_bwd", _p(x), ldx, _p(dy), lddy, _p(dx), C, _p(gamma), _p(dgamma), _p(dbeta), 1, 0, M, C,
              float(eps), dt_code(x.dtype), _stream())
    return dx


def layernorm_bwd_fresh(x, dy, gamma, eps, dres=None):
    """dx (+ dres: the gradient arriving over the skip path of a pre-norm residual block, added in the same launch), dgamma,
    dbeta (fresh f32 [C] tensors).  The kernel can spread the reduction over `nparts` partial accumulators; measured on
    MI355X that does not pay (57 us plain vs 71 us + a fill and a column-sum launch at 30976 x 320), so P = 1."""
    M, C, ldx = rows(x)
    _, _, lddy = rows(dy)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    lddr = 0
    if dres is not None:
        Mr, Cr, lddr = rows(dres)
        assert (Mr, Cr) == (M, C) and dres.dtype == x.dtype
    acc = grad_zeros((1, 2, C), x.device)
    _lib.call("emip_layernorm_bwd_res", _p(x), ldx, _p(dy), lddy, _p(dx), C, _p(dres), lddr, _p(gamma), _p(acc),
              _p(acc[0, 1]), 1, 2 * C, M, C, float(eps), dt_code(x.dtype), _stream())
    return dx, acc[0, 0], acc[0, 1]


# ---- backward building blocks ---------------------------------------------------------------------------------------
def softmax_rows(x, L, scale=1.0, gid_q=None, gid_k=None, period=0, nwin=0, out=None):
    M, _, ld = rows(x)
    if out is None:
        out = torch.empty_like(x)
    _lib.call("emip_softmax_rows", _p(x), _p(out), M, L, ld, float(scale), _p(gid_q), _p(gid_k), period, nwin,
              dt_code(x.dtype), _stream())
    return out


def softmax_bwd_rows(p, dp, L, scale=1.0, out=None):
    M, _, ld = rows(p)
    if out is None:
        out = torch.empty_like(p)
    _lib.call("emip_softmax_bwd_rows", _p(p), _p(dp), _p(out), M, L, ld, float(scale), dt_code(p.dtype), _stream())
    return out


def transpose_pad(x, rpad):
    """x [Z, R, C] (row stride = x.stride(1)) -> [Z, C, rpad] with zero padding of the R axis"""
    _dev(x)
    Z, R, C = x.shape
    y = torch.empty((Z, C, rpad), dtype=x.dtype, device=x.device)
    _lib.call("emip_transpose_pad", _p(x), x.stride(1), x.stride(0), _p(y), C * rpad, Z, R, C, rpad, dt_code(x.dtype),
              _stream())
    return y


def gelu_bwd(z, dy):
    M, C, ldz = rows(z)
    _, _, lddy = rows(dy)
    dz = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    _lib.call("emip_gelu_bwd", _p(z), ldz, _p(dy), lddy, _p(dz), C, M, C, dt_code(z.dtype), _stream())
    return dz


def dwconv3x3_wgrad(x, dy, dw, db=None):
    B, H, W, C = x.shape
    _, _, ldx = rows(x)
    _, _, lddy = rows(dy)
    _lib.call("emip_dwconv3x3_wgrad", _p(x), ldx, _p(dy), lddy, _p(dw), _p(db), B, H, W, C, dt_code(x.dtype), _stream())
    return dw


def dwconv3x3_bwd_fused(x, z, dy, wt, dw, db=None, gelu=True):
    """depthwise 3x3 (+ GELU) backward in one pass (bf16): -> dx; dw f32 [C, 9] (the parameter's order) and db f32 [C] are
    accumulated into.  x: the conv's input, z: its pre-activation output (gelu only), dy: gradient of the block's output"""
    B, H, W, C = x.shape
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
    _, _, ldx = rows(x)
    _, _, lddy = rows(dy)
    ldz = rows(z)[2] if gelu else 0
    dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    _lib.call("emip_dwconv3x3_bwd_fused", _p(x), ldx, _p(z) if gelu else None, ldz, _p(dy), lddy, _p(dx), C, _p(wt), _p(dw),
              _p(db), B, H, W, C, 1 if gelu else 0, _stream())
    return dx


def bn_train_bwd(x, dy, out, fsums, gamma, dgamma, dbeta, eps):
    M, C, ldx = rows(x)
    _, _, lddy = rows(dy)
    ldo = rows(out)[2] if out is not None else 0
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    ws = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    _lib.call("emip_bn_train_bwd", _p(x), ldx, _p(dy), lddy, _p(out), ldo, _p(dx), C, _p(fsums), _p(gamma), _p(dgamma),
              _p(dbeta), _p(ws), M, C, float(eps), dt_code(x.dtype), _stream())
    return dx


def bilinear_bwd(dy, H, W, align_corners, mul=1.0):
    B, Ho, Wo, C = dy.shape
    _, _, lddy = rows(dy)
    dx = torch.zeros((B, H, W, C), dtype=torch.float32, device=dy.device)
    _lib.call("emip_bilinear_bwd", _p(dy), lddy, _p(dx), B, H, W, C, Ho, Wo, int(align_corners), float(mul),
              dt_code(dy.dtype), _stream())
    return dx


def bilinear_planar_bwd(dy, H, W, align_corners, mul=1.0):
    """dy planar f32 [B,C,Ho,Wo] -> f32 channels-last [B,H,W,C]"""
    _dev(dy)
    B, C, Ho, Wo = dy.shape
    dx = torch.zeros((B, H, W, C), dtype=torch.float32, device=dy.device)
    _lib.call("emip_bilinear_planar_bwd", _p(dy.contiguous()), _p(dx), C, 0, B, H, W, C, Ho, Wo, int(align_corners),
              float(mul), _stream())
    return dx


def zero_insert(dy, H, W, stride):
    B, Ho, Wo, C = dy.shape
    _, _, lddy = rows(dy)
    z = torch.empty((B, H, W, C), dtype=dy.dtype, device=dy.device)
    _lib.call("emip_zero_insert", _p(dy), lddy, _p(z), B, Ho, Wo, H, W, C, stride, dt_code(dy.dtype), _stream())
    return z


def depatchify(p, B, Ho, Wo, k, C):
    assert p.is_contiguous()
    dx = torch.empty((B, Ho * k, Wo * k, C), dtype=p.dtype, device=p.device)
    _lib.call("emip_depatchify", _p(p), _p(dx), B, Ho, Wo, k, C, dt_code(p.dtype), _stream())
    return dx


def colsum(x):
    """sum over rows of a channels-last tensor -> f32 [C]"""
    M, C, ldx = rows(x)
    out = grad_zeros((C,), x.device)
    _lib.call("emip_colsum", _p(x), ldx, _p(out), M, C, dt_code(x.dtype), _stream())
    return out


def bn_running_update(sums, running_mean, running_var, tracked, n, momentum):
    """train-mode BatchNorm2d bookkeeping from the f64 sums [1, C, 2] of chan_stats (one group of n values per channel)"""
    C = running_mean.numel()
    assert sums.dtype == torch.float64 and sums.numel() == 2 * C and running_mean.dtype == torch.float32
    assert tracked is None or tracked.dtype == torch.int64
    _lib.call("emip_bn_running_update", _p(sums), _p(running_mean), _p(running_var), _p(tracked), int(n), float(momentum), C,
              _stream())


def gate_fwd(z, ch, cpad):
    M, _, ldz = rows(z)
    y = torch.empty(z.shape[:-1] + (cpad,), dtype=z.dtype, device=z.device)
    _lib.call("emip_gate_fwd", _p(z), ldz, _p(y), cpad, M, ch, cpad, dt_code(z.dtype), _stream())
    return y


def gate_bwd(z, dy, ch):
    M, _, ldz = rows(z)
    _, _, lddy = rows(dy)
    dz = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    _lib.call("emip_gate_bwd", _p(z), ldz, _p(dy), lddy, _p(dz), z.shape[-1], M, ch, dt_code(z.dtype), _stream())
    return dz


def colscale_add(a, b, s, lds, rows_per_group, out):
    """out[r, c] = a[r, c] + s[r // rows_per_group, c] * b[r, c]"""
    M, C, lda = rows(a)
    _, _, ldb = rows(b)
    _, _, ldy = rows(out)
    _lib.call("emip_colscale_add", _p(a), lda, _p(b), ldb, _p(s), lds, _p(out), ldy, M, C, rows_per_group,
              dt_code(a.dtype), _stream())
    return out


def mdta_attn_ws(q, kv_k, temperature, B, heads, P):
    """like mdta_attn, also returning the f32 workspace [G | nq^2 | nk^2] the backward needs"""
    _dev(q)
    ws = torch.empty(_lib.load().emip_mdta_ws_floats(B, heads), dtype=torch.float32, device=q.device)
    attn = torch.empty((B, heads, 64, 64), dtype=q.dtype, device=q.device)
    _lib.call("emip_mdta_attn", _p(q), q.stride(-2), q.stride(0), _p(kv_k), kv_k.stride(-2), kv_k.stride(0),
              _p(temperature), _p(ws), _p(attn), B, heads, P, dt_code(q.dtype), _stream())
    return ws[:B * heads * (4096 + 128)], attn


def mdta_bwd_small(ws, temperature, attn, dA, B, heads):
    nbh = B * heads
    G, nq2, nk2 = ws[:nbh * 4096], ws[nbh * 4096:nbh * 4096 + nbh * 64], ws[nbh * 4096 + nbh * 64:]
    dG = torch.empty_like(attn)
    dGT = torch.empty_like(attn)
    sq = torch.empty((B, heads, 64), dtype=torch.float32, device=attn.device)
    sk = torch.empty((B, heads, 64), dtype=torch.float32, device=attn.device)
    dtau = torch.zeros(heads, dtype=torch.float32, device=attn.device)
    _lib.call("emip_mdta_bwd_small", _p(G), _p(nq2), _p(nk2), _p(temperature), _p(attn), _p(dA), _p(dG), _p(dGT),
              _p(sq), _p(sk), _p(dtau), B, heads, dt_code(attn.dtype), _stream())
    return dG, dGT, sq, sk, dtau


def window_rows(src, table, B, nwin, L, Lp, n, C, scatter=False):
    """gather [B, n, C] -> [B*nwin, Lp, C] (zero pad rows) through table [nwin, L]; scatter: the inverse -> [B, n, C]"""
    _dev(src)
    assert src.is_contiguous()
    if not scatter:
        dst = (torch.zeros if Lp > L else torch.empty)((B * nwin, Lp, C), dtype=src.dtype, device=src.device)
    else:
        dst = torch.empty((B, n, C), dtype=src.dtype, device=src.device)
    _lib.call("emip_window_rows", _p(src), _p(dst), _p(table), B, nwin, L, Lp, n, C, C, C, int(scatter),
              dt_code(src.dtype), _stream())
    return dst


def axpby(a, b, alpha, beta, out=None):
    M, C, lda = rows(a)
    _, _, ldb = rows(b)
    if out is None:
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    _lib.call("emip_axpby", _p(a), lda, _p(b), ldb, _p(out), rows(out)[2], M, C, float(alpha), float(beta),
              dt_code(a.dtype), _stream())
    return out


def act_fwd(x, act):
    M, C, ldx = rows(x)
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _lib.call("emip_act_fwd", _p(x), ldx, _p(y), C, M, C, act, dt_code(x.dtype), _stream())
    return y


def relu_bwd(y, dy):
    M, C, ldy = rows(y)
    _, _, lddy = rows(dy)
    dx = torch.empty(y.shape, dtype=y.dtype, device=y.device)
    _lib.call("emip_relu_bwd", _p(y), ldy, _p(dy), lddy, _p(dx), C, M, C, dt_code(y.dtype), _stream())
    return dx


def convex_upsample_bwd(logits, flow, dY):
    """logits [N,H,W,576+], flow f32 [N,H,W,2], dY planar f32 [N,2,8H,8W] -> (dlogits like logits[..., :576], dflow)"""
    N, H, W = flow.shape[:3]
    _, _, ldl = rows(logits)
    dl = torch.empty((N, H, W, 576), dtype=logits.dtype, device=logits.device)
    df = torch.empty_like(flow)
    _lib.call("emip_convex_upsample_bwd", _p(logits), ldl, _p(flow), _p(dY), _p(dl), 576, _p(df), N, H, W,
              dt_code(logits.dtype), _stream())
    return dl, df


def hybrid_e_loss_fwd_ws(pred, mask):
    B, _, H, W = pred.shape
    ws = torch.empty((B, 8), dtype=torch.float64, device=pred.device)
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    _lib.call("emip_hybrid_e_loss", _p(pred), _p(mask), _p(ws), _p(out), B, H, W, _stream())
    return out, ws


def hybrid_e_loss_bwd(pred, mask, ws, gout):
    B, _, H, W = pred.shape
    dp = torch.empty_like(pred)
    _lib.call("emip_hybrid_e_loss_bwd", _p(pred), _p(mask), _p(ws), _p(gout), _p(dp), B, H, W, _stream())
    return dp


def photometric_loss_ws(im, rec, mask, out, weight, accumulate):
    B, C, H, W = im.shape
    ws = torch.empty(4, dtype=torch.float64, device=im.device)
    _lib.call("emip_photometric_loss", _p(im), _p(rec), _p(mask), _p(ws), _p(out), B, C, H, W, float(weight),
              int(accumulate), _stream())
    return ws


def photometric_loss_bwd(im, rec, mask, ws, gout, weight, drec=None):
    B, C, H, W = im.shape
    abc = torch.empty((B * C * H * W * 3,), dtype=torch.float32, device=im.device)
    acc = drec is not None
    if drec is None:
        drec = torch.empty_like(rec)
    _lib.call("emip_photometric_loss_bwd", _p(im), _p(rec), _p(mask), _p(ws), _p(abc), _p(gout), _p(drec), B, C, H, W,
              float(weight), int(acc), _stream())
    return drec


def flow_warp_bwd(x, flow, dy):
    B, C, H, W = x.shape
    df = torch.empty_like(flow)
    _lib.call("emip_flow_warp_bwd", _p(x), _p(flow), _p(dy), _p(df), B, C, H, W, _stream())
    return df


def gemm_heads(a, w, out, batch, heads, M, N, K, lda, ldw, ldc, bsA, hsA, bsW, hsW, bsC, hsC):
    """per (b, h): out = a w^T on raw views; operand of (b, h) at b * bs + h * hs (batch = B * heads)"""
    _dev(a)
    _lib.call("emip_gemm_heads", _p(a), _p(w), _p(out), M, N, K, lda, ldw, ldc, batch, heads, bsA, hsA, bsW, hsW, bsC, hsC,
              dt_code(a.dtype), _stream())
    return out


def gemm_tn_heads(a, b, out, batch, heads, M, N, K, lda, ldb, ldc, bsA, hsA, bsB, hsB, bsC, hsC):
    """per (b, h): out[n, k] += sum_m a[m, n] b[m, k] into the PRE-ZEROED f32 `out` (may be a column slice, ldc > K)"""
    _dev(a)
    assert out.dtype == torch.float32
    _lib.call("emip_gemm_tn_heads", _p(a), _p(b), _p(out), M, N, K, lda, ldb, ldc, batch, heads, bsA, hsA, bsB, hsB, bsC,
              hsC, dt_code(a.dtype), _stream())
    return out


def transpose_pad_heads(x, batch, heads, R, C, Rpad, ldx, bsx, hsx):
    """slices (b, h) of x (R rows of C columns at b * bsx + h * hsx) -> [batch, C, Rpad], zero padded"""
    _dev(x)
    y = torch.empty((batch, C, Rpad), dtype=x.dtype, device=x.device)
    _lib.call("emip_transpose_pad_heads", _p(x), ldx, bsx, hsx, _p(y), batch, heads, R, C, Rpad, dt_code(x.dtype),
              _stream())
    return y
