"""Host-side plumbing shared by the nn.Module surfaces.

The modules in emip_amd.model / emip_amd.lib keep the reference's constructor
arguments, attribute names and state_dict keys, but hold
parameters only: their forward passes run hand-written HIP kernels through
emip_amd.ops on channels-last activations.  This file provides

  * the compute-dtype switch (float32 = parity mode, bfloat16 = performance mode),
  * a per-module cache of kernel-ready ("packed") weights, invalidated by the
    parameters' version counters (optimizer steps and load_state_dict bump them),
  * packing helpers (conv weights to [Cout][KH][KW][Cin], BatchNorm folding, ...).
"""
import weakref

import torch
import torch.nn as nn

_DEFAULT_DTYPE = [torch.float32]


def set_default_dtype(dtype):
    """Compute/storage dtype of activations for modules created or run afterwards."""
    assert dtype in (torch.float32, torch.bfloat16)
    _DEFAULT_DTYPE[0] = dtype


def get_default_dtype():
    return _DEFAULT_DTYPE[0]


# ---------------------------------------------------------------------------------------------
# Packed-weight cache with in-place refresh.
#
# A pack is rebuilt by its builder (torch ops, a handful of tiny launches) whenever a source parameter's version changes.
# In a training loop that is every step for every trainable weight: ~1 300 launches, 7 ms of a 149 ms EMIP-short step.
# The packing helpers below therefore RECORD what they built as a generalised permutation of the f32 master
# (dst dims, source strides, base, channel padding); an entry whose tensor outputs are all either recorded packs or the
# parameters themselves is `refreshable`, and refresh_packs() -- called by FusedClampAdamW.step() -- rewrites all of them
# in place with ONE emip_repack launch and moves the entries' signatures to the new versions.  Anything else (folded
# norms, permuted channel orders, other optimizers) keeps the rebuild-on-version-change path.

_REC = None            # list while a builder runs under packed(): (src, dst, dims4, strides4, base, valid3)
_REFRESHABLE = {}      # (id(module), cache key) -> weakref(module): the packs die with their module (a strong reference
                       # here kept every discarded model's bf16 / f32 packs in GPU memory for the life of the process)
_TABLE = {}            # device -> (key, recs tensor, blockmap tensor, nblocks)
# Packs of TRAINABLE parameters that are not plain permutations (the mask head's weight padded to 8 output rows, ...): the
# eager forward rebuilds them on first use after an optimizer step -- into NEW tensors.  A captured training step
# (train.GraphedTrainStep) has no Python forward and holds the OLD addresses: it would train against a frozen copy.
# refresh_packs() therefore rebuilds these through their builder and copies the result INTO the existing tensors.
_REBUILD = {}          # (id(module), cache key) -> weakref(module)


class _Entry:
    __slots__ = ("sig", "val", "recs", "tensors", "mode", "builder")

    def __init__(self, sig, val, recs, tensors, mode, builder=None):
        self.sig, self.val, self.recs, self.tensors, self.mode, self.builder = sig, val, recs, tensors, mode, builder


def _record(src, dst, dims, strides, base=0, valid3=None, scale=1.0):
    """dst (contiguous) = scale * src.flatten()[base + sum_i idx_i * strides_i] over `dims` (4 entries, leading ones padded
    with 1), zero where idx_3 >= valid3"""
    if _REC is None or src.dtype != torch.float32 or not src.is_contiguous():
        return
    if dst.untyped_storage().data_ptr() == src.untyped_storage().data_ptr():
        return                                   # f32 mode: the "pack" is the parameter itself
    assert dst.is_contiguous() and dst.numel() >= dims[0] * dims[1] * dims[2] * dims[3]      # a prefix may be enough
    _REC.append((src, dst, tuple(int(d) for d in dims), tuple(int(x) for x in strides), int(base),
                 int(dims[3] if valid3 is None else valid3), float(scale)))


def _flat_tensors(v, out):
    """tensors inside a builder's return value (nested tuples / lists / dicts of tensors and plain scalars); raises
    TypeError on anything else, which the caller reads as `not refreshable`"""
    if torch.is_tensor(v):
        out.append(v)
    elif isinstance(v, (tuple, list)):
        for x in v:
            _flat_tensors(x, out)
    elif isinstance(v, dict):
        for x in v.values():
            _flat_tensors(x, out)
    elif not (v is None or isinstance(v, (int, float, bool, str))):
        raise TypeError(type(v))
    return out


def refresh_packs():
    """Rewrite every refreshable pack of a parameter that requires grad from its f32 master: one launch per device."""
    import struct

    from . import _lib
    live = []
    for rk, ref in list(_REFRESHABLE.items()):
        m = ref()
        e = m._pack_cache.get(rk[1]) if m is not None else None
        if e is None or e.recs is None:
            del _REFRESHABLE[rk]                  # module gone, cache cleared (.to()), or the entry was rebuilt as not refreshable
        else:
            live.append(e)
    todo = [e for e in live if any(t.requires_grad for t in e.tensors)]
    by_dev = {}
    for e in todo:
        by_dev.setdefault(e.tensors[0].device, []).append(e)
    for dev, ents in by_dev.items():
        if dev.type != "cuda":
            continue                              # no device, no launch: those entries are rebuilt on their next use
        key = tuple(id(e) for e in ents) + tuple((r[0].data_ptr(), r[1].data_ptr()) for e in ents for r in e.recs)
        hit = _TABLE.get(dev)
        if hit is None or hit[0] != key:
            chunk = _lib.load().emip_repack_chunk()
            recs, bmap = bytearray(), []
            i = 0
            for e in ents:
                for (src, dst, d, st, base, valid3, scale) in e.recs:
                    n = d[0] * d[1] * d[2] * d[3]
                    recs += struct.pack("<QQqqqqqqiiiiif", src.data_ptr(), dst.data_ptr(), n, st[0], st[1], st[2], st[3],
                                        base, d[1], d[2], d[3], valid3, 1 if dst.dtype == torch.bfloat16 else 0, scale)
                    bmap += [(i, c) for c in range((n + chunk - 1) // chunk)]
                    i += 1
            if not bmap:
                continue
            hit = (key, torch.frombuffer(recs, dtype=torch.uint8).clone().to(dev),
                   torch.tensor(bmap, dtype=torch.int32).to(dev), len(bmap))
            _TABLE[dev] = hit
        with torch.cuda.device(dev):
            _lib.call("emip_repack", hit[1].data_ptr(), hit[2].data_ptr(), hit[3], torch.cuda.current_stream(dev).cuda_stream)
        for e in ents:
            e.sig = tuple((t.data_ptr(), t._version, t.device) for t in e.tensors) + e.mode
    _rebuild_in_place()


def _rebuild_in_place():
    """the non-refreshable packs of trainable parameters: builder again, result copied into the tensors the entry already holds
    (same addresses: a captured graph keeps reading current weights); an entry whose builder now returns another structure
    is left to the rebuild-on-version-change path"""
    global _REC
    for rk, ref in list(_REBUILD.items()):
        m = ref()
        e = m._pack_cache.get(rk[1]) if m is not None else None
        if e is None or e.recs is not None or e.builder is None:
            del _REBUILD[rk]
            continue
        if not any(t.requires_grad for t in e.tensors) or not e.tensors[0].is_cuda:
            continue
        if e.mode != (m.cdtype, m.training):
            continue        # built for the other mode (an eval-only pack such as a folded BatchNorm): rebuilt when that mode uses it again
        sig = tuple((t.data_ptr(), t._version, t.device) for t in e.tensors) + e.mode
        if sig == e.sig:
            continue
        prev, _REC = _REC, []
        try:
            with torch.no_grad():
                new = e.builder(*e.tensors)
        finally:
            _REC = prev
        try:
            old_t, new_t = _flat_tensors(e.val, []), _flat_tensors(new, [])
        except TypeError:
            continue
        if len(old_t) != len(new_t) or any(o.shape != n.shape or o.dtype != n.dtype for o, n in zip(old_t, new_t)):
            continue
        with torch.no_grad():
            for o, n in zip(old_t, new_t):
                if o.data_ptr() != n.data_ptr():
                    o.copy_(n)
        e.sig = sig


def packs_not_kept_current(model):
    """(module name, cache key) of every pack of a trainable parameter that neither refresh path keeps current in place --
    what a captured training step would read stale.  Empty for the EMIP modules (tests/test_train_graph_gpu.py)."""
    out = []
    for name, m in model.named_modules():
        for key, e in getattr(m, "_pack_cache", {}).items():
            if any(t.requires_grad for t in e.tensors) and e.recs is None and (id(m), key) not in _REBUILD:
                out.append((name, key))
    return out


class EmipModule(nn.Module):
    """nn.Module with a packed-weight cache.  `self.cdtype` is the activation dtype."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_pack_cache", {})

    @property
    def cdtype(self):
        return _DEFAULT_DTYPE[0]

    def packed(self, key, tensors, builder):
        """builder(*tensors) -> packed object, cached until a tensor's version/device changes (or kept current in place by
        refresh_packs(), see above)."""
        global _REC
        mode = (self.cdtype, self.training)
        sig = tuple((t.data_ptr(), t._version, t.device) for t in tensors) + mode
        hit = self._pack_cache.get(key)
        if hit is not None and hit.sig == sig:
            return hit.val
        prev, _REC = _REC, []
        try:
            with torch.no_grad():
                val = builder(*tensors)
        finally:
            recs, _REC = _REC, prev
        # refreshable: every tensor the builder returned is a recorded pack of one of `tensors`, or one of them itself
        srcs = {t.untyped_storage().data_ptr() for t in tensors}
        recs = [r for r in recs if r[0].untyped_storage().data_ptr() in srcs]
        dsts = {r[1].data_ptr() for r in recs}
        try:
            ok = all(o.data_ptr() in dsts or o.untyped_storage().data_ptr() in srcs for o in _flat_tensors(val, []))
        except TypeError:
            ok = False
        ent = _Entry(sig, val, recs if ok else None, tuple(tensors), mode, builder)
        self._pack_cache[key] = ent
        if ok:
            _REFRESHABLE[(id(self), key)] = weakref.ref(self)
        elif any(t.requires_grad for t in tensors):
            _REBUILD[(id(self), key)] = weakref.ref(self)
        return val

    def _apply(self, fn, *a, **k):  # .to()/.cuda(): drop packed copies living on the old device
        self._pack_cache.clear()
        return super()._apply(fn, *a, **k)


# ---------------------------------------------------------------------------------------------
# packing helpers (weight preprocessing only)


def f32(t):
    return t.detach().float().contiguous()


def pack_linear(w, dtype):
    out = w.detach().to(dtype).contiguous()
    _record(w, out, (1, 1, 1, w.numel()), (0, 0, 0, 1))
    return out


def lin_packs(w, dtype):
    """forward pack [N,K] and input-gradient pack W^T [K,N] of a Linear weight"""
    n, k = w.shape
    wt = w.detach().t().to(dtype).contiguous()
    _record(w, wt, (1, 1, k, n), (0, 0, 1, k))
    return pack_linear(w, dtype), wt


def lin_packs_kpad(w, dtype, kpad):
    """[N,K] -> ([N,kpad], [kpad,N]): forward and input-gradient packs with the K axis zero-padded"""
    n, k = w.shape
    wo = torch.zeros(n, kpad, dtype=dtype, device=w.device)
    wo[:, :k] = w.detach()
    wt = torch.zeros(kpad, n, dtype=dtype, device=w.device)
    wt[:k] = w.detach().t()
    _record(w, wo, (1, 1, n, kpad), (0, 0, k, 1), 0, k)
    _record(w, wt, (1, 1, k, n), (0, 0, 1, k))            # the zero rows behind it never change
    return wo, wt


def pack_conv(w, dtype, cin_pad=None, perm=None, record=True):
    """[Cout,Cin,KH,KW] -> [Cout, KH*KW*Cin_pad] with ci fastest (matches emip_conv2d).
    perm: optional input-channel permutation applied before padding."""
    src = w
    w = w.detach().float()
    if perm is not None:
        w = w[:, perm]
    co, ci, kh, kw = w.shape
    if cin_pad is not None and cin_pad > ci:
        w = torch.cat([w, w.new_zeros(co, cin_pad - ci, kh, kw)], 1)
    out = w.permute(0, 2, 3, 1).reshape(co, -1).to(dtype).contiguous()
    if perm is None and record:
        _record(src, out, (co, kh, kw, out.shape[1] // (kh * kw)), (ci * kh * kw, kw, 1, kh * kw), 0, ci)
    return out


def conv_dgrad_pack(w, dtype, k, s, p):
    """weights for the input gradient of a conv [Cout,Cin,k,k]: W^T [k*k*Cin, Cout] for non-overlapping patch convs (one
    GEMM + un-patchify), else the spatially flipped, channel-transposed kernel packed like a forward conv [Cin, k*k*Cout]"""
    co, ci, kh, kw = w.shape
    if k == s and p == 0:
        out = pack_conv(w, dtype, record=False).t().contiguous()                          # [(ky,kx,ci)][co]
        _record(w, out, (kh, kw, ci, co), (kw, 1, kh * kw, ci * kh * kw))
        return out
    out = pack_conv(w.detach().flip(2, 3).permute(1, 0, 2, 3), dtype, record=False)         # [ci][ky][kx][co]
    _record(w, out, (ci, kh, kw, co), (kh * kw, -kw, -1, ci * kh * kw), (kh - 1) * kw + (kw - 1))
    return out


def fold_bn(w, b, bn, eps=None):
    """Fold an eval-mode BatchNorm2d into the preceding conv: returns (w', b') in f32."""
    eps = bn.eps if eps is None else eps
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + eps)
    w2 = w.detach().float() * scale.view(-1, 1, 1, 1)
    b0 = b.detach().float() if b is not None else torch.zeros_like(scale)
    b2 = (b0 - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
    return w2, b2.contiguous()


def pack_dw(w, flip=False):
    """depthwise [C,1,3,3] -> f32 [9][C]; flip: taps reversed (the input-gradient kernel)"""
    c = w.shape[0]
    x = w.detach().float()
    out = (x.flip(2, 3) if flip else x).reshape(c, 9).t().contiguous()
    _record(w, out, (1, 1, 9, c), (0, 0, -1 if flip else 1, 9), 8 if flip else 0)
    return out


def to_cl(x, dtype, cpad=None):
    """module-boundary conversion: planar [B,C,H,W] (any float dtype) -> channels-last"""
    from . import ops
    return ops.planar_to_cl(x.detach().float().contiguous(), dtype, cpad)


def to_planar(x, xc=0, C=None):
    from . import ops
    return ops.cl_to_planar(x, xc, C)
