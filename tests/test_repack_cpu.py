"""The packed-weight refresh records (emip_amd/nn_base.py) restated on the CPU: every pack helper must describe the tensor it
just built as dst = src.flatten()[base + sum_i idx_i stride_i] with zeros where idx_3 >= valid3 -- the contract of
emip_repack (include/emip_hip.h), which rewrites the packs after an optimizer step on the device."""
import numpy as np
import torch

from emip_amd import nn_base


def _emulate(rec):
    src, dst, d, st, base, valid3, scale = rec
    flat = src.detach().float().reshape(-1).numpy() * np.float32(scale)
    idx = np.indices(d).reshape(4, -1)
    off = base + sum(idx[i].astype(np.int64) * st[i] for i in range(4))
    ok = idx[3] < valid3
    vals = np.where(ok, flat[np.where(ok, off, 0)], 0.0).astype(np.float32)
    n = vals.size
    got = dst.detach().float().reshape(-1).numpy()
    return vals, got[:n], got[n:]


def _check(build, nrec):
    nn_base._REC = []
    try:
        build()
        recs = nn_base._REC
    finally:
        nn_base._REC = None
    assert len(recs) == nrec, len(recs)
    for r in recs:
        want, got, tail = _emulate(r)
        # bf16 packs: compare after the same rounding
        if r[1].dtype == torch.bfloat16:
            want = torch.from_numpy(want).to(torch.bfloat16).float().numpy()
        assert np.array_equal(want, got), (r[2], r[3])
        assert not tail.any()


def test_linear_and_transposed_packs():
    w = torch.randn(24, 40)
    _check(lambda: nn_base.lin_packs(w, torch.bfloat16), 2)
    _check(lambda: nn_base.lin_packs_kpad(w, torch.bfloat16, 48), 2)


def test_conv_packs():
    w = torch.randn(16, 6, 3, 3)
    _check(lambda: nn_base.pack_conv(w, torch.bfloat16), 1)
    _check(lambda: nn_base.pack_conv(w, torch.bfloat16, cin_pad=8), 1)
    _check(lambda: nn_base.conv_dgrad_pack(w, torch.bfloat16, 3, 1, 1), 1)
    w2 = torch.randn(16, 8, 2, 2)
    _check(lambda: nn_base.conv_dgrad_pack(w2, torch.bfloat16, 2, 2, 0), 1)
    w7 = torch.randn(8, 3, 7, 7)
    _check(lambda: nn_base.conv_dgrad_pack(w7, torch.bfloat16, 7, 4, 3), 1)


def test_depthwise_packs():
    w = torch.randn(32, 1, 3, 3)
    _check(lambda: nn_base.pack_dw(w), 1)
    _check(lambda: nn_base.pack_dw(w, flip=True), 1)


def test_only_entries_fully_covered_by_records_are_refreshable():
    class M(nn_base.EmipModule):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(8, 16))
            self.b = torch.nn.Parameter(torch.randn(8))
    m = M()
    nn_base.set_default_dtype(torch.bfloat16)
    try:
        m.packed("a", (m.w, m.b), lambda w, b: (nn_base.lin_packs(w, torch.bfloat16), nn_base.f32(b)))
        m.packed("b", (m.w,), lambda w: (w.detach() * 2).to(torch.bfloat16))
        m.packed("c", (m.w,), lambda w: dict(a=nn_base.pack_linear(w, torch.bfloat16), folded=(w.detach() * 2).to(torch.bfloat16)))
        m.packed("d", (m.w, m.b), lambda w, b: dict(a=nn_base.pack_linear(w, torch.bfloat16), b=nn_base.f32(b), n=3))
    finally:
        nn_base.set_default_dtype(torch.float32)
    assert m._pack_cache["a"].recs is not None and len(m._pack_cache["a"].recs) == 2
    assert m._pack_cache["b"].recs is None
    assert m._pack_cache["c"].recs is None            # a dict holding one tensor no record covers
    assert m._pack_cache["d"].recs is not None and len(m._pack_cache["d"].recs) == 1
