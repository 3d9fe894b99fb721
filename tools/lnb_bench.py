"""Calibration: LayerNorm backward, the narrow kernel (8-byte loads, 1024 small workgroups) against the wide one (16-byte
loads, one-pass row sums, <= 256 workgroups of 1024 threads)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


dev, dt = "cuda:0", torch.bfloat16
for M, C in [(30976, 320), (123904, 128), (495616, 64), (7744, 512), (123904, 256), (7744, 320)]:
    x = torch.randn(M, C, device=dev).to(dt)
    dy = torch.randn(M, C, device=dev).to(dt)
    g = torch.ones(C, device=dev)
    dg = torch.zeros(C, device=dev)
    db = torch.zeros(C, device=dev)
    _lib.call("emip_debug_set_lnb", 0)
    t0 = timeit(lambda: ops.layernorm_bwd(x, dy, g, 1e-6, dg, db))
    _lib.call("emip_debug_set_lnb", 1)
    t1 = timeit(lambda: ops.layernorm_bwd(x, dy, g, 1e-6, dg, db))
    gb = 3 * M * C * 2 / 1e3
    print(M, C, "narrow %.1f us (%.0f GB/s) | wide %.1f us (%.0f GB/s)" % (t0, gb / t0, t1, gb / t1))
