"""Calibration: LayerNorm backward, plain accumulation (1024 workgroups) against 32 partial accumulators (4096)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import ops  # noqa: E402


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
for M, C in [(30976, 320), (123904, 128), (495616, 64), (7744, 512)]:
    x = torch.randn(M, C, device=dev).to(dt)
    dy = torch.randn(M, C, device=dev).to(dt)
    g = torch.ones(C, device=dev)
    dg = torch.zeros(C, device=dev)
    db = torch.zeros(C, device=dev)
    print(M, C, "plain %.1f us | partial accumulators (incl. fill + column sum) %.1f us" % (
        timeit(lambda: ops.layernorm_bwd(x, dy, g, 1e-6, dg, db)), timeit(lambda: ops.layernorm_bwd_fresh(x, dy, g, 1e-6))))
