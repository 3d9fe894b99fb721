"""Calibration: depthwise-3x3 weight gradient and GELU backward at the PVT Mlp shapes of a 32-pair training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for H, C in [(88, 256), (44, 512), (22, 1280), (11, 2048), (44, 680)]:
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    dy = torch.randn(B, H, H, C, device=dev).to(dt)
    dw = torch.zeros(9, C, device=dev)
    db = torch.zeros(C, device=dev)
    gb = 2 * x.numel() * 2 / 1e3
    res = []
    for chunks in (0, 1, 2, 4, 8, 16, 22):
        _lib.call("emip_debug_set_dww", chunks)
        t = timeit(lambda: ops.dwconv3x3_wgrad(x, dy, dw, db))
        res.append("%d: %.0f us" % (chunks, t))
    _lib.call("emip_debug_set_dww", 0)
    print("dwconv wgrad B=%d %dx%d C=%d (%.0f MB) chunks -> %s" % (B, H, H, C, gb / 1e3, " | ".join(res)))

for M, C in [(64 * 484, 1280), (64 * 1936, 512), (64 * 7744, 256)]:
    z = torch.randn(M, C, device=dev).to(dt)
    dy = torch.randn(M, C, device=dev).to(dt)
    t = timeit(lambda: ops.gelu_bwd(z, dy))
    print("gelu_bwd %d x %d: %.1f us (%.0f GB/s)" % (M, C, t, 3 * M * C * 2 / 1e3 / t))

for B, H, C in [(8, 176, 64), (8, 88, 96), (8, 44, 128), (64, 176, 64)]:
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    r = torch.randn(B, H, H, C, device=dev).to(dt)
    sums = ops.chan_stats(x, B)
    t = timeit(lambda: ops.chan_norm_apply(x, sums, B, 1e-5, True, True, res=r))
    print("instance-norm apply B=%d %dx%d C=%d: %.1f us (%.0f GB/s)" % (B, H, H, C, t, 3 * x.numel() * 2 / 1e3 / t))
