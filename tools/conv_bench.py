#!/usr/bin/env python3
"""Calibration: emip_conv2d on the EMIP conv shapes, 50 captured launches per timing (host launch cost excluded).
DEEP=<n>: workgroup-count threshold of the 3-stage register prefetch (emip_debug_set key 6; 0 = off)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import _lib, ops  # noqa: E402


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(4):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 200 * 1e3


_lib.load()
if os.environ.get("DEEP") is not None:
    _lib.call("emip_debug_set", 6, int(os.environ["DEEP"]))
dev, dt = "cuda:0", torch.bfloat16
for B in (int(b) for b in os.environ.get("BATCHES", "8,32").split(",")):
    for H, Cin, Cout, k, s, p in [(88, 64, 64, 8, 8, 0), (44, 128, 128, 4, 4, 0), (22, 320, 320, 2, 2, 0), (88, 64, 128, 3, 2, 1),
                                  (44, 128, 320, 3, 2, 1), (22, 320, 512, 3, 2, 1), (176, 64, 64, 3, 1, 1), (88, 96, 96, 3, 1, 1),
                                  (44, 128, 128, 3, 1, 1), (44, 128, 32, 3, 1, 1), (11, 512, 32, 3, 1, 1)]:
        x = torch.randn(B, H, H, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, k * k * Cin, device=dev) / (k * k * Cin) ** 0.5).to(dt)
        us = timeit(lambda: ops.conv2d(x, w, k, k, s, p))
        Ho = (H + 2 * p - k) // s + 1
        fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
        print("B%2d %3dx%-3d Cin%4d Cout%4d k%d s%d: %7.1f us %7.1f TF/s" % (B, H, H, Cin, Cout, k, s, us, fl / us / 1e6))
