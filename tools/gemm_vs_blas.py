#!/usr/bin/env python3
"""Calibration (not product code): emip_gemm against the vendor GEMM on the EMIP dense shapes, bias epilogue only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("GRAPH"):          # replay 50 captured launches: host launch cost (~10 us per call) drops out
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
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


dev, dt = "cuda:0", torch.bfloat16
lib = _lib.load()
if os.environ.get("RESIDENT"):
    _lib.call("emip_debug_set", 5, int(os.environ["RESIDENT"]))
if os.environ.get("TILE"):
    _lib.call("emip_debug_set", 1, int(os.environ["TILE"]))
shapes = [tuple(int(v) for v in t.split('x')) for t in os.environ['SHAPES'].split(',')] if os.environ.get('SHAPES') else [(15488, 1280, 320), (15488, 320, 1280), (15488, 320, 320), (15488, 640, 320), (61952, 512, 128),
          (61952, 128, 512), (61952, 128, 128), (61952, 256, 128), (247808, 256, 64), (247808, 64, 256),
          (247808, 64, 64), (247808, 128, 64), (3872, 2048, 512), (3872, 512, 2048), (3872, 512, 512),
          (3872, 1024, 512), (61952, 1024, 256), (61952, 128, 1024), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b = torch.randn(N, device=dev)
    bb = b.to(dt)
    o = torch.empty(M, N, device=dev, dtype=dt)
    us = timeit(lambda: ops.gemm(a, w, bias=b, out=o))
    ub = timeit(lambda: torch.nn.functional.linear(a, w, bb)) if not os.environ.get('NOBLAS') else 1.0
    ref = torch.nn.functional.linear(a, w, bb)
    err = (o.float() - ref.float()).abs().max().item()
    fl = 2.0 * M * N * K
    byt = 2.0 * (M * K + N * K + M * N)
    t = lib.emip_gemm_tile(M, N, 1, K)
    print("%7d %5d %5d tile %dx%d: emip %7.1f us %7.1f TF/s %5.2f TB/s | blas %7.1f us %7.1f TF/s | ratio %.2f err %.3g" % (
        M, N, K, t // 1000, t % 1000, us, fl / us / 1e6, byt / us / 1e6, ub, fl / ub / 1e6, us / ub, err))
