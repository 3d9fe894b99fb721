#!/usr/bin/env python3
"""Calibration: emip_gemm_tn (weight gradients) on the PVTv2-b5 training shapes at batch 32 pairs (64 images)."""
import ctypes
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


lib = _lib.load()
dev, dt = "cuda:0", torch.bfloat16
shapes = [(30976, 320, 320), (30976, 1280, 320), (30976, 320, 1280), (7744, 640, 320), (123904, 512, 128),
          (123904, 128, 512), (123904, 128, 128), (495616, 256, 64), (495616, 64, 256), (495616, 64, 64),
          (7744, 2048, 512), (7744, 512, 2048)]
for target in (int(x) for x in os.environ.get("TARGETS", "1024,512,256").split(",")):
    if hasattr(lib, "emip_debug_set_tn"):
        lib.emip_debug_set_tn(ctypes.c_int(target))
    print("target workgroups", target)
    for M, N, K in shapes:
        a = torch.randn(M, N, device=dev).to(dt)
        b = torch.randn(M, K, device=dev).to(dt)
        us = timeit(lambda: ops.gemm_tn(a, b))
        print("  tn %7d %5d %5d : %7.1f us %7.1f TF/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
