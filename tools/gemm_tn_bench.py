#!/usr/bin/env python3
"""Calibration: emip_gemm_tn (weight gradients) on the PVTv2-b5 training shapes at batch 32 pairs (64 images): the
register-staged body (gemm_tn.hip) against the LDS-DMA ring body (gemm_tn8.hip) at ring depths 2-4 and several workgroup
targets.  Needs the tuning library:  EMIP_TUNING=1 python tools/gemm_tn_bench.py"""
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
ops_ = {}
for M, N, K in shapes:
    ops_[(M, N, K)] = (torch.randn(M, N, device=dev).to(dt), torch.randn(M, K, device=dev).to(dt))
configs = [("staged", -1, 0, 0), ("auto", 0, 0, 0)] + [("ring%d/%d" % (n, t), 0, n, t) for n in (-2, 2, -3, 3, 4) for t in (256, 512)]
if not hasattr(lib, "emip_debug_set_tn8"):
    configs = [("product", None, None, None)]
res = {}
for name, tn, nst, tgt in configs:
    if tn is not None:
        lib.emip_debug_set_tn(ctypes.c_int(tn))
        lib.emip_debug_set_tn8(ctypes.c_int(nst), ctypes.c_int(tgt))
    for sh in shapes:
        a, b = ops_[sh]
        res[(name, sh)] = timeit(lambda: ops.gemm_tn(a, b))
print("%-24s" % "M x N x K" + "".join("%12s" % c[0] for c in configs))
for sh in shapes:
    M, N, K = sh
    print("%-24s" % ("%d x %d x %d" % sh) + "".join("%9.1f us" % res[(c[0], sh)] for c in configs))
    print("%-24s" % "  TFLOP/s" + "".join("%12.0f" % (2.0 * M * N * K / res[(c[0], sh)] / 1e6) for c in configs))
# correctness of the ring body against the staged one on one awkward shape (tails in all three dimensions)
if hasattr(lib, "emip_debug_set_tn8"):
    a = torch.randn(5003, 328, device=dev).to(dt)
    b = torch.randn(5003, 200, device=dev).to(dt)
    lib.emip_debug_set_tn(ctypes.c_int(-1))
    ref = ops.gemm_tn(a, b)
    for nst in (2, 3, 4):
        lib.emip_debug_set_tn(ctypes.c_int(0)); lib.emip_debug_set_tn8(ctypes.c_int(nst), ctypes.c_int(0))
        got, db = ops.gemm_tn(a, b, with_colsum=True)
        print("ring%d vs staged: max |d| %.3e (|ref| max %.1f); db err %.3e" % (
            nst, (got - ref).abs().max().item(), ref.abs().max().item(), (db - a.float().sum(0)).abs().max().item()))
