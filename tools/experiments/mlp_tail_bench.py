#!/usr/bin/env python3
"""Calibration: fused emip_mlp_tail against emip_dwconv3x3 + emip_gemm on the PVTv2-b5 shapes."""
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
for B in (8, 32):
    for H, C in ((88, 64), (44, 128), (22, 320), (11, 512)):
        Ch = 4 * C
        h = torch.randn(B, H, H, Ch, device=dev).to(dt)
        wt = torch.randn(9, Ch, device=dev) * 0.3
        bd = torch.randn(Ch, device=dev) * 0.1
        w2 = (torch.randn(C, Ch, device=dev) / Ch ** 0.5).to(dt)
        b2 = torch.randn(C, device=dev)
        res = torch.randn(B, H, H, C, device=dev).to(dt)
        t = torch.empty_like(h)
        o = torch.empty_like(res)
        u_dw = timeit(lambda: ops.dwconv3x3(h, wt, bd, act=ops.ACT_GELU, out=t))
        u_g = timeit(lambda: ops.gemm(t, w2, bias=b2, res=res, out=o))
        u_f = timeit(lambda: ops.mlp_tail(h, wt, bd, w2, b2, res, out=o))
        x = torch.randn(B, H, H, C, device=dev).to(dt)
        w1 = (torch.randn(Ch, C, device=dev) / C ** 0.5).to(dt)
        b1 = torch.randn(Ch, device=dev)
        u_fc1 = timeit(lambda: ops.gemm(x, w1, bias=b1, out=h))
        u_head = timeit(lambda: ops.mlp_head(x, w1, b1, wt, bd, out=t))
        print("B%2d %3dx%-3d C%3d Ch%4d: dwconv %6.1f + fc2 %6.1f = %6.1f us | tail fused %6.1f us || fc1 %6.1f + dwconv "
              "%6.1f = %6.1f us | head fused %6.1f us" % (B, H, H, C, Ch, u_dw, u_g, u_dw + u_g, u_f, u_fc1, u_dw,
                                                         u_fc1 + u_dw, u_head))
