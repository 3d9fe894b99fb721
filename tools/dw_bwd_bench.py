#!/usr/bin/env python3
"""emip_dwconv3x3_bwd_fused against the three launches it replaces, at the training step's shapes (64 images), graph of 10."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops

def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps * 1e3)
    return best

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for H, C in ((88, 256), (44, 512), (22, 1280), (11, 2048)):
    bf = torch.bfloat16
    x = torch.randn(B, H, H, C, device="cuda").to(bf)
    z = torch.randn(B, H, H, C, device="cuda").to(bf)
    dy = torch.randn(B, H, H, C, device="cuda").to(bf)
    wt = torch.randn(9, C, device="cuda")
    wf = wt.flip(0).contiguous()
    acc = torch.zeros(10 * C, device="cuda")
    dw9 = torch.zeros(9, C, device="cuda"); db = torch.zeros(C, device="cuda")
    def old():
        dz = ops.gelu_bwd(z, dy)
        dx = ops.dwconv3x3(dz, wf)
        ops.dwconv3x3_wgrad(x, dz, dw9, db)
        return dx
    def new():
        return ops.dwconv3x3_bwd_fused(x, z, dy, wt, acc[:9 * C], acc[9 * C:], True)
    to, tn = timed(old), timed(new)
    mb = x.numel() * 2 / 1e6
    print("%3d x %3d x %4d  (%6.1f MB per pass)   three launches %7.1f us   fused %7.1f us (%.2f TB/s over 4 passes)" % (
        H, H, C, mb, to, tn, 4 * mb / tn / 1e6 * 1e6 / 1e6), flush=True)
