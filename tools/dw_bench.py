#!/usr/bin/env python3
"""Depthwise 3x3 + GELU on the PVT Mlp shapes (32 images, bf16): microseconds per launch from a hipGraph of 20 back-to-back
launches, plus max |y - reference| against an f32 torch restatement (erf GELU).  EMIP_HIP_LIB selects the library build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from emip_amd import _lib, ops

_lib.load()
torch.manual_seed(0)
for (B, H, C) in ((32, 88, 256), (32, 44, 512), (32, 22, 1280), (32, 11, 2048), (64, 22, 1280)):
    x = (torch.randn(B, H, H, C, device="cuda") * 1.5).to(torch.bfloat16)
    w = torch.randn(C, 1, 3, 3, device="cuda") * 0.3
    b = torch.randn(C, device="cuda") * 0.1
    wt = w.reshape(C, 9).t().contiguous()
    y = ops.dwconv3x3(x, wt, bias=b, act=ops.ACT_GELU)
    ref = F.gelu(F.conv2d(x.float().permute(0, 3, 1, 2), w, b, padding=1, groups=C)).permute(0, 2, 3, 1)
    err = (y.float() - ref).abs().max().item()
    out = torch.empty_like(x)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ops.dwconv3x3(x, wt, bias=b, act=ops.ACT_GELU, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                ops.dwconv3x3(x, wt, bias=b, act=ops.ACT_GELU, out=out)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100
    byt = 4.0 * x.numel()
    print("B=%d %dx%d C=%d: %.1f us/launch  %.2f TB/s  max|err| %.3e (ref max %.2f)" % (B, H, H, C, us, byt / us / 1e6, err, ref.abs().max().item()))
