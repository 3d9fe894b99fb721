#!/usr/bin/env python3
"""the spatial-reduction conv of the 22 x 22 stage (16 images: 1936 x 320 outputs over K = 1280, per-tap LayerNorm) on the tile
configurations that take it, with and without its row statistics"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops
from tools.gemm8_batched_sweep import timeit
B = int(os.environ.get("B", "16"))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, 22, 22, 320, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(320, 1280, device="cuda", generator=g) / 36).to(torch.bfloat16)
bias = torch.randn(320, device="cuda", generator=g)
tsum = w.float().view(320, 4, 320).sum(2).t().contiguous()
xf = x.float().view(-1, 320)
stats = torch.stack((xf.sum(1), (xf * xf).sum(1)), 1).contiguous().view(-1)
for cfg in (9, 10, 11):
    st = torch.zeros(B * 121 * 2, device="cuda")
    t1 = timeit(lambda: ops.conv8(x, w, 2, 2, 2, 0, bias=bias, ln_stats=stats, tapsum=tsum, ln_eps=1e-6, out_stats=st, cfg=cfg), n=20)
    t0 = timeit(lambda: ops.conv8(x, w, 2, 2, 2, 0, bias=bias, ln_stats=stats, tapsum=tsum, ln_eps=1e-6, cfg=cfg), n=20)
    print("cfg %2d: %5.1f us with row statistics, %5.1f us without" % (cfg, t1, t0), flush=True)
