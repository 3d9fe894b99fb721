#!/usr/bin/env python3
"""emip_mlp_fc1dw against its two-launch form (emip_gemm_lne + emip_dwconv3x3) on the stage-3 / stage-4 Mlp shapes:
max |difference| and microseconds per call from hipGraphs of 10 calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, ops

_lib.load()
torch.manual_seed(0)


def graph_us(fn, n=10, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


for (B, H, K, N) in ((32, 22, 320, 1280), (16, 22, 320, 1280), (3, 22, 320, 1280)):
    x = (torch.randn(B, H, H, K, device="cuda") * 1.3 + 0.2).to(torch.bfloat16)
    xf = x.float().view(-1, K)
    stats = torch.stack((xf.sum(1), (xf * xf).sum(1)), 1).contiguous()
    w1 = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b1 = torch.randn(N, device="cuda") * 0.1
    cs = w1.float().sum(1).contiguous()
    wd = (torch.randn(9, N, device="cuda") * 0.3).contiguous()
    bd = torch.randn(N, device="cuda") * 0.1

    def two():
        t = ops.gemm(x, w1, bias=b1, ln_stats=stats, ln_eps=1e-6, colsum=cs)
        return ops.dwconv3x3(t, wd, bd, act=ops.ACT_GELU)

    def one():
        return ops.mlp_fc1dw(x, w1, b1, cs, stats, 1e-6, wd, bd)
    ref, got = two().float(), one().float()
    err = (ref - got).abs().max().item()
    print("B=%d %dx%d K=%d N=%d: max |fused - two launches| %.3e (max |ref| %.2f);  two launches %.1f us, fused %.1f us" % (
        B, H, H, K, N, err, ref.abs().max().item(), graph_us(two), graph_us(one)))
