#!/usr/bin/env python3
"""emip_conv3x3_halo against the launches it replaces at the GMFlow CNN's first level (32 images of 176 x 176 x 64), from
replayed graphs: conv alone, conv + statistics, normalise + conv + statistics"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops


def timeit(fn, n=20, reps=5):
    """us per call from a replayed graph of n calls"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(reps):
            g.replay()
        b.record(st)
        torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (n * reps)


B, H, W = int(os.environ.get("B", "32")), 176, 176
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, H, W, 64, device="cuda", generator=g).to(torch.bfloat16)
wp = (torch.randn(64, 576, device="cuda", generator=g) / 24).to(torch.bfloat16)
pk = ops.conv3x3_halo_pack(wp)
ws = ops.conv3x3_halo_ws(B, H, W, x.device)
s_in = torch.zeros((B, 64, 2), dtype=torch.float64, device="cuda")
ops.chan_stats(x, B, sums=s_in)
s_out = torch.zeros((B, 64, 2), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
xn = torch.empty_like(x)
fl = 2.0 * B * H * W * 64 * 576
def rep(name, t):
    print("%-64s %7.1f us  %6.0f TFLOP/s" % (name, t, fl / t / 1e6), flush=True)
rep("implicit-GEMM conv (emip_conv2d)", timeit(lambda: ops.conv2d(x, wp, 3, 3, 1, 1, out=y)))
def old_chain():
    ops.chan_norm_apply(x, s_in, B, 1e-5, relu_inner=True, out=xn)
    ops.conv2d(xn, wp, 3, 3, 1, 1, out=y, zero=s_out)
    ops.chan_stats(y, B, sums=s_out)
rep("normalise pass + implicit-GEMM conv + statistics pass", timeit(old_chain))
rep("halo conv", timeit(lambda: ops.conv3x3_halo(x, pk, out=y)))
rep("halo conv + statistics", timeit(lambda: ops.conv3x3_halo(x, pk, out=y, out_sums=s_out, ws=ws)))
rep("halo conv, normalise on staging", timeit(lambda: ops.conv3x3_halo(x, pk, out=y, in_sums=s_in)))
rep("halo conv, normalise on staging + statistics", timeit(lambda: ops.conv3x3_halo(x, pk, out=y, in_sums=s_in, out_sums=s_out, ws=ws)))

# ---- the stem (7 x 7, stride 2, 3 -> 64 channels)
xs = torch.zeros(B, 352, 352, 8, device="cuda", dtype=torch.bfloat16)
xs[..., :3] = torch.randn(B, 352, 352, 3, device="cuda", generator=g).to(torch.bfloat16)
wps = (torch.randn(64, 392, device="cuda", generator=g) / 12).to(torch.bfloat16)
pks = ops.conv_stem_pack(wps)
fl = 2.0 * B * 176 * 176 * 64 * 147
rep("stem: implicit-GEMM conv", timeit(lambda: ops.conv2d(xs, wps, 7, 7, 2, 3, out=y)))
def stem_old():
    ops.conv2d(xs, wps, 7, 7, 2, 3, out=y, zero=s_out)
    ops.chan_stats(y, B, sums=s_out)
rep("stem: implicit-GEMM conv + statistics pass", timeit(stem_old))
rep("stem: direct conv", timeit(lambda: ops.conv_stem(xs, pks, out=y)))
rep("stem: direct conv + statistics", timeit(lambda: ops.conv_stem(xs, pks, out_sums=s_out, ws=ws, out=y)))
