"""Calibration: the q + spatial-reduction-conv pair launch at the three PVT stage shapes of an 8-image sub-batch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emip_amd import ops  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


dev, dt = "cuda:0", torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for H, C, k in [(88, 64, 8), (44, 128, 4), (22, 320, 2)]:
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    wq = (torch.randn(C, C, device=dev) / C ** 0.5).to(dt)
    wsr = (torch.randn(C, k * k * C, device=dev) / (k * k * C) ** 0.5).to(dt)
    bq, bsr = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    xf = x.float().view(-1, C)
    stats = torch.stack([xf.sum(1), (xf ** 2).sum(1)], 1).contiguous()
    q = torch.empty_like(x)
    Ho = H // k
    s_out = torch.empty(B, Ho, Ho, C, device=dev, dtype=dt)
    st = torch.zeros(B * Ho * Ho, 2, device=dev)
    cs = wq.float().sum(1).contiguous()

    def pair():
        ops.conv2d_pair(ops.conv_desc(x, wq, 1, 1, 0, bq, q, stats, 1e-6, colsum=cs),
                        ops.conv_desc(x, wsr, k, k, 0, bsr, s_out, stats, 1e-6, out_stats=st), dt)
    t_pair = timeit(pair)
    t_q = timeit(lambda: ops.gemm(x, wq, bias=bq, ln_stats=stats, ln_eps=1e-6, colsum=cs, out=q))
    t_sr = timeit(lambda: ops.conv2d(x, wsr, k, k, k, 0, bias=bsr, ln_stats=stats, ln_eps=1e-6, out_stats=st, out=s_out))
    print("B=%d %dx%d C=%d sr=%d: pair %.1f us | q alone %.1f us | sr conv alone %.1f us (K tiles %d, sr workgroups %d)" % (
        B, H, H, C, k, t_pair, t_q, t_sr, k * k * C // 64, ((B * Ho * Ho + 63) // 64) * ((C + 63) // 64)))
