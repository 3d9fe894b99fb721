#!/usr/bin/env python3
"""emip_mlp_band against emip_mlp_fc1dw + the fc2 GEMM on the stage-3 shape, replayed from a hipGraph of 20; also four
launches on four streams at once (what the benchmark's steps in flight do to a launch that leaves CUs free)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_mlp_block_gpu import _setup


def timed(fn, reps=20):
    for _ in range(3):
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
    return best, g


_STREAMS = []


def concurrent(graphs, reps=5):
    # one fixed set of streams, created back to back: streams map onto the runtime's 4 hardware queues round-robin, and a fresh
    # set per call can put two of them on one queue (observed: "2 at once" twice as long as "1", "4 at once" no longer than "2")
    while len(_STREAMS) < len(graphs):
        _STREAMS.append(torch.cuda.Stream())
    streams = _STREAMS[:len(graphs)]
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for g, st in zip(graphs, streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                g.replay()
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / 20)
    return best


def main():
    for B in (16, 32, 8):
        sets = []
        for k in range(4):
            x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, 22, 22, 3 + k)
            stg, taps = ops.mlp_band_packs(w1, b1, colsum, w2, wd, bd)
            out = torch.empty_like(x)
            ost = torch.empty((B * 484, 2), device="cuda")
            st = stats.view(-1)
            def new(x=x, stg=stg, taps=taps, b2=b2, st=st, out=out, ost=ost):
                ops.mlp_band(x, stg, taps, b2, st, 1e-6, out, out_stats=ost)
            def old(x=x, w1=w1, b1=b1, colsum=colsum, st=st, wd=wd, bd=bd, w2=w2, b2=b2, out=out, ost=ost):
                t = ops.mlp_fc1dw(x, w1, b1, colsum, st, 1e-6, wd, bd)
                ops.gemm(t, w2, bias=b2, res=x, out=out, out_stats=ost)
            sets.append((new, old))
        gf = 2 * 2 * B * 484 * 320 * 1280 / 1e9
        tn = [timed(n) for n, _ in sets]
        to = [timed(o) for _, o in sets]
        a, b = to[0][0], tn[0][0]
        ca, cb = concurrent([g for _, g in to]), concurrent([g for _, g in tn])
        print("B=%2d images: fc1dw + fc2 %7.1f us (%5.0f TF/s)   mlp_band %7.1f us (%5.0f TF/s algorithmic)   "
              "4 at once, per launch set: %7.1f / %7.1f us" % (B, a, gf / a * 1e3, b, gf / b * 1e3, ca, cb))


if __name__ == '__main__':
    main()
