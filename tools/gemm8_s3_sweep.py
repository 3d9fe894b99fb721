#!/usr/bin/env python3
"""tile sweep of emip_gemm8 at the 16-image shapes of the 22 x 22 stage WITH the row statistics the block needs (a launch whose
rows span more than two column tiles pays a row_stats pass behind it: the sweep times both)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, ops
from tools.gemm8_bench import NCFG


def timeit(fn, n=20, reps=5):
    """us per call from a replayed graph of n calls (eager back-to-back launches of 20 us are host-bound)"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (n * reps)

_lib.load()
for M, N, K, what in [(7744, 320, 1280, "fc2"), (7744, 320, 320, "proj"), (15488, 320, 1280, "fc2 at 32 images")]:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    line = "%-18s %6d x %4d x %5d + res + row statistics |" % (what, M, N, K)
    ref = None
    for c in range(1, NCFG + 1):
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        st = torch.zeros(M, 2, device="cuda")
        try:
            ops.gemm8(a, w, bias=bias, res=res, out=out, out_stats=st, cfg=c)
        except Exception as e:
            line += " c%d  --  " % c
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = (out.clone(), st.clone())
        ok = torch.equal(out, ref[0]) and torch.allclose(st, ref[1], rtol=1e-5, atol=1e-3)
        t = timeit(lambda: ops.gemm8(a, w, bias=bias, res=res, out=out, out_stats=st, cfg=c))
        line += " c%d %5.1f%s" % (c, t, "" if ok else "!")
    print(line + " | auto c%d" % _lib.load().emip_gemm8_auto_cfg(M, N, K), flush=True)
