#!/usr/bin/env python3
"""Where a gemm8 launch spends its time: the same launch with parts switched off (tuning library only -- the product
library has no such switches):  EMIP_HIP_LIB=emip_amd/libemip_hip_tuning.so python tools/gemm8_ablate.py
Each figure = one launch inside a hipGraph of 40 back-to-back launches (so it includes the kernel boundary)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd import _lib, ops  # noqa: E402


def graph_time(fn, n=40, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (n * reps)


def main():
    lib = _lib.load()
    dbg = lib.emip_tuning_gemm8_dbg
    names = {0: "full", 1: "no stores", 2: "no MFMA", 4: "no loads", 3: "loads only", 5: "MFMA only", 6: "stores only",
             7: "empty loop", 8: "bare launch"}
    for (M, N, K) in [(15488, 320, 320), (7744, 320, 320), (15488, 320, 1280), (7744, 320, 1280)]:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for cfg in (9, 3):
            line = "%6d x %5d x %5d c%d |" % (M, N, K, cfg)
            for d in (0, 1, 2, 4, 3, 5, 6, 7, 8):
                dbg(d)
                t = graph_time(lambda: ops.gemm8(a, w, bias=bias, out=out, cfg=cfg), n=40 if M * N * K < 1e12 else 6)
                line += " %s %.1f |" % (names[d], t)
            dbg(0)
            print(line, flush=True)


if __name__ == "__main__":
    main()
