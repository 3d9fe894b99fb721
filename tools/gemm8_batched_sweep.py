#!/usr/bin/env python3
"""tile sweep of emip_gemm8_batched at the two per-image GEMMs of the factored conv_corr.0 (16 images)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops


def timeit(fn, n=10, reps=5):
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


if __name__ == "__main__":
  for batch, M, N, K, shared in [(16, 8712, 128, 1984, True), (16, 1936, 968, 1152, False)]:
      g = torch.Generator(device="cuda").manual_seed(1)
      a = torch.randn(1 if shared else batch, M, K, device="cuda", generator=g).to(torch.bfloat16)
      w = (torch.randn(batch, N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
      out = torch.empty((batch, M, N), device="cuda", dtype=torch.bfloat16)
      bsA = 0 if shared else M * K
      t0 = timeit(lambda: ops.gemm_batched_bias(a, w, out, batch, M, N, K, K, K, N, bsA, N * K, M * N))
      line = "%2d x %5d x %4d x %5d: 4-wave %6.1f us |" % (batch, M, N, K, t0)
      for c in range(1, 12):
          try:
              t = timeit(lambda: ops.gemm8_batched(a, w, out, batch, M, N, K, K, K, N, bsA, N * K, M * N, cfg=c))
              line += " c%d %5.1f" % (c, t)
          except Exception:
              line += " c%d  -- " % c
      print(line + "  (%.0f GFLOP)" % (2.0 * batch * M * N * K / 1e9), flush=True)
