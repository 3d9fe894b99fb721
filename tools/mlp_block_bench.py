#!/usr/bin/env python3
"""emip_mlp_block against emip_mlp_fc1dw + the fc2 GEMM on the stage-3 shape, replayed from a hipGraph of 20."""
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
    return best

def main():
  for B in (32, 16, 64):
      x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, 22, 22, 3)
      cst = ops.mlp_block_consts(wd, bd, b1, colsum)
      out = torch.empty_like(x)
      ost = torch.empty((B * 484, 2), device="cuda")
      st = stats.view(-1)
      def new():
          ops.mlp_block(x, w1, w2, cst, b2, st, 1e-6, out, out_stats=ost)
      def old():
          t = ops.mlp_fc1dw(x, w1, b1, colsum, st, 1e-6, wd, bd)
          ops.gemm(t, w2, bias=b2, res=x, out=out, out_stats=ost)
      gf = 2 * 2 * B * 484 * 320 * 1280 / 1e9
      a, b = timed(old), timed(new)
      print("B=%2d images: fc1dw + fc2 %7.1f us (%5.0f TF/s)   mlp_block %7.1f us (%5.0f TF/s algorithmic)" % (B, a, gf / a * 1e3, b, gf / b * 1e3))


if __name__ == '__main__':
    main()
