#!/usr/bin/env python3
"""a handful of emip_mlp_band launches (for rocprofv3 --pmc / --kernel-trace)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_mlp_block_gpu import _setup
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, 22, 22, 3)
stg, taps = ops.mlp_band_packs(w1, b1, colsum, w2, wd, bd)
out = torch.empty_like(x)
st = stats.view(-1)
for _ in range(5):
    ops.mlp_band(x, stg, taps, b2, st, 1e-6, out)
torch.cuda.synchronize()
