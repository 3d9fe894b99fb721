#!/usr/bin/env python3
"""emip_mlp_band: one launch at 16 .. 80 images (64 .. 320 workgroups of 160 KB LDS each)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_mlp_block_gpu import _setup
from mlp_band_bench import timed  # noqa
for B in (2, 8, 16, 24, 32, 48, 64, 80):
    x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, 22, 22, 3)
    stg, taps = ops.mlp_band_packs(w1, b1, colsum, w2, wd, bd)
    out = torch.empty_like(x)
    st = stats.view(-1)
    t4 = timed(lambda: ops.mlp_band(x, stg, taps, b2, st, 1e-6, out, out_stats=None, bands=4))[0]
    t8 = timed(lambda: ops.mlp_band(x, stg, taps, b2, st, 1e-6, out, out_stats=None, bands=8))[0]
    print("B=%2d: 4 bands (%3d workgroups) %7.1f us, 8 bands (%3d workgroups) %7.1f us" % (B, 4 * B, t4, 8 * B, t8), flush=True)
