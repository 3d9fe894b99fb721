#!/usr/bin/env python3
"""phase ablation of emip_mlp_block (tuning library: EMIP_HIP_LIB=emip_amd/libemip_hip_tuning.so)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, ops
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_mlp_block_gpu import _setup
from mlp_block_bench import timed  # noqa
lib = _lib.load()
B = 32
x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, 22, 22, 3)
cst = ops.mlp_block_consts(wd, bd, b1, colsum)
out = torch.empty_like(x)
ost = torch.empty((B * 484, 2), device="cuda")
st = stats.view(-1)
def new():
    ops.mlp_block(x, w1, w2, cst, b2, st, 1e-6, out, out_stats=ost)
for flags, what in ((0, "full"), (1, "no fc1 MFMA"), (2, "no depthwise pass"), (4, "no fc2 MFMA"), (8, "no weight DMA"), (16, "no H store"),
                    (1 | 16, "no fc1, no H store"), (1 | 2 | 4 | 16, "barriers + DMA only"), (1 | 2 | 4 | 8 | 16, "barriers only")):
    lib.emip_debug_set_mb(flags)
    print("%-28s %7.1f us" % (what, timed(new)))
lib.emip_debug_set_mb(0)
