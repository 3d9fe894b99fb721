#!/usr/bin/env python3
"""phase ablation of emip_mlp_band (tuning library: EMIP_HIP_LIB=emip_amd/libemip_hip_tuning.so)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, ops
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_mlp_block_gpu import _setup
from mlp_band_bench import timed  # noqa
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, 22, 22, 3)
stg, taps = ops.mlp_band_packs(w1, b1, colsum, w2, wd, bd)
out = torch.empty_like(x)
ost = torch.empty((B * 484, 2), device="cuda")
st = stats.view(-1)
def new():
    ops.mlp_band(x, stg, taps, b2, st, 1e-6, out, out_stats=ost)
if len(sys.argv) > 2 and sys.argv[2] == "zeros":       # the same instruction stream on zeros: what the data costs (clocks under load)
    x.zero_(); stg.zero_(); taps.zero_()
for flags, what in ((0, "full"), (1, "no fc1 MFMA"), (2, "no depthwise pass"), (4, "no fc2 MFMA"), (8, "no weight DMA"), (16, "no H stores"),
                    (64, "no G stores"), (16 | 64, "no H / G stores"),
                    (32, "constant taps"), (1 | 4, "no MFMA at all"), (1 | 2 | 4, "DMA + barriers + fc1 epilogue"), (1 | 2 | 4 | 16, "DMA + barriers"),
                    (1 | 2 | 4 | 8 | 16, "barriers only"), (2 | 8, "MFMA phases only, no DMA"), (1 | 4 | 8, "depthwise only, no DMA")):
    lib.emip_debug_set_md(flags)
    print("%-34s %7.1f us" % (what, timed(new)[0]), flush=True)
lib.emip_debug_set_md(0)
