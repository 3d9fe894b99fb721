#!/usr/bin/env python3
"""emip_mlp_band under concurrency (tuning library): 1 / 2 / 4 graphs of 20 launches at once, same or different weight sets,
with phase ablations -- what slows a launch down when the chip is full of other steps' bands"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, ops
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_mlp_block_gpu import _setup
from mlp_band_bench import timed, concurrent  # noqa
lib = _lib.load()
B = 16
sets = []
for k in range(4):
    x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, 22, 22, 3 + k)
    stg, taps = ops.mlp_band_packs(w1, b1, colsum, w2, wd, bd)
    sets.append((x, stg, taps, b2, stats.view(-1), torch.empty_like(x), torch.empty((B * 484, 2), device="cuda")))
def mk(k, shared):
    x, stg, taps, b2, st, out, ost = sets[k]
    if shared:
        stg, taps = sets[0][1], sets[0][2]
    return lambda: ops.mlp_band(x, stg, taps, b2, st, 1e-6, out, out_stats=ost)
for flags, what in ((0, "full"), (8, "no weight DMA"), (2 | 8, "MFMA phases only, no DMA"), (1 | 4 | 8, "depthwise only, no DMA"),
                    (1 | 2 | 4 | 16, "DMA + barriers"), (1 | 2 | 4 | 8 | 16, "barriers only")):
    lib.emip_debug_set_md(flags)
    row = []
    for shared in (False, True):
        gs = [timed(mk(k, shared))[1] for k in range(4)]
        row.append([concurrent(gs[:n]) for n in (1, 2, 4)])
    print("%-28s different weights 1/2/4 at once: %6.1f %6.1f %6.1f us   same weights: %6.1f %6.1f %6.1f us" % (what, *row[0], *row[1]), flush=True)
lib.emip_debug_set_md(0)
