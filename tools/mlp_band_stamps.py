#!/usr/bin/env python3
"""in-kernel cycle stamps of emip_mlp_band (tuning library): where a wave's iterations go"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, ops
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_mlp_block_gpu import _setup
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, 22, 22, 3)
stg, taps = ops.mlp_band_packs(w1, b1, colsum, w2, wd, bd)
out = torch.empty_like(x)
st = stats.view(-1)
prof = torch.zeros((B * 4 * 8, 6), dtype=torch.int64, device="cuda")
for _ in range(3):
    ops.mlp_band(x, stg, taps, b2, st, 1e-6, out)
lib.emip_debug_set_md_prof(prof.data_ptr())
ops.mlp_band(x, stg, taps, b2, st, 1e-6, out)
torch.cuda.synchronize()
lib.emip_debug_set_md_prof(None)
p = prof.view(B * 4, 8, 6).double()
print("cycles per iteration (42 iterations), mean over workgroups; s_memtime ticks (100 MHz?) -- ratios matter")
print("wave   wait    issue   fc1     fc2     dw      total/42")
for w in range(8):
    m = p[:, w].mean(0) / 42
    print("%d   %7.1f %7.1f %7.1f %7.1f %7.1f %8.1f" % (w, *m.tolist()))
print("band 0 vs 1 (total):", p[0::4, :, 5].mean().item(), p[1::4, :, 5].mean().item())
