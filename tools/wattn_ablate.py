#!/usr/bin/env python3
"""phase ablation of emip_window_attention (EMIP_HIP_LIB=.../libemip_hip_tuning.so)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, ops
from emip_amd.model.EMIP_short.motion.gmflow.tables import window_tables
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mlp_block_bench import timed
lib = _lib.load()
B2 = 32; h = w = 44; n = h * w; C = 128
big = (torch.randn(B2, n, 5 * C, device="cuda") * 1.5).to(torch.bfloat16)
q, k, v = big[..., :C], big[..., 3 * C:4 * C], big[..., 4 * C:]
out = torch.empty((B2, n, C), dtype=torch.bfloat16, device="cuda")
rows, gid = window_tables(h, w, 2, False, big.device)
new = lambda: ops.window_attention(q, k, v, out, rows, None, n, C ** -0.5, 16)
for flags, what in ((0, "full"), (1, "no S MFMA"), (2, "no softmax"), (4, "no PV"), (8, "no DMA"), (1 | 2, "no S, no softmax"),
                    (1 | 2 | 4, "barriers + DMA"), (1 | 2 | 4 | 8, "barriers only"), (2 | 4, "S only"), (1 | 2 | 8, "PV only")):
    lib.emip_debug_set_wa(flags)
    print("%-22s %7.1f us" % (what, timed(new)))
lib.emip_debug_set_wa(0)
