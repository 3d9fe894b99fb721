#!/usr/bin/env python3
"""emip_window_attention against the generic attention launch it replaces (32 frames = 16 pairs), hipGraph of 20."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops
from emip_amd.model.EMIP_short.motion.gmflow.tables import window_tables
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mlp_block_bench import timed

for B2 in (32, 16):
    h = w = 44; n = h * w; C = 128
    big = (torch.randn(B2, n, 5 * C, device="cuda") * 1.5).to(torch.bfloat16)
    q, k, v = big[..., :C], big[..., 3 * C:4 * C], big[..., 4 * C:]
    out = torch.empty((B2, n, C), dtype=torch.bfloat16, device="cuda")
    for shift in (False, True):
        rows, gid = window_tables(h, w, 2, shift, big.device)
        L = rows.shape[1]
        g = gid if shift else None
        new = lambda: ops.window_attention(q, k, v, out, rows, g, n, C ** -0.5, B2 // 2)
        old = lambda: ops.attention(q, k, v, out, batch=B2, heads=1, nwin=4, Lq=L, Lk=L, D=C, DV=C, q_bs=n * 5 * C, k_bs=n * 5 * C,
                                    v_bs=n * 5 * C, o_bs=n * C, ldq=5 * C, ldk=5 * C, ldv=5 * C, ldo=C, q_rows=rows, k_rows=rows,
                                    q_gid=g, k_gid=g, scale=C ** -0.5, kv_rot=B2 // 2)
        gf = 4 * B2 * 4 * L * L * C / 1e9
        a, b = timed(old), timed(new)
        print("B2=%2d shift=%-5s generic %6.1f us (%4.0f TF/s)   emip_window_attention %6.1f us (%4.0f TF/s)" % (B2, shift, a, gf / a * 1e3, b, gf / b * 1e3))
