#!/usr/bin/env python3
"""emip_match against the generic attention launches it replaces (8 pairs: Z = 16), replayed from a hipGraph of 20."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops
from emip_amd.model.EMIP_short.motion.gmflow.tables import grid_values

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h = w = 44; n = h * w; C = 128
torch.manual_seed(0)
c0 = torch.randn(2 * B, n, C, device="cuda").to(torch.bfloat16)
corr = torch.empty((B, n, n), dtype=torch.bfloat16, device="cuda")
flow = torch.randn(2 * B, n, 2, device="cuda") * 20
grid = grid_values(h, w, torch.bfloat16, c0.device)
o = torch.empty((2 * B, n, 32), dtype=torch.float32, device="cuda")
common = dict(batch=B, heads=1, nwin=1, Lq=n, Lk=n, D=C, DV=32, q_bs=n * C, k_bs=n * C, v_bs=0, o_bs=n * 32, ldq=C, ldk=C,
              ldv=32, ldo=32, scale=C ** -0.5)

def old_match():
    ops.attention(c0[:B], c0[B:], grid, o[:B], scores=corr, s_bs=n * n, lds=n, **common)
    ops.attention(c0[B:], c0[:B], grid, o[B:], **common)
    return ops.corresp_to_flow(o, 2 * B, h, w, True)

def new_match():
    return ops.match(c0, c0, w, C ** -0.5, scores=corr, kv_rot=B)

def new_fwd_only():
    return ops.match(c0[:B], c0[B:], w, C ** -0.5, scores=corr)

def new_noscores():
    return ops.match(c0, c0, w, C ** -0.5, kv_rot=B)

def new_prop():
    return ops.match(c0, c0, w, C ** -0.5, v=flow, sub_grid=False)

def old_prop():
    v = torch.empty((2 * B, n, 32), dtype=torch.bfloat16, device="cuda")
    ops.copy_cols(flow.view(2 * B * n, 2), 0, 2, v.view(2 * B * n, 32), 0, 32)
    oo = torch.empty((2 * B, n, 32), dtype=torch.float32, device="cuda")
    ops.attention(c0, c0, v, oo, batch=2 * B, heads=1, nwin=1, Lq=n, Lk=n, D=C, DV=32, q_bs=n * C, k_bs=n * C, v_bs=n * 32,
                  o_bs=n * 32, ldq=C, ldk=C, ldv=32, ldo=32, scale=C ** -0.5)
    return ops.corresp_to_flow(oo, 2 * B, h, w, False)

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

byt = B * n * n * 2 + 2 * B * n * C * 2
for name, fn in (("old matching (2 launches + flow)", old_match), ("emip_match both directions + scores", new_match),
                 ("emip_match forward only + scores", new_fwd_only), ("emip_match both directions, no scores", new_noscores),
                 ("old propagation (copy + attention + flow)", old_prop), ("emip_match propagation", new_prop)):
    us = timed(fn)
    print("%-46s %8.1f us   (score volume + features %.0f MB -> %.2f TB/s)" % (name, us, byt / 1e6, byt / us / 1e6))
