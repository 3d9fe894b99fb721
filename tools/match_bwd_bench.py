#!/usr/bin/env python3
"""emip_match_bwd at the training step's shape (32 pairs: Z = 64, n = 1936), with and without the upstream score gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops
from tools.mlp_block_bench import timed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n, W, C = 1936, 44, 128
q = (torch.randn(2 * B, n, C, device="cuda") * 0.9).to(torch.bfloat16)
do = torch.randn(2 * B, n, 2, device="cuda")
ds = (torch.randn(B, n, n, device="cuda") * 0.05).to(torch.bfloat16)
lse = torch.empty(2 * B, n, device="cuda")
corr = torch.empty(B, n, n, dtype=torch.bfloat16, device="cuda")
out = ops.match(q, q, W, C ** -0.5, scores=corr, kv_rot=B, sub_grid=False, lse=lse)
for name, fn in (("forward (both directions + volume + lse)", lambda: ops.match(q, q, W, C ** -0.5, scores=corr, kv_rot=B, sub_grid=False, lse=lse)),
                 ("backward with the volume's gradient", lambda: ops.match_bwd(q, q, W, C ** -0.5, out, do, lse, dscores=ds, kv_rot=B, sub_grid=False, accum=True)),
                 ("backward without", lambda: ops.match_bwd(q, q, W, C ** -0.5, out, do, lse, kv_rot=B, sub_grid=False, accum=True))):
    print("%-45s %7.1f us" % (name, timed(fn, 5)), flush=True)
