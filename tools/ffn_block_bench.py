#!/usr/bin/env python3
"""emip_ffn_block against the two launches it replaces (GMFlow FFN at 32 frames: 61 952 tokens), hipGraph of 10"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops
from tools.mlp_block_bench import timed

for M in (61952, 30976):
    bf = torch.bfloat16
    x1 = torch.randn(M, 128, device="cuda").to(bf); x2 = torch.randn(M, 128, device="cuda").to(bf)
    w0 = torch.randn(1024, 256, device="cuda") / 16; w2 = torch.randn(128, 1024, device="cuda") / 32
    gamma = torch.ones(128, device="cuda"); beta = torch.zeros(128, device="cuda")
    w0b, w2b = w0.to(bf).contiguous(), w2.to(bf).contiguous()
    p0, p2 = ops.ffn_block_packs(w0, w2)
    out = torch.empty_like(x1)
    def two():
        h = ops.gemm(x1, w0b, a2=x2, act=ops.ACT_GELU)
        return ops.gemm_ln_out(h, w2b, gamma, beta, 1e-5, res=x1, out=out)
    def one():
        return ops.ffn_block(x1, x2, p0, p2, gamma, beta, 1e-5, res=x1, out=out)
    ta, tb = timed(two, 10), timed(one, 10)
    gf = 2.0 * M * (1024 * 256 + 128 * 1024) / 1e9
    print("M=%6d: two launches %6.1f us (%4.0f TF/s)   emip_ffn_block %6.1f us (%4.0f TF/s)" % (M, ta, gf / ta * 1e3, tb, gf / tb * 1e3), flush=True)
