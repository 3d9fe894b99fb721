#!/usr/bin/env python3
"""GMFlow global matching (gmflow/matching.py:8-41) as bench.py's forward launches it: softmax(F0 F1^T / sqrt(128)) times the
pixel grid with the raw correlation written as [src][tgt], 16 pairs, 1936 x 1936, bf16.  us per launch from a hipGraph of 10
launches; algorithmic bytes = correlation out + features in; checks the scores against torch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, ops

_lib.load()
torch.manual_seed(0)
B, n, C = 16, 1936, 128
f0 = torch.randn(B, n, C, device="cuda").to(torch.bfloat16)
f1 = torch.randn(B, n, C, device="cuda").to(torch.bfloat16)
grid = torch.zeros(B, n, 32, device="cuda", dtype=torch.bfloat16)
grid[..., 0] = (torch.arange(n, device="cuda") % 44).to(torch.bfloat16)
grid[..., 1] = (torch.arange(n, device="cuda") // 44).to(torch.bfloat16)
out = torch.empty(B, n, 32, device="cuda", dtype=torch.float32)
corr = torch.empty(B, n, n, device="cuda", dtype=torch.bfloat16)
for name, scores, ks in (("scores", corr, 1), ("scores ksplit 2", corr, 2), ("scores ksplit 4", corr, 4), ("no scores", None, 1), ("no scores ksplit 2", None, 2), ("no scores ksplit 4", None, 4)):
    def run():
        ops.attention(f0, f1, grid, out, batch=B, heads=1, nwin=1, Lq=n, Lk=n, D=C, DV=32, q_bs=n * C, k_bs=n * C, v_bs=n * 32,
                      o_bs=n * 32, ldq=C, ldk=C, ldv=32, ldo=32, scale=C ** -0.5, scores=scores, s_bs=n * n, lds=n, ksplit=ks)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = 2.0 * B * n * n * (C + 2)
    byt = 2.0 * B * (2 * n * C + n * 2) + (2.0 * B * n * n if scores is not None else 0)
    print("%-20s %7.1f us  %6.1f TFLOP/s  %6.2f TB/s algorithmic" % (name, us, fl / us / 1e6, byt / us / 1e6))
ref = torch.einsum("bqc,bkc->bqk", f0[:2].float(), f1[:2].float()) * C ** -0.5
ops.attention(f0, f1, grid, out, batch=B, heads=1, nwin=1, Lq=n, Lk=n, D=C, DV=32, q_bs=n * C, k_bs=n * C, v_bs=n * 32,
              o_bs=n * 32, ldq=C, ldk=C, ldv=32, ldo=32, scale=C ** -0.5, scores=corr, s_bs=n * n, lds=n, ksplit=1)
print("scores max |err| vs torch f32: %.3e (max |ref| %.2f)" % ((corr[:2].float() - ref).abs().max().item(), ref.abs().max().item()))
p = torch.softmax(ref, -1)
exp = torch.einsum("bqk,bkc->bqc", p, grid[:2, :, :2].float())
print("expected-coordinate max |err|: %.3e" % (out[:2, :, :2] - exp).abs().max().item())
