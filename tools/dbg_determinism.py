#!/usr/bin/env python3
"""run every fused backward kernel several times on identical inputs: bitwise identical outputs?  (a race on an LDS ring slot or a
short wait would show up here as sporadic differences)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import ops
from emip_amd.model.EMIP_short.motion.gmflow.tables import window_tables

torch.manual_seed(0)
bf = torch.bfloat16
def same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))

# window attention backward (64 frames as in training)
B2, h, w, C = 64, 44, 44, 128
n = h * w
q, k, v, do = [(torch.randn(B2, n, C, device="cuda") * 1.2).to(bf) for _ in range(4)]
for shift, rot in ((False, 0), (True, 32)):
    rows, gid = window_tables(h, w, 2, shift, q.device)
    gm = gid if shift else None
    out = torch.empty_like(q); lse = torch.empty(B2, n, device="cuda")
    ops.window_attention(q, k, v, out, rows, gm, n, C ** -0.5, rot, lse=lse)
    ref = [t.clone() for t in ops.window_attention_bwd(q, k, v, out, do, lse, rows, gm, n, C ** -0.5, rot)]
    bad = sum(0 if same(ref, ops.window_attention_bwd(q, k, v, out, do, lse, rows, gm, n, C ** -0.5, rot)) else 1 for _ in range(20))
    o2 = torch.empty_like(q); l2 = torch.empty_like(lse)
    badf = 0
    for _ in range(20):
        ops.window_attention(q, k, v, o2, rows, gm, n, C ** -0.5, rot, lse=l2)
        badf += 0 if (torch.equal(o2, out) and torch.equal(l2, lse)) else 1
    print("window attention shift=%s rot=%d: forward differs in %d / 20 runs, backward in %d / 20" % (shift, rot, badf, bad), flush=True)

# matching backward
Z, n2, W = 64, 1936, 44
f = (torch.randn(Z, n2, C, device="cuda") * 0.9).to(bf)
dO = torch.randn(Z, n2, 2, device="cuda")
ds = (torch.randn(Z // 2, n2, n2, device="cuda") * 0.05).to(bf)
lse = torch.empty(Z, n2, device="cuda"); corr = torch.empty(Z // 2, n2, n2, dtype=bf, device="cuda")
o = ops.match(f, f, W, C ** -0.5, scores=corr, kv_rot=Z // 2, sub_grid=False, lse=lse)
for up in (True, False):
    ref = ops.match_bwd(f, f, W, C ** -0.5, o, dO, lse, dscores=ds if up else None, kv_rot=Z // 2, sub_grid=False, accum=True)[0].clone()
    bad = sum(0 if torch.equal(ref, ops.match_bwd(f, f, W, C ** -0.5, o, dO, lse, dscores=ds if up else None, kv_rot=Z // 2,
                                                    sub_grid=False, accum=True)[0]) else 1 for _ in range(20))
    print("matching backward upstream=%s: differs in %d / 20 runs" % (up, bad), flush=True)
badf = 0
o_ref, lse_ref, corr_ref = o.clone(), lse.clone(), corr.clone()
for _ in range(20):
    o2 = ops.match(f, f, W, C ** -0.5, scores=corr, kv_rot=Z // 2, sub_grid=False, lse=lse)
    badf += 0 if (torch.equal(o2, o_ref) and torch.equal(lse, lse_ref) and torch.equal(corr, corr_ref)) else 1
print("matching forward: differs in %d / 20 runs" % badf, flush=True)

# depthwise backward: dx must repeat; dW / db are f32 atomics (order-dependent)
x, z, dy = [torch.randn(64, 22, 22, 1280, device="cuda").to(bf) for _ in range(3)]
wt = torch.randn(9, 1280, device="cuda")
acc = torch.zeros(10 * 1280, device="cuda")
ref = ops.dwconv3x3_bwd_fused(x, z, dy, wt, acc[:9 * 1280], acc[9 * 1280:], True).clone()
bad = 0
for _ in range(20):
    a2 = torch.zeros_like(acc)
    bad += 0 if torch.equal(ref, ops.dwconv3x3_bwd_fused(x, z, dy, wt, a2[:9 * 1280], a2[9 * 1280:], True)) else 1
print("depthwise backward dx: differs in %d / 20 runs; dW relative spread %.2e" % (bad, ((a2 - acc).abs().max() / acc.abs().max()).item()), flush=True)

# SRA backward, bf16 dK | dV
B, heads, Lq, Lk = 64, 5, 484, 121
Cs = heads * 64
qs = torch.randn(B, Lq, Cs, device="cuda").to(bf); kvs = torch.randn(B, Lk, 2 * Cs, device="cuda").to(bf); dos = torch.randn(B, Lq, Cs, device="cuda").to(bf)
outs = torch.empty_like(qs)
L = ops.sra_attention_lse(qs, kvs, outs, B, heads, Lq, Lk, 0.125)
ref = [t.clone() for t in ops.sra_attention_bwd(qs, kvs, outs, dos, L, B, heads, Lq, Lk, 0.125)]
bad = sum(0 if same(ref, ops.sra_attention_bwd(qs, kvs, outs, dos, L, B, heads, Lq, Lk, 0.125)) else 1 for _ in range(20))
print("SRA backward (bf16 dK | dV): differs in %d / 20 runs" % bad, flush=True)
