#!/usr/bin/env python3
"""conv_corr.0 + BN + ReLU at B pairs: the direct 3 x 3 conv over the 1936-channel correlation volume (incl. writing the volume in
emip_match) against the factored form (transpose, G = W' F1, patch matrix, per-image GEMM); also the two outputs' difference."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base, ops
from emip_amd.filler import state_dict_from_manifest
from emip_amd.model.EMIP_short.model import CoUpdater

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
torch.manual_seed(0)
tok = (torch.randn(2 * B, 1936, 128, device="cuda") * 0.5).to(torch.bfloat16)
corr = torch.empty((B, 1936, 1936), dtype=torch.bfloat16, device="cuda")

def direct():
    ops.match(tok[:B], tok[B:], 44, 128 ** -0.5, scores=corr)
    return net.run_conv_corr(corr)
def direct_conv_only():
    return net.run_conv_corr(corr)
def match_only():
    ops.match(tok[:B], tok[B:], 44, 128 ** -0.5)
def factored():
    ops.match(tok[:B], tok[B:], 44, 128 ** -0.5)
    return net.run_conv_corr_factored(tok, B, 44, 44)

def timed(fn, reps=5):
    with torch.no_grad():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                fn()
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / reps * 1e3)
    return best
with torch.no_grad():
    a, b = direct().float(), factored().float()
print("B = %d: |direct - factored| max %.4f, rel to max %.2e (bf16 mode)" % (B, (a - b).abs().max().item(), ((a - b).abs().max() / a.abs().max()).item()))
print("match alone %.1f us | match + volume + direct conv_corr %.1f us (conv_corr alone %.1f) | match + factored conv_corr %.1f us" % (
    timed(match_only), timed(direct), timed(direct_conv_only), timed(factored)))
