#!/usr/bin/env python3
"""arena-vs-plain gradient comparison in f32 mode: where do the deviating elements sit?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base, ops
from emip_amd.filler import state_dict_from_manifest, synthetic_gt, synthetic_pair
from emip_amd.loss.loss_pred import hybrid_e_loss
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import freeze_like_reference
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.float32)
net = CoUpdater(margs); net.load_state_dict(sd)
net = freeze_like_reference(net.to("cuda:0").train())
for m in net.modules():
    if hasattr(m, "drop_path_rate"):
        m.drop_path_rate = 0.0
im1, im2 = synthetic_pair(1, seed=5)
gt = synthetic_gt(1, seed=5).cuda()
im1, im2 = im1.cuda(), im2.cuda()

def grads(use):
    net.zero_grad(set_to_none=True)
    if use:
        ops.ARENA.begin(im1.device)
    try:
        with torch.enable_grad():
            hybrid_e_loss(net(im1, im2)[0], gt).backward()
    finally:
        ops.ARENA.end()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}

ref, ref2 = grads(False), grads(False)
grads(True)
for rep in range(3):
    got = grads(True)
    bad = []
    for n in ref:
        d = (got[n] - ref[n]).abs()
        sc = ref[n].abs().max().item() + 1e-30
        nz = (ref2[n] - ref[n]).abs().max().item() / sc
        if d.max().item() / sc > 2e-3:
            idx = (d / sc > 1e-3).flatten().nonzero().flatten()
            bad.append((d.max().item() / sc, nz, n, tuple(ref[n].shape), idx.numel(), idx[:6].tolist(),
                        got[n].flatten()[idx[:3]].tolist(), ref[n].flatten()[idx[:3]].tolist()))
    print("arena pass", rep, "deviating parameters:", len(bad))
    for b in sorted(bad, reverse=True)[:12]:
        print("   ", b)
