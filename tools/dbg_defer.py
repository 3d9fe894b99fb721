#!/usr/bin/env python3
"""ONE bf16 training forward, then its backward three times over the retained graph (immediate, immediate, deferred weight
gradients): with the forward fixed, how far apart are the parameter gradients?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from emip_amd import nn_base, ops
from emip_amd.filler import state_dict_from_manifest, synthetic_pair, synthetic_gt
from emip_amd.loss.loss_flow import unFlowLoss
from emip_amd.loss.loss_pred import hybrid_e_loss
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import freeze_like_reference

g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
B = 2
net = CoUpdater(margs); net.load_state_dict(sd)
net = freeze_like_reference(net.to("cuda:0").train())
for m in net.modules():
    if hasattr(m, "drop_path_rate"):
        m.drop_path_rate = 0.0
im1, im2 = synthetic_pair(B, seed=7)
gt = synthetic_gt(B, seed=7).cuda()
im1, im2 = im1.cuda(), im2.cuda()
fl = unFlowLoss()
for arena in (False, True, True):
    if arena:
        ops.ARENA.begin(im1.device)
    with torch.enable_grad():
        preds = net(im1, im2)
        pair = [torch.cat((preds[1][i], preds[2][i]), 1) for i in range(len(preds[1]))]
        loss = hybrid_e_loss(preds[0], gt) + fl.compute_loss(pair, torch.cat((im1, im2), 1))[0]
        runs = []
        for defer in (False, False, True):
            net.zero_grad(set_to_none=True)
            ops.WGRADS.enabled = defer
            loss.backward(retain_graph=True)
            ops.flush_wgrads()
            ops.WGRADS.fixup()
            runs.append({n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None})
    ops.ARENA.end()
    def rel(a, b):
        return sorted((((a[n] - b[n]).abs().max() / (b[n].abs().max() + 1e-30)).item(), n) for n in b)[::-1]
    print("arena", arena, "fixed", ops.WGRADS.fixed)
    print("  immediate again vs immediate:", rel(runs[1], runs[0])[:4])
    print("  deferred        vs immediate:", rel(runs[2], runs[0])[:4])
