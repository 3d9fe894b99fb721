#!/usr/bin/env python3
"""Soak of train.GraphedTrainStep (batch 4, bf16, DropPath on): N steps of graph replays against N eager steps from the same
weights -- loss statistics of the two runs, parameter drift between them against the drift between two eager runs, and the
device memory in use at the start and the end of the steps (a captured step allocates nothing; the first optimizer step creates
the two AdamW moments, 8 bytes per trainable parameter)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_gt, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import GraphedTrainStep, build_optimizer, freeze_like_reference, train_step

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = 4
_lib.load()
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
def make():
    net = CoUpdater(margs); net.load_state_dict(sd)
    net = freeze_like_reference(net.to("cuda:0").train())
    return net, build_optimizer(net)
batches = [(synthetic_pair(B, seed=100 + i), synthetic_gt(B, seed=200 + i)) for i in range(6)]
batches = [(a.cuda(), b.cuda(), c.cuda()) for (a, b), c in batches]
runs = {}
for name in ("eager", "eager2", "graph"):
    torch.manual_seed(0)
    net, opt = make()
    gs = GraphedTrainStep(net, opt, *batches[0]) if name == "graph" else None
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    losses = []
    for i in range(N):
        im1, im2, gt = batches[i % len(batches)]
        l = gs.step(im1, im2, gt) if gs else train_step(net, opt, None, im1, im2, gt)
        losses.append(float(l[0]))
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_allocated()
    runs[name] = ({n: p.detach().clone() for n, p in net.named_parameters() if p.requires_grad}, losses)
    print("%-7s loss first 5 %s  last 5 %s  mean of last 20 %.4f  finite %s  memory %.1f -> %.1f MiB" % (
        name, ["%.3f" % x for x in losses[:5]], ["%.3f" % x for x in losses[-5:]], sum(losses[-20:]) / 20,
        all(x == x for x in losses), m0 / 2 ** 20, m1 / 2 ** 20), flush=True)
    del net, opt, gs
def drift(a, b):
    return max((a[n] - b[n]).abs().max().item() for n in a)
pe, pg, p2 = runs["eager"][0], runs["graph"][0], runs["eager2"][0]
print("largest parameter difference after %d steps: eager vs graph %.3e, eager vs eager %.3e" % (N, drift(pe, pg), drift(pe, p2)))
