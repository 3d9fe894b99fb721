#!/usr/bin/env python3
"""one step at a time: the linear graph against the forked one (model.FORK_DEEP), the fork stream at normal / high priority"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.graph import _Part
from emip_amd.model.EMIP_short import model as M
from emip_amd.model.EMIP_short.model import CoUpdater
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(16, seed=1234)
parts = []
for name, fork, prio, cf in (("linear", False, 0, False), ("forked", True, 0, False), ("forked, CNN on a third branch", True, 0, "f2")):
    M.FORK_PRIORITY = prio
    M.FORK_CNN = cf == 'f2'
    cf = False
    if hasattr(net, "_fork"):
        object.__delattr__(net, "_fork")
    p = _Part(net, 16, 352, "cuda:0", 1, cnn_first=cf, fork_deep=fork)
    p.im1.copy_(im1.cuda()); p.im2.copy_(im2.cuda())
    parts.append((name, p))
torch.cuda.synchronize()
for rnd in range(3):
    for name, p in parts:
        ts = []
        for _ in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p.graph.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%-24s %.3f ms" % (name, sorted(ts)[len(ts) // 2]), flush=True)
ref = parts[0][1].mask
for name, p in parts[1:]:
    print(name, "== linear:", torch.equal(p.mask, ref))
