#!/usr/bin/env python3
"""is the pipelined replay loop bound by the host?  time of the enqueue loop alone against the wall time to completion"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.graph import PipelinedShort
from emip_amd.model.EMIP_short.model import CoUpdater
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(16, seed=1234)
im1, im2 = im1.cuda(), im2.cuda()
for inflight in (1, 2, 4):
    r = PipelinedShort(net, 16, inflight=inflight); r.load(im1, im2)
    torch.cuda.synchronize()
    for _ in range(6):
        r.replay_free()
    torch.cuda.synchronize()
    steps = 40
    t0 = time.perf_counter()
    for _ in range(steps):
        r.replay_free()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("inflight %d: enqueue loop %.2f ms per step, wall %.2f ms per step (%.0f pairs/s)" % (
        inflight, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3, 16 * steps / (t2 - t0)), flush=True)
    del r
