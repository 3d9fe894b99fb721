#!/usr/bin/env python3
"""steps per second of the pipelined replay against the batch size: a launch-rate bound shows as a constant time per step"""
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
inflight = int(os.environ.get("INFLIGHT", "4"))
for B in (2, 4, 8, 16, 32):
    im1, im2 = synthetic_pair(B, seed=1234)
    r = PipelinedShort(net, B, inflight=inflight); r.load(im1.cuda(), im2.cuda())
    torch.cuda.synchronize()
    for _ in range(8):
        r.replay_free()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); steps = 40
    for _ in range(steps):
        r.replay_free()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("B=%2d pairs per step: %.2f ms per step, %.0f pairs/s" % (B, dt * 1e3, B / dt), flush=True)
    del r
    torch.cuda.empty_cache()
