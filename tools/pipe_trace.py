#!/usr/bin/env python3
"""a short pipelined run (whole-batch graphs, N steps in flight) for `rocprofv3 --kernel-trace`: the dispatch timestamps show
how many kernels run at once, how much a kernel stretches under load and where the device idles"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.graph import PipelinedShort
from emip_amd.model.EMIP_short.model import CoUpdater
import importlib
for kv in os.environ.get("EMIP_DBG", "").split(","):      # e.g. EMIP_DBG="emip_amd.lib.pvt_v2:MLP_BAND=False,emip_amd.ops:MLP_BAND_BANDS=8"
    if "=" in kv:
        k, v = kv.split("=")
        mn, k = k.split(":")
        mod = importlib.import_module(mn)
        assert hasattr(mod, k), (mn, k)
        setattr(mod, k, eval(v))
inflight = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(16, seed=1234)
r = PipelinedShort(net, 16, inflight=inflight); r.load(im1.cuda(), im2.cuda())
torch.cuda.synchronize()
for _ in range(4 * inflight):
    r.replay_free()
torch.cuda.synchronize()
