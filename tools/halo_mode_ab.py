#!/usr/bin/env python3
"""(tuning library) emip_debug_set_halo modes under bench.py's arrangement; EMIP_HIP_LIB=emip_amd/libemip_hip_tuning.so python tools/halo_mode_ab.py 0 1 2
(copy of flag_ab.py:) In-process A/B of a module-level switch under bench.py's arrangement (three 16-pair graphs in flight), interleaved rounds:
   python tools/flag_ab.py emip_amd.lib.pvt_v2 MLP_BLOCK True False"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.graph import PipelinedShort
from emip_amd.model.EMIP_short.model import CoUpdater

from emip_amd import _lib
modname, attr = 'emip_amd.ops', 'HALO_MODE'
vals = [int(v) for v in sys.argv[1:]]
mod = importlib.import_module(modname)
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(16, seed=1234)
im1, im2 = im1.cuda(), im2.cuda()
runners = []
for v in vals:
    setattr(mod, attr, v)
    _lib.load().emip_debug_set_halo(v)
    r = PipelinedShort(net, 16, inflight=int(os.environ.get("INFLIGHT", "4"))); r.load(im1, im2)
    runners.append(r)
torch.cuda.synchronize()

def rate(r, steps=30):
    for _ in range(6):
        r.replay_free()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.replay_free()
    torch.cuda.synchronize()
    return 16 * steps / (time.perf_counter() - t0)

res = [[] for _ in vals]
for rnd in range(5):
    for i, r in enumerate(runners):
        res[i].append(rate(r))
for v, xs in zip(vals, res):
    print("%s.%s = %-8r pairs/s: %s   median %.1f" % (modname, attr, v, " ".join("%.1f" % x for x in xs), sorted(xs)[len(xs) // 2]))
