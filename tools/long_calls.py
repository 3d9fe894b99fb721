#!/usr/bin/env python3
"""Per-call-site listing of one steady-state EMIP-long step (8 streams, window full): like tools/fwd_calls.py."""
import json, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from emip_amd import _lib, nn_base, ops
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.model.EMIP_long.model_long import Model_long

lib = _lib.load()
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "long_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = Model_long(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
S = 8
f0, f1 = synthetic_pair(S, seed=1234); f0, f1 = f0.cuda(), f1.cuda()
k = v = None
with torch.no_grad():
    for i in range(Model_long.WINDOW + 2):
        _, k, v = net.forward_streams(f0, f1, i, k, v)
    mk, mv = net._lookup(k, v)
    torch.cuda.synchronize()
    sites = []
    orig = _lib.call
    def traced(name, *a):
        fr = None
        for f in traceback.extract_stack()[:-1][::-1]:
            if "/emip_amd/" in f.filename and not f.filename.endswith(("ops.py", "_lib.py", "autograd.py")):
                fr = "%s:%s" % (os.path.basename(f.filename), f.name); break
        sites.append(fr)
        return orig(name, *a)
    ba = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16); bo = torch.empty_like(ba)
    for _ in range(60):
        ops.gemm(ba, ba, out=bo)
    rec = []
    _lib.call = traced
    _lib.profile(rec)
    net.step_cl(f0, f1, mk, mv)
    _lib.profile(None)
    torch.cuda.synchronize()
phase = {}
tot = 0.0
for i, (name, a, s, e) in enumerate(rec):
    us = s.elapsed_time(e) * 1e3
    tot += us
    d = phase.setdefault((sites[i], name), [0.0, 0]); d[0] += us; d[1] += 1
print("calls %d, sum %.1f us" % (len(rec), tot))
for (p, n), d in sorted(phase.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%-40s %-26s %9.1f us %4d calls" % (p, n, d[0], d[1]))
