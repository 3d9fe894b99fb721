#!/usr/bin/env python3
"""Per-call listing of one eager EMIP-short forward (HIP-event pair around every C-ABI call, host running ahead behind a
blocker GEMM): index, entry point, kernel symbol, a shape digest, microseconds.  --by-phase sums by the Python call site
(module file:function of the first frame under emip_amd/ that is not ops.py)."""
import argparse, json, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT)
import torch
import bench as B
from emip_amd import _lib, nn_base, ops
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=16)
ap.add_argument("--top", type=int, default=0)
args = ap.parse_args()
lib = _lib.load()
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(args.pairs, seed=1234)
im1, im2 = im1.cuda(), im2.cuda()
sites = []
orig_call = _lib.call
def traced(name, *a):
    fr = None
    for f in traceback.extract_stack()[:-1][::-1]:
        if "/emip_amd/" in f.filename and not f.filename.endswith(("ops.py", "_lib.py", "autograd.py")):
            fr = "%s:%s:%d" % (os.path.basename(f.filename), f.name, f.lineno); break
    sites.append(fr)
    return orig_call(name, *a)
with torch.no_grad():
    for _ in range(2):
        net.run(im1, im2)
    torch.cuda.synchronize()
    ba = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16); bo = torch.empty_like(ba)
    for _ in range(80):
        ops.gemm(ba, ba, out=bo)
    rec = []
    _lib.call = traced
    for m in list(sys.modules.values()):
        if m is not None and getattr(m, "__name__", "").startswith("emip_amd") and getattr(m, "call", None) is orig_call:
            m.call = traced
    _lib.profile(rec)
    net.run(im1, im2)
    _lib.profile(None)
    torch.cuda.synchronize()
tot = 0.0
phase = {}
rows = []
for i, (name, a, s, e) in enumerate(rec):
    us = s.elapsed_time(e) * 1e3
    info = B._launch_info(lib, name, a)
    key = info[1] if info else name
    gf = info[0] / 1e9 if info else 0.0
    dims = [x for x in a if isinstance(x, int) and 0 < x < 10**7][:9]
    rows.append((i, name, key, gf, us, sites[i] if i < len(sites) else None, dims))
    tot += us
    p = (sites[i] or "?").rsplit(":", 1)[0]
    d = phase.setdefault(p, [0.0, 0, 0.0]); d[0] += us; d[1] += 1; d[2] += gf
print("calls %d, sum %.1f us" % (len(rec), tot))
for r in rows:
    print("%4d %-26s %-52s %8.2f GF %8.1f us  %-40s %s" % r)
print("---- by call site")
for p, d in sorted(phase.items(), key=lambda kv: -kv[1][0]):
    print("%-50s %9.1f us %5d calls %9.1f GF  %6.1f TFLOP/s" % (p, d[0], d[1], d[2], d[2] / d[0] * 1e3 if d[0] else 0))
