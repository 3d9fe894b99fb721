#!/usr/bin/env python3
"""Throughput of the 16-pair bf16 forward under different replay arrangements (one process, interleaved rounds):
  A  one step = two 8-pair graphs on two streams (bench.py's arrangement: 16 pairs in flight)
  B  one step = ONE 16-pair graph; consecutive steps alternate between two streams (32 pairs in flight)
  C  the same over three streams (48 pairs in flight)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.graph import GraphedShort
from emip_amd.model.EMIP_short.model import CoUpdater

g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(16, seed=1234)
im1, im2 = im1.cuda(), im2.cuda()
A = GraphedShort(net, 16, splits=2); A.load(im1, im2)
whole = [GraphedShort(net, 16, splits=1) for _ in range(3)]
for r in whole:
    r.load(im1, im2)
torch.cuda.synchronize()

def run_A(steps):
    for _ in range(steps):
        A.replay_free()

def run_multi(k):
    def f(steps):
        for i in range(steps):
            whole[i % k].replay_free()
    return f

def rate(fn, steps=24):
    fn(6); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(steps); torch.cuda.synchronize()
    return 16 * steps / (time.perf_counter() - t0)

res = {"A 2x8": [], "B 2x16": [], "C 3x16": [], "D 1x16": []}
for rnd in range(4):
    res["A 2x8"].append(rate(run_A))
    res["B 2x16"].append(rate(run_multi(2)))
    res["C 3x16"].append(rate(run_multi(3)))
    res["D 1x16"].append(rate(run_multi(1)))
for k, v in res.items():
    print("%-8s pairs/s: %s   median %.1f" % (k, " ".join("%.1f" % x for x in v), sorted(v)[len(v) // 2]))
