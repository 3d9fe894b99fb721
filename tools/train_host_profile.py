#!/usr/bin/env python3
"""Host-side cost of one training step (cProfile around 3 steps at batch 32, after warm-up): the step is host-bound once the
kernel time drops under ~130 ms, so this is where the remaining time is."""
import cProfile, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_gt, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import build_optimizer, freeze_like_reference, train_step

_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd)
net = freeze_like_reference(net.to("cuda:0").train())
opt = build_optimizer(net)
im1, im2 = synthetic_pair(B, seed=1234); gt = synthetic_gt(B, seed=99)
im1, im2, gt = im1.cuda(), im2.cuda(), gt.cuda()
for _ in range(3):
    train_step(net, opt, None, im1, im2, gt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    train_step(net, opt, None, im1, im2, gt)
th = time.perf_counter() - t0                 # host time to ENQUEUE three steps
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print("3 steps: host enqueue %.1f ms/step, wall %.1f ms/step" % (th / 3 * 1e3, tw / 3 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    train_step(net, opt, None, im1, im2, gt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
