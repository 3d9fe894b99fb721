#!/usr/bin/env python3
"""In-process A/B of switches on the TRAINING step (batch 32, bf16), interleaved rounds of timed steps.  A setting is a
comma-separated list of module:attr=value assignments applied together; attr may be dotted (Class.ATTR):
   python tools/train_flag_ab.py "emip_amd.ops:WgradQueue.SIDE=False" "emip_amd.ops:WgradQueue.SIDE=True,emip_amd.ops:WgradQueue.MAX=40"
The switches must be read at step time (not baked in at construction)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_gt, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import build_optimizer, freeze_like_reference, train_step

B = int(os.environ.get("PAIRS", "32"))
settings = []
for arg in sys.argv[1:]:
    one = []
    for part in arg.split(","):
        target, val = part.split("=")
        modname, attr = target.split(":")
        one.append((importlib.import_module(modname), attr.split("."), eval(val)))
    settings.append(one)

def apply(one):
    for mod, path, val in one:
        obj = mod
        for name in path[:-1]:
            obj = getattr(obj, name)
        setattr(obj, path[-1], val)

_lib.load()
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd)
net = freeze_like_reference(net.to("cuda:0").train())
opt = build_optimizer(net)
im1, im2 = synthetic_pair(B, seed=1234)
gt = synthetic_gt(B, seed=99)
im1, im2, gt = im1.cuda(), im2.cuda(), gt.cuda()

def run(n):
    loss = None
    for _ in range(n):
        loss = train_step(net, opt, None, im1, im2, gt)
    torch.cuda.synchronize()
    return loss

apply(settings[0])
run(3)
res = [[] for _ in settings]
last = [None for _ in settings]
for rnd in range(4):
    for i, one in enumerate(settings):
        apply(one)
        run(2)
        t0 = time.perf_counter()
        loss = run(6)
        res[i].append((time.perf_counter() - t0) / 6 * 1e3)
        last[i] = [round(float(x), 4) for x in loss]
for arg, xs, l in zip(sys.argv[1:], res, last):
    print("%-90s ms/step: %s   median %.2f   loss %s" % (arg, " ".join("%.2f" % x for x in xs), sorted(xs)[len(xs) // 2], l), flush=True)
