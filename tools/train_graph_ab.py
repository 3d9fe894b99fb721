#!/usr/bin/env python3
"""The training step at batch 32 (bf16): eager train_step against train.GraphedTrainStep (forward + backward + weight gradients
as one hipGraph, optimizer eager), each with model.FORK_DEEP_TRAIN off / on.  Interleaved rounds in one process."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import emip_amd.model.EMIP_short.model as M
from emip_amd import _lib, nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_gt, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import GraphedTrainStep, build_optimizer, freeze_like_reference, train_step

B = int(os.environ.get("PAIRS", "32"))
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

def eager(fork):
    def run(n):
        M.FORK_DEEP_TRAIN = fork
        for _ in range(n):
            loss = train_step(net, opt, None, im1, im2, gt)
        return loss
    return run

def graphed(fork):
    M.FORK_DEEP_TRAIN = fork
    t0 = time.perf_counter()
    gs = GraphedTrainStep(net, opt, im1, im2, gt)
    torch.cuda.synchronize()
    print("capture (fork %s): %.1f s, pool %.1f GiB" % (fork, time.perf_counter() - t0, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
    def run(n):
        for _ in range(n):
            loss = gs.step()
        return loss
    return run

eager(False)(3)
torch.cuda.synchronize()
modes = [("eager", eager(False)), ("eager + fork", eager(True)), ("graph", graphed(False)), ("graph + fork", graphed(True))]
if len(sys.argv) > 1:
    # python tools/train_graph_ab.py "module:ATTR=value,module:ATTR2=value" ...: one captured forked step per setting instead
    import importlib
    modes = []
    for arg in sys.argv[1:]:
        for part in arg.split(","):
            target, val = part.split("=")
            modname, attr = target.split(":")
            obj = importlib.import_module(modname)
            path = attr.split(".")
            for name in path[:-1]:
                obj = getattr(obj, name)
            setattr(obj, path[-1], eval(val))
        modes.append((arg.split(":")[-1][:40], graphed(True)))
res = {k: [] for k, _ in modes}
last = {}
for rnd in range(4):
    for k, fn in modes:
        fn(2); torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = fn(8)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 8 * 1e3)
        last[k] = [round(float(x), 4) for x in loss]
for k, _ in modes:
    xs = res[k]
    print("%-42s ms/step: %s   median %.2f  (%.1f pairs/s)  loss %s" % (k, " ".join("%.2f" % x for x in xs), sorted(xs)[len(xs) // 2],
                                                                      B / sorted(xs)[len(xs) // 2] * 1e3, last[k]), flush=True)
