#!/usr/bin/env python3
"""Which launch first differs between two bf16 training steps of the same inputs?  Every emip_amd.ops function is wrapped to
checksum the tensors it returns; the traces of N steps are compared with the first one."""
import inspect, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from emip_amd import nn_base, ops
from emip_amd.filler import state_dict_from_manifest, synthetic_pair, synthetic_gt
from emip_amd.loss.loss_flow import unFlowLoss
from emip_amd.loss.loss_pred import hybrid_e_loss
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import freeze_like_reference

TRACE = None
def flat(v, out):
    if isinstance(v, torch.Tensor):
        out.append(v)
    elif isinstance(v, (tuple, list)):
        for x in v:
            flat(x, out)
def wrap(name, fn):
    def w(*a, **k):
        r = fn(*a, **k)
        if TRACE is not None:
            ts = []
            flat(r, ts)
            for i, t in enumerate(ts):
                if t.is_cuda and t.numel():
                    f = t.detach().double()
                    TRACE.append((name, i, tuple(t.shape), f.sum().item(), f.abs().sum().item()))
        return r
    return w
for n, f in list(vars(ops).items()):
    if inspect.isfunction(f) and f.__module__ == ops.__name__ and not n.startswith("_") and n not in ("grad_zeros", "flush_wgrads"):
        setattr(ops, n, wrap(n, f))

g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
B = 2
traces = []
for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    net = CoUpdater(margs); net.load_state_dict(sd)
    net = freeze_like_reference(net.to("cuda:0").train())
    for m in net.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    im1, im2 = synthetic_pair(B, seed=7)
    gt = synthetic_gt(B, seed=7).cuda()
    im1, im2 = im1.cuda(), im2.cuda()
    fl = unFlowLoss()
    TRACE = []
    with torch.enable_grad():
        preds = net(im1, im2)
        pair = [torch.cat((preds[1][i], preds[2][i]), 1) for i in range(len(preds[1]))]
        loss = hybrid_e_loss(preds[0], gt) + fl.compute_loss(pair, torch.cat((im1, im2), 1))[0]
        TRACE.append(("loss", 0, (), loss.item(), 0.0))
        nfw = len(TRACE)
        loss.backward()
    traces.append((nfw, TRACE)); TRACE = None
    del net; torch.cuda.empty_cache()
n0, t0 = traces[0]
print("launch-level records per step:", len(t0), "forward:", n0)
for r, (n, t) in enumerate(traces[1:], 1):
    if len(t) != len(t0):
        print("run", r, "trace length differs", len(t), len(t0))
    first = next((i for i, (a, b) in enumerate(zip(t0, t)) if a != b and not (a[3] != a[3] and b[3] != b[3])), None)
    ndiff = sum(1 for a, b in zip(t0, t) if a != b)
    print("run %d: %d records differ; first at %s" % (r, ndiff, first))
    if first is not None:
        for i in range(max(0, first - 2), min(len(t0), first + 4)):
            print("    ", i, "FW" if i < n0 else "BW", t0[i], "|", t[i][3:], "" if t0[i] == t[i] else "  <-- differs")
