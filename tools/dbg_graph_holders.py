#!/usr/bin/env python3
"""After an eager training pass: which live tensors still carry an autograd graph (they keep the parameters' AccumulateGrad
nodes, and with them the stream those were created under), and who refers to them."""
import gc, json, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, nn_base, ops
from emip_amd.filler import state_dict_from_manifest, synthetic_gt, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd import train as T

_lib.load()
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd)
net = T.freeze_like_reference(net.to("cuda:0").train())
opt = T.build_optimizer(net)
im1, im2 = (t.cuda() for t in synthetic_pair(2, seed=1))
gt = synthetic_gt(2, seed=1).cuda()
opt.zero_grad(set_to_none=True)
T.forward_backward(net, im1, im2, gt)
torch.cuda.synchronize()
for m in net.modules():
    if isinstance(getattr(m, "last", None), dict):
        m.last = {}
gc.collect()
n = 0
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.grad_fn is not None:
            n += 1
            if n <= 12:
                refs = [type(r).__name__ + (":" + ",".join(list(r.keys())[:6]) if isinstance(r, dict) else "") for r in gc.get_referrers(o)][:6]
                print(tuple(o.shape), o.dtype, type(o.grad_fn).__name__, refs, flush=True)
    except Exception as e:
        pass
print("live tensors with a grad_fn after the step:", n, flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    with torch.cuda.stream(s):
        opt.zero_grad(set_to_none=True)
        T.forward_backward(net, im1, im2, gt)
    torch.cuda.synchronize()
    print("stream-mismatch warnings on a side stream:", sum("AccumulateGrad" in str(x.message) for x in w), flush=True)
