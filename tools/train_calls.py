#!/usr/bin/env python3
"""Per-launch listing of one bf16 training step (32 pairs): HIP-event pair around every C-ABI call, grouped by entry point and
its integer arguments (shapes), largest first.  Deferred weight gradients appear as the grouped launch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, nn_base, ops
from emip_amd.filler import state_dict_from_manifest, synthetic_pair, synthetic_gt
from emip_amd.loss.loss_flow import unFlowLoss
from emip_amd.loss.loss_pred import hybrid_e_loss
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import freeze_like_reference

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd)
net = freeze_like_reference(net.to("cuda:0").train())
im1, im2 = synthetic_pair(B, seed=7)
gt = synthetic_gt(B, seed=7).cuda()
im1, im2 = im1.cuda(), im2.cuda()
fl = unFlowLoss()
ba = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16); bo = torch.empty_like(ba)
rec = []
for rep in range(3):
    net.zero_grad(set_to_none=True)
    ops.ARENA.begin(im1.device)
    if rep == 2:
        for _ in range(150):
            ops.gemm(ba, ba, out=bo)
        _lib.profile(rec)
    with torch.enable_grad():
        preds = net(im1, im2)
        pair = [torch.cat((preds[1][i], preds[2][i]), 1) for i in range(len(preds[1]))]
        loss = hybrid_e_loss(preds[0], gt) + fl.compute_loss(pair, torch.cat((im1, im2), 1))[0]
        nfw = len(rec)
        loss.backward()
        ops.flush_wgrads()
    ops.ARENA.end()
    _lib.profile(None)
    torch.cuda.synchronize()
agg = {}
tot = [0.0, 0.0]
for i, (name, a, s, e) in enumerate(rec):
    us = s.elapsed_time(e) * 1e3
    tot[i >= nfw] += us
    key = (("FW" if i < nfw else "BW"), name, tuple(x for x in a if isinstance(x, int) and 0 < x < (1 << 22))[:12])
    d = agg.setdefault(key, [0.0, 0]); d[0] += us; d[1] += 1
print("calls %d (forward %d); forward %.1f us, backward %.1f us" % (len(rec), nfw, tot[0], tot[1]))
for k, d in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TOP", "70"))]:
    print("%s %-30s %9.1f us %4d calls  %7.1f each  %s" % (k[0], k[1], d[0], d[1], d[0] / d[1], k[2]))
