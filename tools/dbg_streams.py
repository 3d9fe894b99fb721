#!/usr/bin/env python3
"""the same forward EAGERLY on N streams at once (no hipGraph): do concurrent forwards still corrupt each other's results?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(16, seed=77)
im1, im2 = im1.cuda(), im2.cuda()
with torch.no_grad():
    ref = net(im1, im2)[0].clone()
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    bad = 0
    for rnd in range(rounds):
        outs = []
        for s in streams:
            with torch.cuda.stream(s):
                outs.append(net(im1, im2)[0])
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            if not torch.equal(o, ref):
                d = (o.float() - ref.float()).abs()
                print("round %d stream %d: max %.4f, images %s" % (rnd, i, d.max().item(), [k for k in range(16) if d[k].max() > 0]), flush=True)
                bad += 1
print("mismatches: %d of %d" % (bad, rounds * n))
