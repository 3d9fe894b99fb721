#!/usr/bin/env python3
"""does the forward depend on the ORDER of independent parts (GMFlow CNN ahead of / behind the PVT backbone) or on what the
allocator hands out?  every kernel is deterministic now, so any difference is an uninitialised read or a race"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.model.EMIP_short import model as M
from emip_amd.model.EMIP_short.model import CoUpdater
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(16, seed=77)
im1, im2 = im1.cuda(), im2.cuda()
def flat(d, prefix=""):
    out = {}
    for k, v in (d.items() if isinstance(d, dict) else enumerate(d)):
        if torch.is_tensor(v): out[f"{prefix}{k}"] = v.clone()
        elif isinstance(v, (dict, list, tuple)): out.update(flat(v, f"{prefix}{k}."))
    return out
res = []
with torch.no_grad():
    for cnn_first, poison in ((False, False), (True, False), (False, True), (True, True)):
        M.CNN_FIRST = cnn_first
        if poison:      # fill the allocator's free blocks with NaN patterns: an uninitialised read shows
            torch.cuda.empty_cache()
            junk = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 28, 1 << 27, 1 << 26, 1 << 25, 1 << 24, 1 << 22, 1 << 20)]
            del junk
        mask = net(im1, im2)[0].clone()
        res.append((cnn_first, poison, mask, flat(net.last)))
M.CNN_FIRST = False
base = res[0]
for cf, po, mask, inter in res[1:]:
    bad = [n for n in base[3] if n in inter and inter[n].shape == base[3][n].shape and not torch.equal(inter[n], base[3][n])]
    print("CNN_FIRST=%s poison=%s: mask equal %s, max |d| %.4f, finite %s; first differing intermediates: %s" % (
        cf, po, torch.equal(mask, base[2]), (mask.float() - base[2].float()).abs().max().item(), torch.isfinite(mask).all().item(), bad[:6]))
