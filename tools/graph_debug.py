import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.graph import GraphedShort
from emip_amd.model.EMIP_short.model import CoUpdater
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
for dt in (torch.float32, torch.bfloat16):
    nn_base.set_default_dtype(dt)
    net = CoUpdater(margs); net.load_state_dict(sd); net = net.cuda().eval()
    im1, im2 = synthetic_pair(2, seed=4321); im1, im2 = im1.cuda(), im2.cuda()
    with torch.no_grad():
        ref, _, _ = net(im1, im2)
        r1a, _, _ = net(im1[:1], im2[:1]); r1b, _, _ = net(im1[1:], im2[1:])
    print(dt, "eager B=1 vs B=2:", (torch.cat([r1a, r1b]) - ref).abs().max().item())
    g1 = GraphedShort(net, 2, splits=1)
    m, _, _ = g1(im1, im2); torch.cuda.synchronize()
    print(dt, "graph splits=1:", (m - ref).abs().max().item())
    g2 = GraphedShort(net, 2, splits=2)
    g2.load(im1, im2)
    for p in g2.parts:
        p.graph.replay(); torch.cuda.synchronize()
    m = torch.cat([p.mask for p in g2.parts]); print(dt, "graph splits=2 sequential:", (m - ref).abs().max().item())
    m, _, _ = g2(im1, im2); torch.cuda.synchronize()
    print(dt, "graph splits=2 concurrent:", (m - ref).abs().max().item())
    m, _, _ = g1(im1, im2); torch.cuda.synchronize()
    print(dt, "graph splits=1 again:", (m - ref).abs().max().item())
