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
nn_base.set_default_dtype(torch.float32)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(2, seed=4321); im1, im2 = im1.cuda(), im2.cuda()
with torch.no_grad():
    ref, rfw, rbw = net(im1, im2)
runner = GraphedShort(net, 2, splits=2)
for it in range(3):
    mask, fw, bw = runner(im1, im2)
    torch.cuda.synchronize()
    print(it, "mask err", (mask - ref).abs().max().item(), "per part", [(p.mask - ref[i:i+1]).abs().max().item() for i, p in enumerate(runner.parts)],
          "inputs ok", [(p.im1 - im1[i:i+1]).abs().max().item() for i, p in enumerate(runner.parts)])
with torch.no_grad():
    for i, p in enumerate(runner.parts):
        m, _ = net.run(p.im1, p.im2)
        print("eager on static inputs part", i, (m - ref[i:i+1]).abs().max().item())
print("---- first divergence, part 1")
p = runner.parts[1]
runner(im1, im2); torch.cuda.synchronize()
cap = {k: (v if not isinstance(v, (list, tuple)) else list(v)) for k, v in p.last.items()}
snap = {}
for k, v in cap.items():
    snap[k] = [t.float().clone() for t in v] if isinstance(v, list) else v.float().clone()
with torch.no_grad():
    net.run(p.im1, p.im2)
torch.cuda.synchronize()
for k, v in net.last.items():
    if isinstance(v, (list, tuple)):
        for i, t in enumerate(v):
            print(k, i, (t.float() - snap[k][i]).abs().max().item(), "scale", t.float().abs().max().item())
    else:
        print(k, (v.float() - snap[k]).abs().max().item(), "scale", v.float().abs().max().item())
