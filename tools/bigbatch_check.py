import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater
g = "tests/golden"
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.cuda().eval()
for B in (32, 48, 64):
    im1, im2 = synthetic_pair(B, seed=5)
    with torch.no_grad():
        m = net(im1.cuda(), im2.cuda())[0]
    torch.cuda.synchronize()
    print(B, "pairs ok", tuple(m.shape), bool(torch.isfinite(m).all()), flush=True)
