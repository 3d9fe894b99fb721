"""Distribution of the bf16 flow error under the well-conditioned filler (tests/golden/short_eval_flow.npz)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd import nn_base
from emip_amd.filler import flow_conditioned, textured_pair, state_dict_from_manifest
from emip_amd.model.EMIP_short.model import CoUpdater
g = np.load(os.path.join(ROOT, "tests/golden/short_eval_flow.npz"))
args = json.load(open(os.path.join(ROOT, "tests/golden/model_args.json")))
sd = flow_conditioned(state_dict_from_manifest(json.load(open(os.path.join(ROOT, "tests/golden/short_state_manifest.json"))), 0))
im1, im2 = textured_pair()
for dt in (torch.float32, torch.bfloat16):
    nn_base.set_default_dtype(dt)
    net = CoUpdater(args); net.load_state_dict(sd); net = net.to("cuda:0").eval()
    with torch.no_grad():
        m, fw, bw = net(im1.cuda(), im2.cuda())
    for name, t, ref in (("fw", fw[0], g["fw"]), ("bw", bw[0], g["bw"])):
        e = (t.float().cpu()[:, :, ::4, ::4].numpy() - ref)
        e = np.sqrt((e ** 2).sum(1)).ravel()
        mag = np.sqrt((ref ** 2).sum(1)).ravel()
        inl = mag < 25
        print(dt, name, "pct 50/90/99/99.9/max:", np.percentile(e, [50, 90, 99, 99.9, 100]).round(4),
              "frac>1px", (e > 1).mean().round(5), "inliers(|ref|<25):", inl.mean().round(4), "inlier max", e[inl].max().round(4),
              "inlier frac>1", (e[inl] > 1).mean().round(5))
    nn_base.set_default_dtype(torch.float32)
