#!/usr/bin/env python3
"""gradients that flow through GMFlow's attention backward (injector.*) in f32 (unfused, reference-validated), bf16 with the fused
backward kernels and bf16 with the unfused chain: which bf16 path is closer to f32?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import autograd as ag, nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_gt, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import build_optimizer, freeze_like_reference, train_step

g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
names = ["injector.transformer.attn.project_out.weight", "injector.transformer.ffn.project_out.weight",
         "injector.transformer.attn.q.weight", "backbone.feat_net.pvtv2_en.block2.1.attn.q.weight"]

def grads(dtype, wattn, match):
    nn_base.set_default_dtype(dtype)
    ag.WATTN_BWD_FUSED, ag.MATCH_BWD_FUSED = wattn, match
    net = CoUpdater(margs); net.load_state_dict(sd)
    net = freeze_like_reference(net.to("cuda:0").train())
    for m in net.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    opt = build_optimizer(net, lr=0.0)
    im1, im2 = synthetic_pair(2, seed=11); gt = synthetic_gt(2, seed=11)
    train_step(net, opt, None, im1.cuda(), im2.cuda(), gt.cuda())
    ps = dict(net.named_parameters())
    out = {n: ps[n].grad.detach().float().clone() for n in names}
    nn_base.set_default_dtype(torch.float32)
    return out

f32 = grads(torch.float32, False, False)
runs = {"bf16 fused": grads(torch.bfloat16, True, True), "bf16 fused again": grads(torch.bfloat16, True, True),
        "bf16 unfused attention": grads(torch.bfloat16, False, True), "bf16 unfused both": grads(torch.bfloat16, False, False),
        "bf16 unfused both again": grads(torch.bfloat16, False, False)}
for n in names:
    ref = f32[n]
    print(n, "f32 |g| max %.3e" % ref.abs().max().item())
    for k, v in runs.items():
        d = (v[n] - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        cos = torch.nn.functional.cosine_similarity(v[n].flatten(), ref.flatten(), dim=0).item()
        print("   %-26s rel max err vs f32 %.3f   cosine %.4f" % (k, d, cos))
