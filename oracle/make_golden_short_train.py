#!/usr/bin/env python3
"""Golden vector for one EMIP-short TRAINING step (train.py:43-60), produced by the reference itself.
TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).

Reference CoUpdater in train() mode with the freeze rule of train.py:340-342, DropPath rates zeroed (deterministic step),
one frame pair: forward, hybrid_e_loss + unFlowLoss, backward.  The fixture holds both loss values and gradient
statistics / leading elements of named parameters across the model.
usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_short_train.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from make_golden import f32, install_placeholders, stats  # noqa: E402
from emip_amd.filler import filled_state_dict, synthetic_gt, synthetic_pair  # noqa: E402

P = "backbone.feat_net.pvtv2_en."
NAMES = [P + "patch_embed1.proj.weight", P + "block1.0.attn.q.weight", P + "block1.2.mlp.dwconv.dwconv.weight",
         P + "block2.3.mlp.fc1.weight", P + "block3.20.attn.kv.weight", P + "block3.39.attn.norm.weight",
         P + "block4.2.mlp.fc2.weight", P + "norm4.weight", "injector.transformer.attn.temperature",
         "injector.transformer.attn.q.weight", "injector.transformer.ffn.project_out.weight",
         "injector1.transformer.attn.kv.weight", "conv_corr.0.weight", "conv_corr.1.weight", "conv_corr.3.weight",
         "dr1.reduce.0.conv.weight", "dr3.reduce.1.bn.weight", "decoder.conv_upsample5.conv.weight", "decoder.conv5.weight",
         "decoder.conv5.bias"]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_placeholders()
    from model.EMIP_short.model import CoUpdater
    from loss import loss_flow, loss_pred
    margs = json.load(open(os.path.join(ROOT, "tests", "golden", "model_args.json")))
    net = CoUpdater(args=margs)
    net.load_state_dict(filled_state_dict(net.state_dict(), seed=0))
    for name, para in net.named_parameters():          # train.py:340-342
        if "GMFlow" in name and 'dwconv' not in name and 'adaptor' not in name:
            para.requires_grad = False
    net.train()
    for m in net.modules():
        if m.__class__.__name__ == "_DropPath":
            m.p = 0.0
    im1, im2 = synthetic_pair(1, seed=99)
    gt = synthetic_gt(1, seed=99)
    preds = net(im1, im2)
    lp = loss_pred.hybrid_e_loss(preds[0], gt)
    flow_pair = [torch.cat((preds[1][i], preds[2][i]), dim=1) for i in range(len(preds[1]))]
    lf = loss_flow.unFlowLoss().compute_loss(flow_pair, torch.cat((im1, im2), dim=1))[0]
    (lp + lf).backward()
    out = {"loss_pred": np.float64(lp.item()), "loss_flow": np.float64(lf.item()), "mask": f32(preds[0][:, :, ::4, ::4])}
    p = dict(net.named_parameters())
    for i, n in enumerate(NAMES):
        g = p[n].grad
        assert g is not None, n
        out["g%d_stats" % i] = stats(g)
        out["g%d_head" % i] = f32(g.reshape(-1)[:64])
    out["names"] = np.array(NAMES)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "short_train_grads.npz"), **out)
    print("short train grads: loss_pred %.6f loss_flow %.6f" % (lp.item(), lf.item()))


if __name__ == "__main__":
    main()
