"""Golden-vector generator.  TEST INFRASTRUCTURE ONLY; runs ONLY in the build
container (needs /root/reference, which never travels to the GPU box).

Imports the reference's own modules on PyTorch-CPU, loads the name-keyed filler
weights (emip_amd/filler.py), runs seeded synthetic inputs and writes small
fixtures to tests/golden/.  The fixtures are data (inputs are regenerated from
seeds, outputs are stored); no reference source is copied.

The reference is unimportable as shipped because it imports timm / mmcv /
mmdet / torchvision / cv2 / matplotlib (absent here) and a package that does
not exist (model.EPFlow_1_feature, PromptInteract.py:4,6).  Following
SURVEY.md appendix B this script registers empty placeholder modules for
those names before the import.  None of them contributes arithmetic in eval
mode: DropPath is identity outside train mode, trunc_normal_ only initialises
weights that the filler overwrites, the rest are decorators/loaders.

usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd.filler import filled_state_dict, synthetic_gt, synthetic_pair  # noqa: E402

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _DropPath(nn.Module):
    """timm semantics: per-sample Bernoulli keep, scaled by 1/keep (train only)."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1 - self.p
        r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0:
            r.div_(keep)
        return x * r


def install_placeholders():
    _mod("timm")
    _mod("timm.models", create_model=None)
    _mod("timm.models.layers", DropPath=_DropPath,
         to_2tuple=lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x),
         trunc_normal_=lambda t, mean=0.0, std=1.0, a=-2.0, b=2.0: nn.init.trunc_normal_(t, mean, std, a, b))
    _mod("timm.models.registry", register_model=lambda f: f)
    _mod("timm.models.vision_transformer", _cfg=lambda **k: {})

    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c

    _mod("mmdet")
    _mod("mmdet.models")
    _mod("mmdet.models.builder", BACKBONES=_Reg())
    _mod("mmdet.utils", get_root_logger=lambda *a, **k: None)
    _mod("mmcv")
    _mod("mmcv.runner", load_checkpoint=lambda *a, **k: None)
    for n in ("torchvision", "torchvision.models", "cv2", "matplotlib", "matplotlib.pyplot"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.path.insert(0, REF)
    import model.EMIP_short.motion.common as _c
    import model.EMIP_short.motion.transformer as _t
    _mod("model.EPFlow_1_feature")
    _mod("model.EPFlow_1_feature.motion")
    sys.modules["model.EPFlow_1_feature.motion.common"] = _c
    sys.modules["model.EPFlow_1_feature.motion.transformer"] = _t


def stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.pow(2).sum().sqrt().item(), t.abs().max().item()], dtype=np.float64)


def f32(t):
    return t.detach().float().contiguous().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_placeholders()
    import yaml
    from model.EMIP_short.model import CoUpdater
    from model.EMIP_long.model_long import Model_long
    from loss import loss_flow, loss_pred, warp_utils

    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "configs.yaml")))
    margs = cfg["model"]["args"]
    with open(os.path.join(args.out, "model_args.json"), "w") as f:
        json.dump(margs, f, indent=1, sort_keys=True)

    # ---------------- EMIP-short ------------------------------------------------
    net = CoUpdater(args=margs)
    sd = filled_state_dict(net.state_dict(), seed=0)
    net.load_state_dict(sd)
    manifest = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()}
    with open(os.path.join(args.out, "short_state_manifest.json"), "w") as f:
        json.dump(manifest, f)
    trainable = [n for n, p in net.named_parameters()
                 if not ("GMFlow" in n and "dwconv" not in n and "adaptor" not in n)]
    with open(os.path.join(args.out, "short_trainable.json"), "w") as f:
        json.dump(trainable, f)

    net.eval()
    cap = {}
    hooks = []

    def grab(name):
        def fn(_m, _i, o):
            cap.setdefault(name, []).append(o)
        return fn

    hooks.append(net.backbone.feat_net.register_forward_hook(grab("pvt")))
    hooks.append(net.GMFlow.backbone.register_forward_hook(grab("gmcnn")))
    hooks.append(net.injector.register_forward_hook(grab("inj")))
    hooks.append(net.GMFlow.transformer.register_forward_hook(grab("gmtr")))
    hooks.append(net.GMFlow.feature_flow_attn.register_forward_hook(grab("flowattn")))
    hooks.append(net.conv_corr.register_forward_hook(grab("conv_corr")))
    hooks.append(net.injector1.register_forward_hook(grab("inj1")))
    hooks.append(net.dr1.register_forward_hook(grab("dr1")))
    hooks.append(net.dr2.register_forward_hook(grab("dr2")))
    hooks.append(net.dr3.register_forward_hook(grab("dr3")))
    hooks.append(net.decoder.conv5.register_forward_hook(grab("pc")))
    hooks.append(net.GMFlow.feature_flow_attn.register_forward_pre_hook(
        lambda _m, i: cap.setdefault("flow_lr", []).append(i[1])))
    hooks.append(net.conv_corr.register_forward_pre_hook(lambda _m, i: cap.setdefault("corr", []).append(i[0])))

    for B in (1, 2):
        cap.clear()
        im1, im2 = synthetic_pair(B, seed=1234)
        with torch.no_grad():
            mask, fw, bw = net(im1, im2)
        g = {}
        g["mask"] = f32(mask)
        g["flow_fw"] = f32(fw[0][:, :, ::4, ::4])
        g["flow_bw"] = f32(bw[0][:, :, ::4, ::4])
        g["flow_fw_stats"] = stats(fw[0])
        g["flow_bw_stats"] = stats(bw[0])
        p1 = cap["pvt"][0]
        g["pvt1_s2"] = f32(p1[0][:, :, ::2, ::2])
        g["pvt1_s3"] = f32(p1[1])
        g["pvt1_s4"] = f32(p1[2])
        g["pvt1_s2_stats"] = stats(p1[0])
        g["pvt2_s2_stats"] = stats(cap["pvt"][1][0])
        g["gm1"] = f32(cap["gmcnn"][0][0][:, :, ::2, ::2])
        g["gm1_stats"] = stats(cap["gmcnn"][0][0])
        g["inj_a"] = f32(cap["inj"][0][:, :, ::2, ::2])
        g["inj_b_stats"] = stats(cap["inj"][1])
        g["f0"] = f32(cap["gmtr"][0][0][:, :, ::2, ::2])
        g["f1_stats"] = stats(cap["gmtr"][0][1])
        corr = cap["corr"][0]  # [B, tgt, 44, 44(src)]
        g["corr_block"] = f32(corr[:, :64].reshape(B, 64, -1)[:, :, :64])
        g["corr_stats"] = stats(corr)
        g["flow_lr"] = f32(cap["flow_lr"][0])
        g["flow_prop"] = f32(cap["flowattn"][0])
        g["conv_corr"] = f32(cap["conv_corr"][0][:, :, ::2, ::2])
        g["conv_corr_stats"] = stats(cap["conv_corr"][0])
        g["inj1"] = f32(cap["inj1"][0][:, :, ::2, ::2])
        g["dr1"] = f32(cap["dr1"][0][:, :, ::2, ::2])
        g["dr2"] = f32(cap["dr2"][0])
        g["dr3"] = f32(cap["dr3"][0])
        g["pc"] = f32(cap["pc"][0])
        if B == 2:  # keep the batch fixture light
            g = {k: v for k, v in g.items() if k in ("mask", "pc", "flow_fw", "flow_bw", "flow_lr", "flow_prop")
                 or k.endswith("_stats")}
        np.savez_compressed(os.path.join(args.out, f"short_eval_b{B}.npz"), **g)
        print("short eval B=%d: mask mean %.6f  absmax %.4f" % (B, mask.mean().item(), mask.abs().max().item()))
    for h in hooks:
        h.remove()

    # ---------------- train-mode forward + both losses (DropPath off) -------------
    # DropPath randomness is removed by zeroing its rate (the Bernoulli stream is
    # not a parity contract); BatchNorm batch statistics and the second (bilinear)
    # flow prediction of train mode are exercised.
    net.train()
    for m in net.modules():
        if isinstance(m, _DropPath):
            m.p = 0.0
    im1, im2 = synthetic_pair(2, seed=77)
    gt = synthetic_gt(2, seed=99)
    with torch.no_grad():
        mask, fw, bw = net(im1, im2)
        lp = loss_pred.hybrid_e_loss(mask, gt)
        fp = [torch.cat([fw[i], bw[i]], 1) for i in range(len(fw))]
        lf = loss_flow.unFlowLoss().compute_loss(fp, torch.cat((im1, im2), 1))[0]
    np.savez_compressed(os.path.join(args.out, "short_train_b2.npz"), mask=f32(mask),
                        flow0_fw=f32(fw[0][:, :, ::4, ::4]), flow1_fw=f32(fw[1][:, :, ::4, ::4]),
                        loss_pred=np.float64(lp.item()), loss_flow=np.float64(lf.item()), n_preds=len(fw))
    print("short train: loss_pred %.6f loss_flow %.6f" % (lp.item(), lf.item()))

    # ---------------- loss-side micro goldens ---------------------------------------
    rs = np.random.RandomState(5)
    H, W = 24, 40
    x = torch.from_numpy(rs.uniform(-1, 1, (2, 3, H, W)).astype(np.float32))
    y = torch.from_numpy(rs.uniform(-1, 1, (2, 3, H, W)).astype(np.float32))
    flow = torch.from_numpy(rs.normal(0, 3.0, (2, 2, H, W)).astype(np.float32))
    flow[0, :, 0, 0] = torch.tensor([-7.5, 2.25])  # out of range corners
    flow[1, :, H - 1, W - 1] = torch.tensor([5.0, 5.0])
    flow[1, :, 3, 3] = torch.tensor([0.0, 0.0])  # exact integer coordinates
    warped = warp_utils.flow_warp(x, flow)
    B = 2
    base = warp_utils.mesh_grid(B, H, W).type_as(flow)
    cmap = warp_utils.get_corresponding_map(base + flow)
    occ = warp_utils.get_occu_mask_backward(flow)
    pred = torch.from_numpy(rs.normal(0, 2.0, (2, 1, H, W)).astype(np.float32))
    gtm = (torch.from_numpy(rs.uniform(0, 1, (2, 1, H, W)).astype(np.float32)) > 0.7).float()
    hl = loss_pred.hybrid_e_loss(pred, gtm)
    flows4 = [torch.cat([flow, -flow * 0.5], 1), torch.cat([flow * 0.9, -flow * 0.4], 1)]
    ul = loss_flow.unFlowLoss().compute_loss(flows4, torch.cat((x, y), 1))[0]
    from loss.loss_blocks import SSIM
    np.savez_compressed(os.path.join(args.out, "loss_micro.npz"), x=f32(x), y=f32(y), flow=f32(flow),
                        warped=f32(warped), cmap=f32(cmap), occ=f32(occ), pred=f32(pred), gt=f32(gtm),
                        hybrid=np.float64(hl.item()), unflow=np.float64(ul.item()), ssim=f32(SSIM(x, y, 1)))

    # full-size warp-index golden (bit-exact contract): indices for a seeded flow at 352x352
    fl = torch.from_numpy(np.random.RandomState(11).normal(0, 6.0, (1, 2, 352, 352)).astype(np.float32))
    base = warp_utils.mesh_grid(1, 352, 352).type_as(fl)
    data = base + fl
    # re-derive indices exactly as warp_utils.py:43-66 evaluates them
    xx = data[:, 0].view(1, -1)
    yy = data[:, 1].view(1, -1)
    x1 = torch.floor(xx); xf = x1.clamp(0, 351); y1 = torch.floor(yy); yf = y1.clamp(0, 351)
    xc = (x1 + 1).clamp(0, 351); yc = (y1 + 1).clamp(0, 351)
    idx = torch.cat([xc + yc * 352, xc + yf * 352, xf + yc * 352, xf + yf * 352], 1).long()
    occ352 = warp_utils.get_occu_mask_backward(fl)
    np.savez_compressed(os.path.join(args.out, "warp_indices_352.npz"), indices=idx.numpy().astype(np.int32),
                        occ=occ352.numpy().astype(np.uint8))

    # ---------------- EMIP-long -------------------------------------------------------
    del net
    lnet = Model_long(args=margs)
    lsd = filled_state_dict(lnet.state_dict(), seed=0)
    lnet.load_state_dict(lsd)
    lman = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in lnet.state_dict().items()}
    with open(os.path.join(args.out, "long_state_manifest.json"), "w") as f:
        json.dump(lman, f)
    lnet.eval()
    T = 8
    frames = []
    for t in range(T):
        a, _ = synthetic_pair(1, seed=500, shift=(0, 0))
        frames.append(None)
    # a drifting sequence: frame t = base field shifted by (t, -t)
    seq = [synthetic_pair(1, seed=500, shift=(t - 4, 4 - t))[1][0] for t in range(T)]
    mk = mv = None
    out = {}
    with torch.no_grad():
        for i in range(T - 1):
            if i == 0:
                m, _, _ = lnet(seq[0], seq[1], 0, None, None)
            else:
                m, mk, mv = lnet(seq[i - 1], seq[i], i, mk, mv)
            out[f"mask_{i}"] = f32(m[:, :, ::2, ::2])
            out[f"mask_{i}_stats"] = stats(m)
            if mk is not None:
                out[f"k_{i}_stats"] = stats(mk)
                out[f"v_{i}_stats"] = stats(mv)
                out[f"T_{i}"] = np.int64(mk.shape[3])
    np.savez_compressed(os.path.join(args.out, "long_eval.npz"), **out)
    print("long: T window", [int(out[k]) for k in sorted(out) if k.startswith("T_")])


if __name__ == "__main__":
    main()
