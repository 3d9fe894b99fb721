#!/usr/bin/env python3
"""What each part of the EMIP-short step costs UNDER bench.py's arrangement: the part is replaced by its cached output (captured
graphs of the remaining launches), and the whole-batch throughput with 4 steps in flight and the one-step-at-a-time latency are
measured again.  The differences to the full step are the marginal costs -- they need not add up (parts overlap)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.graph import PipelinedShort
from emip_amd.model.EMIP_short.model import CoUpdater

g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(16, seed=1234)
im1, im2 = im1.cuda(), im2.cuda()


def cached(obj, name):
    """replace obj.name by a function returning what the real one returned last"""
    real = getattr(obj, name)
    box = {}
    def rec(*a, **k):
        box["v"] = real(*a, **k)
        return box["v"]
    object.__setattr__(obj, name, rec)
    return box, real


pvt = net.backbone.feat_net.pvtv2_en
parts = {
    "gmflow cnn": [(net.GMFlow.backbone, "run")],
    "camouflage feeder": [(net.injector, "run")],
    "gmflow transformer + matching + flows": [(net.GMFlow, "run")],
    "conv_corr": [(net, "run_conv_corr_factored")],
    "injector1 + reductions + decoder": [(net.injector1, "run"), (net.dr1, "run"), (net.dr2, "run"), (net.dr3, "run"), (net.decoder, "run")],
}
boxes = {k: [cached(o, n) + (o, n) for o, n in v] for k, v in parts.items()}
with torch.no_grad():
    net(im1, im2)
torch.cuda.synchronize()


def set_part(key, on):
    if key.startswith("pvt stage"):
        i = int(key.split()[2])
        for blk in getattr(pvt, "block%d" % i):
            if on:
                object.__setattr__(blk, "run_fused", lambda x, stats, buf, alt=None, ws=None: (x, stats, alt))
            else:
                object.__delattr__(blk, "run_fused")
        return
    for box, real, o, n in boxes[key]:
        v = box["v"]
        object.__setattr__(o, n, (lambda *a, _v=v, **k: _v) if on else real)


def measure(inflight, steps):
    r = PipelinedShort(net, 16, inflight=inflight); r.load(im1, im2)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        for _ in range(6):
            r.replay_free()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.replay_free()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    del r
    return sorted(out)[1]


from emip_amd import ops as OPS
_real = {n: getattr(OPS, n) for n in ("mlp_fc1dw", "gemm", "sra_block", "conv8")}
S3 = {
    # launches of the 22 x 22 stage's blocks (C = 320) replaced by nothing (outputs: uninitialised or unchanged tensors)
    "s3 fc1 + depthwise (emip_mlp_fc1dw)": ("mlp_fc1dw", lambda f: (lambda x, w1, *a, **k: torch.empty(x.shape[:3] + (w1.shape[0],), dtype=x.dtype, device=x.device)
                                                              if x.shape[1] == 22 and x.shape[-1] == 320 else f(x, w1, *a, **k))),
    "s3 fc2 GEMM + its row statistics": ("gemm", lambda f: (lambda a, w, *r, **k: k["out"] if (a.shape[-1] == 1280 and w.shape[0] == 320 and k.get("out") is not None)
                                                          else f(a, w, *r, **k))),
    "s3 kv GEMM": ("gemm", lambda f: (lambda a, w, *r, **k: torch.empty(a.shape[:-1] + (640,), dtype=a.dtype, device=a.device)
                                      if (a.shape[-1] == 320 and w.shape[0] == 640) else f(a, w, *r, **k))),
    "s3 q + attention + proj (emip_sra_block)": ("sra_block", lambda f: (lambda x, *a, **k: None if x.shape[-1] == 320 else f(x, *a, **k))),
    "s3 spatial-reduction conv": ("conv8", lambda f: (lambda x, w, kh, kw, stride=1, pad=0, **k:
                                                      torch.empty((x.shape[0], x.shape[1] // stride, x.shape[2] // stride, w.shape[0]), dtype=x.dtype, device=x.device)
                                                      if k.get("cfg") == 10 else f(x, w, kh, kw, stride, pad, **k))),
}
_set_part = set_part


def set_part(key, on):
    if key in S3:
        name, mk = S3[key]
        setattr(OPS, name, mk(_real[name]) if on else _real[name])
        return
    _set_part(key, on)


if "--s3" in sys.argv:
    keys_override = ["(nothing)"] + list(S3)
else:
    keys_override = None
keys = ["(nothing)"] + (list(parts) if "--all" in sys.argv else []) + ["pvt stage %d blocks" % i for i in (1, 2, 3, 4)]
if keys_override:
    keys = keys_override
base = None
for k in keys:
    if k != "(nothing)":
        set_part(k, True)
    t4, t1 = measure(4, 32), measure(1, 12)
    if k != "(nothing)":
        set_part(k, False)
    else:
        base = (t4, t1)
    print("without %-40s: %6.3f ms per step with 4 in flight (%7.1f pairs/s, -%5.3f ms), %6.3f ms alone (-%5.3f ms)"
          % (k, t4, 16e3 / t4, base[0] - t4, t1, base[1] - t1), flush=True)
