#!/usr/bin/env python3
"""Which torch (aten) kernels run inside one training step, by call site: a TorchDispatchMode around train_step #3.
Everything else on the step is libemip_hip.so; this lists what is left to remove."""
import collections, json, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from emip_amd import _lib, nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_gt, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater
from emip_amd.train import build_optimizer, freeze_like_reference, train_step

SKIP = ("aten.view", "aten.detach", "aten.t.", "aten.transpose", "aten.permute", "aten.slice", "aten.select", "aten.alias",
        "aten._unsafe_view", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.as_strided", "aten.reshape",
        "aten.empty", "aten.unbind", "aten.split", "aten.lift_fresh", "aten._local_scalar", "aten.is_", "aten.sym_")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()
        self.shapes = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "?"
            for f in traceback.extract_stack()[:-1][::-1]:
                if ("/emip_amd/" in f.filename or f.filename.endswith("train_aten_ops.py")) and "_python_dispatch" not in f.filename:
                    site = "%s:%s:%d" % (os.path.basename(f.filename), f.name, f.lineno)
                    break
            self.c[(name, site)] += 1
            if name.startswith("aten.add.Tensor") or name.startswith("aten.clone") or name.startswith("aten.zeros"):
                shp = tuple(args[0].shape) if args and hasattr(args[0], "shape") else tuple(args[0]) if args else ()
                self.shapes[(name, site, shp)] += 1
        return func(*args, **(kwargs or {}))


_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd)
net = freeze_like_reference(net.to("cuda:0").train())
opt = build_optimizer(net)
im1, im2 = synthetic_pair(B, seed=1234); gt = synthetic_gt(B, seed=99)
im1, im2, gt = im1.cuda(), im2.cuda(), gt.cuda()
for _ in range(2):
    train_step(net, opt, None, im1, im2, gt)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)      # backward on this thread: a dispatch mode is thread-local
m = Count()
with m:
    train_step(net, opt, None, im1, im2, gt)
torch.cuda.synchronize()
tot = sum(m.c.values())
print("aten ops in one step:", tot)
byop = collections.Counter()
for (n, s), k in m.c.items():
    byop[n] += k
for n, k in byop.most_common(30):
    print("  %5d %s" % (k, n))
print("---- by site")
for (n, s), k in m.c.most_common(90):
    print("  %5d %-40s %s" % (k, n, s))
print("---- add / clone / zeros by shape")
for (n, s, shp), k in m.shapes.most_common(60):
    print("  %5d %-28s %-36s %s" % (k, n, s, shp))
