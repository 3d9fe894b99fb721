#!/usr/bin/env python3
"""Which torch (aten) kernels run inside one EMIP-short inference forward, by call site (TorchDispatchMode around a warm
forward).  Everything else is libemip_hip.so; this lists what is left to remove from the graph."""
import collections, json, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from emip_amd import _lib, nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.model.EMIP_short.model import CoUpdater

SKIP = ("aten.view", "aten.detach", "aten.t.", "aten.transpose", "aten.permute", "aten.slice", "aten.select", "aten.alias",
        "aten._unsafe_view", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.as_strided", "aten.reshape",
        "aten.empty", "aten.unbind", "aten.split", "aten.lift_fresh", "aten._local_scalar", "aten.is_", "aten.sym_")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "?"
            for f in traceback.extract_stack()[:-1][::-1]:
                if "/emip_amd/" in f.filename and "_python_dispatch" not in f.filename:
                    site = "%s:%s:%d" % (os.path.basename(f.filename), f.name, f.lineno)
                    break
            shp = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            self.c[(name, site, str(shp))] += 1
        return func(*args, **(kwargs or {}))


_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.to("cuda:0").eval()
im1, im2 = synthetic_pair(B, seed=1234)
im1, im2 = im1.cuda(), im2.cuda()
with torch.no_grad():
    for _ in range(2):
        net.run(im1, im2)
    torch.cuda.synchronize()
    m = Count()
    with m:
        net.run(im1, im2)
torch.cuda.synchronize()
print("aten ops in one forward:", sum(m.c.values()))
for (n, s, shp), k in sorted(m.c.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("  %3d %-34s %-44s %s" % (k, n, s, shp))
