"""Experiment: does splitting the 16-pair batch over two HIP streams (two hipGraphs replayed concurrently)
raise throughput?  The per-kernel profile says the small PVT kernels are latency-bound, not throughput-bound."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emip_amd import nn_base
from emip_amd.filler import state_dict_from_manifest, synthetic_pair
from emip_amd.graph import GraphedShort
from emip_amd.model.EMIP_short.model import CoUpdater
g = os.path.join(ROOT, "tests", "golden")
margs = json.load(open(os.path.join(g, "model_args.json")))
sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
nn_base.set_default_dtype(torch.bfloat16)
net = CoUpdater(margs); net.load_state_dict(sd); net = net.cuda().eval()
def bench(nsplit, B=16, steps=20):
    runners = [GraphedShort(net, B // nsplit) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    def step():
        for r, s in zip(runners, streams):
            with torch.cuda.stream(s):
                r.replay()
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("splits %d: %.2f ms per %d pairs -> %.1f pairs/s" % (nsplit, dt * 1e3, B, B / dt), flush=True)
for n in (4, 8, 16):
    bench(n)
