#!/usr/bin/env python3
"""Per-launch timeline of one eager EMIP-short forward: every C-ABI call in issue order with its integer arguments and
its HIP-event duration.  Diagnosis only (which launches of a block are far from their floor at a given batch).

  python tools/timeline.py --pairs 16 --out gpurun_out/timeline_b16.tsv"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=16)
    ap.add_argument("--out", default="gpurun_out/timeline.tsv")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from emip_amd import _lib, nn_base, ops
    from emip_amd.filler import state_dict_from_manifest, synthetic_pair
    from emip_amd.model.EMIP_short.model import CoUpdater
    _lib.load()
    g = os.path.join(ROOT, "tests", "golden")
    margs = json.load(open(os.path.join(g, "model_args.json")))
    sd = state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0)
    nn_base.set_default_dtype(torch.bfloat16)
    net = CoUpdater(margs)
    net.load_state_dict(sd)
    net = net.to("cuda:0").eval()
    im1, im2 = synthetic_pair(args.pairs, seed=1234)
    im1, im2 = im1.cuda(), im2.cuda()
    ba = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
    bo = torch.empty_like(ba)
    runs = []
    with torch.no_grad():
        net.run(im1, im2)
        net.run(im1, im2)
        torch.cuda.synchronize()
        for _ in range(args.reps):
            rec = []
            for _ in range(60):                 # blocker: the host runs ahead, event pairs bracket kernels only
                ops.gemm(ba, ba, out=bo)
            _lib.profile(rec)
            net.run(im1, im2)
            _lib.profile(None)
            torch.cuda.synchronize()
            runs.append(rec)
    n = len(runs[0])
    assert all(len(r) == n for r in runs)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    tot = 0.0
    byname = {}
    with open(args.out, "w") as f:
        for i in range(n):
            name, a, _, _ = runs[0][i]
            us = sorted(r[i][2].elapsed_time(r[i][3]) * 1e3 for r in runs)[len(runs) // 2]
            ints = [str(x) for x in a if isinstance(x, int) and not isinstance(x, bool) and abs(x) < (1 << 31)]
            f.write("%d\t%s\t%.2f\t%s\n" % (i, name, us, ",".join(ints)))
            tot += us
            d = byname.setdefault(name, [0.0, 0])
            d[0] += us
            d[1] += 1
    print("pairs %d: %d launches, sum of event durations %.2f ms" % (args.pairs, n, tot / 1e3))
    for k, v in sorted(byname.items(), key=lambda kv: -kv[1][0])[:25]:
        print("  %-28s %6d calls %9.2f ms  avg %7.2f us" % (k, v[1], v[0] / 1e3, v[0] / v[1]))


if __name__ == "__main__":
    main()
