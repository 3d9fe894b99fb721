#!/usr/bin/env python3
"""Eager EMIP-short forwards for a rocprofv3 --kernel-trace run (true device-side kernel durations and the gaps between
them -- HIP-event pairs add ~5 us per launch and hide what a kernel boundary costs):

  rocprofv3 --kernel-trace -d gpurun_out/prof -o fwd -- python3 tools/fwd_profile.py --pairs 16 --reps 3
  python3 tools/fwd_profile.py --parse gpurun_out/prof/.../fwd_kernel_trace.csv --launches 617"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    import torch
    from emip_amd import _lib, nn_base
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
    with torch.no_grad():
        for _ in range(2):
            net.run(im1, im2)
        torch.cuda.synchronize()
        if args.graph:
            from emip_amd.graph import GraphedShort
            r = GraphedShort(net, args.pairs, splits=1)
            r.load(im1, im2)
            torch.cuda.synchronize()
            for _ in range(args.reps):
                r.replay()
            torch.cuda.synchronize()
        else:
            for _ in range(args.reps):
                net.run(im1, im2)
            torch.cuda.synchronize()


def parse(args):
    rows = list(csv.DictReader(open(args.parse)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = args.launches
    last = rows[-n:]                       # the final forward
    t0 = int(last[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
    span = int(last[-1]["End_Timestamp"]) - t0
    print("last %d dispatches: span %.3f ms, sum of kernel durations %.3f ms, gaps %.3f ms" % (n, span / 1e6, busy / 1e6,
                                                                                              (span - busy) / 1e6))
    agg = {}
    prev_end = None
    out = open(args.parse + ".timeline.tsv", "w")
    for i, r in enumerate(last):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"]
        short = name.split("(")[0][-70:]
        gap = 0 if prev_end is None else s - prev_end
        prev_end = e
        out.write("%d\t%s\t%.2f\t%.2f\t%s\n" % (i, short, (e - s) / 1e3, gap / 1e3, r.get("Grid_Size", "")))
        d = agg.setdefault(short, [0, 0, 0])
        d[0] += e - s
        d[1] += 1
        d[2] += gap
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
        print("  %-72s %5d calls %8.3f ms  avg %7.2f us  gap-before avg %5.2f us" % (k, v[1], v[0] / 1e6, v[0] / v[1] / 1e3,
                                                                                     v[2] / v[1] / 1e3))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--parse", default="")
    ap.add_argument("--launches", type=int, default=617)
    a = ap.parse_args()
    parse(a) if a.parse else run(a)
