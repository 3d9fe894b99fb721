#!/usr/bin/env python3
"""reads a rocprofv3 kernel-trace CSV of `tools/pipe_trace.py 1` (one step at a time): the LAST replay as a timeline -- per
kernel name the launches, the summed durations and the summed idle gaps in front of them; then the step's total"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 0
# replays are separated by the longest idle gaps: split at gaps > 40 us between consecutive kernels after the warm-up
cut = [i for i in range(1, len(ev)) if ev[i][0] - ev[i - 1][1] > 40000]
seg = ev[cut[-2]:cut[-1]] if len(cut) >= 2 else ev
agg = collections.defaultdict(lambda: [0, 0, 0])
prev = seg[0][0]
for s, e, n in seg:
    a = agg[n[:64]]
    a[0] += 1; a[1] += e - s; a[2] += max(0, s - prev)
    prev = max(prev, e)
tot = seg[-1][1] - seg[0][0]
print("last replay: %d launches, %.3f ms from first start to last end, busy %.3f ms" % (len(seg), tot / 1e6, sum(a[1] for a in agg.values()) / 1e6))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2])[:28]:
    print("%-64s n=%4d dur %8.1f us  gaps in front %8.1f us" % (n, a[0], a[1] / 1e3, a[2] / 1e3))
