#!/usr/bin/env python3
"""reads a rocprofv3 kernel-trace CSV of `tools/pipe_trace.py 1` (one step at a time): the LAST whole step by kernel symbol --
launches, summed duration, and CHIP time = duration x min(1, workgroups / 256) (what a launch takes from the other steps in
flight: a 31-workgroup launch of 30 us costs them 3.6 us, a chip-filling one all of its duration)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid // wg))
ev.sort()
first = [i for i, e in enumerate(ev) if "planar_to_cl8" in e[2]]
starts = first[::2]                       # two per step (both frames)
seg = ev[starts[-2]:starts[-1]]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0])
for s, e, n, wgs in seg:
    a = agg[n.replace("(anonymous namespace)::", "").replace("void ", "")[:72]]
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += (e - s) / 1e3 * min(1.0, wgs / 256.0); a[3] = wgs
print("last step: %d launches, %.3f ms first start to last end, summed durations %.3f ms, chip time %.3f ms"
      % (len(seg), (seg[-1][1] - seg[0][0]) / 1e6, sum(a[1] for a in agg.values()) / 1e3, sum(a[2] for a in agg.values()) / 1e3))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-72s n=%3d wgs %6d dur %8.1f us chip %8.1f us" % (n, a[0], a[3], a[1], a[2]))
