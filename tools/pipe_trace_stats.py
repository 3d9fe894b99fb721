#!/usr/bin/env python3
"""reads a rocprofv3 kernel-trace CSV of tools/pipe_trace.py: concurrency, stretch, idle time over the last replays"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r.get("Workgroup_Size_X", 1) or 1)
    gx = int(r.get("Grid_Size_X", 1) or 1) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    ev.append((s, e, r["Kernel_Name"], r.get("Queue_Id", "?"), gx // max(1, wg * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))))
ev.sort()
t_end = ev[-1][1]
# the last 40 % of the trace: steady state of the pipelined replays
t0 = ev[0][0] + int(0.6 * (t_end - ev[0][0]))
sel = [e for e in ev if e[0] >= t0]
span = sel[-1][1] - sel[0][0]
busy = sum(e[1] - e[0] for e in sel)
pts = sorted([(e[0], 1) for e in sel] + [(e[1], -1) for e in sel])
cur, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[cur] += t - last
    cur += d
    last = t
print("window %.2f ms, %d dispatches, sum of durations %.2f ms -> average concurrency %.2f" % (span / 1e6, len(sel), busy / 1e6, busy / span))
print("time share by number of kernels running:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
print("queues:", collections.Counter(e[3] for e in sel))
agg = collections.defaultdict(lambda: [0, 0, 0])
for s, e, n, q, wgs in sel:
    a = agg[n[:70]]
    a[0] += 1; a[1] += e - s; a[2] = wgs
for n, (c, d, w) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
    print("%-72s n=%4d avg %7.1f us  wgs %6d  share %.3f" % (n, c, d / c / 1e3, w, d / busy))
