#!/usr/bin/env python3
"""Development aid: per-step kernel timeline from a rocprofv3 --kernel-trace CSV of bench.py
(python tools/timeline.py <kernel_trace.csv>): wall, union-busy and per-kernel time per step, one step listed."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
            for r in rows if "ggnn" in r["Kernel_Name"])
ups = [i for i, e in enumerate(ev) if "step_update" in e[2] or "heads_regressor_update" in e[2]]
a, b = ups[20], ups[28]
win = ev[a:b]
t0, t1 = win[0][0], win[-1][1]
busy, cs, ce = 0, win[0][0], win[0][1]
for s, e, _ in win[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("8 steps: wall %.1f us/step, some kernel running %.1f us/step, idle %.1f us/step" % ((t1 - t0) / 8e3, busy / 8e3, (t1 - t0 - busy) / 8e3))
tot = collections.Counter()
for s, e, n in win:
    tot[n] += e - s
for n, v in tot.most_common():
    print("  %-45s %.1f us/step" % (n, v / 8e3))
base = ev[ups[22]][0]
for s, e, n in ev[ups[22]:ups[23] + 1]:
    print("   start %8.1f  dur %8.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, n))
