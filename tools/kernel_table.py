#!/usr/bin/env python3
"""Per-step kernel table from rocprofv3 *_kernel_stats.csv files of `bench.py --profile` runs."""
import csv, sys
for f in sys.argv[1:]:
    print(f)
    tot = 0.0
    rows = list(csv.DictReader(open(f)))
    # steps of the run (warm-up, settle replays, timed): the regressor's head kernel runs once per step
    STEPS = max(int(r["Calls"]) for r in rows if "heads_regressor" in r["Name"])
    for r in rows:
        if "ggnn" not in r["Name"] or "csr_" in r["Name"]:
            continue
        n = r["Name"].split("(")[0].replace("void ", "")
        per_step = float(r["TotalDurationNs"]) / 1e3 / STEPS
        tot += per_step
        print(f"  {n:42s} calls/step {int(r['Calls']) / STEPS:4.1f}  avg {float(r['AverageNs']) / 1e3:7.1f} us  "
              f"min {float(r['MinNs']) / 1e3:6.1f}  max {float(r['MaxNs']) / 1e3:6.1f}   per step {per_step:7.1f} us")
    print(f"  sum of kernel time per step {tot:.1f} us")
