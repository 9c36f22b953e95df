#!/usr/bin/env python3
"""Development aid: the timeline of ONE training step (the one before the last) from a rocprofv3 kernel trace (rocpd .db)
of tests/bench_train_step.py: start offset, duration and the idle gap in front of every kernel.  Steps are cut at
ggnn::edge_prepare_kernel (one per step, the step's first hand-written kernel).
  python tools/train_timeline.py <results.db>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
cuts = [i for i, r in enumerate(rows) if "edge_prepare" in r[0]]
a, b = cuts[-3], cuts[-2]
# a step starts a few torch kernels in front of edge_prepare; shift the window to the largest idle gap before it
step = rows[a:b]
t0 = step[0][1]
busy = 0.0
end_prev = t0
tot_gap = 0.0
for name, s, e in step:
    m = re.search(r"(ggnn::\w+(<[^>]*>)?|copyBuffer|fillBuffer\w*|FillFunctor|direct_copy|CatArray\w+|reduce_kernel|index_\w+|"
                  r"multi_tensor\w*|Cijk\w{0,30}|CUDAFunctor_add|MulFunctor|tanh\w*|threshold|clamp|pow|neg|MeanOps|fused_adam\w*)", name)
    k = m.group(0) if m else re.sub(r"void at::native::|\(anonymous namespace\)::", "", name)[:50]
    gap = (s - end_prev) / 1e3
    if gap > 0:
        tot_gap += gap
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  gap {gap:6.1f}  {k}")
    busy += (e - s) / 1e3
    end_prev = max(end_prev, e)
print(f"{len(step)} kernels, span {(end_prev - t0) / 1e3:.0f} us, kernel time {busy:.0f} us, idle gaps {tot_gap:.0f} us")
