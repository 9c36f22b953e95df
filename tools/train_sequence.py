#!/usr/bin/env python3
"""Development aid: the kernel sequence of ONE training step from a rocprofv3 kernel trace (rocpd .db) of
`tests/bench_train_step.py --steps N`:  python tools/train_sequence.py <results.db> [steps] [warm]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = db.execute("select name, start, end from kernels order by start").fetchall()
per = len(rows) // (steps + warm)
last = rows[-per * 2:-per]   # the step before the last (the last one carries the end-of-run copies)


def short(n):
    m = re.search(r"(ggnn::\w+|copyBuffer|fillBuffer\w*|FillFunctor|direct_copy|CatArray\w+|reduce_kernel|index_\w+|"
                  r"multi_tensor\w*|Cijk|CUDAFunctor_add|MulFunctor|tanh\w*|threshold|clamp|pow|neg|MeanOps|fused_adam\w*)", n)
    return m.group(0)[:28] if m else re.sub(r"void at::native::|\(anonymous namespace\)::", "", n)[:28]


out, prev, cnt = [], None, 0
for name, s, e in last:
    k = short(name)
    if k == prev:
        cnt += 1
    else:
        if prev is not None:
            out.append(prev + (f" x{cnt}" if cnt > 1 else ""))
        prev, cnt = k, 1
out.append(prev + (f" x{cnt}" if cnt > 1 else ""))
print(f"{per} kernels per step, {sum(e - s for _, s, e in last) / 1e3:.0f} us of kernel time")
print(" | ".join(out))
