#!/usr/bin/env python3
"""Per-kernel breakdown of one training step from a rocprofv3 kernel trace (rocpd .db) of
`tests/bench_train_step.py --steps N`:  python tools/train_kernels.py <results.db> [steps] [warm]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = db.execute("select name, start, end from kernels order by start").fetchall()
per = len(rows) // (steps + warm)
last = rows[-per * steps:]
cat, agg = collections.defaultdict(lambda: [0, 0.0]), collections.defaultdict(lambda: [0, 0.0])
for name, s, e in last:
    d = (e - s) / 1e3
    if name.startswith("Cijk"):
        c = "library GEMM"
    elif "ggnn::" in name:
        c = "ggnn (hand-written)"
    elif "reduce" in name:
        c = "reduce"
    elif "Cat" in name:
        c = "cat"
    elif "multi_tensor" in name:
        c = "optimizer (foreach)"
    elif "elementwise" in name or "fillBuffer" in name:
        c = "pointwise / copy / fill"
    else:
        c = "other"
    cat[c][0] += 1
    cat[c][1] += d
    m = re.search(r"(ggnn::\w+(<[^>]*>)?|direct_copy|FillFunctor|CatArray\w+|CUDAFunctor_add|MulFunctor|"
                  r"sigmoid\w*|tanh\w*|multi_tensor\w*|fillBuffer\w*|reduce_kernel|Cijk_\w{0,40})", name)
    k = m.group(1) if m else name[:60]
    agg[k][0] += 1
    agg[k][1] += d
tot = sum(v[1] for v in cat.values())
print(f"{per} kernels per step, {tot / steps:.0f} us of kernel time per step")
for c, v in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print(f"  {c:26s} {v[1] / steps:8.1f} us {v[0] / steps:6.1f} launches")
print()
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1] / steps:9.1f} us {v[0] / steps:6.1f}x  {k}")
