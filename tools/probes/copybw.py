#!/usr/bin/env python3
"""Development aid: what a plain streaming kernel reaches on this box at the row GEMMs' sizes (cold caches: a 512 MB fill
between calls) -- torch's copy (read + write) and sum (read only), HIP events, median of 20."""
import numpy as np
import torch
big = torch.empty(128 * 1024 * 1024, device="cuda")
def timeit(fn, reps=20):
    ts = []
    for _ in range(reps + 3):
        big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts[3:]))
for mb in (16, 36, 72, 102, 170, 512):
    n = mb * 1024 * 1024 // 4
    a, b = torch.randn(n, device="cuda"), torch.empty(n, device="cuda")
    tc, ts = timeit(lambda: b.copy_(a)), timeit(lambda: a.sum())
    print(f"{mb:4d} MB: copy {tc:6.1f} us = {2 * mb * 1.048576 / tc * 1e3 / 1e3:5.2f} TB/s (read + write)   sum {ts:6.1f} us = {mb * 1.048576 / ts:5.2f} TB/s", flush=True)
