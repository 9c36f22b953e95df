#!/usr/bin/env python3
"""Development aid: ggnn_rowgemm on the training path's shapes at the 10k-grain graph against the BLAS call it
replaces (HIP events, median of 20; cold = a 512 MB fill between calls).  Not part of the product."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from graingraphnn_amd.backend import default_backend

be = default_backend()
dev = "cuda"
big = torch.empty(128 * 1024 * 1024, device=dev)


def timeit(fn, cold=True, reps=20):
    ts = []
    for _ in range(reps + 3):
        if cold:
            big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts[3:]))


shapes = [("gate GEMM, joints (dec)", 20000, 224, 96, 4, False), ("gate GEMM, grains (dec)", 10000, 128, 96, 4, False),
          ("g_agg, joints (dec)", 20000, 96, 224, 4, True), ("g_agg, grains (dec)", 10000, 96, 128, 4, True),
          ("g_h, joints", 20000, 2112, 96, 1, True), ("g_h, grains", 10000, 1248, 96, 1, True)]
for name, M, K, n_out, B, tr in shapes:
    a = torch.randn(B, M, K, device=dev)
    w = torch.randn(B, K, n_out, device=dev) * 0.1 if tr else torch.randn(B, n_out, K, device=dev) * 0.1
    out = torch.empty(B, M, n_out, device=dev)
    lib = (lambda: torch.bmm(a, w, out=out)) if tr else (lambda: torch.bmm(a, w.transpose(1, 2), out=out))
    r = [timeit(lambda: be.rowgemm(a, w, out, K, n_out, batch=B, transposed=tr)),
         timeit(lambda: be.rowgemm(a, w, out, K, n_out, batch=B, transposed=tr, bf16=True)), timeit(lib)]
    print(f"{name:26s} M={M:6d} K={K:5d} n={n_out:4d} x{B}: rowgemm fp32 {r[0]:7.1f}  bf16 {r[1]:7.1f}  library fp32 {r[2]:7.1f} us", flush=True)
