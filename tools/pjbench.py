#!/usr/bin/env python3
"""Development aid: time ggnn_project alone for a range of M (prologue vs per-tile cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graingraphnn_amd.backend import default_backend

be = default_backend()
dev = "cuda"
F, ncols = int(os.environ.get("F", 8)), int(os.environ.get("NCOLS", 2688))
Fp = (F + 3) & ~3
for M in (16, 128, 2048, 4096, 20000, 40000):
    x = torch.randn(M, F, device=dev)
    h = torch.randn(M, 96, device=dev)
    wp = torch.randn(ncols, Fp + 96, device=dev) * 0.1
    bp = torch.randn(ncols, device=dev)
    out = torch.empty(M, ncols, device=dev)
    for _ in range(5):
        be.project(x, F, h, wp, bp, out)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); be.project(x, F, h, wp, bp, out); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    extra = ""
    if os.environ.get("PJ_CLOCK"):
        c, r, n, pro = out[0, :4].tolist()
        extra = f"  wave0: {c:.0f} cycles, {r / 100:.1f} us => {c / max(r, 1) * 100:.0f} MHz, {n:.0f} tile span, prologue {pro / 100:.1f} us"
    print(f"M={M:6d}  med {np.median(ts):7.1f} us  min {min(ts):7.1f} us{extra}")
