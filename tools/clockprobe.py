#!/usr/bin/env python3
"""Is the projection GEMM MFMA-bound at a power-throttled clock?  Time ggnn_project on random
vs all-zero operands (same instruction stream; zeros let the chip hold a higher clock --
MI355X_MICROARCH.md 'DVFS give-back')."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graingraphnn_amd.backend import default_backend

be = default_backend()
dev = "cuda"
M, F, N = 20000, 8, 2688
for name, fill in (("random", None), ("zeros", 0.0), ("random", None), ("zeros", 0.0)):
    x = torch.rand(M, F, device=dev) if fill is None else torch.zeros(M, F, device=dev)
    h = torch.randn(M, 96, device=dev) if fill is None else torch.zeros(M, 96, device=dev)
    w = torch.randn(N, 104, device=dev) * 0.1 if fill is None else torch.zeros(N, 104, device=dev)
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(5):
        be.project(x, F, h, w, b, out)
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, c in e:
        a.record(); be.project(x, F, h, w, b, out); c.record()
    torch.cuda.synchronize()
    t = np.median([a.elapsed_time(c) for a, c in e]) * 1e3
    print(f"{name:7s} {t:7.1f} us  {2*M*104*N/t/1e6:6.1f} TFLOP/s")
