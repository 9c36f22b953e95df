#!/usr/bin/env python3
"""Development aid: time ggnn_wgrad on the shapes of a cfg3 training step against the library product.
    python tools/wgbench.py [libggnn_variant.so]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    os.environ["GGNN_LIB_PATH"] = os.path.abspath(sys.argv[1])
import torch  # noqa: E402

from graingraphnn_amd.backend import default_backend  # noqa: E402

be = default_backend()
shapes = [(20000, 2112, 108, 1), (10000, 1248, 112, 1), (20000, 96, 224, 4), (10000, 96, 128, 4),
          (20000, 480, 12, 1), (20000, 96, 224, 3), (20000, 4, 100, 1)]
for K, M, Nc, batch in shapes:
    a = torch.randn(batch, K, M, device="cuda")
    b = torch.randn(K, batch * Nc, device="cuda")
    f = lambda: be.wgrad(a, b, K, M, Nc, M, batch * Nc, batch=batch, a_bstride=K * M, b_bstride=Nc)
    ref = lambda: torch.stack([a[k].t() @ b[:, k * Nc:(k + 1) * Nc] for k in range(batch)])
    out = []
    for fn in (f, ref):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 20 * 1e6)
    S = be.lib.ggnn_wgrad_splits(K, M, Nc, batch)
    print(f"K {K:6d} M {M:5d} Nc {Nc:4d} batch {batch}: wgrad (+ sum over {S:3d} splits) {out[0]:7.1f} us   library {out[1]:7.1f} us   "
          f"{2 * K * M * Nc * batch / out[0] / 1e6:6.1f} TFLOP/s")
