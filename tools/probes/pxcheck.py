#!/usr/bin/env python3
"""Development aid: checksum + time of the fused plan's value projection (ggnn_project_batch, GGNN_PRECISION_F16X2 |
GGNN_OUT_BLOCK_MAJOR) at the 10k-grain graph's shapes -- run under two builds of the library (GGNN_LIB_PATH)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from graingraphnn_amd import _lib
from graingraphnn_amd.backend import default_backend

be = default_backend()
rs = np.random.RandomState(3)
f = lambda *shape, lo=-1.0, hi=1.0: torch.from_numpy(rs.uniform(lo, hi, shape).astype(np.float32)).cuda()
probs = []
for M, F, ncols in ((20000, 8, 768), (10000, 11, 384)):
    Fp = (F + 3) & ~3
    wp = f(ncols, Fp + 96, lo=-0.2, hi=0.2)
    wp[:, F:Fp] = 0
    probs.append((f(M, F, lo=0.0), F, f(M, 96), wp, f(ncols), torch.empty(M, ncols, device="cuda"),
                  _lib.GGNN_PRECISION_F16X2 | _lib.GGNN_OUT_BLOCK_MAJOR))
h = hashlib.sha256()
be.project_batch(probs)
torch.cuda.synchronize()
for p in probs:
    h.update(p[5].cpu().numpy().tobytes())
for _ in range(5):
    be.project_batch(probs)
ts = []
for _ in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); be.project_batch(probs); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(os.path.basename(os.environ.get("GGNN_LIB_PATH", "libggnn.so")), h.hexdigest()[:24], f"median {np.median(ts):.1f} us  min {min(ts):.1f} us", flush=True)
