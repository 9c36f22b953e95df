"""Development probe (round 5): the fused plan's value projection (joints 768 + grains 384 columns, one launch) with six and
with three products per k-step.  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from graingraphnn_amd import _lib
from graingraphnn_amd.backend import default_backend

be = default_backend()
dev = "cuda"
def prob(M, F, ncols, prec):
    Fp = (F + 3) & ~3
    return (torch.rand(M, F, device=dev), F, torch.tanh(torch.randn(M, 96, device=dev)), torch.randn(ncols, Fp + 96, device=dev) * 0.1,
            torch.randn(ncols, device=dev), torch.empty(M, ncols, device=dev), prec)
for name, prec in (("six products", 0), ("three products", _lib.GGNN_PRECISION_F16X2)):
    probs = [prob(20000, 8, 768, prec), prob(10000, 11, 384, prec)]
    for _ in range(5):
        be.project_batch(probs)
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); be.project_batch(probs); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"{name:16s} med {np.median(ts):6.1f} us  min {min(ts):6.1f} us", flush=True)
