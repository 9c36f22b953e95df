#!/usr/bin/env python3
"""Development aid: time the encoder cell alone at cfg3 sizes -- fused (ggnn_encoder_cell_batch) against
the sweep + gate-epilogue pair it replaces -- for one model and for the regressor + classifier batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _enc_cell_problem

be = default_backend()
dev = "cuda"


def timeit(fn, reps=30, cold=True):
    for _ in range(5):
        fn()
    ts = []
    flush = torch.empty(64 << 20, device=dev)
    for _ in range(reps):
        if cold:
            flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return np.median(ts), min(ts)


rs = np.random.RandomState(0)
J = (20000, [(10000, 11, 60000), (20000, 8, 60000)])
Gr = (10000, [(20000, 8, 60000)])
for name, shapes in (("joint", [J]), ("grain", [Gr]), ("one model (joint + grain)", [J, Gr]),
                     ("R + C (2 x joint + grain)", [J, Gr, J, Gr])):
    probs = [_enc_cell_problem(be, rs, n, ins, regular=True) for n, ins in shapes]
    fused = [p[0] for p in probs]
    sweeps = [s for p in probs for s in p[1]]
    gates = [p[2] for p in probs]
    for cold in (True, False):
        mf, nf = timeit(lambda: be.encoder_cell_batch(fused), cold=cold)
        ms, ns = timeit(lambda: be.aggregate_enc_batch(sweeps), cold=cold)
        mg, ng = timeit(lambda: be.lstm_epilogue_batch(gates), cold=cold)
        print(f"{name:28s} {'cold' if cold else 'warm'}: fused {mf:6.1f} us (min {nf:6.1f})   "
              f"sweep {ms:6.1f} + gates {mg:6.1f} = {ms + mg:6.1f} us", flush=True)
