#!/usr/bin/env python3
"""Development aid: time ggnn_lstm_epilogue alone on cfg3-sized problems (workspace layout: gate
stride padded to 32 floats), single problems and the batches a rollout step launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graingraphnn_amd import _lib
from graingraphnn_amd.backend import default_backend
from graingraphnn_amd.packing import bf16_planes

be = default_backend()
dev = "cuda"


def problem(N, Ka, G):
    gs = (Ka + 31) // 32 * 32
    agg = torch.randn(N, G * gs, device=dev)
    w2 = torch.randn(G, 96, Ka, device=dev) * 0.1
    pd = torch.randn(N, G * 96, device=dev)
    c_in = torch.randn(N, 96, device=dev)
    h, c = torch.empty(N, 96, device=dev), torch.empty(N, 96, device=dev)
    mode = _lib.MODE_LSTM if G == 4 else _lib.MODE_LSTM_H0
    return (agg, w2, pd, 0, c_in if G == 4 else None, h, c, None, G, mode, bf16_planes(w2), gs)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    ts = []
    flush = torch.empty(64 << 20, device=dev)  # 256 MB: the operands leave the Infinity Cache between runs
    for _ in range(reps):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return np.median(ts), min(ts)


def bytes_of(p):
    agg, w2, pd, _, c_in, h, c, _, G, _, pl, gs = p
    N = agg.size(0)
    return 4 * N * (G * gs + G * 96 + (96 if c_in is not None else 0) + 2 * 96)


for N, Ka, G in ((20000, 196, 4), (10000, 100, 4), (20000, 196, 3), (10000, 100, 3), (2086, 196, 4), (1043, 100, 4)):
    p = problem(N, Ka, G)
    med, mn = timeit(lambda: be.lstm_epilogue(*p))
    fl = 2.0 * N * G * 96 * Ka
    print(f"N={N} Ka={Ka} G={G}: med {med:6.1f} us  min {mn:6.1f} us   {fl / med / 1e6:6.1f} TFLOP/s fp32-equivalent, "
          f"{bytes_of(p) / med / 1e3:6.0f} GB/s of operand bytes")
for name, shapes in (("decoder R+C: joint, grain, joint", ((20000, 196, 4), (10000, 100, 4), (20000, 196, 4))),
                     ("decoder R: joint, grain", ((20000, 196, 4), (10000, 100, 4))),
                     ("encoder R+C: joint, grain, joint, grain", ((20000, 196, 3), (10000, 100, 3), (20000, 196, 3), (10000, 100, 3))),
                     ("cfg2 decoder R+C", ((2086, 196, 4), (1043, 100, 4), (2086, 196, 4)))):
    ps = [problem(*s) for s in shapes]
    med, mn = timeit(lambda: be.lstm_epilogue_batch(ps))
    fl = sum(2.0 * N * G * 96 * Ka for N, Ka, G in shapes)
    print(f"batch {name}: med {med:6.1f} us  min {mn:6.1f} us   {fl / med / 1e6:6.1f} TFLOP/s, "
          f"{sum(bytes_of(p) for p in ps) / med / 1e3:6.0f} GB/s")
