#!/usr/bin/env python3
"""Development aid: phase timeline of the gate kernel from the in-kernel stamps of the diagnostic
build (make -C graingraphnn_amd/csrc STAMPS=1; csrc/stamps.h).  Not part of the product.

    GGNN_LIB_PATH=graingraphnn_amd/libggnn_stamps.so python tools/stamps.py
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GGNN_LIB_PATH", os.path.join(ROOT, "graingraphnn_amd", "libggnn_stamps.so"))
import numpy as np
import torch
from graingraphnn_amd import _lib
from graingraphnn_amd.backend import default_backend
from graingraphnn_amd.packing import bf16_planes

SLOTS, WAVES = 20, 8192
be = default_backend()
lib = ctypes.CDLL(_lib.LIB_PATH)
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


names = ["start", "prologue issued", "slice 0 in LDS"] + [f"k{ks} {w}" for ks in range(6) for w in ("mfma done", "barrier")] + ["exchange barrier", "end"]
flush = torch.empty(64 << 20, device=dev)
for shapes in (((20000, 196, 4),), ((10000, 100, 4),), ((20000, 196, 3),), ((20000, 196, 4), (10000, 100, 4))):
    ps = [problem(*s) for s in shapes]
    nks = (shapes[0][1] - 4) // 32
    for cold in (True, False):
        for _ in range(3):
            if cold:
                flush.zero_()
            torch.cuda.synchronize()
            assert lib.ggnn_debug_stamps_clear() == 0
            be.lstm_epilogue_batch(ps)
        torch.cuda.synchronize()
        buf = np.zeros(WAVES * SLOTS, dtype=np.uint64)
        assert lib.ggnn_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        st = buf.reshape(WAVES, SLOTS).astype(np.int64)
        n_w = sum((N + 15) // 16 for N, _, _ in shapes)  # upper bound on waves that ran; keep those with a start stamp
        live = st[:, 0] > 0
        st = st[live]
        t0 = st[:, 0].min()
        print(f"\n{shapes} {'cold (256 MB flushed)' if cold else 'warm (back to back)'}: {live.sum()} waves, "
              f"kernel span {(st[:, 16].max() - t0) / 100:.1f} us")
        idx = [0, 1, 2] + [3 + i for i in range(2 * nks)] + [15, 16]
        prev = None
        for i in idx:
            rel = (st[:, i] - t0) / 100.0
            d = "" if prev is None else f"   delta med {np.median((st[:, i] - st[:, prev]) / 100.0):6.2f}"
            print(f"  {names[i]:18s} med {np.median(rel):7.2f}  min {rel.min():7.2f}  max {rel.max():7.2f} us{d}")
            prev = i
