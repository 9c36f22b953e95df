#!/usr/bin/env python3
"""Development aid: phase times of the fused decoder cell from the in-kernel stamps of the diagnostic build
(make -C graingraphnn_amd/csrc STAMPS=1).  Not part of the product."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GGNN_LIB_PATH", os.path.join(ROOT, "graingraphnn_amd", "libggnn_stamps.so"))
import numpy as np
import torch
from graingraphnn_amd import _lib
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _dec_cell_problem

SLOTS, WAVES = 20, 8192
be = default_backend()
lib = ctypes.CDLL(_lib.LIB_PATH)
rs = np.random.RandomState(0)
J = (20000, [(10000, 11, 60000), (20000, 8, 60000)])
Gr = (10000, [(20000, 8, 60000)])
from graingraphnn_amd import synthetic
_, hei, _ = synthetic.honeycomb(100, 10, 0)          # the benchmark's structure: neighbours are near in index
GJ, JG, JJ = synthetic.EDGE_TYPES
EDGES = {2: [hei[GJ], hei[JJ]], 1: [hei[JG]]}
for name, shapes in (("regressor (joint + grain), honeycomb", [J, Gr]), ("classifier (joint), honeycomb", [J]),
                     ("regressor, random sources", [J, Gr])):
    probs = [_dec_cell_problem(be, rs, n, ins, F_dst=8 if len(ins) == 2 else 11,
                               edges=None if "random" in name else EDGES[len(ins)]) for n, ins in shapes]
    for _ in range(3):
        torch.cuda.synchronize()
        assert lib.ggnn_debug_stamps_clear_dec() == 0
        be.decoder_cell_batch(probs)
    torch.cuda.synchronize()
    buf = np.zeros(WAVES * SLOTS, dtype=np.uint64)
    assert lib.ggnn_debug_stamps_dec(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf.reshape(WAVES, SLOTS).astype(np.int64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    us = lambda x: x / 100.0
    print(f"\n{name}: {len(st)} waves, span {us(st[:, 16].max() - t0):.1f} us")
    for n_in in (2, 1):
        m = st[st[:, 10] == n_in]
        if not len(m):
            continue
        print(f" destination type with {n_in} incoming edge types: {len(m)} waves")
        for i, nm in ((0, "start"), (1, "prologue done"), (16, "end")):
            r = us(m[:, i] - t0)
            print(f"  {nm:16s} med {np.median(r):7.2f}  min {r.min():7.2f}  max {r.max():7.2f} us")
        life = us(m[:, 16] - m[:, 0])
        print(f"  wave life        med {np.median(life):7.2f}  max {life.max():7.2f} us")
        clk = (m[:, 18] - m[:, 17]) / np.maximum(m[:, 16] - m[:, 0], 1) * 100.0   # shader clocks per 100 MHz tick
        print(f"  shader clock     med {np.median(clk):7.0f} MHz (cycle counter / real-time counter over the wave's life)")
        for i, nm in ((5, "sum P1 (scores)"), (6, "sum P2 (sweep)"), (7, "sum P3 (lin_l2)"), (8, "sum P4 (skip)"),
                      (9, "sum LSTM"), (4, "  of which slice wait + barrier"), (11, "  of which slice DMA issue")):
            r = us(m[:, i])
            print(f"  {nm:32s} med {np.median(r):7.2f}  max {r.max():7.2f} us")
