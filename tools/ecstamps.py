#!/usr/bin/env python3
"""Development aid: phase times of the encoder cell from the in-kernel stamps of a diagnostic build
(tools/mkvariant.sh stamps enc_cell.hip -DGGNN_STAMPS).  Not part of the product."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GGNN_LIB_PATH", os.path.join(ROOT, "graingraphnn_amd", "libggnn_stamps.so"))
import numpy as np
import torch
from graingraphnn_amd import _lib, synthetic
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _enc_cell_problem

SLOTS, WAVES = 20, 8192
be = default_backend()
lib = ctypes.CDLL(_lib.LIB_PATH)
rs = np.random.RandomState(0)
J = (20000, [(10000, 11, 60000), (20000, 8, 60000)])
Gr = (10000, [(20000, 8, 60000)])
_, hei, _ = synthetic.honeycomb(100, 10, 0)
GJ, JG, JJ = synthetic.EDGE_TYPES
EDGES = {2: [hei[GJ], hei[JJ]], 1: [hei[JG]]}
probs = [_enc_cell_problem(be, rs, n, ins, F_dst=8 if len(ins) == 2 else 11, edges=EDGES[len(ins)])[0] for n, ins in (J, Gr)]
for _ in range(3):
    torch.cuda.synchronize()
    assert lib.ggnn_debug_stamps_clear_enc() == 0
    be.encoder_cell_batch(probs)
torch.cuda.synchronize()
buf = np.zeros(WAVES * SLOTS, dtype=np.uint64)
assert lib.ggnn_debug_stamps_enc(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(WAVES, SLOTS).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
us = lambda x: x / 100.0
print(f"one model (joint + grain), honeycomb: {len(st)} waves, span {us(st[:, 16].max() - t0):.1f} us")
for n_in in (2, 1):
    m = st[st[:, 10] == n_in]
    if not len(m):
        continue
    print(f" destination type with {n_in} incoming edge types: {len(m)} waves")
    for i, nm in ((0, "start"), (1, "prologue done"), (16, "end")):
        r = us(m[:, i] - t0)
        print(f"  {nm:16s} med {np.median(r):7.2f}  min {r.min():7.2f}  max {r.max():7.2f} us")
    for i, nm in ((5, "sum slice A (u4, scores, values)"), (7, "sum lin_l2"), (8, "sum skip"), (9, "sum LSTM"),
                  (4, "  of which slice wait + barrier")):
        r = us(m[:, i])
        print(f"  {nm:34s} med {np.median(r):7.2f}  max {r.max():7.2f} us")
