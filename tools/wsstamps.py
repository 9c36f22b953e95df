#!/usr/bin/env python3
"""Development aid: where the waves of the specialised-wave decoder cell (dec_cell_ws.hip) spend their time, from
the in-kernel stamps of the diagnostic build (make -C graingraphnn_amd/csrc STAMPS=1).  Not part of the product."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GGNN_LIB_PATH", os.path.join(ROOT, "graingraphnn_amd", "libggnn_stamps.so"))
os.environ["GGNN_DC_KERNEL"] = "ws"
import numpy as np
import torch
from graingraphnn_amd import _lib, synthetic
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _dec_cell_problem

SLOTS, WAVES = 20, 8192
be = default_backend()
lib = ctypes.CDLL(_lib.LIB_PATH)
rs = np.random.RandomState(0)
J = (20000, [(10000, 11, 60000), (20000, 8, 60000)])
Gr = (10000, [(20000, 8, 60000)])
_, hei, _ = synthetic.honeycomb(100, 10, 0)
GJ, JG, JJ = synthetic.EDGE_TYPES
EDGES = {2: [hei[GJ], hei[JJ]], 1: [hei[JG]]}
for name, shapes in (("regressor (joint + grain), honeycomb", [J, Gr]), ("classifier (joint), honeycomb", [J])):
    probs = [_dec_cell_problem(be, rs, n, ins, F_dst=8 if len(ins) == 2 else 11, edges=EDGES[len(ins)]) for n, ins in shapes]
    for _ in range(3):
        torch.cuda.synchronize()
        assert lib.ggnn_debug_stamps_clear_decws() == 0
        be.decoder_cell_batch(probs)
    torch.cuda.synchronize()
    buf = np.zeros(WAVES * SLOTS, dtype=np.uint64)
    assert lib.ggnn_debug_stamps_decws(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf.reshape(WAVES, SLOTS).astype(np.int64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    us = lambda x: x / 100.0
    print(f"\n{name}: {len(st)} waves, span {us(st[:, 16].max() - t0):.1f} us")
    for n_in in (2, 1):
        # (waits inside the slice loop are counted in POLLS -- ~64-cycle sleep + an LDS read each, SCALE below --
        # not clock reads: s_memrealtime per slice costs more than the slice and distorted the first tables)
        for role, rn, cols in ((1, "M", ((4, "polls for slices (L)"), (5, "wait for aggregates (S), us"))),
                               (2, "S", ((4, "wait for u (M), us"),)),
                               (3, "L", ((4, "polls for a free buffer (M)"),))):
            m = st[(st[:, 6] == n_in) & (st[:, 10] == role)]
            if not len(m):
                continue
            life = us(m[:, 16] - m[:, 0])
            print(f" n_in {n_in} {rn}-waves ({len(m)}): life med {np.median(life):7.2f} max {life.max():7.2f} us; "
                  f"end med {np.median(us(m[:, 16] - t0)):7.2f}")
            for i, nm in cols:
                r = m[:, i].astype(float) if "polls" in nm else us(m[:, i])
                print(f"    {nm:34s} med {np.median(r):8.2f}  max {r.max():8.2f}")
