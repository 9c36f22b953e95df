#!/usr/bin/env python3
"""Development aid: phase times of the fused encoder cell from the in-kernel stamps of the diagnostic build
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
from test_hip_parity import _enc_cell_problem

SLOTS, WAVES = 20, 8192
be = default_backend()
lib = ctypes.CDLL(_lib.LIB_PATH)
rs = np.random.RandomState(0)
J = (20000, [(10000, 11, 60000), (20000, 8, 60000)])
Gr = (10000, [(20000, 8, 60000)])
for name, shapes in (("one model", [J, Gr]), ("R + C", [J, Gr, J, Gr])):
    probs = [_enc_cell_problem(be, rs, n, ins, regular=True)[0] for n, ins in shapes]
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
    print(f"\n{name}: {len(st)} waves, span {us(st[:, 16].max() - t0):.1f} us")
    for i, nm in ((0, "start"), (1, "weights in LDS"), (2, "headers landed"), (3, "ring filled"), (16, "end")):
        r = us(st[:, i] - t0)
        print(f"  {nm:16s} med {np.median(r):7.2f}  min {r.min():7.2f}  max {r.max():7.2f} us")
    for i, nm in ((4, "sum wait"), (5, "sum compute"), (6, "sum issue"), (7, "sum gemm")):
        r = us(st[:, i])
        print(f"  {nm:16s} med {np.median(r):7.2f}  max {r.max():7.2f} us")
    print(f"  steps per wave   med {np.median(st[:, 8]):.0f} max {st[:, 8].max()}   tiles per wave med {np.median(st[:, 9]):.0f} max {st[:, 9].max()}")
    # end time per workgroup, grouped in runs of equal step counts (= combinations, in launch order)
    wg_end = us(st[:, 16] - t0).reshape(-1, 8).max(1)
    wg_steps = st[:, 8].reshape(-1, 8).sum(1)
    print("  workgroup end times (us), 16 per row:")
    for i in range(0, len(wg_end), 16):
        print("   ", " ".join(f"{v:5.1f}" for v in wg_end[i:i + 16]), " | steps", wg_steps[i])
    n = np.maximum(st[:, 8], 1)
    print(f"  per step: wait {np.median(us(st[:, 4]) / n):.3f}  compute {np.median(us(st[:, 5]) / n):.3f}  issue {np.median(us(st[:, 6]) / n):.3f} us;"
          f"  per tile gemm {np.median(us(st[:, 7]) / np.maximum(st[:, 9], 1)):.3f} us")
