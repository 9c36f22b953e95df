#!/usr/bin/env python3
"""Development aid: time ggnn_decoder_cell_batch alone on the 10k-grain honeycomb's shapes (HIP events on the launch
stream, median of --reps launches; regressor = joint + grain problem, classifier = joint only, both = all three in one
launch).  GGNN_LIB_PATH=graingraphnn_amd/libggnn_<variant>.so selects a `make VARIANT=` build.  Not part of the product."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from graingraphnn_amd import _lib, synthetic
if os.environ.get("GGNN_ABI"):   # timing an older build of the library (its results on this tree's stream image are not checked here)
    _lib.GGNN_ABI_VERSION = int(os.environ["GGNN_ABI"])
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _dec_cell_problem

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--tag", default=os.path.basename(os.environ.get("GGNN_LIB_PATH", "libggnn.so")))
a = ap.parse_args()
be = default_backend()
rs = np.random.RandomState(0)
J = (20000, [(10000, 11, 60000), (20000, 8, 60000)])
Gr = (10000, [(20000, 8, 60000)])
_, hei, _ = synthetic.honeycomb(100, 10, 0)
GJ, JG, JJ = synthetic.EDGE_TYPES
EDGES = {2: [hei[GJ], hei[JJ]], 1: [hei[JG]]}
mk = lambda n, ins: _dec_cell_problem(be, rs, n, ins, F_dst=8 if len(ins) == 2 else 11, edges=EDGES[len(ins)])
pj, pg, pj2 = mk(*J), mk(*Gr), mk(*J)
out = []
for name, probs in (("regressor", [pj, pg]), ("classifier", [pj2]), ("both", [pj, pg, pj2])):
    for _ in range(3):
        be.decoder_cell_batch(probs)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        be.decoder_cell_batch(probs)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    out.append(f"{name} {np.median(ts):7.1f} (min {min(ts):6.1f})")
print(f"{a.tag:28s} " + "   ".join(out) + "  us", flush=True)
