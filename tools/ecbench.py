#!/usr/bin/env python3
"""Development aid: time ggnn_encoder_cell_batch alone on the 10k-grain honeycomb's shapes (HIP events on the launch
stream, median of --reps launches; one model = joint + grain problem, both = the four problems of regressor and
classifier in one launch).  GGNN_LIB_PATH=graingraphnn_amd/libggnn_<variant>.so selects a tools/mkvariant.sh build.
Not part of the product."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from graingraphnn_amd import synthetic
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _enc_cell_problem

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
mk = lambda n, ins: _enc_cell_problem(be, rs, n, ins, F_dst=8 if len(ins) == 2 else 11, edges=EDGES[len(ins)])[0]
p = [mk(*J), mk(*Gr), mk(*J), mk(*Gr)]
out = []
for name, probs in (("one model", p[:2]), ("joint only", p[:1]), ("grain only", p[1:2]), ("both models", p)):
    for _ in range(3):
        be.encoder_cell_batch(probs)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        be.encoder_cell_batch(probs)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    out.append(f"{name} {np.median(ts):6.1f} (min {min(ts):5.1f})")
print(f"{a.tag:24s} " + "   ".join(out) + "  us", flush=True)
