"""Development probe (round 4; see profiles/r4_dec_cell_ablations.txt / DESIGN.md section 7).  Not part of the product."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from graingraphnn_amd import synthetic
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _dec_cell_problem
be = default_backend(); rs = np.random.RandomState(0)
J = (20000, [(10000, 11, 60000), (20000, 8, 60000)]); Gr = (10000, [(20000, 8, 60000)])
_, hei, _ = synthetic.honeycomb(100, 10, 0)
GJ, JG, JJ = synthetic.EDGE_TYPES
EDGES = {2: [hei[GJ], hei[JJ]], 1: [hei[JG]]}
mk = lambda n, ins: _dec_cell_problem(be, rs, n, ins, F_dst=8 if len(ins) == 2 else 11, edges=EDGES[len(ins)])
pj, pg, pj2 = mk(*J), mk(*Gr), mk(*J)
sets = {"both (3 problems)": [pj, pg, pj2]}
if "half6" in os.environ.get("GGNN_LIB_PATH", ""):
    sets["six half problems, joints first"] = [pj, pj2, pj, pj2, pg, pg]
for name, probs in sets.items():
    for _ in range(3): be.decoder_cell_batch(probs)
    torch.cuda.synchronize(); ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); be.decoder_cell_batch(probs); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(os.path.basename(os.environ.get("GGNN_LIB_PATH", "libggnn.so")), name, f"{np.median(ts):.1f} us")
