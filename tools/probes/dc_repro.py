"""Development probe: run-to-run differences of ggnn_decoder_cell_batch on one random problem."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _dec_cell_problem
be = default_backend()
for (n_dst, ins, hub) in [(118, [(236, 8, 708)], 0), (236, [(118, 11, 708), (236, 8, 708)], 0), (50, [(70, 11, 1300)], 900)]:
    rs = np.random.RandomState(n_dst + 7 * len(ins) + hub)
    prob = _dec_cell_problem(be, rs, n_dst, ins, hub)
    outs = []
    for k in range(6):
        prob[6].fill_(float("nan")); prob[7].fill_(float("nan"))
        be.decoder_cell_batch([prob])
        torch.cuda.synchronize()
        outs.append((prob[6].clone(), prob[7].clone()))
    for k in range(1, 6):
        for name, a, b in (("h", outs[0][0], outs[k][0]), ("c", outs[0][1], outs[k][1])):
            d = (a != b)
            if bool(d.any()):
                rows = d.any(1).nonzero().view(-1).tolist()
                cols = d.any(0).nonzero().view(-1).tolist()
                print(n_dst, len(ins), f"run {k} {name}: {int(d.sum())} elements differ, rows {rows[:20]} cols {cols[:24]} max |d| {float((a-b).abs().max()):.3e}")
    print(n_dst, len(ins), "done")
