#!/usr/bin/env python3
"""Development aid: checksum of ggnn_encoder_cell_batch's outputs on fixed random problems -- run under two builds of the
library (GGNN_LIB_PATH) to see whether a variant changes a bit."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _enc_cell_problem

be = default_backend()
h = hashlib.sha256()
for n_dst, ins, F in [(236, [(118, 11, 708), (236, 8, 708)], 8), (118, [(236, 8, 708)], 11), (5, [(9, 8, 11), (5, 8, 0)], 8),
                      (50, [(70, 8, 400), (70, 11, 1300)], 8), (67, [(30, 8, 500)], 11),
                      (20000, [(10000, 11, 60000), (20000, 8, 60000)], 8), (10000, [(20000, 8, 60000)], 11)]:
    prob = _enc_cell_problem(be, np.random.RandomState(n_dst), n_dst, ins, F_dst=F)[0]
    be.encoder_cell_batch([prob])
    torch.cuda.synchronize()
    for t in prob:
        pass
    outs = [t for t in prob if torch.is_tensor(t) and t.dim() == 2 and t.size(1) == 96 and t.size(0) == n_dst]
    for t in outs[-2:]:
        h.update(t.cpu().numpy().tobytes())
print(os.path.basename(os.environ.get("GGNN_LIB_PATH", "libggnn.so")), h.hexdigest()[:32], flush=True)
