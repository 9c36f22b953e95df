#!/usr/bin/env python3
"""Development aid: checksum of ggnn_decoder_cell_batch's outputs on fixed random problems (hub rows, ragged tiles, empty edge
types, cfg3 sizes) -- run under two builds of the library (GGNN_LIB_PATH) to see whether a variant changes a bit."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _dec_cell_problem

be = default_backend()
h = hashlib.sha256()
for n_dst, ins, hub in [(236, [(118, 11, 708), (236, 8, 708)], 0), (118, [(236, 8, 708)], 0), (5, [(9, 8, 11), (5, 8, 0)], 0),
                        (50, [(70, 8, 400), (70, 11, 1300)], 37), (50, [(70, 11, 1300)], 900), (67, [(30, 8, 500)], 0),
                        (20000, [(10000, 11, 60000), (20000, 8, 60000)], 0), (10000, [(20000, 8, 60000)], 0)]:
    prob = _dec_cell_problem(be, np.random.RandomState(n_dst + hub), n_dst, ins, hub)
    be.decoder_cell_batch([prob])
    torch.cuda.synchronize()
    h.update(prob[6].cpu().numpy().tobytes())
    h.update(prob[7].cpu().numpy().tobytes())
print(os.path.basename(os.environ.get("GGNN_LIB_PATH", "libggnn.so")), h.hexdigest()[:32], flush=True)
