#!/usr/bin/env python3
"""The reference's (G, R) -> span lookup table as plain arrays.

`graph_trajectory.py --mode=generate` does not take the frame span as an argument: it looks it up
(graph_trajectory.py:1308-1316) in `GR_train_grid.pkl` (a dill pickle of a dict made by extract_dz_grid.py:
1 441 normalised (G, R) points with the span the training runs used there, and the normalisation bounds) by
nearest neighbour.  This script, run in THIS container only (needs /root/reference), stores

  * graingraphnn_amd/data/gr_span_grid.npz -- the table itself (data: G, R, span, G_min, G_max, R_min, R_max),
    which `synthetic.span_for(G, R)` reads;
  (ONE-OFF, sandbox only: `dill.load` of that pickle executes whatever it contains -- never run this on a machine or a
  pickle you do not trust; the shipped .npz is the artefact, nobody needs to re-run this.)

  * tests/golden/gr_span_pins.npz -- (G, R) pairs with the span the reference's own expression
    (`scipy.interpolate.griddata(..., method='nearest')` on the unpickled dict, evaluated here) returns for them.

    python tests/golden/make_gr_span_grid.py
"""
import os

import dill
import numpy as np
from scipy.interpolate import griddata

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def main():
    with open(os.path.join(REF, "GR_train_grid.pkl"), "rb") as f:
        g = dill.load(f)
    out = os.path.join(ROOT, "graingraphnn_amd", "data", "gr_span_grid.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, G=np.asarray(g["G"], np.float64), R=np.asarray(g["R"], np.float64),
                        span=np.asarray(g["span"], np.int64),
                        bounds=np.array([g["G_min"], g["G_max"], g["R_min"], g["R_max"]], np.float64))
    # pins: the shipped fixtures' parameters, the CLI defaults, the table's corners, and random draws
    rs = np.random.RandomState(0)
    GR = [(1.904, 0.558), (10.0, 2.0), (2.0, 0.4), (0.5, 0.2), (10.0, 0.2), (0.5, 2.0), (5.0, 1.0)]
    GR += [(float(a), float(b)) for a, b in zip(rs.uniform(0.5, 10.0, 40), rs.uniform(0.2, 2.0, 40))]
    pts = np.array([g["G"], g["R"]]).T
    spans = []
    for G, R in GR:   # graph_trajectory.py:1314-1316, as written there
        G_ = (G - g["G_min"]) / (g["G_max"] - g["G_min"])
        R_ = (R - g["R_min"]) / (g["R_max"] - g["R_min"])
        spans.append(int(griddata(pts, np.array(g["span"]), (G_, R_), method="nearest")))
    np.savez_compressed(os.path.join(HERE, "gr_span_pins.npz"), GR=np.array(GR, np.float64),
                        span=np.array(spans, np.int64))
    print(out, len(g["G"]), "points;", len(GR), "pins:", list(zip(GR[:7], spans[:7])))


if __name__ == "__main__":
    main()
