#!/usr/bin/env python3
"""Development probe: in-place event rollout (graphs) beside an eager rebuild-everything rollout on the reference's event trajectory."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import EDGE_TYPES, load_graph, product_models, tt  # noqa: E402
from graingraphnn_amd import GrainRollout  # noqa: E402

DEV = torch.device("cuda", 0)
x, ei, ea = load_graph("40")
R, Cm = product_models(10020, 1.0, DEV)
mask = {"grain": np.ones((118, 1)), "joint": np.ones((236, 1))}
mode = sys.argv[1] if len(sys.argv) > 1 else "both"


def make(in_place, graph):
    os.environ["GGNN_EVENT_GRAPHS"] = "1" if in_place else "0"
    X = tt(x, DEV)
    ro = GrainRollout(R, Cm, X, tt(ei, DEV), tt(ea, DEV), 6, use_graph=graph, refresh_centres=True)
    ro.enable_events(mask, 1e-4, 0.6)
    return ro, X


ra, Xa = make(True, True)
rb, Xb = (make(False, False) if mode == "both" else (None, None))
for step in range(1, 6):
    try:
        pa, eva, sa = ra.step_events()
    except Exception as exc:
        print("step", step, "in-place rollout raised:", exc)
        break
    msg = f"step {step}: in place: {len(eva)} grains, {len(sa)} switches, E = {[ra.edge_index[et].size(1) for et in EDGE_TYPES]}"
    if rb is not None:
        pb, evb, sb = rb.step_events()
        same = all(torch.equal(Xa[nt], Xb[nt]) for nt in Xa)
        msg += f" | eager: {len(evb)} grains, {len(sb)} switches; state equal: {same}; pred equal: " + \
            str({k: bool(torch.equal(pa[k], pb[k])) for k in ("joint", "grain", "grain_area", "edge_event", "edge")})
    print(msg, flush=True)
