#!/usr/bin/env python3
"""Development probe: fuzz_events case 63 (120 um fixture, seed 69121 x0.5, area < 1e-4, p > 0.7): scan oracle + step_events,
native session + step_events, native session + run_events -- where do they part?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import load_graph, product_models, tt  # noqa: E402
from graingraphnn_amd import GrainRollout, topology as native  # noqa: E402
from graingraphnn_amd.synthetic import EDGE_TYPES  # noqa: E402
from oracle import topology_scan as scan  # noqa: E402

DEV = torch.device("cuda", 0)
x, ei, ea = load_graph("120")
R, Cm = product_models(69121, 0.5, DEV)
mask = {"grain": np.ones((x["grain"].shape[0], 1)), "joint": np.ones((x["joint"].shape[0], 1))}
kw = dict(use_graph=True, refresh_centres=True, joint_launches=False, concurrent=True)


def make(hook):
    X = tt(x, DEV)
    ro = GrainRollout(R, Cm, X, tt(ei, DEV), tt(ea, DEV), 6, **kw)
    ro.enable_events(mask, 1e-4, 0.7)
    if hook:
        def scan_update(*a, **k):
            try:
                return scan.update_topology(*a, **k)
            except (scan.TopologyError, IndexError, ValueError) as err:
                raise native.TopologyError(str(err)) from None
        ro.rewire_hook = scan_update
    return ro, X


(ra, Xa), (rb, Xb), (rc, Xc) = make(True), make(False), make(False)
with torch.no_grad():
    for step in range(5):
        out = []
        for name, ro in (("scan/step", ra), ("native/step", rb)):
            try:
                p, e, sw = ro.step_events()
                out.append((name, len(e), len(sw), None))
            except native.TopologyError as err:
                out.append((name, None, None, str(err)[:60]))
        try:
            e, sw = rc.run_events(1)
            out.append(("native/run", len(e[0]), len(sw[0]), None))
        except native.TopologyError as err:
            out.append(("native/run", None, None, str(err)[:60]))
        torch.cuda.synchronize()
        eq = lambda r1, X1, r2, X2: (all(torch.equal(X1[nt], X2[nt]) for nt in X1),
                                     all(torch.equal(r1.edge_index[et], r2.edge_index[et]) for et in EDGE_TYPES)
                                     if all(r1.edge_index[et].shape == r2.edge_index[et].shape for et in EDGE_TYPES) else False,
                                     bool(np.array_equal(r1.mask["grain"], r2.mask["grain"])))
        nan = lambda X: {nt: int(torch.isnan(X[nt]).sum()) for nt in X}
        print(f"step {step}: {out}\n   a=b (x, lists, mask): {eq(ra, Xa, rb, Xb)}  b=c: {eq(rb, Xb, rc, Xc)}  live grains a/b/c: "
              f"{int(ra.mask['grain'].sum())}/{int(rb.mask['grain'].sum())}/{int(rc.mask['grain'].sum())}  NaNs in x a {nan(Xa)} c {nan(Xc)}; "
              f"area NaNs a {int(torch.isnan(ra.pred['grain_area']).sum())} c {int(torch.isnan(rc.pred['grain_area']).sum())}", flush=True)

print("--- run_events in chunks (the fuzz's rollout b)")
with torch.no_grad():
    for chunks in ((5,), (4, 1), (3, 2), (2, 3), (1, 4), (1, 1, 3), (2, 2, 1)):
        rd, Xd = make(False)
        ev, err = [], None
        try:
            for n in chunks:
                e, sw = rd.run_events(n)
                ev += [len(v) for v in e]
        except native.TopologyError as exc:
            err = str(exc)[:70]
            ev = [len(v) for v in rd.grain_events]
        print(f"chunks {chunks}: events per step {ev}, error: {err}; live grains {int(rd.mask['grain'].sum())}", flush=True)
