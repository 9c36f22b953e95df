#!/usr/bin/env python3
"""GPU fuzz (not collected by pytest): event-driven rollouts on the 40 um and 120 um fixtures under random models and thresholds --
GrainRollout.step_events with the SCAN ORACLE's rewiring (oracle/topology_scan.py, the reference's formulation) against
GrainRollout.run_events with the product's native rewiring (the library's topology session: ggnn_topology_apply): same events, switches, edge lists, masks and
state bit for bit, or the same refusal at the same step.  (Refusals one step apart are counted separately -- everything up to the earlier one
must still be identical: until the end of round 6 the native update refused a structure with a doubly joined junction pair that the scan
formulation, the reference's, rewrites once more before it fails itself -- case 63 of 120, a collapse from 401 to 32 grains in one step; it
follows the reference there now and no such pair is left in 500 cases.)   python tests/fuzz_events.py [n_cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import load_graph, product_models, tt
from graingraphnn_amd import GrainRollout, synthetic
from graingraphnn_amd import rollout as rollout_mod, topology as native
from graingraphnn_amd.synthetic import EDGE_TYPES
from oracle import topology_scan as scan

DEV = torch.device("cuda", 0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rs = np.random.RandomState(7)
tot_ev = tot_sw = refusals = apart = 0
with torch.no_grad():
    for case in range(n_cases):
        name = "40" if case % 3 else "120"
        x, ei, ea = load_graph(name)
        seed, scale = int(rs.randint(1, 10 ** 5)), float(rs.choice([0.5, 1.0, 1.5]))
        area_thr = float(rs.choice([5e-5, 1e-4, 3e-4, 1e-3]))
        edge_thr = float(rs.choice([0.5, 0.55, 0.6, 0.7]))
        steps = int(rs.randint(4, 11))
        R, Cm = product_models(seed, scale, DEV)
        n_g, n_j = x["grain"].shape[0], x["joint"].shape[0]
        mask = {"grain": np.ones((n_g, 1)), "joint": np.ones((n_j, 1))}
        kw = dict(use_graph=bool(case & 1), refresh_centres=True, joint_launches=False, concurrent=True)
        Xa, Xb = tt(x, DEV), tt(x, DEV)
        ra = GrainRollout(R, Cm, Xa, tt(ei, DEV), tt(ea, DEV), 6, **kw)
        rb = GrainRollout(R, Cm, Xb, tt(ei, DEV), tt(ea, DEV), 6, **kw)
        ra.enable_events(mask, area_thr, edge_thr)
        rb.enable_events(mask, area_thr, edge_thr)
        ev_a, sw_a, err_a = [], [], None
        # rollout a: one step at a time, rewiring by the scan oracle

        def scan_update(*args, **kw2):
            try:
                return scan.update_topology(*args, **kw2)
            except (scan.TopologyError, IndexError, ValueError) as err:
                raise native.TopologyError(str(err)) from None
        try:
            ra.rewire_hook = scan_update   # (GrainRollout._apply_events: the update by another implementation)
            for _ in range(steps):
                _, e, sw = ra.step_events()
                ev_a.append(e), sw_a.append(sw)
        except native.TopologyError as err:
            err_a = str(err)
        ev_b, sw_b, err_b = [], [], None
        try:
            left = steps
            while left:
                n = min(left, int(rs.randint(1, 6)))
                e, sw = rb.run_events(n)
                ev_b += e
                sw_b += sw
                left -= n
        except native.TopologyError as err:
            err_b = str(err)
            ev_b, sw_b = rb.grain_events[:], rb.switched[:]
        torch.cuda.synchronize()
        assert (err_a is None) == (err_b is None), (case, err_a, err_b)
        # (run_events keeps a refused step as a quiet one in its lists: not a step to compare)
        n_ok = min(len(ev_a), len(ev_b) - (1 if err_b else 0)) if err_a else steps
        if err_a is not None and len(ev_a) != len(ev_b) - 1:
            apart += 1
        assert err_a is not None or len(ev_a) == len(ev_b) == steps
        for k in range(n_ok):
            if not (np.array_equal(ev_a[k], ev_b[k]) and np.array_equal(sw_a[k], sw_b[k])):
                print(f"case {case} ({name}, seed {seed} x{scale}, area<{area_thr:g} p>{edge_thr:g}, graph={kw['use_graph']}) step {k}: "
                      f"events a {ev_a[k].tolist()} b {ev_b[k].tolist()}; switches a {len(sw_a[k])} b {len(sw_b[k])}; "
                      f"only in a {sorted(set(map(tuple, sw_a[k].tolist())) - set(map(tuple, sw_b[k].tolist())))[:6]}, "
                      f"only in b {sorted(set(map(tuple, sw_b[k].tolist())) - set(map(tuple, sw_a[k].tolist())))[:6]}; "
                      f"per step a {[len(e) for e in ev_a]} b {[len(e) for e in ev_b]}", flush=True)
            assert np.array_equal(ev_a[k], ev_b[k]) and np.array_equal(sw_a[k], sw_b[k]), (case, k)
        if err_a is None:
            for et in EDGE_TYPES:
                assert torch.equal(ra.edge_index[et], rb.edge_index[et]), (case, et)
                assert torch.equal(ra.edge_attr_dict()[et], rb.edge_attr_dict()[et]), (case, et)
            assert np.array_equal(ra.mask["grain"], rb.mask["grain"]) and np.array_equal(ra.mask["joint"], rb.mask["joint"])
            for nt in Xa:
                assert torch.equal(Xa[nt], Xb[nt]), (case, nt)
        else:
            refusals += 1
        ne, ns = sum(len(e) for e in ev_a), sum(len(s) for s in sw_a)
        tot_ev += ne
        tot_sw += ns
        print(f"case {case:2d}: fixture {name:>3s} seed {seed:5d} x{scale} area<{area_thr:g} p>{edge_thr:g} {steps:2d} steps, graph={kw['use_graph']}: "
              f"{ne:4d} grains, {ns:3d} switches per step {[len(e) for e in ev_a]}" + (f"  both refused: {err_a[:50]}" if err_a else ""), flush=True)
print(f"{n_cases} cases, {tot_ev} eliminated grains, {tot_sw} switched edges, {refusals} refused by both"
      + (f" ({apart} of them one step apart: see the docstring)" if apart else "") + ": identical")
