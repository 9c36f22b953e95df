#!/usr/bin/env python3
"""SURVEY 8 f-2 beside the headline (bench.py calls this in a child process): what the event-driven loop costs at the
10k-grain graph -- a quiet step of GrainRollout.run_events against a static step, and an EVENTFUL step piece by piece
(the host-side rewiring ggnn_topology_update, the CSR rebuild, the rest of the round trip).  The area threshold is put
just above the 3rd smallest predicted area (random weights tie, so a few dozen grains vanish per eventful step).
Prints one JSON line.  Not collected by pytest."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from graingraphnn_amd import engine, topology
import graingraphnn_amd.rollout as rmod
from graingraphnn_amd.rollout import GrainRollout

dev = torch.device("cuda", 0)
R, Cm, X, EI, EA, inputs = bench.build(dev, seed=0)
ro = GrainRollout(R, Cm, X, EI, EA, bench.SPAN, refresh_centres=True, domain_factor=inputs[3],
                  domain_offset=None if inputs[4] is None else torch.from_numpy(inputs[4]))
sync = torch.cuda.synchronize


def timed(fn):
    sync()
    t0 = time.perf_counter()
    r = fn()
    sync()
    return (time.perf_counter() - t0) * 1e3, r


ro.run(4 * ro.RUN_UNROLL)
t_static, _ = timed(lambda: ro.run(20 * ro.RUN_UNROLL))
static_us = t_static / (20 * ro.RUN_UNROLL) * 1e3
mask = {"grain": np.ones((X["grain"].size(0), 1), np.int64), "joint": np.ones((X["joint"].size(0), 1), np.int64)}
ro.enable_events(mask, area_threshold=-1.0, edge_threshold=0.999999)
ro.run_events(4 * ro.EVENTS_UNROLL)
t_quiet, _ = timed(lambda: ro.run_events(25 * ro.EVENTS_UNROLL))
quiet_us = t_quiet / (25 * ro.EVENTS_UNROLL) * 1e3
rounds = []
orig_update, orig_graph_for = topology.update_topology, rmod.graph_for
for rnd in range(5):
    ro.area_threshold = -1.0
    ro.run_events(1)
    area = ro.pred["grain_area"].cpu().numpy()
    live = ro.mask["grain"][:, 0] > 0
    ro.area_threshold = float(np.nextafter(np.float32(np.sort(area[live])[2]), np.float32(1)))
    t_fwd, _ = timed(lambda: ro._run_segment("fwd"))
    t = {}

    def upd(*a, **k):
        t0 = time.perf_counter()
        r = orig_update(*a, **k)
        t["rewiring_ms"] = (time.perf_counter() - t0) * 1e3
        return r

    def gf(*a, **k):
        sync()
        t0 = time.perf_counter()
        r = orig_graph_for(*a, **k)
        sync()
        t["csr_rebuild_ms"] = (time.perf_counter() - t0) * 1e3
        return r
    topology.update_topology, rmod.graph_for = upd, gf
    try:
        t_apply, (events, switches) = timed(ro._apply_events)
    finally:
        topology.update_topology, rmod.graph_for = orig_update, orig_graph_for
    t_ref, _ = timed(lambda: ro._run_segment("ref"))
    ro._einfo_fresh = False
    ro.steps_done += 1
    rounds.append({"grains": int(len(events)), "forwards_update_ms": round(t_fwd, 3), "apply_events_ms": round(t_apply, 3),
                   "refresh_ms": round(t_ref, 3), **{k: round(v, 3) for k, v in t.items()}})
steady = rounds[1:]   # (the first event of a process pays one-off allocations)
med = lambda key: round(float(np.median([r[key] for r in steady])), 3)
print(json.dumps({
    "what": "event-driven loop at the 10k-grain graph (SURVEY 8 f-2): GrainRollout.run_events / _apply_events",
    "static_step_us": round(static_us, 1), "quiet_event_step_us": round(quiet_us, 1),
    "quiet_over_static": round(quiet_us / static_us, 3),
    "eventful_step_ms": round(med("forwards_update_ms") + med("apply_events_ms") + med("refresh_ms"), 3),
    "of_which": {"forwards_update_eager_ms": med("forwards_update_ms"), "apply_events_ms": med("apply_events_ms"),
                 "host_rewiring_ggnn_topology_update_ms": med("rewiring_ms"), "csr_rebuild_ms": med("csr_rebuild_ms"),
                 "refresh_ms": med("refresh_ms")},
    "grains_per_eventful_step": [r["grains"] for r in steady],
    "statistic": "median of 4 eventful steps after the first (which pays one-off allocations); each piece bracketed by a synchronisation",
}))
