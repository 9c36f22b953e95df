#!/usr/bin/env python3
"""Development aid: where the time of an EVENTFUL step goes at the 10k-grain graph -- read-backs, the host-side
rewiring (topology.update_topology -> ggnn_topology_update), uploads, the CSR rebuild, buffer allocation, the
refresh on the new topology and the first steps afterwards (graphs are re-captured).  The area threshold is put just above
the k-th smallest predicted area so that a handful of grains vanish.  Not part of the product."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from graingraphnn_amd import engine, synthetic
from graingraphnn_amd.rollout import GrainRollout
from graingraphnn_amd.synthetic import EDGE_TYPES

ap = argparse.ArgumentParser()
ap.add_argument("--grains", type=int, default=3, help="grains to eliminate in the eventful step")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
R, Cm, X, EI, EA, inputs = bench.build(dev, seed=0)
ro = GrainRollout(R, Cm, X, EI, EA, bench.SPAN, refresh_centres=True, domain_factor=inputs[3],
                  domain_offset=None if inputs[4] is None else torch.from_numpy(inputs[4]))
mask = {"grain": np.ones((X["grain"].size(0), 1), np.int64), "joint": np.ones((X["joint"].size(0), 1), np.int64)}
ro.enable_events(mask, area_threshold=-1.0, edge_threshold=0.999999)
sync = torch.cuda.synchronize


def timed(fn):
    sync()
    t0 = time.perf_counter()
    r = fn()
    sync()
    return (time.perf_counter() - t0) * 1e3, r


ro.run_events(16)   # quiet: captures the block graphs
t_quiet, _ = timed(lambda: ro.run_events(64))
print(f"quiet steps (speculative blocks of {ro.EVENTS_UNROLL}): {t_quiet / 64 * 1e3:.1f} us per step")
for rnd in range(a.rounds):
    # one step to see the predicted areas, then the threshold that takes the k smallest live grains
    ro.area_threshold = -1.0
    ro.run_events(1)
    area = ro.pred["grain_area"].cpu().numpy()
    live = ro.mask["grain"][:, 0] > 0
    kth = np.sort(area[live])[a.grains - 1]
    ro.area_threshold = float(np.nextafter(np.float32(kth), np.float32(1)))
    # --- the eventful step, piece by piece (the body of step_events / _apply_events with timers) ---
    t_fwd, _ = timed(lambda: ro._run_segment("fwd"))
    ro.be.detect_events(ro.pred["grain_area"], ro._live_grain, ro.area_threshold, ro.pred["edge_event"],
                        ro.graph.edge_index[("joint", "connect", "joint")], ro._logit_trigger, ro._ev_flags)
    sync()
    assert int(ro._ev_flags[0].item()) > 0, "no grain below the threshold"
    from graingraphnn_amd import topology
    orig_update, orig_graph_for = topology.update_topology, engine.graph_for
    t = {}

    def upd(*args, **kw):
        t0 = time.perf_counter()
        r = orig_update(*args, **kw)
        t["host rewiring (update_topology)        "] = (time.perf_counter() - t0) * 1e3
        return r

    def gf(*args, **kw):
        sync()
        t0 = time.perf_counter()
        r = orig_graph_for(*args, **kw)
        sync()
        t["CSR rebuild (3 edge types, device)"] = (time.perf_counter() - t0) * 1e3
        return r
    topology.update_topology = upd
    import graingraphnn_amd.rollout as rmod
    rmod.graph_for = gf
    t_apply, (events, switches) = timed(ro._apply_events)
    topology.update_topology, rmod.graph_for = orig_update, orig_graph_for
    t_ref, _ = timed(lambda: ro._run_segment("ref"))
    ro.steps_done += 1
    ro.area_threshold = -1.0
    t_next = [timed(lambda: ro.run_events(1))[0] for _ in range(3)]
    t_block, _ = timed(lambda: ro.run_events(2 * ro.EVENTS_UNROLL))
    rest = t_apply - sum(t.values())
    print(f"round {rnd}: {len(events)} grains eliminated, {len(switches)} switches, "
          f"{ro.edge_index[('joint', 'connect', 'joint')].size(1)} junction edges left")
    print(f"  forwards + update (eager)                {t_fwd:8.3f} ms")
    print(f"  _apply_events                            {t_apply:8.3f} ms")
    for k, v in t.items():
        print(f"    {k:38s} {v:8.3f} ms")
    print(f"    read-backs, uploads, buffer allocation {rest:8.3f} ms")
    print(f"  refresh on the new topology              {t_ref:8.3f} ms")
    print(f"  next three single steps                  " + " ".join(f"{v:.3f}" for v in t_next) + " ms")
    print(f"  next {2 * ro.EVENTS_UNROLL} steps (graphs re-captured)        {t_block:8.3f} ms")

# --- a stretch in which EVERY step is eventful: the threshold follows the k-th smallest area the previous step predicted ---
K_GRAINS, N_STEPS = 3, 100
import graingraphnn_amd.topology as topo_mod
done, n_ev, refused = 0, 0, None
sync()
t0 = time.perf_counter()
try:
    for _ in range(N_STEPS):
        area = ro.pred["grain_area"]
        live = ro._live_grain > 0
        kth = torch.kthvalue(area[live], K_GRAINS).values
        ro.area_threshold = float(np.nextafter(np.float32(kth.item()), np.float32(1)))
        ev, _ = ro.run_events(1)
        n_ev += len(ev[0])
        done += 1
except topo_mod.TopologyError as err:
    refused = str(err)
sync()
dt = (time.perf_counter() - t0) * 1e3
print(f"{done} consecutive eventful steps, {n_ev} grains eliminated ({n_ev / max(done, 1):.1f} per step): {dt / max(done, 1):.3f} ms per step "
      f"= {max(done, 1) / dt * 1e3:.0f} steps/s (incl. one threshold read-back per step)" + (f"; stopped: {refused}" if refused else ""))
