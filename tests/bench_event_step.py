#!/usr/bin/env python3
"""SURVEY 8 f-2 beside the headline (bench.py calls this in a child process): what the event-driven loop costs.
  * at the 10k-grain graph (cfg3): a quiet step of GrainRollout.run_events against a static step, and an EVENTFUL step piece
    by piece (eager forwards, the read-back, the library's rewiring ggnn_topology_apply, uploads + CSR rebuild, refresh);
  * STEADY-EVENTFUL records at cfg1 (40 um fixture), cfg2 (120 um fixture, folded) and cfg3: every step eliminates grains,
    as on the reference's own trajectories (README.md:68-69: 75 eliminations in 20 steps at 40 um, 704 at 120 um, i.e. ~4 and
    ~35 per step) -- GrainRollout.step_events in a loop, the area threshold put just above the k-th smallest area the
    previous step predicted (random weights: ties make it k or a few more), edge switching off (threshold 0.999999).
Prints one JSON line.  Not collected by pytest."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from graingraphnn_amd import synthetic
from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor
from graingraphnn_amd.rollout import GrainRollout
from graingraphnn_amd.seeding import load_seeded
from graingraphnn_amd.topology import TopologyError

dev = torch.device("cuda", 0)
sync = torch.cuda.synchronize
GOLD = os.path.join(ROOT, "tests", "golden")


def timed(fn):
    sync()
    t0 = time.perf_counter()
    r = fn()
    sync()
    return (time.perf_counter() - t0) * 1e3, r


def rollout_for(workload):
    if workload == "cfg3":
        R, Cm, X, EI, EA, inputs = bench.build(dev, seed=0)
        factor, off = inputs[3], inputs[4]
    else:
        hp = synthetic.default_hyper(dev)
        R = GrainNN_regressor(hp)
        Cm = GrainNN_classifier(hp, R)
        x, ei, ea = synthetic.load_fixture(os.path.join(GOLD, "graph_40.npz" if workload == "cfg1" else "graph_120.npz"))
        x, ea = {k: v.copy() for k, v in x.items()}, {k: v.copy() for k, v in ea.items()}
        if workload == "cfg2":
            factor, off = 3.0, synthetic.scale_feature_patchs(3.0, x, ea)
            load_seeded(R, 0).eval(), load_seeded(Cm, 1).eval()
        else:
            factor, off = 1.0, None
            load_seeded(R, 10020).eval(), load_seeded(Cm, 10021).eval()
        R, Cm = R.to(dev), Cm.to(dev)
        X, EI, EA = synthetic.to_torch(x, ei, ea, dev)
    ro = GrainRollout(R, Cm, X, EI, EA, bench.SPAN, use_graph=True, refresh_centres=True, domain_factor=factor,
                      domain_offset=None if off is None else torch.from_numpy(off), joint_launches=False, concurrent=True)
    mask = {"grain": np.ones((X["grain"].size(0), 1), np.int64), "joint": np.ones((X["joint"].size(0), 1), np.int64)}
    return ro, X, mask


def kth_area_threshold(ro, k):
    """A threshold just above the k-th smallest predicted area of the live grains.  Random weights tie the predicted areas of
    hundreds of grains: `GrainRollout.max_grain_events` (a probe hook: at most that many grains per step, smallest
    first) keeps a step at k eliminations + whatever they force."""
    area = ro.pred["grain_area"].cpu().numpy()
    live = ro.mask["grain"][:, 0] > 0
    ro.max_grain_events = k
    return float(np.nextafter(np.float32(np.sort(area[live])[k - 1]), np.float32(1))), int(live.sum())


def steady_eventful(workload, k, n_steps):
    """`n_steps` consecutive eventful steps (about k grains each): wall time per step of step_events, everything included
    but the probe's own threshold read-back between steps."""
    ro, X, mask = rollout_for(workload)
    ro.enable_events(mask, area_threshold=-1.0, edge_threshold=0.999999)
    for _ in range(3):
        ro.step_events()
    ms, grains, stopped = [], [], None
    for i in range(n_steps + 2):
        thr, n_live = kth_area_threshold(ro, k)
        if n_live < 6 * k + 12:
            stopped = f"{n_live} grains left"
            break
        ro.area_threshold = thr
        try:
            t, (_, ev, _) = timed(ro.step_events)
        except (TopologyError, IndexError, ValueError) as exc:
            stopped = f"{type(exc).__name__}: {exc}"[:120]
            break
        if i >= 2:   # (the first events of a process pay one-off allocations and the session's start)
            ms.append(t), grains.append(int(len(ev)))
    ro.area_threshold = -1.0
    for _ in range(4):
        ro.step_events()   # (two quiet steps, then the segment graphs are captured)
    t_quiet, _ = timed(lambda: [ro.step_events() for _ in range(10)])
    return {"workload": workload, "grains_start": int(mask["grain"].shape[0]), "eventful_steps_timed": len(ms),
            "grains_per_step": grains, "ms_per_step_median": round(float(np.median(ms)), 3) if ms else None,
            "ms_per_step_min_max": [round(min(ms), 3), round(max(ms), 3)] if ms else None,
            "steps_per_s": round(1e3 / float(np.median(ms)), 1) if ms else None,
            "quiet_step_events_ms_per_step_afterwards": round(t_quiet / 10, 3), "stopped": stopped}


def sparse_events(period, cycles):
    """run_events() with ~3 grains eliminated every `period` steps (cfg3): wall time per step -- what an event costs a loop
    that is otherwise quiet (on the in-place topology the ring of slots and the block graphs survive it)."""
    ro, X, mask = rollout_for("cfg3")
    ro.enable_events(mask, area_threshold=-1.0, edge_threshold=0.999999)

    def cycle():
        ro.area_threshold = -1.0
        ro.run_events(period - 1)
        ro.area_threshold, _ = kth_area_threshold(ro, 3)
        return len(ro.run_events(1)[0][0])
    for _ in range(3):
        cycle()
    t, grains = timed(lambda: [cycle() for _ in range(cycles)])
    return {"workload": "cfg3", "an_event_every_steps": period, "steps_timed": period * cycles, "grains_per_event": sorted(set(grains)),
            "ms_per_step": round(t / (period * cycles), 3), "steps_per_s": round(period * cycles / t * 1e3, 1)}


ro, X, mask = rollout_for("cfg3")
ro.run(4 * ro.RUN_UNROLL)
t_static, _ = timed(lambda: ro.run(20 * ro.RUN_UNROLL))
static_us = t_static / (20 * ro.RUN_UNROLL) * 1e3
ro.enable_events(mask, area_threshold=-1.0, edge_threshold=0.999999)
ro.run_events(4 * ro.EVENTS_UNROLL)
t_quiet, _ = timed(lambda: ro.run_events(25 * ro.EVENTS_UNROLL))
quiet_us = t_quiet / (25 * ro.EVENTS_UNROLL) * 1e3
rounds = []
for rnd in range(6):
    ro.area_threshold = -1.0
    ro.step_events()   # (a quiet step between the rounds; run_events(1) here would leave ANOTHER slot's buffers current every
    #                     round, and the segment graphs -- one per set of buffers -- would be captured inside the timed pieces)
    ro.area_threshold, _ = kth_area_threshold(ro, 3)
    ro._einfo_fresh = False
    t_fwd, _ = timed(lambda: ro._run_segment("fwd"))
    T = ro.event_timing = {}
    t_apply, (events, switches) = timed(ro._apply_events)
    t_ref, _ = timed(lambda: ro._run_segment("ref"))
    ro.event_timing = None
    ro._einfo_fresh = False
    ro._x_written_outside()
    ro.steps_done += 1
    rounds.append({"grains": int(len(events)), "forwards_update_ms": round(t_fwd, 3), "apply_events_ms": round(t_apply, 3),
                   "refresh_ms": round(t_ref, 3), **{k[:-2] + "_ms": round(v * 1e3, 3) for k, v in T.items()}})
steady = rounds[1:]   # (the first event of a process pays one-off allocations and opens the session)
med = lambda key: round(float(np.median([r[key] for r in steady])), 3)
out = {
    "what": "event-driven loop (SURVEY 8 f-2): GrainRollout.run_events / step_events / _apply_events",
    "static_step_us": round(static_us, 1), "quiet_event_step_us": round(quiet_us, 1),
    "quiet_over_static": round(quiet_us / static_us, 3),
    "eventful_step_ms": round(med("forwards_update_ms") + med("apply_events_ms") + med("refresh_ms"), 3),
    "topology_in_place": getattr(ro, "_cap", None) is not None,   # (the segments' hipGraphs survive events: DESIGN section 6)
    "of_which": {"forwards_update_ms": med("forwards_update_ms"), "apply_events_ms": med("apply_events_ms"),
                 "apply_events_host_pieces_ms": {k: med(k) for k in ("readback_ms", "rewiring_ms", "upload_enqueue_ms",
                                                                      "set_topology_ms")},
                 "refresh_ms": med("refresh_ms")},
    "grains_per_eventful_step": [r["grains"] for r in steady],
    "statistic": ("cfg3, median of 5 eventful steps after the first; forwards / apply / refresh each bracketed by a "
                  "synchronisation (apply_events' own pieces are host-side intervals: the uploads and the CSR rebuild they "
                  "enqueue finish inside the apply bracket); with the topology in place the forwards and the refresh are "
                  "graph replays, otherwise (GGNN_EVENT_GRAPHS=0) eager launches"),
    "steady_eventful": [steady_eventful("cfg1", 4, 8), steady_eventful("cfg2", 35, 8), steady_eventful("cfg3", 20, 12)],
    "sparse_events_run_events": [sparse_events(16, 8), sparse_events(64, 3)],
}
print(json.dumps(out))
