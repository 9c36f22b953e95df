#!/usr/bin/env python3
"""Development aid: the pieces of GrainRollout._apply_events at the 10k-grain graph, each bracketed by a synchronisation
(the body of the method re-stated with timers).  Not part of the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from graingraphnn_amd.rollout import GrainRollout, ET_JJ
from graingraphnn_amd.topology import update_topology

dev = torch.device("cuda", 0)
R, Cm, X, EI, EA, inputs = bench.build(dev, seed=0)
ro = GrainRollout(R, Cm, X, EI, EA, bench.SPAN, refresh_centres=True, domain_factor=inputs[3],
                  domain_offset=None if inputs[4] is None else torch.from_numpy(inputs[4]))
mask = {"grain": np.ones((X["grain"].size(0), 1), np.int64), "joint": np.ones((X["joint"].size(0), 1), np.int64)}
ro.enable_events(mask, area_threshold=-1.0, edge_threshold=0.999999)
JG = ("joint", "pull", "grain")
sync = torch.cuda.synchronize
ro.run_events(8)
for rnd in range(4):
    ro.area_threshold = -1.0
    ro.run_events(1)
    area = ro.pred["grain_area"].cpu().numpy()
    live = ro.mask["grain"][:, 0] > 0
    ro.area_threshold = float(np.nextafter(np.float32(np.sort(area[live])[2]), np.float32(1)))
    ro._run_segment("fwd")
    sync()
    T = {}

    def lap(name, t0):
        sync()
        T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()
    p = ro.pred
    t = time.perf_counter()
    area = p["grain_area"].cpu().numpy()
    prob = torch.sigmoid(p["edge_event"]).cpu().numpy()
    xj = ro.x["joint"].cpu().numpy()
    yj, yg = p["joint"].cpu().numpy(), p["grain"].cpu().numpy()
    t = lap("read-backs (5 x .cpu())", t)
    host = getattr(ro, "_ei_host", None)
    if host is not None and host[0] is ro.edge_index[ET_JJ]:
        ei_jj, ei_jg = host[2], host[3]
    else:
        ei_jj, ei_jg = ro.edge_index[ET_JJ].cpu().numpy(), ro.edge_index[JG].cpu().numpy()
    t = lap("edge lists (host mirror or read-back)", t)
    ge = np.flatnonzero(live & (area < np.float32(ro.area_threshold)))
    ge = ge[np.argsort(area[ge], kind="stable")]
    mg, mj = ro.mask["grain"].copy(), ro.mask["joint"].copy()
    t = lap("event list, mask copies", t)
    pp, pq, qp, switches, events = update_topology(xj, ei_jj, ei_jg, yj, yg, prob, ge, mg, mj, ro.edge_threshold)
    t = lap("update_topology", t)
    ro.mask["grain"], ro.mask["joint"] = mg, mj
    ro.x["joint"].copy_(torch.from_numpy(xj))
    p["joint"].copy_(torch.from_numpy(yj))
    ro._live_grain.copy_(torch.from_numpy(ro.mask["grain"][:, 0].astype(np.int32)))
    t = lap("uploads: x, y, live mask", t)
    new_ei = {ET_JJ: torch.from_numpy(pp).to(dev), JG: torch.from_numpy(pq).to(dev),
              ("grain", "push", "joint"): torch.from_numpy(np.ascontiguousarray(qp)).to(dev)}
    t = lap("uploads: three edge lists", t)
    from graingraphnn_amd import engine
    ro.edge_index = {et: new_ei[et] for et in engine.EDGE_TYPES}
    g = engine.graph_for(ro.be, ro.edge_index, ro.n_nodes)
    t = lap("graph_for (CSR x 3)", t)
    ro._set_topology(new_ei, lasting=False)
    t = lap("_set_topology (the rest: buffers)", t)
    jj, jg = new_ei[ET_JJ], new_ei[JG]
    ro._ei_host = (jj, jg, pp, pq, (jj._version, jg._version))
    ro._graph_fwd = ro._graph_ref = None
    ro._run_segment("ref")
    ro._einfo_fresh = False
    ro.steps_done += 1
    t = lap("refresh on the new topology", t)
    print(f"round {rnd}: {len(events)} grains; " + "; ".join(f"{k} {v:.3f}" for k, v in T.items()) + f"; sum {sum(T.values()):.3f} ms", flush=True)
