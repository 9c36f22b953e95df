"""Event-driven topology update of the grain graph, host side: the binding of
`ggnn_topology_update` (include/ggnn.h, csrc/topology.hip -- host code in the C-ABI library).

It restates `GrainNN_classifier.update` of the reference (models.py:612-842 with its helpers
`delete_grain_index` :861-893, `switching_edge_index` :896-1051, `point_in_triangle` :1055-1070,
`periodic_move` :1103-1106) for the periodic, nucleation-free configuration every shipped script
runs (test.py:88 `--nucleation_density 0`).  The work is integer / index work on a few edges per
event and inherently sequential (every event rewires the lists the next one reads), so it runs on
the host between device steps; the device side only sees new `edge_index` tensors, for which
`engine.graph_for` rebuilds the CSR.

The reference answers every lookup with a mask over a whole edge list (57-75 ms of host time per
eventful step at the 10k-grain graph, measured with that formulation: oracle/topology_scan.py, now
the test oracle); the library answers them from column indices and running counts in native code
(profiles/r5_event_step_breakdown.txt).

Bit-exactness contract (tests/golden/golden_cfg1_events.npz, produced by the unmodified
reference): identical `edge_index` COLUMN ORDER (edges are rewritten in place, new edges are
appended, dead columns are dropped at the end -- never re-sorted), identical masks, identical
fp32 junction coordinates.
"""
import ctypes
from typing import Optional, Sequence

import numpy as np

from . import _lib

GJ, JG, JJ = ("grain", "push", "joint"), ("joint", "pull", "grain"), ("joint", "connect", "joint")


class TopologyError(RuntimeError):
    """The lists are not a valid grain graph (the reference asserts / raises KeyError here)."""


def _flags(ids: Optional[Sequence[int]], n: int) -> Optional[np.ndarray]:
    """Index list of an active window -> one byte per node (None = everything is active)."""
    if ids is None:
        return None
    f = np.zeros(n, dtype=np.uint8)
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    f[ids[(ids >= 0) & (ids < n)]] = 1
    return f


def update_topology(x_joint: np.ndarray, ei_jj: np.ndarray, ei_jg: np.ndarray, y_joint: np.ndarray,
                    y_grain: np.ndarray, edge_prob: np.ndarray, grain_event: Sequence[int],
                    mask_grain: np.ndarray, mask_joint: np.ndarray, threshold: float,
                    active_grains: Optional[np.ndarray] = None, active_joints: Optional[np.ndarray] = None):
    """One call of the reference's `Cmodel.update` (nucleation off).  `x_joint` [N_j, 8] fp32,
    `y_joint` [N_j, 2] fp32 and the masks are modified in place.  `edge_prob` = sigmoid of the
    classifier's `edge_event` logits, `grain_event` = grains below the area threshold, smallest
    first (test.py:418-420).  Returns (ei_jj, ei_jg, ei_gj, switching_list [S, 2], grain_event').
    Raises TopologyError when the lists are not a valid grain graph; the in-place arguments are then
    untouched (the library works on copies that are committed together)."""
    lib = _lib.load()
    n_j, n_g = int(mask_joint.shape[0]), int(mask_grain.shape[0])
    for name, a, cols in (("x_joint", x_joint, None), ("y_joint", y_joint, 2)):
        if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.ndim != 2 or a.shape[0] != n_j \
                or (cols is not None and a.shape[1] != cols):
            raise ValueError(f"{name} must be a float32 numpy array [{n_j}, {cols or '>= 8'}]")
    if x_joint.shape[1] < 8:
        raise ValueError("x_joint needs the 8 junction features (columns 6, 7 = the displacement the events reset)")
    ge = np.ascontiguousarray(np.asarray(grain_event, dtype=np.int64).reshape(-1))
    n_pp, n_pq = int(ei_jj.shape[1]), int(ei_jg.shape[1])
    # every removed grain appends two columns before the dead ones are dropped; forced eliminations and two-sided grains
    # come on top of `grain_event`: bounded by the number of grains
    cap = n_pp + 2 * n_g + 2
    pp = np.empty((2, cap), dtype=np.int64)
    pp[:, :n_pp] = ei_jj
    pq = np.array(ei_jg, dtype=np.int64, order="C", copy=True)
    xj, yj = np.array(x_joint, order="C", copy=True), np.array(y_joint, order="C", copy=True)
    mg = np.ascontiguousarray(mask_grain.reshape(-1).astype(np.int64))
    mj = np.ascontiguousarray(mask_joint.reshape(-1).astype(np.int64))
    area = np.ascontiguousarray(np.asarray(y_grain, dtype=np.float32)[:, 0])
    prob = np.ascontiguousarray(np.asarray(edge_prob, dtype=np.float32).reshape(-1))
    if prob.shape[0] != n_pp or area.shape[0] != n_g:
        raise ValueError("edge_prob needs one entry per junction edge, y_grain one row per grain")
    act_g, act_j = _flags(active_grains, n_g), _flags(active_joints, n_j)
    switching = np.empty((max(n_pp, 1), 2), dtype=np.int64)
    extra = np.empty(n_g + 1, dtype=np.int64)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    A = _lib.TopologyArgs()
    A.pp, A.pq, A.n_pp, A.n_pq, A.pp_cap, A.pq_cap = p(pp), p(pq), n_pp, n_pq, cap, n_pq
    A.x_joint, A.y_joint, A.y_grain_area, A.edge_prob = p(xj), p(yj), p(area), p(prob)
    A.grain_event, A.mask_grain, A.mask_joint = p(ge), p(mg), p(mj)
    A.active_grain, A.active_joint, A.switching, A.events_extra = p(act_g), p(act_j), p(switching), p(extra)
    A.n_joint, A.n_grain, A.ldx, A.ldyg, A.n_grain_event = n_j, n_g, xj.shape[1], 1, ge.shape[0]
    A.switching_cap, A.extra_cap, A.threshold = switching.shape[0], extra.shape[0], float(threshold)
    rc = lib.ggnn_topology_update(ctypes.byref(A))
    if rc == _lib.GGNN_ETOPOLOGY:
        raise TopologyError(A.error.decode(errors="replace"))
    _lib.check(rc, "ggnn_topology_update")
    # commit: coordinates, displacement features and masks in place, the lists as new arrays
    x_joint[...] = xj
    y_joint[...] = yj
    mask_grain[...] = mg.reshape(mask_grain.shape)
    mask_joint[...] = mj.reshape(mask_joint.shape)
    new_pp = pp[:, :A.n_pp].copy()
    new_pq = pq[:, :A.n_pq].copy()
    events = np.concatenate([ge, extra[:A.n_extra]]) if A.n_extra else ge
    return new_pp, new_pq, new_pq[::-1].copy(), switching[:A.n_switching].copy(), events
