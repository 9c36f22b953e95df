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
import os
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


def _library():
    """libggnn.so -- or, for the sanitizer run of the host code only (tests/test_topology.py under
    `make -C graingraphnn_amd/csrc host-asan`), the plain-C++ build of csrc/topology.hip named by GGNN_TOPOLOGY_LIB."""
    alt = os.environ.get("GGNN_TOPOLOGY_LIB")
    if not alt:
        return _lib.load()
    lib = _alt_libs.get(alt)
    if lib is None:
        lib = _alt_libs[alt] = ctypes.CDLL(alt)
        c_int, c_int64, c_void_p, P = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER
        lib.ggnn_topology_update.restype = c_int
        lib.ggnn_topology_update.argtypes = [P(_lib.TopologyArgs)]
        lib.ggnn_topology_open.restype = c_int
        lib.ggnn_topology_open.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                           P(c_void_p), ctypes.c_char_p]
        lib.ggnn_topology_apply.restype = c_int
        lib.ggnn_topology_apply.argtypes = [c_void_p, P(_lib.TopologyArgs)]
        lib.ggnn_topology_counts.restype = c_int
        lib.ggnn_topology_counts.argtypes = [c_void_p, P(c_int64), P(c_int64)]
        lib.ggnn_topology_export.restype = c_int
        lib.ggnn_topology_export.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64]
        lib.ggnn_topology_close.restype = None
        lib.ggnn_topology_close.argtypes = [c_void_p]
    return lib


_alt_libs = {}
_p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class TopologySession:
    """The lists of ONE trajectory kept inside the library between updates (ggnn_topology_open / _apply / _export,
    include/ggnn.h): the reference's loop calls `Cmodel.update` at every step (test.py:418-426); the session keeps the
    lookup tables and per-grain counts and patches them as the events rewrite columns, so an update costs what its events
    touch.  Same results, bit for bit, as chained `update_topology` calls (tests/test_topology.py).

    `apply` works IN PLACE on the caller's x_joint / y_joint / masks (a refused update leaves them and the session
    untouched -- the library journals its writes); `export` writes the current lists into caller-provided int64 arrays
    (e.g. pinned staging buffers of an upload)."""

    def __init__(self, ei_jj: np.ndarray, ei_jg: np.ndarray, n_joint: int, n_grain: int):
        self._lib = _library()
        pp = np.ascontiguousarray(ei_jj, dtype=np.int64)
        pq = np.ascontiguousarray(ei_jg, dtype=np.int64)
        if pp.ndim != 2 or pp.shape[0] != 2 or pq.ndim != 2 or pq.shape[0] != 2:
            raise ValueError("edge lists must be [2, E]")
        self.n_joint, self.n_grain = int(n_joint), int(n_grain)
        self._h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(192)
        rc = self._lib.ggnn_topology_open(_p(pp), pp.shape[1], pp.shape[1], _p(pq), pq.shape[1], pq.shape[1],
                                          self.n_joint, self.n_grain, ctypes.byref(self._h), err)
        if rc == _lib.GGNN_ETOPOLOGY:
            raise TopologyError(err.value.decode(errors="replace"))
        _lib.check(rc, "ggnn_topology_open")
        self.n_pp, self.n_pq = pp.shape[1], pq.shape[1]
        # output lists of a call: sized once (a call reports at most every edge / every grain + the dead-column marker)
        self._switching = np.empty((max(self.n_pp, 1), 2), dtype=np.int64)
        self._extra = np.empty(self.n_grain + 1, dtype=np.int64)
        self._args = _lib.TopologyArgs()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ggnn_topology_close(self._h)
            self._h = None

    __del__ = close

    def apply(self, x_joint: np.ndarray, y_joint: np.ndarray, y_grain_area: np.ndarray, edge_prob: np.ndarray,
              grain_event, mask_grain: np.ndarray, mask_joint: np.ndarray, threshold: float,
              active_grain: Optional[np.ndarray] = None, active_joint: Optional[np.ndarray] = None):
        """One update.  x_joint [N_j, >= 8] / y_joint [N_j, 2] fp32 C-contiguous and the int64 masks (any shape with N
        elements, contiguous) are rewritten in place; y_grain_area = a float32 view whose element g is the predicted area
        change of grain g (e.g. y_grain[:, 0]: strided views are taken as they are); edge_prob [n_pp] float32;
        active_* = one byte per node or None.  Returns (grain events incl. forced ones, switching list [S, 2])."""
        n_j, n_g = self.n_joint, self.n_grain
        for name, a, shape in (("x_joint", x_joint, None), ("y_joint", y_joint, (n_j, 2))):
            if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.ndim != 2 or a.shape[0] != n_j \
                    or not a.flags.c_contiguous or (shape is not None and a.shape != shape):
                raise ValueError(f"{name} must be a C-contiguous float32 array [{n_j}, {'2' if shape else '>= 8'}]")
        if x_joint.shape[1] < 8:
            raise ValueError("x_joint needs the 8 junction features (columns 6, 7 = the displacement the events reset)")
        for name, a, n in (("mask_grain", mask_grain, n_g), ("mask_joint", mask_joint, n_j)):
            if not isinstance(a, np.ndarray) or a.dtype != np.int64 or a.size != n or not a.flags.c_contiguous:
                raise ValueError(f"{name} must be a contiguous int64 array of {n} elements")
        area = y_grain_area
        if not isinstance(area, np.ndarray) or area.dtype != np.float32 or area.ndim != 1 or area.shape[0] != n_g \
                or area.strides[0] % 4:
            raise ValueError("y_grain_area must be a float32 vector with one element per grain")
        prob = np.ascontiguousarray(np.asarray(edge_prob, dtype=np.float32).reshape(-1))
        if prob.shape[0] != self.n_pp:
            raise ValueError(f"edge_prob needs one entry per junction edge ({self.n_pp})")
        ge = np.ascontiguousarray(np.asarray(grain_event, dtype=np.int64).reshape(-1))
        A = self._args
        A.pp = A.pq = None
        A.x_joint, A.y_joint, A.y_grain_area, A.edge_prob = _p(x_joint), _p(y_joint), _p(area), _p(prob)
        A.grain_event, A.mask_grain, A.mask_joint = _p(ge), _p(mask_grain), _p(mask_joint)
        A.active_grain, A.active_joint = _p(active_grain), _p(active_joint)
        A.switching, A.events_extra = _p(self._switching), _p(self._extra)
        A.n_joint, A.n_grain, A.ldx, A.ldyg, A.n_grain_event = n_j, n_g, x_joint.shape[1], area.strides[0] // 4, ge.shape[0]
        A.switching_cap, A.extra_cap, A.threshold = self._switching.shape[0], self._extra.shape[0], float(threshold)
        rc = self._lib.ggnn_topology_apply(self._h, ctypes.byref(A))
        if rc == _lib.GGNN_ETOPOLOGY:
            raise TopologyError(A.error.decode(errors="replace"))
        _lib.check(rc, "ggnn_topology_apply")
        self.n_pp, self.n_pq = int(A.n_pp), int(A.n_pq)
        events = np.concatenate([ge, self._extra[:A.n_extra]]) if A.n_extra else ge
        return events, self._switching[:A.n_switching].copy()

    def export(self, pp: Optional[np.ndarray] = None, pq: Optional[np.ndarray] = None, qp: Optional[np.ndarray] = None):
        """The current lists into int64 arrays [2, >= n] whose rows are contiguous (row stride = shape[1] elements); with
        no arguments: three fresh arrays (ei_jj [2, n_pp], ei_jg [2, n_pq], ei_gj [2, n_pq])."""
        fresh = pp is None and pq is None and qp is None
        if fresh:
            pp = np.empty((2, self.n_pp), np.int64)
            pq = np.empty((2, self.n_pq), np.int64)
            qp = np.empty((2, self.n_pq), np.int64)
        for a, n in ((pp, self.n_pp), (pq, self.n_pq), (qp, self.n_pq)):
            if a is not None and (a.dtype != np.int64 or a.ndim != 2 or a.shape[0] != 2 or a.shape[1] < n
                                  or not a.flags.c_contiguous):
                raise ValueError("export targets must be C-contiguous int64 arrays [2, >= n]")
        ld = lambda a: 0 if a is None else a.shape[1]
        _lib.check(self._lib.ggnn_topology_export(self._h, _p(pp), ld(pp), _p(pq), ld(pq), _p(qp), ld(qp)),
                   "ggnn_topology_export")
        return (pp, pq, qp) if fresh else None


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
    lib = _library()
    n_j, n_g = int(mask_joint.shape[0]), int(mask_grain.shape[0])
    for name, a, cols in (("x_joint", x_joint, None), ("y_joint", y_joint, 2)):
        if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.ndim != 2 or a.shape[0] != n_j \
                or (cols is not None and a.shape[1] != cols):
            raise ValueError(f"{name} must be a float32 numpy array [{n_j}, {cols or '>= 8'}]")
    if x_joint.shape[1] < 8:
        raise ValueError("x_joint needs the 8 junction features (columns 6, 7 = the displacement the events reset)")
    ge = np.ascontiguousarray(np.asarray(grain_event, dtype=np.int64).reshape(-1))
    n_pp, n_pq = int(ei_jj.shape[1]), int(ei_jg.shape[1])
    # (room for the result: a removed grain appends two columns and drops more than that)
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
