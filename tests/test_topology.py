"""the host-side topology update against vectors produced by the unmodified
reference `GrainNN_classifier.update` (tests/golden/make_golden_events.py).  Integer work:
bit-exact, including the COLUMN ORDER of every edge list."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from graingraphnn_amd.topology import GJ, JG, JJ, TopologyError, update_topology

EV = dict(np.load(os.path.join(GOLDEN, "golden_cfg1_events.npz")))
# + 12 random scenarios (0-3 eliminations and 0-4 switches each, chained up to 4 deep) recorded
# by tests/golden/fuzz_events.py, which also ran several hundred more against the reference
EV.update(np.load(os.path.join(GOLDEN, "golden_cfg1_events_fuzz.npz")))
SCENARIOS = ["elim1", "switch1", "switch3", "mixed", "mass3", "mass4"] + [f"fuzz{i:02d}" for i in range(12)]


def k(et):
    return "ei_" + "__".join(et)


def run(name):
    i, o = name + "__in_", name + "__out_"
    xj = EV[i + "x_joint"].copy()
    yj = EV[i + "y_joint"].copy()
    mg, mj = EV[i + "mask_grain"].copy(), EV[i + "mask_joint"].copy()
    prob = torch.sigmoid(torch.from_numpy(EV[i + "edge_event"])).numpy()
    res = update_topology(xj, EV[i + k(JJ)], EV[i + k(JG)], yj, EV[i + "y_grain"], prob,
                          EV[i + "grain_event"], mg, mj, 0.6)
    return res, xj, yj, mg, mj, o


@pytest.mark.parametrize("name", SCENARIOS)
def test_update_matches_reference_bit_for_bit(name):
    (pp, pq, qp, sw, events), xj, yj, mg, mj, o = run(name)
    assert np.array_equal(pp, EV[o + k(JJ)]), "joint-joint edge list (values or column order)"
    assert np.array_equal(pq, EV[o + k(JG)]), "joint-grain edge list"
    assert np.array_equal(qp, EV[o + k(GJ)]), "grain-joint edge list"
    assert np.array_equal(sw, EV[o + "switching_list"])
    assert np.array_equal(events, EV[o + "grain_event"])
    assert np.array_equal(mg, EV[o + "mask_grain"]) and np.array_equal(mj, EV[o + "mask_joint"])
    assert np.array_equal(xj, EV[o + "x_joint"]), "junction features (fp32, bit-exact)"
    assert np.array_equal(yj, EV[o + "y_joint"])
    assert np.array_equal(EV[name + "__in_x_grain"], EV[o + "x_grain"])     # grains are not touched here


@pytest.mark.parametrize("name", SCENARIOS)
def test_result_is_a_valid_grain_graph(name):
    """graph_trajectory.py:985-988 invariants survive every update: live junctions have exactly
    three grain and three junction neighbours, edge lists are symmetric, E = 3 N_live."""
    (pp, pq, qp, sw, events), xj, yj, mg, mj, o = run(name)
    live_j = np.flatnonzero(mj[:, 0] > 0)
    assert pp.shape[1] == pq.shape[1] == 3 * len(live_j)
    assert (np.bincount(pp[0], minlength=len(mj))[live_j] == 3).all()
    assert (np.bincount(pq[0], minlength=len(mj))[live_j] == 3).all()
    assert set(map(tuple, pp.T)) == set(map(tuple, pp[::-1].T))
    dead_g = np.flatnonzero(mg[:, 0] == 0)
    assert not np.isin(pq[1], dead_g).any() and set(events.tolist()) <= set(dead_g.tolist())
    # Euler on the torus: junctions = 2 x grains
    assert len(live_j) == 2 * int((mg[:, 0] > 0).sum())


def test_quiet_step_is_the_identity():
    i = "elim1__in_"
    xj, yj = EV[i + "x_joint"].copy(), EV[i + "y_joint"].copy()
    mg, mj = EV[i + "mask_grain"].copy(), EV[i + "mask_joint"].copy()
    pp, pq, qp, sw, ev = update_topology(xj, EV[i + k(JJ)], EV[i + k(JG)], yj, EV[i + "y_grain"],
                                         np.zeros(EV[i + k(JJ)].shape[1], np.float32), [], mg, mj, 0.6)
    assert np.array_equal(pp, EV[i + k(JJ)]) and np.array_equal(pq, EV[i + k(JG)])
    assert np.array_equal(xj, EV[i + "x_joint"]) and len(sw) == 0 and len(ev) == 0


def test_invalid_lists_fail_loudly():
    i = "elim1__in_"
    pq = EV[i + k(JG)].copy()
    pq[1, np.flatnonzero(pq[1] == 44)[0]] = 45          # grain 44 loses a corner: no longer a polygon
    with pytest.raises((TopologyError, IndexError, ValueError)):
        update_topology(EV[i + "x_joint"].copy(), EV[i + k(JJ)], pq, EV[i + "y_joint"].copy(),
                        EV[i + "y_grain"], np.zeros(708, np.float32), [44],
                        EV[i + "mask_grain"].copy(), EV[i + "mask_joint"].copy(), 0.6)


def test_classifier_update_is_a_drop_in():
    """The reference call of test.py:426 on torch tensors: same argument list, same in-place
    effects, same return values as `GrainNN_classifier.update` of the reference."""
    from helpers import product_models
    _, Cm = product_models(10020)
    Cm.threshold = 0.6
    i, o = "mixed__in_", "mixed__out_"
    t = lambda a: torch.from_numpy(np.array(a, copy=True))
    x = {"joint": t(EV[i + "x_joint"]), "grain": t(EV[i + "x_grain"])}
    ei = {et: t(EV[i + k(et)]) for et in (GJ, JG, JJ)}
    y = {"joint": t(EV[i + "y_joint"]), "grain": t(EV[i + "y_grain"]), "edge_event": t(EV[i + "edge_event"]),
         "grain_event": t(EV[i + "grain_event"])}
    mask = {"grain": t(EV[i + "mask_grain"]), "joint": t(EV[i + "mask_joint"])}
    gs = {"active_grains": torch.arange(118), "active_joints": torch.arange(236)}
    old_jj = ei[JJ]
    x2, ei2, sw = Cm.update(x, ei, None, y, mask, gs, 0.0)
    assert x2 is x and ei2 is ei and ei[JJ] is not old_jj
    for et in (GJ, JG, JJ):
        assert torch.equal(ei[et], t(EV[o + k(et)]))
    assert torch.equal(x["joint"], t(EV[o + "x_joint"])) and torch.equal(y["joint"], t(EV[o + "y_joint"]))
    assert torch.equal(y["grain_event"], t(EV[o + "grain_event"])) and torch.equal(sw, t(EV[o + "switching_list"]))
    assert torch.equal(mask["grain"], t(EV[o + "mask_grain"])) and torch.equal(mask["joint"], t(EV[o + "mask_joint"]))
    with pytest.raises(NotImplementedError):
        Cm.update(x, ei, None, y, mask, gs, 0.01)      # nucleation is out of scope
    # a junction outside the active window freezes its events (models.py:640-645, 911)
    x = {"joint": t(EV[i + "x_joint"]), "grain": t(EV[i + "x_grain"])}
    ei = {et: t(EV[i + k(et)]) for et in (GJ, JG, JJ)}
    y = {"joint": t(EV[i + "y_joint"]), "grain": t(EV[i + "y_grain"]),
         "edge_event": torch.full((708,), -8.0), "grain_event": torch.tensor([60])}
    mask = {"grain": t(EV[i + "mask_grain"]), "joint": t(EV[i + "mask_joint"])}
    corner = int(ei[JG][0][ei[JG][1] == 60][0])
    gs = {"active_grains": torch.arange(118), "active_joints": torch.tensor([j for j in range(236) if j != corner])}
    Cm.update(x, ei, None, y, mask, gs, 0.0)
    assert torch.equal(ei[JJ], t(EV[i + k(JJ)])) and int(mask["grain"].sum()) == 118


def _random_events(n, seed, depth, n_elim, n_switch):
    """`depth` chained updates on an n x n periodic honeycomb with random predictions: (arguments of every call)."""
    from graingraphnn_amd import synthetic
    rs = np.random.RandomState(seed)
    x, ei, _ = synthetic.honeycomb(n, 1, seed)
    xj = np.ascontiguousarray(x["joint"], dtype=np.float32)
    n_g, n_j = x["grain"].shape[0], xj.shape[0]
    state = dict(xj=xj, pp=ei[JJ].astype(np.int64), pq=ei[JG].astype(np.int64),
                 mg=np.ones((n_g, 1), np.int64), mj=np.ones((n_j, 1), np.int64))
    for _ in range(depth):
        yj = (rs.normal(0, 0.02 / n, (n_j, 2)) * 5).astype(np.float32)
        yg = rs.normal(0, 1, (n_g, 2)).astype(np.float32)
        live = np.flatnonzero(state["mg"][:, 0] > 0)
        ge = rs.choice(live, size=min(n_elim, len(live)), replace=False)
        prob = np.zeros(state["pp"].shape[1], np.float32)
        up = np.flatnonzero(state["pp"][0] < state["pp"][1])
        prob[rs.choice(up, size=min(n_switch, len(up)), replace=False)] = rs.uniform(0.7, 1.0, min(n_switch, len(up)))
        yield state, yj, yg, prob, ge


@pytest.mark.parametrize("n,seed,n_elim,n_switch", [(12, 0, 3, 4), (12, 1, 6, 0), (20, 2, 8, 10), (40, 3, 20, 25)])
def test_indexed_lookups_equal_the_scan_formulation_on_random_events(n, seed, n_elim, n_switch):
    """The product answers the update's lookups from column indices; oracle/topology_scan.py answers them with the
    reference's full-array masks.  Four chained updates on honeycombs of up to 1 600 grains with random eliminations and
    switches: identical lists (column order included), masks, coordinates, event lists -- or the same refusal."""
    from oracle import topology_scan as scan
    compared = 0
    for state, yj, yg, prob, ge in _random_events(n, seed, 4, n_elim, n_switch):
        a = dict(xj=state["xj"].copy(), yj=yj.copy(), mg=state["mg"].copy(), mj=state["mj"].copy())
        b = dict(xj=state["xj"].copy(), yj=yj.copy(), mg=state["mg"].copy(), mj=state["mj"].copy())
        res = []
        for mod, s in ((scan, a), (__import__("graingraphnn_amd.topology", fromlist=["x"]), b)):
            try:
                res.append(mod.update_topology(s["xj"], state["pp"], state["pq"], s["yj"], yg, prob, ge, s["mg"], s["mj"], 0.6))
            except (mod.TopologyError, IndexError) as err:
                res.append(type(err).__name__)
        if isinstance(res[0], str) or isinstance(res[1], str):
            assert isinstance(res[0], str) and isinstance(res[1], str), (res[0] if isinstance(res[0], str) else "ok",
                                                                         res[1] if isinstance(res[1], str) else "ok")
            break   # (both refused these events: the chain ends here)
        for u, v in zip(res[0], res[1]):
            assert np.array_equal(u, v)
        for key in ("xj", "yj", "mg", "mj"):
            assert np.array_equal(a[key], b[key]), key
        compared += 1
        state.update(xj=b["xj"], pp=res[1][0], pq=res[1][1], mg=b["mg"], mj=b["mj"])
    assert compared >= 1


def test_c_abi_refuses_bad_arguments_and_missing_room():
    """ggnn_topology_update through the C ABI: null pointers and inconsistent sizes are GGNN_EINVAL; an edge list without
    room for the two columns a removed grain appends is GGNN_ETOPOLOGY with a message, not a write past the end."""
    import ctypes
    from graingraphnn_amd import _lib
    lib = _lib.load()
    A = _lib.TopologyArgs()
    assert lib.ggnn_topology_update(None) == -1
    assert lib.ggnn_topology_update(ctypes.byref(A)) == -1          # all pointers null
    i = "elim1__in_"
    pp, pq = np.ascontiguousarray(EV[i + k(JJ)]).copy(), np.ascontiguousarray(EV[i + k(JG)]).copy()
    xj, yj = EV[i + "x_joint"].copy(), EV[i + "y_joint"].copy()
    area = np.ascontiguousarray(EV[i + "y_grain"][:, 0])
    prob = np.zeros(pp.shape[1], np.float32)
    ge = np.ascontiguousarray(EV[i + "grain_event"].astype(np.int64).reshape(-1))
    assert len(ge) >= 1
    mg = np.ascontiguousarray(EV[i + "mask_grain"].reshape(-1).astype(np.int64))
    mj = np.ascontiguousarray(EV[i + "mask_joint"].reshape(-1).astype(np.int64))
    sw, extra = np.empty((8, 2), np.int64), np.empty(8, np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    A.pp, A.pq, A.n_pp, A.n_pq, A.pp_cap, A.pq_cap = p(pp), p(pq), pp.shape[1], pq.shape[1], pp.shape[1], pq.shape[1]
    A.x_joint, A.y_joint, A.y_grain_area, A.edge_prob = p(xj), p(yj), p(area), p(prob)
    A.grain_event, A.mask_grain, A.mask_joint, A.switching, A.events_extra = p(ge), p(mg), p(mj), p(sw), p(extra)
    A.n_joint, A.n_grain, A.ldx, A.ldyg, A.n_grain_event = len(mj), len(mg), xj.shape[1], 1, len(ge)
    A.switching_cap, A.extra_cap, A.threshold = 8, 8, 0.6
    A.ldx = 4
    assert lib.ggnn_topology_update(ctypes.byref(A)) == -1          # fewer than the 8 junction features
    A.ldx = xj.shape[1]
    A.pp_cap = pp.shape[1] - 1
    assert lib.ggnn_topology_update(ctypes.byref(A)) == -1          # capacity below the columns in use
    A.pp_cap = pp.shape[1]                                          # no room for the appended pair
    assert lib.ggnn_topology_update(ctypes.byref(A)) == _lib.GGNN_ETOPOLOGY
    assert b"room" in A.error
