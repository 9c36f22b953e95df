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


@pytest.mark.parametrize("n,seed,n_elim,n_switch", [(12, 5, 3, 4), (20, 6, 8, 10), (40, 7, 12, 20)])
def test_session_equals_chained_stateless_updates(n, seed, n_elim, n_switch):
    """TopologySession (lists, lookup tables and counts kept in the library and patched by every update: ggnn_topology_open /
    _apply / _export) against the stateless call on the lists it returned last time: six chained updates with random
    eliminations and switches give identical lists, masks, coordinates and event lists; a refused update leaves the session
    and the in-place arrays exactly as they were, and the session goes on from there."""
    from graingraphnn_amd.topology import TopologySession
    ses, compared, refused = None, 0, 0
    for state, yj, yg, prob, ge in _random_events(n, seed, 6, n_elim, n_switch):
        if ses is None:
            ses = TopologySession(state["pp"], state["pq"], state["mj"].shape[0], state["mg"].shape[0])
        a = dict(xj=state["xj"].copy(), yj=yj.copy(), mg=state["mg"].copy(), mj=state["mj"].copy())
        b = dict(xj=state["xj"].copy(), yj=yj.copy(), mg=state["mg"].copy(), mj=state["mj"].copy())
        try:
            ra = update_topology(a["xj"], state["pp"], state["pq"], a["yj"], yg, prob, ge, a["mg"], a["mj"], 0.6)
        except TopologyError:
            ra = None
        try:
            ev, sw = ses.apply(b["xj"], b["yj"], yg[:, 0], prob, ge, b["mg"], b["mj"], 0.6)
        except TopologyError:
            assert ra is None
            refused += 1
            for key in ("xj", "yj", "mg", "mj"):   # nothing was touched ...
                assert np.array_equal(b[key], a[key]) and np.array_equal(b[key], state[key] if key != "yj" else yj), key
            pp, pq, qp = ses.export()              # ... and the lists are those of the last accepted update
            assert np.array_equal(pp, state["pp"]) and np.array_equal(pq, state["pq"])
            continue
        assert ra is not None
        pp, pq, qp = ses.export()
        assert np.array_equal(pp, ra[0]) and np.array_equal(pq, ra[1]) and np.array_equal(qp, ra[2])
        assert np.array_equal(sw, ra[3]) and np.array_equal(ev, ra[4])
        for key in ("xj", "yj", "mg", "mj"):
            assert np.array_equal(a[key], b[key]), key
        compared += 1
        state.update(xj=b["xj"], pp=pp, pq=pq, mg=b["mg"], mj=b["mj"])
    assert compared >= 2, (compared, refused)
    # export into wider (staging) arrays: rows at the arrays' own stride
    wide = np.full((2, ses.n_pp + 7), -5, np.int64)
    ses.export(pp=wide)
    assert np.array_equal(wide[:, :ses.n_pp], state["pp"]) and (wide[:, ses.n_pp:] == -5).all()
    ses.close()


def test_session_shrinks_its_tables_when_most_columns_are_dead():
    """Eliminations until fewer than half of the session's columns are live (it then restarts its numbering from the live
    ones: ggnn_topology_session::renumber): still the stateless call's results."""
    from graingraphnn_amd.topology import TopologySession
    from graingraphnn_amd import synthetic
    rs = np.random.RandomState(11)
    x, ei, _ = synthetic.honeycomb(40, 1, 3)
    xj = np.ascontiguousarray(x["joint"], dtype=np.float32)
    n_g, n_j = x["grain"].shape[0], xj.shape[0]
    pp, pq = ei[JJ].astype(np.int64), ei[JG].astype(np.int64)
    mg, mj = np.ones((n_g, 1), np.int64), np.ones((n_j, 1), np.int64)
    ses = TopologySession(pp, pq, n_j, n_g)
    done = 0
    for _ in range(400):
        live = np.flatnonzero(mg[:, 0] > 0)
        if len(live) < 40:
            break
        yj = np.zeros((n_j, 2), np.float32)
        yg = rs.normal(0, 1, (n_g, 2)).astype(np.float32)
        ge = rs.choice(live, size=4, replace=False)
        prob = np.zeros(pp.shape[1], np.float32)
        a = dict(xj=xj.copy(), yj=yj.copy(), mg=mg.copy(), mj=mj.copy())
        try:
            ra = update_topology(a["xj"], pp, pq, a["yj"], yg, prob, ge, a["mg"], a["mj"], 0.6)
        except TopologyError:
            with pytest.raises(TopologyError):
                ses.apply(xj, yj, yg[:, 0], prob, ge, mg, mj, 0.6)
            continue
        ses.apply(xj, yj, yg[:, 0], prob, ge, mg, mj, 0.6)
        pp, pq, _ = ses.export()
        assert np.array_equal(pp, ra[0]) and np.array_equal(pq, ra[1])
        assert np.array_equal(xj, a["xj"]) and np.array_equal(mg, a["mg"]) and np.array_equal(mj, a["mj"])
        done += 1
    assert done >= 20 and pp.shape[1] < ei[JJ].shape[1] // 2, (done, pp.shape)


def test_c_abi_refuses_bad_arguments_and_missing_room():
    """ggnn_topology_update through the C ABI: null pointers and inconsistent sizes are GGNN_EINVAL; an output list without
    room is GGNN_ETOPOLOGY with a message BEFORE anything is rewritten, not a write past the end."""
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
    sw, extra = np.empty((8, 2), np.int64), np.empty(len(mg) + 1, np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    A.pp, A.pq, A.n_pp, A.n_pq, A.pp_cap, A.pq_cap = p(pp), p(pq), pp.shape[1], pq.shape[1], pp.shape[1], pq.shape[1]
    A.x_joint, A.y_joint, A.y_grain_area, A.edge_prob = p(xj), p(yj), p(area), p(prob)
    A.grain_event, A.mask_grain, A.mask_joint, A.switching, A.events_extra = p(ge), p(mg), p(mj), p(sw), p(extra)
    A.n_joint, A.n_grain, A.ldx, A.ldyg, A.n_grain_event = len(mj), len(mg), xj.shape[1], 1, len(ge)
    A.switching_cap, A.extra_cap, A.threshold = 8, len(extra), 0.6
    A.ldx = 4
    assert lib.ggnn_topology_update(ctypes.byref(A)) == -1          # fewer than the 8 junction features
    A.ldx = xj.shape[1]
    A.pp_cap = pp.shape[1] - 1
    assert lib.ggnn_topology_update(ctypes.byref(A)) == -1          # capacity below the columns in use
    A.pp_cap = pp.shape[1]
    A.extra_cap = 8                                                 # no room for the forced eliminations' report
    assert lib.ggnn_topology_update(ctypes.byref(A)) == _lib.GGNN_ETOPOLOGY
    assert b"room" in A.error
    assert np.array_equal(pp, EV[i + k(JJ)]) and np.array_equal(xj, EV[i + "x_joint"])   # refused before anything was rewritten
    A.extra_cap = len(extra)
    assert lib.ggnn_topology_update(ctypes.byref(A)) == 0 and A.n_pp < pp.shape[1]     # (the result fits the columns in use)


def test_host_code_under_sanitizers():
    """csrc/topology.hip -- host code, the only part of the library that is not a kernel -- as plain C++ under
    AddressSanitizer + UndefinedBehaviourSanitizer (`make -C graingraphnn_amd/csrc host-asan`), and this file's tests run
    against that build in a child process (GGNN_TOPOLOGY_LIB redirects topology.py's calls; the sanitizer runtimes are
    preloaded).  CPU only: no GPU sanitizer runs exist on this pool."""
    import shutil
    import subprocess
    import sys
    if os.environ.get("GGNN_TOPOLOGY_LIB"):
        pytest.skip("already inside the sanitizer run")
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    runtimes = [subprocess.run([gxx, f"-print-file-name={n}"], capture_output=True, text=True).stdout.strip()
                for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(r) and os.path.exists(r) for r in runtimes):
        pytest.skip("sanitizer runtimes not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "graingraphnn_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "host-asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    env = dict(os.environ, GGNN_TOPOLOGY_LIB=os.path.join(csrc, "build", "libggnn_topology_asan.so"),
               LD_PRELOAD=" ".join(runtimes), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr \
        and "runtime error" not in r.stderr, r.stdout[-3000:] + r.stderr[-3000:]
