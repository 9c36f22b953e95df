"""The oracle's pin: oracle/grainnn_oracle.py against the golden vectors that
tests/golden/make_golden.py produced by running the unmodified reference (CPU only)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import (EDGE_TYPES, GOLDEN, assert_close, etk, fold_120, golden, load_graph,
                     oracle_models, random_state, tt)
from oracle import grainnn_oracle as oracle

TOL = 2e-5  # bit-exact where generated; slack only for a different BLAS on another host


def test_parameter_counts_and_keys_match_reference_logs():
    """model/regressor0_logfile:40 -> 1 204 612, model/classifier1_logfile:40 -> 1 204 806."""
    R, Cm = oracle_models(1)
    keys = json.load(open(os.path.join(GOLDEN, "keys.json")))
    assert sum(p.numel() for p in R.parameters()) == 1204612 == keys["n_params"]["regressor"]
    assert sum(p.numel() for p in Cm.parameters()) == 1204806 == keys["n_params"]["classifier"]
    for name, m in (("regressor", R), ("classifier", Cm)):
        sd = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert sd == keys[name]
        assert len(sd) == 284


@torch.no_grad()
def test_periodconv_and_cells_cfg1():
    x, ei, ea = load_graph("40")
    g = golden("cfg1_s1")
    R, _ = oracle_models(10020)
    n_nodes = {nt: v.shape[0] for nt, v in x.items()}
    h0, c0 = random_state(n_nodes, 7), random_state(n_nodes, 8)
    X, EI, EA = tt(x), tt(ei), tt(ea)
    xh = {nt: torch.cat([X[nt], torch.from_numpy(h0[nt])], 1) for nt in x}
    for et in EDGE_TYPES:
        conv = R.gclstm_decoder.cell_list[0].conv_i.convs[etk(et)]
        assert_close(conv(xh[et[0]], xh[et[-1]], EI[et], EA[et]), g["conv_" + etk(et)], f"conv {et}", TOL)
    h, c = R.gclstm_encoder.cell_list[0](X, EI, EA, None, None)
    for nt in x:
        assert_close(h[nt], g[f"cell0_h_{nt}"], f"cell0 h {nt}", TOL)
        assert_close(c[nt], g[f"cell0_c_{nt}"], f"cell0 c {nt}", TOL)
    h, c = R.gclstm_decoder.cell_list[0](X, EI, EA, tt(h0), tt(c0))
    for nt in x:
        assert_close(h[nt], g[f"cell1_h_{nt}"], f"cell1 h {nt}", TOL)
        assert_close(c[nt], g[f"cell1_c_{nt}"], f"cell1 c {nt}", TOL)


def _inputs(tag):
    if tag.startswith("cfg1"):
        return load_graph("40")
    x, ei, ea = load_graph("120")
    x, ea = fold_120(x, ea)
    return x, ei, ea


@pytest.mark.parametrize("tag,seed,scale", [("cfg1_s1", 10020, 1.0), ("cfg1_s3", 10020, 3.0),
                                            ("cfg2_s1", 0, 1.0)])
@torch.no_grad()
def test_forward_and_rollout(tag, seed, scale):
    x, ei, ea = _inputs(tag)
    g = golden(tag)
    assert list(g["meta"][:2]) == [seed, scale]
    n_steps, span = int(g["meta"][2]), int(g["meta"][3])
    R, Cm = oracle_models(seed, scale)
    X, EI, EA = tt(x), tt(ei), tt(ea)
    yr, yc = R(X, EI, EA), Cm(X, EI, EA)
    for k in ("joint", "grain", "grain_area"):
        assert_close(yr[k], g["R_" + k], f"{tag} R {k}", TOL)
    for k in ("edge_event", "edge"):
        assert_close(yc[k], g["C_" + k], f"{tag} C {k}", TOL)
    for step in range(1, n_steps + 1):
        _, EA = oracle.rollout_step(R, Cm, X, EI, EA, span)
        if step in (1, n_steps):
            for nt in x:
                assert_close(X[nt], g[f"step{step}_x_{nt}"], f"{tag} step{step} x {nt}", 1e-4)
            for et in EDGE_TYPES:
                assert_close(EA[et], g[f"step{step}_ea_{etk(et)}"], f"{tag} step{step} ea {et}", 1e-4)


@torch.no_grad()
def test_rollout_with_grain_centres():
    """vectors produced by the reference's own graph_trajectory.GNN_update /
    graph.update() driven from graphs/40_40/traj10020.pkl.gz (make_golden.generate_centres)."""
    x, ei, ea = load_graph("40")
    g = golden("cfg1_centres")
    R, Cm = oracle_models(10020, 1.0)
    X, EI, EA = tt(x), tt(ei), tt(ea)
    c0 = oracle.grain_centres(X["joint"][:, :2], EI[EDGE_TYPES[0]], X["grain"].size(0)).numpy()
    assert np.array_equal(c0, g["init_region_center"])          # float64, bit-exact
    for step in range(1, 4):
        _, EA = oracle.rollout_step(R, Cm, X, EI, EA, 6, centres=(1.0, None))
        if step in (1, 3):
            for nt in x:
                assert_close(X[nt], g[f"step{step}_x_{nt}"], f"centres step{step} x {nt}", 1e-4)
            for et in EDGE_TYPES:
                assert_close(EA[et], g[f"step{step}_ea_{etk(et)}"], f"centres step{step} ea {et}", 1e-4)
    # the centres really moved (this golden is not the static-geometry one)
    assert np.abs(g["step3_x_grain"][:, :2] - x["grain"][:, :2]).max() > 1e-3


def test_grain_centres_order_and_wrap_properties():
    """Centres do not depend on which junction starts the chain, land in (-eps, 2) and follow a
    rigid periodic shift of the junctions (a grain straddling the boundary is handled by the
    +1 rule of graph_datastruct.py:696-704)."""
    x, ei, _ = load_graph("40")
    gj = torch.from_numpy(ei[EDGE_TYPES[0]])
    xy = torch.from_numpy(x["joint"][:, :2].copy())
    c = oracle.grain_centres(xy, gj, 118).numpy()
    perm = torch.randperm(gj.size(1), generator=torch.Generator().manual_seed(3))
    assert np.abs(oracle.grain_centres(xy, gj[:, perm], 118).numpy() - c).max() < 1e-6
    assert (c > -1e-12).all() and (c < 2).all()
    shifted = (xy + torch.tensor([0.37, 0.61])) % 1
    cs = oracle.grain_centres(shifted, gj, 118).numpy()
    d = (cs - c - np.array([0.37, 0.61])) % 1
    assert np.minimum(d, 1 - d).max() < 1e-6


def test_fixture_invariants():
    """graph_trajectory.py:985-988: every junction has exactly 3 grain and 3 junction
    neighbours; E = 3 N_j for all three edge types; N_j = 2 N_g on the torus."""
    for name, (ng, nj) in (("40", (118, 236)), ("120", (1043, 2086))):
        x, ei, ea = load_graph(name)
        assert x["grain"].shape == (ng, 11) and x["joint"].shape == (nj, 8) and nj == 2 * ng
        for et in EDGE_TYPES:
            assert ei[et].shape == (2, 3 * nj) and ea[et].shape == (3 * nj, 1)
        assert (np.bincount(ei[EDGE_TYPES[0]][1], minlength=nj) == 3).all()
        assert (np.bincount(ei[EDGE_TYPES[2]][1], minlength=nj) == 3).all()
        gj = set(map(tuple, ei[EDGE_TYPES[0]].T))
        assert all((b, a) in gj for a, b in map(tuple, ei[EDGE_TYPES[1]].T))  # models.py:841


def test_z_clamp_branch():
    """test.py:405-407 fires once z passes 120/121."""
    X = {"grain": torch.zeros(3, 11), "joint": torch.zeros(5, 8)}
    X["grain"][:, 2] = 0.96
    X["joint"][:, 2] = 0.96
    oracle.advance_z(X, 6)
    zmax = torch.tensor(120 / 121, dtype=torch.float32)
    assert (X["grain"][:, 2] == zmax).all() and (X["joint"][:, 2] == zmax).all()
