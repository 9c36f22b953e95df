"""Generate the golden vectors in this directory by running the UNMODIFIED reference
(`/root/reference/models.py`, imported through the PyG-subset stand-in `tools/oracle_stub/`).

Runs only in the build container (needs /root/reference); the GPU box and the test-suite only
read the committed `.npz` outputs.  Usage:  python tests/golden/make_golden.py

What it writes
  graph_40.npz   / graph_120.npz    the reference's two test graphs as neutral arrays
                                    (graphs/40_40/seed10020_*.pkl, graphs/120_120/seed0_*.pkl;
                                    float64 -> float32 exactly as data_loader.py:65,79,86 casts)
  golden_cfg1_s1.npz                40 um graph, weights RandomState(10020) x1.0
  golden_cfg1_s3.npz                40 um graph, weights RandomState(10020) x3.0 (saturating)
  golden_cfg2_s1.npz                120 um graph after x3 patch folding (test.py:29-55), RandomState(0)
  keys.json                         reference state_dict keys/shapes + parameter counts
It also checks the CPU oracle (`oracle/grainnn_oracle.py`) against the reference on every
vector (asserts max|a-b| <= 2e-5 * max|b|) -- that is the oracle's pin.
"""
import ast
import json
import os
import sys
from types import SimpleNamespace

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(ROOT, "tools", "oracle_stub"), REF, ROOT]

import dill  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

import models as ref_models  # noqa: E402  -- the reference itself
from graingraphnn_amd.seeding import seeded_state_dict  # noqa: E402
from oracle import grainnn_oracle as oracle  # noqa: E402

torch.set_num_threads(4)
GJ, JG, JJ = ("grain", "push", "joint"), ("joint", "pull", "grain"), ("joint", "connect", "joint")
ETS = (GJ, JG, JJ)
SPAN = 6


def ref_function(pyfile, name):
    """Execute ONE function definition of a reference script without importing the script
    (test.py pulls in tvtk/h5py at import time).  Nothing is copied into the repo."""
    src = open(os.path.join(REF, pyfile)).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), pyfile, "exec"), ns)
    return ns[name]


ref_scale_feature_patchs = ref_function("test.py", "scale_feature_patchs")


def load_graph(path):
    with open(path, "rb") as f:
        g = dill.load(f)[0]
    x = {k: np.asarray(v).astype(np.float32) for k, v in g.feature_dicts.items()}
    ei = {k: np.asarray(v).astype(np.int64) for k, v in g.edge_index_dicts.items()}
    ea = {k: np.asarray(v).astype(np.float32) for k, v in g.edge_weight_dicts.items()}
    return g, x, ei, ea


def save_graph(name, x, ei, ea):
    d = {"x_grain": x["grain"], "x_joint": x["joint"]}
    for et in ETS:
        d["ei_" + "__".join(et)] = ei[et]
        d["ea_" + "__".join(et)] = ea[et]
    np.savez_compressed(os.path.join(HERE, name), **d)


def tt(d):
    return {k: torch.from_numpy(v.copy()) for k, v in d.items()}


def make_hyper(g):
    return SimpleNamespace(features=g.features, targets=g.targets, layer_size=96, layers=1,
                           metadata=(list(g.features.keys()), [GJ, JG, JJ]), out_win=1, window=1,
                           device="cpu")


def build_reference(hp, x, ei, ea, seed, scale):
    """test.py:177-184 with seeded weights instead of the stripped .pt files."""
    R = ref_models.GrainNN_regressor(hp)
    with torch.no_grad():
        R(tt(x), tt(ei), tt(ea))  # materialise PyG lazy Linear(-1, .) (train.py:99-107)
    Cm = ref_models.GrainNN_classifier(hp, R)
    for m, s in ((R, seed), (Cm, seed + 1)):
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(seeded_state_dict(shapes, s, scale))
        m.eval()
    return R, Cm


def build_oracle(hp, seed, scale):
    R = oracle.GrainNN_regressor(hp)
    Cm = oracle.GrainNN_classifier(hp, R)
    for m, s in ((R, seed), (Cm, seed + 1)):
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(seeded_state_dict(shapes, s, scale))
        m.eval()
    return R, Cm


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


CHECKS = []


def pin(name, got, ref, tol=2e-5):
    e = rel_err(got, ref)
    CHECKS.append((name, e))
    assert e <= tol, f"oracle disagrees with the reference on {name}: rel err {e:.3e}"


def random_state(n_nodes, seed):
    rs = np.random.RandomState(seed)
    return {nt: rs.uniform(-1, 1, (n, 96)).astype(np.float32) for nt, n in sorted(n_nodes.items())}


@torch.no_grad()
def generate(tag, x, ei, ea, hp, seed, scale, full):
    R, Cm = build_reference(hp, x, ei, ea, seed, scale)
    oR, oC = build_oracle(hp, seed, scale)
    out = {}
    n_nodes = {nt: v.shape[0] for nt, v in x.items()}
    h0, c0 = random_state(n_nodes, 7), random_state(n_nodes, 8)

    if full:
        # (1) one PeriodConv per edge type on cat[x, h] (decoder input gate)
        xh = {nt: np.concatenate([x[nt], h0[nt]], 1) for nt in x}
        for et in ETS:
            conv = R.gclstm_decoder.cell_list[0].conv_i.convs["__".join(et)]
            xs, xd = torch.from_numpy(xh[et[0]]), torch.from_numpy(xh[et[-1]])
            arg = xs if et[0] == et[-1] else (xs, xd)
            y = conv(arg, torch.from_numpy(ei[et]), edge_attr=torch.from_numpy(ea[et])).numpy()
            out["conv_" + "__".join(et)] = y
            oconv = oR.gclstm_decoder.cell_list[0].conv_i.convs["__".join(et)]
            pin(f"{tag}/conv/{et}", oconv(xs, xd, torch.from_numpy(ei[et]), torch.from_numpy(ea[et])).numpy(), y)
        # (2) HeteroPGCLSTM: zero state (encoder cell) and non-zero state (decoder cell)
        cell, ocell = R.gclstm_encoder.cell_list[0], oR.gclstm_encoder.cell_list[0]
        h, c = cell(tt(x), tt(ei), tt(ea), None, None)
        oh, oc = ocell(tt(x), tt(ei), tt(ea), None, None)
        for nt in x:
            out[f"cell0_h_{nt}"], out[f"cell0_c_{nt}"] = h[nt].numpy(), c[nt].numpy()
            pin(f"{tag}/cell0/h/{nt}", oh[nt].numpy(), h[nt].numpy())
            pin(f"{tag}/cell0/c/{nt}", oc[nt].numpy(), c[nt].numpy())
        cell, ocell = R.gclstm_decoder.cell_list[0], oR.gclstm_decoder.cell_list[0]
        h, c = cell(tt(x), tt(ei), tt(ea), tt(h0), tt(c0))
        oh, oc = ocell(tt(x), tt(ei), tt(ea), tt(h0), tt(c0))
        for nt in x:
            out[f"cell1_h_{nt}"], out[f"cell1_c_{nt}"] = h[nt].numpy(), c[nt].numpy()
            pin(f"{tag}/cell1/h/{nt}", oh[nt].numpy(), h[nt].numpy())
            pin(f"{tag}/cell1/c/{nt}", oc[nt].numpy(), c[nt].numpy())

    # (3) model forwards (test.py:382-383)
    yr, yc = R(tt(x), tt(ei), tt(ea)), Cm(tt(x), tt(ei), tt(ea))
    oyr, oyc = oR(tt(x), tt(ei), tt(ea)), oC(tt(x), tt(ei), tt(ea))
    for k in ("joint", "grain", "grain_area"):
        out["R_" + k] = yr[k].numpy()
        pin(f"{tag}/R/{k}", oyr[k].numpy(), yr[k].numpy())
    for k in ("edge_event", "edge"):
        out["C_" + k] = yc[k].numpy()
        pin(f"{tag}/C/{k}", oyc[k].numpy(), yc[k].numpy())

    # (4) rollout steps with static topology: reference ops of test.py:382-407, 562-575
    n_steps = 3
    X, EI, EA = tt(x), tt(ei), tt(ea)
    oX, oEA = tt(x), tt(ea)
    for step in range(1, n_steps + 1):
        pred = R(X, EI, EA)
        pred.update(Cm(X, EI, EA))
        R.update(X, pred, {})                                    # test.py:400
        X["grain"][:, 2] += SPAN / (120 + 1)                     # test.py:401-402
        X["joint"][:, 2] += SPAN / (120 + 1)
        if X["grain"][0, 2] > 120 / (120 + 1):                   # test.py:405-407
            X["grain"][:, 2] = 120 / (120 + 1)
            X["joint"][:, 2] = 120 / (120 + 1)
        EA = {}
        for edge_type, index in EI.items():                      # test.py:562-575
            src_x = X[edge_type[0]][index[0], :2]
            dst_x = X[edge_type[-1]][index[-1], :2]
            rel_loc = src_x - dst_x
            rel_loc = -1 * (rel_loc > 0.5) + 1 * (rel_loc < -0.5) + rel_loc
            EA[edge_type] = torch.sqrt(rel_loc[:, 0] ** 2 + rel_loc[:, 1] ** 2).view(-1, 1)
        _, oEA = oracle.rollout_step(oR, oC, oX, EI, oEA, SPAN)
        if step in (1, n_steps):
            for nt in x:
                out[f"step{step}_x_{nt}"] = X[nt].numpy().copy()
                pin(f"{tag}/step{step}/x/{nt}", oX[nt].numpy(), X[nt].numpy(), tol=1e-4)
            for et in ETS:
                out[f"step{step}_ea_" + "__".join(et)] = EA[et].numpy().copy()
                pin(f"{tag}/step{step}/ea/{et}", oEA[et].numpy(), EA[et].numpy(), tol=1e-4)
    out["meta"] = np.array([seed, scale, n_steps, SPAN], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, f"golden_{tag}.npz"), **out)
    return R, Cm


@torch.no_grad()
def generate_centres(x, ei, ea, mask, hp, seed, scale):
    """f-1: the rollout loop of test.py INCLUDING traj.GNN_update -> graph.update() ->
    region_center -> x_grain[:, :2] (test.py:468-478, 556-559), driven by the reference's own
    graph_trajectory object (graphs/40_40/traj10020.pkl.gz).  Cmodel.update (topology surgery,
    SURVEY 8f-2) is skipped like in `generate` (4): topo=False, static connectivity."""
    import gzip
    import __main__
    import graph_trajectory as gt
    __main__.graph_trajectory, __main__.graph = gt.graph_trajectory, gt.graph
    with gzip.open(os.path.join(REF, "graphs/40_40/traj10020.pkl.gz"), "rb") as f:
        traj = dill.load(f)
    traj.raise_err = False
    R, Cm = build_reference(hp, x, ei, ea, seed, scale)
    oR, oC = build_oracle(hp, seed, scale)
    X, EI, EA = tt(x), tt(ei), tt(ea)
    oX, oEA = tt(x), tt(ea)
    M = {k: torch.from_numpy(v.copy()) for k, v in mask.items()}
    M["joint"] = 1 + 0 * M["joint"]                               # test.py:258
    traj.extraV_traj, traj.area_traj = [], traj.area_traj[:1]
    traj.GNN_update(0, {k: v.clone() for k, v in X.items()}, M, True, EI, False)   # test.py:266
    c0 = np.array([traj.region_center[g + 1] for g in range(x["grain"].shape[0])])
    out = {"init_region_center": c0}
    pin("centres/init", oracle.grain_centres(X["joint"][:, :2], EI[GJ], x["grain"].shape[0]).numpy(), c0, tol=0.0)
    for step in range(1, 4):
        pred = R(X, EI, EA)
        pred.update(Cm(X, EI, EA))
        R.update(X, pred, {})
        X["grain"][:, 2] += SPAN / 121
        X["joint"][:, 2] += SPAN / 121
        traj.GNN_update(step * SPAN, {k: v.clone() for k, v in X.items()}, M, False, EI, False)  # :478
        for grain, coor in traj.region_center.items():            # test.py:556-559
            X["grain"][grain - 1, :2] = torch.FloatTensor(coor)
        EA = {}
        for edge_type, index in EI.items():                       # test.py:562-575
            rel_loc = X[edge_type[0]][index[0], :2] - X[edge_type[-1]][index[-1], :2]
            rel_loc = -1 * (rel_loc > 0.5) + 1 * (rel_loc < -0.5) + rel_loc
            EA[edge_type] = torch.sqrt(rel_loc[:, 0] ** 2 + rel_loc[:, 1] ** 2).view(-1, 1)
        _, oEA = oracle.rollout_step(oR, oC, oX, EI, oEA, SPAN, centres=(1.0, None))
        if step in (1, 3):
            for nt in x:
                out[f"step{step}_x_{nt}"] = X[nt].numpy().copy()
                pin(f"centres/step{step}/x/{nt}", oX[nt].numpy(), X[nt].numpy(), tol=1e-5)
            for et in ETS:
                out[f"step{step}_ea_" + "__".join(et)] = EA[et].numpy().copy()
                pin(f"centres/step{step}/ea/{et}", oEA[et].numpy(), EA[et].numpy(), tol=1e-5)
    out["meta"] = np.array([seed, scale, 3, SPAN], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "golden_cfg1_centres.npz"), **out)


def main():
    g40, x40, ei40, ea40 = load_graph(os.path.join(REF, "graphs/40_40/seed10020_G1.904_R0.558_span6.pkl"))
    g120, x120, ei120, ea120 = load_graph(os.path.join(REF, "graphs/120_120/seed0_G10.0_R2.0_span6.pkl"))
    save_graph("graph_40.npz", x40, ei40, ea40)
    save_graph("graph_120.npz", x120, ei120, ea120)
    hp = make_hyper(g40)

    # structural pins: state_dict layout + parameter counts (model/*_logfile:40)
    R, Cm = generate("cfg1_s1", x40, ei40, ea40, hp, 10020, 1.0, full=True)
    n_r = sum(p.numel() for p in R.parameters())
    n_c = sum(p.numel() for p in Cm.parameters())
    assert (n_r, n_c) == (1204612, 1204806), (n_r, n_c)
    keys = {"regressor": {k: list(v.shape) for k, v in R.state_dict().items()},
            "classifier": {k: list(v.shape) for k, v in Cm.state_dict().items()},
            "n_params": {"regressor": n_r, "classifier": n_c},
            "features": g40.features, "targets": g40.targets}
    with open(os.path.join(HERE, "keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)

    generate("cfg1_s3", x40, ei40, ea40, hp, 10020, 3.0, full=False)
    mask40 = {k: np.asarray(v).astype(np.int64) for k, v in g40.mask.items()}
    generate_centres(x40, ei40, ea40, mask40, hp, 10020, 1.0)

    # cfg2: x3 patch folding with the reference's own function (test.py:29-55, 310-312)
    X, EA = tt(x120), tt(ea120)
    ref_scale_feature_patchs(3.0, X, EA, "periodic")
    oX, oEA = tt(x120), tt(ea120)
    oracle.scale_feature_patchs(3.0, oX, oEA)
    for nt in X:
        pin(f"fold/x/{nt}", oX[nt].numpy(), X[nt].numpy(), tol=0.0)
    for et in ETS:
        pin(f"fold/ea/{et}", oEA[et].numpy(), EA[et].numpy(), tol=0.0)
    x120f = {k: v.numpy() for k, v in X.items()}
    ea120f = {k: v.numpy() for k, v in EA.items()}
    # how many edges actually wrap after folding (the point of cfg2)
    rel = x120f["joint"][ei120[JJ][0], :2] - x120f["joint"][ei120[JJ][1], :2]
    print("cfg2 wrapped jj edges:", int((np.abs(rel) > 0.5).any(1).sum()), "of", rel.shape[0])
    generate("cfg2_s1", x120f, ei120, ea120f, make_hyper(g120), 0, 1.0, full=False)

    print(f"{len(CHECKS)} oracle-vs-reference checks, worst:")
    for name, e in sorted(CHECKS, key=lambda t: -t[1])[:8]:
        print(f"  {e:.3e}  {name}")


if __name__ == "__main__":
    main()
