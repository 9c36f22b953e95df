"""Golden vectors for SURVEY 8f-2 (event-driven topology update): inputs and outputs of the
UNMODIFIED reference `GrainNN_classifier.update` (models.py:612-842: grain elimination, neighbour
switching, cleanup; nucleation off as in test.py:88) on the 40 um graph.

Runs only in the build container (needs /root/reference); writes `golden_cfg1_events.npz`.
    python tests/golden/make_golden_events.py

Scenarios (every one starts from reference forwards with weights RandomState(10020) x1.0, after
Rmodel.update + the z advance, exactly where test.py:426 calls Cmodel.update):
  elim1    one grain below the area threshold, no edge above the switching threshold
  switch1  one junction-junction edge above the switching threshold
  switch3  three edges with different probabilities (processed in descending order, test of the
           look-ahead `nxt` rule, models.py:1010-1018)
  mixed    two eliminations and two switches in one call
  mass3    step 3 of the free-running seeded rollout: 22 grains vanish at once
  mass4    step 4 of the same rollout (66 listed grains, 9 more force-eliminated, models.py:685-698)
The per-scenario arrays are `<name>__in_*` and `<name>__out_*`.
"""
import gzip
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (sets up sys.path for the reference + stubs)
import numpy as np  # noqa: E402
import torch  # noqa: E402

GJ, JG, JJ = mg.GJ, mg.JG, mg.JJ
SPAN = 6


def step_to_update_point(R, Cm, X, EI, EA, M):
    """test.py:382-418: forwards, Rmodel.update, z advance, grain-event list."""
    gs = {"domain_offset": 0, "domain_factor": 1.0}
    pred = R(X, EI, EA)
    pred.update(Cm(X, EI, EA))
    R.update(X, pred, gs)
    X["grain"][:, 2] += SPAN / 121
    X["joint"][:, 2] += SPAN / 121
    ge = ((M["grain"][:, 0] > 0) & (pred["grain_area"] < R.threshold)).nonzero().view(-1)
    pred["grain_event"] = ge[torch.argsort(pred["grain_area"][ge])]
    return pred, gs


def refresh_edges(X, EI):
    EA = {}
    for et, index in EI.items():                                    # test.py:562-575
        rel = X[et[0]][index[0], :2] - X[et[-1]][index[-1], :2]
        rel = -1 * (rel > 0.5) + 1 * (rel < -0.5) + rel
        EA[et] = torch.sqrt(rel[:, 0] ** 2 + rel[:, 1] ** 2).view(-1, 1)
    return EA


def record(out, name, Cm, X, EI, EA, pred, M, gs):
    """Save the inputs, call the reference update, save the outputs."""
    i = name + "__in_"
    out[i + "x_grain"], out[i + "x_joint"] = X["grain"].numpy().copy(), X["joint"].numpy().copy()
    for et in (GJ, JG, JJ):
        out[i + "ei_" + "__".join(et)] = EI[et].numpy().copy()
    out[i + "y_joint"], out[i + "y_grain"] = pred["joint"].numpy().copy(), pred["grain"].numpy().copy()
    out[i + "edge_event"] = pred["edge_event"].numpy().copy()
    out[i + "grain_event"] = pred["grain_event"].numpy().copy()
    out[i + "mask_grain"], out[i + "mask_joint"] = M["grain"].numpy().copy(), M["joint"].numpy().copy()
    X2, EI2, pairs = Cm.update(X, EI, EA, pred, M, gs, 0.0)
    o = name + "__out_"
    out[o + "x_grain"], out[o + "x_joint"] = X2["grain"].numpy().copy(), X2["joint"].numpy().copy()
    for et in (GJ, JG, JJ):
        out[o + "ei_" + "__".join(et)] = EI2[et].numpy().copy()
    out[o + "y_joint"] = pred["joint"].numpy().copy()
    out[o + "grain_event"] = pred["grain_event"].numpy().copy()
    out[o + "switching_list"] = np.asarray(pairs.numpy()).reshape(-1, 2).copy()
    out[o + "mask_grain"], out[o + "mask_joint"] = M["grain"].numpy().copy(), M["joint"].numpy().copy()
    print(f"{name}: grain events {out[i + 'grain_event'].tolist()[:8]}{'...' if len(out[i + 'grain_event']) > 8 else ''} "
          f"-> {len(out[o + 'grain_event'])} eliminated, {len(out[o + 'switching_list'])} switches, "
          f"E_jj {out[i + 'ei_' + '__'.join(JJ)].shape[1]} -> {out[o + 'ei_' + '__'.join(JJ)].shape[1]}")
    return X2, EI2


@torch.no_grad()
def main():
    import __main__
    import dill
    import graph_trajectory as gt
    __main__.graph_trajectory, __main__.graph = gt.graph_trajectory, gt.graph
    g40, x, ei, ea = mg.load_graph(os.path.join(mg.REF, "graphs/40_40/seed10020_G1.904_R0.558_span6.pkl"))
    hp = mg.make_hyper(g40)
    R, Cm = mg.build_reference(hp, x, ei, ea, 10020, 1.0)
    R.threshold, Cm.threshold = 1e-4, 0.6                            # test.py:187-188
    mask0 = {k: torch.from_numpy(np.asarray(v).astype(np.int64)) for k, v in g40.mask.items()}
    mask0["joint"] = 1 + 0 * mask0["joint"]                          # test.py:258
    out = {}

    def fresh():
        X, EI, EA = mg.tt(x), mg.tt(ei), mg.tt(ea)
        M = {k: v.clone() for k, v in mask0.items()}
        pred, gs = step_to_update_point(R, Cm, X, EI, EA, M)
        assert len(pred["grain_event"]) == 0                        # step 1 of this rollout is quiet
        pred["edge_event"] = torch.full_like(pred["edge_event"], -8.0)
        return X, EI, EA, M, pred, gs

    def directed_edges(EI, k):
        """k junction-junction edges with src < dst whose end points are pairwise disjoint."""
        src, dst = EI[JJ][0].tolist(), EI[JJ][1].tolist()
        used, picks = set(), []
        for e in range(len(src)):
            if src[e] < dst[e] and not ({src[e], dst[e]} & used):
                picks.append(e)
                used |= {src[e], dst[e]}
                if len(picks) == k:
                    break
        return picks

    # ---- hand-made events on the initial topology ----
    X, EI, EA, M, pred, gs = fresh()
    pred["grain_event"] = torch.tensor([44])
    record(out, "elim1", Cm, X, EI, EA, pred, M, gs)

    X, EI, EA, M, pred, gs = fresh()
    e = directed_edges(EI, 1)
    pred["edge_event"][e[0]] = 3.0
    record(out, "switch1", Cm, X, EI, EA, pred, M, gs)

    X, EI, EA, M, pred, gs = fresh()
    src, dst = EI[JJ][0].tolist(), EI[JJ][1].tolist()
    e0 = directed_edges(EI, 1)[0]
    # a second edge that shares a junction with the first, and a far one
    e1 = next(k for k in range(len(src)) if src[k] < dst[k] and k != e0 and ({src[k], dst[k]} & {src[e0], dst[e0]}))
    e2 = [k for k in directed_edges(EI, 40) if not ({src[k], dst[k]} & {src[e0], dst[e0], src[e1], dst[e1]})][-1]
    pred["edge_event"][e0], pred["edge_event"][e1], pred["edge_event"][e2] = 1.0, 4.0, 2.0
    record(out, "switch3", Cm, X, EI, EA, pred, M, gs)

    X, EI, EA, M, pred, gs = fresh()
    pred["grain_event"] = torch.tensor([60, 23])
    picks = directed_edges(EI, 30)
    pred["edge_event"][picks[10]], pred["edge_event"][picks[25]] = 2.5, 0.9
    record(out, "mixed", Cm, X, EI, EA, pred, M, gs)

    # ---- free-running seeded rollout: the mass eliminations of steps 3 and 4 ----
    with gzip.open(os.path.join(mg.REF, "graphs/40_40/traj10020.pkl.gz"), "rb") as f:
        traj = dill.load(f)
    traj.raise_err = False
    traj.extraV_traj, traj.area_traj = [], traj.area_traj[:1]
    X, EI, EA = mg.tt(x), mg.tt(ei), mg.tt(ea)
    M = {k: v.clone() for k, v in mask0.items()}
    traj.GNN_update(0, {k: v.clone() for k, v in X.items()}, M, True, EI, False)
    for step in range(1, 5):
        pred, gs = step_to_update_point(R, Cm, X, EI, EA, M)
        if step >= 3:
            X, EI = record(out, f"mass{step}", Cm, X, EI, EA, pred, M, gs)
            pairs = out[f"mass{step}__out_switching_list"]
        else:
            X, EI, pairs = Cm.update(X, EI, EA, pred, M, gs, 0.0)
        topo = len(pred["grain_event"]) > 0 or len(pairs) > 0
        traj.GNN_update(step * SPAN, {k: v.clone() for k, v in X.items()}, M, topo, EI, False)
        for grain, coor in traj.region_center.items():               # test.py:556-559
            X["grain"][grain - 1, :2] = torch.FloatTensor(coor)
        EA = refresh_edges(X, EI)
        if step >= 3:
            out[f"mass{step}__next_x_grain"] = X["grain"].numpy().copy()   # after the centre refresh
    np.savez_compressed(os.path.join(HERE, "golden_cfg1_events.npz"), **out)
    print("wrote golden_cfg1_events.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
