"""Fuzz the host-side topology update against the UNMODIFIED reference
`GrainNN_classifier.update` (models.py:612-842) on random event scenarios.

Runs only in the build container (needs /root/reference).  Nothing is written unless
`--save N` is given, which stores the first N scenarios in `golden_cfg1_events_fuzz.npz` (same
key scheme as golden_cfg1_events.npz) for tests/test_topology.py.
    python tests/golden/fuzz_events.py [--n 200] [--seed 0] [--chain 3] [--save 12]

A scenario starts from the 40 um fixture (or, with --chain k, from the topology left by the
previous k-1 scenarios of the chain), draws 0-3 grains to eliminate and 0-4 junction-junction
edges with random super-threshold probabilities, and calls both implementations.  Scenarios on
which the reference itself raises (its asserts at models.py:673, 681, 803, 869) must raise here
too; all others must agree bit for bit, including the column order of every edge list.
"""
import argparse
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (sets up sys.path for the reference + stubs)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from graingraphnn_amd.topology import TopologyError, update_topology  # noqa: E402

GJ, JG, JJ = mg.GJ, mg.JG, mg.JJ


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--chain", type=int, default=3, help="scenarios applied on top of each other before a reset")
    ap.add_argument("--save", type=int, default=0)
    args = ap.parse_args()
    g40, x, ei, ea = mg.load_graph(os.path.join(mg.REF, "graphs/40_40/seed10020_G1.904_R0.558_span6.pkl"))
    hp = mg.make_hyper(g40)
    R, Cm = mg.build_reference(hp, x, ei, ea, 10020, 1.0)
    Cm.threshold = 0.6
    mask0 = {k: torch.from_numpy(np.asarray(v).astype(np.int64)) for k, v in g40.mask.items()}
    mask0["joint"] = 1 + 0 * mask0["joint"]
    rs = np.random.RandomState(args.seed)
    saved, n_ok, n_both_raise, n_events, n_switch = {}, 0, 0, 0, 0
    X = EI = M = None
    for it in range(args.n):
        if it % args.chain == 0 or X is None:
            X, EI = mg.tt(x), mg.tt(ei)
            M = {k: v.clone() for k, v in mask0.items()}
        n_j, E = X["joint"].shape[0], EI[JJ].shape[1]
        live_g = np.flatnonzero(M["grain"][:, 0].numpy() > 0)
        pred = {"joint": torch.from_numpy(rs.uniform(-1, 1, (n_j, 2)).astype(np.float32)),
                "grain": torch.from_numpy(rs.uniform(-1, 1, (X["grain"].shape[0], 2)).astype(np.float32)),
                "edge_event": torch.full((E,), -8.0)}
        for e in rs.choice(E, size=rs.randint(0, 5), replace=False):
            pred["edge_event"][e] = float(rs.uniform(0.5, 4.0))
        pred["grain_event"] = torch.from_numpy(rs.choice(live_g, size=rs.randint(0, 4), replace=False).astype(np.int64))
        gs = {"domain_offset": 0, "domain_factor": 1.0,  # + what Rmodel.update leaves behind (models.py:476-478)
              "active_grains": (pred["grain"][:, 0] > -10).nonzero().view(-1),
              "active_joints": (pred["joint"][:, 0] > -10).nonzero().view(-1)}
        inp = {"x_joint": X["joint"].numpy().copy(), "x_grain": X["grain"].numpy().copy(),
               "y_joint": pred["joint"].numpy().copy(), "y_grain": pred["grain"].numpy().copy(),
               "edge_event": pred["edge_event"].numpy().copy(), "grain_event": pred["grain_event"].numpy().copy(),
               "mask_grain": M["grain"].numpy().copy(), "mask_joint": M["joint"].numpy().copy()}
        for et in (GJ, JG, JJ):
            inp["ei_" + "__".join(et)] = EI[et].numpy().copy()
        # ---- mine (on copies) ----
        xj, yj = inp["x_joint"].copy(), inp["y_joint"].copy()
        mgr, mjo = inp["mask_grain"].copy(), inp["mask_joint"].copy()
        prob = torch.sigmoid(pred["edge_event"]).numpy()
        try:
            mine = update_topology(xj, inp["ei_" + "__".join(JJ)], inp["ei_" + "__".join(JG)], yj, inp["y_grain"],
                                   prob, inp["grain_event"], mgr, mjo, 0.6)
            mine_err = None
        except (TopologyError, IndexError, ValueError, KeyError, AssertionError) as exc:
            mine, mine_err = None, exc
        # ---- the reference (mutates X, EI, M, pred in place) ----
        EA = {et: torch.zeros(EI[et].shape[1], 1) for et in (GJ, JG, JJ)}
        try:
            X2, EI2, pairs = Cm.update(X, EI, EA, pred, M, gs, 0.0)
            ref_err = None
        except Exception as exc:  # noqa: BLE001  (the reference asserts / KeyErrors on degenerate inputs)
            ref_err = exc
        if ref_err is not None or mine_err is not None:
            assert ref_err is not None and mine_err is not None, (it, repr(ref_err), repr(mine_err))
            n_both_raise += 1
            X = None  # the reference may have left its inputs half-updated: restart the chain
            continue
        pp, pq, qp, sw, events = mine
        assert np.array_equal(pp, EI2[JJ].numpy()), (it, "joint-joint")
        assert np.array_equal(pq, EI2[JG].numpy()), (it, "joint-grain")
        assert np.array_equal(qp, EI2[GJ].numpy()), (it, "grain-joint")
        assert np.array_equal(sw, np.asarray(pairs.numpy()).reshape(-1, 2)), (it, "switching list")
        assert np.array_equal(events, pred["grain_event"].numpy()), (it, "grain events")
        assert np.array_equal(mgr, M["grain"].numpy()) and np.array_equal(mjo, M["joint"].numpy()), (it, "masks")
        assert np.array_equal(xj, X2["joint"].numpy()), (it, "x_joint")
        assert np.array_equal(yj, pred["joint"].numpy()), (it, "y_joint")
        n_ok += 1
        n_events += len(events)
        n_switch += len(sw)
        if len(saved) // 21 < args.save and (len(events) or len(sw)):
            name = f"fuzz{len(saved) // 21:02d}"
            for k, v in inp.items():
                saved[f"{name}__in_{k}"] = v
            saved[f"{name}__out_x_joint"], saved[f"{name}__out_x_grain"] = xj, X2["grain"].numpy().copy()
            saved[f"{name}__out_y_joint"] = yj
            saved[f"{name}__out_ei_" + "__".join(JJ)] = pp
            saved[f"{name}__out_ei_" + "__".join(JG)] = pq
            saved[f"{name}__out_ei_" + "__".join(GJ)] = qp
            saved[f"{name}__out_switching_list"], saved[f"{name}__out_grain_event"] = sw, events
            saved[f"{name}__out_mask_grain"], saved[f"{name}__out_mask_joint"] = mgr, mjo
            assert len(saved) % 21 == 0, len(saved)
        X, EI = X2, EI2
    print(f"{args.n} scenarios: {n_ok} agree bit for bit ({n_events} grains eliminated, {n_switch} switches), "
          f"{n_both_raise} rejected by both")
    if args.save:
        np.savez_compressed(os.path.join(HERE, "golden_cfg1_events_fuzz.npz"), **saved)
        print("wrote golden_cfg1_events_fuzz.npz:", len(saved) // 21, "scenarios")


if __name__ == "__main__":
    main()
