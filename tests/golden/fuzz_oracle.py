"""Fuzz the CPU oracle against the UNMODIFIED reference models (forward of both models and two
rollout updates) on random Voronoi grain structures and random weights -- the oracle's pin beyond
the three committed golden configurations.  Build container only (needs /root/reference); writes
nothing.
    python tests/golden/fuzz_oracle.py [--n 20] [--seed 0]
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

from graingraphnn_amd import synthetic  # noqa: E402


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    g40, x40, ei40, ea40 = mg.load_graph(os.path.join(mg.REF, "graphs/40_40/seed10020_G1.904_R0.558_span6.pkl"))
    hp = mg.make_hyper(g40)
    rs = np.random.RandomState(args.seed)
    worst = 0.0
    for it in range(args.n):
        n_g = int(rs.choice([12, 40, 150, 400]))
        fold = int(rs.choice([1, 2, 3])) if n_g >= 150 else 1
        noise = None if rs.rand() < 0.5 else float(rs.uniform(0.05, 0.3))
        wseed, scale = int(rs.randint(1, 10 ** 6)), float(rs.choice([0.3, 1.0, 3.0]))
        x, ei, ea = synthetic.voronoi(n_g, seed=int(rs.randint(1, 10 ** 6)), fold=fold, lattice_noise=noise)
        R, Cm = mg.build_reference(hp, x, ei, ea, wseed, scale)
        oR, oC = mg.build_oracle(hp, wseed, scale)
        Xa, Xb = mg.tt(x), mg.tt(x)
        errs = {}
        for step in range(2):
            ya, ca = R(Xa, mg.tt(ei), mg.tt(ea)), Cm(Xa, mg.tt(ei), mg.tt(ea))
            yb, cb = oR(Xb, mg.tt(ei), mg.tt(ea)), oC(Xb, mg.tt(ei), mg.tt(ea))
            for k in ("joint", "grain", "grain_area"):
                errs[f"{k}{step}"] = mg.rel_err(yb[k].numpy(), ya[k].numpy())
            for k in ("edge_event", "edge"):
                errs[f"{k}{step}"] = mg.rel_err(cb[k].numpy(), ca[k].numpy())
            R.update(Xa, ya, {"domain_offset": 0, "domain_factor": 1.0})
            oR.update(Xb, yb, {"domain_offset": 0, "domain_factor": 1.0})
            for nt in ("joint", "grain"):
                errs[f"x_{nt}{step}"] = mg.rel_err(Xb[nt].numpy(), Xa[nt].numpy())
        w = max(errs.values())
        worst = max(worst, w)
        print(f"{it:3d} grains {x['grain'].shape[0]:4d} fold {fold} weights x{scale}: worst {w:.2e}", flush=True)
        assert w <= 2e-5, errs
    print(f"{args.n} random structures: oracle vs reference worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
