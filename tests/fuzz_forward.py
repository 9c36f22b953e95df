#!/usr/bin/env python3
"""One-off GPU fuzz (not collected by pytest): HIP forward of both models against the CPU oracle on
random Voronoi grain structures (grain degrees 3..12), random sizes / foldings / weight seeds and
scales.  Tolerance: the north_star bar, max|a-b| <= 1e-4 * max|ref| per output tensor.
    python tests/fuzz_forward.py [--n 30] [--seed 0]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import oracle_models, product_models, rel_err, tt  # noqa: E402
from graingraphnn_amd import synthetic  # noqa: E402


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    worst = 0.0
    for it in range(args.n):
        n_g = int(rs.choice([12, 40, 150, 400, 1500]))
        fold = int(rs.choice([1, 1, 2, 3])) if n_g >= 150 else 1
        noise = None if rs.rand() < 0.5 else float(rs.uniform(0.05, 0.3))
        wseed, scale = int(rs.randint(1, 10 ** 6)), float(rs.choice([0.3, 1.0, 2.0]))
        x, ei, ea = synthetic.voronoi(n_g, seed=int(rs.randint(1, 10 ** 6)), fold=fold, lattice_noise=noise)
        R, Cm = product_models(wseed, scale, "cuda")
        oR, oC = oracle_models(wseed, scale)
        ya, ca = R(tt(x, "cuda"), tt(ei, "cuda"), tt(ea, "cuda")), Cm(tt(x, "cuda"), tt(ei, "cuda"), tt(ea, "cuda"))
        yb, cb = oR(tt(x), tt(ei), tt(ea)), oC(tt(x), tt(ei), tt(ea))
        errs = {k: rel_err(ya[k], yb[k]) for k in ("joint", "grain", "grain_area")}
        errs.update({k: rel_err(ca[k], cb[k]) for k in ("edge_event", "edge")})
        w = max(errs.values())
        worst = max(worst, w)
        deg = np.bincount(ei[("joint", "pull", "grain")][1])
        print(f"{it:3d} grains {x['grain'].shape[0]:5d} fold {fold} degree {deg.min()}..{deg.max()} weights x{scale}: "
              f"worst {w:.2e}" + (f"  {({k: float(f'{v:.1e}') for k, v in errs.items()})}" if w > 1e-5 else ""), flush=True)
        assert w <= 1e-4, errs
    print(f"{args.n} random structures: worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
