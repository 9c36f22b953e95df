#!/usr/bin/env python3
"""Per-step drift of the HIP rollout against the CPU oracle on the benchmark's own workload (BASELINE config 3:
10k-grain honeycomb folded x10, weights RandomState(0) x 0.3, R + C + update + grain-centre refresh + edge
refresh per step, hipGraph replay) over N steps (default 100; the suite asserts the first 10).

One CSV line per step: the max-norm error max|a-b| / max|b| of every prediction tensor and of the state
(x_joint, x_grain[:, 2:], the three edge_attr), so that one can see where the per-step 1e-4 contract -- a bound on
ONE step from IDENTICAL inputs -- stops being a statement about the trajectory: the random-weight rollout is a
chaotic map, both sides integrate their own rounding.

    python tools/cfg3_drift.py [--steps 100] [--out gpurun_out/cfg3_drift.csv] [--resync]

--resync: after every step copy the oracle's state into the HIP rollout (then every line is a ONE-step error from
identical inputs: the contract proper, at every point of the trajectory)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import EDGE_TYPES, oracle, oracle_models, product_models, rel_err, tt  # noqa: E402
from graingraphnn_amd import GrainRollout, synthetic  # noqa: E402


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cfg3_drift.csv"))
    ap.add_argument("--resync", action="store_true")
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    dev = "cuda"
    x, ei, ea, off = synthetic.honeycomb(100, 10, 0, return_offset=True)
    R, Cm = product_models(0, 0.3, dev)
    oR, oC = oracle_models(0, 0.3)
    X, EI, EA = tt(x, dev), tt(ei, dev), tt(ea, dev)
    oX, oEI, oEA = tt(x), tt(ei), tt(ea)
    ooff = torch.from_numpy(off)
    ro = GrainRollout(R, Cm, X, EI, EA, 6, use_graph=not a.resync, refresh_centres=True, domain_factor=10.0,
                      domain_offset=ooff)
    torch.set_num_threads(min(a.threads, torch.get_num_threads()))
    keys = ("joint", "grain", "grain_area", "edge_event", "edge")
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    t0 = time.time()
    with open(a.out, "w") as f:
        f.write("step," + ",".join(f"pred_{k}" for k in keys) + ",x_joint,x_grain_2on,"
                + ",".join("ea_" + "__".join(et) for et in EDGE_TYPES) + ",mode\n")
        for step in range(a.steps):
            pred = {k: v.clone() for k, v in ro.step().items()}
            opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6, centres=(10.0, ooff))
            errs = [rel_err(pred[k], opred[k]) for k in keys]
            errs += [rel_err(X["joint"], oX["joint"]), rel_err(X["grain"][:, 2:], oX["grain"][:, 2:])]
            hea = ro.edge_attr_dict()
            errs += [rel_err(hea[et].view(-1), oEA[et].view(-1)) for et in EDGE_TYPES]
            f.write(f"{step}," + ",".join(f"{e:.3e}" for e in errs) + f",{'resync' if a.resync else 'free'}\n")
            f.flush()
            if a.resync:
                for nt in X:
                    X[nt].copy_(oX[nt].to(dev))
                for et in EDGE_TYPES:
                    hea[et].view(-1).copy_(oEA[et].view(-1).to(dev))
            if step % 10 == 9:
                print(f"step {step + 1}: worst prediction error {max(errs[:5]):.2e}, state {max(errs[5:]):.2e} "
                      f"({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
