#!/usr/bin/env python3
"""One-off GPU check (not collected by pytest): the hot path at TEN TIMES the benchmark's size -- a periodic
honeycomb of 99 856 grains / 199 712 junctions / 599 136 edges per edge type (1.6 GB of projections per model) --
one rollout step (regressor + classifier + update + edge refresh) against the CPU oracle, bitwise determinism, then
the replayed rollout's rate.  Looks for 32-bit index arithmetic, grid limits and unit-table sizes that the 10k-grain
benchmark cannot reach.
    python tests/scale_check.py [--n 316] [--steps 40]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import torch  # noqa: E402

from helpers import EDGE_TYPES, assert_close, oracle_models, product_models, tt  # noqa: E402
from graingraphnn_amd import GrainRollout, synthetic  # noqa: E402
from oracle import grainnn_oracle as oracle  # noqa: E402


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=316)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--no-oracle", action="store_true", help="sizes the CPU oracle needs minutes for: determinism and rate only")
    args = ap.parse_args()
    dev = "cuda"
    x, ei, ea = synthetic.honeycomb(args.n, 32, 0)
    print(f"{x['grain'].shape[0]} grains, {x['joint'].shape[0]} junctions, "
          f"{', '.join(str(ei[et].shape[1]) for et in EDGE_TYPES)} edges", flush=True)
    R, Cm = product_models(0, 0.3, dev)
    X, EI, EA = tt(x, dev), tt(ei, dev), tt(ea, dev)
    ro = GrainRollout(R, Cm, X, EI, EA, 6)
    pred = {k: v.clone() for k, v in ro.step().items()}
    torch.cuda.synchronize()
    t_cpu = float("nan")
    if not args.no_oracle:
        torch.set_num_threads(args.threads)
        oR, oC = oracle_models(0, 0.3)
        oX, oEI, oEA = tt(x), tt(ei), tt(ea)
        t0 = time.perf_counter()
        opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6)
        t_cpu = time.perf_counter() - t0
        for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
            assert_close(pred[k], opred[k], f"{k}")
        for nt in x:
            assert_close(X[nt], oX[nt], f"x {nt} after the step")
        for et in EDGE_TYPES:
            assert_close(ro.edge_attr_dict()[et], oEA[et], f"edge_attr {et} after the step")
        print(f"one step matches the oracle (1e-4 max-norm and element-wise bars); oracle step {t_cpu:.1f} s at "
              f"{args.threads} threads", flush=True)
    X2 = tt(x, dev)
    ro2 = GrainRollout(R, Cm, X2, tt(ei, dev), tt(ea, dev), 6)
    pred2 = ro2.step()
    assert all(torch.equal(pred[k], pred2[k]) for k in pred), "two identical runs differ"
    ro.run(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ro.run(args.steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert all(bool(torch.isfinite(v).all()) for v in ro.x.values())
    print(f"replayed rollout: {dt * 1e3:.3f} ms per step = {1 / dt:.1f} steps/s ({x['grain'].shape[0] / dt / 1e6:.1f} M grain-steps/s; "
          f"the 10k-grain benchmark: ~20 M)" + ("" if args.no_oracle else f", {t_cpu / dt:.0f}x the oracle"), flush=True)


if __name__ == "__main__":
    main()
