#!/usr/bin/env python3
"""Development aid: the rollout step at 4x and 10x the benchmark size (size robustness of the
32-bit row offsets, grids and L2 windows; r1: 2.44 ms and 5.75 ms per step, linear in size)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graingraphnn_amd import GrainRollout, synthetic
from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor
from graingraphnn_amd.seeding import load_seeded
dev = torch.device("cuda", 0)
for n, fold in ((200, 20), (316, 32)):
    n -= n % 2
    x, ei, ea, off = synthetic.honeycomb(n, fold, 0, return_offset=True)
    hp = synthetic.default_hyper(dev)
    R = GrainNN_regressor(hp); Cm = GrainNN_classifier(hp, R)
    load_seeded(R, 0, 0.3).eval(); load_seeded(Cm, 1, 0.3).eval()
    X, EI, EA = synthetic.to_torch(x, ei, ea, dev)
    ro = GrainRollout(R.to(dev), Cm.to(dev), X, EI, EA, 6, use_graph=True, refresh_centres=True,
                      domain_factor=float(fold), domain_offset=torch.from_numpy(off))
    for _ in range(5): ro.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ro.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    ok = all(bool(torch.isfinite(v).all()) for v in X.values())
    print(f"{n*n} grains / {2*n*n} junctions: {dt*1e3:.3f} ms/step, {n*n/dt/1e6:.1f} M grain-steps/s, finite={ok}")
