#!/usr/bin/env python3
"""Development probe: where the HOST time of an eventful step goes -- cProfile over a loop of eventful step_events calls
(cfg3, ~3 grains per step)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from graingraphnn_amd.rollout import GrainRollout  # noqa: E402

dev = torch.device("cuda", 0)
R, Cm, X, EI, EA, inputs = bench.build(dev, seed=0)
ro = GrainRollout(R, Cm, X, EI, EA, bench.SPAN, use_graph=True, refresh_centres=True, domain_factor=inputs[3],
                  domain_offset=None if inputs[4] is None else torch.from_numpy(inputs[4]), joint_launches=False, concurrent=True)
mask = {"grain": np.ones((X["grain"].size(0), 1), np.int64), "joint": np.ones((X["joint"].size(0), 1), np.int64)}
ro.enable_events(mask, area_threshold=-1.0, edge_threshold=0.999999)
for _ in range(3):
    ro.step_events()


def threshold(k):
    area = ro.pred["grain_area"].cpu().numpy()
    live = ro.mask["grain"][:, 0] > 0
    ro.max_grain_events = k
    return float(np.nextafter(np.float32(np.sort(area[live])[k - 1]), np.float32(1)))


def loop(n):
    for _ in range(n):
        ro.area_threshold = threshold(3)
        ro.step_events()


loop(4)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
loop(40)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
