#!/usr/bin/env python3
"""Development probe: run_events() with an event every PERIOD steps (cfg3, ~3 grains per event): wall time per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from graingraphnn_amd.rollout import GrainRollout  # noqa: E402

period, cycles = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
R, Cm, X, EI, EA, inputs = bench.build(dev, seed=0)
ro = GrainRollout(R, Cm, X, EI, EA, bench.SPAN, use_graph=True, refresh_centres=True, domain_factor=inputs[3],
                  domain_offset=None if inputs[4] is None else torch.from_numpy(inputs[4]), joint_launches=False, concurrent=True)
mask = {"grain": np.ones((X["grain"].size(0), 1), np.int64), "joint": np.ones((X["joint"].size(0), 1), np.int64)}
ro.enable_events(mask, area_threshold=-1.0, edge_threshold=0.999999)


def cycle():
    ro.area_threshold = -1.0
    ro.run_events(period - 1)
    area = ro.pred["grain_area"].cpu().numpy()
    live = ro.mask["grain"][:, 0] > 0
    ro.max_grain_events = 3
    ro.area_threshold = float(np.nextafter(np.float32(np.sort(area[live])[2]), np.float32(1)))
    ev, _ = ro.run_events(1)
    return len(ev[0])


for _ in range(3):
    cycle()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = [cycle() for _ in range(cycles)]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"in place: {getattr(ro, '_cap', None) is not None}; an event every {period} steps, {cycles} cycles, grains per event {sorted(set(n))}: "
      f"{dt / (period * cycles) * 1e3:.3f} ms per step ({period * cycles / dt:.0f} steps/s)")
