#!/usr/bin/env python3
"""Development aid: the quiet speculative event loop (GrainRollout.run_events, thresholds that never fire) or the static
rollout at the 10k-grain graph, for rocprofv3 --kernel-trace + tools/timeline.py.   evquiet.py [static|events] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from graingraphnn_amd.rollout import GrainRollout

mode = sys.argv[1] if len(sys.argv) > 1 else "events"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
R, Cm, X, EI, EA, inputs = bench.build(dev, seed=0)
ro = GrainRollout(R, Cm, X, EI, EA, bench.SPAN, use_graph=True, refresh_centres=True, domain_factor=inputs[3],
                  domain_offset=torch.from_numpy(inputs[4]))
if mode == "events":
    mask = {"grain": np.ones((X["grain"].size(0), 1), np.int64), "joint": np.ones((X["joint"].size(0), 1), np.int64)}
    ro.enable_events(mask, area_threshold=-1.0, edge_threshold=0.999999)
    run = ro.run_events
else:
    run = ro.run
run(40)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / steps * 1e6:.1f} us per step", flush=True)
