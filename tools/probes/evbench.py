"""Development probe (round 5): cost of a QUIET event-mode step at the 10k-grain graph (thresholds that never fire) through
step_events (one host synchronisation per step) and run_events (speculative, no stall), against the static rollout.
Not part of the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import json
import numpy as np
import torch
import bench
from graingraphnn_amd.rollout import GrainRollout

dev = torch.device("cuda", 0)
R, Cm, X, EI, EA, inputs = bench.build(dev, seed=0)
def make():
    Xc = {k: v.clone() for k, v in X.items()}
    ro = GrainRollout(R, Cm, Xc, EI, EA, bench.SPAN, use_graph=True, concurrent=True, joint_launches=False,
                      refresh_centres=True, domain_factor=inputs[3],
                      domain_offset=None if inputs[4] is None else torch.from_numpy(inputs[4]))
    return Xc, ro
def timed(fn, n):
    fn(48)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
n = 200
_, ro = make()
static = timed(ro.run, n)
out = {"static_us_per_step": round(static, 1)}
for name in ("step_events", "run_events"):
    Xc, ro = make()
    ro.enable_events({"grain": np.ones((Xc["grain"].size(0), 1)), "joint": np.ones((Xc["joint"].size(0), 1))}, -1.0, 0.999999)
    ro._logit_trigger = 1e30     # never fires: quiet steps only
    fn = (lambda k: [ro.step_events() for _ in range(k)]) if name == "step_events" else ro.run_events
    us = timed(fn, n)
    out[name + "_quiet_us_per_step"] = round(us, 1)
    out[name + "_over_static"] = round(us / static, 3)
print(json.dumps(out), flush=True)
