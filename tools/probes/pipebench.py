"""Development probe (round 5): the cfg3 rollout through hipGraph replays at several unroll factors, pipelined plans A/B
(GGNN_PIPE=coupled | default).  Not part of the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from graingraphnn_amd.rollout import GrainRollout

dev = torch.device("cuda", 0)
R, Cm, X, EI, EA, inputs = bench.build(dev, seed=0)
unrolls = [int(a) for a in sys.argv[1:]] or [4, 10, 20]
for u in unrolls:
    Xc = {k: v.clone() for k, v in X.items()}
    ro = GrainRollout(R, Cm, Xc, EI, EA, bench.SPAN, use_graph=True, concurrent=True, joint_launches=False,
                      refresh_centres=True, domain_factor=inputs[3],
                      domain_offset=None if inputs[4] is None else torch.from_numpy(inputs[4]))
    ro.RUN_UNROLL = u
    try:
        ro.run(u * 3)
        torch.cuda.synchronize()
        n = u * max(1, 200 // u)
        t0 = time.perf_counter()
        ro.run(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"unroll {u:3d}: {n / dt:8.1f} steps/s  {dt / n * 1e6:7.1f} us/step  (pipe={os.environ.get('GGNN_PIPE', 'tail under the classifier decoder')})", flush=True)
    except Exception as exc:
        print(f"unroll {u:3d}: FAILED {type(exc).__name__}: {str(exc).splitlines()[0]}", flush=True)
        break
