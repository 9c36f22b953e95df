"""Development probe (round 4; see profiles/r4_dec_cell_ablations.txt / DESIGN.md section 7).  Not part of the product."""
import os
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import product_models, tt
from graingraphnn_amd import GrainRollout, synthetic
x, ei, ea, off = synthetic.honeycomb(100, 10, 0, return_offset=True)
R, Cm = product_models(0, 0.3, "cuda")
X, EI, EA = tt(x, "cuda"), tt(ei, "cuda"), tt(ea, "cuda")
ro = GrainRollout(R, Cm, X, EI, EA, 6, use_graph=True, refresh_centres=True, domain_factor=10.0, domain_offset=torch.from_numpy(off))
ro.run(40)
torch.cuda.synchronize()
for trial in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ro.refresh_weights()
    t1 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ro.run(20)
    e1.record()
    t2 = time.perf_counter()
    st = ro.state()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"refresh {1e3*(t1-t0):.3f} ms | run(20) host {1e3*(t2-t1):.3f} ms | state() {1e3*(t3-t2):.3f} | sync {1e3*(t4-t3):.3f} | total {1e3*(t4-t0):.3f} ms = {20/(t4-t0):.0f} steps/s | GPU e0..e1 {e0.elapsed_time(e1):.3f} ms")
for n in (20, 100, 500):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ro.run(n); ro.state(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(n, f"{n/dt:.0f} steps/s")
