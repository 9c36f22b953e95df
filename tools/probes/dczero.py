"""Development probe (round 5): the decoder cell on random and on all-zero operands (same structure, same instruction stream).
Zeros let the chip hold its clock (MI355X_MICROARCH.md, DVFS give-back): the ratio says how much of the launch time is the
clock the chip holds under this kernel's load.  Not part of the product."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from graingraphnn_amd import synthetic
from graingraphnn_amd.backend import default_backend
from test_hip_parity import _dec_cell_problem, _enc_cell_problem

be = default_backend()
rs = np.random.RandomState(0)
J = (20000, [(10000, 11, 60000), (20000, 8, 60000)])
_, hei, _ = synthetic.honeycomb(100, 10, 0)
GJ, JG, JJ = synthetic.EDGE_TYPES
def timeit(fn, probs):
    for _ in range(5):
        fn(probs)
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(probs); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return np.median(ts)
pj = _dec_cell_problem(be, rs, *J, F_dst=8, edges=[hei[GJ], hei[JJ]])
t_rand = timeit(be.decoder_cell_batch, [pj])
sweeps, xd, h_dst, c_in, wstream, tail, h_out, c_out = pj
for t in (xd, h_dst, c_in, wstream, tail):
    t.zero_()
for csr, einfo, h_src, v_src, v_off, ep in sweeps:
    for t in (einfo, h_src, v_src, ep):
        t.zero_()
t_zero = timeit(be.decoder_cell_batch, [pj])
print(f"decoder cell, 20 000 joints (157 workgroups): random operands {t_rand:.1f} us, all-zero operands {t_zero:.1f} us, ratio {t_rand / t_zero:.3f}", flush=True)
pe = _enc_cell_problem(be, rs, *J, F_dst=8, edges=[hei[GJ], hei[JJ]])[0]
t_rand = timeit(be.encoder_cell_batch, [pe])
for t in pe[1:4]:
    t.zero_()
for csr, einfo in pe[0]:
    einfo.zero_()
t_zero = timeit(be.encoder_cell_batch, [pe])
print(f"encoder cell, 20 000 joints: random operands {t_rand:.1f} us, all-zero operands {t_zero:.1f} us, ratio {t_rand / t_zero:.3f}", flush=True)
