#!/usr/bin/env python3
"""Development aid: where the HOST time of an eager training step goes (cProfile over N steps at cfg3).
    python tools/train_hostprofile.py [steps]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from graingraphnn_amd import synthetic, training  # noqa: E402
from graingraphnn_amd.models import GrainNN_regressor  # noqa: E402
from graingraphnn_amd.seeding import load_seeded  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
x, ei, ea = synthetic.honeycomb(100, 10, 0)
rs = np.random.RandomState(3)
dev = torch.device("cuda", 0)
y = {nt: torch.from_numpy(rs.uniform(-1, 1, (x[nt].shape[0], 2)).astype(np.float32)).to(dev) for nt in x}
mask = {nt: torch.ones(x[nt].shape[0], 1, device=dev) for nt in x}
R = load_seeded(GrainNN_regressor(synthetic.default_hyper(dev)), 0, 1.0).to(dev)
X, EI, EA = synthetic.to_torch(x, ei, ea, dev)
R.train()
opt = torch.optim.Adam(R.parameters(), lr=5e-3)


def one():
    loss = training.regressor_loss(y, R(X, EI, EA), mask)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(5):
    one()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    one()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
