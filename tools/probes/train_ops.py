#!/usr/bin/env python3
"""Development probe: the recorded framework ops (anything that is not a view) of one eager training step (cfg3 regressor,
FusedAdam), in order, with operand shapes and the package frames they come from -- a TorchDispatchMode log; the hand-written
launches appear as the `ggnn_*` entry points the backend calls."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from graingraphnn_amd import synthetic, training  # noqa: E402
from graingraphnn_amd.backend import default_backend  # noqa: E402
from graingraphnn_amd.models import GrainNN_regressor  # noqa: E402
from graingraphnn_amd.seeding import load_seeded  # noqa: E402

dev = torch.device("cuda", 0)
x, ei, ea = synthetic.honeycomb(100, 10, 0)
rs = np.random.RandomState(3)
Y = {nt: torch.from_numpy(rs.uniform(-1, 1, (x[nt].shape[0], 2)).astype(np.float32)).to(dev) for nt in x}
M = {nt: torch.ones(x[nt].shape[0], 1, device=dev) for nt in x}
R = load_seeded(GrainNN_regressor(synthetic.default_hyper(dev)), 0, 1.0).to(dev)
X, EI, EA = synthetic.to_torch(x, ei, ea, dev)
R.train()
opt = training.FusedAdam(R.parameters(), lr=5e-3)
VIEWS = ("view", "empty", "detach", "slice", "select", "split", "transpose", "alias", "aten.t.", "as_strided", "unsqueeze",
         "expand", "unbind", "reshape", "squeeze", "permute", "_unsafe_view", "lift_fresh", "is_")


def one():
    loss = training.regressor_loss(Y, R(X, EI, EA), M)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in VIEWS):
            shp = [tuple(a.shape) for a in args if torch.is_tensor(a)]
            if args and isinstance(args[0], (list, tuple)):
                shp = ["%d tensors" % len(args[0])]
            fr = [f"{f.filename.split('/')[-1]}:{f.lineno}" for f in traceback.extract_stack() if "graingraphnn" in f.filename]
            print(f"  {name:34s} {str(shp)[:60]:60s} {' < '.join(reversed(fr[-3:]))}")
        return func(*args, **(kwargs or {}))


for _ in range(3):
    one()
torch.cuda.synchronize()
be = default_backend()
launch = be._launch


def logged(fn, name, *a):
    print("  " + name)
    return launch(fn, name, *a)


be._launch = logged
with Log():
    one()
torch.cuda.synchronize()
