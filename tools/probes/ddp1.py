#!/usr/bin/env python3
"""Development probe: DistributedDataParallel (world 1, gloo) against the plain backward on the HIP training path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch.nn.parallel import DistributedDataParallel  # noqa: E402

from helpers import load_graph, product_models, tt  # noqa: E402
from graingraphnn_amd import synthetic, training  # noqa: E402
from test_training import _targets  # noqa: E402

dev = torch.device("cuda", 0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
x0, ei0, ea0 = load_graph("40")
x, ei, ea, _ = synthetic.disjoint_union([(synthetic.perturbed_copy(x0, 1e-3, 2000 + t), ei0, ea0) for t in range(4)])
y_np, m_np = _targets(x, ei)
y, mask = tt(y_np, dev), tt(m_np, dev)
X, EI, EA = tt(x, dev), tt(ei, dev), tt(ea, dev)
Rt, _ = product_models(4, 1.0, dev)
Rt.train()
grads = []
for rep in range(3):
    Rt.zero_grad()
    training.regressor_loss(y, Rt(X, EI, EA), mask).backward()
    grads.append({n: p.grad.clone() for n, p in Rt.named_parameters()})
for n in grads[0]:
    for k in (1, 2):
        if not torch.equal(grads[0][n], grads[k][n]):
            print("plain backward not reproducible:", n, float((grads[0][n] - grads[k][n]).abs().max()))
local = grads[0]
Rt.zero_grad()
model = DistributedDataParallel(Rt, device_ids=[0])
for it in range(3):
    model.zero_grad()
    training.regressor_loss(y, model(X, EI, EA), mask).backward()
    torch.cuda.synchronize()
    bad = 0
    for n, p in Rt.named_parameters():
        d = float((p.grad - local[n]).abs().max())
        if d > 1e-6 * max(float(local[n].abs().max()), 1e-6):
            bad += 1
            if bad < 6:
                print(f"iteration {it}: {n}: |ddp - plain| = {d:.3e}, |plain| = {float(local[n].abs().max()):.3e}")
    print("iteration", it, "tensors off:", bad)
dist.destroy_process_group()
