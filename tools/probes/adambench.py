#!/usr/bin/env python3
"""Development aid: training.FusedAdam.step() on the regressor's 284 parameter tensors, 30 steps (run under rocprofv3
--kernel-trace --stats to read ggnn::adam_kernel's duration)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from graingraphnn_amd import synthetic, training
from graingraphnn_amd.models import GrainNN_regressor
R = GrainNN_regressor(synthetic.default_hyper("cuda")).cuda()
opt = training.FusedAdam(R.parameters(), lr=1e-3)
for p in R.parameters():
    p.grad = torch.randn_like(p) * 1e-3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    opt.step()
torch.cuda.synchronize()
ts = []
for _ in range(30):
    torch.cuda._sleep(int(2e6))
    e0.record(); opt.step(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
b = opt._built
print("tensors", len(b["ps"]), "chunks", sum(l[2].numel() for l in b["launches"]), "elements", sum(p.numel() for p, _ in b["ps"]),
      "median us (incl. host)", ts[len(ts) // 2], "min", ts[0])
