#!/usr/bin/env python3
"""Development probe: what a plain, perfectly coalesced store stream reaches on this box for buffers of the decoder
projection's size (169 MB joint + 49 MB grain per model), cycling through enough buffers to defeat the 256 MB
Infinity Cache -- the ceiling to hold ggnn::project_x6_kernel's 188-204 MB of output per launch against."""
import torch
dev = "cuda"
for mb, nbuf in ((188, 1), (188, 8), (32, 1), (1024, 2)):
    bufs = [torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev) for _ in range(nbuf)]
    for b in bufs:
        b.zero_()
    torch.cuda.synchronize()
    n = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        bufs[i % nbuf].fill_(1.0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"fill {mb} MB x {nbuf} buffers: {us:.1f} us per fill = {mb * 1.048576 / us:.2f} TB/s")
    src = torch.empty_like(bufs[0])
    e0.record()
    for i in range(n):
        bufs[i % nbuf].copy_(src)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"copy {mb} MB x {nbuf} buffers: {us:.1f} us per copy = {2 * mb * 1.048576 / us:.2f} TB/s (read + write)")
