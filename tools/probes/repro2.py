#!/usr/bin/env python3
"""Development probe: is the training backward bit-reproducible while ANOTHER process uses the same GPU?
    python tools/probes/repro2.py            # starts two workers and prints what they found
    python tools/probes/repro2.py worker R   # one worker: 40 backward passes compared with the first"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank):
    import torch
    from helpers import load_graph, product_models, tt
    from graingraphnn_amd import synthetic, training
    from test_training import _targets
    dev = torch.device("cuda", 0)
    x0, ei0, ea0 = load_graph("40")
    x, ei, ea, _ = synthetic.disjoint_union([(synthetic.perturbed_copy(x0, 1e-3, 2000 + 4 * rank + t), ei0, ea0) for t in range(4)])
    y_np, m_np = _targets(x, ei)
    y, mask = tt(y_np, dev), tt(m_np, dev)
    X, EI, EA = tt(x, dev), tt(ei, dev), tt(ea, dev)
    Rt, _ = product_models(4, 1.0, dev)
    Rt.train()
    first, junk = None, []
    for it in range(int(os.environ.get("REPRO_PASSES", "40"))):
        Rt.zero_grad()
        training.regressor_loss(y, Rt(X, EI, EA), mask).backward()
        g = {n: p.grad.clone() for n, p in Rt.named_parameters()}
        if first is None:
            first = g
        else:
            for n in g:
                if not torch.equal(g[n], first[n]):
                    d = (g[n] - first[n]).abs()
                    print(f"rank {rank} pass {it}: {n}: max diff {float(d.max()):.3e} at {int(d.argmax())} of {tuple(g[n].shape)}, "
                          f"{int((d > 0).sum())} entries, |g| {float(first[n].abs().max()):.3e}", flush=True)
        # perturb the allocator the way a collective's staging buffers do
        junk.append(torch.randn(1 + (it * 7919) % 300000, device=dev))
        if len(junk) > 3:
            junk.pop(it % 3)
    print(f"rank {rank} done", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(int(sys.argv[2]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r)], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(n)]
        for p in procs:
            out, _ = p.communicate(timeout=600)
            print("\n".join(l for l in out.splitlines() if l.startswith("rank")))
