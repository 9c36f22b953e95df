#!/usr/bin/env python3
"""Development probe: two ranks (gloo, both on cuda:0), DistributedDataParallel against the mean of the ranks' plain gradients,
many iterations, every deviation printed with its pattern.
    python tools/probes/ddp2.py [ITERS]          # starts the two ranks
    python tools/probes/ddp2.py worker PORT RANK ITERS"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def worker(port, rank, iters):
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from helpers import load_graph, product_models, tt
    from graingraphnn_amd import synthetic, training
    from graingraphnn_amd.dist import gather_states
    from test_training import _targets
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    x0, ei0, ea0 = load_graph("40")
    x, ei, ea, _ = synthetic.disjoint_union([(synthetic.perturbed_copy(x0, 1e-3, 2000 + 4 * rank + t), ei0, ea0) for t in range(4)])
    y_np, m_np = _targets(x, ei)
    y, mask = tt(y_np, dev), tt(m_np, dev)
    X, EI, EA = tt(x, dev), tt(ei, dev), tt(ea, dev)
    Rt, _ = product_models(4, 1.0, dev)
    Rt.train()
    training.regressor_loss(y, Rt(X, EI, EA), mask).backward()
    local = {n: p.grad.clone() for n, p in Rt.named_parameters()}
    Rt.zero_grad()
    both = gather_states(local, 2)
    mean = {n: (both[0][n] + both[1][n]) / 2 for n in local}
    model = DistributedDataParallel(Rt, device_ids=[0])
    for it in range(iters):
        model.zero_grad()
        training.regressor_loss(y, model(X, EI, EA), mask).backward()
        torch.cuda.synchronize()
        for n, p in Rt.named_parameters():
            d = (p.grad - mean[n]).abs()
            if float(d.max()) > 1e-6 * max(float(mean[n].abs().max()), 1e-6):
                idx = torch.nonzero(d.reshape(-1) > 1e-7 * float(mean[n].abs().max())).reshape(-1)
                print(f"rank {rank} it {it}: {n} {tuple(p.shape)}: max {float(d.max()):.3e} (|mean| {float(mean[n].abs().max()):.3e}), "
                      f"{idx.numel()} entries off, first {idx[:8].tolist()}, last {idx[-3:].tolist()}; grad is view: {p.grad._base is not None}, "
                      f"offset {p.grad.storage_offset()}", flush=True)
        # the plain backward again: still the first one's bits?
        if it % 5 == 4:
            Rt.zero_grad()
            training.regressor_loss(y, Rt(X, EI, EA), mask).backward()
            for n, p in Rt.named_parameters():
                if not torch.equal(p.grad, local[n]):
                    print(f"rank {rank} it {it}: plain backward differs in {n}: {float((p.grad - local[n]).abs().max()):.3e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} done", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(port), str(r), str(iters)],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        for p in procs:
            out, _ = p.communicate(timeout=900)
            print("\n".join(l for l in out.splitlines() if l.startswith("rank") or "Error" in l))
