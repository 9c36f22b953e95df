#!/usr/bin/env python3
"""Development aid: time one training step (forward, loss, backward, Adam; train.py:158-166) of
the regressor on the HIP training path, and the same step of the CPU oracle (which is why this
script lives under tests/: only tests, smoke() and bench.py's cpu_baseline may touch oracle/).

    python tests/bench_train_step.py [--batch 4] [--steps 20] [--bf16] [--cfg3] [--no-cpu]

Default workload: `--batch` copies of the 40 um fixture as one disjoint-union graph (what PyG's
DataLoader collation does for train.py's batch_size 4); --cfg3: the 10k-grain honeycomb.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from graingraphnn_amd import synthetic, training  # noqa: E402
from graingraphnn_amd.models import GrainNN_regressor  # noqa: E402
from graingraphnn_amd.seeding import load_seeded  # noqa: E402


def run(model, X, EI, EA, y, mask, steps, sync, autocast=False, graph=False, fused=False):
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, capturable=graph, **({"fused": True} if fused else {}))
    losses = []

    def one():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            loss = training.regressor_loss(y, model(X, EI, EA), mask)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    if not graph:
        for _ in range(3):
            one()
        sync()
    else:
        # the whole step (forward, loss, backward, Adam) replayed from one hipGraph: the eager step is
        # bound by the host (~800 launches), the kernels themselves take about half of its time.
        # Warm-up and capture on the same side stream (PyTorch's whole-network capture recipe).
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                static_loss = one()
        torch.cuda.current_stream().wait_stream(s)

        def one():
            g.replay()
            return static_loss.clone()
        sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(one())
    sync()
    dt = (time.perf_counter() - t0) / steps
    return dt, [float(v.detach()) for v in losses]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--cfg3", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the whole training step from one hipGraph")
    ap.add_argument("--fused", action="store_true", help="torch.optim.Adam(fused=True): one optimizer launch instead of ~40")
    ap.add_argument("--json", action="store_true", help="print one JSON line instead of text (bench.py's train_step record)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads of the CPU oracle leg (16 is its best on the 2 x 64-core GPU box)")
    args = ap.parse_args()
    if args.cfg3:
        x, ei, ea = synthetic.honeycomb(100, 10, 0)
        name = "cfg3 honeycomb (10000 grains)"
    else:
        x0, ei0, ea0 = synthetic.load_fixture(os.path.join(ROOT, "tests", "golden", "graph_40.npz"))
        x, ei, ea, _ = synthetic.disjoint_union([(synthetic.perturbed_copy(x0, 1e-3, 1000 + t), ei0, ea0)
                                                 for t in range(args.batch)])
        name = f"{args.batch} x 40 um fixture (118 grains each)"
    rs = np.random.RandomState(3)
    y = {nt: rs.uniform(-1, 1, (x[nt].shape[0], 2)).astype(np.float32) for nt in x}
    mask = {nt: np.ones((x[nt].shape[0], 1), np.float32) for nt in x}
    dev = torch.device("cuda", 0)
    R = load_seeded(GrainNN_regressor(synthetic.default_hyper(dev)), 0, 1.0).to(dev)
    X, EI, EA = synthetic.to_torch(x, ei, ea, dev)
    Y = {k: torch.from_numpy(v).to(dev) for k, v in y.items()}
    M = {k: torch.from_numpy(v).to(dev) for k, v in mask.items()}
    dt, losses = run(R, X, EI, EA, Y, M, args.steps, torch.cuda.synchronize, args.bf16, args.graph, args.fused)
    if args.json:
        import json
        print(json.dumps({"workload": name, "ms_per_step": round(dt * 1e3, 3), "steps": args.steps,
                          "launch": "hipGraph replay (training.GraphedTrainStep recipe)" if args.graph else "eager",
                          "what": "forward, loss (train.py:31-37), backward, Adam step of the regressor; " +
                                  ("torch.autocast(bfloat16): the decoder projection in bf16 MFMA arithmetic "
                                   "(GGNN_PRECISION_BF16), sweeps / softmax / LSTM / gradients fp32" if args.bf16 else "fp32"),
                          "optimizer": "torch.optim.Adam(fused=True)" if args.fused else "torch.optim.Adam (default: foreach)",
                          "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)]}))
        return
    print(f"{name}: HIP training path{' (bf16 autocast GEMMs)' if args.bf16 else ''}"
          f"{' (hipGraph replay)' if args.graph else ''}{' (fused Adam)' if args.fused else ''}: {dt * 1e3:.2f} ms/step, "
          f"loss {losses[0]:.4f} -> {losses[-1]:.4f}")
    if not args.no_cpu:
        from oracle import grainnn_oracle as oracle
        torch.set_num_threads(args.cpu_threads)
        oR = load_seeded(oracle.GrainNN_regressor(synthetic.default_hyper("cpu")), 0, 1.0)
        Xc, EIc, EAc = synthetic.to_torch(x, ei, ea, "cpu")
        Yc = {k: torch.from_numpy(v) for k, v in y.items()}
        Mc = {k: torch.from_numpy(v) for k, v in mask.items()}
        dtc, lc = run(oR, Xc, EIc, EAc, Yc, Mc, max(2, args.steps // 10), lambda: None)
        print(f"{name}: CPU oracle, {torch.get_num_threads()} threads: {dtc * 1e3:.1f} ms/step, "
              f"loss {lc[0]:.4f} -> {lc[-1]:.4f}   (HIP {dtc / dt:.0f}x)")


if __name__ == "__main__":
    main()
