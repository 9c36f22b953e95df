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
from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor  # noqa: E402
from graingraphnn_amd.seeding import load_seeded  # noqa: E402


def run(model, X, EI, EA, y, mask, steps, sync, autocast=False, graph=False, fused=False, whatif=(), stage="regressor"):
    """`whatif` (development): pieces left out to see what they cost -- "noopt" (no optimizer step), "sumloss" (the loss
    replaced by a plain sum of the predictions)."""
    model.train()
    if stage == "classifier":
        # train.py:83-91 (classifier_transfered, parameters.py:97-134): three parameter groups with their own learning rates
        params = [{"params": list(model.gclstm_encoder.parameters()), "lr": 2.5e-3 * 0.1 * 0.1},
                  {"params": list(model.gclstm_decoder.parameters()), "lr": 2.5e-3 * 0.1},
                  {"params": list(model.lin2.parameters())}]
        lr = 2.5e-3
    else:
        params, lr = model.parameters(), 5e-3
    if fused == "ggnn":
        opt = training.FusedAdam(params, lr=lr)
    else:
        opt = torch.optim.Adam(params, lr=lr, capturable=graph, **({"fused": True} if fused else {}))
    losses = []

    def one():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            pred = model(X, EI, EA)
            if stage == "classifier":
                loss = training.classifier_loss(y, pred, 1.0)
            else:
                loss = (pred["joint"].sum() + pred["grain"].sum()) if "sumloss" in whatif else training.regressor_loss(y, pred, mask)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if "noopt" not in whatif:
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


def kernel_rooflines(model, X, EI, EA, y, mask, autocast):
    """Live durations of the training path's heavy C-ABI calls inside real (eager) training steps: HIP events on the
    launch stream bracketing the C call itself (the backend's `_launch`), a spin kernel in front so that the GPU stays
    behind the host.  Returns one record per call kind: the weight-gradient GEMM against the fp32 matrix pipe it runs on,
    the sweep's backward and the row GEMMs against HBM (what they touch at least once)."""
    from graingraphnn_amd.backend import default_backend
    be = default_backend()
    launch = be._launch
    rec = {}

    def timed_launch(fn, name, *cargs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch(fn, name, *cargs)
        e1.record()
        rec.setdefault(name, []).append((e0, e1, cargs))

    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    be._launch = timed_launch
    try:
        for it in range(4):
            if it == 1:
                rec.clear()          # the first step allocates and packs
            torch.cuda._sleep(int(4e7))
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                loss = training.regressor_loss(y, model(X, EI, EA), mask)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
    finally:
        del be._launch
    n_j, n_g = X["joint"].size(0), X["grain"].size(0)
    out = []
    for name, evs in sorted(rec.items()):
        us = [e0.elapsed_time(e1) * 1e3 for e0, e1, _ in evs]
        r = {"call": name, "calls_per_step": round(len(evs) / 3.0, 1), "avg_us": round(sum(us) / len(us), 1),
             "us_per_step": round(sum(us) / 3.0, 1)}
        if name == "ggnn_wgrad":
            # C = A^T B over the nodes: 2 K M Nc batch flops per call, exact fp32 MFMA (157.3 TFLOP/s dense)
            fl = [2.0 * a[0]._obj.K * a[0]._obj.M * a[0]._obj.Nc * a[0]._obj.batch for _, _, a in evs]
            tf = sum(fl) / sum(us) / 1e6
            r.update(bound="mfma", achieved=round(tf, 1), peak=157.3, unit="TFLOP/s", frac=round(tf / 157.3, 4))
        elif name == "ggnn_period_gat_aggregate_backward":
            # per sweep: the destination pass reads u / agg / g_agg rows and per edge the source's value + hidden rows,
            # writes the destination-side gradients and (alpha, ds) records; the source pass gathers them back:
            # >= 4 G 96 (3 n_dst + 2 n_src) + 4 E (2 G 96 + 2 G) bytes touched at least once
            by = []
            for _, _, a in evs:
                o = a[0]._obj
                by.append(4.0 * o.n_gates * 96 * (3 * o.n_dst + 2 * o.n_src) + 4.0 * o.E * (2 * o.n_gates * 96 + 2 * o.n_gates))
            gbs = sum(by) / sum(us) / 1e3
            r.update(bound="hbm", achieved=round(gbs, 1), peak=8000.0, unit="GB/s", frac=round(gbs / 8000.0, 4))
        elif name == "ggnn_rowgemm":
            by = [4.0 * a[0]._obj.M * a[0]._obj.batch * (a[0]._obj.K + a[0]._obj.n_out) for _, _, a in evs]   # A in, C out
            gbs = sum(by) / sum(us) / 1e3
            r.update(bound="hbm", achieved=round(gbs, 1), peak=8000.0, unit="GB/s", frac=round(gbs / 8000.0, 4))
        if name in ("ggnn_rowgemm", "ggnn_wgrad"):
            shapes = {}
            for (e0, e1, a) in evs:
                o = a[0]._obj
                key = "M%d K%d N%d b%d" % (o.M, o.K, o.n_out if name == "ggnn_rowgemm" else o.Nc, o.batch)
                shapes.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
            r["shapes"] = {k: [len(v) // 3, round(sum(v) / len(v), 1)] for k, v in shapes.items()}
        out.append(r)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--cfg3", action="store_true")
    ap.add_argument("--classifier", action="store_true",
                    help="the classifier stage (train.py:40-71, 83-91): BCE-with-logits on the edge events, three parameter groups")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the whole training step from one hipGraph")
    ap.add_argument("--fused", action="store_true", help="torch.optim.Adam(fused=True): 14 optimizer launches instead of ~40")
    ap.add_argument("--ggnn-adam", action="store_true", help="training.FusedAdam: the whole update in one launch (ggnn_adam_step)")
    ap.add_argument("--json", action="store_true", help="print one JSON line instead of text (bench.py's train_step record)")
    ap.add_argument("--whatif", default="", help="development: comma list of noopt, sumloss")
    ap.add_argument("--roofline", action="store_true", help="with --json: add live per-call roofline records (an eager pass)")
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
    stage = "classifier" if args.classifier else "regressor"
    if args.classifier:
        Cm = load_seeded(GrainNN_classifier(synthetic.default_hyper(dev), R), 1, 1.0).to(dev)
        E = EI[("joint", "connect", "joint")].size(1)
        lab = rs.randint(-1, 2, E).astype(np.float32)          # -1: unlabelled edge (train.py:46), 0 / 1: no event / event
        Y = {"edge_event": torch.from_numpy(lab).to(dev)}
        R = Cm
    dt, losses = run(R, X, EI, EA, Y, M, args.steps, torch.cuda.synchronize, args.bf16, args.graph, "ggnn" if args.ggnn_adam else args.fused,
                     tuple(w for w in args.whatif.split(",") if w), stage)
    if args.json:
        import json
        extra = {"kernel_rooflines": kernel_rooflines(R, X, EI, EA, Y, M, args.bf16)} if args.roofline else {}
        print(json.dumps({**extra, "workload": name, "ms_per_step": round(dt * 1e3, 3), "steps": args.steps,
                          "launch": "hipGraph replay (training.GraphedTrainStep recipe)" if args.graph else "eager",
                          "what": ("forward, loss (train.py:40-71: BCE with logits over the labelled edges), backward, Adam step (three "
                                   "parameter groups, train.py:83-91) of the classifier; " if args.classifier else
                                   "forward, loss (train.py:31-37), backward, Adam step of the regressor; ") +
                                  ("torch.autocast(bfloat16): the decoder projection in bf16 MFMA arithmetic "
                                   "(GGNN_PRECISION_BF16), sweeps / softmax / LSTM / gradients fp32" if args.bf16 else "fp32"),
                          "optimizer": "training.FusedAdam (ggnn_adam_step)" if args.ggnn_adam else "torch.optim.Adam(fused=True)" if args.fused else "torch.optim.Adam (default: foreach)",
                          "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)]}))
        return
    print(f"{name}: HIP training path{' (bf16 autocast GEMMs)' if args.bf16 else ''}"
          f"{' (hipGraph replay)' if args.graph else ''}{' (ggnn_adam_step)' if args.ggnn_adam else ' (fused Adam)' if args.fused else ''}: {dt * 1e3:.2f} ms/step, "
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
