#!/usr/bin/env python3
"""Per-kernel micro-benchmark on the cfg3 workload (development aid, GPU only).

    python tools/kbench.py [--reps 30] [--n 100]

Times every launch of one decoder cell and one encoder cell of the regressor in isolation
(HIP events on the launch stream, L2/MALL state as in a real step because the launches run
in their natural order) and prints achieved GB/s (aggregation, algorithmic bytes of SURVEY
8d) and TFLOP/s (GEMMs).  Use under `rocprofv3 --pmc ...` for counters.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from graingraphnn_amd import engine, synthetic  # noqa: E402
from graingraphnn_amd.backend import default_backend  # noqa: E402
from graingraphnn_amd.models import GrainNN_regressor  # noqa: E402
from graingraphnn_amd.packing import EDGE_TYPES, NODE_TYPES  # noqa: E402
from graingraphnn_amd.seeding import load_seeded  # noqa: E402


def alg_bytes(n_src, n_dst, E, G):
    return 4 * (G * 96 * (2 * n_src + 2 * n_dst) + 3 * (n_src + n_dst) + E) + 4 * E + 4 * (n_dst + 1) + 8 * G * n_dst


class Timed:
    def __init__(self, be):
        self.be, self.rec = be, {}

    def __getattr__(self, name):
        fn = getattr(self.be, name)

        def wrapped(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.rec.setdefault((name, self.tag(name, a)), []).append((e0, e1))
            return r
        return wrapped

    @staticmethod
    def tag(name, a):
        if name == "project":
            return f"M={a[0].size(0)} K={a[3].size(1)} N={a[3].size(0)}"
        if name == "aggregate":
            return f"n_dst={a[3].size(0)} n_src={a[2].size(0)} G={a[-1]}"
        if name == "aggregate_batch":
            return f"sweeps={len(a[0])} G={a[0][0][-1]} bytes=" + str(sum(
                alg_bytes(sw[2].size(0), sw[3].size(0), sw[0].E, sw[-1]) for sw in a[0]))
        if name == "lstm_epilogue":
            return f"N={a[0].size(0)} Ka={a[1].size(2)} G={a[8]}"
        return ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--tile", type=int, default=0, help="renumber the nodes in tiles of this many cells (experiment)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    x, ei, ea = synthetic.honeycomb(args.n, 10, 0)
    if args.tile:
        inv = {}
        for nt in NODE_TYPES:
            xy = x[nt][:, :2]
            g = 1.0 / args.tile  # tiles of g x g of the unit square, row-major inside a tile
            tx, ty = np.floor(xy[:, 0] / g).astype(np.int64), np.floor(xy[:, 1] / g).astype(np.int64)
            key = np.lexsort((xy[:, 0], xy[:, 1], tx, ty))
            inv[nt] = np.empty(len(key), np.int64)
            inv[nt][key] = np.arange(len(key))
            x[nt] = x[nt][key]
        for et in list(ei):
            ei[et] = np.stack([inv[et[0]][ei[et][0]], inv[et[-1]][ei[et][1]]])
    R = load_seeded(GrainNN_regressor(synthetic.default_hyper(dev)), 0, 0.3).eval().to(dev)
    X, EI, EA = synthetic.to_torch(x, ei, ea, dev)
    be = Timed(default_backend())
    n_nodes = {nt: X[nt].size(0) for nt in NODE_TYPES}
    graph = engine.graph_for(default_backend(), EI, n_nodes)
    enc = R.gclstm_encoder.cell_list[0].packed(True)
    dec = R.gclstm_decoder.cell_list[0].packed(False)
    ws = engine.Workspace(enc, dec, n_nodes, dev)
    for _ in range(3):
        engine.run_encoder_decoder(default_backend(), enc, dec, graph, ws, X, EA)
    torch.cuda.synchronize()
    for _ in range(args.reps):
        engine.run_encoder_decoder(be, enc, dec, graph, ws, X, EA)
    torch.cuda.synchronize()
    E = {et: EI[et].size(1) for et in EDGE_TYPES}
    print(f"{'call':16s} {'shape':34s} {'med us':>8s} {'min us':>8s}  rate")
    total = 0.0
    for (name, tag), evs in be.rec.items():
        ms = np.array([a.elapsed_time(b) for a, b in evs])
        per_call = len(evs) // args.reps
        for k in range(per_call):  # launches of one forward that share a tag, in order
            t = ms[k::per_call]
            med, mn = float(np.median(t)) * 1e3, float(t.min()) * 1e3
            total += med
            rate = ""
            if name == "aggregate":
                et = [e for e in EDGE_TYPES if f"n_dst={n_nodes[e[-1]]} n_src={n_nodes[e[0]]}" in tag]
                et = et[k % len(et)] if et else EDGE_TYPES[0]
                G = int(tag.split("G=")[1])
                b = alg_bytes(n_nodes[et[0]], n_nodes[et[-1]], E[et], G)
                rate = f"{b / med / 1e3:8.0f} GB/s algorithmic ({b / 1e6:.1f} MB) {et[0][0]}->{et[-1][0]}"
            elif name == "aggregate_batch":
                b = int(tag.split("bytes=")[1])
                rate = f"{b / med / 1e3:8.0f} GB/s algorithmic ({b / 1e6:.1f} MB)"
            elif name == "project":
                M, K, N = (int(s.split("=")[1]) for s in tag.split())
                rate = f"{2 * M * K * N / med / 1e6:8.1f} TFLOP/s, {4 * M * N / med / 1e3:6.0f} GB/s written"
            elif name == "lstm_epilogue":
                N, Ka, G = (int(s.split("=")[1]) for s in tag.split())
                rate = f"{2 * N * Ka * 96 * G / med / 1e6:8.1f} TFLOP/s, {4 * N * (G * Ka + G * 96 + 3 * 96) / med / 1e3:6.0f} GB/s"
            print(f"{name:16s} {tag:34s} {med:8.1f} {mn:8.1f}  {rate}")
    print(f"sum of medians: {total:.1f} us per model forward")


if __name__ == "__main__":
    main()
