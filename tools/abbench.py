#!/usr/bin/env python3
"""Development aid: time the sweep backward (ggnn_period_gat_aggregate_backward) on the cfg3 decoder / encoder shapes.
    GGNN_LIB_PATH=graingraphnn_amd/libggnn_variant.so python tools/abbench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from graingraphnn_amd import synthetic  # noqa: E402
from graingraphnn_amd.backend import default_backend  # noqa: E402
from graingraphnn_amd.engine import alloc_einfo, graph_for  # noqa: E402
from graingraphnn_amd.packing import EDGE_TYPES  # noqa: E402
from graingraphnn_amd.training import train_topology  # noqa: E402

be = default_backend()
x, ei, ea = synthetic.honeycomb(100, 10, 0)
X, EI, EA = synthetic.to_torch(x, ei, ea, "cuda")
graph = graph_for(be, EI, {nt: X[nt].size(0) for nt in X})
topo = train_topology(be, graph)
einfo = alloc_einfo(graph, "cuda")
be.edge_prepare([(graph.csr[et], EA[et].view(-1), X[et[0]], X[et[-1]], einfo[et]) for et in EDGE_TYPES])
for G, has_h in ((4, True), (3, False)):
    for et in EDGE_TYPES:
        ns, nd = X[et[0]].size(0), X[et[-1]].size(0)
        ld = G * 96 + (G * 96 if has_h else 0) + G * 16
        ps, pd = torch.randn(ns, ld, device="cuda"), torch.randn(nd, ld, device="cuda") * 0.1
        h = torch.randn(ns, 96, device="cuda") if has_h else None
        ep = torch.randn(G, 3, 96, device="cuda")
        agg, g_agg = torch.zeros(nd, G * 128, device="cuda"), torch.randn(nd, G * 128, device="cuda")
        offs = (0, G * 96, 2 * G * 96 if has_h else G * 96, 0, 128, 96)
        be.aggregate(graph.csr[et], einfo[et], ps, pd, h, ep, agg, *offs, G)
        f = lambda: be.aggregate_backward(graph.csr[et], topo.rcsr[et], topo.r_slot[et], einfo[et], ps, pd, h, ep, agg,
                                          g_agg, *offs, G)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        print(f"G {G} {'h' if has_h else '-'} {et[0]}->{et[-1]}: backward (both passes + partial sum) {(time.perf_counter() - t0) / 20 * 1e6:7.1f} us")
