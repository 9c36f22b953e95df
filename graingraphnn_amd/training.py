"""Training path (first stage): a differentiable forward of the same modules.

`GrainNN_regressor.forward` / `GrainNN_classifier.forward` come here when autograd is recording
and the module is in training mode, i.e. inside the reference's loop (train.py:158-166:
`model.train(); pred = model(...); loss.backward(); optimizer.step()`).  What runs where:

  * the aggregation (PeriodConv.message + propagate, periodGATconv.py:174-175, 204-236) is the
    HIP sweep `ggnn_period_gat_aggregate` with its hand-written backward
    `ggnn_period_gat_aggregate_backward` (segment-softmax backward, relu mask, atomics-free
    scatter to the source rows) behind one `torch.autograd.Function`;
  * the dense algebra around it (query / key-transpose / value / l2 / skip linears, the LSTM
    update, the heads) is plain library GEMMs and pointwise ops on the same device, recorded by
    autograd.  It uses the key-free form of DESIGN.md section 2 directly on the parameters
    (u = (W_q x_i + b_q) W_k / sqrt(96)), so gradients reach every reference parameter.

The fused inference kernels (ggnn_project, ggnn_lstm_epilogue, heads) have no backward; they are
not used here.  Same results as the inference path up to fp32 re-association.
"""
import math
from typing import Dict

import torch

from .backend import default_backend
from .engine import _check_x, _edge_attr_1d, alloc_einfo, graph_for
from .packing import C, EDGE_TYPES, NODE_TYPES, et_key

_KG = 128  # row pitch of one gate in the sweep's output: 96 values, sum(alpha), sum(alpha * a), padding


class TrainTopology:
    """Forward CSR + reverse (source-grouped) CSR of the three edge types of one topology."""

    def __init__(self, backend, graph):
        self.graph = graph
        self.rcsr, self.r_slot = {}, {}
        for et in EDGE_TYPES:
            csr, ei = graph.csr[et], graph.edge_index[et]
            n_src, n_dst = graph.n_nodes[et[0]], graph.n_nodes[et[-1]]
            self.rcsr[et] = backend.build_csr(ei.flip(0).contiguous(), n_dst, n_src)
            E = csr.E
            if E == 0:  # an edge type without edges: the CSR's perm is an unwritten placeholder
                self.r_slot[et] = torch.zeros(1, dtype=torch.int32, device=ei.device)
                continue
            inv = torch.empty(E, dtype=torch.int32, device=ei.device)
            inv[csr.perm[:E].long()] = torch.arange(E, dtype=torch.int32, device=ei.device)
            self.r_slot[et] = inv[self.rcsr[et].perm[:E].long()].contiguous()


_topo_cache: Dict[int, TrainTopology] = {}


def train_topology(backend, graph) -> TrainTopology:
    t = _topo_cache.get(id(graph))
    if t is None or t.graph is not graph:
        if len(_topo_cache) >= 8:
            _topo_cache.pop(next(iter(_topo_cache)))
        t = _topo_cache[id(graph)] = TrainTopology(backend, graph)
    return t


class _Sweep(torch.autograd.Function):
    """agg = sweep(p_dst = [u_h (G x 96) | u4 (G x 16)], v (G x 96), h_src, edge_params) for one
    edge type; agg is [n_dst, G, 128] = (96 values, sum alpha, sum alpha * a, zeros).  (Single-sweep
    form, kept for op-level tests; the models go through _CellSweeps.)"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)  # the sweep is fp32 under bf16 autocast too
    def forward(ctx, p_dst, v, h_src, ep, backend, topo, et, einfo, G):
        p_dst, v, ep = p_dst.contiguous(), v.contiguous(), ep.contiguous()
        h_src = None if h_src is None else h_src.contiguous()
        agg = torch.zeros(p_dst.size(0), G * _KG, dtype=torch.float32, device=p_dst.device)
        offs = (0, 0, G * C if h_src is not None else 0, 0, _KG, C)  # v, u_h, u4, a, gate stride, scalars
        backend.aggregate(topo.graph.csr[et], einfo, v, p_dst, h_src, ep, agg, *offs, G)
        ctx.save_for_backward(p_dst, v, h_src, ep, agg, einfo)
        ctx.misc = (backend, topo, et, G, offs)
        return agg.view(-1, G, _KG)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_agg):
        p_dst, v, h_src, ep, agg, einfo = ctx.saved_tensors
        backend, topo, et, G, offs = ctx.misc
        g_agg = g_agg.contiguous().view(-1, G * _KG)
        g_p_dst, g_v, g_h, g_ep = backend.aggregate_backward(
            topo.graph.csr[et], topo.rcsr[et], topo.r_slot[et], einfo, v, p_dst, h_src, ep, agg, g_agg, *offs, G)
        return g_p_dst, g_v, g_h, g_ep, None, None, None, None, None


# ---------------------------------------------------------------------------------------
# The cell in its packed (inference) formulation, differentiable.
#
# One projection GEMM per node type produces, for all gates and edge types at once, what the
# sweeps and the gate stage consume (value rows, key-free score rows u_h / u4, summed skip rows:
# DESIGN.md section 2, packing.py); its weight matrix is assembled from the reference parameters
# by differentiable torch ops on weight-sized tensors, so autograd carries the gradient back to
# every lin_key / lin_query / lin_value / lin_skip / lin_l2 / lin_edge / gate bias.  All parameters
# of a cell are first gathered into ONE flat buffer (a single cat kernel); every stacked per-gate
# weight is a view of it.
# ---------------------------------------------------------------------------------------
_KINDS = (("wq", "lin_query", "weight"), ("bq", "lin_query", "bias"), ("wk", "lin_key", "weight"),
          ("bk", "lin_key", "bias"), ("wv", "lin_value", "weight"), ("bv", "lin_value", "bias"),
          ("ws", "lin_skip", "weight"), ("bs", "lin_skip", "bias"), ("wl", "lin_l2", "weight"),
          ("bl", "lin_l2", "bias"), ("we", "lin_edge", "weight"))


def _gather_params(cell, gates):
    """All parameters the cell's forward reads, stacked per gate: get(et, kind) -> [G, *param.shape];
    gate bias get("b", nt) -> [G, 96].  ONE cat kernel gathers them into a flat buffer and ONE
    split hands out the views (a slice per tensor would cost a full-size zero fill + add each in the
    backward; the split's backward is a single cat)."""
    G = len(gates)
    plist, keys, shapes, sizes = [], [], [], []
    for et in EDGE_TYPES:
        for kind, lin, wb in _KINDS:
            ts = [getattr(getattr(getattr(cell, "conv_" + g).convs[et_key(et)], lin), wb) for g in gates]
            keys.append((et, kind))
            shapes.append((G,) + tuple(ts[0].shape))
            sizes.append(G * ts[0].numel())
            plist += ts
    for nt in NODE_TYPES:
        ts = [getattr(cell, "b_" + g)[nt] for g in gates]
        keys.append(("b", nt))
        shapes.append((G, C))
        sizes.append(G * C)
        plist += ts
    parts = torch.split(torch.cat([t.reshape(-1) for t in plist]), sizes)
    table = {k: part.view(shape) for k, part, shape in zip(keys, parts, shapes)}
    return lambda *key: table[key]


_zero_cache = {}


def _zeros(dev, *shape):
    """Constant zero blocks of the packed matrices (no gradient flows into them): made once."""
    key = (str(dev), shape)
    z = _zero_cache.get(key)
    if z is None:
        z = _zero_cache[key] = torch.zeros(*shape, dtype=torch.float32, device=dev)
    return z


def _packed_weights(cell, gates, F, sees_h):
    """-> (layout, wp, bp, ep, w2): the projection weight [ncols, F + k2] / bias [ncols] per node type in
    the column order of packing.node_layout (value rows, u_h rows, skip rows, u4 tails), the reloc
    columns of lin_value per edge type [G, 3, 96] and the gate weight [G, 96, Ka] per node type."""
    from .packing import node_layout
    G, k2 = len(gates), (C if sees_h else 0)
    scale = 1.0 / math.sqrt(C)
    get = _gather_params(cell, gates)
    dev = get(EDGE_TYPES[0], "wq").device
    zeros = lambda *shape: _zeros(dev, *shape)
    layout, wp, bp, ep, w2 = {}, {}, {}, {}, {}
    prod = {}
    cut = lambda t, n, dim: t if t.size(dim) == n else t.narrow(dim, 0, n)   # (the encoder ignores the h columns)
    wv3, wvr = {}, {}
    for et in EDGE_TYPES:  # key-free score rows: M = W_k^T W_q / sqrt(96) and friends, per gate
        Fs, Fd = F[et[0]], F[et[-1]]
        wq, bq = cut(get(et, "wq"), Fd + k2, 2), get(et, "bq")
        wk, bk = cut(get(et, "wk"), Fs + k2, 2), get(et, "bk")
        we = get(et, "we").squeeze(-1)
        wkt = wk.transpose(1, 2)                                     # [G, Fs + k2, 96]
        # (splits, not slices: a slice's backward is a full-size zero fill + add, a split's one cat)
        Mb = torch.bmm(wkt, torch.cat([wq, bq.unsqueeze(-1)], 2)) * scale          # [G, Fs + k2, Fd + k2 + 1]
        Rb = torch.bmm(torch.stack([bk, we], 1), torch.cat([wq, bq.unsqueeze(-1)], 2)) * scale   # [G, 2, . + 1]: s1, s2 rows
        if sees_h:
            Mb_x, Mb_h = torch.split(Mb, [Fs, k2], 1)
        else:
            Mb_x, Mb_h = Mb, None
        tail = torch.cat([Mb_x, zeros(G, 12 - Fs, Fd + k2 + 1), Rb, zeros(G, 2, Fd + k2 + 1)], 1)   # [G, 16, . + 1]
        prod[et] = (Mb_h, tail)
        wv3[et], wvr[et] = torch.split(cut(get(et, "wv"), Fs + k2, 2), [3, Fs + k2 - 3], 2)
    for nt in NODE_TYPES:
        lay = node_layout(nt, F[nt], G, EDGE_TYPES, True, sees_h, True)
        Fn, D = F[nt], F[nt] + k2
        blocks = []                                                  # rows of [W | b]: [n, D + 1]
        for et in lay.src_ets:                                       # value rows, reloc columns zeroed
            blocks.append(torch.cat([zeros(G, C, 3), wvr[et], get(et, "bv").unsqueeze(-1)], 2).reshape(G * C, D + 1))
        if sees_h:
            for et in lay.dst_ets:                                   # hidden-state part of u
                blocks.append(prod[et][0].reshape(G * C, D + 1))
        ws = sum(cut(get(et, "ws"), D, 2) for et in lay.dst_ets)      # HeteroConv aggr 'sum' -> summed skip
        bs = sum(get(et, "bs") for et in lay.dst_ets) + get("b", nt)
        blocks.append(torch.cat([ws, bs.unsqueeze(-1)], 2).reshape(G * C, D + 1))
        for et in lay.dst_ets:
            blocks.append(prod[et][1].reshape(G * 16, D + 1))
        n_rows = sum(b.size(0) for b in blocks)
        blocks.append(zeros(lay.ncols - n_rows, D + 1))
        wp[nt], bp[nt] = torch.split(torch.cat(blocks), [D, 1], 1)
        bp[nt] = bp[nt].reshape(-1)
        layout[nt] = lay
        n_in = len(lay.dst_ets)
        w2[nt] = torch.cat([get(et, "wl") for et in lay.dst_ets]
                           + [torch.cat([get(et, "bl").unsqueeze(-1), get(et, "we")], 2) for et in lay.dst_ets]
                           + [zeros(G, C, lay.Ka - n_in * (C + 2))], 2)   # [G, 96, Ka]
    for et in EDGE_TYPES:
        ep[et] = wv3[et].transpose(1, 2).contiguous()                # [G, 3, 96]
    return layout, wp, bp, ep, w2


class _CellSweeps(torch.autograd.Function):
    """The three aggregation sweeps of one cell on the projections P[nt] (layout as in inference):
    (P_grain, P_joint, h_grain, h_joint, ep_gj, ep_jg, ep_jj) -> (agg_grain, agg_joint).  One launch
    forward; backward = ggnn_period_gat_aggregate_backward per edge type into shared gradient
    buffers (each sweep owns its columns)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)  # the sweeps are fp32 under bf16 autocast too
    def forward(ctx, P_grain, P_joint, h_grain, h_joint, ep_gj, ep_jg, ep_jj, backend, topo, einfo, layout, G):
        P = {"grain": P_grain.contiguous(), "joint": P_joint.contiguous()}
        h = {"grain": None if h_grain is None else h_grain.contiguous(),
             "joint": None if h_joint is None else h_joint.contiguous()}
        ep = dict(zip(EDGE_TYPES, (ep_gj.contiguous(), ep_jg.contiguous(), ep_jj.contiguous())))
        agg = {nt: torch.zeros(P[nt].size(0), G * layout[nt].Kg, dtype=torch.float32, device=P[nt].device)
               for nt in NODE_TYPES}
        sweeps = []
        for et in EDGE_TYPES:
            s, d = et[0], et[-1]
            sweeps.append((topo.graph.csr[et], einfo[et], P[s], P[d], h[s], ep[et], agg[d], layout[s].v_off[et],
                           layout[d].u_off.get(et, 0), layout[d].u4_off[et], layout[d].a_off[et], layout[d].Kg,
                           layout[d].sc_off[et], G))
        backend.aggregate_batch(sweeps)
        ctx.save_for_backward(P["grain"], P["joint"], h["grain"], h["joint"], ep_gj, ep_jg, ep_jj, agg["grain"],
                              agg["joint"], *[einfo[et] for et in EDGE_TYPES])
        ctx.misc = (backend, topo, layout, G, sweeps)
        return agg["grain"], agg["joint"]

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_grain, g_joint):
        Pg, Pj, hg, hj, ep_gj, ep_jg, ep_jj, agg_g, agg_j, *einfos = ctx.saved_tensors
        backend, topo, layout, G, sweeps = ctx.misc
        P, h = {"grain": Pg, "joint": Pj}, {"grain": hg, "joint": hj}
        agg = {"grain": agg_g, "joint": agg_j}
        g_agg = {"grain": g_grain.contiguous(), "joint": g_joint.contiguous()}
        ep = dict(zip(EDGE_TYPES, (ep_gj, ep_jg, ep_jj)))
        gP = {nt: torch.zeros_like(P[nt]) for nt in NODE_TYPES}
        gh = {nt: None for nt in NODE_TYPES}
        g_ep = {}
        for et, einfo in zip(EDGE_TYPES, einfos):
            s, d = et[0], et[-1]
            _, _, g_h, g_ep[et] = backend.aggregate_backward(
                topo.graph.csr[et], topo.rcsr[et], topo.r_slot[et], einfo, P[s], P[d], h[s], ep[et], agg[d], g_agg[d],
                layout[s].v_off[et], layout[d].u_off.get(et, 0), layout[d].u4_off[et], layout[d].a_off[et],
                layout[d].Kg, layout[d].sc_off[et], G, out_p_dst=gP[d], out_p_src=gP[s])
            if g_h is not None:
                gh[s] = g_h if gh[s] is None else gh[s] + g_h
        return (gP["grain"], gP["joint"], gh["grain"], gh["joint"], g_ep[EDGE_TYPES[0]], g_ep[EDGE_TYPES[1]],
                g_ep[EDGE_TYPES[2]], None, None, None, None, None)


def cell_forward(cell, backend, topo, einfo, x, h, c):
    """HeteroPGCLSTM.forward (heteropgclstm.py:101-183), differentiable, in the packed formulation:
    2 projection GEMMs, 1 launch of the three sweeps, 2 batched gate GEMMs, the LSTM update.
    h, c: dicts or None (encoder: zero state; the forget gate multiplies c = 0 and is skipped)."""
    gates = "ifco" if h is not None else "ico"
    G = len(gates)
    F = cell.in_channels_dict
    with torch.autocast(x["joint"].device.type, enabled=False):   # weight-sized products stay fp32
        layout, wp, bp, ep, w2 = _packed_weights(cell, gates, F, h is not None)
    P = {}
    for nt in NODE_TYPES:
        xin = x[nt] if h is None else torch.cat([x[nt], h[nt]], 1)
        P[nt] = torch.nn.functional.linear(xin, wp[nt], bp[nt])       # [N, ncols] (bf16 under autocast)
    agg_g, agg_j = _CellSweeps.apply(P["grain"], P["joint"], None if h is None else h["grain"],
                                     None if h is None else h["joint"], ep[EDGE_TYPES[0]], ep[EDGE_TYPES[1]],
                                     ep[EDGE_TYPES[2]], backend, topo, einfo, layout, G)
    agg = {"grain": agg_g, "joint": agg_j}
    # Encoder: f * c with c = 0.  The reference still runs conv_f, so its parameters receive an
    # exactly zero gradient; they are touched here the same way, which also keeps
    # DistributedDataParallel(model, device_ids=[rank]) (dist_train.py:82) usable as written.
    touch = 0.0 if h is not None else 0.0 * sum(
        q.sum() for q in list(cell.conv_f.parameters()) + list(cell.b_f.parameters()))
    h_new, c_new = {}, {}
    for nt in NODE_TYPES:
        lay = layout[nt]
        n = agg[nt].size(0)
        with torch.autocast(agg[nt].device.type, enabled=False):
            a = agg[nt].view(n, G, lay.Kg).transpose(0, 1)                         # [G, N, Kg] (pad columns meet zero weights)
            w2p = w2[nt] if lay.Kg == lay.Ka else torch.cat([w2[nt], _zeros(w2[nt].device, G, C, lay.Kg - lay.Ka)], 2)
            skip = P[nt].narrow(1, lay.s_off, G * C).float().view(n, G, C)
            pre = torch.bmm(a, w2p.transpose(1, 2)).transpose(0, 1) + skip         # [N, G, 96]
        p = {g: pre[:, k] for k, g in enumerate(gates)}
        cand = torch.sigmoid(p["i"]) * torch.tanh(p["c"])
        c_new[nt] = cand + touch if h is None else torch.sigmoid(p["f"]) * c[nt] + cand
        h_new[nt] = torch.sigmoid(p["o"]) * torch.tanh(c_new[nt])
    return h_new, c_new


def encoder_decoder(model, x_dict, edge_index_dict, edge_attr):
    """models.py:422-426 / 581-585 with autograd.  Returns the decoder's h_dict."""
    be = default_backend()
    for nt in NODE_TYPES:
        _check_x(x_dict[nt], model.in_channels_dict[nt], nt)
    n_nodes = {nt: x_dict[nt].size(0) for nt in NODE_TYPES}
    graph = graph_for(be, edge_index_dict, n_nodes)
    topo = train_topology(be, graph)
    x = {nt: x_dict[nt].detach().contiguous() for nt in NODE_TYPES}
    with torch.no_grad():
        einfo = alloc_einfo(graph, x["joint"].device)
        be.edge_prepare([(graph.csr[et], _edge_attr_1d(edge_attr[et]), x[et[0]], x[et[-1]], einfo[et])
                         for et in EDGE_TYPES])
    h, c = cell_forward(model.gclstm_encoder.cell_list[0], be, topo, einfo, x, None, None)
    h, c = cell_forward(model.gclstm_decoder.cell_list[0], be, topo, einfo, x, h, c)
    return h, graph


def regressor_forward(model, x_dict, edge_index_dict, edge_attr):
    """GrainNN_regressor.forward (models.py:401-467) with autograd."""
    h, _ = encoder_decoder(model, x_dict, edge_index_dict, edge_attr)
    y_joint = torch.tanh(model.linear["joint"](h["joint"]))
    yg = model.linear["grain"](h["grain"])
    y0 = torch.tanh(yg[:, 0])
    area = y0 / model.scaling["grain"] + x_dict["grain"][:, 3]
    y_grain = torch.stack([y0, torch.relu(yg[:, 1])], 1)
    return {"grain": y_grain, "joint": y_joint, "grain_area": area}


def classifier_forward(model, x_dict, edge_index_dict, edge_attr):
    """GrainNN_classifier.forward (models.py:572-611) with autograd."""
    h, graph = encoder_decoder(model, x_dict, edge_index_dict, edge_attr)
    et = ("joint", "connect", "joint")
    src, dst = graph.edge_index[et][0], graph.edge_index[et][1]
    pair = torch.cat([h["joint"][src], h["joint"][dst], edge_attr[et].view(-1, 1)], -1)
    return {"edge_event": model.lin2(pair).view(-1), "edge": torch.tanh(model.lin1(pair))}


def regressor_loss(y_dict, pred, mask):
    """train.py:31-37 (edge_len off): 100 * (mean(mask_j (y_j - p_j)^2) + mean(mask_g (y_g - p_g)^2))."""
    return 100 * (torch.mean(mask["joint"] * (y_dict["joint"] - pred["joint"]) ** 2)
                  + torch.mean(mask["grain"] * (y_dict["grain"] - pred["grain"]) ** 2))


def classifier_loss(y_dict, pred, pos_weight: float = 1.0):
    """train.py:40-70 (edge_len off): BCE-with-logits over the labelled edges (label > -1)."""
    y, z = y_dict["edge_event"], pred["edge_event"]
    keep = y > -1
    return torch.nn.functional.binary_cross_entropy_with_logits(
        z[keep], y[keep].float(), pos_weight=torch.tensor(pos_weight, device=z.device))


def wants_autograd(model) -> bool:
    return torch.is_grad_enabled() and model.training and any(p.requires_grad for p in model.parameters())


class GraphedTrainStep:
    """One optimisation step of train.py:158-166 (forward, loss, backward, optimizer.step) replayed from
    a hipGraph.  The eager step is bound by the host (~700 launches: 10.8 ms at the 10k-grain graph, 5.9 ms of
    it on the GPU); captured once, a step is one graph launch (7.8 ms there, 4.5 ms for a collated batch of
    four 40 um graphs).  Static shapes and topology: the inputs are copied into the buffers of the captured
    step, so every call must bring a batch of the captured sizes on the same edge lists (what a fixed-size
    DataLoader batch of one structure family gives; anything else: run the eager step).

        step = GraphedTrainStep(model, optimizer, lambda pred, y: regressor_loss(y, pred, mask), X, EI, EA, Y)
        loss = step(X, EA, Y)            # same tensors or new values of the same shapes

    `optimizer` must be created with `capturable=True` (torch.optim.Adam(..., capturable=True)); gradients
    are kept as buffers (zero_grad(set_to_none=False)).  Warm-up (3 eager steps, which DO update the model)
    and capture (which only records) run on a side stream, PyTorch's whole-network capture recipe."""

    def __init__(self, model, optimizer, loss_fn, x_dict, edge_index_dict, edge_attr, y_dict, autocast_dtype=None,
                 warmup: int = 3):
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.x = {k: v.detach().clone() for k, v in x_dict.items()}
        self.ei = edge_index_dict
        self.ea = {k: v.detach().clone() for k, v in edge_attr.items()}
        self.y = {k: v.detach().clone() for k, v in y_dict.items()}
        self.autocast_dtype = autocast_dtype
        model.train()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager()
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=s):
                self._loss = self._eager()
        torch.cuda.current_stream().wait_stream(s)

    def _eager(self):
        with torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            loss = self.loss_fn(self.model(self.x, self.ei, self.ea), self.y)
        self.opt.zero_grad(set_to_none=False)
        loss.backward()
        self.opt.step()
        return loss

    @torch.no_grad()
    def __call__(self, x_dict=None, edge_attr=None, y_dict=None):
        """Copies the batch into the captured buffers (tensors that ARE the buffers are skipped), replays
        the step and returns the loss of that step (a tensor the next replay overwrites)."""
        for dst, src in ((self.x, x_dict), (self.ea, edge_attr), (self.y, y_dict)):
            for k, v in (src or {}).items():
                if v is not dst[k]:
                    dst[k].copy_(v)
        self.graph.replay()
        return self._loss
