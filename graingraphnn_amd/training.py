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
    edge type; agg is [n_dst, G, 128] = (96 values, sum alpha, sum alpha * a, zeros)."""

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


def _edge_type_forward(convs, backend, topo, et, einfo, x_src, x_dst, h_src, h_dst, Fs, Fd):
    """The PeriodConvs of all gates of one edge type (HeteroConv entry, heteropgclstm.py:113-138):
    key-free operands from the parameters, one sweep, the linear tail.  -> [N_d, G, 96].
    The gates are batched into single GEMMs (weights stacked along the output dimension)."""
    G = len(convs)
    k2 = 0 if h_src is None else C
    inv = 1.0 / math.sqrt(C)
    n_d = x_dst.size(0)
    st = lambda f: torch.stack([f(cv) for cv in convs])                       # [G, ...]
    Xd = x_dst if not k2 else torch.cat([x_dst, h_dst], 1)
    xz = torch.cat([torch.zeros_like(x_src[:, :3]), x_src[:, 3:]], 1)         # the wrap moves columns 0..2 to the edge
    Xs = xz if not k2 else torch.cat([xz, h_src], 1)
    wq, bq = st(lambda cv: cv.lin_query.weight[:, :Fd + k2]), st(lambda cv: cv.lin_query.bias)
    wk, bk = st(lambda cv: cv.lin_key.weight[:, :Fs + k2]), st(lambda cv: cv.lin_key.bias)
    wv, bv = st(lambda cv: cv.lin_value.weight[:, :Fs + k2]), st(lambda cv: cv.lin_value.bias)
    ws, bs = st(lambda cv: cv.lin_skip.weight[:, :Fd + k2]), st(lambda cv: cv.lin_skip.bias)
    wl, bl = st(lambda cv: cv.lin_l2.weight), st(lambda cv: cv.lin_l2.bias)
    we = st(lambda cv: cv.lin_edge.weight[:, 0])                              # [G, 96]
    q = (Xd @ wq.reshape(G * C, -1).t() + bq.reshape(-1)).view(n_d, G, C)     # [N_d, G, 96]
    # the per-gate batched products stay fp32 under bf16 autocast: they feed the attention logits,
    # and the library's bf16 batched-GEMM backward is three orders of magnitude slower here
    with torch.autocast(q.device.type, enabled=False):
        q = q.float()
        u = torch.einsum("ngc,gcd->ngd", q, wk) * inv                         # u = W_k^T q / sqrt(96)
        s1 = torch.einsum("ngc,gc->ng", q, bk) * inv
        s2 = torch.einsum("ngc,gc->ng", q, we) * inv
    z = torch.zeros(n_d, G, 16, dtype=q.dtype, device=q.device)
    u4 = torch.cat([u[:, :, :Fs], z[:, :, Fs:12], s1.unsqueeze(-1), s2.unsqueeze(-1), z[:, :, 14:]], -1)
    p_dst = u4.reshape(n_d, G * 16) if not k2 else torch.cat([u[:, :, Fs:].reshape(n_d, G * C),
                                                              u4.reshape(n_d, G * 16)], 1)
    val = Xs @ wv.reshape(G * C, -1).t() + bv.reshape(-1)                     # [N_s, G * 96]
    ep = wv[:, :, :3].transpose(1, 2)                                         # [G, 3, 96]
    agg = _Sweep.apply(p_dst, val, h_src, ep, backend, topo, et, einfo, G)    # [N_d, G, 128]
    skip = (Xd @ ws.reshape(G * C, -1).t() + bs.reshape(-1)).view(n_d, G, C)
    with torch.autocast(agg.device.type, enabled=False):
        return (torch.einsum("ngc,gkc->ngk", agg[:, :, :C], wl) + agg[:, :, C:C + 1] * bl
                + agg[:, :, C + 1:C + 2] * we + skip.float())


def cell_forward(cell, backend, topo, einfo, x, h, c):
    """HeteroPGCLSTM.forward (heteropgclstm.py:101-183), differentiable.  h, c: dicts or None
    (encoder: zero state; the forget gate multiplies c = 0 and is skipped, its gradient is 0)."""
    gates = "ifco" if h is not None else "ico"
    F = cell.in_channels_dict
    pre = {nt: 0.0 for nt in NODE_TYPES}
    for et in EDGE_TYPES:
        s, d = et[0], et[-1]
        convs = [getattr(cell, "conv_" + g).convs[et_key(et)] for g in gates]
        pre[d] = pre[d] + _edge_type_forward(convs, backend, topo, et, einfo[et], x[s], x[d],
                                             None if h is None else h[s], None if h is None else h[d], F[s], F[d])
    h_new, c_new = {}, {}
    # Encoder: f * c with c = 0.  The reference still runs conv_f, so its parameters receive an
    # exactly zero gradient; they are touched here the same way, which also keeps
    # DistributedDataParallel(model, device_ids=[rank]) (dist_train.py:82) usable as written.
    touch = 0.0 if h is not None else 0.0 * sum(
        q.sum() for q in list(cell.conv_f.parameters()) + list(cell.b_f.parameters()))
    for nt in NODE_TYPES:
        p = {g: pre[nt][:, k] + getattr(cell, "b_" + g)[nt] for k, g in enumerate(gates)}
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
