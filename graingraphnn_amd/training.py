"""Training path: a differentiable forward of the same modules.

`GrainNN_regressor.forward` / `GrainNN_classifier.forward` come here when autograd is recording
and the module is in training mode, i.e. inside the reference's loop (train.py:158-166:
`model.train(); pred = model(...); loss.backward(); optimizer.step()`).  What runs where:

  * the packed weight matrices of a cell (packing.py's layout) are an autograd function of the reference
    parameters: index-table gathers + one batched product each way (`train_pack._PackWeights`), so gradients
    reach every lin_key / lin_query / lin_value / lin_skip / lin_l2 / lin_edge / gate bias;
  * a cell (heteropgclstm.py:101-183) is ONE autograd function on HIP kernels (`_PackedCell`): the inference
    projection `ggnn_project_batch`, the sweeps `ggnn_period_gat_aggregate_batch` (PeriodConv.message + propagate,
    periodGATconv.py:174-175, 204-236) with their hand-written backward `ggnn_period_gat_aggregate_backward`
    (segment-softmax backward, relu mask, atomics-free scatter to the source rows), the LSTM update
    `ggnn_lstm_train_forward / _backward`, every weight gradient through the split-K `ggnn_wgrad`; the gate GEMM,
    its input gradient and the hidden-state gradient of the projection through `ggnn_rowgemm` (round 4: they were
    library GEMMs) -- no BLAS call is left inside a cell;
  * the regressor's heads are `_RegressorHeads` (the inference head kernel forward, `ggnn_heads_regressor_backward`
    + `ggnn_wgrad` backward); the classifier's pair heads are `_ClassifierHeads` (the inference kernels forward, a
    node-level backward: segment sums over the forward / reverse CSR in a fixed order, [n, 6] products; layer_size < 96:
    `_RowLinear` on the recorded pair matrix).

Under `torch.autocast(bfloat16)` (BASELINE config 5) the cells' dense products -- the decoder projection, the gate
GEMM, its input gradient, the hidden-state gradient -- run in real bf16 arithmetic on the HIP kernels (operands rounded
to bf16, ONE MFMA product per k-step instead of three or six, fp32 accumulate: GGNN_PRECISION_BF16), as do the recorded
2-D GEMMs around the cells; the sweep, the softmax, the LSTM update and the weight gradients stay fp32.
Without autocast everything is fp32-equivalent: same results as the inference path up to fp32 re-association.
"""
from typing import Dict

import torch
from torch.autograd.function import once_differentiable

from . import _lib, _pins
from .backend import default_backend
from .engine import _check_x, _edge_attr_1d, alloc_einfo, graph_for
from .packing import C, EDGE_TYPES, NODE_TYPES
from .train_pack import _ones, _zeros, packed_weights, packed_weights_of

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

    def jj_index(self):
        """int64 index tensors of the junction-junction edges for the classifier heads' backward (made once): the original
        edge of every forward CSR slot, the forward slot of every reverse CSR slot, and the two row pointers."""
        jj = getattr(self, "_jj", None)
        if jj is None:
            et = ("joint", "connect", "joint")
            csr, rcsr, E = self.graph.csr[et], self.rcsr[et], self.graph.csr[et].E
            jj = self._jj = {"slot_edge": csr.perm[:E].long(), "rowptr": csr.rowptr.long(),
                             "r_slot": self.r_slot[et][:E].long(), "r_rowptr": rcsr.rowptr.long()}
        return jj


_topo_cache: Dict[int, TrainTopology] = {}


def train_topology(backend, graph) -> TrainTopology:
    t = _topo_cache.get(id(graph))
    if t is None or t.graph is not graph:
        if len(_topo_cache) >= 8:
            _topo_cache.pop(next(iter(_topo_cache)))
        t = _topo_cache[id(graph)] = TrainTopology(backend, graph)
    return _pins.note(t)


# ---------------------------------------------------------------------------------------
# The cell in its packed (inference) formulation, differentiable: the packed matrices come from
# train_pack.packed_weights (an autograd function of the reference parameters), the cell that
# consumes them is _PackedCell.
# ---------------------------------------------------------------------------------------

def _wgrad2d(backend, a, b, b_ins=None, ins_off=0, defer=None):
    """a^T b for contiguous a [K, M], b [K, Nc] through ggnn_wgrad, in the cheaper of the two orientations: a wave
    computes a (32 or 64) x 112 block whatever part of it is inside the matrix, so a narrow factor (the encoder's
    [x | 1]: 12 columns) belongs on the row side.  `b_ins` [K, w]: b stands for itself with these columns inserted at
    column ins_off (read where they lie, ggnn_wgrad_args.b_ins: no concatenated copy; the product keeps this orientation).
    `defer`: see backend.wgrad (the result is complete after backend.sum_rows_batch(defer))."""
    K, M, Nc = a.size(0), a.size(1), b.size(1)
    if b_ins is not None:
        return backend.wgrad(a, b, K, M, Nc + b_ins.size(1), M, Nc, b_ins=b_ins, ins_off=ins_off, defer=defer)[0]

    def blocks(m, n):  # 16-row tiles a launch computes for an m x n result (wgrad.hip: wgrad_plan)
        ta = 4 if m % 64 == 0 or m >= 512 else 2
        return -(-m // (16 * ta)) * ta * -(-n // 112)
    if blocks(Nc, M) < blocks(M, Nc):
        return backend.wgrad(b, a, K, Nc, M, Nc, M, defer=defer)[0].t()
    return backend.wgrad(a, b, K, M, Nc, M, Nc, defer=defer)[0]


_weight_streams = {}


def _train_streams():
    """What a cell's backward pass puts on a second stream (development): letters of GGNN_TRAIN_STREAMS -- "w" the weight
    gradients, "h" the grains' hidden-state gradient; default "": one stream.  Measured on the replayed cfg3 step (r6): "w"
    +0.04 ms (the kernels overlap and slow each other down by as much), "h" +-0.01 ms at cfg3 and +0.05 ms for four 40 um
    graphs (profiles/r6_train_step_experiments.txt)."""
    import os
    return os.environ.get("GGNN_TRAIN_STREAMS", "")


def _weight_stream(device):
    """The second stream of a cell's backward pass (one per device)."""
    s = _weight_streams.get(device)
    if s is None:
        s = _weight_streams[device] = torch.cuda.Stream(device=device)
    return s


class _PackedCell(torch.autograd.Function):
    """One HeteroPGCLSTM cell (heteropgclstm.py:101-183) on the packed weights, forward and backward written
    out by hand so that a cell is ~10 launches forward and ~25 backward with no autograd bookkeeping between them:

      forward   P[nt] = [x | h] Wp^T + bp             ggnn_project_batch (the inference kernel; one launch)
                agg   = sweeps(P, h)                   ggnn_period_gat_aggregate_batch (one launch)
                z     = agg W2^T (per gate, batched)   ggnn_rowgemm
                h', c' = LSTM(z + skip(P), c)          ggnn_lstm_train_forward
      backward  g_z, gP[skip], g_c                     ggnn_lstm_train_backward
                g_W2 = g_z^T agg                       ggnn_wgrad (reduction over the nodes, split over the chip)
                g_agg = g_z W2                         ggnn_rowgemm
                gP[u, u4, v], g_h (source side), g_ep  ggnn_period_gat_aggregate_backward per edge type
                g_[Wp | bp] = gP^T [x | h | 1]         ggnn_wgrad
                g_h += gP Wp[:, h columns]             ggnn_rowgemm

    Inputs (x_g, x_j, h_g, h_j, c_g, c_j, wp_g, wp_j, bp_g, bp_j, ep x 3, w2_g, w2_j) with the packed matrices as
    train_pack.packed_weights lays them out (wp [ncols, Fp + k2], w2 [G, 96, Kg]); h / c None = zero state
    (encoder, 3 gates).  x carries no gradient (data).  `bf16`: the caller ran under bf16 autocast."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x_g, x_j, h_g, h_j, c_g, c_j, wp_g, wp_j, bp_g, bp_j, ep_gj, ep_jg, ep_jj, w2_g, w2_j,
                backend, topo, einfo, layout, G, bf16=False, xs=None):
        x = {"grain": x_g.contiguous(), "joint": x_j.contiguous()}
        sees_h = h_g is not None
        h = {"grain": h_g.contiguous() if sees_h else None, "joint": h_j.contiguous() if sees_h else None}
        c = {"grain": c_g.contiguous() if sees_h else None, "joint": c_j.contiguous() if sees_h else None}
        wp, bp = {"grain": wp_g.contiguous(), "joint": wp_j.contiguous()}, {"grain": bp_g.contiguous(),
                                                                            "joint": bp_j.contiguous()}
        w2 = {"grain": w2_g.contiguous(), "joint": w2_j.contiguous()}
        ep = dict(zip(EDGE_TYPES, (ep_gj.contiguous(), ep_jg.contiguous(), ep_jj.contiguous())))
        f32 = dict(dtype=torch.float32, device=x_j.device)
        P, agg = {}, {}
        problems = []
        for nt in NODE_TYPES:
            lay, n = layout[nt], x[nt].size(0)
            P[nt] = torch.empty(n, lay.ncols, **f32)
            # under torch.autocast(bfloat16) the decoder projection -- the cell's big linear, [x | h] Wp^T -- runs as
            # bf16 operands / one MFMA product per k-step / fp32 accumulate (GGNN_PRECISION_BF16); the encoder's K <= 12
            # projection, the sweep and the softmax stay fp32
            x6 = getattr(getattr(backend, "lib", None), "ggnn_gemm_mode", lambda: 1)() == 1   # (not under GGNN_GEMM=fp32)
            prec = _lib.GGNN_PRECISION_BF16 if (bf16 and sees_h and x6) else 0
            problems.append((x[nt], lay.F, h[nt], wp[nt], bp[nt], P[nt], prec))
            # the sweeps write every aggregate and scalar column of every row; the pad columns behind them meet zero
            # columns of w2 and only have to be finite: the sweep of the node type's last edge type zeroes them (pad_n)
            agg[nt] = torch.empty(n, G * lay.Kg, **f32)
        backend.project_batch(problems)
        sweeps = []
        for et in EDGE_TYPES:
            s, d = et[0], et[-1]
            used = len(layout[d].dst_ets) * (C + 2)
            pad_n = layout[d].Kg - used if layout[d].sc_off[et] + 2 == used else 0
            sweeps.append((topo.graph.csr[et], einfo[et], P[s], P[d], h[s], ep[et], agg[d], layout[s].v_off[et],
                           layout[d].u_off.get(et, 0), layout[d].u4_off[et], layout[d].a_off[et], layout[d].Kg,
                           layout[d].sc_off[et], G, pad_n))
        backend.aggregate_batch(sweeps)
        # the weight planes of the cell's row GEMMs -- the gate GEMM of both node types now, its input gradient and the
        # hidden-state gradient in the backward pass -- packed by ONE launch (they were ten small launches per step)
        specs = [(w2[nt], layout[nt].Kg, C, G, False, bf16) for nt in NODE_TYPES] \
            + [(w2[nt], C, layout[nt].Kg, G, True, bf16) for nt in NODE_TYPES]
        if sees_h:
            specs += [(wp[nt][:, layout[nt].Fp:layout[nt].Fp + C], layout[nt].ncols, C, 1, True, bf16) for nt in NODE_TYPES]
        planes = backend.rowgemm_pack(specs) if hasattr(backend, "rowgemm_pack") else [None] * len(specs)
        z, out, updates = {}, [], []
        for k, nt in enumerate(NODE_TYPES):
            lay, n = layout[nt], x[nt].size(0)
            # the gate GEMM z_g = agg_g W2_g^T (ggnn_rowgemm: two-piece fp16 = fp32-equivalent; one bf16 product under
            # torch.autocast(bfloat16))
            z[nt] = torch.empty(G, n, C, **f32)
            backend.rowgemm(agg[nt].view(n, G, lay.Kg).transpose(0, 1), w2[nt], z[nt], lay.Kg, C, batch=G, bf16=bf16,
                            planes=planes[k])
            h_new, c_new = torch.empty(n, C, **f32), torch.empty(n, C, **f32)
            updates.append((z[nt], P[nt], lay.s_off, c[nt], h_new, c_new))
            out += [h_new, c_new]
        backend.lstm_train_forward_batch(updates, G)   # (both node types in one launch)
        saved = []
        for nt in NODE_TYPES:
            saved += [x[nt], h[nt], c[nt], wp[nt], w2[nt], P[nt], agg[nt], z[nt]]
        if xs is None:   # (a caller without the step's data rows: made here, one launch)
            xs = dict(zip(NODE_TYPES, backend.train_input_rows([(x[nt], layout[nt].F) for nt in NODE_TYPES])))
        ctx.save_for_backward(*saved, *out, *[ep[et] for et in EDGE_TYPES], *[einfo[et] for et in EDGE_TYPES],
                              *[xs[nt] for nt in NODE_TYPES])
        ctx.misc = (backend, topo, layout, G, sees_h, bf16, planes[2:])   # (the backward's planes: held until then)
        ctx.set_materialize_grads(False)
        return tuple(out)                                           # h_grain, c_grain, h_joint, c_joint

    @staticmethod
    @once_differentiable   # (a second derivative through the hand-written backward fails loudly)
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_hg, g_cg, g_hj, g_cj):
        backend, topo, layout, G, sees_h, bf16, planes = ctx.misc
        t = ctx.saved_tensors
        x, h, c, wp, w2, P, agg, z, c_new = {}, {}, {}, {}, {}, {}, {}, {}, {}
        for k, nt in enumerate(NODE_TYPES):
            x[nt], h[nt], c[nt], wp[nt], w2[nt], P[nt], agg[nt], z[nt] = t[8 * k:8 * k + 8]
            c_new[nt] = t[16 + 2 * k + 1]
        ep = dict(zip(EDGE_TYPES, t[20:23]))
        einfo = dict(zip(EDGE_TYPES, t[23:26]))
        xs = dict(zip(NODE_TYPES, t[26:28]))   # [x | 0 .. | 1 0 0 0] per node type (ggnn_train_input_rows)
        g_h_out = {"grain": g_hg, "joint": g_hj}
        g_c_out = {"grain": g_cg, "joint": g_cj}
        f32 = dict(dtype=torch.float32, device=P["joint"].device)
        gP, g_agg, g_w2, g_c, g_z, updates = {}, {}, {}, {}, {}, []
        ok = lambda g: None if g is None else g.contiguous()
        for k, nt in enumerate(NODE_TYPES):
            lay, n = layout[nt], x[nt].size(0)
            gP[nt] = torch.empty_like(P[nt])
            used = max(lay.u4_off[et] + G * 16 for et in lay.dst_ets)   # columns behind it are padding: zeroed by the update
            pad_n = lay.ncols - used
            if pad_n > C or pad_n % 4 or used % 4:                      # (not a layout packing.node_layout makes)
                gP[nt][:, used:].zero_()
                pad_n = 0
            g_z[nt] = torch.empty_like(z[nt])
            g_c[nt] = torch.empty(n, C, **f32) if sees_h else None
            updates.append((z[nt], c[nt], c_new[nt], ok(g_h_out[nt]), ok(g_c_out[nt]), g_z[nt], gP[nt], lay.s_off, g_c[nt],
                            used, pad_n))
        backend.lstm_train_backward_batch(updates, G)   # (both node types in one launch)
        # Development switch (GGNN_TRAIN_STREAMS, off by default: no gain measured): the WEIGHT gradients ("w") or the grains'
        # hidden-state gradient ("h") on a second stream beside the chain of activation gradients.  Both chains end inside
        # this function (the streams are joined before it returns: autograd sees results that are ready on its stream),
        # every tensor the second stream reads is held by this frame until then, and an episode on it starts behind an
        # event of the main stream -- so memory it allocated and the main stream has since read is not rewritten early.
        main = torch.cuda.current_stream() if P["joint"].is_cuda else None
        mode = _train_streams() if main is not None else ""
        side = _weight_stream(P["joint"].device) if mode else None

        def beside(fn, what="w"):
            if side is None or what not in mode:
                return fn()
            fork = torch.cuda.Event()
            fork.record(main)
            with torch.cuda.stream(side):
                side.wait_event(fork)
                return fn()

        # the sums over the split-K partials of the four weight gradients and over the sweeps' edge-parameter partials are
        # postponed to ONE launch at the end of this backward pass (nothing reads them before)
        red = [] if hasattr(backend, "sum_rows_batch") and "w" not in mode else None

        def gate_weight_gradients():
            for nt in NODE_TYPES:
                lay, n = layout[nt], x[nt].size(0)
                g_w2[nt] = backend.wgrad(g_z[nt], agg[nt], n, C, lay.Kg, C, G * lay.Kg, batch=G, a_bstride=n * C,
                                         b_bstride=lay.Kg, defer=red)                          # [G, 96, Kg]
        beside(gate_weight_gradients)
        for k, nt in enumerate(NODE_TYPES):
            lay, n = layout[nt], x[nt].size(0)
            g_agg[nt] = torch.empty_like(agg[nt])
            backend.rowgemm(g_z[nt], w2[nt], g_agg[nt].view(n, G, lay.Kg).transpose(0, 1), C, lay.Kg, batch=G,
                            transposed=True, bf16=bf16, planes=planes[k])                      # g_agg_g = g_z_g W2_g
        gh_src = {nt: None for nt in NODE_TYPES}
        # the per-workgroup partial sums of the three sweeps' edge-parameter gradients side by side: one reduction
        n_part = [backend.aggregate_bwd_partials(x[et[-1]].size(0)) for et in EDGE_TYPES]
        ep_part = (torch.empty if len(set(n_part)) == 1 else torch.zeros)(len(EDGE_TYPES), max(n_part), G, 3, C, **f32)
        for k, et in enumerate(EDGE_TYPES):
            s, d = et[0], et[-1]
            _, _, g_h, _ = backend.aggregate_backward(
                topo.graph.csr[et], topo.rcsr[et], topo.r_slot[et], einfo[et], P[s], P[d], h[s], ep[et], agg[d],
                g_agg[d], layout[s].v_off[et], layout[d].u_off.get(et, 0), layout[d].u4_off[et], layout[d].a_off[et],
                layout[d].Kg, layout[d].sc_off[et], G, out_p_dst=gP[d], out_p_src=gP[s], ep_partial_out=ep_part[k],
                g_h_into=gh_src[s])   # (the second sweep out of a node type adds to the first one's rows in place)
            if g_h is not None:
                gh_src[s] = g_h
        g_wp, g_bp, g_h, g_ep = {}, {}, {}, {}

        def projection_weight_gradients():
            if red is None:
                g_ep.update(zip(EDGE_TYPES, backend.sum_rows(ep_part.view(len(EDGE_TYPES), max(n_part), -1)).view(-1, G, 3, C)))
            else:
                sums = torch.empty(len(EDGE_TYPES), G * 3 * C, **f32)
                red.append((ep_part.view(len(EDGE_TYPES), max(n_part), -1), sums))
                g_ep.update(zip(EDGE_TYPES, sums.view(-1, G, 3, C)))
            for nt in NODE_TYPES:
                lay = layout[nt]
                Kp = lay.Fp + (C if sees_h else 0)                      # columns of wp: [x (F) | 0 (Fp - F) | h]
                # the other factor [x | 0 | h | 1 0 0 0] = the step's data rows with the hidden state inserted at column Fp
                g_wpb = _wgrad2d(backend, gP[nt], xs[nt], h[nt] if sees_h else None, lay.Fp, defer=red)   # [ncols, Kp + 4]
                g_wp[nt], g_bp[nt] = g_wpb[:, :Kp], g_wpb[:, Kp]
        beside(projection_weight_gradients)
        # g_h = (the sweeps' source-side gradient) + gP Wp[:, h columns]   [N, 96] -- the two node types' products side by side
        # in ONE grid (ggnn_rowgemm_pair): a wave of ggnn_rowgemm walks the whole reduction of its 16 rows, so they are 79
        # and 157 workgroups on 256 compute units (one after the other: 44 + 68 us at cfg3)
        products = [(gP[nt], wp[nt][:, layout[nt].Fp:layout[nt].Fp + C], torch.empty(x[nt].size(0), C, **f32), layout[nt].ncols,
                     C, gh_src[nt], True, bf16, planes[2 + k]) for k, nt in enumerate(NODE_TYPES)] if sees_h else []
        for nt in NODE_TYPES:
            g_h[nt] = None
        if sees_h and hasattr(backend, "rowgemm_pair") and "h" not in mode:
            g_h.update(zip(NODE_TYPES, backend.rowgemm_pair(*products)))
        elif sees_h:
            for k, (nt, (a, w, out, K, n_out, c_in, tr, b16, pl)) in enumerate(zip(NODE_TYPES, products)):
                def hidden_state_gradient(nt=nt, a=a, w=w, out=out, K=K, n_out=n_out, c_in=c_in, tr=tr, b16=b16, pl=pl):
                    g_h[nt] = backend.rowgemm(a, w, out, K, n_out, c_in=c_in, transposed=tr, bf16=b16, planes=pl)
                if k == 0:   # (mode "h", development: the grains' product on a second stream beside the joints')
                    beside(hidden_state_gradient, "h")
                else:
                    hidden_state_gradient()
        if side is not None:
            main.wait_stream(side)
        if red:
            backend.sum_rows_batch(red)
        return (None, None, g_h["grain"], g_h["joint"], g_c["grain"], g_c["joint"], g_wp["grain"], g_wp["joint"],
                g_bp["grain"], g_bp["joint"], g_ep[EDGE_TYPES[0]], g_ep[EDGE_TYPES[1]], g_ep[EDGE_TYPES[2]],
                g_w2["grain"], g_w2["joint"], None, None, None, None, None, None, None)


def cell_forward(cell, backend, topo, einfo, x, h, c, xs=None, packed=None):
    """HeteroPGCLSTM.forward (heteropgclstm.py:101-183), differentiable, in the packed formulation: the packed
    weights are assembled from the parameters by recorded torch ops, the cell itself is _PackedCell.
    h, c: dicts or None (encoder: zero state; the forget gate multiplies c = 0 and is skipped).
    (Measured and dropped, round 4: the decoder's packed weights assembled on a side stream beside the encoder cell -- a
    replayed hipGraph ran the two branches one after the other, with launch gaps between their small kernels.)"""
    gates = "ifco" if h is not None else "ico"
    G = len(gates)
    F = cell.in_channels_dict
    bf16 = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
    with torch.autocast(x["joint"].device.type, enabled=False):   # weight-sized products stay fp32
        layout, wp, bp, ep, w2 = packed if packed is not None else packed_weights(cell, gates, F, h is not None)
    hg, cg, hj, cj = _PackedCell.apply(
        x["grain"], x["joint"], None if h is None else h["grain"], None if h is None else h["joint"],
        None if c is None else c["grain"], None if c is None else c["joint"], wp["grain"], wp["joint"], bp["grain"],
        bp["joint"], ep[EDGE_TYPES[0]], ep[EDGE_TYPES[1]], ep[EDGE_TYPES[2]], w2["grain"], w2["joint"], backend, topo,
        einfo, layout, G, bf16, xs)
    return {"grain": hg, "joint": hj}, {"grain": cg, "joint": cj}


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
        # (ggnn_edge_prepare writes the zero padding records behind the E edges itself: no fill)
        einfo = alloc_einfo(graph, x["joint"].device, zero=False)
        be.edge_prepare([(graph.csr[et], _edge_attr_1d(edge_attr[et]), x[et[0]], x[et[-1]], einfo[et])
                         for et in EDGE_TYPES])
        # the data part of the weight gradients' second factor, [x | 0 .. | 1 0 0 0] per node type: one launch per step,
        # shared by the two cells' backward passes
        xs = dict(zip(NODE_TYPES, be.train_input_rows([(x[nt], model.in_channels_dict[nt]) for nt in NODE_TYPES])))
    enc, dec = model.gclstm_encoder.cell_list[0], model.gclstm_decoder.cell_list[0]
    # both cells' packed weights by one autograd function: its launches (three forward, four backward) are shared
    with torch.autocast(x["joint"].device.type, enabled=False):   # weight-sized products stay fp32
        pk_enc, pk_dec = packed_weights_of([(enc, "ico", enc.in_channels_dict, False), (dec, "ifco", dec.in_channels_dict, True)])
    h, c = cell_forward(enc, be, topo, einfo, x, None, None, xs, pk_enc)
    h, c = cell_forward(dec, be, topo, einfo, x, h, c, xs, pk_dec)
    return h, graph


class _RowLinear(torch.autograd.Function):
    """y = x W^T + b for many rows (nodes / edges) and a head of 1-3 outputs (models.py:427-433, 600-603).  The
    weight and bias gradients are reductions over all rows: one ggnn_wgrad launch ([g_y | 0]^T [x | 1 | 0]) instead
    of the BLAS call that runs such a shape on half a dozen workgroups (50-80 us at the 10k-grain graph)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias, backend):
        ctx.save_for_backward(x, weight)
        ctx.backend = backend
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    @once_differentiable   # (a second derivative through the hand-written backward fails loudly)
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        n, n_in, n_out = x.size(0), x.size(1), weight.size(0)
        g = g.contiguous()
        gp = torch.nn.functional.pad(g, (0, (-n_out) % 4))
        xin = torch.cat([x, _ones(x.device, n, 1 + (-(n_in + 1)) % 4)], 1)
        gw = ctx.backend.wgrad(gp, xin, n, gp.size(1), xin.size(1), gp.size(1), xin.size(1))[0]
        return g @ weight, gw[:n_out, :n_in], gw[:n_out, n_in], None


class _RegressorHeads(torch.autograd.Function):
    """The regressor's output heads (models.py:427-452): (h_joint, h_grain, x_grain, W_j, b_j, W_g, b_g) ->
    (y_joint, y_grain, grain_area).  Forward = the inference kernel `ggnn_heads_regressor`; backward =
    `ggnn_heads_regressor_backward` (pre-activation and hidden-state gradients in one launch) + one `ggnn_wgrad` per
    node type for the weights and biases: 12 launches each way where the recorded ops needed 32."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, h_joint, h_grain, x_grain, w_j, b_j, w_g, b_g, backend):
        h_joint, h_grain = h_joint.contiguous(), h_grain.contiguous()
        ctx.width = w_j.size(1)
        if ctx.width == C:    # [w_joint | w_grain | b_joint | b_grain] by one launch
            wb = torch.cat([w_j.reshape(-1), w_g.reshape(-1), b_j, b_g])
            w, b = wb[:4 * C].view(2, 2, C), wb[4 * C:]                   # [2, 2, 96], [4] (packing.pack_regressor_heads)
        else:                 # layer_size < 96: zero columns for the padded (exactly zero) channels of h
            w, b = torch.nn.functional.pad(torch.stack([w_j, w_g]), (0, C - ctx.width)), torch.cat([b_j, b_g])
        nj, ng = h_joint.size(0), h_grain.size(0)
        f32 = dict(dtype=torch.float32, device=h_joint.device)
        y_joint, y_grain, area = torch.empty(nj, 2, **f32), torch.empty(ng, 2, **f32), torch.empty(ng, **f32)
        backend.heads_regressor(h_joint, h_grain, x_grain, w, b, y_joint, y_grain, area)
        ctx.save_for_backward(h_joint, h_grain, w, y_joint, y_grain)
        ctx.backend = backend
        ctx.set_materialize_grads(False)
        return y_joint, y_grain, area

    @staticmethod
    @once_differentiable
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_yj, g_yg, g_area):
        h_joint, h_grain, w, y_joint, y_grain = ctx.saved_tensors
        be = ctx.backend
        ok = lambda g: None if g is None else g.contiguous()
        gpj, gpg, ghj, ghg = be.heads_regressor_backward(w, y_joint, y_grain, ok(g_yj), ok(g_yg), ok(g_area))
        rows, wd = [], ctx.width
        red = [] if hasattr(be, "sum_rows_batch") else None   # (the two products' split-K sums in one launch)
        for gp, h in ((gpj, h_joint), (gpg, h_grain)):   # [g_pre | 0]^T [h | 1 0 0 0]: weight and bias gradient in one product,
            # the second factor = the constant [1 0 0 0] rows with h inserted in front of them (read where it lies)
            gw = _wgrad2d(be, gp, _ones(h.device, h.size(0), 4), h, 0, defer=red)          # [4, 100]
            rows += [gw[0, :wd], gw[1, :wd], gw[0, C:C + 1], gw[1, C:C + 1]]
        if red:
            be.sum_rows_batch(red)
        # the four parameter gradients as views of one buffer made by one launch (row pieces of the two products): a view is
        # adopted as .grad without a copy, where four strided slices cost four copies
        flat = torch.cat(rows)
        n = 2 * wd + 2
        return ghj, ghg, None, flat[:2 * wd].view(2, wd), flat[2 * wd:n], flat[n:n + 2 * wd].view(2, wd), flat[n + 2 * wd:], None


def regressor_forward(model, x_dict, edge_index_dict, edge_attr):
    """GrainNN_regressor.forward (models.py:401-467) with autograd."""
    h, _ = encoder_decoder(model, x_dict, edge_index_dict, edge_attr)
    lin = model.linear
    y_joint, y_grain, area = _RegressorHeads.apply(h["joint"], h["grain"], x_dict["grain"].detach().contiguous(),
                                                   lin["joint"].weight, lin["joint"].bias, lin["grain"].weight,
                                                   lin["grain"].bias, default_backend())
    return {"grain": y_grain, "joint": y_joint, "grain_area": area}


class _ClassifierHeads(torch.autograd.Function):
    """The classifier's pair heads (models.py:595-609): pair = [h_j[src] | h_j[dst] | edge length], edge = tanh(lin1(pair)),
    edge_event = lin2(pair) -- (h_joint, edge length [E], W [3, 193] = rows of lin1 then lin2, b [3]) -> (edge [E, 2],
    edge_event [E]).  Forward = the inference kernels (`ggnn_heads_classifier`: six partial dots per NODE, combined per
    edge); the backward stays at the node level too: the pre-activation gradients [E, 3] are summed per source and per
    destination junction over the reverse / forward CSR (segment sums in a fixed order: reproducible, where autograd's
    backward of h[src] is an atomic scatter), and the hidden-state and weight gradients are [n, 6] products.  The recorded
    formulation it replaces materialises the [E, 193] pair matrix each way (46 MB at the 10k-grain graph)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, h, ea, W, b, backend, topo):
        h, ea = h.contiguous(), ea.contiguous().view(-1)
        et = ("joint", "connect", "joint")
        ei = topo.graph.edge_index[et]
        n, E = h.size(0), ei.size(1)
        w_node = torch.cat([W[:, :C], W[:, C:2 * C]]).contiguous()           # [6, 96]: packing.pack_classifier_heads' order
        w_edge = torch.cat([W[:, 2 * C], b]).contiguous()                    # [6]
        f32 = dict(dtype=torch.float32, device=h.device)
        node_tmp, edge_event, edge = torch.empty(n, 8, **f32), torch.empty(E, **f32), torch.empty(E, 2, **f32)
        backend.heads_classifier(h, ei, ea, w_node, w_edge, node_tmp, edge_event, edge)
        ctx.save_for_backward(h, ea, w_node, edge)
        ctx.misc = (backend, topo)
        ctx.set_materialize_grads(False)
        return edge, edge_event

    @staticmethod
    @once_differentiable
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_edge, g_event):
        h, ea, w_node, edge = ctx.saved_tensors
        backend, topo = ctx.misc
        et = ("joint", "connect", "joint")
        n, E = h.size(0), edge.size(0)
        g_pre = torch.zeros(E, 4, dtype=torch.float32, device=h.device)      # [lin1 (2) | lin2 (1) | 0]
        if g_edge is not None:
            g_pre[:, :2] = g_edge * (1.0 - edge * edge)
        if g_event is not None:
            g_pre[:, 2] = g_event
        jj = topo.jj_index()
        by_dst = g_pre[jj["slot_edge"]]                                       # forward CSR order: grouped by destination
        as_dst = torch.segment_reduce(by_dst, "sum", offsets=jj["rowptr"])    # [n, 4]
        as_src = torch.segment_reduce(by_dst[jj["r_slot"]], "sum", offsets=jj["r_rowptr"])
        g_tmp = torch.cat([as_src[:, :3], as_dst[:, :3], as_dst.new_zeros(n, 2)], 1)   # [n, 8]: the node kernel's partial dots
        g_h = g_tmp[:, :6] @ w_node
        g_w_node = backend.wgrad(g_tmp, h, n, 8, C, 8, C)[0][:6]              # [6, 96]
        g_W = torch.cat([g_w_node[:3], g_w_node[3:], (g_pre[:, :3] * ea.unsqueeze(1)).sum(0).unsqueeze(1)], 1)   # [3, 193]
        return g_h, None, g_W, g_pre[:, :3].sum(0), None, None


def classifier_forward(model, x_dict, edge_index_dict, edge_attr):
    """GrainNN_classifier.forward (models.py:572-611) with autograd."""
    h, graph = encoder_decoder(model, x_dict, edge_index_dict, edge_attr)
    et = ("joint", "connect", "joint")
    if model.out_channels == C:
        be = default_backend()
        edge, edge_event = _ClassifierHeads.apply(h["joint"], edge_attr[et], torch.cat([model.lin1.weight, model.lin2.weight]),
                                                  torch.cat([model.lin1.bias, model.lin2.bias]), be, train_topology(be, graph))
        return {"edge_event": edge_event, "edge": edge}
    src, dst = graph.edge_index[et][0], graph.edge_index[et][1]
    hj = h["joint"] if model.out_channels == C else h["joint"][:, :model.out_channels]   # (padded channels: zero)
    pair = torch.cat([hj[src], hj[dst], edge_attr[et].view(-1, 1)], -1)
    y = _RowLinear.apply(pair, torch.cat([model.lin1.weight, model.lin2.weight]),
                         torch.cat([model.lin1.bias, model.lin2.bias]), default_backend())   # both heads in one product
    return {"edge_event": y[:, 2].contiguous(), "edge": torch.tanh(y[:, :2])}


class _MaskedMSE(torch.autograd.Function):
    """100 * sum_k mean(mask_k (y_k - p_k)^2) with its gradient from ONE launch (`ggnn_masked_mse`): the recorded ops are
    ten kernels forward and twenty backward for a scalar."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, scale, backend, n_terms, *tensors):
        preds, ys, masks = tensors[:n_terms], tensors[n_terms:2 * n_terms], tensors[2 * n_terms:]
        c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=preds[0].device)
        grads = backend.masked_mse([(c(p), c(y), c(m)) for p, y, m in zip(preds, ys, masks)], scale, loss,
                                   want_grad=any(ctx.needs_input_grad[3:3 + n_terms]))
        ctx.n_terms = n_terms
        ctx.save_for_backward(*[g for g in grads if g is not None])
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss):
        grads = ctx.saved_tensors
        out = []
        if grads:   # (one multi-tensor launch for the terms' gradients)
            scaled = torch._foreach_mul(list(grads), g_loss)
            out = [g if ctx.needs_input_grad[3 + k] else None for k, g in enumerate(scaled)]
        out += [None] * (ctx.n_terms - len(out))
        return (None, None, None, *out, *([None] * (2 * ctx.n_terms)))


def regressor_loss(y_dict, pred, mask):
    """train.py:31-37 (edge_len off): 100 * (mean(mask_j (y_j - p_j)^2) + mean(mask_g (y_g - p_g)^2)).  On the GPU one
    launch computes the loss and its gradient (`_MaskedMSE`); `regressor_loss_recorded` is the same expression as recorded
    torch ops."""
    pj, pg = pred["joint"], pred["grain"]
    mj, mg = mask["joint"], mask["grain"]
    fits = lambda m, p: m.shape == p.shape or (m.dim() >= 1 and m.numel() == p.size(0))   # elementwise or one per row
    if pj.is_cuda and pj.dtype == torch.float32 and pg.dtype == torch.float32 and fits(mj, pj) and fits(mg, pg) \
            and y_dict["joint"].shape == pj.shape and y_dict["grain"].shape == pg.shape \
            and not (y_dict["joint"].requires_grad or y_dict["grain"].requires_grad or mj.requires_grad or mg.requires_grad):
        return _MaskedMSE.apply(100.0, default_backend(), 2, pj, pg, y_dict["joint"], y_dict["grain"], mj, mg)
    return regressor_loss_recorded(y_dict, pred, mask)


def regressor_loss_recorded(y_dict, pred, mask):
    return 100 * (torch.mean(mask["joint"] * (y_dict["joint"] - pred["joint"]) ** 2)
                  + torch.mean(mask["grain"] * (y_dict["grain"] - pred["grain"]) ** 2))


def classifier_loss(y_dict, pred, pos_weight: float = 1.0):
    """train.py:40-70 (edge_len off): BCE-with-logits over the labelled edges (label > -1)."""
    y, z = y_dict["edge_event"], pred["edge_event"]
    # the mean over the labelled edges as a weighted sum over all of them: no boolean indexing (its size is data: a host
    # synchronisation per step, and not capturable in a hipGraph); same value up to the order of the fp32 sum
    # An unlabelled edge is masked BEFORE the loss (train.py:44-47 indexes it away): its logit, finite or not, reaches
    # neither the value (inf * 0 = NaN otherwise) nor the gradients.
    labelled = y > -1
    keep = labelled.to(z.dtype)
    per_edge = torch.nn.functional.binary_cross_entropy_with_logits(
        torch.where(labelled, z, torch.zeros_like(z)), y.clamp(min=0).to(z.dtype),
        pos_weight=z.new_full((), float(pos_weight)), reduction="none")   # (a fill, not a host copy: capturable)
    return torch.where(labelled, per_edge, torch.zeros_like(per_edge)).sum() / keep.sum()


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (train.py:82-91: per-group learning rates, StepLR on top) with the update of ALL parameter tensors
    in ONE call (`ggnn_adam_step`: the update kernel + a one-workgroup launch that advances the step counts): the model has
    284 small parameter tensors, which torch's multi-tensor kernels take in 14 launches (~0.1 ms of a 2 ms training step
    at the 10k-grain graph).  What is fixed about a tensor (addresses of
    the parameter and its moments, size, group) sits in a table in device memory; what changes from step to step -- the
    gradients' addresses (new tensors after every backward), the groups' learning rates (a scheduler edits
    `param_groups[i]["lr"]`) -- travels in the launch's arguments.  The step counts live on the device and the call
    increments them itself, so `step()` can be captured in a hipGraph (`GraphedTrainStep`).  The groups' learning rates and
    weight decays are read from a small DEVICE array (`ggnn_adam_args.hyper`) that `sync_hyperparams()` refreshes from
    `param_groups` with an uncaptured copy whenever they changed -- step() does it outside a capture, GraphedTrainStep
    before every replay -- so a scheduler on top (train.py:91: StepLR) is followed by the replayed steps too.

    Same arithmetic as torch.optim.Adam(amsgrad=False, maximize=False): tests compare the two.  State per parameter
    (`state_dict`): "step", "exp_avg", "exp_avg_sq" -- views of three flat device buffers."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0.0 or lr < 0.0 or weight_decay < 0.0:
            raise ValueError("FusedAdam: lr, eps, weight_decay >= 0 and betas in [0, 1)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len({(tuple(g["betas"]), g["eps"]) for g in self.param_groups}) != 1:
            raise ValueError("FusedAdam: betas and eps must be the same in every parameter group (lr and weight_decay may differ)")
        self._built = None

    def _build(self):
        import ctypes
        import numpy as np
        ps = [(p, gi) for gi, g in enumerate(self.param_groups) for p in g["params"] if p.requires_grad]
        if not ps:
            raise ValueError("FusedAdam: no parameter requires a gradient")
        dev = ps[0][0].device
        for p, _ in ps:
            if not p.is_cuda or p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.GGNNError("FusedAdam: parameters must be contiguous float32 tensors on one MI355X device "
                                     "(no CPU fallback exists)")
        sizes = [p.numel() for p, _ in ps]
        offs = np.concatenate([[0], np.cumsum([-(-n // 4) * 4 for n in sizes])])   # 16-byte aligned views
        m, v = torch.zeros(int(offs[-1]), device=dev), torch.zeros(int(offs[-1]), device=dev)
        step = torch.zeros(len(ps), device=dev)   # a count per tensor, as torch keeps it
        table = np.zeros(len(ps), dtype=np.dtype([("param", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("n", "<i8"),
                                                  ("group", "<i4"), ("reserved", "<i4")]))
        assert table.itemsize == ctypes.sizeof(_lib.AdamTensor)
        for k, ((p, gi), n) in enumerate(zip(ps, sizes)):
            o = int(offs[k])
            new = {"step": step[k], "exp_avg": m[o:o + n].view_as(p), "exp_avg_sq": v[o:o + n].view_as(p)}
            old = self.state.get(p)
            if old and "exp_avg" in old:   # a rebuild (load_state_dict, add_param_group): what was there moves in
                new["exp_avg"].copy_(old["exp_avg"])
                new["exp_avg_sq"].copy_(old["exp_avg_sq"])
                new["step"].copy_(torch.as_tensor(old["step"]).to(device=dev, dtype=torch.float32))
            self.state[p] = new
            table[k] = (p.data_ptr(), m.data_ptr() + 4 * o, v.data_ptr() + 4 * o, n, gi, 0)
        dev_table = torch.from_numpy(table.view(np.uint8)).to(dev)
        launches = []   # a launch takes GGNN_ADAM_MAX_TENSORS tensors: (first tensor, count, chunk maps)
        for t0 in range(0, len(ps), _lib.GGNN_ADAM_MAX_TENSORS):
            cnt = min(_lib.GGNN_ADAM_MAX_TENSORS, len(ps) - t0)
            ct, ci = [], []
            for k in range(cnt):
                nc = -(-sizes[t0 + k] // _lib.GGNN_ADAM_CHUNK)
                ct += [k] * nc
                ci += list(range(nc))
            launches.append((t0, cnt, torch.tensor(ct, dtype=torch.int32, device=dev),
                             torch.tensor(ci, dtype=torch.int32, device=dev)))
        hyper = torch.zeros(2 * _lib.GGNN_ADAM_MAX_GROUPS, device=dev)
        hyper_host = torch.zeros(2 * _lib.GGNN_ADAM_MAX_GROUPS).pin_memory()
        self._built = dict(ps=ps, m=m, v=v, step=step, dev_table=dev_table, launches=launches, params_at=[p.data_ptr() for p, _ in ps],
                           hyper=hyper, hyper_host=hyper_host, hyper_seen=None)

    def sync_hyperparams(self):
        """Copy the groups' lr / weight_decay into the device array the update kernel reads, if they changed since the last
        copy.  Not capturable (a captured copy would replay the captured values): called by step() outside a capture and
        by GraphedTrainStep before every replay."""
        if self._built is None:
            self._build()
        b = self._built
        G = _lib.GGNN_ADAM_MAX_GROUPS
        now = tuple((float(g["lr"]), float(g["weight_decay"])) for g in self.param_groups)
        if now == b["hyper_seen"]:
            return
        if torch.cuda.is_current_stream_capturing():
            raise _lib.GGNNError("FusedAdam: lr / weight_decay changed inside a hipGraph capture: call sync_hyperparams() before it")
        torch.cuda.current_stream().synchronize()   # (an earlier copy from the pinned staging array may still be queued)
        for gi, (lr, wd) in enumerate(now):
            b["hyper_host"][gi], b["hyper_host"][G + gi] = lr, wd
        b["hyper"].copy_(b["hyper_host"], non_blocking=True)
        b["hyper_seen"] = now

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._built is None:
            self._build()
        self.sync_hyperparams()
        b = self._built
        be = default_backend()
        g0 = self.param_groups[0]
        for t0, cnt, ct, ci in b["launches"]:
            a = _lib.AdamArgs()
            a.table = b["dev_table"].data_ptr() + t0 * ctypes.sizeof(_lib.AdamTensor)
            a.chunk_tensor, a.chunk_index = ct.data_ptr(), ci.data_ptr()
            a.step = b["step"].data_ptr() + 4 * t0
            for k in range(cnt):
                p = b["ps"][t0 + k][0]
                gr = p.grad
                if p.data_ptr() != b["params_at"][t0 + k]:
                    raise _lib.GGNNError("FusedAdam: a parameter's storage moved after the first step (model.to(...) / "
                                         "p.data = ...): build a new optimizer")
                if gr is not None and (gr.dtype != torch.float32 or gr.is_sparse or not gr.is_contiguous() or gr.device != p.device):
                    raise _lib.GGNNError("FusedAdam: gradients must be dense contiguous float32 tensors on the parameters' device")
                a.grad[k] = None if gr is None else gr.data_ptr()
            for gi, g in enumerate(self.param_groups):
                a.lr[gi], a.weight_decay[gi] = g["lr"], g["weight_decay"]
            a.beta1, a.beta2, a.eps = g0["betas"][0], g0["betas"][1], g0["eps"]
            a.hyper = b["hyper"].data_ptr()
            a.n_chunks, a.n_tensors = ct.numel(), cnt
            be.adam_step(a)
        return loss

    def load_state_dict(self, state_dict):
        """The loaded moments and step counts are copied INTO the flat buffers (the device table points there)."""
        super().load_state_dict(state_dict)
        self._built = None
        with torch.no_grad():
            self._build()

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if len(self.param_groups) > _lib.GGNN_ADAM_MAX_GROUPS:
            raise ValueError("FusedAdam: at most %d parameter groups" % _lib.GGNN_ADAM_MAX_GROUPS)
        self._built = None   # rebuilt (state carried over) by the next step


_warned_eval_grad = False


def wants_autograd(model, x_dict=None) -> bool:
    """The differentiable path is taken in training mode with autograd recording (train.py:158-166).  In
    `.eval()` mode the forward is the fused inference path and its outputs carry no grad_fn -- unlike the
    reference, which is differentiable in either mode: asking for input gradients there is refused, and a
    forward in eval mode with autograd recording warns once (wrap inference in torch.no_grad(), as test.py:240
    does, or call model.train() to differentiate)."""
    global _warned_eval_grad
    if not torch.is_grad_enabled():
        return False
    if x_dict is not None and any(torch.is_tensor(v) and v.requires_grad for v in x_dict.values()):
        raise NotImplementedError("gradients with respect to x_dict are not built (the reference's training "
                                  "loop never asks for them, train.py:158-166): detach the inputs")
    if model.training:
        return any(p.requires_grad for p in model.parameters())
    if not _warned_eval_grad and any(p.requires_grad for p in model.parameters()):
        _warned_eval_grad = True
        import warnings
        warnings.warn("graingraphnn_amd: forward in .eval() mode with autograd recording takes the fused "
                      "inference path, whose outputs have no grad_fn; use torch.no_grad() for inference or "
                      "model.train() to differentiate", stacklevel=3)
    return False


class GraphedTrainStep:
    """One optimisation step of train.py:158-166 (forward, loss, backward, optimizer.step) replayed from
    a hipGraph.  The eager step is bound by the host (~270 launches: 7 ms at the 10k-grain graph, 3 ms of it on
    the GPU); captured once, a step is one graph launch (3.8 ms there, 2.2 ms for a collated batch of
    four 40 um graphs).  Static shapes and topology: the inputs are copied into the buffers of the captured
    step, so every call must bring a batch of the captured sizes on the same edge lists (what a fixed-size
    DataLoader batch of one structure family gives; anything else: run the eager step).

        step = GraphedTrainStep(model, optimizer, lambda pred, y: regressor_loss(y, pred, mask), X, EI, EA, Y)
        loss = step(X, EA, Y)            # same tensors or new values of the same shapes

    `optimizer` must be created with `capturable=True` (torch.optim.Adam(..., capturable=True)); add `fused=True`:
    the default foreach Adam is ~40 launches over the model's 284 parameter tensors and costs 1.3 ms of a replayed
    step (3.7 -> 2.3 ms at the 10k-grain graph, 2.2 -> 0.9 ms for four 40 um graphs; same update).  The .grad
    tensors are those of the captured step (its memory pool): valid after every call, replaced by none.  Warm-up
    (3 eager steps, which DO update the model) and capture (which only records) run on a side stream, PyTorch's
    whole-network capture recipe."""

    def __init__(self, model, optimizer, loss_fn, x_dict, edge_index_dict, edge_attr, y_dict, autocast_dtype=None,
                 warmup: int = 3):
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.x = {k: v.detach().clone() for k, v in x_dict.items()}
        self.ei = edge_index_dict
        self.ea = {k: v.detach().clone() for k, v in edge_attr.items()}
        self.y = {k: v.detach().clone() for k, v in y_dict.items()}
        self.autocast_dtype = autocast_dtype
        if warmup < 1:
            raise ValueError("GraphedTrainStep needs at least one eager warm-up step before the capture")
        model.train()
        # Everything the captured launches read out of the package's evictable caches (constant blocks, CSR and
        # reverse-CSR tables) is pinned here for the lifetime of the step object: the graph holds raw addresses.
        self._pins = []
        self._one = None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), _pins.collect(self._pins):
            for _ in range(warmup):
                self._eager()
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=s):
                self._loss = self._eager()
        torch.cuda.current_stream().wait_stream(s)
        self._pins = list({id(o): o for o in self._pins}.values())

    def _eager(self):
        with torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            loss = self.loss_fn(self.model(self.x, self.ei, self.ea), self.y)
        self.opt.zero_grad(set_to_none=True)   # (inside the capture the new .grad tensors come from the graph's pool)
        if self._one is None or self._one.shape != loss.shape or self._one.dtype != loss.dtype:
            self._one = torch.ones_like(loss)   # (the root gradient, made once: backward() would fill a new one per step)
        loss.backward(self._one)
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
        if hasattr(self.opt, "sync_hyperparams"):   # FusedAdam: a scheduler's new learning rates reach the replayed update
            self.opt.sync_hyperparams()
        self.graph.replay()
        return self._loss
