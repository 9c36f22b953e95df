"""Packed weights of one HeteroPGCLSTM cell for the TRAINING path, differentiable.

The cell runs in the packed formulation of the inference path (packing.py): one projection
per node type whose weight rows are, for all gates and edge types at once, the value rows of lin_value, the
key-free score rows  M = W_k^T W_q / sqrt(96)  (hidden-state part and the 16-float `u4` tails), the summed
lin_skip rows + gate bias; the relocation columns of lin_value per edge type; the gate weight (lin_l2 | its bias |
lin_edge) per node type.  Every entry of these matrices is either a SUM OF UP TO THREE reference parameters or an
entry of one of three small batched products, so the whole assembly is

    flat2  = [ all parameters | (scale K_ext) Q_ext per edge type | 0 ]        1 cat, 1 gather, 1 bmm
    packed = flat2[idx3].sum(1)                                                1 gather, 1 sum

with index tables made once per cell configuration, and its backward is the same two gathers through the inverse
tables plus 2 bmm (`_PackWeights`): ~15 launches each way where recorded torch ops needed ~120.  The tables are
produced by running the readable definition of the layout (`assemble`) on index tensors; `packed_weights_ops` runs the
same definition on the parameter values with autograd and is what tests compare the tables with.
"""
import math
from typing import Dict

import torch

from . import _lib, _pins
from torch.autograd.function import once_differentiable

from .packing import C, EDGE_TYPES, NODE_TYPES, et_key, node_layout

_KINDS = (("wq", "lin_query", "weight"), ("bq", "lin_query", "bias"), ("wk", "lin_key", "weight"),
          ("bk", "lin_key", "bias"), ("wv", "lin_value", "weight"), ("bv", "lin_value", "bias"),
          ("ws", "lin_skip", "weight"), ("bs", "lin_skip", "bias"), ("wl", "lin_l2", "weight"),
          ("bl", "lin_l2", "bias"), ("we", "lin_edge", "weight"))
SCALE = 1.0 / math.sqrt(C)   # periodGATconv.py:226

_zero_cache, _ones_cache = {}, {}


def _zeros(dev, *shape):
    """Constant zero blocks (no gradient flows into them): made once."""
    key = (str(dev), shape)
    z = _zero_cache.get(key)
    if z is None:
        if len(_zero_cache) >= 64:
            _zero_cache.clear()
        z = _zero_cache[key] = torch.zeros(*shape, dtype=torch.float32, device=dev)
    return _pins.note(z)


def _ones(dev, n, width):
    """[n, width]: a column of ones followed by zero columns (constant)."""
    key = (str(dev), n, width)
    o = _ones_cache.get(key)
    if o is None:
        if len(_ones_cache) >= 16:
            _ones_cache.clear()
        o = _ones_cache[key] = torch.zeros(n, width, dtype=torch.float32, device=dev)
        o[:, 0] = 1.0
    return _pins.note(o)


def cell_params(cell, gates):
    """The parameters the cell's forward reads, in the order of the flat buffer: per edge type, per kind, per
    gate; then the gate biases per node type.  -> (list, {key: (offset, stacked shape [G, *param.shape])}, size).
    Runs on every forward: the walk through the module tree (132 getattr chains, 0.4 ms) is done once per cell and
    gate set; afterwards the list is read from the owning modules' parameter dicts, which nn.Module updates in
    place when a Parameter object is replaced."""
    cache = cell.__dict__.setdefault("_ggnn_train_params", {})
    hit = cache.get(gates)
    if hit is not None:
        slots, table, off = hit
        return [d[k] for d, k in slots], table, off
    slots, table, off = [], {}, 0
    G = len(gates)
    for et in EDGE_TYPES:
        for kind, lin, wb in _KINDS:
            owners = [getattr(getattr(cell, "conv_" + g).convs[et_key(et)], lin) for g in gates]
            first = owners[0]._parameters[wb]
            table[(et, kind)] = (off, (G,) + tuple(first.shape))
            off += G * first.numel()
            slots += [(o._parameters, wb) for o in owners]
    for nt in NODE_TYPES:
        owners = [getattr(cell, "b_" + g) for g in gates]
        width = owners[0]._parameters[nt].numel()       # layer_size (96 in the shipped models)
        table[("b", nt)] = (off, (G, width))
        off += G * width
        slots += [(o._parameters, nt) for o in owners]
    cache[gates] = (slots, table, off)
    return [d[k] for d, k in slots], table, off


def kq_operands(get, F, k2, et):
    """K_ext [G, Fs + k2 + 2, 96] = rows (W_k^T, b_k, w_edge), Q_ext [G, 96, Fd + k2 + 1] = [W_q | b_q]: their
    product (times 1/sqrt(96)) holds every key-free score coefficient of the edge type (the encoder cuts the
    hidden-state columns off the weights)."""
    Fs, Fd = F[et[0]], F[et[-1]]
    wq, wk = get(et, "wq")[:, :, :Fd + k2], get(et, "wk")[:, :, :Fs + k2]
    K = torch.cat([wk.transpose(1, 2), get(et, "bk").unsqueeze(1), get(et, "we").squeeze(-1).unsqueeze(1)], 1)
    Q = torch.cat([wq, get(et, "bq").unsqueeze(-1)], 2)
    return K, Q


def assemble(get, get_s, mr, zeros, sum_of, F, G, sees_h):
    """The layout, on values or on indices.  get(et, kind): stacked parameters [G, *shape] that are the single
    source of an entry; get_s(et, kind) / get_s("b", nt): the same for the terms of a sum; mr(et): the product block
    [G, Fs + k2 + 2, Fd + k2 + 1]; zeros(*shape); sum_of(list): elementwise sum (HeteroConv aggr='sum').
    -> (layout, {nt: W [ncols, Fp + k2]}, {nt: b [ncols]}, {et: ep [G, 3, 96]}, {nt: w2 [G, 96, Kg]})."""
    k2 = C if sees_h else 0
    layout, wp, bp, ep, w2 = {}, {}, {}, {}, {}
    for nt in NODE_TYPES:
        lay = node_layout(nt, F[nt], G, EDGE_TYPES, True, sees_h, True)
        Fn, Fp, D = F[nt], lay.Fp, F[nt] + k2
        blocks = []                                                  # rows of [W | b]: [n, D + 1]
        for et in lay.src_ets:                                       # value rows, relocation columns zeroed
            wvr = get(et, "wv")[:, :, 3:D]
            blocks.append(torch.cat([zeros(G, C, 3), wvr, get(et, "bv").unsqueeze(-1)], 2).reshape(G * C, D + 1))
        if sees_h:
            for et in lay.dst_ets:                                   # hidden-state part of u
                Fs = F[et[0]]
                blocks.append(mr(et)[:, Fs:Fs + k2].reshape(G * C, D + 1))
        ws = sum_of([get_s(et, "ws")[:, :, :D] for et in lay.dst_ets])  # summed skip rows + gate bias
        bs = sum_of([get_s(et, "bs") for et in lay.dst_ets] + [get_s("b", nt)])
        blocks.append(torch.cat([ws, bs.unsqueeze(-1)], 2).reshape(G * C, D + 1))
        for et in lay.dst_ets:                                       # u4 tails: feature part of u, s1, s2 rows
            Fs, m = F[et[0]], mr(et)
            tail = torch.cat([m[:, :Fs], zeros(G, 12 - Fs, D + 1), m[:, Fs + k2:Fs + k2 + 2], zeros(G, 2, D + 1)], 1)
            blocks.append(tail.reshape(G * 16, D + 1))
        n_rows = sum(b.size(0) for b in blocks)
        blocks.append(zeros(lay.ncols - n_rows, D + 1))
        rows = torch.cat(blocks)
        wp[nt] = torch.cat([rows[:, :Fn], zeros(lay.ncols, Fp - Fn), rows[:, Fn:D]], 1)   # ggnn_project's input order
        bp[nt] = rows[:, D]
        layout[nt] = lay
        n_in = len(lay.dst_ets)
        w2[nt] = torch.cat([get(et, "wl") for et in lay.dst_ets]
                           + [torch.cat([get(et, "bl").unsqueeze(-1), get(et, "we")], 2) for et in lay.dst_ets]
                           + [zeros(G, C, lay.Kg - n_in * (C + 2))], 2)   # [G, 96, Kg]: agg's pad columns meet zeros
    for et in EDGE_TYPES:
        ep[et] = get(et, "wv")[:, :, :3].transpose(1, 2)              # [G, 3, 96]
    return layout, wp, bp, ep, w2


def _outputs_in_order(wp, bp, ep, w2):
    return [wp["grain"], wp["joint"], bp["grain"], bp["joint"], ep[EDGE_TYPES[0]], ep[EDGE_TYPES[1]],
            ep[EDGE_TYPES[2]], w2["grain"], w2["joint"]]


def packed_weights_ops(cell, gates, F, sees_h):
    """The readable definition: `assemble` on the parameter values, recorded by autograd (~120 launches each way)."""
    G, k2 = len(gates), (C if sees_h else 0)
    plist, table, _ = cell_params(cell, gates)
    flat = torch.cat([p.reshape(-1) for p in plist])
    get = lambda *key: flat[table[key][0]:table[key][0] + math.prod(table[key][1])].view(table[key][1])
    dev = flat.device
    prod = {}

    def mr(et):
        if et not in prod:
            K, Q = kq_operands(get, F, k2, et)
            prod[et] = torch.bmm(K * SCALE, Q)
        return prod[et]
    layout, wp, bp, ep, w2 = assemble(get, get, mr, lambda *s: _zeros(dev, *s), lambda ts: sum(ts), F, G, sees_h)
    return layout, wp, bp, {et: v.contiguous() for et, v in ep.items()}, w2


def _widen(t, key, c, F):
    """Index tensor of a stacked parameter [G, *shape] of a layer_size-c cell -> the shape it has at layer_size 96, the
    padding reading slot -1 (the plan's zero slot): output rows c -> 96, the hidden-state part of the input columns
    [F | c] -> [F | 96] (packing.padded_cell is the same padding on values, for the inference path)."""
    et, kind = key
    pad = lambda x, dim, lead=0: torch.cat(
        [x, torch.full(x.shape[:dim] + (lead + C - x.size(dim),) + x.shape[dim + 1:], -1, dtype=x.dtype)], dim)
    if et == "b" or kind in ("bq", "bk", "bv", "bs", "bl"):
        return pad(t, 1)                                           # [G, c] -> [G, 96]
    t = pad(t, 1)                                                  # output rows
    if kind == "wl":
        return pad(t, 2)                                           # [G, 96, c] -> [G, 96, 96]
    if kind == "we":
        return t                                                   # [G, 96, 1]
    Fin = F[et[0]] if kind in ("wk", "wv") else F[et[-1]]          # (wq, ws: destination features)
    return pad(t, 2, Fin)                                          # [G, 96, F + c] -> [G, 96, F + 96]


class PackPlan:
    """Index tables of one cell configuration (feature widths, gates, sees_h, layer_size), built on the CPU once.
    layer_size c < 96 (parameters.py:19): the tables address the cell's c-wide parameters directly and read the zero slot
    wherever the 96-wide layout has a padded row or hidden-state column -- the packed matrices come out zero-padded (the
    padded channels stay exactly zero through a cell, packing.py), at no run-time cost; the score scale is 1 / sqrt(c)."""

    def __init__(self, cell, gates, F, sees_h, device):
        G, k2 = len(gates), (C if sees_h else 0)
        c = getattr(cell, "out_channels", C)
        plist, table, n_flat = cell_params(cell, gates)
        self.shapes = [tuple(p.shape) for p in plist]
        self.sizes = [p.numel() for p in plist]
        self.n_flat = n_flat

        def idx_of(*key):   # (padding = -1 until the zero slot's index is known)
            t = torch.arange(table[key][0], table[key][0] + math.prod(table[key][1])).view(table[key][1])
            return t if c == C else _widen(t, key, c, F)
        # the products, batched over the three edge types: operands padded to common shapes (pad entries read the
        # zero slot), one bmm [3 G, r, 96] x [3 G, 96, c]; coefficient = the 1/sqrt(96) on the K side
        ops = [kq_operands(idx_of, F, k2, et) for et in EDGE_TYPES]
        r_max, c_max = max(K.size(1) for K, _ in ops), max(Q.size(2) for _, Q in ops)
        nE = len(EDGE_TYPES)
        n_mr = nE * G * r_max * c_max
        self.zero, self.n_flat2 = n_flat + n_mr, n_flat + n_mr + 1
        Z = self.zero
        K_all = torch.full((nE * G, r_max, C), Z)
        Q_all = torch.full((nE * G, C, c_max), Z)
        mr_all = torch.arange(n_flat, n_flat + n_mr).view(nE * G, r_max, c_max)
        mr_idx = {}
        for e, (et, (K, Q)) in enumerate(zip(EDGE_TYPES, ops)):
            K_all[e * G:(e + 1) * G, :K.size(1)] = K
            Q_all[e * G:(e + 1) * G, :, :Q.size(2)] = Q
            mr_idx[et] = mr_all[e * G:(e + 1) * G, :K.size(1), :Q.size(2)]
        self.k_shape, self.q_shape, self.mr_shape = tuple(K_all.shape), tuple(Q_all.shape), tuple(mr_all.shape)
        self.n_k, self.n_kq = K_all.numel(), K_all.numel() + Q_all.numel()
        K_all[K_all < 0] = Z
        Q_all[Q_all < 0] = Z
        kq_idx = torch.cat([K_all.reshape(-1), Q_all.reshape(-1)])
        self.k_coef = 1.0 / math.sqrt(c)                                                                       # periodGATconv.py:226
        kq_coef = torch.cat([torch.full((K_all.numel(),), self.k_coef), torch.ones(Q_all.numel())])
        blank = lambda *key: torch.full_like(idx_of(*key), Z)
        fill = lambda *shape: torch.full(shape, Z)
        layers = []
        for layer in range(3):
            # term l of a sum sits on layer l; every other entry has its single source on layer 0 and reads the
            # zero slot on the upper layers
            sum_of = lambda ts, l=layer: ts[l] if l < len(ts) else torch.full_like(ts[0], Z)
            if layer == 0:
                layers.append(assemble(idx_of, idx_of, lambda et: mr_idx[et], fill, sum_of, F, G, sees_h))
            else:
                layers.append(assemble(blank, idx_of, lambda et: torch.full_like(mr_idx[et], Z), fill, sum_of, F, G,
                                       sees_h))
        self.layout = layers[0][0]
        outs = [_outputs_in_order(*lay[1:]) for lay in layers]
        self.out_shapes = [tuple(t.shape) for t in outs[0]]
        self.out_sizes = [t.numel() for t in outs[0]]
        idx3 = torch.stack([torch.cat([t.reshape(-1) for t in o]) for o in outs], 1)      # [n_packed, 3]
        idx3[idx3 < 0] = Z
        while idx3.size(1) > 1 and bool((idx3[:, -1] == Z).all()):
            idx3 = idx3[:, :-1]
        self.n_packed = idx3.size(0)
        # inverse tables: which packed entries read a flat2 element / which operand entry reads a parameter
        inv = _inverse(idx3, self.n_flat2, Z, self.n_packed)
        inv_kq = _inverse(kq_idx.view(-1, 1), n_flat, Z, self.n_kq)
        to = lambda t: t.to(device)
        # the same two tables for the device side reading every parameter where it lies (ggnn_pack_args.params): an entry
        # below n_flat becomes ((tensor + 1) << 40) | offset inside the tensor
        starts = torch.cumsum(torch.tensor([0] + self.sizes), 0)

        def encoded(idx):
            t = torch.searchsorted(starts, idx.clamp(max=n_flat - 1), right=True) - 1
            return torch.where(idx < n_flat, ((t + 1) << _lib.GGNN_PACK_TENSOR_SHIFT) | (idx - starts[t]), idx)
        self.idx3_enc, self.kq_idx_enc = to(encoded(idx3)), to(encoded(kq_idx))
        self._ptables = {}
        self.idx3, self.kq_idx, self.kq_coef = to(idx3), to(kq_idx), to(kq_coef)
        self.inv, self.inv_kq = to(inv), to(inv_kq)


def _param_table(plan, params):
    """DEVICE int64 array of the parameter tensors' addresses (ggnn_pack_args.params), made when the set of addresses is
    new (model.to(...), a fresh model on the same plan) -- an upload, so not inside a hipGraph capture: the eager warm-up
    steps before a capture have made it."""
    key = tuple(p.data_ptr() for p in params)
    t = plan._ptables.get(key)
    if t is None:
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.GGNNError("training path: parameters must be contiguous float32 tensors")
        if torch.cuda.is_current_stream_capturing():
            raise _lib.GGNNError("training path: a parameter's storage moved between the warm-up steps and the hipGraph capture")
        if len(plan._ptables) >= 8:
            plan._ptables.pop(next(iter(plan._ptables)))
        t = plan._ptables[key] = torch.tensor(key, dtype=torch.int64).to(params[0].device)
    return _pins.note(t)


def _inverse(idx, n_src, skip, n_dst):
    """idx [n_dst, L] (entries == skip ignored) -> inv [n_src, m]: the positions (row of idx) that read each source
    element, padded with n_dst (a zero slot behind the gathered vector)."""
    rows = torch.arange(idx.size(0)).unsqueeze(1).expand_as(idx).reshape(-1)
    flat = idx.reshape(-1)
    keep = flat != skip
    src, dst = flat[keep], rows[keep]
    order = torch.argsort(src, stable=True)
    src, dst = src[order], dst[order]
    counts = torch.bincount(src, minlength=n_src)
    m = max(int(counts.max()) if counts.numel() else 1, 1)
    start = torch.cumsum(counts, 0) - counts
    rank = torch.arange(src.numel()) - start[src]
    inv = torch.full((n_src, m), n_dst, dtype=torch.int64)
    inv[src, rank] = dst
    return inv


_plans: Dict[tuple, PackPlan] = {}


def pack_plan(cell, gates, F, sees_h, device) -> PackPlan:
    key = (tuple(sorted(F.items())), gates, sees_h, str(device), getattr(cell, "out_channels", C))
    p = _plans.get(key)
    if p is None:
        p = _plans[key] = PackPlan(cell, gates, F, sees_h, device)
    return p


class _PackWeights(torch.autograd.Function):
    """(plans, counts, *parameters) -> the nine packed matrices of every cell (views of one buffer per cell).  plans: one
    PackPlan per cell (the encoder's and the decoder's of a step share every launch: ggnn_pack_weights_batch); counts[k] =
    (n_used, n_all): cell k's parameters come next in `parameters`, the first n_used of them in the plan's order, the others
    are read by the reference but contribute nothing (the encoder's forget gate: f * c with c = 0): zero gradient."""

    @staticmethod
    def forward(ctx, plans, counts, *params):
        dev = params[0].device
        ctx.plans, ctx.counts = plans, counts
        ctx.set_materialize_grads(False)
        cells, at = [], 0
        for plan, (n_used, n_all) in zip(plans, counts):
            cells.append((plan, params[at:at + n_used], [tuple(p.shape) for p in params[at + n_used:at + n_all]]))
            at += n_all
        ctx.unused_shapes = [c[2] for c in cells]
        ctx.hip = dev.type == "cuda"
        flat2s = [torch.empty(plan.n_flat2, dtype=torch.float32, device=dev) for plan, _, _ in cells]
        if ctx.hip:
            # the device side in three launches for all cells (csrc/pack.hip): the products' operands and the products
            # straight from the parameters through the index tables, then the gather-sum -- where the recorded ops below are
            # a cat, a fill, two gathers, a multiply, a library GEMM and a reduction per cell.  The parameters are read where
            # they lie (a device table of their addresses): no concatenated copy.
            from .backend import default_backend
            packed = [torch.empty(plan.n_packed, dtype=torch.float32, device=dev) for plan, _, _ in cells]
            kqs = [torch.empty(plan.n_kq, dtype=torch.float32, device=dev) for plan, _, _ in cells]
            default_backend().pack_weights_batch([(plan, f2, kq, pk, _param_table(plan, used))
                                                  for (plan, used, _), f2, kq, pk in zip(cells, flat2s, kqs, packed)])
            ctx.flat2s = flat2s   # (only their sizes and the zero slot's index matter to the backward: kept for the argument check)
        else:
            packed, kqs = [], []
            for (plan, used, _), flat2 in zip(cells, flat2s):
                torch.cat([p.reshape(-1) for p in used], out=flat2[:plan.n_flat])
                flat2[plan.zero:].zero_()
                kq = flat2[plan.kq_idx] * plan.kq_coef
                torch.bmm(kq[:plan.n_k].view(plan.k_shape), kq[plan.n_k:].view(plan.q_shape),
                          out=flat2[plan.n_flat:plan.zero].view(plan.mr_shape))
                packed.append(flat2[plan.idx3].sum(1) if plan.idx3.size(1) > 1 else flat2[plan.idx3[:, 0]])
                kqs.append(kq)
        ctx.save_for_backward(*kqs)
        outs = []
        for (plan, _, _), pk in zip(cells, packed):
            outs += [o.view(sh) for o, sh in zip(torch.split(pk, plan.out_sizes), plan.out_shapes)]
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        kqs = ctx.saved_tensors
        dev = kqs[0].device
        n_out = len(grads) // len(ctx.plans)
        f32 = dict(dtype=torch.float32, device=dev)
        if ctx.hip:   # four launches for all cells (csrc/pack.hip) instead of ~12 recorded ops per cell, two of them library GEMMs
            from .backend import default_backend
            work, result = [], []
            for k, (plan, kq, flat2, unused) in enumerate(zip(ctx.plans, kqs, ctx.flat2s, ctx.unused_shapes)):
                # one buffer for every parameter's gradient -- the last launch writes the used parameters' part and zeros
                # behind it (the encoder's forget gate) -- handed out as VIEWS: AccumulateGrad adopts a view as .grad without
                # a copy (it is the only reference), where round 5 paid three multi-tensor launches per cell for separate
                # tensors
                unused_sizes = [math.prod(sh) for sh in unused]
                g_flat = torch.empty(plan.n_flat + sum(unused_sizes), **f32)
                # (the projection's weight and bias gradients arrive as column blocks of one [ncols, K + 1] product: read in place)
                gs = [None if g is None else (g if g.is_contiguous() or g.dim() <= 2 else g.contiguous())
                      for g in grads[k * n_out:(k + 1) * n_out]]
                work.append((plan, flat2, kq, gs, torch.empty(plan.n_flat2, **f32), torch.empty(plan.n_kq, **f32), g_flat))
                pieces = torch.split(g_flat, plan.sizes + unused_sizes)
                result += [p.view(sh) for p, sh in zip(pieces, plan.shapes + list(unused))]
            default_backend().pack_weights_backward_batch(work)
            return (None, None, *result)
        result = []
        for k, (plan, kq, unused) in enumerate(zip(ctx.plans, kqs, ctx.unused_shapes)):
            cell_grads = grads[k * n_out:(k + 1) * n_out]
            g_packed = torch.empty(plan.n_packed + 1, **f32)
            slots = [s.view(sh) for s, sh in zip(torch.split(g_packed[:plan.n_packed], plan.out_sizes), plan.out_shapes)]
            have = [(s, g) for s, g in zip(slots, cell_grads) if g is not None]
            if have:
                torch._foreach_copy_([s for s, _ in have], [g for _, g in have])
            for s, g in zip(slots, cell_grads):
                if g is None:
                    s.zero_()
            g_packed[plan.n_packed:].zero_()
            g_flat2 = g_packed[plan.inv].sum(1) if plan.inv.size(1) > 1 else g_packed[plan.inv[:, 0]]
            g_kq = torch.empty(plan.n_kq + 1, **f32)
            K, Q = kq[:plan.n_k].view(plan.k_shape), kq[plan.n_k:].view(plan.q_shape)
            g_m = g_flat2[plan.n_flat:plan.zero].view(plan.mr_shape)          # (pad entries: nobody reads them -> zero)
            torch.bmm(g_m, Q.transpose(1, 2), out=g_kq[:plan.n_k].view(plan.k_shape))
            torch.bmm(K.transpose(1, 2), g_m, out=g_kq[plan.n_k:plan.n_kq].view(plan.q_shape))
            g_kq[:plan.n_kq].mul_(plan.kq_coef)
            g_kq[plan.n_kq:].zero_()
            g_via_kq = g_kq[plan.inv_kq].sum(1) if plan.inv_kq.size(1) > 1 else g_kq[plan.inv_kq[:, 0]]
            g_flat = g_flat2[:plan.n_flat] + g_via_kq
            # every parameter gets a fresh tensor made by ONE multi-tensor op (a functional foreach allocates its results
            # on the C++ side): AccumulateGrad adopts it as .grad without a kernel
            result += torch._foreach_mul([p.view(sh) for p, sh in zip(torch.split(g_flat, plan.sizes), plan.shapes)], 1.0)
            zeros = [g_flat.new_empty(sh) for sh in unused]
            if zeros:
                torch._foreach_zero_(zeros)
            result += zeros
        return (None, None, *result)


def _forget_gate_params(cell):
    slots = cell.__dict__.get("_ggnn_train_unused")
    if slots is None:
        mods = list(cell.conv_f.modules()) + [cell.b_f]
        slots = cell.__dict__["_ggnn_train_unused"] = [(m._parameters, k) for m in mods for k in m._parameters]
    return [d[k] for d, k in slots if d[k] is not None]


def _cell_inputs(cell, gates, F, sees_h):
    plist, _, _ = cell_params(cell, gates)
    # Encoder: f * c with c = 0.  The reference still runs conv_f, so its parameters receive an exactly zero
    # gradient; they get one here too, which also keeps DistributedDataParallel(model, device_ids=[rank])
    # (dist_train.py:82) usable as written.
    unused = [] if "f" in gates else _forget_gate_params(cell)
    plan = pack_plan(cell, gates, dict(F), sees_h, plist[0].device)
    return plan, plist, unused


def _as_dicts(plan, o):
    nt_g, nt_j = "grain", "joint"
    return (plan.layout, {nt_g: o[0], nt_j: o[1]}, {nt_g: o[2], nt_j: o[3]}, dict(zip(EDGE_TYPES, o[4:7])),
            {nt_g: o[7], nt_j: o[8]})


def packed_weights(cell, gates, F, sees_h):
    """-> (layout, wp {nt: [ncols, Fp + k2]}, bp {nt: [ncols]}, ep {et: [G, 3, 96]}, w2 {nt: [G, 96, Kg]}) of the
    cell, differentiable with respect to every parameter the reference's forward reads."""
    plan, plist, unused = _cell_inputs(cell, gates, F, sees_h)
    return _as_dicts(plan, _PackWeights.apply((plan,), ((len(plist), len(plist) + len(unused)),), *plist, *unused))


def packed_weights_of(cells):
    """`packed_weights` for several cells at once, [(cell, gates, F, sees_h)] -> [its result per cell]: one autograd function
    whose launches all cells share (the encoder's and the decoder's packing of a training step)."""
    ins = [_cell_inputs(*c) for c in cells]
    o = _PackWeights.apply(tuple(plan for plan, _, _ in ins), tuple((len(pl), len(pl) + len(un)) for _, pl, un in ins),
                           *[p for _, pl, un in ins for p in (*pl, *un)])
    n = len(o) // len(ins)
    return [_as_dicts(plan, o[k * n:(k + 1) * n]) for k, (plan, _, _) in enumerate(ins)]
