"""Parameter holders with the reference's module tree (so `.pt` state_dicts load unchanged)
whose `forward` runs on the HIP kernels.

Reference classes mirrored (constructor arguments, attribute names, forward signatures):
  PeriodConv     periodGATconv.py:15-236   (heads=1, concat, edge_dim=1, no beta, dropout 0)
  HeteroConv     torch_geometric 2.1.0 as used at heteropgclstm.py:49-82 (ModuleDict `convs`,
                 keys '__'.join(edge_type), aggr='sum')
  HeteroPGCLSTM  heteropgclstm.py:18-183
  SeqGCLSTM      models.py:151-301 (layers == 1, seq_len == 1)
"""
import math
from collections import OrderedDict
from typing import Dict

import torch
import torch.nn as nn

from . import _lib
from .backend import default_backend
from .engine import _check_x, _edge_attr_1d, graph_for, prepare_edges, run_cell
from .packing import C, EDGE_TYPES, NODE_TYPES, bf16_planes, et_key, pack_cell, pack_conv, padded_cell, padded_conv


def _param_version(module: nn.Module):
    """(storage, version) of every parameter: changes when a parameter is updated in place, moved
    or replaced.  Runs on every forward, so it walks the per-module parameter dicts collected
    once (`module.parameters()` re-traverses the module tree: 0.3 ms per cell) -- the dicts
    are updated in place by nn.Module when a Parameter object is replaced."""
    dicts = module.__dict__.get("_ggnn_param_dicts")
    if dicts is None:
        dicts = [m._parameters for m in module.modules() if m._parameters]
        module.__dict__["_ggnn_param_dicts"] = dicts
    return tuple((p.data_ptr(), p._version) for d in dicts for p in d.values() if p is not None)


def _param_version_sample(module: nn.Module):
    """The same for a SAMPLE of the parameters (every 24th tensor, the first and the last included): what an optimizer step,
    load_state_dict or .to() changes, it changes in all of them -- a check per step of a loop can afford this (a few
    microseconds where the full walk is ~0.07 ms per model), with the full walk every 16th step."""
    dicts = module.__dict__.get("_ggnn_param_dicts")
    if dicts is None:
        _param_version(module)
        dicts = module.__dict__["_ggnn_param_dicts"]
    sample = module.__dict__.get("_ggnn_param_sample")
    if sample is None or sample[0] != sum(len(d) for d in dicts):
        slots = [(d, k) for d in dicts for k in d]
        picked = slots[::24] + slots[-1:]
        sample = module.__dict__["_ggnn_param_sample"] = (len(slots), picked)
    return tuple((d[k].data_ptr(), d[k]._version) for d, k in sample[1] if d[k] is not None)


class PeriodConv(nn.Module):
    """periodGATconv.py:90-154.  `in_channels` = (source width, destination width)."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        if not 1 <= out_channels <= C:
            raise NotImplementedError(f"the HIP path is built for layer_size <= {C} (got {out_channels})")
        self.in_channels = tuple(in_channels)
        self.out_channels = out_channels
        self.lin_key = nn.Linear(in_channels[0], out_channels)
        self.lin_query = nn.Linear(in_channels[1], out_channels)
        self.lin_value = nn.Linear(in_channels[0], out_channels)
        self.lin_l2 = nn.Linear(out_channels, out_channels, bias=bias)
        self.lin_edge = nn.Linear(1, out_channels, bias=False)
        self.lin_skip = nn.Linear(in_channels[1], out_channels, bias=bias)

    @torch.no_grad()
    def forward(self, x, edge_index, edge_attr=None):
        """x: Tensor or (x_src, x_dst), rows = cat[features, h] (width F + layer_size) or bare
        features (width F <= 12); edge_index [2, E] int64; edge_attr [E, 1].  -> [N_dst, layer_size].
        (layer_size < 96: on the 96-wide kernels with zero-padded parameters and hidden columns, packing.padded_conv.)"""
        be = default_backend()
        x_src, x_dst = (x, x) if isinstance(x, torch.Tensor) else x
        Ds, Dd = self.in_channels
        c = self.out_channels
        k2 = C if Ds > 12 else 0
        Fs, Fd = (Ds - c, Dd - c) if k2 else (Ds, Dd)
        wps, bps, wpd, bpd, ep, w2 = pack_conv(self if c == C else padded_conv(self, c), Fs, Fd, k2)
        dev = x_src.device
        xs, xd = x_src[:, :Fs].contiguous(), x_dst[:, :Fd].contiguous()
        widen = lambda h: h.contiguous() if c == C else torch.nn.functional.pad(h, (0, C - c)).contiguous()
        hs = widen(x_src[:, Fs:]) if k2 else None
        hd = widen(x_dst[:, Fd:]) if k2 else None
        ps = torch.empty(x_src.size(0), C, device=dev)          # [V]
        pd = torch.empty(x_dst.size(0), 3 * C, device=dev)      # [u_h | S | u4, zero rows]
        be.project(xs, Fs, hs, wps, bps, ps)
        be.project(xd, Fd, hd, wpd, bpd, pd)
        csr = be.build_csr(edge_index, x_src.size(0), x_dst.size(0))
        einfo = torch.zeros(edge_index.size(1) + _lib.GGNN_UNIT_EDGES, _lib.GGNN_EINFO_ROW, device=dev)
        be.edge_prepare([(csr, _edge_attr_1d(edge_attr), xs, xd, einfo)])
        agg = torch.zeros(x_dst.size(0), 100, device=dev)
        be.aggregate(csr, einfo, ps, pd, hs, ep, agg, 0, 0, 2 * C, 0, 100, C, 1)
        out = torch.empty(x_dst.size(0), C, device=dev)
        be.lstm_epilogue(agg, w2, pd, C, None, None, None, out, 1, _lib.MODE_RAW, bf16_planes(w2))
        return out if c == C else out[:, :c].contiguous()


class HeteroConv(nn.Module):
    def __init__(self, convs: Dict[tuple, nn.Module]):
        super().__init__()
        self.convs = nn.ModuleDict(OrderedDict((et_key(et), m) for et, m in convs.items()))


class HeteroPGCLSTM(nn.Module):
    """heteropgclstm.py:30-99: conv_{i,f,c,o} (HeteroConv of 3 PeriodConv on cat[x, h]) and
    per-node-type gate biases b_{i,f,c,o} [1, C] (glorot)."""

    def __init__(self, in_channels_dict, out_channels, metadata, bias=True, device="cpu"):
        super().__init__()
        if not 1 <= out_channels <= C:
            raise NotImplementedError(f"the HIP path is built for layer_size <= {C} (got {out_channels})")
        self.in_channels_dict = dict(in_channels_dict)
        self.out_channels = out_channels
        self.metadata = metadata
        edge_types = [tuple(et) for et in metadata[1]]
        if sorted(edge_types) != sorted(EDGE_TYPES) or sorted(in_channels_dict) != sorted(NODE_TYPES):
            raise NotImplementedError(
                f"graph schema must be node types {NODE_TYPES} and edge types {EDGE_TYPES}")
        for nt, F in in_channels_dict.items():
            if not 3 <= F <= 12:
                raise NotImplementedError(f"{nt}: feature width {F} outside the supported 3..12")
        self.edge_types = edge_types
        for g in "ifco":
            c = out_channels
            setattr(self, "conv_" + g, HeteroConv({
                et: PeriodConv((in_channels_dict[et[0]] + c, in_channels_dict[et[-1]] + c), c, bias)
                for et in edge_types}))
            b = nn.ParameterDict({nt: nn.Parameter(torch.empty(1, c)) for nt in in_channels_dict})
            for p in b.values():
                bound = math.sqrt(6.0 / (p.size(-2) + p.size(-1)))  # glorot, heteropgclstm.py:90-99
                nn.init.uniform_(p, -bound, bound)
            setattr(self, "b_" + g, b)
        self._packed = {}

    def packed(self, encoder: bool, live=NODE_TYPES):
        """Fused device weights, re-packed when parameters were replaced or updated.  `live`:
        node types whose new (h, c) is read afterwards (dead ones are not computed)."""
        ver = _param_version(self)
        key = (encoder, tuple(live))
        hit = self._packed.get(key)
        if hit is None or hit[0] != ver:
            # (layer_size < 96: the 96-wide kernels on zero-padded parameters, packing.padded_cell)
            src = self if self.out_channels == C else padded_cell(self, self.out_channels)
            hit = (ver, pack_cell(src, self.in_channels_dict, encoder, live=tuple(live)))
            self._packed[key] = hit
        return hit[1]

    @torch.no_grad()
    def forward(self, x_dict, edge_index_dict, edge_attr=None, h_dict=None, c_dict=None):
        be = default_backend()
        n_nodes = {nt: x_dict[nt].size(0) for nt in NODE_TYPES}
        for nt in NODE_TYPES:
            _check_x(x_dict[nt], self.in_channels_dict[nt], nt)
        graph = graph_for(be, edge_index_dict, n_nodes)
        encoder = h_dict is None
        pc = self.packed(encoder)
        dev = x_dict["joint"].device
        f32 = dict(dtype=torch.float32, device=dev)
        proj = {nt: torch.empty(n_nodes[nt], pc.layout[nt].ncols, **f32) for nt in NODE_TYPES}
        agg = {nt: torch.zeros(n_nodes[nt], pc.G * pc.layout[nt].Kg, **f32) for nt in NODE_TYPES}
        h_out = {nt: torch.empty(n_nodes[nt], C, **f32) for nt in NODE_TYPES}
        c_out = {nt: torch.empty(n_nodes[nt], C, **f32) for nt in NODE_TYPES}
        ea = {et: _edge_attr_1d(edge_attr[et]) for et in EDGE_TYPES}
        cw = self.out_channels
        widen = lambda t: t.contiguous() if cw == C else torch.nn.functional.pad(t, (0, C - cw)).contiguous()
        if not encoder:
            if c_dict is None:
                c_dict = {nt: torch.zeros(n_nodes[nt], cw, **f32) for nt in NODE_TYPES}
            h_dict = {nt: widen(h_dict[nt]) for nt in NODE_TYPES}
            c_dict = {nt: widen(c_dict[nt]) for nt in NODE_TYPES}
        einfo = prepare_edges(be, graph, x_dict, ea, None)
        run_cell(be, pc, graph, x_dict, einfo, h_dict, c_dict, proj, agg, h_out, c_out)
        if cw != C:   # (the padded channels are exactly zero)
            h_out = {nt: t[:, :cw].contiguous() for nt, t in h_out.items()}
            c_out = {nt: t[:, :cw].contiguous() for nt, t in c_out.items()}
        return h_out, c_out


class SeqGCLSTM(nn.Module):
    """models.py:173-216 with layers == 1 (parameters.py:49,90,130)."""

    def __init__(self, in_channels_dict, out_channels, num_layers, metadata, device,
                 bias=True, return_all_layers=True):
        super().__init__()
        if num_layers != 1:
            raise NotImplementedError("only layers == 1 is on the shipped path (parameters.py:49)")
        self.in_channels_dict = dict(in_channels_dict)
        self.num_layers = num_layers
        self.cell_list = nn.ModuleList(
            [HeteroPGCLSTM(in_channels_dict, out_channels, metadata, bias, device)])

    def forward(self, x_dict, edge_index_dict, edge_attr, hidden_state):
        h, c = (None, None) if hidden_state is None else hidden_state[0]
        h, c = self.cell_list[0](x_dict, edge_index_dict, edge_attr, h, c)
        return [[h, c]]
