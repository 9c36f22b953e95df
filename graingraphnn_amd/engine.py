"""Launch plan of one model forward on the HIP kernels.

One HeteroPGCLSTM cell (heteropgclstm.py:148-183: 4 gates x 3 PeriodConv + LSTM update) is
THREE launches: the projection GEMMs of both node types, the three aggregation sweeps, the gate
GEMM + LSTM epilogues of both node types -- for one model, or for the regressor and the classifier
together (they run on the same x_dict / graph, test.py:382-383).  A model forward (edge geometry,
encoder cell with h = c = 0, decoder cell, heads; models.py:422-452, 581-609) is 8-9 launches
instead of the ~600 framework kernels the reference issues; a rollout step of both models is 13.
"""
from typing import Dict, Optional, Tuple

import torch

from . import _lib, _pins
from .packing import C, EDGE_TYPES, NODE_TYPES, PackedCell

ET = Tuple[str, str, str]


class GraphCSR:
    """Destination-grouped neighbour lists of the three edge types of one topology."""

    def __init__(self, backend, edge_index_dict, n_nodes: Dict[str, int], trusted: bool = False, into=None, counts=None):
        """trusted: the lists come from the library's own topology update (validated on the host, topology.py): the
        range check of the build -- a read-back, i.e. a host synchronisation -- is skipped.
        into = a backend.CsrInPlace: the tables are refilled IN PLACE (same device tensors, same addresses);
        counts = {et: int64 [1] device tensor}: where the per-edge kernels find the number of edges at run time (CSR.E_dev)."""
        self.csr = {}
        self.edge_index = {}
        self.n_nodes = dict(n_nodes)
        for et in EDGE_TYPES:
            if et not in edge_index_dict:
                raise KeyError(f"edge_index_dict lacks edge type {et}")
            self.edge_index[et] = edge_index_dict[et].contiguous()
        lists = [(self.edge_index[et], n_nodes[et[0]], n_nodes[et[-1]]) for et in EDGE_TYPES]
        if into is not None:   # a backend.CsrInPlace: the same tables, refilled
            built = into.rebuild([l[0] for l in lists])
            for et, csr in zip(EDGE_TYPES, built):
                csr.E_dev = None if counts is None else counts[et]
        elif hasattr(backend, "build_csr_batch"):   # the three edge types in one sequence of launches, one synchronisation
            built = backend.build_csr_batch(lists, check=not trusted) if trusted else backend.build_csr_batch(lists)
        else:
            built = [backend.build_csr(*l) for l in lists]
        self.csr = dict(zip(EDGE_TYPES, built))

    def n_edges(self, et):
        return self.edge_index[et].size(1)


_graph_cache: Dict[tuple, GraphCSR] = {}
_GRAPH_CACHE_MAX = 8


def graph_for(backend, edge_index_dict, n_nodes, trusted: bool = False) -> GraphCSR:
    """CSR of `edge_index_dict`, rebuilt only when a tensor is replaced or modified in place
    (Cmodel.update swaps the tensors after a topological event, models.py:841-845)."""
    key = tuple((et, edge_index_dict[et].data_ptr(), edge_index_dict[et]._version,
                 tuple(edge_index_dict[et].shape)) for et in EDGE_TYPES if et in edge_index_dict)
    key = key + tuple(sorted(n_nodes.items()))
    g = _graph_cache.get(key)
    if g is None:
        g = GraphCSR(backend, edge_index_dict, n_nodes, trusted)
        if len(_graph_cache) >= _GRAPH_CACHE_MAX:
            _graph_cache.pop(next(iter(_graph_cache)))
        _graph_cache[key] = g
    return _pins.note(g)


class Workspace:
    """Per-model device scratch, sized for one (n_grain, n_joint) and reused every step.
    Aggregate buffers are zero-initialised once: their padding columns are never written and
    meet zero weights in the gate GEMM, so they must stay finite."""

    def __init__(self, enc: PackedCell, dec: PackedCell, n_nodes: Dict[str, int], device):
        self.n_nodes = dict(n_nodes)
        f32 = dict(dtype=torch.float32, device=device)
        self.proj, self.agg_enc, self.agg_dec, self.h1, self.c1, self.h2, self.c2 = {}, {}, {}, {}, {}, {}, {}
        self.einfo = None  # edge type -> [E + 3, 20], (re)allocated by prepare_edges
        self.ea = None     # edge type -> [E]: the model's own copy of edge_attr (launch tape, models.py)
        for nt in NODE_TYPES:
            n = n_nodes[nt]
            ncols = max(enc.layout[nt].ncols, dec.layout[nt].ncols)
            self.proj[nt] = torch.empty(n, ncols, **f32)
            self.agg_enc[nt] = torch.zeros(n, enc.G * enc.layout[nt].Kg, **f32)
            self.agg_dec[nt] = torch.zeros(n, dec.G * dec.layout[nt].Kg, **f32)
            for d in (self.h1, self.c1, self.h2, self.c2):
                d[nt] = torch.empty(n, C, **f32)


def _edge_attr_1d(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _lib.GGNNError("edge_attr must be float32")
    if t.dim() == 2 and t.size(1) != 1:
        raise _lib.GGNNError("edge_attr must be [E, 1] (edge_dim is forced to 1, periodGATconv.py:105)")
    return t.contiguous().view(-1)


def _check_x(x: torch.Tensor, F: int, name: str):
    if x.dtype != torch.float32 or x.dim() != 2 or x.size(1) != F or x.stride(1) != 1:
        raise _lib.GGNNError(f"x_dict['{name}'] must be float32 [N, {F}] with unit column stride")


def alloc_einfo(graph: GraphCSR, device, zero: bool = True):
    """[E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW] per edge type: the tail rows pad the last unit's reads.  zero=False: for a
    buffer that goes through ggnn_edge_prepare before anything reads it (that call writes the tail rows as zeros)."""
    make = torch.zeros if zero else torch.empty
    return {et: make(graph.n_edges(et) + _lib.GGNN_UNIT_EDGES, _lib.GGNN_EINFO_ROW, dtype=torch.float32, device=device)
            for et in EDGE_TYPES}


def prepare_edges(backend, graph: GraphCSR, x: Dict[str, torch.Tensor],
                  edge_attr: Dict[ET, torch.Tensor], einfo: Optional[Dict[ET, torch.Tensor]]):
    """One launch: min-image offsets + edge lengths of all three edge types in CSR order."""
    dev = x["joint"].device
    if einfo is None or any(einfo[et].size(0) != graph.n_edges(et) + _lib.GGNN_UNIT_EDGES for et in EDGE_TYPES):
        einfo = alloc_einfo(graph, dev)
    backend.edge_prepare([(graph.csr[et], edge_attr[et], x[et[0]], x[et[-1]], einfo[et])
                          for et in EDGE_TYPES])
    return einfo


def run_cells(backend, cells, graph: GraphCSR, x: Dict[str, torch.Tensor], einfo: Dict[ET, torch.Tensor],
              after_projection=None, after_sweeps=None, range_flag=None):
    """One HeteroPGCLSTM.forward for every entry of `cells` -- (pc, h_in, c_in, proj, agg, h_out,
    c_out), the same cell (encoder or decoder) of one or more models on the same graph, x and edge
    geometry (test.py:382-383 runs the regressor and the classifier on the same x_dict) -- in THREE
    launches: all projections, all aggregation sweeps, all gate GEMM + LSTM epilogues.
    Encoder cells (pc.k2 == 0) ignore h_in / c_in (zeros).  `after_projection`: called right behind the last
    launch of the cell that reads x (the projection; with the fused decoder cell, that kernel) and `after_sweeps`
    behind the last launch that reads the edge records (a caller may record stream events there)."""
    projs, sweeps, enc_sweeps, gates, enc_cells, dec_cells = [], [], [], [], [], []
    flag = () if range_flag is None else (range_flag,)   # the caller's own range-flag word for the fused cells
    for pc, h_in, c_in, proj, agg, h_out, c_out in cells:
        lay = pc.layout
        if pc.ecs and getattr(backend, "fused_encoder", False):
            # encoder: everything of a destination node type in one kernel (ggnn_encoder_cell_batch): no projection, no
            # aggregate or pre-activation buffer
            for nt in NODE_TYPES:
                if lay[nt].live:
                    enc_cells.append(([(graph.csr[et], einfo[et]) for et in lay[nt].dst_ets], x[nt], pc.ecs[nt],
                                      pc.ect[nt], h_out[nt], c_out[nt], *flag))
            continue
        fd = getattr(backend, "fused_decoder", False)
        if fd not in (False, True):   # one model's decoder only: the classifier's has one live destination type
            fd = fd == ("classifier" if sum(bool(lay[nt].live) for nt in NODE_TYPES) == 1 else "regressor")
        if fd and x["joint"].size(0) < getattr(backend, "fused_decoder_min_joints", 0):
            fd = False   # a small graph: three short kernels beat one long dependent chain per tile (backend.py)
        if pc.dcs and fd:
            # decoder: everything on the destination side in one kernel (ggnn_decoder_cell_batch); the projection only
            # emits the source-side value rows
            # the value rows: written once by the projection, gathered 96 columns (one edge type and gate) at a time by the
            # cell -- as [blocks][N][96] (GGNN_OUT_BLOCK_MAJOR) every workgroup of the projection stores one contiguous run
            vbm = getattr(backend, "value_rows_block_major", False)
            for nt in NODE_TYPES:
                if nt in pc.wpv:
                    projs.append((x[nt], lay[nt].F, h_in[nt], pc.wpv[nt], pc.bpv[nt], proj[nt][:, :pc.wpv[nt].size(0)],
                                  (_lib.GGNN_PRECISION_F16X2 if pc.wpv_f16 and backend.f16_projection() else 0)
                                  | (_lib.GGNN_OUT_BLOCK_MAJOR if vbm else 0)))
            for nt in NODE_TYPES:
                if lay[nt].live:
                    dec_cells.append(([(graph.csr[et], einfo[et], h_in[et[0]], proj[et[0]], pc.vof[et], pc.ep[et], vbm)
                                       for et in lay[nt].dst_ets], x[nt], h_in[nt], c_in[nt], pc.dcs[nt], pc.dct[nt],
                                      h_out[nt], c_out[nt], *flag))
            continue
        for nt in NODE_TYPES:
            P = proj[nt][:, :lay[nt].ncols] if proj[nt].size(1) != lay[nt].ncols else proj[nt]
            projs.append((x[nt], lay[nt].F, h_in[nt] if pc.k2 else None, pc.wp[nt], pc.bp[nt], P))
        for et in EDGE_TYPES:  # fewer sweeps when a destination type is dead
            s, d = et[0], et[-1]
            if not lay[d].live:
                continue
            if pc.wvf:  # encoder: values from the edge records on the matrix cores
                enc_sweeps.append((graph.csr[et], einfo[et], proj[d], pc.wvf[et], agg[d], lay[d].u4_off[et],
                                   lay[d].a_off[et], lay[d].Kg, lay[d].sc_off[et], pc.G))
                continue
            sweeps.append((graph.csr[et], einfo[et], proj[s], proj[d], h_in[s] if pc.k2 else None,
                           pc.ep[et], agg[d], lay[s].v_off[et], lay[d].u_off.get(et, 0), lay[d].u4_off[et],
                           lay[d].a_off[et], lay[d].Kg, lay[d].sc_off[et], pc.G))
        gates += gate_problems(pc, proj, agg, c_in, h_out, c_out)
    if projs:
        backend.project_batch(projs)
    if after_projection is not None and not dec_cells:
        after_projection()
    if enc_cells:
        backend.encoder_cell_batch(enc_cells)
    if enc_sweeps:
        backend.aggregate_enc_batch(enc_sweeps)
    if sweeps:
        backend.aggregate_batch(sweeps)
    if dec_cells:
        backend.decoder_cell_batch(dec_cells)   # reads the destination nodes' features: x's last reader
        if after_projection is not None:
            after_projection()
    if after_sweeps is not None:
        after_sweeps()
    if gates:
        backend.lstm_epilogue_batch(gates)


def run_cell(backend, pc: PackedCell, graph: GraphCSR, x: Dict[str, torch.Tensor],
             einfo: Dict[ET, torch.Tensor], h_in: Optional[Dict[str, torch.Tensor]],
             c_in: Optional[Dict[str, torch.Tensor]], proj, agg, h_out, c_out):
    """One HeteroPGCLSTM.forward of one model."""
    run_cells(backend, [(pc, h_in, c_in, proj, agg, h_out, c_out)], graph, x, einfo)


def gate_problems(pc: PackedCell, proj, agg, c_in, h_out, c_out):
    """Argument tuples of ggnn_lstm_epilogue for the live node types of one cell."""
    mode = _lib.MODE_LSTM if pc.k2 else _lib.MODE_LSTM_H0
    lay = pc.layout
    return [(agg[nt], pc.w2[nt], proj[nt], lay[nt].s_off, c_in[nt] if pc.k2 else None, h_out[nt], c_out[nt],
             None, pc.G, mode, pc.w2p.get(nt), lay[nt].Kg) for nt in NODE_TYPES if lay[nt].live]


def run_encoder_decoder(backend, enc: PackedCell, dec: PackedCell, graph: GraphCSR, ws: Workspace,
                        x: Dict[str, torch.Tensor], edge_attr: Dict[ET, torch.Tensor],
                        einfo: Optional[Dict[ET, torch.Tensor]] = None, x_read=None, einfo_read=None, after_encoder=None):
    """models.py:422-426 / 581-585: encoder from zero state, decoder from the encoder's (h, c),
    both on the same x_dict.  `einfo` (from prepare_edges) may be shared by several models that
    see the same x / edge_attr; when absent it is computed here.  Returns the decoder's (h, c).
    `after_encoder`: called once the encoder cell's launches are enqueued (a caller may record a stream event there)."""
    if einfo is None:
        ea = {et: _edge_attr_1d(edge_attr[et]) for et in EDGE_TYPES}
        einfo = ws.einfo = prepare_edges(backend, graph, x, ea, ws.einfo)
    run_encoder_decoder_multi(backend, [(enc, dec, ws)], graph, x, einfo, x_read, einfo_read, after_encoder)
    return ws.h2, ws.c2


def run_encoder_decoder_multi(backend, models, graph: GraphCSR, x: Dict[str, torch.Tensor],
                              einfo: Dict[ET, torch.Tensor], x_read=None, einfo_read=None, after_encoder=None):
    """The encoder cells of all `models` = [(enc, dec, workspace), ...] in three launches, then
    their decoder cells in three more (every model keeps its own weights, workspace and state)."""
    flag = getattr(models[0][2], "range_flag", None)   # (models launched together belong to one rollout: one word)
    run_cells(backend, [(enc, None, None, ws.proj, ws.agg_enc, ws.h1, ws.c1) for enc, _, ws in models],
              graph, x, einfo, range_flag=flag)
    if after_encoder is not None:
        after_encoder()
    # (x_read / einfo_read: called once the last launch that reads x -- the decoder projection -- / the edge
    # records -- the decoder sweeps -- is enqueued)
    run_cells(backend, [(dec, ws.h1, ws.c1, ws.proj, ws.agg_dec, ws.h2, ws.c2) for _, dec, ws in models],
              graph, x, einfo, after_projection=x_read, after_sweeps=einfo_read, range_flag=flag)
