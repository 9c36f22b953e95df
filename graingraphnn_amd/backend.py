"""Thin tensor-level wrapper over the C ABI (include/ggnn.h).  One method per entry point;
tensors in, launches enqueued on the current HIP stream, nothing returned but the outputs
written in place.  `HipBackend` is the only backend the package ships: it refuses CPU
tensors and raises if libggnn.so cannot be loaded -- there is no CPU fallback."""
import ctypes
import os

import torch

from . import _lib
from ._lib import (AggregateArgs, AggregateBwdArgs, AggregateEncArgs, DecCellArgs, EncCellArgs, EpilogueArgs, PrepareEdge, ProjectArgs,
                   RefreshEdge, check, ptr)



class CSR:
    """Destination-grouped edge list of one edge type (all int32, device resident)."""
    __slots__ = ("rowptr", "col", "perm", "row", "unit_ptr", "units", "E", "E_dev")

    def __init__(self, rowptr, col, perm, row, unit_ptr, units, E, E_dev=None):
        self.rowptr, self.col, self.perm, self.row = rowptr, col, perm, row
        self.unit_ptr, self.units, self.E = unit_ptr, units, E
        # int64 [1] on the device or None: the number of edges the per-edge kernels read at RUN time (tables that are
        # rebuilt in place under captured launches: GrainRollout's event loop; include/ggnn.h, ggnn_prepare_edge.E_dev)
        self.E_dev = E_dev


class CsrInPlace:
    """See HipBackend.csr_in_place."""

    def __init__(self, backend, shapes, device):
        if not 1 <= len(shapes) <= 4:
            raise _lib.GGNNError("csr_in_place: one to four lists")
        self.be, self.shapes = backend, [tuple(int(v) for v in s) for s in shapes]
        lib = backend.lib
        r4 = lambda n: (n + 3) & ~3
        words = sum(backend.csr_arena_words(cap, n_dst) for cap, _, n_dst in self.shapes)
        self.arena = torch.empty(words, dtype=torch.int32, device=device)
        self.args = (_lib.CsrArgs * len(self.shapes))()
        self.csr, at = [], 0

        def take(n, shape=None):
            nonlocal at
            t = self.arena[at:at + n]
            at += r4(n)
            return t if shape is None else t.view(shape)
        for a, (cap, n_src, n_dst) in zip(self.args, self.shapes):
            rowptr, col, perm, row = take(n_dst + 1), take(max(cap, 1)), take(max(cap, 1)), take(max(cap, 1))
            unit_ptr = take(n_dst + 1)
            n_units = lib.ggnn_csr_max_units(cap, n_dst)
            units, flags = take(8 * n_units, (n_units, 8)), take(2)
            nbytes = lib.ggnn_csr_workspace_bytes(cap, n_dst)
            ws = take(nbytes // 4 + 1)
            a.n_src, a.n_dst = n_src, n_dst
            a.rowptr, a.col, a.perm, a.row = rowptr.data_ptr(), col.data_ptr(), perm.data_ptr(), row.data_ptr()
            a.unit_ptr, a.units, a.flags = unit_ptr.data_ptr(), units.data_ptr(), flags.data_ptr()
            a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes   # (sized for the capacity: enough for any shorter list)
            self.csr.append(CSR(rowptr, col, perm, row, unit_ptr, units, 0))

    def rebuild(self, lists):
        for a, csr, ei, (cap, _, _) in zip(self.args, self.csr, lists, self.shapes):
            if ei.dtype != torch.int64 or ei.dim() != 2 or ei.size(0) != 2 or not ei.is_contiguous() or not ei.is_cuda:
                raise _lib.GGNNError("edge_index must be a contiguous int64 [2, E] device tensor")
            E = ei.size(1)
            if E > cap:
                raise _lib.GGNNError("csr_in_place: a list grew beyond its capacity")
            a.edge_index, a.E = ei.data_ptr(), E
            csr.E = E
        _lib.check(self.be.lib.ggnn_build_csr_batch(self.args, len(self.shapes), _lib.current_stream()), "ggnn_build_csr_batch")
        return self.csr


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.GGNNError(
                "graingraphnn_amd runs on MI355X only: got a CPU tensor (no CPU fallback exists; "
                "move the model and the graph to 'cuda')")


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.GGNNError(f"{name} must be a contiguous float32 tensor")
    return t


class HipBackend:
    name = "hip"
    FUSED_DECODER_MIN_JOINTS = 6000

    def __init__(self):
        self.lib = _lib.load()
        self._tape = None  # while a list: every hot-path launch is also recorded as (cfunc, name, cargs)
        self._range_flags = {}  # device -> int32 word the fp16 two-piece kernels report clamped activations in
        self._rowgemm_ws = {}   # (device, stream, bytes) -> weight-plane workspace of ggnn_rowgemm
        self._mse_ws = {}       # (device, stream) -> partial sums + arrival counter of ggnn_masked_mse
        # encoder cell as ONE fused sweep + gate GEMM launch (ggnn_encoder_cell_batch; bf16x6 arithmetic
        # only).  GGNN_ENC=split keeps the sweep and the gate GEMM as separate launches (development).
        self.fused_encoder = (self.lib.ggnn_gemm_mode() == 1 and os.environ.get("GGNN_ENC", "") != "split")
        # The decoder cell of a model: ONE kernel (ggnn_decoder_cell_batch, the default with the bf16 / fp16 matrix-core
        # GEMM mode) or projection + sweeps + gate GEMM (GGNN_DEC=split; always under GGNN_GEMM=fp32).
        # GGNN_DEC=fused-classifier / fused-regressor: the fused cell for that model only (development).
        # By default (GGNN_DEC unset) the plan follows the graph's size (engine.run_cells): a fused cell is one dependent
        # chain of ~110 us per 16-node tile however few tiles there are, so below FUSED_DECODER_MIN_JOINTS junctions the
        # three short kernels win (1 043-grain fixture: 8 080 vs 5 620 steps/s; 64 x 118 grains batched: 205 k vs 220 k
        # trajectory-steps/s; 10 000 grains: 2 380 vs 2 920 steps/s).  GGNN_DEC=fused: the fused cell whatever the size.
        dec = os.environ.get("GGNN_DEC", "auto")
        self.fused_decoder = False
        self.fused_decoder_min_joints = 0
        if self.lib.ggnn_gemm_mode() == 1 and (dec == "auto" or dec.startswith("fused")):
            self.fused_decoder = {"auto": True, "fused": True, "fused-classifier": "classifier",
                                  "fused-regressor": "regressor"}.get(dec, False)
            if dec == "auto":
                self.fused_decoder_min_joints = int(os.environ.get("GGNN_DEC_MIN_JOINTS", str(self.FUSED_DECODER_MIN_JOINTS)))

        # the fused decoder plan's value rows as [blocks][N][96] (GGNN_OUT_BLOCK_MAJOR: every workgroup of the projection
        # stores contiguous runs, and a (edge type, gate) pass of the cell gathers from one dense [N, 96] matrix --
        # dec_cell_kernel 110.2 -> 105.5 us, project_x6_kernel 22.8 -> 21.9 us per launch, 3 020 -> 3 054 steps/s on one box,
        # profiles/r6_value_rows_layout.txt); GGNN_VLAYOUT=rows: [N, ncols] (A/B runs)
        self.value_rows_block_major = os.environ.get("GGNN_VLAYOUT", "block") != "rows"

    def f16_projection(self) -> bool:
        """The fused decoder plan's value projection in the cells' three-product arithmetic (GGNN_PRECISION_F16X2) when the
        packed weights allow; GGNN_PROJ=x6 keeps the six-product split (development, A/B runs)."""
        return os.environ.get("GGNN_PROJ", "") != "x6"

    # -- launch tape: the drop-in forward() issues the same dozen launches with the same arguments
    # step after step (test.py:382-383); re-issuing the recorded C calls skips the per-launch
    # Python work (argument checks, ctypes struct filling), which otherwise outweighs the kernels.
    def _launch(self, fn, name, *cargs):
        check(fn(*cargs), name)
        if self._tape is not None:
            self._tape.append((fn, name, cargs))

    def start_tape(self):
        self._tape = []

    def stop_tape(self):
        tape, self._tape = self._tape, None
        return tape

    @staticmethod
    def replay(tape):
        for fn, name, cargs in tape:
            rc = fn(*cargs)
            if rc:
                check(rc, name)

    # -- CSR ---------------------------------------------------------------------------
    def build_csr(self, edge_index, n_src, n_dst):
        """edge_index [2, E] int64 (cuda) -> CSR.  Raises IndexError on out-of-range indices
        (one host sync, only when a topology is first seen)."""
        return self.build_csr_batch([(edge_index, n_src, n_dst)])[0]

    def csr_arena_words(self, E_cap, n_dst):
        """int32 words of one list's tables (row pointers, columns, permutation, rows, unit tables, flags, workspace) for up
        to E_cap edges, every table starting on a 16-byte boundary."""
        r4 = lambda n: (n + 3) & ~3
        return 2 * r4(n_dst + 1) + 3 * r4(max(int(E_cap), 1)) + 8 * self.lib.ggnn_csr_max_units(int(E_cap), n_dst) + 4 \
            + r4(self.lib.ggnn_csr_workspace_bytes(int(E_cap), n_dst) // 4 + 1)

    def csr_in_place(self, shapes, device):
        """Tables that are rebuilt in place: `shapes` = [(E_cap, n_src, n_dst)] (at most four lists) -> a CsrInPlace whose
        `rebuild([edge_index [2, E <= E_cap]])` fills the SAME device tables for the new lists (unchecked: validated lists)
        and returns the same CSR objects with their E updated -- tensors of capacity size, addresses that never change, so
        launches captured on an earlier, longer version of a list keep reading valid tables."""
        return CsrInPlace(self, shapes, device)

    def build_csr_batch(self, lists, check=True):
        """[(edge_index [2, E] int64 cuda, n_src, n_dst)] -> [CSR]: ggnn_build_csr_batch, up to four lists per sequence of
        launches (engine.GraphCSR builds the three edge types of a topology in one; a topological event rebuilds them),
        one range check = one host synchronisation behind the last (`check=False`: lists the caller has validated -- the
        kernels skip an out-of-range edge either way --: no read-back, the host goes on enqueueing).  (Tables that are
        refilled IN PLACE per event: `csr_in_place`.)"""
        out, todo = [], list(lists)
        while todo:
            chunk, todo = todo[:4], todo[4:]
            arr = (_lib.CsrArgs * len(chunk))()
            keep = []
            arena = None
            if not check:
                # unchecked builds run once per topological event: every table of the chunk out of ONE allocation (two
                # dozen allocator calls and six fills otherwise; the unit table's unused tail and the range flags that
                # nobody reads stay uninitialised)
                need = sum(self.csr_arena_words(int(ei_.size(1)), n_dst) for ei_, _, n_dst in chunk)
                arena = [torch.empty(need, dtype=torch.int32, device=chunk[0][0].device), 0]

            def take(n, like_zeros=False, shape=None, dev=None):
                if arena is None:
                    t = (torch.zeros if like_zeros else torch.empty)(n, dtype=torch.int32, device=dev)
                else:
                    t = arena[0][arena[1]:arena[1] + n]
                    arena[1] += (n + 3) & ~3
                return t if shape is None else t.view(shape)
            for a, (edge_index, n_src, n_dst) in zip(arr, chunk):
                _require_cuda(edge_index)
                if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
                    raise _lib.GGNNError("edge_index must be int64 [2, E]")
                ei = edge_index.contiguous()
                E, dev = ei.size(1), ei.device
                rowptr = take(n_dst + 1, dev=dev)
                col = take(max(E, 1), dev=dev)
                perm = take(max(E, 1), dev=dev)
                row = take(max(E, 1), dev=dev)
                unit_ptr = take(n_dst + 1, dev=dev)
                n_units = self.lib.ggnn_csr_max_units(E, n_dst)
                units = take(8 * n_units, True, (n_units, 8), dev)
                flags = take(2, True, dev=dev)
                nbytes = self.lib.ggnn_csr_workspace_bytes(E, n_dst)
                ws = take(nbytes // 4 + 1, dev=dev).view(torch.uint8)[:nbytes] if arena is not None else \
                    torch.empty(nbytes, dtype=torch.uint8, device=dev)
                a.edge_index, a.E, a.n_src, a.n_dst = ei.data_ptr(), E, n_src, n_dst
                a.rowptr, a.col, a.perm, a.row = rowptr.data_ptr(), col.data_ptr(), perm.data_ptr(), row.data_ptr()
                a.unit_ptr, a.units, a.flags = unit_ptr.data_ptr(), units.data_ptr(), flags.data_ptr()
                a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
                keep.append((ei, ws, flags, n_src, n_dst))
                out.append(CSR(rowptr, col, perm, row, unit_ptr, units, E))
            _lib.check(self.lib.ggnn_build_csr_batch(arr, len(chunk), _lib.current_stream()), "ggnn_build_csr_batch")
            if not check:   # (ei / ws / flags are only used by launches on this stream: the allocator keeps them until those ran)
                continue
            bad = torch.stack([k[2][0] for k in keep]).cpu()   # (the synchronisation; also keeps ei / ws alive until here)
            for k, f in zip(keep, bad.tolist()):
                if f & 1:
                    raise IndexError(f"edge_index has entries outside [0,{k[3]}) x [0,{k[4]})")
        return out

    # -- per-edge geometry -------------------------------------------------------------
    def edge_prepare(self, items):
        """items: list of (csr, edge_attr [E] COO order, x_src, x_dst, einfo_out [E + 3, GGNN_EINFO_ROW])."""
        arr = (PrepareEdge * max(len(items), 1))()
        for k, (csr, ea, xs, xd, einfo) in enumerate(items):
            _require_cuda(csr.col, ea, xs, xd, einfo)
            if einfo.size(0) < ea.numel() + _lib.GGNN_UNIT_EDGES or einfo.size(1) != _lib.GGNN_EINFO_ROW:
                raise _lib.GGNNError("einfo must be [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW]")
            a = arr[k]
            a.col, a.perm, a.row = csr.col.data_ptr(), csr.perm.data_ptr(), csr.row.data_ptr()
            a.edge_attr, a.x_src, a.x_dst = ea.data_ptr(), xs.data_ptr(), xd.data_ptr()
            a.einfo = einfo.data_ptr()
            a.ldx_src, a.ldx_dst, a.E, a.f_src = xs.stride(0), xd.stride(0), ea.numel(), xs.size(1)
            a.E_dev = ptr(getattr(csr, "E_dev", None))   # (a topology that shrinks in place under captured launches)
        self._launch(self.lib.ggnn_edge_prepare, "ggnn_edge_prepare", arr, len(items), _lib.current_stream())

    # -- projection --------------------------------------------------------------------
    def project(self, x, F, h, wp, bp, out):
        _require_cuda(x, h, wp, bp, out)
        M = x.size(0)
        k2 = 0 if h is None else h.size(1)
        self._launch(self.lib.ggnn_project, "ggnn_project", ptr(x), x.stride(0), F, ptr(h),
                     0 if h is None else h.stride(0), k2, ptr(wp), ptr(bp), M, wp.size(0), ptr(out),
                     out.stride(0), _lib.current_stream())

    def project_batch(self, problems):
        """Up to four projections in one launch (ggnn_project_batch); each item is the argument tuple
        of `project`: (x, F, h, wp, bp, out[, precision]).  All with h or all without."""
        arr = (ProjectArgs * len(problems))()
        if len({pr[2] is None for pr in problems}) > 1:
            raise _lib.GGNNError("ggnn_project_batch: every problem with a hidden state or none of them "
                                 "(problems with k2 = 96 and k2 = 0 cannot share a launch)")
        for a, (x, F, h, wp, bp, out, *rest) in zip(arr, problems):
            a.precision = rest[0] if rest else 0   # optional 7th element: _lib.GGNN_PRECISION_BF16 (training, autocast)
            _require_cuda(x, h, wp, bp, out)
            # what the C side cannot see behind a raw pointer: dtypes, unit column strides, the weight's shape
            k2 = 0 if h is None else h.size(1)
            for t, name in ((x, "x"), (h, "h"), (wp, "wp"), (bp, "bp"), (out, "out")):
                if t is not None and t.dtype != torch.float32:
                    raise _lib.GGNNError(f"ggnn_project_batch: {name} must be float32")
            if x.dim() != 2 or x.stride(1) != 1 or x.size(1) < F or out.dim() != 2 or out.stride(1) != 1:
                raise _lib.GGNNError("ggnn_project_batch: x [M, >= F] and out [M, ncols] need unit column stride")
            if h is not None and (h.dim() != 2 or h.stride(1) != 1 or h.size(0) != x.size(0)):
                raise _lib.GGNNError("ggnn_project_batch: h must be [M, k2] with unit column stride")
            if not wp.is_contiguous() or wp.dim() != 2 or wp.size(1) != ((F + 3) & ~3) + k2:
                raise _lib.GGNNError(f"ggnn_project_batch: wp must be contiguous [ncols, roundup4(F) + k2] = "
                                     f"[*, {((F + 3) & ~3) + k2}], got {tuple(wp.shape)}")
            if bp.numel() != wp.size(0) or not bp.is_contiguous():
                raise _lib.GGNNError("ggnn_project_batch: bp must hold ncols contiguous values")
            if out.size(0) != x.size(0) or out.size(1) < wp.size(0):
                raise _lib.GGNNError("ggnn_project_batch: out must be [M, >= ncols]")
            a.X, a.Wp, a.bias, a.out = x.data_ptr(), wp.data_ptr(), bp.data_ptr(), out.data_ptr()
            a.H = None if h is None else h.data_ptr()
            a.ldx, a.ldh, a.M, a.ldo = x.stride(0), 0 if h is None else h.stride(0), x.size(0), out.stride(0)
            a.F, a.k2, a.ncols = F, 0 if h is None else h.size(1), wp.size(0)
        self._launch(self.lib.ggnn_project_batch, "ggnn_project_batch", arr, len(problems), _lib.current_stream())

    # -- aggregation -------------------------------------------------------------------
    @staticmethod
    def _sweep_args(a, csr, einfo, p_src, p_dst, h_src, ep, agg, v_off, u_off, u4_off, a_off,
                    a_gstride, sc_off, n_gates, pad_n=0):
        _require_cuda(csr.unit_ptr, einfo, p_src, p_dst, h_src, ep, agg)
        a.unit_ptr, a.units, a.einfo = csr.unit_ptr.data_ptr(), csr.units.data_ptr(), einfo.data_ptr()
        a.p_src, a.p_dst = p_src.data_ptr(), p_dst.data_ptr()
        a.h_src = None if h_src is None else h_src.data_ptr()
        a.edge_params, a.agg = ep.data_ptr(), agg.data_ptr()
        a.ldp_src, a.ldp_dst, a.ld_agg = p_src.stride(0), p_dst.stride(0), agg.stride(0)
        a.ldh_src = 0 if h_src is None else h_src.stride(0)
        a.n_src, a.n_dst, a.E = p_src.size(0), p_dst.size(0), csr.E
        a.v_off, a.u_off, a.u4_off, a.a_off, a.a_gstride, a.sc_off, a.n_gates = (
            v_off, u_off, u4_off, a_off, a_gstride, sc_off, n_gates)
        a.pad_n = pad_n

    def aggregate(self, *sweep):
        """One sweep of ggnn_period_gat_aggregate (include/ggnn.h): (csr, einfo, p_src, p_dst, h_src,
        ep, agg, v_off, u_off, u4_off, a_off, a_gstride, sc_off, n_gates[, pad_n]).  h_src: the source node
        type's hidden state [n_src, 96] or None (encoder); pad_n: columns behind the sweep's scalars that it zeroes in
        every gate row (the training path's padded gate rows)."""
        a = AggregateArgs()
        self._sweep_args(a, *sweep)
        check(self.lib.ggnn_period_gat_aggregate(ctypes.byref(a), _lib.current_stream()),
              "ggnn_period_gat_aggregate")

    def aggregate_batch(self, sweeps):
        """The 1..3 sweeps of one cell in one launch (ggnn_period_gat_aggregate_batch); each item
        is the argument tuple of `aggregate`."""
        arr = (AggregateArgs * len(sweeps))()
        for a, sweep in zip(arr, sweeps):
            self._sweep_args(a, *sweep)
        self._launch(self.lib.ggnn_period_gat_aggregate_batch, "ggnn_period_gat_aggregate_batch", arr,
                     len(sweeps), _lib.current_stream())

    def aggregate_enc_batch(self, sweeps):
        """Encoder sweeps (h = 0) with the edge values on the matrix cores
        (ggnn_period_gat_aggregate_enc_batch); each item: (csr, einfo, p_dst, wv_frag, agg, u4_off, a_off,
        a_gstride, sc_off, n_gates)."""
        arr = (AggregateEncArgs * len(sweeps))()
        for a, (csr, einfo, p_dst, wvf, agg, u4_off, a_off, a_gstride, sc_off, n_gates) in zip(arr, sweeps):
            _require_cuda(csr.unit_ptr, einfo, p_dst, wvf, agg)
            if wvf.dtype != torch.float32 or wvf.numel() != 6 * n_gates * 3 * 64:
                raise _lib.GGNNError("wv_frag does not match n_gates (see packing.value_fragments)")
            a.unit_ptr, a.units, a.einfo = csr.unit_ptr.data_ptr(), csr.units.data_ptr(), einfo.data_ptr()
            a.p_dst, a.wv_frag, a.agg = p_dst.data_ptr(), wvf.data_ptr(), agg.data_ptr()
            a.ldp_dst, a.ld_agg, a.n_dst, a.E = p_dst.stride(0), agg.stride(0), p_dst.size(0), csr.E
            a.u4_off, a.a_off, a.a_gstride, a.sc_off, a.n_gates = u4_off, a_off, a_gstride, sc_off, n_gates
        self._launch(self.lib.ggnn_period_gat_aggregate_enc_batch, "ggnn_period_gat_aggregate_enc_batch", arr,
                     len(sweeps), _lib.current_stream())

    def encoder_cell_batch(self, problems):
        """ggnn_encoder_cell_batch (include/ggnn.h): the encoder cell (h = c = 0) of up to four (model, destination
        node type) problems in one launch.  Each item: (sweeps, x_dst, wstream, w2_tail, h_out, c_out) with sweeps =
        [(csr, einfo)] for the 1 or 2 incoming edge types; wstream / w2_tail: packing.encoder_cell_stream."""
        arr = (EncCellArgs * len(problems))()
        for a, (sweeps, x_dst, wstream, w2_tail, h_out, c_out, *rest) in zip(arr, problems):
            _require_cuda(x_dst, wstream, w2_tail, h_out, c_out)
            n, n_in = x_dst.size(0), len(sweeps)
            for t, name in ((x_dst, "x_dst"), (h_out, "h_out"), (c_out, "c_out"), (w2_tail, "w2_tail")):
                if t.dtype != torch.float32 or t.dim() < 2 or t.stride(-1) != 1:
                    raise _lib.GGNNError(f"ggnn_encoder_cell_batch: {name} must be float32 with unit column stride")
            if n_in not in (1, 2) or tuple(h_out.shape) != (n, 96) or not h_out.is_contiguous() \
                    or tuple(c_out.shape) != (n, 96) or not c_out.is_contiguous():
                raise _lib.GGNNError("ggnn_encoder_cell_batch: h_out / c_out must be contiguous [n_dst, 96], 1 or 2 "
                                     "incoming edge types")
            if wstream.dtype != torch.int16 or not wstream.is_contiguous() \
                    or wstream.numel() * 2 != 3 * (4 * n_in + 1) * _lib.GGNN_DC_SLICE_BYTES:
                raise _lib.GGNNError("ggnn_encoder_cell_batch: wstream is not packing.encoder_cell_stream of this "
                                     "number of incoming edge types")
            if tuple(w2_tail.shape) != (3, n_in, 6, 64) or not w2_tail.is_contiguous():
                raise _lib.GGNNError("ggnn_encoder_cell_batch: w2_tail must be contiguous [3, n_in, 6, 64]")
            for sw, (csr, einfo) in zip(a.sweeps, sweeps):
                _require_cuda(csr.rowptr, einfo)
                if csr.rowptr.numel() != n + 1:
                    raise _lib.GGNNError("the sweep's CSR does not have one row per destination node")
                if einfo.dtype != torch.float32 or not einfo.is_contiguous() or einfo.dim() != 2 \
                        or einfo.size(1) != _lib.GGNN_EINFO_ROW or einfo.size(0) < csr.E + _lib.GGNN_UNIT_EDGES:
                    raise _lib.GGNNError("einfo must be contiguous float32 [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW]")
                sw.rowptr, sw.einfo, sw.E = csr.rowptr.data_ptr(), einfo.data_ptr(), csr.E
            a.x_dst, a.h_out, a.c_out = x_dst.data_ptr(), h_out.data_ptr(), c_out.data_ptr()
            a.wstream, a.w2_tail = wstream.data_ptr(), w2_tail.data_ptr()
            # (optional last element of a problem: the caller's own flag word -- a rollout's -- instead of the device-wide one)
            a.flags = (rest[0] if rest and rest[0] is not None else self.range_flag(x_dst.device)).data_ptr()
            a.n_dst, a.ldx, a.n_in, a.f_dst = n, x_dst.stride(0), n_in, x_dst.size(1)
        self._launch(self.lib.ggnn_encoder_cell_batch, "ggnn_encoder_cell_batch", arr, len(problems),
                     _lib.current_stream())

    def range_flag(self, device):
        """The device word the two-piece fp16 kernels OR GGNN_FLAG_F16_RANGE into when they clamp an activation
        (include/ggnn.h, OPERAND RANGE); one per device, never cleared by the kernels."""
        key = torch.device(device)
        if key.index is None:
            key = torch.device(key.type, torch.cuda.current_device())
        flag = self._range_flags.get(key)
        if flag is None:
            flag = self._range_flags[key] = torch.zeros(1, dtype=torch.int32, device=key)
        return flag

    def range_exceeded(self, device="cuda", clear=True, flag=None) -> bool:
        """True when a fused cell has clamped an activation to fp16's range since the flag was last cleared: its
        results are then NOT the reference's (re-run with GGNN_DEC=split / GGNN_ENC=split, whose bf16 x 3 split
        covers fp32's range).  One host sync.  `flag`: a caller's own word (GrainRollout keeps one per rollout, so that
        two rollouts on a device do not consume each other's reports); default: the device-wide word."""
        flag = self.range_flag(device) if flag is None else flag
        hit = bool(int(flag.item()) & _lib.GGNN_FLAG_F16_RANGE)
        if hit and clear:
            flag.zero_()
        return hit

    def decoder_cell_batch(self, problems):
        """ggnn_decoder_cell_batch (include/ggnn.h): the decoder cell of up to four (model, destination node type)
        problems in one launch.  Each item: (sweeps, x_dst, h_dst, c_in, wstream, w2_tail, h_out, c_out) with
        sweeps = [(csr, einfo, h_src, v_src, v_off, edge_params)] for the 1 or 2 incoming edge types; wstream /
        w2_tail: packing.decoder_cell_stream."""
        arr = (DecCellArgs * len(problems))()
        for a, (sweeps, x_dst, h_dst, c_in, wstream, w2_tail, h_out, c_out, *rest) in zip(arr, problems):
            _require_cuda(x_dst, h_dst, c_in, wstream, w2_tail, h_out, c_out)
            n, n_in = x_dst.size(0), len(sweeps)
            for t, name in ((x_dst, "x_dst"), (h_dst, "h_dst"), (c_in, "c_in"), (h_out, "h_out"), (c_out, "c_out"),
                            (w2_tail, "w2_tail")):
                if t.dtype != torch.float32 or t.dim() < 2 or t.stride(-1) != 1:
                    raise _lib.GGNNError(f"ggnn_decoder_cell_batch: {name} must be float32 with unit column stride")
            if n_in not in (1, 2) or tuple(h_dst.shape) != (n, 96) or tuple(c_in.shape) != (n, 96) \
                    or not c_in.is_contiguous() or tuple(h_out.shape) != (n, 96) or not h_out.is_contiguous() \
                    or tuple(c_out.shape) != (n, 96) or not c_out.is_contiguous():
                raise _lib.GGNNError("ggnn_decoder_cell_batch: h_dst / c_in / h_out / c_out must be [n_dst, 96] "
                                     "(c_in, h_out, c_out contiguous), 1 or 2 incoming edge types")
            if wstream.dtype != torch.int16 or not wstream.is_contiguous() \
                    or wstream.numel() * 2 != 4 * (7 * n_in + 4) * _lib.GGNN_DC_SLICE_BYTES:
                raise _lib.GGNNError("ggnn_decoder_cell_batch: wstream is not packing.decoder_cell_stream of this "
                                     "number of incoming edge types")
            if tuple(w2_tail.shape) != (4, n_in, 6, 64) or not w2_tail.is_contiguous():
                raise _lib.GGNNError("ggnn_decoder_cell_batch: w2_tail must be contiguous [4, n_in, 6, 64]")
            for sw, (csr, einfo, h_src, v_src, v_off, ep, *v_bm) in zip(a.sweeps, sweeps):
                v_bm = bool(v_bm and v_bm[0])   # optional 7th element: v_src is GGNN_OUT_BLOCK_MAJOR ([blocks][n_src][96])
                _require_cuda(csr.rowptr, csr.col, einfo, h_src, v_src, ep)
                if csr.rowptr.numel() != n + 1:
                    raise _lib.GGNNError("the sweep's CSR does not have one row per destination node")
                if einfo.dtype != torch.float32 or not einfo.is_contiguous() or einfo.dim() != 2 \
                        or einfo.size(1) != _lib.GGNN_EINFO_ROW or einfo.size(0) < csr.E + _lib.GGNN_UNIT_EDGES:
                    raise _lib.GGNNError("einfo must be contiguous float32 [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW]")
                for t, name in ((h_src, "h_src"), (v_src, "v_src")):
                    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.size(0) != h_src.size(0):
                        raise _lib.GGNNError(f"ggnn_decoder_cell_batch: {name} must be float32 [n_src, *] with unit "
                                             "column stride")
                if v_bm and (not v_src.is_contiguous() or v_off % 96):
                    raise _lib.GGNNError("ggnn_decoder_cell_batch: block-major value rows live in a contiguous buffer, "
                                         "v_off a multiple of 96")
                if h_src.size(1) < 96 or v_off < 0 or v_off + 4 * 96 > v_src.size(1):
                    raise _lib.GGNNError("ggnn_decoder_cell_batch: h_src needs 96 columns, the four gates' value rows "
                                         "must lie inside a v_src row")
                if ep.dtype != torch.float32 or not ep.is_contiguous() or tuple(ep.shape) != (4, 3, 96):
                    raise _lib.GGNNError("edge_params must be contiguous float32 [4, 3, 96]")
                sw.rowptr, sw.col, sw.einfo = csr.rowptr.data_ptr(), csr.col.data_ptr(), einfo.data_ptr()
                sw.h_src, sw.v_src, sw.edge_params = h_src.data_ptr(), v_src.data_ptr(), ep.data_ptr()
                sw.E, sw.n_src, sw.ldh_src, sw.ldv, sw.v_off = csr.E, h_src.size(0), h_src.stride(0), v_src.stride(0), v_off
                sw.v_block_major = int(v_bm)
            a.x_dst, a.h_dst, a.c_in = x_dst.data_ptr(), h_dst.data_ptr(), c_in.data_ptr()
            a.h_out, a.c_out = h_out.data_ptr(), c_out.data_ptr()
            a.wstream, a.w2_tail = wstream.data_ptr(), w2_tail.data_ptr()
            # (optional last element of a problem: the caller's own flag word -- a rollout's -- instead of the device-wide one)
            a.flags = (rest[0] if rest and rest[0] is not None else self.range_flag(x_dst.device)).data_ptr()
            a.n_dst, a.ldx, a.ldh, a.n_in, a.f_dst = n, x_dst.stride(0), h_dst.stride(0), n_in, x_dst.size(1)
        self._launch(self.lib.ggnn_decoder_cell_batch, "ggnn_decoder_cell_batch", arr, len(problems),
                     _lib.current_stream())

    def aggregate_backward(self, csr, rcsr, r_slot, einfo, p_src, p_dst, h_src, ep, agg, g_agg,
                           v_off, u_off, u4_off, a_off, a_gstride, sc_off, n_gates, out_p_dst=None, out_p_src=None,
                           ep_partial_out=None, g_h_into=None):
        """ggnn_period_gat_aggregate_backward (include/ggnn.h).  `rcsr`: CSR of the flipped
        edge_index (grouped by source), `r_slot` [E] int32: forward CSR slot of every reverse slot.
        Returns (g_p_dst, g_p_src, g_h_src or None, g_ep [n_gates, 3, 96]); the gradient tensors
        have the layout of their operands, columns the sweep does not read are zero.
        `ep_partial_out` [>= aggregate_bwd_partials(n_dst), n_gates, 3, 96]: the per-workgroup partial sums of g_ep go
        there and g_ep is returned as None -- the caller sums them (one reduction for the sweeps of a cell).
        `g_h_into` [n_src, 96]: an earlier sweep's g_h_src of the same source node type -- this sweep ADDS its own to it in
        place (g_h_accumulate) and returns it."""
        _require_cuda(csr.rowptr, rcsr.rowptr, r_slot, einfo, p_src, p_dst, h_src, ep, agg, g_agg)
        dev = p_src.device
        E, G = csr.E, n_gates
        a = AggregateBwdArgs()
        a.rowptr, a.col, a.einfo = csr.rowptr.data_ptr(), csr.col.data_ptr(), einfo.data_ptr()
        a.p_src, a.p_dst = p_src.data_ptr(), p_dst.data_ptr()
        a.h_src = None if h_src is None else h_src.data_ptr()
        a.edge_params, a.agg, a.g_agg = ep.data_ptr(), agg.data_ptr(), g_agg.data_ptr()
        a.r_rowptr, a.r_dst, a.r_slot = rcsr.rowptr.data_ptr(), rcsr.col.data_ptr(), r_slot.data_ptr()
        if agg.stride(0) != g_agg.stride(0) or agg.shape != g_agg.shape:
            raise _lib.GGNNError("g_agg must have the layout of agg")
        n_part = self.lib.ggnn_aggregate_bwd_partials(p_dst.size(0))
        f32 = dict(dtype=torch.float32, device=dev)
        scratch = torch.empty(2, max(E, 1) * G, **f32)
        if ep_partial_out is None:
            ep_partial = torch.empty(n_part, G, _lib.GGNN_EDGE_PARAM_ROWS, 96, **f32)
        else:
            ep_partial = ep_partial_out
            if not ep_partial.is_contiguous() or ep_partial.dtype != torch.float32 \
                    or tuple(ep_partial.shape[1:]) != (G, _lib.GGNN_EDGE_PARAM_ROWS, 96) or ep_partial.size(0) < n_part:
                raise _lib.GGNNError("ep_partial_out must be a contiguous [>= partials, n_gates, 3, 96] float32 tensor")
        # out_p_dst / out_p_src: zero-initialised gradient buffers shared by the sweeps of a cell (every
        # sweep writes its own columns only)
        g_p_dst = torch.zeros_like(p_dst) if out_p_dst is None else out_p_dst
        g_p_src = torch.zeros_like(p_src) if out_p_src is None else out_p_src
        if g_p_dst.shape != p_dst.shape or g_p_src.shape != p_src.shape or g_p_dst.stride(0) != p_dst.stride(0) \
                or g_p_src.stride(0) != p_src.stride(0):
            raise _lib.GGNNError("gradient buffers must have the layout of their operands")
        if g_h_into is not None and (h_src is None or g_h_into.shape != h_src.shape or not g_h_into.is_contiguous()
                                     or g_h_into.dtype != torch.float32):
            raise _lib.GGNNError("g_h_into must be a contiguous float32 tensor of h_src's shape")
        g_h_src = None if h_src is None else (g_h_into if g_h_into is not None else torch.empty(h_src.shape, **f32))
        a.g_h_accumulate = 1 if g_h_into is not None else 0
        a.edge_alpha, a.edge_ds, a.ep_partial = scratch[0].data_ptr(), scratch[1].data_ptr(), ep_partial.data_ptr()
        a.g_p_dst, a.g_p_src = g_p_dst.data_ptr(), g_p_src.data_ptr()
        a.g_h_src = None if g_h_src is None else g_h_src.data_ptr()
        a.ldp_src, a.ldp_dst, a.ld_agg = p_src.stride(0), p_dst.stride(0), agg.stride(0)
        a.ldh_src = 0 if h_src is None else h_src.stride(0)
        if g_h_src is not None and h_src.stride(0) != g_h_src.stride(0):
            raise _lib.GGNNError("aggregate_backward: h_src rows must be contiguous (g_h_src has its layout)")
        a.n_src, a.n_dst, a.E, a.n_partials = p_src.size(0), p_dst.size(0), E, n_part
        a.v_off, a.u_off, a.u4_off, a.a_off, a.a_gstride, a.sc_off, a.n_gates = (
            v_off, u_off, u4_off, a_off, a_gstride, sc_off, n_gates)
        self._launch(self.lib.ggnn_period_gat_aggregate_backward, "ggnn_period_gat_aggregate_backward", ctypes.byref(a),
                     _lib.current_stream())
        return g_p_dst, g_p_src, g_h_src, (ep_partial.sum(0) if ep_partial_out is None else None)

    def aggregate_bwd_partials(self, n_dst):
        """Rows of the partial-sum output one backward call writes (ggnn_aggregate_bwd_partials)."""
        return self.lib.ggnn_aggregate_bwd_partials(n_dst)

    # -- gate GEMM + LSTM --------------------------------------------------------------
    @staticmethod
    def _epilogue_args(a, agg, w2, p_dst, s_off, c_in, h_out, c_out, raw_out, n_gates, mode,
                       w2_planes=None, g_stride=0):
        _require_cuda(agg, w2, p_dst, c_in, h_out, c_out, raw_out, w2_planes)
        a.agg, a.w2, a.p_dst = agg.data_ptr(), w2.data_ptr(), p_dst.data_ptr()
        a.c_in = None if c_in is None else c_in.data_ptr()
        a.h_out = None if h_out is None else h_out.data_ptr()
        a.c_out = None if c_out is None else c_out.data_ptr()
        a.raw_out = None if raw_out is None else raw_out.data_ptr()
        a.ldp, a.N = p_dst.stride(0), agg.size(0)
        a.Ka, a.s_off, a.n_gates, a.mode = w2.size(2), s_off, n_gates, mode
        if g_stride:  # agg gate stride (floats) when the rows are padded; 0 = packed [N, n_gates * Ka]
            a.g_stride, a.ld_agg = g_stride, agg.stride(0)
        if w2_planes is not None:
            if w2_planes.dtype != torch.int16 or w2_planes.numel() != 2 * w2.size(0) * _lib.GGNN_C * (w2.size(2) - 4):
                raise _lib.GGNNError("w2_planes does not match w2 (see packing.bf16_planes)")
            a.w2_planes = w2_planes.data_ptr()

    def lstm_epilogue(self, *problem, **kw):
        """(agg, w2, p_dst, s_off, c_in, h_out, c_out, raw_out, n_gates, mode, w2_planes=None, g_stride=0);
        w2_planes: packing.bf16_planes(w2) (selects the bf16x6 kernel unless GGNN_GEMM=fp32)."""
        a = EpilogueArgs()
        self._epilogue_args(a, *problem, **kw)
        self._launch(self.lib.ggnn_lstm_epilogue, "ggnn_lstm_epilogue", ctypes.byref(a), _lib.current_stream())

    def lstm_epilogue_batch(self, problems):
        """Up to four gate GEMM + LSTM problems (node types of a cell and / or both models) in one
        launch (ggnn_lstm_epilogue_batch); each item is the argument tuple of `lstm_epilogue`."""
        arr = (EpilogueArgs * len(problems))()
        for a, prob in zip(arr, problems):
            self._epilogue_args(a, *prob)
        self._launch(self.lib.ggnn_lstm_epilogue_batch, "ggnn_lstm_epilogue_batch", arr, len(problems),
                     _lib.current_stream())

    # -- LSTM update of the training path -----------------------------------------------
    def lstm_train_forward(self, z, p_dst, s_off, c_in, h_out, c_out):
        """ggnn_lstm_train_forward: z [G, N, 96] (gate GEMM output in, pre-activations out)."""
        _require_cuda(z, p_dst, c_in, h_out, c_out)
        G, N = z.size(0), z.size(1)
        if not z.is_contiguous() or z.size(2) != _lib.GGNN_C or not (h_out.is_contiguous() and c_out.is_contiguous()) \
                or (c_in is not None and not c_in.is_contiguous()):
            raise _lib.GGNNError("z [G, N, 96], c_in, h_out, c_out [N, 96] must be contiguous")
        self._launch(self.lib.ggnn_lstm_train_forward, "ggnn_lstm_train_forward", ptr(z), ptr(p_dst), p_dst.stride(0),
                     s_off, ptr(c_in), ptr(h_out), ptr(c_out), N, G, _lib.current_stream())

    def lstm_train_backward(self, z, c_in, c_out, g_h, g_c, g_z, g_p_dst, s_off, g_c_in):
        """ggnn_lstm_train_backward: g_h / g_c / g_p_dst / g_c_in may be None."""
        _require_cuda(z, c_in, c_out, g_h, g_c, g_z, g_p_dst, g_c_in)
        G, N = z.size(0), z.size(1)
        for t in (z, c_in, c_out, g_h, g_c, g_z, g_c_in):
            if t is not None and not t.is_contiguous():
                raise _lib.GGNNError("lstm_train_backward: z, g_z [G, N, 96] and the [N, 96] operands must be contiguous")
        if g_z.shape != z.shape:
            raise _lib.GGNNError("g_z must have the shape of z")
        self._launch(self.lib.ggnn_lstm_train_backward, "ggnn_lstm_train_backward", ptr(z), ptr(c_in), ptr(c_out),
                     ptr(g_h), ptr(g_c), ptr(g_z), ptr(g_p_dst), 0 if g_p_dst is None else g_p_dst.stride(0), s_off,
                     ptr(g_c_in), N, G, _lib.current_stream())

    def lstm_train_forward_batch(self, problems, n_gates):
        """ggnn_lstm_train_forward_batch: [(z [G, N, 96], p_dst, s_off, c_in or None, h_out, c_out)] -- the node types of a
        cell in one launch."""
        arr = (_lib.LstmTrainProblem * len(problems))()
        for a, (z, p_dst, s_off, c_in, h_out, c_out) in zip(arr, problems):
            _require_cuda(z, p_dst, c_in, h_out, c_out)
            if not z.is_contiguous() or z.dim() != 3 or z.size(0) != n_gates or z.size(2) != _lib.GGNN_C \
                    or not (h_out.is_contiguous() and c_out.is_contiguous()) or (c_in is not None and not c_in.is_contiguous()):
                raise _lib.GGNNError("z [G, N, 96], c_in, h_out, c_out [N, 96] must be contiguous")
            a.z, a.p_dst, a.c_in, a.h_out, a.c_out = ptr(z), ptr(p_dst), ptr(c_in), ptr(h_out), ptr(c_out)
            a.ldp, a.N, a.s_off = p_dst.stride(0), z.size(1), s_off
        self._launch(self.lib.ggnn_lstm_train_forward_batch, "ggnn_lstm_train_forward_batch", arr, len(problems), n_gates,
                     _lib.current_stream())

    def lstm_train_backward_batch(self, problems, n_gates):
        """ggnn_lstm_train_backward_batch: [(z, c_in, c_out, g_h, g_c, g_z, g_p_dst, s_off, g_c_in, pad_off, pad_n)]; the
        call also zeroes g_p_dst[:, pad_off : pad_off + pad_n] (pad_n <= 96, multiples of 4)."""
        arr = (_lib.LstmTrainProblem * len(problems))()
        for a, (z, c_in, c_out, g_h, g_c, g_z, g_p_dst, s_off, g_c_in, pad_off, pad_n) in zip(arr, problems):
            _require_cuda(z, c_in, c_out, g_h, g_c, g_z, g_p_dst, g_c_in)
            for t in (z, c_in, c_out, g_h, g_c, g_z, g_c_in):
                if t is not None and not t.is_contiguous():
                    raise _lib.GGNNError("lstm_train_backward: z, g_z [G, N, 96] and the [N, 96] operands must be contiguous")
            if g_z.shape != z.shape or z.dim() != 3 or z.size(0) != n_gates:
                raise _lib.GGNNError("g_z must have the shape of z [G, N, 96]")
            a.z, a.c_in, a.c_out, a.g_h, a.g_c = ptr(z), ptr(c_in), ptr(c_out), ptr(g_h), ptr(g_c)
            a.g_z, a.g_p_dst, a.g_c_in = ptr(g_z), ptr(g_p_dst), ptr(g_c_in)
            a.ldp, a.N, a.s_off = 0 if g_p_dst is None else g_p_dst.stride(0), z.size(1), s_off
            a.pad_off, a.pad_n = pad_off, pad_n
        self._launch(self.lib.ggnn_lstm_train_backward_batch, "ggnn_lstm_train_backward_batch", arr, len(problems), n_gates,
                     _lib.current_stream())

    def train_input_rows(self, problems):
        """ggnn_train_input_rows: [(x [N, >= F], F)] -> [out [N, roundup4(F) + 4] = [x[:, :F] | 0 .. | 1 0 0 0]] in one launch."""
        arr = (_lib.TrainRowsProblem * len(problems))()
        outs = []
        for a, (x, F) in zip(arr, problems):
            _require_cuda(x)
            if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1 or x.size(1) < F:
                raise _lib.GGNNError("ggnn_train_input_rows: x must be float32 [N, >= F] with unit column stride")
            out = torch.empty(x.size(0), ((F + 3) & ~3) + 4, dtype=torch.float32, device=x.device)
            outs.append(out)
            a.x, a.out, a.ldx, a.ldo, a.N, a.F = ptr(x), ptr(out), x.stride(0), out.stride(0), x.size(0), F
        self._launch(self.lib.ggnn_train_input_rows, "ggnn_train_input_rows", arr, len(problems), _lib.current_stream())
        return outs

    def wgrad(self, a, b, K, M, Nc, lda, ldb, batch=1, a_bstride=0, b_bstride=0, b_ins=None, ins_off=0, defer=None):
        """ggnn_wgrad: C[batch, M, Nc] = A_k^T B_k; A_k = K rows of M floats, row pitch lda, from element
        k * a_bstride of the contiguous tensor `a` on (B_k likewise).  Returns the sum over the splits.
        `b_ins` [K, w] (contiguous): B is `b` with these w columns inserted at column ins_off (Nc counts them).
        `defer` (a list): the sum over the splits is NOT made by this call -- the returned tensor is filled by
        `sum_rows_batch(defer)`, which the caller runs once for several postponed reductions."""
        _require_cuda(a, b, b_ins)
        ins_w = 0
        if b_ins is not None:
            if b_ins.dtype != torch.float32 or b_ins.dim() != 2 or not b_ins.is_contiguous() or b_ins.size(0) < K or batch != 1:
                raise _lib.GGNNError("wgrad: b_ins must be a contiguous float32 [K, w] matrix (batch 1)")
            ins_w = b_ins.size(1)
        for t, ld, bs, w in ((a, lda, a_bstride, M), (b, ldb, b_bstride, Nc - ins_w)):
            if t.dtype != torch.float32 or not t.is_contiguous() or ld < w \
                    or (batch - 1) * bs + (K - 1) * ld + w > t.numel():
                raise _lib.GGNNError("wgrad: operands must be contiguous float32 tensors that hold every addressed row")
        S = self.lib.ggnn_wgrad_splits(K, M, Nc, batch)
        partial = torch.empty(S, batch, M, Nc, dtype=torch.float32, device=a.device)
        out = torch.empty(batch, M, Nc, dtype=torch.float32, device=a.device) if S > 1 else None
        w = _lib.WgradArgs()
        w.a, w.b, w.partial, w.out = a.data_ptr(), b.data_ptr(), partial.data_ptr(), (None if defer is not None else ptr(out))
        if defer is not None and S > 1:
            defer.append((partial.view(1, S, -1), out.view(1, -1)))
        w.lda, w.ldb, w.a_bstride, w.b_bstride, w.K = lda, ldb, a_bstride, b_bstride, K
        w.M, w.Nc, w.batch, w.n_split = M, Nc, batch, S
        if b_ins is not None:
            w.b_ins, w.ld_ins, w.ins_off, w.ins_w = b_ins.data_ptr(), b_ins.stride(0), ins_off, ins_w
        self._launch(self.lib.ggnn_wgrad, "ggnn_wgrad", ctypes.byref(w), _lib.current_stream())
        return out if S > 1 else partial[0]

    @staticmethod
    def _rowgemm_weight(g, w, K, n_out, batch, transposed, bf16):
        if w.dim() == 2:
            w = w.unsqueeze(0)
        if w.dtype != torch.float32 or w.dim() != 3 or w.stride(2) != 1 or w.size(0) != batch:
            raise _lib.GGNNError("ggnn_rowgemm: w must be a float32 [batch, rows, cols] view with unit column stride")
        if (transposed and (w.size(1) < K or w.size(2) < n_out)) or (not transposed and (w.size(1) < n_out or w.size(2) < K)):
            raise _lib.GGNNError("ggnn_rowgemm: w does not hold [n_out, K] (or its transpose)")
        g.w, g.w_bstride = w.data_ptr(), (w.stride(0) if batch > 1 else 0)
        g.w_nstride, g.w_kstride = (1, w.stride(1)) if transposed else (w.stride(1), 1)
        g.K, g.n_out, g.batch, g.precision = K, n_out, batch, _lib.GGNN_PRECISION_BF16 if bf16 else 0

    def rowgemm_pack(self, weights):
        """ggnn_rowgemm_pack: the weight planes of several ggnn_rowgemm products in ONE launch.  `weights` = [(w, K, n_out,
        batch, transposed, bf16)] as `rowgemm` takes them; returns one plane tensor per product for `rowgemm(..., planes=)`
        (fresh tensors: they may be kept until the backward pass)."""
        out = []
        for i0 in range(0, len(weights), _lib.GGNN_ROWGEMM_MAX_PACK):
            chunk = weights[i0:i0 + _lib.GGNN_ROWGEMM_MAX_PACK]
            arr = (_lib.RowGemmArgs * len(chunk))()
            for g, (w, K, n_out, batch, transposed, bf16) in zip(arr, chunk):
                _require_cuda(w)
                self._rowgemm_weight(g, w, K, n_out, batch, transposed, bf16)
                nbytes = self.lib.ggnn_rowgemm_workspace_bytes(K, n_out, batch)
                ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=w.device)
                g.workspace, g.workspace_bytes = ws.data_ptr(), nbytes
                out.append(ws)
            self._launch(self.lib.ggnn_rowgemm_pack, "ggnn_rowgemm_pack", arr, len(chunk), _lib.current_stream())
        return out

    def _rowgemm_args(self, a, w, out, K, n_out, batch, c_in, transposed, bf16, planes):
        """The filled ggnn_rowgemm_args of one product (checks included) and the tensor holding its weight planes."""
        _require_cuda(a, w, out, c_in)
        if a.dim() == 2:
            a = a.unsqueeze(0)
        if out.dim() == 2:
            out = out.unsqueeze(0)
        if c_in is not None and c_in.dim() == 2:
            c_in = c_in.unsqueeze(0)
        for t, name in ((a, "a"), (out, "out"), (c_in, "c_in")):
            if t is not None and (t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1):
                raise _lib.GGNNError(f"ggnn_rowgemm: {name} must be a float32 [batch, rows, cols] view with unit column stride")
        M = a.size(1)
        if a.size(0) != batch or out.size(0) != batch or out.size(1) != M or a.size(2) < K \
                or out.size(2) < n_out or (c_in is not None and (c_in.shape != out.shape or c_in.stride() != out.stride())):
            raise _lib.GGNNError("ggnn_rowgemm: operand shapes do not match (batch, M, K, n_out)")
        g = _lib.RowGemmArgs()
        self._rowgemm_weight(g, w, K, n_out, batch, transposed, bf16)
        nbytes = self.lib.ggnn_rowgemm_workspace_bytes(K, n_out, batch)
        if planes is not None:
            if planes.dtype != torch.uint8 or planes.numel() < nbytes or planes.device != a.device:
                raise _lib.GGNNError("ggnn_rowgemm: `planes` is not the rowgemm_pack output of this product")
            ws, g.prepacked = planes, 1
        else:
            # one plane buffer per (device, stream, size): the calls on a stream are ordered
            key = (a.device, torch.cuda.current_stream(a.device).cuda_stream, nbytes)
            ws = self._rowgemm_ws.get(key)
            if ws is None:
                ws = self._rowgemm_ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=a.device)
        g.a, g.c = a.data_ptr(), out.data_ptr()
        g.c_in = None if c_in is None else c_in.data_ptr()
        g.workspace, g.workspace_bytes = ws.data_ptr(), nbytes
        g.M, g.lda, g.ldc = M, a.stride(1), out.stride(1)
        g.a_bstride, g.c_bstride = a.stride(0) if batch > 1 else 0, out.stride(0) if batch > 1 else 0
        return g, ws

    def rowgemm(self, a, w, out, K, n_out, batch=1, c_in=None, transposed=False, bf16=False, planes=None):
        """ggnn_rowgemm (include/ggnn.h): out[b] = a[b] . W[b]^T (+ c_in[b]) for b < batch.
        `a`   : [M, lda] (batch 1) or [batch, M, lda] float32 view with unit column stride: the first K columns;
        `w`   : [batch, n_out, >= K] (transposed=False: W[b] = w[b][:, :K]) or [batch, K, >= n_out] (transposed=True:
                W[b] = w[b][:, :n_out]^T -- a gradient uses the transpose of the forward's weight), or 2-D for batch 1;
        `out` : [M, ldc] / [batch, M, ldc] float32 view with unit column stride: its first n_out columns are written
                (returned as given: a [M, n] result must not come back as [1, M, n] -- autograd would sum_to_size it);
        `c_in`: optional, same layout as out (may be out itself);
        `planes`: the weight planes of exactly this (w, K, n_out, batch, transposed, bf16) from `rowgemm_pack`."""
        g, _ = self._rowgemm_args(a, w, out, K, n_out, batch, c_in, transposed, bf16, planes)
        self._launch(self.lib.ggnn_rowgemm, "ggnn_rowgemm", ctypes.byref(g), _lib.current_stream())
        return out

    def rowgemm_pair(self, first, second):
        """ggnn_rowgemm_pair: two LONG products side by side in one grid; each = (a, w, out, K, n_out, c_in, transposed, bf16,
        planes) with the planes from `rowgemm_pack` (batch 1).  Falls back to two calls where the C entry point would refuse
        (a product whose weight planes stay in LDS, different output widths or precisions)."""
        args = [self._rowgemm_args(a, w, out, K, n_out, 1, c_in, transposed, bf16, planes)[0]
                for (a, w, out, K, n_out, c_in, transposed, bf16, planes) in (first, second)]
        tiles = lambda n: 6 if n <= 96 else (8 if n <= 128 else 14)
        streams = lambda g: not (g.K // 32 <= (8 if g.n_out == 96 else 4) and g.n_out in (96, 128, 224)) \
            and not (g.precision == 0 and g.n_out > 128)
        if all(g.prepacked and streams(g) for g in args) and tiles(args[0].n_out) == tiles(args[1].n_out) \
                and args[0].precision == args[1].precision:
            arr = (_lib.RowGemmArgs * 2)(*args)
            self._launch(self.lib.ggnn_rowgemm_pair, "ggnn_rowgemm_pair", arr, _lib.current_stream())
        else:
            for g in args:
                self._launch(self.lib.ggnn_rowgemm, "ggnn_rowgemm", ctypes.byref(g), _lib.current_stream())
        return first[2], second[2]

    def sum_rows(self, t):
        """ggnn_sum_rows: [batch, rows, cols] contiguous float32 -> [batch, cols], the sum over the rows in a fixed order."""
        _require_cuda(t)
        if t.dtype != torch.float32 or t.dim() != 3 or not t.is_contiguous() or t.size(2) % 4:
            raise _lib.GGNNError("ggnn_sum_rows: a contiguous float32 [batch, rows, cols] tensor with cols % 4 == 0")
        out = torch.empty(t.size(0), t.size(2), dtype=torch.float32, device=t.device)
        self._launch(self.lib.ggnn_sum_rows, "ggnn_sum_rows", t.data_ptr(), out.data_ptr(), t.size(1), t.size(2), t.size(0),
                     _lib.current_stream())
        return out

    def sum_rows_batch(self, problems):
        """ggnn_sum_rows_batch: [(in [batch, rows, cols], out [batch, cols])] contiguous float32 -- out = the sum over the rows,
        all problems in one launch (chunks of GGNN_SUM_ROWS_MAX)."""
        for i0 in range(0, len(problems), _lib.GGNN_SUM_ROWS_MAX):
            chunk = problems[i0:i0 + _lib.GGNN_SUM_ROWS_MAX]
            arr = (_lib.SumRowsProblem * len(chunk))()
            for a, (t, out) in zip(arr, chunk):
                _require_cuda(t, out)
                if t.dtype != torch.float32 or t.dim() != 3 or not t.is_contiguous() or t.size(2) % 4 or out.dtype != torch.float32 \
                        or not out.is_contiguous() or out.numel() != t.size(0) * t.size(2):
                    raise _lib.GGNNError("ggnn_sum_rows_batch: contiguous float32 [batch, rows, cols] (cols % 4 == 0) -> [batch, cols]")
                a.in_, a.out, a.n_rows, a.n_cols, a.batch = t.data_ptr(), out.data_ptr(), t.size(1), t.size(2), t.size(0)
            self._launch(self.lib.ggnn_sum_rows_batch, "ggnn_sum_rows_batch", arr, len(chunk), _lib.current_stream())

    def adam_step(self, args):
        """ggnn_adam_step (include/ggnn.h) on a filled `_lib.AdamArgs` (training.FusedAdam builds it)."""
        self._launch(self.lib.ggnn_adam_step, "ggnn_adam_step", ctypes.byref(args), _lib.current_stream())

    @staticmethod
    def _pack_args(plan, flat2, kq, packed=None, params=None):
        a = _lib.PackArgs()
        if params is None:
            a.flat2, a.kq, a.kq_idx, a.idx3 = flat2.data_ptr(), kq.data_ptr(), plan.kq_idx.data_ptr(), plan.idx3.data_ptr()
        else:   # the parameters read where they lie: encoded index tables + the DEVICE table of their addresses
            if params.dtype != torch.int64 or params.numel() != len(plan.sizes) or not params.is_cuda:
                raise _lib.GGNNError("ggnn_pack_weights: params must be a device int64 table with one address per parameter tensor")
            a.flat2, a.kq, a.kq_idx, a.idx3 = flat2.data_ptr(), kq.data_ptr(), plan.kq_idx_enc.data_ptr(), plan.idx3_enc.data_ptr()
            a.params = params.data_ptr()
        a.packed = 0 if packed is None else packed.data_ptr()
        a.n_flat, a.zero, a.n_packed = plan.n_flat, plan.zero, plan.n_packed
        a.nb, a.r, a.c, a.L = plan.mr_shape[0], plan.mr_shape[1], plan.mr_shape[2], plan.idx3.size(1)
        a.coef = plan.k_coef
        return a

    def pack_weights(self, plan, flat2, kq, packed, params=None):
        """ggnn_pack_weights: `flat2` [plan.n_flat2] holds the parameters in its first n_flat entries -- or, with `params` (a
        device int64 table of the parameter tensors' addresses), they are read where they lie and that part of flat2 is not
        touched; fills the operands `kq` [plan.n_kq], the products and `packed` [plan.n_packed] (train_pack._PackWeights)."""
        self.pack_weights_batch([(plan, flat2, kq, packed, params)])

    def pack_weights_batch(self, cells):
        """ggnn_pack_weights_batch: [(plan, flat2, kq, packed, params)] -- the cells of a step with every launch shared."""
        arr = (_lib.PackArgs * len(cells))()
        for k, (plan, flat2, kq, packed, params) in enumerate(cells):
            _require_cuda(flat2, kq, packed)
            if flat2.dtype != torch.float32 or packed.dtype != torch.float32 or kq.dtype != torch.float32 \
                    or flat2.numel() != plan.n_flat2 or kq.numel() != plan.n_kq or packed.numel() != plan.n_packed \
                    or not flat2.is_contiguous() or not packed.is_contiguous() or not kq.is_contiguous():
                raise _lib.GGNNError("ggnn_pack_weights: flat2 [n_flat2], kq [n_kq] and packed [n_packed] must be contiguous float32")
            arr[k] = self._pack_args(plan, flat2, kq, packed, params)
        self._launch(self.lib.ggnn_pack_weights_batch, "ggnn_pack_weights_batch", arr, len(cells), _lib.current_stream())

    def pack_weights_backward(self, plan, flat2, kq, grads, g_flat2, g_kq, g_flat):
        """ggnn_pack_weights_backward: `grads` = the gradients of the nine packed outputs (None: zero), contiguous float32 of
        plan.out_sizes; workspaces g_flat2 [n_flat2], g_kq [n_kq]; g_flat [>= n_flat] receives the parameters' gradient in its
        first n_flat entries and zeros behind them (n_tail: the parameters without effect)."""
        self.pack_weights_backward_batch([(plan, flat2, kq, grads, g_flat2, g_kq, g_flat)])

    def pack_weights_backward_batch(self, cells):
        """ggnn_pack_weights_backward_batch: [(plan, flat2, kq, grads, g_flat2, g_kq, g_flat)], every launch shared."""
        arr = (_lib.PackBwdArgs * len(cells))()
        for k, (plan, flat2, kq, grads, g_flat2, g_kq, g_flat) in enumerate(cells):
            self._pack_bwd_args(arr[k], plan, flat2, kq, grads, g_flat2, g_kq, g_flat)
        self._launch(self.lib.ggnn_pack_weights_backward_batch, "ggnn_pack_weights_backward_batch", arr, len(cells),
                     _lib.current_stream())

    def _pack_bwd_args(self, b, plan, flat2, kq, grads, g_flat2, g_kq, g_flat):
        _require_cuda(flat2, g_flat2, g_kq, g_flat, *[g for g in grads if g is not None])
        if len(grads) != _lib.GGNN_PACK_OUTPUTS or g_flat2.numel() != plan.n_flat2 or g_kq.numel() != plan.n_kq \
                or g_flat.numel() < plan.n_flat or not g_flat.is_contiguous():
            raise _lib.GGNNError("ggnn_pack_weights_backward: nine output gradients and workspaces of the plan's sizes")
        b.fwd = self._pack_args(plan, flat2, kq)
        b.fwd.packed = g_flat.data_ptr()   # (unused by the backward; must not be NULL)
        off = 0
        for s, (g, n) in enumerate(zip(grads, plan.out_sizes)):
            if g is not None and (g.dtype != torch.float32 or g.numel() != n):
                raise _lib.GGNNError("ggnn_pack_weights_backward: output gradients must be float32 of the outputs' sizes")
            b.g_out[s] = None if g is None else g.data_ptr()
            b.g_off[s] = off
            off += n
            if g is None or g.is_contiguous():
                b.g_w[s], b.g_rs[s], b.g_cs[s] = max(n, 1), 0, 1
            elif g.dim() == 2:          # a column block of a wider matrix (or of its transpose): read in place
                b.g_w[s], b.g_rs[s], b.g_cs[s] = g.size(1), g.stride(0), g.stride(1)
            elif g.dim() == 1:          # a column of a matrix
                b.g_w[s], b.g_rs[s], b.g_cs[s] = 1, g.stride(0), 1
            else:
                raise _lib.GGNNError("ggnn_pack_weights_backward: a non-contiguous output gradient must be 1-D or 2-D")
        b.g_off[_lib.GGNN_PACK_OUTPUTS] = off
        b.inv, b.inv_kq = plan.inv.data_ptr(), plan.inv_kq.data_ptr()
        b.g_flat2, b.g_kq, b.g_flat = g_flat2.data_ptr(), g_kq.data_ptr(), g_flat.data_ptr()
        b.n_flat2, b.n_kq, b.inv_m, b.inv_kq_m = plan.n_flat2, plan.n_kq, plan.inv.size(1), plan.inv_kq.size(1)
        b.n_tail = g_flat.numel() - plan.n_flat

    def masked_mse(self, terms, scale, loss, want_grad=True):
        """ggnn_masked_mse: `terms` = [(pred, target, mask or None)], contiguous float32 CUDA tensors; mask has the shape
        of pred or one entry per leading row of it.  Returns the gradients [g_pred] (None each with want_grad=False)."""
        a = _lib.MseArgs()
        if not 1 <= len(terms) <= _lib.GGNN_MSE_MAX_TERMS:
            raise _lib.GGNNError("ggnn_masked_mse: 1..%d terms" % _lib.GGNN_MSE_MAX_TERMS)
        grads = []
        for k, (p, y, m) in enumerate(terms):
            _require_cuda(p, y, m)
            for t in (p, y, m):
                if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
                    raise _lib.GGNNError("ggnn_masked_mse: operands must be contiguous float32 tensors")
            if y.shape != p.shape:
                raise _lib.GGNNError("ggnn_masked_mse: pred and target differ in shape")
            div = 1
            if m is not None and m.numel() != p.numel():
                if p.dim() < 1 or m.numel() != p.size(0) or p.numel() % max(m.numel(), 1):
                    raise _lib.GGNNError("ggnn_masked_mse: mask must match pred or its leading dimension")
                div = p.numel() // m.numel()
            g = torch.empty_like(p) if want_grad else None
            grads.append(g)
            a.pred[k], a.target[k], a.mask[k], a.g_pred[k] = p.data_ptr(), y.data_ptr(), ptr(m), ptr(g)
            a.n[k], a.mask_div[k] = p.numel(), div
        # one workspace per (device, stream): two losses in flight on different streams (a capture on a side stream beside
        # eager work, two models training in threads) must not share the arrival counter and the partial sums
        key = (loss.device, torch.cuda.current_stream(loss.device).cuda_stream)
        ws = self._mse_ws.get(key)
        if ws is None:
            ws = self._mse_ws[key] = torch.zeros(_lib.GGNN_MSE_BLOCKS + 1, dtype=torch.float64, device=loss.device)
        a.workspace, a.loss, a.scale, a.n_terms = ws.data_ptr(), loss.data_ptr(), scale, len(terms)
        self._launch(self.lib.ggnn_masked_mse, "ggnn_masked_mse", ctypes.byref(a), _lib.current_stream())
        return grads

    # -- heads -------------------------------------------------------------------------
    def heads_regressor(self, h_joint, h_grain, x_grain, w, b, y_joint, y_grain, grain_area):
        _require_cuda(h_joint, h_grain, x_grain, w, b, y_joint, y_grain, grain_area)
        check(self.lib.ggnn_heads_regressor(ptr(h_joint), h_joint.size(0), ptr(h_grain),
                                            h_grain.size(0), ptr(x_grain), x_grain.stride(0), ptr(w),
                                            ptr(b), ptr(y_joint), ptr(y_grain), ptr(grain_area),
                                            _lib.current_stream()), "ggnn_heads_regressor")

    def heads_regressor_backward(self, w, y_joint, y_grain, g_y_joint, g_y_grain, g_grain_area):
        """ggnn_heads_regressor_backward -> (g_pre_joint [n, 4], g_pre_grain [n, 4], g_h_joint, g_h_grain [n, 96]);
        the three incoming gradients may be None."""
        _require_cuda(w, y_joint, y_grain, g_y_joint, g_y_grain, g_grain_area)
        for t in (w, y_joint, y_grain, g_y_joint, g_y_grain, g_grain_area):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
                raise _lib.GGNNError("heads_regressor_backward: contiguous float32 operands")
        nj, ng = y_joint.size(0), y_grain.size(0)
        f32 = dict(dtype=torch.float32, device=w.device)
        gpj, gpg = torch.empty(nj, 4, **f32), torch.empty(ng, 4, **f32)
        ghj, ghg = torch.empty(nj, 96, **f32), torch.empty(ng, 96, **f32)
        self._launch(self.lib.ggnn_heads_regressor_backward, "ggnn_heads_regressor_backward", nj, ng, ptr(w), ptr(y_joint),
                     ptr(y_grain), ptr(g_y_joint), ptr(g_y_grain), ptr(g_grain_area), ptr(gpj), ptr(gpg), ptr(ghj),
                     ptr(ghg), _lib.current_stream())
        return gpj, gpg, ghj, ghg

    def heads_regressor_update(self, h_joint, h_grain, x_joint, x_grain, w, b, y_joint, y_grain, grain_area, dz, zmax,
                               flags):
        """heads_regressor + step_update in one launch (ggnn_heads_regressor_update)."""
        _require_cuda(h_joint, h_grain, x_joint, x_grain, w, b, y_joint, y_grain, grain_area, flags)
        self._launch(self.lib.ggnn_heads_regressor_update, "ggnn_heads_regressor_update", ptr(h_joint), h_joint.size(0),
                     ptr(h_grain), h_grain.size(0), ptr(x_joint), x_joint.stride(0), ptr(x_grain), x_grain.stride(0),
                     x_grain.size(1), ptr(w), ptr(b), ptr(y_joint), ptr(y_grain), ptr(grain_area), dz, zmax,
                     ptr(flags), _lib.current_stream())

    def step_refresh_prepare(self, x_joint, x_grain, zmax, flags, items, mirror=None):
        """step_refresh + edge_prepare of the next forward in one launch (ggnn_step_refresh_prepare); items as
        edge_prepare: (csr, edge_attr [E] COO order -- WRITTEN here --, x_src, x_dst, einfo_out).  `mirror` = (x_joint',
        x_grain'): second copies of the node features as they stand behind this call (include/ggnn.h)."""
        _require_cuda(x_joint, x_grain, flags)
        mj = mg = None
        if mirror is not None:
            mj, mg = mirror
            _require_cuda(mj, mg)
            if mj.shape != x_joint.shape or mg.shape != x_grain.shape or mj.stride() != x_joint.stride() \
                    or mg.stride() != x_grain.stride() or mj.dtype != torch.float32 or mg.dtype != torch.float32:
                raise _lib.GGNNError("mirror tensors must have the shape and strides of x_joint / x_grain")
        arr = (PrepareEdge * max(len(items), 1))()
        for k, (csr, ea, xs, xd, einfo) in enumerate(items):
            _require_cuda(csr.col, ea, xs, xd, einfo)
            if einfo.size(0) < ea.numel() + _lib.GGNN_UNIT_EDGES or einfo.size(1) != _lib.GGNN_EINFO_ROW:
                raise _lib.GGNNError("einfo must be [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW]")
            a = arr[k]
            a.col, a.perm, a.row = csr.col.data_ptr(), csr.perm.data_ptr(), csr.row.data_ptr()
            a.edge_attr, a.x_src, a.x_dst = ea.data_ptr(), xs.data_ptr(), xd.data_ptr()
            a.einfo = einfo.data_ptr()
            a.ldx_src, a.ldx_dst, a.E, a.f_src = xs.stride(0), xd.stride(0), ea.numel(), xs.size(1)
            a.E_dev = ptr(getattr(csr, "E_dev", None))   # (a topology that shrinks in place under captured launches)
        self._launch(self.lib.ggnn_step_refresh_prepare, "ggnn_step_refresh_prepare", ptr(x_joint), x_joint.size(0),
                     x_joint.stride(0), ptr(x_grain), x_grain.size(0), x_grain.stride(0), zmax, ptr(flags), arr,
                     len(items), None if mj is None else ptr(mj), None if mg is None else ptr(mg), _lib.current_stream())

    def heads_classifier(self, h_joint, edge_index_jj, edge_attr_jj, w_node, w_edge, node_tmp,
                         edge_event, edge, E_dev=None):
        """`E_dev` (int64 [1], device): the number of edges at RUN time (ggnn_heads_classifier_n; edge_index_jj's own width is
        then the capacity the launch is sized for)."""
        _require_cuda(h_joint, edge_index_jj, edge_attr_jj, w_node, w_edge, node_tmp, edge_event, edge, E_dev)
        E = edge_index_jj.size(1)
        check(self.lib.ggnn_heads_classifier_n(ptr(h_joint), h_joint.size(0), ptr(edge_index_jj), E, ptr(E_dev),
                                               ptr(edge_attr_jj), ptr(w_node), ptr(w_edge),
                                               ptr(node_tmp), ptr(edge_event), ptr(edge),
                                               _lib.current_stream()), "ggnn_heads_classifier")

    # -- rollout-step glue -------------------------------------------------------------
    def step_update(self, x_joint, x_grain, y_joint, y_grain, dz, zmax, flags):
        _require_cuda(x_joint, x_grain, y_joint, y_grain, flags)
        check(self.lib.ggnn_step_update(ptr(x_joint), x_joint.size(0), x_joint.stride(0),
                                        ptr(x_grain), x_grain.size(0), x_grain.stride(0),
                                        x_grain.size(1), ptr(y_joint), ptr(y_grain), dz, zmax,
                                        ptr(flags), _lib.current_stream()), "ggnn_step_update")

    def grain_centres(self, csr_jg, x_joint, x_grain, domain_factor=1.0, domain_offset=None, centres_before=None):
        """x_grain[:, :2] <- region centres of the grains' junction polygons (graph.update(),
        graph_datastruct.py:681-708 + test.py:556-559).  csr_jg: CSR of (joint, pull, grain).  centres_before
        ([n_grain, 2] fp32, contiguous): receives x_grain[:, :2] as the call found them."""
        _require_cuda(x_joint, x_grain, csr_jg.rowptr, domain_offset, centres_before)
        if centres_before is not None:
            _f32c(centres_before, "centres_before")
            if tuple(centres_before.shape) != (x_grain.size(0), 2):
                raise _lib.GGNNError("centres_before must be [n_grain, 2]")
        if csr_jg.rowptr.numel() != x_grain.size(0) + 1:
            raise _lib.GGNNError("csr_jg must have one row per grain")
        if domain_offset is not None:
            _f32c(domain_offset, "domain_offset")
            if tuple(domain_offset.shape) != (x_joint.size(0), 2):
                raise _lib.GGNNError("domain_offset must be [n_joint, 2]")
        elif domain_factor > 1:
            raise _lib.GGNNError("domain_factor > 1 needs the domain_offset of scale_feature_patchs")
        check(self.lib.ggnn_grain_centres(ptr(csr_jg.rowptr), ptr(csr_jg.col), ptr(x_joint),
                                          x_joint.size(0), x_joint.stride(0),
                                          ptr(domain_offset) if domain_offset is not None else None,
                                          float(domain_factor), ptr(x_grain), x_grain.size(0),
                                          x_grain.stride(0), ptr(centres_before), _lib.current_stream()),
              "ggnn_grain_centres")

    def detect_events(self, grain_area, live_grain, area_threshold, edge_event, edge_index_jj,
                      logit_threshold, flags, range_word=None, E_dev=None):
        """flags[0:2] (int32, device) <- (#grain events, #switch candidates); with `range_word` (int32 [1]) flags[2] <- the
        word, which is cleared; `E_dev` (int64 [1], device): the number of junction edges at RUN time; see ggnn.h."""
        _require_cuda(grain_area, live_grain, edge_event, edge_index_jj, flags, range_word, E_dev)
        if live_grain.dtype != torch.int32 or flags.dtype != torch.int32 or flags.numel() < (2 if range_word is None else 3):
            raise _lib.GGNNError("live_grain / flags must be int32 (flags: two words, three with a range word)")
        if range_word is not None and (range_word.dtype != torch.int32 or range_word.numel() < 1):
            raise _lib.GGNNError("range_word must be an int32 word")
        check(self.lib.ggnn_detect_events_n(ptr(grain_area), ptr(live_grain), grain_area.numel(),
                                            float(area_threshold), ptr(edge_event), ptr(edge_index_jj),
                                            edge_index_jj.size(1), ptr(E_dev), float(logit_threshold), ptr(flags),
                                            ptr(range_word), _lib.current_stream()), "ggnn_detect_events")

    def step_refresh(self, x_joint, x_grain, zmax, flags, edges):
        """edges: list of (edge_index [2,E] int64, x_src, x_dst, edge_attr_out [E][, E_dev int64 [1] or None])."""
        _require_cuda(x_joint, x_grain, flags)
        arr = (RefreshEdge * max(len(edges), 1))()
        for k, (ei, xs, xd, ea, *rest) in enumerate(edges):
            _require_cuda(ei, xs, xd, ea)
            arr[k].E_dev = ptr(rest[0]) if rest else None
            arr[k].edge_index, arr[k].x_src, arr[k].x_dst = ei.data_ptr(), xs.data_ptr(), xd.data_ptr()
            arr[k].edge_attr = ea.data_ptr()
            arr[k].ldx_src, arr[k].ldx_dst = xs.stride(0), xd.stride(0)
            arr[k].n_src, arr[k].n_dst, arr[k].E = xs.size(0), xd.size(0), ei.size(1)
        check(self.lib.ggnn_step_refresh(ptr(x_joint), x_joint.size(0), x_joint.stride(0),
                                         ptr(x_grain), x_grain.size(0), x_grain.stride(0), zmax,
                                         ptr(flags), arr, len(edges), _lib.current_stream()),
              "ggnn_step_refresh")


_default = None


def default_backend():
    global _default
    if _default is None:
        _default = HipBackend()
    return _default
