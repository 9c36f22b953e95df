"""Static-topology rollout driver: the hot loop of test.py:353-577 with everything that
runs per step kept on the device.

One step = Rmodel.forward + Cmodel.forward (test.py:382-383) + Rmodel.update (:400)
+ z advance and clamp (:401-407) + [grain-centre refresh, :468-478 + :556-559 via
graph.update(), when `refresh_centres=True`] + edge-length refresh (:562-575).  Grain-event /
edge-event topology surgery (Cmodel.update, :426) is host code outside this path; the topology
is therefore static here.

The whole step is 13 kernel launches (the regressor and the classifier share every launch of
their cells) with no host synchronisation and no allocation, so it can be replayed from a hipGraph
(`use_graph=True`) to remove launch overhead on small graphs.
"""
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .backend import default_backend
from .engine import (Workspace, _check_x, alloc_einfo, graph_for, prepare_edges, run_encoder_decoder,
                     run_encoder_decoder_multi)
from .modules import _param_version, _param_version_sample
from .packing import EDGE_TYPES, NODE_TYPES, pack_classifier_heads, pack_regressor_heads

TRAIN_FRAMES = 120  # test.py:190
ET_JJ = ("joint", "connect", "joint")


class GrainRollout:
    JOINT_LAUNCH_MAX_JOINTS = 8000

    def __init__(self, rmodel, cmodel, x_dict: Dict[str, torch.Tensor], edge_index_dict,
                 edge_attr_dict, span: int, use_graph: bool = False, concurrent: bool = True,
                 refresh_centres: bool = False,
                 domain_factor: float = 1.0, domain_offset: Optional[torch.Tensor] = None,
                 joint_launches: Optional[bool] = None):
        """refresh_centres: also recompute x_grain[:, :2] from the junction polygons every step,
        like the reference's traj.GNN_update + test.py:556-559 (default off = the static-geometry
        goldens).  domain_factor / domain_offset: `geometry_scaling` of test.py:310-312 when the
        domain was folded by scale_feature_patchs (offset [n_joint, 2], floor of the scaled xy).
        joint_launches: the regressor and the classifier see the same x, graph and edge geometry,
        so every stage of their cells can go out as ONE launch for both (13 launches per step);
        False = one set of launches per model, on two streams when `concurrent`.  Default (None):
        joint launches below JOINT_LAUNCH_MAX_JOINTS junctions, where a step is launch-bound
        (cfg2: 0.139 vs 0.162 ms per step); above it every kernel fills the chip by itself and the
        two-stream plan wins by overlapping kernels of different kinds -- one model's matrix-bound
        projection beside the other's memory-bound sweep (cfg3: 0.61 vs 0.66 ms per step)."""
        self.be = default_backend()
        self.rmodel, self.cmodel = rmodel, cmodel
        self.x = {nt: x_dict[nt] for nt in NODE_TYPES}  # mutated in place, like the reference
        for nt in NODE_TYPES:
            _check_x(self.x[nt], rmodel.in_channels_dict[nt], nt)
            if not self.x[nt].is_contiguous():
                raise _lib.GGNNError("x_dict tensors must be contiguous")
        dev = self.x["joint"].device
        self.n_nodes = {nt: self.x[nt].size(0) for nt in NODE_TYPES}
        self._set_topology(edge_index_dict, edge_attr_dict)
        self.span = span
        # test.py:401-406 computes in fp32: z += fp32(span/121); clamp at fp32(120/121)
        self.dz = float(np.float32(span / (TRAIN_FRAMES + 1)))
        self.zmax = float(np.float32(TRAIN_FRAMES / (TRAIN_FRAMES + 1)))
        self.flags = torch.zeros(2, dtype=torch.int32, device=dev)
        self.packed = {}
        self.ws = {}
        self._pack_weights()
        # this rollout's own range-flag word (include/ggnn.h, OPERAND RANGE): the fused cells of its launches report here,
        # so that two rollouts on a device neither consume nor raise each other's reports
        self._range_word = torch.zeros(1, dtype=torch.int32, device=dev)
        for name in ("R", "C"):
            self.ws[name] = Workspace(*self.packed[name], self.n_nodes, dev)
            self.ws[name].range_flag = self._range_word
        nj, ng = self.n_nodes["joint"], self.n_nodes["grain"]
        f32 = dict(dtype=torch.float32, device=dev)
        self.pred.update({"joint": torch.empty(nj, 2, **f32), "grain": torch.empty(ng, 2, **f32),
                          "grain_area": torch.empty(ng, **f32)})
        self._tmp = torch.empty(nj, 8, **f32)
        # the regressor and the classifier are independent given (x, edge geometry): run them
        # on two HIP streams so one model's launch tails overlap the other's kernels
        if joint_launches is None:
            joint_launches = self.n_nodes["joint"] < self.JOINT_LAUNCH_MAX_JOINTS
        self.joint_launches = joint_launches
        self.concurrent = concurrent and not joint_launches
        # GGNN_TAIL=join (development, A/B runs): the plain two-stream plan -- update, centres, refresh and edge
        # records behind a join of both streams -- instead of the pipelined one
        self.pipeline_tail = os.environ.get("GGNN_TAIL", "") != "join"
        # GGNN_PIPE=r4 (development, A/B runs): the round-4 pipelined plan, whose Rmodel.update waits for the classifier's
        # decoder; default: the regressor's tail runs UNDER the classifier's decoder (_enqueue_steps_overlapped)
        self.overlap_tail = os.environ.get("GGNN_PIPE", "") != "r4"
        # where the classifier's chain of a step starts relative to the regressor's: "none" = together with it (default),
        # "enc" / "dec" = behind its encoder / decoder cell (development switch, see _enqueue_steps_overlapped)
        self.classifier_lead = os.environ.get("GGNN_C_AFTER", "none")
        self._side = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)) if self.concurrent else None
        self.refresh_centres = refresh_centres
        self.domain_factor = float(domain_factor)
        self.domain_offset = None
        if refresh_centres and self.domain_factor > 1:
            if domain_offset is None:
                raise _lib.GGNNError("domain_factor > 1 needs domain_offset")
            self.domain_offset = domain_offset.to(dev, torch.float32).contiguous()
        self.steps_done = 0
        self.use_graph = use_graph

    @torch.no_grad()
    def set_process_parameters(self, G: float, R: float):
        """The reference's `--temporal` schedule (test.py:345-346, 376-378): thermal gradient G and pulling speed R
        of the step to come, written into the junction features (`x_joint[:, 3] = 1 - G / 10`, `x_joint[:, 4] = R / 2`)
        before the forwards.  The edge records of the next step carry the sources' features, so they are rebuilt: the
        in-place write bumps the tensor's version counter, which step() / run() check (`_ensure_edge_records`).  The
        values themselves (graph_trajectory.py:129-175, GR_seq_from_time) are the caller's: call this between steps."""
        self.x["joint"][:, 3] = 1.0 - float(G) / 10.0
        self.x["joint"][:, 4] = float(R) / 2.0

    def _pack_weights(self):
        """Fused device weights of both models, and the parameter versions they were packed from."""
        self._wver = (_param_version(self.rmodel), _param_version(self.cmodel))
        self._wver_sample = (_param_version_sample(self.rmodel), _param_version_sample(self.cmodel))
        for name, m in (("R", self.rmodel), ("C", self.cmodel)):
            self.packed[name] = (m.gclstm_encoder.cell_list[0].packed(True),
                                 m.gclstm_decoder.cell_list[0].packed(False, m._live_out))
        self.w_reg = pack_regressor_heads(self.rmodel.linear)
        self.w_cls = pack_classifier_heads(self.cmodel.lin1, self.cmodel.lin2)

    def refresh_weights(self, force: bool = False, sample: bool = False):
        """Re-pack the weights and drop the captured hipGraphs (which hold the old buffers'
        addresses) if a parameter of either model was updated, moved or replaced since they were
        packed (load_state_dict, an optimizer step, .to()).  Checked on every run() / run_events() call and
        on every 16th step() / step_events() call (the check walks 568 tensors, ~0.15 ms: a seventh of an
        eventful step); the step_events() calls between look at a sample of the tensors (`sample`: every 24th --
        what changes a model changes all of it); call it directly after changing single parameters
        between two step() / step_events() calls."""
        if sample and not force:
            now = (_param_version_sample(self.rmodel), _param_version_sample(self.cmodel))
            if now == self._wver_sample:
                return
        if force or self._wver != (_param_version(self.rmodel), _param_version(self.cmodel)):
            self._pack_weights()
            self._graphs = None
            self._drop_segment_graphs()
            self._spec = None   # (run_events' captured blocks hold the old buffers' addresses too)

    def _set_topology(self, edge_index_dict, edge_attr_dict=None, lasting=True, trusted=False):
        """(Re)build everything that depends on the edge lists: CSR + unit tables, the edge-length
        buffers (refreshed in place every step), the per-edge geometry records and the per-edge
        outputs.  Called once at construction and after every topological event (`trusted`: lists the
        library's own update produced -- no range check, no read-back)."""
        dev = self.x["joint"].device
        if getattr(self, "_cap", None) is not None:
            if trusted and edge_attr_dict is None and all(edge_index_dict[et].size(1) <= self._cap["cap"][et] for et in EDGE_TYPES):
                return self._install_topology_in_place(edge_index_dict)
            self._cap = None   # (a caller's own topology, or one that grew: back to buffers of its own; the segment graphs go)
            self._drop_segment_graphs()
        self.edge_index = {et: edge_index_dict[et] for et in EDGE_TYPES}
        self.graph = graph_for(self.be, self.edge_index, self.n_nodes, trusted)
        if trusted and edge_attr_dict is None:
            return self._set_topology_buffers_from_one_allocation(dev)
        if edge_attr_dict is not None:
            self.edge_attr = {et: edge_attr_dict[et].detach().clone().contiguous().view(-1).float()
                              for et in EDGE_TYPES}
        else:  # lengths are recomputed by the refresh that follows an event (every element: nothing to initialise)
            self.edge_attr = {et: torch.empty(self.edge_index[et].size(1), device=dev) for et in EDGE_TYPES}
        # second buffer of the pipelined step (_enqueue_step_pipelined): the refresh writes the lengths of step
        # k + 1 while the classifier's head still reads those of step k
        self._ea_other = {et: torch.empty_like(self.edge_attr[et]) for et in EDGE_TYPES}
        self._einfo_fresh = False   # True: self.einfo already holds the records of the step to come
        E = self.graph.edge_index[ET_JJ].size(1)
        if not hasattr(self, "pred"):
            self.pred = {}
        self.pred["edge_event"] = torch.empty(E, dtype=torch.float32, device=dev)
        self.pred["edge"] = torch.empty(E, 2, dtype=torch.float32, device=dev)
        self.einfo = alloc_einfo(self.graph, dev)
        # second set of edge records (_enqueue_steps_overlapped): made on first use; the classifier's copies of x keep
        # their buffers (the node sets never change) but no longer mirror x
        self._einfo_other = None
        if not hasattr(self, "_xc"):
            self._xc = self._xc_other = None
        self._x_written_outside()
        self._graphs = None
        self._seen = None

    def _set_topology_buffers_from_one_allocation(self, dev):
        """The per-edge buffers of a topology the event loop installs (every step, on the reference's trajectories): both
        sets of edge lengths and edge records and the per-edge predictions as views of ONE allocation, nothing
        initialised -- the refresh that follows an event writes every length, ggnn_edge_prepare / ggnn_step_refresh_prepare
        every record (zero padding included) before anything reads them."""
        E = {et: self.edge_index[et].size(1) for et in EDGE_TYPES}
        r4 = lambda n: (n + 3) & ~3
        rec = {et: (E[et] + _lib.GGNN_UNIT_EDGES) * _lib.GGNN_EINFO_ROW for et in EDGE_TYPES}
        total = sum(2 * r4(E[et]) + 2 * r4(rec[et]) for et in EDGE_TYPES) + r4(E[ET_JJ]) + r4(2 * E[ET_JJ])
        buf, o = torch.empty(total, dtype=torch.float32, device=dev), [0]

        def take(n, shape=None):
            v = buf[o[0]:o[0] + n]
            o[0] += r4(n)
            return v if shape is None else v.view(shape)
        self.edge_attr = {et: take(E[et]) for et in EDGE_TYPES}
        self._ea_other = {et: take(E[et]) for et in EDGE_TYPES}
        shape = lambda et: (E[et] + _lib.GGNN_UNIT_EDGES, _lib.GGNN_EINFO_ROW)
        self.einfo = {et: take(rec[et], shape(et)) for et in EDGE_TYPES}
        self._einfo_other = {et: take(rec[et], shape(et)) for et in EDGE_TYPES}
        self._einfo_fresh = False
        self.pred["edge_event"] = take(E[ET_JJ])
        self.pred["edge"] = take(2 * E[ET_JJ], (E[ET_JJ], 2))
        self._x_written_outside()
        self._graphs = None
        self._seen = None

    # -- a topology that changes IN PLACE (event mode) ---------------------------------------------------------------------
    def _enter_capacity_mode(self):
        """enable_events(): the edge lists, the CSR tables and every per-edge buffer move into allocations of the CURRENT
        lists' size -- the capacity: grain eliminations only remove edges, neighbour switches keep their number -- and stay
        there: an event rewrites them in place (_install_topology_in_place), the per-edge kernels read the number of edges from
        device memory (CSR.E_dev), and the hipGraphs of step_events()' two segments keep replaying across events (SURVEY 8
        f-2: "keep the device rollout running between events"; round 5 dropped every graph with the topology and ran the
        steps around an event eagerly).  The fused cells need nothing: their grids follow the node sets, their edge windows
        come from the rebuilt row pointers, and an edge count that is too large only widens a clamp onto stale, valid entries."""
        be, dev = self.be, self.x["joint"].device
        if not hasattr(be, "csr_in_place") or os.environ.get("GGNN_EVENT_GRAPHS", "1") == "0":
            self._cap = None
            return
        cap = {et: int(self.edge_index[et].size(1)) for et in EDGE_TYPES}
        r4 = lambda n: (n + 3) & ~3
        rec = {et: (cap[et] + _lib.GGNN_UNIT_EDGES) * _lib.GGNN_EINFO_ROW for et in EDGE_TYPES}
        off, at = {}, 0
        for name, size in [(("ea", et), cap[et]) for et in EDGE_TYPES] + [(("ea2", et), cap[et]) for et in EDGE_TYPES] \
                + [(("rec", et), rec[et]) for et in EDGE_TYPES] + [(("rec2", et), rec[et]) for et in EDGE_TYPES] \
                + [("edge_event", cap[ET_JJ]), ("edge", 2 * cap[ET_JJ])]:
            off[name] = at
            at += r4(max(size, 1))
        old_ea = {et: self.edge_attr[et] for et in EDGE_TYPES}
        old_ei = {et: self.edge_index[et] for et in EDGE_TYPES}
        self._cap = {
            "cap": cap, "off": off, "buf": torch.empty(at, dtype=torch.float32, device=dev),
            "lists": {et: torch.empty(2 * max(cap[et], 1), dtype=torch.int64, device=dev) for et in EDGE_TYPES},
            "csr": be.csr_in_place([(cap[et], self.n_nodes[et[0]], self.n_nodes[et[-1]]) for et in EDGE_TYPES], dev),
            "counts": torch.zeros(len(EDGE_TYPES), dtype=torch.int64, device=dev),
            "counts_host": torch.zeros(len(EDGE_TYPES), dtype=torch.int64).pin_memory(),
        }
        self._install_topology_in_place(old_ei)
        for et in EDGE_TYPES:
            self.edge_attr[et].copy_(old_ea[et])
        self._drop_segment_graphs()

    def _install_topology_in_place(self, edge_index_dict):
        """The lists of `edge_index_dict` (device tensors, or already views of the list buffers) become the topology: copied
        into the list buffers, their sizes into the device-side counts, the CSR tables rebuilt inside the arena, the per-edge
        tensors re-cut as views of the same allocations.  No address changes: the segment graphs stay."""
        C, dev = self._cap, self.x["joint"].device
        E = {et: int(edge_index_dict[et].size(1)) for et in EDGE_TYPES}
        self.edge_index = {}
        for k, et in enumerate(EDGE_TYPES):
            dst = C["lists"][et][:2 * E[et]].view(2, E[et])
            src = edge_index_dict[et]
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
            self.edge_index[et] = dst
            C["counts_host"][k] = E[et]
        C["counts"].copy_(C["counts_host"], non_blocking=True)
        counts = {et: C["counts"][k:k + 1] for k, et in enumerate(EDGE_TYPES)}
        from .engine import GraphCSR
        self.graph = GraphCSR(self.be, self.edge_index, self.n_nodes, trusted=True, into=C["csr"], counts=counts)
        buf, off = C["buf"], C["off"]
        cut = lambda name, n, shape=None: buf[off[name]:off[name] + n] if shape is None else buf[off[name]:off[name] + n].view(shape)
        rows = lambda et: E[et] + _lib.GGNN_UNIT_EDGES
        sets = [{et: cut(("ea", et), E[et]) for et in EDGE_TYPES}, {et: cut(("ea2", et), E[et]) for et in EDGE_TYPES}]
        recs = [{et: cut(("rec", et), rows(et) * _lib.GGNN_EINFO_ROW, (rows(et), _lib.GGNN_EINFO_ROW)) for et in EDGE_TYPES},
                {et: cut(("rec2", et), rows(et) * _lib.GGNN_EINFO_ROW, (rows(et), _lib.GGNN_EINFO_ROW)) for et in EDGE_TYPES}]
        # which of the two sets is the current one follows the speculative loop's slot parity while its state lives
        # (run_events' graphs alternate the sets by slot: _spec_state); step_events() alone stays on the first
        S = getattr(self, "_spec", None)
        par = (S["cur"] & 1) if (S is not None and S.get("in_place")) else 0
        C["sets"], C["recs"] = sets, recs
        self.edge_attr, self._ea_other = sets[par], sets[1 - par]
        self.einfo, self._einfo_other = recs[par], recs[1 - par]
        self._einfo_fresh = False
        if not hasattr(self, "pred"):
            self.pred = {}
        self.pred["edge_event"] = cut("edge_event", E[ET_JJ])
        self.pred["edge"] = cut("edge", 2 * E[ET_JJ], (E[ET_JJ], 2))
        self._x_written_outside()
        self._graphs = None
        self._seen = None
        # a segment graph was captured for the sizes of ITS moment: it stays valid while nothing has grown since
        got = C.get("captured")
        if got is not None and any(E[et] > got[et] for et in EDGE_TYPES):
            self._drop_segment_graphs()
            C["captured"] = None
            if S is not None:
                S["graphs"], S["captured"] = {}, None

    def _x_written_outside(self):
        """x was (or is about to be) advanced by something other than the overlapped two-stream step / the speculative
        event step: their mirrors of x (the copies the classifier's forward reads) are stale."""
        self._xc_fresh = False
        if getattr(self, "_spec", None) is not None:
            self._spec["xs_valid"] = None

    # -- one step, enqueued on the current stream --------------------------------------
    def _pipelined(self):
        """Two streams, static topology: update, grain centres, edge refresh and the NEXT step's edge records
        run on the regressor's stream beside the classifier's last kernels (_enqueue_step_pipelined)."""
        return self._side is not None and not self.joint_launches and self.pipeline_tail

    def _enqueue_step(self):
        self._enqueue_steps(1)

    def _enqueue_steps(self, n_steps: int):
        if self._pipelined():
            if self.overlap_tail:
                return self._enqueue_steps_overlapped(n_steps)
            return self._enqueue_steps_pipelined(n_steps)
        for _ in range(n_steps):
            self._enqueue_forward_update()
            self._enqueue_refresh()
        self._einfo_fresh = False
        self._x_written_outside()

    def _enqueue_steps_pipelined(self, n_steps: int):
        """`n_steps` static-topology steps on two streams with the small launches hidden and no join between
        steps.  After both forwards only four small kernels remain -- Rmodel.update, grain centres, z clamp + edge
        lengths, and the edge records of the next forward -- during which the chip is nearly idle (~34 us of a
        ~530 us step at 10 000 grains, plus a join and a fork).  None of them needs the classifier's results: they
        run on the regressor's stream, beside the classifier's gate GEMM and heads, and the regressor's next forward
        follows them directly.  Cross-stream edges per step (each costs a few us inside a hipGraph, so there are
        exactly two):
          * regressor stream waits for the classifier's decoder sweeps of this step (event `swept`) before
            Rmodel.update: by then the classifier has read x for the last time (its decoder projection) and the edge
            records too, so x (update, centres, z clamp) and einfo (next step's records) may be overwritten;
          * classifier stream waits for the regressor stream's edge records (event `ready`) before its next forward.
        The refreshed edge lengths go to the OTHER edge_attr buffer: the classifier's head still reads this step's
        (models.py:595-609); the buffers swap every step.  Needs self.einfo to hold the records of the first step
        already (step() / run() see to that)."""
        be, x, p = self.be, self.x, self.pred
        einfo = self.einfo
        # the regressor's chain stays on the current stream (it carries every step-to-step dependency), the
        # classifier forks from it at the top of every step and is joined once, behind the last one
        main = torch.cuda.current_stream()
        st_c = self._side[1]
        events = []  # kept alive until the streams are joined (and a capture has ended)
        for _ in range(n_steps):
            ea, ea_next = self.edge_attr, self._ea_other
            swept = torch.cuda.Event()
            events.append(swept)
            st_c.wait_stream(main)   # x, einfo and edge_attr of this step are final on `main`
            enc, dec = self.packed["R"]
            hr, _ = run_encoder_decoder(be, enc, dec, self.graph, self.ws["R"], x, ea, einfo)
            with torch.cuda.stream(st_c):
                enc, dec = self.packed["C"]
                h, _ = run_encoder_decoder(be, enc, dec, self.graph, self.ws["C"], x, ea, einfo,
                                           None, lambda: swept.record(st_c))
                be.heads_classifier(h["joint"], self.graph.edge_index[ET_JJ], ea[ET_JJ], self.w_cls[0],
                                    self.w_cls[1], self._tmp, p["edge_event"], p["edge"])
            main.wait_event(swept)
            # heads + Rmodel.update in one launch; z clamp + edge lengths + next records in one launch
            be.heads_regressor_update(hr["joint"], hr["grain"], x["joint"], x["grain"], self.w_reg[0], self.w_reg[1],
                                      p["joint"], p["grain"], p["grain_area"], self.dz, self.zmax, self.flags)
            if self.refresh_centres:
                be.grain_centres(self.graph.csr[("joint", "pull", "grain")], x["joint"], x["grain"],
                                 self.domain_factor, self.domain_offset)
            be.step_refresh_prepare(x["joint"], x["grain"], self.zmax, self.flags,
                                    [(self.graph.csr[et], ea_next[et], x[et[0]], x[et[-1]], einfo[et])
                                     for et in EDGE_TYPES])
            self.edge_attr, self._ea_other = ea_next, ea
        main.wait_stream(st_c)
        self._einfo_fresh = True
        self._x_written_outside()
        return events

    def _overlap_buffers(self):
        """The second set of edge records and the classifier's two alternating copies of x (outside any capture:
        _capture calls this first)."""
        if self._pipelined() and self.overlap_tail:
            if self._einfo_other is None:
                self._einfo_other = alloc_einfo(self.graph, self.x["joint"].device)
            if self._xc is None:
                self._xc = {nt: torch.empty_like(self.x[nt]) for nt in NODE_TYPES}
                self._xc_other = {nt: torch.empty_like(self.x[nt]) for nt in NODE_TYPES}
                self._xc_fresh = False

    def _enqueue_steps_overlapped(self, n_steps: int):
        """`n_steps` static-topology steps on two streams with the regressor's tail -- heads + Rmodel.update, grain
        centres, z clamp + edge lengths + the NEXT step's edge records -- UNDER the classifier's decoder cell.  In the
        round-4 plan (_enqueue_steps_pipelined) that tail waited for the classifier's decoder, the last reader of x and of
        the edge records: at 10 000 grains the two decoder cells are 393 workgroups in two rounds of the chip, the
        regressor's finishes ~100 us before the classifier's, and the tail (three launch-bound kernels, ~25 us + the
        cross-stream hand-overs) then ran on an idle chip (profiles/r5_step_timeline.txt).  Here the classifier reads a
        private copy of x and the copies / edge lengths / edge records alternate between two sets -- the refresh of step
        k writes the set of step k + 1 (the copy of x as a by-product of its pass over the nodes: ggnn_step_refresh_prepare's
        mirror, ABI 24; round 5 copied x at the top of the classifier's step, two copy kernels at the head of its chain)
        while the classifier still reads the set of step k -- so nothing the tail writes is read by the classifier's
        forward of the same step.  Cross-stream edges per step:
          * the classifier waits for `ready` (its copy of x, edge lengths and records of this step are final; recorded by
            the regressor's stream at the top of the step);
          * the NEXT step's refresh (the first launch that writes into the set this step's classifier reads: copy of x,
            edge lengths, edge records) waits for `headed` (the classifier's heads -- the last reader, of the edge lengths
            -- are done).  Round 5 held the whole next regressor forward back until the classifier's decoder had
            finished ("the chip is free"); when the classifier's decoder is the one that ends last -- which of the two
            decoder cells gets the compute units first is the hardware's choice -- that left 60 us per step with only
            its last workgroups running (profiles/r6_step_timeline.txt): now the regressor's next encoder fills them.
        Needs self._xc to mirror x already (step() / run() see to that: _ensure_edge_records).
        Same kernels on the same operands as the single-stream plan: bit-identical results
        (test_pipelined_two_stream_rollout_equals_the_single_stream_plan)."""
        be, x, p = self.be, self.x, self.pred
        self._overlap_buffers()
        main = torch.cuda.current_stream()
        st_c = self._side[1]
        events = []        # kept alive until the streams are joined (and a capture has ended)
        headed_prev = None    # the classifier's heads of the previous step of this block
        for _ in range(n_steps):
            ea, ea_next = self.edge_attr, self._ea_other
            einfo, einfo_next = self.einfo, self._einfo_other
            xc, xc_next = self._xc, self._xc_other
            ready, headed, lead = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            events += [ready, headed, lead]
            ready.record(main)
            enc, dec = self.packed["R"]
            # GGNN_C_AFTER=enc / dec (development; default: none) start the classifier's chain BEHIND the regressor's encoder
            # / decoder cell instead of beside it (`lead`), so that the regressor's tail -- three launch-bound kernels,
            # ~40 us -- runs under the classifier's decoder cell rather than on an idle chip.  Measured a wash: one box
            # 3 288-3 491 steps/s (none) vs 3 460-3 503 (enc) vs 3 176-3 201 (dec), a second box 3 523-3 626 (none) vs
            # 3 466-3 536 (enc) -- profiles/r6_step_timeline.txt; the results are bit-identical either way.
            stage = self.classifier_lead
            hr, _ = run_encoder_decoder(be, enc, dec, self.graph, self.ws["R"], x, ea, einfo,
                                        x_read=(lambda: lead.record(main)) if stage == "dec" else None,
                                        after_encoder=(lambda: lead.record(main)) if stage == "enc" else None)
            with torch.cuda.stream(st_c):
                st_c.wait_event(ready)
                if stage in ("enc", "dec"):
                    st_c.wait_event(lead)
                enc, dec = self.packed["C"]
                h, _ = run_encoder_decoder(be, enc, dec, self.graph, self.ws["C"], xc, ea, einfo)
                be.heads_classifier(h["joint"], self.graph.edge_index[ET_JJ], ea[ET_JJ], self.w_cls[0],
                                    self.w_cls[1], self._tmp, p["edge_event"], p["edge"])
                headed.record(st_c)
            # heads + Rmodel.update in one launch; z clamp + edge lengths + next records + next copy of x in one launch
            be.heads_regressor_update(hr["joint"], hr["grain"], x["joint"], x["grain"], self.w_reg[0], self.w_reg[1],
                                      p["joint"], p["grain"], p["grain_area"], self.dz, self.zmax, self.flags)
            if self.refresh_centres:
                be.grain_centres(self.graph.csr[("joint", "pull", "grain")], x["joint"], x["grain"],
                                 self.domain_factor, self.domain_offset)
            if headed_prev is not None:
                main.wait_event(headed_prev)   # the previous step's classifier has read the set this refresh writes
            be.step_refresh_prepare(x["joint"], x["grain"], self.zmax, self.flags,
                                    [(self.graph.csr[et], ea_next[et], x[et[0]], x[et[-1]], einfo_next[et])
                                     for et in EDGE_TYPES], mirror=(xc_next["joint"], xc_next["grain"]))
            headed_prev = headed
            self.edge_attr, self._ea_other = ea_next, ea
            self.einfo, self._einfo_other = einfo_next, einfo
            self._xc, self._xc_other = xc_next, xc
        main.wait_stream(st_c)
        self._einfo_fresh = True
        self._xc_fresh = True
        if getattr(self, "_spec", None) is not None:
            self._spec["xs_valid"] = None
        return events

    def _ensure_edge_records(self, mirror=True):
        """Before a pipelined step outside a capture: einfo must hold the records of the step to come and (overlapped
        plan, `mirror`) the classifier's copy of x must equal x.  They are stale after construction, a topology change,
        an event-mode or single-stream step, and when the caller wrote into x / edge_attr (in-place Python writes bump
        the tensors' version counters; the kernels do not)."""
        seen = tuple(t._version for t in self.x.values()) + tuple(sorted(
            t._version for t in (*self.edge_attr.values(), *self._ea_other.values())))
        if seen != self._seen:
            self._x_written_outside()
        if not self._einfo_fresh or seen != self._seen:
            prepare_edges(self.be, self.graph, self.x, self.edge_attr, self.einfo)
            self._einfo_fresh = True
        self._seen = seen
        if mirror and self.overlap_tail and not self._xc_fresh:
            self._overlap_buffers()
            for nt in NODE_TYPES:
                self._xc[nt].copy_(self.x[nt])
            self._xc_fresh = True

    def _enqueue_forward_update(self, joint=None):
        """test.py:382-402: both forwards, Rmodel.update, z advance.  `joint`: True = the regressor and the classifier in the
        same launches whatever the rollout's plan (same kernels on the same operands, bit-identical results)."""
        be, x, ea, p = self.be, self.x, self.edge_attr, self.pred
        joint = self.joint_launches if joint is None else joint
        # edge geometry once per step, shared by both models and all four cells
        einfo = prepare_edges(be, self.graph, x, ea, self.einfo)


        def regressor():
            enc, dec = self.packed["R"]
            h, _ = run_encoder_decoder(be, enc, dec, self.graph, self.ws["R"], x, ea, einfo)
            be.heads_regressor(h["joint"], h["grain"], x["grain"], self.w_reg[0], self.w_reg[1],
                               p["joint"], p["grain"], p["grain_area"])

        def classifier(x_read=None):
            enc, dec = self.packed["C"]
            h, _ = run_encoder_decoder(be, enc, dec, self.graph, self.ws["C"], x, ea, einfo, x_read)
            be.heads_classifier(h["joint"], self.graph.edge_index[ET_JJ], ea[ET_JJ], self.w_cls[0],
                                self.w_cls[1], self._tmp, p["edge_event"], p["edge"], E_dev=self.graph.csr[ET_JJ].E_dev)

        if joint:
            run_encoder_decoder_multi(be, [(*self.packed["R"], self.ws["R"]), (*self.packed["C"], self.ws["C"])],
                                      self.graph, x, einfo)
            be.heads_regressor(self.ws["R"].h2["joint"], self.ws["R"].h2["grain"], x["grain"], self.w_reg[0],
                               self.w_reg[1], p["joint"], p["grain"], p["grain_area"])
            be.heads_classifier(self.ws["C"].h2["joint"], self.graph.edge_index[ET_JJ], ea[ET_JJ], self.w_cls[0],
                                self.w_cls[1], self._tmp, p["edge_event"], p["edge"], E_dev=self.graph.csr[ET_JJ].E_dev)
        elif self._side is None:
            regressor()
            classifier()
        else:
            # Two streams.  Rmodel.update (and the grain centres) need the regressor's heads only, and they may
            # overwrite x as soon as the classifier's last reader of x -- its decoder projection -- has run:
            # they go on the regressor's stream behind that event and hide beside the classifier's sweep, gate
            # GEMM and heads (the chip is otherwise nearly idle during these small launches).
            main = torch.cuda.current_stream()
            st_r, st_c = self._side
            x_free = torch.cuda.Event()
            st_r.wait_stream(main)
            st_c.wait_stream(main)
            with torch.cuda.stream(st_r):
                regressor()
            with torch.cuda.stream(st_c):
                classifier(lambda: x_free.record(st_c))
            with torch.cuda.stream(st_r):
                st_r.wait_event(x_free)
                be.step_update(x["joint"], x["grain"], p["joint"], p["grain"], self.dz, self.zmax, self.flags)
            main.wait_stream(st_r)
            main.wait_stream(st_c)
            return
        be.step_update(x["joint"], x["grain"], p["joint"], p["grain"], self.dz, self.zmax, self.flags)

    def _enqueue_refresh(self):
        """test.py:405-407, 468-478 + 556-559, 562-575: z clamp, grain centres, edge lengths."""
        be, x, ea = self.be, self.x, self.edge_attr
        if self.refresh_centres:
            be.grain_centres(self.graph.csr[("joint", "pull", "grain")], x["joint"], x["grain"],
                             self.domain_factor, self.domain_offset)
        be.step_refresh(x["joint"], x["grain"], self.zmax, self.flags,
                        [(self.graph.edge_index[et], x[et[0]], x[et[-1]], ea[et], self.graph.csr[et].E_dev) for et in EDGE_TYPES])

    def _capture(self, n_steps: int = 1):
        """Record `n_steps` steps into a hipGraph (torch.cuda.CUDAGraph is hipGraph on ROCm); the
        kernels are launched through the C ABI on the capturing stream.  A pipelined step swaps the two
        edge_attr buffers: a graph is only valid from the buffer it was captured on, so graphs are kept per
        buffer (`_graphs[(n_steps, id of the current buffer)]`), and an odd number of steps leaves the other one
        current after every replay."""
        self._overlap_buffers()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        state = (self.edge_attr, self._ea_other, self._einfo_fresh, self.einfo, self._einfo_other, self._xc,
                 self._xc_other, self._xc_fresh)
        with torch.cuda.stream(s):
            # the capture records, it does not execute: x / edge_attr are left untouched
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                keep = self._enqueue_steps(n_steps)   # (its stream events outlive the capture)
        torch.cuda.current_stream().wait_stream(s)
        del keep
        (self.edge_attr, self._ea_other, self._einfo_fresh, self.einfo, self._einfo_other, self._xc, self._xc_other,
         self._xc_fresh) = state
        return g

    def _replay(self, n_steps: int):
        """`n_steps` steps from the hipGraph captured for this step count and the current edge_attr buffer."""
        if self._graphs is None:
            self._graphs = {}
        key = (n_steps, self.edge_attr[ET_JJ].data_ptr(), self.einfo[ET_JJ].data_ptr(),
               self._xc["joint"].data_ptr() if self._xc is not None else 0)
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = self._capture(n_steps)
        g.replay()
        if self._pipelined():
            if n_steps % 2:
                self.edge_attr, self._ea_other = self._ea_other, self.edge_attr
                if self.overlap_tail:   # (the edge records and the classifier's copies of x alternate with the edge lengths)
                    self.einfo, self._einfo_other = self._einfo_other, self.einfo
                    self._xc, self._xc_other = self._xc_other, self._xc
            self._einfo_fresh = True
            if self.overlap_tail:   # (the last step's refresh left its mirror of x in the now-current copy)
                self._xc_fresh = True
                if getattr(self, "_spec", None) is not None:
                    self._spec["xs_valid"] = None
            else:
                self._x_written_outside()
        else:
            self._einfo_fresh = False
            self._x_written_outside()

    # -- event-driven mode ------------------------------------------------
    def enable_events(self, mask, area_threshold: float = 1e-4, edge_threshold: float = 0.6):
        """Switch to the full loop of test.py:382-575: after the device forwards + Rmodel.update,
        grains whose predicted area fell below `area_threshold` (Rmodel.threshold, test.py:187,
        418) are eliminated and junction edges with sigmoid(edge_event) above `edge_threshold`
        (Cmodel.threshold, :188) are switched by the host-side `topology.update_topology`; then
        grain centres and edge lengths are refreshed on the NEW topology.  `mask` = the
        reference's `data['mask']` ({'grain': [N_g, 1], 'joint': [N_j, 1]}, any integer dtype)."""
        self.mask = {k: np.array(torch.as_tensor(mask[k]).cpu().numpy(), dtype=np.int64, copy=True).reshape(-1, 1)
                     for k in ("grain", "joint")}
        dev = self.x["joint"].device
        self._live_grain = torch.from_numpy(self.mask["grain"][:, 0].astype(np.int32)).to(dev)
        self.area_threshold, self.edge_threshold = float(area_threshold), float(edge_threshold)
        # device-side trigger: slightly wider than the host's exact sigmoid(x) > threshold test,
        # so a borderline edge always reaches the host, which then decides exactly
        self._logit_trigger = float(np.log(edge_threshold / (1.0 - edge_threshold)) - 1e-4)
        self._ev_flags = torch.zeros(2, dtype=torch.int32, device=dev)
        self._ev_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._quiet_steps = 0
        self._drop_segment_graphs()
        self.grain_events, self.switched = [], []
        self._enter_capacity_mode()

    def _drop_segment_graphs(self):
        """The hipGraphs of step_events()' two segments, all variants (one per set of buffers they were captured on)."""
        self._graph_fwd = self._graph_ref = None
        self._segment_graphs = {}

    def _run_segment(self, which):
        """The two halves of a step, replayed from their own hipGraphs: on the in-place topology (_enter_capacity_mode) from the
        first step on and across events; otherwise once the topology has been quiet for two steps (an event then drops the
        graphs with the topology, and a capture is not worth it while events fire every step)."""
        fn = self._enqueue_forward_update if which == "fwd" else self._enqueue_refresh
        attr = "_graph_fwd" if which == "fwd" else "_graph_ref"
        in_place = getattr(self, "_cap", None) is not None   # (the graphs survive events: captured once, at the first step)
        if self.use_graph and (self._quiet_steps >= 2 or in_place):
            # a segment graph holds the addresses of the buffers that were current when it was captured; run_events() leaves
            # other ones current (its slots' predictions, the other set of edge lengths / records): a graph per set
            key = (which, self.edge_attr[ET_JJ].data_ptr(), self.einfo[ET_JJ].data_ptr(), self.pred["joint"].data_ptr(),
                   self.pred["edge_event"].data_ptr(), self.graph.csr[ET_JJ].rowptr.data_ptr())
            graphs = self.__dict__.setdefault("_segment_graphs", {})
            if getattr(self, attr) is None:   # (dropped from outside: `ro._graph_fwd = None`)
                graphs.pop(key, None)
            if key not in graphs:
                if in_place:   # valid for as long as no list is longer than now (_install_topology_in_place)
                    now = {et: int(self.edge_index[et].size(1)) for et in EDGE_TYPES}
                    got = self._cap.get("captured")
                    self._cap["captured"] = now if got is None else {et: min(now[et], got[et]) for et in EDGE_TYPES}
                st = torch.cuda.Stream()
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        fn()
                torch.cuda.current_stream().wait_stream(st)
                graphs[key] = g
            setattr(self, attr, graphs[key])
            getattr(self, attr).replay()
        elif which == "fwd":
            # eager launches (GGNN_EVENT_GRAPHS=0 / a caller's own topology: the steps around an event, which replaces the
            # topology the graphs were captured on) are bound by the HOST's launch rate at any graph size -- ~20 launches + stream forks of the two-stream plan take
            # 0.49 ms where the kernels take 0.35 (profiles/r6_event_step_breakdown.txt): R and C share every launch here
            self._enqueue_forward_update(joint=True)
        else:
            fn()

    def step_events(self):
        """One step with topological events.  Returns (pred, grain_events, switching_list); the
        last two are empty numpy arrays on a quiet step.  One 8-byte read-back per step is the only
        host synchronisation unless an event fires."""
        if not hasattr(self, "mask"):
            raise _lib.GGNNError("call enable_events(mask, ...) first")
        self.refresh_weights(sample=self.steps_done % 16 != 0)
        self._einfo_fresh = False   # this mode prepares its edge records at the start of every step
        self._x_written_outside()
        self._run_segment("fwd")
        p = self.pred
        self.be.detect_events(p["grain_area"], self._live_grain, self.area_threshold, p["edge_event"],
                              self.graph.edge_index[ET_JJ], self._logit_trigger, self._ev_flags)
        self._ev_host.copy_(self._ev_flags, non_blocking=True)
        # behind an eventful step the next one is eventful too, on the reference's trajectories (README.md:68-69: events at
        # nearly every step): what the rewiring reads travels to the host behind the counts, one synchronisation instead of two
        payload = self._quiet_steps == 0 and getattr(self, "_evb", None) is not None and getattr(self, "rewire_hook", None) is None
        if payload:
            self._enqueue_event_readback()
        torch.cuda.current_stream().synchronize()
        events, switches = np.zeros(0, np.int64), np.zeros((0, 2), np.int64)
        pred = self.pred
        if int(self._ev_host[0]) or int(self._ev_host[1]):
            # _apply_events may replace the topology and with it the per-edge output buffers:
            # the caller gets this step's predictions on the PRE-event edge list, as the
            # reference's loop does (test.py:383-426 keeps `pred` across Cmodel.update)
            pred = dict(self.pred)
            events, switches = self._apply_events(payload_ready=payload)
        if len(events) or len(switches):
            self._quiet_steps = 0
        else:
            self._quiet_steps += 1
        self._run_segment("ref")
        self.steps_done += 1
        self.grain_events.append(events)
        self.switched.append(switches)
        return pred, events, switches

    # -- event-driven mode without a host stall per quiet step ------------------------------------------------------
    # steps per block in run_events while the topology is quiet (a graph-to-graph boundary costs ~25 us at the 10k-grain
    # graph); after an eventful step the blocks start again at one step and double while the steps stay quiet (on the in-place
    # topology their graphs are the ones captured before the event; otherwise every event drops them with the topology)
    EVENTS_UNROLL = 8

    def _spec_state(self):
        """Buffers of the speculative event loop (run_events): a ring of 2 x EVENTS_UNROLL slots, slot = step index mod
        ring size -- x as the step found it (the copy the classifier's forward reads anyway: written by the refresh of
        the step before, ggnn_step_refresh_prepare's mirror), the step's predictions, the grain centres as they were
        before the step's refresh, its z-clamp flag (test.py:405: written by the step's Rmodel.update, read by its
        refresh), its event counts and its own fp16-range word (device + pinned host: [grains, edges, range, -]) --
        plus the two alternating sets of edge lengths / edge records (set = slot parity).  Graphs are captured per
        (first slot, number of steps).  On the in-place topology (_enter_capacity_mode) the ring and the graphs survive events
        (the views are re-cut); otherwise the per-edge half and the graphs are rebuilt after every topology change."""
        S = getattr(self, "_spec", None)
        if S is not None and S["topology"] is self.graph and S["ea"][S["cur"] & 1] is self.edge_attr \
                and S["einfo"][S["cur"] & 1] is self.einfo:
            return S
        self._overlap_buffers()
        dev = self.x["joint"].device
        D = 2 * max(1, int(self.EVENTS_UNROLL))
        per_edge = ("edge_event", "edge")   # the predictions that follow the junction edge list
        C = getattr(self, "_cap", None)
        if C is not None:
            # The topology lives in place (_enter_capacity_mode): the two sets of edge lengths / records are the SAME
            # allocations before and after an event, the slots' per-edge predictions are cut from one allocation of the
            # capacity, and the graphs of the blocks -- whose per-edge kernels read the number of edges from device memory --
            # stay valid across events for as long as no list has grown: an event re-cuts the views, nothing else.
            Ecap, E = C["cap"][ET_JJ], self.pred["edge_event"].numel()
            Ea = (Ecap + 3) & ~3
            if S is None or not S.get("in_place") or S["D"] != D or S["flat"].numel() != 3 * D * Ea:
                S = self._spec = {
                    "in_place": True, "cur": 0, "D": D, "graphs": {}, "captured": None, "xs_valid": None,
                    "flat": torch.empty(3 * D * Ea, dtype=torch.float32, device=dev),
                    "xs": [{nt: torch.empty_like(self.x[nt]) for nt in NODE_TYPES} for _ in range(D)],
                    "cen": [torch.empty(self.n_nodes["grain"], 2, device=dev) for _ in range(D)],
                    "evf": [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(D)],
                    "evh": [torch.zeros(4, dtype=torch.int32).pin_memory() for _ in range(D)],
                    "zf": [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(D)],
                    "rw": [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(D)],
                    "pred": [{k: torch.empty_like(v) for k, v in self.pred.items() if k not in per_edge} for _ in range(D)]}
            flat = S["flat"]
            for i, slot in enumerate(S["pred"]):
                slot["edge_event"] = flat[3 * i * Ea:3 * i * Ea + E]
                slot["edge"] = flat[3 * i * Ea + Ea:3 * i * Ea + Ea + 2 * E].view(E, 2)
            par = S["cur"] & 1
            S["topology"], S["ea"], S["einfo"] = self.graph, C["sets"], C["recs"]
            if self.edge_attr is not C["sets"][par]:   # (step_events() in between left the first set current: move over)
                for et in EDGE_TYPES:
                    C["sets"][par][et].copy_(self.edge_attr[et])
                self._einfo_fresh = False
            self.edge_attr, self._ea_other = C["sets"][par], C["sets"][1 - par]
            self.einfo, self._einfo_other = C["recs"][par], C["recs"][1 - par]
            sizes = {et: int(self.edge_index[et].size(1)) for et in EDGE_TYPES}
            if S["captured"] is not None and any(sizes[et] > S["captured"][et] for et in EDGE_TYPES):
                S["graphs"], S["captured"] = {}, None
            return S
        if S is not None and S["D"] == D:
            # a new topology (after an event): the per-node slots, the centre snapshots and the (pinned) count words stay --
            # the node sets never change -- only the per-edge predictions follow the new edge list
            keep = {k: S[k] for k in ("xs", "cen", "evf", "evh", "zf", "rw")}
            E = self.pred["edge_event"].numel()
            Ea = (E + 3) & ~3                                                     # (16-byte aligned segments)
            flat = torch.empty(3 * D * Ea, dtype=torch.float32, device=dev)   # one allocation for all slots
            edge_bufs = [{"edge_event": flat[3 * i * Ea:3 * i * Ea + E],
                          "edge": flat[3 * i * Ea + Ea:3 * i * Ea + Ea + 2 * E].view(E, 2)} for i in range(D)]
            pred = [{k: (edge_bufs[i][k] if k in per_edge else v) for k, v in slot.items()}
                    for i, slot in enumerate(S["pred"])]
        else:
            keep = {"xs": [{nt: torch.empty_like(self.x[nt]) for nt in NODE_TYPES} for _ in range(D)],
                    "cen": [torch.empty(self.n_nodes["grain"], 2, device=dev) for _ in range(D)],
                    "evf": [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(D)],
                    "evh": [torch.zeros(4, dtype=torch.int32).pin_memory() for _ in range(D)],
                    "zf": [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(D)],
                    "rw": [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(D)]}
            pred = [{k: torch.empty_like(v) for k, v in self.pred.items()} for _ in range(D)]
        S = self._spec = {
            "topology": self.graph, "cur": 0, "D": D,
            "ea": [self.edge_attr, self._ea_other], "einfo": [self.einfo, self._einfo_other],
            "pred": pred, "graphs": {}, "xs_valid": None, **keep}
        return S

    def _enqueue_spec_step(self, slot: int, headed_prev=None):
        """One step with events ASSUMED ABSENT: the overlapped two-stream step (_enqueue_steps_overlapped) on the edge
        set of the slot's parity + a snapshot of the grain centres before they are refreshed + the event counts of this
        step's predictions (ggnn_detect_events) copied to pinned host memory.  Nothing an event would need is
        overwritten by the steps enqueued behind it: they use other slots of the ring (the copy of x the step leaves
        for the next one, its z-clamp flag and its range word included).  Needs S["xs"][slot] to equal x (_spec_launch)."""
        S, be, x = self._spec, self.be, self.x
        s = slot & 1
        ea, ea_next, einfo, einfo_next = S["ea"][s], S["ea"][1 - s], S["einfo"][s], S["einfo"][1 - s]
        p, xc, xc_next, zf = S["pred"][slot], S["xs"][slot], S["xs"][(slot + 1) % S["D"]], S["zf"][slot]
        main, st_c = torch.cuda.current_stream(), self._side[1]
        ready, updated, headed = (torch.cuda.Event() for _ in range(3))
        st_d = self._side[0]   # the event counts go out on a stream of their own: neither model's chain waits for them
        for name in ("R", "C"):   # the fused cells of this step report to the slot's own word (a void step's report is dropped)
            self.ws[name].range_flag = S["rw"][slot]
        ready.record(main)
        lead, stage = torch.cuda.Event(), self.classifier_lead   # (the classifier behind the regressor's encoder cell: see _enqueue_steps_overlapped)
        enc, dec = self.packed["R"]
        hr, _ = run_encoder_decoder(be, enc, dec, self.graph, self.ws["R"], x, ea, einfo,
                                    x_read=(lambda: lead.record(main)) if stage == "dec" else None,
                                    after_encoder=(lambda: lead.record(main)) if stage == "enc" else None)
        with torch.cuda.stream(st_c):
            st_c.wait_event(ready)
            if stage in ("enc", "dec"):
                st_c.wait_event(lead)
            enc, dec = self.packed["C"]
            h, _ = run_encoder_decoder(be, enc, dec, self.graph, self.ws["C"], xc, ea, einfo)
            be.heads_classifier(h["joint"], self.graph.edge_index[ET_JJ], ea[ET_JJ], self.w_cls[0],
                                self.w_cls[1], self._tmp, p["edge_event"], p["edge"], E_dev=self.graph.csr[ET_JJ].E_dev)
            headed.record(st_c)
        be.heads_regressor_update(hr["joint"], hr["grain"], x["joint"], x["grain"], self.w_reg[0], self.w_reg[1],
                                  p["joint"], p["grain"], p["grain_area"], self.dz, self.zmax, zf)
        updated.record(main)
        if self.refresh_centres:   # (the snapshot of the centres the events must see rides with the launch that replaces them)
            be.grain_centres(self.graph.csr[("joint", "pull", "grain")], x["joint"], x["grain"],
                             self.domain_factor, self.domain_offset, centres_before=S["cen"][slot])
        else:
            S["cen"][slot].copy_(x["grain"][:, :2])
        if headed_prev is not None:
            main.wait_event(headed_prev)   # the previous step's classifier has read the edge set this refresh writes
        be.step_refresh_prepare(x["joint"], x["grain"], self.zmax, zf,
                                [(self.graph.csr[et], ea_next[et], x[et[0]], x[et[-1]], einfo_next[et])
                                 for et in EDGE_TYPES], mirror=(xc_next["joint"], xc_next["grain"]))
        with torch.cuda.stream(st_d):
            st_d.wait_event(updated)   # grain_area of this step (and every cell of the regressor has reported its range)
            st_d.wait_event(headed)    # ... and its edge_event (the classifier's cells have, too)
            # (the slot's range word travels in flags[2] and is cleared by the same launch: it is sticky on the device)
            be.detect_events(p["grain_area"], self._live_grain, self.area_threshold, p["edge_event"],
                             self.graph.edge_index[ET_JJ], self._logit_trigger, S["evf"][slot], S["rw"][slot],
                             E_dev=self.graph.csr[ET_JJ].E_dev)
            S["evh"][slot].copy_(S["evf"][slot], non_blocking=True)
        return [ready, lead, updated, headed]

    def _enqueue_spec_steps(self, slots):
        """The steps of a block back to back (a step's refresh waits for the previous step's classifier heads -- the last
        reader of the edge set it writes --, nothing else of a step waits for the step before on the other stream), the
        streams joined behind the last one."""
        keep, headed = [], None
        try:
            for sl in slots:
                ev = self._enqueue_spec_step(sl, headed)
                headed = ev[-1]
                keep += ev
        finally:
            for name in ("R", "C"):
                self.ws[name].range_flag = self._range_word
        torch.cuda.current_stream().wait_stream(self._side[1])
        torch.cuda.current_stream().wait_stream(self._side[0])
        return keep

    def _spec_launch(self, n: int):
        """Enqueue `n` speculative steps from the current slot on; returns (their slots, the event behind them)."""
        S = self._spec
        slots = [(S["cur"] + i) % S["D"] for i in range(n)]
        if S["xs_valid"] != slots[0]:   # the first step's copy of x (later ones get theirs from the refresh before them)
            for nt in NODE_TYPES:
                S["xs"][slots[0]][nt].copy_(self.x[nt])
        S["xs_valid"] = (slots[-1] + 1) % S["D"]
        self._xc_fresh = False
        if self.use_graph and n > 1:
            g = S["graphs"].get((slots[0], n))
            if g is None:
                if S.get("in_place"):   # valid for as long as no list is longer than now (_spec_state)
                    now = {et: int(self.edge_index[et].size(1)) for et in EDGE_TYPES}
                    got = S["captured"]
                    S["captured"] = now if got is None else {et: min(now[et], got[et]) for et in EDGE_TYPES}
                st = torch.cuda.Stream()
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        keep = self._enqueue_spec_steps(slots)
                torch.cuda.current_stream().wait_stream(st)
                del keep
                S["graphs"][(slots[0], n)] = g
            g.replay()
        else:
            self._enqueue_spec_steps(slots)
        done = torch.cuda.Event()
        done.record()
        self._spec_adopt(slots[-1])
        self._einfo_fresh = True   # (the last step's refresh prepared the records of the step to come)
        return slots, done

    def _spec_adopt(self, slot: int):
        """The rollout's current buffers := the state behind the step of `slot`."""
        S = self._spec
        nxt = (slot + 1) & 1
        self.pred = S["pred"][slot]
        self.edge_attr, self._ea_other = S["ea"][nxt], S["ea"][1 - nxt]
        self.einfo, self._einfo_other = S["einfo"][nxt], S["einfo"][1 - nxt]
        S["cur"] = (slot + 1) % S["D"]

    def run_events(self, n_steps: int):
        """`n_steps` steps of the full loop of test.py:382-575 (forwards, Rmodel.update, topological events, grain
        centres, edge refresh) WITHOUT a host stall per quiet step: the steps are enqueued EVENTS_UNROLL at a time as if
        they had no events (one captured graph per block), each step's event counts travel to pinned host memory, and
        the host reads the counts of a block only after the next block is enqueued -- the device never waits for the
        host while the topology is quiet.  When a step did have events, every step enqueued behind it is void: x is
        taken back from the copy the following step made at its top, the grain centres from the snapshot taken before
        the eventful step's refresh, that step's predictions are still in their own slot; the events are then applied
        exactly as step_events does (host-side topology update, refresh on the new topology) and the loop goes on from
        the next step.  Same results, bit for bit, as n_steps x step_events()
        (test_speculative_event_loop_equals_step_events).  Needs the two-stream launch plan (joint_launches=False,
        concurrent=True); otherwise runs step_events in a loop.  Returns (grain_events, switching_lists): one entry per step."""
        if not hasattr(self, "mask"):
            raise _lib.GGNNError("call enable_events(mask, ...) first")
        if not (self._pipelined() and self.overlap_tail):
            out = [self.step_events()[1:] for _ in range(n_steps)]
            return [e for e, _ in out], [sw for _, sw in out]
        self.refresh_weights()
        ev_out, sw_out = [], []
        none = (np.zeros(0, np.int64), np.zeros((0, 2), np.int64))

        def finish(step_events):   # book-keeping of one completed step
            ev_out.append(step_events[0])
            sw_out.append(step_events[1])
            self.grain_events.append(step_events[0])
            self.switched.append(step_events[1])
            self.steps_done += 1

        K = max(1, int(self.EVENTS_UNROLL))
        quiet_blocks = getattr(self, "_spec_quiet_blocks", 0)
        blocks = []   # enqueued, unchecked: (slots, done event), oldest first
        while len(ev_out) < n_steps:
            in_flight = sum(len(b[0]) for b in blocks)
            if len(blocks) < 2 and len(ev_out) + in_flight < n_steps:
                self._spec_state()
                self._ensure_edge_records(mirror=False)
                size = min(K, 1 << min(quiet_blocks + len(blocks), 16))
                size = min(size, K - self._spec["cur"] % K)   # blocks end on multiples of K: full blocks reuse two graphs
                blocks.append(self._spec_launch(min(size, n_steps - len(ev_out) - in_flight)))
                if len(blocks) < 2 and len(ev_out) + in_flight + len(blocks[-1][0]) < n_steps:
                    continue   # keep one block queued behind the one whose counts are read
            S = self._spec
            slots, done = blocks.pop(0)
            done.synchronize()
            hit = next((i for i, sl in enumerate(slots) if int(S["evh"][sl][0]) or int(S["evh"][sl][1])), None)
            # the fp16-range reports of the steps that stand (a void step ran on a topology the trajectory never had)
            if any(int(S["evh"][sl][2]) for sl in (slots if hit is None else slots[:hit + 1])):
                self._range_hit = True
            if hit is None:
                for _ in slots:
                    finish(none)
                quiet_blocks += 1
                self._spec_quiet_blocks = quiet_blocks
                continue
            quiet_blocks = self._spec_quiet_blocks = 0
            for _ in range(hit):
                finish(none)
            slot = slots[hit]
            # the step of `slot` saw events: whatever was enqueued behind it is void
            torch.cuda.current_stream().synchronize()
            void = len(slots) - hit - 1 + sum(len(b[0]) for b in blocks)
            blocks = []
            self._x_written_outside()
            # the z-clamp flag as the eventful step's Rmodel.update left it (test.py:405): the refreshes below read it
            self.flags.copy_(S["zf"][slot])
            if void:
                nxt = S["xs"][(slot + 1) % S["D"]]
                for nt in NODE_TYPES:
                    self.x[nt].copy_(nxt[nt])                # x as the step behind found it (= after the eventful step)
                self._spec_adopt(slot)
                # the void steps refreshed the edge sets past this point: lengths and records are recomputed from x below
                self._einfo_fresh = False
            keep_centres = self.x["grain"][:, :2].clone()
            self.x["grain"][:, :2].copy_(S["cen"][slot])     # the centres the events must see: before the refresh
            def stands():   # the step stands as enqueued (its refresh ran on the unchanged topology)
                self.x["grain"][:, :2].copy_(keep_centres)
                if void:   # ... but its edge lengths were overwritten by the void steps: the same kernel on the same x
                    self.be.step_refresh(self.x["joint"], self.x["grain"], self.zmax, self.flags,
                                         [(self.graph.edge_index[et], self.x[et[0]], self.x[et[-1]], self.edge_attr[et])
                                          for et in EDGE_TYPES])
            try:
                events, switches = self._apply_events()
            except Exception:
                # the host-side update refused the events (it leaves masks, coordinates and edge lists untouched): the step
                # is kept as a quiet one, so that a caller who catches the error can go on from a consistent state
                stands()
                finish(none)
                raise
            if len(events) or len(switches):
                self._quiet_steps = 0
                self._run_segment("ref")                     # centres + edge lengths on the new topology
                self._einfo_fresh = False
            else:   # the device-side trigger was conservative
                stands()
            finish((events, switches))
        return ev_out, sw_out

    def _event_buffers(self):
        """Pinned host staging of the event round trip, sized once (the lists only shrink: a removed grain appends two
        junction columns and drops at least eight; a list that grows past its room gets new buffers)."""
        E, n_pq = self.edge_index[ET_JJ].size(1), self.edge_index[("joint", "pull", "grain")].size(1)
        B = getattr(self, "_evb", None)
        if B is not None and B["cap"] >= 2 * (E + n_pq) and B["prob_cap"] >= E:
            return B
        nj, ng = self.n_nodes["joint"], self.n_nodes["grain"]
        fj = self.x["joint"].size(1)
        cap = 2 * (E + n_pq) + 1024
        f = torch.empty(ng + (E + 512) + nj * fj + 2 * nj + 2 * ng, dtype=torch.float32).pin_memory()
        o = [0]

        def take(n, shape):
            v = f[o[0]:o[0] + n].view(shape)
            o[0] += n
            return v
        B = self._evb = {"cap": cap, "prob_cap": E + 512, "area": take(ng, (ng,)), "prob": take(E + 512, (E + 512,)),
                         "xj": take(nj * fj, (nj, fj)), "yj": take(2 * nj, (nj, 2)), "yg": take(2 * ng, (ng, 2)),
                         "lists": torch.empty(cap, dtype=torch.int64).pin_memory(),
                         "live": torch.empty(ng, dtype=torch.int32).pin_memory()}
        B["np"] = {k: B[k].numpy() for k in ("area", "prob", "xj", "yj", "yg", "lists", "live")}
        return B

    def _topology_session(self):
        """The library-side lists of this trajectory (topology.TopologySession): opened from the device lists at the first
        event (one read-back), afterwards patched by every update -- as long as the rollout's edge lists are the ones the
        session produced last."""
        from .topology import TopologySession
        jj, jg = self.edge_index[ET_JJ], self.edge_index[("joint", "pull", "grain")]
        T = getattr(self, "_topo", None)
        if T is not None and T[1] is jj and T[2] is jg and T[3] == (jj._version, jg._version):
            return T[0]
        if T is not None:
            T[0].close()
        ses = TopologySession(jj.cpu().numpy(), jg.cpu().numpy(), self.n_nodes["joint"], self.n_nodes["grain"])
        self._topo = (ses, jj, jg, (jj._version, jg._version))
        return ses

    def _enqueue_event_readback(self):
        """What the rewiring reads -- predicted areas, switching probabilities, junction coordinates and displacements -- as
        asynchronous copies into the pinned staging buffers (the caller synchronises)."""
        B, p = self._event_buffers(), self.pred
        E = self.edge_index[ET_JJ].size(1)
        prob_d = torch.sigmoid(p["edge_event"])
        B["area"].copy_(p["grain_area"], non_blocking=True)
        B["prob"][:E].copy_(prob_d, non_blocking=True)
        B["xj"].copy_(self.x["joint"], non_blocking=True)
        B["yj"].copy_(p["joint"], non_blocking=True)
        B["yg"].copy_(p["grain"], non_blocking=True)

    def _apply_events(self, payload_ready=False):
        """Host round trip of an eventful step: the predictions and junction coordinates travel to pinned host memory in
        one batch of asynchronous copies behind ONE synchronisation, the library's session rewires its lists in place
        (ggnn_topology_apply: a refused update leaves everything as it was), the new lists, coordinates and masks travel
        back asynchronously and the CSR tables are rebuilt without a read-back -- the host returns to enqueueing the next
        step while the device is still uploading (round 5: six synchronous read-backs, three pageable uploads, the lists
        copied five times on the host and every lookup table rebuilt per call; profiles/r6_event_step_breakdown.txt)."""
        from .topology import TopologyError  # noqa: F401  (raised by the session; callers catch it from here)
        import time
        p, dev = self.pred, self.x["joint"].device
        JG, GJ = ("joint", "pull", "grain"), ("grain", "push", "joint")
        hook = getattr(self, "rewire_hook", None)   # test infrastructure: see below
        T = getattr(self, "event_timing", None)      # a dict: host-side seconds of the pieces (tests/bench_event_step.py)
        t0 = time.perf_counter()
        ses = self._topology_session() if hook is None else None
        B = self._event_buffers()
        N = B["np"]
        E = self.edge_index[ET_JJ].size(1)
        if ses is not None and ses.n_pp != E:
            raise _lib.GGNNError("the topology session and the rollout's junction edge list disagree")
        if not payload_ready:
            self._enqueue_event_readback()
            torch.cuda.current_stream().synchronize()
        t1 = time.perf_counter()
        area, prob = N["area"], N["prob"][:E]
        live = self.mask["grain"][:, 0] > 0
        ge = np.flatnonzero(live & (area < np.float32(self.area_threshold)))
        ge = ge[np.argsort(area[ge], kind="stable")]                         # test.py:418-420
        if getattr(self, "max_grain_events", None) is not None:   # probe hook (tests/bench_event_step.py): random weights
            ge = ge[:int(self.max_grain_events)]                  # tie the predicted areas of hundreds of grains
        # (the session ignores edges at or below the threshold and the (dst, src) twin of every pair: with no grain below
        # the area threshold either, the update is the identity -- the device-side trigger was conservative)
        mg, mj = self.mask["grain"], self.mask["joint"]
        if not (mg.dtype == np.int64 and mg.flags.c_contiguous and mj.dtype == np.int64 and mj.flags.c_contiguous):
            mg, mj = self.mask["grain"], self.mask["joint"] = np.ascontiguousarray(mg, np.int64), np.ascontiguousarray(mj, np.int64)
        lists = N["lists"]
        if hook is None:
            events, switches = ses.apply(N["xj"], N["yj"], N["yg"][:, 0], prob, ge, mg, mj, self.edge_threshold)
            if len(events) == 0 and len(switches) == 0:
                return np.zeros(0, np.int64), np.zeros((0, 2), np.int64)
            n_pp, n_pq = ses.n_pp, ses.n_pq
            ses.export(lists[:2 * n_pp].reshape(2, n_pp), lists[2 * n_pp:2 * (n_pp + n_pq)].reshape(2, n_pq))
        else:
            # `rewire_hook`: another implementation of the update with topology.update_topology's signature (the tests'
            # scan oracle, tests/fuzz_events.py) on copies that are committed together, like the reference's call
            if len(ge) == 0 and not np.any(prob > np.float32(self.edge_threshold)):
                return np.zeros(0, np.int64), np.zeros((0, 2), np.int64)
            xj, yj, mg2, mj2 = N["xj"].copy(), N["yj"].copy(), mg.copy(), mj.copy()
            pp, pq, _, switches, events = hook(xj, self.edge_index[ET_JJ].cpu().numpy(), self.edge_index[JG].cpu().numpy(),
                                               yj, N["yg"], prob.copy(), ge, mg2, mj2, self.edge_threshold)
            if len(events) == 0 and len(switches) == 0:
                return np.zeros(0, np.int64), np.zeros((0, 2), np.int64)
            N["xj"][...], N["yj"][...], mg[...], mj[...] = xj, yj, mg2, mj2
            n_pp, n_pq = pp.shape[1], pq.shape[1]
            lists[:2 * n_pp] = pp.reshape(-1)
            lists[2 * n_pp:2 * (n_pp + n_pq)] = pq.reshape(-1)
        t2 = time.perf_counter()
        self.x["joint"].copy_(B["xj"], non_blocking=True)
        p["joint"].copy_(B["yj"], non_blocking=True)
        if len(events):
            N["live"][:] = mg[:, 0]
            self._live_grain.copy_(B["live"], non_blocking=True)
        C = getattr(self, "_cap", None)
        if C is not None and n_pp <= C["cap"][ET_JJ] and n_pq <= C["cap"][JG] and n_pq <= C["cap"][GJ]:
            # the lists go straight into the buffers the topology lives in (_install_topology_in_place finds them there)
            jj, jg = C["lists"][ET_JJ][:2 * n_pp].view(2, n_pp), C["lists"][JG][:2 * n_pq].view(2, n_pq)
            jj.copy_(B["lists"][:2 * n_pp].view(2, n_pp), non_blocking=True)
            jg.copy_(B["lists"][2 * n_pp:2 * (n_pp + n_pq)].view(2, n_pq), non_blocking=True)
            gj = C["lists"][GJ][:2 * n_pq].view(2, n_pq)
            torch.stack((jg[1], jg[0]), out=gj)
            new_ei = {ET_JJ: jj, JG: jg, GJ: gj}
        else:
            d = torch.empty(2 * (n_pp + n_pq), dtype=torch.int64, device=dev)
            d.copy_(B["lists"][:2 * (n_pp + n_pq)], non_blocking=True)
            jj, jg = d[:2 * n_pp].view(2, n_pp), d[2 * n_pp:].view(2, n_pq)
            new_ei = {ET_JJ: jj, JG: jg, GJ: torch.stack((jg[1], jg[0]))}
        t3 = time.perf_counter()
        self._set_topology(new_ei, lasting=False, trusted=True)
        if ses is not None:
            jj, jg = self.edge_index[ET_JJ], self.edge_index[JG]
            self._topo = (ses, jj, jg, (jj._version, jg._version))
        if getattr(self, "_cap", None) is None:   # (in place: the segment graphs stay)
            self._drop_segment_graphs()
        if T is not None:
            T.update(readback_s=t1 - t0, rewiring_s=t2 - t1, upload_enqueue_s=t3 - t2, set_topology_s=time.perf_counter() - t3)
        return events, switches

    def step(self):
        """Advance one rollout step; returns the prediction dict (tensors are reused)."""
        if self.steps_done % 16 == 0:
            self.refresh_weights()
        if self._pipelined():
            self._ensure_edge_records()
        if self.use_graph:
            self._replay(1)
        else:
            self._enqueue_step()
        self.steps_done += 1
        return self.pred

    # steps per graph in run(): a graph-to-graph boundary costs ~10 us on the GPU (cfg3, 500 steps: 4 steps per graph 2 867
    # steps/s, 10 steps per graph 2 897); bench.py measures with this default
    RUN_UNROLL = 10

    def run(self, n_steps: int):
        """`n_steps` static-topology steps.  With hipGraph replay the bulk goes through a graph of
        RUN_UNROLL consecutive steps (same kernels, same order, same results as step() x n)."""
        self.refresh_weights()
        if self.use_graph and n_steps >= self.RUN_UNROLL:
            if self._pipelined():
                self._ensure_edge_records()
            for _ in range(n_steps // self.RUN_UNROLL):
                self._replay(self.RUN_UNROLL)
            self.steps_done += n_steps - n_steps % self.RUN_UNROLL
            n_steps %= self.RUN_UNROLL
        for _ in range(n_steps):
            self.step()
        return self.pred

    def range_exceeded(self, clear=True) -> bool:
        """True when a fused cell has clamped an activation to fp16's range since the last check (include/ggnn.h,
        OPERAND RANGE): the trajectory since then is NOT the reference's -- re-run with GGNN_DEC=split GGNN_ENC=split.
        state() checks it (it synchronises anyway); one 4-byte read-back."""
        hit = getattr(self, "_range_hit", False)   # (reports of run_events' committed steps, read with their event counts)
        if clear:
            self._range_hit = False
        return self.be.range_exceeded(self.x["joint"].device, clear, flag=self._range_word) or hit

    def state(self):
        """Final state a caller gathers across ranks: joint xy and grain (area, extraV).  Raises when the fused
        cells reported an operand beyond their arithmetic's range on the way here."""
        if self.range_exceeded():
            raise _lib.GGNNError("an activation reached fp16's range (+-65504) in a fused cell: this trajectory was computed "
                            "with clamped operands; re-run with GGNN_DEC=split GGNN_ENC=split (full fp32 range)")
        return {"joint_xy": self.x["joint"][:, :2].clone(), "grain_area_v": self.x["grain"][:, 3:5].clone()}

    def edge_attr_dict(self):
        return {et: self.edge_attr[et].view(-1, 1) for et in EDGE_TYPES}
