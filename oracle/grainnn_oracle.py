"""CPU oracle for the GrainGNN rollout hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm in the
*reference formulation* (per-edge linears, COO gather / scatter-add, four separate gate
convolutions per cell).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it, and only as the checker / the timed CPU baseline.  The
product (`graingraphnn_amd/`) never imports it and has no CPU fallback.

Pinning: checked op-for-op against the unmodified reference (`/root/reference/models.py`
imported in the build container through `tools/oracle_stub/`) by
`tests/golden/make_golden.py`; the resulting vectors live in `tests/golden/*.npz` and are
re-checked by `tests/test_oracle_golden.py` on every run.  The trained checkpoints are not
available (stripped blobs), so the README end-to-end accuracy numbers are *unpinned*; the
pins are seeded-weight goldens + the parameter counts 1 204 612 / 1 204 806 from
`model/regressor0_logfile:40`, `model/classifier1_logfile:40`.

Third-party semantics restated here (absent from /root/reference, pinned in README.md:24-25
as torch-geometric==2.1.0 / torch-scatter==2.1.0): `HeteroConv(aggr='sum')`,
`MessagePassing.propagate` (COO, source->target, aggr='add'), `utils.softmax`
(`exp(x - segmax) / (segsum + 1e-16)`), `dense.linear.Linear`, `inits.glorot`.

All `file:line` citations are into /root/reference.
"""
import copy
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

EDGE_TYPES = (("grain", "push", "joint"), ("joint", "pull", "grain"), ("joint", "connect", "joint"))
SCALING = {"grain": 20, "joint": 5}  # models.py:398


# ----------------------------------------------------------------------------------------
# PyG pieces (third-party, restated)
# ----------------------------------------------------------------------------------------
def segment_softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax (2.1.0): exp(src - max_seg) / (sum_seg + 1e-16)."""
    seg_max = torch.full((num_nodes,), float("-inf"), dtype=src.dtype)
    seg_max = seg_max.scatter_reduce(0, index, src, reduce="amax", include_self=True)
    out = (src - seg_max[index]).exp()
    seg_sum = torch.zeros(num_nodes, dtype=src.dtype).scatter_add_(0, index, out)
    return out / (seg_sum[index] + 1e-16)


class _Linear(nn.Module):
    """PyG `Linear`: weight [out, in]; U(+-1/sqrt(in)) init for weight and bias."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        bound = 1.0 / math.sqrt(in_channels)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels).uniform_(-bound, bound))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


# ----------------------------------------------------------------------------------------
# a1-a3: PeriodConv (periodGATconv.py:157-236), heads=1, concat, no beta, dropout 0
# ----------------------------------------------------------------------------------------
class PeriodConv(nn.Module):
    def __init__(self, in_src, in_dst, out_channels):
        super().__init__()
        C = out_channels
        self.out_channels = C
        self.lin_key = _Linear(in_src, C)            # periodGATconv.py:117
        self.lin_query = _Linear(in_dst, C)          # :118
        self.lin_value = _Linear(in_src, C)          # :119
        self.lin_l2 = _Linear(C, C)                  # :120
        self.lin_edge = _Linear(1, C, bias=False)    # :105 (edge_dim forced to 1), :123
        self.lin_skip = _Linear(in_dst, C)           # :128

    def forward(self, x_src, x_dst, edge_index, edge_attr):
        C = self.out_channels
        src, dst = edge_index[0], edge_index[1]
        x_j = x_src.index_select(0, src)             # PyG propagate: x_j = x[0][ei[0]]
        x_i = x_dst.index_select(0, dst)             #                x_i = x[1][ei[1]]
        # message(), periodGATconv.py:209-211: min-image wrap of the first three columns
        rel_loc = x_j[:, :3] - x_i[:, :3]
        reloc = -1 * (rel_loc > 0.5) + 1 * (rel_loc < -0.5) + rel_loc
        x_j = torch.cat([reloc, x_j[:, 3:]], dim=1)
        query = self.lin_query(x_i)                                  # :216
        key = self.lin_key(x_j)                                      # :217
        value = self.lin_l2(F.relu(self.lin_value(x_j)))             # :218
        e = self.lin_edge(edge_attr)                                 # :222
        key = key + e                                                # :224
        alpha = (query * key).sum(dim=-1) / math.sqrt(C)             # :226
        alpha = segment_softmax(alpha, dst, x_dst.size(0))           # :227
        out = (value + e) * alpha.view(-1, 1)                        # :231-235
        agg = torch.zeros(x_dst.size(0), C, dtype=out.dtype).index_add_(0, dst, out)  # aggr='add'
        return agg + self.lin_skip(x_dst)                            # :186,192


# ----------------------------------------------------------------------------------------
# a4: HeteroConv(aggr='sum') as used at heteropgclstm.py:49-82
# ----------------------------------------------------------------------------------------
class HeteroConv(nn.Module):
    def __init__(self, in_channels, out_channels, edge_types):
        super().__init__()
        self.edge_types = [tuple(et) for et in edge_types]
        self.convs = nn.ModuleDict(OrderedDict(
            ("__".join(et), PeriodConv(in_channels[et[0]], in_channels[et[-1]], out_channels))
            for et in self.edge_types))

    def forward(self, x_dict, edge_index_dict, edge_attr_dict):
        outs = {}
        for et, ei in edge_index_dict.items():
            et = tuple(et)
            key = "__".join(et)
            if key not in self.convs:
                continue
            o = self.convs[key](x_dict[et[0]], x_dict[et[-1]], ei, edge_attr_dict[et])
            outs.setdefault(et[-1], []).append(o)
        return {k: torch.stack(v, 0).sum(0) for k, v in outs.items()}


# ----------------------------------------------------------------------------------------
# a5: HeteroPGCLSTM (heteropgclstm.py:18-183)
# ----------------------------------------------------------------------------------------
class HeteroPGCLSTM(nn.Module):
    def __init__(self, in_channels_dict, out_channels, edge_types):
        super().__init__()
        self.out_channels = out_channels
        cat_dims = {k: v + out_channels for k, v in in_channels_dict.items()}  # cat[x, h]
        for g in "ifco":   # creation order i, f, c, o (heteropgclstm.py:84-88)
            setattr(self, "conv_" + g, HeteroConv(cat_dims, out_channels, edge_types))
            b = nn.ParameterDict({nt: nn.Parameter(torch.empty(1, out_channels))
                                  for nt in in_channels_dict})
            for p in b.values():  # glorot, heteropgclstm.py:90-99
                bound = math.sqrt(6.0 / (p.size(-2) + p.size(-1)))
                p.data.uniform_(-bound, bound)
            setattr(self, "b_" + g, b)

    def _gate(self, g, xh, ei, ea):
        conv = getattr(self, "conv_" + g)(xh, ei, ea)
        b = getattr(self, "b_" + g)
        return {nt: conv[nt] + b[nt] for nt in xh}

    def forward(self, x_dict, edge_index_dict, edge_attr, h_dict=None, c_dict=None):
        C = self.out_channels
        if h_dict is None:
            h_dict = {nt: torch.zeros(x.shape[0], C) for nt, x in x_dict.items()}
        if c_dict is None:
            c_dict = {nt: torch.zeros(x.shape[0], C) for nt, x in x_dict.items()}
        xh = {nt: torch.cat([x, h_dict[nt]], dim=1) for nt, x in x_dict.items()}  # :112 (x4)
        i = {nt: torch.sigmoid(v) for nt, v in self._gate("i", xh, edge_index_dict, edge_attr).items()}
        f = {nt: torch.sigmoid(v) for nt, v in self._gate("f", xh, edge_index_dict, edge_attr).items()}
        t = {nt: torch.tanh(v) for nt, v in self._gate("c", xh, edge_index_dict, edge_attr).items()}
        c_new = {nt: f[nt] * c + i[nt] * t[nt] for nt, c in c_dict.items()}       # :132
        o = {nt: torch.sigmoid(v) for nt, v in self._gate("o", xh, edge_index_dict, edge_attr).items()}
        h_new = {nt: o[nt] * torch.tanh(c) for nt, c in c_new.items()}            # :146
        return h_new, c_new


# ----------------------------------------------------------------------------------------
# a6: SeqGCLSTM (models.py:151-301), layers == 1, seq_len == 1
# ----------------------------------------------------------------------------------------
class SeqGCLSTM(nn.Module):
    def __init__(self, in_channels_dict, out_channels, num_layers, edge_types):
        super().__init__()
        if num_layers != 1:
            raise ValueError("only layers == 1 is on the shipped path (parameters.py:49)")
        self.cell_list = nn.ModuleList([HeteroPGCLSTM(in_channels_dict, out_channels, edge_types)])

    def forward(self, x_dict, edge_index_dict, edge_attr, hidden_state):
        h, c = (None, None) if hidden_state is None else hidden_state[0]
        h, c = self.cell_list[0](x_dict, edge_index_dict, edge_attr, h, c)
        return [[h, c]]


def _hyper_fields(hyper):
    in_ch = {nt: len(f) for nt, f in hyper.features.items()}
    return in_ch, hyper.layer_size, hyper.layers, [tuple(et) for et in hyper.metadata[1]]


# ----------------------------------------------------------------------------------------
# a7: GrainNN_regressor.forward / update (models.py:351-516, periodic branch)
# ----------------------------------------------------------------------------------------
class GrainNN_regressor(nn.Module):
    def __init__(self, hyper, history=False, edge_len=False):
        super().__init__()
        if history or edge_len:
            raise NotImplementedError("history / edge_len are disabled in every shipped config")
        in_ch, C, L, ets = _hyper_fields(hyper)
        self.gclstm_encoder = SeqGCLSTM(in_ch, C, L, ets)
        self.gclstm_decoder = SeqGCLSTM(in_ch, C, L, ets)
        self.linear = nn.ModuleDict({nt: nn.Linear(C, len(t)) for nt, t in hyper.targets.items()})
        self.scaling = dict(SCALING)

    def forward(self, x_dict, edge_index_dict, edge_attr):
        hidden = self.gclstm_encoder(x_dict, edge_index_dict, edge_attr, None)     # models.py:422
        hidden = self.gclstm_decoder(x_dict, edge_index_dict, edge_attr, hidden)   # :424
        h_dict, _ = hidden[-1]
        y = {nt: self.linear[nt](h) for nt, h in h_dict.items()}                   # :433
        y["joint"] = torch.tanh(y["joint"])                                        # :443
        y["grain_area"] = torch.tanh(y["grain"][:, 0]) / self.scaling["grain"] + x_dict["grain"][:, 3]
        y["grain"][:, 0] = torch.tanh(y["grain"][:, 0])                            # :450
        y["grain"][:, 1] = F.relu(y["grain"][:, 1])                                # :452
        return y

    def update(self, x_dict, y_dict, geometry_scaling=None):
        """Periodic / no-melt-pool branch of models.py:473-516 (in place on x_dict)."""
        dj, dg, dv = y_dict["joint"], y_dict["grain"][:, 0], y_dict["grain"][:, 1]
        if geometry_scaling is not None:
            geometry_scaling["active_grains"] = (dg > -10).nonzero().view(-1)
            geometry_scaling["active_joints"] = (dj[:, 0] > -10).nonzero().view(-1)
        x_dict["joint"][:, :2] += dj / self.scaling["joint"]
        x_dict["grain"][:, 3] += dg / self.scaling["grain"]
        x_dict["grain"][:, 4] = dv
        x_dict["joint"][:, 6:8] = dj
        x_dict["grain"][:, -1] = dg


# ----------------------------------------------------------------------------------------
# a8: GrainNN_classifier.forward (models.py:529-611)
# ----------------------------------------------------------------------------------------
class GrainNN_classifier(nn.Module):
    def __init__(self, hyper, regressor=None, history=False):
        super().__init__()
        if history:
            raise NotImplementedError("history is disabled in every shipped config")
        in_ch, C, L, ets = _hyper_fields(hyper)
        if regressor is not None:
            self.gclstm_encoder = copy.deepcopy(regressor.gclstm_encoder)   # models.py:551
            self.gclstm_decoder = copy.deepcopy(regressor.gclstm_decoder)   # :552
        else:
            self.gclstm_encoder = SeqGCLSTM(in_ch, C, L, ets)
            self.gclstm_decoder = SeqGCLSTM(in_ch, C, L, ets)
        self.lin1 = nn.Linear(2 * C + 1, 2)   # :568
        self.lin2 = nn.Linear(2 * C + 1, 1)   # :569

    def forward(self, x_dict, edge_index_dict, edge_attr):
        hidden = self.gclstm_encoder(x_dict, edge_index_dict, edge_attr, None)
        hidden = self.gclstm_decoder(x_dict, edge_index_dict, edge_attr, hidden)
        h_dict, _ = hidden[-1]
        et = ("joint", "connect", "joint")
        hj = h_dict["joint"]
        src, dst = edge_index_dict[et][0], edge_index_dict[et][1]
        pair = torch.cat([hj[src], hj[dst], edge_attr[et]], dim=-1)                 # :600
        return {"edge_event": self.lin2(pair).view(-1), "edge": torch.tanh(self.lin1(pair))}


# ----------------------------------------------------------------------------------------
# a9: rollout-step glue (test.py:363-407, 562-575), static topology
# ----------------------------------------------------------------------------------------
TRAIN_FRAMES = 120  # test.py:190


def advance_z(x_dict, span):
    """test.py:401-407: z += span/121 on both node types; clamp all to 120/121 once grain 0 passes it."""
    x_dict["grain"][:, 2] += span / (TRAIN_FRAMES + 1)
    x_dict["joint"][:, 2] += span / (TRAIN_FRAMES + 1)
    if x_dict["grain"][0, 2] > TRAIN_FRAMES / (TRAIN_FRAMES + 1):
        x_dict["grain"][:, 2] = TRAIN_FRAMES / (TRAIN_FRAMES + 1)
        x_dict["joint"][:, 2] = TRAIN_FRAMES / (TRAIN_FRAMES + 1)


def refresh_edge_attr(x_dict, edge_index_dict):
    """test.py:562-575: edge_attr = || min-image(src_xy - dst_xy) ||_2, shape [E, 1]."""
    out = {}
    for et, index in edge_index_dict.items():
        src_x = x_dict[et[0]][index[0], :2]
        dst_x = x_dict[et[-1]][index[-1], :2]
        rel = src_x - dst_x
        rel = -1 * (rel > 0.5) + 1 * (rel < -0.5) + rel
        out[et] = torch.sqrt(rel[:, 0] ** 2 + rel[:, 1] ** 2).view(-1, 1)
    return out


def scale_feature_patchs(factor, x_dict, edge_attr_dict):
    """test.py:29-55, periodic boundary: fold a large domain onto [0,1) training-size patches."""
    for et in edge_attr_dict:
        edge_attr_dict[et] *= factor
    x_dict["grain"][:, :2] *= factor
    x_dict["joint"][:, :2] *= factor
    domain_offset = torch.floor(x_dict["joint"][:, :2])
    x_dict["joint"][:, :2] = x_dict["joint"][:, :2] - domain_offset
    grain_coor_offset = x_dict["grain"][:, :2] - x_dict["grain"][:, :2] % 1
    x_dict["grain"][:, :2] = x_dict["grain"][:, :2] - grain_coor_offset
    return domain_offset, grain_coor_offset


def grain_centres(x_joint_global, gj_edge_index, n_grain):
    """graph_datastruct.py:654-708 (`graph.update`, periodic BC) restated for tensors.  Junction
    order inside a grain is the reference's: `joint2vertex` as `GNN_update(topo=True)` rebuilds it
    (graph_trajectory.py:1066-1085) = junctions in order of first appearance in the grain->joint
    edge list.  Each junction is min-imaged to the PREVIOUS (already moved) one (`periodic_move`,
    :55-72), 1 is added to a coordinate if any vertex of the grain has it below -eps (:698-704),
    then the mean.  Arithmetic types follow the reference's numpy scalars: the vertices are
    float32 (rows of x_dict['joint'].numpy(), graph_trajectory.py:1021,1038); `periodic_move`
    adds an int64 to them, which promotes every vertex but the first to float64, and np.mean of
    the mixed list is float64.  Returns [n_grain, 2] float64; grains with <= 1 junction are NaN
    (the reference skips them, :685)."""
    import numpy as np
    eps = 1e-12  # graph_datastruct.py:36
    xj = x_joint_global.detach().numpy().astype(np.float32)
    tri = {}
    for g, j in zip(gj_edge_index[0].tolist(), gj_edge_index[1].tolist()):
        tri.setdefault(j, set()).add(g)
    members = [[] for _ in range(n_grain)]
    for j in tri:                      # dict order = first appearance
        for g in tri[j]:
            members[g].append(j)
    out = np.full((n_grain, 2), np.nan)
    for g, js in enumerate(members):
        if len(js) <= 1:
            continue
        verts = [[xj[js[0], 0], xj[js[0], 1]]]
        for j in js[1:]:
            p = []
            for d in (0, 1):
                rel = xj[j, d] - verts[-1][d]
                p.append(xj[j, d] + np.int64(-1 * (rel > 0.5) + 1 * (rel < -0.5)))
            verts.append(p)
        inbound = [all(v[d] > -eps for v in verts) for d in (0, 1)]
        moved = [[v[d] + 1 * (not inbound[d]) for d in (0, 1)] for v in verts]
        out[g] = [np.mean([v[0] for v in moved]), np.mean([v[1] for v in moved])]
    return torch.from_numpy(out)


def refresh_grain_centres(x_dict, edge_index_dict, domain_factor=1.0, domain_offset=None):
    """test.py:468-478 + 556-559: junction coordinates back to the global frame
    ((x + domain_offset) / domain_factor when the domain was folded), region centres from
    `graph.update`, written to x_grain[:, :2] ((c * domain_factor) % 1 when folded)."""
    xj = x_dict["joint"][:, :2]
    if domain_factor > 1:
        xj = (xj + domain_offset) / domain_factor
    c = grain_centres(xj, edge_index_dict[("grain", "push", "joint")], x_dict["grain"].size(0))
    c = c.float()  # torch.FloatTensor(coor), test.py:557
    if domain_factor > 1:
        c = (c * domain_factor) % 1
    ok = ~torch.isnan(c[:, 0])
    x_dict["grain"][ok, :2] = c[ok]


@torch.no_grad()
def rollout_step(rmodel, cmodel, x_dict, edge_index_dict, edge_attr_dict, span, centres=None):
    """One static-topology rollout step: R.forward + C.forward + R.update + z advance +
    [grain-centre refresh when `centres` = (domain_factor, domain_offset) is given] +
    edge-length refresh.  Mutates x_dict in place; returns (pred, new edge_attr_dict)."""
    pred = rmodel(x_dict, edge_index_dict, edge_attr_dict)       # test.py:382
    pred.update(cmodel(x_dict, edge_index_dict, edge_attr_dict))  # :383-384
    rmodel.update(x_dict, pred, None)                             # :400
    advance_z(x_dict, span)                                       # :401-407
    if centres is not None:                                       # :468-478, 556-559
        refresh_grain_centres(x_dict, edge_index_dict, *centres)
    return pred, refresh_edge_attr(x_dict, edge_index_dict)       # :562-575
