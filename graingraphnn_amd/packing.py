"""Host-side weight packing: reference `state_dict` layout -> the fused device layouts the
HIP kernels consume.  Pure tensor reshuffling (no arithmetic except summing the `lin_skip`
weights of edge types that share a destination, which is what HeteroConv(aggr='sum') does to
their outputs -- heteropgclstm.py:49-82 -- and folding the gate bias b_{i,f,c,o} in).

Layouts (C = 96, G = number of gates, F = features of the node type, Fp = roundup4(F)):

* projection weight of node type T, `Wp [ncols, Kp]`, Kp = Fp + (96 if the cell sees h else 0):
    columns of the OUTPUT (rows of Wp), in blocks of 96:
      for each edge type with source T (canonical order):  for g: [K_g | V_g]
      for each edge type with destination T:               for g: [Q_g]
      summed skip + gate bias:                             for g: [S_g]
    K/V rows have their first three input columns zeroed: the aggregation kernel re-adds
    `W[:, :3] . minimg(x_j - x_i)` per edge (periodGATconv.py:209-211).
* edge parameters of an edge type, `EP [G][7][96]`: W_key[:, 0..2], W_value[:, 0..2], w_edge.
* gate weight of node type T, `W2 [G][96][Ka]`, Ka = roundup4(98 * n_in):
      [lin_l2.weight of incoming edge type 0 | ... | (b_l2, w_edge) of type 0 | ...]
"""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch

C = 96
NODE_TYPES = ("grain", "joint")
EDGE_TYPES = (("grain", "push", "joint"), ("joint", "pull", "grain"), ("joint", "connect", "joint"))
GATES_DEC = ("i", "f", "c", "o")
GATES_ENC = ("i", "c", "o")  # h = c = 0: the forget gate multiplies c = 0 (heteropgclstm.py:132)


def et_key(et) -> str:
    return "__".join(et)


def roundup4(n: int) -> int:
    return (n + 3) & ~3


@dataclass
class NodeLayout:
    """Column offsets inside the projection buffer / aggregate buffer of one node type."""
    F: int
    G: int
    src_ets: List[Tuple[str, str, str]]
    dst_ets: List[Tuple[str, str, str]]
    kv_off: Dict[Tuple[str, str, str], int] = field(default_factory=dict)
    q_off: Dict[Tuple[str, str, str], int] = field(default_factory=dict)
    a_off: Dict[Tuple[str, str, str], int] = field(default_factory=dict)
    sc_off: Dict[Tuple[str, str, str], int] = field(default_factory=dict)
    s_off: int = 0
    ncols: int = 0
    Ka: int = 0
    live: bool = True

    @property
    def Fp(self):
        return roundup4(self.F)


ENC_W_ROW = 40  # floats per channel in the fused-encoder weight record (include/ggnn.h)
# Fused encoder sweep (ggnn_period_gat_aggregate_enc): recompute K0/V0/Q per edge from the
# feature rows instead of projecting them to HBM.  Parity-green, but measured SLOWER on cfg3
# (r1: 3 sweeps 133 us + projection 21 us vs 84 us + 47 us for projection + gather): the sweep
# is VALU-issue-bound at 150 VGPRs / 3 waves per SIMD.  Kept behind this switch for tuning.
FUSE_ENCODER = False
# (f_src, f_dst) pairs the fused encoder sweep is instantiated for
ENC_FUSED_SHAPES = ((11, 8), (8, 11), (8, 8))


def node_layout(node_type: str, F: int, G: int, edge_types=EDGE_TYPES, live: bool = True,
                fused: bool = False) -> NodeLayout:
    """`live=False`: the new (h, c) of this node type is never read (the classifier's decoder only
    feeds h_joint to its head, models.py:595-609), so the type keeps only its role as a message
    SOURCE: no query / skip columns, no aggregation into it, no gate update."""
    src_ets = [tuple(et) for et in edge_types if et[0] == node_type]
    dst_ets = [tuple(et) for et in edge_types if et[-1] == node_type] if live else []
    lay = NodeLayout(F=F, G=G, src_ets=src_ets, dst_ets=dst_ets, live=live)
    off = 0
    if not fused:  # fused encoder: key / value / query never leave the aggregation kernel
        for et in src_ets:
            lay.kv_off[et] = off
            off += G * 2 * C
        for et in dst_ets:
            lay.q_off[et] = off
            off += G * C
    lay.s_off = off
    lay.ncols = off + (G * C if live else 0)
    n_in = len(dst_ets)
    lay.Ka = roundup4(n_in * C + 2 * n_in)
    for d, et in enumerate(dst_ets):
        lay.a_off[et] = d * C
        lay.sc_off[et] = n_in * C + 2 * d
    return lay


@dataclass
class PackedCell:
    """Device-resident fused parameters of one HeteroPGCLSTM cell."""
    G: int
    k2: int  # 96 (decoder: input is cat[x, h]) or 0 (encoder: h = 0)
    layout: Dict[str, NodeLayout]
    wp: Dict[str, torch.Tensor]     # node type -> [ncols, Kp]
    bp: Dict[str, torch.Tensor]     # node type -> [ncols]
    ep: Dict[Tuple[str, str, str], torch.Tensor]  # edge type -> [G, 7, 96]
    w2: Dict[str, torch.Tensor]     # node type -> [G, 96, Ka]
    w2p: Dict[str, torch.Tensor] = field(default_factory=dict)  # node type -> bf16 planes of w2 (bf16_planes)
    enc_w: Dict[Tuple[str, str, str], torch.Tensor] = field(default_factory=dict)  # fused encoder: [G, 96, 40]
    fused: bool = False


@torch.no_grad()
def bf16_planes(w2: torch.Tensor) -> torch.Tensor:
    """`ggnn_epilogue_args.w2_planes` (include/ggnn.h): w2[:, :, :Ka-4] split exactly into three
    bf16 pieces (hi = rne(w), mid = rne(w - hi), lo = rne(w - hi - mid); hi + mid + lo == w) and
    laid out in MFMA fragment order [G][(Ka-4)/32][3][6][64][8] as int16 bit patterns."""
    G, nch, Ka = w2.shape
    KM = Ka - 4
    assert nch == C and KM % 32 == 0
    w = w2[:, :, :KM].float()
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    mid = r1.to(torch.bfloat16)
    r2 = r1 - mid.float()
    lo = r2.to(torch.bfloat16)
    assert torch.equal(hi.float() + mid.float() + lo.float(), w)
    pl = torch.stack([hi, mid, lo], 0).view(3, G, 6, 16, KM // 32, 4, 8)   # p g ct i ks kq j
    pl = pl.permute(1, 4, 0, 2, 5, 3, 6).contiguous()                      # g ks p ct kq i j
    return pl.view(torch.int16).view(-1)


def _conv(cell, gate, et):
    return getattr(cell, "conv_" + gate).convs[et_key(et)]


@torch.no_grad()
def pack_cell(cell, in_channels: Dict[str, int], encoder: bool, edge_types=EDGE_TYPES,
              live=NODE_TYPES) -> PackedCell:
    """`cell` is a HeteroPGCLSTM parameter holder (same attribute tree as heteropgclstm.py:30-99).
    `live`: node types whose new state is consumed downstream (see node_layout)."""
    gates = GATES_ENC if encoder else GATES_DEC
    G = len(gates)
    k2 = 0 if encoder else C
    some = _conv(cell, "i", edge_types[0]).lin_key.weight
    dev, dt = some.device, torch.float32
    fused = (FUSE_ENCODER and encoder and
             all((in_channels[et[0]], in_channels[et[-1]]) in ENC_FUSED_SHAPES for et in edge_types))
    layout = {nt: node_layout(nt, in_channels[nt], G, edge_types, nt in live, fused) for nt in NODE_TYPES}
    wp, bp, w2, ep, enc_w = {}, {}, {}, {}, {}

    def put(dst_w, dst_b, row0, F, Fp, weight, bias, zero_xyz):
        """weight: [96, F + 96] reference layout -> rows row0..row0+95 of the packed matrix."""
        w = weight.detach().to(dt)
        blk = dst_w[row0:row0 + C]
        blk[:, :F] = w[:, :F]
        if zero_xyz:
            blk[:, :3] = 0.0
        if k2:
            blk[:, Fp:Fp + C] = w[:, F:F + C]
        dst_b[row0:row0 + C] += bias.detach().to(dt)

    for nt in NODE_TYPES:
        lay = layout[nt]
        F, Fp = lay.F, lay.Fp
        W = torch.zeros(lay.ncols, Fp + k2, dtype=dt, device=dev)
        B = torch.zeros(lay.ncols, dtype=dt, device=dev)
        for et in (() if fused else lay.src_ets):
            for g, gate in enumerate(gates):
                conv = _conv(cell, gate, et)
                base = lay.kv_off[et] + g * 2 * C
                put(W, B, base, F, Fp, conv.lin_key.weight, conv.lin_key.bias, True)
                put(W, B, base + C, F, Fp, conv.lin_value.weight, conv.lin_value.bias, True)
        for et in (() if fused else lay.dst_ets):
            for g, gate in enumerate(gates):
                conv = _conv(cell, gate, et)
                put(W, B, lay.q_off[et] + g * C, F, Fp, conv.lin_query.weight, conv.lin_query.bias, False)
        for g, gate in enumerate(gates if lay.live else ()):
            row0 = lay.s_off + g * C
            for et in lay.dst_ets:  # HeteroConv sums the outputs -> sum the skip weights
                conv = _conv(cell, gate, et)
                w = conv.lin_skip.weight.detach().to(dt)
                W[row0:row0 + C, :F] += w[:, :F]
                if k2:
                    W[row0:row0 + C, Fp:Fp + C] += w[:, F:F + C]
                B[row0:row0 + C] += conv.lin_skip.bias.detach().to(dt)
            B[row0:row0 + C] += getattr(cell, "b_" + gate)[nt].detach().to(dt).view(-1)
        wp[nt], bp[nt] = W.contiguous(), B.contiguous()

        n_in = len(lay.dst_ets)
        W2 = torch.zeros(G, C, max(lay.Ka, 4), dtype=dt, device=dev)
        for d, et in enumerate(lay.dst_ets):
            for g, gate in enumerate(gates):
                conv = _conv(cell, gate, et)
                W2[g, :, d * C:(d + 1) * C] = conv.lin_l2.weight.detach().to(dt)
                W2[g, :, n_in * C + 2 * d] = conv.lin_l2.bias.detach().to(dt)
                W2[g, :, n_in * C + 2 * d + 1] = conv.lin_edge.weight.detach().to(dt)[:, 0]
        w2[nt] = W2.contiguous()

    for et in edge_types:
        et = tuple(et)
        if not layout[et[-1]].live:
            continue
        E = torch.zeros(G, 7, C, dtype=dt, device=dev)
        for g, gate in enumerate(gates):
            conv = _conv(cell, gate, et)
            E[g, 0:3] = conv.lin_key.weight.detach().to(dt)[:, 0:3].t()
            E[g, 3:6] = conv.lin_value.weight.detach().to(dt)[:, 0:3].t()
            E[g, 6] = conv.lin_edge.weight.detach().to(dt)[:, 0]
        ep[et] = E.contiguous()
        if fused:
            Fs, Fd = in_channels[et[0]], in_channels[et[-1]]
            Wf = torch.zeros(G, C, ENC_W_ROW, dtype=dt, device=dev)
            for g, gate in enumerate(gates):
                conv = _conv(cell, gate, et)
                wq, wk, wv = (m.weight.detach().to(dt) for m in (conv.lin_query, conv.lin_key, conv.lin_value))
                Wf[g, :, 0:Fd] = wq[:, :Fd]
                Wf[g, :, 12] = conv.lin_query.bias.detach().to(dt)
                Wf[g, :, 13:13 + Fs - 3] = wk[:, 3:Fs]
                Wf[g, :, 21] = conv.lin_key.bias.detach().to(dt)
                Wf[g, :, 22:22 + Fs - 3] = wv[:, 3:Fs]
                Wf[g, :, 30] = conv.lin_value.bias.detach().to(dt)
                Wf[g, :, 31:34] = wk[:, 0:3]
                Wf[g, :, 34:37] = wv[:, 0:3]
                Wf[g, :, 37] = conv.lin_edge.weight.detach().to(dt)[:, 0]
            enc_w[et] = Wf.contiguous()
    w2p = {nt: bf16_planes(t) for nt, t in w2.items()}
    return PackedCell(G=G, k2=k2, layout=layout, wp=wp, bp=bp, ep=ep, w2=w2, enc_w=enc_w, fused=fused,
                      w2p=w2p)


@torch.no_grad()
def pack_conv(conv, F_src: int, F_dst: int, k2: int):
    """Single PeriodConv (one gate, one edge type) for op-level parity tests.  Returns
    (wp_src [192, Kp_s], bp_src, wp_dst [192, Kp_d], bp_dst, ep [1,7,96], w2 [1,96,100]):
    source projection = [K | V], destination projection = [Q | S]."""
    dt = torch.float32
    dev = conv.lin_key.weight.device

    def pack(weight, F, zero_xyz):
        Fp = roundup4(F)
        w = weight.detach().to(dt)
        out = torch.zeros(C, Fp + k2, dtype=dt, device=dev)
        out[:, :F] = w[:, :F]
        if zero_xyz:
            out[:, :3] = 0.0
        if k2:
            out[:, Fp:Fp + C] = w[:, F:F + C]
        return out

    wp_src = torch.cat([pack(conv.lin_key.weight, F_src, True), pack(conv.lin_value.weight, F_src, True)])
    bp_src = torch.cat([conv.lin_key.bias, conv.lin_value.bias]).detach().to(dt)
    wp_dst = torch.cat([pack(conv.lin_query.weight, F_dst, False), pack(conv.lin_skip.weight, F_dst, False)])
    bp_dst = torch.cat([conv.lin_query.bias, conv.lin_skip.bias]).detach().to(dt)
    ep = torch.zeros(1, 7, C, dtype=dt, device=dev)
    ep[0, 0:3] = conv.lin_key.weight.detach().to(dt)[:, 0:3].t()
    ep[0, 3:6] = conv.lin_value.weight.detach().to(dt)[:, 0:3].t()
    ep[0, 6] = conv.lin_edge.weight.detach().to(dt)[:, 0]
    w2 = torch.zeros(1, C, 100, dtype=dt, device=dev)
    w2[0, :, :C] = conv.lin_l2.weight.detach().to(dt)
    w2[0, :, C] = conv.lin_l2.bias.detach().to(dt)
    w2[0, :, C + 1] = conv.lin_edge.weight.detach().to(dt)[:, 0]
    return (wp_src.contiguous(), bp_src.contiguous(), wp_dst.contiguous(), bp_dst.contiguous(),
            ep.contiguous(), w2.contiguous())


@torch.no_grad()
def pack_regressor_heads(linear):
    """`linear` = ModuleDict {'grain','joint'} of Linear(96, 2) (models.py:393-394)."""
    w = torch.stack([linear["joint"].weight, linear["grain"].weight]).detach().float().contiguous()
    b = torch.cat([linear["joint"].bias, linear["grain"].bias]).detach().float().contiguous()
    return w, b  # [2, 2, 96], [4]


@torch.no_grad()
def pack_classifier_heads(lin1, lin2):
    """lin1: Linear(193, 2), lin2: Linear(193, 1) (models.py:568-569)."""
    w1, w2 = lin1.weight.detach().float(), lin2.weight.detach().float()
    w_node = torch.stack([w1[0, :C], w1[1, :C], w2[0, :C], w1[0, C:2 * C], w1[1, C:2 * C], w2[0, C:2 * C]])
    w_edge = torch.stack([w1[0, 2 * C], w1[1, 2 * C], w2[0, 2 * C], lin1.bias[0].detach().float(),
                          lin1.bias[1].detach().float(), lin2.bias[0].detach().float()])
    return w_node.contiguous(), w_edge.contiguous()  # [6, 96], [6]
