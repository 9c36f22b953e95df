"""Host-side weight packing: reference `state_dict` layout -> the fused device layouts the
HIP kernels consume.

Layouts (C = 96, G = number of gates, F = features of the node type, Fp = roundup4(F)):

* projection weight of node type T, `Wp [ncols, Kp]`, Kp = Fp + (96 if the cell sees h else 0).
  Rows of Wp = columns of the projection output, in this order:
      for each edge type with source T:       for g: V_g   [96]  lin_value, first three input
                                              columns zeroed (the sweep re-adds W[:, :3] . reloc);
                                              decoder only -- the encoder sweep forms its values
                                              from the edge records (value_fragments below)
      for each edge type with destination T:  for g: u_h_g [96]  (only when the cell sees h)
      summed skip + gate bias:                for g: S_g   [96]  (lin_skip summed over incoming
                                              edge types = HeteroConv aggr 'sum', + b_{i,f,c,o})
      for each edge type with destination T:  for g: u4_g  [16]
      zero rows up to a multiple of 96
  u_h / u4 replace the reference's query AND key (periodGATconv.py:216-217, 226): with
  x~_j = [reloc, x_j[3:], h_j] the score q_i . (W_k x~_j + b_k + w_e a_e) / sqrt(96) equals
  u_i . x~_j + s1_i + a_e s2_i for u_i = W_k^T q_i / sqrt(96), s1_i = b_k . q_i / sqrt(96),
  s2_i = w_e . q_i / sqrt(96); all three are affine in [x_i | h_i], i.e. rows of the destination's
  projection: u_h = the 96 hidden-state components of u, u4 = (u[0:F_src], 0.., s1 @12, s2 @13).
  The products W_k^T W_q are formed in float64 and rounded once.
* edge parameters of an edge type, `EP [G][3][96]`: W_value[:, 0..2].
* gate weight of node type T, `W2 [G][96][Ka]`, Ka = roundup4(98 * n_in):
      [lin_l2.weight of incoming edge type 0 | ... | (b_l2, w_edge) of type 0 | ...]
"""
import math
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
    v_off: Dict[Tuple[str, str, str], int] = field(default_factory=dict)
    u_off: Dict[Tuple[str, str, str], int] = field(default_factory=dict)
    u4_off: Dict[Tuple[str, str, str], int] = field(default_factory=dict)
    a_off: Dict[Tuple[str, str, str], int] = field(default_factory=dict)
    sc_off: Dict[Tuple[str, str, str], int] = field(default_factory=dict)
    s_off: int = 0
    ncols: int = 0
    Ka: int = 0   # width of the gate weight W2 (aggregate columns the gate GEMM reads)
    Kg: int = 0   # gate stride inside an aggregate row: Ka rounded up to 32 floats, so that every
                  # 384-byte block the sweep stores is three whole 128-byte lines
    live: bool = True

    @property
    def Fp(self):
        return roundup4(self.F)


U4 = 16  # width of the per-(destination, gate) tail record: u[0:F_src], s1 @12, s2 @13


def node_layout(node_type: str, F: int, G: int, edge_types=EDGE_TYPES, live: bool = True,
                sees_h: bool = True, with_values: bool = True, live_types=NODE_TYPES) -> NodeLayout:
    """`live=False`: the new (h, c) of this node type is never read (the classifier's decoder only
    feeds h_joint to its head, models.py:595-609), so the type keeps only its role as a message
    SOURCE: no score / skip columns, no aggregation into it, no gate update.
    `sees_h=False` (encoder, h = 0): the hidden-state part of u is not needed.
    `with_values=False` (encoder): no value columns -- the encoder sweep forms the values from the
    edge records on the matrix cores (ggnn_period_gat_aggregate_enc_batch).
    `live_types`: the node types that are live in this cell (value rows towards a dead type are not projected:
    the classifier's decoder never sweeps joint -> grain)."""
    # value rows only for edge types somebody sweeps: a dead destination type (`live_types`) takes no messages
    src_ets = [tuple(et) for et in edge_types if et[0] == node_type and et[-1] in live_types]
    dst_ets = [tuple(et) for et in edge_types if et[-1] == node_type] if live else []
    lay = NodeLayout(F=F, G=G, src_ets=src_ets, dst_ets=dst_ets, live=live)
    off = 0
    for et in src_ets if with_values else ():
        lay.v_off[et] = off
        off += G * C
    if sees_h:
        for et in dst_ets:
            lay.u_off[et] = off
            off += G * C
    lay.s_off = off
    off += G * C if live else 0
    for et in dst_ets:
        lay.u4_off[et] = off
        off += G * U4
    lay.ncols = (off + C - 1) // C * C
    n_in = len(dst_ets)
    lay.Ka = roundup4(n_in * C + 2 * n_in)
    lay.Kg = (max(lay.Ka, 4) + 31) // 32 * 32
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
    ep: Dict[Tuple[str, str, str], torch.Tensor]  # edge type -> [G, 3, 96]
    w2: Dict[str, torch.Tensor]     # node type -> [G, 96, Ka]
    w2p: Dict[str, torch.Tensor] = field(default_factory=dict)  # node type -> fragment-ordered w2 (bf16_planes)
    wvf: Dict[Tuple[str, str, str], torch.Tensor] = field(default_factory=dict)  # encoder: edge type -> value_fragments
    # decoder, fused cell (ggnn_decoder_cell_batch): the projection only emits the source-side value rows ...
    wpv: Dict[str, torch.Tensor] = field(default_factory=dict)   # node type -> [96 G n_src_ets, Kp] value rows of wp
    wpv_f16: bool = False   # the value rows fit the two-piece fp16 arithmetic (finite, below 65504): GGNN_PRECISION_F16X2
    bpv: Dict[str, torch.Tensor] = field(default_factory=dict)
    vof: Dict[Tuple[str, str, str], int] = field(default_factory=dict)  # edge type -> value column in that projection
    # ... and everything on the destination side streams past the tiles as fp16 planes
    # encoder, ONE-kernel cell (ggnn_encoder_cell_batch): node type -> encoder_cell_stream / its (b_l2, w_edge) tails
    ecs: Dict[str, torch.Tensor] = field(default_factory=dict)
    ect: Dict[str, torch.Tensor] = field(default_factory=dict)
    dcs: Dict[str, torch.Tensor] = field(default_factory=dict)   # node type -> decoder_cell_stream (int16)
    dct: Dict[str, torch.Tensor] = field(default_factory=dict)   # node type -> decoder_cell_tail [4, n_in, 6, 64]


@torch.no_grad()
def bf16_planes(w2: torch.Tensor) -> torch.Tensor:
    """`ggnn_epilogue_args.w2_planes` (include/ggnn.h): w2[:, :, :Ka-4] in MFMA A-fragment order, fp32,
    [G][(Ka-4)/32][6][2][64][4]: element [g][ks][ct][h][l][j] = w2[g][16 ct + (l & 15)][32 ks + 8 (l >> 4)
    + 4 h + j], so that one 16-byte load per lane of a wave is 1 KB of contiguous memory.  The kernel
    splits every value exactly into three bf16 pieces (hi = rne(w), mid = rne(w - hi), lo = rne(w - hi
    - mid); hi + mid + lo == w) on the fly; that this split is exact for the given weights is checked
    here.  Returned as int16 bit patterns (two per float), the dtype the binding checks."""
    G, nch, Ka = w2.shape
    KM = Ka - 4
    assert nch == C and KM % 32 == 0
    w = w2[:, :, :KM].float()
    if not bool(torch.isfinite(w).all()):
        raise ValueError("gate weights (lin_l2 / lin_edge) contain non-finite values: cannot be packed")
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    # hi + mid + lo == w exactly unless the last piece underflows bf16's exponent range
    # (|w| < ~2^-110): such entries lose bits far below anything fp32 arithmetic could show
    # (their products are subnormal in the fp32 accumulator) and are accepted
    resid = (hi.float() + mid.float() + lo.float() - w).abs()
    if resid.numel() and float(resid.max()) > 2.0 ** -120:
        raise ValueError(f"bf16 split of the gate weights is not exact (residual {float(resid.max()):.3e})")
    fr = w.view(G, 6, 16, KM // 32, 4, 2, 4)          # g ct i ks kq h j
    fr = fr.permute(0, 3, 1, 5, 4, 2, 6).contiguous()  # g ks ct h kq i j   (lane l = 16 kq + i)
    return fr.view(torch.int16).view(-1)


@torch.no_grad()
def value_fragments(weights, biases, F_src: int) -> torch.Tensor:
    """`ggnn_aggregate_enc_args.wv_frag` (include/ggnn.h): the encoder's lin_value of the G gates of
    one edge type (weights[g]: [96, >= F_src], biases[g]: [96]) as MFMA B fragments of
    Bp [12, G * 96]: Bp[k][g*96 + ch] = W_g[ch][k] (k < F_src), Bp[11] = b_g; element
    [t][s][l] = Bp[4 s + (l >> 4)][(t / 6) * 96 + 32 ((t % 6) / 2) + 2 (l & 15) + t % 2]."""
    G = len(weights)
    if F_src > 11:
        raise ValueError("the encoder sweep keeps its bias in record slot 11: at most 11 source features")
    dev = weights[0].device
    Bp = torch.zeros(12, G * C, dtype=torch.float32, device=dev)
    for g in range(G):
        Bp[:F_src, g * C:(g + 1) * C] = weights[g].detach().float()[:, :F_src].t()
        Bp[11, g * C:(g + 1) * C] = biases[g].detach().float()
    fr = Bp.view(3, 4, G, 3, 16, 2)                  # s kq g m2 j e   (k = 4 s + kq, column = g*96 + 32 m2 + 2 j + e)
    fr = fr.permute(2, 3, 5, 0, 1, 4).contiguous()   # g m2 e s kq j   (tile t = 6 g + 2 m2 + e, lane l = 16 kq + j)
    return fr.view(-1)


DC_PLANES = 2               # fp16 planes of the decoder cell's weight stream: hi, (w - hi) 2^11 (csrc/common.h)
DC_LO_SCALE = 2048.0
DC_SLICE_I16 = 7 * DC_PLANES * 1024 // 2   # GGNN_DC_SLICE_BYTES / 2: int16 elements per slice of the decoder cell's weight stream
DC_GATE_ORDER = (0, 2, 1, 3)  # the fused decoder cell walks the gates i, c~, f, o (weights are indexed i, f, c, o)


def dc_edge_order(gi: int, items):
    """(index, item) pairs in the order the fused decoder cell walks its incoming edge types for the gi-th gate of the
    stream: forwards for gi = 0, 2, backwards for gi = 1, 3 (a pass re-gathers what the pass before it gathered)."""
    seq = list(enumerate(items))
    return seq[::-1] if gi & 1 else seq


@torch.no_grad()
def split2_f16(w: torch.Tensor):
    """fp32 -> the two fp16 pieces of the decoder cell's arithmetic (csrc/common.h: split_f16x2): hi = rne16(w),
    lo' = rne16((w - hi) 2^11); w = hi + lo' / 2^11 up to 2^-22 |w|.  Raises when a weight is not finite or beyond
    fp16's range (the kernel would saturate it)."""
    w = w.float()
    if not bool(torch.isfinite(w).all()) or (w.numel() and float(w.abs().max()) >= 65504.0):
        raise ValueError("weights are not finite or beyond fp16's range: cannot be packed for the decoder cell")
    hi = w.half()
    return hi, ((w - hi.float()) * DC_LO_SCALE).half()


@torch.no_grad()
def _plane_slices(W: torch.Tensor) -> torch.Tensor:
    """[16 NB, 32 NKS] fp32 -> [NKS, DC_SLICE_I16] int16: per k-step the slice image of include/ggnn.h
    (ggnn_dec_cell_args.wstream): [column tile nb][plane hi, lo'][lane l = 16 kq + m][8 fp16] with lane (m, kq)
    of (nb, plane) holding W[16 nb + m][32 ks + 8 kq .. + 7]; slices with fewer than 7 column tiles end in zeros."""
    rows, K = W.shape
    NB, NKS = rows // 16, K // 32
    assert rows == 16 * NB and K == 32 * NKS and NB <= 7
    planes = torch.stack([t.view(torch.int16) for t in split2_f16(W)])          # [2, rows, K]
    P = planes.size(0)
    fr = planes.view(P, NB, 16, NKS, 4, 8)                                      # p nb m ks kq j
    fr = fr.permute(3, 1, 0, 4, 2, 5).reshape(NKS, NB * P * 64 * 8)             # ks | nb p kq m j
    out = torch.zeros(NKS, DC_SLICE_I16, dtype=torch.int16, device=W.device)
    out[:, :fr.size(1)] = fr
    return out


@torch.no_grad()
def decoder_cell_stream(wp, bp, w2, lay: "NodeLayout"):
    """`ggnn_dec_cell_args.wstream` and `.w2_tail` of one destination node type (include/ggnn.h) from the packed
    projection rows `wp` [ncols, Fp + 96] / `bp` and the gate weight `w2` [4, 96, Ka] of pack_cell.
    Reduction index of the score (P1) and skip (P4) blocks: [h 0..95 | x 0..F-1 | 1 (bias) | 0 ..] = 128."""
    F, Fp, G = lay.F, lay.Fp, lay.G
    assert G == 4 and wp.size(1) == Fp + C and F + 1 <= 16
    n_in = len(lay.dst_ets)

    def block(rows, bias):   # [n, Fp + 96] in the projection's input order [x | h] (+ bias) -> [n, 128]
        out = torch.zeros(rows.size(0), 128, dtype=torch.float32, device=wp.device)
        out[:, :C] = rows[:, Fp:Fp + C]
        out[:, C:C + F] = rows[:, :F]
        out[:, C + F] = bias
        return out

    slices = []
    for gi, g in enumerate(DC_GATE_ORDER):
        for d, et in dc_edge_order(gi, lay.dst_ets):
            u = slice(lay.u_off[et] + g * C, lay.u_off[et] + (g + 1) * C)
            t = slice(lay.u4_off[et] + g * U4, lay.u4_off[et] + (g + 1) * U4)
            slices.append(_plane_slices(block(torch.cat([wp[u], wp[t]]), torch.cat([bp[u], bp[t]]))))   # P1: 4 slices
            slices.append(_plane_slices(w2[g][:, d * C:(d + 1) * C].contiguous()))                        # P3: 3 slices
        sk = slice(lay.s_off + g * C, lay.s_off + (g + 1) * C)
        slices.append(_plane_slices(block(wp[sk], bp[sk])))                                                # P4: 4 slices
    stream = torch.cat(slices).contiguous()
    assert stream.size(0) == 4 * (7 * n_in + 4)
    tail = torch.zeros(G, n_in, 6, 4, 16, dtype=torch.float32, device=wp.device)   # g e ct k m   (lane l = 16 k + m)
    for d in range(n_in):
        for k in range(2):
            tail[:, d, :, k, :] = w2[:, :, n_in * C + 2 * d + k].view(G, 6, 16)
    return stream.view(-1), tail.view(G, n_in, 6, 64).contiguous()


# Aggregate channel a lin_l2 column of the fused ENCODER cell's stream multiplies (include/ggnn.h, GGNN_CELL_P3_CHANNEL):
# the kernel's aggregates leave the value MFMAs in lane (node, k-group kq) as channels 16 nb + 4 kq ..+3, nb = 0..5, and
# feed the lin_l2 k-steps as they stand.
CELL_P3_CHANNEL = tuple(32 * (k // 32) + 16 * ((k % 8) // 4) + 4 * ((k % 32) // 8) + k % 4 for k in range(96))


def _spread16(S: torch.Tensor) -> torch.Tensor:
    """[rows, 16 slots] -> [rows, 32]: one k-step of the encoder cell's 16-slot products, k = 8 q + j holds slot
    4 q + j for j < 4 and zero for j >= 4 (a lane's fragment: its four slots in the low half)."""
    out = torch.zeros(S.size(0), 32, dtype=S.dtype, device=S.device)
    out.view(-1, 4, 8)[:, :, :4] = S.view(-1, 4, 4)
    return out


@torch.no_grad()
def encoder_cell_stream(wp, bp, w2, lay: "NodeLayout", values, F_src):
    """`ggnn_enc_cell_args.wstream` and `.w2_tail` of one destination node type (include/ggnn.h) from the packed
    projection rows `wp` [ncols, Fp] / `bp` (score tails u4, summed skip) and the gate weight `w2` [3, 96, Ka] of
    pack_cell; `values[et][g]` = (lin_value.weight [96, F_src], lin_value.bias) of the incoming edge types, `F_src[et]`
    their source feature counts.  Slot 12 of a feature row / edge record is 1: the bias column."""
    F, G = lay.F, lay.G
    assert G == 3 and F <= 12
    n_in = len(lay.dst_ets)
    dev = wp.device

    def dst_block(rows):   # packed projection rows (input = the destination's features) -> [n, 16 slots]
        out = torch.zeros(rows.numel(), 16, dtype=torch.float32, device=dev)
        out[:, :F] = wp[rows][:, :F]
        out[:, 12] = bp[rows]
        return out

    p3 = torch.tensor(CELL_P3_CHANNEL, device=dev)
    slices = []
    for g in range(G):
        for d, et in enumerate(lay.dst_ets):
            Fs = F_src[et]
            assert Fs <= 12
            wv, bv = values[et][g]
            V = torch.zeros(C, 16, dtype=torch.float32, device=dev)
            V[:, :Fs] = wv.detach().float()[:, :Fs]     # record slots 0..2 = reloc, 3..Fs-1 = x_src[3:Fs]
            V[:, 12] = bv.detach().float()
            T = dst_block(torch.arange(lay.u4_off[et] + g * U4, lay.u4_off[et] + (g + 1) * U4, device=dev))
            slices.append(_plane_slices(_spread16(torch.cat([V, T]))))                      # A(e, g): 1 slice
            slices.append(_plane_slices(w2[g][:, p3 + d * C].contiguous()))                  # lin_l2(e, g): 3 slices
        slices.append(_plane_slices(_spread16(dst_block(torch.arange(lay.s_off + g * C, lay.s_off + (g + 1) * C,
                                                                     device=dev)))))        # S(g): 1 slice
    stream = torch.cat(slices).contiguous()
    assert stream.size(0) == G * (4 * n_in + 1)
    tail = torch.zeros(G, n_in, 6, 4, 16, dtype=torch.float32, device=dev)   # g e ct k m   (lane l = 16 k + m)
    for d in range(n_in):
        for k, slot in enumerate((0, 3)):   # b_l2 meets sum alpha in k-group 0, w_edge meets sum alpha a_e in k-group 3
            tail[:, d, :, slot, :] = w2[:, :, n_in * C + 2 * d + k].view(G, 6, 16)
    return stream.view(-1), tail.view(G, n_in, 6, 64).contiguous()


def _conv(cell, gate, et):
    return getattr(cell, "conv_" + gate).convs[et_key(et)]


# ---- layer_size < 96 (parameters.py:19: the regressor's grid also holds 64 and 32) ---------------------------------
# The kernels are built for 96 hidden channels.  A narrower model runs on them ZERO-PADDED: every [c]-sized axis of a
# parameter (output rows, the hidden-state part of the input columns) is padded to 96, and the query side is scaled by
# sqrt(96 / c) (periodGATconv.py:226 divides the scores by sqrt(out_channels); the packed weights fold in 1 / sqrt(96)).
# The padded channels stay exactly zero through a cell -- zero rows of lin_value / lin_skip / lin_l2 / lin_edge and a
# zero gate bias give pre-activations 0: i = f = o = 1/2, c~ = 0, so c' = c / 2 = 0 and h' = tanh(0) / 2 = 0 -- and meet
# zero columns everywhere they are read.  Same results as the c-wide model up to fp32 summation order; same speed as
# the 96-wide one.
class _Holder:
    """Attribute bag standing in for an nn.Module in the packers (`.weight`, `.bias`, `.convs[...]`, ...)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _pad_axis(t, dim, c, F=None):
    """Zero-pad axis `dim` of `t` from c to C; with F: the axis is [F features | c hidden] -> [F | C]."""
    if t is None:
        return None
    t = t.detach()
    lead = 0 if F is None else F
    if t.size(dim) != lead + c:
        raise ValueError(f"parameter axis of {t.size(dim)} entries, expected {lead + c}")
    shape = list(t.shape)
    shape[dim] = lead + C
    out = t.new_zeros(shape)
    out.narrow(dim, 0, lead + c).copy_(t)
    return out


def padded_conv(conv, c: int):
    """A PeriodConv of width c < 96 as a holder of 96-wide tensors (see above)."""
    Ds, Dd = conv.in_channels
    has_h = Ds > 12
    Fs, Fd = (Ds - c, Dd - c) if has_h else (Ds, Dd)
    q = math.sqrt(C / c)

    def lin(m, F, scale=1.0, square=False):
        w = _pad_axis(m.weight, 0, c)
        if square:
            w = _pad_axis(w, 1, c)
        elif has_h and F is not None:
            w = _pad_axis(w, 1, c, F)
        b = _pad_axis(m.bias, 0, c) if m.bias is not None else None
        return _Holder(weight=w * scale, bias=None if b is None else b * scale)
    return _Holder(in_channels=(Fs + C, Fd + C) if has_h else (Ds, Dd), out_channels=C,
                   lin_key=lin(conv.lin_key, Fs), lin_query=lin(conv.lin_query, Fd, q), lin_value=lin(conv.lin_value, Fs),
                   lin_l2=lin(conv.lin_l2, None, square=True), lin_edge=lin(conv.lin_edge, None),
                   lin_skip=lin(conv.lin_skip, Fd))


@torch.no_grad()
def padded_cell(cell, c: int):
    """A HeteroPGCLSTM of width c < 96 as a holder tree of 96-wide tensors with the attribute names pack_cell reads."""
    out = _Holder()
    for g in "ifco":
        hc = getattr(cell, "conv_" + g)
        setattr(out, "conv_" + g, _Holder(convs={k: padded_conv(m, c) for k, m in hc.convs.items()}))
        setattr(out, "b_" + g, {nt: _pad_axis(b, 1, c) for nt, b in getattr(cell, "b_" + g).items()})
    return out


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
    F_of = dict(in_channels)
    # encoder: the sweep forms the values from the edge records (no value columns in the projection)
    enc_mfma = encoder and all(F <= 11 for F in in_channels.values())
    layout = {nt: node_layout(nt, in_channels[nt], G, edge_types, nt in live, not encoder, not enc_mfma, tuple(live))
              for nt in NODE_TYPES}
    wp, bp, w2, ep = {}, {}, {}, {}

    def put(dst_w, dst_b, row0, F, Fp, weight, bias, zero_xyz=False):
        """weight: [n, F + 96] in the reference's input order [x | h] -> n rows of the packed matrix
        from row0 on (input order [x, pad to Fp | h]); bias is ADDED to the packed bias."""
        w = weight.detach().to(dt)
        n = w.size(0)
        blk = dst_w[row0:row0 + n]
        blk[:, :F] += w[:, :F]
        if zero_xyz:
            blk[:, :3] = 0.0
        if k2:
            blk[:, Fp:Fp + C] += w[:, F:F + C]
        dst_b[row0:row0 + n] += bias.detach().to(dt)

    scale = 1.0 / math.sqrt(C)  # periodGATconv.py:226
    for nt in NODE_TYPES:
        lay = layout[nt]
        F, Fp = lay.F, lay.Fp
        W = torch.zeros(lay.ncols, Fp + k2, dtype=dt, device=dev)
        B = torch.zeros(lay.ncols, dtype=dt, device=dev)
        for et in lay.src_ets if not enc_mfma else ():
            for g, gate in enumerate(gates):
                conv = _conv(cell, gate, et)
                put(W, B, lay.v_off[et] + g * C, F, Fp, conv.lin_value.weight, conv.lin_value.bias, True)
        for et in lay.dst_ets:
            Fs = F_of[et[0]]
            for g, gate in enumerate(gates):
                conv = _conv(cell, gate, et)
                wq, bq = conv.lin_query.weight.detach().double(), conv.lin_query.bias.detach().double()
                wk, bk = conv.lin_key.weight.detach().double(), conv.lin_key.bias.detach().double()
                we = conv.lin_edge.weight.detach().double()[:, 0]
                if not k2:  # encoder: both sides see the bare feature row
                    wq, wk = wq[:, :F], wk[:, :Fs]
                M, mb = (wk.t() @ wq) * scale, (wk.t() @ bq) * scale         # [Fs (+96), F (+96)], [Fs (+96)]
                tail_w = torch.zeros(U4, wq.size(1), dtype=torch.float64, device=dev)
                tail_b = torch.zeros(U4, dtype=torch.float64, device=dev)
                tail_w[:Fs], tail_b[:Fs] = M[:Fs], mb[:Fs]
                tail_w[12], tail_b[12] = (bk @ wq) * scale, (bk @ bq) * scale
                tail_w[13], tail_b[13] = (we @ wq) * scale, (we @ bq) * scale
                put(W, B, lay.u4_off[et] + g * U4, F, Fp, tail_w, tail_b)
                if k2:
                    put(W, B, lay.u_off[et] + g * C, F, Fp, M[Fs:], mb[Fs:])
        for g, gate in enumerate(gates if lay.live else ()):
            row0 = lay.s_off + g * C
            for et in lay.dst_ets:  # HeteroConv sums the outputs -> sum the skip weights
                conv = _conv(cell, gate, et)
                put(W, B, row0, F, Fp, conv.lin_skip.weight, conv.lin_skip.bias)
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
        E = torch.zeros(G, 3, C, dtype=dt, device=dev)
        for g, gate in enumerate(gates):
            E[g] = _conv(cell, gate, et).lin_value.weight.detach().to(dt)[:, 0:3].t()
        ep[et] = E.contiguous()
    w2p = {nt: bf16_planes(t) for nt, t in w2.items()}
    wvf = {}
    if enc_mfma:
        for et in ep:
            convs = [_conv(cell, gate, et) for gate in gates]
            wvf[et] = value_fragments([cv.lin_value.weight for cv in convs], [cv.lin_value.bias for cv in convs],
                                      F_of[et[0]])
    ecs, ect = {}, {}
    if encoder and all(F <= 12 for F in in_channels.values()):   # the encoder cell as ONE kernel: its weight stream
        try:
            for nt in NODE_TYPES:
                lay = layout[nt]
                if lay.live:
                    vals = {et: [(_conv(cell, gate, et).lin_value.weight, _conv(cell, gate, et).lin_value.bias)
                                 for gate in gates] for et in lay.dst_ets}
                    ecs[nt], ect[nt] = encoder_cell_stream(wp[nt], bp[nt], w2[nt], lay, vals,
                                                           {et: F_of[et[0]] for et in lay.dst_ets})
        except ValueError:   # a weight beyond fp16's range (or not finite): sweep + gate GEMM launches instead
            ecs, ect = {}, {}
    wpv, bpv, vof, dcs, dct = {}, {}, {}, {}, {}
    if k2 and all(F + 1 <= 16 for F in in_channels.values()):   # decoder: the fused cell's operands
        for nt in NODE_TYPES:
            lay = layout[nt]
            ets = [et for et in lay.src_ets if layout[et[-1]].live]   # value rows nobody sweeps are not projected
            if ets:
                idx = torch.cat([torch.arange(lay.v_off[et], lay.v_off[et] + G * C, device=dev) for et in ets])
                wpv[nt], bpv[nt] = wp[nt][idx].contiguous(), bp[nt][idx].contiguous()
                for k, et in enumerate(ets):
                    vof[et] = k * G * C
        try:
            for nt in NODE_TYPES:
                if layout[nt].live:
                    dcs[nt], dct[nt] = decoder_cell_stream(wp[nt], bp[nt], w2[nt], layout[nt])
        except ValueError:   # a weight that is not finite or beyond fp16's range: the cell runs on the three-kernel
            dcs, dct = {}, {}   # plan (projection + sweeps + gate GEMM, bf16 x 3: the full fp32 range, NaNs propagate)
    # the value projection of the fused plan runs in the cells' arithmetic (three products) when its weights allow
    wpv_f16 = bool(dcs) and all(bool(torch.isfinite(w).all()) and float(w.abs().max()) < 65504.0 for w in wpv.values())
    return PackedCell(G=G, k2=k2, layout=layout, wp=wp, bp=bp, ep=ep, w2=w2, w2p=w2p, wvf=wvf, ecs=ecs, ect=ect,
                      wpv=wpv, bpv=bpv, vof=vof, dcs=dcs, dct=dct, wpv_f16=wpv_f16)


@torch.no_grad()
def pack_conv(conv, F_src: int, F_dst: int, k2: int):
    """Single PeriodConv (one gate, one edge type) for op-level parity tests.  Returns
    (wp_src [96, Kp_s], bp_src, wp_dst [288, Kp_d], bp_dst, ep [1,3,96], w2 [1,96,100]):
    source projection = [V]; destination projection = [u_h | S | u4, zero rows] (u_h unused and
    zero when k2 == 0)."""
    dt = torch.float32
    dev = conv.lin_key.weight.device

    def pack(weight, F, zero_xyz):
        """[n, F (+96)] reference input order -> [n, roundup4(F) + k2]."""
        Fp = roundup4(F)
        w = weight.detach().to(dt)
        out = torch.zeros(w.size(0), Fp + k2, dtype=dt, device=dev)
        out[:, :F] = w[:, :F]
        if zero_xyz:
            out[:, :3] = 0.0
        if k2:
            out[:, Fp:Fp + C] = w[:, F:F + C]
        return out

    wp_src = pack(conv.lin_value.weight, F_src, True)
    bp_src = conv.lin_value.bias.detach().to(dt)
    scale = 1.0 / math.sqrt(C)
    wq, bq = conv.lin_query.weight.detach().double(), conv.lin_query.bias.detach().double()
    wk, bk = conv.lin_key.weight.detach().double(), conv.lin_key.bias.detach().double()
    we = conv.lin_edge.weight.detach().double()[:, 0]
    if not k2:
        wq, wk = wq[:, :F_dst], wk[:, :F_src]
    M, mb = (wk.t() @ wq) * scale, (wk.t() @ bq) * scale
    tail_w = torch.zeros(U4, wq.size(1), dtype=torch.float64, device=dev)
    tail_b = torch.zeros(U4, dtype=torch.float64, device=dev)
    tail_w[:F_src], tail_b[:F_src] = M[:F_src], mb[:F_src]
    tail_w[12], tail_b[12] = (bk @ wq) * scale, (bk @ bq) * scale
    tail_w[13], tail_b[13] = (we @ wq) * scale, (we @ bq) * scale
    uh_w = M[F_src:] if k2 else torch.zeros(C, wq.size(1), dtype=torch.float64, device=dev)
    uh_b = mb[F_src:] if k2 else torch.zeros(C, dtype=torch.float64, device=dev)
    wp_dst = torch.cat([pack(uh_w, F_dst, False), pack(conv.lin_skip.weight, F_dst, False),
                        pack(tail_w, F_dst, False),
                        torch.zeros(C - U4, roundup4(F_dst) + k2, dtype=dt, device=dev)])
    bp_dst = torch.cat([uh_b.to(dt), conv.lin_skip.bias.detach().to(dt), tail_b.to(dt),
                        torch.zeros(C - U4, dtype=dt, device=dev)])
    ep = conv.lin_value.weight.detach().to(dt)[:, 0:3].t().reshape(1, 3, C).contiguous()
    w2 = torch.zeros(1, C, 100, dtype=dt, device=dev)
    w2[0, :, :C] = conv.lin_l2.weight.detach().to(dt)
    w2[0, :, C] = conv.lin_l2.bias.detach().to(dt)
    w2[0, :, C + 1] = conv.lin_edge.weight.detach().to(dt)[:, 0]
    return (wp_src.contiguous(), bp_src.contiguous(), wp_dst.contiguous(), bp_dst.contiguous(), ep, w2.contiguous())


@torch.no_grad()
def pack_regressor_heads(linear):
    """`linear` = ModuleDict {'grain','joint'} of Linear(layer_size, 2) (models.py:393-394); narrower than 96: zero
    columns for the padded channels."""
    c = linear["joint"].weight.size(1)
    wj, wg = (_pad_axis(linear[nt].weight, 1, c) if c != C else linear[nt].weight for nt in ("joint", "grain"))
    w = torch.stack([wj, wg]).detach().float().contiguous()
    b = torch.cat([linear["joint"].bias, linear["grain"].bias]).detach().float().contiguous()
    return w, b  # [2, 2, 96], [4]


@torch.no_grad()
def pack_classifier_heads(lin1, lin2):
    """lin1: Linear(2 layer_size + 1, 2), lin2: Linear(2 layer_size + 1, 1) (models.py:568-569); input order
    [h_src | h_dst | edge length].  Narrower than 96: zero columns for the padded channels of both hidden blocks."""
    w1, w2 = lin1.weight.detach().float(), lin2.weight.detach().float()
    c = (w1.size(1) - 1) // 2
    if c != C:
        widen = lambda w: torch.cat([_pad_axis(w[:, :c], 1, c), _pad_axis(w[:, c:2 * c], 1, c), w[:, 2 * c:]], 1)
        w1, w2 = widen(w1), widen(w2)
    w_node = torch.stack([w1[0, :C], w1[1, :C], w2[0, :C], w1[0, C:2 * C], w1[1, C:2 * C], w2[0, C:2 * C]])
    w_edge = torch.stack([w1[0, 2 * C], w1[1, 2 * C], w2[0, 2 * C], lin1.bias[0].detach().float(),
                          lin1.bias[1].detach().float(), lin2.bias[0].detach().float()])
    return w_node.contiguous(), w_edge.contiguous()  # [6, 96], [6]
