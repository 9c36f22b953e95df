// Decoder HeteroPGCLSTM cell with everything that belongs to a destination node in ONE kernel
// (ggnn_decoder_cell_batch, include/ggnn.h): the destination-side projections (u_h | u4 per edge type and
// gate, the summed skip term), the periodic-boundary GAT sweep (PeriodConv.message, periodGATconv.py:204-236,
// + propagate's gather / scatter-add), lin_l2 + the value-side lin_edge term, HeteroConv's sum over the edge
// types and the LSTM update (heteropgclstm.py:111-146).  Replaces two thirds of the decoder projection's
// columns, ggnn_period_gat_aggregate_batch and ggnn_lstm_epilogue_batch: per model forward at the 10k-grain
// graph 142 MB of u_h / u4 / S rows and 92 MB of aggregates that were written and read back once each.
//
// Why it is organised around a 16-node tile per WAVE with the weights streaming past:
//   * the three GEMMs of a destination node (score weights K = 104, lin_l2 K = 96, skip K = 104) chain through
//     the MFMA layouts without a transpose when the NODES are the B operand: D[out][node] leaves lane
//     (node l & 15, out rows 4 (l >> 4) ..+3), and a B fragment wants lane (node l & 15, k = 8 (l >> 4) ..+7);
//   * the sweep wants the other layout (a node's 96 channels across the 16 lanes of a DPP row: whole 384-byte
//     rows per gather, dot products closed with four DPP adds), so u and the aggregates cross a wave-private
//     LDS stage ([16][116] floats) once each way -- 7 KB per wave instead of 57 + 50 KB of u / agg per tile if
//     all gates were kept at once: the gates are walked ONE AFTER THE OTHER (i, c~, f, o) and the LSTM update is
//     folded in as they arrive (sig(i) -> sig(i) tanh(c~) -> c' -> h'), so a wave holds one gate's 16 x 96
//     pre-activations (24 VGPRs) plus the running LSTM term (24), never four;
//   * the weights of a destination type are 0.84 MB as two fp16 planes (joints: 2 x 4 score blocks [112 x 128],
//     8 lin_l2 blocks, 4 skip blocks) -- five times the LDS.  They arrive as k-step slices (one 32-deep
//     k-step of one block: 14 KB) through a double-buffered LDS region shared by the workgroup's eight waves,
//     fetched one slice ahead by LDS-DMA, one workgroup barrier per slice.  Host-side pre-split planes: no
//     splitting arithmetic on the weight side in the kernel (the gate kernel's split cost 44 VALU per fragment).
//   * arithmetic: every fp32 operand as TWO fp16 pieces (hi = rne16(x), lo' = rne16((x - hi) 2^11)) and THREE MFMA
//     products per k-step (hi hi into the main accumulator; hi lo' + lo' hi into a cross accumulator that is folded
//     in with 2^-11 behind the k-loop): 22 significand bits per operand, against an fp64 product 5e-8 of sum |x||w|
//     (a plain fp32 fma chain: 2e-7; the six-product bf16 split of the other GEMM kernels: 2e-8) -- common.h.  Until
//     round 3's last version this kernel used the bf16 split too: 21 KB slices, 42 MFMAs per P1 slice, 144 us per
//     launch at the 10k-grain graph against 122 us now.
//   * the tile's input rows [h | x | 1 | 0] stay in LDS for the whole tile (B operand of P1 and P4 of every
//     gate): as registers they had to be reloaded behind every sweep, in front of P3's first slice.
// A tile's CSR window (17 row pointers + up to 111 source indices per edge type) is fetched once into LDS
// and reused by the four gate passes; h_src rows are gathered once per gate (they stay in L2), V rows once.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "cell_common.h"
#define GGNN_STAMP_SUFFIX _dec
#include "stamps.h"

namespace ggnn {

constexpr int DC_WAVES = 8;                        // one workgroup of 128 nodes per compute unit, two waves per SIMD
constexpr int DC_MAX_PROBLEMS = 4;
constexpr int DC_SLICE = GGNN_DC_SLICE_BYTES;       // 14 pieces of 1 KB
static_assert(DC_SLICE == 7 * DC_PL * 1024, "slice = 7 column tiles x planes x 1 KB");
constexpr int DC_NP1 = 7 * DC_PL, DC_NP3 = 6 * DC_PL;   // pieces of a P1 slice / of a P3 or P4 slice
constexpr int DC_S = 116;                           // stage row stride in floats (52 mod 64 banks: rows spread)
constexpr int DC_STAGE = 16 * DC_S * 4;             // 7 424 B
constexpr int DC_XF = 16 * DC_S * 4;                // the tile's input rows [h 96 | x F | 1 | 0 ..] as B-fragment planes (7 KB used): B operand of P1 / P4
constexpr int DC_CW = 111;                          // source indices of a tile kept in LDS per edge type
constexpr int DC_CSR = (17 + DC_CW) * 4;            // 512 B
constexpr int DC_WAVE_LDS = DC_STAGE + DC_XF + 2 * DC_CSR;
constexpr int DC_LDS = 2 * DC_SLICE + DC_WAVES * DC_WAVE_LDS;   // 155 648 B
static_assert(DC_LDS <= 160 * 1024, "LDS");

struct DecCellBatch {
  ggnn_dec_cell_args a[DC_MAX_PROBLEMS];
  int wg_off[DC_MAX_PROBLEMS + 1];
  int n;
};

__device__ __forceinline__ void dec_cell_body(const ggnn_dec_cell_args& A, const int tileset,
                                              unsigned char* __restrict__ smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;  // matrix view: node lr of the tile, k-group / output rows 4 kq ..
  const int ch = 3 * lr;                     // sweep view: lane lr of DPP row kq owns channels ch..ch+2, 48+ch..
  constexpr int CH2 = C / 2;

  unsigned char* __restrict__ wbase = smem + 2 * DC_SLICE + wave * DC_WAVE_LDS;
  float* __restrict__ stage = reinterpret_cast<float*>(wbase);
  float* __restrict__ xf = reinterpret_cast<float*>(wbase + DC_STAGE);
  int* __restrict__ csr = reinterpret_cast<int*>(wbase + DC_STAGE + DC_XF);   // [e][17 + DC_CW]

  const int n_dst = (int)A.n_dst, n_in = A.n_in, F = A.f_dst;
  // a ragged last tile slides back over rows the previous tile also produces (identical duplicate results);
  // tiles past the end (a workgroup's surplus waves) repeat the last one: every wave runs the whole program,
  // so the workgroup barriers of the slice stream need no special case
  const int row0 = max(0, min((tileset * DC_WAVES + wave) * 16, n_dst - 16));
  const int node_m = min(row0 + lr, n_dst - 1);    // this lane's node in the matrix view (n_dst < 16: clamped)

  // ---- the weight stream: slice s -> buffer s & 1, fetched one slice ahead by all the workgroup's waves ----
  // LDS-DMA, a wave's share of the next slice (the slice's 14 or 12 one-KB pieces dealt round-robin) requested at the top
  // of a k-step; one counted wait + one workgroup barrier per slice.  The ISSUE of a piece costs the wave ~150
  // cycles (in-kernel stamps: 0.38 us per slice with four waves sharing a slice = 27 of a tile's 140 us), which is
  // why the workgroup has eight waves: half the pieces per wave and slice.  (Tried: 16-byte loads to registers +
  // ds_write_b128 at the end of the k-step -- the loads' latency then sits in front of the barrier: slower.)
  const unsigned char* __restrict__ wsrc = reinterpret_cast<const unsigned char*>(A.wstream) + lane * 16;
  const uint32_t slice_lds =
      __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)));
  int s_cur = 0;
  [[maybe_unused]] unsigned long long st_wait = 0, st_p1 = 0, st_p2 = 0, st_p3 = 0, st_p4 = 0, st_lstm = 0;
  [[maybe_unused]] unsigned long long st_dma = 0, st_t0 = 0;
  GGNN_STAMP(0);
  auto dma_slice = [&](int s, int np) {
    const unsigned char* src = wsrc + (size_t)s * DC_SLICE;
    const uint32_t dst = slice_lds + (s & 1) * DC_SLICE;
    for (int p = wave; p < np; p += DC_WAVES) dc_dma16(src + p * 1024, dst + p * 1024);
  };
  // `np_next`: pieces of the slice after the current one (14: a P1 slice, 12: P3 / P4, 0: none)
#ifdef DC_EXP_DMAMID
  // development (profiles/r6_dec_cell_experiments.txt): the wave's share of the next slice requested BETWEEN two column
  // tiles' MFMAs of the k-step (dc_kstep's `mid`) instead of at its top, in front of the first fragment read
  int np_pending = 0;
  auto begin_slice = [&](int np_next) -> const u32x4* {
    np_pending = np_next;
    return reinterpret_cast<const u32x4*>(smem + (s_cur & 1) * DC_SLICE) + lane;
  };
  auto dma_mid = [&]() __attribute__((always_inline)) {
    if (np_pending > 0) dma_slice(s_cur + 1, np_pending);
  };
#define DC_MID , dma_mid
#else
  auto begin_slice = [&](int np_next) -> const u32x4* {   // the slice about to be used landed at the previous end_slice
    st_t0 = GGNN_STAMP_NOW();
    if (np_next > 0) dma_slice(s_cur + 1, np_next);
    st_dma += GGNN_STAMP_NOW() - st_t0;
    return reinterpret_cast<const u32x4*>(smem + (s_cur & 1) * DC_SLICE) + lane;
  };
#define DC_MID
#endif
  auto end_slice = [&]() {
    [[maybe_unused]] const unsigned long long w0 = GGNN_STAMP_NOW();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next slice are in LDS
    __syncthreads();                                   // ... everybody's are, and nobody reads the old one any more
    st_wait += GGNN_STAMP_NOW() - w0;
    ++s_cur;
  };
  dma_slice(0, DC_NP1);
#ifdef DC_EXP_PRIO
  // development: the second-dispatched half of the workgroup loses every arbitration against its SIMD partner
  // (MI355X_MICROARCH.md, two waves per SIMD, item 4): one static priority for waves 4-7
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  // Range flag (ggnn.h, OPERAND RANGE): the operands of the two-piece fp16 split are checked where they are made --
  // the tile's input rows here in the prologue, the aggregates when a row is closed -- and reported at once: no
  // state is carried (the kernel has no register to spare).
  auto report_range = [&](bool bad) __attribute__((always_inline)) {
    if (A.flags != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(A.flags, GGNN_FLAG_F16_RANGE);
  };

  // ---- tile prologue: the tile's input rows [h | x | 1 | 0] -> LDS AS THE TWO fp16 PLANES of their B fragments
  // (read by P1 and P4 of every gate: 12 x 4 k-steps per tile; split once here instead of at every read -- round 4:
  // the splits were ~100 vector instructions per k-step, 9 us of a tile's 120).  A lane (node lr, k-group kq) only ever
  // reads the 16-byte slots it writes: [k-step 0..2][plane][lane] and, for the 16 feature slots, [plane][lanes of
  // k-groups 0 and 1] = 7 KB of the wave's DC_XF bytes.  CSR windows -> LDS. ----
  unsigned char* __restrict__ xpl = reinterpret_cast<unsigned char*>(xf) + lane * 16;
  auto x_planes = [&](int ks, u32x4 (&out)[DC_PL]) __attribute__((always_inline)) {
    if (ks < 3) {
#pragma unroll
      for (int p = 0; p < DC_PL; ++p) out[p] = *reinterpret_cast<const u32x4*>(xpl + (ks * DC_PL + p) * 1024);
    } else {   // the feature slots: k-groups 0 and 1 (512 B per plane), zeros behind
#pragma unroll
      for (int p = 0; p < DC_PL; ++p) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(xpl + 3 * DC_PL * 1024 + p * 512 - (kq >= 2 ? 512 : 0));
        out[p] = kq < 2 ? v : (u32x4){0u, 0u, 0u, 0u};
      }
    }
  };
  {
    uint32_t in_max = 0u;   // (dc_track: as bits, so that a NaN or an inf in the tile's rows is reported)
    const float* hrow = A.h_dst + (int64_t)min(row0 + lr, n_dst - 1) * A.ldh + 8 * kq;
    f32x4 hv[3][2];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      hv[ks][0] = *reinterpret_cast<const f32x4*>(hrow + 32 * ks);
      hv[ks][1] = *reinterpret_cast<const f32x4*>(hrow + 32 * ks + 4);
    }
    // features: slots 8 (kq & 1) ..+7 of [x_0 .. x_{F-1}, 1 (bias), 0 ..]
    const float* xrow = A.x_dst + (int64_t)min(row0 + lr, n_dst - 1) * A.ldx;
    f32x4 xv[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sl = 8 * (kq & 1) + j;
      const float v = xrow[min(sl, F - 1)];   // unconditional (clamped) load
      xv[j >> 2][j & 3] = sl < F ? v : (sl == F ? 1.0f : 0.0f);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 r0 = ks < 3 ? hv[ks][0] : xv[0], r1 = ks < 3 ? hv[ks][1] : xv[1];
      u32x4 pl[DC_PL];
      dc_split(r0, r1, pl, in_max);
      if (ks < 3) {
#pragma unroll
        for (int p = 0; p < DC_PL; ++p) *reinterpret_cast<u32x4*>(xpl + (ks * DC_PL + p) * 1024) = pl[p];
      } else if (kq < 2) {
#pragma unroll
        for (int p = 0; p < DC_PL; ++p) *reinterpret_cast<u32x4*>(xpl + 3 * DC_PL * 1024 + p * 512) = pl[p];
      }
    }
    report_range(in_max >= DC_RANGE_LIMIT);
    for (int e = 0; e < n_in; ++e) {
      const ggnn_dec_cell_sweep& Sw = A.in[e];
      int* __restrict__ rp = csr + e * (17 + DC_CW);
      if (lane < 17) rp[lane] = Sw.rowptr[min(row0 + lane, n_dst)];
      __builtin_amdgcn_wave_barrier();
      const int pbase = rp[0], e_last = (int)Sw.E - 1;
      if (Sw.E > 0) {
        for (int k = lane; k < DC_CW; k += 64) rp[17 + k] = Sw.col[min(pbase + k, e_last)];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // slice 0 is in LDS
  GGNN_STAMP(1);

  f32x4 run[6];   // the LSTM update as the gates arrive: sig(i) -> sig(i) tanh(c~) -> c' -> (h')
#pragma unroll
  for (int ct = 0; ct < 6; ++ct) run[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int gi = 0; gi < 4; ++gi) {
    const int g = gi == 1 ? 2 : (gi == 2 ? 1 : gi);   // weights are indexed i, f, c, o; processed i, c~, f, o
    f32x4 pre[6], cin[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) pre[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the edge types are walked forwards for the first and third gate of the stream and backwards for the second and
    // fourth: the hidden rows and edge records a pass gathered are gathered again by the very next pass (the same edge
    // type, the next gate) while the XCD's L2 still holds them -- in a fixed order every re-gather came two passes later
    for (int ei = 0; ei < n_in; ++ei) {
      const int e = (gi & 1) ? n_in - 1 - ei : ei;
      const ggnn_dec_cell_sweep& Sw = A.in[e];
      // ================= P1: u_h | u4 of the tile's 16 nodes for (e, g) =================
      [[maybe_unused]] const unsigned long long t_a = GGNN_STAMP_NOW();
      {
        DcAcc u[7];
#pragma unroll
        for (int nb = 0; nb < 7; ++nb) u[nb].zero();
        u32x4 xb[2][DC_PL];
        x_planes(0, xb[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4* pw = begin_slice(ks < 3 ? DC_NP1 : DC_NP3);   // behind P1: the sweep, then P3's first slice
          dc_kstep<7>(pw, xb[ks & 1], u DC_MID);
          if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);
          end_slice();
        }
        // D layout -> stage[node][column]
#pragma unroll
        for (int nb = 0; nb < 7; ++nb) *reinterpret_cast<f32x4*>(&stage[lr * DC_S + 16 * nb + 4 * kq]) = u[nb].value();
      }
      __builtin_amdgcn_wave_barrier();
      [[maybe_unused]] const unsigned long long t_b = GGNN_STAMP_NOW();

      // ================= P2: the sweep of (e, g) over the tile's rows, one node per 16-lane row =================
      // Two rows per 16-lane group in flight (nodes 8 half + kq and 8 half + 4 + kq): the gathers of both are issued
      // back to back before either is folded, so a (gate, edge type) pass exposes two memory round trips, not four.
      {
        const float* __restrict__ ep = Sw.edge_params + g * GGNN_EDGE_PARAM_ROWS * C;
        f3 wv[6];
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          const float* w = ep + ch + (cc < 3 ? cc : CH2 + cc - 3);
          wv[cc] = {w[0], w[C], w[2 * C]};
        }
        // value rows of (edge type, gate): 96 columns of a [n_src, ldv] row, or (v_block_major) a [n_src, 96] matrix of their own
        const float* __restrict__ vbase = Sw.v_block_major ? Sw.v_src + (int64_t)(Sw.v_off / C + g) * Sw.n_src * C + ch
                                                           : Sw.v_src + Sw.v_off + g * C + ch;
        const float* __restrict__ hbase = Sw.h_src + ch;
        const float* __restrict__ einfo = Sw.einfo;
        const uint32_t ldv = Sw.v_block_major ? (uint32_t)C : (uint32_t)Sw.ldv, ldh = (uint32_t)Sw.ldh_src;
        const int* __restrict__ rp = csr + e * (17 + DC_CW);
        const int* __restrict__ colw = rp + 17;
        const int pbase = rp[0], e_last = max((int)Sw.E - 1, 0);
        const bool has_edges = Sw.E > 0;
        struct Row {      // one destination row being folded
          f3 uh0, uh1;
          float u4, mx, den, sae, acc[6];
          int p, pe;
        };
        struct Unit {     // the gathered operands of <= 3 of its in-edges
          f3 hh[GGNN_UNIT_EDGES][2], vv[GGNN_UNIT_EDGES][2];
          float x4[GGNN_UNIT_EDGES];
          // (reloc_e = slots 0..2 of the edge record, i.e. x4 of the row's lanes 0..2: broadcast at fold time, no load
          // and no register of its own; the edge length a_e is slot 13: lane 13 sums alpha a_e)
        };
        auto open_row = [&](Row& r, int n) __attribute__((always_inline)) {
          const float* __restrict__ su = stage + n * DC_S;
          r.uh0 = {su[ch], su[ch + 1], su[ch + 2]};
          r.uh1 = {su[CH2 + ch], su[CH2 + ch + 1], su[CH2 + ch + 2]};
          r.u4 = su[C + lr];
          const int nl = min(row0 + n, n_dst - 1) - row0;   // (n_dst < 16: rows past the end repeat the last node)
          r.p = rp[nl];
          r.pe = rp[nl + 1];
          r.mx = -INFINITY;
          r.den = r.sae = 0.f;
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) r.acc[cc] = 0.f;
        };
        // A tile whose in-edges fit the LDS index window (all but hub tiles) finds every source index there; a load
        // under `if` would drag a full wait to the branch merge and serialise the edges, so the choice is made once
        // per tile, wave-uniformly, between two straight-line variants of the gather.
        // (two entries of the window stay spare: a row's three consecutive indices are read from ONE clamped position)
        const bool in_window = __builtin_amdgcn_readfirstlane(rp[16] - pbase) <= DC_CW - 2;
        // the source nodes of a row's next three in-edges: ALL index reads of the rows in flight go out before the first
        // address is formed (interleaved with the gathers -- rounds 3-5 -- every edge paid its own LDS round trip in front
        // of its loads: six exposed ds_read latencies per pair of rows)
        auto sources = [&](const Row& r, int (&j)[GGNN_UNIT_EDGES], auto window_tag) __attribute__((always_inline)) {
          constexpr bool WINDOW = decltype(window_tag)::value;
          if constexpr (WINDOW) {
            // three consecutive window entries from one address (an active edge's entry is col[p + t]; behind the row's
            // end they belong to the next rows or repeat the list's last entry: valid nodes, weighted with exact zeros)
            const int* __restrict__ w = colw + min(max(r.p - pbase, 0), DC_CW - GGNN_UNIT_EDGES);
#pragma unroll
            for (int t = 0; t < GGNN_UNIT_EDGES; ++t) j[t] = has_edges ? w[t] : 0;
          } else {
#pragma unroll
            for (int t = 0; t < GGNN_UNIT_EDGES; ++t) j[t] = has_edges ? Sw.col[min(r.p + t, e_last)] : 0;
          }
        };
        auto gather = [&](const Row& r, Unit& U, const int (&j)[GGNN_UNIT_EDGES]) __attribute__((always_inline)) {   // unconditional (clamped) loads, back to back
#pragma unroll
          for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
            const int pt = min(r.p + t, e_last);
            U.hh[t][0] = ld3(hbase + (uint32_t)j[t] * ldh);
            U.hh[t][1] = ld3(hbase + (uint32_t)j[t] * ldh + CH2);
            U.x4[t] = einfo[(uint32_t)pt * GGNN_EINFO_ROW + lr];
            U.vv[t][0] = ld3(vbase + (uint32_t)j[t] * ldv);
            U.vv[t][1] = ld3(vbase + (uint32_t)j[t] * ldv + CH2);
          }
        };
        // Every product-sum is an explicit fma and contraction is off: the two variants of the sweep (and a row computed
        // by two overlapping tiles of a ragged end) must give the same bits, whatever the compiler would have chosen to
        // fuse in each inlined copy.
        // NO BRANCH (round 6): every gathered operand is consumed UNCONDITIONALLY (an edge beyond the row's end holds the
        // clamped load of a real edge: finite, weighted with an exact zero), inactive edges and exhausted rows are selects.
        // With the uses under `if (t < nact)` -- rounds 3-5 -- the compiler SANK gathers into those blocks: the ISA had the
        // hidden-row loads of a row's second edge behind the branch with an s_waitcnt vmcnt(0) of their own, a third
        // dependent memory round trip per pair of rows in every pass.  Same operations on the same operands for the active
        // edges, exact zeros added for the others: bit-identical results (tools/probes/dccheck.py), classifier launch
        // 128 -> 118 us in tools/dcbench.py, 3 160-3 190 -> 3 290-3 360 steps/s on one box.
        auto fold = [&](Row& r, const Unit& U) __attribute__((always_inline)) {
#pragma clang fp contract(off)
          const int nact = min(max(r.pe - r.p, 0), GGNN_UNIT_EDGES);
          float s[GGNN_UNIT_EDGES];
          float mnew = r.mx;
#pragma unroll
          for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
            float part = r.u4 * U.x4[t];
            part = __builtin_fmaf(r.uh0.x, U.hh[t][0].x, part);
            part = __builtin_fmaf(r.uh0.y, U.hh[t][0].y, part);
            part = __builtin_fmaf(r.uh0.z, U.hh[t][0].z, part);
            part = __builtin_fmaf(r.uh1.x, U.hh[t][1].x, part);
            part = __builtin_fmaf(r.uh1.y, U.hh[t][1].y, part);
            part = __builtin_fmaf(r.uh1.z, U.hh[t][1].z, part);
            const float st = row_sum(part);   // 1 / sqrt(96) is folded into u
            s[t] = t < nact ? st : -INFINITY;
            mnew = fmaxf(mnew, s[t]);
          }
          const float scale = nact > 0 ? __expf(r.mx - mnew) : 1.0f;   // exp(-inf) = 0 on a row's first unit
          r.den = r.den * scale;
          r.sae = r.sae * scale;
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) r.acc[cc] = r.acc[cc] * scale;
#pragma unroll
          for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
            const int xi = __builtin_bit_cast(int, U.x4[t]);
            // lane k of every 16-lane row -> the whole row: DPP row_newbcast (a vector-ALU move that the consumer can carry
            // as an operand modifier; rounds 3-5 used ds_swizzle: an LDS-pipe operation with a wait of its own)
            const float rx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x150, 0xF, 0xF, true));
            const float ry = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x151, 0xF, 0xF, true));
            const float rz = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x152, 0xF, 0xF, true));
            const float pw_ = t < nact ? __expf(s[t] - mnew) : 0.f;
            r.den = r.den + pw_;
            r.sae = __builtin_fmaf(pw_, U.x4[t], r.sae);
            const float v[6] = {U.vv[t][0].x, U.vv[t][0].y, U.vv[t][0].z, U.vv[t][1].x, U.vv[t][1].y, U.vv[t][1].z};
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {
              const float val = __builtin_fmaf(wv[cc].z, rz, __builtin_fmaf(wv[cc].y, ry, __builtin_fmaf(wv[cc].x, rx, v[cc])));
              r.acc[cc] = __builtin_fmaf(pw_, fmaxf(val, 0.f), r.acc[cc]);
            }
          }
          r.mx = mnew;
          r.p += GGNN_UNIT_EDGES;
        };
        auto close_row = [&](const Row& r, int n) __attribute__((always_inline)) {  // the row's aggregate over the u it was computed from
#pragma clang fp contract(off)
          const float inv = 1.0f / (r.den + 1e-16f);   // PyG softmax denominator
          float* __restrict__ so = stage + n * DC_S;
          float o[6];
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) o[cc] = r.acc[cc] * inv;
          so[ch] = o[0];
          so[ch + 1] = o[1];
          so[ch + 2] = o[2];
          so[CH2 + ch] = o[3];
          so[CH2 + ch + 1] = o[4];
          so[CH2 + ch + 2] = o[5];
          if (lr == 0) so[C] = r.den * inv;
          if (lr == 13) so[C + 1] = r.sae * inv;
          uint32_t amx = 0u;   // (aggregates are sums of relu outputs; a NaN among the gathered operands ends up here)
          dc_track(amx, o[0], o[1]);
          dc_track(amx, o[2], o[3]);
          dc_track(amx, o[4], o[5]);
          report_range(amx >= DC_RANGE_LIMIT);
        };
        auto sweep = [&](auto window_tag) __attribute__((always_inline)) {
#pragma unroll 1
          for (int half = 0; half < 2; ++half) {
            const int na = 8 * half + kq, nb = na + 4;     // tile rows of this DPP row
            Row ra, rb;
            open_row(ra, na);
            open_row(rb, nb);
            do {
              Unit ua, ub;
              int ja[GGNN_UNIT_EDGES], jb[GGNN_UNIT_EDGES];
              sources(ra, ja, window_tag);
              sources(rb, jb, window_tag);
              gather(ra, ua, ja);
              gather(rb, ub, jb);
              fold(ra, ua);
              fold(rb, ub);
            } while (__builtin_amdgcn_ballot_w64(ra.p < ra.pe || rb.p < rb.pe) != 0);
            close_row(ra, na);
            close_row(rb, nb);
          }
        };
        if (in_window) sweep(std::true_type{});
        else sweep(std::false_type{});
      }
      __builtin_amdgcn_wave_barrier();
      [[maybe_unused]] const unsigned long long t_c = GGNN_STAMP_NOW();
      // ================= P3: pre += lin_l2(e, g) . agg + (b_l2, w_edge) . (sum alpha, sum alpha a) =================
      {
        u32x4 xb[2][DC_PL];
        auto a_planes = [&](int ks, u32x4 (&out)[DC_PL]) __attribute__((always_inline)) {
          const float* sr = &stage[lr * DC_S + 32 * ks + 8 * kq];
          dc_split(*reinterpret_cast<const f32x4*>(sr), *reinterpret_cast<const f32x4*>(sr + 4), out);
        };
        a_planes(0, xb[0]);
        DcAcc part[6];   // this phase's contribution (the fp16 split's cross terms live only here)
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) part[ct].zero();
        // the exact fp32 tail's weight fragments: requested here, used behind the three slices (requested there,
        // their round trip stood in front of the tail's MFMAs in every phase)
        float wtail[6];
        {
          const float* __restrict__ wt = A.w2_tail + (size_t)((g * n_in + e) * 6) * 64 + lane;
#pragma unroll
          for (int ct = 0; ct < 6; ++ct) wtail[ct] = wt[ct * 64];
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const u32x4* pw = begin_slice(ks < 2 ? DC_NP3 : (ei + 1 < n_in ? DC_NP1 : DC_NP3));   // next: P3, the next edge type's P1, or P4
          dc_kstep<6>(pw, xb[ks & 1], part DC_MID);
          if (ks + 1 < 3) a_planes(ks + 1, xb[(ks + 1) & 1]);
          end_slice();
        }
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] += part[ct].value();
        const float xt = kq < 2 ? stage[lr * DC_S + C + kq] : 0.f;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wtail[ct], xt, pre[ct], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
      [[maybe_unused]] const unsigned long long t_d = GGNN_STAMP_NOW();
      st_p1 += t_b - t_a;
      st_p2 += t_c - t_b;
      st_p3 += t_d - t_c;
    }
    [[maybe_unused]] const unsigned long long t_e = GGNN_STAMP_NOW();

    // the old cell state (only the forget gate uses it): in flight during P4, live nowhere else
    if (gi == 2) {
      const float* crow = A.c_in + (int64_t)node_m * C + 4 * kq;
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) cin[ct] = *reinterpret_cast<const f32x4*>(crow + 16 * ct);
    }
    // ================= P4: the summed skip term + gate bias of gate g =================
    {
      u32x4 xb[2][DC_PL];
      x_planes(0, xb[0]);
      DcAcc part[6];
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) part[ct].zero();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4* pw = begin_slice(ks < 3 ? DC_NP3 : (gi < 3 ? DC_NP1 : 0));   // next: P4, the next gate's P1, or nothing
        dc_kstep<6>(pw, xb[ks & 1], part DC_MID);
        if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);
        end_slice();
      }
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) pre[ct] += part[ct].value();
    }

    [[maybe_unused]] const unsigned long long t_f = GGNN_STAMP_NOW();
    st_p4 += t_f - t_e;
    // ================= LSTM update, folded in gate by gate (heteropgclstm.py:140-146) =================
    // (the gate loop is a real loop -- unrolled four times the register allocator gave up --: the four updates sit
    // behind wave-uniform branches)
    f32x4 (&pv)[6] = pre;
    if (gi == 0) {
#pragma unroll
      for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) run[ct][r] = sigmoidf_(pv[ct][r]);
    } else if (gi == 1) {
#pragma unroll
      for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) run[ct][r] *= tanhf_(pv[ct][r]);
    } else if (gi == 2) {
#pragma unroll
      for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) run[ct][r] = sigmoidf_(pv[ct][r]) * cin[ct][r] + run[ct][r];
    } else {
#pragma unroll
      for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[ct][r] = sigmoidf_(pv[ct][r]) * tanhf_(run[ct][r]);
    }
    if (gi == 2) {
      float* crow = A.c_out + (int64_t)node_m * C + 4 * kq;
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) *reinterpret_cast<f32x4*>(crow + 16 * ct) = run[ct];
    }
    if (gi == 3) {
      float* hrow = A.h_out + (int64_t)node_m * C + 4 * kq;
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) *reinterpret_cast<f32x4*>(hrow + 16 * ct) = pv[ct];
    }
    st_lstm += GGNN_STAMP_NOW() - t_f;
  }
  GGNN_STAMP_VAL(4, st_wait);
  GGNN_STAMP_VAL(5, st_p1);
  GGNN_STAMP_VAL(6, st_p2);
  GGNN_STAMP_VAL(7, st_p3);
  GGNN_STAMP_VAL(8, st_p4);
  GGNN_STAMP_VAL(9, st_lstm);
  GGNN_STAMP_VAL(10, n_in);
  GGNN_STAMP_VAL(11, st_dma);
  GGNN_STAMP(16);
}

// (155 648 B of LDS: one workgroup per compute unit, two waves per SIMD, <= 256 registers)
__global__ __launch_bounds__(DC_WAVES * 64, 2) void dec_cell_kernel(const DecCellBatch B) {
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[DC_LDS];
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const int nwg = B.wg_off[k + 1] - B.wg_off[k];
  // workgroups that share an XCD take one contiguous range of tile sets: neighbouring rows, whose in-edges
  // come from the same source rows, meet in the same L2 (speed only)
  const int ts = xcd_remap((int)blockIdx.x - B.wg_off[k], nwg);
  dec_cell_body(B.a[k], ts, s_raw);
}

}  // namespace ggnn

extern "C" int ggnn_decoder_cell_batch(const ggnn_dec_cell_args* args, int n_problems, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_problems < 1 || n_problems > DC_MAX_PROBLEMS) return GGNN_EINVAL;
  DecCellBatch B;
  B.n = n_problems;
  B.wg_off[0] = 0;
  for (int k = 0; k < DC_MAX_PROBLEMS; ++k) {
    B.a[k] = args[k < n_problems ? k : 0];
    if (k >= n_problems) {
      B.wg_off[k + 1] = B.wg_off[k];
      continue;
    }
    const ggnn_dec_cell_args& A = B.a[k];
    if (A.n_in < 1 || A.n_in > 2 || A.n_dst <= 0 || A.f_dst < 1 || A.f_dst > 12 || A.ldx < A.f_dst) return GGNN_EINVAL;
    if (!A.x_dst || !A.h_dst || !A.c_in || !A.h_out || !A.c_out || !A.wstream || !A.w2_tail) return GGNN_EINVAL;
    if (A.ldh < C || (A.ldh & 3) || !aligned16(A.h_dst) || !aligned16(A.c_in) || !aligned16(A.h_out) ||
        !aligned16(A.c_out) || !aligned16(A.wstream))
      return GGNN_EINVAL;
    if (A.n_dst >= INT32_MAX - 64) return GGNN_EINVAL;
    for (int e = 0; e < A.n_in; ++e) {
      const ggnn_dec_cell_sweep& Sw = A.in[e];
      if (!Sw.rowptr || !Sw.einfo || !Sw.h_src || !Sw.v_src || !Sw.edge_params || !aligned16(Sw.einfo)) return GGNN_EINVAL;
      if (Sw.E < 0 || Sw.n_src <= 0 || (Sw.E > 0 && !Sw.col)) return GGNN_EINVAL;
      if (Sw.ldh_src < C || Sw.v_off < 0) return GGNN_EINVAL;
      if (Sw.v_block_major ? (Sw.v_block_major != 1 || Sw.v_off % C != 0) : Sw.v_off + 4 * C > Sw.ldv) return GGNN_EINVAL;
      if (Sw.n_src * Sw.ldh_src >= INT32_MAX || Sw.n_src * (Sw.v_block_major ? (int64_t)C : Sw.ldv) >= INT32_MAX ||
          (Sw.E + GGNN_UNIT_EDGES) * GGNN_EINFO_ROW >= INT32_MAX)
        return GGNN_EINVAL;  // gathered rows are addressed with 32-bit offsets
    }
    const int64_t n_ts = (A.n_dst + 16 * DC_WAVES - 1) / (16 * DC_WAVES);
    if (B.wg_off[k] + n_ts >= INT32_MAX) return GGNN_EINVAL;
    B.wg_off[k + 1] = B.wg_off[k] + (int)n_ts;
  }
  hipLaunchKernelGGL(dec_cell_kernel, dim3((unsigned)B.wg_off[DC_MAX_PROBLEMS]), dim3(DC_WAVES * 64), 0,
                     (hipStream_t)stream, B);
  return launch_status();
}
