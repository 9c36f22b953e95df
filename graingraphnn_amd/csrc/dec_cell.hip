// Decoder HeteroPGCLSTM cell with everything that belongs to a destination node in ONE kernel
// (ggnn_decoder_cell_batch, include/ggnn.h): the destination-side projections (u_h | u4 per edge type and
// gate, the summed skip term), the periodic-boundary GAT sweep (PeriodConv.message, periodGATconv.py:204-236,
// + propagate's gather / scatter-add), lin_l2 + the value-side lin_edge term, HeteroConv's sum over the edge
// types and the LSTM update (heteropgclstm.py:111-146).
//
// Round 4: the tile never leaves the matrix layouts.  Timing ablations of the round-3 kernel (sweep / GEMMs /
// weight stream / barriers compiled out one at a time, profiles/r4_dec_cell_ablations.txt) showed a wave's life to
// be ONE dependency chain that nothing on the chip contends with -- two workgroups per compute unit in antiphase
// ran no faster than in phase -- and the sweep's share of it (52 of 145 us) to be its sixteen exposed gather round
// trips: the 16-lanes-per-row sweep needed u and the aggregates in another layout than the GEMMs (an LDS stage
// each way) and 78 landing registers per half tile, so its gathers could only be issued behind P1 and in two
// halves.  Now a lane (node lr = l & 15, k-group kq = l >> 4) owns 24 channels of ONE node throughout:
//   * P1 leaves u[node][column 16 nb + 4 kq + i] in lane (node, kq): with the rows of the score weights permuted
//     on the host that IS an A fragment of u (k = hidden channel 32 ks + 8 kq + j), so the 48 scores of a tile and
//     edge type are 36 more MFMAs -- D[node][edge], edge tile t = the t-th in-edge of every node, the wanted
//     entries on the diagonal -- against B fragments of the SOURCE rows: hidden states arrive as the two fp16
//     planes of the arithmetic (ggnn_hidden_planes; 2 x 192 B per node = the bytes of the fp32 row), gathered
//     as 16-byte pieces that need no splitting and, not depending on u, issued in front of P1;
//   * the value rows are gathered ONCE per (edge type, gate) for all 16 nodes (16-byte pieces of the lane's
//     channels 16 nb + 4 kq ..+3, straight into the C operand of an exact fp32 MFMA that adds W_value[:, 0:3] . reloc),
//     requested right behind P1's last k-step: their round trip runs beside the score MFMAs and the softmax;
//   * relu, alpha-weighted sum and normalisation happen in that D layout, which -- with the columns of lin_l2
//     permuted on the host -- is the B fragment of P3.  No LDS stage, no CSR window: the tile's LDS is its own
//     input planes (8 KB per wave).
// What stays: one 16-node tile per wave, eight waves per workgroup in step on a double-buffered LDS-DMA stream of
// pre-split fp16 weight slices (72 for a tile with two incoming edge types), gates walked i, c~, f, o with the LSTM
// update folded in as they arrive, two fp16 pieces / three products per fp32 operand (common.h), explicit fmas with
// contraction off so that a row computed by two overlapping tiles of a ragged end gets the same bits.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#define GGNN_STAMP_SUFFIX _dec
#include "stamps.h"

namespace ggnn {

#ifndef DC_WAVES_
#define DC_WAVES_ 8
#endif
constexpr int DC_WAVES = DC_WAVES_;                        // one workgroup of 128 nodes per compute unit, two waves per SIMD
constexpr int DC_MAX_PROBLEMS = 4;
constexpr int DC_PL = 2;                            // weight / operand planes: fp16 hi and scaled residual (common.h)
constexpr int DC_SLICE = GGNN_DC_SLICE_BYTES;       // 14 pieces of 1 KB
static_assert(DC_SLICE == 7 * DC_PL * 1024, "slice = 7 column tiles x planes x 1 KB");
constexpr int DC_NP1 = 7 * DC_PL, DC_NP3 = 6 * DC_PL;   // pieces of a P1 slice / of a P3 or P4 slice
constexpr int DC_XP = 3 * DC_PL * 1024 + DC_PL * 512;             // a tile's input planes: [k-step][plane][lane] 16 B (B operand of P1 / P4)
constexpr int DC_PARK = 6 * 1024;                  // the running LSTM term of the tile, parked while a (edge type, gate) pass needs the registers
constexpr int DC_WAVE_LDS = DC_XP + DC_PARK;
constexpr int DC_NBUF = 3;                          // slice buffers: the stream runs two slices ahead of the one in use
constexpr int DC_LDS = DC_NBUF * DC_SLICE + DC_WAVES * DC_WAVE_LDS;   // 143 360 B
static_assert(DC_LDS <= 160 * 1024, "LDS");
constexpr int DC_HP_ROW = 2 * C * 2;                // bytes of a node's hidden planes: [hi | lo'][96] fp16
constexpr int UE = GGNN_UNIT_EDGES;                 // edge tiles (t-th in-edge of every node) handled together

struct DecCellBatch {
  ggnn_dec_cell_args a[DC_MAX_PROBLEMS];
  int wg_off[DC_MAX_PROBLEMS + 1];
  int n;
  int stagger;   // start delay of workgroup b: ((b >> 3) & 7) * stagger * ~1 us (speed only; see the kernel)
};

// LDS-DMA: every lane copies 16 bytes from its own global address to lds_base + lane * 16 (wave-uniform base
// in M0).  Not tracked by the compiler: completion = s_waitcnt vmcnt (in issue order with every other
// vector-memory operation of the wave).
__device__ __forceinline__ void dc_dma16(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base))
               : "memory");
}

// eight fp32 values (r0 | r1) -> the two fp16 planes of an MFMA operand fragment; `amax` follows the largest
// magnitude that went through a split (range flag, ggnn.h)
__device__ __forceinline__ void dc_split(const f32x4 r0, const f32x4 r1, u32x4 (&xb)[DC_PL], float& amax) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 h = e < 2 ? r0 : r1;
    const float a = h[2 * (e & 1)], b = h[2 * (e & 1) + 1];
    amax = fmaxf(amax, fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)));
    uint32_t q0, q1;
    split_f16x2(a, b, q0, q1);
    xb[0][e] = q0;
    xb[1][e] = q1;
  }
}
__device__ __forceinline__ void dc_split_half(const f32x4 r0, u32x4 (&xb)[DC_PL], float& amax) {   // r1 = 0
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float a = r0[2 * e], b = r0[2 * e + 1];
    amax = fmaxf(amax, fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)));
    uint32_t q0, q1;
    split_f16x2(a, b, q0, q1);
    xb[0][e] = q0;
    xb[1][e] = q1;
  }
  xb[0][2] = xb[0][3] = xb[1][2] = xb[1][3] = 0u;
}

// An accumulator of a 16 x 16 output tile: main + cross terms of the fp16 split.
struct DcAcc {
  f32x4 m, c;
  __device__ __forceinline__ void zero() { m = c = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ f32x4 value() const { return m + c * (1.0f / F16X2_SCALE); }
};
// One k-step of a GEMM phase: acc[nb] += W[nb] . x for the NB column tiles of the slice at `pw` (= slice base +
// lane; piece (nb, plane) at (nb * 2 + plane) * 64).  The weight fragments of tile nb + 1 are read while the
// three MFMAs of tile nb run.
template <int NB>
__device__ __forceinline__ void dc_kstep(const u32x4* __restrict__ pw, const u32x4 (&xb)[DC_PL], DcAcc (&acc)[NB]) {
  u32x4 wf[2][DC_PL];
#pragma unroll
  for (int p = 0; p < DC_PL; ++p) wf[0][p] = pw[p * 64];
  __builtin_amdgcn_sched_group_barrier(0x100, DC_PL, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (nb + 1 < NB) {
#pragma unroll
      for (int p = 0; p < DC_PL; ++p) wf[(nb + 1) & 1][p] = pw[((nb + 1) * DC_PL + p) * 64];
    }
    mfma_x3h(wf[nb & 1], xb, acc[nb].m, acc[nb].c);
    if (nb + 1 < NB) __builtin_amdgcn_sched_group_barrier(0x100, DC_PL, 0);  // DS read
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                        // MFMA
  }
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ f32x4 ld16f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ void dec_cell_body(const ggnn_dec_cell_args& A, const int tileset,
                                              unsigned char* __restrict__ smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;   // node lr of the tile; k-group of a B / A fragment = output rows 4 kq .. of a D tile
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // this lane's 16-byte slot in every 1 KB piece of the wave's input planes
  unsigned char* __restrict__ xpl = smem + DC_NBUF * DC_SLICE + wave * DC_WAVE_LDS + lane * 16;

  const int n_dst = (int)A.n_dst, n_in = A.n_in, F = A.f_dst;
  // a ragged last tile slides back over rows the previous tile also produces (identical duplicate results);
  // tiles past the end (a workgroup's surplus waves) repeat the last one: every wave runs the whole program,
  // so the workgroup barriers of the slice stream need no special case
  const int row0 = max(0, min((tileset * DC_WAVES + wave) * 16, n_dst - 16));
  const int node_m = min(row0 + lr, n_dst - 1);    // this lane's node (n_dst < 16: the last node repeats)

  // ---- the weight stream: slice s -> buffer s & 1, fetched one slice ahead by all the workgroup's waves ----
  // LDS-DMA, a wave's share of the next slice (its 14 or 12 one-KB pieces dealt round-robin) requested at the top
  // of a k-step; one counted wait + one workgroup barrier per slice.
  const unsigned char* __restrict__ wsrc = reinterpret_cast<const unsigned char*>(A.wstream) + lane * 16;
  const uint32_t slice_lds =
      __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)));
  int s_cur = 0;
  [[maybe_unused]] unsigned long long st_wait = 0, st_p1 = 0, st_p2 = 0, st_p3 = 0, st_p4 = 0, st_lstm = 0;
  [[maybe_unused]] unsigned long long st_dma = 0, st_t0 = 0, st_k[4] = {0, 0, 0, 0}, st_split = 0, st_score = 0, st_fold = 0;
  GGNN_STAMP(0);
  // slice s -> buffer s % 3, requested TWO slices ahead: at the top of the k-step on slice s every wave requests its
  // share of slice s + 2 (into the buffer slice s - 1 left at the last barrier); at the end of the k-step it waits
  // for its pieces of slice s + 1 -- requested a whole k-step earlier -- with a COUNTED wait that leaves this k-step's
  // own requests in flight, then the workgroup barrier.
  const int n_slices = 4 * (7 * n_in + 4);
  auto pieces_of = [&](int s) {   // 14: a P1 slice (7 column tiles), 12: P3 / P4, 0: past the end
    const int r = s % (7 * n_in + 4);
    return s >= n_slices ? 0 : ((r < 7 * n_in && (r % 7) < 4) ? DC_NP1 : DC_NP3);
  };
  auto dma_slice = [&](int s) {
    const int np = pieces_of(s);
    const unsigned char* src = wsrc + (size_t)s * DC_SLICE;
    const uint32_t dst = slice_lds + (s % DC_NBUF) * DC_SLICE;
    for (int p = wave; p < np; p += DC_WAVES) dc_dma16(src + p * 1024, dst + p * 1024);
  };
  auto begin_slice = [&]() -> const u32x4* {   // the slice about to be used landed at the previous end_slice
    return reinterpret_cast<const u32x4*>(smem + (s_cur % DC_NBUF) * DC_SLICE) + lane;
  };
  // Issue order inside a k-step: the k-step's own register loads (gathers) FIRST, then request_next().  At its end,
  // `younger` = the number of those register loads: with the <= 2 pieces just requested they may stay in flight.
  auto request_next = [&]() {
    st_t0 = GGNN_STAMP_NOW();
    dma_slice(s_cur + 2);
    st_dma += GGNN_STAMP_NOW() - st_t0;
  };
  auto end_slice = [&](auto younger) {
    [[maybe_unused]] const unsigned long long w0 = GGNN_STAMP_NOW();
    // everything older than (this k-step's register loads + one piece) has landed: in particular this wave's pieces
    // of slice s + 1 (a wave that requested two pieces of slice s + 2 waits for the first of them too: harmless)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(younger)::value + 1) : "memory");
    __syncthreads();                                   // ... everybody's have, and nobody reads slice s any more
    st_wait += GGNN_STAMP_NOW() - w0;
    ++s_cur;
  };
  constexpr std::integral_constant<int, 0> none{};
  dma_slice(0);
  dma_slice(1);

  float amax = 0.f;   // largest magnitude this lane has split into fp16 pieces

  // ---- tile prologue: the tile's input rows [h | x | 1 | 0] as B-fragment planes -> LDS (B operand of P1 and P4 of
  // every gate; a lane only ever reads the slots it wrote); this node's CSR row and first in-edges -> registers ----
  {
    const unsigned char* hp = reinterpret_cast<const unsigned char*>(A.hp_dst) + (int64_t)node_m * DC_HP_ROW + 16 * kq;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
      for (int p = 0; p < DC_PL; ++p)
        *reinterpret_cast<u32x4*>(xpl + (ks * DC_PL + p) * 1024) = ld16(hp + p * (C * 2) + 64 * ks);
    // k-step 3: the 16 slots [x_0 .. x_{F-1}, 1 (bias), 0 ..] in k-groups 0 and 1, zeros behind
    const float* xrow = A.x_dst + (int64_t)node_m * A.ldx;
    f32x4 r[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sl = 8 * (kq & 1) + j;
      const float xv = xrow[min(sl, F - 1)];   // unconditional (clamped) load
      r[j >> 2][j & 3] = kq >= 2 ? 0.f : (sl < F ? xv : (sl == F ? 1.0f : 0.0f));
    }
    u32x4 xb[DC_PL];
    dc_split(r[0], r[1], xb, amax);
    if (kq < 2) {
#pragma unroll
      for (int p = 0; p < DC_PL; ++p) *reinterpret_cast<u32x4*>(xpl + 3 * DC_PL * 1024 + p * 512) = xb[p];
    }
  }
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
  int p_first[2], p_end[2], j_first[2][UE];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    p_first[e] = p_end[e] = 0;
#pragma unroll
    for (int t = 0; t < UE; ++t) j_first[e][t] = 0;
    if (e < n_in) {
      const ggnn_dec_cell_sweep& Sw = A.in[e];
      p_first[e] = Sw.rowptr[node_m];
      p_end[e] = Sw.rowptr[node_m + 1];
      if (Sw.E > 0) {
#pragma unroll
        for (int t = 0; t < UE; ++t) j_first[e][t] = Sw.col[min(p_first[e] + t, (int)Sw.E - 1)];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // slice 0 is in LDS
  GGNN_STAMP(1);

  // the diagonal of a score tile D[node][edge]: node lr's entry sits in lane (lr, kq = lr >> 2), register lr & 3
  const int diag_addr = 4 * (16 * (lr >> 2) + lr), diag_sub = lr & 3;

  // the LSTM update as the gates arrive: sig(i) -> sig(i) tanh(c~) -> c' -> (h'); lives in this lane's LDS slots
  // between the gates (24 registers the gather landing zones need)
  f32x4* __restrict__ park = reinterpret_cast<f32x4*>(xpl + DC_XP);   // + ct * 64
#pragma unroll 1
  for (int gi = 0; gi < 4; ++gi) {
    const int g = gi == 1 ? 2 : (gi == 2 ? 1 : gi);   // weights are indexed i, f, c, o; processed i, c~, f, o
    f32x4 pre[6], cin[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) pre[ct] = zero4;

#pragma unroll 1
    for (int e = 0; e < n_in; ++e) {
      const ggnn_dec_cell_sweep& Sw = A.in[e];
      [[maybe_unused]] const unsigned long long t_a = GGNN_STAMP_NOW();
      const unsigned char* __restrict__ hsrc = reinterpret_cast<const unsigned char*>(Sw.hp_src) + 16 * kq;
      const float* __restrict__ einfo = Sw.einfo;
      const float* __restrict__ vbase = Sw.v_src + Sw.v_off + g * C + 4 * kq;
      const uint32_t ldv = (uint32_t)Sw.ldv;
      const int e_last = max((int)Sw.E - 1, 0);
      const bool has_edges = Sw.E > 0;
      const int pe = e == 0 ? p_end[0] : p_end[1];
      int p = e == 0 ? p_first[0] : p_first[1];

      // The gathered operands of one unit = the t-th in-edges (t = 0..2 from p) of the tile's 16 nodes:
      struct Land {               // ... what the scores need: B fragments of the source's hidden planes, the edge record
        u32x4 hb[UE][3][DC_PL];
        f32x4 x4[UE];             // record slots 4 kq ..+3 (k-step 3 of the score)
        float rel[UE];            // record slot 16 + kq: reloc_x, reloc_y, reloc_z | edge length
      };
      struct VLand {              // ... and the value rows: channels 16 nb + 4 kq ..+3 (C operand of the reloc MFMA)
        f32x4 v[UE][6];
      };
      auto load_scores = [&](int p_, const int (&j)[UE], Land& L, const int t) __attribute__((always_inline)) {   // 8 loads
        const unsigned char* hrow = hsrc + (uint32_t)j[t] * (uint32_t)DC_HP_ROW;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
          for (int pl = 0; pl < DC_PL; ++pl) L.hb[t][ks][pl] = ld16(hrow + pl * (C * 2) + 64 * ks);
        const float* er = einfo + (uint32_t)min(p_ + t, e_last) * GGNN_EINFO_ROW;
        L.x4[t] = ld16f(er + 4 * kq);
        L.rel[t] = er[16 + kq];
      };
      auto load_values = [&](const int (&j)[UE], VLand& V, const int t) __attribute__((always_inline)) {
        const float* vr = vbase + (uint32_t)j[t] * ldv;
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) V.v[t][nb] = ld16f(vr + 16 * nb);
      };

      int j0[UE];
#pragma unroll
      for (int t = 0; t < UE; ++t) j0[t] = e == 0 ? j_first[0][t] : j_first[1][t];
      Land L;
      // ================= P1: u_h | u4 of the tile's 16 nodes for (e, g) =================
      u32x4 ua[4][DC_PL];   // ... as the A fragments of the score MFMAs
      {
        DcAcc u[7];
#pragma unroll
        for (int nb = 0; nb < 7; ++nb) u[nb].zero();
        u32x4 xb[2][DC_PL];
        x_planes(0, xb[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          [[maybe_unused]] const unsigned long long tk0 = GGNN_STAMP_NOW();
          const u32x4* pw = begin_slice();
          // the first unit's score operands do not depend on u: one edge tile's eight loads per k-step, requested
          // BEHIND the k-step's slice pieces -- they stay in flight across its barrier (the address unit of the
          // compute unit takes ~40 cycles per gather instruction: all at once, the eight waves' requests stood in
          // front of the next slice's pieces for 3.6 us)
          if (ks < UE) load_scores(p, j0, L, ks);
          request_next();
          if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);
          dc_kstep<7>(pw, xb[ks & 1], u);
          if (ks < UE) end_slice(std::integral_constant<int, 8>{});
          else end_slice(none);
          st_k[ks] += GGNN_STAMP_NOW() - tk0;
        }
        [[maybe_unused]] const unsigned long long ts0 = GGNN_STAMP_NOW();
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) dc_split(u[2 * ks].value(), u[2 * ks + 1].value(), ua[ks], amax);
        dc_split_half(u[6].value(), ua[3], amax);
        st_split += GGNN_STAMP_NOW() - ts0;
      }
      [[maybe_unused]] const unsigned long long t_b = GGNN_STAMP_NOW();

      // ================= P2: scores, softmax, values -- all in the matrix layouts =================
      // the exact fp32 fragments of this (e, g): W_value[:, 0:3] as A operand (lane (channel 16 nb + lr, k = kq))
      float w3[6];
      {
        const float* __restrict__ ep = Sw.edge_params + g * GGNN_EDGE_PARAM_ROWS * C + min(kq, 2) * C + lr;
#pragma unroll
        for (int nb = 0; nb < 6; ++nb) {
          const float w = ep[16 * nb];
          w3[nb] = kq < 3 ? w : 0.f;
        }
      }
      // the value rows: one round trip for the whole tile, requested edge tile by edge tile as the score MFMAs
      // free the landing registers of the hidden planes
      VLand V;
      float mx = -INFINITY, den = 0.f, sae = 0.f;
      f32x4 acc[6];
#pragma unroll
      for (int nb = 0; nb < 6; ++nb) acc[nb] = zero4;

      auto scores = [&](const Land& L_, float (&s)[UE], const int (&j)[UE], VLand& V_) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < UE; ++t) {
          load_values(j, V_, t);
          asm volatile("" ::: "memory");
          DcAcc S;
          S.zero();
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) mfma_x3h(ua[ks], L_.hb[t][ks], S.m, S.c);
          u32x4 xb4[DC_PL];
          dc_split_half(L_.x4[t], xb4, amax);
          mfma_x3h(ua[3], xb4, S.m, S.c);
          const f32x4 sv = S.value();   // (1 / sqrt(96) is folded into u)
          const float sel = diag_sub == 0 ? sv[0] : (diag_sub == 1 ? sv[1] : (diag_sub == 2 ? sv[2] : sv[3]));
          s[t] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(diag_addr, __builtin_bit_cast(int, sel)));
        }
      };
      // online-max softmax over the units of a row; every product-sum is an explicit fma and contraction is off: a
      // row computed by two overlapping tiles of a ragged end must give the same bits
      auto fold = [&](const float (&s)[UE], const Land& L_, const VLand& V_, const int nact, auto first_tag) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        constexpr bool FIRST = decltype(first_tag)::value;
        float mnew = mx;
#pragma unroll
        for (int t = 0; t < UE; ++t) mnew = t < nact ? fmaxf(mnew, s[t]) : mnew;
        if constexpr (!FIRST) {
          const float scale = nact > 0 ? __expf(mx - mnew) : 1.0f;
          den = den * scale;
          sae = sae * scale;
#pragma unroll
          for (int nb = 0; nb < 6; ++nb) acc[nb] = acc[nb] * scale;
        }
#pragma unroll
        for (int t = 0; t < UE; ++t) {
          const float pw_ = t < nact ? __expf(s[t] - mnew) : 0.f;
          den = den + pw_;
          sae = __builtin_fmaf(pw_, L_.rel[t], sae);   // k-group 3 holds the edge length: sum alpha a_e there
#pragma unroll
          for (int nb = 0; nb < 6; ++nb) {
            const f32x4 val = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[nb], L_.rel[t], V_.v[t][nb], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nb][i] = __builtin_fmaf(pw_, fmaxf(val[i], 0.f), acc[nb][i]);
          }
        }
        mx = mnew;
      };
      {
        float s[UE];
        [[maybe_unused]] const unsigned long long tq0 = GGNN_STAMP_NOW();
        scores(L, s, j0, V);
        [[maybe_unused]] const unsigned long long tq1 = GGNN_STAMP_NOW();
        fold(s, L, V, min(max(pe - p, 0), UE), std::true_type{});
        st_score += tq1 - tq0;
        st_fold += GGNN_STAMP_NOW() - tq1;
      }
      p += UE;
      // rows of more than three in-edges (grains; hubs): further units, their loads exposed
      while (__builtin_amdgcn_ballot_w64(p < pe) != 0) {
        int j[UE];
#pragma unroll
        for (int t = 0; t < UE; ++t) j[t] = has_edges ? Sw.col[min(p + t, e_last)] : 0;
        Land L2;
        VLand V2;
#pragma unroll
        for (int t = 0; t < UE; ++t) load_scores(p, j, L2, t);
        float s[UE];
        scores(L2, s, j, V2);
        fold(s, L2, V2, min(max(pe - p, 0), UE), std::false_type{});
        p += UE;
      }
      [[maybe_unused]] const unsigned long long t_c = GGNN_STAMP_NOW();

      // ================= P3: pre += lin_l2(e, g) . agg + (b_l2, w_edge) . (sum alpha, sum alpha a) =================
      {
        u32x4 ab[3][DC_PL];
        float xt, wtail[6];   // the (b_l2, w_edge) tail of lin_l2: requested here, used behind the three slices
        {
          const float* __restrict__ wt = A.w2_tail + (size_t)((g * n_in + e) * 6) * 64 + lane;
#pragma unroll
          for (int ct = 0; ct < 6; ++ct) wtail[ct] = wt[ct * 64];
        }
        {
#pragma clang fp contract(off)
          const float inv = 1.0f / (den + 1e-16f);   // PyG softmax denominator
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) dc_split(acc[2 * ks] * inv, acc[2 * ks + 1] * inv, ab[ks], amax);
          xt = kq == 0 ? den * inv : (kq == 3 ? sae * inv : 0.f);
        }
        DcAcc part[6];   // this phase's contribution (the fp16 split's cross terms live only here)
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) part[ct].zero();
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const u32x4* pw = begin_slice();
          request_next();
          dc_kstep<6>(pw, ab[ks], part);
          end_slice(none);
        }
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] += part[ct].value();
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wtail[ct], xt, pre[ct], 0, 0, 0);
      }
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
        const u32x4* pw = begin_slice();
        request_next();
        if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);
        dc_kstep<6>(pw, xb[ks & 1], part);
        end_slice(none);
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
    f32x4 run[6];
    if (gi > 0) {
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) run[ct] = park[ct * 64];
    }
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
    if (gi < 3) {
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) park[ct * 64] = run[ct];
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
  // range flag: an operand at or beyond fp16's range was clamped somewhere in this tile
  if (A.flags != nullptr && __builtin_amdgcn_ballot_w64(!(amax < 65504.0f)) != 0 && lane == 0) atomicOr(A.flags, GGNN_FLAG_F16_RANGE);
  GGNN_STAMP_VAL(4, st_wait);
  GGNN_STAMP_VAL(5, st_p1);
  GGNN_STAMP_VAL(6, st_p2);
  GGNN_STAMP_VAL(7, st_p3);
  GGNN_STAMP_VAL(8, st_p4);
  GGNN_STAMP_VAL(9, st_lstm);
  GGNN_STAMP_VAL(10, n_in);
  GGNN_STAMP_VAL(11, st_dma);
  GGNN_STAMP_VAL(12, st_k[0]);
  GGNN_STAMP_VAL(13, st_k[1]);
  GGNN_STAMP_VAL(14, st_k[2]);
  GGNN_STAMP_VAL(15, st_k[3]);
  GGNN_STAMP_VAL(17, st_split);
  GGNN_STAMP_VAL(18, st_score);
  GGNN_STAMP_VAL(19, st_fold);
  GGNN_STAMP(16);
}

__global__ __launch_bounds__(DC_WAVES * 64, DC_WAVES == 4 ? 2 : 1) void dec_cell_kernel(const DecCellBatch B) {
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[DC_LDS];
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const int nwg = B.wg_off[k + 1] - B.wg_off[k];
  // workgroups that share an XCD take one contiguous range of tile sets: neighbouring rows, whose in-edges
  // come from the same source rows, meet in the same L2 (speed only)
  const int ts = xcd_remap((int)blockIdx.x - B.wg_off[k], nwg);
  // Every workgroup runs the same program from the same start: left alone, the whole chip gathers in the same
  // microseconds and sits in its GEMM phases in the same microseconds.  A start delay that differs between the
  // workgroups of an XCD spreads the gather bursts over the period of one (edge type, gate) pass.
  for (int i = ((int)(blockIdx.x >> 3) & 7) * B.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(32);
  dec_cell_body(B.a[k], ts, s_raw);
}

// fp32 rows -> the two fp16 planes of the decoder cell's arithmetic, [n][hi | lo'][96]; one thread per channel pair
__global__ __launch_bounds__(256) void hidden_planes_kernel(const float* __restrict__ h, int64_t n, int64_t ldh,
                                                            uint32_t* __restrict__ out, int32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool over = false;
  if (i < n * (C / 2)) {
    const int64_t node = i / (C / 2);
    const int c2 = (int)(i - node * (C / 2));
    const float a = h[node * ldh + 2 * c2], b = h[node * ldh + 2 * c2 + 1];
    over = !(fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)) < 65504.0f);
    uint32_t hi, lo;
    split_f16x2(a, b, hi, lo);
    out[node * C + c2] = hi;
    out[node * C + C / 2 + c2] = lo;
  }
  if (flags != nullptr && __builtin_amdgcn_ballot_w64(over) != 0 && (threadIdx.x & 63) == 0) atomicOr(flags, GGNN_FLAG_F16_RANGE);
}

}  // namespace ggnn

extern "C" int ggnn_hidden_planes(const float* h, int64_t n, int64_t ldh, void* planes, int32_t* flags,
                                  ggnn_stream_t stream) {
  using namespace ggnn;
  if (!h || !planes || n <= 0 || ldh < C || !aligned16(planes)) return GGNN_EINVAL;
  const int64_t nblk = (n * (C / 2) + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(hidden_planes_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, h, n, ldh,
                     reinterpret_cast<uint32_t*>(planes), flags);
  return launch_status();
}

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
    if (!A.x_dst || !A.hp_dst || !A.c_in || !A.h_out || !A.c_out || !A.wstream || !A.w2_tail) return GGNN_EINVAL;
    if (!aligned16(A.hp_dst) || !aligned16(A.c_in) || !aligned16(A.h_out) || !aligned16(A.c_out) || !aligned16(A.wstream))
      return GGNN_EINVAL;
    if (A.n_dst >= INT32_MAX - 64) return GGNN_EINVAL;
    for (int e = 0; e < A.n_in; ++e) {
      const ggnn_dec_cell_sweep& Sw = A.in[e];
      if (!Sw.rowptr || !Sw.einfo || !Sw.hp_src || !Sw.v_src || !Sw.edge_params) return GGNN_EINVAL;
      if (!aligned16(Sw.einfo) || !aligned16(Sw.hp_src) || !aligned16(Sw.v_src)) return GGNN_EINVAL;
      if (Sw.E < 0 || Sw.n_src <= 0 || (Sw.E > 0 && !Sw.col)) return GGNN_EINVAL;
      if (Sw.v_off < 0 || Sw.v_off + 4 * C > Sw.ldv || (Sw.v_off & 3) || (Sw.ldv & 3)) return GGNN_EINVAL;   // 16-byte pieces
      if (Sw.n_src * DC_HP_ROW >= INT32_MAX || Sw.n_src * Sw.ldv >= INT32_MAX ||
          (Sw.E + GGNN_UNIT_EDGES) * GGNN_EINFO_ROW >= INT32_MAX)
        return GGNN_EINVAL;  // gathered rows are addressed with 32-bit offsets
    }
    const int64_t n_ts = (A.n_dst + 16 * DC_WAVES - 1) / (16 * DC_WAVES);
    if (B.wg_off[k] + n_ts >= INT32_MAX) return GGNN_EINVAL;
    B.wg_off[k + 1] = B.wg_off[k] + (int)n_ts;
  }
  static const int stagger = [] {
    const char* e = getenv("GGNN_DC_STAGGER");
    return e ? atoi(e) : 0;
  }();
  B.stagger = stagger;
  hipLaunchKernelGGL(dec_cell_kernel, dim3((unsigned)B.wg_off[DC_MAX_PROBLEMS]), dim3(DC_WAVES * 64), 0,
                     (hipStream_t)stream, B);
  return launch_status();
}
