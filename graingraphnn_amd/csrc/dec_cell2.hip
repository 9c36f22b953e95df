// Decoder HeteroPGCLSTM cell, SCORE-ONCE variant of dec_cell.hip (ggnn_decoder_cell2_batch, include/ggnn.h): the same
// cell -- destination-side projections, the periodic-boundary GAT sweep (PeriodConv.message, periodGATconv.py:204-236, +
// propagate's gather / scatter-add), lin_l2 + the value-side lin_edge term, HeteroConv's sum over the edge types and the
// LSTM update (heteropgclstm.py:111-146) -- with the sweep of an edge type split in two:
//   phase A (once per edge type): the score operands u_h | u4 of ALL FOUR gates, then ONE pass over the tile's in-edges
//     that gathers a source's hidden row and the edge record once and folds the four gates' attention scores out of
//     them (dec_cell.hip gathers them once per gate).  Raw scores -> a scratch array [edge][4] (written and read by the
//     same wave); a row's softmax statistics (max, 1 / sum, sum alpha, sum alpha a_e per gate) -> 2 KB of LDS per wave.
//   phase B (per gate, per edge type): a pass that gathers only the source's VALUE row, the edge's score and relocation
//     and accumulates alpha relu(v + W_e r) with alpha = exp(s - max) / sum final from the start (no rescaling).
// P1 runs with the tile's rows as the A operand (dc_kstep_xa) so that u leaves the matrix pipe in the sweep's layout --
// a node's 96 channels across the 16 lanes of a DPP row (host permutation packing.DC_P1_ROW) -- and the aggregates return
// to the matrix layout through 24 exact v_mfma_f32_16x16x4_f32 transposes against 0/1 selectors (packing.DC_P3_COL):
// no LDS stage.  Weight stream (same slices, other order): for e: P1(e, i, c~, f, o); for g: for e: P3(e, g); P4(g).
// Arithmetic as dec_cell.hip: the GEMM phases on two fp16 pieces and three MFMA products, the sweep in fp32 with
// explicit fmas.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "cell_common.h"
#define GGNN_STAMP_SUFFIX _dec2
#include "stamps.h"

namespace ggnn {

#ifndef D2_ROWS_A_
#define D2_ROWS_A_ 2   // rows of a 16-lane group whose gathers are in flight together, score pass
#endif
#ifndef D2_ROWS_B_
#define D2_ROWS_B_ 2   // ... value pass (1, 2 or 4)
#endif
constexpr int D2_WAVES = 8;                         // 16-node tiles per workgroup
constexpr int D2_MAX_PROBLEMS = 4;
constexpr int DC_SLICE = GGNN_DC_SLICE_BYTES;       // 14 pieces of 1 KB
static_assert(DC_SLICE == 7 * DC_PL * 1024, "slice = 7 column tiles x planes x 1 KB");
constexpr int DC_NP1 = 7 * DC_PL, DC_NP3 = 6 * DC_PL;   // pieces of a P1 slice / of a P3 or P4 slice
constexpr int D2_XF = 7 * 1024;                     // the tile's input rows as fragment planes
constexpr int D2_CW = 111;                          // source indices of a tile kept in LDS per edge type
constexpr int D2_CSR = (17 + D2_CW) * 4;            // 512 B
constexpr int D2_RSTAT = 2 * 4 * 16 * 4 * 4;        // [e][gate][node][4] floats = 2 KB
constexpr int D2_WAVE_LDS = D2_XF + 2 * D2_CSR + D2_RSTAT;      // 10 KB
constexpr int D2_LDS = 2 * DC_SLICE + D2_WAVES * D2_WAVE_LDS;   // 110 592 B
static_assert(D2_LDS <= 160 * 1024, "LDS");

struct DecCell2Batch {
  ggnn_dec_cell2_args a[D2_MAX_PROBLEMS];
  int wg_off[D2_MAX_PROBLEMS + 1];
  int n;
};

// The k-step of cell_common.h with the operands in the other order: acc[nb] += x . W[nb] -- the tile's rows are the A
// operand and the weight fragments (the same bytes) the B operand, so a lane holds D[row 4 (l >> 4) ..+3][column l & 15]
// of every column tile: a node's columns across the 16 lanes of a DPP row.
template <int NB>
__device__ __forceinline__ void dc_kstep_xa(const u32x4* __restrict__ pw, const u32x4 (&xb)[DC_PL], DcAcc (&acc)[NB]) {
  constexpr int AH = GGNN_KSTEP_AHEAD < NB ? GGNN_KSTEP_AHEAD : NB - 1;
  u32x4 wf[AH + 1][DC_PL];
#pragma unroll
  for (int a = 0; a < AH; ++a)
#pragma unroll
    for (int p = 0; p < DC_PL; ++p) wf[a][p] = pw[(a * DC_PL + p) * 64];
  __builtin_amdgcn_sched_group_barrier(0x100, AH * DC_PL, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (nb + AH < NB) {
#pragma unroll
      for (int p = 0; p < DC_PL; ++p) wf[(nb + AH) % (AH + 1)][p] = pw[((nb + AH) * DC_PL + p) * 64];
    }
    mfma_x3h(xb, wf[nb % (AH + 1)], acc[nb].m, acc[nb].c);
    if (nb + AH < NB) __builtin_amdgcn_sched_group_barrier(0x100, DC_PL, 0);  // DS read
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                         // MFMA
  }
}

// lane k of every 16-lane row -> the whole row (ds_swizzle bit mode: and 0x10, or k)
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (K << 5) | 0x10));
}

__device__ __forceinline__ void dec_cell2_body(const ggnn_dec_cell2_args& A2, const int tileset,
                                               unsigned char* __restrict__ smem) {
  const ggnn_dec_cell_args& A = A2.cell;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // matrix view (P3, P4, LSTM): node lr of the tile, k-group / output rows 4 kq ..
  // sweep view (P1's result, P2): lane lr of DPP row kq owns channels ch..ch+2, 48+ch.. of nodes 4 kq .. 4 kq + 3
  const int lr = lane & 15, kq = lane >> 4;
  const int ch = 3 * lr;
  constexpr int CH2 = C / 2;

  unsigned char* __restrict__ wbase = smem + 2 * DC_SLICE + wave * D2_WAVE_LDS;
  int* __restrict__ csr = reinterpret_cast<int*>(wbase + D2_XF);   // [e][17 + D2_CW]
  float* __restrict__ rstat = reinterpret_cast<float*>(wbase + D2_XF + 2 * D2_CSR);   // [e][gate][node][mx, inv, sum alpha, sum alpha a_e]

  const int n_dst = (int)A.n_dst, n_in = A.n_in, F = A.f_dst;
  // a ragged last tile slides back over rows the previous tile also produces (identical duplicate results);
  // tiles past the end (a workgroup's surplus waves) repeat the last one: every wave runs the whole program,
  // so the workgroup barriers of the slice stream need no special case
  const int row0 = max(0, min((tileset * D2_WAVES + wave) * 16, n_dst - 16));
  const int node_m = min(row0 + lr, n_dst - 1);    // this lane's node in the matrix view (n_dst < 16: clamped)

  // ---- the weight stream: slice s -> buffer s & 1, fetched one slice ahead by all the workgroup's waves ----
  const unsigned char* __restrict__ wsrc = reinterpret_cast<const unsigned char*>(A.wstream) + lane * 16;
  const uint32_t slice_lds =
      __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)));
  int s_cur = 0;
  [[maybe_unused]] unsigned long long st_wait = 0, st_p1 = 0, st_p2 = 0, st_p3 = 0, st_p4 = 0, st_lstm = 0;
  [[maybe_unused]] unsigned long long st_dma = 0, st_t0 = 0, st_pa = 0;
  GGNN_STAMP(0);
  auto dma_slice = [&](int s, int np) {
    const unsigned char* src = wsrc + (size_t)s * DC_SLICE;
    const uint32_t dst = slice_lds + (s & 1) * DC_SLICE;
    for (int p = wave; p < np; p += D2_WAVES) dc_dma16(src + p * 1024, dst + p * 1024);
  };
  // `np_next`: pieces of the slice after the current one (14: a P1 slice, 12: P3 / P4, 0: none)
  auto begin_slice = [&](int np_next) -> const u32x4* {   // the slice about to be used landed at the previous end_slice
    st_t0 = GGNN_STAMP_NOW();
    if (np_next > 0) dma_slice(s_cur + 1, np_next);
    st_dma += GGNN_STAMP_NOW() - st_t0;
    return reinterpret_cast<const u32x4*>(smem + (s_cur & 1) * DC_SLICE) + lane;
  };
  auto end_slice = [&]() {
    [[maybe_unused]] const unsigned long long w0 = GGNN_STAMP_NOW();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next slice are in LDS
    __syncthreads();                                   // ... everybody's are, and nobody reads the old one any more
    st_wait += GGNN_STAMP_NOW() - w0;
    ++s_cur;
  };
  dma_slice(0, DC_NP1);
  // Range flag (ggnn.h, OPERAND RANGE): the operands of the two-piece fp16 split are checked where they are made --
  // the tile's input rows here in the prologue, the aggregates when a row is closed -- and reported at once.
  auto report_range = [&](bool bad) __attribute__((always_inline)) {
    if (A.flags != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(A.flags, GGNN_FLAG_F16_RANGE);
  };

  // ---- tile prologue: the tile's input rows [h | x | 1 | 0] -> LDS AS THE TWO fp16 PLANES of their fragments (a
  // lane (node lr, k-group kq) holds x[node][32 ks + 8 kq ..+7]: the A fragment of P1 and the B fragment of P4 are the
  // same registers).  A lane only ever reads the 16-byte slots it writes: [k-step 0..2][plane][lane] and, for the 16
  // feature slots, [plane][lanes of k-groups 0 and 1] = 7 KB.  CSR windows -> LDS. ----
  unsigned char* __restrict__ xpl = wbase + lane * 16;
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
    const float* hrow = A.h_dst + (int64_t)node_m * A.ldh + 8 * kq;
    f32x4 hv[3][2];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      hv[ks][0] = *reinterpret_cast<const f32x4*>(hrow + 32 * ks);
      hv[ks][1] = *reinterpret_cast<const f32x4*>(hrow + 32 * ks + 4);
    }
    // features: slots 8 (kq & 1) ..+7 of [x_0 .. x_{F-1}, 1 (bias), 0 ..]
    const float* xrow = A.x_dst + (int64_t)node_m * A.ldx;
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
      int* __restrict__ rp = csr + e * (17 + D2_CW);
      if (lane < 17) rp[lane] = Sw.rowptr[min(row0 + lane, n_dst)];
      __builtin_amdgcn_wave_barrier();
      const int pbase = rp[0], e_last = (int)Sw.E - 1;
      if (Sw.E > 0) {
        for (int k = lane; k < D2_CW; k += 64) rp[17 + k] = Sw.col[min(pbase + k, e_last)];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // slice 0 is in LDS
  GGNN_STAMP(1);


  // ======================= phase A: per edge type, the four gates' scores of the tile's in-edges =======================
  const int n_p1 = 16 * n_in;   // P1 slices in front of the stream
#pragma unroll 1
  for (int e = 0; e < n_in; ++e) {
    const ggnn_dec_cell_sweep& Sw = A.in[e];
    [[maybe_unused]] const unsigned long long t_a = GGNN_STAMP_NOW();
    // uv[gi][cc][r] (cc < 6) = u_h[node 4 kq + r][channel ch + cc | 48 + ch + cc - 3], uv[gi][6][r] = u4[node 4 kq + r][slot lr]
    f32x4 uv[4][7];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      DcAcc u[7];
#pragma unroll
      for (int nb = 0; nb < 7; ++nb) u[nb].zero();
      u32x4 xb[2][DC_PL];
      x_planes(0, xb[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4* pw = begin_slice(s_cur + 1 < n_p1 ? DC_NP1 : DC_NP3);
        dc_kstep_xa<7>(pw, xb[ks & 1], u);
        if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);
        end_slice();
      }
#pragma unroll
      for (int nb = 0; nb < 7; ++nb) uv[gi][nb] = u[nb].value();
    }
    [[maybe_unused]] const unsigned long long t_b = GGNN_STAMP_NOW();
    {
      const float* __restrict__ hbase = Sw.h_src + ch;
      const float* __restrict__ einfo = Sw.einfo;
      float* __restrict__ sws = A2.score_ws[e];
      const uint32_t ldh = (uint32_t)Sw.ldh_src;
      const int* __restrict__ rp = csr + e * (17 + D2_CW);
      const int* __restrict__ colw = rp + 17;
      const int pbase = rp[0], e_last = max((int)Sw.E - 1, 0);
      const bool has_edges = Sw.E > 0;
      float* __restrict__ rs_e = rstat + e * (4 * 16 * 4);
      struct Row {
        float mx[4], den[4], sae[4];
        int p, pe;
      };
      struct Unit {
        f3 hh[GGNN_UNIT_EDGES][2];
        float x4[GGNN_UNIT_EDGES];
      };
      auto open_row = [&](Row& r, int n) __attribute__((always_inline)) {
        const int nl = min(row0 + n, n_dst - 1) - row0;   // (n_dst < 16: rows past the end repeat the last node)
        r.p = rp[nl];
        r.pe = rp[nl + 1];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          r.mx[g] = -INFINITY;
          r.den[g] = r.sae[g] = 0.f;
        }
      };
      const bool in_window = __builtin_amdgcn_readfirstlane(rp[16] - pbase) <= D2_CW;
      auto gather = [&](const Row& r, Unit& U, auto window_tag) __attribute__((always_inline)) {
        constexpr bool WINDOW = decltype(window_tag)::value;
#pragma unroll
        for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
          const int pt = min(r.p + t, e_last);
          int j;
          if constexpr (WINDOW) j = colw[min(max(pt - pbase, 0), D2_CW - 1)];
          else j = has_edges ? Sw.col[pt] : 0;
          if (!has_edges) j = 0;
          U.hh[t][0] = ld3(hbase + (uint32_t)j * ldh);
          U.hh[t][1] = ld3(hbase + (uint32_t)j * ldh + CH2);
          U.x4[t] = einfo[(uint32_t)pt * GGNN_EINFO_ROW + lr];
        }
      };
      auto fold = [&](Row& r, const Unit& U, const int rr) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        const int nact = min(max(r.pe - r.p, 0), GGNN_UNIT_EDGES);
        if (nact > 0) {
          float s[4][GGNN_UNIT_EDGES];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float mnew = r.mx[g];
#pragma unroll
            for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
              s[g][t] = -INFINITY;
              if (t < nact) {
                float part = uv[g][6][rr] * U.x4[t];
                part = __builtin_fmaf(uv[g][0][rr], U.hh[t][0].x, part);
                part = __builtin_fmaf(uv[g][1][rr], U.hh[t][0].y, part);
                part = __builtin_fmaf(uv[g][2][rr], U.hh[t][0].z, part);
                part = __builtin_fmaf(uv[g][3][rr], U.hh[t][1].x, part);
                part = __builtin_fmaf(uv[g][4][rr], U.hh[t][1].y, part);
                part = __builtin_fmaf(uv[g][5][rr], U.hh[t][1].z, part);
                s[g][t] = row_sum(part);   // 1 / sqrt(96) is folded into u
                mnew = fmaxf(mnew, s[g][t]);
              }
            }
            const float scale = __expf(r.mx[g] - mnew);   // exp(-inf) = 0 on a row's first unit
            float den = r.den[g] * scale, sae = r.sae[g] * scale;
#pragma unroll
            for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
              if (t < nact) {
                const float pw_ = __expf(s[g][t] - mnew);
                den = den + pw_;
                sae = __builtin_fmaf(pw_, U.x4[t], sae);   // lane 13: sum alpha a_e (other lanes: unused)
              }
            }
            r.den[g] = den;
            r.sae[g] = sae;
            r.mx[g] = mnew;
          }
          // lane t of the row writes edge p + t's four scores (gate order of the stream)
#pragma unroll
          for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
            if (t < nact && lr == t) {
              float* dst = sws + (size_t)(r.p + t) * 4;
              __hip_atomic_store(reinterpret_cast<uint32_t*>(dst) + 0, __float_as_uint(s[0][t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(reinterpret_cast<uint32_t*>(dst) + 1, __float_as_uint(s[1][t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(reinterpret_cast<uint32_t*>(dst) + 2, __float_as_uint(s[2][t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(reinterpret_cast<uint32_t*>(dst) + 3, __float_as_uint(s[3][t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        r.p += GGNN_UNIT_EDGES;
      };
      auto close_row = [&](const Row& r, int n) __attribute__((always_inline)) {
#pragma clang fp contract(off)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float inv = 1.0f / (r.den[g] + 1e-16f);   // PyG softmax denominator
          float* q = rs_e + (g * 16 + n) * 4;
          if (lr == 0) {
            q[0] = r.mx[g];
            q[1] = inv;
            q[2] = r.den[g] * inv;
          }
          if (lr == 13) q[3] = r.sae[g] * inv;
        }
      };
      auto sweep = [&](auto window_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int r0 = 0; r0 < 4; r0 += D2_ROWS_A_) {
          Row rr[D2_ROWS_A_];
#pragma unroll
          for (int q = 0; q < D2_ROWS_A_; ++q) open_row(rr[q], 4 * kq + r0 + q);
          bool more;
          do {
            Unit un[D2_ROWS_A_];
#pragma unroll
            for (int q = 0; q < D2_ROWS_A_; ++q) gather(rr[q], un[q], window_tag);
            more = false;
#pragma unroll
            for (int q = 0; q < D2_ROWS_A_; ++q) {
              fold(rr[q], un[q], r0 + q);
              more |= rr[q].p < rr[q].pe;
            }
          } while (__builtin_amdgcn_ballot_w64(more) != 0);
#pragma unroll
          for (int q = 0; q < D2_ROWS_A_; ++q) close_row(rr[q], 4 * kq + r0 + q);
        }
      };
      if (in_window) sweep(std::true_type{});
      else sweep(std::false_type{});
    }
    st_p1 += t_b - t_a;
    st_pa += GGNN_STAMP_NOW() - t_b;
  }
  // (the scores are read back by the wave that wrote them: its stores are acknowledged by the slice stream's next
  // s_waitcnt vmcnt(0) -- P3 has three before the first value pass could start -- and both sides go to the L2)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();

  // ======================= phase B: per gate, values -> lin_l2 -> skip -> LSTM =======================
  f32x4 run[6];   // the LSTM update as the gates arrive: sig(i) -> sig(i) tanh(c~) -> c' -> (h')
#pragma unroll
  for (int ct = 0; ct < 6; ++ct) run[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int gi = 0; gi < 4; ++gi) {
    const int g = gi == 1 ? 2 : (gi == 2 ? 1 : gi);   // weights are indexed i, f, c, o; processed i, c~, f, o
    f32x4 pre[6], cin[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) pre[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int e = 0; e < n_in; ++e) {
      const ggnn_dec_cell_sweep& Sw = A.in[e];
      [[maybe_unused]] const unsigned long long t_b = GGNN_STAMP_NOW();
      const float* __restrict__ rs_eg = rstat + (e * 4 + gi) * (16 * 4);
      f32x4 ag[6];   // ag[cc][r]: aggregate channel slot cc of node 4 kq + r
      {
        const float* __restrict__ ep = Sw.edge_params + g * GGNN_EDGE_PARAM_ROWS * C;
        f3 wv[6];
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          const float* w = ep + ch + (cc < 3 ? cc : CH2 + cc - 3);
          wv[cc] = {w[0], w[C], w[2 * C]};
        }
        const float* __restrict__ vbase = Sw.v_src + Sw.v_off + g * C + ch;
        const float* __restrict__ einfo = Sw.einfo;
        const float* __restrict__ sws = A2.score_ws[e] + gi;
        const uint32_t ldv = (uint32_t)Sw.ldv;
        const int* __restrict__ rp = csr + e * (17 + D2_CW);
        const int* __restrict__ colw = rp + 17;
        const int pbase = rp[0], e_last = max((int)Sw.E - 1, 0);
        const bool has_edges = Sw.E > 0;
        struct Row {
          float mx, inv, acc[6];
          int p, pe;
        };
        struct Unit {
          f3 vv[GGNN_UNIT_EDGES][2];
          float sc[GGNN_UNIT_EDGES], rl[GGNN_UNIT_EDGES];
        };
        auto open_row = [&](Row& r, int n) __attribute__((always_inline)) {
          const int nl = min(row0 + n, n_dst - 1) - row0;
          r.p = rp[nl];
          r.pe = rp[nl + 1];
          const float2 st = *reinterpret_cast<const float2*>(rs_eg + n * 4);
          r.mx = st.x;
          r.inv = st.y;
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) r.acc[cc] = 0.f;
        };
        const bool in_window = __builtin_amdgcn_readfirstlane(rp[16] - pbase) <= D2_CW;
        auto gather = [&](const Row& r, Unit& U, auto window_tag) __attribute__((always_inline)) {
          constexpr bool WINDOW = decltype(window_tag)::value;
#pragma unroll
          for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
            const int pt = min(r.p + t, e_last);
            int j;
            if constexpr (WINDOW) j = colw[min(max(pt - pbase, 0), D2_CW - 1)];
            else j = has_edges ? Sw.col[pt] : 0;
            if (!has_edges) j = 0;
            U.vv[t][0] = ld3(vbase + (uint32_t)j * ldv);
            U.vv[t][1] = ld3(vbase + (uint32_t)j * ldv + CH2);
            U.sc[t] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(sws + (size_t)pt * 4), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT));
            U.rl[t] = einfo[(uint32_t)pt * GGNN_EINFO_ROW + (lr & 3)];   // slots 0..2 = reloc_e
          }
        };
        auto fold = [&](Row& r, const Unit& U) __attribute__((always_inline)) {
#pragma clang fp contract(off)
          const int nact = min(max(r.pe - r.p, 0), GGNN_UNIT_EDGES);
#pragma unroll
          for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
            if (t < nact) {
              const float rx = row_bcast<0>(U.rl[t]), ry = row_bcast<1>(U.rl[t]), rz = row_bcast<2>(U.rl[t]);
              const float pw_ = __expf(U.sc[t] - r.mx) * r.inv;
              const float v[6] = {U.vv[t][0].x, U.vv[t][0].y, U.vv[t][0].z, U.vv[t][1].x, U.vv[t][1].y, U.vv[t][1].z};
#pragma unroll
              for (int cc = 0; cc < 6; ++cc) {
                const float val = __builtin_fmaf(wv[cc].z, rz, __builtin_fmaf(wv[cc].y, ry, __builtin_fmaf(wv[cc].x, rx, v[cc])));
                r.acc[cc] = __builtin_fmaf(pw_, fmaxf(val, 0.f), r.acc[cc]);
              }
            }
          }
          r.p += GGNN_UNIT_EDGES;
        };
        auto sweep = [&](auto window_tag) __attribute__((always_inline)) {
          uint32_t amx = 0u;   // (aggregates are sums of relu outputs; a NaN among the gathered operands ends up here)
#pragma unroll
          for (int r0 = 0; r0 < 4; r0 += D2_ROWS_B_) {
            Row rr[D2_ROWS_B_];
#pragma unroll
            for (int q = 0; q < D2_ROWS_B_; ++q) open_row(rr[q], 4 * kq + r0 + q);
            bool more;
            do {
              Unit un[D2_ROWS_B_];
#pragma unroll
              for (int q = 0; q < D2_ROWS_B_; ++q) gather(rr[q], un[q], window_tag);
              more = false;
#pragma unroll
              for (int q = 0; q < D2_ROWS_B_; ++q) {
                fold(rr[q], un[q]);
                more |= rr[q].p < rr[q].pe;
              }
            } while (__builtin_amdgcn_ballot_w64(more) != 0);
#pragma unroll
            for (int q = 0; q < D2_ROWS_B_; ++q) {
#pragma unroll
              for (int cc = 0; cc < 6; ++cc) ag[cc][r0 + q] = rr[q].acc[cc];
              dc_track(amx, rr[q].acc[0], rr[q].acc[1]);
              dc_track(amx, rr[q].acc[2], rr[q].acc[3]);
              dc_track(amx, rr[q].acc[4], rr[q].acc[5]);
            }
          }
          report_range(amx >= DC_RANGE_LIMIT);
        };
        if (in_window) sweep(std::true_type{});
        else sweep(std::false_type{});
      }
      [[maybe_unused]] const unsigned long long t_c = GGNN_STAMP_NOW();
      // ================= P3: pre += lin_l2(e, g) . agg + (b_l2, w_edge) . (sum alpha, sum alpha a) =================
      {
        // sweep layout -> matrix layout on the matrix pipe: aT[cc][i] = agg[node lr][channel slot cc of lane 4 kq + i]
        // D[i][j] += sum_k A[i][k] B[k][j] with A[i = lr][k = kq] = agg of node 4 kq + r in lane lr and
        // B[k][j] = (j == 4 k + r): exact (one non-zero product per element).
        f32x4 aT[6];
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) aT[cc] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sel = lr == 4 * kq + r ? 1.0f : 0.0f;
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) aT[cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[cc][r], sel, aT[cc], 0, 0, 0);
        }
        float wtail[6];
        {
          const float* __restrict__ wt = A.w2_tail + (size_t)((g * n_in + e) * 6) * 64 + lane;
#pragma unroll
          for (int ct = 0; ct < 6; ++ct) wtail[ct] = wt[ct * 64];
        }
        u32x4 ab[3][DC_PL];   // lin_l2's B fragments: k-step ks, k slot 8 kq + j = aT[2 ks + (j >> 2)][j & 3]
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) dc_split(aT[2 * ks], aT[2 * ks + 1], ab[ks]);
        // sum alpha | sum alpha a_e of node lr meet b_l2 | w_edge in k-groups 0 | 3
        const float xs = rs_eg[lr * 4 + (kq == 0 ? 2 : 3)];
        const float xt = (kq == 0 || kq == 3) ? xs : 0.f;
        DcAcc part[6];
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) part[ct].zero();
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const u32x4* pw = begin_slice(DC_NP3);   // next: P3 or P4
          dc_kstep<6>(pw, ab[ks], part);
          end_slice();
        }
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] += part[ct].value();
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wtail[ct], xt, pre[ct], 0, 0, 0);
      }
      st_p2 += t_c - t_b;
      st_p3 += GGNN_STAMP_NOW() - t_c;
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
        const u32x4* pw = begin_slice((ks < 3 || gi < 3) ? DC_NP3 : 0);   // next: P4, the next gate's P3, or nothing
        dc_kstep<6>(pw, xb[ks & 1], part);
        if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);
        end_slice();
      }
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) pre[ct] += part[ct].value();
    }

    [[maybe_unused]] const unsigned long long t_f = GGNN_STAMP_NOW();
    st_p4 += t_f - t_e;
    // ================= LSTM update, folded in gate by gate (heteropgclstm.py:140-146) =================
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
  GGNN_STAMP_VAL(12, st_pa);
  GGNN_STAMP(16);
}

__global__ __launch_bounds__(D2_WAVES * 64, 1) void dec_cell2_kernel(const DecCell2Batch B) {
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[D2_LDS];
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const int nwg = B.wg_off[k + 1] - B.wg_off[k];
  // workgroups that share an XCD take one contiguous range of tile sets (speed only)
  const int ts = xcd_remap((int)blockIdx.x - B.wg_off[k], nwg);
  dec_cell2_body(B.a[k], ts, s_raw);
}

}  // namespace ggnn

extern "C" int ggnn_decoder_cell2_batch(const ggnn_dec_cell2_args* args, int n_problems, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_problems < 1 || n_problems > D2_MAX_PROBLEMS) return GGNN_EINVAL;
  DecCell2Batch B;
  B.n = n_problems;
  B.wg_off[0] = 0;
  for (int k = 0; k < D2_MAX_PROBLEMS; ++k) {
    B.a[k] = args[k < n_problems ? k : 0];
    if (k >= n_problems) {
      B.wg_off[k + 1] = B.wg_off[k];
      continue;
    }
    const ggnn_dec_cell_args& A = B.a[k].cell;
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
      if (Sw.E > 0 && (!B.a[k].score_ws[e] || !aligned16(B.a[k].score_ws[e]))) return GGNN_EINVAL;
      if (Sw.ldh_src < C || Sw.v_off < 0 || Sw.v_off + 4 * C > Sw.ldv) return GGNN_EINVAL;
      if (Sw.n_src * Sw.ldh_src >= INT32_MAX || Sw.n_src * Sw.ldv >= INT32_MAX ||
          (Sw.E + GGNN_UNIT_EDGES) * GGNN_EINFO_ROW >= INT32_MAX)
        return GGNN_EINVAL;  // gathered rows are addressed with 32-bit offsets
    }
    const int64_t n_ts = (A.n_dst + 16 * D2_WAVES - 1) / (16 * D2_WAVES);
    if (B.wg_off[k] + n_ts >= INT32_MAX) return GGNN_EINVAL;
    B.wg_off[k + 1] = B.wg_off[k] + (int)n_ts;
  }
  hipLaunchKernelGGL(dec_cell2_kernel, dim3((unsigned)B.wg_off[D2_MAX_PROBLEMS]), dim3(D2_WAVES * 64), 0,
                     (hipStream_t)stream, B);
  return launch_status();
}
