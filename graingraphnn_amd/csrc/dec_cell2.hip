// Decoder HeteroPGCLSTM cell, PHASE-SHIFTED variant of dec_cell.hip (ggnn_decoder_cell2_batch, include/ggnn.h): the same
// cell -- destination-side projections, the periodic-boundary GAT sweep (PeriodConv.message, periodGATconv.py:204-236, +
// propagate's gather / scatter-add), lin_l2 + the value-side lin_edge term, HeteroConv's sum over the edge types and the
// LSTM update (heteropgclstm.py:111-146) -- a 16-node tile per wave, eight waves per workgroup.
//
// dec_cell.hip's waves walk their tiles in lockstep (one workgroup barrier per weight slice): all eight are in a GEMM
// phase (matrix pipe + LDS reads; the memory pipe idles) or all eight are in a sweep (gathers + VALU; the matrix pipe idles).
// Here the workgroup is two HALVES of four waves, one per SIMD, that run the same program ONE ITEM APART: a tile's program is
//     G_0 S_0 G_1 S_1 ... S_{L-1} G_L,   L = 4 n_in sweeps S, GEMM blocks G (P1 | P3 + P1 | P3 + P4 + LSTM + P1 | P3 + P4 + LSTM),
// and while one half is in block G_i (n_i weight slices, a workgroup barrier behind each) the other is in a sweep that
// passes the same n_i barriers between issuing its gathers and folding them: the barrier is the rendezvous of both halves,
// a SIMD always has one wave on the matrix pipe and one on the memory pipe, and a slice is read from LDS by four waves.
// Each half has its own position in the weight stream (the stream is fetched twice per workgroup) and three slice buffers.
// Layouts as the stage-free experiment of round 5 (profiles/r5_dec_cell_experiments.txt): P1 with the tile's rows as the
// A operand leaves u in the sweep's layout (packing.DC_P1_ROW), the aggregates return to the matrix layout through exact
// v_mfma_f32_16x16x4_f32 transposes (packing.DC_P3_COL): no LDS stage, 8 KB of LDS per wave.
// Arithmetic as dec_cell.hip: the GEMM phases on two fp16 pieces and three MFMA products, the sweep in fp32 with explicit fmas.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "cell_common.h"
#define GGNN_STAMP_SUFFIX _dec2
#include "stamps.h"

namespace ggnn {

constexpr int D2_WAVES = 8, D2_HALF = 4;            // 16-node tiles per workgroup, waves per half
#ifndef D2_INTERLEAVED_
#define D2_INTERLEAVED_ 0   // 0: waves 0..3 | 4..7 are the halves (consecutive waves on different SIMDs); 1: even | odd
#endif
#if D2_INTERLEAVED_
#define D2_HALF_OF(w) ((w) & 1)
#define D2_RANK_OF(w) ((w) >> 1)
#else
#define D2_HALF_OF(w) ((w) >> 2)
#define D2_RANK_OF(w) ((w) & 3)
#endif
constexpr int D2_NBUF = 3;                          // slice buffers per half
constexpr int D2_MAX_PROBLEMS = 4;
constexpr int DC_SLICE = GGNN_DC_SLICE_BYTES;       // 14 pieces of 1 KB
static_assert(DC_SLICE == 7 * DC_PL * 1024, "slice = 7 column tiles x planes x 1 KB");
constexpr int DC_NP1 = 7 * DC_PL, DC_NP3 = 6 * DC_PL;   // pieces of a P1 slice / of a P3 or P4 slice
constexpr int D2_XF = 7 * 1024;                     // the tile's input rows as fragment planes
constexpr int D2_CW = 111;                          // source indices of a tile kept in LDS per edge type
constexpr int D2_CSR = (17 + D2_CW) * 4;            // 512 B
constexpr int D2_WAVE_LDS = D2_XF + 2 * D2_CSR;     // 8 KB
constexpr int D2_LDS = 2 * D2_NBUF * DC_SLICE + D2_WAVES * D2_WAVE_LDS;   // 151 552 B
static_assert(D2_LDS <= 160 * 1024, "LDS");

struct DecCell2Batch {
  ggnn_dec_cell_args a[D2_MAX_PROBLEMS];
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

__device__ __forceinline__ void dec_cell2_body(const ggnn_dec_cell_args& A, const int tileset,
                                               unsigned char* __restrict__ smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // matrix view (P3, P4, LSTM): node lr of the tile, k-group / output rows 4 kq ..
  // sweep view (P1's result, P2): lane lr of DPP row kq owns channels ch..ch+2, 48+ch.. of nodes 4 kq .. 4 kq + 3
  const int lr = lane & 15, kq = lane >> 4;
  const int half = D2_HALF_OF(wave), hw = D2_RANK_OF(wave);   // which half of the workgroup, rank in it
  const int ch = 3 * lr;
  constexpr int CH2 = C / 2;

  unsigned char* __restrict__ wbase = smem + 2 * D2_NBUF * DC_SLICE + wave * D2_WAVE_LDS;
  int* __restrict__ csr = reinterpret_cast<int*>(wbase + D2_XF);   // [e][17 + D2_CW]

  const int n_dst = (int)A.n_dst, n_in = A.n_in, F = A.f_dst;
  // a ragged last tile slides back over rows the previous tile also produces (identical duplicate results);
  // tiles past the end (a workgroup's surplus waves) repeat the last one: every wave runs the whole program,
  // so the workgroup barriers of the slice stream need no special case
  const int row0 = max(0, min((tileset * D2_WAVES + wave) * 16, n_dst - 16));
  const int node_m = min(row0 + lr, n_dst - 1);    // this lane's node in the matrix view (n_dst < 16: clamped)

  // ---- the weight stream, ONE POSITION PER HALF: slice s -> buffer s % 3 of the half's LDS region, fetched TWO slices
  // ahead by the half's four waves (a sweep of this half falls between two blocks: the next two slices wait in LDS) ----
  const unsigned char* __restrict__ wsrc = reinterpret_cast<const unsigned char*>(A.wstream) + lane * 16;
  unsigned char* __restrict__ sbase = smem + half * (D2_NBUF * DC_SLICE);
  const uint32_t slice_lds =
      __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(sbase)));
  const int per_gate = 7 * n_in + 4, n_slices = 4 * per_gate;
  int s_cur = 0, b_cur = 0, n_ahead = 0;   // slice in use, its buffer, this wave's DMA operations for slice s_cur + 2
  [[maybe_unused]] unsigned long long st_wait = 0, st_p1 = 0, st_p2 = 0, st_p3 = 0, st_p4 = 0, st_lstm = 0;
  [[maybe_unused]] unsigned long long st_dma = 0, st_t0 = 0, st_spin = 0;
  GGNN_STAMP(0);
  auto dma_slice = [&](int s, int buf) -> int {
    if (s >= n_slices) return 0;
    const int q = s % per_gate;   // within a gate: (P1 x 4, P3 x 3) per edge type, P4 x 4
    const int np = (q < 7 * n_in && (q % 7) < 4) ? DC_NP1 : DC_NP3;
    const unsigned char* src = wsrc + (size_t)s * DC_SLICE;
    const uint32_t dst = slice_lds + buf * DC_SLICE;
    int n = 0;
    for (int p = hw; p < np; p += D2_HALF) {
      dc_dma16(src + p * 1024, dst + p * 1024);
      ++n;
    }
    return n;
  };
  auto begin_slice = [&]() -> const u32x4* {
    st_t0 = GGNN_STAMP_NOW();
    // slice s_cur + 2 -> the buffer slice s_cur - 1 was read from: the half left it before the last barrier
    n_ahead = dma_slice(s_cur + 2, b_cur == 0 ? D2_NBUF - 1 : b_cur - 1);
    st_dma += GGNN_STAMP_NOW() - st_t0;
    return reinterpret_cast<const u32x4*>(sbase + b_cur * DC_SLICE) + lane;
  };
  // The workgroup barrier is a rendezvous of BOTH halves: the half in a GEMM block ends every slice with one, the
  // half in a sweep passes the same number between its gathers and its folds (spin).  Raw s_barrier: a __syncthreads
  // would wait for every vector-memory operation in flight, the slice two ahead and the other half's gathers included.
  auto end_slice = [&]() {
    [[maybe_unused]] const unsigned long long w0 = GGNN_STAMP_NOW();
    // slice s_cur + 1 was issued before the n_ahead operations of slice s_cur + 2 (loads return in order)
    if (n_ahead >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n_ahead == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    st_wait += GGNN_STAMP_NOW() - w0;
    ++s_cur;
    b_cur = b_cur == D2_NBUF - 1 ? 0 : b_cur + 1;
  };
  auto spin = [&](int n) {
    [[maybe_unused]] const unsigned long long w0 = GGNN_STAMP_NOW();
    for (int i = 0; i < n; ++i) asm volatile("s_barrier" ::: "memory");
    st_spin += GGNN_STAMP_NOW() - w0;
  };
  dma_slice(0, 0);
  dma_slice(1, 1);
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
  __syncthreads();   // slices 0 and 1 of both halves are in LDS
  GGNN_STAMP(1);


  f32x4 run[6];   // the LSTM update as the gates arrive: sig(i) -> sig(i) tanh(c~) -> c' -> (h')
  f32x4 pre[6];   // the gate's pre-activation
#pragma unroll
  for (int ct = 0; ct < 6; ++ct) run[ct] = pre[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 uv[7];    // u of the pass, then its aggregates (sweep layout)
  f32x4 ys;       // the pass's rank-1 tail scalars: sum alpha in lane 0, sum alpha a_e in lane 13 of a row group
#pragma unroll
  for (int cc = 0; cc < 7; ++cc) uv[cc] = (f32x4){0.f, 0.f, 0.f, 0.f};
  ys = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int L = 4 * n_in;
  if (half == 1) spin(4);   // the first half's block G_0
  int pe = 0, pg = 0;       // edge type and gate (weight index) of the previous pass
#pragma unroll 1
  for (int it = 0; it <= L; ++it) {
    const int gi = it / n_in, ei = it - gi * n_in;           // (gi == 4: the closing block)
    const int g = gi == 1 ? 2 : (gi == 2 ? 1 : gi);          // weights are indexed i, f, c, o; processed i, c~, f, o
    // the edge types are walked forwards for the first and third gate of the stream and backwards for the second and
    // fourth: the rows a pass gathered are gathered again by the very next pass while the XCD's L2 still holds them
    const int e = (gi & 1) ? n_in - 1 - ei : ei;
    [[maybe_unused]] const unsigned long long t_a = GGNN_STAMP_NOW();
    // ================= block G_it =================
    if (it > 0) {
      // ---- P3 of the previous pass: pre += lin_l2(pe, pg) . agg + (b_l2, w_edge) . (sum alpha, sum alpha a) ----
      // sweep layout -> matrix layout on the matrix pipe: aT[cc][i] = agg[node lr][channel slot cc of lane 4 kq + i]
      // D[i][j] += sum_k A[i][k] B[k][j] with A[i = lr][k = kq] = agg of node 4 kq + r in lane lr and
      // B[k][j] = (j == 4 k + r): exact (one non-zero product per element).
      f32x4 aT[7];
#pragma unroll
      for (int cc = 0; cc < 7; ++cc) aT[cc] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sel = lr == 4 * kq + r ? 1.0f : 0.0f;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) aT[cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(uv[cc][r], sel, aT[cc], 0, 0, 0);
        aT[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(ys[r], sel, aT[6], 0, 0, 0);
      }
      float wtail[6];   // the exact fp32 tail's weight fragments: requested here, used behind the three slices
      {
        const float* __restrict__ wt = A.w2_tail + (size_t)((pg * n_in + pe) * 6) * 64 + lane;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) wtail[ct] = wt[ct * 64];
      }
      u32x4 ab[3][DC_PL];   // lin_l2's B fragments: k-step ks, k slot 8 kq + j = aT[2 ks + (j >> 2)][j & 3]
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) dc_split(aT[2 * ks], aT[2 * ks + 1], ab[ks]);
      const float xt = kq == 0 ? aT[6][0] : (kq == 3 ? aT[6][1] : 0.f);   // sum alpha | sum alpha a_e of node lr
      DcAcc part[6];
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) part[ct].zero();
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const u32x4* pw = begin_slice();
        dc_kstep<6>(pw, ab[ks], part);
        end_slice();
      }
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) pre[ct] += part[ct].value();
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) pre[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wtail[ct], xt, pre[ct], 0, 0, 0);
    }
    [[maybe_unused]] const unsigned long long t_b = GGNN_STAMP_NOW();
    st_p3 += t_b - t_a;
    if (it > 0 && ei == 0) {
      const int qi = gi - 1;   // the gate that is complete but for its skip term
      f32x4 cin[6];
      // the old cell state (only the forget gate uses it): in flight during P4, live nowhere else
      if (qi == 2) {
        const float* crow = A.c_in + (int64_t)node_m * C + 4 * kq;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) cin[ct] = *reinterpret_cast<const f32x4*>(crow + 16 * ct);
      }
      // ---- P4: the summed skip term + gate bias ----
      {
        u32x4 xb[2][DC_PL];
        x_planes(0, xb[0]);
        DcAcc part[6];
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) part[ct].zero();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4* pw = begin_slice();
          dc_kstep<6>(pw, xb[ks & 1], part);
          if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);
          end_slice();
        }
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] += part[ct].value();
      }
      [[maybe_unused]] const unsigned long long t_f = GGNN_STAMP_NOW();
      st_p4 += t_f - t_b;
      // ---- LSTM update, folded in gate by gate (heteropgclstm.py:140-146) ----
      if (qi == 0) {
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) run[ct][r] = sigmoidf_(pre[ct][r]);
      } else if (qi == 1) {
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) run[ct][r] *= tanhf_(pre[ct][r]);
      } else if (qi == 2) {
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) run[ct][r] = sigmoidf_(pre[ct][r]) * cin[ct][r] + run[ct][r];
        float* crow = A.c_out + (int64_t)node_m * C + 4 * kq;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) *reinterpret_cast<f32x4*>(crow + 16 * ct) = run[ct];
      } else {
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) pre[ct][r] = sigmoidf_(pre[ct][r]) * tanhf_(run[ct][r]);
        float* hrow = A.h_out + (int64_t)node_m * C + 4 * kq;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) *reinterpret_cast<f32x4*>(hrow + 16 * ct) = pre[ct];
      }
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) pre[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      st_lstm += GGNN_STAMP_NOW() - t_f;
    }
    if (it == L) break;
    const ggnn_dec_cell_sweep& Sw = A.in[e];
    [[maybe_unused]] const unsigned long long t_c = GGNN_STAMP_NOW();
    // ---- P1: u_h | u4 of the tile's 16 nodes for (e, g), in the sweep's layout ----
    // uv[cc][r] (cc < 6) = u_h[node 4 kq + r][channel ch + cc | 48 + ch + cc - 3], uv[6][r] = u4[node 4 kq + r][slot lr]
    {
      DcAcc u[7];
#pragma unroll
      for (int nb = 0; nb < 7; ++nb) u[nb].zero();
      u32x4 xb[2][DC_PL];
      x_planes(0, xb[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4* pw = begin_slice();
        dc_kstep_xa<7>(pw, xb[ks & 1], u);
        if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);
        end_slice();
      }
#pragma unroll
      for (int nb = 0; nb < 7; ++nb) uv[nb] = u[nb].value();
    }
    [[maybe_unused]] const unsigned long long t_d = GGNN_STAMP_NOW();
    st_p1 += t_d - t_c;
    // ================= sweep S_it of (e, g): one node per 16-lane row at a time; the other half is in a block of
    // `nbar` slices: the first half's sweep runs beside the other's copy of the block in front of it, the second
    // half's beside the first's next block =================
    const int nbar = half == 0 ? (ei == 0 ? (gi == 0 ? 4 : 11) : 7) : (ei == n_in - 1 ? (gi == 3 ? 7 : 11) : 7);
    {
        const float* __restrict__ ep = Sw.edge_params + g * GGNN_EDGE_PARAM_ROWS * C;
        f3 wv[6];
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          const float* w = ep + ch + (cc < 3 ? cc : CH2 + cc - 3);
          wv[cc] = {w[0], w[C], w[2 * C]};
        }
        const float* __restrict__ vbase = Sw.v_src + Sw.v_off + g * C + ch;
        const float* __restrict__ hbase = Sw.h_src + ch;
        const float* __restrict__ einfo = Sw.einfo;
        const uint32_t ldv = (uint32_t)Sw.ldv, ldh = (uint32_t)Sw.ldh_src;
        const int* __restrict__ rp = csr + e * (17 + D2_CW);
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
        auto open_row = [&](Row& r, int n, const f3 uh0, const f3 uh1, const float u4) __attribute__((always_inline)) {
          r.uh0 = uh0;
          r.uh1 = uh1;
          r.u4 = u4;
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
        const bool in_window = __builtin_amdgcn_readfirstlane(rp[16] - pbase) <= D2_CW;
        auto gather = [&](const Row& r, Unit& U, auto window_tag) __attribute__((always_inline)) {   // unconditional (clamped) loads, back to back
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
            U.vv[t][0] = ld3(vbase + (uint32_t)j * ldv);
            U.vv[t][1] = ld3(vbase + (uint32_t)j * ldv + CH2);
          }
        };
        auto fold = [&](Row& r, const Unit& U) __attribute__((always_inline)) {
          // Every product-sum is an explicit fma and contraction is off: the two variants of the sweep (and a row
          // computed by two overlapping tiles of a ragged end) must give the same bits, whatever the compiler would
          // have chosen to fuse in each inlined copy.
#pragma clang fp contract(off)
          const int nact = min(max(r.pe - r.p, 0), GGNN_UNIT_EDGES);
          if (nact > 0) {
            float s[GGNN_UNIT_EDGES];
            float mnew = r.mx;
#pragma unroll
            for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
              s[t] = -INFINITY;
              if (t < nact) {
                float part = r.u4 * U.x4[t];
                part = __builtin_fmaf(r.uh0.x, U.hh[t][0].x, part);
                part = __builtin_fmaf(r.uh0.y, U.hh[t][0].y, part);
                part = __builtin_fmaf(r.uh0.z, U.hh[t][0].z, part);
                part = __builtin_fmaf(r.uh1.x, U.hh[t][1].x, part);
                part = __builtin_fmaf(r.uh1.y, U.hh[t][1].y, part);
                part = __builtin_fmaf(r.uh1.z, U.hh[t][1].z, part);
                s[t] = row_sum(part);   // 1 / sqrt(96) is folded into u
                mnew = fmaxf(mnew, s[t]);
              }
            }
            const float scale = __expf(r.mx - mnew);   // exp(-inf) = 0 on a row's first unit
            r.den = r.den * scale;
            r.sae = r.sae * scale;
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) r.acc[cc] = r.acc[cc] * scale;
#pragma unroll
            for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
              if (t < nact) {
                // lane k of every 16-lane row -> the whole row (ds_swizzle bit mode: and 0x10, or k)
                const int xi = __builtin_bit_cast(int, U.x4[t]);
                const float rx = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(xi, (0 << 5) | 0x10));
                const float ry = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(xi, (1 << 5) | 0x10));
                const float rz = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(xi, (2 << 5) | 0x10));
                const float pw_ = __expf(s[t] - mnew);
                r.den = r.den + pw_;
                r.sae = __builtin_fmaf(pw_, U.x4[t], r.sae);   // lane 13: sum alpha a_e (other lanes: unused)
                const float v[6] = {U.vv[t][0].x, U.vv[t][0].y, U.vv[t][0].z, U.vv[t][1].x, U.vv[t][1].y, U.vv[t][1].z};
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                  const float val = __builtin_fmaf(wv[cc].z, rz, __builtin_fmaf(wv[cc].y, ry, __builtin_fmaf(wv[cc].x, rx, v[cc])));
                  r.acc[cc] = __builtin_fmaf(pw_, fmaxf(val, 0.f), r.acc[cc]);
                }
              }
            }
            r.mx = mnew;
          }
          r.p += GGNN_UNIT_EDGES;
        };
        // the row's aggregate (over the u it was computed from) and its two scalars; the aggregates are sums of relu
        // outputs (non-negative): their range check is on the largest
        auto close_row = [&](const Row& r, float (&out)[6], float& y) __attribute__((always_inline)) {
#pragma clang fp contract(off)
          const float inv = 1.0f / (r.den + 1e-16f);   // PyG softmax denominator
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) out[cc] = r.acc[cc] * inv;
          y = lr == 0 ? r.den * inv : (lr == 13 ? r.sae * inv : 0.f);
          uint32_t amx = 0u;   // (aggregates are sums of relu outputs; a NaN among the gathered operands ends up here)
          dc_track(amx, out[0], out[1]);
          dc_track(amx, out[2], out[3]);
          dc_track(amx, out[4], out[5]);
          report_range(amx >= DC_RANGE_LIMIT);
        };
        auto sweep = [&](auto window_tag) __attribute__((always_inline)) {
#pragma unroll
          for (int r0 = 0; r0 < 4; r0 += 2) {
            Row rr[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int r = r0 + q;
              open_row(rr[q], 4 * kq + r, {uv[0][r], uv[1][r], uv[2][r]}, {uv[3][r], uv[4][r], uv[5][r]}, uv[6][r]);
            }
            bool first = true, more;
            do {
              Unit un[2];
#pragma unroll
              for (int q = 0; q < 2; ++q) gather(rr[q], un[q], window_tag);
              // the other half's slices pass while the gathers are in flight
              if (first) spin(r0 == 0 ? nbar / 2 : nbar - nbar / 2);
              first = false;
              more = false;
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                fold(rr[q], un[q]);
                more |= rr[q].p < rr[q].pe;
              }
            } while (__builtin_amdgcn_ballot_w64(more) != 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int r = r0 + q;
              float out[6], y;
              close_row(rr[q], out, y);
#pragma unroll
              for (int cc = 0; cc < 6; ++cc) uv[cc][r] = out[cc];
              ys[r] = y;
            }
          }
        };
        if (in_window) sweep(std::true_type{});
        else sweep(std::false_type{});
    }
    st_p2 += GGNN_STAMP_NOW() - t_d;
    pe = e;
    pg = g;
  }
  if (half == 0) spin(7);   // the second half's closing block
  GGNN_STAMP_VAL(4, st_wait);
  GGNN_STAMP_VAL(5, st_p1);
  GGNN_STAMP_VAL(6, st_p2);
  GGNN_STAMP_VAL(7, st_p3);
  GGNN_STAMP_VAL(8, st_p4);
  GGNN_STAMP_VAL(9, st_lstm);
  GGNN_STAMP_VAL(10, n_in);
  GGNN_STAMP_VAL(11, st_dma);
  GGNN_STAMP_VAL(12, st_spin);
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

extern "C" int ggnn_decoder_cell2_batch(const ggnn_dec_cell_args* args, int n_problems, ggnn_stream_t stream) {
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
