// Gate GEMM + LSTM cell update with the K = 192 / 96 aggregate columns on the bf16 matrix cores
// at fp32-equivalent numerics (exact 3 x bf16 split, 6 products per k-step: common.h) and the 4
// trailing rank-1 columns (sum alpha, sum alpha*a per incoming edge type) on one exact
// v_mfma_f32_16x16x4_f32.  Same contract as gates_kernel (gates.hip):
//   pre[g] = agg[:, g, 0:Ka] . W2[g]^T + skip[g];  LSTM cell of heteropgclstm.py:111-146.
// Matrix-pipe cycles per 16x16 tile and gate: 6 x 6 x 16 + 32 = 608 instead of 49 x 32 = 1568,
// which moves the kernel from the matrix pipe to the HBM stream of its `agg` operand.
//
// Structure: WEIGHT-STATIONARY, no workgroup barrier after the prologue (the pass-and-barrier
// structure of gates.hip spent more time in barriers and pipeline refills than in MFMAs).
//   * A workgroup owns a 32-channel slice (16 channels on small graphs, see GX slices below) for ALL gates and the whole reduction: its weights,
//     already split into bf16 planes in MFMA fragment order by the host-side packing
//     (ggnn_epilogue_args.w2_planes), are copied once into LDS (G x 6 k-steps x 3 planes x 2 column
//     tiles x 1 KB = 144 KB at G = 4, Ka = 196) and read back lane-linearly (conflict-free).
//   * A wave owns 16 nodes x those 32 channels (accumulators for all gates in registers, so the
//     LSTM update needs no exchange) and streams the nodes' `agg` rows: per (gate, k-step) a lane
//     loads the 32 bytes of its MFMA B fragment (node l&15, k = 32 ks + 8 (l>>4)..) straight into
//     registers, GX_DEPTH steps ahead, and splits them into planes when their turn comes.
//   * A wave processes T (1 or 2) node tiles as ONE unrolled stream of T x G x 6 steps, so the
//     prefetch runs across the tile boundary and every wait is an exact counted vmcnt (a runtime
//     tile loop would drain vmcnt to 0 at its header: project_x6.hip).
//   * The three channel slices of a node range read the same `agg` rows; they are given the same
//     XCD (workgroup id mod 8) and neighbouring dispatch slots, so two of the three reads hit L2.
#include "common.h"

namespace ggnn {

constexpr int GX_BM = 16;     // nodes per wave tile
constexpr int GX_WAVES = 8;   // waves per workgroup (two per SIMD)
// Channel slices: NCT column tiles of 16 channels per workgroup, 6 / NCT slices.  NCT = 2 (three
// 32-channel slices) reads `agg` three times; NCT = 1 (six 16-channel slices) reads it six times but
// halves the LDS fill and the MFMA chain per step and doubles the workgroups: used while the graph
// is too small to give every CU a workgroup (cfg2, 2 086 joints: 0.180 -> 0.169 ms per rollout step).
// (gate, k-step) steps of `agg` in flight per wave.  Measured (MI355X, 20 000 joints, isolated
// launches): G = 4: 48 us at depth 4..8, 60 at 12; G = 3: 46 / 39 / 74 / 36 us at 4 / 6 / 8 / 12.
constexpr int gx_depth(int G) { return G == 3 ? 12 : 8; }

template <int G, int MODE, int KA, int T, int NCT>
__global__ __launch_bounds__(GX_WAVES * 64, 1) void gates_x6_kernel(const ggnn_epilogue_args A, int n_ranges) {
  constexpr int GX_SLICES = 6 / NCT;
  constexpr int SW = 16 * NCT;  // channels per slice
  constexpr int KM = KA - 4;    // columns on the bf16 path (192 / 96)
  constexpr int NKS = KM / 32;  // k-steps per gate (6 / 3)
  constexpr int NSTEP = G * NKS;
  constexpr int GX_DEPTH = gx_depth(G);
  constexpr int NPIECE = NSTEP * 3 * NCT * 64;  // 16-byte weight pieces of one slice
  static_assert(KM % 32 == 0, "Ka - 4 must be a multiple of 32");
  __shared__ u32x4 s_w[NPIECE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
#ifdef GX_VAR_CLOCK
  const uint64_t tr_start = __builtin_amdgcn_s_memrealtime();
#endif
  // workgroup -> (node range, channel slice): the slices of a range share blockIdx mod 8 (= XCD)
  const int j = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int slice = j % GX_SLICES, range = (j / GX_SLICES) * 8 + xcd;
  if (range >= n_ranges) return;
  const int64_t ld_agg = A.ld_agg;
  const int gs_ = A.g_stride;

  // ---- prologue: this slice's weight planes -> LDS (a linear copy per (gate, k-step, plane)) ----
  {
    const u32x4* wpl = reinterpret_cast<const u32x4*>(A.w2_planes);
    constexpr int NIT = (NPIECE + GX_WAVES * 64 - 1) / (GX_WAVES * 64);
    u32x4 wreg[NIT];  // all loads first: one L2 round trip, not one per piece
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = min(tid + it * GX_WAVES * 64, NPIECE - 1), gkp = idx / (NCT * 64), rem = idx % (NCT * 64);  // NCT column tiles x 64 lanes
      wreg[it] = wpl[((int64_t)gkp * 6 + NCT * slice) * 64 + rem];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (tid + it * GX_WAVES * 64 < NPIECE) s_w[tid + it * GX_WAVES * 64] = wreg[it];
  }
  // operands of the exact fp32 tail (columns KM .. KM+3): weight side, 16x16x4 fragment layout
  float wt[G][NCT];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int a = 0; a < NCT; ++a) wt[g][a] = A.w2[((int64_t)g * C + slice * SW + a * 16 + lr) * KA + KM + kq];
  __syncthreads();  // the only workgroup barrier
#ifdef GX_VAR_CLOCK
  const uint64_t tr_pro = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- this wave's T node tiles; a ragged last tile slides back (identical duplicate stores) ----
  const int64_t n_mt = (A.N + GX_BM - 1) / GX_BM;
  const int64_t m_last = max(A.N - GX_BM, (int64_t)0);
  const int row_l = (int)min((int64_t)lr, A.N - 1);  // N < 16: clamp the lane's row instead
  int64_t m0[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
    m0[t] = min(min(((int64_t)range * T + t) * GX_WAVES + wave, n_mt - 1) * GX_BM, m_last);

#ifdef GX_VAR_NOSPLIT  // timing experiment only: `agg` taken as if it already were three bf16 planes
  f32x4 raw[GX_DEPTH][3];
#else
  f32x4 raw[GX_DEPTH][2];
#endif
  auto load_step = [&](int gs) {  // global step -> (tile, gate, k-step); nothing here uses a loaded value
    const int t = gs / NSTEP, st = gs % NSTEP, g = st / NKS, ks = st % NKS;
#ifdef GX_VAR_COALESCED  // timing experiment only (wrong lane <-> data assignment)
    const float* a = A.agg + (m0[t] + (lane >> 2)) * ld_agg + g * gs_ + 32 * ks + 8 * (lane & 3);
#else
    const float* a = A.agg + (m0[t] + row_l) * ld_agg + g * gs_ + 32 * ks + 8 * kq;
#endif
#ifdef GX_VAR_BLOCKED  // timing experiment only: a step's fragment as 2 KB of contiguous memory
    const float* ab = A.agg + m0[t] * ld_agg + st * 512 + lane * 4;
    raw[gs % GX_DEPTH][0] = *reinterpret_cast<const f32x4*>(ab);
    raw[gs % GX_DEPTH][1] = *reinterpret_cast<const f32x4*>(ab + 256);
    (void)a;
#else
    raw[gs % GX_DEPTH][0] = *reinterpret_cast<const f32x4*>(a);  // default cache policy: the other
    raw[gs % GX_DEPTH][1] = *reinterpret_cast<const f32x4*>(a + 4);  // two slices find these lines in L2
#endif
#ifdef GX_VAR_NOSPLIT
    raw[gs % GX_DEPTH][2] = *reinterpret_cast<const f32x4*>(a + 8);
#endif
  };
#pragma unroll
  for (int gs = 0; gs < GX_DEPTH && gs < T * NSTEP; ++gs) load_step(gs);

  const u32x4* pw = &s_w[lane];
  f32x4 acc[G][NCT], skip[G][NCT], cold[NCT];
  float xt[G];
#pragma unroll
  for (int gs = 0; gs < T * NSTEP; ++gs) {
    const int t = gs / NSTEP, st = gs % NSTEP, g = st / NKS;
    const int64_t m = m0[t] + row_l;
    if (st == 0) {
      // per-tile side inputs: requested now, consumed in the tile's epilogue
#pragma unroll
      for (int g2 = 0; g2 < G; ++g2) {
#pragma unroll
        for (int a = 0; a < NCT; ++a) {
          acc[g2][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
          skip[g2][a] = *reinterpret_cast<const f32x4*>(A.p_dst + m * A.ldp + A.s_off + g2 * C + slice * SW +
                                                        a * 16 + 4 * kq);
        }
        xt[g2] = A.agg[m * ld_agg + g2 * gs_ + KM + kq];
      }
      if (MODE == GGNN_MODE_LSTM) {
#pragma unroll
        for (int a = 0; a < NCT; ++a)
          cold[a] = *reinterpret_cast<const f32x4*>(A.c_in + m * C + slice * SW + a * 16 + 4 * kq);
      }
    }
    // split this step's fragment, then reuse its ring slot for the step GX_DEPTH ahead
    u32x4 xb[3];
#ifdef GX_VAR_NOSPLIT
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int d = 0; d < 4; ++d) xb[q][d] = __float_as_uint(raw[gs % GX_DEPTH][q][d]);
#else
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint32_t p0, p1, p2;
      split_bf16x3(raw[gs % GX_DEPTH][d >> 1][2 * (d & 1)], raw[gs % GX_DEPTH][d >> 1][2 * (d & 1) + 1], p0, p1, p2);
      xb[0][d] = p0;
      xb[1][d] = p1;
      xb[2][d] = p2;
    }
#endif
    __builtin_amdgcn_sched_barrier(0);  // pin the issue point: hipcc otherwise sinks or bunches the ring loads
    if (gs + GX_DEPTH < T * NSTEP) load_step(gs + GX_DEPTH);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < NCT; ++a) {
      u32x4 wf[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) wf[q] = pw[((st * 3 + q) * NCT + a) * 64];
      acc[g][a] = mfma_x6(wf, xb, acc[g][a]);
    }
    if (st == NSTEP - 1) {
      // ---- tile epilogue: exact fp32 tail, + skip, LSTM; lane holds channels n..n+3 (twice) ----
#pragma unroll
      for (int a = 0; a < NCT; ++a) {
        const int n = slice * SW + a * 16 + 4 * kq;
#pragma unroll
        for (int g2 = 0; g2 < G; ++g2)
          acc[g2][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[g2][a], xt[g2], acc[g2][a], 0, 0, 0) + skip[g2][a];
        if (MODE == GGNN_MODE_RAW) {
#pragma unroll
          for (int g2 = 0; g2 < G; ++g2)
            *reinterpret_cast<f32x4*>(A.raw_out + m * (int64_t)(G * C) + g2 * C + n) = acc[g2][a];
        } else {
          constexpr int GI = 0, GF = 1, GC = (MODE == GGNN_MODE_LSTM) ? 2 : 1,
                        GO = (MODE == GGNN_MODE_LSTM) ? 3 : 2;
          f32x4 hn, cn;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ig = sigmoidf_(acc[GI][a][r]);
            const float tg = tanhf_(acc[GC][a][r]);
            float cv = ig * tg;
            if (MODE == GGNN_MODE_LSTM) cv = sigmoidf_(acc[GF < G ? GF : 0][a][r]) * cold[a][r] + cv;
            const float og = sigmoidf_(acc[GO][a][r]);
            cn[r] = cv;
            hn[r] = og * tanhf_(cv);
          }
          *reinterpret_cast<f32x4*>(A.c_out + m * C + n) = cn;
          *reinterpret_cast<f32x4*>(A.h_out + m * C + n) = hn;
        }
      }
    }
  }
#ifdef GX_VAR_CLOCK  // diagnostic build: (start, prologue end, end) in 10 ns ticks per wave, into c_out
  if (lane == 0 && MODE != GGNN_MODE_RAW) {
    __builtin_amdgcn_s_waitcnt(0);
    float* o = A.c_out + ((int64_t)blockIdx.x * GX_WAVES + wave) * 4;
    o[0] = (float)(tr_start & 0xffffff);
    o[1] = (float)(tr_pro - tr_start);
    o[2] = (float)(__builtin_amdgcn_s_memrealtime() - tr_start);
  }
#endif
}

}  // namespace ggnn

// Called by ggnn_lstm_epilogue (gates.hip) after argument validation, when w2_planes is given
// and GGNN_GEMM != fp32.
int ggnn_lstm_epilogue_x6(const ggnn_epilogue_args& A, hipStream_t s) {
  using namespace ggnn;
  const int G = A.n_gates;
  const bool wide = A.Ka == 196;
  const int64_t n_mt = (A.N + GX_BM - 1) / GX_BM;
  const int64_t n_wg_rows = (n_mt + GX_WAVES - 1) / GX_WAVES;  // workgroups along the nodes at one tile per wave
  // narrow slices (6 x 16 channels) while three wide ones leave more than half of the CUs without a workgroup
  const int NCT = 3 * n_wg_rows <= 128 ? 1 : 2;
  const int n_slices = 6 / NCT;
  // tiles per wave: 1 while one round of workgroups (<= 256) covers the nodes, else 2
  const int T = n_slices * n_wg_rows <= 256 ? 1 : 2;
  const int64_t n_ranges = (n_mt + GX_WAVES * T - 1) / (GX_WAVES * T);
  const int64_t nblk = 8 * n_slices * ((n_ranges + 7) / 8);
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  const dim3 grid((unsigned)nblk), block(GX_WAVES * 64);
#define GGNN_GX_LAUNCH2(G_, MODE_, KA_)                                                                           \
  do {                                                                                                            \
    if (NCT == 1) hipLaunchKernelGGL((gates_x6_kernel<G_, MODE_, KA_, 1, 1>), grid, block, 0, s, A, (int)n_ranges);      \
    else if (T == 1) hipLaunchKernelGGL((gates_x6_kernel<G_, MODE_, KA_, 1, 2>), grid, block, 0, s, A, (int)n_ranges);   \
    else hipLaunchKernelGGL((gates_x6_kernel<G_, MODE_, KA_, 2, 2>), grid, block, 0, s, A, (int)n_ranges);        \
  } while (0)
#define GGNN_GX_LAUNCH(G_, MODE_)              \
  do {                                         \
    if (wide) GGNN_GX_LAUNCH2(G_, MODE_, 196); \
    else GGNN_GX_LAUNCH2(G_, MODE_, 100);      \
  } while (0)
  if (A.mode == GGNN_MODE_LSTM) GGNN_GX_LAUNCH(4, GGNN_MODE_LSTM);
  else if (A.mode == GGNN_MODE_LSTM_H0) GGNN_GX_LAUNCH(3, GGNN_MODE_LSTM_H0);
  else if (G == 4) GGNN_GX_LAUNCH(4, GGNN_MODE_RAW);
  else if (G == 3) GGNN_GX_LAUNCH(3, GGNN_MODE_RAW);
  else GGNN_GX_LAUNCH(1, GGNN_MODE_RAW);
#undef GGNN_GX_LAUNCH
#undef GGNN_GX_LAUNCH2
  return launch_status();
}
