// Gate GEMM + LSTM cell update with the K = 192 / 96 aggregate columns on the bf16 matrix cores
// at fp32-equivalent numerics (exact 3 x bf16 split, 6 products per k-step: common.h) and the 4
// trailing rank-1 columns (sum alpha, sum alpha*a per incoming edge type) on one exact
// v_mfma_f32_16x16x4_f32.  Same contract as gates_kernel (gates.hip):
//   pre[g] = agg[:, g, 0:Ka] . W2[g]^T + skip[g];  LSTM cell of heteropgclstm.py:111-146.
//
// Structure (round 2; the round-1 kernel kept a 32-channel weight slice per workgroup, so three
// workgroups each re-read and re-split the same `agg` rows: 3 460 VALU against 584 MFMA
// instructions per wave, matrix pipe 30 % busy):
//   * the gates are INDEPENDENT GEMMs (gate g reads only agg[:, g, :]), so the unit of work is a
//     JOB = (16-node tile, gate): one wave reads the tile's 16 x Ka aggregate block of that gate
//     ONCE from memory, splits every fragment ONCE (48 VALU) and uses it against all 96 output
//     channels (36 MFMAs per fragment instead of 12).  5 000 jobs of 216 MFMAs at cfg3 spread
//     evenly over the 1 024 SIMDs (a whole tile x 4 gates per wave would not: 1 250 tiles);
//   * a workgroup (one per CU; 8 waves at G = 4: 2 tile slots x 4 gates, 6 waves at G = 3) =
//     SLOTS tile slots x G gates, up to three tiles per wave and pass; wave w = (gate w % G,
//     slot w / G), so with 5 tiles per workgroup the two waves of a SIMD carry 3 + 2 jobs.  Two
//     waves per SIMD leave each wave 256 VGPRs: 72 accumulators, a 48-register `agg` ring and 36
//     weight-staging registers fit without spilling (a 12-wave variant at 168 VGPRs spilled 200);
//   * the weights (442 KB as bf16 planes at G = 4, Ka = 196: more than the LDS) are STREAMED:
//     k-step slice s of all gates (G x 18 KB of planes in fragment order) sits in one half of a
//     double buffer while every wave stages its share of slice s + 1 through registers: fp32
//     fragments from L2 at the top of the k-step, split into planes and ds_write_b128 at its end,
//     one workgroup barrier per k-step.  (Not LDS-DMA: the wave's `agg` prefetch shares the vmcnt
//     queue, and the compiler only emits counted waits for plain loads.);
//   * `agg` fragments are prefetched two k-steps ahead (ring of two register sets per tile);
//   * the LSTM update needs all gates of a (node, channel): the waves leave their 16 x 96
//     pre-activation blocks in LDS (the weight buffers are free by then; rows padded to 100 floats:
//     conflict-free ds_write_b128 from the MFMA D layout), and after one barrier every thread
//     handles whole (node, 4-channel) quads: c_in is read and h / c written as contiguous
//     384-byte rows (the D layout would touch 64-byte pieces); these side loads are issued before
//     the last k-step (into the registers the weight staging and the `agg` ring no longer need).
//   * up to four problems (node types x models) per launch: workgroups are dealt to the problems
//     in proportion to their MFMA work (ggnn_lstm_epilogue_batch).
#include <algorithm>

#include "common.h"
#include "stamps.h"

namespace ggnn {

// Waves per workgroup / tiles per wave.  (Measured at G = 3, 20 000 joints, cold caches: 6 waves x 3
// tiles, 9 x 2, 12 x 2 all 36 us, 12 x 1 45 us -- the kernel is bound by the CU's load path, not by
// how its jobs are spread over the SIMDs; loading `agg` in pairs of k-steps was slower, 57 vs 50 us.)
constexpr int gw_waves(int G) { return G == 3 ? 6 : 8; }  // two per SIMD (G = 3: 2, 2, 1, 1)
constexpr int gw_tmax(int G) { return 3; }
constexpr int GW_BM = 16;      // nodes per tile
constexpr int GW_PRE_LD = 100; // floats per node row of the pre-activation exchange image
constexpr int GW_LDS_BYTES = 8 * 3 * GW_BM * GW_PRE_LD * 4;  // 153 600 (>= 2 x 4 x 18 KB of weight slices)
constexpr int GW_MAX_PROBLEMS = 4;

struct GateBatch {
  ggnn_epilogue_args a[GW_MAX_PROBLEMS];
  int wg_off[GW_MAX_PROBLEMS + 1];  // first workgroup of every problem
  int tpw[GW_MAX_PROBLEMS];         // tiles per workgroup
  int n;
};

// All global traffic of the kernel goes through raw buffer instructions: a wave-uniform resource
// (base in SGPRs) + a 32-bit lane offset + a wave-uniform scalar offset.  With plain pointers hipcc
// formed one 64-bit VGPR address per (weight piece, k-step) and per fragment load, hoisted all of
// them to the top of the kernel and spilled 200 registers.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);  // raw, no bounds clamp in range
}
__device__ __forceinline__ u32x4 bld128(rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ f32x4 bld128f(rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(f32x4, bld128(r, voff, soff));
}
__device__ __forceinline__ float bld32f(rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bst128f(rsrc_t r, uint32_t voff, uint32_t soff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)soff, 0);
}

// One workgroup = one pass over tiles [mt_lo, mt_hi), mt_hi - mt_lo <= SLOTS * GW_TMAX.
template <int G, int MODE, int KA>
__device__ __forceinline__ void gates_body(const ggnn_epilogue_args& A, int64_t mt_lo, int64_t mt_hi,
                                           u32x4* __restrict__ smem) {
  constexpr int KM = KA - 4;                 // columns on the bf16 path (192 / 96)
  constexpr int NKS = KM / 32;               // k-steps (6 / 3)
  constexpr int GW_WAVES = gw_waves(G), GW_THREADS = GW_WAVES * 64, GW_TMAX = gw_tmax(G);
  constexpr int SLOTS = GW_WAVES / G;        // tile slots
  constexpr int TP = SLOTS * GW_TMAX;        // tiles per workgroup, at most
  constexpr int NPIECE = G * 18;             // 1 KB pieces of one k-step slice: [gate][plane 3][column tile 6]
  constexpr int NFRAG = G * 6;               // (gate, column tile) fp32 fragments of one k-step slice, 2 KB each
  constexpr int FPW = (NFRAG + GW_WAVES - 1) / GW_WAVES;  // fragments a wave stages per k-step
  constexpr int SLICE = NPIECE * 64;         // u32x4 per slice
  static_assert(KM % 32 == 0 && GW_WAVES % G == 0, "shape");
  static_assert(2 * SLICE * 16 <= GW_LDS_BYTES && TP * G * GW_BM * GW_PRE_LD * 4 <= GW_LDS_BYTES, "LDS");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  GGNN_STAMP(0);
  const int lr = lane & 15, kq = lane >> 4;
  const int g = wave % G, slot = wave / G;
  const int64_t m_last = max(A.N - GW_BM, (int64_t)0);   // a ragged last tile slides back (identical duplicate stores)
  const int row_l = (int)min((int64_t)lr, A.N - 1);      // N < 16: clamp the lane's row instead
  float* __restrict__ pre = reinterpret_cast<float*>(smem);
  const int n_t = (int)(mt_hi - mt_lo);                  // tiles of this workgroup; tile j -> slot j % SLOTS, t = j / SLOTS
  const int nt_w = (n_t > slot ? 1 : 0) + (n_t > slot + SLOTS ? 1 : 0) + (GW_TMAX > 2 && n_t > slot + 2 * SLOTS ? 1 : 0);
  const int64_t row0 = min(mt_lo * GW_BM, m_last);       // first row of the workgroup: every offset below is relative to it
  const uint32_t ld_agg = (uint32_t)A.ld_agg, ldp = (uint32_t)A.ldp;

  const rsrc_t r_w2 = make_rsrc(A.w2);
  const rsrc_t r_agg = make_rsrc(A.agg + row0 * A.ld_agg), r_skip = make_rsrc(A.p_dst + row0 * A.ldp + A.s_off);
  const rsrc_t r_c = make_rsrc(MODE == GGNN_MODE_LSTM ? A.c_in + row0 * C : A.p_dst);
  // lane offsets (bytes) inside a tile; a tile's first row and the gate go into the scalar offset
  const uint32_t lo_agg = (uint32_t)row_l * ld_agg * 4u + 32u * kq;
  const uint32_t lo_tail = (uint32_t)row_l * ld_agg * 4u + 4u * kq;
  // absent tiles repeat the wave's first tile (or the workgroup's first): their loads hit L1, their MFMAs are skipped
  uint32_t so_agg[GW_TMAX];
#pragma unroll
  for (int t = 0; t < GW_TMAX; ++t) {
    const int j = t < nt_w ? slot + t * SLOTS : (nt_w ? slot : 0);
    const uint32_t dm = (uint32_t)(min((mt_lo + j) * GW_BM, m_last) - row0);
    so_agg[t] = (dm * ld_agg + (uint32_t)(g * A.g_stride)) * 4u;
  }

  // Weight staging.  The weights arrive as fp32 MFMA A fragments in fragment order
  // (ggnn_epilogue_args.w2_planes: [g][ks][ct][half][lane][4 floats], 1 KB of contiguous memory per
  // load instruction) -- lane (i = l & 15, kq = l >> 4) of fragment (gate gq, column tile ct, k-step
  // ks) holds w2[gq][16 ct + i][32 ks + 8 kq ..+7] --, are split into the three bf16 planes here
  // (44 VALU per fragment and lane) and stored lane-linearly at [gq][plane][ct][lane]: 4 bytes per
  // weight travel from L2 instead of the 6 of pre-split planes (every CU streams all weights: the
  // XCD's L2 bandwidth, not the matrix pipe, is what this stream costs).
  // Surplus slots repeat the last fragment (same bytes to the same place).
  const uint32_t lo_w = 16u * lane;
  const rsrc_t r_wf = make_rsrc(A.w2_planes);
  f32x4 stage[FPW][2];
  auto stage_load = [&](int ks) {
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const int f = min(wave + i * GW_WAVES, NFRAG - 1), gq = f / 6, ct = f - gq * 6;
      const uint32_t so = (uint32_t)((((gq * NKS + ks) * 6 + ct) * 2) * 1024);
      stage[i][0] = bld128f(r_wf, lo_w, so);
      stage[i][1] = bld128f(r_wf, lo_w, so + 1024u);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const int f = min(wave + i * GW_WAVES, NFRAG - 1), gq = f / 6, ct = f - gq * 6;
      u32x4 pl[3];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint32_t q0, q1, q2;
        split_bf16x3(stage[i][d >> 1][2 * (d & 1)], stage[i][d >> 1][2 * (d & 1) + 1], q0, q1, q2);
        pl[0][d] = q0;
        pl[1][d] = q1;
        pl[2][d] = q2;
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) smem[buf * SLICE + ((gq * 3 + q) * 6 + ct) * 64 + lane] = pl[q];
    }
  };
  f32x4 raw[2][GW_TMAX][2];
  auto agg_load = [&](int ks, int t) {
    raw[ks & 1][t][0] = bld128f(r_agg, lo_agg, so_agg[t] + 128u * ks);
    raw[ks & 1][t][1] = bld128f(r_agg, lo_agg, so_agg[t] + 128u * ks + 16u);
  };

  // ---- prologue: slice 0 -> LDS; the first two k-steps of `agg` and the tail operands in flight ----
  stage_load(0);
  float xt[GW_TMAX], wt[6];
  f32x4 acc[GW_TMAX][6];
#pragma unroll
  for (int t = 0; t < GW_TMAX; ++t) {
    agg_load(0, t);
    if (NKS > 1) agg_load(1, t);
  }
#pragma unroll
  for (int t = 0; t < GW_TMAX; ++t) {
    xt[t] = bld32f(r_agg, lo_tail, so_agg[t] + KM * 4u);
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) acc[t][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // weight side of the exact fp32 tail (columns KM .. KM+3), 16x16x4 fragment layout
#pragma unroll
  for (int ct = 0; ct < 6; ++ct)
    wt[ct] = bld32f(r_w2, ((uint32_t)lr * KA + kq) * 4u, (uint32_t)(((g * C + ct * 16) * KA + KM) * 4));
  GGNN_STAMP(1);
  stage_store(0);
  __syncthreads();
  GGNN_STAMP(2);

  // the LSTM phase: this thread's (node, 4-channel) quads, as byte offsets from row0
  constexpr int NQ = (TP * GW_BM * 24 + GW_THREADS - 1) / GW_THREADS;  // 5 (G = 4), 6 (G = 3)
  f32x4 cold[NQ], skipv[NQ][G];
  uint32_t qo[NQ], qs[NQ];   // ((row - row0) * 96 + 4 c4) * 4, ((row - row0) * ldp + 4 c4) * 4
  int qoff[NQ];
  bool qok[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = tid + i * GW_THREADS;
    const int j = q / (GW_BM * 24), rem = q - j * (GW_BM * 24), node = rem / 24, c4 = rem - node * 24;
    qok[i] = j < n_t;
    const int jj = qok[i] ? j : 0;
    const uint32_t row = (uint32_t)(min((mt_lo + jj) * GW_BM, m_last) - row0 + min((int64_t)node, A.N - 1));
    qo[i] = (row * C + 4 * c4) * 4u;
    qs[i] = (row * ldp + 4 * c4) * 4u;
    qoff[i] = (jj * G * GW_BM + node) * GW_PRE_LD + 4 * c4;
  }

  constexpr int NQ_EARLY = NQ < 2 ? NQ : 2;
  auto side_load = [&](int i0, int i1) {
#pragma unroll
    for (int i = i0; i < i1; ++i) {
#pragma unroll
      for (int g2 = 0; g2 < G; ++g2) skipv[i][g2] = bld128f(r_skip, qs[i], (uint32_t)(g2 * C * 4));
      if constexpr (MODE == GGNN_MODE_LSTM) cold[i] = bld128f(r_c, qo[i], 0);
    }
  };

#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (ks + 1 < NKS) stage_load(ks + 1);
    // side inputs of the LSTM phase: the first quads' behind the last k-step, the rest behind the
    // tail / exchange (all at once they would fill the CU's load queue in front of the last MFMAs)
    if (ks == NKS - 1 && MODE != GGNN_MODE_RAW) side_load(0, NQ_EARLY);
    const u32x4* pw = smem + (ks & 1) * SLICE + g * (18 * 64) + lane;
#pragma unroll
    for (int t = 0; t < GW_TMAX; ++t) {
      // split this k-step's fragment, then reuse its registers for the fragment two k-steps ahead
      u32x4 xb[3];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint32_t q0, q1, q2;
        split_bf16x3(raw[ks & 1][t][d >> 1][2 * (d & 1)], raw[ks & 1][t][d >> 1][2 * (d & 1) + 1], q0, q1, q2);
        xb[0][d] = q0;
        xb[1][d] = q1;
        xb[2][d] = q2;
      }
      if (ks + 2 < NKS) agg_load(ks + 2, t);
      if (t < nt_w) {
        // the three weight fragments of column tile ct + 1 are read while the six MFMAs of ct run
        u32x4 wf[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) wf[0][q] = pw[(q * 6) * 64];
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // DS reads of ct = 0, then [reads ct + 1 | MFMAs ct] ...
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) {
          if (ct + 1 < 6) {
#pragma unroll
            for (int q = 0; q < 3; ++q) wf[(ct + 1) & 1][q] = pw[(q * 6 + ct + 1) * 64];
          }
          acc[t][ct] = mfma_x6(wf[ct & 1], xb, acc[t][ct]);
          if (ct + 1 < 6) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // DS read
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                  // MFMA
        }
      }
    }
    GGNN_STAMP(3 + 2 * ks);
    if (ks + 1 < NKS) stage_store((ks + 1) & 1);
    __syncthreads();  // slice ks + 1 complete and visible; every wave is done with slice ks
    GGNN_STAMP(4 + 2 * ks);
  }

  if constexpr (MODE != GGNN_MODE_RAW) side_load(NQ_EARLY, NQ);
  // ---- exact fp32 tail, then the pre-activation blocks -> LDS (the weight buffers are free) ----
#pragma unroll
  for (int t = 0; t < GW_TMAX; ++t) {
    if (t < nt_w) {
      const int j = slot + t * SLOTS;
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) {
        acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[ct], xt[t], acc[t][ct], 0, 0, 0);
        *reinterpret_cast<f32x4*>(&pre[((j * G + g) * GW_BM + lr) * GW_PRE_LD + ct * 16 + 4 * kq]) = acc[t][ct];
      }
    }
  }
  __syncthreads();
  GGNN_STAMP(15);

  // ---- LSTM phase: whole (node, 4-channel) quads per thread, contiguous rows in memory ----
  const rsrc_t r_h = make_rsrc(MODE == GGNN_MODE_RAW ? A.raw_out + row0 * (G * C) : A.h_out + row0 * C);
  const rsrc_t r_co = make_rsrc(MODE == GGNN_MODE_RAW ? A.raw_out : A.c_out + row0 * C);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    if (!qok[i]) continue;
    f32x4 p[G];
#pragma unroll
    for (int g2 = 0; g2 < G; ++g2) {
      if constexpr (MODE == GGNN_MODE_RAW) skipv[i][g2] = bld128f(r_skip, qs[i], (uint32_t)(g2 * C * 4));  // parity-test mode
      p[g2] = *reinterpret_cast<const f32x4*>(&pre[qoff[i] + g2 * GW_BM * GW_PRE_LD]) + skipv[i][g2];
    }
    if constexpr (MODE == GGNN_MODE_RAW) {
      const uint32_t row = qo[i] / (C * 4u), cb = qo[i] - row * (C * 4u);
#pragma unroll
      for (int g2 = 0; g2 < G; ++g2) bst128f(r_h, row * (G * C * 4u) + cb, (uint32_t)(g2 * C * 4), p[g2]);
    } else {
      constexpr int GI = 0, GF = 1, GC = (MODE == GGNN_MODE_LSTM) ? 2 : 1, GO = (MODE == GGNN_MODE_LSTM) ? 3 : 2;
      f32x4 hn, cn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = sigmoidf_(p[GI][r]);
        const float tg = tanhf_(p[GC][r]);
        float cv = ig * tg;
        if constexpr (MODE == GGNN_MODE_LSTM) cv = sigmoidf_(p[GF][r]) * cold[i][r] + cv;
        const float og = sigmoidf_(p[GO][r]);
        cn[r] = cv;
        hn[r] = og * tanhf_(cv);
      }
      bst128f(r_co, qo[i], 0, cn);
      bst128f(r_h, qo[i], 0, hn);
    }
  }
  GGNN_STAMP(16);
}

template <int G, int MODE>
__global__ __launch_bounds__(gw_waves(G) * 64, 1) void gates_x6_kernel(const GateBatch B) {
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[GW_LDS_BYTES];
  u32x4* smem = reinterpret_cast<u32x4*>(s_raw);
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const ggnn_epilogue_args& A = B.a[k];
  const int64_t n_mt = (A.N + GW_BM - 1) / GW_BM;
  const int64_t mt_lo = (int64_t)((int)blockIdx.x - B.wg_off[k]) * B.tpw[k];
  const int64_t mt_hi = min(n_mt, mt_lo + B.tpw[k]);
  if (mt_lo >= mt_hi) return;
  if (A.Ka == 196) gates_body<G, MODE, 196>(A, mt_lo, mt_hi, smem);
  else gates_body<G, MODE, 100>(A, mt_lo, mt_hi, smem);
}

}  // namespace ggnn

// Called by ggnn_lstm_epilogue_batch (gates.hip) after argument validation, when every problem has
// w2_planes and GGNN_GEMM != fp32.  All problems share mode and n_gates.
int ggnn_lstm_epilogue_x6(const ggnn_epilogue_args* args, int n, hipStream_t s) {
  using namespace ggnn;
  GateBatch B;
  B.n = n;
  // Workgroups are dealt in proportion to the MFMA work: a workgroup takes `tpw` tiles of one
  // problem with tpw * k-steps ~ W, the smallest W for which one round of workgroups (one per CU)
  // covers everything.
  const int ncu = num_cu();
  int64_t n_mt[GW_MAX_PROBLEMS], nks[GW_MAX_PROBLEMS];
  for (int k = 0; k < n; ++k) {
    n_mt[k] = (args[k].N + GW_BM - 1) / GW_BM;
    nks[k] = (args[k].Ka - 4) / 32;
  }
  const int G = args[0].n_gates, mode = args[0].mode;
  const int64_t tp_max = (int64_t)(gw_waves(G) / G) * gw_tmax(G);  // tiles a workgroup can take
  int64_t total = 0;
  for (int W = 6;; W += 3) {
    total = 0;
    bool capped = true;
    for (int k = 0; k < n; ++k) {
      const int64_t tpw = std::min(tp_max, std::max<int64_t>(1, W / nks[k]));
      capped = capped && tpw == tp_max;
      B.tpw[k] = (int)tpw;
      total += (n_mt[k] + tpw - 1) / tpw;
    }
    if (total <= ncu || capped) break;  // more than one round of workgroups: full workgroups, dealt by the hardware
  }
  if (total >= INT32_MAX) return GGNN_EINVAL;
  B.wg_off[0] = 0;
  for (int k = 0; k < GW_MAX_PROBLEMS; ++k) {
    B.a[k] = args[k < n ? k : 0];
    if (k < n) B.wg_off[k + 1] = B.wg_off[k] + (int)((n_mt[k] + B.tpw[k] - 1) / B.tpw[k]);
    else {
      B.wg_off[k + 1] = B.wg_off[k];
      B.tpw[k] = 1;
    }
  }
  const dim3 grid((unsigned)total), block(gw_waves(G) * 64);
  if (mode == GGNN_MODE_LSTM) hipLaunchKernelGGL((gates_x6_kernel<4, GGNN_MODE_LSTM>), grid, block, 0, s, B);
  else if (mode == GGNN_MODE_LSTM_H0) hipLaunchKernelGGL((gates_x6_kernel<3, GGNN_MODE_LSTM_H0>), grid, block, 0, s, B);
  else if (G == 4) hipLaunchKernelGGL((gates_x6_kernel<4, GGNN_MODE_RAW>), grid, block, 0, s, B);
  else if (G == 3) hipLaunchKernelGGL((gates_x6_kernel<3, GGNN_MODE_RAW>), grid, block, 0, s, B);
  else hipLaunchKernelGGL((gates_x6_kernel<1, GGNN_MODE_RAW>), grid, block, 0, s, B);
  return launch_status();
}
