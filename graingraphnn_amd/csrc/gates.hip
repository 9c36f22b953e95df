// Gate GEMM + LSTM cell update on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
// For every node and gate g:
//   pre[g] = agg[:, g, 0:Ka] . W2[g]^T + skip[g]
// W2[g] packs, per incoming edge type, lin_l2.weight (periodGATconv.py:218) and the two
// rank-1 columns (lin_l2.bias x sum alpha, lin_edge.weight x sum alpha*a, :231-235); skip[g]
// (lin_skip summed over incoming edge types, :186, HeteroConv aggr='sum', + b_g) was written
// by ggnn_project.  The epilogue is the cell of heteropgclstm.py:111-146:
//   i = sig(pre_i); f = sig(pre_f); c' = f*c + i*tanh(pre_c); o = sig(pre_o); h' = o*tanh(c')
// (encoder: h = c = 0, so f is never needed and c' = i*tanh(pre_c)).
//
// Structure (same operand orientation as project.hip: weight tile = MFMA A operand, node
// tile = B operand, so a lane ends with 4 consecutive channels of one node; LDS rows are
// (kc + 2) floats apart => conflict-free ds_read_b32):
//   * a wave owns 16 nodes x 32 channels and keeps their accumulators for ALL gates in
//     registers (G x 2 tiles), so the LSTM update needs no exchange; three waves share a
//     16-node row group;
//   * ONE workgroup per CU and ONE round of workgroups: a workgroup takes NG = ceil(N / 16 /
//     256) row groups (20 000 joints -> 5 groups = 15 waves, 250 workgroups; 10 000 grains ->
//     3 groups = 9 waves, 209 workgroups).  Rounding the grid to the CU count matters more
//     than anything else here: 313 equal workgroups on 256 CUs take as long as 512;
//   * the K dimension is cut into passes (gate, <= 100 columns), fully unrolled; the 96 x kc
//     weight chunk and the row groups' 16 x kc node chunks are double-buffered in LDS: while
//     a pass is swept (50 MFMAs per wave) the next pass's chunks fly into registers and are
//     written to the other buffer after the sweep -- one workgroup barrier per pass.  All
//     global loads are unconditional (clamped indices; a load under `if` drags a wait to the
//     branch merge and serialises the stage).
#include "common.h"

namespace ggnn {

constexpr int GT_BM = 16;     // nodes per row group (three waves per group)
constexpr int GT_MAXG = 5;    // row groups per workgroup, upper bound (15 waves)
constexpr int GT_MING = 3;    // ... lower bound (sizes the weight staging registers: 5 pieces per lane)
constexpr int GT_KC = 100;    // K chunk per pass (Ka = 196 -> 100 + 96, Ka = 100 -> 100)
constexpr int GT_LD = GT_KC + 2;

template <int G, int MODE, int KA>
__global__ __launch_bounds__(GT_MAXG * 192, 1) void gates_kernel(const ggnn_epilogue_args A) {
  __shared__ float s_w[2][C * GT_LD];
  __shared__ float s_a[2][GT_MAXG][GT_BM * GT_LD];
  constexpr int NCH = (KA + GT_KC - 1) / GT_KC;  // K chunks per gate (2 for Ka = 196, 1 for 100)
  constexpr int NPASS = G * NCH;
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave / 3, third = wave - 3 * rg;  // row group, channel third
  const int ng = nthr / 192;                        // row groups in this workgroup
  const int64_t ld_agg = A.ld_agg;
  const int64_t m0 = ((int64_t)blockIdx.x * ng + rg) * GT_BM;  // first node of this row group
  const int lr = lane & 15, lq = lane >> 4;
  const int t192 = tid - rg * 192;  // index inside the row group's three waves

  // register stage of the next pass: the row group's node chunk (16 x 25 pieces over 192
  // lanes) and the weight chunk (96 x 25 pieces over the workgroup: 7 per lane at 2 groups)
  constexpr int NA = (GT_BM * (GT_KC / 4) + 191) / 192;                      // 3
  constexpr int NW = (C * (GT_KC / 4) + GT_MING * 192 - 1) / (GT_MING * 192);  // 7
  f32x4 ra[NA], rw[NW];
  auto load_pass = [&](int g, int kb, int kc) {
    const int nv = kc >> 2;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int idx = min(t192 + it * 192, GT_BM * nv - 1), r = idx / nv, c4 = idx - r * nv;
      const int64_t m = min(m0 + r, A.N - 1);
      ra[it] = *reinterpret_cast<const f32x4*>(A.agg + m * ld_agg + g * A.g_stride + kb + 4 * c4);
    }
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int idx = min(tid + it * nthr, C * nv - 1), r = idx / nv, c4 = idx - r * nv;
      rw[it] = *reinterpret_cast<const f32x4*>(A.w2 + ((int64_t)g * C + r) * KA + kb + 4 * c4);
    }
  };
  auto store_pass = [&](int buf, int kc) {
    const int nv = kc >> 2, ld = kc + 2;
    float* sa = s_a[buf][rg];
    float* sw = s_w[buf];
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int idx = t192 + it * 192, r = idx / nv, c4 = idx - r * nv;
      if (idx < GT_BM * nv) {
        float2* dst = reinterpret_cast<float2*>(&sa[r * ld + 4 * c4]);
        dst[0] = make_float2(ra[it].x, ra[it].y);
        dst[1] = make_float2(ra[it].z, ra[it].w);
      }
    }
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int idx = tid + it * nthr, r = idx / nv, c4 = idx - r * nv;
      if (idx < C * nv) {
        float2* dst = reinterpret_cast<float2*>(&sw[r * ld + 4 * c4]);
        dst[0] = make_float2(rw[it].x, rw[it].y);
        dst[1] = make_float2(rw[it].z, rw[it].w);
      }
    }
  };
  // chunk c of a gate covers columns [c * 100, min(KA, c * 100 + 100))
  auto kc_of = [](int c) { return (c + 1) * GT_KC <= KA ? GT_KC : KA - c * GT_KC; };

  // The accumulators START from the skip / bias term written by ggnn_project (pre = skip +
  // agg . W2^T): its loads fly together with the first chunk's, and the epilogue has nothing
  // left to fetch but c.
  const int64_t m = m0 + lr, m_c = min(m, A.N - 1);
  f32x4 acc[G][2];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int a = 0; a < 2; ++a)
      acc[g][a] = *reinterpret_cast<const f32x4*>(A.p_dst + m_c * A.ldp + A.s_off + g * C + third * 32 +
                                                  a * 16 + 4 * lq);
  f32x4 cold[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    cold[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (MODE == GGNN_MODE_LSTM)
      cold[a] = *reinterpret_cast<const f32x4*>(A.c_in + m_c * C + third * 32 + a * 16 + 4 * lq);
  }

  load_pass(0, 0, kc_of(0));
  store_pass(0, kc_of(0));
  __syncthreads();
  // passes are fully unrolled: gate index, chunk width, buffer and every stride are compile-time
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int p = g * NCH + c;
      const int kc = kc_of(c), ld = kc + 2;
      const bool has_next = p + 1 < NPASS;
      const int gn = (p + 1) / NCH, cn = (p + 1) % NCH;
      if (has_next) load_pass(gn, cn * GT_KC, kc_of(cn));  // in flight during the sweep below
      __builtin_amdgcn_sched_barrier(0);  // keep the loads above the sweep (hipcc sinks them otherwise)
      const float* pw = &s_w[p & 1][(third * 32 + lr) * ld + lq];
      const float* px = &s_a[p & 1][rg][lr * ld + lq];
#pragma unroll 5
      for (int k0 = 0; k0 < kc; k0 += 4) {
        const float xf = px[k0];
#pragma unroll
        for (int a = 0; a < 2; ++a)
          acc[g][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[a * 16 * ld + k0], xf, acc[g][a], 0, 0, 0);
      }
      if (has_next) store_pass((p + 1) & 1, kc_of(cn));  // the other buffer: last read in pass p - 1
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds channels n..n+3 (twice) of node m for every gate ----
  if (m >= A.N) return;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int n = third * 32 + a * 16 + 4 * lq;
    if (MODE == GGNN_MODE_RAW) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        *reinterpret_cast<f32x4*>(A.raw_out + m * (int64_t)(G * C) + g * C + n) = acc[g][a];
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

}  // namespace ggnn

int ggnn_lstm_epilogue_x6(const ggnn_epilogue_args* args, int n, hipStream_t s);

namespace ggnn {
// Argument checks of one gate-GEMM problem; normalises the packed agg layout.
static int check_epilogue(ggnn_epilogue_args& A) {
  if (!A.agg || !A.w2 || !A.p_dst || A.N <= 0) return GGNN_EINVAL;
  if (A.g_stride == 0 && A.ld_agg == 0) {  // packed layout
    A.g_stride = A.Ka;
    A.ld_agg = (int64_t)A.n_gates * A.Ka;
  }
  if (A.g_stride < A.Ka || (A.g_stride & 3) || (A.ld_agg & 3) || A.ld_agg < (int64_t)A.n_gates * A.g_stride)
    return GGNN_EINVAL;
  if (A.Ka != 196 && A.Ka != 100) return GGNN_EINVAL;  // two / one incoming edge types (packing.py)
  const int G = A.n_gates;
  if (A.s_off < 0 || (A.s_off & 3) || (A.ldp & 3) || A.s_off + (int64_t)G * C > A.ldp) return GGNN_EINVAL;
  if (!aligned16(A.agg) || !aligned16(A.w2) || !aligned16(A.p_dst)) return GGNN_EINVAL;
  if (A.w2_planes && !aligned16(A.w2_planes)) return GGNN_EINVAL;
  if (A.mode == GGNN_MODE_LSTM) {
    if (G != 4 || !A.c_in || !A.h_out || !A.c_out) return GGNN_EINVAL;
    if (!aligned16(A.c_in) || !aligned16(A.h_out) || !aligned16(A.c_out)) return GGNN_EINVAL;
  } else if (A.mode == GGNN_MODE_LSTM_H0) {
    if (G != 3 || !A.h_out || !A.c_out) return GGNN_EINVAL;
    if (!aligned16(A.h_out) || !aligned16(A.c_out)) return GGNN_EINVAL;
  } else if (A.mode == GGNN_MODE_RAW) {
    if (!A.raw_out || !aligned16(A.raw_out)) return GGNN_EINVAL;
    if (G != 4 && G != 3 && G != 1) return GGNN_EINVAL;
  } else {
    return GGNN_EINVAL;
  }
  return GGNN_OK;
}

// native fp32 MFMA kernel, one problem
static int launch_fp32(const ggnn_epilogue_args& A, hipStream_t s) {
  // row groups per workgroup: one round of workgroups, at most one per CU (2..5 groups)
  const int G = A.n_gates;
  const int64_t n16 = (A.N + GT_BM - 1) / GT_BM;
  int64_t ng = (n16 + num_cu() - 1) / num_cu();
  ng = ng < GT_MING ? GT_MING : (ng > GT_MAXG ? GT_MAXG : ng);
  const int64_t nblk = (n16 + ng - 1) / ng;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  const dim3 grid((unsigned)nblk), block((unsigned)(192 * ng));
  const bool wide = A.Ka == 196;
#define GGNN_GT_LAUNCH(G_, MODE_)                                                         \
  do {                                                                                    \
    if (wide) hipLaunchKernelGGL((gates_kernel<G_, MODE_, 196>), grid, block, 0, s, A);   \
    else hipLaunchKernelGGL((gates_kernel<G_, MODE_, 100>), grid, block, 0, s, A);        \
  } while (0)
  if (A.mode == GGNN_MODE_LSTM) GGNN_GT_LAUNCH(4, GGNN_MODE_LSTM);
  else if (A.mode == GGNN_MODE_LSTM_H0) GGNN_GT_LAUNCH(3, GGNN_MODE_LSTM_H0);
  else if (G == 4) GGNN_GT_LAUNCH(4, GGNN_MODE_RAW);
  else if (G == 3) GGNN_GT_LAUNCH(3, GGNN_MODE_RAW);
  else GGNN_GT_LAUNCH(1, GGNN_MODE_RAW);
#undef GGNN_GT_LAUNCH
  return launch_status();
}
}  // namespace ggnn

extern "C" int ggnn_lstm_epilogue_batch(const ggnn_epilogue_args* args, int n_problems, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_problems < 1 || n_problems > 4) return GGNN_EINVAL;
  ggnn_epilogue_args A[4];
  bool x6 = gemm_mode() == GGNN_GEMM_BF16X6;
  for (int k = 0; k < n_problems; ++k) {
    A[k] = args[k];
    const int rc = check_epilogue(A[k]);
    if (rc != GGNN_OK) return rc;
    if (A[k].mode != A[0].mode || A[k].n_gates != A[0].n_gates) return GGNN_EINVAL;
    x6 = x6 && A[k].w2_planes != nullptr;
  }
  hipStream_t s = (hipStream_t)stream;
  if (x6) return ggnn_lstm_epilogue_x6(A, n_problems, s);
  for (int k = 0; k < n_problems; ++k) {
    const int rc = launch_fp32(A[k], s);
    if (rc != GGNN_OK) return rc;
  }
  return GGNN_OK;
}

extern "C" int ggnn_lstm_epilogue(const ggnn_epilogue_args* args, ggnn_stream_t stream) {
  return ggnn_lstm_epilogue_batch(args, 1, stream);
}
