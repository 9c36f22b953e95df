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
//   * a workgroup of W waves (4..8, chosen so that one round of workgroups covers the chip)
//     owns 16 x W nodes; every wave keeps the accumulators of its own 16 nodes x 96 channels
//     for ALL gates in registers (G x 6 tiles), so the LSTM update needs no exchange;
//   * the K dimension is cut into passes (gate, <=100 columns); the 96 x kc weight chunk of a
//     pass is shared through a double-buffered LDS tile, the 16 x kc node chunk is
//     wave-private.  While a pass is swept (150 MFMAs per wave) the next pass's chunks are in
//     flight into registers and are written to LDS right after the sweep: one workgroup
//     barrier per pass, no exposed global latency after the prologue.
#include "common.h"

namespace ggnn {

constexpr int GT_BM = 16;    // nodes per wave
constexpr int GT_MAXW = 8;   // waves per workgroup (upper bound)
constexpr int GT_MINW = 4;   // ... and lower bound (sizes the weight staging registers)
constexpr int GT_KC = 100;   // K chunk per pass (Ka = 196 -> 100 + 96, Ka = 100 -> 100)
constexpr int GT_LD = GT_KC + 2;
constexpr int GT_NUM_CU = 256;

template <int G, int MODE, int KA>
__global__ __launch_bounds__(GT_MAXW * 64, 1) void gates_kernel(const ggnn_epilogue_args A) {
  __shared__ float s_w[2][C * GT_LD];
  __shared__ float s_a[GT_MAXW][GT_BM * GT_LD];
  constexpr int NCH = (KA + GT_KC - 1) / GT_KC;  // K chunks per gate (2 for Ka = 196, 1 for 100)
  constexpr int NPASS = G * NCH;
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int64_t ld_agg = (int64_t)G * KA;
  const int64_t m0 = ((int64_t)blockIdx.x * (nthr >> 6) + wave) * GT_BM;  // first node of this wave
  const int lr = lane & 15, lq = lane >> 4;
  float* sa = s_a[wave];

  // Register stage of the next pass: node chunk (<= 7 pieces per lane) and weight chunk
  // (96 x 25 pieces over the workgroup: <= 10 per lane at 4 waves, 5 at 8).  Every load is
  // UNCONDITIONAL (indices clamped into range; only the LDS writes are predicated), so all of
  // them are in flight together -- a load under `if` drags a wait to the branch merge.
  constexpr int NA = (GT_BM * (GT_KC / 4) + 63) / 64;  // 7
  constexpr int NW = (C * (GT_KC / 4) + GT_MINW * 64 - 1) / (GT_MINW * 64);  // 10
  f32x4 ra[NA], rw[NW];
  auto load_pass = [&](int g, int kb, int kc) {
    const int nv = kc >> 2;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int idx = min(lane + it * 64, GT_BM * nv - 1), r = idx / nv, c4 = idx - r * nv;
      const int64_t m = min(m0 + r, A.N - 1);
      ra[it] = *reinterpret_cast<const f32x4*>(A.agg + m * ld_agg + g * KA + kb + 4 * c4);
    }
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int idx = min(tid + it * nthr, C * nv - 1), r = idx / nv, c4 = idx - r * nv;
      rw[it] = *reinterpret_cast<const f32x4*>(A.w2 + ((int64_t)g * C + r) * KA + kb + 4 * c4);
    }
  };
  auto store_pass = [&](int p, int kc) {
    const int nv = kc >> 2, ld = kc + 2;
    float* sw = s_w[p & 1];
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int idx = lane + it * 64, r = idx / nv, c4 = idx - r * nv;
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

  f32x4 acc[G][6];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[g][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_pass(0, 0, kc_of(0));
  store_pass(0, kc_of(0));
  __syncthreads();
  // passes are fully unrolled: gate index, chunk width and every stride are compile-time
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int p = g * NCH + c;
      const int kc = kc_of(c), ld = kc + 2;
      const bool has_next = p + 1 < NPASS;
      const int gn = (p + 1) / NCH, cn = (p + 1) % NCH;
      if (has_next) load_pass(gn, cn * GT_KC, kc_of(cn));  // in flight during the sweep below
      const float* pw = &s_w[p & 1][lr * ld + lq];
      const float* px = &sa[lr * ld + lq];
#ifndef GT_VAR_NO_MFMA
#pragma unroll 5
      for (int k0 = 0; k0 < kc; k0 += 4) {
        const float xf = px[k0];
#pragma unroll
        for (int a = 0; a < 6; ++a)
          acc[g][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[a * 16 * ld + k0], xf, acc[g][a], 0, 0, 0);
      }
#else
      acc[g][0][0] += pw[0] + px[0];
#endif
      if (has_next) store_pass(p + 1, kc_of(cn));  // node chunk: same wave, in-order LDS; weights: other buffer
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds channels n..n+3 (six times) of node m for every gate ----
  const int64_t m = m0 + lr;
  if (m >= A.N) return;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int n = a * 16 + 4 * lq;
    const float* srow = A.p_dst + m * A.ldp + A.s_off + n;
    f32x4 pre[G];
#pragma unroll
    for (int g = 0; g < G; ++g) pre[g] = acc[g][a] + *reinterpret_cast<const f32x4*>(srow + g * C);
    if (MODE == GGNN_MODE_RAW) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        *reinterpret_cast<f32x4*>(A.raw_out + m * (int64_t)(G * C) + g * C + n) = pre[g];
    } else {
      constexpr int GI = 0, GF = 1, GC = (MODE == GGNN_MODE_LSTM) ? 2 : 1,
                    GO = (MODE == GGNN_MODE_LSTM) ? 3 : 2;
      f32x4 cold = {0.f, 0.f, 0.f, 0.f};
      if (MODE == GGNN_MODE_LSTM) cold = *reinterpret_cast<const f32x4*>(A.c_in + m * C + n);
      f32x4 hn, cn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#ifdef GT_VAR_NO_LSTM
        cn[r] = pre[GI][r] + pre[GC][r] + cold[r];
        hn[r] = pre[GO][r] + pre[GF < G ? GF : 0][r];
        continue;
#endif
        const float ig = sigmoidf_(pre[GI][r]);
        const float tg = tanhf(pre[GC][r]);
        float cv = ig * tg;
        if (MODE == GGNN_MODE_LSTM) cv = sigmoidf_(pre[GF < G ? GF : 0][r]) * cold[r] + cv;
        const float og = sigmoidf_(pre[GO][r]);
        cn[r] = cv;
        hn[r] = og * tanhf(cv);
      }
      *reinterpret_cast<f32x4*>(A.c_out + m * C + n) = cn;
      *reinterpret_cast<f32x4*>(A.h_out + m * C + n) = hn;
    }
  }
}

}  // namespace ggnn

extern "C" int ggnn_lstm_epilogue(const ggnn_epilogue_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  const ggnn_epilogue_args& A = *args;
  if (!A.agg || !A.w2 || !A.p_dst || A.N <= 0) return GGNN_EINVAL;
  if (A.Ka < 4 || (A.Ka & 3) || A.Ka > 2 * GT_KC) return GGNN_EINVAL;
  const int G = A.n_gates;
  if (A.s_off < 0 || (A.s_off & 3) || (A.ldp & 3) || A.s_off + (int64_t)G * C > A.ldp) return GGNN_EINVAL;
  if (!aligned16(A.agg) || !aligned16(A.w2) || !aligned16(A.p_dst)) return GGNN_EINVAL;
  // waves per workgroup: as few as cover the chip in one round of workgroups (4..8)
  const int64_t n16 = (A.N + GT_BM - 1) / GT_BM;
  int64_t W = (n16 + GT_NUM_CU - 1) / GT_NUM_CU;
  W = W < GT_MINW ? GT_MINW : (W > GT_MAXW ? GT_MAXW : W);
#ifdef GT_VAR_W
  W = GT_VAR_W;
#endif
  const int64_t nblk = (n16 + W - 1) / W;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  const dim3 grid((unsigned)nblk), block((unsigned)(64 * W));
  hipStream_t s = (hipStream_t)stream;
  if (A.Ka != 196 && A.Ka != 100) return GGNN_EINVAL;  // two / one incoming edge types (packing.py)
  const bool wide = A.Ka == 196;
#define GGNN_GT_LAUNCH(G_, MODE_)                                                         \
  do {                                                                                    \
    if (wide) hipLaunchKernelGGL((gates_kernel<G_, MODE_, 196>), grid, block, 0, s, A);   \
    else hipLaunchKernelGGL((gates_kernel<G_, MODE_, 100>), grid, block, 0, s, A);        \
  } while (0)
  if (A.mode == GGNN_MODE_LSTM) {
    if (G != 4 || !A.c_in || !A.h_out || !A.c_out) return GGNN_EINVAL;
    if (!aligned16(A.c_in) || !aligned16(A.h_out) || !aligned16(A.c_out)) return GGNN_EINVAL;
    GGNN_GT_LAUNCH(4, GGNN_MODE_LSTM);
  } else if (A.mode == GGNN_MODE_LSTM_H0) {
    if (G != 3 || !A.h_out || !A.c_out) return GGNN_EINVAL;
    if (!aligned16(A.h_out) || !aligned16(A.c_out)) return GGNN_EINVAL;
    GGNN_GT_LAUNCH(3, GGNN_MODE_LSTM_H0);
  } else if (A.mode == GGNN_MODE_RAW) {
    if (!A.raw_out || !aligned16(A.raw_out)) return GGNN_EINVAL;
    if (G == 4) GGNN_GT_LAUNCH(4, GGNN_MODE_RAW);
    else if (G == 3) GGNN_GT_LAUNCH(3, GGNN_MODE_RAW);
    else if (G == 1) GGNN_GT_LAUNCH(1, GGNN_MODE_RAW);
    else return GGNN_EINVAL;
  } else {
    return GGNN_EINVAL;
  }
#undef GGNN_GT_LAUNCH
  return launch_status();
}
