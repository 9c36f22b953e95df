// Gate GEMM + LSTM cell update on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
// For a block of 32 nodes and every gate g:
//   pre[g] = agg[:, g, 0:Ka] . W2[g]^T + skip[g]
// W2[g] packs, per incoming edge type, lin_l2.weight (periodGATconv.py:218) and the two
// rank-1 columns (lin_l2.bias x sum alpha, lin_edge.weight x sum alpha*a, :231-235); skip[g]
// (lin_skip summed over incoming edge types, :186, HeteroConv aggr='sum', + b_g) was written
// by ggnn_project.  The epilogue is the cell of heteropgclstm.py:111-146:
//   i = sig(pre_i); f = sig(pre_f); c' = f*c + i*tanh(pre_c); o = sig(pre_o); h' = o*tanh(c')
// (encoder: h = c = 0, so f is never needed and c' = i*tanh(pre_c)).
//
// Same operand orientation and LDS layout as project.hip: the weight tile is the MFMA A
// operand, the node tile the B operand, each lane ends up with 4 consecutive channels of one
// node, rows in LDS are (kc + 2) floats apart (2 x odd => conflict-free ds_read_b32).
#include "common.h"

namespace ggnn {

constexpr int GT_BM = 32;    // nodes per workgroup (52 KB of LDS -> three workgroups per CU)
constexpr int GT_KC = 100;   // K chunk staged per pass (Ka = 196 -> 100 + 96, Ka = 100 -> 100)
constexpr int GT_LD = GT_KC + 2;

template <int G, int MODE>
__global__ __launch_bounds__(256, 3) void gates_kernel(const ggnn_epilogue_args A) {
  __shared__ float s_a[GT_BM * GT_LD];
  __shared__ float s_w[C * GT_LD];

  const int tid = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.x * GT_BM;
  const int Ka = A.Ka;
  const int64_t ld_agg = (int64_t)G * Ka;

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;  // wave tile: 16 nodes x 48 channels
  const int lr = lane & 15, lq = lane >> 4;

  f32x4 acc[G][3];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int a = 0; a < 3; ++a) acc[g][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int g = 0; g < G; ++g) {
    for (int kb = 0; kb < Ka; kb += GT_KC) {
      const int kc = min(GT_KC, Ka - kb);  // multiple of 4
      const int ld = kc + 2;
      const int nv = kc >> 2;
      __syncthreads();  // previous pass has finished reading LDS
      for (int idx = tid; idx < GT_BM * nv; idx += 256) {
        const int r = idx / nv, c4 = idx - r * nv;
        const int64_t m = m0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < A.N) v = *reinterpret_cast<const f32x4*>(A.agg + m * ld_agg + g * Ka + kb + 4 * c4);
        float2* dst = reinterpret_cast<float2*>(&s_a[r * ld + 4 * c4]);
        dst[0] = make_float2(v.x, v.y);
        dst[1] = make_float2(v.z, v.w);
      }
      for (int idx = tid; idx < C * nv; idx += 256) {
        const int r = idx / nv, c4 = idx - r * nv;
        const f32x4 v =
            *reinterpret_cast<const f32x4*>(A.w2 + ((int64_t)g * C + r) * Ka + kb + 4 * c4);
        float2* dst = reinterpret_cast<float2*>(&s_w[r * ld + 4 * c4]);
        dst[0] = make_float2(v.x, v.y);
        dst[1] = make_float2(v.z, v.w);
      }
      __syncthreads();
      const float* pw = &s_w[(wn * 48 + lr) * ld + lq];
      const float* px = &s_a[(wm * 16 + lr) * ld + lq];
#pragma unroll 2
      for (int k0 = 0; k0 < kc; k0 += 4) {
        float wf[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) wf[a] = pw[a * 16 * ld + k0];
        const float xf = px[k0];
#pragma unroll
        for (int a = 0; a < 3; ++a)
          acc[g][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[a], xf, acc[g][a], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds channels n..n+3 of node m for every gate ----
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int n = wn * 48 + a * 16 + 4 * lq;
    {
      const int64_t m = m0 + wm * 16 + lr;
      if (m >= A.N) continue;
      const float* srow = A.p_dst + m * A.ldp + A.s_off + n;
      f32x4 pre[G];
#pragma unroll
      for (int g = 0; g < G; ++g)
        pre[g] = acc[g][a] + *reinterpret_cast<const f32x4*>(srow + g * C);
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
  const int64_t nblk = (A.N + GT_BM - 1) / GT_BM;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  const dim3 grid((unsigned)nblk), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (A.mode == GGNN_MODE_LSTM) {
    if (G != 4 || !A.c_in || !A.h_out || !A.c_out) return GGNN_EINVAL;
    if (!aligned16(A.c_in) || !aligned16(A.h_out) || !aligned16(A.c_out)) return GGNN_EINVAL;
    hipLaunchKernelGGL((gates_kernel<4, GGNN_MODE_LSTM>), grid, block, 0, s, A);
  } else if (A.mode == GGNN_MODE_LSTM_H0) {
    if (G != 3 || !A.h_out || !A.c_out) return GGNN_EINVAL;
    if (!aligned16(A.h_out) || !aligned16(A.c_out)) return GGNN_EINVAL;
    hipLaunchKernelGGL((gates_kernel<3, GGNN_MODE_LSTM_H0>), grid, block, 0, s, A);
  } else if (A.mode == GGNN_MODE_RAW) {
    if (!A.raw_out || !aligned16(A.raw_out)) return GGNN_EINVAL;
    if (G == 4)
      hipLaunchKernelGGL((gates_kernel<4, GGNN_MODE_RAW>), grid, block, 0, s, A);
    else if (G == 3)
      hipLaunchKernelGGL((gates_kernel<3, GGNN_MODE_RAW>), grid, block, 0, s, A);
    else if (G == 1)
      hipLaunchKernelGGL((gates_kernel<1, GGNN_MODE_RAW>), grid, block, 0, s, A);
    else
      return GGNN_EINVAL;
  } else {
    return GGNN_EINVAL;
  }
  return launch_status();
}
