// LSTM update of HeteroPGCLSTM.forward (heteropgclstm.py:140-183) for the TRAINING path: a forward
// that keeps what the backward needs, and the backward.  (The inference path fuses the update into
// the gate GEMM, gates*.hip / enc_cell.hip; those kernels keep nothing.)
//
//   z_g = gemm_g + skip_g                      g in (i, f, c, o), or (i, c, o) with zero state (encoder)
//   c'  = sig(z_f) c + sig(z_i) tanh(z_c)      h' = sig(z_o) tanh(c')
//
// forward : z [G, N, 96] holds the gate GEMM's output on entry and the pre-activations z on exit
//           (saved for the backward); the skip rows come from the projection, p_dst[n, s_off + g*96 ..].
// backward: from (z, c, c') and the gradients of h' and c' (either may be absent = zero) the gradient
//           of every z_g -- written twice: as g_z [G, N, 96] for the gate GEMM's backward and into the
//           skip columns of the projection's gradient -- and of c.
// Pointwise, one float4 per thread: HBM-bound (forward 4G+3 rows of 384 B per node, backward 5G+4).
#include "common.h"

namespace ggnn {

constexpr int LT_Q = C / 4;  // float4s per row

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// The library's expf / tanhf (1-2 ulp), not the inference path's hardware approximations: gradients are sums
// with cancellation over all nodes, and these kernels are HBM-bound anyway.
__device__ __forceinline__ float sig(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float at(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

struct LstmTrainBatch {
  ggnn_lstm_train_problem p[GGNN_LSTM_TRAIN_MAX];
  int blk_off[GGNN_LSTM_TRAIN_MAX + 1];   // first workgroup of every problem
  int n;
};

template <int G>
__global__ __launch_bounds__(256) void lstm_train_fwd_kernel(const LstmTrainBatch B) {
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.blk_off[k + 1]) ++k;
  const ggnn_lstm_train_problem& P = B.p[k];
  const int64_t N = P.N, ldp = P.ldp;
  const int64_t t = (int64_t)((int)blockIdx.x - B.blk_off[k]) * 256 + threadIdx.x;
  if (t >= N * LT_Q) return;
  const int64_t n = t / LT_Q;
  const int q = (int)(t - n * LT_Q) * 4;
  constexpr int GI = 0, GF = 1, GC = G - 2, GO = G - 1;
  float4 zz[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float* zp = P.z + ((int64_t)g * N + n) * C + q;
    const float4 a = ld4(zp), s = ld4(P.p_dst + n * ldp + P.s_off + g * C + q);
    zz[g] = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
    st4(zp, zz[g]);
  }
  float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
  if (G == 4) c = ld4(P.c_in + n * C + q);
  float cn[4], hn[4];
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) {
    float v = sig(at(zz[GI], k2)) * tanhf(at(zz[GC], k2));
    if (G == 4) v += sig(at(zz[GF], k2)) * at(c, k2);
    cn[k2] = v;
    hn[k2] = sig(at(zz[GO], k2)) * tanhf(v);
  }
  st4(P.c_out + n * C + q, make_float4(cn[0], cn[1], cn[2], cn[3]));
  st4(P.h_out + n * C + q, make_float4(hn[0], hn[1], hn[2], hn[3]));
}

template <int G>
__global__ __launch_bounds__(256) void lstm_train_bwd_kernel(const LstmTrainBatch B) {
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.blk_off[k + 1]) ++k;
  const ggnn_lstm_train_problem& P = B.p[k];
  const int64_t N = P.N, ldp = P.ldp;
  const int64_t t = (int64_t)((int)blockIdx.x - B.blk_off[k]) * 256 + threadIdx.x;
  if (t >= N * LT_Q) return;
  const int64_t n = t / LT_Q;
  const int q = (int)(t - n * LT_Q) * 4;
  constexpr int GI = 0, GF = 1, GC = G - 2, GO = G - 1;
  float4 zz[G];
#pragma unroll
  for (int g = 0; g < G; ++g) zz[g] = ld4(P.z + ((int64_t)g * N + n) * C + q);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 cn = ld4(P.c_out + n * C + q);
  const float4 c = G == 4 ? ld4(P.c_in + n * C + q) : zero;
  const float4 gh = P.g_h ? ld4(P.g_h + n * C + q) : zero;
  const float4 gc = P.g_c ? ld4(P.g_c + n * C + q) : zero;
  float gz[G][4], gcin[4];
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) {
    const float o = sig(at(zz[GO], k2)), tc = tanhf(at(cn, k2));
    const float dh = at(gh, k2);
    gz[GO][k2] = dh * tc * o * (1.f - o);
    const float dc = at(gc, k2) + dh * o * (1.f - tc * tc);
    const float i = sig(at(zz[GI], k2)), ct = tanhf(at(zz[GC], k2));
    gz[GI][k2] = dc * ct * i * (1.f - i);
    gz[GC][k2] = dc * i * (1.f - ct * ct);
    if (G == 4) {
      const float f = sig(at(zz[GF], k2));
      gz[GF][k2] = dc * at(c, k2) * f * (1.f - f);
      gcin[k2] = dc * f;
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float4 v = make_float4(gz[g][0], gz[g][1], gz[g][2], gz[g][3]);
    st4(P.g_z + ((int64_t)g * N + n) * C + q, v);
    if (P.g_p_dst) st4(P.g_p_dst + n * ldp + P.s_off + g * C + q, v);
  }
  // the padding columns of the projection's gradient (<= 96 of them: one float4 of this row's 24 threads each)
  if (P.g_p_dst && q < P.pad_n) st4(P.g_p_dst + n * ldp + P.pad_off + q, zero);
  if (G == 4 && P.g_c_in) st4(P.g_c_in + n * C + q, make_float4(gcin[0], gcin[1], gcin[2], gcin[3]));
}

static int lstm_train_batch(const ggnn_lstm_train_problem* problems, int n_problems, int n_gates, bool backward, LstmTrainBatch& B) {
  if (!problems || n_problems < 1 || n_problems > GGNN_LSTM_TRAIN_MAX || (n_gates != 3 && n_gates != 4)) return GGNN_EINVAL;
  B.n = 0;
  B.blk_off[0] = 0;
  for (int k = 0; k < n_problems; ++k) {
    const ggnn_lstm_train_problem& P = problems[k];
    if (P.N < 0) return GGNN_EINVAL;
    if (P.N == 0) continue;
    if (!backward) {
      if (!P.z || !P.p_dst || !P.h_out || !P.c_out || (n_gates == 4) != (P.c_in != nullptr)) return GGNN_EINVAL;
      if (P.s_off < 0 || (P.s_off & 3) || (P.ldp & 3) || P.ldp < (int64_t)P.s_off + n_gates * C) return GGNN_EINVAL;
      if (!aligned16(P.z) || !aligned16(P.p_dst) || !aligned16(P.h_out) || !aligned16(P.c_out) || !aligned16(P.c_in)) return GGNN_EINVAL;
    } else {
      if (!P.z || !P.c_out || !P.g_z || (n_gates == 4 && !P.c_in)) return GGNN_EINVAL;
      if (P.g_p_dst && (P.s_off < 0 || (P.s_off & 3) || (P.ldp & 3) || P.ldp < (int64_t)P.s_off + n_gates * C)) return GGNN_EINVAL;
      if (P.pad_n < 0 || P.pad_n > C || (P.pad_n & 3) || (P.pad_n > 0 && (!P.g_p_dst || P.pad_off < 0 || (P.pad_off & 3) ||
                                                                           (int64_t)P.pad_off + P.pad_n > P.ldp)))
        return GGNN_EINVAL;
      if (!aligned16(P.z) || !aligned16(P.c_in) || !aligned16(P.c_out) || !aligned16(P.g_h) || !aligned16(P.g_c) ||
          !aligned16(P.g_z) || !aligned16(P.g_p_dst) || !aligned16(P.g_c_in))
        return GGNN_EINVAL;
    }
    const int64_t blocks = (P.N * LT_Q + 255) / 256;
    if (B.blk_off[B.n] + blocks >= 0x7fffffff) return GGNN_EINVAL;
    B.p[B.n] = P;
    B.blk_off[B.n + 1] = B.blk_off[B.n] + (int)blocks;
    ++B.n;
  }
  return GGNN_OK;
}

}  // namespace ggnn

extern "C" int ggnn_lstm_train_forward_batch(const ggnn_lstm_train_problem* problems, int n_problems, int n_gates,
                                             ggnn_stream_t stream_) {
  using namespace ggnn;
  LstmTrainBatch B;
  if (const int rc = lstm_train_batch(problems, n_problems, n_gates, false, B)) return rc;
  if (B.n == 0) return 0;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_gates == 4)
    hipLaunchKernelGGL(lstm_train_fwd_kernel<4>, dim3((unsigned)B.blk_off[B.n]), dim3(256), 0, stream, B);
  else
    hipLaunchKernelGGL(lstm_train_fwd_kernel<3>, dim3((unsigned)B.blk_off[B.n]), dim3(256), 0, stream, B);
  return hipGetLastError() == hipSuccess ? 0 : GGNN_ELAUNCH;
}

extern "C" int ggnn_lstm_train_backward_batch(const ggnn_lstm_train_problem* problems, int n_problems, int n_gates,
                                              ggnn_stream_t stream_) {
  using namespace ggnn;
  LstmTrainBatch B;
  if (const int rc = lstm_train_batch(problems, n_problems, n_gates, true, B)) return rc;
  if (B.n == 0) return 0;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_gates == 4)
    hipLaunchKernelGGL(lstm_train_bwd_kernel<4>, dim3((unsigned)B.blk_off[B.n]), dim3(256), 0, stream, B);
  else
    hipLaunchKernelGGL(lstm_train_bwd_kernel<3>, dim3((unsigned)B.blk_off[B.n]), dim3(256), 0, stream, B);
  return hipGetLastError() == hipSuccess ? 0 : GGNN_ELAUNCH;
}

extern "C" int ggnn_lstm_train_forward(float* z, const float* p_dst, int64_t ldp, int s_off, const float* c_in,
                                       float* h_out, float* c_out, int64_t N, int n_gates, ggnn_stream_t stream) {
  ggnn_lstm_train_problem P = {};
  P.z = z, P.p_dst = p_dst, P.c_in = c_in, P.h_out = h_out, P.c_out = c_out;
  P.ldp = ldp, P.N = N, P.s_off = s_off;
  return ggnn_lstm_train_forward_batch(&P, 1, n_gates, stream);
}

extern "C" int ggnn_lstm_train_backward(const float* z, const float* c_in, const float* c_out, const float* g_h,
                                        const float* g_c, float* g_z, float* g_p_dst, int64_t ldp, int s_off,
                                        float* g_c_in, int64_t N, int n_gates, ggnn_stream_t stream) {
  ggnn_lstm_train_problem P = {};
  P.z = const_cast<float*>(z), P.c_in = c_in, P.c_out = const_cast<float*>(c_out), P.g_h = g_h, P.g_c = g_c;
  P.g_z = g_z, P.g_p_dst = g_p_dst, P.g_c_in = g_c_in;
  P.ldp = ldp, P.N = N, P.s_off = s_off;
  return ggnn_lstm_train_backward_batch(&P, 1, n_gates, stream);
}
