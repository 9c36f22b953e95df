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

template <int G>
__global__ __launch_bounds__(256) void lstm_train_fwd_kernel(
    float* __restrict__ z, const float* __restrict__ p_dst, int64_t ldp, int s_off,
    const float* __restrict__ c_in, float* __restrict__ h_out, float* __restrict__ c_out, int64_t N) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= N * LT_Q) return;
  const int64_t n = t / LT_Q;
  const int q = (int)(t - n * LT_Q) * 4;
  constexpr int GI = 0, GF = 1, GC = G - 2, GO = G - 1;
  float4 zz[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float* zp = z + ((int64_t)g * N + n) * C + q;
    const float4 a = ld4(zp), s = ld4(p_dst + n * ldp + s_off + g * C + q);
    zz[g] = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
    st4(zp, zz[g]);
  }
  float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
  if (G == 4) c = ld4(c_in + n * C + q);
  float cn[4], hn[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v = sig(at(zz[GI], k)) * tanhf(at(zz[GC], k));
    if (G == 4) v += sig(at(zz[GF], k)) * at(c, k);
    cn[k] = v;
    hn[k] = sig(at(zz[GO], k)) * tanhf(v);
  }
  st4(c_out + n * C + q, make_float4(cn[0], cn[1], cn[2], cn[3]));
  st4(h_out + n * C + q, make_float4(hn[0], hn[1], hn[2], hn[3]));
}

template <int G>
__global__ __launch_bounds__(256) void lstm_train_bwd_kernel(
    const float* __restrict__ z, const float* __restrict__ c_in, const float* __restrict__ c_out,
    const float* __restrict__ g_h, const float* __restrict__ g_c, float* __restrict__ g_z,
    float* __restrict__ g_p_dst, int64_t ldp, int s_off, float* __restrict__ g_c_in, int64_t N) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= N * LT_Q) return;
  const int64_t n = t / LT_Q;
  const int q = (int)(t - n * LT_Q) * 4;
  constexpr int GI = 0, GF = 1, GC = G - 2, GO = G - 1;
  float4 zz[G];
#pragma unroll
  for (int g = 0; g < G; ++g) zz[g] = ld4(z + ((int64_t)g * N + n) * C + q);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 cn = ld4(c_out + n * C + q);
  const float4 c = G == 4 ? ld4(c_in + n * C + q) : zero;
  const float4 gh = g_h ? ld4(g_h + n * C + q) : zero;
  const float4 gc = g_c ? ld4(g_c + n * C + q) : zero;
  float gz[G][4], gcin[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float o = sig(at(zz[GO], k)), tc = tanhf(at(cn, k));
    const float dh = at(gh, k);
    gz[GO][k] = dh * tc * o * (1.f - o);
    const float dc = at(gc, k) + dh * o * (1.f - tc * tc);
    const float i = sig(at(zz[GI], k)), ct = tanhf(at(zz[GC], k));
    gz[GI][k] = dc * ct * i * (1.f - i);
    gz[GC][k] = dc * i * (1.f - ct * ct);
    if (G == 4) {
      const float f = sig(at(zz[GF], k));
      gz[GF][k] = dc * at(c, k) * f * (1.f - f);
      gcin[k] = dc * f;
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float4 v = make_float4(gz[g][0], gz[g][1], gz[g][2], gz[g][3]);
    st4(g_z + ((int64_t)g * N + n) * C + q, v);
    if (g_p_dst) st4(g_p_dst + n * ldp + s_off + g * C + q, v);
  }
  if (G == 4 && g_c_in) st4(g_c_in + n * C + q, make_float4(gcin[0], gcin[1], gcin[2], gcin[3]));
}

}  // namespace ggnn

extern "C" int ggnn_lstm_train_forward(float* z, const float* p_dst, int64_t ldp, int s_off, const float* c_in,
                                       float* h_out, float* c_out, int64_t N, int n_gates, ggnn_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  using namespace ggnn;
  if (N < 0 || (n_gates != 3 && n_gates != 4)) return GGNN_EINVAL;
  if (N == 0) return 0;
  if (!z || !p_dst || !h_out || !c_out || (n_gates == 4) != (c_in != nullptr)) return GGNN_EINVAL;
  if (s_off < 0 || (s_off & 3) || (ldp & 3) || ldp < (int64_t)s_off + n_gates * C) return GGNN_EINVAL;
  if (!aligned16(z) || !aligned16(p_dst) || !aligned16(h_out) || !aligned16(c_out) || !aligned16(c_in))
    return GGNN_EINVAL;
  const int64_t blocks = (N * LT_Q + 255) / 256;
  if (blocks > 0x7fffffff) return GGNN_EINVAL;
  if (n_gates == 4)
    hipLaunchKernelGGL(lstm_train_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, z, p_dst, ldp, s_off,
                       c_in, h_out, c_out, N);
  else
    hipLaunchKernelGGL(lstm_train_fwd_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, stream, z, p_dst, ldp, s_off,
                       c_in, h_out, c_out, N);
  return hipGetLastError() == hipSuccess ? 0 : GGNN_ELAUNCH;
}

extern "C" int ggnn_lstm_train_backward(const float* z, const float* c_in, const float* c_out, const float* g_h,
                                        const float* g_c, float* g_z, float* g_p_dst, int64_t ldp, int s_off,
                                        float* g_c_in, int64_t N, int n_gates, ggnn_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  using namespace ggnn;
  if (N < 0 || (n_gates != 3 && n_gates != 4)) return GGNN_EINVAL;
  if (N == 0) return 0;
  if (!z || !c_out || !g_z || (n_gates == 4 && !c_in)) return GGNN_EINVAL;
  if (g_p_dst && (s_off < 0 || (s_off & 3) || (ldp & 3) || ldp < (int64_t)s_off + n_gates * C)) return GGNN_EINVAL;
  if (!aligned16(z) || !aligned16(c_in) || !aligned16(c_out) || !aligned16(g_h) || !aligned16(g_c) ||
      !aligned16(g_z) || !aligned16(g_p_dst) || !aligned16(g_c_in))
    return GGNN_EINVAL;
  const int64_t blocks = (N * LT_Q + 255) / 256;
  if (blocks > 0x7fffffff) return GGNN_EINVAL;
  if (n_gates == 4)
    hipLaunchKernelGGL(lstm_train_bwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, z, c_in, c_out, g_h,
                       g_c, g_z, g_p_dst, ldp, s_off, g_c_in, N);
  else
    hipLaunchKernelGGL(lstm_train_bwd_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, stream, z, c_in, c_out, g_h,
                       g_c, g_z, g_p_dst, ldp, s_off, g_c_in, N);
  return hipGetLastError() == hipSuccess ? 0 : GGNN_ELAUNCH;
}
