// Output heads of GrainNN_regressor.forward (models.py:433-452) and
// GrainNN_classifier.forward (models.py:595-609).  Sixteen lanes own one node (6 channels
// each, three coalesced 8-byte loads), reduce their partial dots with four xor-shuffles.
#include "common.h"

namespace ggnn {

template <int R>
__device__ __forceinline__ void node_dots(const float* __restrict__ hrow,
                                          const float* __restrict__ w, int l16, float (&out)[R]) {
  float2 h[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) h[k] = *reinterpret_cast<const float2*>(hrow + 32 * k + 2 * l16);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float2 wv = *reinterpret_cast<const float2*>(w + r * C + 32 * k + 2 * l16);
      s += h[k].x * wv.x + h[k].y * wv.y;
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m, 16);
    out[r] = s;
  }
}

__global__ __launch_bounds__(256) void heads_regressor_kernel(
    const float* __restrict__ h_joint, int64_t n_joint, const float* __restrict__ h_grain,
    int64_t n_grain, const float* __restrict__ x_grain, int64_t ldx_grain,
    const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y_joint,
    float* __restrict__ y_grain, float* __restrict__ grain_area) {
  const int64_t node = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int l16 = threadIdx.x & 15;
  if (node >= n_joint + n_grain) return;  // uniform per 16-lane group
  float d[2];
  if (node < n_joint) {
    node_dots<2>(h_joint + node * C, w, l16, d);
    if (l16 == 0) {  // models.py:443
      y_joint[2 * node] = tanhf_(d[0] + b[0]);  // v_exp / v_rcp form: libm tanhf on one lane in 16 cost 15 us
      y_joint[2 * node + 1] = tanhf_(d[1] + b[1]);
    }
  } else {
    const int64_t g = node - n_joint;
    node_dots<2>(h_grain + g * C, w + 2 * C, l16, d);
    if (l16 == 0) {
      const float t0 = tanhf_(d[0] + b[2]);
      grain_area[g] = t0 / 20.0f + x_grain[g * ldx_grain + 3];  // models.py:445
      y_grain[2 * g] = t0;                                      // :450
      y_grain[2 * g + 1] = fmaxf(d[1] + b[3], 0.f);             // :452
    }
  }
}

// The regressor's heads followed, for the same node, by GrainNN_regressor.update's periodic branch and the z
// advance (step_update_kernel, step.hip): one launch instead of two behind the last gate GEMM of a step.
__global__ __launch_bounds__(256) void heads_regressor_update_kernel(
    const float* __restrict__ h_joint, int64_t n_joint, const float* __restrict__ h_grain, int64_t n_grain,
    float* __restrict__ x_joint, int64_t ldxj, float* __restrict__ x_grain, int64_t ldxg, int f_grain,
    const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y_joint,
    float* __restrict__ y_grain, float* __restrict__ grain_area, float dz, float zmax, int32_t* __restrict__ flags) {
  const int64_t node = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int l16 = threadIdx.x & 15;
  if (node >= n_joint + n_grain) return;  // uniform per 16-lane group
  float d[2];
  if (node < n_joint) {
    node_dots<2>(h_joint + node * C, w, l16, d);
    if (l16 == 0) {
      const float dx = tanhf_(d[0] + b[0]), dy = tanhf_(d[1] + b[1]);  // models.py:443
      y_joint[2 * node] = dx;
      y_joint[2 * node + 1] = dy;
      float* x = x_joint + node * ldxj;
      x[0] += dx / 5.0f;  // models.py:505 (scaling['joint'] = 5)
      x[1] += dy / 5.0f;
      x[2] += dz;         // test.py:402
      x[6] = dx;          // models.py:510
      x[7] = dy;
    }
  } else {
    const int64_t g = node - n_joint;
    node_dots<2>(h_grain + g * C, w + 2 * C, l16, d);
    if (l16 == 0) {
      float* x = x_grain + g * ldxg;
      const float da = tanhf_(d[0] + b[2]), dv = fmaxf(d[1] + b[3], 0.f);
      grain_area[g] = da / 20.0f + x[3];  // models.py:445, on the area the forward saw
      y_grain[2 * g] = da;                // :450
      y_grain[2 * g + 1] = dv;            // :452
      const float z = x[2] + dz;          // test.py:401
      x[2] = z;
      x[3] += da / 20.0f;                 // models.py:506 (scaling['grain'] = 20)
      x[4] = dv;                          // :507
      x[f_grain - 1] = da;                // :511
      if (g == 0) flags[1] = z > zmax ? 1 : 0;  // test.py:405
    }
  }
}

__global__ __launch_bounds__(256) void heads_classifier_node_kernel(
    const float* __restrict__ h_joint, int64_t n_joint, const float* __restrict__ w_node,
    float* __restrict__ node_tmp) {
  const int64_t node = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int l16 = threadIdx.x & 15;
  if (node >= n_joint) return;
  float d[6];
  node_dots<6>(h_joint + node * C, w_node, l16, d);
  if (l16 < 6) {
    float v = d[0];
#pragma unroll
    for (int r = 1; r < 6; ++r) v = (l16 == r) ? d[r] : v;
    node_tmp[node * 8 + l16] = v;
  }
}

__global__ __launch_bounds__(256) void heads_classifier_edge_kernel(
    const float* __restrict__ node_tmp, int64_t n_joint, const int64_t* __restrict__ ei, int64_t E_cap,
    const int64_t* __restrict__ E_dev, const float* __restrict__ edge_attr, const float* __restrict__ w_edge,
    float* __restrict__ edge_event, float* __restrict__ edge) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t E = E_dev ? *E_dev : E_cap;   // (E_dev: include/ggnn.h, ggnn_prepare_edge)
  if (e >= E) return;
  const int64_t s = ei[e], d = ei[E + e];
  if ((uint64_t)s >= (uint64_t)n_joint || (uint64_t)d >= (uint64_t)n_joint) {
    edge_event[e] = NAN;  // never reached when the same edge_index passed ggnn_build_csr
    edge[2 * e] = NAN;
    edge[2 * e + 1] = NAN;
    return;
  }
  const float a = edge_attr[e];
  const float* ts = node_tmp + s * 8;
  const float* td = node_tmp + d * 8;
  edge[2 * e] = tanhf(ts[0] + td[3] + w_edge[0] * a + w_edge[3]);      // models.py:609
  edge[2 * e + 1] = tanhf(ts[1] + td[4] + w_edge[1] * a + w_edge[4]);
  edge_event[e] = ts[2] + td[5] + w_edge[2] * a + w_edge[5];           // models.py:607
}

// Backward of heads_regressor_kernel (training path): from the gradients of (y_joint, y_grain, grain_area) and the
// saved outputs the gradient of the pre-activations (4 floats per node, columns 2-3 zero: the [N, 4] operand of
// ggnn_wgrad for the head weights) and of the hidden states (g_h = g_pre W).  Sixteen lanes own one node.
__global__ __launch_bounds__(256) void heads_regressor_bwd_kernel(
    int64_t n_joint, int64_t n_grain, const float* __restrict__ w, const float* __restrict__ y_joint,
    const float* __restrict__ y_grain, const float* __restrict__ g_yj, const float* __restrict__ g_yg,
    const float* __restrict__ g_area, float* __restrict__ g_pre_joint, float* __restrict__ g_pre_grain,
    float* __restrict__ g_h_joint, float* __restrict__ g_h_grain) {
  const int64_t node = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int l16 = threadIdx.x & 15;
  if (node >= n_joint + n_grain) return;
  const bool joint = node < n_joint;
  const int64_t i = joint ? node : node - n_joint;
  float p0, p1;
  if (joint) {
    const float y0 = y_joint[2 * i], y1 = y_joint[2 * i + 1];
    p0 = (g_yj ? g_yj[2 * i] : 0.f) * (1.f - y0 * y0);  // models.py:443
    p1 = (g_yj ? g_yj[2 * i + 1] : 0.f) * (1.f - y1 * y1);
  } else {
    const float t0 = y_grain[2 * i];
    const float gt = (g_yg ? g_yg[2 * i] : 0.f) + (g_area ? g_area[i] / 20.0f : 0.f);  // models.py:445, 450
    p0 = gt * (1.f - t0 * t0);
    p1 = y_grain[2 * i + 1] > 0.f ? (g_yg ? g_yg[2 * i + 1] : 0.f) : 0.f;                // :452
  }
  float* gp = (joint ? g_pre_joint : g_pre_grain) + 4 * i;
  if (l16 < 4) gp[l16] = l16 == 0 ? p0 : l16 == 1 ? p1 : 0.f;
  const float* wr = w + (joint ? 0 : 2 * C);
  float* gh = (joint ? g_h_joint : g_h_grain) + i * C;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float2 w0 = *reinterpret_cast<const float2*>(wr + 32 * k + 2 * l16);
    const float2 w1 = *reinterpret_cast<const float2*>(wr + C + 32 * k + 2 * l16);
    *reinterpret_cast<float2*>(gh + 32 * k + 2 * l16) = make_float2(p0 * w0.x + p1 * w1.x, p0 * w0.y + p1 * w1.y);
  }
}


}  // namespace ggnn

extern "C" int ggnn_heads_regressor(const float* h_joint, int64_t n_joint, const float* h_grain,
                                    int64_t n_grain, const float* x_grain, int64_t ldx_grain,
                                    const float* w, const float* b, float* y_joint, float* y_grain,
                                    float* grain_area, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!h_joint || !h_grain || !x_grain || !w || !b || !y_joint || !y_grain || !grain_area)
    return GGNN_EINVAL;
  if (n_joint <= 0 || n_grain <= 0 || ldx_grain < 4) return GGNN_EINVAL;
  const int64_t nblk = ((n_joint + n_grain) * 16 + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(heads_regressor_kernel, dim3((unsigned)nblk), dim3(256), 0,
                     (hipStream_t)stream, h_joint, n_joint, h_grain, n_grain, x_grain, ldx_grain, w,
                     b, y_joint, y_grain, grain_area);
  return launch_status();
}

extern "C" int ggnn_heads_regressor_backward(int64_t n_joint, int64_t n_grain, const float* w, const float* y_joint,
                                             const float* y_grain, const float* g_y_joint, const float* g_y_grain,
                                             const float* g_grain_area, float* g_pre_joint, float* g_pre_grain,
                                             float* g_h_joint, float* g_h_grain, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!w || !y_joint || !y_grain || !g_pre_joint || !g_pre_grain || !g_h_joint || !g_h_grain) return GGNN_EINVAL;
  if (n_joint <= 0 || n_grain <= 0) return GGNN_EINVAL;
  const int64_t nblk = ((n_joint + n_grain) * 16 + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(heads_regressor_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, n_joint,
                     n_grain, w, y_joint, y_grain, g_y_joint, g_y_grain, g_grain_area, g_pre_joint, g_pre_grain,
                     g_h_joint, g_h_grain);
  return launch_status();
}

extern "C" int ggnn_heads_regressor_update(const float* h_joint, int64_t n_joint, const float* h_grain,
                                           int64_t n_grain, float* x_joint, int64_t ldx_joint, float* x_grain,
                                           int64_t ldx_grain, int f_grain, const float* w, const float* b,
                                           float* y_joint, float* y_grain, float* grain_area, float dz, float zmax,
                                           int32_t* flags, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!h_joint || !h_grain || !x_joint || !x_grain || !w || !b || !y_joint || !y_grain || !grain_area || !flags)
    return GGNN_EINVAL;
  if (n_joint <= 0 || n_grain <= 0 || ldx_joint < 8 || f_grain < 6 || ldx_grain < f_grain) return GGNN_EINVAL;
  const int64_t nblk = ((n_joint + n_grain) * 16 + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(heads_regressor_update_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     h_joint, n_joint, h_grain, n_grain, x_joint, ldx_joint, x_grain, ldx_grain, f_grain, w, b,
                     y_joint, y_grain, grain_area, dz, zmax, flags);
  return launch_status();
}

extern "C" int ggnn_heads_classifier(const float* h_joint, int64_t n_joint,
                                     const int64_t* edge_index_jj, int64_t E,
                                     const float* edge_attr_jj, const float* w_node,
                                     const float* w_edge, float* node_tmp, float* edge_event,
                                     float* edge, ggnn_stream_t stream) {
  return ggnn_heads_classifier_n(h_joint, n_joint, edge_index_jj, E, nullptr, edge_attr_jj, w_node, w_edge, node_tmp, edge_event,
                                 edge, stream);
}

extern "C" int ggnn_heads_classifier_n(const float* h_joint, int64_t n_joint, const int64_t* edge_index_jj, int64_t E,
                                       const int64_t* E_dev, const float* edge_attr_jj, const float* w_node,
                                       const float* w_edge, float* node_tmp, float* edge_event, float* edge,
                                       ggnn_stream_t stream) {
  using namespace ggnn;
  if (!h_joint || !w_node || !w_edge || !node_tmp || n_joint <= 0 || E < 0) return GGNN_EINVAL;
  if (E > 0 && (!edge_index_jj || !edge_attr_jj || !edge_event || !edge)) return GGNN_EINVAL;
  const int64_t nb_node = (n_joint * 16 + 255) / 256, nb_edge = (E + 255) / 256;
  if (nb_node >= INT32_MAX || nb_edge >= INT32_MAX) return GGNN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(heads_classifier_node_kernel, dim3((unsigned)nb_node), dim3(256), 0, s,
                     h_joint, n_joint, w_node, node_tmp);
  if (E > 0)
    hipLaunchKernelGGL(heads_classifier_edge_kernel, dim3((unsigned)nb_edge), dim3(256), 0, s,
                       node_tmp, n_joint, edge_index_jj, E, E_dev, edge_attr_jj, w_edge, edge_event, edge);
  return launch_status();
}
