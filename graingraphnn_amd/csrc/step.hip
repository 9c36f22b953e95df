// Rollout-step glue on device (static topology):
//   ggnn_step_update  = GrainNN_regressor.update, periodic branch (models.py:503-516)
//                       + z advance (test.py:401-402)
//   ggnn_step_refresh = z clamp (test.py:405-407) + edge-length refresh (test.py:562-575)
//   ggnn_grain_centres = region centres of graph.update() (graph_datastruct.py:681-708) written
//                       to x_grain[:, :2] (test.py:468-478, 556-559), between the two
// Separate launches because each stage needs every node's updated coordinates.
#include "common.h"

namespace ggnn {

__global__ __launch_bounds__(256) void step_update_kernel(
    float* __restrict__ x_joint, int64_t n_joint, int64_t ldxj, float* __restrict__ x_grain,
    int64_t n_grain, int64_t ldxg, int f_grain, const float* __restrict__ y_joint,
    const float* __restrict__ y_grain, float dz, float zmax, int32_t* __restrict__ flags) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n_joint) {
    float* x = x_joint + t * ldxj;
    const float dx = y_joint[2 * t], dy = y_joint[2 * t + 1];
    x[0] += dx / 5.0f;  // models.py:505 (scaling['joint'] = 5)
    x[1] += dy / 5.0f;
    x[2] += dz;         // test.py:402
    x[6] = dx;          // models.py:510
    x[7] = dy;
  } else if (t < n_joint + n_grain) {
    const int64_t g = t - n_joint;
    float* x = x_grain + g * ldxg;
    const float da = y_grain[2 * g], dv = y_grain[2 * g + 1];
    const float z = x[2] + dz;  // test.py:401
    x[2] = z;
    x[3] += da / 20.0f;  // models.py:506 (scaling['grain'] = 20)
    x[4] = dv;           // :507
    x[f_grain - 1] = da;  // :511
    if (g == 0) flags[1] = z > zmax ? 1 : 0;  // test.py:405
  }
}

// One thread per grain: walk the grain's junctions (CSR row of the joint->grain edge type),
// min-image each to the previous moved one, shift by +1 where any vertex is below -eps, mean.
// ~6 junctions x 8 B per grain: the launch is latency-, not bandwidth-bound.
__global__ __launch_bounds__(256) void grain_centres_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ x_joint, int64_t ldxj, const float* __restrict__ offset,
    float factor, float* __restrict__ x_grain, int64_t ldxg, int64_t n_grain, int64_t n_joint,
    float* __restrict__ centres_before) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n_grain) return;
  if (centres_before != nullptr) {   // the centres as this call found them (the speculative event loop's snapshot)
    centres_before[2 * g] = x_grain[g * ldxg];
    centres_before[2 * g + 1] = x_grain[g * ldxg + 1];
  }
  const int p0 = rowptr[g], p1 = rowptr[g + 1];
  if (p1 - p0 <= 1) return;  // graph_datastruct.py:685: such a region keeps its centre
  const bool folded = factor > 1.0f;
  float prev[2], sum[2] = {0.f, 0.f}, lo[2] = {INFINITY, INFINITY};
  // batches of 8 junctions: the index loads, then the coordinate loads, are issued together; only
  // the min-image chain itself is sequential
  for (int q0 = p0; q0 < p1; q0 += 8) {
    int64_t js[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) js[k] = min((int64_t)max(col[min(q0 + k, p1 - 1)], 0), n_joint - 1);
    float raw[8][2];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float t = x_joint[js[k] * ldxj + c];
        if (folded) t = (t + (offset ? offset[2 * js[k] + c] : 0.f)) / factor;  // test.py:474
        raw[k][c] = t;
      }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (q0 + k >= p1) break;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float t = raw[k][c];
        if (q0 + k > p0) {  // periodic_move, graph_datastruct.py:55-72
          const float rel = t - prev[c];
          t += rel > 0.5f ? -1.0f : (rel < -0.5f ? 1.0f : 0.0f);
        }
        prev[c] = t;
        sum[c] += t;
        lo[c] = fminf(lo[c], t);
      }
    }
  }
  const float inv = 1.0f / (float)(p1 - p0);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float m = sum[c] * inv;
    if (!(lo[c] > -1e-12f)) m += 1.0f;  // inbound test, graph_datastruct.py:696-704
    if (folded) {                        // test.py:558-559
      m *= factor;
      m -= floorf(m);
    }
    x_grain[g * ldxg + c] = m;
  }
}

// Counts the events the host-side topology update would act on (test.py:418, models.py:624-626):
// flags[0] += #live grains with grain_area < area_threshold, flags[1] += #directed junction edges
// (src < dst) with edge_event > logit_threshold.  One pass over 10k + 60k values; the rollout reads
// the two words back once per step and only touches the host path when one is non-zero.
__global__ __launch_bounds__(256) void detect_events_kernel(
    const float* __restrict__ grain_area, const int32_t* __restrict__ live_grain, int64_t n_grain,
    float area_threshold, const float* __restrict__ edge_event, const int64_t* __restrict__ ei_jj,
    int64_t E_cap, const int64_t* __restrict__ E_dev, float logit_threshold, int32_t* __restrict__ flags,
    int32_t* __restrict__ range_word) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t E = E_dev ? *E_dev : E_cap;   // (E_dev: include/ggnn.h, ggnn_prepare_edge)
  // (the caller's fp16-range word travels with the counts and starts its next use clean: one thread moves it)
  if (range_word != nullptr && t == 0) flags[2] = atomicExch(range_word, 0);
  bool g = false, e = false;
  if (t < n_grain) g = live_grain[t] > 0 && grain_area[t] < area_threshold;
  else if (t - n_grain < E) {
    const int64_t k = t - n_grain;
    e = edge_event[k] > logit_threshold && ei_jj[k] < ei_jj[E + k];
  }
  const int ng = __popcll(__ballot(g)), ne = __popcll(__ballot(e));
  if ((threadIdx.x & 63) == 0) {
    if (ng) atomicAdd(&flags[0], ng);
    if (ne) atomicAdd(&flags[1], ne);
  }
}

struct RefreshArgs {
  ggnn_refresh_edge et[3];
  int64_t e_off[4];  // prefix sums of E over the edge types
  int n_et;
};

__global__ __launch_bounds__(256) void step_refresh_kernel(
    float* __restrict__ x_joint, int64_t n_joint, int64_t ldxj, float* __restrict__ x_grain,
    int64_t n_grain, int64_t ldxg, float zmax, const int32_t* __restrict__ flags,
    const RefreshArgs R) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_nodes = n_joint + n_grain;
  if (t < n_nodes) {
    if (flags[1]) {  // test.py:405-407
      if (t < n_joint)
        x_joint[t * ldxj + 2] = zmax;
      else
        x_grain[(t - n_joint) * ldxg + 2] = zmax;
    }
    return;
  }
  const int64_t eg = t - n_nodes;
  if (eg >= R.e_off[R.n_et]) return;
  int k = 0;
  while (k + 1 < R.n_et && eg >= R.e_off[k + 1]) ++k;
  const ggnn_refresh_edge& T = R.et[k];
  const int64_t e = eg - R.e_off[k];
  const int64_t E = T.E_dev ? *T.E_dev : T.E;   // (E_dev: include/ggnn.h, ggnn_prepare_edge)
  if (e >= E) return;
  const int64_t s = T.edge_index[e], d = T.edge_index[E + e];
  if ((uint64_t)s >= (uint64_t)T.n_src || (uint64_t)d >= (uint64_t)T.n_dst) {
    T.edge_attr[e] = NAN;  // never reached for an edge_index that passed ggnn_build_csr
    return;
  }
  const float* xs = T.x_src + s * T.ldx_src;
  const float* xd = T.x_dst + d * T.ldx_dst;
  float r[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {  // test.py:570-571
    const float rel = xs[c] - xd[c];
    const float w = rel > 0.5f ? -1.0f : (rel < -0.5f ? 1.0f : 0.0f);
    r[c] = w + rel;
  }
  T.edge_attr[e] = sqrtf(r[0] * r[0] + r[1] * r[1]);  // test.py:572
}

}  // namespace ggnn

extern "C" int ggnn_step_update(float* x_joint, int64_t n_joint, int64_t ldx_joint, float* x_grain,
                                int64_t n_grain, int64_t ldx_grain, int f_grain,
                                const float* y_joint, const float* y_grain, float dz, float zmax,
                                int32_t* flags, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!x_joint || !x_grain || !y_joint || !y_grain || !flags) return GGNN_EINVAL;
  if (n_joint <= 0 || n_grain <= 0 || ldx_joint < 8 || f_grain < 6 || ldx_grain < f_grain)
    return GGNN_EINVAL;
  const int64_t nblk = (n_joint + n_grain + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(step_update_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     x_joint, n_joint, ldx_joint, x_grain, n_grain, ldx_grain, f_grain, y_joint,
                     y_grain, dz, zmax, flags);
  return launch_status();
}

extern "C" int ggnn_grain_centres(const int32_t* rowptr, const int32_t* col, const float* x_joint,
                                  int64_t n_joint, int64_t ldx_joint, const float* domain_offset,
                                  float domain_factor, float* x_grain, int64_t n_grain,
                                  int64_t ldx_grain, float* centres_before, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!rowptr || !col || !x_joint || !x_grain) return GGNN_EINVAL;
  if (n_joint <= 0 || n_grain <= 0 || ldx_joint < 2 || ldx_grain < 2) return GGNN_EINVAL;
  if (!(domain_factor >= 1.0f)) return GGNN_EINVAL;
  const int64_t nblk = (n_grain + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(grain_centres_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     rowptr, col, x_joint, ldx_joint, domain_offset, domain_factor, x_grain,
                     ldx_grain, n_grain, n_joint, centres_before);
  return launch_status();
}

extern "C" int ggnn_detect_events(const float* grain_area, const int32_t* live_grain, int64_t n_grain,
                                  float area_threshold, const float* edge_event,
                                  const int64_t* edge_index_jj, int64_t E, float logit_threshold,
                                  int32_t* flags, int32_t* range_word, ggnn_stream_t stream) {
  return ggnn_detect_events_n(grain_area, live_grain, n_grain, area_threshold, edge_event, edge_index_jj, E, nullptr,
                              logit_threshold, flags, range_word, stream);
}

extern "C" int ggnn_detect_events_n(const float* grain_area, const int32_t* live_grain, int64_t n_grain,
                                    float area_threshold, const float* edge_event, const int64_t* edge_index_jj,
                                    int64_t E, const int64_t* E_dev, float logit_threshold, int32_t* flags,
                                    int32_t* range_word, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!grain_area || !live_grain || !flags || n_grain <= 0 || E < 0) return GGNN_EINVAL;
  if (E > 0 && (!edge_event || !edge_index_jj)) return GGNN_EINVAL;
  const int64_t nblk = (n_grain + E + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  if (hipMemsetAsync(flags, 0, (range_word ? 3 : 2) * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return GGNN_ELAUNCH;
  hipLaunchKernelGGL(detect_events_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     grain_area, live_grain, n_grain, area_threshold, edge_event, edge_index_jj, E, E_dev,
                     logit_threshold, flags, range_word);
  return launch_status();
}

extern "C" int ggnn_step_refresh(float* x_joint, int64_t n_joint, int64_t ldx_joint,
                                 float* x_grain, int64_t n_grain, int64_t ldx_grain, float zmax,
                                 const int32_t* flags, const ggnn_refresh_edge* edges,
                                 int n_edge_types, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!x_joint || !x_grain || !flags || n_joint <= 0 || n_grain <= 0) return GGNN_EINVAL;
  if (ldx_joint < 3 || ldx_grain < 3) return GGNN_EINVAL;
  if (n_edge_types < 0 || n_edge_types > 3 || (n_edge_types > 0 && !edges)) return GGNN_EINVAL;
  RefreshArgs R;
  R.n_et = n_edge_types;
  R.e_off[0] = 0;
  for (int k = 0; k < 3; ++k) {
    if (k < n_edge_types) {
      const ggnn_refresh_edge& T = edges[k];
      if (T.E < 0 || T.ldx_src < 2 || T.ldx_dst < 2 || T.n_src <= 0 || T.n_dst <= 0) return GGNN_EINVAL;
      if (T.E > 0 && (!T.edge_index || !T.x_src || !T.x_dst || !T.edge_attr)) return GGNN_EINVAL;
      R.et[k] = T;
      R.e_off[k + 1] = R.e_off[k] + T.E;
    } else {
      R.et[k] = ggnn_refresh_edge{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
      R.e_off[k + 1] = R.e_off[k];
    }
  }
  const int64_t total = n_joint + n_grain + R.e_off[n_edge_types];
  const int64_t nblk = (total + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(step_refresh_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     x_joint, n_joint, ldx_joint, x_grain, n_grain, ldx_grain, zmax, flags, R);
  return launch_status();
}
