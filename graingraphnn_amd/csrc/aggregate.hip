// Periodic-boundary GAT aggregation for one edge type, all gates fused (HBM-bound).
// Replaces PeriodConv.message (periodGATconv.py:204-236) and the gather / scatter-add of
// PyG MessagePassing.propagate (periodGATconv.py:174-175) with an atomics-free CSR sweep.
//
// Per edge e = (j -> i) and gate g, with reloc_e = minimg(x_j[:3] - x_i[:3]) computed
// exactly as periodGATconv.py:209-210 does, a_e = edge_attr_e, and the node-level
// projections K0_j, V0_j (key/value WITHOUT their first three input columns) and Q_i:
//   k_e   = K0_j + Wk3 . reloc_e + w_edge * a_e
//   s_e   = (Q_i . k_e) / sqrt(96)
//   alpha = exp(s_e - max_i) / (sum_i exp(.) + 1e-16)            (PyG softmax)
//   r_e   = relu(V0_j + Wv3 . reloc_e)
//   agg_i = sum_e alpha_e r_e,  sa_i = sum_e alpha_e,  sae_i = sum_e alpha_e a_e
//
// Mapping: one workgroup (4 waves) owns AG_BD consecutive destination rows; their CSR
// segment (neighbour ids, min-image offsets, edge lengths) is staged through LDS once with
// coalesced reads and then broadcast-read by the compute lanes.  A half-wave (32 lanes x 3
// channels = one 384-byte row fragment per 12-byte-per-lane load) owns one (destination,
// gate) pair and walks the in-edges four at a time with an online (running max) softmax, so
// any degree works with bounded registers and nothing is ever re-read.
#include "common.h"

namespace ggnn {

constexpr int AG_BD = 16;    // destination rows per workgroup
constexpr int AG_CAP = 512;  // CSR slots staged in LDS per workgroup (beyond: direct reads)
constexpr int AG_CH = 4;     // edges per softmax chunk

__device__ __forceinline__ float halfwave_sum(float v) {
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 32);
  return v;
}

struct EdgeInfo {
  int j;
  float rx, ry, rz, a;
};

__device__ __forceinline__ EdgeInfo edge_from_global(const ggnn_aggregate_args& A, int p,
                                                     int64_t i) {
  EdgeInfo e;
  e.j = A.col[p];
  const float* xs = A.x_src + (int64_t)e.j * A.ldx_src;
  const float* xd = A.x_dst + i * A.ldx_dst;
  float r[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float rel = xs[c] - xd[c];
    const float w = rel > 0.5f ? -1.0f : (rel < -0.5f ? 1.0f : 0.0f);
    r[c] = w + rel;  // periodGATconv.py:210
  }
  e.rx = r[0];
  e.ry = r[1];
  e.rz = r[2];
  e.a = A.edge_attr[A.perm[p]];
  return e;
}

template <int G>
__global__ __launch_bounds__(256) void aggregate_kernel(const ggnn_aggregate_args A) {
  __shared__ float s_ep[G * GGNN_EDGE_PARAM_ROWS * C];
  __shared__ __attribute__((aligned(16))) float s_edge[AG_CAP * 4];
  __shared__ int s_col[AG_CAP];
  __shared__ int s_rowptr[AG_BD + 1];

  const int tid = threadIdx.x;
  const int b = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t i0 = (int64_t)b * AG_BD;
  const int nd = (int)min((int64_t)AG_BD, A.n_dst - i0);

  for (int t = tid; t < G * GGNN_EDGE_PARAM_ROWS * C; t += 256) s_ep[t] = A.edge_params[t];
  if (tid <= nd) s_rowptr[tid] = A.rowptr[i0 + tid];
  __syncthreads();
  const int e_begin = s_rowptr[0];
  const int n_stage = min(s_rowptr[nd] - e_begin, AG_CAP);
  for (int t = tid; t < n_stage; t += 256) {
    const int p = e_begin + t;
    int d = 0;
    while (d + 1 < nd && s_rowptr[d + 1] <= p) ++d;
    const EdgeInfo e = edge_from_global(A, p, i0 + d);
    s_col[t] = e.j;
    *reinterpret_cast<f32x4*>(&s_edge[4 * t]) = (f32x4){e.rx, e.ry, e.rz, e.a};
  }
  __syncthreads();

  const int hw = tid >> 5;        // half-wave id 0..7
  const int ch = 3 * (tid & 31);  // first of this lane's three channels
  const float inv_sqrt_c = 0.10206207261596577f;  // 1/sqrt(96), periodGATconv.py:226

  // Work item = (destination d, gate g), d-major: the half-waves that share a destination
  // read the 4 x 768-byte K|V fragments of a neighbour row back to back.
  for (int item = hw; item < nd * G; item += 8) {
    const int d = item / G, g = item - d * G;
    const int64_t i = i0 + d;
    const int beg = s_rowptr[d], end = s_rowptr[d + 1];

    const f3 q = ld3(A.p_dst + i * A.ldp_dst + A.q_off + g * C + ch);
    const float* ep = &s_ep[g * GGNN_EDGE_PARAM_ROWS * C + ch];
    const f3 wkx = ld3(ep), wky = ld3(ep + C), wkz = ld3(ep + 2 * C);
    const f3 wvx = ld3(ep + 3 * C), wvy = ld3(ep + 4 * C), wvz = ld3(ep + 5 * C);
    const f3 we = ld3(ep + 6 * C);
    const float* kvbase = A.p_src + A.kv_off + g * 2 * C + ch;

    float mx = -INFINITY, den = 0.f, sae = 0.f;
    f3 acc = {0.f, 0.f, 0.f};

    for (int p = beg; p < end; p += AG_CH) {
      const int nact = min(AG_CH, end - p);
      EdgeInfo ed[AG_CH];
      f3 kk[AG_CH], vv[AG_CH];
#pragma unroll
      for (int t = 0; t < AG_CH; ++t) {
        ed[t] = {0, 0.f, 0.f, 0.f, 0.f};
        kk[t] = {0.f, 0.f, 0.f};
        vv[t] = {0.f, 0.f, 0.f};
        if (t < nact) {
          const int s = p + t - e_begin;
          if (s < AG_CAP) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&s_edge[4 * s]);
            ed[t] = {s_col[s], v.x, v.y, v.z, v.w};
          } else {
            ed[t] = edge_from_global(A, p + t, i);
          }
          const float* row = kvbase + (int64_t)ed[t].j * A.ldp_src;
          kk[t] = ld3(row);
          vv[t] = ld3(row + C);
        }
      }
      float s[AG_CH];
      float mnew = mx;
#pragma unroll
      for (int t = 0; t < AG_CH; ++t) {
        s[t] = -INFINITY;
        if (t < nact) {
          const float rx = ed[t].rx, ry = ed[t].ry, rz = ed[t].rz, a = ed[t].a;
          const float k0 = kk[t].x + wkx.x * rx + wky.x * ry + wkz.x * rz + we.x * a;
          const float k1 = kk[t].y + wkx.y * rx + wky.y * ry + wkz.y * rz + we.y * a;
          const float k2 = kk[t].z + wkx.z * rx + wky.z * ry + wkz.z * rz + we.z * a;
          s[t] = halfwave_sum(q.x * k0 + q.y * k1 + q.z * k2) * inv_sqrt_c;
          mnew = fmaxf(mnew, s[t]);
        }
      }
      const float scale = expf(mx - mnew);  // exp(-inf) = 0 on the first chunk
      den *= scale;
      sae *= scale;
      acc = {acc.x * scale, acc.y * scale, acc.z * scale};
#pragma unroll
      for (int t = 0; t < AG_CH; ++t) {
        if (t < nact) {
          const float rx = ed[t].rx, ry = ed[t].ry, rz = ed[t].rz;
          const float pe = expf(s[t] - mnew);
          den += pe;
          sae += pe * ed[t].a;
          acc.x += pe * fmaxf(vv[t].x + wvx.x * rx + wvy.x * ry + wvz.x * rz, 0.f);
          acc.y += pe * fmaxf(vv[t].y + wvx.y * rx + wvy.y * ry + wvz.y * rz, 0.f);
          acc.z += pe * fmaxf(vv[t].z + wvx.z * rx + wvy.z * ry + wvz.z * rz, 0.f);
        }
      }
      mx = mnew;
    }

    const float inv = 1.0f / (den + 1e-16f);  // PyG softmax denominator
    float* orow = A.agg + i * A.ld_agg + g * A.a_gstride;
    st3(orow + A.a_off + ch, {acc.x * inv, acc.y * inv, acc.z * inv});
    if ((tid & 31) == 0) {
      orow[A.sc_off] = den * inv;
      orow[A.sc_off + 1] = sae * inv;
    }
  }
}

}  // namespace ggnn

extern "C" int ggnn_period_gat_aggregate(const ggnn_aggregate_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  const ggnn_aggregate_args& A = *args;
  if (!A.rowptr || !A.x_src || !A.x_dst || !A.p_src || !A.p_dst || !A.edge_params || !A.agg)
    return GGNN_EINVAL;
  if (A.E > 0 && (!A.col || !A.perm || !A.edge_attr)) return GGNN_EINVAL;
  if (A.n_dst <= 0 || A.n_src <= 0 || A.E < 0 || A.ldx_src < 3 || A.ldx_dst < 3) return GGNN_EINVAL;
  const int G = A.n_gates;
  if (G != 1 && G != 3 && G != 4) return GGNN_EINVAL;
  if (A.kv_off < 0 || A.q_off < 0 || A.a_off < 0 || A.sc_off < 0 || A.a_gstride < C) return GGNN_EINVAL;
  if (A.kv_off + (int64_t)G * 2 * C > A.ldp_src || A.q_off + (int64_t)G * C > A.ldp_dst) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.a_off + C > A.ld_agg) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.sc_off + 2 > A.ld_agg) return GGNN_EINVAL;
  const int64_t nblk = (A.n_dst + AG_BD - 1) / AG_BD;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  const dim3 grid((unsigned)nblk), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (G == 4)
    hipLaunchKernelGGL(aggregate_kernel<4>, grid, block, 0, s, A);
  else if (G == 3)
    hipLaunchKernelGGL(aggregate_kernel<3>, grid, block, 0, s, A);
  else
    hipLaunchKernelGGL(aggregate_kernel<1>, grid, block, 0, s, A);
  return launch_status();
}
