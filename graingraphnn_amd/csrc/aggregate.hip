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
// ggnn_edge_prepare computes (reloc_e, a_e) once per forward in CSR order (16 bytes per
// edge, shared by the 7 gate sweeps of encoder + decoder).
//
// Mapping of the sweep: a half-wave (32 lanes x 3 channels = one 384-byte row fragment per
// 12-byte-per-lane load) owns one (destination, gate) item: it reads the row bounds, the
// neighbour ids and edge records (same address in all lanes -> one request), issues all K|V
// fragment loads of up to four in-edges at once, and folds them with an online (running
// max) softmax, so any degree works with bounded registers and nothing is re-read.  No
// workgroup barrier sits between loads: latency is hidden by 8 independent half-waves per
// workgroup and up to 8 workgroups per CU.  The grid is sized to the resident capacity and
// every workgroup owns one contiguous item range, XCD-contiguous so neighbouring rows meet
// in the same L2.  Only the per-gate edge parameters (7 x 96 floats) go through LDS.
#include "common.h"

namespace ggnn {

constexpr int AG_CH = 4;            // edges per softmax chunk
constexpr int AG_BLOCKS_PER_CU = 8; // resident workgroups per CU the grid is sized for
constexpr int AG_NUM_CU = 256;

// Sum over the 32 lanes of a half-wave, result in every lane: four DPP row steps inside each
// 16-lane row, then one ds_swizzle (xor 16) across the two rows.
__device__ __forceinline__ float halfwave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));                    // lane ^ 16
  return v;
}

// ---------------------------------------------------------------------------------------
// edge_prepare: einfo[p] = (reloc_x, reloc_y, reloc_z, edge_attr[perm[p]]) in CSR order
// ---------------------------------------------------------------------------------------
struct PrepareArgs {
  ggnn_prepare_edge et[3];
  int64_t e_off[4];
  int n_et;
};

__global__ __launch_bounds__(256) void edge_prepare_kernel(const PrepareArgs P) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= P.e_off[P.n_et]) return;
  int k = 0;
  while (k + 1 < P.n_et && t >= P.e_off[k + 1]) ++k;
  const ggnn_prepare_edge& T = P.et[k];
  const int64_t p = t - P.e_off[k];
  const float* xs = T.x_src + (int64_t)T.col[p] * T.ldx_src;
  const float* xd = T.x_dst + (int64_t)T.row[p] * T.ldx_dst;
  f32x4 o;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float rel = xs[c] - xd[c];
    const float w = rel > 0.5f ? -1.0f : (rel < -0.5f ? 1.0f : 0.0f);
    o[c] = w + rel;  // periodGATconv.py:210
  }
  o[3] = T.edge_attr[T.perm[p]];
  *reinterpret_cast<f32x4*>(T.einfo + 4 * p) = o;
}

// ---------------------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void aggregate_kernel(const ggnn_aggregate_args A) {
  // [g][channel][8]: 7 parameters of one channel contiguous (32 B) -> two 16-byte LDS reads
  __shared__ __attribute__((aligned(16))) float s_ep[G * C * 8];
  const int tid = threadIdx.x;
  for (int t = tid; t < G * GGNN_EDGE_PARAM_ROWS * C; t += 256) {
    const int g = t / (GGNN_EDGE_PARAM_ROWS * C), r = (t / C) % GGNN_EDGE_PARAM_ROWS, c = t % C;
    s_ep[(g * C + c) * 8 + r] = A.edge_params[t];
  }
  __syncthreads();

  const int hw = tid >> 5;        // half-wave id 0..7
  const int ch = 3 * (tid & 31);  // first of this lane's three channels
  const float inv_sqrt_c = 0.10206207261596577f;  // 1/sqrt(96), periodGATconv.py:226

  const int64_t n_items = A.n_dst * G;
  const int nblk = gridDim.x;
  const int b = xcd_remap(blockIdx.x, nblk);
  const int64_t per_blk = (n_items + nblk - 1) / nblk;
  const int64_t it_end = min(n_items, (b + 1) * per_blk);

  // Work item = (destination i, gate g), i-major: the half-waves that share a destination
  // read the G x 768-byte K|V fragments of a neighbour row back to back.
  for (int64_t item = b * per_blk + hw; item < it_end; item += 8) {
    const int64_t i = item / G;
    const int g = (int)(item - i * G);
    const f3 q = ld3(A.p_dst + i * A.ldp_dst + A.q_off + g * C + ch);
    const int beg = A.rowptr[i], end = A.rowptr[i + 1];
    const float* kvbase = A.p_src + A.kv_off + g * 2 * C + ch;

    // this lane's 3 channels x 8 parameters: [wkx wky wkz wvx wvy wvz we -]
    const f32x4* epp = reinterpret_cast<const f32x4*>(&s_ep[(g * C + ch) * 8]);
    const f32x4 e0a = epp[0], e0b = epp[1], e1a = epp[2], e1b = epp[3], e2a = epp[4], e2b = epp[5];

    float mx = -INFINITY, den = 0.f, sae = 0.f;
    f3 acc = {0.f, 0.f, 0.f};

    for (int p = beg; p < end; p += AG_CH) {
      const int nact = min(AG_CH, end - p);
      f32x4 ed[AG_CH];
      f3 kk[AG_CH], vv[AG_CH];
#pragma unroll
      for (int t = 0; t < AG_CH; ++t) {
        ed[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        kk[t] = {0.f, 0.f, 0.f};
        vv[t] = {0.f, 0.f, 0.f};
        if (t < nact) {
          const int j = A.col[p + t];
          ed[t] = *reinterpret_cast<const f32x4*>(A.einfo + 4 * (int64_t)(p + t));
          const float* row = kvbase + (int64_t)j * A.ldp_src;
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
          const float rx = ed[t].x, ry = ed[t].y, rz = ed[t].z, a = ed[t].w;
          const float k0 = kk[t].x + e0a.x * rx + e0a.y * ry + e0a.z * rz + e0b.z * a;
          const float k1 = kk[t].y + e1a.x * rx + e1a.y * ry + e1a.z * rz + e1b.z * a;
          const float k2 = kk[t].z + e2a.x * rx + e2a.y * ry + e2a.z * rz + e2b.z * a;
          s[t] = halfwave_sum(q.x * k0 + q.y * k1 + q.z * k2) * inv_sqrt_c;
          mnew = fmaxf(mnew, s[t]);
        }
      }
      const float scale = __expf(mx - mnew);  // exp(-inf) = 0 on the first chunk
      den *= scale;
      sae *= scale;
      acc = {acc.x * scale, acc.y * scale, acc.z * scale};
#pragma unroll
      for (int t = 0; t < AG_CH; ++t) {
        if (t < nact) {
          const float rx = ed[t].x, ry = ed[t].y, rz = ed[t].z;
          const float pe = __expf(s[t] - mnew);
          den += pe;
          sae += pe * ed[t].w;
          acc.x += pe * fmaxf(vv[t].x + e0a.w * rx + e0b.x * ry + e0b.y * rz, 0.f);
          acc.y += pe * fmaxf(vv[t].y + e1a.w * rx + e1b.x * ry + e1b.y * rz, 0.f);
          acc.z += pe * fmaxf(vv[t].z + e2a.w * rx + e2b.x * ry + e2b.y * rz, 0.f);
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

extern "C" int ggnn_edge_prepare(const ggnn_prepare_edge* edges, int n_edge_types,
                                 ggnn_stream_t stream) {
  using namespace ggnn;
  if (!edges || n_edge_types < 1 || n_edge_types > 3) return GGNN_EINVAL;
  PrepareArgs P;
  P.n_et = n_edge_types;
  P.e_off[0] = 0;
  for (int k = 0; k < 3; ++k) {
    if (k < n_edge_types) {
      const ggnn_prepare_edge& T = edges[k];
      if (T.E < 0 || T.ldx_src < 3 || T.ldx_dst < 3) return GGNN_EINVAL;
      if (T.E > 0 && (!T.col || !T.perm || !T.row || !T.edge_attr || !T.x_src || !T.x_dst ||
                      !T.einfo || !aligned16(T.einfo)))
        return GGNN_EINVAL;
      P.et[k] = T;
      P.e_off[k + 1] = P.e_off[k] + T.E;
    } else {
      P.et[k] = ggnn_prepare_edge{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
      P.e_off[k + 1] = P.e_off[k];
    }
  }
  const int64_t total = P.e_off[n_edge_types];
  if (total == 0) return GGNN_OK;
  const int64_t nblk = (total + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(edge_prepare_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, P);
  return launch_status();
}

extern "C" int ggnn_period_gat_aggregate(const ggnn_aggregate_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  const ggnn_aggregate_args& A = *args;
  if (!A.rowptr || !A.p_src || !A.p_dst || !A.edge_params || !A.agg) return GGNN_EINVAL;
  if (A.E > 0 && (!A.col || !A.einfo || !aligned16(A.einfo))) return GGNN_EINVAL;
  if (A.n_dst <= 0 || A.n_src <= 0 || A.E < 0) return GGNN_EINVAL;
  const int G = A.n_gates;
  if (G != 1 && G != 3 && G != 4) return GGNN_EINVAL;
  if (A.kv_off < 0 || A.q_off < 0 || A.a_off < 0 || A.sc_off < 0 || A.a_gstride < C) return GGNN_EINVAL;
  if (A.kv_off + (int64_t)G * 2 * C > A.ldp_src || A.q_off + (int64_t)G * C > A.ldp_dst) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.a_off + C > A.ld_agg) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.sc_off + 2 > A.ld_agg) return GGNN_EINVAL;
  const int64_t n_items = A.n_dst * G;
  const int64_t want = (n_items + 7) / 8;  // one item per half-wave
  const int64_t nblk = want < (int64_t)AG_NUM_CU * AG_BLOCKS_PER_CU ? want : (int64_t)AG_NUM_CU * AG_BLOCKS_PER_CU;
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
