// Backward of the periodic-boundary GAT aggregation sweep (training path): the
// gradient of ggnn_period_gat_aggregate with respect to its floating-point operands.  Replaces
// what autograd does for PeriodConv.message + PyG propagate (periodGATconv.py:174-175, 204-236):
// segment-softmax backward, the relu mask, scatter of the value gradient to the source rows.
//
// Forward, per destination i, gate g, in-edge e (source j):
//   s_e = u4_i . x4_e + u_h,i . h_j          alpha_e = exp(s_e - max) / (sum + 1e-16)
//   val_e = relu(V_j + W3 r_e)               out_i = sum alpha_e val_e
//   den_i = sum alpha_e                      sae_i = sum alpha_e a_e
// Backward, given (g_out, g_den, g_sae) per destination and gate:
//   dalpha_e = g_out . val_e + g_den + g_sae a_e
//   ds_e     = alpha_e (dalpha_e - S_i),  S_i = sum_e alpha_e dalpha_e = g_out . out_i + g_den den_i + g_sae sae_i
//                                         (taken from the saved forward output: no extra pass)
//   du4_i   += ds_e x4_e      du_h,i += ds_e h_j      dh_j += ds_e u_h,i  (summed over gates)
//   dV_j    += alpha_e g_out [val_e > 0]              dW3  += (alpha_e g_out [val_e > 0]) r_e^T
//
// Two atomics-free passes, both with the forward sweep's lane layout (a 16-lane row per gate,
// channels ch..ch+2 and 48+ch..48+ch+2 per lane):
//   dst pass  one wave per destination row (grid-stride): recomputes the scores (online max and
//             sum, then the gradients), writes the destination-side gradients, the per-edge
//             (alpha, ds) records and this wave's partial sum of dW3 (reduced by the caller:
//             a fixed grid, so the result is reproducible);
//   src pass  one wave per source row over the REVERSE (source-grouped) CSR: gathers the
//             destination rows' g_out / u_h and the edge records, writes dV_j and dh_j.
// Round 3: both passes take the edges in units of three with all loads of a unit in flight together.
#include "common.h"

namespace ggnn {

constexpr int AB_WAVES = 4;

struct BwdLane {
  int wave, g, gc, l16, ch;
  bool active;
};

template <int G>
__device__ __forceinline__ BwdLane bwd_lane() {
  BwdLane L;
  const int tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  L.g = (tid >> 4) & 3;
  L.active = L.g < G;
  L.gc = L.active ? L.g : 0;
  L.l16 = tid & 15;
  L.ch = 3 * L.l16;
  return L;
}

__device__ __forceinline__ void ld6(const float* p, float (&v)[6]) {
  const f3 a = ld3(p), b = ld3(p + C / 2);
  v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = b.x, v[4] = b.y, v[5] = b.z;
}
__device__ __forceinline__ void st6(float* p, const float (&v)[6]) {
  st3(p, {v[0], v[1], v[2]});
  st3(p + C / 2, {v[3], v[4], v[5]});
}

// Destination pass, round 3: the in-edges of a row are taken in UNITS of three with every load of a unit issued
// back to back and unconditionally (clamped indices: a load under `if` drags a wait to the branch merge), so a
// row of degree <= 3 -- every junction -- costs two memory round trips (hidden rows + tails, then values + records)
// instead of six; the scores of such a row stay in registers between the two halves.  Rows of higher degree run the
// same unit code twice (online max / sum, then the gradients).  Arithmetic and summation order of the first version.
template <int G, bool HAS_H>
__global__ __launch_bounds__(256) void aggregate_bwd_dst_kernel(const ggnn_aggregate_bwd_args A) {
  const BwdLane L = bwd_lane<G>();
  const int64_t w = (int64_t)blockIdx.x * AB_WAVES + L.wave, n_w = (int64_t)gridDim.x * AB_WAVES;
  constexpr int U = GGNN_UNIT_EDGES;
  float wv[6][3];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const float* e = A.edge_params + L.gc * GGNN_EDGE_PARAM_ROWS * C + L.ch + (c < 3 ? c : C / 2 + c - 3);
    wv[c][0] = e[0], wv[c][1] = e[C], wv[c][2] = e[2 * C];
  }
  float dwv[6][3] = {};
  const int e_last = (int)max(A.E - 1, (int64_t)0);
  const float* __restrict__ vsrc = A.p_src + A.v_off + L.gc * C + L.ch;
  const float* __restrict__ hsrc = A.h_src + L.ch;
  for (int64_t i = w; i < A.n_dst; i += n_w) {
    const int beg = A.rowptr[i], end = A.rowptr[i + 1];
    float uh[6] = {}, u4 = 0.f, go[6] = {}, out[6] = {}, gden = 0.f, gsae = 0.f, oden = 0.f, osae = 0.f;
    if (L.active) {
      const float* pd = A.p_dst + i * A.ldp_dst;
      if (HAS_H) ld6(pd + A.u_off + L.gc * C + L.ch, uh);
      u4 = pd[A.u4_off + L.gc * 16 + L.l16];
      const int64_t o = i * A.ld_agg + (int64_t)L.gc * A.a_gstride;
      ld6(A.g_agg + o + A.a_off + L.ch, go);
      ld6(A.agg + o + A.a_off + L.ch, out);
      gden = A.g_agg[o + A.sc_off], gsae = A.g_agg[o + A.sc_off + 1];
      oden = A.agg[o + A.sc_off], osae = A.agg[o + A.sc_off + 1];
    }
    // ---- scores: online max and sum over the units ----
    float mx = -INFINITY, den = 0.f;
    float h[U][6], x4[U], sc[U];
    int64_t jj[U];
    auto load_scores = [&](int p0) {   // hidden rows and score tails of the unit at p0
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int p = min(p0 + t, e_last);
        jj[t] = A.E > 0 ? (int64_t)A.col[p] : 0;
        x4[t] = A.einfo[(int64_t)p * GGNN_EINFO_ROW + L.l16];
        if (HAS_H) ld6(hsrc + jj[t] * A.ldh_src, h[t]);
      }
    };
    auto unit_scores = [&](int p0) {
#pragma unroll
      for (int t = 0; t < U; ++t) {
        float part = u4 * x4[t];
        if (HAS_H) {
#pragma unroll
          for (int c = 0; c < 6; ++c) part += uh[c] * h[t][c];
        }
        sc[t] = row_sum(part);
        if (p0 + t < end) {
          const float mn = fmaxf(mx, sc[t]);
          den = den * __expf(mx - mn) + __expf(sc[t] - mn);
          mx = mn;
        }
      }
    };
    const bool one_unit = end - beg <= U;
    for (int p0 = beg; p0 < end; p0 += U) {
      load_scores(p0);
      unit_scores(p0);
    }
    const float inv = 1.0f / (den + 1e-16f);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < 6; ++c) dot += go[c] * out[c];
    const float S = row_sum(dot) + gden * oden + gsae * osae;
    // ---- gradients, unit by unit (a single unit still has its hidden rows, tails and scores in registers) ----
    float du4 = 0.f, duh[6] = {};
    for (int p0 = beg; p0 < end; p0 += U) {
      float v[U][6], rec[U][4];
      if (!one_unit) load_scores(p0);
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int p = min(p0 + t, e_last);
        const float* r = A.einfo + (int64_t)p * GGNN_EINFO_ROW + 16;
        rec[t][0] = r[0], rec[t][1] = r[1], rec[t][2] = r[2], rec[t][3] = r[3];
        ld6(vsrc + jj[t] * A.ldp_src, v[t]);
      }
      if (!one_unit) {   // (scores again, without touching the softmax state)
#pragma unroll
        for (int t = 0; t < U; ++t) {
          float part = u4 * x4[t];
          if (HAS_H) {
#pragma unroll
            for (int c = 0; c < 6; ++c) part += uh[c] * h[t][c];
          }
          sc[t] = row_sum(part);
        }
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int p = p0 + t;
        const bool live = p < end;
        const float rx = rec[t][0], ry = rec[t][1], rz = rec[t][2], ae = rec[t][3];
        const float alpha = live ? __expf(sc[t] - mx) * inv : 0.f;
        float gv[6], dpart = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const float val = v[t][c] + wv[c][0] * rx + wv[c][1] * ry + wv[c][2] * rz;
          gv[c] = val > 0.f ? go[c] : 0.f;  // g_out masked by the relu
          dpart += gv[c] * val;
        }
        const float dalpha = row_sum(dpart) + gden + gsae * ae;
        const float ds = live ? alpha * (dalpha - S) : 0.f;
        du4 += ds * x4[t];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          if (HAS_H) duh[c] += ds * h[t][c];
          const float tt = alpha * gv[c];
          dwv[c][0] += tt * rx, dwv[c][1] += tt * ry, dwv[c][2] += tt * rz;
        }
        if (live && L.active && L.l16 == 0) {
          A.edge_alpha[(int64_t)p * G + L.g] = alpha;
          A.edge_ds[(int64_t)p * G + L.g] = ds;
        }
      }
    }
    if (L.active) {
      float* gd = A.g_p_dst + i * A.ldp_dst;
      gd[A.u4_off + L.g * 16 + L.l16] = du4;
      if (HAS_H) st6(gd + A.u_off + L.g * C + L.ch, duh);
    }
  }
  // this workgroup's share of dW3, [n_partials = workgroups][G][3][96]: the four waves' sums are added in wave order
  // through LDS (a fixed grid and a fixed order: the caller's reduction over the partials is reproducible)
  __shared__ float red[AB_WAVES][G][GGNN_EDGE_PARAM_ROWS * C];
  if (L.active) {
    float* o = &red[L.wave][L.g][L.ch];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 6; ++c) o[k * C + (c < 3 ? c : C / 2 + c - 3)] = dwv[c][k];
  }
  __syncthreads();
  constexpr int NP = G * GGNN_EDGE_PARAM_ROWS * C;
  float* __restrict__ po = A.ep_partial + (int64_t)blockIdx.x * NP;
  for (int idx = threadIdx.x; idx < NP; idx += 256) {
    const float* r = &red[0][0][0] + idx;
    po[idx] = ((r[0] + r[NP]) + r[2 * NP]) + r[3 * NP];
  }
}

// Source pass, round 3: the out-edges of a source row in units of three, every load of a unit in flight together
// (the reverse-CSR entries first, then the destination rows' g_out / u_h and the edge records they point to).
template <int G, bool HAS_H>
__global__ __launch_bounds__(256) void aggregate_bwd_src_kernel(const ggnn_aggregate_bwd_args A) {
  const BwdLane L = bwd_lane<G>();
  const int64_t w = (int64_t)blockIdx.x * AB_WAVES + L.wave, n_w = (int64_t)gridDim.x * AB_WAVES;
  constexpr int U = GGNN_UNIT_EDGES;
  float wv[6][3];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const float* e = A.edge_params + L.gc * GGNN_EDGE_PARAM_ROWS * C + L.ch + (c < 3 ? c : C / 2 + c - 3);
    wv[c][0] = e[0], wv[c][1] = e[C], wv[c][2] = e[2 * C];
  }
  const int e_last = (int)max(A.E - 1, (int64_t)0);
  const float* __restrict__ gagg = A.g_agg + (int64_t)L.gc * A.a_gstride + A.a_off + L.ch;
  const float* __restrict__ udst = A.p_dst + A.u_off + L.gc * C + L.ch;
  for (int64_t j = w; j < A.n_src; j += n_w) {
    const int beg = A.r_rowptr[j], end = A.r_rowptr[j + 1];
    float v[6], dv[6] = {}, dh[6] = {};
    ld6(A.p_src + j * A.ldp_src + A.v_off + L.gc * C + L.ch, v);
    // g_h_accumulate: the cell's earlier sweep from this source node type wrote the row -- add to it (requested here, used
    // behind the row's edges)
    float prev[6] = {};
    if (HAS_H && A.g_h_accumulate) ld6(A.g_h_src + j * A.ldh_src + L.ch, prev);
    for (int q0 = beg; q0 < end; q0 += U) {
      int64_t pp[U], ii[U];
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int q = min(q0 + t, e_last);
        pp[t] = A.E > 0 ? (int64_t)A.r_slot[q] : 0;
        ii[t] = A.E > 0 ? (int64_t)A.r_dst[q] : 0;
      }
      float rec[U][3], alpha[U], ds[U], go[U][6], uh[U][6];
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const float* r = A.einfo + pp[t] * GGNN_EINFO_ROW + 16;
        rec[t][0] = r[0], rec[t][1] = r[1], rec[t][2] = r[2];
        alpha[t] = A.edge_alpha[pp[t] * G + L.gc];
        ds[t] = A.edge_ds[pp[t] * G + L.gc];
        ld6(gagg + ii[t] * A.ld_agg, go[t]);
        if (HAS_H) ld6(udst + ii[t] * A.ldp_dst, uh[t]);
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const bool live = q0 + t < end && L.active;
        const float al = live ? alpha[t] : 0.f, dd = live ? ds[t] : 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const float val = v[c] + wv[c][0] * rec[t][0] + wv[c][1] * rec[t][1] + wv[c][2] * rec[t][2];
          // (the relu mask as a factor, not a branch around the product: under `val > 0 ? al * go : 0` the compiler sank one
          // element of the gathered gradient row behind the branch, a dword load with an s_waitcnt vmcnt(0) of its own --
          // a dependent memory round trip per unit; same value for finite operands)
          dv[c] += (val > 0.f ? al : 0.f) * go[t][c];
          if (HAS_H) dh[c] += dd * uh[t][c];
        }
      }
    }
    if (L.active) st6(A.g_p_src + j * A.ldp_src + A.v_off + L.g * C + L.ch, dv);
    if (HAS_H) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {  // sum over the four gate rows of the wave
        dh[c] += __shfl_xor(dh[c], 16, 64);
        dh[c] += __shfl_xor(dh[c], 32, 64);
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) dh[c] += prev[c];   // (outside the branch below: the load stays at the top of the row)
      if (L.g == 0) st6(A.g_h_src + j * A.ldh_src + L.ch, dh);
    }
  }
}

}  // namespace ggnn

// = workgroups of the destination pass: three per compute unit (the pass is a chain of dependent gathers per row and
// holds ~160 registers: twelve waves per compute unit are what fits, and all of them should be resident at once)
extern "C" int64_t ggnn_aggregate_bwd_partials(int64_t n_dst) {
  const int64_t want = (n_dst + ggnn::AB_WAVES - 1) / ggnn::AB_WAVES;
  return want < 768 ? (want > 0 ? want : 1) : 768;
}

extern "C" int ggnn_period_gat_aggregate_backward(const ggnn_aggregate_bwd_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  ggnn_aggregate_bwd_args A = *args;
  if (!A.rowptr || !A.einfo || !A.p_src || !A.p_dst || !A.edge_params || !A.agg || !A.g_agg || !A.r_rowptr ||
      !A.edge_alpha || !A.edge_ds || !A.ep_partial || !A.g_p_dst || !A.g_p_src)
    return GGNN_EINVAL;
  if (A.n_dst <= 0 || A.n_src <= 0 || A.E < 0) return GGNN_EINVAL;
  if (A.E > 0 && (!A.col || !A.r_dst || !A.r_slot)) return GGNN_EINVAL;
  const int G = A.n_gates;
  if (G != 1 && G != 3 && G != 4) return GGNN_EINVAL;
  const bool has_h = A.h_src != nullptr;
  if (has_h && (!A.g_h_src || A.ldh_src < C || A.u_off < 0)) return GGNN_EINVAL;
  if (!has_h) {
    A.h_src = A.p_src;
    A.ldh_src = 0;
    A.u_off = 0;
  }
  if (A.v_off < 0 || A.u4_off < 0 || A.a_off < 0 || A.sc_off < 0 || A.a_gstride < C) return GGNN_EINVAL;
  if (A.ldp_src <= 0 || A.ldp_dst <= 0) return GGNN_EINVAL;
  if (A.v_off + (int64_t)G * C > A.ldp_src || A.u4_off + (int64_t)G * 16 > A.ldp_dst) return GGNN_EINVAL;
  if (has_h && A.u_off + (int64_t)G * C > A.ldp_dst) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.a_off + C > A.ld_agg) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.sc_off + 2 > A.ld_agg) return GGNN_EINVAL;
  if (A.n_partials != ggnn_aggregate_bwd_partials(A.n_dst)) return GGNN_EINVAL;
  if (A.g_h_accumulate != 0 && A.g_h_accumulate != 1) return GGNN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid_d((unsigned)A.n_partials);
  const int64_t want_s = (A.n_src + AB_WAVES - 1) / AB_WAVES;
  const dim3 grid_s((unsigned)(want_s < 2048 ? want_s : 2048));
#define GGNN_AB_LAUNCH(G_)                                                                          \
  do {                                                                                              \
    if (has_h) {                                                                                    \
      hipLaunchKernelGGL((aggregate_bwd_dst_kernel<G_, true>), grid_d, dim3(256), 0, s, A);         \
      hipLaunchKernelGGL((aggregate_bwd_src_kernel<G_, true>), grid_s, dim3(256), 0, s, A);         \
    } else {                                                                                        \
      hipLaunchKernelGGL((aggregate_bwd_dst_kernel<G_, false>), grid_d, dim3(256), 0, s, A);        \
      hipLaunchKernelGGL((aggregate_bwd_src_kernel<G_, false>), grid_s, dim3(256), 0, s, A);        \
    }                                                                                               \
  } while (0)
  if (G == 4) GGNN_AB_LAUNCH(4);
  else if (G == 3) GGNN_AB_LAUNCH(3);
  else GGNN_AB_LAUNCH(1);
#undef GGNN_AB_LAUNCH
  return launch_status();
}
