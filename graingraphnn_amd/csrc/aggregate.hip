// Periodic-boundary GAT aggregation for one edge type, all gates fused (HBM-bound).
// Replaces PeriodConv.message (periodGATconv.py:204-236) and the gather / scatter-add of
// PyG MessagePassing.propagate (periodGATconv.py:174-175) with an atomics-free CSR sweep.
//
// Per edge e = (j -> i) and gate g, with reloc_e = minimg(x_j[:3] - x_i[:3]) computed exactly as
// periodGATconv.py:209-210 does, a_e = edge_attr_e and x~_j = [reloc_e, x_j[3:F], h_j]:
//   s_e   = q_i . (W_k x~_j + b_k + w_edge a_e) / sqrt(96)                       (:216-226)
//         = u_i . x~_j + s1_i + a_e s2_i       with  u_i = W_k^T q_i / sqrt(96),
//                                                    s1_i = b_k . q_i / sqrt(96), s2_i = w_edge . q_i / sqrt(96)
//   alpha = exp(s_e - max_i) / (sum_i exp(.) + 1e-16)                            (PyG softmax)
//   r_e   = relu(V0_j + Wv3 . reloc_e)          V0 = value WITHOUT its first three input columns
//   agg_i = sum_e alpha_e r_e,  sa_i = sum_e alpha_e,  sae_i = sum_e alpha_e a_e
// The KEY never exists: the destination-side projection (ggnn_project) delivers u_i, s1_i, s2_i
// (all affine in the destination's [x_i | h_i]: one row of W_k^T W_q per source feature), and
// the sweep dots u_i with the source's RAW [x_j | h_j] row.  Against gathering a projected key
// this removes a quarter of the projection's output columns, and the 384-byte h_j row is shared
// by all gates where the four 384-byte key fragments were not; in the encoder (h = 0) the key
// side shrinks to the 8 / 11 feature floats.
//   per (destination, gate):  u_h [96] | u4 [16] = (u_x[0..F), 0.., s1 @12, s2 @13, 0, 0)
//   per edge:                 h_j [96] | x4 [16] = (reloc, x_j[3..F), 0.., 1 @12, a_e @13, 0, 0)
//   s_e = sum over the half-wave of  u_h . h_j (3 channels per lane) + u4 * x4 (lanes 0..15)
//
// ggnn_edge_prepare computes (reloc_e, a_e) once per forward in CSR order (16 bytes per
// edge, shared by the 7 gate sweeps of encoder + decoder).
//
// The sweep walks the *unit table* built with the CSR (a unit = one destination row x up to
// 3 in-edges, 32-byte descriptor {i, p0, nact|first|last, -, j0, j1, j2, -}).  The workgroups
// that share an XCD own one contiguous eighth of the destination rows and deal them
// round-robin to their waves, so an XCD always works inside a short sliding window of
// neighbouring rows and the 3-6 re-reads of a source row hit its L2 (rocprofv3: 63 % L2
// misses and 2x the algorithmic bytes fetched with per-workgroup contiguous ranges).
// A WAVE takes such a row stream and a pair of gates (one gate per half-wave: 32 lanes x 3
// channels, so one global_load_dwordx3 is one 384-byte row fragment).  Both halves see the same
// units, so everything that describes a unit is wave-uniform and lives on the scalar side:
//   * descriptor and the 3 edge records (reloc, a_e) come through scalar loads (SMEM) into
//     SGPRs; the scalar chain for unit u+1 runs while the vector loads of unit u are in flight;
//   * the vector side of a unit is issued back to back and unconditionally (absent edges repeat
//     j0; a load under `if` would drag a wait to the branch merge): u_h, u4 and per edge h_j, x4,
//     V -- one exposed round trip per unit;
//   * scores are folded into an online-max softmax carried in registers across the units of one
//     row (any degree, bounded registers, nothing re-read); the row is stored once, when its last
//     unit is done.  No atomics => bit-reproducible.
// (Measured alternatives, in git history and profiles/: v1 block-staged CSR 64 us; v2
// per-half-wave CSR walk 53 us; v3 per-wave LDS-DMA gather ring 63 us; v4 unit table + vector-side
// descriptor prefetch 41 us; v7 projected keys gathered per gate 31 us; a fused encoder sweep
// recomputing K0 / V0 / Q per edge from feature rows: VALU-bound, slower.)
#include "common.h"

namespace ggnn {

#ifndef AG_VAR_BPC
#define AG_VAR_BPC 7
#endif
constexpr int AG_BLOCKS_PER_CU = AG_VAR_BPC;  // resident workgroups per CU (68 VGPRs -> 7 waves per SIMD)
constexpr int AG_NUM_CU = 256;
constexpr int UE = GGNN_UNIT_EDGES;

typedef int i32x4 __attribute__((ext_vector_type(4)));

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
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (p < T.E) {  // the last GGNN_UNIT_EDGES records are zero padding
    const float* xs = T.x_src + (int64_t)T.col[p] * T.ldx_src;
    const float* xd = T.x_dst + (int64_t)T.row[p] * T.ldx_dst;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float rel = xs[c] - xd[c];
      const float w = rel > 0.5f ? -1.0f : (rel < -0.5f ? 1.0f : 0.0f);
      o[c] = w + rel;  // periodGATconv.py:210
    }
    o[3] = T.edge_attr[T.perm[p]];
  }
  *reinterpret_cast<f32x4*>(T.einfo + 4 * p) = o;
}

// ---------------------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef const i32x8 __attribute__((address_space(4))) * const_i32x8_ptr;  // uniform index -> s_load_dwordx8
typedef const f32x4 __attribute__((address_space(4))) * const_f32x4_ptr;  // uniform index -> s_load_dwordx4
typedef const int __attribute__((address_space(4))) * const_i32_ptr;

template <int G> struct Shape {
  static constexpr int pairs = (G + 1) / 2;  // gate pairs; a wave sweeps one pair
  static constexpr int waves = 4;
  static constexpr int subs = waves / pairs;  // row sub-ranges per workgroup
};

template <int G, bool HAS_H>
__global__ __launch_bounds__(256) void aggregate_kernel(const ggnn_aggregate_args A) {
  using SH = Shape<G>;
  // [g][channel][4]: W_value[:, 0..2] of one channel (16 B) -> one 16-byte LDS read per channel
  __shared__ __attribute__((aligned(16))) float s_ep[G * C * 4];
  const int tid = threadIdx.x;
  for (int t = tid; t < G * GGNN_EDGE_PARAM_ROWS * C; t += 256) {
    const int g = t / (GGNN_EDGE_PARAM_ROWS * C), r = (t / C) % GGNN_EDGE_PARAM_ROWS, c = t % C;
    s_ep[(g * C + c) * 4 + r] = A.edge_params[t];
  }
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave % SH::pairs, sub = wave / SH::pairs;
  const int g = pair * 2 + ((tid >> 5) & 1);  // the gate of this half-wave
  const bool active = g < G;                  // odd G: the last pair's upper half idles
  const int gc = active ? g : 0;
  const int ch = 3 * (tid & 31);              // first of this lane's three channels
  // lanes 0..15 of the half-wave also carry one element of the 16-wide tail (see the header)
  const int l16 = tid & 15;
  const bool tail = (tid & 16) == 0;
  const float c_rx = tail && l16 == 0, c_ry = tail && l16 == 1, c_rz = tail && l16 == 2;
  const float c_x = tail && l16 >= 3 && l16 < A.f_src, c_one = tail && l16 == 12, c_a = tail && l16 == 13;
  const int xi = min(l16, A.f_src - 1);

  // Row assignment (all wave-uniform).  Workgroups with equal blockIdx % 8 share an XCD and
  // its L2 (observed placement; speed only).  Each such group owns one contiguous eighth of
  // the rows and its streams (one per wave pair-slot) take rows round-robin, so at any moment
  // the whole group works inside one short sliding window of neighbouring rows: the 3-6
  // readers of a source row then run close together in time and the re-reads hit that L2
  // instead of going back to memory.
  const int nblk = gridDim.x;
  const int ngrp = min(nblk, 8);
  const int grp = blockIdx.x % ngrp, lb = blockIdx.x / ngrp;
  const int nb_grp = (nblk - grp + ngrp - 1) / ngrp;            // workgroups in this group
  const int64_t x_lo = A.n_dst * grp / ngrp, x_hi = A.n_dst * (grp + 1) / ngrp;
  const int64_t n_streams = (int64_t)nb_grp * SH::subs;
  int64_t r = x_lo + (int64_t)lb * SH::subs + sub;
  if (sub >= SH::subs || r >= x_hi) return;
  const const_i32_ptr uptr = (const_i32_ptr)(uintptr_t)A.unit_ptr;
  const const_i32x8_ptr udesc = (const_i32x8_ptr)(uintptr_t)A.units;
  const const_f32x4_ptr einfo = (const_f32x4_ptr)(uintptr_t)A.einfo;

  // this lane's 3 channels x (wvx, wvy, wvz)
  const f32x4* epp = reinterpret_cast<const f32x4*>(&s_ep[(gc * C + ch) * 4]);
  const f32x4 w0 = epp[0], w1 = epp[1], w2 = epp[2];
  const float* vbase = A.p_src + A.v_off + gc * C + ch;
  const float* uhbase = A.p_dst + A.u_off + gc * C + ch;
  const float* u4base = A.p_dst + A.u4_off + gc * 16 + l16;
  const float* hbase = A.h_src + ch;
  const float* xbase = A.x_src + xi;
  const uint32_t ldp_src = (uint32_t)A.ldp_src, ldp_dst = (uint32_t)A.ldp_dst;
  const uint32_t ldh = (uint32_t)A.ldh_src, ldx = (uint32_t)A.ldx_src;

  float mx = -INFINITY, den = 0.f, sae = 0.f;
  f3 acc = {0.f, 0.f, 0.f};

  // scalar state of the current unit
  int u = uptr[r], u_end = uptr[r + 1];
  i32x8 d = udesc[u];
  f32x4 ed[UE];
#pragma unroll
  for (int t = 0; t < UE; ++t) ed[t] = einfo[(int64_t)d[1] + t];  // einfo is padded by UE records

  while (true) {
    const int i = d[0], nact = d[2] & 0xFF;
    const bool first = (d[2] >> 8) & 1, last = (d[2] >> 9) & 1;
    // ---- vector side, unconditional and back to back (host checked: n * ld < 2^31) ----
    f3 uh = {0.f, 0.f, 0.f}, hh[UE], vv[UE];
    float u4 = 0.f, x4[UE];
#pragma unroll
    for (int t = 0; t < UE; ++t) {
      hh[t] = {0.f, 0.f, 0.f};
      vv[t] = {0.f, 0.f, 0.f};
      x4[t] = 0.f;
    }
    if (active) {
      if (HAS_H) uh = ld3_nt(uhbase + (uint32_t)i * ldp_dst);  // read once
      u4 = __builtin_nontemporal_load(u4base + (uint32_t)i * ldp_dst);
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        const uint32_t j = (uint32_t)d[4 + t];
        if (HAS_H) hh[t] = ld3(hbase + j * ldh);
        x4[t] = xbase[j * ldx];
        vv[t] = ld3(vbase + j * ldp_src);
      }
    }
    // ---- scalar side for the next unit (same row, or the first unit of this stream's next
    // row), hidden under the vector round trip above ----
    bool more = true;
    int un = u + 1, un_end = u_end;
    int64_t rn = r;
    if (un >= u_end) {
      rn = r + n_streams;
      if (rn < x_hi) {
        un = uptr[rn];
        un_end = uptr[rn + 1];
      } else {
        more = false;
        un = u;
      }
    }
    const i32x8 dn = udesc[un];
    f32x4 edn[UE];
#pragma unroll
    for (int t = 0; t < UE; ++t) edn[t] = einfo[(int64_t)dn[1] + t];

    if (first) {
      mx = -INFINITY;
      den = 0.f;
      sae = 0.f;
      acc = {0.f, 0.f, 0.f};
    }
    if (active && nact > 0) {  // (empty rows have a single unit with nact == 0: zeros are stored)
      float s[UE];
      float mnew = mx;
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        s[t] = -INFINITY;
        if (t < nact) {
          // this lane's element of the tail: reloc / raw feature / 1 / a_e / 0
          const float xe = c_x * x4[t] + c_rx * ed[t].x + c_ry * ed[t].y + c_rz * ed[t].z + c_a * ed[t].w + c_one;
          float part = u4 * xe;
          if (HAS_H) part += uh.x * hh[t].x + uh.y * hh[t].y + uh.z * hh[t].z;
          s[t] = halfwave_sum(part);  // 1/sqrt(96) is folded into u
          mnew = fmaxf(mnew, s[t]);
        }
      }
      const float scale = __expf(mx - mnew);  // exp(-inf) = 0 on a row's first unit
      den *= scale;
      sae *= scale;
      acc = {acc.x * scale, acc.y * scale, acc.z * scale};
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        if (t < nact) {
          const float rx = ed[t].x, ry = ed[t].y, rz = ed[t].z;
          const float pe = __expf(s[t] - mnew);
          den += pe;
          sae += pe * ed[t].w;
          acc.x += pe * fmaxf(vv[t].x + w0.x * rx + w0.y * ry + w0.z * rz, 0.f);
          acc.y += pe * fmaxf(vv[t].y + w1.x * rx + w1.y * ry + w1.z * rz, 0.f);
          acc.z += pe * fmaxf(vv[t].z + w2.x * rx + w2.y * ry + w2.z * rz, 0.f);
        }
      }
      mx = mnew;
    }
    if (active && last) {
      const float inv = 1.0f / (den + 1e-16f);  // PyG softmax denominator
      float* orow = A.agg + (int64_t)i * A.ld_agg + g * A.a_gstride;
      st3_nt(orow + A.a_off + ch, {acc.x * inv, acc.y * inv, acc.z * inv});
      if ((tid & 31) == 0) {
        __builtin_nontemporal_store(den * inv, orow + A.sc_off);
        __builtin_nontemporal_store(sae * inv, orow + A.sc_off + 1);
      }
    }
    if (!more) break;
    u = un;
    u_end = un_end;
    r = rn;
    d = dn;
#pragma unroll
    for (int t = 0; t < UE; ++t) ed[t] = edn[t];
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
      if (T.E < 0 || T.ldx_src < 3 || T.ldx_dst < 3 || !T.einfo || !aligned16(T.einfo)) return GGNN_EINVAL;
      if (T.E > 0 && (!T.col || !T.perm || !T.row || !T.edge_attr || !T.x_src || !T.x_dst))
        return GGNN_EINVAL;
      P.et[k] = T;
      P.e_off[k + 1] = P.e_off[k] + T.E + GGNN_UNIT_EDGES;  // + zero padding records
    } else {
      P.et[k] = ggnn_prepare_edge{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
      P.e_off[k + 1] = P.e_off[k];
    }
  }
  const int64_t total = P.e_off[n_edge_types];
  const int64_t nblk = (total + 255) / 256;
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(edge_prepare_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, P);
  return launch_status();
}

extern "C" int ggnn_period_gat_aggregate(const ggnn_aggregate_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  ggnn_aggregate_args A = *args;
  if (!A.unit_ptr || !A.units || !A.einfo || !A.p_src || !A.p_dst || !A.x_src || !A.edge_params || !A.agg)
    return GGNN_EINVAL;
  if (!aligned16(A.units) || !aligned16(A.einfo)) return GGNN_EINVAL;
  if (A.n_dst <= 0 || A.n_src <= 0 || A.E < 0) return GGNN_EINVAL;
  const int G = A.n_gates;
  if (G != 1 && G != 3 && G != 4) return GGNN_EINVAL;
  if (A.f_src < 3 || A.f_src > 12 || A.ldx_src < A.f_src) return GGNN_EINVAL;
  const bool has_h = A.h_src != nullptr;
  if (has_h && (A.ldh_src < C || A.u_off < 0)) return GGNN_EINVAL;
  if (!has_h) {  // never dereferenced, but the address arithmetic must stay in range
    A.h_src = A.x_src;
    A.ldh_src = 0;
    A.u_off = 0;
  }
  if (A.v_off < 0 || A.u4_off < 0 || A.a_off < 0 || A.sc_off < 0 || A.a_gstride < C) return GGNN_EINVAL;
  if (A.ldp_src <= 0 || A.ldp_dst <= 0 || A.n_src * A.ldp_src >= INT32_MAX || A.n_dst * A.ldp_dst >= INT32_MAX ||
      A.n_src * A.ldx_src >= INT32_MAX || A.n_src * A.ldh_src >= INT32_MAX)
    return GGNN_EINVAL;  // the sweep forms row offsets in 32 bits
  if (A.v_off + (int64_t)G * C > A.ldp_src || A.u4_off + (int64_t)G * 16 > A.ldp_dst) return GGNN_EINVAL;
  if (has_h && A.u_off + (int64_t)G * C > A.ldp_dst) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.a_off + C > A.ld_agg) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.sc_off + 2 > A.ld_agg) return GGNN_EINVAL;
  // persistent grid: at least 4 rows per workgroup, at most the resident capacity
  const int64_t want = (A.n_dst + 3) / 4;
  const int64_t cap = (int64_t)AG_NUM_CU * AG_BLOCKS_PER_CU;
  const dim3 grid((unsigned)(want < cap ? want : cap));
  hipStream_t s = (hipStream_t)stream;
#define GGNN_AG_LAUNCH(G_)                                                                   \
  do {                                                                                       \
    if (has_h) hipLaunchKernelGGL((aggregate_kernel<G_, true>), grid, dim3(256), 0, s, A);   \
    else hipLaunchKernelGGL((aggregate_kernel<G_, false>), grid, dim3(256), 0, s, A);        \
  } while (0)
  if (G == 4) GGNN_AG_LAUNCH(4);
  else if (G == 3) GGNN_AG_LAUNCH(3);
  else GGNN_AG_LAUNCH(1);
#undef GGNN_AG_LAUNCH
  return launch_status();
}
