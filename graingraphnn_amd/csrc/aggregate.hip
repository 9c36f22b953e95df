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
// The sweep walks the *unit table* built with the CSR (a unit = one destination row x up to
// 3 in-edges, 32-byte descriptor {i, p0, nact|first|last, -, j0, j1, j2, -}).  The workgroups
// that share an XCD own one contiguous eighth of the destination rows and deal them
// round-robin to their waves, so an XCD always works inside a short sliding window of
// neighbouring rows and the 3-6 re-reads of a source row hit its L2 (rocprofv3: 63 % L2
// misses and 2x the algorithmic bytes fetched with per-workgroup contiguous ranges).
// A WAVE takes such a row stream and a pair of gates (one gate per half-wave: 32 lanes x 3 channels, so one
// global_load_dwordx3 is one 384-byte row fragment) and streams through the sub-range's
// units.  Both halves see the same units, so everything that describes a unit is
// wave-uniform and lives on the scalar side:
//   * descriptor and the 3 edge records (reloc, a_e) come through scalar loads (SMEM) into
//     SGPRs; the scalar chain for unit u+1 (descriptor -> its edge records) runs while the
//     vector loads of unit u are in flight, and does not touch the vector-memory counter;
//   * the vector side of a unit is exactly 7 loads issued back to back: Q and the 3 x (K, V)
//     fragments -- one exposed round trip per unit, ~21 VGPRs of load state, which keeps the
//     kernel at 8 waves per SIMD (64 half-wave streams, ~170 KB requested per CU);
//   * scores are folded into an online-max softmax carried in registers across the units
//     of one row (any degree, bounded registers, nothing re-read); the row is stored once,
//     when its last unit is done.  No atomics => bit-reproducible.
// Only the 7 x 96 per-gate edge parameters go through LDS.  (Measured alternatives, in git
// history and profiles/: v1 block-staged CSR 64 us; v2 per-half-wave CSR walk 53 us; v3
// per-wave LDS-DMA gather ring 63 us -- 86 KB/CU in flight but issue-bound at one wave per
// SIMD; v4 unit table + vector-side descriptor prefetch 41 us.)
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

template <int G>
__global__ __launch_bounds__(256) void aggregate_kernel(const ggnn_aggregate_args A) {
  using SH = Shape<G>;
  // [g][channel][8]: 7 parameters of one channel contiguous (32 B) -> two 16-byte LDS reads
  __shared__ __attribute__((aligned(16))) float s_ep[G * C * 8];
  const int tid = threadIdx.x;
  for (int t = tid; t < G * GGNN_EDGE_PARAM_ROWS * C; t += 256) {
    const int g = t / (GGNN_EDGE_PARAM_ROWS * C), r = (t / C) % GGNN_EDGE_PARAM_ROWS, c = t % C;
    s_ep[(g * C + c) * 8 + r] = A.edge_params[t];
  }
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave % SH::pairs, sub = wave / SH::pairs;
  const int g = pair * 2 + ((tid >> 5) & 1);  // the gate of this half-wave
  const bool active = g < G;                  // odd G: the last pair's upper half idles
  const int gc = active ? g : 0;
  const int ch = 3 * (tid & 31);              // first of this lane's three channels
  const float inv_sqrt_c = 0.10206207261596577f;  // 1/sqrt(96), periodGATconv.py:226

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

  // this lane's 3 channels x 8 parameters: [wkx wky wkz wvx | wvy wvz we -]
  const f32x4* epp = reinterpret_cast<const f32x4*>(&s_ep[(gc * C + ch) * 8]);
  const f32x4 e0a = epp[0], e0b = epp[1], e1a = epp[2], e1b = epp[3], e2a = epp[4], e2b = epp[5];
  const float* kvbase = A.p_src + A.kv_off + gc * 2 * C + ch;
  const float* qbase = A.p_dst + A.q_off + gc * C + ch;
  const uint32_t ldp_src = (uint32_t)A.ldp_src, ldp_dst = (uint32_t)A.ldp_dst;

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
    // ---- vector side: Q + 3 x (K, V), unconditional and back to back (absent edges repeat
    // j0 in the descriptor, so every address is valid; they are masked when folded) ----
    f3 q = {0.f, 0.f, 0.f}, kk[UE], vv[UE];
#pragma unroll
    for (int t = 0; t < UE; ++t) {
      kk[t] = {0.f, 0.f, 0.f};
      vv[t] = {0.f, 0.f, 0.f};
    }
    if (active) {
      q = ld3_nt(qbase + (uint32_t)i * ldp_dst);  // read once; host checked: n * ld < 2^31
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        const float* row = kvbase + (uint32_t)d[4 + t] * ldp_src;
        kk[t] = ld3(row);
        vv[t] = ld3(row + C);
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
          const float rx = ed[t].x, ry = ed[t].y, rz = ed[t].z, a = ed[t].w;
          const float k0 = kk[t].x + e0a.x * rx + e0a.y * ry + e0a.z * rz + e0b.z * a;
          const float k1 = kk[t].y + e1a.x * rx + e1a.y * ry + e1a.z * rz + e1b.z * a;
          const float k2 = kk[t].z + e2a.x * rx + e2a.y * ry + e2a.z * rz + e2b.z * a;
          s[t] = halfwave_sum(q.x * k0 + q.y * k1 + q.z * k2) * inv_sqrt_c;
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
          acc.x += pe * fmaxf(vv[t].x + e0a.w * rx + e0b.x * ry + e0b.y * rz, 0.f);
          acc.y += pe * fmaxf(vv[t].y + e1a.w * rx + e1b.x * ry + e1b.y * rz, 0.f);
          acc.z += pe * fmaxf(vv[t].z + e2a.w * rx + e2b.x * ry + e2b.y * rz, 0.f);
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

// ---------------------------------------------------------------------------------------
// encoder sweep: h = c = 0, so K0 / V0 / Q are affine maps of the 8- or 11-float feature rows.
// Every input of a unit (descriptor, edge records, the destination's and the three sources'
// feature rows) is wave-uniform and arrives through scalar loads; the lanes only hold the
// 120 weights of their gate and three channels.  The inner loop has NO vector-memory load:
// it is pure VALU (scalar x vector FMAs) plus the row store.
// ---------------------------------------------------------------------------------------
template <int FS, int FD>
__global__ __launch_bounds__(256) void aggregate_enc_kernel(const ggnn_aggregate_enc_args A) {
  constexpr int G = 3;
  using SH = Shape<G>;
  constexpr int KS = FS - 3;  // source features beyond xyz
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave % SH::pairs, sub = wave / SH::pairs;
  const int g = pair * 2 + ((tid >> 5) & 1);
  const bool active = g < G;
  const int gc = active ? g : 0;
  const int ch = 3 * (tid & 31);
  const float inv_sqrt_c = 0.10206207261596577f;

  // this lane's weights: 3 channels x 40 floats
  f32x4 w[3][GGNN_ENC_W_ROW / 4];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < GGNN_ENC_W_ROW / 4; ++k)
      w[c][k] = *reinterpret_cast<const f32x4*>(A.enc_w + ((int64_t)(gc * C + ch + c)) * GGNN_ENC_W_ROW + 4 * k);
  auto W = [&](int c, int idx) { return w[c][idx >> 2][idx & 3]; };

  const int nblk = gridDim.x;
  const int ngrp = min(nblk, 8);
  const int grp = blockIdx.x % ngrp, lb = blockIdx.x / ngrp;
  const int nb_grp = (nblk - grp + ngrp - 1) / ngrp;
  const int64_t x_lo = A.n_dst * grp / ngrp, x_hi = A.n_dst * (grp + 1) / ngrp;
  const int64_t n_streams = (int64_t)nb_grp * SH::subs;
  int64_t r = x_lo + (int64_t)lb * SH::subs + sub;
  if (sub >= SH::subs || r >= x_hi) return;
  const const_i32_ptr uptr = (const_i32_ptr)(uintptr_t)A.unit_ptr;
  const const_i32x8_ptr udesc = (const_i32x8_ptr)(uintptr_t)A.units;
  const const_f32x4_ptr einfo = (const_f32x4_ptr)(uintptr_t)A.einfo;
  typedef const float __attribute__((address_space(4))) * const_f32_ptr;
  const const_f32_ptr xs = (const_f32_ptr)(uintptr_t)A.x_src;
  const const_f32_ptr xd = (const_f32_ptr)(uintptr_t)A.x_dst;

  float mx = -INFINITY, den = 0.f, sae = 0.f;
  f3 acc = {0.f, 0.f, 0.f};
  f3 q = {0.f, 0.f, 0.f};

  int u = uptr[r], u_end = uptr[r + 1];
  i32x8 d = udesc[u];
  f32x4 ed[UE];
#pragma unroll
  for (int t = 0; t < UE; ++t) ed[t] = einfo[(int64_t)d[1] + t];
  // feature rows of a unit (scalar loads; absent edges repeat j0, so always valid)
  float xi[FD], xj[UE][KS];
#pragma unroll
  for (int f = 0; f < FD; ++f) xi[f] = xd[(int64_t)d[0] * FD + f];
#pragma unroll
  for (int t = 0; t < UE; ++t)
#pragma unroll
    for (int f = 0; f < KS; ++f) xj[t][f] = xs[(int64_t)d[4 + t] * FS + 3 + f];

  while (true) {
    const int i = d[0], nact = d[2] & 0xFF;
    const bool first = (d[2] >> 8) & 1, last = (d[2] >> 9) & 1;
    // scalar side of the next unit
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
    // ... including its feature rows: the whole scalar chain of unit u+1 runs under unit u's math
    float xin[FD], xjn[UE][KS];
#pragma unroll
    for (int f = 0; f < FD; ++f) xin[f] = xd[(int64_t)dn[0] * FD + f];
#pragma unroll
    for (int t = 0; t < UE; ++t)
#pragma unroll
      for (int f = 0; f < KS; ++f) xjn[t][f] = xs[(int64_t)dn[4 + t] * FS + 3 + f];

    if (first) {
      mx = -INFINITY;
      den = 0.f;
      sae = 0.f;
      acc = {0.f, 0.f, 0.f};
      // query of this row: lin_query(x_i) (periodGATconv.py:216), x_i is NOT wrapped
      float qq[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float a = W(c, 12);
#pragma unroll
        for (int f = 0; f < FD; ++f) a += W(c, f) * xi[f];
        qq[c] = a;
      }
      q = {qq[0], qq[1], qq[2]};
    }
    if (active && nact > 0) {
      float s[UE];
      float vv[UE][3];
      float mnew = mx;
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        s[t] = -INFINITY;
        if (t < nact) {
          const float rx = ed[t].x, ry = ed[t].y, rz = ed[t].z, a = ed[t].w;
          float part = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float k = W(c, 21) + W(c, 31) * rx + W(c, 32) * ry + W(c, 33) * rz + W(c, 37) * a;
            float v = W(c, 30) + W(c, 34) * rx + W(c, 35) * ry + W(c, 36) * rz;
#pragma unroll
            for (int f = 0; f < KS; ++f) {
              k += W(c, 13 + f) * xj[t][f];
              v += W(c, 22 + f) * xj[t][f];
            }
            vv[t][c] = fmaxf(v, 0.f);
            part += (c == 0 ? q.x : (c == 1 ? q.y : q.z)) * k;
          }
          s[t] = halfwave_sum(part) * inv_sqrt_c;
          mnew = fmaxf(mnew, s[t]);
        }
      }
      const float scale = __expf(mx - mnew);
      den *= scale;
      sae *= scale;
      acc = {acc.x * scale, acc.y * scale, acc.z * scale};
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        if (t < nact) {
          const float pe = __expf(s[t] - mnew);
          den += pe;
          sae += pe * ed[t].w;
          acc.x += pe * vv[t][0];
          acc.y += pe * vv[t][1];
          acc.z += pe * vv[t][2];
        }
      }
      mx = mnew;
    }
    if (active && last) {
      const float inv = 1.0f / (den + 1e-16f);
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
#pragma unroll
    for (int f = 0; f < FD; ++f) xi[f] = xin[f];
#pragma unroll
    for (int t = 0; t < UE; ++t)
#pragma unroll
      for (int f = 0; f < KS; ++f) xj[t][f] = xjn[t][f];
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
  const ggnn_aggregate_args& A = *args;
  if (!A.unit_ptr || !A.units || !A.einfo || !A.p_src || !A.p_dst || !A.edge_params || !A.agg)
    return GGNN_EINVAL;
  if (!aligned16(A.units) || !aligned16(A.einfo)) return GGNN_EINVAL;
  if (A.n_dst <= 0 || A.n_src <= 0 || A.E < 0) return GGNN_EINVAL;
  const int G = A.n_gates;
  if (G != 1 && G != 3 && G != 4) return GGNN_EINVAL;
  if (A.kv_off < 0 || A.q_off < 0 || A.a_off < 0 || A.sc_off < 0 || A.a_gstride < C) return GGNN_EINVAL;
  if (A.ldp_src <= 0 || A.ldp_dst <= 0 || A.n_src * A.ldp_src >= INT32_MAX || A.n_dst * A.ldp_dst >= INT32_MAX)
    return GGNN_EINVAL;  // the sweep forms row offsets in 32 bits
  if (A.kv_off + (int64_t)G * 2 * C > A.ldp_src || A.q_off + (int64_t)G * C > A.ldp_dst) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.a_off + C > A.ld_agg) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.sc_off + 2 > A.ld_agg) return GGNN_EINVAL;
  // persistent grid: at least 4 rows per workgroup, at most the resident capacity
  const int64_t want = (A.n_dst + 3) / 4;
  const int64_t cap = (int64_t)AG_NUM_CU * AG_BLOCKS_PER_CU;
  const dim3 grid((unsigned)(want < cap ? want : cap));
  hipStream_t s = (hipStream_t)stream;
  if (G == 4)
    hipLaunchKernelGGL(aggregate_kernel<4>, grid, dim3(256), 0, s, A);
  else if (G == 3)
    hipLaunchKernelGGL(aggregate_kernel<3>, grid, dim3(256), 0, s, A);
  else
    hipLaunchKernelGGL(aggregate_kernel<1>, grid, dim3(256), 0, s, A);
  return launch_status();
}

extern "C" int ggnn_period_gat_aggregate_enc(const ggnn_aggregate_enc_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  const ggnn_aggregate_enc_args& A = *args;
  if (!A.unit_ptr || !A.units || !A.einfo || !A.x_src || !A.x_dst || !A.enc_w || !A.agg) return GGNN_EINVAL;
  if (!aligned16(A.units) || !aligned16(A.einfo) || !aligned16(A.enc_w)) return GGNN_EINVAL;
  if (A.n_dst <= 0 || A.n_src <= 0 || A.E < 0 || A.n_gates != 3) return GGNN_EINVAL;
  if (A.a_off < 0 || A.sc_off < 0 || A.a_gstride < C) return GGNN_EINVAL;
  if (2LL * A.a_gstride + A.a_off + C > A.ld_agg || 2LL * A.a_gstride + A.sc_off + 2 > A.ld_agg) return GGNN_EINVAL;
  const int64_t want = (A.n_dst + 3) / 4;
  const int64_t cap = (int64_t)AG_NUM_CU * 3;  // ~150 VGPRs -> three workgroups per CU
  const dim3 grid((unsigned)(want < cap ? want : cap)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (A.f_src == 11 && A.f_dst == 8)
    hipLaunchKernelGGL((aggregate_enc_kernel<11, 8>), grid, block, 0, s, A);
  else if (A.f_src == 8 && A.f_dst == 11)
    hipLaunchKernelGGL((aggregate_enc_kernel<8, 11>), grid, block, 0, s, A);
  else if (A.f_src == 8 && A.f_dst == 8)
    hipLaunchKernelGGL((aggregate_enc_kernel<8, 8>), grid, block, 0, s, A);
  else
    return GGNN_EINVAL;
  return launch_status();
}
