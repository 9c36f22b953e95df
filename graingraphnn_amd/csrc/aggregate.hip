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
//   s_e = sum over a 16-lane row of  u_h . h_j (6 channels per lane) + u4 * x4 (1 element per lane)
//
// ggnn_edge_prepare writes, once per forward and in CSR order, the 80-byte record of every edge:
// the 16-float tail x4 above and (reloc_e, a_e) again as one aligned float4 for the scalar side;
// it is shared by the 7 gate sweeps of encoder + decoder of both models.
//
// The sweep walks the *unit table* built with the CSR (a unit = one destination row x up to
// 3 in-edges, 32-byte descriptor {i, p0, nact|first|last, -, j0, j1, j2, -}).  The workgroups
// that share an XCD own one contiguous eighth of the destination rows and deal them
// round-robin to their waves, so an XCD always works inside a short sliding window of
// neighbouring rows and the 3-6 re-reads of a source row hit its L2 (rocprofv3: 63 % L2
// misses and 2x the algorithmic bytes fetched with per-workgroup contiguous ranges).
// A WAVE takes such a row stream and ALL gates of its units: one gate per 16-lane row, 6 channels
// per lane (a 384-byte fragment is two global_load_dwordx3 per lane; a row's dot product closes with
// four DPP steps inside the row -- no cross-row traffic; the 16-wide tail is one element per lane).
// Everything that describes a unit is wave-uniform and lives on the scalar side:
//   * descriptor and the 3 edge records (reloc, a_e) come through scalar loads (SMEM) into
//     SGPRs; the scalar chain for unit u+1 runs while the vector loads of unit u are in flight;
//   * the vector side of a unit is issued back to back and unconditionally (absent edges repeat
//     j0; a load under `if` would drag a wait to the branch merge): u_h, u4 and per edge h_j (the
//     SAME 384 bytes for every gate row), the edge's tail record, V -- one exposed round trip;
//   * scores are folded into an online-max softmax carried in registers across the units of one
//     row (any degree, bounded registers, nothing re-read); the row is stored once, when its last
//     unit is done.  No atomics => bit-reproducible.
// The kernel is VALU-issue-bound, not bandwidth-bound (with every vector load AND store removed
// the previous two-gates-per-wave version still took 60 % of its time): what made it faster was
// fewer instructions per (edge, gate) -- the scalar work of a unit is now shared by four gates
// instead of two, the tail arrives ready-made from ggnn_edge_prepare, no half-wave swizzles.
// (Measured alternatives, in git history and profiles/: v1 block-staged CSR 64 us; v2
// per-half-wave CSR walk 53 us; v3 per-wave LDS-DMA gather ring 63 us; v4 unit table + vector-side
// descriptor prefetch 41 us; v7 projected keys gathered per gate 31 us; v12 key-free, two gates per
// wave 31 us, the same with two units in flight 32 us; a fused encoder sweep recomputing
// K0 / V0 / Q per edge from feature rows: VALU-bound, slower.)
#include <algorithm>

#include "common.h"

namespace ggnn {

// Workgroups per CU of the persistent grid (measured on the batched launch, cfg3: 2 -> 80 us,
// 3 -> 70, 4 -> 72, 5 -> 76: fewer streams keep the sliding window of rows, and with it the
// re-reads of source rows, inside the XCD's L2; 116 VGPRs allow 4 waves per SIMD)
constexpr int AG_BLOCKS_PER_CU = 3;
constexpr int AG_BLOCKS_PER_CU_NOH = 6;  // encoder sweep (no hidden rows, 78 VGPRs): 4 -> 47.5 us, 5 -> 46.5, 6 -> 46, 7 -> 51
constexpr int UE = GGNN_UNIT_EDGES;

typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// edge_prepare: the 20-float record of every edge, in CSR order (GGNN_EINFO_ROW floats):
//   [0..15]  tail of the score dot product: reloc_xyz, x_src[3 .. f_src), 0.., 1 @12, a_e @13, 0, 0
//   [16..19] (reloc_x, reloc_y, reloc_z, a_e) for the scalar side, a_e = edge_attr[perm[p]]
// ---------------------------------------------------------------------------------------
struct PrepareArgs {
  ggnn_prepare_edge et[3];
  int b_off[4];  // first workgroup of every edge type (256 records per workgroup)
  int n_et;
};

// REFRESH = true: ggnn_step_refresh_prepare -- the record's edge length is not read from edge_attr but
// recomputed from the xy offsets (the same w + rel the record holds: test.py:570-572) and ALSO written to
// edge_attr[perm[p]]; the z offset uses the clamped z when flags[1] is set (the node part of the launch,
// below, clamps x itself: test.py:405-407).
template <bool REFRESH>
__device__ __forceinline__ void edge_prepare_body(const PrepareArgs& P, int blk, float zmax, bool clamp) {
  // One thread builds one record (one gather of the source row, the destination xyz and the edge
  // length); the workgroup's 256 records are contiguous in memory and leave through LDS as 1 280
  // consecutive 16-byte pieces, so every store instruction covers 1 KB.
  __shared__ __attribute__((aligned(16))) float s_rec[256 * GGNN_EINFO_ROW];
  int k = 0;
  while (k + 1 < P.n_et && blk >= P.b_off[k + 1]) ++k;
  const ggnn_prepare_edge& T = P.et[k];
  const int64_t E = T.E_dev ? *T.E_dev : T.E;   // (E_dev: a captured launch sized for T.E edges on a list that has shrunk in place)
  const int64_t p0 = (int64_t)(blk - P.b_off[k]) * 256, p = p0 + threadIdx.x;
  if (p0 >= E + GGNN_UNIT_EDGES) return;        // (uniform per workgroup: before the barrier below)
  float rec[GGNN_EINFO_ROW];
#pragma unroll
  for (int c = 0; c < GGNN_EINFO_ROW; ++c) rec[c] = 0.f;
  if (p < E) {  // the last GGNN_UNIT_EDGES records are zero padding
    const float* xs = T.x_src + (int64_t)T.col[p] * T.ldx_src;
    const float* xd = T.x_dst + (int64_t)T.row[p] * T.ldx_dst;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float rel = (REFRESH && clamp && c == 2) ? 0.f : xs[c] - xd[c];
      const float w = rel > 0.5f ? -1.0f : (rel < -0.5f ? 1.0f : 0.0f);
      rec[c] = rec[16 + c] = w + rel;  // periodGATconv.py:210
    }
#pragma unroll
    for (int c = 3; c < 12; ++c)
      if (c < T.f_src) rec[c] = xs[c];
    if (T.f_src <= 11) rec[11] = 1.0f;  // bias row of the encoder sweep's value product (aggregate_enc.hip)
    rec[12] = 1.0f;
    if (REFRESH) {
      const float len = sqrtf(rec[0] * rec[0] + rec[1] * rec[1]);  // test.py:572
      const_cast<float*>(T.edge_attr)[T.perm[p]] = len;
      rec[13] = rec[19] = len;
    } else {
      rec[13] = rec[19] = T.edge_attr[T.perm[p]];
    }
  }
  f32x4* mine = reinterpret_cast<f32x4*>(&s_rec[threadIdx.x * GGNN_EINFO_ROW]);
#pragma unroll
  for (int c = 0; c < GGNN_EINFO_ROW / 4; ++c) mine[c] = (f32x4){rec[4 * c], rec[4 * c + 1], rec[4 * c + 2], rec[4 * c + 3]};
  __syncthreads();
  const int64_t n_rec = min((int64_t)256, E + GGNN_UNIT_EDGES - p0);  // records of this workgroup
  const f32x4* src = reinterpret_cast<const f32x4*>(s_rec);
  f32x4* dst = reinterpret_cast<f32x4*>(T.einfo + p0 * GGNN_EINFO_ROW);
#pragma unroll
  for (int c = 0; c < GGNN_EINFO_ROW / 4; ++c) {
    const int i = threadIdx.x + c * 256;
    if (i < n_rec * (GGNN_EINFO_ROW / 4)) dst[i] = src[i];
  }
}

__global__ __launch_bounds__(256) void edge_prepare_kernel(const PrepareArgs P) {
  edge_prepare_body<false>(P, (int)blockIdx.x, 0.f, false);
}

// z clamp of every node (test.py:405-407) in the first workgroups, then refresh + records per edge
// (x_joint_mirror / x_grain_mirror: optional second copies of the node features as they stand BEHIND this step -- every
// column, the clamped z included -- for a reader that runs beside the next step's in-place updates of x: rollout.py)
__global__ __launch_bounds__(256) void refresh_prepare_kernel(const PrepareArgs P, float* __restrict__ x_joint,
                                                              int64_t n_joint, int64_t ldxj, float* __restrict__ x_grain,
                                                              int64_t n_grain, int64_t ldxg, float zmax,
                                                              const int32_t* __restrict__ flags, int node_blocks,
                                                              float* __restrict__ x_joint_mirror,
                                                              float* __restrict__ x_grain_mirror) {
  const bool clamp = flags[1] != 0;
  if ((int)blockIdx.x < node_blocks) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n_joint + n_grain) {
      const bool jt = t < n_joint;
      float* row = jt ? x_joint + t * ldxj : x_grain + (t - n_joint) * ldxg;
      if (clamp) row[2] = zmax;
      float* mir = jt ? x_joint_mirror : x_grain_mirror;
      if (mir != nullptr) {
        const int64_t ld = jt ? ldxj : ldxg;
        mir += (jt ? t : t - n_joint) * ld;
        for (int64_t c = 0; c < ld; ++c) mir[c] = (clamp && c == 2) ? zmax : row[c];
      }
    }
    return;
  }
  edge_prepare_body<true>(P, (int)blockIdx.x - node_blocks, zmax, clamp);
}

// ---------------------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef const i32x8 __attribute__((address_space(4))) * const_i32x8_ptr;  // uniform index -> s_load_dwordx8
typedef const f32x4 __attribute__((address_space(4))) * const_f32x4_ptr;  // uniform index -> s_load_dwordx4
typedef const int __attribute__((address_space(4))) * const_i32_ptr;

constexpr int AG_WAVES = 4;  // independent row streams per workgroup

// Up to six sweeps in one launch (the three edge types of one cell, for one model or for the
// regressor and the classifier together): every workgroup walks its rows of sweep 0, then of
// sweep 1, ... (no barrier in between: a stream that runs out of rows moves on, so the tail of
// one sweep is filled by the head of the next, and a cell pays one launch instead of three).
constexpr int AG_MAX_SWEEPS = 6;
struct AggregateBatch {
  ggnn_aggregate_args a[AG_MAX_SWEEPS];
  int n;
};

template <int G, bool HAS_H>
__global__ __launch_bounds__(256)
void aggregate_kernel(const AggregateBatch B) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = (tid >> 4) & 3;     // the gate of this 16-lane row
  constexpr bool ALL = G == 4;      // every row of the wave has a gate
  const bool active = ALL || g < G;
  const int gc = active ? g : 0;
  const int l16 = tid & 15;
  // this lane's six channels: ch..ch+2 and 48+ch..48+ch+2, so that each of the two dwordx3
  // accesses of a fragment covers a CONTIGUOUS 192-byte span per row (6 consecutive channels per
  // lane would leave 12-byte holes in every instruction: half-filled lines on loads and stores)
  const int ch = 3 * l16;
  constexpr int CH2 = C / 2;

  // Row assignment (all wave-uniform).  Workgroups with equal blockIdx % 8 share an XCD and
  // its L2 (observed placement; speed only).  Each such group owns one contiguous eighth of
  // the rows and its streams (one per wave) take rows round-robin, so at any moment the whole
  // group works inside one short sliding window of neighbouring rows: the 3-6 readers of a
  // source row then run close together in time and the re-reads hit that L2 instead of going
  // back to memory.
  const int nblk = gridDim.x;
  const int ngrp = min(nblk, 8);
  const int grp = blockIdx.x % ngrp, lb = blockIdx.x / ngrp;
  const int nb_grp = (nblk - grp + ngrp - 1) / ngrp;            // workgroups in this group
  for (int k = 0; k < B.n; ++k) {
  const ggnn_aggregate_args& A = B.a[k];
  const int64_t x_lo = A.n_dst * grp / ngrp, x_hi = A.n_dst * (grp + 1) / ngrp;
  const int64_t n_streams = (int64_t)nb_grp * AG_WAVES;
  int64_t r = x_lo + (int64_t)lb * AG_WAVES + wave;
  if (r >= x_hi) continue;
  const const_i32_ptr uptr = (const_i32_ptr)(uintptr_t)A.unit_ptr;
  const const_i32x8_ptr udesc = (const_i32x8_ptr)(uintptr_t)A.units;
  const const_f32x4_ptr erec = (const_f32x4_ptr)(uintptr_t)A.einfo;  // 5 float4 per edge; [4] = (reloc, a)

  // this lane's 6 channels x W_value[:, 0..2] (edge_params [G][3][96], L2-resident, read once)
  f3 wv[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const float* e = A.edge_params + gc * GGNN_EDGE_PARAM_ROWS * C + ch + (c < 3 ? c : CH2 + c - 3);
    wv[c] = {e[0], e[C], e[2 * C]};
  }
  const float* vbase = A.p_src + A.v_off + gc * C + ch;
  const float* uhbase = A.p_dst + A.u_off + gc * C + ch;
  const float* u4base = A.p_dst + A.u4_off + gc * 16 + l16;
  const float* hbase = A.h_src + ch;
  const float* tbase = A.einfo + l16;
  const uint32_t ldp_src = (uint32_t)A.ldp_src, ldp_dst = (uint32_t)A.ldp_dst, ldh = (uint32_t)A.ldh_src;

  float mx = -INFINITY, den = 0.f, sae = 0.f;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // scalar state of the current unit
  int u = uptr[r], u_end = uptr[r + 1];
  i32x8 d = udesc[u];
  f32x4 ed[UE];
#pragma unroll
  for (int t = 0; t < UE; ++t) ed[t] = erec[((int64_t)d[1] + t) * (GGNN_EINFO_ROW / 4) + 4];  // padded by UE records

  while (true) {
    const int i = d[0], nact = d[2] & 0xFF;
    const bool first = (d[2] >> 8) & 1, last = (d[2] >> 9) & 1;
    // ---- vector side, unconditional and back to back (host checked: n * ld < 2^31) ----
    f3 uh[2], hh[UE][2], vv[UE][2];
    float u4 = 0.f, x4[UE];
    if (!ALL) {
      uh[0] = uh[1] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        hh[t][0] = hh[t][1] = vv[t][0] = vv[t][1] = {0.f, 0.f, 0.f};
        x4[t] = 0.f;
      }
    }
    if (ALL || active) {
      if (HAS_H) {
        uh[0] = ld3_nt(uhbase + (uint32_t)i * ldp_dst);  // read once
        uh[1] = ld3_nt(uhbase + (uint32_t)i * ldp_dst + CH2);
      }
      u4 = __builtin_nontemporal_load(u4base + (uint32_t)i * ldp_dst);
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        const uint32_t j = (uint32_t)d[4 + t];
        if (HAS_H) {
          hh[t][0] = ld3(hbase + j * ldh);
          hh[t][1] = ld3(hbase + j * ldh + CH2);
        }
        x4[t] = tbase[((uint32_t)d[1] + t) * GGNN_EINFO_ROW];
        vv[t][0] = ld3(vbase + j * ldp_src);
        vv[t][1] = ld3(vbase + j * ldp_src + CH2);
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
    for (int t = 0; t < UE; ++t) edn[t] = erec[((int64_t)dn[1] + t) * (GGNN_EINFO_ROW / 4) + 4];

    if (first) {
      mx = -INFINITY;
      den = 0.f;
      sae = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[c] = 0.f;
    }
    if ((ALL || active) && nact > 0) {  // (empty rows have a single unit with nact == 0: zeros are stored)
      float s[UE];
      float mnew = mx;
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        s[t] = -INFINITY;
        if (t < nact) {
          float part = u4 * x4[t];
          if (HAS_H)
            part += uh[0].x * hh[t][0].x + uh[0].y * hh[t][0].y + uh[0].z * hh[t][0].z +
                    uh[1].x * hh[t][1].x + uh[1].y * hh[t][1].y + uh[1].z * hh[t][1].z;
          s[t] = row_sum(part);  // 1/sqrt(96) is folded into u
          mnew = fmaxf(mnew, s[t]);
        }
      }
      const float scale = __expf(mx - mnew);  // exp(-inf) = 0 on a row's first unit
      den *= scale;
      sae *= scale;
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[c] *= scale;
#pragma unroll
      for (int t = 0; t < UE; ++t) {
        if (t < nact) {
          const float rx = ed[t].x, ry = ed[t].y, rz = ed[t].z;
          const float pe = __expf(s[t] - mnew);
          den += pe;
          sae += pe * ed[t].w;
          const float v[6] = {vv[t][0].x, vv[t][0].y, vv[t][0].z, vv[t][1].x, vv[t][1].y, vv[t][1].z};
#pragma unroll
          for (int c = 0; c < 6; ++c)
            acc[c] += pe * fmaxf(v[c] + wv[c].x * rx + wv[c].y * ry + wv[c].z * rz, 0.f);
        }
      }
      mx = mnew;
    }
    if ((ALL || active) && last) {
      const float inv = 1.0f / (den + 1e-16f);  // PyG softmax denominator
      float* orow = A.agg + (int64_t)i * A.ld_agg + g * A.a_gstride;
      st3_nt(orow + A.a_off + ch, {acc[0] * inv, acc[1] * inv, acc[2] * inv});
      st3_nt(orow + A.a_off + ch + CH2, {acc[3] * inv, acc[4] * inv, acc[5] * inv});
      if (l16 == 0) {  // 8 bytes of a line the other edge type's sweep also writes into: through L2
        orow[A.sc_off] = den * inv;
        orow[A.sc_off + 1] = sae * inv;
      }
      // (training path: the gate row's padding behind this sweep's scalars -- it meets zero weight columns of the gate GEMM and
      // must be finite -- is zeroed here instead of by a fill launch of the caller)
      for (int k = l16; k < A.pad_n; k += 16) orow[A.sc_off + 2 + k] = 0.f;
    }
    if (!more) break;
    u = un;
    u_end = un_end;
    r = rn;
    d = dn;
#pragma unroll
    for (int t = 0; t < UE; ++t) ed[t] = edn[t];
  }
  }  // sweeps of the batch
}

}  // namespace ggnn

extern "C" int ggnn_edge_prepare(const ggnn_prepare_edge* edges, int n_edge_types,
                                 ggnn_stream_t stream) {
  using namespace ggnn;
  if (!edges || n_edge_types < 1 || n_edge_types > 3) return GGNN_EINVAL;
  PrepareArgs P;
  P.n_et = n_edge_types;
  P.b_off[0] = 0;
  for (int k = 0; k < 3; ++k) {
    if (k < n_edge_types) {
      const ggnn_prepare_edge& T = edges[k];
      if (T.E < 0 || T.ldx_src < 3 || T.ldx_dst < 3 || !T.einfo || !aligned16(T.einfo)) return GGNN_EINVAL;
      if (T.f_src < 3 || T.f_src > 12 || T.ldx_src < T.f_src) return GGNN_EINVAL;
      if (T.E > 0 && (!T.col || !T.perm || !T.row || !T.edge_attr || !T.x_src || !T.x_dst))
        return GGNN_EINVAL;
      const int64_t nb = (T.E + GGNN_UNIT_EDGES + 255) / 256;  // + zero padding records
      if (P.b_off[k] + nb >= INT32_MAX) return GGNN_EINVAL;
      P.et[k] = T;
      P.b_off[k + 1] = P.b_off[k] + (int)nb;
    } else {
      P.et[k] = ggnn_prepare_edge{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
      P.b_off[k + 1] = P.b_off[k];
    }
  }
  hipLaunchKernelGGL(edge_prepare_kernel, dim3((unsigned)P.b_off[n_edge_types]), dim3(256), 0, (hipStream_t)stream, P);
  return launch_status();
}

extern "C" int ggnn_step_refresh_prepare(float* x_joint, int64_t n_joint, int64_t ldx_joint, float* x_grain,
                                         int64_t n_grain, int64_t ldx_grain, float zmax, const int32_t* flags,
                                         const ggnn_prepare_edge* edges, int n_edge_types, float* x_joint_mirror,
                                         float* x_grain_mirror, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!x_joint || !x_grain || !flags || n_joint <= 0 || n_grain <= 0 || ldx_joint < 3 || ldx_grain < 3) return GGNN_EINVAL;
  if ((x_joint_mirror == nullptr) != (x_grain_mirror == nullptr) || x_joint_mirror == x_joint || x_grain_mirror == x_grain)
    return GGNN_EINVAL;
  if (!edges || n_edge_types < 1 || n_edge_types > 3) return GGNN_EINVAL;
  PrepareArgs P;
  P.n_et = n_edge_types;
  P.b_off[0] = 0;
  for (int k = 0; k < 3; ++k) {
    if (k < n_edge_types) {
      const ggnn_prepare_edge& T = edges[k];
      if (T.E < 0 || T.ldx_src < 3 || T.ldx_dst < 3 || !T.einfo || !aligned16(T.einfo)) return GGNN_EINVAL;
      if (T.f_src < 3 || T.f_src > 12 || T.ldx_src < T.f_src) return GGNN_EINVAL;
      if (T.E > 0 && (!T.col || !T.perm || !T.row || !T.edge_attr || !T.x_src || !T.x_dst)) return GGNN_EINVAL;
      const int64_t nb = (T.E + GGNN_UNIT_EDGES + 255) / 256;
      if (P.b_off[k] + nb >= INT32_MAX) return GGNN_EINVAL;
      P.et[k] = T;
      P.b_off[k + 1] = P.b_off[k] + (int)nb;
    } else {
      P.et[k] = ggnn_prepare_edge{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
      P.b_off[k + 1] = P.b_off[k];
    }
  }
  const int64_t node_blocks = (n_joint + n_grain + 255) / 256;
  if (node_blocks + P.b_off[n_edge_types] >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(refresh_prepare_kernel, dim3((unsigned)(node_blocks + P.b_off[n_edge_types])), dim3(256), 0,
                     (hipStream_t)stream, P, x_joint, n_joint, ldx_joint, x_grain, n_grain, ldx_grain, zmax, flags,
                     (int)node_blocks, x_joint_mirror, x_grain_mirror);
  return launch_status();
}

namespace ggnn {
// Argument checks of one sweep; normalises the encoder form (h_src == NULL).
static int check_sweep(ggnn_aggregate_args& A) {
  if (!A.unit_ptr || !A.units || !A.einfo || !A.p_src || !A.p_dst || !A.edge_params || !A.agg)
    return GGNN_EINVAL;
  if (!aligned16(A.units) || !aligned16(A.einfo)) return GGNN_EINVAL;
  if (A.n_dst <= 0 || A.n_src <= 0 || A.E < 0) return GGNN_EINVAL;
  const int G = A.n_gates;
  if (G != 1 && G != 3 && G != 4) return GGNN_EINVAL;
  const bool has_h = A.h_src != nullptr;
  if (has_h && (A.ldh_src < C || A.u_off < 0)) return GGNN_EINVAL;
  if (!has_h) {  // never dereferenced, but the address arithmetic must stay in range
    A.h_src = A.p_src;
    A.ldh_src = 0;
    A.u_off = 0;
  }
  if (A.v_off < 0 || A.u4_off < 0 || A.a_off < 0 || A.sc_off < 0 || A.a_gstride < C) return GGNN_EINVAL;
  if (A.ldp_src <= 0 || A.ldp_dst <= 0 || A.n_src * A.ldp_src >= INT32_MAX || A.n_dst * A.ldp_dst >= INT32_MAX ||
      A.n_src * A.ldh_src >= INT32_MAX || (A.E + GGNN_UNIT_EDGES) * GGNN_EINFO_ROW >= INT32_MAX)
    return GGNN_EINVAL;  // the sweep forms row offsets in 32 bits
  if (A.v_off + (int64_t)G * C > A.ldp_src || A.u4_off + (int64_t)G * 16 > A.ldp_dst) return GGNN_EINVAL;
  if (has_h && A.u_off + (int64_t)G * C > A.ldp_dst) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.a_off + C > A.ld_agg) return GGNN_EINVAL;
  if (A.pad_n < 0 || (int64_t)(G - 1) * A.a_gstride + A.sc_off + 2 + A.pad_n > A.ld_agg) return GGNN_EINVAL;
  return GGNN_OK;
}
}  // namespace ggnn

extern "C" int ggnn_period_gat_aggregate_batch(const ggnn_aggregate_args* args, int n_sweeps,
                                               ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_sweeps < 1 || n_sweeps > AG_MAX_SWEEPS) return GGNN_EINVAL;
  AggregateBatch B;
  B.n = n_sweeps;
  const int G = args[0].n_gates;
  const bool has_h = args[0].h_src != nullptr;
  int64_t want = 1;
  for (int k = 0; k < AG_MAX_SWEEPS; ++k) {
    B.a[k] = args[k < n_sweeps ? k : 0];
    if (B.a[k].n_gates != G || (B.a[k].h_src != nullptr) != has_h) return GGNN_EINVAL;
    const int rc = check_sweep(B.a[k]);
    if (rc != GGNN_OK) return rc;
    want = std::max<int64_t>(want, (B.a[k].n_dst + 3) / 4);
  }
  // persistent grid: at least 4 rows per workgroup, at most the resident capacity
  const int64_t cap = (int64_t)num_cu() * (has_h ? AG_BLOCKS_PER_CU : AG_BLOCKS_PER_CU_NOH);
  const dim3 grid((unsigned)(want < cap ? want : cap));
  hipStream_t s = (hipStream_t)stream;
#define GGNN_AG_LAUNCH(G_)                                                                   \
  do {                                                                                       \
    if (has_h) hipLaunchKernelGGL((aggregate_kernel<G_, true>), grid, dim3(256), 0, s, B);   \
    else hipLaunchKernelGGL((aggregate_kernel<G_, false>), grid, dim3(256), 0, s, B);        \
  } while (0)
  if (G == 4) GGNN_AG_LAUNCH(4);
  else if (G == 3) GGNN_AG_LAUNCH(3);
  else GGNN_AG_LAUNCH(1);
#undef GGNN_AG_LAUNCH
  return launch_status();
}

extern "C" int ggnn_period_gat_aggregate(const ggnn_aggregate_args* args, ggnn_stream_t stream) {
  return ggnn_period_gat_aggregate_batch(args, 1, stream);
}
