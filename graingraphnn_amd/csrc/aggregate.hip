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
// The sweep is a per-wave LDS-DMA gather ring.  Work is cut into *units* (one destination
// row x up to 3 in-edges, descriptor table built with the CSR).  A wave owns a contiguous
// range of destination rows and runs a 3-deep software pipeline over its units:
//   issue  : 13 `global_load_lds_dwordx4` per unit copy the unit descriptor, its 3 edge
//            records, the destination's Q fragments of all gates (G x 384 B) and the 3
//            neighbour rows' K|V fragments of all gates (G x 768 B contiguous each) straight
//            from HBM into the wave's LDS slot -- no VGPRs are held while they fly, so each
//            wave keeps 2 units (~21 KB) in flight and a CU ~86 KB, 2x what 8 TB/s needs;
//   wait   : a counted `s_waitcnt vmcnt(13 * units_issued_after)` (LDS-DMA completes in
//            order), never vmcnt(0) in steady state, no workgroup barrier;
//   compute: the two half-waves take gates {h, h+2}; each lane owns channels {l, l+32, l+64}
//            (conflict-free ds_read_b32), scores its edges, folds them into an online-max
//            softmax carried in registers across the units of one row, and stores the row
//            when its last unit is done.
// Unit descriptors are fetched with scalar loads (SMEM, lgkmcnt) one iteration ahead, so the
// only vector-memory traffic in the loop is the DMA stream and the output stores.
// No atomics, one owner per output row => bit-reproducible.  Rows of any degree work.
#include "common.h"

namespace ggnn {

constexpr int AG_WAVES = 4;   // waves per workgroup (one workgroup per CU: LDS-limited)
constexpr int AG_DEPTH = 3;   // ring slots per wave
constexpr int AG_NUM_CU = 256;
constexpr int UE = GGNN_UNIT_EDGES;

// LDS slot layout (bytes)
constexpr int SL_DESC = 0;     // 32 B unit descriptor
constexpr int SL_EINFO = 64;   // 3 x 16 B edge records
constexpr int SL_Q = 128;      // G x 384 B
template <int G> struct Slot {
  static constexpr int q_bytes = G * C * 4;
  static constexpr int row_bytes = G * 2 * C * 4;
  static constexpr int rows_off = SL_Q + ((q_bytes + 63) / 64) * 64;
  static constexpr int bytes = ((rows_off + UE * row_bytes + 127) / 128) * 128;
  static constexpr int nq = (q_bytes + 1023) / 1024;     // DMA instructions for Q
  static constexpr int nr = (row_bytes + 1023) / 1024;   // DMA instructions per neighbour row
  static constexpr int dma_per_unit = 2 + nq + UE * nr;  // desc + einfo + Q + rows
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef const i32x8 __attribute__((address_space(4))) * const_i32x8_ptr;  // -> s_load_dwordx8
typedef const int __attribute__((address_space(4))) * const_i32_ptr;

// One LDS-DMA instruction: every active lane copies 16 bytes from its own global address to
// lds_base + lane * 16 (wave-uniform base in M0).  Not tracked by the compiler: completion is
// awaited with dma_wait<N>() below.  (cdna_hip_programming.md 5.7: M0 is written and
// restored inside the same statement.)
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}

template <int N> __device__ __forceinline__ void dma_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

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
template <int G>
__device__ __forceinline__ void issue_unit(const ggnn_aggregate_args& A, int u, const i32x8 d,
                                           uint32_t slot, int lane) {
  using S = Slot<G>;
  const int i = d[0], p0 = d[1];
  // descriptor (2 lanes) and the three edge records (3 lanes)
  if (lane < 2) dma16(reinterpret_cast<const char*>(A.units) + (int64_t)u * 32 + lane * 16, slot + SL_DESC);
  if (lane < UE) dma16(reinterpret_cast<const char*>(A.einfo) + ((int64_t)p0 + lane) * 16, slot + SL_EINFO);
  // Q fragments of all gates of destination i
  const char* qsrc = reinterpret_cast<const char*>(A.p_dst + (int64_t)i * A.ldp_dst + A.q_off) + lane * 16;
#pragma unroll
  for (int m = 0; m < S::nq; ++m)
    if (m * 1024 + lane * 16 < S::q_bytes) dma16(qsrc + m * 1024, slot + SL_Q + m * 1024);
  // K|V fragments of all gates of the three neighbour rows
#pragma unroll
  for (int t = 0; t < UE; ++t) {
    const char* rsrc = reinterpret_cast<const char*>(A.p_src + (int64_t)d[4 + t] * A.ldp_src + A.kv_off) + lane * 16;
#pragma unroll
    for (int m = 0; m < S::nr; ++m)
      if (m * 1024 + lane * 16 < S::row_bytes)
        dma16(rsrc + m * 1024, slot + S::rows_off + t * S::row_bytes + m * 1024);
  }
}

struct GateState {
  float mx, den, sae, a0, a1, a2;
};

template <int G>
__global__ __launch_bounds__(AG_WAVES * 64, 1) void aggregate_kernel(const ggnn_aggregate_args A) {
  using S = Slot<G>;
  __shared__ float s_ep[G * GGNN_EDGE_PARAM_ROWS * C];
  __shared__ __attribute__((aligned(128))) char s_ring[AG_WAVES * AG_DEPTH * S::bytes];

  const int tid = threadIdx.x;
  for (int t = tid; t < G * GGNN_EDGE_PARAM_ROWS * C; t += AG_WAVES * 64) s_ep[t] = A.edge_params[t];
  __syncthreads();  // the only workgroup barrier; no DMA has been issued yet

  const int lane = tid & 63, l32 = tid & 31, half = (tid >> 5) & 1;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = gridDim.x;
  const int b = xcd_remap(blockIdx.x, nblk);
  // contiguous destination range of this wave
  const int64_t n_waves = (int64_t)nblk * AG_WAVES;
  const int64_t dpw = (A.n_dst + n_waves - 1) / n_waves;
  const int64_t d_lo = ((int64_t)b * AG_WAVES + wave) * dpw;
  if (d_lo >= A.n_dst) return;
  const int64_t d_hi = min(A.n_dst, d_lo + dpw);
  const const_i32_ptr uptr = (const_i32_ptr)(uintptr_t)A.unit_ptr;
  const int u_lo = uptr[d_lo], u_hi = uptr[d_hi];
  const int n_u = u_hi - u_lo;
  const const_i32x8_ptr udesc = (const_i32x8_ptr)(uintptr_t)A.units;

  const uint32_t ring =
      __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s_ring))) +
      wave * (AG_DEPTH * S::bytes);
  const char* ring_ptr = s_ring + wave * (AG_DEPTH * S::bytes);
  const float inv_sqrt_c = 0.10206207261596577f;  // 1/sqrt(96), periodGATconv.py:226

  // gates owned by this half-wave: half, half + 2
  GateState st[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) st[s] = {-INFINITY, 0.f, 0.f, 0.f, 0.f, 0.f};

  // ---- prologue: AG_DEPTH-1 units in flight, descriptor of the next one on its way ----
  i32x8 dnext = udesc[u_lo];
#pragma unroll
  for (int k = 0; k < AG_DEPTH - 1; ++k) {
    if (k < n_u) {
      const i32x8 d = dnext;
      if (k + 1 < n_u) dnext = udesc[u_lo + k + 1];
      issue_unit<G>(A, u_lo + k, d, ring + k * S::bytes, lane);
    }
  }

  int slot_c = 0;                 // slot of the unit being consumed
  int slot_i = AG_DEPTH - 1;      // slot the next issue goes to
  for (int k = 0; k < n_u; ++k) {
    // ---- issue unit k + DEPTH - 1 into the slot freed in the previous iteration ----
    const int ki = k + AG_DEPTH - 1;
    if (ki < n_u) {
      const i32x8 d = dnext;
      if (ki + 1 < n_u) dnext = udesc[u_lo + ki + 1];
      issue_unit<G>(A, u_lo + ki, d, ring + slot_i * S::bytes, lane);
      slot_i = slot_i + 1 == AG_DEPTH ? 0 : slot_i + 1;
    }
    // ---- wait for unit k: everything but the DMAs of the units issued after it ----
    const int younger = min(n_u - 1 - k, AG_DEPTH - 1);
    if (younger >= 2) dma_wait<2 * S::dma_per_unit>();
    else if (younger == 1) dma_wait<S::dma_per_unit>();
    else dma_wait<0>();

    // ---- compute unit k from LDS ----
    const char* sl = ring_ptr + slot_c * S::bytes;
    slot_c = slot_c + 1 == AG_DEPTH ? 0 : slot_c + 1;
    const i32x4 dd = *reinterpret_cast<const i32x4*>(sl + SL_DESC);
    const int i = dd.x, nact = dd.z & 0xFF;
    const bool first = (dd.z >> 8) & 1, last = (dd.z >> 9) & 1;
    f32x4 ed[UE];
#pragma unroll
    for (int t = 0; t < UE; ++t) ed[t] = *reinterpret_cast<const f32x4*>(sl + SL_EINFO + 16 * t);
    const float* qf = reinterpret_cast<const float*>(sl + SL_Q);
    const float* rows = reinterpret_cast<const float*>(sl + S::rows_off);

#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int g = half + 2 * s;
      if (g < G) {
        GateState z = st[s];
        if (first) z = {-INFINITY, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* ep = &s_ep[g * GGNN_EDGE_PARAM_ROWS * C + l32];
        const float q0 = qf[g * C + l32], q1 = qf[g * C + l32 + 32], q2 = qf[g * C + l32 + 64];
        float sc[UE];
        float mnew = z.mx;
#pragma unroll
        for (int t = 0; t < UE; ++t) {
          sc[t] = -INFINITY;
          if (t < nact) {
            const float* kr = rows + t * (S::row_bytes / 4) + g * 2 * C + l32;
            const float rx = ed[t].x, ry = ed[t].y, rz = ed[t].z, a = ed[t].w;
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float kc = kr[32 * c] + ep[32 * c] * rx + ep[C + 32 * c] * ry + ep[2 * C + 32 * c] * rz +
                               ep[6 * C + 32 * c] * a;
              part += (c == 0 ? q0 : (c == 1 ? q1 : q2)) * kc;
            }
            sc[t] = halfwave_sum(part) * inv_sqrt_c;
            mnew = fmaxf(mnew, sc[t]);
          }
        }
        // (empty rows have a single unit with nact == 0: nothing to fold, zeros are stored)
        const float scale = nact > 0 ? __expf(z.mx - mnew) : 1.0f;  // exp(-inf) = 0 on a row's first unit
        z.den *= scale;
        z.sae *= scale;
        z.a0 *= scale;
        z.a1 *= scale;
        z.a2 *= scale;
#pragma unroll
        for (int t = 0; t < UE; ++t) {
          if (t < nact) {
            const float* vr = rows + t * (S::row_bytes / 4) + g * 2 * C + C + l32;
            const float rx = ed[t].x, ry = ed[t].y, rz = ed[t].z;
            const float pe = __expf(sc[t] - mnew);
            z.den += pe;
            z.sae += pe * ed[t].w;
            z.a0 += pe * fmaxf(vr[0] + ep[3 * C] * rx + ep[4 * C] * ry + ep[5 * C] * rz, 0.f);
            z.a1 += pe * fmaxf(vr[32] + ep[3 * C + 32] * rx + ep[4 * C + 32] * ry + ep[5 * C + 32] * rz, 0.f);
            z.a2 += pe * fmaxf(vr[64] + ep[3 * C + 64] * rx + ep[4 * C + 64] * ry + ep[5 * C + 64] * rz, 0.f);
          }
        }
        if (nact > 0) z.mx = mnew;
        st[s] = z;
        if (last) {
          const float inv = 1.0f / (z.den + 1e-16f);  // PyG softmax denominator
          float* orow = A.agg + (int64_t)i * A.ld_agg + g * A.a_gstride;
          orow[A.a_off + l32] = z.a0 * inv;
          orow[A.a_off + l32 + 32] = z.a1 * inv;
          orow[A.a_off + l32 + 64] = z.a2 * inv;
          if (l32 == 0) {
            orow[A.sc_off] = z.den * inv;
            orow[A.sc_off + 1] = z.sae * inv;
          }
        }
      }
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
  if (!aligned16(A.units) || !aligned16(A.einfo) || !aligned16(A.p_src) || !aligned16(A.p_dst))
    return GGNN_EINVAL;
  if (A.n_dst <= 0 || A.n_src <= 0 || A.E < 0) return GGNN_EINVAL;
  const int G = A.n_gates;
  if (G != 1 && G != 3 && G != 4) return GGNN_EINVAL;
  if (A.kv_off < 0 || A.q_off < 0 || A.a_off < 0 || A.sc_off < 0 || A.a_gstride < C) return GGNN_EINVAL;
  if ((A.kv_off & 3) || (A.q_off & 3) || (A.ldp_src & 3) || (A.ldp_dst & 3)) return GGNN_EINVAL;  // 16-byte DMA granules
  if (A.kv_off + (int64_t)G * 2 * C > A.ldp_src || A.q_off + (int64_t)G * C > A.ldp_dst) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.a_off + C > A.ld_agg) return GGNN_EINVAL;
  if ((int64_t)(G - 1) * A.a_gstride + A.sc_off + 2 > A.ld_agg) return GGNN_EINVAL;
  const int64_t want = (A.n_dst + AG_WAVES - 1) / AG_WAVES;  // at least one row per wave
  const int64_t nblk = want < AG_NUM_CU ? want : AG_NUM_CU;
  const dim3 grid((unsigned)nblk), block(AG_WAVES * 64);
  hipStream_t s = (hipStream_t)stream;
  if (G == 4)
    hipLaunchKernelGGL(aggregate_kernel<4>, grid, block, 0, s, A);
  else if (G == 3)
    hipLaunchKernelGGL(aggregate_kernel<3>, grid, block, 0, s, A);
  else
    hipLaunchKernelGGL(aggregate_kernel<1>, grid, block, 0, s, A);
  return launch_status();
}
