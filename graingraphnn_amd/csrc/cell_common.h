// Pieces shared by the fused cell kernels (dec_cell.hip, enc_cell.hip): the LDS-DMA weight stream, the two-piece
// fp16 operand split with its range tracking, the k-step of a streamed GEMM phase.  gfx950 only.
#pragma once
#include "common.h"

namespace ggnn {

constexpr int DC_PL = 2;   // weight / operand planes: fp16 hi and scaled residual (common.h)

// LDS-DMA: every lane copies 16 bytes from its own global address to lds_base + lane * 16 (wave-uniform base
// in M0).  Not tracked by the compiler: completion = s_waitcnt vmcnt (in issue order with every other
// vector-memory operation of the wave).
__device__ __forceinline__ void dc_dma16(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base))
               : "memory");
}

// Range tracking of what goes through a split (range flag, ggnn.h): the largest |x| AS A BIT PATTERN -- for non-negative
// floats the unsigned order of the bits is the order of the values, with +inf above every finite value and every NaN
// above +inf (an fp max would drop a NaN) -- so one v_max3_u32 per pair keeps a non-finite operand visible.
constexpr uint32_t DC_RANGE_LIMIT = 0x477FE000u;   // bits of 65504.0f: in range <=> tracked < limit
__device__ __forceinline__ void dc_track(uint32_t& amax, float a, float b) {
  amax = max(amax, max(__float_as_uint(a) & 0x7fffffffu, __float_as_uint(b) & 0x7fffffffu));
}
// eight fp32 values (r0 | r1) -> the two fp16 planes of an MFMA operand fragment; `amax` follows the largest
// magnitude that went through a split (dc_track)
__device__ __forceinline__ void dc_split(const f32x4 r0, const f32x4 r1, u32x4 (&xb)[DC_PL], uint32_t& amax) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 h = e < 2 ? r0 : r1;
    const float a = h[2 * (e & 1)], b = h[2 * (e & 1) + 1];
    dc_track(amax, a, b);
    uint32_t q0, q1;
    split_f16x2(a, b, q0, q1);
    xb[0][e] = q0;
    xb[1][e] = q1;
  }
}
// ... without the tracking (a kernel that checks its operands where they are produced)
__device__ __forceinline__ void dc_split(const f32x4 r0, const f32x4 r1, u32x4 (&xb)[DC_PL]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 h = e < 2 ? r0 : r1;
    uint32_t q0, q1;
    split_f16x2(h[2 * (e & 1)], h[2 * (e & 1) + 1], q0, q1);
    xb[0][e] = q0;
    xb[1][e] = q1;
  }
}
// four fp32 values in k slots 0..3 of a fragment, zeros in slots 4..7
__device__ __forceinline__ void dc_split_half(const f32x4 r0, u32x4 (&xb)[DC_PL], uint32_t& amax) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float a = r0[2 * e], b = r0[2 * e + 1];
    dc_track(amax, a, b);
    uint32_t q0, q1;
    split_f16x2(a, b, q0, q1);
    xb[0][e] = q0;
    xb[1][e] = q1;
  }
  xb[0][2] = xb[0][3] = xb[1][2] = xb[1][3] = 0u;
}

// An accumulator of a 16 x 16 output tile: main + cross terms of the fp16 split.
struct DcAcc {
  f32x4 m, c;
  __device__ __forceinline__ void zero() { m = c = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ f32x4 value() const { return m + c * (1.0f / F16X2_SCALE); }
};
// One k-step of a GEMM phase: acc[nb] += W[nb] . x for the NB column tiles of the slice at `pw` (= slice base +
// lane; piece (nb, plane) at (nb * 2 + plane) * 64).  The weight fragments are read AHEAD column tiles in front of
// the three MFMAs that use them (1: measured the same as 2 and 3 on both cell kernels -- the k-step is not bound by the
// latency of its fragment reads -- and 2 costs the encoder cell 8 spilled registers).
#ifndef GGNN_KSTEP_AHEAD
#define GGNN_KSTEP_AHEAD 1
#endif
struct DcNoMid {
  __device__ __forceinline__ void operator()() const {}
};
// `mid`: called once behind the MFMAs of column tile GGNN_KSTEP_MID (development: where in a k-step the wave issues its share
// of the next slice's LDS-DMA -- at the top, in front of the first fragment read, or between two column tiles' MFMAs)
#ifndef GGNN_KSTEP_MID
#define GGNN_KSTEP_MID 0
#endif
template <int NB, typename Mid = DcNoMid>
__device__ __forceinline__ void dc_kstep(const u32x4* __restrict__ pw, const u32x4 (&xb)[DC_PL], DcAcc (&acc)[NB],
                                         Mid mid = Mid()) {
  constexpr int AH = GGNN_KSTEP_AHEAD < NB ? GGNN_KSTEP_AHEAD : NB - 1;
  u32x4 wf[AH + 1][DC_PL];
#pragma unroll
  for (int a = 0; a < AH; ++a)
#pragma unroll
    for (int p = 0; p < DC_PL; ++p) wf[a][p] = pw[(a * DC_PL + p) * 64];
  __builtin_amdgcn_sched_group_barrier(0x100, AH * DC_PL, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (nb + AH < NB) {
#pragma unroll
      for (int p = 0; p < DC_PL; ++p) wf[(nb + AH) % (AH + 1)][p] = pw[((nb + AH) * DC_PL + p) * 64];
    }
    mfma_x3h(wf[nb % (AH + 1)], xb, acc[nb].m, acc[nb].c);
    if (nb + AH < NB) __builtin_amdgcn_sched_group_barrier(0x100, DC_PL, 0);  // DS read
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                         // MFMA
    if (nb == GGNN_KSTEP_MID) mid();
  }
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ f32x4 ld16f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

}  // namespace ggnn
