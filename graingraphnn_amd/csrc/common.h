// Shared device/host helpers for libggnn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ggnn.h"

namespace ggnn {

constexpr int C = GGNN_C;  // 96 hidden channels

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int launch_status() {
  return hipGetLastError() == hipSuccess ? GGNN_OK : GGNN_ELAUNCH;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Blocks b and b+8 share an XCD (round-robin dispatch, observed not contractual): give every
// XCD one contiguous range of logical blocks so neighbouring rows meet in the same L2.
// Bijective for any grid size; affects speed only.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Gate non-linearities on the hardware exp / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).
// Absolute error <= ~2e-7 on outputs in [-1, 1]; the parity bar is 1e-4 relative.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2 / (1 + exp(2x)); exp overflow -> rcp(inf) = 0 -> 1, underflow -> -1
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}

struct f3 {
  float x, y, z;
};
// 12-byte row-fragment load/store; p must be 4-byte aligned.  The three adjacent dword
// accesses are merged into one global_load_dwordx3 / global_store_dwordx3 by the backend.
__device__ __forceinline__ f3 ld3(const float* __restrict__ p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* __restrict__ p, f3 v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}

// Streaming (read-once / write-once) variants: the `nt` hint keeps these rows from evicting
// the gathered rows that ARE re-read out of the XCD's L2.
__device__ __forceinline__ f3 ld3_nt(const float* __restrict__ p) {
  return {__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2)};
}
__device__ __forceinline__ void st3_nt(float* __restrict__ p, f3 v) {
  __builtin_nontemporal_store(v.x, p);
  __builtin_nontemporal_store(v.y, p + 1);
  __builtin_nontemporal_store(v.z, p + 2);
}

}  // namespace ggnn
