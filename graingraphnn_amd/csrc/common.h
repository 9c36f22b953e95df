// Shared device/host helpers for libggnn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ggnn.h"

namespace ggnn {

constexpr int C = GGNN_C;  // 96 hidden channels

typedef float f32x4 __attribute__((ext_vector_type(4)));

// How the K >= 100 GEMMs (decoder projection, gate epilogue) run: GGNN_GEMM=fp32 in the
// environment selects the native v_mfma_f32_16x16x4_f32 kernels, anything else (default) the
// 3 x bf16 split kernels.  Read once per process.
int gemm_mode();

// Compute units of the current device (256 on MI355X), queried once per process: sizes the
// persistent / one-round grids.  Affects speed only.
int num_cu();

// out[b][j] = sum_r in[b][r][j] in a fixed order, many short rows (train_misc.hip: ggnn_sum_rows; also the reduction of
// ggnn_wgrad's partial results when there are many of them).  n_cols % 4 == 0, 16-byte aligned operands.
int launch_sum_rows(const float* in, float* out, int64_t n_rows, int64_t n_cols, int batch, hipStream_t stream);

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

// ---- fp32 GEMMs on the bf16 matrix cores: exact 3-way split + 6 products ----------------
// x = hi + mid + lo EXACTLY (three round-to-nearest bf16 pieces hold the 24-bit significand),
// and  x*w = hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi  up to the three dropped
// products mid*lo, lo*mid, lo*lo <= 2^-25 |x*w|: below one fp32 rounding of the product.  Every
// kept product of two 8-bit significands is exact in the fp32 accumulator, so the result is
// fp32-equivalent (tests compare both GEMM modes with an fp64 product) at 6/16 of the MFMA
// cycles of v_mfma_f32_16x16x4_f32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE)
  const f32x2_ v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}
// two fp32 values -> three dwords of packed (a, b) bf16 pieces
__device__ __forceinline__ void split_bf16x3(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = pack_bf16(a, b);
  a -= __uint_as_float(hi << 16);
  b -= __uint_as_float(hi & 0xffff0000u);
  mid = pack_bf16(a, b);
  a -= __uint_as_float(mid << 16);
  b -= __uint_as_float(mid & 0xffff0000u);
  lo = pack_bf16(a, b);
}
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// all six kept products of one k-step, smallest first
__device__ __forceinline__ f32x4 mfma_x6(const u32x4 (&w)[3], const u32x4 (&x)[3], f32x4 c) {
  c = mfma_bf16(w[0], x[2], c);
  c = mfma_bf16(w[2], x[0], c);
  c = mfma_bf16(w[1], x[1], c);
  c = mfma_bf16(w[0], x[1], c);
  c = mfma_bf16(w[1], x[0], c);
  c = mfma_bf16(w[0], x[0], c);
  return c;
}

// ---- the same GEMMs with TWO fp16 pieces and THREE products (development: dec_cell.hip -DDC_F16X2) ----
// x = hi + lo' / 2^11 + e with hi = rne16(x), lo' = rne16((x - hi) 2^11): 2 x 11 significand bits, |e| <= 2^-22 |x|;
// the scaling keeps the residual in fp16's normal range down to |x| ~ 2^-13.  x*w ~ hi*hi + (hi*lo' + lo'*hi) / 2^11:
// the cross terms go to a second accumulator that is folded in once, after the k-loop.  Against an fp64 product,
// normalised by sum |x||w|: 4.6e-8 .. 6e-8 (K = 104 / 196), the six-product bf16 split 2e-8, a plain fp32 fma chain
// 2e-7 -- half the MFMAs and two thirds of the weight bytes of the bf16 split.  |x| > 65504 saturates (clamped).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
constexpr float F16X2_SCALE = 2048.0f;
__device__ __forceinline__ void split_f16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
  a = __builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f);
  b = __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f);
  const f16x2_ h = {(_Float16)a, (_Float16)b};
  const f16x2_ l = {(_Float16)((a - (float)h[0]) * F16X2_SCALE), (_Float16)((b - (float)h[1]) * F16X2_SCALE)};
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the three kept products of one k-step: main += hi hi, cross += hi lo' + lo' hi
__device__ __forceinline__ void mfma_x3h(const u32x4 (&w)[2], const u32x4 (&x)[2], f32x4& main, f32x4& cross) {
  cross = mfma_f16(w[0], x[1], cross);
  cross = mfma_f16(w[1], x[0], cross);
  main = mfma_f16(w[0], x[0], main);
}

// Sum over the 16 lanes of a DPP row, result in every lane of the row (four v_add_f32 with DPP).
__device__ __forceinline__ float row_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
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
