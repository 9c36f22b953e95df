// In-kernel time stamps for development (tools/stamps.py).  The product build defines nothing: every
// GGNN_STAMP(i) is empty.  `make STAMPS=1` builds a diagnostic library whose instrumented kernels
// write s_memrealtime (100 MHz) of phase i for every wave into a device array that
// ggnn_debug_stamps() copies out; no output of a kernel depends on a stamp.
#pragma once
#ifdef GGNN_STAMPS
// One buffer per translation unit: a file that stamps defines GGNN_STAMP_SUFFIX (e.g. _enc) before the
// include unless it is the gate kernel (no suffix: tools/stamps.py).
#ifndef GGNN_STAMP_SUFFIX
#define GGNN_STAMP_SUFFIX
#endif
#define GGNN_STAMP_CAT2(a, b) a##b
#define GGNN_STAMP_CAT(a, b) GGNN_STAMP_CAT2(a, b)
#define ggnn_stamp_buf GGNN_STAMP_CAT(ggnn_stamp_buf_, GGNN_STAMP_SUFFIX)
#define GGNN_STAMP_SLOTS 20
#define GGNN_STAMP_WAVES 8192
static __device__ unsigned long long ggnn_stamp_buf[GGNN_STAMP_WAVES * GGNN_STAMP_SLOTS];
#define GGNN_STAMP(i)                                                                              \
  do {                                                                                             \
    if ((threadIdx.x & 63) == 0) {                                                                 \
      const unsigned w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                     \
      if (w_ < GGNN_STAMP_WAVES) {                                                                 \
        ggnn_stamp_buf[w_ * GGNN_STAMP_SLOTS + (i)] = __builtin_amdgcn_s_memrealtime();            \
        /* slots 17 / 18: the shader clock counter beside stamps 0 / 16 (the clock a kernel held) */ \
        if ((i) == 0) ggnn_stamp_buf[w_ * GGNN_STAMP_SLOTS + 17] = __builtin_readcyclecounter();     \
        if ((i) == 16) ggnn_stamp_buf[w_ * GGNN_STAMP_SLOTS + 18] = __builtin_readcyclecounter();    \
      }                                                                                            \
    }                                                                                              \
  } while (0)
#define GGNN_STAMP_VAL(i, v)                                                                       \
  do {                                                                                             \
    if ((threadIdx.x & 63) == 0) {                                                                 \
      const unsigned w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                     \
      if (w_ < GGNN_STAMP_WAVES) ggnn_stamp_buf[w_ * GGNN_STAMP_SLOTS + (i)] = (unsigned long long)(v); \
    }                                                                                              \
  } while (0)
#define GGNN_STAMP_NOW() __builtin_amdgcn_s_memrealtime()
extern "C" int GGNN_STAMP_CAT(ggnn_debug_stamps, GGNN_STAMP_SUFFIX)(unsigned long long* host_dst) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ggnn_stamp_buf), sizeof(ggnn_stamp_buf)) == hipSuccess ? 0 : -2;
}
extern "C" int GGNN_STAMP_CAT(ggnn_debug_stamps_clear, GGNN_STAMP_SUFFIX)(void) {
  void* p = nullptr;
  return hipGetSymbolAddress(&p, HIP_SYMBOL(ggnn_stamp_buf)) == hipSuccess &&
                 hipMemset(p, 0, sizeof(ggnn_stamp_buf)) == hipSuccess ? 0 : -2;
}
#else
#define GGNN_STAMP(i) do { } while (0)
#define GGNN_STAMP_VAL(i, v) do { } while (0)
#define GGNN_STAMP_NOW() 0ull
#endif
