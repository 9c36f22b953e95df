// Development probe: issue rate of bf16 MFMA shapes on gfx950 (cycles per instruction per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(uint64_t* out, int iters, uint32_t seed) {
  const int l = threadIdx.x;
  u32x4 a = {seed * (l + 1), seed ^ l, 0x3f803f80u, 0x3f813f82u}, b = {0x3f803f80u + l, 0x3f853f80u, seed, l * 77u};
  f32x4 acc[4] = {};
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (SHAPE == 32)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
      else if (SHAPE == 16)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, (u32x2){a[0], a[1]}), __builtin_bit_cast(s16x4, (u32x2){b[0], b[1]}), acc[j], 0, 0, 0);
      else
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a[0]), __builtin_bit_cast(float, b[0]), acc[j], 0, 0, 0);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (l == 0) out[blockIdx.x] = t1 - t0;
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1.2345f) out[0] = 0;
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 8 * 4096);
  const int iters = 20000;
  for (int shape : {32, 16, 4}) {
    for (int blocks : {1, 1024}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        else if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      uint64_t c;
      hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
      printf("shape 16x16x%-2d blocks %4d: %.1f cycles / MFMA (one wave per SIMD), %.3f ms => %.0f MHz\n", shape, blocks,
             (double)c / (iters * 4.0), ms, c / (ms * 1e3));
    }
  }
  return 0;
}
