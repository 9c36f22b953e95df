// Training path: the packed weight matrices of a cell from its parameters, and the way back.
// train_pack.PackPlan describes the packing as index tables -- every entry of the nine packed matrices is the sum of up to
// L (<= 3) elements of  flat2 = [ all parameters | the batched products (scale K_ext) Q_ext | 0 ]  -- and the recorded torch
// ops for it were a cat, two gathers, a bmm and a sum each way plus fills: ~9 launches forward and ~17 backward per cell, one
// of them a library GEMM, at ~5 us each inside a replayed step.  Here: two launches forward (the products straight from the
// parameters through the operands' index tables; the gather-sum), four backward (the gradient of flat2 through the inverse
// table, the two operand gradients of the products, the gradient of the parameters).  Products of [r <= 110, 96] x [96, c <=
// 110] blocks (c <= 128): plain fp32 fmas on operands staged in LDS.
#include <algorithm>

#include "common.h"

namespace ggnn {

constexpr int PK_THREADS = 128;

// flat2[j], the zero slot reading 0 -- as an unconditional (clamped) load and a select: a load under a branch makes every
// gathered element its own dependent round trip, which is what the first version of these kernels spent its time in
// With A.params (ABI 25) an entry that addresses a parameter carries (tensor + 1) in its high bits and the offset inside that
// tensor below them, and is read where the tensor lies (no concatenated copy of the parameters in flat2[0 .. n_flat)): one more
// dependent load, of the tensor's address out of a table of a few hundred entries.  Entries of the product region and the
// zero slot are plain flat2 indices either way.
__device__ __forceinline__ float pk_read(const ggnn_pack_args& A, int64_t e) {
  const int64_t t = e >> GGNN_PACK_TENSOR_SHIFT, lo = e & (((int64_t)1 << GGNN_PACK_TENSOR_SHIFT) - 1);
  const bool is_zero = e == A.zero;
  const float* base = A.flat2;
  if (A.params) {                                              // (uniform branch on a kernel argument)
    const float* pt = A.params[t ? t - 1 : 0];                 // (unconditional, clamped: no load under a per-lane branch)
    base = t ? pt : A.flat2;
  }
  const float x = base[is_zero ? A.n_flat : lo];               // (flat2[n_flat]: the first product entry, always there)
  return is_zero ? 0.f : x;
}
// flat2[n_flat + (b r + row) c + col] = sum_k coef flat2[kq_idx[(b r + row) 96 + k]] * flat2[kq_idx[n_k + (b 96 + k) c + col]]
// One workgroup per (product b, slab of PK_ROWS rows): Q_b [96, c] is gathered into LDS once (40 KB) and serves the slab's rows;
// a thread owns one column and walks the slab's rows.  (First version: a workgroup per output row reading Q through the index
// table for every row -- 96 dependent index -> value round trips per thread: 40 us per cell where this takes a few.)
constexpr int PK_ROWS = 4;   // rows of a product per workgroup: 12 x 27 workgroups per launch at the shipped shapes
// kq[i] = flat2[kq_idx[i]] (x coef on the K side): the operands of the products, dense (also what the backward reads)
// The cells of a step in one launch each (ggnn_pack_weights_batch: the encoder's and the decoder's packing are independent and
// each of these kernels is a few microseconds of latency): workgroup -> (cell, its block) through the offsets.
struct PackBatch {
  ggnn_pack_args a[GGNN_PACK_MAX];
  int off[GGNN_PACK_MAX + 1];
  int n;
};
struct PackBwdBatch {
  ggnn_pack_bwd_args a[GGNN_PACK_MAX];
  int off[GGNN_PACK_MAX + 1];
  int n;
};
template <class B>
__device__ __forceinline__ int pk_cell(const B& b, int& blk) {
  int k = 0;
  while (k + 1 < b.n && (int)blockIdx.x >= b.off[k + 1]) ++k;
  blk = (int)blockIdx.x - b.off[k];
  return k;
}
__global__ __launch_bounds__(256) void pack_operands_kernel(const PackBatch B) {
  int blk;
  const ggnn_pack_args& A = B.a[pk_cell(B, blk)];
  const int64_t n_k = (int64_t)A.nb * A.r * GGNN_C, n_kq = n_k + (int64_t)A.nb * GGNN_C * A.c;
  const int64_t i = (int64_t)blk * 256 + threadIdx.x;
  if (i < n_kq) A.kq[i] = pk_read(A, A.kq_idx[i]) * (i < n_k ? A.coef : 1.0f);
}
__global__ __launch_bounds__(PK_THREADS) void pack_products_kernel(const PackBatch B) {
  extern __shared__ float lds[];
  int blk;
  const ggnn_pack_args& A = B.a[pk_cell(B, blk)];
  float* __restrict__ q = lds;                       // [96][c]
  float* __restrict__ kr = lds + GGNN_C * A.c;       // [PK_ROWS][96]
  const int64_t b = blk % A.nb, row0 = (int64_t)(blk / A.nb) * PK_ROWS;
  const int tid = threadIdx.x, nrow = (int)min((int64_t)PK_ROWS, A.r - row0);
  const int64_t n_k = (int64_t)A.nb * A.r * GGNN_C;
  const float* __restrict__ qd = A.kq + n_k + b * GGNN_C * A.c;
  for (int i = tid; i < GGNN_C * A.c; i += PK_THREADS) q[i] = qd[i];
  const float* __restrict__ kd = A.kq + (b * A.r + row0) * GGNN_C;
  for (int i = tid; i < nrow * GGNN_C; i += PK_THREADS) kr[i] = kd[i];
  __syncthreads();
  for (int col = tid; col < A.c; col += PK_THREADS) {
    float acc[PK_ROWS];
#pragma unroll
    for (int j = 0; j < PK_ROWS; ++j) acc[j] = 0.f;
    for (int k = 0; k < GGNN_C; ++k) {
      const float qv = q[k * A.c + col];
#pragma unroll
      for (int j = 0; j < PK_ROWS; ++j) acc[j] = __builtin_fmaf(kr[j * GGNN_C + k], qv, acc[j]);   // (rows past nrow: stale LDS, not stored)
    }
    for (int j = 0; j < nrow; ++j) A.flat2[A.n_flat + (b * A.r + row0 + j) * A.c + col] = acc[j];
  }
}

// packed[i] = sum_t flat2[idx3[i L + t]]   (terms in the order t = 0, 1, 2: torch's sum over the last dimension of [n, L])
__global__ __launch_bounds__(256) void pack_gather_kernel(const PackBatch B) {
  int blk;
  const ggnn_pack_args& A = B.a[pk_cell(B, blk)];
  const int64_t i = (int64_t)blk * 256 + threadIdx.x;
  if (i >= A.n_packed) return;
  float s = pk_read(A, A.idx3[i * A.L]);
  for (int t = 1; t < A.L; ++t) s += pk_read(A, A.idx3[i * A.L + t]);
  A.packed[i] = s;
}

// ---- backward ----
__device__ __forceinline__ float pk_gpacked(const ggnn_pack_bwd_args& A, int64_t i) {
  if (i >= A.fwd.n_packed) return 0.f;   // (the inverse table's padding)
  int s = 0;
  while (s + 1 < GGNN_PACK_OUTPUTS && i >= A.g_off[s + 1]) ++s;
  if (A.g_out[s] == nullptr) return 0.f;
  const int64_t e = i - A.g_off[s], row = e / A.g_w[s], col = e - row * A.g_w[s];   // (a view of a wider matrix: no copy)
  return A.g_out[s][row * A.g_rs[s] + col * A.g_cs[s]];
}
// g_flat2[j] = sum_m g_packed[inv[j M + m]]
__global__ __launch_bounds__(256) void pack_bwd_flat2_kernel(const PackBwdBatch B) {
  int blk;
  const ggnn_pack_bwd_args& A = B.a[pk_cell(B, blk)];
  const int64_t j = (int64_t)blk * 256 + threadIdx.x;
  if (j >= A.n_flat2) return;
  float s = pk_gpacked(A, A.inv[j * A.inv_m]);
  for (int m = 1; m < A.inv_m; ++m) s += pk_gpacked(A, A.inv[j * A.inv_m + m]);
  A.g_flat2[j] = s;
}
// g_kq[(b r + row) 96 + k] = coef sum_col g_m[b, row, col] Q[b, k, col]        (gradient of the K operand, coefficient applied)
// Same slabs as the forward: Q_b in LDS, PK_ROWS rows of g_m beside it; a thread owns one k and walks the slab's rows.
__global__ __launch_bounds__(PK_THREADS) void pack_bwd_k_kernel(const PackBwdBatch B) {
  extern __shared__ float lds[];
  int blk;
  const ggnn_pack_bwd_args& A = B.a[pk_cell(B, blk)];
  const ggnn_pack_args& F = A.fwd;
  const int cp = F.c | 1;                             // odd row stride: a thread per k reads q[k * cp + col] conflict-free
  float* __restrict__ q = lds;                        // [96][cp]
  float* __restrict__ gm = lds + GGNN_C * cp;         // [PK_ROWS][c]
  const int64_t b = blk % F.nb, row0 = (int64_t)(blk / F.nb) * PK_ROWS;
  const int tid = threadIdx.x, nrow = (int)min((int64_t)PK_ROWS, F.r - row0);
  const int64_t n_k = (int64_t)F.nb * F.r * GGNN_C;
  const float* __restrict__ qd = F.kq + n_k + b * GGNN_C * F.c;
  for (int i = tid; i < GGNN_C * F.c; i += PK_THREADS) q[(i / F.c) * cp + i % F.c] = qd[i];
  for (int i = tid; i < nrow * F.c; i += PK_THREADS) gm[i] = A.g_flat2[F.n_flat + (b * F.r + row0) * F.c + i];
  __syncthreads();
  if (tid < GGNN_C) {
    float acc[PK_ROWS];
#pragma unroll
    for (int j = 0; j < PK_ROWS; ++j) acc[j] = 0.f;
    for (int col = 0; col < F.c; ++col) {
      const float qv = q[tid * cp + col];
#pragma unroll
      for (int j = 0; j < PK_ROWS; ++j) acc[j] = __builtin_fmaf(gm[j * F.c + col], qv, acc[j]);
    }
    for (int j = 0; j < nrow; ++j) A.g_kq[(b * F.r + row0 + j) * GGNN_C + tid] = acc[j] * F.coef;
  }
}
// g_kq[n_k + (b 96 + k) c + col] = sum_row (coef K[b, row, k]) g_m[b, row, col]   (gradient of the Q operand)
// One workgroup per (product b, slab of PK_ROWS values of k): the slab's K columns [r][PK_ROWS] in LDS; g_m rows stream
// from memory, coalesced over the column a thread owns.
__global__ __launch_bounds__(PK_THREADS) void pack_bwd_q_kernel(const PackBwdBatch B) {
  extern __shared__ float lds[];
  int blk;
  const ggnn_pack_bwd_args& A = B.a[pk_cell(B, blk)];
  const ggnn_pack_args& F = A.fwd;
  float* __restrict__ kc = lds;                       // [r][PK_ROWS]
  const int64_t b = blk % F.nb, k0 = (int64_t)(blk / F.nb) * PK_ROWS;
  const int tid = threadIdx.x;
  for (int i = tid; i < F.r * PK_ROWS; i += PK_THREADS)   // (kq's K side already carries the coefficient)
    kc[i] = F.kq[(b * F.r + i / PK_ROWS) * GGNN_C + k0 + i % PK_ROWS];
  __syncthreads();
  const int64_t n_k = (int64_t)F.nb * F.r * GGNN_C;
  for (int col = tid; col < F.c; col += PK_THREADS) {
    float acc[PK_ROWS];
#pragma unroll
    for (int j = 0; j < PK_ROWS; ++j) acc[j] = 0.f;
    for (int row = 0; row < F.r; ++row) {
      const float g = A.g_flat2[F.n_flat + (b * F.r + row) * F.c + col];
#pragma unroll
      for (int j = 0; j < PK_ROWS; ++j) acc[j] = __builtin_fmaf(kc[row * PK_ROWS + j], g, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < PK_ROWS; ++j) A.g_kq[n_k + (b * GGNN_C + k0 + j) * F.c + col] = acc[j];
  }
}
// g_flat[p] = g_flat2[p] + sum_m g_kq[inv_kq[p M + m]]
__global__ __launch_bounds__(256) void pack_bwd_params_kernel(const PackBwdBatch B) {
  int blk;
  const ggnn_pack_bwd_args& A = B.a[pk_cell(B, blk)];
  const int64_t p = (int64_t)blk * 256 + threadIdx.x;
  if (p >= A.fwd.n_flat) {
    if (p < A.fwd.n_flat + A.n_tail) A.g_flat[p] = 0.f;   // parameters read without effect (the encoder's forget gate)
    return;
  }
  float s = 0.f;
  for (int m = 0; m < A.inv_kq_m; ++m) {
    const int64_t i = A.inv_kq[p * A.inv_kq_m + m];
    if (i < A.n_kq) s += A.g_kq[i];
  }
  A.g_flat[p] = A.g_flat2[p] + s;
}

static bool pack_args_ok(const ggnn_pack_args& A) {
  return A.flat2 && A.kq_idx && A.kq && A.idx3 && A.packed && A.n_flat > 0 && A.nb > 0 && A.r > 0 && A.c > 0 && A.r <= 4096 &&
         A.c <= 128 && A.L >= 1 && A.L <= 8 && A.n_packed > 0 && A.zero == A.n_flat + (int64_t)A.nb * A.r * A.c &&
         (int64_t)A.nb * A.r < INT32_MAX && (A.n_packed + 255) / 256 < INT32_MAX;
}

static bool pack_bwd_args_ok(const ggnn_pack_bwd_args& A) {
  const ggnn_pack_args& F = A.fwd;
  if (!pack_args_ok(F)) return false;
  if (!A.inv || !A.inv_kq || !A.g_flat2 || !A.g_kq || !A.g_flat || A.inv_m < 1 || A.inv_m > 16 || A.inv_kq_m < 1 || A.inv_kq_m > 16)
    return false;
  if (A.n_tail < 0 || A.n_flat2 != F.zero + 1 || A.n_kq != (int64_t)F.nb * GGNN_C * (F.r + F.c) || A.g_off[0] != 0 ||
      A.g_off[GGNN_PACK_OUTPUTS] != F.n_packed)
    return false;
  for (int s = 0; s < GGNN_PACK_OUTPUTS; ++s)
    if (A.g_off[s + 1] < A.g_off[s] || (A.g_out[s] != nullptr && A.g_w[s] < 1)) return false;
  return true;
}

// block offsets of a launch whose cell k takes count(k) workgroups; false if they do not fit an int
template <class B, class Count>
static bool pk_offsets(B& b, int n, Count count) {
  b.n = n;
  b.off[0] = 0;
  for (int k = 0; k < n; ++k) {
    const int64_t c = count(k);
    if (c <= 0 || b.off[k] + c >= INT32_MAX) return false;
    b.off[k + 1] = b.off[k] + (int)c;
  }
  return true;
}

}  // namespace ggnn

extern "C" int ggnn_pack_weights_batch(const ggnn_pack_args* args, int n_cells, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_cells < 1 || n_cells > GGNN_PACK_MAX) return GGNN_EINVAL;
  PackBatch B;
  size_t lds = 0;
  for (int k = 0; k < n_cells; ++k) {
    if (!pack_args_ok(args[k])) return GGNN_EINVAL;
    B.a[k] = args[k];
    lds = std::max(lds, (size_t)(GGNN_C * args[k].c + PK_ROWS * GGNN_C) * sizeof(float));
  }
  hipStream_t st = (hipStream_t)stream;
  auto A = [&](int k) -> const ggnn_pack_args& { return args[k]; };
  if (!pk_offsets(B, n_cells, [&](int k) { return ((int64_t)A(k).nb * GGNN_C * (A(k).r + A(k).c) + 255) / 256; })) return GGNN_EINVAL;
  hipLaunchKernelGGL(pack_operands_kernel, dim3((unsigned)B.off[n_cells]), dim3(256), 0, st, B);
  if (!pk_offsets(B, n_cells, [&](int k) { return (int64_t)A(k).nb * ((A(k).r + PK_ROWS - 1) / PK_ROWS); })) return GGNN_EINVAL;
  hipLaunchKernelGGL(pack_products_kernel, dim3((unsigned)B.off[n_cells]), dim3(PK_THREADS), lds, st, B);
  if (!pk_offsets(B, n_cells, [&](int k) { return (A(k).n_packed + 255) / 256; })) return GGNN_EINVAL;
  hipLaunchKernelGGL(pack_gather_kernel, dim3((unsigned)B.off[n_cells]), dim3(256), 0, st, B);
  return launch_status();
}

extern "C" int ggnn_pack_weights_backward_batch(const ggnn_pack_bwd_args* args, int n_cells, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_cells < 1 || n_cells > GGNN_PACK_MAX) return GGNN_EINVAL;
  PackBwdBatch B;
  size_t lds_k = 0, lds_q = 0;
  for (int k = 0; k < n_cells; ++k) {
    if (!pack_bwd_args_ok(args[k])) return GGNN_EINVAL;
    B.a[k] = args[k];
    const ggnn_pack_args& F = args[k].fwd;
    lds_k = std::max(lds_k, (size_t)(GGNN_C * (F.c | 1) + PK_ROWS * F.c) * sizeof(float));
    lds_q = std::max(lds_q, (size_t)F.r * PK_ROWS * sizeof(float));
  }
  hipStream_t st = (hipStream_t)stream;
  auto F = [&](int k) -> const ggnn_pack_args& { return args[k].fwd; };
  if (!pk_offsets(B, n_cells, [&](int k) { return (args[k].n_flat2 + 255) / 256; })) return GGNN_EINVAL;
  hipLaunchKernelGGL(pack_bwd_flat2_kernel, dim3((unsigned)B.off[n_cells]), dim3(256), 0, st, B);
  if (!pk_offsets(B, n_cells, [&](int k) { return (int64_t)F(k).nb * ((F(k).r + PK_ROWS - 1) / PK_ROWS); })) return GGNN_EINVAL;
  hipLaunchKernelGGL(pack_bwd_k_kernel, dim3((unsigned)B.off[n_cells]), dim3(PK_THREADS), lds_k, st, B);
  static_assert(GGNN_C % PK_ROWS == 0, "slabs of k");
  if (!pk_offsets(B, n_cells, [&](int k) { return (int64_t)F(k).nb * (GGNN_C / PK_ROWS); })) return GGNN_EINVAL;
  hipLaunchKernelGGL(pack_bwd_q_kernel, dim3((unsigned)B.off[n_cells]), dim3(PK_THREADS), lds_q, st, B);
  if (!pk_offsets(B, n_cells, [&](int k) { return (F(k).n_flat + args[k].n_tail + 255) / 256; })) return GGNN_EINVAL;
  hipLaunchKernelGGL(pack_bwd_params_kernel, dim3((unsigned)B.off[n_cells]), dim3(256), 0, st, B);
  return launch_status();
}

extern "C" int ggnn_pack_weights(const ggnn_pack_args* args, ggnn_stream_t stream) { return ggnn_pack_weights_batch(args, 1, stream); }

extern "C" int ggnn_pack_weights_backward(const ggnn_pack_bwd_args* args, ggnn_stream_t stream) {
  return ggnn_pack_weights_backward_batch(args, 1, stream);
}
