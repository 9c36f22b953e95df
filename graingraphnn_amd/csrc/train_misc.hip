// Training path, the two ends of a step that are not a cell: the loss (train.py:31-37: the masked
// squared error of the regressor, with its gradient) and the optimizer (train.py:82-91: torch.optim.Adam, possibly with
// per-group learning rates) -- each ONE launch where the recorded ops are ~30 (loss forward + backward) and the
// multi-tensor library kernels 14 (the model has ~150 parameter tensors; a launch of theirs carries at most ~30 tensors
// in its kernel arguments).  Here the tensors' fixed description is a table in DEVICE memory and a launch carries only the
// gradients' addresses.
#include "common.h"

namespace ggnn {

constexpr int AD_THREADS = 256;
constexpr int AD_CHUNK = GGNN_ADAM_CHUNK;   // elements per workgroup

// step[t]: updates tensor t has had so far (a float per tensor, as torch keeps it: a tensor without a gradient in some step
// does not advance).  What changes from step to step -- the gradients' addresses, the learning rates -- travels in the kernel
// arguments (3 KB): nothing is staged through host memory that a later step could overwrite while an earlier one is queued.
__global__ __launch_bounds__(AD_THREADS) void adam_kernel(const ggnn_adam_args A) {
  const int ti = A.chunk_tensor[blockIdx.x];
  const ggnn_adam_tensor T = A.table[ti];
  const float* __restrict__ grad = A.grad[ti];
  const int64_t i0 = (int64_t)A.chunk_index[blockIdx.x] * AD_CHUNK;
  const float s = A.step[ti] + 1.0f;
  if (grad != nullptr) {
    // torch.optim.Adam (single-tensor formulation): step_size = lr / (1 - beta1^s); denom = sqrt(v) / sqrt(1 - beta2^s) + eps
    const float lr = A.hyper != nullptr ? A.hyper[T.group] : A.lr[T.group];
    const float wd = A.hyper != nullptr ? A.hyper[GGNN_ADAM_MAX_GROUPS + T.group] : A.weight_decay[T.group];
    const float bc1 = 1.0f - powf(A.beta1, s), bc2 = 1.0f - powf(A.beta2, s);
    const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    const int64_t i1 = min(T.n, i0 + AD_CHUNK);
    // a thread's 16 elements in two batches of 8 with every load of a batch requested before the first use: one element at
    // a time the kernel was a chain of 16 memory round trips per thread
    constexpr int NB = 8;
    static_assert(AD_CHUNK % (NB * AD_THREADS) == 0, "chunk = whole batches");
#pragma unroll 1
    for (int64_t base = i0 + threadIdx.x; base < i1; base += NB * AD_THREADS) {
      float p[NB], g[NB], m[NB], v[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int64_t i = min(base + j * AD_THREADS, i1 - 1);   // (clamped: loaded, not stored)
        p[j] = T.param[i], g[j] = grad[i], m[j] = T.exp_avg[i], v[j] = T.exp_avg_sq[i];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int64_t i = base + j * AD_THREADS;
        if (i < i1) {
          float gg = g[j];
          if (wd != 0.f) gg = __builtin_fmaf(wd, p[j], gg);
          const float mn = __builtin_fmaf(A.beta1, m[j], (1.0f - A.beta1) * gg);
          const float vn = __builtin_fmaf(A.beta2, v[j], (1.0f - A.beta2) * gg * gg);
          T.exp_avg[i] = mn;
          T.exp_avg_sq[i] = vn;
          T.param[i] = p[j] - step_size * (mn / (sqrtf(vn) * inv_sqrt_bc2 + A.eps));
        }
      }
    }
  }
}

// ... and the counts advance in a launch of their own behind it (one workgroup).  (First version: the last workgroup of the
// update to arrive at a counter advanced them -- 580 atomics on one address, 17 of the kernel's 26 us.)
__global__ __launch_bounds__(AD_THREADS) void adam_advance_kernel(const ggnn_adam_args A) {
  for (int t = threadIdx.x; t < A.n_tensors; t += AD_THREADS)
    if (A.grad[t] != nullptr) A.step[t] += 1.0f;
}

// loss = scale * sum_k mean_i(mask_k[i / mask_div_k] * (pred_k[i] - target_k[i])^2), g_pred_k = d loss / d pred_k.
// GGNN_MSE_BLOCKS workgroups, each over a fixed share of the elements; their partial sums (doubles) go through the
// workspace and the last workgroup to arrive adds them in index order: the result does not depend on the arrival order.
constexpr int MSE_THREADS = 256;
__global__ __launch_bounds__(MSE_THREADS) void masked_mse_kernel(const ggnn_mse_args A) {
  __shared__ double red[MSE_THREADS / 64];
  const int64_t gtid = (int64_t)blockIdx.x * MSE_THREADS + threadIdx.x, gstride = (int64_t)GGNN_MSE_BLOCKS * MSE_THREADS;
  double total = 0.0;
#pragma unroll 1
  for (int k = 0; k < A.n_terms; ++k) {
    const int64_t n = A.n[k];
    if (n <= 0) continue;
    const float* __restrict__ p = A.pred[k];
    const float* __restrict__ y = A.target[k];
    const float* __restrict__ m = A.mask[k];
    float* __restrict__ g = A.g_pred[k];
    const int64_t div = A.mask_div[k];
    const float gs = 2.0f * A.scale / (float)n;
    float acc = 0.f;
    for (int64_t i = gtid; i < n; i += gstride) {
      const float d = p[i] - y[i], w = m ? m[div == 1 ? i : i / div] : 1.0f;
      acc = __builtin_fmaf(w * d, d, acc);
      if (g) g[i] = gs * w * d;
    }
    total += (double)acc / (double)n;
  }
  for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < MSE_THREADS / 64; ++w) t += red[w];
    A.workspace[blockIdx.x] = t;
    __threadfence();
    unsigned* counter = reinterpret_cast<unsigned*>(A.workspace + GGNN_MSE_BLOCKS);
    if (atomicAdd(counter, 1u) == GGNN_MSE_BLOCKS - 1) {
      __threadfence();
      double sum = 0.0;   // (device-scope loads: the other workgroups' partial sums come from L2)
      for (int b = 0; b < GGNN_MSE_BLOCKS; ++b) sum += __hip_atomic_load(A.workspace + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *A.loss = (float)(sum * (double)A.scale);
      *counter = 0u;
    }
  }
}

// out[b][j] = sum_r in[b][r][j] for a TALL stack of short rows (the sweep backward's per-workgroup partial sums of the
// edge-parameter gradient: 768 rows of 1 152 floats per edge type).  A workgroup owns 32 columns (8 float4) of one batch
// entry: 32 row groups x 8 column quads; a thread adds every 32nd row, eight loads in flight, and the row groups are
// combined through LDS in index order -- a fixed summation tree.
constexpr int SR_QUADS = 8, SR_GROUPS = 32;
__global__ __launch_bounds__(SR_QUADS * SR_GROUPS) void sum_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                         int64_t n_rows, int64_t n_cols) {
  __shared__ f32x4 red[SR_GROUPS][SR_QUADS];
  const int q = threadIdx.x % SR_QUADS, rg = threadIdx.x / SR_QUADS;
  const int64_t n4 = n_cols / 4, col = (int64_t)blockIdx.x * SR_QUADS + q;
  const bool live = col < n4;
  const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(in) + (int64_t)blockIdx.y * n_rows * n4 + (live ? col : 0);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int64_t r = rg;
  for (; r + 7 * SR_GROUPS < n_rows; r += 8 * SR_GROUPS) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(r + j * SR_GROUPS) * n4];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; r < n_rows; r += SR_GROUPS) acc += p[r * n4];
  red[rg][q] = acc;
  __syncthreads();
  if (rg == 0 && live) {
    f32x4 t = red[0][q];
    for (int g = 1; g < SR_GROUPS; ++g) t += red[g][q];
    reinterpret_cast<f32x4*>(out)[(int64_t)blockIdx.y * n4 + col] = t;
  }
}

// ... several such sums in one launch (ggnn_sum_rows_batch: the split-K partials of a cell's four weight gradients and the
// edge-parameter partials of its sweeps, which nothing reads before the cell's backward pass ends)
// Two ways through a problem, the ones ggnn_wgrad chooses between for its own reduction (wgrad.hip), so that a postponed
// reduction gives the bits of an immediate one: MANY rows of a small result (sum_rows_tree: the summation tree above), or a
// thread per result quad that walks the rows in index order, eight loads in flight (few rows of a large result).
__host__ __device__ inline bool sum_rows_tree(int64_t n_rows, int64_t n4) { return n_rows > 48 && n4 <= 16384; }
struct SumRowsBatch {
  ggnn_sum_rows_problem p[GGNN_SUM_ROWS_MAX];
  int blk_off[GGNN_SUM_ROWS_MAX + 1];   // first workgroup of every problem: (column blocks) x batch of them each
  int n;
};
__global__ __launch_bounds__(SR_QUADS * SR_GROUPS) void sum_rows_batch_kernel(const SumRowsBatch B) {
  __shared__ f32x4 red[SR_GROUPS][SR_QUADS];
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.blk_off[k + 1]) ++k;
  const ggnn_sum_rows_problem& P = B.p[k];
  const int q = threadIdx.x % SR_QUADS, rg = threadIdx.x / SR_QUADS;
  const int64_t n4 = P.n_cols / 4, n_rows = P.n_rows, blk = (int)blockIdx.x - B.blk_off[k];
  if (!sum_rows_tree(n_rows, n4)) {   // (uniform per workgroup)
    const int64_t nbw = (n4 + 255) / 256, bw = blk / nbw, i = (blk - bw * nbw) * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4* __restrict__ pw = reinterpret_cast<const f32x4*>(P.in) + bw * n_rows * n4 + i;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int64_t s = 0;
    for (; s + 8 <= n_rows; s += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = pw[(s + j) * n4];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
    for (; s < n_rows; ++s) acc += pw[s * n4];
    reinterpret_cast<f32x4*>(P.out)[bw * n4 + i] = acc;
    return;
  }
  const int64_t nb = (n4 + SR_QUADS - 1) / SR_QUADS;
  const int64_t b = blk / nb, col = (blk - b * nb) * SR_QUADS + q;
  const bool live = col < n4;
  const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(P.in) + b * n_rows * n4 + (live ? col : 0);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int64_t r = rg;
  for (; r + 7 * SR_GROUPS < n_rows; r += 8 * SR_GROUPS) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(r + j * SR_GROUPS) * n4];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; r < n_rows; r += SR_GROUPS) acc += p[r * n4];
  red[rg][q] = acc;
  __syncthreads();
  if (rg == 0 && live) {
    f32x4 t = red[0][q];
    for (int g = 1; g < SR_GROUPS; ++g) t += red[g][q];
    reinterpret_cast<f32x4*>(P.out)[b * n4 + col] = t;
  }
}

int launch_sum_rows(const float* in, float* out, int64_t n_rows, int64_t n_cols, int batch, hipStream_t stream) {
  if (!in || !out || n_rows <= 0 || n_cols <= 0 || (n_cols & 3) || batch < 1 || batch > 65535) return GGNN_EINVAL;
  if (!aligned16(in) || !aligned16(out)) return GGNN_EINVAL;
  const int64_t nb = (n_cols / 4 + SR_QUADS - 1) / SR_QUADS;
  if (nb >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)nb, (unsigned)batch), dim3(SR_QUADS * SR_GROUPS), 0, stream, in, out, n_rows,
                     n_cols);
  return launch_status();
}

}  // namespace ggnn

extern "C" int ggnn_sum_rows(const float* in, float* out, int64_t n_rows, int64_t n_cols, int32_t batch, ggnn_stream_t stream) {
  return ggnn::launch_sum_rows(in, out, n_rows, n_cols, batch, (hipStream_t)stream);
}

extern "C" int ggnn_sum_rows_batch(const ggnn_sum_rows_problem* problems, int n_problems, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!problems || n_problems < 1 || n_problems > GGNN_SUM_ROWS_MAX) return GGNN_EINVAL;
  SumRowsBatch B;
  B.n = n_problems;
  B.blk_off[0] = 0;
  for (int k = 0; k < n_problems; ++k) {
    const ggnn_sum_rows_problem& P = problems[k];
    if (!P.in || !P.out || P.n_rows <= 0 || P.n_cols <= 0 || (P.n_cols & 3) || P.batch < 1 || !aligned16(P.in) || !aligned16(P.out))
      return GGNN_EINVAL;
    const int64_t n4 = P.n_cols / 4;
    const int64_t nb = (sum_rows_tree(P.n_rows, n4) ? (n4 + SR_QUADS - 1) / SR_QUADS : (n4 + 255) / 256) * P.batch;
    if (B.blk_off[k] + nb >= INT32_MAX) return GGNN_EINVAL;
    B.p[k] = P;
    B.blk_off[k + 1] = B.blk_off[k] + (int)nb;
  }
  hipLaunchKernelGGL(sum_rows_batch_kernel, dim3((unsigned)B.blk_off[n_problems]), dim3(SR_QUADS * SR_GROUPS), 0, (hipStream_t)stream, B);
  return launch_status();
}

extern "C" int ggnn_adam_step(const ggnn_adam_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  const ggnn_adam_args& A = *args;
  if (!A.table || !A.chunk_tensor || !A.chunk_index || !A.step || A.n_chunks <= 0) return GGNN_EINVAL;
  if (A.n_tensors < 1 || A.n_tensors > GGNN_ADAM_MAX_TENSORS) return GGNN_EINVAL;
  if (!(A.beta1 >= 0.f && A.beta1 < 1.f) || !(A.beta2 >= 0.f && A.beta2 < 1.f) || !(A.eps >= 0.f)) return GGNN_EINVAL;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)A.n_chunks), dim3(AD_THREADS), 0, (hipStream_t)stream, A);
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(AD_THREADS), 0, (hipStream_t)stream, A);
  return launch_status();
}

extern "C" int ggnn_masked_mse(const ggnn_mse_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || !args->loss || !args->workspace || args->n_terms < 1 || args->n_terms > GGNN_MSE_MAX_TERMS) return GGNN_EINVAL;
  for (int k = 0; k < args->n_terms; ++k) {
    if (args->n[k] < 0 || (args->n[k] > 0 && (!args->pred[k] || !args->target[k]))) return GGNN_EINVAL;
    if (args->mask[k] && args->mask_div[k] < 1) return GGNN_EINVAL;
  }
  hipLaunchKernelGGL(masked_mse_kernel, dim3(GGNN_MSE_BLOCKS), dim3(MSE_THREADS), 0, (hipStream_t)stream, *args);
  return launch_status();
}

// ---- ggnn_train_input_rows: [x | 0 .. | 1 0 0 0], the data part of the weight gradients' B operands (include/ggnn.h) ----
namespace ggnn {
struct TrainRowsBatch {
  ggnn_train_rows_problem p[GGNN_TRAIN_ROWS_MAX];
  int blk_off[GGNN_TRAIN_ROWS_MAX + 1];
  int n;
};
// a thread per output float4 (Fp / 4 + 1 <= 4 of them per row)
__global__ __launch_bounds__(256) void train_input_rows_kernel(const TrainRowsBatch B) {
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.blk_off[k + 1]) ++k;
  const ggnn_train_rows_problem& P = B.p[k];
  const int Fp = (P.F + 3) & ~3, Q = Fp / 4 + 1;
  const int64_t t = (int64_t)((int)blockIdx.x - B.blk_off[k]) * 256 + threadIdx.x;
  if (t >= P.N * Q) return;
  const int64_t n = t / Q;
  const int c0 = (int)(t - n * Q) * 4;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j;
    const float xv = P.x[n * P.ldx + min(c, P.F - 1)];
    v[j] = c < P.F ? xv : (c == Fp ? 1.0f : 0.0f);
  }
  *reinterpret_cast<f32x4*>(P.out + n * P.ldo + c0) = f32x4{v[0], v[1], v[2], v[3]};
}
}  // namespace ggnn

extern "C" int ggnn_train_input_rows(const ggnn_train_rows_problem* problems, int n_problems, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!problems || n_problems < 1 || n_problems > GGNN_TRAIN_ROWS_MAX) return GGNN_EINVAL;
  TrainRowsBatch B;
  B.n = 0;
  B.blk_off[0] = 0;
  for (int k = 0; k < n_problems; ++k) {
    const ggnn_train_rows_problem& P = problems[k];
    if (P.N < 0) return GGNN_EINVAL;
    if (P.N == 0) continue;
    const int Fp = (P.F + 3) & ~3;
    if (!P.x || !P.out || P.F < 3 || P.F > 12 || P.ldx < P.F || P.ldo < Fp + 4 || (P.ldo & 3) || !aligned16(P.out)) return GGNN_EINVAL;
    const int64_t blocks = (P.N * (Fp / 4 + 1) + 255) / 256;
    if (B.blk_off[B.n] + blocks >= 0x7fffffff) return GGNN_EINVAL;
    B.p[B.n] = P;
    B.blk_off[B.n + 1] = B.blk_off[B.n] + (int)blocks;
    ++B.n;
  }
  if (B.n == 0) return 0;
  hipLaunchKernelGGL(train_input_rows_kernel, dim3((unsigned)B.blk_off[B.n]), dim3(256), 0, (hipStream_t)stream, B);
  return launch_status();
}
