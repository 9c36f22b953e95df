// Node-level projection GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32):
//   out[M, ncols] = [X[:, :F] | H] . Wp^T + bias
// One launch per node type and cell produces, for every gate and edge type, the per-node
// value (as source), key-free score operands u = W_k^T q / sqrt(96) (as destination) and summed-skip
// pre-activations that replace the reference's per-EDGE linears (periodGATconv.py:216-218, :186;
// what the rows of Wp hold: graingraphnn_amd/packing.py).
//
// K = roundup4(F) + 96 <= 108 is short and M is long, so the kernel is WEIGHT-STATIONARY and
// persistent: one 8-wave workgroup per CU keeps a 96-column x K weight tile in LDS for its
// whole life (read-only after the prologue), and every WAVE streams its own 16-node input
// tiles past it through a wave-private LDS stage.  Nothing is shared between waves except
// the constant weight tile, so the main loop has NO workgroup barrier: two waves per SIMD
// run decoupled and one wave's staging / epilogue hides under the other's MFMA sweep:
//   * while the sweep of tile t runs (26 k-steps x 6 MFMAs), the global loads of tile t+1
//     are in flight into registers (6 x 16 B + <= 3 x 4 B per lane); they are written to the
//     same LDS stage right after the sweep (LDS accesses of one wave execute in order);
//   * the sweep is fully unrolled (K is a template parameter) so operand fragments are read
//     from LDS well ahead of the MFMA that consumes them;
//   * the 16-byte output stores of tile t drain while tile t+1 is swept.
// The grid is (ncols / 96) column tiles x as many row splits as fit 256 CUs.
//
// Orientation: the WEIGHT tile is the MFMA A operand and the NODE tile the B operand, so
// that each lane ends up with 4 consecutive output columns of one node -> 16-byte stores.
//   A[i][k] = Wp[n0+i][k]   lane l holds i = l&15, k = k0 + (l>>4)
//   B[k][j] = XH[m0+j][k]   lane l holds j = l&15, k = k0 + (l>>4)
//   D[i][j] -> lane l, reg r: i = 4*(l>>4) + r, j = l&15
// LDS rows have stride Kp + 2 floats (= 2 * odd), which makes the ds_read_b32 operand
// reads bank-conflict-free (bank = (row*ld + k) mod 32 covers 0..31 over 16 rows x 2 k).
#include <algorithm>

#include "common.h"
#include "project_batch.h"

namespace ggnn {

constexpr int PJ_BM = 16;      // nodes per wave tile
constexpr int PJ_BN = 96;      // output columns per workgroup
constexpr int PJ_WAVES = 8;    // waves per workgroup (two per SIMD)

constexpr int pj_lds_floats(int k2) { return (PJ_BN + PJ_WAVES * PJ_BM) * (12 + k2 + 2); }

template <int FP, int K2>
__device__ __forceinline__ void project_body(const ggnn_project_args& A, int blk, int m_splits, float* smem) {
  constexpr int KP = FP + K2;
  constexpr int LD = KP + 2;
  constexpr int NVH = K2 > 0 ? K2 / 4 : 1;                   // 16-byte pieces per hidden row (24)
  constexpr int NH = K2 > 0 ? (PJ_BM * NVH) / 64 : 0;        // hidden pieces per lane per tile (6 or 0)
  constexpr int NX = (PJ_BM * FP) / 64;                      // feature floats per lane per tile (1..3)
  static_assert((K2 == 0 || (PJ_BM * NVH) % 64 == 0) && (PJ_BM * FP) % 64 == 0, "tile / lane mismatch");
  float* s_w = smem;                                          // [PJ_BN][LD]
  float* s_xs = smem + PJ_BN * LD;                            // [PJ_WAVES][PJ_BM * LD]
  const float* __restrict__ X = A.X;
  const float* __restrict__ H = A.H;
  const float* __restrict__ Wp = A.Wp;
  const float* __restrict__ bias = A.bias;
  float* __restrict__ out = A.out;
  const int64_t ldx = A.ldx, ldh = A.ldh, M = A.M, ldo = A.ldo;
  const int F = A.F, ncols = A.ncols;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb_n = ncols / PJ_BN;
  const int bn = blk % nb_n, ms = blk / nb_n;
  const int n0 = bn * PJ_BN;

  // ---- prologue: the weight tile, shared and constant from here on ----
  {
    constexpr int nv = KP / 4;
    for (int idx = tid; idx < PJ_BN * nv; idx += PJ_WAVES * 64) {
      const int r = idx / nv, c4 = idx - r * nv;
      const f32x4 v = *reinterpret_cast<const f32x4*>(Wp + (int64_t)(n0 + r) * KP + 4 * c4);
      float2* dst = reinterpret_cast<float2*>(&s_w[r * LD + 4 * c4]);
      dst[0] = make_float2(v.x, v.y);
      dst[1] = make_float2(v.z, v.w);
    }
  }
  __syncthreads();  // the only workgroup barrier

  // 16-node tiles of this workgroup's row split, dealt round-robin to its waves
  const int64_t n_mt = (M + PJ_BM - 1) / PJ_BM;
  const int64_t per = (n_mt + m_splits - 1) / m_splits;
  const int64_t mt_lo = ms * per + wave, mt_hi = min(n_mt, (ms + 1) * per);
  if (mt_lo >= mt_hi) return;
  float* sx = s_xs + wave * (PJ_BM * LD);

  // ---- global -> register stage of one node tile (issued early, written to LDS late) ----
  f32x4 rh[NH > 0 ? NH : 1];
  float rx[NX];
  auto load_tile = [&](int64_t mt) {
    const int64_t m0 = mt * PJ_BM;
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      // unconditional (clamped) loads: rows past M are never stored, feature columns past F
      // meet zero weight columns -- a load under `if` would drag a wait to the branch merge
      const int idx = lane + it * 64, r = idx / FP, k = min(idx - r * FP, F - 1);
      const int64_t m = min(m0 + r, M - 1);
      rx[it] = X[m * ldx + k];
    }
#pragma unroll
    for (int it = 0; it < NH; ++it) {
      const int idx = lane + it * 64, r = idx / NVH, c4 = idx - r * NVH;
      const int64_t m = min(m0 + r, M - 1);
      rh[it] = *reinterpret_cast<const f32x4*>(H + m * ldh + 4 * c4);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int idx = lane + it * 64, r = idx / FP, k = idx - r * FP;
      sx[r * LD + k] = rx[it];
    }
#pragma unroll
    for (int it = 0; it < NH; ++it) {
      const int idx = lane + it * 64, r = idx / NVH, c4 = idx - r * NVH;
      float2* dst = reinterpret_cast<float2*>(&sx[r * LD + FP + 4 * c4]);
      dst[0] = make_float2(rh[it].x, rh[it].y);
      dst[1] = make_float2(rh[it].z, rh[it].w);
    }
  };

  const int lr = lane & 15, lq = lane >> 4;
  const float* pw = &s_w[lr * LD + lq];
  const float* px = &sx[lr * LD + lq];
  f32x4 bv[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) bv[a] = *reinterpret_cast<const f32x4*>(bias + n0 + a * 16 + 4 * lq);

  // Waves w and w + 4 share a SIMD and run the same program: started together they would sweep
  // together and reach their load / store phases together, leaving the matrix pipe idle.  A
  // half-sweep head start for one of them makes the phases complementary for the whole kernel
  // (there is no barrier to re-align them).  Speed only.
  if (K2 > 0 && wave >= 4) __builtin_amdgcn_s_sleep(40);  // 40 x 64 clocks ~ half a sweep
  load_tile(mt_lo);
  store_tile();
  for (int64_t mt = mt_lo; mt < mt_hi; mt += PJ_WAVES) {
    const bool has_next = mt + PJ_WAVES < mt_hi;
    if (has_next) load_tile(mt + PJ_WAVES);  // in flight during the sweep below
    __builtin_amdgcn_sched_barrier(0);       // loads stay above the sweep, the stage write above the reads

    // ---- sweep: 16 nodes x 96 columns = 6 accumulator tiles, K fully unrolled.  Operand
    // fragments are double-buffered in registers: the 7 LDS reads of k-step s+1 are issued
    // between the 6 MFMAs of k-step s (sched_group_barrier pins the interleave), so no MFMA
    // waits on an LDS read issued right in front of it. ----
    f32x4 acc[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wf[2][6], xf[2];
#pragma unroll
    for (int a = 0; a < 6; ++a) wf[0][a] = pw[a * 16 * LD];
    xf[0] = px[0];
#pragma unroll
    for (int ks = 0; ks < KP / 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < KP / 4) {
#pragma unroll
        for (int a = 0; a < 6; ++a) wf[nxt][a] = pw[a * 16 * LD + 4 * (ks + 1)];
        xf[nxt] = px[4 * (ks + 1)];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cur][a], xf[cur], acc[a], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA ...
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // ... two DS reads
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_wave_barrier();
    if (has_next) store_tile();  // same wave, in-order LDS: lands after the reads above

    // ---- epilogue: + bias, 16-byte stores (4 consecutive columns of one node per lane) ----
    const int64_t m = mt * PJ_BM + lr;
    if (m < M) {
#pragma unroll
      for (int a = 0; a < 6; ++a)
        *reinterpret_cast<f32x4*>(out + m * ldo + n0 + a * 16 + 4 * lq) = acc[a] + bv[a];
    }
  }
}

template <int K2>
__global__ __launch_bounds__(PJ_WAVES * 64, 1) void project_kernel(const ProjectBatch B) {
  __shared__ __attribute__((aligned(16))) float smem[pj_lds_floats(K2)];
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const ggnn_project_args& A = B.a[k];
  const int blk = (int)blockIdx.x - B.wg_off[k], Fp = (A.F + 3) & ~3;
  if (Fp == 4) project_body<4, K2>(A, blk, B.m_splits[k], smem);
  else if (Fp == 8) project_body<8, K2>(A, blk, B.m_splits[k], smem);
  else project_body<12, K2>(A, blk, B.m_splits[k], smem);
}

}  // namespace ggnn

int ggnn_project_x6(const ggnn::ProjectBatch& B, int n_wg, hipStream_t s);

extern "C" int ggnn_project_batch(const ggnn_project_args* args, int n_problems, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_problems < 1 || n_problems > PJ_MAX_PROBLEMS) return GGNN_EINVAL;
  ProjectBatch B;
  B.n = n_problems;
  const int k2 = args[0].k2;
  double work[PJ_MAX_PROBLEMS], total = 0;
  for (int k = 0; k < n_problems; ++k) {
    const ggnn_project_args& A = args[k];
    if (!A.X || !A.Wp || !A.bias || !A.out || A.M <= 0) return GGNN_EINVAL;
    if (A.F < 1 || A.F > 12 || A.ldx < A.F) return GGNN_EINVAL;
    if (A.k2 != k2 || (k2 != 0 && k2 != C)) return GGNN_EINVAL;
    if (k2 != 0 && (!A.H || A.ldh < k2 || (A.ldh & 3) || !aligned16(A.H))) return GGNN_EINVAL;
    if (A.ncols <= 0 || A.ncols % PJ_BN != 0 || A.ldo < A.ncols || (A.ldo & 3)) return GGNN_EINVAL;
    const int prec = A.precision & ~GGNN_OUT_BLOCK_MAJOR;
    if (A.precision != 0 &&
        ((prec != 0 && prec != GGNN_PRECISION_BF16 && prec != GGNN_PRECISION_F16X2) || k2 != C || gemm_mode() != GGNN_GEMM_BF16X6))
      return GGNN_EINVAL;   // (the one- and three-product forms and the block-major output exist for the decoder shape on the split kernel only)
    if (!aligned16(A.Wp) || !aligned16(A.bias) || !aligned16(A.out)) return GGNN_EINVAL;
    // a tile's first row is a 64-bit base; only the offsets INSIDE a 16-row tile are formed in 32 bits
    if (16 * std::max(A.ldo, std::max(A.ldx, A.ldh)) >= INT32_MAX || A.M >= ((int64_t)1 << 40)) return GGNN_EINVAL;
    work[k] = (double)A.M * A.ncols;
    total += work[k];
  }
  // decoder (K ~ 104): MFMA-bound, one workgroup per CU; encoder (K <= 12): store-bound with a
  // tiny LDS footprint, so two workgroups per CU keep more stores in flight.  The workgroups are
  // dealt to the problems in proportion to their output size.
  const int budget = k2 ? num_cu() : 2 * num_cu();
  // row splits per problem: proportional to the output size, rounded DOWN (one workgroup more than
  // the budget would wait for a second round and double the launch), then the spare workgroups
  // go, one row split at a time, to the problem whose workgroups have the most rows
  int nb_n[PJ_MAX_PROBLEMS];
  int64_t n_mt[PJ_MAX_PROBLEMS], ms[PJ_MAX_PROBLEMS];
  int used = 0;
  for (int k = 0; k < n_problems; ++k) {
    nb_n[k] = args[k].ncols / PJ_BN;
    n_mt[k] = (args[k].M + PJ_BM * PJ_WAVES - 1) / (PJ_BM * PJ_WAVES);  // 128-node groups
    ms[k] = std::max<int64_t>(1, std::min<int64_t>((int64_t)(budget * work[k] / total / nb_n[k]), n_mt[k]));
    used += (int)(nb_n[k] * ms[k]);
  }
  for (;;) {
    int best = -1;
    double best_rows = 0;
    for (int k = 0; k < n_problems; ++k) {
      const double rows = (double)args[k].M / ms[k];
      if (ms[k] < n_mt[k] && used + nb_n[k] <= budget && rows > best_rows) {
        best = k;
        best_rows = rows;
      }
    }
    if (best < 0) break;
    ms[best] += 1;
    used += nb_n[best];
  }
  B.wg_off[0] = 0;
  for (int k = 0; k < PJ_MAX_PROBLEMS; ++k) {
    B.a[k] = args[k < n_problems ? k : 0];
    B.m_splits[k] = k < n_problems ? (int)ms[k] : 1;
    B.wg_off[k + 1] = B.wg_off[k] + (k < n_problems ? (int)(nb_n[k] * ms[k]) : 0);
  }
  const int n_wg = B.wg_off[PJ_MAX_PROBLEMS];
  hipStream_t s = (hipStream_t)stream;
  if (k2 != 0 && gemm_mode() == GGNN_GEMM_BF16X6) return ggnn_project_x6(B, n_wg, s);
  if (k2 == 0) hipLaunchKernelGGL((project_kernel<0>), dim3(n_wg), dim3(PJ_WAVES * 64), 0, s, B);
  else hipLaunchKernelGGL((project_kernel<96>), dim3(n_wg), dim3(PJ_WAVES * 64), 0, s, B);
  return launch_status();
}

extern "C" int ggnn_project(const float* X, int64_t ldx, int F, const float* H, int64_t ldh,
                            int k2, const float* Wp, const float* bias, int64_t M, int ncols,
                            float* out, int64_t ldo, ggnn_stream_t stream) {
  const ggnn_project_args a = {X, H, Wp, bias, out, ldx, ldh, M, ldo, F, k2, ncols, 0};
  return ggnn_project_batch(&a, 1, stream);
}
