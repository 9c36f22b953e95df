// Decoder projection GEMM (K = F + 96) on the bf16 matrix cores with fp32-equivalent numerics:
//   out[M, ncols] = [X[:, :F] | H] . Wp^T + bias
// Same contract and launch geometry as project_kernel (project.hip); the arithmetic is the exact
// 3 x bf16 split of common.h (6 products per k-step, fp32 accumulate), which needs 4 k-steps of
// v_mfma_f32_16x16x32_bf16 x 6 = 384 matrix-pipe cycles per 16x16 output tile instead of
// 26 x 32 = 832 for v_mfma_f32_16x16x4_f32.  With the MFMA time more than halved the kernel sits
// at the HBM write roofline of its 215 MB output.
//
// Layout of a k-step: the reduction index is re-ordered to [H 0..95 | X 0..FP-1 | zeros] so
// that k-steps 0..2 are pure hidden state and k-step 3 holds the <= 12 features.
//   * WEIGHTS (A operand): split once per workgroup in the prologue into three bf16 planes
//     [96 columns][128 k] in LDS (72 KB), 256-byte rows with the 16-byte slot XOR-swizzled by
//     (row & 15): the ds_read_b128 fragment reads (lane: row l&15, k = 32 ks + 8 (l>>4) ..+8)
//     are bank-conflict-free.
//   * NODES (B operand): every wave streams its own 16-node tiles: coalesced 16-byte row loads
//     (in flight during the previous sweep) -> a wave-private fp32 LDS stage -> per-lane fragment
//     reads (node l&15, the 8 consecutive k = 32 ks + 8 (l>>4)..) -> split into three planes in
//     registers, where they stay for the whole 96-column sweep.  No workgroup barrier in the loop.
#include "common.h"
#include "project_batch.h"

namespace ggnn {

constexpr int PX_BM = 16, PX_BN = 96, PX_WAVES = 12, PX_KS = 4;
constexpr int PX_LD = 112;  // floats per staged node row (96 + 16); 12 stages + weights = 160 128 B of LDS

struct ProjectX6Lds {
  u32x4 w[3][PX_BN][16];
  f32x4 b[PX_BN / 4];
  int next;  // tile queue of the workgroup: waves take the next 16-node tile when free
  __attribute__((aligned(16))) float x[PX_WAVES][PX_BM * PX_LD];
};

// NPROD = 6: the exact split (fp32-equivalent); NPROD = 1 (GGNN_PRECISION_BF16): only the leading bf16 piece of both
// operands, one product per k-step -- bf16 arithmetic with fp32 accumulation, as autocast defines a linear;
// NPROD = 3 (GGNN_PRECISION_F16X2): the fused cells' arithmetic -- two fp16 pieces per operand, hi hi into the main
// accumulator, hi lo' + lo' hi into a cross accumulator folded in with 2^-11 behind the four k-steps (common.h):
// 5e-8 of sum |x||w| against an fp64 product for operands below 65504, at half the matrix-pipe cycles of the six-product
// split.  At the chip's clock under this load (~1.55 GHz) the six-product sweep is 17.6 us of matrix pipe per model at the
// 10k-grain graph, which is what kept the kernel at 26 us; with three products its 61.7 MB of stores are the bound.
template <int FP, int NPROD>
__device__ __forceinline__ void project_x6_body(const ggnn_project_args& A, int blk, int m_splits, ProjectX6Lds& L) {
  constexpr int KP = FP + 96;  // row length of Wp: [X(FP) | H(96)]
  auto& s_w = L.w;
  auto& s_b = L.b;
  int& s_next = L.next;
  auto& s_x = L.x;
  const float* __restrict__ X = A.X;
  const float* __restrict__ H = A.H;
  const float* __restrict__ Wp = A.Wp;
  const float* __restrict__ bias = A.bias;
  const int64_t ldx = A.ldx, ldh = A.ldh, M = A.M;
  const int F = A.F, ncols = A.ncols;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb_n = ncols / PX_BN;
  const int bn = blk % nb_n, ms = blk / nb_n;
  const int n0 = bn * PX_BN;
  // GGNN_OUT_BLOCK_MAJOR (ggnn.h): the 96 columns this workgroup produces are a contiguous [M, 96] block of their own -- every
  // tile's 6 KB leave as one contiguous run instead of sixteen 384-byte pieces 4 * ldo bytes apart
  const bool block_major = (A.precision & GGNN_OUT_BLOCK_MAJOR) != 0;
  float* __restrict__ out = block_major ? A.out + (int64_t)bn * M * PX_BN : A.out;
  const int64_t ldo = block_major ? PX_BN : A.ldo;
  const int n0_out = block_major ? 0 : n0;

  // ---- prologue: split the weight tile into its three planes ----
  // All eight loads of a lane FIRST, unconditional (the k slots behind the row's end clamp to its last pair and are zeroed by a
  // select), then the splits: under `if (k < 96 + FP)` -- rounds 2-5 -- the compiler kept every load inside its branch with an
  // s_waitcnt vmcnt(0) behind it: eight dependent memory round trips at the head of every workgroup of a 22 us kernel.
  constexpr int PX_PRO = PX_BN * 64 / (PX_WAVES * 64);
  f32x2_ wab[PX_PRO];
#pragma unroll
  for (int it = 0; it < PX_PRO; ++it) {
    const int idx = tid + it * PX_WAVES * 64;
    const int r = idx >> 6, k = min(2 * (idx & 63), 96 + FP - 2);  // k, k+1 never straddle a segment (96, FP even)
    wab[it] = *reinterpret_cast<const f32x2_*>(Wp + (int64_t)(n0 + r) * KP + (k < 96 ? FP + k : k - 96));
  }
#pragma unroll
  for (int it = 0; it < PX_PRO; ++it) {
    const int idx = tid + it * PX_WAVES * 64;
    const int r = idx >> 6, kp = idx & 63, k = 2 * kp;
    const float a = k < 96 + FP ? wab[it][0] : 0.f, b = k < 96 + FP ? wab[it][1] : 0.f;
    uint32_t p0, p1, p2 = 0u;
    if constexpr (NPROD == 3) split_f16x2(a, b, p0, p1);
    else split_bf16x3(a, b, p0, p1, p2);
    const int slot = (kp >> 2) ^ (r & 15), d = kp & 3;
    reinterpret_cast<uint32_t*>(&s_w[0][r][slot])[d] = p0;
    reinterpret_cast<uint32_t*>(&s_w[1][r][slot])[d] = p1;
    if constexpr (NPROD != 3) reinterpret_cast<uint32_t*>(&s_w[2][r][slot])[d] = p2;
  }
  if (tid < PX_BN / 4) s_b[tid] = *reinterpret_cast<const f32x4*>(bias + n0 + 4 * tid);
  if (tid == 0) s_next = PX_WAVES;
  __syncthreads();  // the only workgroup barrier

  const int64_t n_mt = (M + PX_BM - 1) / PX_BM;
  const int64_t per = (n_mt + m_splits - 1) / m_splits;
  const int64_t mt_lo = ms * per, mt_hi = min(n_mt, (ms + 1) * per);
  if (mt_lo + wave >= mt_hi) return;
  auto grab = [&]() {  // wave-uniform ticket
    int v = 0;
    if (lane == 0) v = atomicAdd(&s_next, 1);
    return (int64_t)__builtin_amdgcn_readfirstlane(v);
  };

  // ---- node tiles: coalesced row loads -> wave-private LDS stage -> fragment reads ----
  // (Fragment-shaped global loads -- 16 rows x 16 B per quarter-wave -- touch 64 cache lines
  // per instruction and made the L1 tag pipe the bottleneck of the whole kernel.)
  const int lr = lane & 15, kq = lane >> 4;
  float* sx = s_x[wave];
  constexpr int NH = (PX_BM * 24) / 64;        // 16-byte hidden pieces per lane per tile (6)
  constexpr int NX = (PX_BM * FP) / 64;        // feature floats per lane per tile (1..3)
  static_assert((PX_BM * FP) % 64 == 0, "tile / lane mismatch");
  f32x4 gh[NH];
  float gx[NX];
  // Addressing: a tile's first row is wave-uniform -- min(16 mt, M - 16), so a ragged last tile
  // slides back over rows the previous tile also produced (identical values, benign duplicate
  // stores) -- and every lane adds a constant 32-bit offset: no per-tile vector address math.
  // (M < 16: one tile at row 0 whose lanes clamp their constant row to M - 1.)
  const int64_t m_last = max(M - PX_BM, (int64_t)0);
  int oh[NH], ox[NX];
#pragma unroll
  for (int it = 0; it < NH; ++it) {
    const int idx = lane + it * 64, r = idx / 24, c4 = idx - r * 24;
    oh[it] = (int)min((int64_t)r, M - 1) * (int)ldh + 4 * c4;
  }
#pragma unroll
  for (int it = 0; it < NX; ++it) {
    const int idx = lane + it * 64, r = idx / FP, k = min(idx - r * FP, F - 1);
    ox[it] = (int)min((int64_t)r, M - 1) * (int)ldx + k;
  }
  const int oo = (int)min((int64_t)lr, M - 1) * (int)ldo + n0_out + 4 * kq;
  auto load_tile = [&](int64_t mt) {  // unconditional: nothing here uses a loaded value
    const int64_t m0 = min(mt * PX_BM, m_last);
    const float* hb = H + m0 * ldh;
    const float* xb0 = X + m0 * ldx;
#pragma unroll
    for (int it = 0; it < NH; ++it) gh[it] = *reinterpret_cast<const f32x4*>(hb + oh[it]);
#pragma unroll
    for (int it = 0; it < NX; ++it) gx[it] = xb0[ox[it]];
  };
  auto stage_tile = [&]() {  // columns [96 + F, PX_LD) of the stage stay zero for the whole kernel
#pragma unroll
    for (int it = 0; it < NH; ++it) {
      const int idx = lane + it * 64, r = idx / 24, c4 = idx - r * 24;
      *reinterpret_cast<f32x4*>(&sx[r * PX_LD + 4 * c4]) = gh[it];
    }
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int idx = lane + it * 64, r = idx / FP, k = idx - r * FP;
      sx[r * PX_LD + 96 + k] = k < F ? gx[it] : 0.f;  // a select, not a branch: keeps the wait exact
    }
  };
  u32x4 xb[PX_KS][3];
  auto split_tile = [&]() {
#pragma unroll
    for (int ks = 0; ks < PX_KS; ++ks) {
      // k-step 3 holds the features in k = 96 .. 96+F-1 <= 107: quarter-waves 2, 3 see zeros
      const float* src = &sx[lr * PX_LD + (ks < 3 ? 32 * ks + 8 * kq : 96 + 8 * (kq & 1))];
      f32x4 v[2] = {*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4)};
      if (ks == 3 && kq >= 2) v[0] = v[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint32_t p0, p1, p2 = 0u;
        if constexpr (NPROD == 3) split_f16x2(v[d >> 1][2 * (d & 1)], v[d >> 1][2 * (d & 1) + 1], p0, p1);
        else split_bf16x3(v[d >> 1][2 * (d & 1)], v[d >> 1][2 * (d & 1) + 1], p0, p1, p2);
        xb[ks][0][d] = p0;
        xb[ks][1][d] = p1;
        xb[ks][2][d] = p2;
      }
    }
  };
  for (int i = lane; i < PX_BM * (PX_LD - 96); i += 64) {
    const int r = i / (PX_LD - 96);
    sx[r * PX_LD + 96 + (i - r * (PX_LD - 96))] = 0.f;
  }
  // waves w and w + 4 share a SIMD: a head start for one of them keeps their split / store
  // phases out of step so the matrix pipe always has a sweep to run (speed only)
  if (wave >= 8) __builtin_amdgcn_s_sleep(32);
  else if (wave >= 4) __builtin_amdgcn_s_sleep(16);
  const u32x4* pw = &s_w[0][lr][0];
  int64_t mt = mt_lo + wave;
  load_tile(mt);
  stage_tile();
  for (;;) {
    split_tile();  // LDS stage -> register planes
    // the next tile's rows are in flight during the sweep and are written to the stage right
    // after it: that wait sits in straight-line code behind the sweep's stores, so it is an exact
    // vmcnt(#stores) and the stores keep draining into the next tile
    __builtin_amdgcn_sched_barrier(0);
    const int64_t mt_next = mt_lo + grab();
    const bool has_next = mt_next < mt_hi;
    load_tile(has_next ? mt_next : mt);  // the last iteration re-reads a resident tile
    __builtin_amdgcn_sched_barrier(0);

    // ---- sweep: 6 column tiles x 4 k-steps, one flat software pipeline.  The three weight
    // fragments of step s+1 are read from LDS between the six MFMAs of step s; the accumulator
    // starts from the bias; a finished column tile is stored while the next one is swept. ----
    // no store predicate: see "Addressing" above
    float* orow = out + min(mt * PX_BM, m_last) * ldo + oo;
    constexpr int NSTEP = (PX_BN / 16) * PX_KS;
    constexpr int NPL = NPROD == 3 ? 2 : 3;   // operand planes in use
    u32x4 wf[2][3];
    f32x4 acc = s_b[kq];
    [[maybe_unused]] f32x4 cross = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NPL; ++p) wf[0][p] = pw[p * PX_BN * 16 + (kq ^ lr)];
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int ct = st / PX_KS, ks = st % PX_KS, cur = st & 1, nxt = cur ^ 1;
      f32x4 acc_next = acc;
      if (st + 1 < NSTEP) {
        const int ct1 = (st + 1) / PX_KS, ks1 = (st + 1) % PX_KS;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          wf[nxt][p] = pw[(p * PX_BN + ct1 * 16) * 16 + ((4 * ks1 + kq) ^ lr)];
        if (ks1 == 0) acc_next = s_b[ct1 * 4 + kq];
      }
      if constexpr (NPROD == 6) {
        acc = mfma_x6(wf[cur], xb[ks], acc);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      } else if constexpr (NPROD == 3) {
        cross = mfma_f16(wf[cur][0], xb[ks][1], cross);
        cross = mfma_f16(wf[cur][1], xb[ks][0], cross);
        acc = mfma_f16(wf[cur][0], xb[ks][0], acc);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      } else {
        acc = mfma_bf16(wf[cur][0], xb[ks][0], acc);
      }
      if (ks == PX_KS - 1) {
        if constexpr (NPROD == 3) {
          acc += cross * (1.0f / F16X2_SCALE);
          cross = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<f32x4*>(orow + ct * 16) = acc;
        acc = acc_next;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_tile();
    if (!has_next) break;
    mt = mt_next;
  }
}

__global__ __launch_bounds__(PX_WAVES * 64, 1) void project_x6_kernel(const ProjectBatch B) {
  __shared__ ProjectX6Lds lds;
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const ggnn_project_args& A = B.a[k];
  // The column tiles of one row split are consecutive logical blocks; dealt to the XCDs in contiguous ranges
  // (xcd_remap) they run side by side behind ONE L2, which then reads their node rows from memory once instead of once
  // per column tile (speed only).
  const int blk = xcd_remap((int)blockIdx.x - B.wg_off[k], B.wg_off[k + 1] - B.wg_off[k]), Fp = (A.F + 3) & ~3;
  const int precision = A.precision & ~GGNN_OUT_BLOCK_MAJOR;
  if (precision == GGNN_PRECISION_F16X2) {
    if (Fp == 4) project_x6_body<4, 3>(A, blk, B.m_splits[k], lds);
    else if (Fp == 8) project_x6_body<8, 3>(A, blk, B.m_splits[k], lds);
    else project_x6_body<12, 3>(A, blk, B.m_splits[k], lds);
    return;
  }
  if (precision == GGNN_PRECISION_BF16) {
    if (Fp == 4) project_x6_body<4, 1>(A, blk, B.m_splits[k], lds);
    else if (Fp == 8) project_x6_body<8, 1>(A, blk, B.m_splits[k], lds);
    else project_x6_body<12, 1>(A, blk, B.m_splits[k], lds);
    return;
  }
  if (Fp == 4) project_x6_body<4, 6>(A, blk, B.m_splits[k], lds);
  else if (Fp == 8) project_x6_body<8, 6>(A, blk, B.m_splits[k], lds);
  else project_x6_body<12, 6>(A, blk, B.m_splits[k], lds);
}

}  // namespace ggnn

// Called by ggnn_project_batch (project.hip) for k2 == 96 unless GGNN_GEMM=fp32.
int ggnn_project_x6(const ggnn::ProjectBatch& B, int n_wg, hipStream_t s) {
  using namespace ggnn;
  hipLaunchKernelGGL(project_x6_kernel, dim3((unsigned)n_wg), dim3(PX_WAVES * 64), 0, s, B);
  return launch_status();
}
