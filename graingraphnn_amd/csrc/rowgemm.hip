// Training path: the three dense products of a HeteroPGCLSTM cell that round 2 left to the BLAS
// library -- the gate GEMM z_g = agg_g W2_g^T (forward), its input gradient g_agg_g = g_z_g W2_g and the hidden-state
// gradient g_h += gP Wp[:, h] (training.py) -- as one hand-written kernel: C[b] = A[b] . W[b]^T (+ C_in) for a TALL
// A [M, K] (M = nodes: 10^4 .. 10^5, K = 96 .. 2112) and a SMALL W [n_out <= 224, K].
//
// Same machinery as the fused cells (cell_common.h): a wave owns ONE 16-row tile of A for the whole reduction and
// keeps its 16 x n_out output tile in registers; the weights are k-step slices of pre-split planes ([column tile][plane]
// [64 lanes][8 halfs], made from the fp32 parameters by ggnn_rowgemm_pack -- by the call itself or, for the five
// products of a training cell, by one launch ahead of them: they change with every optimizer step).  Two kernels:
//   * rowgemm_resident_kernel (K <= 128, or <= 256 with 96 outputs: eight of the ten products of a training step): the
//     planes of one batch entry stay in LDS for the whole launch, grid = (row chunks, batch), no barrier behind the prologue;
//   * rowgemm_kernel (longer reductions: g_h, K = 1248 / 2112): the slices stream through a double-buffered LDS region
//     shared by the workgroup's eight waves (LDS-DMA one group of slices ahead, one barrier per group).
// In both the rows arrive as B-fragment-shaped 32-byte pieces (the four k-groups of a row = one 128-byte line), a whole
// tile or group ahead in registers.
// Arithmetic: fp32 mode = two fp16 pieces / three products per operand pair (common.h: 5e-8 of sum |a||w| against
// fp64, an fp32 fma chain: 2e-7); GGNN_PRECISION_BF16 (torch.autocast(bfloat16), BASELINE config 5) = ONE bf16 plane,
// one product, fp32 accumulation: a third of the matrix work and half the weight bytes.
#include <algorithm>

#include "common.h"
#include "cell_common.h"

namespace ggnn {

// ---- per-row power-of-two scaling of the streamed operand (fp32-equivalent mode) ----
// The two-piece fp16 split has an ABSOLUTE resolution of 2^-36 (common.h): fine for activations, not for the gradient rows
// the backward products stream (1e-4 .. 1e-10 per node: g_z, gP).  Every row is therefore multiplied by 2^(141 - e), e =
// biased exponent of the row's largest magnitude SO FAR (exact; largest element in [2^14, 2^15)), and the accumulators
// carry the same factor: a row element 2^-k below the row's largest keeps a relative resolution of max(2^-22, 2^(k - 51)),
// whatever the row's magnitude, and no finite fp32 row saturates.  A row holding an inf or a NaN poisons its outputs
// (scale = NaN), as an fp32 product would: gradient overflow checks keep working.
struct RowScale {
  int e = 0;                 // biased exponent the scale belongs to (0: nothing but zeros seen)
  float s = 0.f, inv = 0.f;  // 2^(141 - e) and its reciprocal
};
// the largest magnitude (as bits: dc_track) over the four k-group lanes that hold pieces of this lane's row
__device__ __forceinline__ uint32_t rg_row_amax(uint32_t am, int lane) {
  am = max(am, (uint32_t)__builtin_amdgcn_ds_swizzle((int)am, 0x401F));               // lane ^ 16 (bit mode: xor 0x10)
  am = max(am, (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)am));    // lane ^ 32
  return am;
}
// Adopt `am` into the row's scale; returns the exact factor (a power of two, or 1) the accumulators have to be multiplied
// with because the scale went down.
__device__ __forceinline__ float rg_rescale(RowScale& R, uint32_t am) {
  const int e = (int)(am >> 23);
  if (e <= R.e && R.s != 0.f) return 1.0f;
  const bool had = R.s != 0.f;
  const int es_old = (int)(__float_as_uint(R.s) >> 23);   // (a finite power of two when `had`: a NaN scale returns above)
  float f = 1.0f;
  if (e == 255) {
    if (had) f = e - R.e > 126 ? 0.f : __uint_as_float((uint32_t)(127 - (e - R.e)) << 23);
    R.s = R.inv = __builtin_nanf("");
  } else {
    // the factor follows the CLAMPED scale exponents: for rows at or below 2^-112 the scale stays at 2^126 while the
    // row's exponent still rises, and the accumulators must then stay as they are
    const int es = min(max(268 - e, 1), 253);
    if (had) f = es_old - es > 126 ? 0.f : __uint_as_float((uint32_t)(127 - (es_old - es)) << 23);
    R.s = __uint_as_float((uint32_t)es << 23);
    R.inv = __uint_as_float((uint32_t)(254 - es) << 23);
  }
  R.e = e;
  return f;
}
__device__ __forceinline__ void rg_track8(uint32_t& am, const f32x4 a, const f32x4 b) {
  dc_track(am, a[0], a[1]);
  dc_track(am, a[2], a[3]);
  dc_track(am, b[0], b[1]);
  dc_track(am, b[2], b[3]);
}


constexpr int RG_WAVES = 8;
constexpr int RG_MAX_CT = 14;                 // n_out <= 224

// lane l = 16 kq + m of piece (b, ks, ct, plane): W[b][16 ct + m][32 ks + 8 kq .. +7]
// ... for several products in one launch (ggnn_rowgemm_pack): workgroup -> product through the offsets
struct RowGemmPackBatch {
  ggnn_rowgemm_args a[GGNN_ROWGEMM_MAX_PACK];
  int blk_off[GGNN_ROWGEMM_MAX_PACK + 1];
  int nks[GGNN_ROWGEMM_MAX_PACK], nct[GGNN_ROWGEMM_MAX_PACK];
  int n;
};
__device__ __forceinline__ void rowgemm_pack_item(const ggnn_rowgemm_args& A, u32x4* __restrict__ out, int nks, int nct,
                                                  int64_t i, bool bf16) {
  if (i >= (int64_t)A.batch * nks * nct * 64) return;
  const int lane = (int)(i & 63);
  int64_t r = i >> 6;
  const int ct = (int)(r % nct);
  r /= nct;
  const int ks = (int)(r % nks), b = (int)(r / nks);
  const int n = 16 * ct + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
  const float* w = A.w + (int64_t)b * A.w_bstride + (int64_t)n * A.w_nstride;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (n < A.n_out && k0 + j < A.K) ? w[(int64_t)(k0 + j) * A.w_kstride] : 0.f;
  const int P = bf16 ? 1 : 2;
  u32x4* dst = out + ((((int64_t)b * nks + ks) * nct + ct) * P) * 64 + lane;
  if (bf16) {
    u32x4 q;
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] = pack_bf16(v[2 * e], v[2 * e + 1]);
    dst[0] = q;
  } else {
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t q0, q1;
      split_f16x2(v[2 * e], v[2 * e + 1], q0, q1);
      hi[e] = q0;
      lo[e] = q1;
    }
    dst[0] = hi;
    dst[64] = lo;
  }
}
__global__ __launch_bounds__(256) void rowgemm_pack_batch_kernel(const RowGemmPackBatch B) {
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.blk_off[k + 1]) ++k;
  const ggnn_rowgemm_args& A = B.a[k];
  rowgemm_pack_item(A, reinterpret_cast<u32x4*>(A.workspace), B.nks[k], B.nct[k],
                    (int64_t)(blockIdx.x - B.blk_off[k]) * 256 + threadIdx.x, A.precision == GGNN_PRECISION_BF16);
}

// Weight slices (one k-step each) travel in GROUPS of GS: one barrier and one counted wait per group, and the rows of
// the whole NEXT group (GS x 32 bytes per lane) are requested at the top of the current one -- ~100 KB of rows in
// flight per compute unit: the kernel streams A from HBM once, and a request per k-step (the first version) had
// every k-step wait out a full memory round trip (125 us for the 169 MB of the joints' g_h; the library: 102).
// One or TWO products per launch (ggnn_rowgemm_pair): a wave walks the whole reduction of its 16 rows, so the grid is M / 128
// workgroups -- 79 and 157 for the grains' and the joints' hidden-state gradient on 256 compute units, one after the other;
// side by side in one grid they take the time of the longer one.
struct RowGemmPair {
  ggnn_rowgemm_args a[2];
  const u32x4* planes[2];
  int nks[2];
  int n_wg0;   // workgroups of the first product (the second one's follow)
};
template <int NCT, bool BF16>
__global__ __launch_bounds__(RG_WAVES * 64) void rowgemm_kernel(const RowGemmPair B) {
  const int which = (int)blockIdx.x >= B.n_wg0 ? 1 : 0;
  const ggnn_rowgemm_args& A = B.a[which];
  const u32x4* __restrict__ planes = B.planes[which];
  const int nks = B.nks[which];
  const int64_t wg = (int64_t)blockIdx.x - (which ? B.n_wg0 : 0);
  constexpr int P = BF16 ? 1 : 2;
  constexpr int SLICE = NCT * P * 1024;
  // k-steps per group: 4 (n_out <= 128); n_out = 224: 2 in bf16, 1 in fp32 mode (112 accumulator registers: no room for
  // more rows in flight; its reductions are 3 k-steps long)
  constexpr int GS = NCT > 8 ? (BF16 ? 2 : 1) : 4;
  static_assert(2 * GS * SLICE <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * GS * SLICE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  const int64_t M = A.M;
  // a ragged last tile slides back over rows the previous tile also produces (identical duplicate stores); surplus
  // waves repeat the last tile: every wave runs the whole program (the group barriers need no special case)
  const int64_t row0 = std::max<int64_t>(0, std::min<int64_t>((wg * RG_WAVES + wave) * 16, M - 16));
  const int64_t row = std::min<int64_t>(row0 + lr, M - 1);

  const unsigned char* __restrict__ wsrc = reinterpret_cast<const unsigned char*>(planes) + lane * 16;
  const uint32_t slice_lds = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)));
  const int gpb = (nks + GS - 1) / GS;             // groups per batch entry (the last one may be short)
  const int n_groups = A.batch * gpb;
  auto group_steps = [&](int q) { return std::min(GS, nks - (q % gpb) * GS); };
  auto dma_group = [&](int q) {
    if (q >= n_groups) return;
    const int b = q / gpb, ks0 = (q - b * gpb) * GS;
    const unsigned char* src = wsrc + (size_t)(b * nks + ks0) * SLICE;       // the slices of a batch entry are contiguous
    const uint32_t dst = slice_lds + (q & 1) * (GS * SLICE);
    const int np = group_steps(q) * NCT * P;
    for (int p = wave; p < np; p += RG_WAVES) dc_dma16(src + p * 1024, dst + p * 1024);
  };
  f32x4 rows[2][GS][2];                            // [parity][k-step of the group][half]: this lane's 32-byte row pieces
  auto load_rows = [&](int q, f32x4 (&r)[GS][2]) __attribute__((always_inline)) {
    const int qq = std::min(q, n_groups - 1);      // (past the end: a harmless repeat of the last group's loads)
    const int b = qq / gpb, ks0 = (qq - b * gpb) * GS;
    const float* __restrict__ arow = A.a + (int64_t)b * A.a_bstride + row * A.lda + 8 * kq;
#pragma unroll
    for (int i = 0; i < GS; ++i) {
      const int ks = std::min(ks0 + i, nks - 1);   // (a short last group repeats its last k-step: loaded, not used)
      r[i][0] = ld16f(arow + 32 * ks);
      r[i][1] = ld16f(arow + 32 * ks + 4);
    }
  };
  dma_group(0);
  load_rows(0, rows[0]);

  int q = 0;
  for (int b = 0; b < A.batch; ++b) {
    float* __restrict__ crow = A.c + (int64_t)b * A.c_bstride + row * A.ldc + 4 * kq;
    DcAcc acc[NCT];   // (bf16 mode: the cross terms stay zero and the compiler drops them)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct].zero();
    [[maybe_unused]] RowScale rs;   // fp32-equivalent mode: the accumulators hold rs.s x the product
    if (BF16 && A.c_in != nullptr) {
      const float* __restrict__ cin = A.c_in + (int64_t)b * A.c_bstride + row * A.ldc + 4 * kq;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
        if (16 * ct < A.n_out) acc[ct].m = ld16f(cin + 16 * ct);
    }
    for (int g = 0; g < gpb; ++g, ++q) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of group q and its rows have landed
      __syncthreads();                                   // ... everybody's have; nobody reads group q - 1 any more
      dma_group(q + 1);
      const u32x4* pw = reinterpret_cast<const u32x4*>(smem + (q & 1) * (GS * SLICE)) + lane;
      const int ns = group_steps(q);
      auto run_group = [&](f32x4 (&cur)[GS][2], f32x4 (&nxt)[GS][2]) __attribute__((always_inline)) {
        load_rows(q + 1, nxt);                           // the next group's rows: in flight during this group's MFMAs
        if constexpr (!BF16) {   // this group's share of the row: its largest magnitude may lower the row's scale
          uint32_t am = 0u;
#pragma unroll
          for (int i = 0; i < GS; ++i)
            if (i < ns) rg_track8(am, cur[i][0], cur[i][1]);
          const float f = rg_rescale(rs, rg_row_amax(am, lane));
          if (f != 1.0f) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
              acc[ct].m *= f;
              acc[ct].c *= f;
            }
          }
        }
#pragma unroll
        for (int i = 0; i < GS; ++i) {
          if (i < ns) {
            u32x4 xb[DC_PL];
            if constexpr (BF16) {
              xb[0] = (u32x4){pack_bf16(cur[i][0][0], cur[i][0][1]), pack_bf16(cur[i][0][2], cur[i][0][3]),
                              pack_bf16(cur[i][1][0], cur[i][1][1]), pack_bf16(cur[i][1][2], cur[i][1][3])};
            } else {
              dc_split(cur[i][0] * rs.s, cur[i][1] * rs.s, xb);
            }
            const u32x4* ps = pw + i * (SLICE / 16);
            if constexpr (BF16) {
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct) {
                acc[ct].m = mfma_bf16(ps[ct * 64], xb[0], acc[ct].m);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              }
            } else {
              dc_kstep<NCT>(ps, xb, acc);
            }
          }
        }
      };
      if (q & 1) run_group(rows[1], rows[0]);
      else run_group(rows[0], rows[1]);
    }
    if constexpr (BF16) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
        if (16 * ct < A.n_out) *reinterpret_cast<f32x4*>(crow + 16 * ct) = acc[ct].m;
    } else {
      const float* __restrict__ cin = A.c_in != nullptr ? A.c_in + (int64_t)b * A.c_bstride + row * A.ldc + 4 * kq : nullptr;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        if (16 * ct < A.n_out) {
          f32x4 v = acc[ct].value() * rs.inv;
          if (cin != nullptr) v += ld16f(cin + 16 * ct);
          *reinterpret_cast<f32x4*>(crow + 16 * ct) = v;
        }
      }
    }
  }
}

// Short reductions (K <= 128 .. 256: the gate GEMM and its input gradient -- eight of the ten products of a training
// step): the planes of ONE batch entry's W fit the LDS (K / 32 slices), so a workgroup fetches them once and its waves
// then stream row tiles with no barrier at all -- the next tile's rows (K / 32 x 32 bytes per lane) are requested before
// the current tile's MFMAs.  Grid = (row chunks, batch): the streamed kernel above walked the batch entries inside one
// workgroup per 128 rows, 79 / 157 workgroups on 256 compute units at the 10k-grain graph.
template <int NCT, int NKSM, bool BF16>
__global__ __launch_bounds__(RG_WAVES * 64) void rowgemm_resident_kernel(const ggnn_rowgemm_args A,
                                                                         const u32x4* __restrict__ planes, const int nks) {
  constexpr int P = BF16 ? 1 : 2;
  constexpr int SLICE = NCT * P * 1024;
  static_assert(NKSM * SLICE <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NKSM * SLICE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  const int64_t M = A.M, n_tiles = (M + 15) / 16, stride = (int64_t)gridDim.x * RG_WAVES;
  {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(planes) + (size_t)b * nks * SLICE + lane * 16;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)));
    const int np = nks * NCT * P;
    for (int p = wave; p < np; p += RG_WAVES) dc_dma16(src + p * 1024, dst + p * 1024);
  }
  const float* __restrict__ abase = A.a + (int64_t)b * A.a_bstride + 8 * kq;
  // a ragged last tile slides back over rows the previous tile also produces (identical duplicate stores)
  auto tile_row = [&](int64_t t) { return std::min<int64_t>(std::max<int64_t>(0, std::min<int64_t>(t * 16, M - 16)) + lr, M - 1); };
  auto load_rows = [&](int64_t t, f32x4 (&r)[NKSM][2]) __attribute__((always_inline)) {
    const float* __restrict__ arow = abase + tile_row(std::min(t, n_tiles - 1)) * A.lda;   // (past the end: a harmless repeat)
    // unconditional (clamped) loads: a load under `if` makes the compiler wait for EVERYTHING at the branch merge, the
    // next tile's rows included
#pragma unroll
    for (int i = 0; i < NKSM; ++i) {
      const int ic = std::min(i, nks - 1);
      r[i][0] = ld16f(arow + 32 * ic);
      r[i][1] = ld16f(arow + 32 * ic + 4);
    }
  };
  f32x4 rows[2][NKSM][2];
  int64_t t = (int64_t)blockIdx.x * RG_WAVES + wave;
  load_rows(t, rows[0]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // W[b] is in LDS; from here on the waves run free
  const u32x4* pw = reinterpret_cast<const u32x4*>(smem) + lane;
  auto run_tile = [&](f32x4 (&cur)[NKSM][2], f32x4 (&nxt)[NKSM][2]) __attribute__((always_inline)) {
    load_rows(t + stride, nxt);
    const int64_t row = tile_row(t);
    DcAcc acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct].zero();
    [[maybe_unused]] RowScale rs;   // the whole row is in registers: one scale (RowScale above)
    if constexpr (!BF16) {
      uint32_t am = 0u;
#pragma unroll
      for (int i = 0; i < NKSM; ++i)
        if (i < nks) rg_track8(am, cur[i][0], cur[i][1]);
      rg_rescale(rs, rg_row_amax(am, lane));
    }
#pragma unroll
    for (int i = 0; i < NKSM; ++i) {
      if (i < nks) {
        u32x4 xb[DC_PL];
        const u32x4* ps = pw + i * (SLICE / 16);
        if constexpr (BF16) {
          xb[0] = (u32x4){pack_bf16(cur[i][0][0], cur[i][0][1]), pack_bf16(cur[i][0][2], cur[i][0][3]),
                          pack_bf16(cur[i][1][0], cur[i][1][1]), pack_bf16(cur[i][1][2], cur[i][1][3])};
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            acc[ct].m = mfma_bf16(ps[ct * 64], xb[0], acc[ct].m);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        } else {
          dc_split(cur[i][0] * rs.s, cur[i][1] * rs.s, xb);
          dc_kstep<NCT>(ps, xb, acc);
        }
      }
    }
    // (n_out = 16 NCT on this path: no store under a branch)
    const int64_t coff = (int64_t)b * A.c_bstride + row * A.ldc + 4 * kq;
    f32x4 cin[NCT];
    if (A.c_in != nullptr) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) cin[ct] = ld16f(A.c_in + coff + 16 * ct);
    } else {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) cin[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
      *reinterpret_cast<f32x4*>(A.c + coff + 16 * ct) = (BF16 ? acc[ct].m : acc[ct].value() * rs.inv) + cin[ct];
  };
  while (t < n_tiles) {
    run_tile(rows[0], rows[1]);
    t += stride;
    if (t >= n_tiles) break;
    run_tile(rows[1], rows[0]);
    t += stride;
  }
}

template <int NCT, int NKSM>
static int rowgemm_resident_launch(const ggnn_rowgemm_args& A, const u32x4* planes, int nks, hipStream_t st) {
  // row tiles per wave so that the whole grid is resident at once (one workgroup per compute unit: the planes take most
  // of the LDS), the waves of a batch entry sharing its tiles evenly
  const int64_t n_tiles = (A.M + 15) / 16;
  const int64_t per_wave = std::max<int64_t>(1, (n_tiles * A.batch + RG_WAVES * 256 - 1) / (RG_WAVES * 256));
  const int64_t n_wg = (n_tiles + RG_WAVES * per_wave - 1) / (RG_WAVES * per_wave);
  if (n_wg >= INT32_MAX || A.batch > 65535) return GGNN_EINVAL;
  const dim3 grid((unsigned)n_wg, (unsigned)A.batch);
  if (A.precision == GGNN_PRECISION_BF16)
    hipLaunchKernelGGL((rowgemm_resident_kernel<NCT, NKSM, true>), grid, dim3(RG_WAVES * 64), 0, st, A, planes, nks);
  else
    hipLaunchKernelGGL((rowgemm_resident_kernel<NCT, NKSM, false>), grid, dim3(RG_WAVES * 64), 0, st, A, planes, nks);
  return launch_status();
}

// (the second product may be absent: A1 == nullptr)
template <int NCT>
static int rowgemm_launch(const ggnn_rowgemm_args& A0, const ggnn_rowgemm_args* A1, hipStream_t st) {
  RowGemmPair B;
  const int64_t n0 = (A0.M + 16 * RG_WAVES - 1) / (16 * RG_WAVES), n1 = A1 ? (A1->M + 16 * RG_WAVES - 1) / (16 * RG_WAVES) : 0;
  if (n0 + n1 >= INT32_MAX) return GGNN_EINVAL;
  B.a[0] = A0, B.planes[0] = reinterpret_cast<const u32x4*>(A0.workspace), B.nks[0] = A0.K / 32;
  B.a[1] = A1 ? *A1 : A0, B.planes[1] = reinterpret_cast<const u32x4*>(B.a[1].workspace), B.nks[1] = B.a[1].K / 32;
  B.n_wg0 = (int)n0;
  if (A0.precision == GGNN_PRECISION_BF16)
    hipLaunchKernelGGL((rowgemm_kernel<NCT, true>), dim3((unsigned)(n0 + n1)), dim3(RG_WAVES * 64), 0, st, B);
  else
    hipLaunchKernelGGL((rowgemm_kernel<NCT, false>), dim3((unsigned)(n0 + n1)), dim3(RG_WAVES * 64), 0, st, B);
  return launch_status();
}

}  // namespace ggnn

// column tiles per weight slice: the kernel is instantiated for 6, 8 and 14 (n_out = 96, 128, 224: the shapes of the
// training path); other widths run on the next larger one (their surplus column tiles are zero planes, never stored)
static int rowgemm_tiles(int n_out) {
  const int nct = (n_out + 15) / 16;
  return nct <= 6 ? 6 : (nct <= 8 ? 8 : 14);
}

extern "C" size_t ggnn_rowgemm_workspace_bytes(int32_t K, int32_t n_out, int32_t batch) {
  if (K <= 0 || n_out <= 0 || n_out > 16 * ggnn::RG_MAX_CT || batch <= 0) return 0;
  return (size_t)batch * ((size_t)(K + 31) / 32) * rowgemm_tiles(n_out) * 2 * 1024;
}

static int rowgemm_check_weights(const ggnn_rowgemm_args& A) {
  using namespace ggnn;
  if (!A.workspace || A.batch < 1 || !aligned16(A.workspace)) return GGNN_EINVAL;
  if (A.K <= 0 || (A.K & 31) || A.n_out <= 0 || (A.n_out & 15) || A.n_out > 16 * RG_MAX_CT) return GGNN_EINVAL;
  if (A.precision != 0 && A.precision != GGNN_PRECISION_BF16) return GGNN_EINVAL;
  if (A.workspace_bytes < ggnn_rowgemm_workspace_bytes(A.K, A.n_out, A.batch)) return GGNN_EINVAL;
  return GGNN_OK;
}

extern "C" int ggnn_rowgemm_pack(const ggnn_rowgemm_args* args, int n_products, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_products < 1 || n_products > GGNN_ROWGEMM_MAX_PACK) return GGNN_EINVAL;
  RowGemmPackBatch B;
  B.n = n_products;
  B.blk_off[0] = 0;
  for (int k = 0; k < GGNN_ROWGEMM_MAX_PACK; ++k) {
    B.a[k] = args[k < n_products ? k : 0];
    B.nks[k] = B.nct[k] = 0;
    if (k >= n_products) {
      B.blk_off[k + 1] = B.blk_off[k];
      continue;
    }
    const ggnn_rowgemm_args& A = B.a[k];
    if (!A.w || rowgemm_check_weights(A) != GGNN_OK) return GGNN_EINVAL;
    B.nks[k] = A.K / 32;
    B.nct[k] = rowgemm_tiles(A.n_out);
    const int64_t nb = ((int64_t)A.batch * B.nks[k] * B.nct[k] * 64 + 255) / 256;
    if (B.blk_off[k] + nb >= INT32_MAX) return GGNN_EINVAL;
    B.blk_off[k + 1] = B.blk_off[k] + (int)nb;
  }
  hipLaunchKernelGGL(rowgemm_pack_batch_kernel, dim3((unsigned)B.blk_off[GGNN_ROWGEMM_MAX_PACK]), dim3(256), 0,
                     (hipStream_t)stream, B);
  return launch_status();
}

extern "C" int ggnn_rowgemm(const ggnn_rowgemm_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  const ggnn_rowgemm_args& A = *args;
  if (!A.a || (!A.w && !A.prepacked) || !A.c || A.M <= 0 || rowgemm_check_weights(A) != GGNN_OK) return GGNN_EINVAL;
  if (A.lda < A.K || A.ldc < A.n_out || (A.lda & 3) || (A.ldc & 3) || (A.a_bstride & 3) || (A.c_bstride & 3)) return GGNN_EINVAL;
  if (!aligned16(A.a) || !aligned16(A.c) || (A.c_in && !aligned16(A.c_in))) return GGNN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // W[b]'s planes fit the LDS (and n_out is one of the instantiated widths)
  const bool resident = A.n_out == 16 * rowgemm_tiles(A.n_out) && A.K / 32 <= (A.n_out == 96 ? 8 : 4);
  if (A.precision == 0 && A.n_out > 128 && !resident) {
    if (A.prepacked) return GGNN_EINVAL;   // (the halves have plane images of their own)
    // fp32 mode wider than 8 column tiles: 14 x (main + cross) accumulators leave no registers for rows in flight
    // (82 spilled, slower than the library) -- two passes over column halves instead (A is read twice: the wide shape
    // is the gate GEMM's input gradient, whose reduction is 96 long)
    ggnn_rowgemm_args H = A;
    const int half = ((A.n_out / 16 + 1) / 2) * 16;
    H.n_out = half;
    int rc = ggnn_rowgemm(&H, stream);
    if (rc != GGNN_OK) return rc;
    H.n_out = A.n_out - half;
    H.w = A.w + (int64_t)half * A.w_nstride;
    H.c = A.c + half;
    H.c_in = A.c_in ? A.c_in + half : nullptr;
    return ggnn_rowgemm(&H, stream);
  }
  const int nks = A.K / 32, nct = rowgemm_tiles(A.n_out);
  u32x4* planes = reinterpret_cast<u32x4*>(A.workspace);
  if (!A.prepacked) {
    const int rc = ggnn_rowgemm_pack(&A, 1, stream);
    if (rc != GGNN_OK) return rc;
  }
  if (resident) {
    switch (nct) {
      case 6: return rowgemm_resident_launch<6, 8>(A, planes, nks, st);
      case 8: return rowgemm_resident_launch<8, 4>(A, planes, nks, st);
      case 14: return rowgemm_resident_launch<14, 4>(A, planes, nks, st);
      default: return GGNN_EINVAL;
    }
  }
  switch (nct) {
    case 6: return rowgemm_launch<6>(A, nullptr, st);
    case 8: return rowgemm_launch<8>(A, nullptr, st);
    case 14: return rowgemm_launch<14>(A, nullptr, st);
    default: return GGNN_EINVAL;
  }
}

static bool rowgemm_streams(const ggnn_rowgemm_args& A) {   // the streamed kernel in ONE pass (see ggnn_rowgemm)
  const bool resident = A.n_out == 16 * rowgemm_tiles(A.n_out) && A.K / 32 <= (A.n_out == 96 ? 8 : 4);
  return !resident && !(A.precision == 0 && A.n_out > 128);
}

extern "C" int ggnn_rowgemm_pair(const ggnn_rowgemm_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  for (int k = 0; k < 2; ++k) {
    const ggnn_rowgemm_args& A = args[k];
    if (!A.a || !A.c || A.M <= 0 || !A.prepacked || A.batch != 1 || rowgemm_check_weights(A) != GGNN_OK) return GGNN_EINVAL;
    if (A.lda < A.K || A.ldc < A.n_out || (A.lda & 3) || (A.ldc & 3)) return GGNN_EINVAL;
    if (!aligned16(A.a) || !aligned16(A.c) || (A.c_in && !aligned16(A.c_in)) || !rowgemm_streams(A)) return GGNN_EINVAL;
  }
  if (rowgemm_tiles(args[0].n_out) != rowgemm_tiles(args[1].n_out) || args[0].precision != args[1].precision) return GGNN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (rowgemm_tiles(args[0].n_out)) {
    case 6: return rowgemm_launch<6>(args[0], &args[1], st);
    case 8: return rowgemm_launch<8>(args[0], &args[1], st);
    case 14: return rowgemm_launch<14>(args[0], &args[1], st);
    default: return GGNN_EINVAL;
  }
}
