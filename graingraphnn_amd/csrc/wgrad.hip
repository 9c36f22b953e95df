// Weight-gradient GEMM of the training path:  C[b] = A[b]^T B[b],  A [K, M], B [K, Nc] with K = number of
// nodes (10^4..10^5) and a small M x Nc result (the gradient of a packed projection / gate weight matrix:
// train.py:158-166's loss.backward() reaching lin_* through the packed formulation).
// The BLAS library runs these shapes on a few dozen workgroups (no split over K): 150-290 us each at the
// 10k-grain graph, a quarter of the whole training step.  Here K is split over the chip:
//
//   * one wave owns a (16 TA) x 112 block of C over one K range and accumulates it in registers with
//     v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate);
//   * both operands of that MFMA are K-major (lane l supplies row k0 + l/16), so the fragments come straight from
//     the row-major operands -- no LDS, no transpose.  A lane loads TA consecutive floats of A and 4 + 2 + 1 of B
//     per row (dwordx4 / x2 / x1: 256-byte row pieces per 16 lanes), and component j of a load is the lane's
//     element of MFMA tile j: a "tile" is the strided column set {c0 + w i + j}, which the MFMA does not care about
//     and the store undoes;
//   * a ring of four 4-row groups: a group's registers are reloaded for 16 rows ahead as soon as its 7 TA MFMAs
//     have issued, so loads have ~1 us of matrix work to land;
//   * logical blocks are dealt to the XCDs in contiguous ranges (xcd_remap): the waves that share a K range -- and
//     so the rows of B -- sit behind one L2;
//   * every wave writes its partial block to partial[split]; the caller sums over the split axis (a fixed
//     decomposition: results are reproducible run to run).
// Lanes whose columns fall beyond M / Nc load a clamped address and store nothing.
// Measured and dropped: the four waves of a workgroup on four K ranges of one block, summed through LDS (a quarter
// of the partials for the caller to add): 176 instead of 102 us for the largest call -- the waves of a workgroup
// then read different rows, and the 1 KB row pieces they shared become four 256-byte pieces far apart.
#include "common.h"

namespace ggnn {

constexpr int WG_TB = 7;   // 16-column tiles of B per wave: 4 (dwordx4) + 2 (dwordx2) + 1 (dword)
constexpr int WG_NB = 16 * WG_TB;
constexpr int WG_U = 4;    // 4-row MFMA groups per pipeline stage
#ifndef WG_CFG_WAVES
#define WG_CFG_WAVES 1024
#endif
constexpr int WG_WAVES = WG_CFG_WAVES;  // waves a launch aims for (overridable for measurements: make VARIANT=..)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int W> struct wg_vec;
template <> struct wg_vec<4> { typedef f32x4 type; };
template <> struct wg_vec<2> { typedef f32x2 type; };

template <int TA>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const ggnn_wgrad_args W, int n_mt, int n_nb, int64_t chunk) {
  typedef typename wg_vec<TA>::type avec;
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lq = lane >> 4;
  int64_t w = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t total = (int64_t)W.batch * n_nb * W.n_split * n_mt;
  if (w >= total) return;
  const int mt = (int)(w % n_mt);
  w /= n_mt;
  const int nb = (int)(w % n_nb);
  w /= n_nb;
  const int b = (int)(w % W.batch), s = (int)(w / W.batch);
  const int m0 = mt * 16 * TA, n0 = nb * WG_NB;
  const int64_t k_begin = s * chunk, k_end = min(W.K, k_begin + chunk);

  // this lane's columns (clamped into the matrix; clamped lanes do not store)
  const int ca = min(m0 + TA * li, W.M - TA);
  const int cb4 = min(n0 + 4 * li, W.Nc - 4), cb2 = min(n0 + 64 + 2 * li, W.Nc - 2), cb1 = min(n0 + 96 + li, W.Nc - 1);
  const float* A = W.a + (int64_t)b * W.a_bstride + (int64_t)lq * W.lda + ca;
  // this lane's three column pieces of B; with an inserted operand (b_ins, ABI 25: boundaries are multiples of 4, so a piece
  // lies in one of the two matrices) each piece has its own base and row pitch
  const float *B4, *B2, *B1;
  int64_t ld4, ld2, ld1;
  auto piece = [&](int c, const float*& p, int64_t& ld) {
    const bool in = W.b_ins != nullptr && c >= W.ins_off && c < W.ins_off + W.ins_w;
    const int cc = in ? c - W.ins_off : (W.b_ins != nullptr && c >= W.ins_off + W.ins_w ? c - W.ins_w : c);
    ld = in ? W.ld_ins : W.ldb;
    p = (in ? W.b_ins : W.b + (int64_t)b * W.b_bstride) + (int64_t)lq * ld + cc;
  };
  piece(cb4, B4, ld4);
  piece(cb2, B2, ld2);
  piece(cb1, B1, ld1);

  f32x4 acc[TA][WG_TB];
#pragma unroll
  for (int t = 0; t < TA; ++t)
#pragma unroll
    for (int u = 0; u < WG_TB; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A ring of WG_U row groups: group j's registers are reloaded (rows 4 WG_U further on) right after its MFMAs
  // have issued, so every load has the MFMAs of the other WG_U - 1 groups (~1 us) to land.  The loads are inline
  // asm with counted waits: hipcc's own waits drain every outstanding load at the top of the loop (it cannot
  // carry the count over the back edge), which exposes a full memory latency per 16 rows.
  avec ra[WG_U];
  f32x4 rb4[WG_U];
  f32x2 rb2[WG_U];
  float rb1[WG_U];
  auto load = [&](int j, int64_t k) {  // the 4 rows from k (all below k_end): 4 loads
    const float* pa = A + k * W.lda;
    if constexpr (TA == 4)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[j]) : "v"(pa));
    else
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ra[j]) : "v"(pa));
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb4[j]) : "v"(B4 + k * ld4));
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rb2[j]) : "v"(B2 + k * ld2));
    asm volatile("global_load_dword %0, %1, off" : "=v"(rb1[j]) : "v"(B1 + k * ld1));
  };
  auto compute = [&](int j) {
#pragma unroll
    for (int t = 0; t < TA; ++t) {
      const float a = ra[j][t];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rb4[j][u], acc[t][u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
        acc[t][4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rb2[j][u], acc[t][4 + u], 0, 0, 0);
      acc[t][6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rb1[j], acc[t][6], 0, 0, 0);
    }
  };
#define WG_WAIT(n)                                    \
  do {                                                \
    asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                \
  } while (0)

  const int64_t n_grp = (k_end - k_begin) / 4;  // full 4-row groups
  int64_t gi = 0;
#pragma unroll
  for (int j = 0; j < WG_U; ++j)
    if (j < n_grp) load(j, k_begin + 4 * j);
  __builtin_amdgcn_sched_barrier(0);
  for (; gi + 2 * WG_U <= n_grp; gi += WG_U) {  // steady state: every group of the ring has a successor in flight
#pragma unroll
    for (int j = 0; j < WG_U; ++j) {
      WG_WAIT(12);  // 4 (WG_U - 1) younger loads may stay in flight
      compute(j);
      __builtin_amdgcn_sched_barrier(0);
      load(j, k_begin + 4 * (gi + j + WG_U));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (; gi < n_grp; gi += WG_U) {  // drain
#pragma unroll
    for (int j = 0; j < WG_U; ++j) {
      if (gi + j < n_grp) {
        WG_WAIT(0);
        compute(j);
        __builtin_amdgcn_sched_barrier(0);
        if (gi + j + WG_U < n_grp) load(j, k_begin + 4 * (gi + j + WG_U));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (k_begin + 4 * n_grp < k_end) {  // 1-3 rows left: lanes of the rows at and behind k_end contribute zero
    const int64_t k = k_begin + 4 * n_grp;
    const bool ok = k + lq < k_end;
    load(0, ok ? k : k_end - 1 - lq);
    WG_WAIT(0);
#pragma unroll
    for (int t = 0; t < TA; ++t) ra[0][t] = ok ? ra[0][t] : 0.f;
    compute(0);
  }
#undef WG_WAIT

  // acc[t][u][r]: C row = A column of (tile t, lane 4 lq + r), C column = B column of (tile u, lane li)
  float* out = W.partial + ((int64_t)s * W.batch + b) * W.M * W.Nc;
#pragma unroll
  for (int u = 0; u < WG_TB; ++u) {
    const int n = u < 4 ? n0 + 4 * li + u : u < 6 ? n0 + 64 + 2 * li + (u - 4) : n0 + 96 + li;
    if (n >= W.Nc) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mrow = m0 + TA * (4 * lq + r);
      if (mrow >= W.M) continue;
#pragma unroll
      for (int t = 0; t < TA; ++t) out[(int64_t)(mrow + t) * W.Nc + n] = acc[t][u][r];
    }
  }
}

// ---- the TALL results on the bf16 matrix cores (round 4; GGNN_GEMM=fp32 keeps the kernel above for everything) ----
// v_mfma_f32_16x16x4_f32 made the large weight gradients matrix-pipe bound (101 TFLOP/s of the 157 the fp32 pipe has: 107 us
// with its reduction for the joints' [2112 x 108] over 20 000 rows, whose operands are 169 MB).  Here both operands are split
// in registers into three bf16 pieces and multiplied with six 16x16x32 products per tile pair (common.h: fp32-equivalent, 2e-8
// of sum |a||b|, fp32's RANGE -- gradients of 1e-10 keep their 24 bits, which the two-piece fp16 split of the cells would not
// give them): 6 x 16 cycles per 32 rows where the fp32 MFMA needs 8 x 32.  Same decomposition and output as above; what
// differs is the row -> lane map (a lane holds rows k0 + 8 (lane / 16) .. +7 of its columns: the 8 k-values of a 32-deep MFMA
// operand), the block (32 x 112 per wave: 56 accumulator registers) and where B comes from.
// A first version kept B private to a wave (a register ring like the kernel above): it re-read and re-split the 32 x 112 block
// of B once per 32-row tile of the result, 66 times for the joints' projection gradient -- 590 MB through the compute units'
// vector-memory path for an 8.6 MB operand: 92 us.  Now a workgroup's four waves take four NEIGHBOURING 32-row tiles of the
// result over the same K range: each wave fetches and splits a quarter of the B block (rows 8 w .. 8 w + 7 of the 32-row
// group, two columns per lane: exactly k-group `w` of every MFMA operand fragment), parks the bf16 pieces in LDS as operand
// fragments ([column tile][piece][64 lanes][8 values], double-buffered: one barrier per 32 rows), and all four read them
// back with 21 ds_read_b128.  A quarter of the B traffic and of the B splitting per wave; no register ring (the next group's
// rows are requested before this group's MFMAs and used behind them: plain loads, the compiler counts the waits); 145
// registers: three workgroups per compute unit.  [2112 x 108] over 20 000 rows: 72-77 us with its reduction; [1248 x 112]
// over 10 000: 36 (fp32 MFMA: 46).  The 96-row results (gate weights) gain nothing from it (41 against 43 us) and stay above.
constexpr int WX_TA = 2;
#ifndef WX_CFG_WAVES
#define WX_CFG_WAVES 3072
#endif
constexpr int WX_WAVES = WX_CFG_WAVES;
constexpr int WS_LDS = WG_TB * 3 * 1024;   // one group's B fragments: 21 KB
__global__ __launch_bounds__(256, 3) void wgrad_x6s_kernel(const ggnn_wgrad_args W, int n_mt, int n_nb, int64_t chunk) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WS_LDS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int n_mtg = (n_mt + 3) / 4;
  int64_t g = xcd_remap(blockIdx.x, gridDim.x);          // workgroup -> (tile quad, column block, batch entry, K range)
  const int mtg = (int)(g % n_mtg);
  g /= n_mtg;
  const int nb = (int)(g % n_nb);
  g /= n_nb;
  const int b = (int)(g % W.batch), s = (int)(g / W.batch);
  const int mt = 4 * mtg + wave;
  const bool active = mt < n_mt;                          // (a surplus wave still does its share of B and meets the barriers)
  const int m0 = mt * 16 * WX_TA, n0 = nb * WG_NB;
  const int64_t k_begin = s * chunk, k_end = min(W.K, k_begin + chunk);

  const int ca = min(m0 + WX_TA * li, W.M - WX_TA);
  const float* A = W.a + (int64_t)b * W.a_bstride + ca;
  // B duty: rows 8 wave + j of a group, columns n0 + lane and n0 + 64 + lane (clamped: columns beyond Nc feed result
  // columns that are never stored)
  const int c0 = min(n0 + lane, W.Nc - 1), c1 = min(n0 + 64 + lane, W.Nc - 1);
  // (with an inserted operand, b_ins: a column comes from one of two matrices)
  const float *Bq0, *Bq1;
  int64_t ldq0, ldq1;
  auto column = [&](int c, const float*& p, int64_t& ld) {
    const bool in = W.b_ins != nullptr && c >= W.ins_off && c < W.ins_off + W.ins_w;
    const int cc = in ? c - W.ins_off : (W.b_ins != nullptr && c >= W.ins_off + W.ins_w ? c - W.ins_w : c);
    ld = in ? W.ld_ins : W.ldb;
    p = (in ? W.b_ins : W.b + (int64_t)b * W.b_bstride) + cc;
  };
  column(c0, Bq0, ldq0);
  column(c1, Bq1, ldq1);
  // where this lane's two columns go: tile = column / 16, fragment lane = 16 wave + column % 16
  const int frag0 = ((lane >> 4) * 3 * 64 + 16 * wave + li) * 16, frag1 = frag0 + 4 * 3 * 64 * 16;
  const bool has_c1 = lane < WG_NB - 64;

  f32x4 acc[WX_TA][WG_TB];
#pragma unroll
  for (int t = 0; t < WX_TA; ++t)
#pragma unroll
    for (int u = 0; u < WG_TB; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto split8 = [&](const float (&v)[8], u32x4 (&f)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t h, m, l;
      split_bf16x3(v[2 * e], v[2 * e + 1], h, m, l);
      f[0][e] = h, f[1][e] = m, f[2][e] = l;
    }
  };
  struct Raw {
    f32x2 a[8];
    float b0[8], b1[8];
  };
  // rows of the group at k: A rows k + 8 lq + j for this lane's operand fragment, B rows k + 8 wave + j for the shared block
  // (clamped below k_end; a ragged last group zeroes the A values it clamped: 0 x finite = 0)
  auto load = [&](int64_t k, Raw& r) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t rb = min(k + 8 * wave + j, k_end - 1);
      r.b0[j] = Bq0[rb * ldq0];
      r.b1[j] = Bq1[rb * ldq1];
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t ra = min(k + 8 * lq + j, k_end - 1);
        r.a[j] = *reinterpret_cast<const f32x2*>(A + ra * W.lda);
      }
    }
  };
  auto park_b = [&](const Raw& r, int buf) __attribute__((always_inline)) {
    unsigned char* base = smem + buf * WS_LDS;
    u32x4 f[3];
    split8(r.b0, f);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(base + frag0 + p * 1024) = f[p];
    if (has_c1) {
      split8(r.b1, f);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(base + frag1 + p * 1024) = f[p];
    }
  };
  auto multiply = [&](const Raw& r, int64_t k, int buf) __attribute__((always_inline)) {
    if (!active) return;
    u32x4 fa[WX_TA][3];
#pragma unroll
    for (int t = 0; t < WX_TA; ++t) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = k + 8 * lq + j < k_end ? r.a[j][t] : 0.f;
      split8(v, fa[t]);
    }
    const u32x4* fbp = reinterpret_cast<const u32x4*>(smem + buf * WS_LDS) + lane;
#pragma unroll
    for (int u = 0; u < WG_TB; ++u) {
      u32x4 fb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[p] = fbp[(u * 3 + p) * 64];
#pragma unroll
      for (int t = 0; t < WX_TA; ++t) acc[t][u] = mfma_x6(fa[t], fb, acc[t][u]);
    }
  };

  Raw cur, nxt;
  load(k_begin, cur);
  park_b(cur, 0);
  __syncthreads();
  int buf = 0;
  for (int64_t k = k_begin; k < k_end; k += 32, buf ^= 1) {
    const bool more = k + 32 < k_end;
    if (more) load(k + 32, nxt);                  // in flight during this group's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    multiply(cur, k, buf);
    __builtin_amdgcn_sched_barrier(0);
    if (more) park_b(nxt, buf ^ 1);
    __syncthreads();                              // the next group's fragments are complete; this group's are free
    if (more) cur = nxt;
  }
  if (!active) return;

  // acc[t][u][r]: C row = A column of (tile t, lane 4 lq + r), C column = n0 + 16 u + li
  float* out = W.partial + ((int64_t)s * W.batch + b) * W.M * W.Nc;
#pragma unroll
  for (int u = 0; u < WG_TB; ++u) {
    const int n = n0 + 16 * u + li;
    if (n >= W.Nc) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mrow = m0 + WX_TA * (4 * lq + r);
      if (mrow >= W.M) continue;
#pragma unroll
      for (int t = 0; t < WX_TA; ++t) out[(int64_t)(mrow + t) * W.Nc + n] = acc[t][u][r];
    }
  }
}

#ifndef WX_MIN_M
#define WX_MIN_M 512
#endif
static bool wgrad_uses_x6(int M) { return ggnn::gemm_mode() == 1 && M >= WX_MIN_M; }

struct WgradPlan {
  int ta, n_mt, n_nb, n_split;
  int64_t chunk;
};

static WgradPlan wgrad_plan(int64_t K, int M, int Nc, int batch) {
  WgradPlan p;
  // bf16 x 3 pieces on the matrix cores for the TALL results (the packed projection's gradient: M = 672 .. 2112 rows --
  // matrix-pipe bound on the fp32 MFMA: 107 -> 92 us for [2112 x 108] over 20 000 rows, 46 -> 41 for [1248 x 112] over 10 000);
  // the 96-row results (gate weights) are not bound by the pipe and stay on the exact fp32 MFMA, as everything does under
  // GGNN_GEMM=fp32
  const bool x6 = wgrad_uses_x6(M);
  p.ta = x6 ? WX_TA : (M % 64 == 0 || M >= 512 ? 4 : 2);
  p.n_mt = (M + 16 * p.ta - 1) / (16 * p.ta);
  p.n_nb = (Nc + WG_NB - 1) / WG_NB;
  // (shared-B kernel: a workgroup = four neighbouring row tiles, surplus waves included)
  const int64_t tiles = (int64_t)batch * (x6 ? (p.n_mt + 3) / 4 * 4 : p.n_mt) * p.n_nb;
  // one wave per SIMD of a 256-CU part (the kernel is bound by the matrix pipe: a second wave per SIMD only
  // shares it), the K ranges at least 64 rows long
  int64_t want = (x6 ? WX_WAVES : WG_WAVES) / tiles;
  const int64_t most = (K + 63) / 64;
  if (want > most) want = most;
  if (want < 1) want = 1;
  const int64_t q = x6 ? 32 : 4;   // whole row groups per K range (the last range takes the ragged end)
  p.chunk = ((K + want - 1) / want + q - 1) / q * q;
  p.n_split = (int)((K + p.chunk - 1) / p.chunk);
  if (p.n_split < 1) p.n_split = 1;
  return p;
}

// out[i] = sum_s partial[s][i], s in index order (a fixed order: reproducible), four elements per thread, the loads of
// eight splits in flight together.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int64_t n4, int n_split) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(partial) + i;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= n_split; s += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(s + j) * n4];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; s < n_split; ++s) acc += p[(int64_t)s * n4];
  reinterpret_cast<f32x4*>(out)[i] = acc;
}

}  // namespace ggnn

extern "C" int ggnn_wgrad_splits(int64_t K, int M, int Nc, int batch) {
  if (K <= 0 || M <= 0 || Nc <= 0 || batch <= 0) return 0;
  return ggnn::wgrad_plan(K, M, Nc, batch).n_split;
}

extern "C" int ggnn_wgrad(const ggnn_wgrad_args* args, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args) return GGNN_EINVAL;
  ggnn_wgrad_args W = *args;
  if (!W.a || !W.b || !W.partial || W.K <= 0 || W.M <= 0 || W.Nc <= 0 || W.batch <= 0) return GGNN_EINVAL;
  if (W.b_ins) {   // B = b with the columns of b_ins inserted (include/ggnn.h)
    if (W.batch != 1 || W.ins_off < 0 || W.ins_w <= 0 || (W.ins_off & 3) || (W.ins_w & 3) || (W.ld_ins & 3) || W.ld_ins < W.ins_w ||
        W.ins_off + W.ins_w > W.Nc || !aligned16(W.b_ins))
      return GGNN_EINVAL;
  } else {
    W.ins_off = W.ins_w = 0;
    W.ld_ins = 0;
  }
  if ((W.M & 3) || (W.Nc & 3) || W.lda < W.M || W.ldb < W.Nc - W.ins_w) return GGNN_EINVAL;
  if ((W.lda & 3) || (W.ldb & 3) || (W.a_bstride & 3) || (W.b_bstride & 3) || !aligned16(W.a) || !aligned16(W.b))
    return GGNN_EINVAL;  // dwordx4 row pieces
  const WgradPlan p = wgrad_plan(W.K, W.M, W.Nc, W.batch);
  if (W.n_split != p.n_split) return GGNN_EINVAL;  // the caller sized `partial` with ggnn_wgrad_splits
  const bool shared = wgrad_uses_x6(W.M);
  const int64_t waves = (int64_t)W.batch * p.n_nb * p.n_split * (shared ? (p.n_mt + 3) / 4 * 4 : p.n_mt);
  const int64_t blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffff) return GGNN_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (W.out && !aligned16(W.out)) return GGNN_EINVAL;
  if (shared)
    hipLaunchKernelGGL(wgrad_x6s_kernel, dim3((unsigned)blocks), dim3(256), 0, st, W, p.n_mt, p.n_nb, p.chunk);
  else if (p.ta == 4)
    hipLaunchKernelGGL(wgrad_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, W, p.n_mt, p.n_nb, p.chunk);
  else
    hipLaunchKernelGGL(wgrad_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, W, p.n_mt, p.n_nb, p.chunk);
  if (hipGetLastError() != hipSuccess) return GGNN_ELAUNCH;
  if (W.out) {   // (M, Nc multiples of 4: whole float4s)
    const int64_t n4 = (int64_t)W.batch * W.M * W.Nc / 4;
    // many partials of a small result (the heads' [4 x 100] over 313 K ranges, the encoder's [16 x 672] over 170): a thread
    // per result quad would walk all of them in a chain of n_split / 8 memory round trips (20 us for 0.5 MB) -- the row-sum
    // kernel spreads the partials over 32 thread groups
    if (W.n_split > 48 && n4 <= 16384) return launch_sum_rows(W.partial, W.out, W.n_split, 4 * n4, 1, st);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, W.partial, W.out, n4,
                       W.n_split);
  }
  return hipGetLastError() == hipSuccess ? 0 : GGNN_ELAUNCH;
}
