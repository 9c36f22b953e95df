// Encoder cell (h = c = 0) with the aggregation sweep and the gate GEMM in ONE kernel: the
// aggregates never leave the compute unit.  Same arithmetic as ggnn_period_gat_aggregate_enc_batch
// (aggregate_enc.hip: PeriodConv.message, periodGATconv.py:204-236, + propagate's gather / scatter-add
// for h = 0) followed by ggnn_lstm_epilogue in GGNN_MODE_LSTM_H0 (gates_x6.hip: lin_l2, the value-side
// lin_edge term, lin_skip and the cell update of heteropgclstm.py:111-146), but organised around the
// gate GEMM's weights instead of around an aggregate buffer:
//   * a PeriodConv of one (edge type, gate) is independent of every other up to the sum HeteroConv
//     takes over the edge types and the LSTM update over the gates, and its lin_l2 weights as two
//     fp16 planes (96 x 96 x 4 B = 36 KB; hi and scaled residual, common.h) fit the LDS: a workgroup belongs to one (problem, edge type,
//     gate), splits those weights into LDS ONCE and never streams a weight again -- no k-step slices,
//     no workgroup barrier after the prologue, the four waves (one per SIMD) run independently;
//   * a wave owns whole 16-node tiles: it sweeps the tile's in-edges (4 units per block as in
//     aggregate_enc.hip: edge records = the 16 rows of an fp32 MFMA A operand, the value weights of
//     the gate stationary in registers, bias as the accumulator's initial value so that 8 source
//     features are two k-steps, scores on one more MFMA chain, online-max softmax per row), leaves the
//     tile's 16 x 98 aggregate block in a wave-private LDS stage, reads it back as MFMA B fragments
//     (row stride = 8 mod 16 floats and k-groups interleaved by 4: conflict-free ds_read_b128), splits
//     every fragment once into two fp16 pieces and multiplies it with the resident weights (three
//     products per k-step, main + cross accumulators: common.h); the two rank-1 columns go through one exact fp32 MFMA; the
//     16 x 96 block of partial pre-activations is stored;
//   * the operands of a block (16 edge records, 4 score tails) arrive by LDS-DMA in a per-wave ring
//     of EC_U slots, requested EC_U blocks ahead of their use ACROSS tile boundaries -- a wave alone on
//     its SIMD has nobody to hide a memory round trip behind, registers are not held while the loads
//     fly, and the ring is the only vector-memory traffic of the loop besides the output stores, so
//     its completion is one counted s_waitcnt vmcnt per pair of blocks; a group of 16 lanes walks
//     FOUR CONSECUTIVE rows of the tile, cutting them into units of <= 3 edges arithmetically from the
//     17 rowptr entries of the tile, which arrive by LDS-DMA as well (a header ring, fetched EC_HMAX
//     tiles ahead): no scalar load and no register-destination load in the loop (a scalar load would
//     share lgkmcnt with the LDS traffic: every LDS wait would also wait for it);
//   * a second, element-wise launch sums the partial pre-activations over the incoming edge types,
//     adds the skip term and applies the LSTM update.
// What it saves against sweep + gate kernel: the aggregate round trip (69 MB written and read per
// model at cfg3), the weight stream (every CU streamed all weights of all gates per pass), one
// barrier per k-step, and the prologue / epilogue of a second full-chip launch.
#include <algorithm>

#include "common.h"
#define GGNN_STAMP_SUFFIX _enc
#include "stamps.h"

namespace ggnn {

#ifndef EC_CFG_WAVES       // (development: make VARIANT=.. EXTRA="-DEC_CFG_WAVES=.. -DEC_CFG_U=.. -DEC_CFG_HMAX=.. -DEC_CFG_MINW=..")
#define EC_CFG_WAVES 8
#define EC_CFG_U 4
#define EC_CFG_HMAX 6
#define EC_CFG_MINW 2
#endif
constexpr int EC_WAVES = EC_CFG_WAVES;       // two per SIMD: a wave alone issues one instruction per >= 4 cycles
constexpr int EC_MAX_PROBLEMS = 4;
constexpr int EC_MAX_COMBOS = EC_MAX_PROBLEMS * 2 * 3;  // (problem, incoming edge type, gate)
constexpr int EC_G = 3;                      // i, c, o
constexpr int EC_S = 104;                    // stage row stride in floats: 8 mod 16
constexpr int EC_U = EC_CFG_U;               // ring slots per wave = blocks in flight
constexpr int EC_SLOT = 1024 + 256 + 32;     // 16 records x 64 B | 4 tails x 64 B | control words
constexpr int EC_PL = 2;                     // fp16 pieces per operand (common.h: split_f16x2 / mfma_x3h)
constexpr int EC_PLANES = EC_PL * 18 * 1024; // 36 864 B
constexpr int EC_HMAX = EC_CFG_HMAX, EC_HR = EC_HMAX + 1 + (EC_HMAX & 1);  // tiles fetched ahead of the front cursor, header ring slots
constexpr int EC_HSLOT = 80;                 // 17 rowptr entries
constexpr int EC_WAVE_LDS = 16 * EC_S * 4 + EC_U * EC_SLOT + EC_HR * EC_HSLOT;  // 6 656 + 5 248 + 640
constexpr int EC_LDS_BYTES = EC_PLANES + EC_WAVES * EC_WAVE_LDS;  // 137 216
static_assert(EC_LDS_BYTES <= 160 * 1024, "LDS");
static_assert(2 * (EC_U - 2) < 64 && 2 * EC_HMAX > 2 * (EC_U - 2) + 4, "vmcnt is a 6-bit counter; headers land in time");
static_assert(EC_HMAX >= EC_U && EC_HMAX < EC_HR, "a header must be requested a ring length ahead of its use");

struct EncCellBatch {
  ggnn_enc_cell_args a[EC_MAX_PROBLEMS];
  int wg_off[EC_MAX_COMBOS + 1];  // first workgroup of every combination
  int combo[EC_MAX_COMBOS];       // problem | edge type << 2 | gate << 3
  int n;                          // combinations
};

typedef int ec_i32x4 __attribute__((ext_vector_type(4)));
typedef int ec_i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ec_bperm(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float ec_relu(float x) {  // one v_max_f32
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}
__device__ __forceinline__ int ec_group_max(int v) {  // max over the four 16-lane groups of a group-uniform value
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ float ec_quad_lane3(float v) {  // value of the quad's 4th lane in all four
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xFF, 0xF, 0xF, true));
}
// LDS-DMA: every lane copies 16 / 4 bytes from its own global address to lds_base + lane * 16 / 4
// (wave-uniform base in M0, written and restored inside the statement).  Not tracked by the compiler:
// completion is awaited with ec_dma_wait<N>() (LDS-DMA completes in issue order).
__device__ __forceinline__ void ec_dma16(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base))
               : "memory");
}
__device__ __forceinline__ void ec_dma4(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base))
               : "memory");
}
template <int N> __device__ __forceinline__ void ec_dma_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void enc_cell_body(const ggnn_enc_cell_args& A, const int d, const int g, const int wg,
                                              const int nwg, unsigned char* __restrict__ smem) {
  const ggnn_enc_cell_sweep& Sw = A.in[d];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int qa = c >> 2, ra = c & 3;       // sweep: A row / score column c belongs to unit qa, edge ra
  const int src_sm = (20 * q) * 4;         // lane 16 q + 4 q: the softmax lane of group q (ds_bpermute)
  u32x4* __restrict__ wpl = reinterpret_cast<u32x4*>(smem);  // [3][6][2][64]: lin_l2 of (edge type, gate) as two fp16 planes
  unsigned char* __restrict__ wbase = smem + EC_PLANES + wave * EC_WAVE_LDS;
  float* __restrict__ stage = reinterpret_cast<float*>(wbase);   // wave-private [16][EC_S]
  unsigned char* __restrict__ ring = wbase + 16 * EC_S * 4;                       // wave-private [EC_U][EC_SLOT]
  const unsigned char* __restrict__ hdr = ring + EC_U * EC_SLOT;                   // wave-private [EC_HR][EC_HSLOT]
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ring)));
  const uint32_t hdr_lds = ring_lds + EC_U * EC_SLOT;
  const int n_dst = (int)A.n_dst;
  const int nks_all = 3 * A.n_in;          // k-steps of the whole gate weight (96 columns per incoming edge type)
  const int kmt = 96 * A.n_in;

  // ---- this wave's tiles: t_lo + wave, + EC_WAVES, ... < t_hi ----
  const int n_t = (n_dst + 15) >> 4;
  const int t_lo = (int)((int64_t)wg * n_t / nwg), t_hi = (int)((int64_t)(wg + 1) * n_t / nwg);
  const int row_last = max(n_dst - 16, 0);  // a ragged last tile slides back (identical duplicate results)
  const int e_last = (int)Sw.E + GGNN_UNIT_EDGES - 1;
  const bool nk3 = Sw.f_src > 8;
  const int32_t* __restrict__ rowptr = Sw.rowptr;
  const float* __restrict__ einfo = Sw.einfo;
  const float* __restrict__ tails = A.p_dst + Sw.u4_off + 16 * g;
  const int ldp32 = (int)A.ldp;            // n_dst * ldp < 2^31 (checked by the host)
  float* __restrict__ pre = A.pre + (int64_t)d * n_dst * (EC_G * C) + g * C;
  const int src_grp = (16 * qa) * 4;       // byte address of a lane of group qa (ds_bpermute)

  // ---- tile headers: rowptr[row0 .. row0 + 16] of the wave's idx-th tile -> header ring (one DMA) ----
  auto hdr_issue = [&](int idx) {
    const int tile = t_lo + wave + EC_WAVES * idx;
    const int row0 = tile < t_hi ? min(tile * 16, row_last) : 0;
    if (lane < 17) ec_dma4(rowptr + min(row0 + lane, n_dst), hdr_lds + (idx % EC_HR) * EC_HSLOT);
  };
  int h_idx = 0;                 // next tile whose header has not been requested yet
  for (; h_idx < EC_HMAX; ++h_idx) hdr_issue(h_idx);  // in flight while the weights are staged
  GGNN_STAMP(0);
  // ---- prologue: lin_l2 of this (edge type, gate) -> two fp16 planes in LDS (once per workgroup) ----
  {
    const f32x4* __restrict__ wf =
        reinterpret_cast<const f32x4*>(A.w2_frag) + (size_t)((g * nks_all + 3 * d) * 6) * 2 * 64;
    for (int f = wave; f < 18; f += EC_WAVES) {
      const f32x4 h0 = wf[(f * 2) * 64 + lane], h1 = wf[(f * 2 + 1) * 64 + lane];
      u32x4 pl[EC_PL];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 h = e < 2 ? h0 : h1;
        uint32_t q0, q1;
        split_f16x2(h[2 * (e & 1)], h[2 * (e & 1) + 1], q0, q1);
        pl[0][e] = q0;
        pl[1][e] = q1;
      }
#pragma unroll
      for (int p = 0; p < EC_PL; ++p) wpl[(f * EC_PL + p) * 64 + lane] = pl[p];
    }
    // the stage starts as zeros (rows of a tile that do not exist are never written)
    for (int i = lane; i < 16 * EC_S / 4; i += 64) reinterpret_cast<f32x4*>(stage)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // value weights of this gate (B fragments, 16x16x4: k = 4 s + q, column c of tile t) and bias, stationary
  float bw[6][3], bb[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const float* __restrict__ w = Sw.wv_frag + (size_t)((g * 6 + t) * 4) * 64 + lane;
#pragma unroll
    for (int s = 0; s < 3; ++s) bw[t][s] = w[s * 64];
    bb[t] = w[3 * 64];
  }
  // weight side of the exact fp32 tail (b_l2 and w_edge of this edge type), 16x16x4 A fragments, k = q < 2
  float wt[6];
#pragma unroll
  for (int ct = 0; ct < 6; ++ct)
    wt[ct] = q < 2 ? A.w2[(size_t)(g * C + ct * 16 + c) * A.Ka + kmt + 2 * d + q] : 0.f;
  __syncthreads();
  GGNN_STAMP(1);

  // ---- front cursor: EC_U blocks ahead of the compute.  Group q walks rows row0 + 4 q .. + 3 of the tile,
  // row k in units of <= 3 edges (rows without edges: one empty unit, so that zeros are stored) ----
  int f_idx = 0, f_b = 0, f_nblk = 0, f_row0 = 0;
  bool f_ok = false;
  const int* hp = reinterpret_cast<const int*>(hdr) + 4 * q;  // rowptr of the group's rows in the cursor's header slot
  int f_k = 0, f_j = 0;          // row / unit inside the row at the cursor (group-uniform)
  auto units_of = [](int deg) { return max(1, (deg + GGNN_UNIT_EDGES - 1) / GGNN_UNIT_EDGES); };
  auto enter_tile = [&]() {      // header of tile f_idx -> cursor state (the header landed a ring length ago)
    const int tile = t_lo + wave + EC_WAVES * f_idx;
    f_ok = tile < t_hi;
    f_row0 = min(tile * 16, row_last);
    hp = reinterpret_cast<const int*>(hdr + (f_idx % EC_HR) * EC_HSLOT) + 4 * q;
    int tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) tot += units_of(hp[k + 1] - hp[k]);
    f_nblk = ec_group_max(tot);
    f_b = 0;
    f_k = f_j = 0;
  };
  // fills ring slot `slot` with the block at the cursor (two LDS-DMAs + the control words), requests one
  // more header, then advances the cursor
  auto issue = [&](int slot) {
    const bool alive = f_ok && f_k < 4;
    const int rpk = hp[f_k], rpk1 = hp[f_k + 1];   // (f_k == 4, a finished group: inside the slot, unused)
    const int nuk = units_of(rpk1 - rpk);
    const int p0 = alive ? rpk + GGNN_UNIT_EDGES * f_j : 0;
    const int nact = min(max(rpk1 - p0, 0), GGNN_UNIT_EDGES);
    const int rowl = 4 * q + f_k;
    // row inside the tile | nact << 4 | first << 6 | last << 7
    const int word = alive ? (rowl | (nact << 4) | ((f_j == 0) << 6) | ((f_j + 1 == nuk) << 7)) : 0;
    const int i_q = alive ? min(f_row0 + rowl, n_dst - 1) : 0;
    const int meta = f_ok ? (f_row0 | ((f_b + 1 == f_nblk) << 28) | (1 << 30)) : 0;
    const uint32_t base = ring_lds + slot * EC_SLOT;
    const int p0_a = __builtin_amdgcn_ds_bpermute(src_grp, p0);
    // record of edge ra of unit qa (slot 3 is padding: whatever record follows, clamped to the buffer), piece q
    ec_dma16(einfo + (uint32_t)min(p0_a + ra, e_last) * GGNN_EINFO_ROW + 4 * q, base);
    // score tail of unit q's row, element c
    ec_dma4(tails + (uint32_t)(i_q * ldp32 + c), base + 1024);
    int* __restrict__ cw = reinterpret_cast<int*>(ring + slot * EC_SLOT + 1280);
    if (c == 0) cw[q] = word;
    if (lane == 0) cw[4] = meta;
    // headers are requested as the cursor moves on (one per tile in the steady state): the loop's counted
    // wait assumes none in its window, so a recent one only makes it wait for one DMA more than needed
    if (h_idx - f_idx < EC_HMAX) hdr_issue(h_idx++);
    // advance
    const bool row_done = alive && f_j + 1 == nuk;
    f_j = row_done ? 0 : f_j + (alive ? 1 : 0);
    f_k += row_done ? 1 : 0;
    if (f_ok && ++f_b == f_nblk) {
      ++f_idx;
      enter_tile();
    }
  };

  // ---- the GEMM of a finished tile: stage -> B fragments -> two fp16 pieces, three products against the resident
  // planes (main + cross accumulators: common.h; until round 3's last version three bf16 pieces and six products) ----
  auto gemm = [&](int row0) {
    f32x4 acc[6], accx[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) acc[ct] = accx[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ srow = stage + c * EC_S + 4 * q;  // node c of the tile, k-group q
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(srow + 32 * ks);
      const f32x4 r1 = *reinterpret_cast<const f32x4*>(srow + 32 * ks + 16);
      u32x4 xb[EC_PL];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 h = e < 2 ? r0 : r1;
        uint32_t q0, q1;
        split_f16x2(h[2 * (e & 1)], h[2 * (e & 1) + 1], q0, q1);
        xb[0][e] = q0;
        xb[1][e] = q1;
      }
      const u32x4* __restrict__ pw = wpl + (ks * 6 * EC_PL) * 64 + lane;
      // the weight fragments of column tile ct + 1 are read while the three MFMAs of ct run
      u32x4 wf[2][EC_PL];
#pragma unroll
      for (int p = 0; p < EC_PL; ++p) wf[0][p] = pw[p * 64];
      __builtin_amdgcn_sched_group_barrier(0x100, EC_PL, 0);  // DS reads of ct = 0, then [reads ct + 1 | MFMAs ct] ...
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) {
        if (ct + 1 < 6) {
#pragma unroll
          for (int p = 0; p < EC_PL; ++p) wf[(ct + 1) & 1][p] = pw[((ct + 1) * EC_PL + p) * 64];
        }
        mfma_x3h(wf[ct & 1], xb, acc[ct], accx[ct]);
        if (ct + 1 < 6) __builtin_amdgcn_sched_group_barrier(0x100, EC_PL, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                      // MFMA
      }
    }
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) acc[ct] += accx[ct] * (1.0f / F16X2_SCALE);
    const float xt = q < 2 ? stage[c * EC_S + 96 + q] : 0.f;
    const bool ok = row0 + c < n_dst;
    float* __restrict__ o = pre + (int64_t)(row0 + (ok ? c : 0)) * (EC_G * C) + 4 * q;
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) {
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[ct], xt, acc[ct], 0, 0, 0);
      if (ok) *reinterpret_cast<f32x4*>(o + 16 * ct) = acc[ct];
    }
  };

  // ---- start-up: the first EC_HMAX headers have landed by now; then the ring (blocks 0 .. EC_U - 1) ----
  ec_dma_wait<0>();
  GGNN_STAMP(2);
  enter_tile();
  for (int r = 0; r < EC_U; ++r) issue(r);
  GGNN_STAMP(3);
  [[maybe_unused]] unsigned long long st_wait = 0, st_comp = 0, st_issue = 0, st_gemm = 0, st_n = 0, st_t = 0;

  float mx = -INFINITY, den = 0.f, sae = 0.f;  // softmax state of the row group q is folding (mx: lane 4 q)
  float acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) acc[t] = 0.f;

  // Two blocks per iteration: their MFMA chains are independent (a wave is bound by the latency of its own
  // chain -- LDS round trips, dependent MFMAs --, not by the SIMD), only the softmax fold is sequential.
  static_assert(EC_U % 2 == 0, "two blocks per iteration");
  for (int slot = 0;; slot = slot + 2 == EC_U ? 0 : slot + 2) {
    [[maybe_unused]] const unsigned long long st0 = GGNN_STAMP_NOW();
    // everything but the two DMAs of each of the EC_U - 2 blocks issued after these two (a header DMA issued
    // EC_HMAX tiles ahead of its use has >= 2 EC_HMAX younger DMAs by then: complete as well)
    ec_dma_wait<2 * (EC_U - 2)>();
    [[maybe_unused]] const unsigned long long st1 = GGNN_STAMP_NOW();
    const unsigned char* __restrict__ sl = ring + slot * EC_SLOT;
    int my[2], meta[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int* __restrict__ cwp = reinterpret_cast<const int*>(sl + j * EC_SLOT + 1280);
      my[j] = cwp[q];
      meta[j] = __builtin_amdgcn_readfirstlane(cwp[4]);
    }
    if (!((meta[0] >> 30) & 1)) break;
    // (a block past the end -- only the second of a pair can be one -- has real operands, no edges and
    // neither first nor last: it changes nothing)
    f32x4 sc[2], v[2][6];
    float ae[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float* __restrict__ sa = reinterpret_cast<const float*>(sl + j * EC_SLOT) + 4 * c + q;
      const float* __restrict__ sb = reinterpret_cast<const float*>(sl + j * EC_SLOT + 1024) + 16 * qa + q;
      float a[4], bs[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = sa[64 * s];
#pragma unroll
      for (int s = 0; s < 4; ++s) bs[s] = sb[4 * s];  // (score columns other than 4 u: never read)
      // a_e = x4[13] of the three edges of unit q: piece 3 of records 4 q + r (group-uniform reads)
#pragma unroll
      for (int r = 0; r < 3; ++r) ae[j][r] = reinterpret_cast<const float*>(sl + j * EC_SLOT)[4 * (48 + 4 * q + r) + 1];
      // scores: D[edge 4 q + r][column c]; lane c = 4 q holds unit q's scores
      sc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) sc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bs[s], sc[j], 0, 0, 0);
      // values: per column tile D[edge 4 q + r][column c] = W x~ + b
#pragma unroll
      for (int t = 0; t < 6; ++t) v[j][t] = (f32x4){bb[t], bb[t], bb[t], bb[t]};
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 6; ++t) v[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw[t][s], v[j][t], 0, 0, 0);
      if (nk3) {
#pragma unroll
        for (int t = 0; t < 6; ++t) v[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], bw[t][2], v[j][t], 0, 0, 0);
      }
    }
    [[maybe_unused]] const unsigned long long st2 = GGNN_STAMP_NOW();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row0 = meta[j] & 0x7FFFFF;
      const int rowl = my[j] & 15, nact = (my[j] >> 4) & 3;
      const bool first = (my[j] >> 6) & 1, last = (my[j] >> 7) & 1;
      if (first) {
        mx = -INFINITY;
        den = sae = 0.f;
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t] = 0.f;
      }
      const float s0 = nact > 0 ? sc[j][0] : -INFINITY, s1 = nact > 1 ? sc[j][1] : -INFINITY,
                  s2 = nact > 2 ? sc[j][2] : -INFINITY;
      const float mnew = fmaxf(fmaxf(mx, s0), fmaxf(s1, s2));
      const bool any = nact > 0;  // (an empty row has one unit with nact == 0: zeros are stored)
      const float scale_l = any ? __expf(mx - mnew) : 1.0f;  // exp(-inf) = 0 on a row's first unit
      const float p0_l = any ? __expf(s0 - mnew) : 0.f, p1_l = any ? __expf(s1 - mnew) : 0.f,
                  p2_l = any ? __expf(s2 - mnew) : 0.f;
      if (any) mx = mnew;
      const float scale = ec_bperm(src_sm, scale_l), p0 = ec_bperm(src_sm, p0_l), p1 = ec_bperm(src_sm, p1_l),
                  p2 = ec_bperm(src_sm, p2_l);
      const float e0 = ae[j][0], e1 = ae[j][1], e2 = ae[j][2];
      den = den * scale + (p0 + p1 + p2);
      sae = sae * scale + (p0 * e0 + p1 * e1 + p2 * e2);
      // relu, alpha-weighted sum
#pragma unroll
      for (int t = 0; t < 6; ++t)
        acc[t] = acc[t] * scale + (p0 * ec_relu(v[j][t][0]) + p1 * ec_relu(v[j][t][1]) + p2 * ec_relu(v[j][t][2]));
      if (last) {  // the row's 96 aggregate channels -> stage (8 bytes per lane: 128 per group)
        const float inv = __builtin_amdgcn_rcpf(den + 1e-16f);  // PyG softmax denominator (v_rcp_f32: 1 ulp)
        float* __restrict__ srow = stage + rowl * EC_S;
#pragma unroll
        for (int m = 0; m < 6; m += 2) {
          const f32x2_ w2 = {acc[m] * inv, acc[m + 1] * inv};
          *reinterpret_cast<f32x2_*>(srow + 2 * c + 16 * m) = w2;
        }
        if (c == 0) {
          const f32x2_ t2 = {den * inv, sae * inv};
          *reinterpret_cast<f32x2_*>(srow + 96) = t2;
        }
      }
      if ((meta[j] >> 28) & 1) {
        [[maybe_unused]] const unsigned long long sg0 = GGNN_STAMP_NOW();
        gemm(row0);
        st_gemm += GGNN_STAMP_NOW() - sg0;
        ++st_t;
      }
    }
    // ---- the two slots are free: blocks n + EC_U, n + EC_U + 1 take them over ----
    [[maybe_unused]] const unsigned long long st3 = GGNN_STAMP_NOW();
    issue(slot);
    issue(slot + 1);
    [[maybe_unused]] const unsigned long long st4 = GGNN_STAMP_NOW();
    st_wait += st1 - st0;
    st_comp += st3 - st1;
    st_issue += st4 - st3;
    st_n += 2;
  }
  ec_dma_wait<0>();  // no LDS-DMA may outlive the wave
  GGNN_STAMP_VAL(4, st_wait);
  GGNN_STAMP_VAL(5, st_comp);
  GGNN_STAMP_VAL(6, st_issue);
  GGNN_STAMP_VAL(7, st_gemm);
  GGNN_STAMP_VAL(8, st_n);
  GGNN_STAMP_VAL(9, st_t);
  GGNN_STAMP(16);
}

__global__ __launch_bounds__(EC_WAVES * 64, EC_CFG_MINW) void enc_cell_kernel(const EncCellBatch B) {
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[EC_LDS_BYTES];
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const int wg = (int)blockIdx.x - B.wg_off[k], nwg = B.wg_off[k + 1] - B.wg_off[k];
  const int cb = B.combo[k];
  enc_cell_body(B.a[cb & 3], (cb >> 2) & 1, cb >> 3, wg, nwg, s_raw);
}

// ---- LSTM update from zero state (heteropgclstm.py:111-146 with h = c = 0): c' = sig(i) tanh(c~),
// h' = sig(o) tanh(c'); pre-activation = sum over the incoming edge types of the gate GEMM results + the
// summed skip / gate-bias term, which is formed HERE from the node's 8 / 11 features (weights in LDS)
// instead of being written and read back as 288 projection columns ----
struct EncLstmBatch {
  ggnn_enc_cell_args a[EC_MAX_PROBLEMS];
  int blk_off[EC_MAX_PROBLEMS + 1];
  int n;
};
constexpr int EL_MAXF = 12;

// (Tried: no LDS, the weight table through the vector cache, so that this launch of one model could share
// compute units with the other model's enc_cell_kernel in the two-stream rollout, whose workgroups leave 8 KB of
// LDS: 26.8 instead of 20.4 us alone, and no overlap gained -- the issue-bound encoder cell starves it anyway.)
__global__ __launch_bounds__(256) void enc_lstm_kernel(const EncLstmBatch B) {
  __shared__ __attribute__((aligned(16))) float s_w[(EL_MAXF + 1) * EC_G * C];  // [k][288] skip weights, then the bias row
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.blk_off[k + 1]) ++k;
  const ggnn_enc_cell_args& A = B.a[k];
  const int F = A.f_dst;
  for (int t = threadIdx.x; t < (F + 1) * EC_G * C; t += 256) s_w[t] = A.ws_t[t];
  __syncthreads();
  // a workgroup keeps the weights for a whole range of nodes: (node, 4-channel) quads, 256 per pass
  const int nblk = B.blk_off[k + 1] - B.blk_off[k], blk = (int)blockIdx.x - B.blk_off[k];
  const int64_t n_q = A.n_dst * 24, per = (n_q + nblk - 1) / nblk;
  const int64_t q_hi = min(n_q, (int64_t)(blk + 1) * per);
  const int64_t part = A.n_dst * (int64_t)(EC_G * C);  // one partial per incoming edge type (HeteroConv sums them)
  for (int64_t t = (int64_t)blk * per + threadIdx.x; t < q_hi; t += 256) {
    const int64_t node = t / 24;
    const int c4 = (int)(t - node * 24);
    const float* __restrict__ pr = A.pre + node * (EC_G * C) + 4 * c4;
    f32x4 p[EC_G];
#pragma unroll
    for (int g = 0; g < EC_G; ++g) {
      p[g] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pr + g * C)) +
             *reinterpret_cast<const f32x4*>(&s_w[F * EC_G * C + g * C + 4 * c4]);
      if (A.n_in == 2) p[g] += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pr + part + g * C));
    }
    const float* __restrict__ x = A.x_dst + node * A.ldx;
    float xv[EL_MAXF];  // all features requested at once (clamped index: no load under a branch)
#pragma unroll
    for (int f = 0; f < EL_MAXF; ++f) xv[f] = x[min(f, F - 1)];
#pragma unroll
    for (int f = 0; f < EL_MAXF; ++f) {
      if (f < F) {
#pragma unroll
        for (int g = 0; g < EC_G; ++g) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(&s_w[f * EC_G * C + g * C + 4 * c4]);
#pragma unroll
          for (int r = 0; r < 4; ++r) p[g][r] = fmaf(xv[f], w[r], p[g][r]);
        }
      }
    }
    f32x4 hn, cn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float cv = sigmoidf_(p[0][r]) * tanhf_(p[1][r]);
      cn[r] = cv;
      hn[r] = sigmoidf_(p[2][r]) * tanhf_(cv);
    }
    *reinterpret_cast<f32x4*>(A.c_out + node * C + 4 * c4) = cn;
    *reinterpret_cast<f32x4*>(A.h_out + node * C + 4 * c4) = hn;
  }
}

}  // namespace ggnn

extern "C" int ggnn_encoder_cell_batch(const ggnn_enc_cell_args* args, int n_problems, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_problems < 1 || n_problems > EC_MAX_PROBLEMS) return GGNN_EINVAL;
  EncCellBatch B;
  EncLstmBatch L;
  L.n = n_problems;
  double cost[EC_MAX_COMBOS];   // relative time of one (problem, edge type) sweep + GEMM, per gate
  int64_t n_t[EC_MAX_COMBOS];
  int kd[EC_MAX_COMBOS];
  int n_kd = 0;
  double total = 0.0;
  for (int k = 0; k < EC_MAX_PROBLEMS; ++k) {
    B.a[k] = L.a[k] = args[k < n_problems ? k : 0];
    if (k >= n_problems) continue;
    const ggnn_enc_cell_args& A = B.a[k];
    if (A.n_in < 1 || A.n_in > 2 || A.n_dst <= 0 || A.Ka != 96 * A.n_in + 4) return GGNN_EINVAL;
    if (!A.p_dst || !A.w2_frag || !A.w2 || !A.pre || !A.h_out || !A.c_out || !A.x_dst || !A.ws_t) return GGNN_EINVAL;
    if (!aligned16(A.p_dst) || !aligned16(A.w2_frag) || !aligned16(A.pre) || !aligned16(A.h_out) || !aligned16(A.c_out))
      return GGNN_EINVAL;
    if (A.ldp <= 0 || A.f_dst < 1 || A.f_dst > EL_MAXF || A.ldx < A.f_dst || !aligned16(A.ws_t)) return GGNN_EINVAL;
    if (A.n_dst >= (1 << 23) || A.n_dst * A.ldp >= INT32_MAX) return GGNN_EINVAL;  // row0 field of the meta word
    for (int e = 0; e < A.n_in; ++e) {
      const ggnn_enc_cell_sweep& Sw = A.in[e];
      if (!Sw.rowptr || !Sw.einfo || !Sw.wv_frag || !aligned16(Sw.einfo)) return GGNN_EINVAL;
      if (Sw.E < 0 || Sw.f_src < 3 || Sw.f_src > 12 || Sw.u4_off < 0 || Sw.u4_off + EC_G * 16 > A.ldp)
        return GGNN_EINVAL;
      if ((Sw.E + GGNN_UNIT_EDGES + 1) * GGNN_EINFO_ROW >= INT32_MAX) return GGNN_EINVAL;
      n_t[n_kd] = (A.n_dst + 15) / 16;
      // time of the combination in units of one sweep block (four units; measured: the loop is bound by
      // instruction issue, a third k-step adds ~12 %, a tile's GEMM costs ~1.6 blocks); a row has
      // max(1, ceil(deg / 3)) units: ~ max(n_dst, E / 3) for the degrees of a grain structure
      const double blocks = Sw.n_blocks > 0 ? (double)Sw.n_blocks
                                            : (double)std::max<int64_t>(A.n_dst, Sw.E / GGNN_UNIT_EDGES) / 4.0;
      cost[n_kd] = blocks * (Sw.f_src > 8 ? 1.12 : 1.0) +
                   (double)n_t[n_kd] * 1.6;
      total += EC_G * cost[n_kd];
      kd[n_kd++] = k | (e << 2);
    }
  }
  // One persistent workgroup per compute unit, dealt to the (problem, edge type, gate) combinations in
  // proportion to their matrix-core work; never more workgroups than tiles.
  const int ncu = num_cu();
  int nwg[EC_MAX_COMBOS];
  int used = 0;
  for (int j = 0; j < n_kd; ++j) {
    nwg[j] = (int)std::min<int64_t>(n_t[j], std::max<int64_t>(1, (int64_t)(ncu * cost[j] / total)));
    used += EC_G * nwg[j];
  }
  for (;;) {  // left-over compute units go to the combination with the most work per workgroup
    int best = -1;
    for (int j = 0; j < n_kd; ++j)
      if (nwg[j] < n_t[j] && (best < 0 || cost[j] / nwg[j] > cost[best] / nwg[best])) best = j;
    if (best < 0 || used + EC_G > ncu) break;
    ++nwg[best];
    used += EC_G;
  }
  B.n = EC_G * n_kd;
  B.wg_off[0] = 0;
  for (int j = 0; j < EC_MAX_COMBOS; ++j) {
    const int jj = j / EC_G, g = j % EC_G;
    B.combo[j] = jj < n_kd ? (kd[jj] | (g << 3)) : 0;
    B.wg_off[j + 1] = B.wg_off[j] + (jj < n_kd ? nwg[jj] : 0);
  }
  // LSTM launch: at most eight workgroups per compute unit in all, dealt by node count, 256 quads per pass
  int64_t quads = 0;
  for (int k = 0; k < n_problems; ++k) quads += B.a[k].n_dst * 24;
  L.blk_off[0] = 0;
  for (int k = 0; k < EC_MAX_PROBLEMS; ++k) {
    int64_t nb = 0;
    if (k < n_problems) {
      const int64_t full = (B.a[k].n_dst * 24 + 255) / 256;
      nb = std::max<int64_t>(1, std::min<int64_t>(full, (int64_t)8 * ncu * (B.a[k].n_dst * 24) / quads));
    }
    L.blk_off[k + 1] = L.blk_off[k] + (int)nb;
  }
  hipLaunchKernelGGL(enc_cell_kernel, dim3((unsigned)B.wg_off[EC_MAX_COMBOS]), dim3(EC_WAVES * 64), 0,
                     (hipStream_t)stream, B);
  hipLaunchKernelGGL(enc_lstm_kernel, dim3((unsigned)L.blk_off[EC_MAX_PROBLEMS]), dim3(256), 0, (hipStream_t)stream, L);
  return launch_status();
}
