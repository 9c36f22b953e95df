// The fused decoder cell (dec_cell.hip: same arguments, same weight stream, same arithmetic) with the three
// kinds of work of a tile given to three kinds of WAVES of one workgroup instead of to one wave in turn:
//
//   M-waves (4, one per SIMD)  the matrix chain of one 16-node tile each: P1 (u_h | u4), P3 (lin_l2 over the
//                              aggregate), P4 (skip), the LSTM update.  Nothing but ds_read + split + MFMA.
//   S-waves (4)                the periodic-boundary GAT sweep of the same tile, (gate, edge type) phase k while
//                              the tile's M-wave already computes u of phase k + 1.
//   L-waves (4)                the weight stream: LDS-DMA of the k-step slices into a ring of three buffers,
//                              up to two slices ahead of the slowest M-wave.
//
// dec_cell.hip's stamps say why: there one wave walks P1 -> sweep -> P3 -> P4 of its tile, every phase is
// latency-bound on its own (MFMA chains with one wave per SIMD slot, two exposed gather round trips per sweep,
// 150 issue cycles per DMA piece), and with ~1 tile per wave slot nothing else is there to fill the gaps.  Here
// the three chains of ONE tile overlap each other.
//
// Hand-offs are generation counters in LDS (monotonic, never reset), no workgroup barrier after the start:
//   l_done[j] = t + 1 when L-wave j's pieces of slice t have landed         (M waits for all four >= t + 1)
//   m_fin[m]  = t + 1 when M-wave m has finished reading slice t            (L waits for all four >= t - 2 before
//                                                                           overwriting buffer t % 3)
//   (one counter per wave: a sum over the waves would let two fast waves stand in for a slow one)
//   u_gen[m][b] = phase + 1 when the M-wave of tile m has stored u of that phase in stage slot b = phase & 1
//   a_gen[m][b] = phase + 1 when its S-wave has replaced those rows by the aggregates
//   csr_gen[m]  = tile set + 1 when the tile's CSR windows are in LDS
// LDS operations of a wave execute in order, so data-then-flag stores and flag-then-data loads need no fence;
// DMA pieces are confirmed by the issuing wave's counted s_waitcnt before its flag.  No cycle: L(t) waits for M
// on slices <= t - 3 only; M publishes u(k) before it waits for A(k); S waits for u(k) only.  Every spin is
// BOUNDED all the same: a wave that gives up raises `err`, every other spin then falls through, the kernel
// drains and the outputs' first row is NaN -- a protocol bug fails the parity tests, it does not hang the GPU.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#define GGNN_STAMP_SUFFIX _decws
#include "stamps.h"

namespace ggnn {

constexpr int WS_TILES = 4;                         // tiles of a tile set = M-waves = S-waves = L-waves
constexpr int WS_WAVES = 3 * WS_TILES;
constexpr int WS_RING = 3;
constexpr int WS_MAX_PROBLEMS = 4;
constexpr int WS_SLICE = GGNN_DC_SLICE_BYTES;
constexpr int WS_S = 116;                           // stage row stride in floats
constexpr int WS_STAGE = 16 * WS_S * 4;
constexpr int WS_CW = 111;
constexpr int WS_CSR = (17 + WS_CW) * 4;
constexpr int WS_WV = 16 * 20 * 4;                  // a phase's lin_edge weights, [16 lanes of a DPP row][18 + pad]
constexpr int WS_TILE_LDS = 3 * WS_STAGE + 2 * WS_CSR + WS_WV;   // u / aggregate slots 0, 1; [h | x | 1 | 0] rows; CSR; wv
constexpr int WS_CTRL_INTS = 32;
constexpr int WS_LDS = WS_RING * WS_SLICE + WS_TILES * WS_TILE_LDS + WS_CTRL_INTS * 4;   // 162 944 B
static_assert(WS_LDS <= 160 * 1024, "LDS");
#ifndef WS_LONG_SLEEP
#define WS_LONG_SLEEP 8
#endif
constexpr int WS_SPIN_LIMIT = 1 << 17;              // x (s_sleep + LDS read) ~ 20 ms

// control words
constexpr int CW_ERR = 0, CW_UGEN = 4, CW_AGEN = 12, CW_CSRGEN = 20, CW_LDONE = 24, CW_MFIN = 28;
typedef __attribute__((address_space(3))) volatile int lds_vint;   // control words are read and written as LDS, not flat

struct WsBatch {
  ggnn_dec_cell_args a[WS_MAX_PROBLEMS];
  int wg_off[WS_MAX_PROBLEMS + 1];
  int n, tsw;   // tile sets per workgroup
};

__device__ __forceinline__ void ws_dma16(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base))
               : "memory");
}

__device__ __forceinline__ void ws_split(const f32x4 r0, const f32x4 r1, u32x4 (&xb)[3]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 h = e < 2 ? r0 : r1;
    uint32_t q0, q1, q2;
    split_bf16x3(h[2 * (e & 1)], h[2 * (e & 1) + 1], q0, q1, q2);
    xb[0][e] = q0;
    xb[1][e] = q1;
    xb[2][e] = q2;
  }
}

// One k-step of a GEMM phase: acc[nb] += W[nb] . x for the NB column tiles of the slice at `pw` (= slice base +
// lane; piece (nb, plane) at (nb * 3 + plane) * 64).  The three weight fragments of tile nb + 1 are read while the
// six MFMAs of tile nb run.  (Tried: the next k-step's bf16 split interleaved two VALU per MFMA gap, and the
// landed-flag of the next slice requested one k-step early: no change -- the stream, not this wave's issue
// order, paces the slices)
template <int NB>
__device__ __forceinline__ void ws_kstep(const u32x4* __restrict__ pw, const u32x4 (&xb)[3], f32x4 (&acc)[NB]) {
  u32x4 wf[2][3];
#pragma unroll
  for (int p = 0; p < 3; ++p) wf[0][p] = pw[p * 64];
  __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (nb + 1 < NB) {
#pragma unroll
      for (int p = 0; p < 3; ++p) wf[(nb + 1) & 1][p] = pw[((nb + 1) * 3 + p) * 64];
    }
    acc[nb] = mfma_x6(wf[nb & 1], xb, acc[nb]);
    if (nb + 1 < NB) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
  }
}

// wait until *p >= v (wave-uniform); gives up -- and makes everybody give up -- after WS_SPIN_LIMIT polls
template <int SLEEP = 1>
__device__ __forceinline__ int ws_wait_ge(lds_vint* p, int v, lds_vint* ctrl) {
  int it = 0;
  while (__builtin_amdgcn_readfirstlane(*p) < v) {
    if (__builtin_amdgcn_readfirstlane(ctrl[CW_ERR]) != 0) break;
    if (++it > WS_SPIN_LIMIT) {
      ctrl[CW_ERR] = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(SLEEP);
  }
  asm volatile("" ::: "memory");
  return it;
}
// wait until the four per-wave counters at p[0..3] are all >= v
template <int SLEEP = 1>
__device__ __forceinline__ int ws_wait_all4(lds_vint* p, int v, lds_vint* ctrl) {
  int it = 0;
  for (;;) {
    const int a = p[0], b = p[1], c = p[2], d = p[3];
    if (__builtin_amdgcn_readfirstlane(min(min(a, b), min(c, d))) >= v) break;
    if (__builtin_amdgcn_readfirstlane(ctrl[CW_ERR]) != 0) break;
    if (++it > WS_SPIN_LIMIT) {
      ctrl[CW_ERR] = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(SLEEP);
  }
  asm volatile("" ::: "memory");
  return it;
}
__device__ __forceinline__ void ws_set(lds_vint* p, int v, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) *p = v;
  asm volatile("" ::: "memory");
}

struct WsTile {   // one tile's LDS
  float* slots;   // two stage slots of [16][WS_S] floats
  float* xin;     // [16][WS_S]: h (96) | features (F) | 1 | 0 ..   -- the B operand of P1 and P4
  int* csr;
  float* wv;
  __device__ __forceinline__ float* slot(int b) const { return slots + b * (16 * WS_S); }
};
__device__ __forceinline__ WsTile ws_tile(unsigned char* smem, int m) {
  unsigned char* tb = smem + WS_RING * WS_SLICE + m * WS_TILE_LDS;
  WsTile t;
  t.slots = reinterpret_cast<float*>(tb);
  t.xin = reinterpret_cast<float*>(tb + 2 * WS_STAGE);
  t.csr = reinterpret_cast<int*>(tb + 3 * WS_STAGE);
  t.wv = reinterpret_cast<float*>(tb + 3 * WS_STAGE + 2 * WS_CSR);
  return t;
}

// ------------------------------------------------------------------------------------------------------------
// M: the matrix chain of tile m
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ws_m_role(const ggnn_dec_cell_args& A, const int ts0, const int n_ts, const int m,
                                          unsigned char* __restrict__ smem, lds_vint* ctrl) {
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  const WsTile T = ws_tile(smem, m);
  const int n_dst = (int)A.n_dst, n_in = A.n_in, F = A.f_dst;
  const int K = 4 * n_in;
  int tau = 0, buf = 0;
  [[maybe_unused]] unsigned long long st_slice = 0, st_agg = 0, st_t = 0;
  GGNN_STAMP(0);
  GGNN_STAMP_VAL(10, 1);
  GGNN_STAMP_VAL(6, A.n_in);
  auto acquire = [&]() -> const u32x4* {
    st_slice += ws_wait_all4(ctrl + CW_LDONE, tau + 1, ctrl);   // (stamps count polls: reading the clock per slice would cost more than the slice)
    return reinterpret_cast<const u32x4*>(smem + buf * WS_SLICE) + lane;
  };
  auto release = [&]() {
    ++tau;
    ws_set(ctrl + CW_MFIN + m, tau, lane);
    buf = buf == WS_RING - 1 ? 0 : buf + 1;
  };

  for (int ti = 0; ti < n_ts; ++ti) {
    const int row0 = max(0, min(((ts0 + ti) * WS_TILES + m) * 16, n_dst - 16));
    const int node_m = min(row0 + lr, n_dst - 1);
    const int kk0 = ti * K;
    {
      // the tile's input rows -> LDS: lane l copies 16-byte pieces l, l + 64, .. of the 16 x 24 pieces of h
      for (int q = lane; q < 16 * 24; q += 64) {
        const int n = q / 24, c4 = (q - n * 24) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(A.h_dst + (int64_t)min(row0 + n, n_dst - 1) * A.ldh + c4);
        *reinterpret_cast<f32x4*>(&T.xin[n * WS_S + c4]) = v;
      }
      const int fn = lane >> 2, fq = (lane & 3) * 4;
      const float* xrow = A.x_dst + (int64_t)min(row0 + fn, n_dst - 1) * A.ldx;
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xv = xrow[min(fq + j, F - 1)];
        v[j] = fq + j < F ? xv : (fq + j == F ? 1.0f : 0.0f);
      }
      *reinterpret_cast<f32x4*>(&T.xin[fn * WS_S + C + fq]) = v;
      for (int e = 0; e < n_in; ++e) {
        const ggnn_dec_cell_sweep& Sw = A.in[e];
        int* __restrict__ rp = T.csr + e * (17 + WS_CW);
        if (lane < 17) rp[lane] = Sw.rowptr[min(row0 + lane, n_dst)];
        __builtin_amdgcn_wave_barrier();
        const int pbase = rp[0], e_last = (int)Sw.E - 1;
        if (Sw.E > 0) {
          for (int k = lane; k < WS_CW; k += 64) rp[17 + k] = Sw.col[min(pbase + k, e_last)];
        }
      }
      ws_set(ctrl + CW_CSRGEN + m, ti + 1, lane);
    }
    // B-fragment planes of k-step ks of [h | x | 1 | 0]: k-steps 0..2 are the h rows, k-step 3 the 16 feature slots
    // (k-groups 0 and 1; zeros behind)
    auto x_planes = [&](int ks, u32x4 (&out)[3]) __attribute__((always_inline)) {
      const float* fr = &T.xin[lr * WS_S + 32 * ks + 8 * (ks < 3 ? kq : (kq & 1))];
      f32x4 r0 = *reinterpret_cast<const f32x4*>(fr);
      f32x4 r1 = *reinterpret_cast<const f32x4*>(fr + 4);
      if (ks == 3 && kq >= 2) r0 = r1 = (f32x4){0.f, 0.f, 0.f, 0.f};
      ws_split(r0, r1, out);
    };
    // P1 of phase k: u_h | u4 of the tile -> stage slot k & 1
    auto p1 = [&](int k) __attribute__((always_inline)) {
      f32x4 u[7];
#pragma unroll
      for (int nb = 0; nb < 7; ++nb) u[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      u32x4 xb[2][3];
      x_planes(0, xb[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4* pw = acquire();
        ws_kstep<7>(pw, xb[ks & 1], u);
        if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);   // the next k-step's split
        release();
      }
      float* __restrict__ stage = T.slot(k & 1);
#pragma unroll
      for (int nb = 0; nb < 7; ++nb) *reinterpret_cast<f32x4*>(&stage[lr * WS_S + 16 * nb + 4 * kq]) = u[nb];
      ws_set(ctrl + CW_UGEN + 2 * m + (k & 1), kk0 + k + 1, lane);
    };

    f32x4 run[6], pre[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) run[ct] = pre[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    p1(0);
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const int gi = n_in == 2 ? k >> 1 : k, e = n_in == 2 ? (k & 1) : 0;
      const int g = gi == 1 ? 2 : (gi == 2 ? 1 : gi);
      if (k + 1 < K) p1(k + 1);
      if (e == 0) {
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const bool gate_ends = e == n_in - 1;
      st_t = GGNN_STAMP_NOW();
      ws_wait_ge<WS_LONG_SLEEP>(ctrl + CW_AGEN + 2 * m + (k & 1), kk0 + k + 1, ctrl);
      st_agg += GGNN_STAMP_NOW() - st_t;
      // P3: pre += lin_l2(e, g) . agg + (b_l2, w_edge) . (sum alpha, sum alpha a)
      {
        const float* __restrict__ stage = T.slot(k & 1);
        u32x4 xb[2][3];
        auto a_planes = [&](int ks, u32x4 (&out)[3]) __attribute__((always_inline)) {
          const float* sr = &stage[lr * WS_S + 32 * ks + 8 * kq];
          ws_split(*reinterpret_cast<const f32x4*>(sr), *reinterpret_cast<const f32x4*>(sr + 4), out);
        };
        a_planes(0, xb[0]);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const u32x4* pw = acquire();
          ws_kstep<6>(pw, xb[ks & 1], pre);
          if (ks + 1 < 3) a_planes(ks + 1, xb[(ks + 1) & 1]);
          release();
        }
        const float xt = kq < 2 ? stage[lr * WS_S + C + kq] : 0.f;
        const float* __restrict__ wt = A.w2_tail + (size_t)((g * n_in + e) * 6) * 64 + lane;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[ct * 64], xt, pre[ct], 0, 0, 0);
      }
      if (gate_ends) {
        // the old cell state (the forget gate's; other gates read the same rows and ignore them): in flight
        // during P4, live nowhere else
        f32x4 cin[6];
        {
          const float* crow = A.c_in + (int64_t)node_m * C + 4 * kq;
#pragma unroll
          for (int ct = 0; ct < 6; ++ct) cin[ct] = *reinterpret_cast<const f32x4*>(crow + 16 * ct);
        }
        // P4: the summed skip term + gate bias
        {
          u32x4 xb[2][3];
          x_planes(0, xb[0]);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const u32x4* pw = acquire();
            ws_kstep<6>(pw, xb[ks & 1], pre);
            if (ks + 1 < 4) x_planes(ks + 1, xb[(ks + 1) & 1]);
            release();
          }
        }
        // LSTM update, gate by gate (heteropgclstm.py:140-146)
        if (gi == 0) {
#pragma unroll
          for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) run[ct][r] = sigmoidf_(pre[ct][r]);
        } else if (gi == 1) {
#pragma unroll
          for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) run[ct][r] *= tanhf_(pre[ct][r]);
        } else if (gi == 2) {
#pragma unroll
          for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) run[ct][r] = sigmoidf_(pre[ct][r]) * cin[ct][r] + run[ct][r];
          float* crow = A.c_out + (int64_t)node_m * C + 4 * kq;
#pragma unroll
          for (int ct = 0; ct < 6; ++ct) *reinterpret_cast<f32x4*>(crow + 16 * ct) = run[ct];
        } else {
#pragma unroll
          for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) pre[ct][r] = sigmoidf_(pre[ct][r]) * tanhf_(run[ct][r]);
          float* hrow = A.h_out + (int64_t)node_m * C + 4 * kq;
#pragma unroll
          for (int ct = 0; ct < 6; ++ct) *reinterpret_cast<f32x4*>(hrow + 16 * ct) = pre[ct];
        }
      }
    }
  }
  GGNN_STAMP_VAL(4, st_slice);
  GGNN_STAMP_VAL(5, st_agg);
  GGNN_STAMP(16);
  if (__builtin_amdgcn_readfirstlane(ctrl[CW_ERR]) != 0 && m == 0 && lane == 0) A.h_out[0] = __builtin_nanf("");
}

// ------------------------------------------------------------------------------------------------------------
// S: the sweeps of tile m
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ws_s_role(const ggnn_dec_cell_args& A, const int ts0, const int n_ts, const int m,
                                          unsigned char* __restrict__ smem, lds_vint* ctrl) {
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  const int ch = 3 * lr;
  constexpr int CH2 = C / 2;
  const WsTile T = ws_tile(smem, m);
  const int n_dst = (int)A.n_dst, n_in = A.n_in;
  const int K = 4 * n_in;
  [[maybe_unused]] unsigned long long st_u = 0, st_t = 0;
  GGNN_STAMP(0);
  GGNN_STAMP_VAL(10, 2);
  GGNN_STAMP_VAL(6, A.n_in);

  for (int ti = 0; ti < n_ts; ++ti) {
    const int row0 = max(0, min(((ts0 + ti) * WS_TILES + m) * 16, n_dst - 16));
    const int kk0 = ti * K;
    ws_wait_ge(ctrl + CW_CSRGEN + m, ti + 1, ctrl);
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const int gi = n_in == 2 ? k >> 1 : k, e = n_in == 2 ? (k & 1) : 0;
      const int g = gi == 1 ? 2 : (gi == 2 ? 1 : gi);
      const ggnn_dec_cell_sweep& Sw = A.in[e];
      float* __restrict__ stage = T.slot(k & 1);
      lds_vint* uflag = ctrl + CW_UGEN + 2 * m + (k & 1);
      const int target = kk0 + k + 1;

      // the phase's lin_edge weights of this lane's six channels -> LDS (read back per fold: registers are what
      // this wave is short of); the four DPP rows hold the same 16 x 18 values
      float* __restrict__ wvl = T.wv + lr * 20;
      {
        const float* __restrict__ ep = Sw.edge_params + g * GGNN_EDGE_PARAM_ROWS * C;
        if (kq == 0) {
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) {
            const float* w = ep + ch + (cc < 3 ? cc : CH2 + cc - 3);
            wvl[3 * cc] = w[0];
            wvl[3 * cc + 1] = w[C];
            wvl[3 * cc + 2] = w[2 * C];
          }
        }
      }
      const float* __restrict__ vbase = Sw.v_src + Sw.v_off + g * C + ch;
      const float* __restrict__ hbase = Sw.h_src + ch;
      const float* __restrict__ einfo = Sw.einfo;
      const uint32_t ldv = (uint32_t)Sw.ldv, ldh = (uint32_t)Sw.ldh_src;
      const int* __restrict__ rp = T.csr + e * (17 + WS_CW);
      const int* __restrict__ colw = rp + 17;
      const int pbase = rp[0], e_last = max((int)Sw.E - 1, 0);
      const bool has_edges = Sw.E > 0;
      struct Unit {     // the gathered operands of <= 3 in-edges of one destination row
        f3 hh[GGNN_UNIT_EDGES][2], vv[GGNN_UNIT_EDGES][2];
        float x4[GGNN_UNIT_EDGES];
        f3 ed[GGNN_UNIT_EDGES];   // reloc_e (the edge length a_e is slot 13 of x4: lane 13 sums alpha a_e)
      };
      const bool in_window = __builtin_amdgcn_readfirstlane(rp[16] - pbase) <= WS_CW;
      // unconditional (clamped) loads, back to back; the choice between the LDS index window and the global index
      // array is wave-uniform per tile (a load under `if` would drag a full wait to the branch merge)
      auto gather = [&](int p, Unit& U, auto window_tag) __attribute__((always_inline)) {
        constexpr bool WINDOW = decltype(window_tag)::value;
#pragma unroll
        for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
          const int pt = min(p + t, e_last);
          int j;
          if constexpr (WINDOW) j = colw[min(max(pt - pbase, 0), WS_CW - 1)];
          else j = has_edges ? Sw.col[pt] : 0;
          if (!has_edges) j = 0;
          U.hh[t][0] = ld3(hbase + (uint32_t)j * ldh);
          U.hh[t][1] = ld3(hbase + (uint32_t)j * ldh + CH2);
          U.x4[t] = einfo[(uint32_t)pt * GGNN_EINFO_ROW + lr];
          U.ed[t] = ld3(einfo + (uint32_t)pt * GGNN_EINFO_ROW + 16);
          U.vv[t][0] = ld3(vbase + (uint32_t)j * ldv);
          U.vv[t][1] = ld3(vbase + (uint32_t)j * ldv + CH2);
        }
      };
      struct Row {      // a destination row being folded (its u stays in the stage)
        float mx, den, sae, acc[6];
        int p, pe, n;
      };
      auto open_row = [&](Row& r, int n) __attribute__((always_inline)) {
        const int nl = min(row0 + n, n_dst - 1) - row0;   // (n_dst < 16: rows past the end repeat the last node)
        r.n = n;
        r.p = rp[nl];
        r.pe = rp[nl + 1];
        r.mx = -INFINITY;
        r.den = r.sae = 0.f;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) r.acc[cc] = 0.f;
      };
      auto fold = [&](Row& r, const Unit& Uc) __attribute__((always_inline)) {
        // explicit fma, contraction off: identical bits to dec_cell.hip's sweep and between the two variants
#pragma clang fp contract(off)
        const int nact = min(max(r.pe - r.p, 0), GGNN_UNIT_EDGES);
        if (nact > 0) {
          const float* __restrict__ su = stage + r.n * WS_S;
          float sc[GGNN_UNIT_EDGES];
          float mnew = r.mx;
          {
            const f3 uh0 = {su[ch], su[ch + 1], su[ch + 2]};
            const f3 uh1 = {su[CH2 + ch], su[CH2 + ch + 1], su[CH2 + ch + 2]};
            const float u4 = su[C + lr];
#pragma unroll
            for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
              sc[t] = -INFINITY;
              if (t < nact) {
                float part = u4 * Uc.x4[t];
                part = __builtin_fmaf(uh0.x, Uc.hh[t][0].x, part);
                part = __builtin_fmaf(uh0.y, Uc.hh[t][0].y, part);
                part = __builtin_fmaf(uh0.z, Uc.hh[t][0].z, part);
                part = __builtin_fmaf(uh1.x, Uc.hh[t][1].x, part);
                part = __builtin_fmaf(uh1.y, Uc.hh[t][1].y, part);
                part = __builtin_fmaf(uh1.z, Uc.hh[t][1].z, part);
                sc[t] = row_sum(part);   // 1 / sqrt(96) is folded into u
                mnew = fmaxf(mnew, sc[t]);
              }
            }
          }
          f3 wv[6];
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) wv[cc] = {wvl[3 * cc], wvl[3 * cc + 1], wvl[3 * cc + 2]};
          const float scale = __expf(r.mx - mnew);   // exp(-inf) = 0 on a row's first unit
          r.den = r.den * scale;
          r.sae = r.sae * scale;
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) r.acc[cc] = r.acc[cc] * scale;
#pragma unroll
          for (int t = 0; t < GGNN_UNIT_EDGES; ++t) {
            if (t < nact) {
              const float rx = Uc.ed[t].x, ry = Uc.ed[t].y, rz = Uc.ed[t].z;
              const float pw_ = __expf(sc[t] - mnew);
              r.den = r.den + pw_;
              r.sae = __builtin_fmaf(pw_, Uc.x4[t], r.sae);
              const float v[6] = {Uc.vv[t][0].x, Uc.vv[t][0].y, Uc.vv[t][0].z, Uc.vv[t][1].x, Uc.vv[t][1].y, Uc.vv[t][1].z};
#pragma unroll
              for (int cc = 0; cc < 6; ++cc) {
                const float val = __builtin_fmaf(wv[cc].z, rz, __builtin_fmaf(wv[cc].y, ry, __builtin_fmaf(wv[cc].x, rx, v[cc])));
                r.acc[cc] = __builtin_fmaf(pw_, fmaxf(val, 0.f), r.acc[cc]);
              }
            }
          }
          r.mx = mnew;
        }
        r.p += GGNN_UNIT_EDGES;
      };
      auto close_row = [&](const Row& r) __attribute__((always_inline)) {   // the row's aggregate replaces its u
#pragma clang fp contract(off)
        const float inv = 1.0f / (r.den + 1e-16f);   // PyG softmax denominator
        float* __restrict__ so = stage + r.n * WS_S;
        so[ch] = r.acc[0] * inv;
        so[ch + 1] = r.acc[1] * inv;
        so[ch + 2] = r.acc[2] * inv;
        so[CH2 + ch] = r.acc[3] * inv;
        so[CH2 + ch + 1] = r.acc[4] * inv;
        so[CH2 + ch + 2] = r.acc[5] * inv;
        if (lr == 0) so[C] = r.den * inv;
        if (lr == 13) so[C + 1] = r.sae * inv;
      };
      // Two rows per 16-lane group in flight (tile rows 8 half + kq and 8 half + 4 + kq): the gathers of both are
      // requested back to back before either is folded -- two memory round trips per phase.  The first pair of a
      // phase is requested BEFORE the wait for u (it needs the CSR window only).
      auto sweep = [&](auto window_tag) __attribute__((always_inline)) {
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
          Row ra, rb;
          Unit ua, ub;
          open_row(ra, 8 * half + kq);
          open_row(rb, 8 * half + 4 + kq);
          gather(ra.p, ua, window_tag);
          gather(rb.p, ub, window_tag);
          if (half == 0) {
            st_t = GGNN_STAMP_NOW();
            ws_wait_ge(uflag, target, ctrl);
            st_u += GGNN_STAMP_NOW() - st_t;
          }
          for (;;) {
            fold(ra, ua);
            fold(rb, ub);
            if (__builtin_amdgcn_ballot_w64(ra.p < ra.pe || rb.p < rb.pe) == 0) break;
            gather(ra.p, ua, window_tag);
            gather(rb.p, ub, window_tag);
          }
          close_row(ra);
          close_row(rb);
        }
      };
#ifdef WS_SKIP_SWEEP   // timing experiment only (wrong results): the floor set by the matrix and stream waves
      ws_wait_ge(uflag, target, ctrl);
#else
      if (in_window) sweep(std::true_type{});
      else sweep(std::false_type{});
#endif
      ws_set(ctrl + CW_AGEN + 2 * m + (k & 1), target, lane);
    }
  }
  GGNN_STAMP_VAL(4, st_u);
  GGNN_STAMP(16);
}

// ------------------------------------------------------------------------------------------------------------
// L: the weight stream
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ws_l_role(const ggnn_dec_cell_args& A, const int n_ts, const int j,
                                          unsigned char* __restrict__ smem, lds_vint* ctrl) {
  const int lane = threadIdx.x & 63;
  const int n_in = A.n_in, K = 4 * n_in, per_gate = 7 * n_in + 4;
  const unsigned char* __restrict__ wsrc = reinterpret_cast<const unsigned char*>(A.wstream) + lane * 16;
  const uint32_t slice_lds = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)));
  int tau = 0, buf = 0;
  [[maybe_unused]] unsigned long long st_free = 0, st_land = 0, st_t = 0;
  GGNN_STAMP(0);
  GGNN_STAMP_VAL(10, 3);
  GGNN_STAMP_VAL(6, A.n_in);
  // slice `sid` of the stream (np pieces) is the tau-th one consumed
  auto emit = [&](int sid, int np) {
    if (tau >= WS_RING) st_free += ws_wait_all4<WS_LONG_SLEEP>(ctrl + CW_MFIN, tau - WS_RING + 1, ctrl);
    const unsigned char* src = wsrc + (size_t)sid * WS_SLICE;
    const uint32_t dst = slice_lds + buf * WS_SLICE;
    int cnt = 0;
    for (int p = j; p < np; p += WS_TILES, ++cnt) ws_dma16(src + p * 1024, dst + p * 1024);
    if (tau >= 1) {   // the pieces of the slice before this one (issued earlier, complete in order)
      if (cnt >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (cnt == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      ws_set(ctrl + CW_LDONE + j, tau, lane);   // slices 0 .. tau - 1
    }
    ++tau;
    buf = buf == WS_RING - 1 ? 0 : buf + 1;
  };
  auto p1 = [&](int k) {
    const int gi = n_in == 2 ? k >> 1 : k, e = n_in == 2 ? (k & 1) : 0;
    for (int s = 0; s < 4; ++s) emit(gi * per_gate + 7 * e + s, 21);
  };
  for (int ti = 0; ti < n_ts; ++ti) {
    p1(0);
    for (int k = 0; k < K; ++k) {
      const int gi = n_in == 2 ? k >> 1 : k, e = n_in == 2 ? (k & 1) : 0;
      if (k + 1 < K) p1(k + 1);
      for (int s = 0; s < 3; ++s) emit(gi * per_gate + 7 * e + 4 + s, 18);
      if (e == n_in - 1)
        for (int s = 0; s < 4; ++s) emit(gi * per_gate + 7 * n_in + s, 18);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ws_set(ctrl + CW_LDONE + j, tau, lane);
  GGNN_STAMP_VAL(4, st_free);
  GGNN_STAMP_VAL(5, st_land);
  GGNN_STAMP(16);
}

__global__ __launch_bounds__(WS_WAVES * 64, 3) void dec_cell_ws_kernel(const WsBatch B) {
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[WS_LDS];
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const ggnn_dec_cell_args& A = B.a[k];
  const int nwg = B.wg_off[k + 1] - B.wg_off[k];
  const int wg = xcd_remap((int)blockIdx.x - B.wg_off[k], nwg);
  const int ts_total = (int)((A.n_dst + 16 * WS_TILES - 1) / (16 * WS_TILES));
  const int ts0 = wg * B.tsw, n_ts = min(B.tsw, ts_total - ts0);
  lds_vint* ctrl = (lds_vint*)(s_raw + WS_RING * WS_SLICE + WS_TILES * WS_TILE_LDS);
  if (threadIdx.x < WS_CTRL_INTS) ctrl[threadIdx.x] = 0;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#ifndef WS_NO_M
  if (wave < WS_TILES) ws_m_role(A, ts0, n_ts, wave, s_raw, ctrl);
#endif
#ifndef WS_NO_S
  if (wave >= WS_TILES && wave < 2 * WS_TILES) ws_s_role(A, ts0, n_ts, wave - WS_TILES, s_raw, ctrl);
#endif
  if (wave >= 2 * WS_TILES) ws_l_role(A, n_ts, wave - 2 * WS_TILES, s_raw, ctrl);
}

// the arguments were validated by ggnn_decoder_cell_batch (dec_cell.hip)
int dec_cell_ws_launch(const ggnn_dec_cell_args* args, int n_problems, hipStream_t stream) {
  WsBatch B;
  B.n = n_problems;
  // tile sets per workgroup: the fewest that put the launch on the chip in one round (one workgroup per CU)
  int tsw = 1;
  for (;; ++tsw) {
    int64_t wgs = 0;
    for (int k = 0; k < n_problems; ++k) {
      const int64_t ts = (args[k].n_dst + 16 * WS_TILES - 1) / (16 * WS_TILES);
      wgs += (ts + tsw - 1) / tsw;
    }
    if (wgs <= num_cu() || tsw >= 4) break;
  }
  static const int tsw_env = [] {
    const char* e = getenv("GGNN_DC_TSW");   // development knob
    return e ? atoi(e) : 0;
  }();
  if (tsw_env > 0) tsw = tsw_env;
  B.tsw = tsw;
  B.wg_off[0] = 0;
  for (int k = 0; k < WS_MAX_PROBLEMS; ++k) {
    B.a[k] = args[k < n_problems ? k : 0];
    if (k >= n_problems) {
      B.wg_off[k + 1] = B.wg_off[k];
      continue;
    }
    const int64_t ts = (args[k].n_dst + 16 * WS_TILES - 1) / (16 * WS_TILES);
    const int64_t wgs = (ts + tsw - 1) / tsw;
    if (B.wg_off[k] + wgs >= INT32_MAX) return GGNN_EINVAL;
    B.wg_off[k + 1] = B.wg_off[k] + (int)wgs;
  }
  hipLaunchKernelGGL(dec_cell_ws_kernel, dim3((unsigned)B.wg_off[WS_MAX_PROBLEMS]), dim3(WS_WAVES * 64), 0, stream, B);
  return launch_status();
}

}  // namespace ggnn
