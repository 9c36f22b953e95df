// Encoder HeteroPGCLSTM cell (h = c = 0: models.py:422-424, heteropgclstm.py:148-183 without the forget gate) with
// EVERYTHING of a destination node in ONE kernel and ONE launch (ggnn_encoder_cell_batch, include/ggnn.h): the
// score tails u4, the periodic-boundary GAT sweep of every incoming edge type (PeriodConv.message,
// periodGATconv.py:204-236), lin_l2 + the value-side lin_edge term, HeteroConv's sum over the edge types, the
// summed skip term and the LSTM update.  Round 4: replaces the round-2 pair enc_cell_kernel + enc_lstm_kernel
// (a workgroup per (edge type, gate), 57.6 MB of partial pre-activations per model written by one launch and
// summed by the next) and the slim encoder projection (u4 tails through memory).
//
// With h = 0 a PeriodConv of the encoder sees nothing but 16-float rows: the destination's features and, per
// in-edge, the record ggnn_edge_prepare made of the source's features (reloc, x_j[3:F], 1, edge length).  So the
// whole cell is matrix-core work on operands a lane already holds -- the layouts are those of the round-4 decoder
// experiment (profiles/r4_dec_cell_ablations.txt), which lost to its gathers; the encoder has none:
//   * a wave owns a 16-node tile; lane (node lr = l & 15, k-group kq = l >> 4) keeps four of the 16 slots of its
//     node's feature row, and of the record of its node's t-th in-edge (t = 0..2: one "unit"), as the two fp16 planes
//     of a B fragment (k slot 8 kq + j = row slot 4 kq + j, j < 4) -- loaded and split ONCE per tile, reused by all
//     three gates;
//   * per (edge type e, gate g) ONE weight slice holds [value rows 0..95 | u4 rows 96..111] x 16 slots: the u4 column
//     tile against the tile's feature planes gives u4[node] in lane (node, kq) = the A fragment of the score MFMA
//     against the edge planes -> D[node][edge], the wanted entries on the diagonal (ds_bpermute); the six value
//     column tiles against the SAME edge planes give relu's argument W_value x~_e + b_value in lane (edge = node lr,
//     channels 16 nb + 4 kq ..+3); softmax (online over units for rows of more than three in-edges), relu,
//     alpha-weighted sum and normalisation stay in that layout, which -- lin_l2's columns permuted on the host -- is
//     the B fragment of the three lin_l2 k-steps that follow; the pre-activations of a gate accumulate over the
//     edge types in registers, the skip term is one more k-step against the feature planes, the LSTM update is
//     folded in gate by gate (i, c~, o);
//   * weights: 3 (4 n_in + 1) slices of pre-split fp16 planes per destination type in dec_cell.hip's slice image,
//     streamed by LDS-DMA in GROUPS: the 4 slices of a (e, g) pass (5 with the gate's skip slice behind its last edge
//     type) are requested together, one whole pass ahead, into one of two 70 KB buffers shared by the workgroup's
//     waves -- one barrier per pass (9 per joint tile, where a barrier per slice would be 27) and ~4 us for a group
//     to land.  The kernel's only other LDS traffic is ds_bpermute.
// Arithmetic: two fp16 pieces / three products per fp32 operand (common.h), the rank-1 columns (b_l2, w_edge) on one
// exact fp32 MFMA; explicit fmas with contraction off wherever a row could be computed twice (ragged last tile).
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "cell_common.h"
#define GGNN_STAMP_SUFFIX _enc
#include "stamps.h"

namespace ggnn {

#ifndef EC_WAVES_
#define EC_WAVES_ 8
#endif
constexpr int EC_WAVES = EC_WAVES_;                 // 16-node tiles per workgroup
constexpr int EC_MAX_PROBLEMS = 4;
constexpr int EC_SLICE = GGNN_DC_SLICE_BYTES;       // 14 pieces of 1 KB
constexpr int EC_NPA = 7 * DC_PL, EC_NP3 = 6 * DC_PL;   // pieces of a value | u4 slice / of a lin_l2 or skip slice
constexpr int EC_GROUP = 5;                         // slices of a group: value | u4, 3 x lin_l2, (skip)
constexpr int EC_LDS = 2 * EC_GROUP * EC_SLICE;     // 143 360 B
static_assert(EC_LDS <= 160 * 1024, "LDS");
constexpr int EU = GGNN_UNIT_EDGES;                 // edge tiles (t-th in-edge of every node) handled together

struct EncCellBatch {
  ggnn_enc_cell_args a[EC_MAX_PROBLEMS];
  int wg_off[EC_MAX_PROBLEMS + 1];
  int n;
};

__device__ __forceinline__ void enc_cell_body(const ggnn_enc_cell_args& A, const int tileset,
                                              unsigned char* __restrict__ smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;   // node lr of the tile; k-group of a fragment = output rows 4 kq .. of a D tile
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  const int n_dst = (int)A.n_dst, n_in = A.n_in, F = A.f_dst;
  // a ragged last tile slides back over rows the previous tile also produces (identical duplicate results);
  // tiles past the end (a workgroup's surplus waves) repeat the last one: every wave runs the whole program,
  // so the workgroup barriers of the weight stream need no special case
  const int row0 = max(0, min((tileset * EC_WAVES + wave) * 16, n_dst - 16));
  const int node_m = min(row0 + lr, n_dst - 1);    // this lane's node (n_dst < 16: the last node repeats)

  // ---- the weight stream: group q = (gate g, edge type e) -> buffer q & 1, requested one group ahead ----
  const unsigned char* __restrict__ wsrc = reinterpret_cast<const unsigned char*>(A.wstream) + lane * 16;
  const uint32_t slice_lds =
      __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)));
  const int per_gate = 4 * n_in + 1, n_groups = 3 * n_in;
  int q_cur = 0;
  [[maybe_unused]] unsigned long long st_wait = 0, st_a = 0, st_p3 = 0, st_p4 = 0, st_lstm = 0;
  GGNN_STAMP(0);
#ifndef EC_DMA_PARTS
#define EC_DMA_PARTS 1
#endif
  // `part` of EC_DMA_PARTS: the wave's pieces of a group are requested in that many instalments through the pass that
  // precedes it (development, profiles/r6_enc_cell_experiments.txt; 1 = all of them at the top of the pass)
  auto dma_group = [&](int q, int part = 0) {
    if (q >= n_groups) return;
    const int g = q / n_in, e = q - g * n_in;
    const int s0 = g * per_gate + 4 * e;                  // first slice of the group in the stream
    const int ns = 4 + (e == n_in - 1 ? 1 : 0);           // ... and their number (the gate's skip slice rides with its last pass)
    const int np = EC_NPA + (ns - 1) * EC_NP3;
    const unsigned char* src = wsrc + (size_t)s0 * EC_SLICE;
    const uint32_t dst = slice_lds + (q & 1) * (EC_GROUP * EC_SLICE);
    constexpr int per_part = (8 + EC_DMA_PARTS - 1) / EC_DMA_PARTS;   // (a wave has at most 8 pieces of a 62-piece group)
    const int p_lo = EC_DMA_PARTS == 1 ? wave : wave + EC_WAVES * per_part * part;
    const int p_hi = EC_DMA_PARTS == 1 ? np : min(np, wave + EC_WAVES * per_part * (part + 1));
    for (int p = p_lo; p < p_hi; p += EC_WAVES) {
      // piece p of the group: the 14 pieces of its first slice, then 12 per slice (the last 2 KB of those are unused)
      const int sl = p < EC_NPA ? 0 : 1 + (p - EC_NPA) / EC_NP3;
      const int off = sl * EC_SLICE + (p < EC_NPA ? p : (p - EC_NPA) % EC_NP3) * 1024;
      dc_dma16(src + off, dst + off);
    }
  };
  auto begin_group = [&]() -> const u32x4* {   // the group about to be used landed at the previous end_group
    dma_group(q_cur + 1, 0);
    return reinterpret_cast<const u32x4*>(smem + (q_cur & 1) * (EC_GROUP * EC_SLICE)) + lane;
  };
  auto end_group = [&]() {
    [[maybe_unused]] const unsigned long long w0 = GGNN_STAMP_NOW();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next group are in LDS
    __syncthreads();                                   // ... everybody's are, and nobody reads the old one any more
    st_wait += GGNN_STAMP_NOW() - w0;
    ++q_cur;
  };
  for (int k = 0; k < EC_DMA_PARTS; ++k) dma_group(0, k);
#ifdef EC_EXP_PRIO
  // development (profiles/r6_enc_cell_experiments.txt): one static priority for the second-dispatched half of the workgroup
  if (wave >= EC_WAVES / 2) __builtin_amdgcn_s_setprio(1);
#endif

  uint32_t amax = 0u;   // largest magnitude this lane has split into fp16 pieces, as bits (dc_track: NaN and inf stay visible)

  // ---- tile prologue: this node's feature slots and the records of its first in-edges, as B-fragment planes ----
  // slots of a feature row: x_0 .. x_{F-1}, 0 .., 1 at 12 (bias), 0 ..; of an edge record: ggnn_edge_prepare's
  // (reloc, x_j[3:F_src), 0 .., 1 at 12, edge length at 13, ..): lane (., kq) holds slots 4 kq ..+3 in k slots 8 kq ..+3
  u32x4 xs[DC_PL];
  {
    const float* xrow = A.x_dst + (int64_t)node_m * A.ldx;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sl = 4 * kq + j;
      const float xv = xrow[min(sl, F - 1)];   // unconditional (clamped) load
      v[j] = sl < F ? xv : (sl == 12 ? 1.0f : 0.0f);
    }
    dc_split_half(v, xs, amax);
  }
  // First units of both incoming edge types (or, with ONE incoming edge type, its first two units): a joint's three
  // in-edges per edge type and a grain's first six never touch memory again.
  int p_first[2], p_end[2];
  uint32_t xr[2][EU][DC_PL][2];   // [set][edge tile][plane][packed pair]: the two non-zero dwords of a fragment
  float ael[2][EU];               // record slot 4 kq + 1: the edge length in k-group 3
  auto load_unit = [&](const ggnn_enc_cell_sweep& Sw, int p_, uint32_t (&r)[EU][DC_PL][2], float (&al)[EU]) __attribute__((always_inline)) {
    const int e_last = max((int)Sw.E - 1, 0);
#pragma unroll
    for (int t = 0; t < EU; ++t) {
      const f32x4 rec = ld16f(Sw.einfo + (uint32_t)min(p_ + t, e_last) * GGNN_EINFO_ROW + 4 * kq);
      u32x4 pl[DC_PL];
      dc_split_half(rec, pl, amax);
#pragma unroll
      for (int q = 0; q < DC_PL; ++q) {
        r[t][q][0] = pl[q][0];
        r[t][q][1] = pl[q][1];
      }
      al[t] = rec[1];
    }
  };
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const ggnn_enc_cell_sweep& Sw = A.in[e < n_in ? e : 0];
    p_first[e] = Sw.rowptr[node_m];
    p_end[e] = Sw.rowptr[node_m + 1];
    load_unit(Sw, p_first[e] + (e < n_in ? 0 : EU), xr[e], ael[e]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // group 0 is in LDS
  GGNN_STAMP(1);

  // the diagonal of a score tile D[node][edge]: node lr's entry sits in lane (lr, kq = lr >> 2), register lr & 3
  const int diag_addr = 4 * (16 * (lr >> 2) + lr), diag_sub = lr & 3;

  f32x4 run[6];   // the LSTM update as the gates arrive: sig(i) -> sig(i) tanh(c~) = c' -> (h')
#pragma unroll
  for (int ct = 0; ct < 6; ++ct) run[ct] = zero4;
#pragma unroll 1
  for (int g = 0; g < 3; ++g) {   // gates i, c~, o (the encoder's weights are indexed in that order)
    f32x4 pre[6];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) pre[ct] = zero4;

#pragma unroll 1
    for (int e = 0; e < n_in; ++e) {
      const ggnn_enc_cell_sweep& Sw = A.in[e];
      [[maybe_unused]] const unsigned long long t_a = GGNN_STAMP_NOW();
      const int pe = e == 0 ? p_end[0] : p_end[1];
      int p = e == 0 ? p_first[0] : p_first[1];
      float wtail[6];   // the (b_l2, w_edge) tail of lin_l2: requested here, used behind the three lin_l2 k-steps
      {
        const float* __restrict__ wt = A.w2_tail + (size_t)((g * n_in + e) * 6) * 64 + lane;
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) wtail[ct] = wt[ct * 64];
      }
      const u32x4* pw = begin_group();

      // ================= slice A: u4 of the tile, scores, softmax, values of (e, g) =================
      float mx = -INFINITY, den = 0.f, sae = 0.f;
      f32x4 acc[6];
#pragma unroll
      for (int nb = 0; nb < 6; ++nb) acc[nb] = zero4;
      {
        u32x4 ua[DC_PL];   // u4[node][4 kq ..+3] as an A fragment (1 / sqrt(96) is folded into the weights)
        {
          u32x4 wf[DC_PL];
#pragma unroll
          for (int q = 0; q < DC_PL; ++q) wf[q] = pw[(6 * DC_PL + q) * 64];
          DcAcc U;
          U.zero();
          mfma_x3h(wf, xs, U.m, U.c);
          dc_split_half(U.value(), ua, amax);
        }
        // one unit = the t-th in-edges (t = 0..2 from p) of the tile's 16 nodes, records as planes `r`
        auto unit = [&](const uint32_t (&r)[EU][DC_PL][2], const float (&al)[EU], const int nact, auto first_tag) __attribute__((always_inline)) {
#pragma clang fp contract(off)
          constexpr bool FIRST = decltype(first_tag)::value;
          u32x4 xb[EU][DC_PL];
          float s[EU];
#pragma unroll
          for (int t = 0; t < EU; ++t) {
#pragma unroll
            for (int q = 0; q < DC_PL; ++q) xb[t][q] = (u32x4){r[t][q][0], r[t][q][1], 0u, 0u};
            DcAcc S;
            S.zero();
            mfma_x3h(ua, xb[t], S.m, S.c);
            const f32x4 sv = S.value();
            const float sel = diag_sub == 0 ? sv[0] : (diag_sub == 1 ? sv[1] : (diag_sub == 2 ? sv[2] : sv[3]));
            s[t] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(diag_addr, __builtin_bit_cast(int, sel)));
          }
          // online-max softmax over the units of a row (PyG's softmax: exp(s - max) / (sum + 1e-16))
          float mnew = mx;
#pragma unroll
          for (int t = 0; t < EU; ++t) mnew = t < nact ? fmaxf(mnew, s[t]) : mnew;
          if constexpr (!FIRST) {
            const float scale = nact > 0 ? __expf(mx - mnew) : 1.0f;
            den = den * scale;
            sae = sae * scale;
#pragma unroll
            for (int nb = 0; nb < 6; ++nb) acc[nb] = acc[nb] * scale;
          }
          float pw_[EU];
#pragma unroll
          for (int t = 0; t < EU; ++t) {
            pw_[t] = t < nact ? __expf(s[t] - mnew) : 0.f;
            den = den + pw_[t];
            sae = __builtin_fmaf(pw_[t], al[t], sae);   // k-group 3 holds the edge length: sum alpha a_e there
          }
          mx = mnew;
          // values: relu(W_value x~_e + b_value) in lane (edge = node lr, channels 16 nb + 4 kq ..+3)
#pragma unroll
          for (int nb = 0; nb < 6; ++nb) {
            u32x4 wf[DC_PL];
#pragma unroll
            for (int q = 0; q < DC_PL; ++q) wf[q] = pw[(nb * DC_PL + q) * 64];
#pragma unroll
            for (int t = 0; t < EU; ++t) {
              DcAcc Vv;
              Vv.zero();
              mfma_x3h(wf, xb[t], Vv.m, Vv.c);
              const f32x4 val = Vv.value();
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[nb][i] = __builtin_fmaf(pw_[t], fmaxf(val[i], 0.f), acc[nb][i]);
            }
          }
        };
        if (e == 0) unit(xr[0], ael[0], min(max(pe - p, 0), EU), std::true_type{});
        else unit(xr[1], ael[1], min(max(pe - p, 0), EU), std::true_type{});
        p += EU;
        // rows of more than three in-edges (grains; hubs): further units.  With one incoming edge type the second
        // unit's records are in the registers the second edge type does not need.
        bool second = n_in == 1;
        while (__builtin_amdgcn_ballot_w64(p < pe) != 0) {
          if (second) {
            unit(xr[1], ael[1], min(max(pe - p, 0), EU), std::false_type{});
          } else {
            uint32_t r2[EU][DC_PL][2];
            float al2[EU];
            load_unit(Sw, p, r2, al2);
            unit(r2, al2, min(max(pe - p, 0), EU), std::false_type{});
          }
          second = false;
          p += EU;
        }
      }
      [[maybe_unused]] const unsigned long long t_b = GGNN_STAMP_NOW();
      if (EC_DMA_PARTS > 1) dma_group(q_cur + 1, 1);

      // ================= P3: pre += lin_l2(e, g) . agg + (b_l2, w_edge) . (sum alpha, sum alpha a) =================
      {
        u32x4 ab[3][DC_PL];
        float xt;
        {
#pragma clang fp contract(off)
          const float inv = 1.0f / (den + 1e-16f);   // PyG softmax denominator
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) dc_split(acc[2 * ks] * inv, acc[2 * ks + 1] * inv, ab[ks], amax);
          xt = kq == 0 ? den * inv : (kq == 3 ? sae * inv : 0.f);
        }
        DcAcc part[6];   // this phase's contribution (the fp16 split's cross terms live only here)
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) part[ct].zero();
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          dc_kstep<6>(pw + (1 + ks) * (EC_SLICE / 16), ab[ks], part);
          if (EC_DMA_PARTS > 2 && ks + 2 < EC_DMA_PARTS) dma_group(q_cur + 1, ks + 2);
        }
        // ... and, behind the gate's last edge type, P4: the summed skip term + gate bias (16 feature slots: one k-step)
        if (e == n_in - 1) dc_kstep<6>(pw + 4 * (EC_SLICE / 16), xs, part);
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] += part[ct].value();
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) pre[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wtail[ct], xt, pre[ct], 0, 0, 0);
      }
      end_group();
      st_a += t_b - t_a;
      st_p3 += GGNN_STAMP_NOW() - t_b;
    }
    [[maybe_unused]] const unsigned long long t_f = GGNN_STAMP_NOW();

    // ================= LSTM update with c = 0 (heteropgclstm.py:140-146), folded in gate by gate =================
    if (g == 0) {
#pragma unroll
      for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) run[ct][r] = sigmoidf_(pre[ct][r]);
    } else if (g == 1) {
#pragma unroll
      for (int ct = 0; ct < 6; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) run[ct][r] *= tanhf_(pre[ct][r]);
      float* crow = A.c_out + (int64_t)node_m * C + 4 * kq;
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) *reinterpret_cast<f32x4*>(crow + 16 * ct) = run[ct];
    } else {
      float* hrow = A.h_out + (int64_t)node_m * C + 4 * kq;
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) {
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = sigmoidf_(pre[ct][r]) * tanhf_(run[ct][r]);
        *reinterpret_cast<f32x4*>(hrow + 16 * ct) = h;
      }
    }
    st_lstm += GGNN_STAMP_NOW() - t_f;
  }
  // range flag: an operand at or beyond fp16's range was clamped somewhere in this tile
  if (A.flags != nullptr && __builtin_amdgcn_ballot_w64(amax >= DC_RANGE_LIMIT) != 0 && lane == 0) atomicOr(A.flags, GGNN_FLAG_F16_RANGE);
  GGNN_STAMP_VAL(4, st_wait);
  GGNN_STAMP_VAL(5, st_a);
  GGNN_STAMP_VAL(7, st_p3);
  GGNN_STAMP_VAL(8, st_p4);
  GGNN_STAMP_VAL(9, st_lstm);
  GGNN_STAMP_VAL(10, n_in);
  GGNN_STAMP(16);
}

// (143 360 B of LDS: one workgroup per compute unit, two waves per SIMD, <= 256 registers)
__global__ __launch_bounds__(EC_WAVES * 64) void enc_cell_kernel(const EncCellBatch B) {
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[EC_LDS];
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const int nwg = B.wg_off[k + 1] - B.wg_off[k];
  const int ts = xcd_remap((int)blockIdx.x - B.wg_off[k], nwg);
  enc_cell_body(B.a[k], ts, s_raw);
}

}  // namespace ggnn

extern "C" int ggnn_encoder_cell_batch(const ggnn_enc_cell_args* args, int n_problems, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_problems < 1 || n_problems > EC_MAX_PROBLEMS) return GGNN_EINVAL;
  EncCellBatch B;
  B.n = n_problems;
  B.wg_off[0] = 0;
  for (int k = 0; k < EC_MAX_PROBLEMS; ++k) {
    B.a[k] = args[k < n_problems ? k : 0];
    if (k >= n_problems) {
      B.wg_off[k + 1] = B.wg_off[k];
      continue;
    }
    const ggnn_enc_cell_args& A = B.a[k];
    if (A.n_in < 1 || A.n_in > 2 || A.n_dst <= 0 || A.f_dst < 1 || A.f_dst > 12 || A.ldx < A.f_dst) return GGNN_EINVAL;
    if (!A.x_dst || !A.h_out || !A.c_out || !A.wstream || !A.w2_tail) return GGNN_EINVAL;
    if (!aligned16(A.h_out) || !aligned16(A.c_out) || !aligned16(A.wstream)) return GGNN_EINVAL;
    if (A.n_dst >= INT32_MAX - 64) return GGNN_EINVAL;
    for (int e = 0; e < A.n_in; ++e) {
      const ggnn_enc_cell_sweep& Sw = A.in[e];
      if (!Sw.rowptr || !Sw.einfo || !aligned16(Sw.einfo) || Sw.E < 0) return GGNN_EINVAL;
      if ((Sw.E + GGNN_UNIT_EDGES) * GGNN_EINFO_ROW >= INT32_MAX) return GGNN_EINVAL;  // 32-bit record offsets
    }
    const int64_t n_ts = (A.n_dst + 16 * EC_WAVES - 1) / (16 * EC_WAVES);
    if (B.wg_off[k] + n_ts >= INT32_MAX) return GGNN_EINVAL;
    B.wg_off[k + 1] = B.wg_off[k] + (int)n_ts;
  }
  hipLaunchKernelGGL(enc_cell_kernel, dim3((unsigned)B.wg_off[EC_MAX_PROBLEMS]), dim3(EC_WAVES * 64), 0,
                     (hipStream_t)stream, B);
  return launch_status();
}
