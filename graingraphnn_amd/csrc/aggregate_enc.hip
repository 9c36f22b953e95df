// Encoder aggregation sweep (h = c = 0: the cell sees only the 8 / 11 node features) with the edge
// values on the matrix cores.  Same contract as the h_src == NULL form of
// ggnn_period_gat_aggregate (aggregate.hip) -- PeriodConv.message (periodGATconv.py:204-236) +
// propagate's gather / scatter-add, all gates of one edge type --, but nothing is gathered from a
// projected source row: with h = 0 the value of edge e = (j -> i) is
//     r_e = relu(W_value . x~_e + b_value),   x~_e = [reloc_e, x_j[3:F]]
// a 96 x (<= 11) product of the edge's own 16-float record (ggnn_edge_prepare: reloc, x_j[3:F], .., 1
// at 11 and 12, a_e at 13), so the source-side projection of the encoder (3 x 96 value columns per
// edge type and node: 46 + 23 + 46 MB written and gathered again per model at cfg3) disappears:
//   * a wave works on FOUR units at a time (a unit = one destination row x <= 3 in-edges; 16-lane
//     group q of the wave owns unit q); their 4 x (3 + 1 pad) edge records are the 16 rows of an
//     MFMA A operand (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains);
//   * VALUES: A x [12 x 288] weight fragments that stay in 54 VGPRs for the whole sweep (bias in
//     row 11) -> per column tile a 16 x 16 block D[edge][column]; the columns are ordered so that
//     lane c of group q ends up with channels 32 m + 2c, 32 m + 2c + 1 (m = 0..2) of every gate of
//     ITS unit: the alpha-weighted sum over the unit's edges is lane-local (registers r = 0..2 of
//     D) and every 8-byte store instruction of a group covers one whole 128-byte line;
//   * SCORES: the same A x a [16 x 16] operand whose column 4u + g is the destination-side tail
//     u4 of unit u's row for gate g (ggnn_project: u . x + s1 + a_e s2, see aggregate.hip) and
//     whose column 4u + 3 picks a_e: one more MFMA chain; the softmax of a unit is local to lane
//     4q + g of group q, its 3 weights and the rescale factor reach the group through
//     ds_bpermute;
//   * rows of any degree: units of one row are folded with the online-max softmax carried in
//     registers, exactly as in aggregate.hip; no atomics, one owner per output row.
// The four unit streams of a wave (rows r = 4 (W + k NW) + q for group q) keep their cursors in
// VGPRs (group-uniform), advanced with selects and clamped vector loads one block ahead.
#include <algorithm>

#include "common.h"

namespace ggnn {

constexpr int AE_MAX_SWEEPS = 6;
constexpr int AE_BLOCKS_PER_CU = 3;  // 4 waves each: 12 waves per CU, 3 per SIMD (148 VGPRs)

// Every workgroup belongs to ONE sweep (its waves load that edge type's weight fragments once and
// then only walk rows); the workgroups are dealt to the sweeps in proportion to their rows.
struct EncSweepBatch {
  ggnn_aggregate_enc_args a[AE_MAX_SWEEPS];
  int wg_off[AE_MAX_SWEEPS + 1];  // first workgroup of every sweep
  int n;
};

typedef int ae_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bperm(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float relu_(float x) {  // ONE v_max_f32 (fmaxf / fmed3 add a canonicalising v_max per operand)
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}
__device__ __forceinline__ float quad_lane3(float v) {  // value of the quad's 4th lane in all four
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xFF, 0xF, 0xF, true));
}

template <int G>
__global__ __launch_bounds__(256, AE_BLOCKS_PER_CU) void aggregate_enc_kernel(const EncSweepBatch B) {
  constexpr int NT = 6 * G;  // column tiles: gate t / 6; lane c holds channel 32 ((t % 6) / 2) + 2 c + t % 2 of tile t
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int qa = c >> 2, ra = c & 3;  // A / score-B role: row or column c belongs to unit qa
  const int src_lane4 = (20 * q) * 4;  // byte address of lane 16 q + 4 q for ds_bpermute
  int k = 0;
  while (k + 1 < B.n && (int)blockIdx.x >= B.wg_off[k + 1]) ++k;
  const int64_t W = (int64_t)((int)blockIdx.x - B.wg_off[k]) * 4 + wave;  // this wave among the sweep's waves
  const int64_t NWV = (int64_t)(B.wg_off[k + 1] - B.wg_off[k]) * 4;

  {
    const ggnn_aggregate_enc_args& A = B.a[k];
    if (4 * W >= A.n_dst) return;
    const float* __restrict__ einfo = A.einfo;
    const uint32_t ldp = (uint32_t)A.ldp_dst;
    const int e_last = (int)A.E + GGNN_UNIT_EDGES - 1;  // einfo holds E + GGNN_UNIT_EDGES records

    // value weights: B fragments, stationary in registers
    float bw[NT][3];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 3; ++s) bw[t][s] = A.wv_frag[(t * 3 + s) * 64 + lane];

    // ---- cursor of the unit stream of group q (group-uniform values in VGPRs; vector loads with
    // clamped addresses and selects instead of branches: scalar loads would share lgkmcnt with the
    // ds_bpermutes below, and loads under branches drag waits to the merge points):
    // current row r with units u .. ue, the next row's (un .. une) one row ahead ----
    const int n_dst = (int)A.n_dst, stride = (int)(4 * NWV);
    const int32_t* __restrict__ uptr = A.unit_ptr;
    const ae_i32x4* __restrict__ udesc = reinterpret_cast<const ae_i32x4*>(A.units);  // 2 x int4 per unit; [0] = {i, p0, flags, -}
    // Software pipeline over the blocks b = 0, 1, .. of this wave: at the top of iteration b the
    // operands of block b + 1 (addresses from its descriptor) and the descriptor of block b + 2 are
    // requested; both are taken over at the END of the iteration, in front of the stores.
    int r = (int)(4 * W) + q;     // cursor: the unit of the block whose descriptor was requested last
    bool alive = r < n_dst;
    int u, ue, un, une;
    {
      const int rc = min(r, n_dst - 1), rnc = min(r + stride, n_dst - 1);
      u = uptr[rc];
      ue = uptr[rc + 1];
      un = uptr[rnc];
      une = uptr[rnc + 1];
    }
    const int src_grp = (16 * qa) * 4;  // byte address of a lane of group qa for ds_bpermute
    int un_ld = 0, une_ld = 0;
    bool row_done = false;
    auto advance = [&]() {  // cursor -> next unit of the stream; requests the row pointers one row further ahead
      row_done = u + 1 >= ue;
      r = row_done ? r + stride : r;
      u = row_done ? un : u + 1;
      ue = row_done ? une : ue;
      alive = alive && r < n_dst;
      const int rnc = min(r + stride, n_dst - 1);
      un_ld = uptr[rnc];
      une_ld = uptr[rnc + 1];
    };
    auto settle = [&]() {  // takes the requested row pointers over
      un = row_done ? un_ld : un;
      une = row_done ? une_ld : une;
    };
    auto load_desc = [&]() {
      ae_i32x4 v = udesc[2 * (int64_t)(alive ? u : 0)];
      return v;
    };
    auto operands = [&](const ae_i32x4& dd, float (&a)[4], float (&bs)[4]) {
      const int p0_a = __builtin_amdgcn_ds_bpermute(src_grp, dd[1]);
      const int i_b = __builtin_amdgcn_ds_bpermute(src_grp, dd[0]);
      // record of edge ra of unit qa (slot 3 is padding: whatever record follows, clamped to the buffer)
      const float* arow = einfo + (uint32_t)min(p0_a + ra, e_last) * GGNN_EINFO_ROW + q;
      const float* brow = A.p_dst + (uint32_t)i_b * ldp + A.u4_off + 16 * min(ra, G - 1) + q;
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = arow[4 * s];
#pragma unroll
      for (int s = 0; s < 4; ++s) bs[s] = __builtin_nontemporal_load(brow + 4 * s);
    };
    // prologue: descriptors of blocks 0 and 1, operands of block 0
    bool alive_c = alive;
    ae_i32x4 d = load_desc();
    if (!alive_c) d[0] = d[1] = d[2] = 0;  // a dead stream: row 0, no edges, neither first nor last
    advance();
    settle();
    bool alive_n = alive;
    ae_i32x4 dn = load_desc();
    if (!alive_n) dn[0] = dn[1] = dn[2] = 0;
    float a[4], bs[4];
    operands(d, a, bs);

    // per-lane softmax / accumulation state of the row group q is folding
    float mx = -INFINITY;  // meaningful in lanes c = 4 q + g
    float den[G], sae[G], acc[NT];
#pragma unroll
    for (int g = 0; g < G; ++g) den[g] = sae[g] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;

    while (__builtin_amdgcn_ballot_w64(alive_c) != 0) {
      // ---- requests for the blocks ahead ----
      float a_n[4], bs_n[4];
      operands(dn, a_n, bs_n);
      advance();
      const bool alive_nn = alive;
      ae_i32x4 dnn = load_desc();
      const int i_q = d[0], fl_q = d[2];
      if (ra >= G) {  // column 4 u + 3: picks a_e = x4[13] (k-step 3, k = 12 + q)
#pragma unroll
        for (int s = 0; s < 4; ++s) bs[s] = (s == 3 && q == 1) ? 1.0f : 0.0f;
      }

      const int nact = fl_q & 0xFF;
      const bool first = (fl_q >> 8) & 1, last = (fl_q >> 9) & 1;
      if (first) {
        mx = -INFINITY;
#pragma unroll
        for (int g = 0; g < G; ++g) den[g] = sae[g] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = 0.f;
      }

      // ---- scores: D[edge 4 q + r][column c]; lane c = 4 q + g holds unit q's scores for gate g ----
      f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bs[s], sc, 0, 0, 0);
      const float ae0 = quad_lane3(sc[0]), ae1 = quad_lane3(sc[1]), ae2 = quad_lane3(sc[2]);
      const float s0 = nact > 0 ? sc[0] : -INFINITY, s1 = nact > 1 ? sc[1] : -INFINITY,
                  s2 = nact > 2 ? sc[2] : -INFINITY;
      const float mnew = fmaxf(fmaxf(mx, s0), fmaxf(s1, s2));
      const bool any = nact > 0;  // (an empty row has one unit with nact == 0: zeros are stored)
      const float scale_l = any ? __expf(mx - mnew) : 1.0f;  // exp(-inf) = 0 on a row's first unit
      const float p0_l = any ? __expf(s0 - mnew) : 0.f, p1_l = any ? __expf(s1 - mnew) : 0.f,
                  p2_l = any ? __expf(s2 - mnew) : 0.f;
      if (any) mx = mnew;
      // the softmax lanes' results -> every lane of the group
      float scale[G], p0[G], p1[G], p2[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        scale[g] = bperm(src_lane4 + 4 * g, scale_l);
        p0[g] = bperm(src_lane4 + 4 * g, p0_l);
        p1[g] = bperm(src_lane4 + 4 * g, p1_l);
        p2[g] = bperm(src_lane4 + 4 * g, p2_l);
      }
      const float e0 = bperm(src_lane4, ae0), e1 = bperm(src_lane4, ae1), e2 = bperm(src_lane4, ae2);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        den[g] = den[g] * scale[g] + (p0[g] + p1[g] + p2[g]);
        sae[g] = sae[g] * scale[g] + (p0[g] * e0 + p1[g] * e1 + p2[g] * e2);
      }

      // ---- values: per column tile D[edge 4 q + r][column c] = W x~ + b, relu, alpha-weighted sum ----
#pragma unroll
      for (int t0 = 0; t0 < NT; t0 += 3) {
        // three column tiles (of one gate) as three independent MFMA chains, issued k-step by k-step
        const int g = t0 / 6;
        f32x4 v[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) v[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int m = 0; m < 3; ++m) v[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw[t0 + m][s], v[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 3; ++m)
          acc[t0 + m] = acc[t0 + m] * scale[g] +
                        (p0[g] * relu_(v[m][0]) + p1[g] * relu_(v[m][1]) + p2[g] * relu_(v[m][2]));
      }

      // what was requested at the top is taken over HERE, in front of the stores: a wait placed behind
      // them (or carried over the loop's back edge) would drain the stores as well
      __builtin_amdgcn_sched_barrier(0);  // (and not earlier: the scheduler would hoist the selects to the loop head)
      settle();
      d = dn;
      alive_c = alive_n;
      dn = dnn;
      alive_n = alive_nn;
      if (!alive_n) dn[0] = dn[1] = dn[2] = 0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = a_n[s];
        bs[s] = bs_n[s];
      }
      asm volatile("" : "+v"(un), "+v"(une), "+v"(dn), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(bs[0]),
                   "+v"(bs[1]), "+v"(bs[2]), "+v"(bs[3]));  // keep the waits here
      if (last) {
        float* orow = A.agg + (int64_t)i_q * A.ld_agg + A.a_off + 2 * c;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float inv = 1.0f / (den[g] + 1e-16f);  // PyG softmax denominator
          float* o = orow + g * A.a_gstride;
#pragma unroll
          for (int m = 0; m < 6; m += 2) {  // every store instruction covers one whole 128-byte line per group
            const f32x2_ w2 = {acc[6 * g + m] * inv, acc[6 * g + m + 1] * inv};
            __builtin_nontemporal_store(w2, reinterpret_cast<f32x2_*>(o + 16 * m));
          }
          if (c == 0) {  // 8 bytes of a line the other edge type's sweep also writes into: through L2
            float* sp = A.agg + (int64_t)i_q * A.ld_agg + g * A.a_gstride + A.sc_off;
            sp[0] = den[g] * inv;
            sp[1] = sae[g] * inv;
          }
        }
      }
    }
  }
}

}  // namespace ggnn

extern "C" int ggnn_period_gat_aggregate_enc_batch(const ggnn_aggregate_enc_args* args, int n_sweeps,
                                                   ggnn_stream_t stream) {
  using namespace ggnn;
  if (!args || n_sweeps < 1 || n_sweeps > AE_MAX_SWEEPS) return GGNN_EINVAL;
  EncSweepBatch B;
  B.n = n_sweeps;
  int64_t rows = 0;
  for (int k = 0; k < AE_MAX_SWEEPS; ++k) {
    B.a[k] = args[k < n_sweeps ? k : 0];
    const ggnn_aggregate_enc_args& A = B.a[k];
    if (!A.unit_ptr || !A.units || !A.einfo || !A.p_dst || !A.wv_frag || !A.agg) return GGNN_EINVAL;
    if (!aligned16(A.units) || !aligned16(A.einfo) || (reinterpret_cast<uintptr_t>(A.agg) & 7u)) return GGNN_EINVAL;
    if (A.n_dst <= 0 || A.E < 0 || A.n_gates != 3) return GGNN_EINVAL;
    if (A.u4_off < 0 || A.a_off < 0 || (A.a_off & 1) || A.sc_off < 0 || A.a_gstride < C || (A.a_gstride & 1) ||
        (A.ld_agg & 1))
      return GGNN_EINVAL;
    if (A.ldp_dst <= 0 || A.n_dst * A.ldp_dst >= INT32_MAX || (A.E + GGNN_UNIT_EDGES + 1) * GGNN_EINFO_ROW >= INT32_MAX)
      return GGNN_EINVAL;  // 32-bit row offsets
    if (A.u4_off + (int64_t)A.n_gates * 16 > A.ldp_dst) return GGNN_EINVAL;
    if ((int64_t)(A.n_gates - 1) * A.a_gstride + A.a_off + C > A.ld_agg) return GGNN_EINVAL;
    if ((int64_t)(A.n_gates - 1) * A.a_gstride + A.sc_off + 2 > A.ld_agg) return GGNN_EINVAL;
    if (k < n_sweeps) rows += A.n_dst + A.E / GGNN_UNIT_EDGES;  // ~ units
  }
  // persistent grid: at most the resident capacity, dealt to the sweeps by their (approximate) unit
  // counts; a workgroup covers at least 16 rows (4 waves x 4 streams)
  const int64_t cap = (int64_t)num_cu() * AE_BLOCKS_PER_CU;
  B.wg_off[0] = 0;
  for (int k = 0; k < AE_MAX_SWEEPS; ++k) {
    int64_t n = 0;
    if (k < n_sweeps) {
      const ggnn_aggregate_enc_args& A = B.a[k];
      n = std::max<int64_t>(1, cap * (A.n_dst + A.E / GGNN_UNIT_EDGES) / rows);
      n = std::min<int64_t>(n, (A.n_dst + 15) / 16);
    }
    B.wg_off[k + 1] = B.wg_off[k] + (int)n;
  }
  const dim3 grid((unsigned)B.wg_off[AE_MAX_SWEEPS]);
  hipLaunchKernelGGL((aggregate_enc_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, B);
  return launch_status();
}
