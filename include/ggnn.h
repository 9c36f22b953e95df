/*
 * ggnn.h -- C ABI of libggnn.so: the MI355X (gfx950) kernels behind the GrainGNN rollout
 * hot path (GrainNN_regressor.forward + GrainNN_classifier.forward + rollout-step glue).
 *
 * Every entry point is `extern "C"`, takes raw DEVICE pointers + sizes + a hipStream_t
 * (passed as void*), never allocates, never synchronises, never throws, keeps no global
 * state, and returns 0 or a negative GGNN_E* code.  All work is enqueued on `stream`, so
 * a caller may capture any sequence of calls into a hipGraph.
 *
 * Citations (file:line) are into the reference repository YigongQin/GrainGraphNN.  The
 * reference has no FFI of its own (it is pure Python on PyTorch + PyG); each entry point
 * names the Python function(s) whose work it replaces.  INTEGRATION.md shows the ctypes
 * binding a reference maintainer would add.
 *
 * Fixed by the shipped models (parameters.py:18-50, 97-134): hidden width C = 96, one
 * attention head, fp32 everywhere.  Features per node <= 12 (grain 11, joint 8).
 */
#ifndef GGNN_H_
#define GGNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGNN_ABI_VERSION 25
#define GGNN_C 96              /* hidden width (hyper.layer_size of every shipped model) */
#define GGNN_MAX_GATES 4       /* i, f, c, o */
#define GGNN_EDGE_PARAM_ROWS 3 /* per gate: W_value[:, 0:3] (the value side of the min-image correction) */
#define GGNN_UNIT_EDGES 3      /* in-edges per aggregation unit (every junction has exactly 3) */

#define GGNN_OK 0
#define GGNN_EINVAL (-1)  /* bad argument (null pointer, size, alignment, unsupported width) */
#define GGNN_ELAUNCH (-2) /* hipLaunch / hipMemsetAsync reported an error */
#define GGNN_ETOPOLOGY (-3) /* ggnn_topology_update: the lists are not a valid grain graph (message in args->error) */

typedef void* ggnn_stream_t; /* hipStream_t */

/* Gate-epilogue modes of ggnn_lstm_epilogue */
#define GGNN_MODE_LSTM 0    /* 4 gates (i,f,c,o), c_in given: c' = f*c + i*tanh(.), h' = o*tanh(c') */
#define GGNN_MODE_LSTM_H0 1 /* 3 gates (i,c,o), h = c = 0 (encoder): c' = i*tanh(.), h' = o*tanh(c') */
#define GGNN_MODE_RAW 2     /* no LSTM: write the n_gates pre-activations (used for PeriodConv parity) */

int ggnn_version(void);
const char* ggnn_error_string(int code);
/* Arithmetic of the K >= 100 GEMMs (decoder ggnn_project, ggnn_lstm_epilogue), fixed per process
 * by the environment variable GGNN_GEMM: GGNN_GEMM_BF16X6 (default) = every fp32 operand split
 * exactly into three bf16 pieces, six bf16 MFMA products per k-step, fp32 accumulate (dropped
 * terms <= 2^-25 of a product: fp32-equivalent); GGNN_GEMM_FP32 ("fp32") = native fp32 MFMA. */
#define GGNN_GEMM_FP32 0
#define GGNN_GEMM_BF16X6 1
int ggnn_gemm_mode(void);

/* ------------------------------------------------------------------------------------
 * CSR build.  Replaces the COO bookkeeping inside PyG MessagePassing.propagate as used at
 * periodGATconv.py:174-175 (gather by edge_index[0]/[1], scatter-add by edge_index[1]):
 * edges are grouped by destination once per topology, so aggregation needs no atomics.
 *   edge_index : [2, E] int64, row 0 = source node, row 1 = destination node (device)
 *   rowptr     : [n_dst + 1] int32 out
 *   col        : [E] int32 out, source node of each CSR slot
 *   perm       : [E] int32 out, original COO edge id of each CSR slot (ascending inside a
 *                row => the result is deterministic and order-stable)
 *   row        : [E] int32 out, destination node of each CSR slot
 *   unit_ptr   : [n_dst + 1] int32 out, first unit of every destination row
 *   units      : [ggnn_csr_max_units(E, n_dst), 8] int32 out, 16-byte aligned.  A unit is a
 *                destination row restricted to <= GGNN_UNIT_EDGES consecutive in-edges:
 *                {i, p0, nact | first<<8 | last<<9, 0, j0, j1, j2, 0}; rows without edges
 *                get one empty unit; absent edges repeat j0.  unit_ptr[n_dst] = #units.
 *   flags      : [1] int32 device word, bit 0 is OR-ed in when an index is out of range
 *                (such edges are dropped; the host wrapper raises)
 *   workspace  : ggnn_csr_workspace_bytes(E, n_dst) bytes of device scratch
 */
size_t ggnn_csr_workspace_bytes(int64_t E, int64_t n_dst);
int64_t ggnn_csr_max_units(int64_t E, int64_t n_dst); /* upper bound: n_dst + E / GGNN_UNIT_EDGES */
int ggnn_build_csr(const int64_t* edge_index, int64_t E, int64_t n_src, int64_t n_dst,
                   int32_t* rowptr, int32_t* col, int32_t* perm, int32_t* row, int32_t* unit_ptr,
                   int32_t* units, int32_t* flags, void* workspace, size_t workspace_bytes,
                   ggnn_stream_t stream);
/* The same build for up to four lists in one sequence of launches (a topological event rebuilds the three edge types'
 * tables: seven launches instead of twenty-four).  Fields as the arguments of ggnn_build_csr. */
typedef struct ggnn_csr_args {
  const int64_t* edge_index;
  int64_t E, n_src, n_dst;
  int32_t* rowptr;
  int32_t* col;
  int32_t* perm;
  int32_t* row;
  int32_t* unit_ptr;
  int32_t* units;
  int32_t* flags;
  void* workspace;
  size_t workspace_bytes;
} ggnn_csr_args;
int ggnn_build_csr_batch(const ggnn_csr_args* problems, int n_problems, ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Per-edge record in CSR order, computed once per forward and shared by every gate of the
 * encoder and decoder cells (GGNN_EINFO_ROW = 20 floats per edge):
 *   einfo[p, 0..15]  = (reloc_xyz, x_src[col[p], 3 .. f_src), 0.., 1 at 12, edge_attr at 13, 0, 0)
 *                      -- the non-hidden part of the source row the attention score is taken with;
 *                      when f_src <= 11 slot 11 is 1 as well (the bias row of the encoder sweep's
 *                      value product; the score tail u4 is 0 there)
 *   einfo[p, 16..19] = (reloc_x, reloc_y, reloc_z, edge_attr)
 * with reloc = min-image(x_src[col[p], :3] - x_dst[row[p], :3]) exactly as periodGATconv.py:209-210
 * (rel > 0.5 -> rel - 1, rel < -0.5 -> rel + 1) and edge_attr taken from the ORIGINAL COO
 * order through perm (edge_attr_dict[et][:, 0]).  Up to three edge types per launch.
 */
#define GGNN_EINFO_ROW 20
typedef struct ggnn_prepare_edge {
  const int32_t* col;     /* [E] */
  const int32_t* perm;    /* [E] */
  const int32_t* row;     /* [E] */
  const float* edge_attr; /* [E] COO order */
  const float* x_src;     /* [n_src, ldx_src] */
  const float* x_dst;     /* [n_dst, ldx_dst] */
  float* einfo;           /* [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW] out, 16-byte aligned (tail rows are padding) */
  int64_t ldx_src, ldx_dst, E, f_src; /* 3 <= f_src <= 12 source features */
  const int64_t* E_dev;   /* ABI 25, or NULL: the number of edges read from DEVICE memory when the kernel runs (E is then the
                             capacity the launch is sized for, *E_dev <= E) -- a launch captured in a hipGraph follows an
                             edge list that SHRINKS in place (GrainRollout's event loop: grain eliminations remove edges;
                             the lists, CSR tables and per-edge buffers keep their addresses and capacity) */
} ggnn_prepare_edge;
int ggnn_edge_prepare(const ggnn_prepare_edge* edges, int n_edge_types, ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Node-level projection: out[M, ncols] = [X[:, :F] | H] . Wp^T + bias.
 * Replaces, for all gates and edge types at once, the per-edge lin_query / lin_key /
 * lin_value (periodGATconv.py:216-218) and lin_skip (:186) applied to cat[x, h]
 * (heteropgclstm.py:112,120,128,137), using linearity to hoist them from edges to nodes.  What
 * the rows of Wp hold is the caller's business (graingraphnn_amd/packing.py: value rows for the
 * node as a source; W_k^T W_q / sqrt(96) rows for it as a destination, see
 * ggnn_period_gat_aggregate; summed skip rows).
 *   X  : [M, ldx] node features, first F columns used (F <= 12)
 *   H  : [M, ldh] hidden state (k2 = 96) or NULL (k2 = 0, encoder: h = 0)
 *   Wp : [ncols, Kp] packed weight rows, Kp = roundup4(F) + k2; columns [F, roundup4(F)) zero
 *   ncols % 96 == 0; ldo % 4 == 0; H, Wp, bias, out 16-byte aligned.
 */
int ggnn_project(const float* X, int64_t ldx, int F, const float* H, int64_t ldh, int k2,
                 const float* Wp, const float* bias, int64_t M, int ncols, float* out,
                 int64_t ldo, ggnn_stream_t stream);
/* Up to four projections in ONE launch: the node types of one cell and / or the same cell of the
 * regressor and the classifier (both see the same x_dict, test.py:382-383).  Fields as the
 * arguments of ggnn_project; all problems must share k2.  Rows are addressed with a 64-bit tile base
 * and 32-bit offsets inside a 16-row tile: 16 * max(ldx, ldh, ldo) < 2^31 is the only size limit.
 * Same result as n single calls. */
typedef struct ggnn_project_args {
  const float* X;
  const float* H; /* NULL when k2 == 0 */
  const float* Wp;
  const float* bias;
  float* out;
  int64_t ldx, ldh, M, ldo;
  int32_t F, k2, ncols;
  int32_t precision; /* 0: fp32-equivalent over fp32's whole range (exact 3-piece bf16 split, 6 products per k-step);
                        GGNN_PRECISION_F16X2 (k2 == 96 only): fp32-equivalent for |x|, |h|, |w| < 65504 -- the fused
                        cells' arithmetic, two fp16 pieces per operand and THREE products per k-step (operands beyond
                        the range are clamped: the caller checks its weights, packing.pack_cell; the node rows are the
                        ones ggnn_decoder_cell_batch checks and reports) -- the value rows of the fused decoder plan;
                        GGNN_PRECISION_BF16 (k2 == 96 only): operands rounded to bf16, ONE bf16 MFMA product per
                        k-step, fp32 accumulate -- what torch.autocast(bfloat16) asks of a linear (training path).
                        | GGNN_OUT_BLOCK_MAJOR (k2 == 96 only, ABI 24): `out` is [ncols / 96][M][96] -- every block of 96
                        columns a contiguous [M, 96] matrix of its own (ldo is ignored) -- instead of [M, ldo]: the layout
                        of the fused decoder plan's value rows, which are written once here and gathered 96 columns
                        (one edge type and gate) at a time by ggnn_decoder_cell_batch (v_block_major) */
} ggnn_project_args;
int ggnn_project_batch(const ggnn_project_args* args, int n_problems, ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Periodic-boundary GAT aggregation for one edge type, all gates fused.  Replaces
 * PeriodConv.message (periodGATconv.py:204-236) + PyG propagate's gather and scatter-add:
 * min-image wrap of the first three coordinates, scaled dot of query and key, segment softmax
 * (+1e-16), relu of the value, alpha-weighted sum.  The key is never formed: with
 * x~_j = [reloc_e, x_j[3:F], h_j] the score q_i.(W_k x~_j + b_k + w_edge a_e)/sqrt(96) equals
 * u_i . x~_j + s1_i + a_e s2_i, where u_i = W_k^T q_i/sqrt(96), s1_i = b_k.q_i/sqrt(96) and
 * s2_i = w_edge.q_i/sqrt(96) are affine in the destination's [x_i | h_i] and come from
 * ggnn_project.  Per destination i and gate g the projections hold
 *   p_dst[i, u_off  + g*96 + 0..95] = u_i[F ..]   (hidden-state part; absent when h_src == NULL)
 *   p_dst[i, u4_off + g*16 + 0..15] = (u_i[0..F-1], 0.., s1_i at 12, s2_i at 13, 0, 0), dotted
 *                                     with einfo[e, 0..15]
 *   p_src[j, v_off  + g*96 + 0..95] = lin_value(x_j with its first three columns zeroed, h_j)
 * and the sweep writes
 *   agg[i, g*a_gstride + a_off + 0..95] = sum_e alpha_e * relu(lin_value(x~_j))
 *   agg[i, g*a_gstride + sc_off + 0]    = sum_e alpha_e            (1, or 0 if no in-edge)
 *   agg[i, g*a_gstride + sc_off + 1]    = sum_e alpha_e * edge_attr_e
 * (lin_l2, its bias, the value-side lin_edge term and lin_skip are applied afterwards by
 * ggnn_lstm_epilogue, which is exact because they are linear in these sums.)
 */
typedef struct ggnn_aggregate_args {
  const int32_t* unit_ptr;  /* [n_dst + 1] from ggnn_build_csr */
  const int32_t* units;     /* [n_units, 8] from ggnn_build_csr */
  const float* einfo;       /* [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW] from ggnn_edge_prepare */
  const float* p_src;       /* [n_src, ldp_src] projections of the source node type */
  const float* p_dst;       /* [n_dst, ldp_dst] projections of the destination node type */
  const float* h_src;       /* [n_src, ldh_src] source hidden state, or NULL (encoder: h = 0) */
  const float* edge_params; /* [n_gates][GGNN_EDGE_PARAM_ROWS][96] = W_value[:, 0..2] */
  float* agg;               /* [n_dst, ld_agg] */
  int64_t ldp_src, ldp_dst, ld_agg, ldh_src, n_src, n_dst, E;
  int32_t v_off, u_off, u4_off, a_off, a_gstride, sc_off, n_gates;
  int32_t pad_n;            /* ABI 25: the sweep also writes zeros into agg[i, g a_gstride + sc_off + 2 .. + pad_n) of every gate
                               row (the padding between the last edge type's scalars and the next gate row, which the training
                               path's gate GEMM multiplies with zero weight columns: it has to be finite); 0 = nothing */
} ggnn_aggregate_args;
int ggnn_period_gat_aggregate(const ggnn_aggregate_args* args, ggnn_stream_t stream);
/* The 1..6 sweeps of one cell (HeteroConv over the edge types, heteropgclstm.py:148-183; of one
 * model, or of the regressor and the classifier, which see the same graph: test.py:382-383) in ONE
 * launch: args[0..n_sweeps).  All must have the same n_gates and agree on h_src == NULL; they may
 * write disjoint columns of the same agg rows.  Same result as n_sweeps single calls. */
int ggnn_period_gat_aggregate_batch(const ggnn_aggregate_args* args, int n_sweeps,
                                    ggnn_stream_t stream);

/* Encoder form of the sweep (h = c = 0, heteropgclstm.py:101-110 with the zero state of
 * models.py:422: the cell sees the bare features), values on the matrix cores: nothing is gathered
 * from a projected source row.  With h = 0 the value of an edge is relu(W_value . [reloc, x_j[3:F]]
 * + b_value), a product of the edge's own einfo record, so p_src / v_off / edge_params of
 * ggnn_aggregate_args are replaced by
 *   wv_frag: [6 * n_gates][3][64] fp32, MFMA B fragments of Bp [12, n_gates * 96] with
 *            Bp[k][g*96 + ch] = W_value_g[ch][k] for k < F_src (columns 0..2 act on reloc),
 *            Bp[11][.] = b_value_g, 0 elsewhere (F_src <= 11; einfo[:, 11] = 1):
 *            element [t][s][l] = Bp[4 s + (l >> 4)][(t / 6) * 96 + 32 ((t % 6) / 2) + 2 (l & 15) + t % 2].
 * p_dst, u4_off, agg, a_off, a_gstride, sc_off and the result are those of the h_src == NULL form
 * of ggnn_period_gat_aggregate (same sums up to fp32 re-association).  n_gates = 3; a_off, a_gstride,
 * ld_agg even.  Up to six sweeps per launch (three edge types x two models). */
typedef struct ggnn_aggregate_enc_args {
  const int32_t* unit_ptr; /* [n_dst + 1] from ggnn_build_csr */
  const int32_t* units;    /* [n_units, 8] from ggnn_build_csr */
  const float* einfo;      /* [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW] from ggnn_edge_prepare */
  const float* p_dst;      /* [n_dst, ldp_dst] destination projections (the u4 tails) */
  const float* wv_frag;    /* [6 * n_gates][3][64] */
  float* agg;              /* [n_dst, ld_agg] */
  int64_t ldp_dst, ld_agg, n_dst, E;
  int32_t u4_off, a_off, a_gstride, sc_off, n_gates, reserved;
} ggnn_aggregate_enc_args;
int ggnn_period_gat_aggregate_enc_batch(const ggnn_aggregate_enc_args* args, int n_sweeps,
                                        ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Backward of one sweep (training path).  Replaces what autograd does for
 * PeriodConv.message + propagate (periodGATconv.py:174-175, 204-236) in train.py:158-166:
 * segment-softmax backward, relu mask, scatter of the value / hidden-state gradients to the
 * source rows -- without atomics (a destination-grouped pass, then a source-grouped pass over
 * the reverse CSR), so gradients are reproducible run to run.
 * Forward operands exactly as in ggnn_aggregate_args (same offsets and strides) plus the saved
 * forward output `agg` and the incoming gradient `g_agg` (same layout as agg; only the 96 value
 * columns and the two scalars of every gate are read).  Outputs, in the layout of their operand:
 *   g_p_dst[i, u_off + g*96 ..], g_p_dst[i, u4_off + g*16 ..]   (other columns untouched)
 *   g_p_src[j, v_off + g*96 ..]                                  (other columns untouched)
 *   g_h_src[j, 0..95]                                            (when h_src != NULL)
 *   ep_partial[w, g, k, c]: partial sums of d edge_params; the gradient is their sum over w.
 * The geometry (einfo) is data, not a parameter: no gradient.
 */
typedef struct ggnn_aggregate_bwd_args {
  const int32_t* rowptr;   /* [n_dst + 1] destination-grouped CSR (ggnn_build_csr) */
  const int32_t* col;      /* [E] source node of every CSR slot */
  const float* einfo;      /* [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW] */
  const float* p_src;      /* forward operands, as ggnn_aggregate_args */
  const float* p_dst;
  const float* h_src;      /* or NULL */
  const float* edge_params;
  const float* agg;        /* [n_dst, ld_agg] forward output */
  const float* g_agg;      /* [n_dst, ld_agg] gradient with respect to agg */
  const int32_t* r_rowptr; /* [n_src + 1] source-grouped (reverse) CSR of the same edges */
  const int32_t* r_dst;    /* [E] destination node of every reverse slot */
  const int32_t* r_slot;   /* [E] forward CSR slot of every reverse slot */
  float* edge_alpha;       /* [E, n_gates] scratch: attention weights */
  float* edge_ds;          /* [E, n_gates] scratch: score gradients */
  float* ep_partial;       /* [n_partials, n_gates, 3, 96] out */
  float* g_p_dst;          /* [n_dst, ldp_dst] out */
  float* g_p_src;          /* [n_src, ldp_src] out */
  float* g_h_src;          /* [n_src, ldh_src] out, or NULL */
  int64_t ldp_src, ldp_dst, ld_agg, ldh_src, n_src, n_dst, E, n_partials;
  int32_t v_off, u_off, u4_off, a_off, a_gstride, sc_off, n_gates;
  int32_t g_h_accumulate;  /* ABI 25: 1 = g_h_src += (the sweeps of a cell that share a source node type add into one buffer
                              the first of them wrote), 0 = g_h_src is written */
} ggnn_aggregate_bwd_args;
int64_t ggnn_aggregate_bwd_partials(int64_t n_dst); /* rows of ep_partial the call writes (one per workgroup of its destination pass) */
int ggnn_period_gat_aggregate_backward(const ggnn_aggregate_bwd_args* args, ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Gate GEMM + LSTM epilogue (arithmetic per ggnn_gemm_mode).  For every node and gate:
 *   pre[g] = agg[:, g, 0:Ka] . W2[g]^T + p_dst[:, s_off + g*96 ...]
 * where W2[g] = [lin_l2.weight of each incoming edge type | b_l2, w_edge per edge type]
 * (periodGATconv.py:218, 231-235), the skip/bias term was produced by ggnn_project
 * (lin_skip summed over incoming edge types = HeteroConv aggr 'sum', plus b_{i,f,c,o}),
 * followed by the cell update of heteropgclstm.py:111-146.
 *   mode GGNN_MODE_LSTM    : n_gates = 4, needs c_in, writes h_out, c_out   [N, 96]
 *   mode GGNN_MODE_LSTM_H0 : n_gates = 3 (i, c, o), writes h_out, c_out
 *   mode GGNN_MODE_RAW     : writes raw_out [N, n_gates*96] = pre
 * Ka % 4 == 0, Ka <= 200.
 */
typedef struct ggnn_epilogue_args {
  const float* agg;   /* [N, n_gates*Ka] */
  const float* w2;    /* [n_gates][96][Ka] */
  const float* p_dst; /* [N, ldp] */
  const float* c_in;  /* [N, 96] or NULL */
  float* h_out;       /* [N, 96] */
  float* c_out;       /* [N, 96] */
  float* raw_out;     /* [N, n_gates*96] (GGNN_MODE_RAW) */
  int64_t ldp, N;
  int32_t Ka, s_off, n_gates, mode;
  /* Optional (NULL = native fp32 MFMA kernel): w2[:, :, 0:Ka-4] re-ordered into MFMA A fragments,
   * fp32, [n_gates][(Ka-4)/32][6][2][64][4]: element [g][ks][ct][h][l][j] =
   * w2[g][16 ct + (l & 15)][32 ks + 8 (l >> 4) + 4 h + j] (one 16-byte load per lane of a wave =
   * 1 KB of contiguous memory).  The kernel splits every value exactly into three bf16 pieces
   * (x = hi + mid + lo, each the bf16 round-to-nearest of what is left) on the fly.  16-byte
   * aligned; declared as uint16 pairs for historical reasons (ABI <= 12 passed pre-split planes).
   * Used when ggnn_gemm_mode() == GGNN_GEMM_BF16X6; Ka - 4 must be a multiple of 32. */
  const uint16_t* w2_planes;
  /* Layout of agg: row stride ld_agg and gate stride g_stride, in floats (0, 0 = packed:
   * g_stride = Ka, ld_agg = n_gates * Ka).  Both multiples of 4, g_stride >= Ka,
   * ld_agg >= n_gates * g_stride.  Multiples of 32 (128-byte aligned k-steps) are what the
   * workspace uses: the fragment loads of the bf16x6 kernel then touch one cache line each. */
  int64_t ld_agg;
  int32_t g_stride, reserved;
} ggnn_epilogue_args;
int ggnn_lstm_epilogue(const ggnn_epilogue_args* args, ggnn_stream_t stream);
/* Up to four gate GEMM + LSTM problems in ONE launch: the node types of one cell
 * (heteropgclstm.py:111-146 runs the update per node type) and / or the same cell of the
 * regressor and the classifier (test.py:382-383 calls both models on the same x_dict).  All
 * problems must share `mode` and `n_gates`; Ka may differ.  Same result as n single calls. */
int ggnn_lstm_epilogue_batch(const ggnn_epilogue_args* args, int n_problems, ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Encoder cell (h = c = 0; HeteroPGCLSTM.forward on the zero state of models.py:422,
 * heteropgclstm.py:101-183: no forget gate) with EVERYTHING that belongs to a destination node in one kernel and
 * one launch: the score tails u4, the sweeps of the incoming edge types (PeriodConv.message,
 * periodGATconv.py:204-236), lin_l2 + the lin_edge value term, HeteroConv's sum over the edge types
 * (heteropgclstm.py:113-138), the summed skip term and the LSTM update.  Replaces the encoder's share of
 * ggnn_project_batch, ggnn_period_gat_aggregate_enc_batch and ggnn_lstm_epilogue_batch(GGNN_MODE_LSTM_H0) -- same
 * sums up to fp32 re-association; nothing but x, the edge records and the weights is read, nothing but (h, c) is
 * written.  One problem = one destination node type of one model, with its 1 or 2 incoming edge types; up to four
 * problems (two node types x regressor and classifier, test.py:382-383) per call.
 *
 * With h = 0 a PeriodConv only sees 16-slot rows: of a destination node (x_0 .. x_{f_dst-1}, 0 .., 1 at slot 12)
 * and of an in-edge (ggnn_edge_prepare's record: reloc, x_src[3 .. f_src), 0 .., 1 at 12, edge length at 13).
 * A workgroup of eight waves owns 128 consecutive destination nodes, one 16-node tile per wave, and walks the
 * gates i, c~, o; per gate g and incoming edge type e: u4[node] = T(e, g) . feature slots; score of an in-edge =
 * u4[node] . record; online-max softmax over the node's in-edges (PyG: exp(s - max) / (sum + 1e-16)); value of an
 * in-edge = relu(V(e, g) . record) (its row 12 column is b_value); aggregate = sum alpha value; pre-activation +=
 * lin_l2(e, g) . aggregate + b_l2 sum alpha + w_edge sum alpha a_e; then + S(g) . feature slots (summed lin_skip +
 * gate bias); i = sig, c' = i tanh(c~), h' = sig(o) tanh(c').  Arithmetic and OPERAND RANGE as for
 * ggnn_decoder_cell_batch (two fp16 pieces, three products; |x| < 65504, reported through *flags).
 *
 * Per incoming edge type: rowptr [n_dst + 1] (ggnn_build_csr), einfo (ggnn_edge_prepare), E.
 * Per problem:
 *   x_dst   : [n_dst, ldx] features of the destination nodes (first f_dst <= 12 columns)
 *   wstream : the weight slices in the order the kernel consumes them (packing.encoder_cell_stream):
 *             for g in (i, c~, o): for e: 1 slice A(e, g) | 3 slices lin_l2(e, g); then 1 slice S(g).
 *             Every slice is GGNN_DC_SLICE_BYTES in ggnn_dec_cell_args.wstream's image
 *             ([column tile nb][plane hi, lo'][64 lanes][8 fp16], lane l = 16 kq + m of (nb, plane) holds
 *             W[16 nb + m][32 ks + 8 kq .. + 7]).  A(e, g) = [V(e, g) rows 0..95 | T(e, g) rows 96..111] and S(g)
 *             [96 rows] are ONE k-step over the 16 slots: k = 8 q + j holds slot 4 q + j for j < 4 and zero for
 *             j >= 4.  lin_l2(e, g) is [96 x 96] in three k-steps, column k = the lin_l2 column of aggregate
 *             channel GGNN_CELL_P3_CHANNEL(k).
 *   w2_tail : [3][n_in][6][64] fp32: (b_l2, w_edge) of (g, e) as v_mfma_f32_16x16x4_f32 A fragments
 *             ([ct][l]: k = l >> 4; k = 0 -> b_l2[16 ct + (l & 15)], k = 3 -> w_edge[16 ct + (l & 15)], else 0)
 *   h_out, c_out : [n_dst, 96]
 *   flags   : optional int32 device word (GGNN_FLAG_F16_RANGE) */
/* column k = 32 ks + 8 kq + j of a lin_l2 block <-> aggregate channel 32 ks + 16 (j / 4) + 4 kq + j % 4 */
#define GGNN_CELL_P3_CHANNEL(k) (32 * ((k) / 32) + 16 * (((k) % 8) / 4) + 4 * (((k) % 32) / 8) + (k) % 4)
typedef struct ggnn_enc_cell_sweep {
  const int32_t* rowptr;   /* [n_dst + 1] */
  const float* einfo;      /* [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW] */
  int64_t E;
} ggnn_enc_cell_sweep;
typedef struct ggnn_enc_cell_args {
  ggnn_enc_cell_sweep in[2];
  const float* x_dst;   /* [n_dst, ldx] */
  float* h_out;         /* [n_dst, 96] */
  float* c_out;         /* [n_dst, 96] */
  const void* wstream;  /* [3 * (4 n_in + 1)][GGNN_DC_SLICE_BYTES] */
  const float* w2_tail; /* [3][n_in][6][64] */
  int32_t* flags;       /* optional */
  int64_t n_dst, ldx;
  int32_t n_in, f_dst;
} ggnn_enc_cell_args;
int ggnn_encoder_cell_batch(const ggnn_enc_cell_args* args, int n_problems, ggnn_stream_t stream);

/* Decoder HeteroPGCLSTM cell (h, c from the encoder; heteropgclstm.py:101-183 with the PeriodConv of
 * periodGATconv.py:204-236) with EVERYTHING that belongs to a destination node in one kernel: replaces the
 * destination-side columns of ggnn_project_batch (u_h, u4, S), ggnn_period_gat_aggregate_batch and
 * ggnn_lstm_epilogue_batch(GGNN_MODE_LSTM) -- same sums up to fp32 re-association.  Only the source-side value
 * rows V (one ggnn_project_batch with the value rows alone) go through memory; the score operands, the
 * aggregates and the gates' pre-activations never leave the compute unit.
 *
 * A workgroup of eight waves owns 128 consecutive destination nodes, one 16-node tile per wave, and walks the
 * gates in the order i, c~, f, o (the LSTM update is folded in as the gates arrive).  Per gate g and incoming
 * edge type e a wave (P1) multiplies its tile's [h | x | 1] rows with the (e, g) score weights -> u_h | u4 of its
 * 16 nodes, (P2) sweeps the tile's in-edges of that edge type (gathers of h_src and V rows, periodic min-image
 * correction, online-max softmax, relu, alpha-weighted sum: exactly ggnn_period_gat_aggregate's arithmetic),
 * (P3) multiplies the 16 x 98 aggregate block with lin_l2 | (b_l2, w_edge) of (e, g) into the gate's
 * pre-activation, then (P4) adds the summed skip term of the gate.  Arithmetic of the three GEMMs: every fp32
 * operand as TWO fp16 pieces, hi = rne16(x) and lo' = rne16((x - hi) * 2048), and three of the four products
 * (hi hi + (hi lo' + lo' hi) / 2048) accumulated in fp32 on v_mfma_f32_16x16x32_f16: 22 significand bits per
 * operand, against an fp64 product 5e-8 of sum |x||w| (a plain fp32 fma chain: 2e-7); the rank-1 columns
 * (b_l2, w_edge) run on one exact fp32 MFMA.  The weights arrive as k-step slices of host-side pre-split planes
 * through a double-buffered LDS region shared by the eight waves (LDS-DMA).
 * OPERAND RANGE of the two-piece arithmetic: finite and |x| < 65504 for the weights, the tile's [h | x] rows and the
 * aggregates; the residual of an operand below 2^-13 in magnitude is subnormal (absolute resolution 2^-36).  A
 * weight outside the range cannot be packed (packing.split2_f16 refuses; packing.pack_cell then leaves the fused
 * operands out and the cell runs on projection + sweeps + gate GEMM, whose bf16 x 3 split covers fp32's range); an
 * activation outside it is clamped to +-65504 AND reported: the kernel ORs GGNN_FLAG_F16_RANGE into *flags (when
 * given), which graingraphnn_amd's backend reads at the end of a rollout / on request (range_exceeded).
 * FINITE OPERANDS are assumed by the sweep (round 6): it folds a row's in-edges three at a time WITHOUT a branch, the
 * slots behind the row's last edge filled with the (clamped) gather of a neighbouring edge and weighted with an exact
 * zero -- bit-identical to skipping them while the gathered hidden / value rows are finite; a NaN or inf in ANOTHER row
 * of h_src / v_src can then reach this row one step earlier than message passing would carry it (0 x inf).  The
 * flag word reports non-finite operands either way.
 *
 * Per incoming edge type:
 *   rowptr, col : destination-grouped CSR of the edge type (ggnn_build_csr)
 *   einfo       : edge records in CSR order (ggnn_edge_prepare)
 *   h_src       : [n_src, ldh_src] hidden state of the source node type
 *   v_src       : [n_src, ldv] projection of the source node type; columns v_off + g * 96 .. + 95 = the value
 *                 rows of gate g for this edge type (first three input columns zeroed, as for the sweep)
 *   edge_params : [4][GGNN_EDGE_PARAM_ROWS][96] = W_value[:, 0..2] per gate
 * per problem (one destination node type of one model):
 *   x_dst [n_dst, ldx], h_dst [n_dst, ldh] (the encoder's h), c_in [n_dst, 96]; h_out, c_out [n_dst, 96]
 *   wstream : the weight slices in the order the kernel consumes them (packing.decoder_cell_stream):
 *             for g in (i, c~, f, o): for e: 4 slices P1(e, g) | 3 slices P3(e, g); then 4 slices P4(g) -- with e running
 *             FORWARDS over the incoming edge types for the first and third gate and BACKWARDS for the second and fourth
 *             (a pass re-gathers the hidden rows and edge records the pass before it gathered while L2 still holds them).
 *             Every slice is GGNN_DC_SLICE_BYTES: [column tile nb][plane hi, lo'][64 lanes][8 fp16] -- the two
 *             fp16 pieces of a weight w are hi = rne16(w) and lo' = rne16((w - hi) * 2048) (finite, |w| < 65504) --
 *             lane l = 16 kq + m of (nb, plane) holds W[16 nb + m][32 ks + 8 kq .. + 7]; P1 has 7 column tiles
 *             (u_h 0..95 | u4 96..111), P3 / P4 six (the tail of their slice is unused).  The reduction index
 *             of P1 / P4 is [h 0..95 | x 0..f_dst-1 | 1 (bias) | 0 ..] padded to 128, of P3 the 96 aggregate channels.
 *             The kernel splits its own operands the same way (OPERAND RANGE above).
 *   w2_tail : [4][n_in][6][64] fp32: (b_l2, w_edge) of (g, e) as v_mfma_f32_16x16x4_f32 A fragments
 *             ([ct][l] = k < 2 ? tail[16 ct + (l & 15)][k = l >> 4] : 0)
 *   flags   : optional int32 device word, see OPERAND RANGE
 * Gates are indexed i, f, c, o (GGNN_MODE_LSTM's order) in wstream's g, w2_tail, edge_params and the V columns.
 * n_src * ld < 2^31 for every gathered operand; up to four problems per call. */
#define GGNN_PRECISION_BF16 1
#define GGNN_PRECISION_F16X2 2
#define GGNN_OUT_BLOCK_MAJOR 0x100 /* or-ed into ggnn_project_args.precision: see there */
#define GGNN_DC_SLICE_BYTES 14336 /* 7 column tiles x 2 planes x 1 KB */
#define GGNN_FLAG_F16_RANGE 1     /* an activation at or beyond +-65504 was clamped in a two-piece fp16 split */
typedef struct ggnn_dec_cell_sweep {
  const int32_t* rowptr;     /* [n_dst + 1] */
  const int32_t* col;        /* [E] source node of every edge, CSR order */
  const float* einfo;        /* [E + GGNN_UNIT_EDGES, GGNN_EINFO_ROW] */
  const float* h_src;        /* [n_src, ldh_src] */
  const float* v_src;        /* [n_src, ldv] */
  const float* edge_params;  /* [4][GGNN_EDGE_PARAM_ROWS][96] */
  int64_t E, n_src, ldh_src, ldv;
  int32_t v_off;         /* first of the four gates' 96 value columns inside a v_src row (a multiple of 96 when block-major) */
  int32_t v_block_major; /* ABI 24: 0 = v_src is [n_src, ldv]; 1 = v_src is GGNN_OUT_BLOCK_MAJOR: [blocks][n_src][96], the
                            four gates' blocks v_off / 96 .. + 3 (ldv is ignored) */
} ggnn_dec_cell_sweep;
typedef struct ggnn_dec_cell_args {
  ggnn_dec_cell_sweep in[2];
  const float* x_dst;    /* [n_dst, ldx] */
  const float* h_dst;    /* [n_dst, ldh] */
  const float* c_in;     /* [n_dst, 96] */
  float* h_out;          /* [n_dst, 96] */
  float* c_out;          /* [n_dst, 96] */
  const void* wstream;   /* [n_slices][GGNN_DC_SLICE_BYTES], n_slices = 4 * (7 n_in + 4) */
  const float* w2_tail;  /* [4][n_in][6][64] */
  int32_t* flags;        /* optional: range flag word */
  int64_t n_dst, ldx, ldh;
  int32_t n_in, f_dst;
} ggnn_dec_cell_args;
int ggnn_decoder_cell_batch(const ggnn_dec_cell_args* args, int n_problems, ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Training path: C[b] = A[b] . W[b]^T (+ C_in[b]) for a TALL A and a SMALL W -- the gate GEMM of a cell
 * (z_g = agg_g W2_g^T), its input gradient (g_agg_g = g_z_g W2_g) and the hidden-state gradient
 * (g_h += gP Wp[:, h columns]) of graingraphnn_amd/training.py, which replaces torch.bmm / torch.addmm there.
 *   a        : element (b, m, k) at a + b * a_bstride + m * lda + k        (M rows, K % 32 == 0, lda % 4 == 0)
 *   w        : element (b, n, k) at w + b * w_bstride + n * w_nstride + k * w_kstride  (n_out <= 224, n_out % 16 == 0;
 *              either orientation of a stored matrix: a gradient uses the transpose of the forward's weight)
 *   c, c_in  : element (b, m, n) at c + b * c_bstride + m * ldc + n; c_in optional (may equal c)
 *   precision: 0 = fp32-equivalent (two fp16 pieces per operand, three MFMA products; |w| < 65504; the rows of `a` may be
 *              ANY finite fp32 values -- every row is scaled by a power of two taken from its largest magnitude before it
 *              is split and the accumulators carry the factor, so gradient rows of 1e-10 keep the resolution activations
 *              of 1 have; a row holding an inf or a NaN gives NaN outputs, as an fp32 product would);
 *              GGNN_PRECISION_BF16 = one bf16 product, fp32 accumulation (what torch.autocast(bfloat16) defines)
 *   workspace: ggnn_rowgemm_workspace_bytes(K, n_out, batch) bytes, 16-byte aligned: the weights as MFMA operand
 *              planes (they change with every optimizer step).  prepacked == 0: the call packs them itself (two launches:
 *              pack, product); prepacked != 0: `workspace` already holds the planes of exactly this (w, K, n_out, batch,
 *              precision) from ggnn_rowgemm_pack -- one launch packs the weights of up to GGNN_ROWGEMM_MAX_PACK products
 *              (a training step packs the five products of a cell at once), `w` is not read.  (Not for fp32 products
 *              with n_out > 128 and K > 128, which run as two passes over column halves.)
 * a, c, c_in 16-byte aligned.  The product keeps a wave's 16 x n_out output tile in registers; the weight planes stay in
 * LDS for the whole launch (K <= 128, or <= 256 with n_out = 96) or stream through it. */
typedef struct ggnn_rowgemm_args {
  const float* a;
  const float* w;
  float* c;
  const float* c_in;
  void* workspace;
  size_t workspace_bytes;
  int64_t M, lda, ldc, a_bstride, c_bstride, w_bstride, w_nstride, w_kstride;
  int32_t K, n_out, batch, precision;
  int32_t prepacked, reserved;
} ggnn_rowgemm_args;
#define GGNN_ROWGEMM_MAX_PACK 8
size_t ggnn_rowgemm_workspace_bytes(int32_t K, int32_t n_out, int32_t batch);
int ggnn_rowgemm_pack(const ggnn_rowgemm_args* args, int n_products, ggnn_stream_t stream); /* reads w, K, n_out, batch, precision, workspace */
int ggnn_rowgemm(const ggnn_rowgemm_args* args, ggnn_stream_t stream);
/* ABI 25: TWO long products (args[0], args[1]: the streamed form -- K / 32 beyond what stays in LDS --, batch == 1, prepacked
 * planes, the same precision and n_out rounded to the same instantiated width) side by side in one grid: a wave walks the whole
 * reduction of its 16 rows, so such a product is M / 128 workgroups -- the hidden-state gradients of a training cell's two node
 * types (79 and 157 workgroups at the 10k-grain graph) then take the time of the longer one.  Same results as two calls;
 * anything else is refused (GGNN_EINVAL: make two calls). */
int ggnn_rowgemm_pair(const ggnn_rowgemm_args* args, ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Output heads.
 * Regressor (models.py:433-452): y_joint = tanh(W_j h_j + b_j); y_grain = W_g h_g + b_g;
 * grain_area = tanh(y_grain[:,0])/20 + x_grain[:,3]; y_grain[:,0] = tanh; y_grain[:,1] = relu.
 *   w : [2][2][96] (joint rows, then grain rows), b : [2][2].
 */
int ggnn_heads_regressor(const float* h_joint, int64_t n_joint, const float* h_grain,
                         int64_t n_grain, const float* x_grain, int64_t ldx_grain,
                         const float* w, const float* b, float* y_joint, float* y_grain,
                         float* grain_area, ggnn_stream_t stream);
/* ggnn_heads_regressor followed by ggnn_step_update (below) in ONE launch: the heads write y_joint / y_grain /
 * grain_area (grain_area from the area the forward saw), then the same thread applies Rmodel.update and the z
 * advance to its node.  For a rollout step in which nothing reads x after this point (GrainRollout's
 * two-stream plan orders it behind the classifier's last reader of x). */
int ggnn_heads_regressor_update(const float* h_joint, int64_t n_joint, const float* h_grain,
                                int64_t n_grain, float* x_joint, int64_t ldx_joint, float* x_grain,
                                int64_t ldx_grain, int f_grain, const float* w, const float* b,
                                float* y_joint, float* y_grain, float* grain_area, float dz, float zmax,
                                int32_t* flags, ggnn_stream_t stream);
/* Classifier (models.py:595-609): pair = [h_j[src] | h_j[dst] | edge_attr];
 * edge_event = lin2(pair); edge = tanh(lin1(pair)).  Computed as per-node partial dots
 * (node_tmp [n_joint, 8] scratch) + a per-edge combine in the original COO order.
 *   w_node : [6][96] = lin1.w[0,0:96], lin1.w[1,0:96], lin2.w[0,0:96], lin1.w[0,96:192], lin1.w[1,96:192], lin2.w[0,96:192]
 *   w_edge : [6]     = lin1.w[0,192], lin1.w[1,192], lin2.w[0,192], lin1.b[0], lin1.b[1], lin2.b[0]
 */
int ggnn_heads_classifier(const float* h_joint, int64_t n_joint, const int64_t* edge_index_jj,
                          int64_t E, const float* edge_attr_jj, const float* w_node,
                          const float* w_edge, float* node_tmp, float* edge_event, float* edge,
                          ggnn_stream_t stream);
/* ABI 25: the same with the number of edges read from device memory when the kernels run (E_dev != NULL: edge_index_jj is
 * [2, *E_dev], E the capacity the launch is sized for; see ggnn_prepare_edge.E_dev). */
int ggnn_heads_classifier_n(const float* h_joint, int64_t n_joint, const int64_t* edge_index_jj,
                            int64_t E, const int64_t* E_dev, const float* edge_attr_jj, const float* w_node,
                            const float* w_edge, float* node_tmp, float* edge_event, float* edge,
                            ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Rollout-step glue on device.
 * ggnn_step_update = GrainNN_regressor.update, periodic branch (models.py:503-516) + the z
 * advance of test.py:401-402:  x_j[:, :2] += y_j/5; x_j[:, 6:8] = y_j; x_g[:, 3] += y_g0/20;
 * x_g[:, 4] = y_g1; x_g[:, f_grain-1] = y_g0; z (column 2 of both) += dz.  flags[1] is set
 * to (x_grain[0, 2] > zmax) after the advance.
 * ggnn_step_refresh = the clamp of test.py:405-407 (if flags[1]: z = zmax on every node)
 * + the edge-length refresh of test.py:562-575 for up to three edge types:
 * edge_attr[e] = || min-image(src_xy - dst_xy) ||_2 in the original COO order.
 */
int ggnn_step_update(float* x_joint, int64_t n_joint, int64_t ldx_joint, float* x_grain,
                     int64_t n_grain, int64_t ldx_grain, int f_grain, const float* y_joint,
                     const float* y_grain, float dz, float zmax, int32_t* flags,
                     ggnn_stream_t stream);
/* Grain-centre refresh: replaces traj.GNN_update -> graph.update()
 * (graph_trajectory.py:1010-1085, graph_datastruct.py:681-708; periodic BC) + the write-back of
 * test.py:556-559.  For every grain with >= 2 junctions (CSR row of the joint->grain edge type
 * from ggnn_build_csr): junction xy, brought to the global frame ((x + domain_offset[j]) /
 * domain_factor when domain_factor > 1, test.py:474), are chained by min-image to the previous
 * junction, shifted by +1 in a coordinate where any of them is below -1e-12, and averaged;
 * x_grain[g, 0:2] = centre, or frac(centre * domain_factor) when domain_factor > 1.
 * fp32 (the reference's numpy scalars promote to fp64: differences are <= 1 ulp of fp32).
 * domain_offset: [n_joint, 2] or NULL (= 0).  Runs between ggnn_step_update and
 * ggnn_step_refresh so that the refreshed edge lengths see the new centres.
 * centres_before (ABI 24): NULL, or [n_grain, 2] that receives x_grain[:, 0:2] as the call found them (what a topological
 * event of this step must see: the speculative event loop keeps it per step instead of copying the columns out). */
int ggnn_grain_centres(const int32_t* rowptr, const int32_t* col, const float* x_joint,
                       int64_t n_joint, int64_t ldx_joint, const float* domain_offset,
                       float domain_factor, float* x_grain, int64_t n_grain, int64_t ldx_grain,
                       float* centres_before, ggnn_stream_t stream);
/* Event detection for the host-side topology update (test.py:418, models.py:624-626):
 * flags[0] = number of grains with live_grain > 0 and grain_area < area_threshold,
 * flags[1] = number of junction-junction edges with src < dst and edge_event (a logit) >
 * logit_threshold.  flags: [2] int32 device words (zeroed by the call).
 * range_word (ABI 24; NULL = none): an OPERAND RANGE word (below) of the step whose predictions these are -- flags must
 * then hold THREE words: flags[2] = the word's value, and the word is cleared for its next use (the speculative event loop
 * reads a step's counts and its range report in one copy and drops the report of a step it voids: rollout.py). */
int ggnn_detect_events(const float* grain_area, const int32_t* live_grain, int64_t n_grain,
                       float area_threshold, const float* edge_event, const int64_t* edge_index_jj,
                       int64_t E, float logit_threshold, int32_t* flags, int32_t* range_word, ggnn_stream_t stream);
/* ABI 25: the same with the number of junction edges read from device memory when the kernel runs (E_dev != NULL:
 * edge_index_jj is [2, *E_dev], E the capacity the launch is sized for; see ggnn_prepare_edge.E_dev). */
int ggnn_detect_events_n(const float* grain_area, const int32_t* live_grain, int64_t n_grain,
                         float area_threshold, const float* edge_event, const int64_t* edge_index_jj,
                         int64_t E, const int64_t* E_dev, float logit_threshold, int32_t* flags,
                         int32_t* range_word, ggnn_stream_t stream);
/* The host-side topology update those counts trigger: one call of the reference's `Cmodel.update`
 * (models.py:612-842 with delete_grain_index :861-893, switching_edge_index :896-1051, point_in_triangle :1055-1070,
 * periodic_move :1103-1106), nucleation off.  HOST memory throughout, no stream: grains of `grain_event` (those below the
 * area threshold, smallest first, test.py:418-420) are eliminated, junction edges with edge_prob > threshold are switched
 * (most probable first), grains left with two junctions are removed.  Bit-exact contract: the edge lists keep the
 * reference's COLUMN ORDER (columns rewritten in place, new columns appended, dead columns dropped at the end), masks and
 * fp32 junction coordinates are the reference's.
 *   pp, pq      : [2][cap] int64 junction->junction / junction->grain lists (row r at pp + r * pp_cap), n_pp / n_pq columns in
 *                 use; updated IN PLACE, n_pp / n_pq = the new column counts.  pp_cap >= n_pp + 2 (removed grains): every
 *                 removed grain appends two columns before the dead ones are dropped.
 *   x_joint     : [n_joint, ldx >= 8] fp32, columns 0, 1 (position) and 6, 7 (displacement feature) are rewritten for the
 *                 junctions an event moves; y_joint [n_joint, 2] likewise
 *   y_grain_area: element g at y_grain_area[g * ldyg] (the regressor's y_grain[:, 0]);  edge_prob [n_pp] = sigmoid(edge_event)
 *   mask_grain / mask_joint : [n] int64, 1 = live; eliminated grains and their junctions are cleared
 *   active_grain / active_joint : optional [n] bytes, 0 = outside the active window (models.py:640-645, 911): frozen
 *   switching   : out [switching_cap][2] the switched junction pairs, n_switching of them
 *   events_extra: out [extra_cap] grains eliminated beyond `grain_event` (forced / two-sided), n_extra of them
 * Returns GGNN_OK, GGNN_EINVAL, or GGNN_ETOPOLOGY with a message in `error` (the reference asserts / raises there); on an error
 * the in/out arrays are in an undefined state: pass copies (graingraphnn_amd/topology.py does).  The room of the output lists
 * (switching_cap >= the edges above the threshold, extra_cap >= n_grain + 1) is checked before anything is rewritten.
 * Refusals where the reference would raise an IndexError of its own (reachable on degenerate lists only): a junction with
 * fewer than three grain columns or fewer than two other neighbours in a switch.  (A junction pair of an eliminated grain
 * joined by MORE than one column is not refused: the reference indexes the concatenated hits with the order of the edges all
 * the same, and so does this -- rounds 5-6 refused there, one step before the reference's formulation failed by itself.) */
typedef struct ggnn_topology_args {
  int64_t* pp;
  int64_t* pq;
  int64_t n_pp, n_pq, pp_cap, pq_cap;
  float* x_joint;
  float* y_joint;
  const float* y_grain_area;
  const float* edge_prob;
  const int64_t* grain_event;
  int64_t* mask_grain;
  int64_t* mask_joint;
  const uint8_t* active_grain;
  const uint8_t* active_joint;
  int64_t* switching;
  int64_t* events_extra;
  int64_t n_joint, n_grain, ldx, ldyg, n_grain_event, switching_cap, extra_cap, n_switching, n_extra;
  double threshold;
  char error[192];
} ggnn_topology_args;
int ggnn_topology_update(ggnn_topology_args* args);
/* The same update on lists that live in the library between calls (ABI 24; graingraphnn_amd/rollout.py keeps one session per
 * trajectory): the reference's loop calls Cmodel.update at EVERY step (test.py:418-426), and a stateless call rebuilds its
 * lookup tables from the whole lists each time (four O(E) sorts for a handful of events).  A session keeps the lists, the
 * tables and the per-grain counts and patches them as the events rewrite columns; a call costs what its events touch plus
 * one pass that renumbers the live columns.  HOST memory, no stream, one thread per session at a time.
 *   ggnn_topology_open   : pp [2][n_pp] (row r at pp + r * pp_ld), pq likewise -> *session.  GGNN_ETOPOLOGY (message in
 *                          `error`, 192 bytes, may be NULL) for an index outside [0, n_joint) / [0, n_grain).
 *   ggnn_topology_apply  : one update; of `args` pp / pq / *_cap are ignored (the session holds the lists), everything else
 *                          as for ggnn_topology_update; n_pp / n_pq = the new column counts.  A refused call
 *                          (GGNN_ETOPOLOGY) leaves the session AND x_joint / y_joint / the masks exactly as they were.
 *   ggnn_topology_export : the current lists in the reference's column order into the caller's arrays (any may be NULL):
 *                          pp [2][>= n_pp], pq [2][>= n_pq], qp = pq with its rows exchanged (the grain -> junction list,
 *                          models.py:841).
 *   ggnn_topology_counts, ggnn_topology_close. */
typedef struct ggnn_topology_session ggnn_topology_session;
int ggnn_topology_open(const int64_t* pp, int64_t n_pp, int64_t pp_ld, const int64_t* pq, int64_t n_pq, int64_t pq_ld,
                       int64_t n_joint, int64_t n_grain, ggnn_topology_session** session, char* error);
int ggnn_topology_apply(ggnn_topology_session* session, ggnn_topology_args* args);
int ggnn_topology_counts(const ggnn_topology_session* session, int64_t* n_pp, int64_t* n_pq);
int ggnn_topology_export(const ggnn_topology_session* session, int64_t* pp, int64_t pp_ld, int64_t* pq, int64_t pq_ld,
                         int64_t* qp, int64_t qp_ld);
void ggnn_topology_close(ggnn_topology_session* session);
typedef struct ggnn_refresh_edge {
  const int64_t* edge_index; /* [2, E] */
  const float* x_src;
  const float* x_dst;
  float* edge_attr; /* [E] out */
  int64_t ldx_src, ldx_dst, n_src, n_dst, E;
  const int64_t* E_dev;      /* ABI 25, or NULL: as in ggnn_prepare_edge (edge_index is [2, *E_dev] then) */
} ggnn_refresh_edge;
int ggnn_step_refresh(float* x_joint, int64_t n_joint, int64_t ldx_joint, float* x_grain,
                      int64_t n_grain, int64_t ldx_grain, float zmax, const int32_t* flags,
                      const ggnn_refresh_edge* edges, int n_edge_types, ggnn_stream_t stream);

/* ggnn_step_refresh + ggnn_edge_prepare of the NEXT forward in ONE launch: z clamp of every node when
 * flags[1] is set, then per CSR slot the edge length from the min-image xy offsets (written to
 * edge_attr[perm[p]], the COO order: edges[k].edge_attr is an OUTPUT here) and the einfo record with that
 * length.  Same values as the two calls in sequence (the length is the same expression on the same operands).
 * x_joint_mirror / x_grain_mirror (ABI 24; both or neither, NULL = none; same shape and leading dimension as x_joint /
 * x_grain, not x itself): every row of x as it stands behind this call (the clamped z included) is also written there --
 * the copy of the node features that the classifier's forward of the NEXT step reads (test.py:382-383 hands both models
 * the same x_dict) while Rmodel.update of that step already rewrites x in place (graingraphnn_amd/rollout.py). */
int ggnn_step_refresh_prepare(float* x_joint, int64_t n_joint, int64_t ldx_joint, float* x_grain,
                              int64_t n_grain, int64_t ldx_grain, float zmax, const int32_t* flags,
                              const ggnn_prepare_edge* edges, int n_edge_types, float* x_joint_mirror,
                              float* x_grain_mirror, ggnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Training path: the LSTM update of HeteroPGCLSTM.forward (heteropgclstm.py:140-183) with the
 * state its backward needs, and that backward.  Replaces what autograd records for the sigmoid / tanh /
 * multiply / add chain of `i, f, c, o` (one launch each way instead of ~40).  n_gates = 4: (i, f, c, o) with
 * c_in; n_gates = 3: (i, c, o), zero state (encoder), c_in = NULL.
 *   forward : z [n_gates, N, 96] holds the gate GEMM's output (lin_l2 part) on entry; on exit the
 *             pre-activations z_g = gemm_g + p_dst[n, s_off + g*96 ..] (skip rows of ggnn_project); writes
 *             h_out, c_out [N, 96].
 *   backward: g_h / g_c (gradients of h_out / c_out; NULL = zero) -> g_z [n_gates, N, 96], the same values
 *             into g_p_dst[n, s_off + g*96 ..] (gradient of the skip rows; NULL = skip), g_c_in (NULL = skip).
 * ldp % 4 == 0, s_off % 4 == 0, every pointer 16-byte aligned. */
int ggnn_lstm_train_forward(float* z, const float* p_dst, int64_t ldp, int s_off, const float* c_in,
                            float* h_out, float* c_out, int64_t N, int n_gates, ggnn_stream_t stream);
int ggnn_lstm_train_backward(const float* z, const float* c_in, const float* c_out, const float* g_h,
                             const float* g_c, float* g_z, float* g_p_dst, int64_t ldp, int s_off,
                             float* g_c_in, int64_t N, int n_gates, ggnn_stream_t stream);
/* ABI 25: the same two updates for the node types of a cell (1..GGNN_LSTM_TRAIN_MAX problems of one gate count) in ONE launch
 * each way.  The forward reads (z, p_dst, ldp, s_off, c_in) and writes (z, h_out, c_out); the backward reads (z, c_in,
 * c_out, g_h, g_c) and writes (g_z, g_p_dst, g_c_in), fields as in the single calls.  Backward only: it also writes zeros
 * into g_p_dst[n, pad_off .. pad_off + pad_n) (the projection gradient's padding columns, which no kernel of the cell's
 * backward writes and the weight gradient reads: pad_off % 4 == 0, pad_n % 4 == 0, pad_n <= 96; 0 = nothing). */
#define GGNN_LSTM_TRAIN_MAX 4
typedef struct ggnn_lstm_train_problem {
  float* z;
  const float* p_dst;
  const float* c_in;
  float* h_out;
  float* c_out;
  const float* g_h;
  const float* g_c;
  float* g_z;
  float* g_p_dst;
  float* g_c_in;
  int64_t ldp, N;
  int32_t s_off, pad_off, pad_n, reserved;
} ggnn_lstm_train_problem;
int ggnn_lstm_train_forward_batch(const ggnn_lstm_train_problem* problems, int n_problems, int n_gates, ggnn_stream_t stream);
int ggnn_lstm_train_backward_batch(const ggnn_lstm_train_problem* problems, int n_problems, int n_gates, ggnn_stream_t stream);

/* Training path: backward of ggnn_heads_regressor.  From the gradients of (y_joint [n_joint, 2], y_grain [n_grain, 2],
 * grain_area [n_grain]; any may be NULL = zero) and the forward's saved y_joint / y_grain: g_pre_* [n, 4] = gradient of
 * the heads' pre-activations (columns 2-3 zero: the operand of ggnn_wgrad for the head weights and biases) and
 * g_h_* [n, 96] = g_pre W.  w as in the forward. */
int ggnn_heads_regressor_backward(int64_t n_joint, int64_t n_grain, const float* w, const float* y_joint,
                                  const float* y_grain, const float* g_y_joint, const float* g_y_grain,
                                  const float* g_grain_area, float* g_pre_joint, float* g_pre_grain,
                                  float* g_h_joint, float* g_h_grain, ggnn_stream_t stream);

/* Weight-gradient GEMM of the training path: C[b] = A[b]^T B[b] with a long reduction (K = nodes) and a small
 * M x Nc result -- the gradient of a packed projection ([ncols, F + 97]) or gate ([96, Kg] per gate) weight
 * matrix, replacing the BLAS call autograd would make for x.t() @ g (which does not split K).  The reduction is
 * split over the chip in a fixed way; the call writes partial[s][b][M][Nc], s < ggnn_wgrad_splits(K, M, Nc, batch),
 * and sums over s into `out` (in index order; out == NULL: left to the caller).  Arithmetic: fp32-equivalent over fp32's
 * whole range either way -- exact fp32 products (v_mfma_f32_16x16x4_f32) for the short results, and for the TALL ones
 * (M >= 512: the packed projection's gradient) under the default GEMM mode both operands as three exact bf16 pieces with
 * six products per k-step (2e-8 of sum |a||b| against an fp64 product; gradients of 1e-10 keep their 24 bits, which a
 * two-piece fp16 split would not give them).  M, Nc, lda, ldb, a_bstride, b_bstride multiples of 4 (pad B with zero columns
 * otherwise); a, b 16-byte aligned.
 * ABI 25, b_ins != NULL: B is `b` with the ins_w columns of `b_ins` [K, ld_ins] INSERTED at column ins_off -- column c of B is
 * b[c] for c < ins_off, b_ins[c - ins_off] for the next ins_w columns, b[c - ins_w] behind them (Nc counts all of them; b holds
 * Nc - ins_w columns) -- so that the weight gradient of a projection reads [x | 0 | h | 1 0 0 0] from the data skeleton
 * [x | 0 | 1 0 0 0] (ggnn_train_input_rows, once per step) and the hidden state where it lies, without a concatenated copy.
 * ins_off, ins_w, ld_ins multiples of 4, b_ins 16-byte aligned, batch == 1. */
typedef struct ggnn_wgrad_args {
  const float* a;  /* [batch] x [K, lda] row-major, first M columns used; batch b starts at a + b * a_bstride */
  const float* b;  /* [batch] x [K, ldb], first Nc columns used; batch b starts at b + b * b_bstride */
  float* partial;  /* [n_split, batch, M, Nc] out */
  float* out;      /* [batch, M, Nc] out: the sum over the splits, or NULL (16-byte aligned) */
  int64_t lda, ldb, a_bstride, b_bstride, K;
  int32_t M, Nc, batch, n_split;
  const float* b_ins;  /* or NULL */
  int64_t ld_ins;
  int32_t ins_off, ins_w;
} ggnn_wgrad_args;
int ggnn_wgrad_splits(int64_t K, int M, int Nc, int batch);
int ggnn_wgrad(const ggnn_wgrad_args* args, ggnn_stream_t stream);

/* Training path (ABI 25): the data part of the weight gradients' B operands, for the node types of a step in one launch:
 * out[n, 0 .. Fp + 4) = [x[n, 0 .. F) | 0 .. (Fp - F zeros) | 1 0 0 0], Fp = F rounded up to 4 -- what the encoder's projection
 * gradient multiplies with as it is, and the decoder's with the hidden state inserted at column Fp (ggnn_wgrad, b_ins).
 * out 16-byte aligned, ldo >= Fp + 4, ldo % 4 == 0; 3 <= F <= 12. */
#define GGNN_TRAIN_ROWS_MAX 4
typedef struct ggnn_train_rows_problem {
  const float* x;
  float* out;
  int64_t ldx, ldo, N;
  int32_t F, reserved;
} ggnn_train_rows_problem;
int ggnn_train_input_rows(const ggnn_train_rows_problem* problems, int n_problems, ggnn_stream_t stream);

/* Training path: out[b][j] = sum_r in[b][r][j], b < batch -- the caller's reduction over the partial sums of
 * ggnn_period_gat_aggregate_backward (ep_partial: n_rows = ggnn_aggregate_bwd_partials rows of n_cols = n_gates * 288 floats
 * per edge type), in a fixed order.  in: [batch, n_rows, n_cols] contiguous, n_cols % 4 == 0, 16-byte aligned. */
int ggnn_sum_rows(const float* in, float* out, int64_t n_rows, int64_t n_cols, int32_t batch, ggnn_stream_t stream);
/* ABI 25: 1..GGNN_SUM_ROWS_MAX such sums in one launch, each with its own shape -- the reductions a cell's backward pass can
 * postpone to its end: the split-K partials of its weight gradients (ggnn_wgrad with out == NULL: in = partial, n_rows =
 * n_split, n_cols = batch M Nc, batch = 1) and the edge-parameter partials of its sweeps.  Same fixed summation tree as
 * ggnn_sum_rows (32 interleaved row groups, combined in index order). */
#define GGNN_SUM_ROWS_MAX 8
typedef struct ggnn_sum_rows_problem {
  const float* in;   /* [batch, n_rows, n_cols] contiguous, 16-byte aligned */
  float* out;        /* [batch, n_cols] */
  int64_t n_rows, n_cols;
  int32_t batch, reserved;
} ggnn_sum_rows_problem;
int ggnn_sum_rows_batch(const ggnn_sum_rows_problem* problems, int n_problems, ggnn_stream_t stream);

/* Training path: the nine packed weight matrices of a cell (projection weights and biases of both node types, relocation
 * columns per edge type, gate weights: graingraphnn_amd/packing.py) from its parameters, differentiable -- the device side of
 * train_pack._PackWeights.  The packing is described by index tables (train_pack.PackPlan): with
 *   flat2 = [ the cell's parameters, concatenated (n_flat) | nb products [r, c] | one zero slot ],  zero = n_flat + nb r c,
 * every packed entry is the sum of L elements of flat2 (idx3 [n_packed, L]; the index `zero` reads 0), and product b is
 * (coef K_b) Q_b with K_b [r, 96], Q_b [96, c] themselves gathered from flat2: kq_idx = [K entries (nb r 96) | Q entries
 * (nb 96 c)].  ggnn_pack_weights: the caller has written the parameters to flat2[0 .. n_flat); the call fills the products
 * and `packed`.  ggnn_pack_weights_backward: g_out[s] = gradient of output s (NULL: zero), output s being
 * packed[g_off[s] .. g_off[s + 1]); inv [n_flat2, inv_m] = the packed entries that read each element of flat2 (padding:
 * n_packed), inv_kq [n_flat, inv_kq_m] = the operand entries that read each parameter (padding: n_kq = nb 96 (r + c));
 * g_flat2 [n_flat2] and g_kq [n_kq] are workspaces, g_flat [n_flat] receives the parameters' gradient.  Plain fp32 fmas. */
#define GGNN_PACK_OUTPUTS 9
typedef struct ggnn_pack_args {
  float* flat2;
  float* kq;                /* [nb 96 (r + c)] the products' operands gathered from flat2 (K side x coef): written by the forward, read by the backward */
  const int64_t* kq_idx;
  const int64_t* idx3;
  float* packed;
  int64_t n_flat, zero, n_packed;
  int32_t nb, r, c, L;
  float coef;
  int32_t reserved;
  const float* const* params;   /* ABI 25, or NULL: DEVICE array of the parameter tensors' addresses.  With it the entries of kq_idx
                                   and idx3 that address a parameter are ENCODED as ((t + 1) << GGNN_PACK_TENSOR_SHIFT) | (offset
                                   inside tensor t) and read from params[t] where the tensor lies (flat2[0 .. n_flat) is then not
                                   read: no concatenated copy of the parameters); entries >= n_flat (products, zero slot) stay
                                   plain flat2 indices.  NULL: every entry is a plain flat2 index. */
} ggnn_pack_args;
#define GGNN_PACK_TENSOR_SHIFT 40
typedef struct ggnn_pack_bwd_args {
  ggnn_pack_args fwd;                       /* as in the forward call (packed is not used) */
  const float* g_out[GGNN_PACK_OUTPUTS];
  int64_t g_off[GGNN_PACK_OUTPUTS + 1];
  int64_t g_w[GGNN_PACK_OUTPUTS], g_rs[GGNN_PACK_OUTPUTS], g_cs[GGNN_PACK_OUTPUTS];   /* element e of output s = g_out[s][(e / w) rs + (e % w) cs]: a
                                                                                         contiguous gradient is (w = its size, rs = 0, cs = 1), a column
                                                                                         block of a wider matrix (w = its width, rs = the row stride) */
  const int64_t* inv;
  const int64_t* inv_kq;
  float* g_flat2;
  float* g_kq;
  float* g_flat;
  int64_t n_flat2, n_kq;
  int32_t inv_m, inv_kq_m;
  int64_t n_tail;   /* ABI 25: g_flat has n_flat + n_tail elements; the tail is written with zeros (the gradient of the parameters
                       the reference's forward reads without effect: the encoder's forget gate) */
} ggnn_pack_bwd_args;
int ggnn_pack_weights(const ggnn_pack_args* args, ggnn_stream_t stream);
int ggnn_pack_weights_backward(const ggnn_pack_bwd_args* args, ggnn_stream_t stream);
/* ABI 25: the same for 1..GGNN_PACK_MAX cells (the encoder's and the decoder's of a training step) with every launch shared:
 * three launches forward and four backward whatever the number of cells (each of these kernels is a few microseconds of
 * latency on ~1 MB of data).  Same results as the single calls. */
#define GGNN_PACK_MAX 2
int ggnn_pack_weights_batch(const ggnn_pack_args* args, int n_cells, ggnn_stream_t stream);
int ggnn_pack_weights_backward_batch(const ggnn_pack_bwd_args* args, int n_cells, ggnn_stream_t stream);

/* Training path: torch.optim.Adam's update (train.py:82-91; amsgrad / maximize off) for up to GGNN_ADAM_MAX_TENSORS parameter
 * tensors in one launch.  `table` (DEVICE memory, n_tensors entries, built once) holds what is fixed: the addresses of a
 * parameter and its two moment buffers, its size, its parameter group.  Workgroup c of the launch updates elements
 * [chunk_index[c] * GGNN_ADAM_CHUNK, +GGNN_ADAM_CHUNK) of tensor chunk_tensor[c] (DEVICE arrays of n_chunks entries, built once
 * from the sizes).  What changes per step comes by value: grad[t] (NULL: tensor t is skipped), lr / weight_decay per group --
 * or, with `hyper` set, lr and weight_decay are read from DEVICE memory (hyper[g] = lr of group g, hyper[GGNN_ADAM_MAX_GROUPS + g]
 * = its weight_decay): a call captured in a hipGraph then follows a learning-rate schedule (train.py:91, StepLR) through
 * an uncaptured copy into that array between replays, where by-value arguments would replay the captured rates for ever.
 * step (DEVICE, n_tensors floats, 0 before the first call): step[t] = updates tensor t has had so far -- read for the bias
 * corrections and incremented by the call itself (a second small launch) where grad[t] != NULL (torch keeps a count per
 * parameter: one without a gradient does not advance), so a captured call replays correctly.  A model with more tensors
 * takes several calls per update, each on its own slice of the table and of step.  Per element:
 *   g += weight_decay p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
 *   p -= lr / (1 - b1^s) * m / (sqrt(v) / sqrt(1 - b2^s) + eps),   s = step[t] + 1. */
#define GGNN_ADAM_CHUNK 4096
#define GGNN_ADAM_MAX_TENSORS 384
#define GGNN_ADAM_MAX_GROUPS 8
typedef struct ggnn_adam_tensor {
  float* param;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;
  int32_t group, reserved;
} ggnn_adam_tensor;
typedef struct ggnn_adam_args {
  const ggnn_adam_tensor* table;
  const int32_t* chunk_tensor;
  const int32_t* chunk_index;
  float* step;
  const float* grad[GGNN_ADAM_MAX_TENSORS];
  float lr[GGNN_ADAM_MAX_GROUPS], weight_decay[GGNN_ADAM_MAX_GROUPS];
  float beta1, beta2, eps;
  int32_t n_chunks, n_tensors;
  const float* hyper;   /* optional, DEVICE: [2][GGNN_ADAM_MAX_GROUPS] lr | weight_decay per group, read instead of the by-value ones */
} ggnn_adam_args;
int ggnn_adam_step(const ggnn_adam_args* args, ggnn_stream_t stream);

/* Training path: the regressor's loss (train.py:31-37, edge_len off) with its gradient in one launch:
 *   loss = scale * sum_k mean_i(mask_k[i / mask_div_k] (pred_k[i] - target_k[i])^2),   g_pred_k[i] = d loss / d pred_k[i]
 * over n_terms <= GGNN_MSE_MAX_TERMS terms of n[k] elements (mask NULL = ones; mask_div = elements of pred per mask entry;
 * g_pred NULL = not wanted).  workspace: GGNN_MSE_BLOCKS + 1 doubles of DEVICE memory, zero before the first call (the call
 * leaves its last entry zero again).  The sum is taken in a fixed order. */
#define GGNN_MSE_MAX_TERMS 4
#define GGNN_MSE_BLOCKS 64
typedef struct ggnn_mse_args {
  const float* pred[GGNN_MSE_MAX_TERMS];
  const float* target[GGNN_MSE_MAX_TERMS];
  const float* mask[GGNN_MSE_MAX_TERMS];
  float* g_pred[GGNN_MSE_MAX_TERMS];
  int64_t n[GGNN_MSE_MAX_TERMS];
  int64_t mask_div[GGNN_MSE_MAX_TERMS];
  double* workspace;
  float* loss; /* [1] out */
  float scale;
  int32_t n_terms;
} ggnn_mse_args;
int ggnn_masked_mse(const ggnn_mse_args* args, ggnn_stream_t stream);

/* Bytes of device scratch one model forward needs (projections + aggregates + h/c), so a
 * caller can size a single arena; the Python host allocates the same amounts as tensors. */
size_t ggnn_workspace_bytes(int64_t n_grain, int64_t n_joint, int64_t E);

#ifdef __cplusplus
}
#endif
#endif /* GGNN_H_ */
