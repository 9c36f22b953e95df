// CSR build: COO edge_index [2,E] int64 -> (rowptr, col, perm, row) grouped by destination,
// plus the unit table (unit_ptr, units) the aggregation sweep walks.
// Replaces the per-call COO gather/scatter bookkeeping of PyG MessagePassing.propagate as
// used at periodGATconv.py:174-175.  Deterministic: inside a row the slots are ordered by
// original edge id, whatever order the atomics happened to land in.
#include "common.h"

namespace ggnn {

// A unit (the unit table the LDS-DMA aggregation sweep walks) is one destination row restricted to at most
// GGNN_UNIT_EDGES consecutive in-edges (rows without edges still get one empty unit so that their output is written).
// The descriptor is 32 bytes: {i, p0, nact | first<<8 | last<<9, 0, j0, j1, j2, 0}; absent edges repeat j0 (or 0) so that
// every unit issues the same number of row loads.
//
// ---- up to four lists per launch (ggnn_build_csr_batch): blockIdx.y = the list; a topological event rebuilds the
// three edge types' tables, and eight launches per list -- two of them single-workgroup scans -- were most of that ----
constexpr int CSR_MAX_BATCH = 4;
struct CsrBatch {
  ggnn_csr_args p[CSR_MAX_BATCH];
};
__device__ __forceinline__ int32_t* csr_counts(const ggnn_csr_args& P) { return reinterpret_cast<int32_t*>(P.workspace); }
__device__ __forceinline__ int32_t* csr_cursor(const ggnn_csr_args& P) { return csr_counts(P) + P.n_dst + 1; }

__global__ __launch_bounds__(256) void csr_zero_batch_kernel(const CsrBatch B) {
  const ggnn_csr_args& P = B.p[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P.n_dst) csr_counts(P)[i] = 0;
}
__global__ __launch_bounds__(256) void csr_count_batch_kernel(const CsrBatch B) {
  const ggnn_csr_args& P = B.p[blockIdx.y];
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.E) return;
  const int64_t s = P.edge_index[e], d = P.edge_index[P.E + e];
  if ((uint64_t)s >= (uint64_t)P.n_src || (uint64_t)d >= (uint64_t)P.n_dst) {
    atomicOr(P.flags, 1);
    return;
  }
  atomicAdd(&csr_counts(P)[d], 1);
}
// Single-workgroup exclusive scan over a list's n_dst counts (a few 10^4 here), one workgroup per list: writes out[0..n] and
// a copy into cursor[0..n-1] for the fill pass.  UNITS: counts -> unit_ptr, else counts -> rowptr
template <bool UNITS>
__global__ __launch_bounds__(1024) void csr_scan_batch_kernel(const CsrBatch B) {
  const ggnn_csr_args& P = B.p[blockIdx.x];
  const int32_t* __restrict__ counts = csr_counts(P);
  int32_t* __restrict__ out = UNITS ? P.unit_ptr : P.rowptr;
  int32_t* __restrict__ cursor = csr_cursor(P);
  const int64_t n = P.n_dst;
  __shared__ int32_t s_wave[16];
  __shared__ int32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + tid;
    const int32_t v = i < n ? counts[i] : 0;
    int32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int32_t wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
    const int32_t carry = s_carry;
    const int32_t excl = carry + wave_off + incl - v;
    if (i < n) {
      out[i] = excl;
      cursor[i] = excl;
    }
    __syncthreads();
    if (tid == 1023) s_carry = carry + wave_off + incl;
    __syncthreads();
  }
  if (tid == 0) out[n] = s_carry;
}
__global__ __launch_bounds__(256) void csr_fill_batch_kernel(const CsrBatch B) {
  const ggnn_csr_args& P = B.p[blockIdx.y];
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.E) return;
  const int64_t s = P.edge_index[e], d = P.edge_index[P.E + e];
  if ((uint64_t)s >= (uint64_t)P.n_src || (uint64_t)d >= (uint64_t)P.n_dst) return;
  const int32_t pos = atomicAdd(&csr_cursor(P)[d], 1);
  P.perm[pos] = (int32_t)e;
}
// One thread per destination row: order the row's slots by original edge id (rows are 3..12 long on grain graphs), resolve
// the source node of each slot, and leave the row's unit count in `counts` for the second scan
__global__ __launch_bounds__(256) void csr_sort_batch_kernel(const CsrBatch B) {
  const ggnn_csr_args& P = B.p[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_dst) return;
  const int32_t beg = P.rowptr[i], end = P.rowptr[i + 1];
  int32_t* __restrict__ perm = P.perm;
  for (int32_t a = beg + 1; a < end; ++a) {
    const int32_t key = perm[a];
    int32_t b = a - 1;
    while (b >= beg && perm[b] > key) {
      perm[b + 1] = perm[b];
      --b;
    }
    perm[b + 1] = key;
  }
  for (int32_t a = beg; a < end; ++a) {
    P.col[a] = (int32_t)P.edge_index[perm[a]];
    P.row[a] = (int32_t)i;
  }
  const int32_t deg = end - beg;
  csr_counts(P)[i] = deg == 0 ? 1 : (deg + GGNN_UNIT_EDGES - 1) / GGNN_UNIT_EDGES;
}
__global__ __launch_bounds__(256) void csr_unit_fill_batch_kernel(const CsrBatch B) {
  const ggnn_csr_args& P = B.p[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_dst) return;
  const int32_t beg = P.rowptr[i], end = P.rowptr[i + 1];
  int32_t u = P.unit_ptr[i];
  int32_t p = beg;
  do {
    const int32_t nact = min(GGNN_UNIT_EDGES, end - p);
    const int32_t first = p == beg, last = p + GGNN_UNIT_EDGES >= end;
    int32_t* d = P.units + 8 * (int64_t)u;
    const int32_t j0 = nact > 0 ? P.col[p] : 0;
    d[0] = (int32_t)i;
    d[1] = p;
    d[2] = nact | (first << 8) | (last << 9);
    d[3] = 0;
    d[4] = j0;
    d[5] = nact > 1 ? P.col[p + 1] : j0;
    d[6] = nact > 2 ? P.col[p + 2] : j0;
    d[7] = 0;
    ++u;
    p += GGNN_UNIT_EDGES;
  } while (p < end);
}

}  // namespace ggnn

extern "C" size_t ggnn_csr_workspace_bytes(int64_t E, int64_t n_dst) {
  (void)E;
  return (size_t)(2 * (n_dst > 0 ? n_dst : 0) + 2) * sizeof(int32_t);
}

extern "C" int64_t ggnn_csr_max_units(int64_t E, int64_t n_dst) {
  return (n_dst > 0 ? n_dst : 0) + (E > 0 ? E : 0) / GGNN_UNIT_EDGES + 1;
}

extern "C" int ggnn_build_csr_batch(const ggnn_csr_args* problems, int n_problems, ggnn_stream_t stream_) {
  using namespace ggnn;
  hipStream_t stream = (hipStream_t)stream_;
  if (!problems || n_problems < 1 || n_problems > CSR_MAX_BATCH) return GGNN_EINVAL;
  CsrBatch B;
  int64_t max_E = 0, max_n = 0;
  for (int k = 0; k < CSR_MAX_BATCH; ++k) {
    B.p[k] = problems[k < n_problems ? k : 0];
    if (k >= n_problems) continue;
    const ggnn_csr_args& P = B.p[k];
    if (P.E < 0 || P.n_src < 0 || P.n_dst <= 0 || !P.rowptr || !P.flags || !P.workspace) return GGNN_EINVAL;
    if (P.E > 0 && (!P.edge_index || !P.col || !P.perm || !P.row)) return GGNN_EINVAL;
    if (!P.unit_ptr || !P.units || !aligned16(P.units)) return GGNN_EINVAL;
    if (P.E >= INT32_MAX || P.n_dst >= INT32_MAX || P.n_src >= INT32_MAX) return GGNN_EINVAL;
    if (P.workspace_bytes < ggnn_csr_workspace_bytes(P.E, P.n_dst) || ((uintptr_t)P.workspace & 3)) return GGNN_EINVAL;
    max_E = P.E > max_E ? P.E : max_E;
    max_n = P.n_dst > max_n ? P.n_dst : max_n;
  }
  const unsigned ny = (unsigned)n_problems;
  const unsigned eb = (unsigned)((max_E + 255) / 256), nb = (unsigned)((max_n + 255) / 256);
  hipLaunchKernelGGL(csr_zero_batch_kernel, dim3(nb, ny), dim3(256), 0, stream, B);
  if (max_E > 0) hipLaunchKernelGGL(csr_count_batch_kernel, dim3(eb, ny), dim3(256), 0, stream, B);
  hipLaunchKernelGGL(csr_scan_batch_kernel<false>, dim3(ny), dim3(1024), 0, stream, B);
  if (max_E > 0) hipLaunchKernelGGL(csr_fill_batch_kernel, dim3(eb, ny), dim3(256), 0, stream, B);
  // (rows sorted, sources resolved, and the rows' unit counts left in `counts` for the second scan; a list without edges
  // still gets its unit counts: one empty unit per row)
  hipLaunchKernelGGL(csr_sort_batch_kernel, dim3(nb, ny), dim3(256), 0, stream, B);
  hipLaunchKernelGGL(csr_scan_batch_kernel<true>, dim3(ny), dim3(1024), 0, stream, B);
  hipLaunchKernelGGL(csr_unit_fill_batch_kernel, dim3(nb, ny), dim3(256), 0, stream, B);
  return launch_status();
}

extern "C" int ggnn_build_csr(const int64_t* edge_index, int64_t E, int64_t n_src, int64_t n_dst,
                              int32_t* rowptr, int32_t* col, int32_t* perm, int32_t* row,
                              int32_t* unit_ptr, int32_t* units, int32_t* flags, void* workspace, size_t workspace_bytes, ggnn_stream_t stream_) {
  ggnn_csr_args P;
  P.edge_index = edge_index, P.E = E, P.n_src = n_src, P.n_dst = n_dst;
  P.rowptr = rowptr, P.col = col, P.perm = perm, P.row = row, P.unit_ptr = unit_ptr, P.units = units, P.flags = flags;
  P.workspace = workspace, P.workspace_bytes = workspace_bytes;
  return ggnn_build_csr_batch(&P, 1, stream_);
}
