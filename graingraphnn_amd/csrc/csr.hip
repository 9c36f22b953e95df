// CSR build: COO edge_index [2,E] int64 -> (rowptr, col, perm, row) grouped by destination,
// plus the unit table (unit_ptr, units) the aggregation sweep walks.
// Replaces the per-call COO gather/scatter bookkeeping of PyG MessagePassing.propagate as
// used at periodGATconv.py:174-175.  Deterministic: inside a row the slots are ordered by
// original edge id, whatever order the atomics happened to land in.
#include "common.h"

namespace ggnn {

__global__ __launch_bounds__(256) void csr_count_kernel(const int64_t* __restrict__ ei, int64_t E,
                                                        int64_t n_src, int64_t n_dst,
                                                        int32_t* __restrict__ counts,
                                                        int32_t* __restrict__ flags) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = ei[e], d = ei[E + e];
  if ((uint64_t)s >= (uint64_t)n_src || (uint64_t)d >= (uint64_t)n_dst) {
    atomicOr(flags, 1);
    return;
  }
  atomicAdd(&counts[d], 1);
}

// Single-workgroup exclusive scan over n counts (n is at most a few 10^5 here); writes
// rowptr[0..n] and a copy into cursor[0..n-1] for the fill pass.
__global__ __launch_bounds__(1024) void csr_scan_kernel(const int32_t* __restrict__ counts,
                                                        int64_t n, int32_t* __restrict__ rowptr,
                                                        int32_t* __restrict__ cursor) {
  __shared__ int32_t s_wave[16];
  __shared__ int32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + tid;
    const int32_t v = i < n ? counts[i] : 0;
    int32_t incl = v;  // inclusive scan inside the wave
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
      rowptr[i] = excl;
      cursor[i] = excl;
    }
    __syncthreads();
    if (tid == 1023) s_carry = carry + wave_off + incl;
    __syncthreads();
  }
  if (tid == 0) rowptr[n] = s_carry;
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const int64_t* __restrict__ ei, int64_t E,
                                                       int64_t n_src, int64_t n_dst,
                                                       int32_t* __restrict__ cursor,
                                                       int32_t* __restrict__ perm) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = ei[e], d = ei[E + e];
  if ((uint64_t)s >= (uint64_t)n_src || (uint64_t)d >= (uint64_t)n_dst) return;
  const int32_t pos = atomicAdd(&cursor[d], 1);
  perm[pos] = (int32_t)e;
}

// One thread per destination row: order the row's slots by original edge id (rows are
// 3..12 long on grain graphs), then resolve the source node of each slot.
__global__ __launch_bounds__(256) void csr_sort_kernel(const int64_t* __restrict__ ei,
                                                       const int32_t* __restrict__ rowptr,
                                                       int64_t n_dst, int32_t* __restrict__ perm,
                                                       int32_t* __restrict__ col,
                                                       int32_t* __restrict__ row) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_dst) return;
  const int32_t beg = rowptr[i], end = rowptr[i + 1];
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
    col[a] = (int32_t)ei[perm[a]];
    row[a] = (int32_t)i;
  }
}

// ---- unit table for the LDS-DMA aggregation sweep -------------------------------------
// A unit is one destination row restricted to at most GGNN_UNIT_EDGES consecutive in-edges
// (rows without edges still get one empty unit so that their output is written).  The
// descriptor is 32 bytes: {i, p0, nact | first<<8 | last<<9, 0, j0, j1, j2, 0}; absent
// edges repeat j0 (or 0) so that every unit issues the same number of row loads.
__global__ __launch_bounds__(256) void csr_unit_count_kernel(const int32_t* __restrict__ rowptr,
                                                             int64_t n_dst,
                                                             int32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_dst) return;
  const int32_t deg = rowptr[i + 1] - rowptr[i];
  cnt[i] = deg == 0 ? 1 : (deg + GGNN_UNIT_EDGES - 1) / GGNN_UNIT_EDGES;
}

__global__ __launch_bounds__(256) void csr_unit_fill_kernel(const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ unit_ptr,
                                                            int64_t n_dst,
                                                            int32_t* __restrict__ units) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_dst) return;
  const int32_t beg = rowptr[i], end = rowptr[i + 1];
  int32_t u = unit_ptr[i];
  int32_t p = beg;
  do {
    const int32_t nact = min(GGNN_UNIT_EDGES, end - p);
    const int32_t first = p == beg, last = p + GGNN_UNIT_EDGES >= end;
    int32_t* d = units + 8 * (int64_t)u;
    const int32_t j0 = nact > 0 ? col[p] : 0;
    d[0] = (int32_t)i;
    d[1] = p;
    d[2] = nact | (first << 8) | (last << 9);
    d[3] = 0;
    d[4] = j0;
    d[5] = nact > 1 ? col[p + 1] : j0;
    d[6] = nact > 2 ? col[p + 2] : j0;
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

extern "C" int ggnn_build_csr(const int64_t* edge_index, int64_t E, int64_t n_src, int64_t n_dst,
                              int32_t* rowptr, int32_t* col, int32_t* perm, int32_t* row,
                              int32_t* unit_ptr, int32_t* units, int32_t* flags, void* workspace, size_t workspace_bytes, ggnn_stream_t stream_) {
  using namespace ggnn;
  hipStream_t stream = (hipStream_t)stream_;
  if (E < 0 || n_src < 0 || n_dst <= 0 || !rowptr || !flags || !workspace) return GGNN_EINVAL;
  if (E > 0 && (!edge_index || !col || !perm || !row)) return GGNN_EINVAL;
  if (!unit_ptr || !units || !aligned16(units)) return GGNN_EINVAL;
  if (E >= INT32_MAX || n_dst >= INT32_MAX || n_src >= INT32_MAX) return GGNN_EINVAL;
  if (workspace_bytes < ggnn_csr_workspace_bytes(E, n_dst)) return GGNN_EINVAL;
  int32_t* counts = (int32_t*)workspace;
  int32_t* cursor = counts + n_dst + 1;
  if (hipMemsetAsync(counts, 0, (size_t)n_dst * sizeof(int32_t), stream) != hipSuccess)
    return GGNN_ELAUNCH;
  const unsigned eb = (unsigned)((E + 255) / 256), nb = (unsigned)((n_dst + 255) / 256);
  if (E > 0)
    hipLaunchKernelGGL(csr_count_kernel, dim3(eb), dim3(256), 0, stream, edge_index, E, n_src,
                       n_dst, counts, flags);
  hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, n_dst, rowptr,
                     cursor);
  if (E > 0) {
    hipLaunchKernelGGL(csr_fill_kernel, dim3(eb), dim3(256), 0, stream, edge_index, E, n_src,
                       n_dst, cursor, perm);
    hipLaunchKernelGGL(csr_sort_kernel, dim3(nb), dim3(256), 0, stream, edge_index, rowptr,
                       n_dst, perm, col, row);
  }
  // unit table: per-row unit counts -> exclusive scan (cursor is free again) -> descriptors
  hipLaunchKernelGGL(csr_unit_count_kernel, dim3(nb), dim3(256), 0, stream, rowptr, n_dst, counts);
  hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, n_dst, unit_ptr,
                     cursor);
  hipLaunchKernelGGL(csr_unit_fill_kernel, dim3(nb), dim3(256), 0, stream, rowptr, col, unit_ptr,
                     n_dst, units);
  return launch_status();
}
