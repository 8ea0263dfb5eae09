// Device-side batch construction: one-hop relabelling of a partition batch straight from the
// HBM-resident global CSR, and the transposed CSR of the result for the backward
// aggregation.  Same results as the host functions in batch_build.cpp (and therefore as
// csrc/cpu/relabel_cpu.cpp:3-108,111-214 of the reference), in the device format the
// aggregation kernels consume (int32 rowptr / col, fp32 values, int64 n_id).
//
// The reference defines the local id of an out-of-batch neighbour by the order in which a
// serial scan over the batch's edges first meets it (relabel_cpu.cpp:56-69).  In parallel
// that order is recovered exactly: every edge position p of the output proposes itself
// for its column with an atomic "smallest p wins" into a dense table indexed by global id;
// the winners are the first occurrences, and an exclusive prefix sum over the winner flags
// in edge order numbers them 0, 1, 2, ... precisely as the serial scan would.
//
// All passes stream over the E batch edges (or the B batch rows); the two dense id tables
// (int32[n_nodes], caller-owned, all -1 between calls) play the role of the host
// implementation's epoch-stamped map and are reset entry by entry at the end, so a call costs
// O(E), not O(n_nodes).
#include "common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <climits>

namespace incagg {

constexpr int kBlock = 256;

struct BuildStatus {    // lives in the scratch buffer; copied to the host at the end
  long long n_edges;    // edges of the output (0 when an error stopped the build)
  long long n_new;      // out-of-batch nodes appended to n_id
  long long error;      // 0 ok, 1 idx out of range, 2 column out of range, 3 capacity exceeded
  long long distinct;   // distinct ids in idx
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// deg[i] = length of row idx[i]; map_a[idx[i]] = i (largest i wins, as with the serial
// "later duplicates overwrite" of the reference map); n_id[i] = idx[i].
__global__ void degrees_kernel(const int64_t *rowptr, int64_t n_nodes, const int64_t *idx,
                               int64_t n_idx, int *map_a, long long *deg, int64_t *out_n_id,
                               BuildStatus *st) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i > n_idx) return;
  if (i == n_idx) {
    deg[i] = 0;
    return;
  }
  const int64_t v = idx[i];
  if (v < 0 || v >= n_nodes) {
    st->error = 1;
    deg[i] = 0;
    return;
  }
  deg[i] = rowptr[v + 1] - rowptr[v];
  out_n_id[i] = v;
  atomicMax(map_a + v, (int)i);
}

// Number of distinct ids in idx (needed for the non-bipartite row-pointer tail of the
// within-batch variant): position i counts when it is the one the map kept.
__global__ void distinct_kernel(const int64_t *idx, int64_t n_idx, int64_t n_nodes,
                                const int *map_a, BuildStatus *st) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool mine = false;
  if (i < n_idx) {
    const int64_t v = idx[i];
    mine = v >= 0 && v < n_nodes && map_a[v] == (int)i;
  }
  const unsigned long long b = __ballot(mine);
  if (lane_id() == 0 && b) atomicAdd((unsigned long long *)&st->distinct, (unsigned long long)__popcll(b));
}

__global__ void bound_degrees_kernel(const int64_t *rowptr, int64_t n_nodes, const int64_t *idx,
                                     int64_t n_idx, long long *deg, BuildStatus *st) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n_idx) return;
  const int64_t v = idx[i];
  if (v < 0 || v >= n_nodes) {
    st->error = 1;
    deg[i] = 0;
  } else {
    deg[i] = rowptr[v + 1] - rowptr[v];
  }
}

// flag[p] = 0 for p in [live edges, capacity): the prefix sum runs over the whole capacity
__global__ void clear_tail_kernel(int *flag, long long cap, const BuildStatus *st) {
  for (long long p = st->n_edges + blockIdx.x * (long long)blockDim.x + threadIdx.x; p < cap;
       p += (long long)gridDim.x * blockDim.x)
    flag[p] = 0;
}

// rowptr64 -> int32 output, capacity check, edge count into the status block.
__global__ void finish_rowptr_kernel(const long long *rp64, int64_t n_idx, int64_t edge_capacity,
                                     int *out_rowptr, BuildStatus *st) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const long long total = rp64[n_idx];
  const bool ok = total <= edge_capacity && st->error == 0;
  if (i <= n_idx) out_rowptr[i] = ok ? (int)rp64[i] : 0;
  if (i == 0) {
    if (total > edge_capacity && st->error == 0) st->error = 3;
    st->n_edges = ok ? total : 0;
  }
}

// One wave per batch row: copies the row's global column ids (and values) to their output
// positions and proposes each position as "first occurrence" of its column.
__global__ void expand_kernel(const int64_t *rowptr, const int64_t *col, const float *value,
                              const int64_t *idx, int64_t n_idx, int64_t n_nodes,
                              const int *out_rowptr, const int *map_a, int *map_b, int *gcol,
                              float *out_value, BuildStatus *st) {
  if (st->n_edges == 0) return;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= n_idx) return;
  const int64_t v = idx[row];
  const int64_t rs = rowptr[v], re = rowptr[v + 1];
  const int base = out_rowptr[row];
  for (int64_t e = rs + lane_id(); e < re; e += 64) {
    const int p = base + (int)(e - rs);
    const int64_t c = col[e];
    if (c < 0 || c >= n_nodes) {
      st->error = 2;
      gcol[p] = -1;
      continue;
    }
    gcol[p] = (int)c;
    if (out_value) out_value[p] = value[e];
    if (map_a[c] < 0) atomicMax(map_b + c, INT_MAX - p);  // smallest p wins
  }
}

__global__ void first_flags_kernel(const int *gcol, const int *map_a, const int *map_b, int *flag,
                                   const BuildStatus *st) {
  const long long n = st->n_edges;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n;
       p += (long long)gridDim.x * blockDim.x) {
    const int c = gcol[p];
    flag[p] = c >= 0 && map_a[c] < 0 && INT_MAX - map_b[c] == p;
  }
}

// The winners take their numbers: n_id[n_idx + k] = column, and the dense table now holds the
// local id (it held the winning position until here -- only the winner itself rewrites it).
__global__ void assign_new_kernel(const int *gcol, const int *flag, const int *rank, int64_t n_idx,
                                  int64_t node_capacity, int *map_b, int64_t *out_n_id,
                                  BuildStatus *st) {
  const long long n = st->n_edges;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n;
       p += (long long)gridDim.x * blockDim.x) {
    if (!flag[p]) continue;
    const long long label = n_idx + rank[p];
    if (label < node_capacity) {
      out_n_id[label] = gcol[p];
      map_b[gcol[p]] = (int)label;
    } else {
      st->error = 3;
    }
    atomicMax((unsigned long long *)&st->n_new, (unsigned long long)(rank[p] + 1));
  }
}

__global__ void relabel_kernel(const int *gcol, const int *map_a, const int *map_b, int *out_col,
                               const BuildStatus *st) {
  const long long n = st->n_edges;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n;
       p += (long long)gridDim.x * blockDim.x) {
    const int c = gcol[p];
    if (c < 0) {
      out_col[p] = 0;
      continue;
    }
    const int a = map_a[c];
    out_col[p] = a >= 0 ? a : map_b[c];
  }
}

// Leaves both tables all -1 again.
__global__ void reset_kernel(const int *gcol, const int64_t *idx, int64_t n_idx, int64_t n_nodes,
                             int *map_a, int *map_b, const BuildStatus *st, long long gcol_len) {
  const long long n = gcol ? gcol_len : 0;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long live = st->n_edges;
  for (long long p = tid; p < n && p < live; p += stride)
    if (gcol[p] >= 0) map_b[gcol[p]] = -1;
  for (long long i = tid; i < n_idx; i += stride) {
    const int64_t v = idx[i];
    if (v >= 0 && v < n_nodes) map_a[v] = -1;
  }
}

// ---- within-batch variant: edges whose column is a batch node, order kept
__global__ void count_kept_kernel(const int64_t *rowptr, const int64_t *col, const int64_t *idx,
                                  int64_t n_idx, int64_t n_nodes, const int *map_a,
                                  long long *deg, BuildStatus *st) {
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row > n_idx) return;
  if (row == n_idx) {
    if (lane_id() == 0) deg[row] = 0;
    return;
  }
  const int64_t v = idx[row];
  long long kept = 0;
  if (v >= 0 && v < n_nodes) {
    const int64_t rs = rowptr[v], re = rowptr[v + 1];
    for (int64_t e0 = rs; e0 < re; e0 += 64) {
      const int64_t e = e0 + lane_id();
      bool keep = false;
      if (e < re) {
        const int64_t c = col[e];
        if (c < 0 || c >= n_nodes)
          st->error = 2;
        else
          keep = map_a[c] >= 0;
      }
      kept += __popcll(__ballot(keep));
    }
  }
  if (lane_id() == 0) deg[row] = kept;
}

__global__ void fill_kept_kernel(const int64_t *rowptr, const int64_t *col, const float *value,
                                 const int64_t *idx, int64_t n_idx, int64_t n_nodes,
                                 const int *map_a, const int *out_rowptr, int *out_col,
                                 float *out_value, const BuildStatus *st) {
  if (st->n_edges == 0) return;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= n_idx) return;
  const int64_t v = idx[row];
  const int64_t rs = rowptr[v], re = rowptr[v + 1];
  int pos = out_rowptr[row];
  const unsigned long long below = (1ull << lane_id()) - 1ull;
  for (int64_t e0 = rs; e0 < re; e0 += 64) {
    const int64_t e = e0 + lane_id();
    int local = -1;
    if (e < re) {
      const int64_t c = col[e];
      if (c >= 0 && c < n_nodes) local = map_a[c];
    }
    const unsigned long long b = __ballot(local >= 0);
    if (local >= 0) {
      const int p = pos + __popcll(b & below);
      out_col[p] = local;
      if (out_value) out_value[p] = value[e];
    }
    pos += __popcll(b);
  }
}

// ---- transposed CSR
__global__ void coo_rows_kernel(const int *rowptr, int n_rows, int *row_of) {
  const long long row = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  if (row >= n_rows) return;
  for (int e = rowptr[row] + lane_id(); e < rowptr[row + 1]; e += 64) row_of[e] = (int)row;
}

__global__ void col_histogram_kernel(const int *col, long long nnz, int n_cols, int *count, int *err) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < nnz;
       e += (long long)gridDim.x * blockDim.x) {
    const int c = col[e];
    if ((unsigned)c >= (unsigned)n_cols)
      *err = 1;
    else
      atomicAdd(count + c, 1);
  }
}

__global__ void transposed_fill_kernel(const int *sorted_edge, const int *row_of, const int *rowptr,
                                       const float *val, long long nnz, int mean, int *t_col,
                                       float *t_val) {
  for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < nnz;
       k += (long long)gridDim.x * blockDim.x) {
    const int e = sorted_edge[k];
    const int r = row_of[e];
    t_col[k] = r;
    float v = val ? val[e] : 1.0f;
    if (mean) v *= 1.0f / (float)max(rowptr[r + 1] - rowptr[r], 1);
    t_val[k] = v;
  }
}

__global__ void iota_kernel(int *a, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    a[i] = (int)i;
}

static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
static inline int grid_for(long long items) {
  return (int)std::max<long long>(1, std::min<long long>((items + kBlock - 1) / kBlock, 1 << 16));
}

struct RelabelScratch {
  BuildStatus *status;
  long long *deg, *rp64;
  int *gcol, *flag, *rank;
  void *cub;
  size_t cub_bytes;
};

static size_t relabel_cub_bytes(int64_t n_idx, int64_t edge_capacity) {
  size_t a = 0, b = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, a, (long long *)nullptr, (long long *)nullptr,
                                         (int)(n_idx + 1));
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (int *)nullptr, (int *)nullptr,
                                         (int)std::max<int64_t>(edge_capacity, 1));
  return std::max(a, b);
}

static size_t carve_relabel(char *base, int64_t n_idx, int64_t edge_capacity, RelabelScratch *s) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char *p = base ? base + off : nullptr;
    off += up256(bytes);
    return p;
  };
  const size_t cub = relabel_cub_bytes(n_idx, edge_capacity);
  char *st = take(sizeof(BuildStatus));
  char *deg = take(sizeof(long long) * (size_t)(n_idx + 1));
  char *rp = take(sizeof(long long) * (size_t)(n_idx + 1));
  char *gc = take(sizeof(int) * (size_t)std::max<int64_t>(edge_capacity, 1));
  char *fl = take(sizeof(int) * (size_t)std::max<int64_t>(edge_capacity, 1));
  char *rk = take(sizeof(int) * (size_t)std::max<int64_t>(edge_capacity, 1));
  char *cb = take(cub);
  if (s) {
    s->status = reinterpret_cast<BuildStatus *>(st);
    s->deg = reinterpret_cast<long long *>(deg);
    s->rp64 = reinterpret_cast<long long *>(rp);
    s->gcol = reinterpret_cast<int *>(gc);
    s->flag = reinterpret_cast<int *>(fl);
    s->rank = reinterpret_cast<int *>(rk);
    s->cub = cb;
    s->cub_bytes = cub;
  }
  return off;
}

}  // namespace incagg

using namespace incagg;

extern "C" int64_t incagg_relabel_device_scratch_bytes(int64_t n_idx, int64_t edge_capacity) {
  if (n_idx < 0 || edge_capacity < 0 || n_idx >= INT_MAX || edge_capacity >= INT_MAX)
    return fail(INCAGG_E_INVALID, "sizes out of range");
  return (int64_t)carve_relabel(nullptr, n_idx, edge_capacity, nullptr);
}

extern "C" int incagg_relabel_bound_device(const int64_t *rowptr, int64_t n_nodes,
                                           const int64_t *idx, int64_t n_idx, void *scratch,
                                           int64_t scratch_bytes, int64_t *bound, void *stream_) {
  INCAGG_REQUIRE(rowptr && bound && n_idx >= 0 && n_nodes >= 0, "bad arguments");
  INCAGG_REQUIRE(n_idx < INT_MAX - 1, "batch too large");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  *bound = 0;
  if (n_idx == 0) return INCAGG_OK;
  INCAGG_REQUIRE(idx && scratch, "null idx or scratch");
  // scratch: BuildStatus, then one int64 per batch row and the reduction's temporary storage
  size_t cub = 0;
  (void)hipcub::DeviceReduce::Sum(nullptr, cub, (long long *)nullptr, (long long *)nullptr, (int)n_idx);
  const size_t need = up256(sizeof(BuildStatus)) + up256(sizeof(long long) * (size_t)(n_idx + 1)) +
                      up256(sizeof(long long)) + up256(cub);
  INCAGG_REQUIRE((size_t)scratch_bytes >= need, "scratch too small: %lld < %zu", (long long)scratch_bytes, need);
  char *base = static_cast<char *>(scratch);
  BuildStatus *st = reinterpret_cast<BuildStatus *>(base);
  long long *deg = reinterpret_cast<long long *>(base + up256(sizeof(BuildStatus)));
  long long *sum = reinterpret_cast<long long *>(reinterpret_cast<char *>(deg) +
                                                 up256(sizeof(long long) * (size_t)(n_idx + 1)));
  void *tmp = reinterpret_cast<char *>(sum) + up256(sizeof(long long));
  INCAGG_HIP_TRY(hipMemsetAsync(st, 0, sizeof(BuildStatus), stream));
  hipLaunchKernelGGL(bound_degrees_kernel, dim3(grid_for(n_idx)), dim3(kBlock), 0, stream, rowptr, n_nodes, idx,
                     n_idx, deg, st);
  INCAGG_HIP_TRY(hipGetLastError());
  INCAGG_HIP_TRY(hipcub::DeviceReduce::Sum(tmp, cub, deg, sum, (int)n_idx, stream));
  long long host[2] = {0, 0};
  INCAGG_HIP_TRY(hipMemcpyAsync(&host[0], sum, sizeof(long long), hipMemcpyDeviceToHost, stream));
  INCAGG_HIP_TRY(hipMemcpyAsync(&host[1], &st->error, sizeof(long long), hipMemcpyDeviceToHost, stream));
  INCAGG_HIP_TRY(hipStreamSynchronize(stream));
  if (host[1]) return fail(INCAGG_E_INDEX, "batch node id outside [0, %lld)", (long long)n_nodes);
  *bound = host[0];
  return INCAGG_OK;
}

extern "C" int incagg_relabel_one_hop_device(
    const int64_t *rowptr, int64_t n_nodes, const int64_t *col, const float *value,
    const int64_t *idx, int64_t n_idx, int within_batch, int64_t edge_capacity,
    int64_t node_capacity, int32_t *map_a, int32_t *map_b, void *scratch, int64_t scratch_bytes,
    int32_t *out_rowptr, int32_t *out_col, float *out_value, int64_t *out_n_id, int64_t *counts,
    void *stream_) {
  INCAGG_REQUIRE(rowptr && counts && out_rowptr, "null rowptr / counts / out_rowptr");
  INCAGG_REQUIRE(n_nodes >= 0 && n_nodes < INT_MAX && n_idx >= 0 && n_idx < INT_MAX - 1,
                 "sizes out of range (nodes %lld, batch %lld)", (long long)n_nodes, (long long)n_idx);
  INCAGG_REQUIRE(edge_capacity >= 0 && edge_capacity < INT_MAX && node_capacity >= n_idx,
                 "bad capacities (edges %lld, nodes %lld)", (long long)edge_capacity,
                 (long long)node_capacity);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  counts[0] = 0;
  counts[1] = n_idx;
  counts[2] = 0;
  if (n_idx == 0) {
    INCAGG_HIP_TRY(hipMemsetAsync(out_rowptr, 0, sizeof(int32_t), stream));
    return INCAGG_OK;
  }
  INCAGG_REQUIRE(idx && out_n_id && map_a && scratch, "null idx / out_n_id / map / scratch");
  INCAGG_REQUIRE(within_batch || map_b, "null second id table");
  INCAGG_REQUIRE(edge_capacity == 0 || (col && out_col), "null col with edges possible");
  INCAGG_REQUIRE(!value || edge_capacity == 0 || out_value, "null out_value");
  RelabelScratch s{};
  const size_t need = carve_relabel(static_cast<char *>(scratch), n_idx, edge_capacity, &s);
  INCAGG_REQUIRE((size_t)scratch_bytes >= need, "scratch too small: %lld < %zu", (long long)scratch_bytes, need);

  INCAGG_HIP_TRY(hipMemsetAsync(s.status, 0, sizeof(BuildStatus), stream));
  const dim3 block(kBlock);
  const dim3 rows_grid(grid_for(n_idx + 1));
  const long long wave_threads = (n_idx + 1) * 64;
  const dim3 wave_grid((unsigned)((wave_threads + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(degrees_kernel, rows_grid, block, 0, stream, rowptr, n_nodes, idx, n_idx, map_a,
                     s.deg, out_n_id, s.status);
  INCAGG_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(distinct_kernel, rows_grid, block, 0, stream, idx, n_idx, n_nodes, map_a, s.status);
  if (within_batch) {
    hipLaunchKernelGGL(count_kept_kernel, wave_grid, block, 0, stream, rowptr, col, idx, n_idx, n_nodes,
                       map_a, s.deg, s.status);
    INCAGG_HIP_TRY(hipGetLastError());
  }
  size_t cub = s.cub_bytes;
  INCAGG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(s.cub, cub, s.deg, s.rp64, (int)(n_idx + 1), stream));
  hipLaunchKernelGGL(finish_rowptr_kernel, rows_grid, block, 0, stream, s.rp64, n_idx, edge_capacity,
                     out_rowptr, s.status);
  INCAGG_HIP_TRY(hipGetLastError());
  const dim3 edge_grid(grid_for(std::max<int64_t>(edge_capacity, 1)));
  if (within_batch) {
    hipLaunchKernelGGL(fill_kept_kernel, wave_grid, block, 0, stream, rowptr, col, value, idx, n_idx,
                       n_nodes, map_a, out_rowptr, out_col, out_value, s.status);
    hipLaunchKernelGGL(reset_kernel, rows_grid, block, 0, stream, (const int *)nullptr, idx, n_idx,
                       n_nodes, map_a, map_b, s.status, 0ll);
    INCAGG_HIP_TRY(hipGetLastError());
  } else if (edge_capacity > 0) {
    hipLaunchKernelGGL(expand_kernel, wave_grid, block, 0, stream, rowptr, col, value, idx, n_idx,
                       n_nodes, out_rowptr, map_a, map_b, s.gcol, out_value, s.status);
    hipLaunchKernelGGL(first_flags_kernel, edge_grid, block, 0, stream, s.gcol, map_a, map_b, s.flag,
                       s.status);
    INCAGG_HIP_TRY(hipGetLastError());
    // the scan runs over the whole capacity: entries past the live edges are cleared first
    cub = s.cub_bytes;
    hipLaunchKernelGGL(clear_tail_kernel, edge_grid, block, 0, stream, s.flag, (long long)edge_capacity, s.status);
    INCAGG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(s.cub, cub, s.flag, s.rank, (int)edge_capacity, stream));
    hipLaunchKernelGGL(assign_new_kernel, edge_grid, block, 0, stream, s.gcol, s.flag, s.rank, n_idx,
                       node_capacity, map_b, out_n_id, s.status);
    hipLaunchKernelGGL(relabel_kernel, edge_grid, block, 0, stream, s.gcol, map_a, map_b, out_col, s.status);
    hipLaunchKernelGGL(reset_kernel, edge_grid, block, 0, stream, s.gcol, idx, n_idx, n_nodes, map_a,
                       map_b, s.status, (long long)edge_capacity);
    INCAGG_HIP_TRY(hipGetLastError());
  } else {
    hipLaunchKernelGGL(reset_kernel, rows_grid, block, 0, stream, (const int *)nullptr, idx, n_idx,
                       n_nodes, map_a, map_b, s.status, 0ll);
    INCAGG_HIP_TRY(hipGetLastError());
  }
  BuildStatus host{};
  INCAGG_HIP_TRY(hipMemcpyAsync(&host, s.status, sizeof(BuildStatus), hipMemcpyDeviceToHost, stream));
  INCAGG_HIP_TRY(hipStreamSynchronize(stream));
  if (host.error == 1) return fail(INCAGG_E_INDEX, "batch node id outside [0, %lld)", (long long)n_nodes);
  if (host.error == 2) return fail(INCAGG_E_INDEX, "column id outside [0, %lld)", (long long)n_nodes);
  if (host.error == 3)
    return fail(INCAGG_E_INVALID, "batch does not fit the capacities (edges %lld, nodes %lld)",
                (long long)edge_capacity, (long long)node_capacity);
  counts[0] = host.n_edges;
  counts[1] = n_idx + host.n_new;
  counts[2] = host.distinct;
  return INCAGG_OK;
}

extern "C" int64_t incagg_csr_transpose_device_scratch_bytes(int64_t n_cols, int64_t nnz) {
  if (n_cols < 0 || nnz < 0 || n_cols >= INT_MAX || nnz >= INT_MAX)
    return fail(INCAGG_E_INVALID, "sizes out of range");
  size_t scan = 0, sort = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (int *)nullptr, (int *)nullptr, (int)(n_cols + 1));
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort, (int *)nullptr, (int *)nullptr,
                                           (int *)nullptr, (int *)nullptr, (int)std::max<int64_t>(nnz, 1));
  const size_t e = up256(sizeof(int) * (size_t)std::max<int64_t>(nnz, 1));
  return (int64_t)(up256(sizeof(int) * (size_t)(n_cols + 1)) + 4 * e + up256(std::max(scan, sort)));
}

extern "C" int incagg_csr_transpose_device(const int32_t *rowptr, const int32_t *col,
                                           const float *val, int64_t n_rows, int64_t n_cols,
                                           int64_t nnz, int reduce, void *scratch,
                                           int64_t scratch_bytes, int32_t *t_rowptr, int32_t *t_col,
                                           float *t_val, void *stream_) {
  INCAGG_REQUIRE(rowptr && t_rowptr && n_rows >= 0 && n_cols >= 0 && nnz >= 0, "bad arguments");
  INCAGG_REQUIRE(reduce == INCAGG_REDUCE_SUM || reduce == INCAGG_REDUCE_MEAN, "bad reduce %d", reduce);
  INCAGG_REQUIRE(nnz < INT_MAX && n_rows < INT_MAX && n_cols < INT_MAX - 1, "CSR too large for int32");
  INCAGG_REQUIRE(nnz == 0 || (col && t_col && t_val), "null pointer with edges present");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t need = incagg_csr_transpose_device_scratch_bytes(n_cols, nnz);
  INCAGG_REQUIRE(scratch && scratch_bytes >= need, "scratch too small: %lld < %lld",
                 (long long)scratch_bytes, (long long)need);
  int *err = nullptr;
  int rc = error_flag_address(&err);
  if (rc) return rc;
  char *base = static_cast<char *>(scratch);
  const size_t e = up256(sizeof(int) * (size_t)std::max<int64_t>(nnz, 1));
  int *count = reinterpret_cast<int *>(base);
  base += up256(sizeof(int) * (size_t)(n_cols + 1));
  int *row_of = reinterpret_cast<int *>(base);
  int *edge_in = reinterpret_cast<int *>(base + e);
  int *edge_out = reinterpret_cast<int *>(base + 2 * e);
  int *col_sorted = reinterpret_cast<int *>(base + 3 * e);
  void *cub = base + 4 * e;
  size_t cub_bytes = (size_t)scratch_bytes - (size_t)(reinterpret_cast<char *>(cub) - static_cast<char *>(scratch));

  INCAGG_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * (size_t)(n_cols + 1), stream));
  const dim3 block(kBlock);
  if (nnz > 0) {
    hipLaunchKernelGGL(col_histogram_kernel, dim3(grid_for(nnz)), block, 0, stream, col, (long long)nnz,
                       (int)n_cols, count, err);
    INCAGG_HIP_TRY(hipGetLastError());
  }
  size_t tmp = cub_bytes;
  INCAGG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub, tmp, count, t_rowptr, (int)(n_cols + 1), stream));
  if (nnz == 0) return INCAGG_OK;
  const long long wave_threads = (long long)std::max<int64_t>(n_rows, 1) * 64;
  hipLaunchKernelGGL(coo_rows_kernel, dim3((unsigned)((wave_threads + kBlock - 1) / kBlock)), block, 0,
                     stream, rowptr, (int)n_rows, row_of);
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(nnz)), block, 0, stream, edge_in, (long long)nnz);
  INCAGG_HIP_TRY(hipGetLastError());
  // stable sort of the edge numbers by column: transposed row c lists its edges in the
  // order of the original rows, exactly like the host counting sort
  int bits = 1;
  while (bits < 31 && (1ll << bits) < n_cols) bits++;
  tmp = cub_bytes;
  INCAGG_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, tmp, col, col_sorted, edge_in, edge_out,
                                                    (int)nnz, 0, bits, stream));
  hipLaunchKernelGGL(transposed_fill_kernel, dim3(grid_for(nnz)), block, 0, stream, edge_out, row_of,
                     rowptr, val, (long long)nnz, reduce == INCAGG_REDUCE_MEAN ? 1 : 0, t_col, t_val);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}
