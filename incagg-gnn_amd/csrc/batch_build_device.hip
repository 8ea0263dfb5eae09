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
// Passes over edges are edge-centric: a thread takes an output position and finds its row with
// a binary search in the row pointers (a wave per row would leave a 17 000-edge hub row to ONE
// wave: 270 dependent rounds, 0.4 ms on an ogbn-products-sized batch against 20 us balanced).
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
  long long n_full;     // edges of the batch rows before any filtering (within-batch variant)
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

// Row of output position p: the largest r with rp[r] <= p (rp non-decreasing, rp[n_rows] > p).
__device__ __forceinline__ int row_of_position(const int *rp, int n_rows, long long p) {
  int lo = 0, hi = n_rows;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rp[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// Every output position copies its global column id (and value) from the graph and proposes
// itself as "first occurrence" of its column.
__global__ void expand_kernel(const int64_t *rowptr, const int64_t *col, const float *value,
                              const int64_t *idx, int64_t n_idx, int64_t n_nodes,
                              const int *out_rowptr, const int *map_a, int *map_b, int *gcol,
                              float *out_value, BuildStatus *st) {
  const long long n = st->n_edges;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n;
       p += (long long)gridDim.x * blockDim.x) {
    const int row = row_of_position(out_rowptr, (int)n_idx, p);
    const int64_t e = rowptr[idx[row]] + (p - out_rowptr[row]);
    const int64_t c = col[e];
    if (c < 0 || c >= n_nodes) {
      st->error = 2;
      gcol[p] = -1;
      continue;
    }
    gcol[p] = (int)c;
    if (out_value) out_value[p] = value[e];
    if (map_a[c] < 0) atomicMax(map_b + c, INT_MAX - (int)p);  // smallest p wins
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
  }
  // the number of winners: the exclusive prefix sum's last entry plus the last flag
  if (blockIdx.x == 0 && threadIdx.x == 0 && n > 0) st->n_new = rank[n - 1] + flag[n - 1];
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
// int32 row pointers over ALL edges of the batch rows + capacity check (the work arrays hold
// one entry per such edge).
__global__ void full_rowptr_kernel(const long long *rp64, int64_t n_idx, int64_t edge_capacity,
                                   int *rp_full, BuildStatus *st) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const long long total = rp64[n_idx];
  const bool ok = total <= edge_capacity && st->error == 0;
  if (i <= n_idx) rp_full[i] = ok ? (int)rp64[i] : 0;
  if (i == 0) {
    if (total > edge_capacity && st->error == 0) st->error = 3;
    st->n_full = ok ? total : 0;
  }
}

// keep[q] = the column of edge q is a batch node; local[q] = its local id (or -1)
__global__ void keep_flags_kernel(const int64_t *rowptr, const int64_t *col, const int64_t *idx,
                                  int64_t n_idx, int64_t n_nodes, const int *rp_full,
                                  const int *map_a, int *keep, int *local, BuildStatus *st) {
  const long long n = st->n_full;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < n;
       q += (long long)gridDim.x * blockDim.x) {
    const int row = row_of_position(rp_full, (int)n_idx, q);
    const int64_t c = col[rowptr[idx[row]] + (q - rp_full[row])];
    int l = -1;
    if (c < 0 || c >= n_nodes)
      st->error = 2;
    else
      l = map_a[c];
    local[q] = l;
    keep[q] = l >= 0;
  }
}

// flag[q] = 0 for q in [edges of the batch rows, capacity)
__global__ void clear_full_tail_kernel(int *flag, long long cap, const BuildStatus *st) {
  for (long long q = st->n_full + blockIdx.x * (long long)blockDim.x + threadIdx.x; q < cap;
       q += (long long)gridDim.x * blockDim.x)
    flag[q] = 0;
}

// Row pointers of the kept edges: what the exclusive prefix sum of the keep flags holds at the
// rows' first edges; the total goes into the status block.
__global__ void kept_rowptr_kernel(const int *rp_full, const int *keep, const int *rank,
                                   int64_t n_idx, int *out_rowptr, BuildStatus *st) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const long long n = st->n_full;
  const int total = n > 0 && st->error == 0 ? rank[n - 1] + keep[n - 1] : 0;
  if (i <= n_idx) {
    const int q = rp_full[i];
    out_rowptr[i] = st->error != 0 ? 0 : (q < n ? rank[q] : total);
  }
  if (i == 0) st->n_edges = total;
}

__global__ void fill_kept_kernel(const int64_t *rowptr, const float *value, const int64_t *idx,
                                 int64_t n_idx, const int *rp_full, const int *keep, const int *rank,
                                 const int *local, int *out_col, float *out_value,
                                 const BuildStatus *st) {
  if (st->error != 0) return;
  const long long n = st->n_full;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < n;
       q += (long long)gridDim.x * blockDim.x) {
    if (!keep[q]) continue;
    const int pos = rank[q];
    out_col[pos] = local[q];
    if (out_value) {
      const int row = row_of_position(rp_full, (int)n_idx, q);
      out_value[pos] = value[rowptr[idx[row]] + (q - rp_full[row])];
    }
  }
}

// ---- transposed CSR
// edge numbers 0 .. nnz-1 (the values of the sort), the row of every edge and the range check of
// the columns
__global__ void edge_numbers_kernel(const int *rowptr, int n_rows, const int *col, long long nnz,
                                    int n_cols, int *edge, int *row_of, int *err) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < nnz;
       e += (long long)gridDim.x * blockDim.x) {
    if ((unsigned)col[e] >= (unsigned)n_cols) *err = 1;
    edge[e] = (int)e;
    row_of[e] = row_of_position(rowptr, n_rows, e);
  }
}

// Row pointers of the transposed matrix from the sorted columns: t_rowptr[c] = number of edges
// with a column below c = the first position in the sorted list that holds a column >= c.
// (A binary search per column over an array the sort has just written: no atomics -- a
// histogram queues a hub column's thousands of increments on one address.)
__global__ void rowptr_from_sorted_kernel(const int *col_sorted, int nnz, int n_cols, int *t_rowptr) {
  for (long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x; c <= n_cols;
       c += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = nnz;
    while (lo < hi) {
      const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
      if (col_sorted[mid] < (int)c) lo = mid + 1; else hi = mid;
    }
    t_rowptr[c] = lo;
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

static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
static inline int grid_for(long long items) {
  return (int)std::max<long long>(1, std::min<long long>((items + kBlock - 1) / kBlock, 1 << 16));
}

struct RelabelScratch {
  BuildStatus *status;
  long long *deg, *rp64;
  int *gcol, *flag, *rank, *rp_full;
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
  char *rf = take(sizeof(int) * (size_t)(n_idx + 1));
  char *cb = take(cub);
  if (s) {
    s->status = reinterpret_cast<BuildStatus *>(st);
    s->deg = reinterpret_cast<long long *>(deg);
    s->rp64 = reinterpret_cast<long long *>(rp);
    s->gcol = reinterpret_cast<int *>(gc);
    s->flag = reinterpret_cast<int *>(fl);
    s->rank = reinterpret_cast<int *>(rk);
    s->rp_full = reinterpret_cast<int *>(rf);
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
  hipLaunchKernelGGL(degrees_kernel, rows_grid, block, 0, stream, rowptr, n_nodes, idx, n_idx, map_a,
                     s.deg, out_n_id, s.status);
  INCAGG_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(distinct_kernel, rows_grid, block, 0, stream, idx, n_idx, n_nodes, map_a, s.status);
  size_t cub = s.cub_bytes;
  INCAGG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(s.cub, cub, s.deg, s.rp64, (int)(n_idx + 1), stream));
  const dim3 edge_grid(grid_for(std::max<int64_t>(edge_capacity, 1)));
  if (within_batch) {
    hipLaunchKernelGGL(full_rowptr_kernel, rows_grid, block, 0, stream, s.rp64, n_idx, edge_capacity,
                       s.rp_full, s.status);
    INCAGG_HIP_TRY(hipGetLastError());
    if (edge_capacity > 0) {
      // keep flags and local ids per edge of the batch rows, positions of the kept ones by a
      // prefix sum over the whole capacity (entries past the live edges cleared first)
      hipLaunchKernelGGL(keep_flags_kernel, edge_grid, block, 0, stream, rowptr, col, idx, n_idx,
                         n_nodes, s.rp_full, map_a, s.flag, s.gcol, s.status);
      hipLaunchKernelGGL(clear_full_tail_kernel, edge_grid, block, 0, stream, s.flag,
                         (long long)edge_capacity, s.status);
      INCAGG_HIP_TRY(hipGetLastError());
      cub = s.cub_bytes;
      INCAGG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(s.cub, cub, s.flag, s.rank, (int)edge_capacity, stream));
    }
    hipLaunchKernelGGL(kept_rowptr_kernel, rows_grid, block, 0, stream, s.rp_full, s.flag, s.rank, n_idx,
                       out_rowptr, s.status);
    if (edge_capacity > 0)
      hipLaunchKernelGGL(fill_kept_kernel, edge_grid, block, 0, stream, rowptr, value, idx, n_idx,
                         s.rp_full, s.flag, s.rank, s.gcol, out_col, out_value, s.status);
    hipLaunchKernelGGL(reset_kernel, rows_grid, block, 0, stream, (const int *)nullptr, idx, n_idx,
                       n_nodes, map_a, map_b, s.status, 0ll);
    INCAGG_HIP_TRY(hipGetLastError());
  } else {
    hipLaunchKernelGGL(finish_rowptr_kernel, rows_grid, block, 0, stream, s.rp64, n_idx, edge_capacity,
                       out_rowptr, s.status);
    INCAGG_HIP_TRY(hipGetLastError());
    if (edge_capacity > 0) {
      hipLaunchKernelGGL(expand_kernel, edge_grid, block, 0, stream, rowptr, col, value, idx, n_idx,
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

  const dim3 block(kBlock);
  (void)count;    // (the layout keeps the slot of the former histogram)
  if (nnz == 0) {
    INCAGG_HIP_TRY(hipMemsetAsync(t_rowptr, 0, sizeof(int) * (size_t)(n_cols + 1), stream));
    return INCAGG_OK;
  }
  hipLaunchKernelGGL(edge_numbers_kernel, dim3(grid_for(nnz)), block, 0, stream, rowptr, (int)n_rows,
                     col, (long long)nnz, (int)n_cols, edge_in, row_of, err);
  INCAGG_HIP_TRY(hipGetLastError());
  // stable sort of the edge numbers by column: transposed row c lists its edges in the
  // order of the original rows, exactly like the host counting sort
  int bits = 1;
  while (bits < 31 && (1ll << bits) < n_cols) bits++;
  size_t tmp = cub_bytes;
  INCAGG_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, tmp, col, col_sorted, edge_in, edge_out,
                                                    (int)nnz, 0, bits, stream));
  hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3(grid_for(n_cols + 1)), block, 0, stream,
                     col_sorted, (int)nnz, (int)n_cols, t_rowptr);
  hipLaunchKernelGGL(transposed_fill_kernel, dim3(grid_for(nnz)), block, 0, stream, edge_out, row_of,
                     rowptr, val, (long long)nnz, reduce == INCAGG_REDUCE_MEAN ? 1 : 0, t_col, t_val);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}
