// Work plan of the aggregation: a pure function of the row pointers, built once per adjacency
// (and once per window of a transposed adjacency) at batch construction, on the host or on the
// GPU.  Integer work; both builders produce the same arrays, element for element.
//
//   order[n_rows]      the rows longer than hub_threshold first (n_hub of them), then the others;
//                      in both parts by descending length, ties in ascending row id (stable).
//                      Matrices of more than plan_local_rows rows (32,768) are ordered chunk by
//                      chunk instead: inside both parts the rows of chunk 0 (row ids 0..1023)
//                      come first, then chunk 1, ..., each chunk by descending length.  Rows of a
//                      batch are its partitions one after another and most edges stay inside a
//                      partition: rows processed at about the same time then gather from the same
//                      few thousand source rows, which a 4 MiB L2 holds (measured on a reddit-
//                      sized batch of 116 k rows at width 41: 494 -> 291 us per launch; 257 us
//                      with the sliced kernel's per-XCD runs of workgroups).  Below that size
//                      a slice of all in-batch source rows fits the L2 anyway and chunks only
//                      cost balance (products batch, 16 k rows, width 256: 58 -> 65 us).
//   items[2 * n_items] (row, segment) pairs, rows in the order above, segments ascending;
//                      a segment is segment_edges edges (the last one shorter)
//   multi_rows[n_multi] the rows with more than one segment
#include "common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <climits>
#include <mutex>
#include <numeric>
#include <vector>

namespace incagg {

constexpr int kBlock = 256;

static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
static inline unsigned grid_for(long long n) { return (unsigned)std::max<long long>(1, (n + kBlock - 1) / kBlock); }

struct PlanStatus {
  int n_hub;
  int n_items;
  int n_multi;
  int error;  // 1: negative row length, 2: capacity exceeded
};

// Sort key of a row (ascending, stable): short rows after long ones, then the chunk of the row,
// then descending length.
typedef unsigned long long PlanKey;
constexpr int kPlanKeyBits = 63;
__host__ __device__ inline PlanKey plan_key(int row, int d, int hub_threshold, int chunk_rows) {
  return (d > hub_threshold ? 0ull : 1ull << 62) |
         ((PlanKey)(chunk_rows > 0 ? row / chunk_rows : 0) << 31) | (PlanKey)(0x7fffffff - d);
}
__host__ __device__ inline int key_length(PlanKey k) { return 0x7fffffff - (int)(k & 0x7fffffffull); }
__host__ __device__ inline bool key_long(PlanKey k) { return (k >> 62) == 0; }

__global__ __launch_bounds__(kBlock) void plan_keys_kernel(const int *rowptr, int n_rows,
                                                          int hub_threshold, int chunk_rows,
                                                          PlanKey *keys, int *ids, PlanStatus *st) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_rows) return;
  int d = rowptr[i + 1] - rowptr[i];
  if (d < 0) {
    st->error = 1;
    d = 0;
  }
  keys[i] = plan_key(i, d, hub_threshold, chunk_rows);
  ids[i] = i;
}

// Over the sorted positions: segments and "several segments" flag of every position.
__global__ __launch_bounds__(kBlock) void plan_segments_kernel(const PlanKey *sorted, int n_rows,
                                                              int seg, int *nseg, int *multi) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= n_rows) return;
  const int d = key_length(sorted[j]);
  const int k = key_long(sorted[j]) ? (d + seg - 1) / seg : 0;
  nseg[j] = k;
  multi[j] = k > 1 ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void plan_fill_kernel(const int *order, const PlanKey *sorted,
                                                          const int *nseg, const int *item_off,
                                                          const int *multi_off, int n_rows,
                                                          long long items_cap,
                                                          long long multi_cap, int2 *items,
                                                          int *multi_rows, PlanStatus *st) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= n_rows) return;
  const int k = nseg[j];
  if (j == n_rows - 1) {  // totals: exclusive sums + the last element
    st->n_items = item_off[j] + k;
    st->n_multi = multi_off[j] + (k > 1 ? 1 : 0);
  }
  // the long rows are a prefix: it ends at the last long position
  if (key_long(sorted[j]) && (j == n_rows - 1 || !key_long(sorted[j + 1]))) st->n_hub = j + 1;
  if (k == 0) return;
  const int row = order[j];
  const long long base = item_off[j];
  if (base + k > items_cap) {
    st->error = 2;
    return;
  }
  for (int s = 0; s < k; s++) items[base + s] = make_int2(row, s);
  if (k > 1) {
    if (multi_off[j] >= multi_cap) {
      st->error = 2;
      return;
    }
    multi_rows[multi_off[j]] = row;
  }
}

static int resolve(int32_t &hub_threshold, int32_t &segment_edges) {
  int t = 0, s = 0;
  plan_defaults(&t, &s);
  if (hub_threshold <= 0) hub_threshold = t;
  if (segment_edges <= 0) segment_edges = s;
  INCAGG_REQUIRE(segment_edges % 64 == 0, "segment_edges %d is not a multiple of 64", segment_edges);
  return INCAGG_OK;
}

}  // namespace incagg

using namespace incagg;

extern "C" int incagg_spmm_plan_defaults(int32_t *hub_threshold, int32_t *segment_edges) {
  INCAGG_REQUIRE(hub_threshold && segment_edges, "null output");
  int t = 0, s = 0;
  plan_defaults(&t, &s);
  *hub_threshold = t;
  *segment_edges = s;
  return INCAGG_OK;
}

extern "C" int32_t incagg_spmm_plan_chunk_rows(int64_t n_rows) { return plan_chunk_rows(n_rows); }

extern "C" int64_t incagg_spmm_plan_items_bound(int64_t n_rows, int64_t nnz, int32_t hub_threshold,
                                                int32_t segment_edges) {
  if (n_rows < 0 || nnz < 0) return fail(INCAGG_E_INVALID, "negative size");
  int rc = resolve(hub_threshold, segment_edges);
  if (rc) return rc;
  // every long row has more than hub_threshold edges and ceil(len / segment) <= 1 + len / segment
  return std::min<int64_t>(n_rows, nnz / ((int64_t)hub_threshold + 1)) + nnz / segment_edges;
}

extern "C" int incagg_spmm_plan_host(const int32_t *rowptr, int64_t n_rows, int32_t hub_threshold,
                                     int32_t segment_edges, int32_t *order, int32_t *items,
                                     int64_t items_cap, int32_t *multi_rows, int64_t multi_cap,
                                     int64_t *counts) {
  INCAGG_REQUIRE(counts != nullptr, "null counts");
  INCAGG_REQUIRE(n_rows >= 0 && n_rows < INT_MAX - 64, "row count out of range");
  int rc = resolve(hub_threshold, segment_edges);
  if (rc) return rc;
  counts[0] = counts[1] = counts[2] = 0;
  counts[3] = hub_threshold;
  counts[4] = segment_edges;
  if (n_rows == 0) return INCAGG_OK;
  INCAGG_REQUIRE(rowptr && order, "null rowptr or order");
  for (int64_t i = 0; i < n_rows; i++)
    if (rowptr[i + 1] < rowptr[i]) return fail(INCAGG_E_INVALID, "row pointers decrease at row %lld", (long long)i);
  const int chunk_rows = plan_chunk_rows(n_rows);
  std::iota(order, order + n_rows, 0);
  std::stable_sort(order, order + n_rows, [&](int32_t a, int32_t b) {
    return plan_key(a, rowptr[a + 1] - rowptr[a], hub_threshold, chunk_rows) <
           plan_key(b, rowptr[b + 1] - rowptr[b], hub_threshold, chunk_rows);
  });
  int64_t n_hub = 0, n_items = 0, n_multi = 0;
  for (int64_t j = 0; j < n_rows; j++) {
    const int32_t row = order[j];
    const int64_t d = (int64_t)rowptr[row + 1] - rowptr[row];
    if (d <= hub_threshold) break;
    n_hub++;
    const int64_t k = (d + segment_edges - 1) / segment_edges;
    if (n_items + k > items_cap || !items)
      return fail(INCAGG_E_INVALID, "items capacity %lld too small", (long long)items_cap);
    for (int64_t s = 0; s < k; s++) {
      items[2 * (n_items + s)] = row;
      items[2 * (n_items + s) + 1] = (int32_t)s;
    }
    n_items += k;
    if (k > 1) {
      if (n_multi >= multi_cap || !multi_rows)
        return fail(INCAGG_E_INVALID, "multi_rows capacity %lld too small", (long long)multi_cap);
      multi_rows[n_multi++] = row;
    }
  }
  counts[0] = n_hub;
  counts[1] = n_items;
  counts[2] = n_multi;
  return INCAGG_OK;
}

extern "C" int64_t incagg_spmm_plan_device_scratch_bytes(int64_t n_rows) {
  if (n_rows < 0 || n_rows >= INT_MAX - 64) return fail(INCAGG_E_INVALID, "row count out of range");
  const int n = (int)std::max<int64_t>(n_rows, 1);
  size_t sort = 0, scan = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort, (PlanKey *)nullptr, (PlanKey *)nullptr,
                                           (int *)nullptr, (int *)nullptr, n, 0, kPlanKeyBits);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (int *)nullptr, (int *)nullptr, n);
  return (int64_t)(up256(sizeof(PlanStatus)) + 2 * up256(sizeof(PlanKey) * (size_t)n) +
                   5 * up256(sizeof(int) * (size_t)n) + up256(std::max(sort, scan)));
}

// Enqueues the device builder; the counts stay in *status_out (device memory inside scratch).
static int plan_device_enqueue(const int32_t *rowptr, int n, int32_t hub_threshold,
                               int32_t segment_edges, void *scratch, int64_t scratch_bytes,
                               int32_t *order, int32_t *items, int64_t items_cap,
                               int32_t *multi_rows, int64_t multi_cap, PlanStatus **status_out,
                               hipStream_t stream) {
  char *base = static_cast<char *>(scratch);
  const size_t e = up256(sizeof(int) * (size_t)n), k = up256(sizeof(PlanKey) * (size_t)n);
  PlanStatus *st = reinterpret_cast<PlanStatus *>(base);
  base += up256(sizeof(PlanStatus));
  PlanKey *keys = reinterpret_cast<PlanKey *>(base);
  PlanKey *sorted = reinterpret_cast<PlanKey *>(base + k);
  base += 2 * k;
  int *ids = reinterpret_cast<int *>(base);
  int *nseg = reinterpret_cast<int *>(base + e);
  int *multi = reinterpret_cast<int *>(base + 2 * e);
  int *item_off = reinterpret_cast<int *>(base + 3 * e);
  int *multi_off = reinterpret_cast<int *>(base + 4 * e);
  void *cub = base + 5 * e;
  const size_t cub_bytes = (size_t)scratch_bytes - (size_t)(static_cast<char *>(cub) - static_cast<char *>(scratch));

  INCAGG_HIP_TRY(hipMemsetAsync(st, 0, sizeof(PlanStatus), stream));
  const dim3 grid(grid_for(n)), block(kBlock);
  hipLaunchKernelGGL(plan_keys_kernel, grid, block, 0, stream, rowptr, n, hub_threshold,
                     plan_chunk_rows(n), keys, ids, st);
  INCAGG_HIP_TRY(hipGetLastError());
  // radix sort, stable: ties keep ascending row ids, like the host builder
  size_t tmp = cub_bytes;
  INCAGG_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, tmp, keys, sorted, ids, order, n, 0,
                                                    kPlanKeyBits, stream));
  hipLaunchKernelGGL(plan_segments_kernel, grid, block, 0, stream, sorted, n, segment_edges, nseg,
                     multi);
  INCAGG_HIP_TRY(hipGetLastError());
  tmp = cub_bytes;
  INCAGG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub, tmp, nseg, item_off, n, stream));
  tmp = cub_bytes;
  INCAGG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub, tmp, multi, multi_off, n, stream));
  hipLaunchKernelGGL(plan_fill_kernel, grid, block, 0, stream, order, sorted, nseg, item_off,
                     multi_off, n, (long long)items_cap, (long long)multi_cap,
                     reinterpret_cast<int2 *>(items), multi_rows, st);
  INCAGG_HIP_TRY(hipGetLastError());
  *status_out = st;
  return INCAGG_OK;
}

extern "C" int incagg_spmm_plan_device(const int32_t *rowptr, int64_t n_rows, int32_t hub_threshold,
                                       int32_t segment_edges, void *scratch, int64_t scratch_bytes,
                                       int32_t *order, int32_t *items, int64_t items_cap,
                                       int32_t *multi_rows, int64_t multi_cap, int64_t *counts,
                                       void *stream_) {
  INCAGG_REQUIRE(counts != nullptr, "null counts");
  INCAGG_REQUIRE(n_rows >= 0 && n_rows < INT_MAX - 64, "row count out of range");
  int rc = resolve(hub_threshold, segment_edges);
  if (rc) return rc;
  counts[0] = counts[1] = counts[2] = 0;
  counts[3] = hub_threshold;
  counts[4] = segment_edges;
  if (n_rows == 0) return INCAGG_OK;
  INCAGG_REQUIRE(rowptr && order, "null rowptr or order");
  INCAGG_REQUIRE(items_cap >= 0 && multi_cap >= 0 && (items_cap == 0 || items) &&
                     (multi_cap == 0 || multi_rows), "bad capacities");
  const int64_t need = incagg_spmm_plan_device_scratch_bytes(n_rows);
  INCAGG_REQUIRE(scratch && scratch_bytes >= need, "scratch too small: %lld < %lld",
                 (long long)scratch_bytes, (long long)need);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PlanStatus *st = nullptr;
  rc = plan_device_enqueue(rowptr, (int)n_rows, hub_threshold, segment_edges, scratch, scratch_bytes,
                           order, items, items_cap, multi_rows, multi_cap, &st, stream);
  if (rc) return rc;
  PlanStatus host{};
  INCAGG_HIP_TRY(hipMemcpyAsync(&host, st, sizeof(PlanStatus), hipMemcpyDeviceToHost, stream));
  INCAGG_HIP_TRY(hipStreamSynchronize(stream));
  if (host.error == 1) return fail(INCAGG_E_INVALID, "row pointers decrease");
  if (host.error == 2)
    return fail(INCAGG_E_INVALID, "plan does not fit the capacities (items %lld, multi %lld)",
                (long long)items_cap, (long long)multi_cap);
  counts[0] = host.n_hub;
  counts[1] = host.n_items;
  counts[2] = host.n_multi;
  return INCAGG_OK;
}

// ---------------------------------------------------------------- the builder that allocates nothing
// Same kernels, same outputs as incagg_spmm_plan_device, with the work arrays in a fixed array
// of the code object instead of a caller-provided workspace, and the counts left on the device
// (no host copy, no wait).  Every GPU holds its own copy of the array, at its own address,
// until the process ends; calls on one GPU share it, so calls that overlap in time race.
namespace incagg {

constexpr int64_t kStaticPlanRows = 262144;
constexpr size_t kStaticPlanBytes = 16u << 20;
static __device__ __align__(256) unsigned char scratch[kStaticPlanBytes];

static constexpr int kMaxPlanDevices = 64;
static unsigned char *s_scratch_ptr[kMaxPlanDevices] = {};
static std::mutex s_scratch_mutex;

// Device address of `scratch` on the current device (its name is only a host-side handle).
static int static_scratch(unsigned char **out) {
  int dev = 0;
  INCAGG_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxPlanDevices) return fail(INCAGG_E_INVALID, "device id %d unsupported", dev);
  std::lock_guard<std::mutex> lock(s_scratch_mutex);
  if (!s_scratch_ptr[dev]) {
    void *p = nullptr;
    INCAGG_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    s_scratch_ptr[dev] = static_cast<unsigned char *>(p);
  }
  *out = s_scratch_ptr[dev];
  return INCAGG_OK;
}

__global__ void plan_publish_kernel(const PlanStatus *st, int hub_threshold, int segment_edges,
                                    int *counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    counts[0] = st->n_hub;
    counts[1] = st->n_items;
    counts[2] = st->n_multi;
    counts[3] = hub_threshold;
    counts[4] = segment_edges;
    counts[5] = st->error;
  }
}

}  // namespace incagg

extern "C" int64_t incagg_spmm_plan_device_static_max_rows(void) { return kStaticPlanRows; }

extern "C" int incagg_spmm_plan_device_static(const int32_t *rowptr, int64_t n_rows,
                                              int32_t hub_threshold, int32_t segment_edges,
                                              int32_t *order, int32_t *items, int64_t items_cap,
                                              int32_t *multi_rows, int64_t multi_cap,
                                              int32_t *counts_dev, void *stream_) {
  INCAGG_REQUIRE(counts_dev != nullptr, "null counts");
  INCAGG_REQUIRE(n_rows >= 1, "the static builder needs at least one row");
  INCAGG_REQUIRE(n_rows <= kStaticPlanRows,
                 "%lld rows exceed the %lld the fixed work array is sized for: use "
                 "incagg_spmm_plan_device with a workspace", (long long)n_rows,
                 (long long)kStaticPlanRows);
  int rc = resolve(hub_threshold, segment_edges);
  if (rc) return rc;
  INCAGG_REQUIRE(rowptr && order, "null rowptr or order");
  INCAGG_REQUIRE(items_cap >= 0 && multi_cap >= 0 && (items_cap == 0 || items) &&
                     (multi_cap == 0 || multi_rows), "bad capacities");
  const int64_t need = incagg_spmm_plan_device_scratch_bytes(n_rows);
  INCAGG_REQUIRE(need >= 0 && (size_t)need <= kStaticPlanBytes,
                 "work arrays for %lld rows need %lld bytes, the fixed array has %zu",
                 (long long)n_rows, (long long)need, kStaticPlanBytes);
  unsigned char *ws = nullptr;
  rc = static_scratch(&ws);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PlanStatus *st = nullptr;
  // (the status word and every work array are (re)written by the kernels below before they are
  // read: nothing of an earlier call's contents is used)
  rc = plan_device_enqueue(rowptr, (int)n_rows, hub_threshold, segment_edges, ws,
                           (int64_t)kStaticPlanBytes, order, items, items_cap, multi_rows, multi_cap,
                           &st, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(plan_publish_kernel, dim3(1), dim3(64), 0, stream, st, hub_threshold,
                     segment_edges, counts_dev);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}
