// Last step of a tall "transposed times normal" product a^T b ([rows, m]^T [rows, n], rows >> m, n:
// every weight gradient of the models here).  The caller runs the bulk as ONE batched GEMM over S
// equal row slabs (hipBLASLt) and hands over the S partial [m, n] products plus the R < S
// remainder rows that did not fit a slab; this kernel adds everything up:
//   out[i, j] = sum_s parts[s, i, j] + sum_r a_tail[r, i] * b_tail[r, j]
// in place of a slab-sum kernel followed by a GEMM with a reduction of length R (which costs a
// full hipBLASLt launch, ~14 us, for a few thousand multiply-adds).  fp32, fixed summation order.
#include "common.h"

#include <algorithm>
#include <climits>
#include <cmath>

namespace incagg {

__global__ __launch_bounds__(256) void tn_finish_kernel(const float *parts, int n_parts,
                                                        long long part_stride, const float *a_tail,
                                                        long long lda, const float *b_tail,
                                                        long long ldb, int n_tail, int m, int n,
                                                        float *out, long long ldo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)m * n) return;
  const int i = (int)(idx / n), j = (int)(idx - (long long)i * n);
  float acc = 0.f;
  for (int s = 0; s < n_parts; s++) acc += parts[s * part_stride + (long long)i * n + j];
  for (int r = 0; r < n_tail; r++) acc += a_tail[r * lda + i] * b_tail[r * ldb + j];
  out[i * ldo + j] = acc;
}

// The same for several products of one shape at once (the weight gradients of all layers of a
// model, finished by ONE launch at the end of the backward pass instead of one launch per
// layer): workgroup b works on product b / blocks_per_product; every product's slab partials
// and remainder rows come from a pointer table.
constexpr int kFinishBatch = 16;
struct FinishTable {
  const float *parts[kFinishBatch];
  const float *a[kFinishBatch];
  const float *b[kFinishBatch];
  long long lda[kFinishBatch];
  long long ldb[kFinishBatch];
};

__global__ __launch_bounds__(256) void tn_finish_batched_kernel(const FinishTable t, int n_parts,
                                                                int n_tail, int m, int n,
                                                                int blocks_per_product, float *out) {
  const int l = blockIdx.x / blocks_per_product;
  const long long idx = (long long)(blockIdx.x - l * blocks_per_product) * blockDim.x + threadIdx.x;
  const long long mn = (long long)m * n;
  if (idx >= mn) return;
  const int i = (int)(idx / n), j = (int)(idx - (long long)i * n);
  const float *p = t.parts[l] + idx;
  float acc = 0.f;
  for (int s = 0; s < n_parts; s++) acc += p[s * mn];
  const float *a = t.a[l], *b = t.b[l];
  const long long lda = t.lda[l], ldb = t.ldb[l];
  for (int r = 0; r < n_tail; r++) acc += a[r * lda + i] * b[r * ldb + j];
  out[l * mn + idx] = acc;
}

}  // namespace incagg

using namespace incagg;

extern "C" int incagg_tn_finish_batched_f32(const float *const *parts, int64_t n_batch,
                                            int64_t n_parts, const float *const *a_tails,
                                            const int64_t *lda, const float *const *b_tails,
                                            const int64_t *ldb, int64_t n_tail, int64_t m, int64_t n,
                                            float *out, void *stream) {
  INCAGG_REQUIRE(m >= 0 && n >= 0 && n_parts >= 0 && n_tail >= 0 && n_batch >= 0, "negative size");
  INCAGG_REQUIRE(m < INT_MAX && n < INT_MAX && m * n < (1ll << 40) && n_parts < INT_MAX &&
                     n_tail < INT_MAX, "size out of range");
  INCAGG_REQUIRE(n_batch <= kFinishBatch, "at most %d products per call", kFinishBatch);
  if (m == 0 || n == 0 || n_batch == 0) return INCAGG_OK;
  INCAGG_REQUIRE(out && parts, "null output or parts");
  FinishTable t{};
  for (int64_t l = 0; l < n_batch; l++) {
    INCAGG_REQUIRE(n_parts == 0 || parts[l], "null parts of product %lld", (long long)l);
    t.parts[l] = parts[l];
    if (n_tail == 0) continue;
    INCAGG_REQUIRE(a_tails && b_tails && lda && ldb && a_tails[l] && b_tails[l] && lda[l] >= m &&
                       ldb[l] >= n, "bad remainder rows of product %lld", (long long)l);
    t.a[l] = a_tails[l];
    t.b[l] = b_tails[l];
    t.lda[l] = lda[l];
    t.ldb[l] = ldb[l];
  }
  const long long per = (m * n + 255) / 256;
  INCAGG_REQUIRE(per * n_batch < INT_MAX, "grid too large");
  hipLaunchKernelGGL(tn_finish_batched_kernel, dim3((unsigned)(per * n_batch)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), t, (int)n_parts, (int)n_tail, (int)m, (int)n,
                     (int)per, out);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

extern "C" int incagg_tn_finish_f32(const float *parts, int64_t n_parts, const float *a_tail,
                                    int64_t lda, const float *b_tail, int64_t ldb, int64_t n_tail,
                                    int64_t m, int64_t n, float *out, int64_t ldo, void *stream) {
  INCAGG_REQUIRE(m >= 0 && n >= 0 && n_parts >= 0 && n_tail >= 0, "negative size");
  INCAGG_REQUIRE(m < INT_MAX && n < INT_MAX && m * n < (1ll << 40) && n_parts < INT_MAX &&
                     n_tail < INT_MAX, "size out of range");
  if (m == 0 || n == 0) return INCAGG_OK;
  INCAGG_REQUIRE(out && ldo >= n, "null output or ldo < n");
  INCAGG_REQUIRE(n_parts == 0 || parts, "null parts");
  INCAGG_REQUIRE(n_tail == 0 || (a_tail && b_tail && lda >= m && ldb >= n), "bad remainder rows");
  const long long total = m * n;
  const long long blocks = (total + 255) / 256;
  INCAGG_REQUIRE(blocks < INT_MAX, "grid too large");
  hipLaunchKernelGGL(tn_finish_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), parts, (int)n_parts, m * n, a_tail, lda,
                     b_tail, ldb, (int)n_tail, (int)m, (int)n, out, ldo);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

// =====================================================================================
// The rest of the dense side's glue: reductions and element-wise passes that torch runs in
// several launches with little parallelism each.  None of them has a counterpart in the
// reference's own sources (it calls torch.nn.functional / torch.optim, main.py:60-90).
// =====================================================================================
namespace incagg {

// ---- column sums of a tall matrix (bias gradients): out[c] = sum_r x[r, c] ----------------
// Two passes over a caller-provided workspace: every workgroup sums `rows_per_block` rows (a
// wave reads 64 x 16 bytes of one row per instruction, the four waves take rows r, r+4, ...)
// into one partial row; the second pass sums the partial rows.  Fixed summation order.
constexpr int kColsumCols = 256;  // columns per workgroup: 64 lanes x 4

__global__ __launch_bounds__(256) void colsum_kernel(const float *x, long long ldx, int rows,
                                                     int width, int rows_per_block, float *out,
                                                     long long ldo, int vec) {
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * kColsumCols + lane * 4;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec) {  // rows 16-byte aligned, width a multiple of 4
    if (c < width) {
      int r = r0 + wave;
      for (; r + 12 < r1; r += 16) {  // four loads in flight
        const float4 a = *reinterpret_cast<const float4 *>(x + (long long)r * ldx + c);
        const float4 b = *reinterpret_cast<const float4 *>(x + (long long)(r + 4) * ldx + c);
        const float4 d = *reinterpret_cast<const float4 *>(x + (long long)(r + 8) * ldx + c);
        const float4 e = *reinterpret_cast<const float4 *>(x + (long long)(r + 12) * ldx + c);
        acc.x += (a.x + b.x) + (d.x + e.x);
        acc.y += (a.y + b.y) + (d.y + e.y);
        acc.z += (a.z + b.z) + (d.z + e.z);
        acc.w += (a.w + b.w) + (d.w + e.w);
      }
      for (; r < r1; r += 4) {
        const float4 a = *reinterpret_cast<const float4 *>(x + (long long)r * ldx + c);
        acc.x += a.x;
        acc.y += a.y;
        acc.z += a.z;
        acc.w += a.w;
      }
    }
  } else {
    for (int r = r0 + wave; r < r1; r += 4) {
      const float *p = x + (long long)r * ldx + c;
      if (c < width) acc.x += p[0];
      if (c + 1 < width) acc.y += p[1];
      if (c + 2 < width) acc.z += p[2];
      if (c + 3 < width) acc.w += p[3];
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  const float4 a = part[0][lane], b = part[1][lane], d = part[2][lane], e = part[3][lane];
  float *o = out + (long long)blockIdx.x * ldo + c;
  if (c < width) o[0] = (a.x + b.x) + (d.x + e.x);
  if (c + 1 < width) o[1] = (a.y + b.y) + (d.y + e.y);
  if (c + 2 < width) o[2] = (a.z + b.z) + (d.z + e.z);
  if (c + 3 < width) o[3] = (a.w + b.w) + (d.w + e.w);
}

// Second pass: few rows (the partial sums), so 64 threads share a group of 16 columns -- thread
// (r, q) adds rows r, r + 64, ... of column quad q -- and meet in LDS.
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float *x, long long ldx, int rows,
                                                            int width, float *out, int vec) {
  __shared__ float4 part[64][4];
  const int q = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = blockIdx.x * 16 + q * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = rl; r < rows; r += 64) {
    const float *p = x + (long long)r * ldx + c;
    if (vec) {
      if (c < width) {
        const float4 a = *reinterpret_cast<const float4 *>(p);
        acc.x += a.x;
        acc.y += a.y;
        acc.z += a.z;
        acc.w += a.w;
      }
    } else {
      if (c < width) acc.x += p[0];
      if (c + 1 < width) acc.y += p[1];
      if (c + 2 < width) acc.z += p[2];
      if (c + 3 < width) acc.w += p[3];
    }
  }
  part[rl][q] = acc;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (rl < o) {
      const float4 b = part[rl + o][q];
      float4 a = part[rl][q];
      a.x += b.x;
      a.y += b.y;
      a.z += b.z;
      a.w += b.w;
      part[rl][q] = a;
    }
    __syncthreads();
  }
  if (rl != 0) return;
  const float4 a = part[0][q];
  if (c < width) out[c] = a.x;
  if (c + 1 < width) out[c + 1] = a.y;
  if (c + 2 < width) out[c + 2] = a.z;
  if (c + 3 < width) out[c + 3] = a.w;
}

static inline int colsum_rows_per_block(long long rows) {
  return (int)std::max<long long>(64, (rows + 1023) / 1024);  // at most 1024 partial rows
}

// ---- mean softmax cross entropy over rows, and its gradient --------------------------------
// One wave per row.  Forward: lse[r] = log sum_c exp(logits[r, c]) and one partial loss sum per
// workgroup (summed by a single workgroup afterwards: fixed order).  Backward:
// dlogits[r, c] = dloss * (exp(logits[r, c] - lse[r]) - [c == label[r]]) / rows.
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void xent_forward_kernel(const float *logits, long long ld,
                                                           const long long *labels, int rows,
                                                           int classes, float *lse, float *part,
                                                           int *err) {
  __shared__ float row_loss[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  float loss = 0.f;
  if (r < rows) {
    const float *p = logits + (long long)r * ld;
    float mx = -INFINITY;
    for (int c = lane; c < classes; c += 64) mx = fmaxf(mx, p[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < classes; c += 64) s += __expf(p[c] - mx);
    const float l = mx + __logf(wave_sum(s));
    const long long y = labels[r];
    if (y < 0 || y >= classes) {
      if (lane == 0) *err = 1;  // reported; the row counts as loss 0
    } else {
      loss = l - p[y];
    }
    if (lane == 0) lse[r] = l;
  }
  if (lane == 0) row_loss[wave] = loss;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (row_loss[0] + row_loss[1]) + (row_loss[2] + row_loss[3]);
}

__global__ __launch_bounds__(256) void xent_finish_kernel(const float *part, int n, float inv_rows,
                                                          float *loss) {
  __shared__ float acc[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  acc[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) acc[threadIdx.x] += acc[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = acc[0] * inv_rows;
}

__global__ __launch_bounds__(256) void xent_backward_kernel(const float *logits, long long ld,
                                                            const long long *labels,
                                                            const float *lse, int rows, int classes,
                                                            const float *dloss, float *dlogits,
                                                            long long ldd) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float g = *dloss / (float)rows, l = lse[r];
  const long long y = labels[r];
  const bool ok = y >= 0 && y < classes;
  const float *p = logits + (long long)r * ld;
  float *d = dlogits + (long long)r * ldd;
  for (int c = lane; c < classes; c += 64)
    d[c] = ok ? g * (__expf(p[c] - l) - (c == y ? 1.f : 0.f)) : 0.f;
}

// ---- Adam step over a list of parameter tensors, one launch -------------------------------
// torch.optim.Adam's update (no amsgrad, no maximize): g' = g + wd p; m += (g' - m)(1 - b1);
// v = b2 v + (1 - b2) g'^2; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), with
// t = *step + 1, every tensor's own count, read from device memory (the launch can sit in a HIP
// graph) and advanced by a one-wave launch behind it (counting arrivals on one address to find
// the last workgroup costs more: ~1000 same-address atomics take 20 us).  Workgroup b handles
// 1024 consecutive elements of ONE tensor.
constexpr int kAdamSegments = 24;
struct AdamTable {
  float *p[kAdamSegments];
  const float *g[kAdamSegments];
  float *m[kAdamSegments];
  float *v[kAdamSegments];
  float *step[kAdamSegments];
  int n[kAdamSegments];
  int first_block[kAdamSegments + 1];
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamTable t, int n_seg, float lr, float b1,
                                                   float b2, float eps, float wd) {
  int seg = 0;
  while (seg + 1 < n_seg && (int)blockIdx.x >= t.first_block[seg + 1]) seg++;  // scalar
  const float tt = *t.step[seg] + 1.f;
  const int n = t.n[seg];
  const int i0 = ((int)blockIdx.x - t.first_block[seg]) * 1024 + (int)threadIdx.x * 4;
  const float bc1 = 1.f - powf(b1, tt), bc2_sqrt = sqrtf(1.f - powf(b2, tt));
  const float step_size = lr / bc1;
  float *p = t.p[seg], *m = t.m[seg], *v = t.v[seg];
  const float *g = t.g[seg];
  auto one = [&](float &pp, float gg, float &mm, float &vv) {
    gg += wd * pp;
    mm += (gg - mm) * (1.f - b1);
    vv = vv * b2 + (1.f - b2) * gg * gg;
    pp -= step_size * mm / (sqrtf(vv) / bc2_sqrt + eps);
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                     reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
  if (i0 + 4 <= n && vec) {
    float4 pp = *reinterpret_cast<float4 *>(p + i0), mm = *reinterpret_cast<float4 *>(m + i0);
    float4 vv = *reinterpret_cast<float4 *>(v + i0);
    const float4 gg = *reinterpret_cast<const float4 *>(g + i0);
    one(pp.x, gg.x, mm.x, vv.x);
    one(pp.y, gg.y, mm.y, vv.y);
    one(pp.z, gg.z, mm.z, vv.z);
    one(pp.w, gg.w, mm.w, vv.w);
    *reinterpret_cast<float4 *>(p + i0) = pp;
    *reinterpret_cast<float4 *>(m + i0) = mm;
    *reinterpret_cast<float4 *>(v + i0) = vv;
  } else {
    for (int i = i0; i < min(i0 + 4, n); i++) one(p[i], g[i], m[i], v[i]);
  }
}

// Advances the step counts of one launch's tensors, after that launch (a counter shared by
// several tensors is advanced once, by the first segment that names it).
__global__ void adam_bump_kernel(const AdamTable t, int n_seg) {
  const int k = threadIdx.x;
  if (k >= n_seg) return;
  for (int j = 0; j < k; j++)
    if (t.step[j] == t.step[k]) return;
  *t.step[k] += 1.f;
}

}  // namespace incagg

extern "C" int64_t incagg_column_sums_ws_floats(int64_t rows, int64_t width) {
  if (rows < 0 || width < 0) return fail(INCAGG_E_INVALID, "negative size");
  const int rpb = colsum_rows_per_block(rows);
  return ((rows + rpb - 1) / rpb) * width;
}

extern "C" int incagg_column_sums_f32(const float *x, int64_t ldx, int64_t rows, int64_t width,
                                      float *ws, int64_t ws_floats, float *out, void *stream_) {
  INCAGG_REQUIRE(rows >= 0 && width >= 0 && rows < INT_MAX && width < INT_MAX, "size out of range");
  if (width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(out != nullptr, "null output");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (rows == 0) {
    INCAGG_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)width, stream));
    return INCAGG_OK;
  }
  INCAGG_REQUIRE(x && ldx >= width, "null input or ldx < width");
  const int rpb = colsum_rows_per_block(rows);
  const int n_part = (int)((rows + rpb - 1) / rpb);
  INCAGG_REQUIRE(ws && ws_floats >= (int64_t)n_part * width, "workspace too small: %lld < %lld",
                 (long long)ws_floats, (long long)n_part * width);
  const unsigned col_blocks = (unsigned)((width + kColsumCols - 1) / kColsumCols);
  const int vec = aligned16(x) && ldx % 4 == 0 && width % 4 == 0;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)n_part, col_blocks), dim3(256), 0, stream, x, ldx,
                     (int)rows, (int)width, rpb, ws, width, vec);
  INCAGG_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((width + 15) / 16)), dim3(256), 0, stream,
                     ws, width, n_part, (int)width, out, (int)(aligned16(ws) && width % 4 == 0));
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

extern "C" int64_t incagg_softmax_xent_ws_floats(int64_t rows) {
  if (rows < 0) return fail(INCAGG_E_INVALID, "negative size");
  return rows + (rows + 3) / 4;  // lse per row + one partial loss per workgroup
}

extern "C" int incagg_softmax_xent_forward_f32(const float *logits, int64_t ld,
                                               const int64_t *labels, int64_t rows,
                                               int64_t classes, float *ws, int64_t ws_floats,
                                               float *loss, void *stream_) {
  INCAGG_REQUIRE(rows >= 1 && classes >= 1 && rows < INT_MAX - 4 && classes < INT_MAX,
                 "size out of range");
  INCAGG_REQUIRE(logits && labels && loss && ld >= classes, "null argument or ld < classes");
  INCAGG_REQUIRE(ws && ws_floats >= incagg_softmax_xent_ws_floats(rows), "workspace too small");
  int *err = nullptr;
  int rc = error_flag_address(&err);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int blocks = (int)((rows + 3) / 4);
  hipLaunchKernelGGL(xent_forward_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, logits, ld,
                     reinterpret_cast<const long long *>(labels), (int)rows, (int)classes, ws,
                     ws + rows, err);
  INCAGG_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(xent_finish_kernel, dim3(1), dim3(256), 0, stream, ws + rows, blocks,
                     1.0f / (float)rows, loss);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

extern "C" int incagg_softmax_xent_backward_f32(const float *logits, int64_t ld,
                                                const int64_t *labels, int64_t rows,
                                                int64_t classes, const float *ws,
                                                const float *dloss, float *dlogits, int64_t ldd,
                                                void *stream_) {
  INCAGG_REQUIRE(rows >= 1 && classes >= 1 && rows < INT_MAX - 4 && classes < INT_MAX,
                 "size out of range");
  INCAGG_REQUIRE(logits && labels && ws && dloss && dlogits && ld >= classes && ldd >= classes,
                 "null argument or a pitch below the class count");
  hipLaunchKernelGGL(xent_backward_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), logits, ld,
                     reinterpret_cast<const long long *>(labels), ws, (int)rows, (int)classes, dloss,
                     dlogits, ldd);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

extern "C" int incagg_adam_step_f32(const incagg_adam_segment *segs, int64_t n_segs, float lr,
                                    float beta1, float beta2, float eps, float weight_decay,
                                    void *stream_) {
  INCAGG_REQUIRE(n_segs >= 0 && (n_segs == 0 || segs), "bad segment list");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // launches of up to kAdamSegments tensors each
  int64_t first = 0;
  while (first < n_segs) {
    AdamTable t{};
    int k = 0, blocks = 0;
    int64_t i = first;
    for (; i < n_segs && k < kAdamSegments; i++) {
      const incagg_adam_segment &s = segs[i];
      INCAGG_REQUIRE(s.n >= 0 && s.n < INT_MAX - 1024, "segment length out of range");
      if (s.n == 0) continue;
      INCAGG_REQUIRE(s.p && s.g && s.m && s.v && s.step, "null segment pointer");
      t.step[k] = s.step;
      t.p[k] = s.p;
      t.g[k] = s.g;
      t.m[k] = s.m;
      t.v[k] = s.v;
      t.n[k] = (int)s.n;
      t.first_block[k] = blocks;
      blocks += (int)((s.n + 1023) / 1024);
      INCAGG_REQUIRE(blocks < (1 << 30), "grid too large");
      k++;
    }
    t.first_block[k] = blocks;
    first = i;
    if (k == 0) break;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t, k, lr, beta1,
                       beta2, eps, weight_decay);
    INCAGG_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(64), 0, stream, t, k);
    INCAGG_HIP_TRY(hipGetLastError());
  }
  return INCAGG_OK;
}

// =====================================================================================
// Element-wise passes of the dense side that torch runs as two launches or at half the
// streaming rate, and the weight fold of the GCNII layers.
// =====================================================================================
namespace incagg {

// x = max(x, 0) in place over [rows, width] (pitch ld).
template <int VEC>
__global__ __launch_bounds__(256) void relu_rows_kernel(float *x, long long ld, long long rows,
                                                        int width) {
  const int per_row = width / VEC;
  const long long total = rows * per_row;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / per_row;
    float *p = x + r * ld + (int)(i - r * per_row) * VEC;
    if (VEC == 4) {
      float4 v = *reinterpret_cast<float4 *>(p);
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
      *reinterpret_cast<float4 *>(p) = v;
    } else {
      *p = fmaxf(*p, 0.f);
    }
  }
}

// ADD: dst += src wherever gate > 0 (the backward of a relu added onto an existing gradient);
// otherwise dst = src wherever gate > 0, else 0 (the backward of a relu; dst may be src).
template <int VEC, bool ADD>
__global__ __launch_bounds__(256) void gate_add_rows_kernel(float *dst, long long ldd,
                                                            const float *src, long long lds,
                                                            const float *gate, long long ldg,
                                                            long long rows, int width) {
  const int per_row = width / VEC;
  const long long total = rows * per_row;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / per_row;
    const int c = (int)(i - r * per_row) * VEC;
    float *d = dst + r * ldd + c;
    const float *s = src + r * lds + c, *g = gate + r * ldg + c;
    if (VEC == 4) {
      float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ADD) dv = *reinterpret_cast<float4 *>(d);
      const float4 sv = *reinterpret_cast<const float4 *>(s);
      const float4 gv = *reinterpret_cast<const float4 *>(g);
      dv.x += gv.x > 0.f ? sv.x : 0.f;
      dv.y += gv.y > 0.f ? sv.y : 0.f;
      dv.z += gv.z > 0.f ? sv.z : 0.f;
      dv.w += gv.w > 0.f ? sv.w : 0.f;
      *reinterpret_cast<float4 *>(d) = dv;
    } else {
      const float v = *g > 0.f ? *s : 0.f;
      *d = ADD ? *d + v : v;
    }
  }
}

// out = scale * (src_0 + src_1 + ... ), every operand with a row pitch of its own: the gradients
// that the K propagation steps of APPNP hand back for their shared x_0 term summed in one pass
// (and, with one source, a scaled copy at the output's pitch).
constexpr int kSumSources = 16;
struct SumSources {
  const float *p[kSumSources];
  long long ld[kSumSources];
};

template <int VEC>
__global__ __launch_bounds__(256) void rows_sum_kernel(const SumSources s, int n, float scale,
                                                       long long rows, int width, float *out,
                                                       long long ldo) {
  const int per_row = width / VEC;
  const long long total = rows * per_row;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / per_row;
    const int c = (int)(i - r * per_row) * VEC;
    if (VEC == 4) {
      float4 acc = *reinterpret_cast<const float4 *>(s.p[0] + r * s.ld[0] + c);
      for (int k = 1; k < n; k++) {   // (k is uniform: the pointers come from scalar loads)
        const float4 v = *reinterpret_cast<const float4 *>(s.p[k] + r * s.ld[k] + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
      *reinterpret_cast<float4 *>(out + r * ldo + c) = acc;
    } else {
      float acc = s.p[0][r * s.ld[0] + c];
      for (int k = 1; k < n; k++) acc += s.p[k][r * s.ld[k] + c];
      out[r * ldo + c] = acc * scale;
    }
  }
}

// GCNII with separate weights (models/gcn2.py:46-48 through PyG's GCN2Conv):
//   out = (1 - alpha) (A x) ((1 - beta_l) I + beta_l W1_l) + alpha x_0 ((1 - beta_l) I + beta_l W2_l).
// The identity parts are folded into the weights of all L layers by one launch:
//   w1f[l]               = (1 - alpha) (beta_l W1_l + (1 - beta_l) I)         [L, C, C]
//   w2cat[:, lC:(l+1)C]  = alpha (beta_l W2_l + (1 - beta_l) I)               [C, L C]
// from ws = [W1_0 .. W1_{L-1}, W2_0 .. W2_{L-1}] ([2L, C, C]); the backward kernel maps the
// gradients of w1f / w2cat back onto that stack.
constexpr int kFoldLayers = 64;
struct FoldBetas {
  float beta[kFoldLayers];
};

__global__ __launch_bounds__(256) void fold_weights_kernel(const float *ws, int n_layers, int c,
                                                           float alpha, const FoldBetas b,
                                                           float *w1f, float *w2cat) {
  const long long cc = (long long)c * c, half = n_layers * cc;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * half) return;
  const bool second = idx >= half;
  const long long k = second ? idx - half : idx;
  const int l = (int)(k / cc), i = (int)((k - l * cc) / c), j = (int)(k - l * cc - (long long)i * c);
  const float beta = b.beta[l];
  const float v = (second ? alpha : 1.f - alpha) * (beta * ws[idx] + (i == j ? 1.f - beta : 0.f));
  if (second)
    w2cat[(long long)i * n_layers * c + (long long)l * c + j] = v;
  else
    w1f[k] = v;
}

__global__ __launch_bounds__(256) void fold_weights_backward_kernel(const float *dw1f,
                                                                    const float *dw2cat,
                                                                    int n_layers, int c, float alpha,
                                                                    const FoldBetas b, float *dws) {
  const long long cc = (long long)c * c, half = n_layers * cc;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * half) return;
  const bool second = idx >= half;
  const long long k = second ? idx - half : idx;
  const int l = (int)(k / cc), i = (int)((k - l * cc) / c), j = (int)(k - l * cc - (long long)i * c);
  const float g = second ? dw2cat[(long long)i * n_layers * c + (long long)l * c + j] : dw1f[k];
  dws[idx] = (second ? alpha : 1.f - alpha) * b.beta[l] * g;
}

static inline bool rows_vec4_ok(const void *p, long long ld, long long width) {
  return aligned16(p) && ld % 4 == 0 && width % 4 == 0;
}

}  // namespace incagg

extern "C" int incagg_relu_rows_f32(float *x, int64_t ld, int64_t rows, int64_t width,
                                    void *stream) {
  INCAGG_REQUIRE(rows >= 0 && width >= 0 && width < INT_MAX, "bad size");
  if (rows == 0 || width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(x && ld >= width, "null pointer or ld < width");
  const bool vec = rows_vec4_ok(x, ld, width);
  const long long items = rows * (width / (vec ? 4 : 1));
  const int blocks = (int)std::min<long long>((items + 255) / 256, 8192);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec)
    hipLaunchKernelGGL((relu_rows_kernel<4>), dim3(blocks), dim3(256), 0, s, x, ld, rows, (int)width);
  else
    hipLaunchKernelGGL((relu_rows_kernel<1>), dim3(blocks), dim3(256), 0, s, x, ld, rows, (int)width);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

static int gate_rows(bool add, float *dst, int64_t ldd, const float *src, int64_t lds,
                     const float *gate, int64_t ldg, int64_t rows, int64_t width, void *stream) {
  INCAGG_REQUIRE(rows >= 0 && width >= 0 && width < INT_MAX, "bad size");
  if (rows == 0 || width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(dst && src && gate && ldd >= width && lds >= width && ldg >= width,
                 "null pointer or a pitch below the width");
  const bool vec = rows_vec4_ok(dst, ldd, width) && rows_vec4_ok(src, lds, width) &&
                   rows_vec4_ok(gate, ldg, width);
  const long long items = rows * (width / (vec ? 4 : 1));
  const int blocks = (int)std::min<long long>((items + 255) / 256, 8192);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec && add)
    hipLaunchKernelGGL((gate_add_rows_kernel<4, true>), dim3(blocks), dim3(256), 0, s, dst, ldd, src,
                       lds, gate, ldg, rows, (int)width);
  else if (vec)
    hipLaunchKernelGGL((gate_add_rows_kernel<4, false>), dim3(blocks), dim3(256), 0, s, dst, ldd, src,
                       lds, gate, ldg, rows, (int)width);
  else if (add)
    hipLaunchKernelGGL((gate_add_rows_kernel<1, true>), dim3(blocks), dim3(256), 0, s, dst, ldd, src,
                       lds, gate, ldg, rows, (int)width);
  else
    hipLaunchKernelGGL((gate_add_rows_kernel<1, false>), dim3(blocks), dim3(256), 0, s, dst, ldd, src,
                       lds, gate, ldg, rows, (int)width);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

extern "C" int incagg_gate_add_rows_f32(float *dst, int64_t ldd, const float *src, int64_t lds,
                                        const float *gate, int64_t ldg, int64_t rows, int64_t width,
                                        void *stream) {
  return gate_rows(true, dst, ldd, src, lds, gate, ldg, rows, width, stream);
}

extern "C" int incagg_gate_rows_f32(float *dst, int64_t ldd, const float *src, int64_t lds,
                                    const float *gate, int64_t ldg, int64_t rows, int64_t width,
                                    void *stream) {
  return gate_rows(false, dst, ldd, src, lds, gate, ldg, rows, width, stream);
}

extern "C" int incagg_rows_sum_f32(const float *const *srcs, const int64_t *lds, int64_t n_srcs,
                                   float scale, int64_t rows, int64_t width, float *out,
                                   int64_t ldo, void *stream) {
  INCAGG_REQUIRE(rows >= 0 && width >= 0 && width < INT_MAX, "bad size");
  INCAGG_REQUIRE(n_srcs >= 1 && n_srcs <= kSumSources, "%lld sources: 1..%d are supported",
                 (long long)n_srcs, kSumSources);
  if (rows == 0 || width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(srcs && lds && out && ldo >= width, "null pointer or ldo < width");
  SumSources ss{};
  bool vec = rows_vec4_ok(out, ldo, width);
  for (int64_t k = 0; k < n_srcs; k++) {
    INCAGG_REQUIRE(srcs[k] && lds[k] >= width, "source %lld: null or pitch below the width",
                   (long long)k);
    INCAGG_REQUIRE((const void *)srcs[k] != (const void *)out, "out must not alias a source");
    ss.p[k] = srcs[k];
    ss.ld[k] = lds[k];
    vec = vec && rows_vec4_ok(srcs[k], lds[k], width);
  }
  const long long items = rows * (width / (vec ? 4 : 1));
  const int blocks = (int)std::min<long long>((items + 255) / 256, 8192);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec)
    hipLaunchKernelGGL((rows_sum_kernel<4>), dim3(blocks), dim3(256), 0, s, ss, (int)n_srcs, scale,
                       rows, (int)width, out, ldo);
  else
    hipLaunchKernelGGL((rows_sum_kernel<1>), dim3(blocks), dim3(256), 0, s, ss, (int)n_srcs, scale,
                       rows, (int)width, out, ldo);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

static int fold_args(int64_t n_layers, int64_t c, const float *beta, FoldBetas &b, long long &blocks) {
  INCAGG_REQUIRE(n_layers >= 1 && n_layers <= kFoldLayers, "1..%d layers", kFoldLayers);
  INCAGG_REQUIRE(c >= 1 && c <= 16384 && beta, "bad width or null beta");
  for (int64_t l = 0; l < n_layers; l++) b.beta[l] = beta[l];
  blocks = (2 * n_layers * c * c + 255) / 256;
  INCAGG_REQUIRE(blocks < INT_MAX, "grid too large");
  return INCAGG_OK;
}

extern "C" int incagg_gcn2_fold_weights_f32(const float *ws, int64_t n_layers, int64_t c,
                                            float alpha, const float *beta, float *w1f,
                                            float *w2cat, void *stream) {
  FoldBetas b{};
  long long blocks = 0;
  int rc = fold_args(n_layers, c, beta, b, blocks);
  if (rc) return rc;
  INCAGG_REQUIRE(ws && w1f && w2cat, "null pointer");
  hipLaunchKernelGGL(fold_weights_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), ws, (int)n_layers, (int)c, alpha, b, w1f,
                     w2cat);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

extern "C" int incagg_gcn2_fold_weights_backward_f32(const float *dw1f, const float *dw2cat,
                                                     int64_t n_layers, int64_t c, float alpha,
                                                     const float *beta, float *dws, void *stream) {
  FoldBetas b{};
  long long blocks = 0;
  int rc = fold_args(n_layers, c, beta, b, blocks);
  if (rc) return rc;
  INCAGG_REQUIRE(dw1f && dw2cat && dws, "null pointer");
  hipLaunchKernelGGL(fold_weights_backward_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dw1f, dw2cat, (int)n_layers, (int)c, alpha, b,
                     dws);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}
