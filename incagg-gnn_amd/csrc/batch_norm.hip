// Batch normalisation over the rows of a tall row-major matrix ([rows, C], rows >> C: the layer
// outputs of a batch; models/gcn.py:145,181,248, models/gcn2.py:132,161,267,
// models/graphsage.py:191 call torch.nn.BatchNorm1d there), training mode: statistics of the
// batch, running statistics updated, and the backward pass.
//
// Every pass streams the matrix the way it lies in memory: a wave reads 64 x 16 bytes of ONE row
// per instruction, the four waves of a workgroup take rows r, r+4, ..., a workgroup sums a slab of
// rows into one partial row, a second small launch adds the partial rows up (in double).  (The
// framework's own kernel for this layout gives every channel to one workgroup, which then walks
// down a column of a row-major matrix: 13 ms for the backward pass of a 169k x 256 matrix on
// MI355X against 0.2 ms here.)
//
// Moments are accumulated on data shifted by the matrix's first row, x - x[0, :]: the shift is
// within a few standard deviations of the column mean, so E[d^2] - E[d]^2 does not cancel the
// way it would on raw data with a large mean.
//
//   forward : mean, rstd = 1 / sqrt(var + eps) (biased variance), y = (x - mean) rstd gamma + beta
//             running_mean = (1 - m) running_mean + m mean,
//             running_var  = (1 - m) running_var  + m var rows / (rows - 1)
//   backward: dbeta = sum dy, dgamma = sum dy xhat, xhat = (x - mean) rstd,
//             dx = gamma rstd (dy - dbeta / rows - xhat dgamma / rows)
#include "common.h"

#include <algorithm>
#include <climits>

namespace incagg {

constexpr int kBnCols = 256;  // columns per workgroup: 64 lanes x 4

static inline int bn_rows_per_block(long long rows) {
  return (int)std::max<long long>(64, (rows + 1023) / 1024);  // at most 1024 partial rows
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// part[(2 * block + k) * width + c]: k = 0 sum of a, k = 1 sum of b over the block's rows, where
// MOMENTS: a = x - shift, b = a^2;   otherwise: a = dy, b = dy * (x - mean) * rstd
template <bool MOMENTS>
__global__ __launch_bounds__(256) void bn_sums_kernel(const float *u, long long ldu, const float *x,
                                                      long long ldx, int rows, int width,
                                                      int rows_per_block, const float *mean,
                                                      const float *rstd, float *part) {
  __shared__ float4 red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * kBnCols + lane * 4;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
  if (c < width) {
    // MOMENTS: the shift (row 0 of x); otherwise mean and rstd of the column quad
    const float4 m = MOMENTS ? ld4(x + c) : ld4(mean + c);
    const float4 s = MOMENTS ? make_float4(1.f, 1.f, 1.f, 1.f) : ld4(rstd + c);
    int r = r0 + wave;
    for (; r + 4 < r1; r += 8) {  // two rows (four loads) in flight
      const float4 x0 = ld4(x + (long long)r * ldx + c), x1 = ld4(x + (long long)(r + 4) * ldx + c);
      float4 a0, a1, b0, b1;
      if (MOMENTS) {
        a0 = make_float4(x0.x - m.x, x0.y - m.y, x0.z - m.z, x0.w - m.w);
        a1 = make_float4(x1.x - m.x, x1.y - m.y, x1.z - m.z, x1.w - m.w);
        b0 = make_float4(a0.x * a0.x, a0.y * a0.y, a0.z * a0.z, a0.w * a0.w);
        b1 = make_float4(a1.x * a1.x, a1.y * a1.y, a1.z * a1.z, a1.w * a1.w);
      } else {
        a0 = ld4(u + (long long)r * ldu + c);
        a1 = ld4(u + (long long)(r + 4) * ldu + c);
        b0 = make_float4(a0.x * (x0.x - m.x) * s.x, a0.y * (x0.y - m.y) * s.y,
                         a0.z * (x0.z - m.z) * s.z, a0.w * (x0.w - m.w) * s.w);
        b1 = make_float4(a1.x * (x1.x - m.x) * s.x, a1.y * (x1.y - m.y) * s.y,
                         a1.z * (x1.z - m.z) * s.z, a1.w * (x1.w - m.w) * s.w);
      }
      sa.x += a0.x + a1.x; sa.y += a0.y + a1.y; sa.z += a0.z + a1.z; sa.w += a0.w + a1.w;
      sb.x += b0.x + b1.x; sb.y += b0.y + b1.y; sb.z += b0.z + b1.z; sb.w += b0.w + b1.w;
    }
    for (; r < r1; r += 4) {
      const float4 x0 = ld4(x + (long long)r * ldx + c);
      float4 a0, b0;
      if (MOMENTS) {
        a0 = make_float4(x0.x - m.x, x0.y - m.y, x0.z - m.z, x0.w - m.w);
        b0 = make_float4(a0.x * a0.x, a0.y * a0.y, a0.z * a0.z, a0.w * a0.w);
      } else {
        a0 = ld4(u + (long long)r * ldu + c);
        b0 = make_float4(a0.x * (x0.x - m.x) * s.x, a0.y * (x0.y - m.y) * s.y,
                         a0.z * (x0.z - m.z) * s.z, a0.w * (x0.w - m.w) * s.w);
      }
      sa.x += a0.x; sa.y += a0.y; sa.z += a0.z; sa.w += a0.w;
      sb.x += b0.x; sb.y += b0.y; sb.z += b0.z; sb.w += b0.w;
    }
  }
  red[0][wave][lane] = sa;
  red[1][wave][lane] = sb;
  __syncthreads();
  if (wave >= 2 || c >= width) return;  // wave 0 writes the sums of a, wave 1 those of b
  const float4 p = red[wave][0][lane], q = red[wave][1][lane], s = red[wave][2][lane],
               t = red[wave][3][lane];
  float *o = part + ((long long)(2 * blockIdx.x + wave)) * width + c;
  *reinterpret_cast<float4 *>(o) =
      make_float4((p.x + q.x) + (s.x + t.x), (p.y + q.y) + (s.y + t.y), (p.z + q.z) + (s.z + t.z),
                  (p.w + q.w) + (s.w + t.w));
}

// Adds the partial rows up in double: 16 threads share a column (thread g of them takes partial
// rows g, g + 16, ...) and meet in LDS; a workgroup covers 16 columns.
// MOMENTS: mean / rstd out, running statistics updated; otherwise dbeta / dgamma out.
template <bool MOMENTS>
__global__ __launch_bounds__(256) void bn_finish_kernel(const float *part, int n_part, int width,
                                                        int rows, const float *x, float eps,
                                                        float momentum, float *running_mean,
                                                        float *running_var, float *out_a,
                                                        float *out_b) {
  __shared__ double red[2][16][17];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double sa = 0.0, sb = 0.0;
  if (c < width) {
    for (int p = g; p < n_part; p += 16) {
      sa += (double)part[(long long)(2 * p) * width + c];
      sb += (double)part[(long long)(2 * p + 1) * width + c];
    }
  }
  red[0][g][cl] = sa;
  red[1][g][cl] = sb;
  __syncthreads();
  if (g != 0 || c >= width) return;
  sa = sb = 0.0;
  for (int k = 0; k < 16; k++) {   // fixed order
    sa += red[0][k][cl];
    sb += red[1][k][cl];
  }
  if (!MOMENTS) {
    out_a[c] = (float)sa;   // dbeta
    out_b[c] = (float)sb;   // dgamma
    return;
  }
  const double n = (double)rows;
  const double md = sa / n;
  const double var = fmax(sb / n - md * md, 0.0);
  const double mean = (double)x[c] + md;
  out_a[c] = (float)mean;
  out_b[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
  if (running_var) {
    const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

// y = (x - mean) rstd gamma + beta
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *x, long long ldx, long long rows,
                                                       int width, const float *mean,
                                                       const float *rstd, const float *gamma,
                                                       const float *beta, float *y, long long ldy) {
  const int quads = width >> 2;
  const long long total = rows * quads;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / quads;
    const int c = (int)(i - r * quads) * 4;
    const float4 v = ld4(x + r * ldx + c), m = ld4(mean + c), s = ld4(rstd + c);
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gamma) g = ld4(gamma + c);
    if (beta) b = ld4(beta + c);
    *reinterpret_cast<float4 *>(y + r * ldy + c) =
        make_float4((v.x - m.x) * s.x * g.x + b.x, (v.y - m.y) * s.y * g.y + b.y,
                    (v.z - m.z) * s.z * g.z + b.z, (v.w - m.w) * s.w * g.w + b.w);
  }
}

// dx = gamma rstd (dy - dbeta / rows - xhat dgamma / rows)
__global__ __launch_bounds__(256) void bn_backward_apply_kernel(
    const float *dy, long long lddy, const float *x, long long ldx, long long rows, int width,
    const float *mean, const float *rstd, const float *gamma, const float *dbeta,
    const float *dgamma, float *dx, long long lddx) {
  const int quads = width >> 2;
  const long long total = rows * quads;
  const float inv = 1.0f / (float)rows;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / quads;
    const int c = (int)(i - r * quads) * 4;
    const float4 g = ld4(dy + r * lddy + c), v = ld4(x + r * ldx + c), m = ld4(mean + c),
                 s = ld4(rstd + c), db = ld4(dbeta + c), dg = ld4(dgamma + c);
    float4 w = make_float4(1.f, 1.f, 1.f, 1.f);
    if (gamma) w = ld4(gamma + c);
    *reinterpret_cast<float4 *>(dx + r * lddx + c) = make_float4(
        w.x * s.x * (g.x - db.x * inv - (v.x - m.x) * s.x * dg.x * inv),
        w.y * s.y * (g.y - db.y * inv - (v.y - m.y) * s.y * dg.y * inv),
        w.z * s.z * (g.z - db.z * inv - (v.z - m.z) * s.z * dg.z * inv),
        w.w * s.w * (g.w - db.w * inv - (v.w - m.w) * s.w * dg.w * inv));
  }
}

static inline bool bn_rows_ok(const void *p, long long ld) { return aligned16(p) && ld % 4 == 0; }

static inline unsigned bn_stream_grid(long long quads) {
  return (unsigned)std::max<long long>(1, std::min<long long>((quads + 255) / 256, 1 << 16));
}

}  // namespace incagg

using namespace incagg;

extern "C" int incagg_batch_norm_supported(int64_t rows, int64_t width) {
  return rows >= 2 && rows < INT_MAX && width >= 4 && width % 4 == 0 && width < (1 << 20);
}

extern "C" int64_t incagg_batch_norm_ws_floats(int64_t rows, int64_t width) {
  if (rows < 0 || width < 0) return fail(INCAGG_E_INVALID, "negative size");
  const int rpb = bn_rows_per_block(rows);
  return 2 * ((rows + rpb - 1) / rpb) * width;
}

extern "C" int incagg_batch_norm_forward_f32(const float *x, int64_t ldx, int64_t rows,
                                             int64_t width, const float *gamma, const float *beta,
                                             float eps, float momentum, float *running_mean,
                                             float *running_var, float *ws, int64_t ws_floats,
                                             float *mean, float *rstd, float *y, int64_t ldy,
                                             void *stream_) {
  INCAGG_REQUIRE(incagg_batch_norm_supported(rows, width),
                 "needs at least 2 rows and a width that is a multiple of 4 (rows %lld, width %lld)",
                 (long long)rows, (long long)width);
  INCAGG_REQUIRE(x && y && mean && rstd && ws, "null pointer");
  INCAGG_REQUIRE(ldx >= width && ldy >= width && bn_rows_ok(x, ldx) && bn_rows_ok(y, ldy),
                 "rows of x and y must be 16-byte aligned and at least `width` wide");
  INCAGG_REQUIRE(aligned16(mean) && aligned16(rstd) && aligned16(ws) && (!gamma || aligned16(gamma)) &&
                     (!beta || aligned16(beta)),
                 "per-column arrays must be 16-byte aligned");
  INCAGG_REQUIRE(eps >= 0.f && momentum >= 0.f && momentum <= 1.f, "bad eps / momentum");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rpb = bn_rows_per_block(rows);
  const int n_part = (int)((rows + rpb - 1) / rpb);
  INCAGG_REQUIRE(ws_floats >= 2ll * n_part * width, "workspace too small: %lld < %lld",
                 (long long)ws_floats, 2ll * n_part * width);
  const unsigned col_blocks = (unsigned)((width + kBnCols - 1) / kBnCols);
  hipLaunchKernelGGL(bn_sums_kernel<true>, dim3((unsigned)n_part, col_blocks), dim3(256), 0, stream,
                     (const float *)nullptr, 0ll, x, (long long)ldx, (int)rows, (int)width, rpb,
                     (const float *)nullptr, (const float *)nullptr, ws);
  hipLaunchKernelGGL(bn_finish_kernel<true>, dim3((unsigned)((width + 15) / 16)), dim3(256), 0, stream,
                     ws, n_part, (int)width, (int)rows, x, eps, momentum, running_mean, running_var,
                     mean, rstd);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_stream_grid(rows * (width / 4))), dim3(256), 0, stream, x,
                     (long long)ldx, (long long)rows, (int)width, mean, rstd, gamma, beta, y,
                     (long long)ldy);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

extern "C" int incagg_batch_norm_backward_f32(const float *dy, int64_t lddy, const float *x,
                                              int64_t ldx, int64_t rows, int64_t width,
                                              const float *gamma, const float *mean,
                                              const float *rstd, float *ws, int64_t ws_floats,
                                              float *dgamma, float *dbeta, float *dx, int64_t lddx,
                                              void *stream_) {
  INCAGG_REQUIRE(incagg_batch_norm_supported(rows, width),
                 "needs at least 2 rows and a width that is a multiple of 4 (rows %lld, width %lld)",
                 (long long)rows, (long long)width);
  INCAGG_REQUIRE(dy && x && mean && rstd && ws && dgamma && dbeta && dx, "null pointer");
  INCAGG_REQUIRE(lddy >= width && ldx >= width && lddx >= width && bn_rows_ok(dy, lddy) &&
                     bn_rows_ok(x, ldx) && bn_rows_ok(dx, lddx),
                 "rows of dy, x and dx must be 16-byte aligned and at least `width` wide");
  INCAGG_REQUIRE(aligned16(mean) && aligned16(rstd) && aligned16(ws) && aligned16(dgamma) &&
                     aligned16(dbeta) && (!gamma || aligned16(gamma)),
                 "per-column arrays must be 16-byte aligned");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rpb = bn_rows_per_block(rows);
  const int n_part = (int)((rows + rpb - 1) / rpb);
  INCAGG_REQUIRE(ws_floats >= 2ll * n_part * width, "workspace too small: %lld < %lld",
                 (long long)ws_floats, 2ll * n_part * width);
  const unsigned col_blocks = (unsigned)((width + kBnCols - 1) / kBnCols);
  hipLaunchKernelGGL(bn_sums_kernel<false>, dim3((unsigned)n_part, col_blocks), dim3(256), 0, stream, dy,
                     (long long)lddy, x, (long long)ldx, (int)rows, (int)width, rpb, mean, rstd, ws);
  hipLaunchKernelGGL(bn_finish_kernel<false>, dim3((unsigned)((width + 15) / 16)), dim3(256), 0, stream,
                     ws, n_part, (int)width, (int)rows, (const float *)nullptr, 0.f, 0.f,
                     (float *)nullptr, (float *)nullptr, dbeta, dgamma);
  hipLaunchKernelGGL(bn_backward_apply_kernel, dim3(bn_stream_grid(rows * (width / 4))), dim3(256), 0,
                     stream, dy, (long long)lddy, x, (long long)ldx, (long long)rows, (int)width, mean,
                     rstd, gamma, dbeta, dgamma, dx, (long long)lddx);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}
