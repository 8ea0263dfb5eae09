// Last step of a tall "transposed times normal" product a^T b ([rows, m]^T [rows, n], rows >> m, n:
// every weight gradient of the models here).  The caller runs the bulk as ONE batched GEMM over S
// equal row slabs (hipBLASLt) and hands over the S partial [m, n] products plus the R < S
// remainder rows that did not fit a slab; this kernel adds everything up:
//   out[i, j] = sum_s parts[s, i, j] + sum_r a_tail[r, i] * b_tail[r, j]
// in place of a slab-sum kernel followed by a GEMM with a reduction of length R (which costs a
// full hipBLASLt launch, ~14 us, for a few thousand multiply-adds).  fp32, fixed summation order.
#include "common.h"

#include <climits>

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

}  // namespace incagg

using namespace incagg;

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
