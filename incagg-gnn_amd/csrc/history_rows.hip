// Row movement for the per-layer History tables on gfx950: gather (pull), indexed
// scatter (push) and contiguous-chunk copies with zero padding.  One wave moves
// whole rows with 16-B-per-lane accesses, four rows in flight per wave, so random
// rows of >= 256 B stream at close to the HBM rate.  All three are pure copies:
// algorithmic traffic is width*4 B read + width*4 B written per row (+ 8 B of
// index for gather / scatter).
#include "common.h"

#include <algorithm>
#include <climits>

namespace incagg {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kRowsInFlight = 4;

template <int VEC> struct RV;
template <> struct RV<1> { using type = float; };
template <> struct RV<4> { using type = f4; };

// GATHER: out[k] = table[idx[k]]   (SCATTER == false)
// SCATTER: table[idx[k]] = packed[k]
template <int VEC, bool SCATTER>
__global__ __launch_bounds__(256) void indexed_rows_kernel(float *table, long long ldt,
                                                          long long table_rows,
                                                          const long long *idx, long long k_rows,
                                                          int width, float *packed,
                                                          long long ldp, int *err) {
  using V = typename RV<VEC>::type;
  const int lane = threadIdx.x & 63;
  const long long wave = blockIdx.x * 4ll + (threadIdx.x >> 6);
  const long long n_waves = gridDim.x * 4ll;
  for (long long k0 = wave * kRowsInFlight; k0 < k_rows; k0 += n_waves * kRowsInFlight) {
    long long r[kRowsInFlight];
    bool ok[kRowsInFlight];
#pragma unroll
    for (int u = 0; u < kRowsInFlight; u++) {
      ok[u] = k0 + u < k_rows;
      r[u] = ok[u] ? idx[k0 + u] : 0;
      if (ok[u] && (unsigned long long)r[u] >= (unsigned long long)table_rows) {
        if (lane == 0) *err = 1;  // skip instead of faulting
        ok[u] = false;
      }
    }
    for (int c = lane * VEC; c < width; c += 64 * VEC) {
      V v[kRowsInFlight];
#pragma unroll
      for (int u = 0; u < kRowsInFlight; u++) {
        if (!ok[u]) continue;
        const float *src = SCATTER ? packed + (k0 + u) * ldp + c : table + r[u] * ldt + c;
        v[u] = *reinterpret_cast<const V *>(src);
      }
#pragma unroll
      for (int u = 0; u < kRowsInFlight; u++) {
        if (!ok[u]) continue;
        float *dst = SCATTER ? table + r[u] * ldt + c : packed + (k0 + u) * ldp + c;
        *reinterpret_cast<V *>(dst) = v[u];
      }
    }
  }
}

__device__ __forceinline__ float relu_of(float v) { return fmaxf(v, 0.f); }
__device__ __forceinline__ f4 relu_of(f4 v) {
  f4 r;
  r.x = fmaxf(v.x, 0.f); r.y = fmaxf(v.y, 0.f); r.z = fmaxf(v.z, 0.f); r.w = fmaxf(v.w, 0.f);
  return r;
}

constexpr int kMaxChunks = 64;   // (one boundary per lane of a wave: see chunk_rows_kernel)

struct ChunkTable {
  long long dst_row[kMaxChunks];
  long long src_row[kMaxChunks];
  long long first[kMaxChunks + 1];  // running row count; first[n] = rows in this launch
  int n;
};

// dst[dst_row[c] + i, :] = src[src_row[c] + i, :src_width], zero-filled up to dst_width.
template <int VEC>
__global__ __launch_bounds__(256) void chunk_rows_kernel(float *dst, long long ldd, int dst_width,
                                                        const float *src, long long lds,
                                                        int src_width, const ChunkTable t,
                                                        int relu) {
  using V = typename RV<VEC>::type;
  const int lane = threadIdx.x & 63;
  const long long wave = blockIdx.x * 4ll + (threadIdx.x >> 6);
  const long long n_waves = gridDim.x * 4ll;
  const long long total = t.first[t.n];
  // lane l keeps the first row of chunk l + 1: the chunk of a row is then the number of lanes
  // whose boundary it has reached -- one compare and a ballot instead of a walk through the
  // table (up to 32 dependent scalar loads per row: 31 us per launch over 100 partitions)
  const long long boundary = lane + 1 < t.n ? t.first[lane + 1] : LLONG_MAX;
  for (long long r0 = wave; r0 < total; r0 += n_waves * kRowsInFlight) {
    const float *s[kRowsInFlight];
    float *d[kRowsInFlight];
    bool ok[kRowsInFlight];
#pragma unroll
    for (int u = 0; u < kRowsInFlight; u++) {
      const long long r = r0 + u * n_waves;
      ok[u] = r < total;
      const int c = ok[u] ? __popcll(__ballot(r >= boundary)) : 0;
      const long long i = ok[u] ? r - t.first[c] : 0;
      s[u] = src + (t.src_row[c] + i) * lds;
      d[u] = dst + (t.dst_row[c] + i) * ldd;
    }
    for (int col = lane * VEC; col < dst_width; col += 64 * VEC) {
      V v[kRowsInFlight];
      // VEC == 4: rows start 16-byte aligned on both sides; widths need not be multiples of 4
      // (41- and 47-wide tables): the quad that straddles the end of a row goes element by element
      const bool whole_src = VEC == 1 || col + VEC <= src_width;
      const bool whole_dst = VEC == 1 || col + VEC <= dst_width;
#pragma unroll
      for (int u = 0; u < kRowsInFlight; u++) {
        v[u] = V(0.f);
        if (ok[u]) {
          if (whole_src) {
            v[u] = *reinterpret_cast<const V *>(s[u] + col);
          } else if constexpr (VEC == 4) {
            if (col < src_width) v[u].x = s[u][col];
            if (col + 1 < src_width) v[u].y = s[u][col + 1];
            if (col + 2 < src_width) v[u].z = s[u][col + 2];
          }
        }
        if (relu) v[u] = relu_of(v[u]);
      }
#pragma unroll
      for (int u = 0; u < kRowsInFlight; u++) {
        if (!ok[u]) continue;
        if (whole_dst) {
          *reinterpret_cast<V *>(d[u] + col) = v[u];
        } else if constexpr (VEC == 4) {
          d[u][col] = v[u].x;
          if (col + 1 < dst_width) d[u][col + 1] = v[u].y;
          if (col + 2 < dst_width) d[u][col + 2] = v[u].z;
        }
      }
    }
  }
}

static int launch_blocks(long long rows, int rows_per_wave) {
  const long long waves = (rows + rows_per_wave - 1) / rows_per_wave;
  return (int)std::max<long long>(1, std::min<long long>((waves + 3) / 4, 4096));
}

static int indexed_rows(bool scatter, float *table, int64_t ldt, int64_t table_rows,
                        const int64_t *idx, int64_t k_rows, int64_t width, float *packed,
                        int64_t ldp, hipStream_t stream) {
  INCAGG_REQUIRE(k_rows >= 0 && width >= 0 && table_rows >= 0 && width < INT_MAX, "bad size");
  if (k_rows == 0 || width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(table && idx && packed, "null pointer");
  INCAGG_REQUIRE(ldt >= width && ldp >= width, "leading dimension < width");
  int *err = nullptr;
  int rc = error_flag_address(&err);
  if (rc) return rc;
  const bool vec4 = width % 4 == 0 && aligned16(table) && aligned16(packed) && ldt % 4 == 0 &&
                    ldp % 4 == 0;
  const int blocks = launch_blocks(k_rows, kRowsInFlight);
  const long long *i64 = reinterpret_cast<const long long *>(idx);
#define INCAGG_LAUNCH_IDX(VEC, SC)                                                              \
  hipLaunchKernelGGL((indexed_rows_kernel<VEC, SC>), dim3(blocks), dim3(256), 0, stream, table, \
                     (long long)ldt, (long long)table_rows, i64, (long long)k_rows, (int)width, \
                     packed, (long long)ldp, err)
  if (vec4) {
    if (scatter) INCAGG_LAUNCH_IDX(4, true); else INCAGG_LAUNCH_IDX(4, false);
  } else {
    if (scatter) INCAGG_LAUNCH_IDX(1, true); else INCAGG_LAUNCH_IDX(1, false);
  }
#undef INCAGG_LAUNCH_IDX
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

}  // namespace incagg

using namespace incagg;

extern "C" int incagg_gather_rows_f32(const float *emb, int64_t lde, int64_t emb_rows,
                                      const int64_t *n_id, int64_t k_rows, int64_t width,
                                      float *out, int64_t ldo, void *stream) {
  return indexed_rows(false, const_cast<float *>(emb), lde, emb_rows, n_id, k_rows, width, out, ldo,
                      static_cast<hipStream_t>(stream));
}

extern "C" int incagg_scatter_rows_f32(float *emb, int64_t lde, int64_t emb_rows,
                                       const int64_t *n_id, int64_t k_rows, int64_t width,
                                       const float *x, int64_t ldx, void *stream) {
  return indexed_rows(true, emb, lde, emb_rows, n_id, k_rows, width, const_cast<float *>(x), ldx,
                      static_cast<hipStream_t>(stream));
}

extern "C" int incagg_copy_row_chunks_f32(float *dst, int64_t ldd, int64_t dst_rows,
                                          int64_t dst_width, const float *src, int64_t lds,
                                          int64_t src_rows, int64_t src_width,
                                          const int64_t *offset, const int64_t *count,
                                          int64_t n_chunks, int packed_is_src, int relu,
                                          void *stream) {
  INCAGG_REQUIRE(n_chunks >= 0 && dst_width >= 0 && src_width >= 0, "bad size");
  INCAGG_REQUIRE(src_width <= dst_width && dst_width < INT_MAX,
                 "source rows (%lld wide) do not fit destination rows (%lld wide)",
                 (long long)src_width, (long long)dst_width);
  if (n_chunks == 0 || dst_width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(dst && src && offset && count, "null pointer");
  INCAGG_REQUIRE(ldd >= dst_width && lds >= src_width, "leading dimension < width");
  // Validate every chunk before anything is enqueued (the reference asserts per
  // chunk, csrc/cuda/async_cuda.cu:77-78,148-149).
  const int64_t table_rows = packed_is_src ? dst_rows : src_rows;
  const int64_t packed_rows = packed_is_src ? src_rows : dst_rows;
  int64_t total = 0;
  for (int64_t c = 0; c < n_chunks; c++) {
    if (offset[c] < 0 || count[c] < 0 || offset[c] + count[c] > table_rows)
      return fail(INCAGG_E_INDEX, "Invalid index: chunk %lld = [%lld, +%lld) outside %lld rows",
                  (long long)c, (long long)offset[c], (long long)count[c], (long long)table_rows);
    total += count[c];
  }
  if (total > packed_rows)
    return fail(INCAGG_E_INDEX, "Invalid index: chunks hold %lld rows, packed side has %lld",
                (long long)total, (long long)packed_rows);
  const bool vec4 = aligned16(dst) && aligned16(src) && ldd % 4 == 0 && lds % 4 == 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int64_t packed_row = 0;
  for (int64_t c0 = 0; c0 < n_chunks; c0 += kMaxChunks) {
    ChunkTable t;
    t.n = (int)std::min<int64_t>(kMaxChunks, n_chunks - c0);
    t.first[0] = 0;
    for (int c = 0; c < t.n; c++) {
      t.dst_row[c] = packed_is_src ? offset[c0 + c] : packed_row;
      t.src_row[c] = packed_is_src ? packed_row : offset[c0 + c];
      t.first[c + 1] = t.first[c] + count[c0 + c];
      packed_row += count[c0 + c];
    }
    if (t.first[t.n] == 0) continue;
    const int blocks = launch_blocks(t.first[t.n], kRowsInFlight);
    if (vec4)
      hipLaunchKernelGGL((chunk_rows_kernel<4>), dim3(blocks), dim3(256), 0, s, dst, (long long)ldd,
                         (int)dst_width, src, (long long)lds, (int)src_width, t, relu);
    else
      hipLaunchKernelGGL((chunk_rows_kernel<1>), dim3(blocks), dim3(256), 0, s, dst, (long long)ldd,
                         (int)dst_width, src, (long long)lds, (int)src_width, t, relu);
    INCAGG_HIP_TRY(hipGetLastError());
  }
  return INCAGG_OK;
}
