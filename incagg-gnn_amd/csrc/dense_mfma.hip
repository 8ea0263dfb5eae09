// Dense update of a layer on the matrix cores of gfx950, in exact fp32:
//
//     Y[rows, N] = (accumulate ? Y : 0) + bias[N] + X[rows, K] @ W   (W: [K, N], or W^T: [N, K])
//     table[table_row0 + i, :N] = Y[i, :]  (optionally max(0, .))   -- the History push, fused
//
// i.e. what follows the aggregation in a GCNII / GCN / GraphSAGE layer (models/gcn2.py:257,306:
// conv.forward_after_propagate; models/gcn.py:242-244) together with the push of the layer's
// output into the next layer's table (history.py:41-65), and the input-gradient product of the
// backward pass (dX = dY @ W^T).  rows is 10^4..10^5, K and N are the hidden width (<= 256): the
// weight is tiny and shared by every workgroup, the row dimension gives the parallelism.
//
// One workgroup (8 waves) owns 64 rows x all N <= 256 columns; wave w computes rows
// [32 (w / 4), +32) x columns [64 (w % 4), +64) as two tiles of v_mfma_f32_32x32x2_f32 (32
// accumulator registers), two waves per SIMD taking turns on the matrix pipe.  K is walked in
// chunks of 32 through two LDS buffers: the loads of chunk c + 1 are in flight (in registers)
// while chunk c is multiplied.  The matrix pipe, not memory, bounds the kernel: a chunk is 40 KB
// of operands against 2048 cycles of MFMA per wave.
//
// Operand fragments.  The instruction wants A[i = lane % 32][k = lane / 32]: two values of k per
// step.  Sums over k may run in any order, so step i of a group of 8 k's pairs k = 8 j + i (lanes
// 0..31) with k = 8 j + 4 + i (lanes 32..63): a lane then reads its 4 steps' worth of A with ONE
// 16-byte LDS load from the row-major [row][k] image (row pitch 36 floats: the 16 lanes of a
// 128-bit read phase hit 64 distinct banks).  The weight is kept the same way, as a [column][k]
// image: copied as it is when W^T is given, transposed on its way into LDS otherwise.  The
// fragments of group j + 1 are fetched before the MFMAs of group j are issued.
//
// Numerics: an MFMA is a k-ordered chain of fp32 fused multiply-adds, one rounding per product.
#include "common.h"

#include <algorithm>
#include <climits>

namespace incagg {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kTileRows = 64;
constexpr int kGemmThreads = 512;
constexpr int kChunk = 32;              // k per LDS buffer
constexpr int kPitch = kChunk + 4;      // row pitch of both LDS images
constexpr int kMaxN = 256;

struct GemmArgs {
  const float *x;
  long long ldx;
  const float *w;      // TRANS_W: [N, K] (row pitch ldw), else [K, N]
  long long ldw;
  const float *bias;   // optional, [N]: added once (on top of the accumulated-into y, if any)
  float *y;
  long long ldy;
  float *table;        // optional: rows [table_row0, table_row0 + rows) receive Y (or max(0, Y))
  long long ldt;
  long long table_row0;
  int rows, k, n;
  int accumulate, relu_push;
  int skip_y;          // the result goes to the table only (y is read when accumulating, never written)
};

struct __align__(16) GemmLds {
  float a[2][kTileRows * kPitch];     // [row][k]
  float b[2][kMaxN * kPitch];         // [column][k]
};

template <bool TRANS_W>
__global__ __launch_bounds__(kGemmThreads) void rows_gemm_kernel(const GemmArgs g) {
  __shared__ GemmLds lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * kTileRows;
  const int half = lane >> 5, l32 = lane & 31;
  const int n_chunks = g.k / kChunk;
  const int wrow = 32 * (wave >> 2), wcol = 64 * (wave & 3);
  const bool wave_live = wcol < g.n;   // (N < 256: the other waves only help with loads)

  // ---- global -> register staging of one chunk.  Rows / columns past the end repeat the last
  // one (results there are never stored), so no load is conditional.
  // A: 64 rows x 32 k = 512 float4: thread t takes row t / 8, k 4 (t % 8) ..
  // W^T [N][K]: N x 32: thread t takes column t / 8 + 64 i (i < 4), k 4 (t % 8) ..
  // W   [K][N]: thread t takes column t % 256, k 16 (t / 256) + 4 i .. (four 4-byte loads each,
  //             coalesced along N), i < 4: the transposition happens in these registers
  f4 ra, rb[4];
  const float *xa = g.x + (long long)min(row0 + (tid >> 3), g.rows - 1) * g.ldx + 4 * (tid & 7);
  auto load_chunk = [&](int c) {
    const int k0 = c * kChunk;
    ra = *reinterpret_cast<const f4 *>(xa + k0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if constexpr (TRANS_W) {
        const int col = min((tid >> 3) + 64 * i, g.n - 1);
        rb[i] = *reinterpret_cast<const f4 *>(g.w + (long long)col * g.ldw + k0 + 4 * (tid & 7));
      } else {
        const int col = min(tid & 255, g.n - 1), kk = k0 + 16 * (tid >> 8) + 4 * i;
        const float *p = g.w + (long long)kk * g.ldw + col;
        rb[i].x = p[0]; rb[i].y = p[g.ldw]; rb[i].z = p[2 * g.ldw]; rb[i].w = p[3 * g.ldw];
      }
    }
  };
  auto store_chunk = [&](int buf) {
    *reinterpret_cast<f4 *>(&lds.a[buf][(tid >> 3) * kPitch + 4 * (tid & 7)]) = ra;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if constexpr (TRANS_W)
        *reinterpret_cast<f4 *>(&lds.b[buf][((tid >> 3) + 64 * i) * kPitch + 4 * (tid & 7)]) = rb[i];
      else
        *reinterpret_cast<f4 *>(&lds.b[buf][(tid & 255) * kPitch + 16 * (tid >> 8) + 4 * i]) = rb[i];
    }
  };

  load_chunk(0);

  // ---- accumulators: tile cb of the wave's 32 x 64 block; register r of lane l is row
  // 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32 of the tile
  f16v acc[2];
  auto tile_row = [&](int r) { return row0 + wrow + 8 * (r >> 2) + 4 * half + (r & 3); };
  const bool full = row0 + kTileRows <= g.rows && wcol + 64 <= g.n;   // (wave-uniform)
#pragma unroll
  for (int cb = 0; cb < 2; cb++) {
    const int col = min(wcol + 32 * cb + l32, g.n - 1);
    const float b = g.bias && wave_live ? g.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[cb][r] = b;
  }
  if (g.accumulate && wave_live) {
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = min(tile_row(r), g.rows - 1), col = min(wcol + 32 * cb + l32, g.n - 1);
        acc[cb][r] += g.y[(long long)row * g.ldy + col];
      }
  }

  store_chunk(0);
  if (n_chunks > 1) load_chunk(1);
  __syncthreads();

  // Schedule of chunk c (buffer c % 2), fragments always fetched one group ahead:
  //   groups 0, 1 | chunk c + 1: registers -> the other buffer, chunk c + 2: global -> registers
  //   | group 2 (fetches group 3) | barrier | first fragments of chunk c + 1 | group 3.
  // After the barrier nobody reads buffer c % 2 any more (group 3 is in registers) and the other
  // buffer is complete, so one barrier per chunk is enough and the matrix pipe has work on
  // both sides of it.
  const int frag_a = (wrow + l32) * kPitch + 4 * half, frag_b = (wcol + l32) * kPitch + 4 * half;
  f4 a[2], b0[2], b1[2];
  auto fetch = [&](int buf, int j, int slot) {
    a[slot] = *reinterpret_cast<const f4 *>(&lds.a[buf][frag_a + 8 * j]);
    b0[slot] = *reinterpret_cast<const f4 *>(&lds.b[buf][frag_b + 8 * j]);
    b1[slot] = *reinterpret_cast<const f4 *>(&lds.b[buf][frag_b + 32 * kPitch + 8 * j]);
  };
  auto multiply = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][i], b0[slot][i], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][i], b1[slot][i], acc[1], 0, 0, 0);
    }
  };
  static_assert(kChunk == 32, "the schedule below is written for four groups of 8 per chunk");
  if (wave_live) fetch(0, 0, 0);
  // (the two waves of a SIMD -- w and w + 4 -- move their share of the next chunk at different
  // points of the schedule, so that one of them always has MFMAs to issue)
  const bool early = wave < 4;
  for (int c = 0; c < n_chunks; c++) {
    const int buf = c & 1;
    const bool more = c + 1 < n_chunks;
    auto stage = [&]() {
      if (more) {
        store_chunk(buf ^ 1);
        if (c + 2 < n_chunks) load_chunk(c + 2);
      }
    };
    if (wave_live) {
      fetch(buf, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(0);
    }
    if (early) stage();
    if (wave_live) {
      fetch(buf, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      multiply(1);
    }
    if (!early) stage();
    if (wave_live) {
      fetch(buf, 3, 1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(0);
    }
    if (more) __syncthreads();
    if (wave_live) {
      if (more) fetch(buf ^ 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      multiply(1);
    }
  }

  if (!wave_live) return;
#pragma unroll
  for (int cb = 0; cb < 2; cb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = tile_row(r), col = wcol + 32 * cb + l32;
      if (!full && (row >= g.rows || col >= g.n)) continue;
      const float v = acc[cb][r];
      if (!g.skip_y) g.y[(long long)row * g.ldy + col] = v;
      if (g.table)
        g.table[(g.table_row0 + row) * g.ldt + col] = g.relu_push ? fmaxf(v, 0.f) : v;
    }
}

}  // namespace incagg

using namespace incagg;

extern "C" int incagg_rows_gemm_supported(int64_t k, int64_t n) {
  return k >= kChunk && k % kChunk == 0 && n >= 4 && n % 4 == 0 && n <= kMaxN;
}

extern "C" int incagg_rows_gemm_f32(const float *x, int64_t ldx, const float *w, int64_t ldw,
                                    int trans_w, const float *bias, int64_t rows, int64_t k,
                                    int64_t n, int accumulate, float *y, int64_t ldy, float *table,
                                    int64_t ldt, int64_t table_rows, int64_t table_row0,
                                    int relu_push, int table_only, void *stream) {
  INCAGG_REQUIRE(rows >= 0 && rows < INT_MAX - kTileRows, "rows %lld out of range", (long long)rows);
  INCAGG_REQUIRE(incagg_rows_gemm_supported(k, n),
                 "unsupported shape: k = %lld must be a positive multiple of %d, n = %lld a multiple "
                 "of 4 up to %d", (long long)k, kChunk, (long long)n, kMaxN);
  if (rows == 0) return INCAGG_OK;
  INCAGG_REQUIRE(x && w && y, "null pointer");
  INCAGG_REQUIRE(ldx >= k && ldy >= n && ldw >= (trans_w ? k : n), "leading dimension too small");
  INCAGG_REQUIRE(aligned16(x) && aligned16(w) && ldx % 4 == 0 && ldw % 4 == 0,
                 "x and w must have 16-byte aligned rows");
  INCAGG_REQUIRE((const void *)x != (const void *)y, "y must not alias x");
  if (table) {
    INCAGG_REQUIRE(ldt >= n && table_row0 >= 0 && table_row0 + rows <= table_rows,
                   "Invalid index: rows [%lld, +%lld) outside a table of %lld rows",
                   (long long)table_row0, (long long)rows, (long long)table_rows);
    INCAGG_REQUIRE((const void *)table != (const void *)y && (const void *)table != (const void *)x,
                   "the table must not alias x or y");
  }
  GemmArgs g{};
  g.x = x; g.ldx = ldx; g.w = w; g.ldw = ldw; g.bias = bias; g.y = y; g.ldy = ldy;
  g.table = table; g.ldt = ldt; g.table_row0 = table_row0;
  g.rows = (int)rows; g.k = (int)k; g.n = (int)n;
  g.accumulate = accumulate != 0; g.relu_push = relu_push != 0;
  g.skip_y = table_only != 0 && table != nullptr;
  const dim3 grid((unsigned)((rows + kTileRows - 1) / kTileRows)), block(kGemmThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (trans_w) hipLaunchKernelGGL(rows_gemm_kernel<true>, grid, block, 0, s, g);
  else hipLaunchKernelGGL(rows_gemm_kernel<false>, grid, block, 0, s, g);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}
