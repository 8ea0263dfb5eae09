// CSR neighbour aggregation for gfx950: one 64-lane wave per destination row, the
// lanes spanning the feature width so that every gathered source row is one fully
// coalesced 16-B-per-lane load.  Column ids / edge weights of a row are fetched
// 64 at a time with one coalesced load and broadcast with v_readlane, which makes
// every gather address "scalar base + lane offset".  Rows longer than
// kHubThreshold are cut into kHubSlice-edge slices that other waves (on other
// CUs) pick up in a second launch and combine with fp32 atomics: one wave keeps only
// 8 row loads in flight, so a 10^3..10^4-edge row (power-law graphs) would otherwise
// run for hundreds of dependent memory round trips while the rest of the chip idles.
// Measured on an ogbn-products-sized partition batch (MI355X): threshold/slice
// 1024/256 -> 164 us per launch pair, 512/256 -> 123 us, 128/64 -> 93 us, 64/32 -> 106 us.
//
// HBM roofline: the kernel moves E*(4+4) index/weight bytes, (B+1)*4 row pointers,
// one read of every distinct source row (width*4 B each) and one write of every
// destination row; flops are 2*E*width, i.e. ~0.5 flop per gathered byte.
#include "common.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <utility>

namespace incagg {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
// Rows with more than hub_threshold edges are cut into hub_slice-edge slices.  Compile-time
// defaults; INCAGG_HUB_THRESHOLD / INCAGG_HUB_SLICE override them for tuning runs.
constexpr int kHubThreshold = 128;
constexpr int kHubSlice = 64;
constexpr int kQueueCap = 1 << 20;  // slices per call that the queue can hold

template <int VEC> struct VT;
template <> struct VT<1> { using type = float; };
template <> struct VT<4> { using type = f4; };

template <int VEC>
__device__ __forceinline__ typename VT<VEC>::type ldv(const float *p) {
  return *reinterpret_cast<const typename VT<VEC>::type *>(p);
}
template <int VEC>
__device__ __forceinline__ void stv(float *p, typename VT<VEC>::type v) {
  *reinterpret_cast<typename VT<VEC>::type *>(p) = v;
}

struct SpmmArgs {
  const int *rowptr;
  const int *col;
  const float *val;
  const float *x;
  long long ldx;
  const float *x2;  // source rows [split, n_cols) live here (row c at x2 + (c - split) * ldx2)
  long long ldx2;
  int split;
  const float *m_in;
  long long ldin;
  const float *m_ag;
  long long ldag;
  const float *add;
  long long ldadd;
  float *y;
  long long ldy;
  int n_rows;
  int n_cols;
  int width;
  int mean;
  float scale;
  float add_scale;
  int hub_threshold;
  int hub_slice;
  int2 *queue;
  int *q_count;
  int *q_count_next;
  int q_cap;
  int *err;
  const int *row_order;  // grouped kernel: processing order of the rows (nullptr = identity)
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// U edges starting at local edge j of the current 64-edge window: all gathers are
// issued before the first FMA so that U (or 2U) row loads are in flight per wave.
template <int VEC, int CH, bool INCAGG, int U>
__device__ __forceinline__ void edge_step(const SpmmArgs &a, int my_c, float my_v, int j,
                                          const int (&off)[CH],
                                          typename VT<VEC>::type (&acc)[CH]) {
  using V = typename VT<VEC>::type;
  V xv[U][CH];
  V mv[U][CH];
  float w[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int c = __builtin_amdgcn_readlane(my_c, j + u);
    w[u] = readlane_f(my_v, j + u);
    const float *xr = c < a.split ? a.x + (size_t)c * a.ldx : a.x2 + (size_t)(c - a.split) * a.ldx2;
#pragma unroll
    for (int ch = 0; ch < CH; ch++) xv[u][ch] = ldv<VEC>(xr + off[ch]);
    if (INCAGG) {
      const float *mr = a.m_in + (size_t)c * a.ldin;
#pragma unroll
      for (int ch = 0; ch < CH; ch++) mv[u][ch] = ldv<VEC>(mr + off[ch]);
    }
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
      if (INCAGG)
        acc[ch] += w[u] * (xv[u][ch] - mv[u][ch]);
      else
        acc[ch] += w[u] * xv[u][ch];
    }
  }
}

// acc += sum over edges [rs, re) of one row.  rs/re are wave-uniform.
template <int VEC, int CH, bool INCAGG>
__device__ __forceinline__ void accumulate_edges(const SpmmArgs &a, int rs, int re, int lane,
                                                 const int (&off)[CH],
                                                 typename VT<VEC>::type (&acc)[CH]) {
  constexpr int UMAX = INCAGG ? 4 : 8;
  for (int e0 = rs; e0 < re; e0 += kWave) {
    const int cnt = min(kWave, re - e0);
    int my_c = 0;
    float my_v = 0.f;
    if (lane < cnt) {
      my_c = a.col[e0 + lane];
      my_v = a.val ? a.val[e0 + lane] : 1.0f;
      if ((unsigned)my_c >= (unsigned)a.n_cols) {  // never fault on a bad id
        *a.err = 1;
        my_c = 0;
        my_v = 0.f;
      }
    }
    int j = 0;
    for (; j + UMAX <= cnt; j += UMAX) edge_step<VEC, CH, INCAGG, UMAX>(a, my_c, my_v, j, off, acc);
    if (UMAX > 4 && j + 4 <= cnt) {
      edge_step<VEC, CH, INCAGG, 4>(a, my_c, my_v, j, off, acc);
      j += 4;
    }
    if (j + 2 <= cnt) {
      edge_step<VEC, CH, INCAGG, 2>(a, my_c, my_v, j, off, acc);
      j += 2;
    }
    if (j < cnt) edge_step<VEC, CH, INCAGG, 1>(a, my_c, my_v, j, off, acc);
  }
}

// Lanes whose columns lie beyond `width` read column 0 instead (a valid address,
// result discarded), which keeps the gather loop free of per-lane branches.
template <int VEC, int CH>
__device__ __forceinline__ void column_offsets(int lane, int width, int (&off)[CH],
                                               bool (&live)[CH]) {
#pragma unroll
  for (int ch = 0; ch < CH; ch++) {
    const int c = lane * VEC + ch * kWave * VEC;
    live[ch] = c < width;
    off[ch] = live[ch] ? c : 0;
  }
}

template <int VEC, int CH, bool INCAGG>
__device__ __forceinline__ void store_row(const SpmmArgs &a, int row, float s,
                                          const int (&off)[CH], const bool (&live)[CH],
                                          const typename VT<VEC>::type (&acc)[CH]) {
  using V = typename VT<VEC>::type;
#pragma unroll
  for (int ch = 0; ch < CH; ch++) {
    if (!live[ch]) continue;
    V out = acc[ch] * s;
    if (INCAGG) out += a.scale * ldv<VEC>(a.m_ag + (size_t)row * a.ldag + off[ch]);
    if (a.add) out += a.add_scale * ldv<VEC>(a.add + (size_t)row * a.ldadd + off[ch]);
    stv<VEC>(a.y + (size_t)row * a.ldy + off[ch], out);
  }
}

template <int VEC, int CH, bool INCAGG>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_rows_kernel(const SpmmArgs a) {
  using V = typename VT<VEC>::type;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  // Blocks b and b + 8 share an XCD (and its L2): give every XCD one contiguous
  // range of rows, since neighbouring rows of a partition share source rows.
  const int nb = gridDim.x, b = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = b & 7;
  const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int row = lb * kWavesPerBlock + wave;
  if (row >= a.n_rows) return;

  const int rs = rfl(a.rowptr[row]);
  const int re = rfl(a.rowptr[row + 1]);
  const int deg = re - rs;

  int off[CH];
  bool live[CH];
  column_offsets<VEC, CH>(lane, a.width, off, live);
  V acc[CH];
#pragma unroll
  for (int ch = 0; ch < CH; ch++) acc[ch] = V(0.f);

  if (deg > a.hub_threshold) {
    const int ns = (deg + a.hub_slice - 1) / a.hub_slice;
    int base = 0;
    if (lane == 0) base = atomicAdd(a.q_count, ns);
    base = rfl(base);
    const bool fits = base + ns <= a.q_cap;
    for (int k = lane; k < ns && base + k < a.q_cap; k += kWave)
      a.queue[base + k] = fits ? make_int2(row, k) : make_int2(-1, 0);
    if (fits) {  // slices add onto this base value in the second launch
      store_row<VEC, CH, INCAGG>(a, row, 0.f, off, live, acc);
      return;
    }
  }
  accumulate_edges<VEC, CH, INCAGG>(a, rs, re, lane, off, acc);
  const float s = a.scale * (a.mean ? 1.0f / (float)max(deg, 1) : 1.0f);
  store_row<VEC, CH, INCAGG>(a, row, s, off, live, acc);
}

template <int VEC, int CH, bool INCAGG>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_hub_kernel(const SpmmArgs a) {
  using V = typename VT<VEC>::type;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int n = min(rfl(*a.q_count), a.q_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.q_count_next = 0;  // for the next call
  int off[CH];
  bool live[CH];
  column_offsets<VEC, CH>(lane, a.width, off, live);
  const int nw = gridDim.x * kWavesPerBlock;
  for (int sidx = blockIdx.x * kWavesPerBlock + wave; sidx < n; sidx += nw) {
    const int2 ent = a.queue[sidx];
    const int row = rfl(ent.x);
    if (row < 0) continue;
    const int k = rfl(ent.y);
    // an entry is followed only if it names a row and a slice of THIS call's matrix: the
    // queue outlives the call, and an entry that is not ours must be reported, not read
    if (row >= a.n_rows || k < 0) {
      *a.err = 1;
      continue;
    }
    const int rs0 = rfl(a.rowptr[row]);
    const int re0 = rfl(a.rowptr[row + 1]);
    if ((long long)k * a.hub_slice >= (long long)(re0 - rs0)) {
      *a.err = 1;
      continue;
    }
    const int rs = rs0 + k * a.hub_slice;
    const int re = min(re0, rs + a.hub_slice);
    V acc[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ch++) acc[ch] = V(0.f);
    accumulate_edges<VEC, CH, INCAGG>(a, rs, re, lane, off, acc);
    const float s = a.scale * (a.mean ? 1.0f / (float)max(re0 - rs0, 1) : 1.0f);
    float *yr = a.y + (size_t)row * a.ldy;
    if constexpr (VEC == 1) {
#pragma unroll
      for (int ch = 0; ch < CH; ch++)
        if (live[ch]) atomicAdd(yr + off[ch], acc[ch] * s);
    } else {
      // Regroup so that each atomic wave-instruction covers 256 contiguous bytes:
      // lane l adds column 64 t + l, held by lane 16 t + l / 4 in component l % 4.
#pragma unroll
      for (int ch = 0; ch < CH; ch++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int src = 16 * t + (lane >> 2);
          const float v0 = __shfl(acc[ch].x, src), v1 = __shfl(acc[ch].y, src);
          const float v2 = __shfl(acc[ch].z, src), v3 = __shfl(acc[ch].w, src);
          const int sel = lane & 3;
          const float v = sel == 0 ? v0 : sel == 1 ? v1 : sel == 2 ? v2 : v3;
          const int c = ch * kWave * VEC + kWave * t + lane;
          if (c < a.width) atomicAdd(yr + c, v * s);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- feature-sliced variant
// The 8 XCDs of an MI355X have private L2s.  With whole rows per wave, a source row that
// is referenced from destination rows running on different XCDs is fetched into every one
// of those L2s (measured: fabric traffic ~4x the distinct source bytes at 7 references per
// source row).  Here the FEATURE dimension is cut into 128-byte slices and slice s is
// processed only by workgroups with blockIdx % n_slices == s, i.e. (with round-robin
// dispatch) by one XCD: every byte of the source matrix then has exactly one home L2 and the
// per-XCD working set shrinks by the number of slices.  Placement is a speed matter only.
//
// A wave works on one destination row and one slice: 8 lanes cover the 128 bytes of a
// gathered piece (one full cache line per edge), the 8 lane groups take 8 different
// edges per load instruction, and the groups are summed with cross-lane shuffles at the end.
constexpr int kSliceFloats = 32;

__device__ __forceinline__ f4 shfl_xor4(f4 v, int m) {
  f4 r;
  r.x = __shfl_xor(v.x, m); r.y = __shfl_xor(v.y, m);
  r.z = __shfl_xor(v.z, m); r.w = __shfl_xor(v.w, m);
  return r;
}

// Sum over edges [rs, re) of val[e] * (x[col[e], piece] (- m_in[col[e], piece])), for the
// lane's 16-byte part of the piece; the result still has to be summed over the 8 lane
// groups.  Indices of the next 8*R edges are loaded while the current gathers are in flight.
// Column ids / weights of the 64-edge window starting at e0 (lane l: edge e0 + l), one
// coalesced load each; edges at or beyond `re` get weight 0.
__device__ __forceinline__ void load_window(const SpmmArgs &a, int e0, int re, int lane, int &c,
                                            float &w) {
  // Lane 8 g + j holds window edge 8 j + g: round j of lane group g (see gather_window).
  // The 64 addresses still fall into the same two cache lines, so the load stays coalesced.
  // Lanes beyond the row's end re-read its last edge with weight 0, so every lane holds a
  // real neighbour and the gather loop needs no per-lane guards.  (re > e0 always.)
  const int slot = e0 + 8 * (lane & 7) + (lane >> 3);
  const int e = min(slot, re - 1);
  c = a.col[e];
  w = a.val ? a.val[e] : 1.0f;
  if (slot >= re) w = 0.f;
  if ((unsigned)c >= (unsigned)a.n_cols) {  // never fault on a bad id
    *a.err = 1;
    c = 0;
    w = 0.f;
  }
}

// Broadcast of lane 8 * (l / 8) + R inside every group of 8 lanes: a ds_swizzle in
// bit-mask mode (and = 0b11000, or = R), which needs no address register.
template <int R>
__device__ __forceinline__ int group_bcast(int v) {
  return __builtin_amdgcn_ds_swizzle(v, (R << 5) | 0x18);
}

// Rounds [R0, R0 + N) of a window: in round r lane group g (8 lanes = the 128 bytes of one
// piece) gathers window edge 8 r + g, so one load instruction fetches 8 different source
// rows; all N gathers are issued before the first FMA.  The indices reach the group through
// the cross-lane network: the vector-memory pipe only sees the row gathers.
// WIDE: row offsets need 64 bits or the source is split in two matrices; otherwise a
// 32-bit byte offset per lane on top of a scalar base is enough.
template <bool INCAGG, bool WIDE, int R0, int N>
__device__ __forceinline__ void gather_rounds(const SpmmArgs &a, int my_c, float my_w, int coloff,
                                              f4 &acc) {
  int c[N];
  float w[N];
  f4 xv[N], mv[N];
#pragma unroll
  for (int r = 0; r < N; r++) {
    // (template recursion keeps the swizzle pattern an immediate)
    c[r] = r == 0 ? group_bcast<R0>(my_c) : r == 1 ? group_bcast<R0 + 1>(my_c)
         : r == 2 ? group_bcast<R0 + 2>(my_c) : r == 3 ? group_bcast<R0 + 3>(my_c)
         : r == 4 ? group_bcast<R0 + 4>(my_c) : r == 5 ? group_bcast<R0 + 5>(my_c)
         : r == 6 ? group_bcast<R0 + 6>(my_c) : group_bcast<R0 + 7>(my_c);
    const int wi = __float_as_int(my_w);
    w[r] = __int_as_float(
        r == 0 ? group_bcast<R0>(wi) : r == 1 ? group_bcast<R0 + 1>(wi)
        : r == 2 ? group_bcast<R0 + 2>(wi) : r == 3 ? group_bcast<R0 + 3>(wi)
        : r == 4 ? group_bcast<R0 + 4>(wi) : r == 5 ? group_bcast<R0 + 5>(wi)
        : r == 6 ? group_bcast<R0 + 6>(wi) : group_bcast<R0 + 7>(wi));
  }
#pragma unroll
  for (int r = 0; r < N; r++) {
    if constexpr (WIDE) {
      const float *xr = c[r] < a.split ? a.x + (size_t)c[r] * a.ldx
                                       : a.x2 + (size_t)(c[r] - a.split) * a.ldx2;
      xv[r] = ldv<4>(xr + coloff);
      if (INCAGG) mv[r] = ldv<4>(a.m_in + (size_t)c[r] * a.ldin + coloff);
    } else {
      // byte offsets computed in 32 bits: "scalar base + 32-bit lane offset" addressing
      const unsigned off = ((unsigned)c[r] * (unsigned)a.ldx + (unsigned)coloff) * 4u;
      xv[r] = *reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(a.x) + off);
      if (INCAGG) {
        const unsigned moff = ((unsigned)c[r] * (unsigned)a.ldin + (unsigned)coloff) * 4u;
        mv[r] = *reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(a.m_in) + moff);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < N; r++) acc += w[r] * (INCAGG ? xv[r] - mv[r] : xv[r]);
}

// acc += contributions of the window whose indices sit in (my_c, my_w); cnt = real edges
// in it (wave-uniform).  Short rows only pay for the rounds they fill: 1, 2, 4 or 8.
template <bool INCAGG, bool WIDE>
__device__ __forceinline__ void gather_window(const SpmmArgs &a, int my_c, float my_w, int cnt,
                                              int coloff, f4 &acc) {
  if (cnt > 32) {
    if constexpr (INCAGG) {  // two operands per edge: 4 rounds = 8 gathers in flight
      gather_rounds<INCAGG, WIDE, 0, 4>(a, my_c, my_w, coloff, acc);
      gather_rounds<INCAGG, WIDE, 4, 4>(a, my_c, my_w, coloff, acc);
    } else {
      gather_rounds<INCAGG, WIDE, 0, 8>(a, my_c, my_w, coloff, acc);
    }
  } else if (cnt > 16) {
    gather_rounds<INCAGG, WIDE, 0, 4>(a, my_c, my_w, coloff, acc);
  } else if (cnt > 8) {
    gather_rounds<INCAGG, WIDE, 0, 2>(a, my_c, my_w, coloff, acc);
  } else {
    gather_rounds<INCAGG, WIDE, 0, 1>(a, my_c, my_w, coloff, acc);
  }
}

__device__ __forceinline__ f4 sum_lane_groups(f4 acc) {
  acc += shfl_xor4(acc, 8);
  acc += shfl_xor4(acc, 16);
  acc += shfl_xor4(acc, 32);
  return acc;
}

template <bool INCAGG>
__device__ __forceinline__ void sliced_store(const SpmmArgs &a, int row, f4 acc, float s, int g,
                                             bool live, int coloff) {
  if (g == 0 && live) {
    f4 out = acc * s;
    if (INCAGG) out += a.scale * ldv<4>(a.m_ag + (size_t)row * a.ldag + coloff);
    if (a.add) out += a.add_scale * ldv<4>(a.add + (size_t)row * a.ldadd + coloff);
    stv<4>(a.y + (size_t)row * a.ldy + coloff, out);
  }
}

// Row pointers of the wave's rows live in lanes 0..n (lane i: rowptr[r0 + i]).
__device__ __forceinline__ int row_start(int my_rp, int i) { return __builtin_amdgcn_readlane(my_rp, i); }

template <bool INCAGG, bool WIDE>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_sliced_kernel(const SpmmArgs a,
                                                                            int n_slices,
                                                                            int rows_per_wave) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 3, t = lane & 7;
  const int slice = blockIdx.x % n_slices;
  const int r0 = ((blockIdx.x / n_slices) * kWavesPerBlock + wave) * rows_per_wave;
  if (r0 >= a.n_rows) return;
  const int nr = min(rows_per_wave, a.n_rows - r0);
  const int col0 = slice * kSliceFloats + t * 4;
  const bool live = col0 < a.width;
  const int coloff = live ? col0 : 0;
  const int my_rp = lane <= nr ? a.rowptr[r0 + lane] : 0;

  // A wave walks its rows window by window (64 edges each) and always has the NEXT
  // window's indices in flight while it gathers the current one -- also across the end
  // of a row, which is what hides the rowptr -> indices -> gather dependency chain of
  // the (many) short rows.  Rows without work for this loop (no edges, or so long that
  // they are handed to the slice queue) are finished on the spot when the walk skips them.
  auto finish_skipped = [&](int i) {
    const int rs = row_start(my_rp, i), re = row_start(my_rp, i + 1);
    const int deg = a.n_cols > 0 ? re - rs : 0;
    if (deg > a.hub_threshold && slice == 0) {  // one wave per row enqueues its slices
      const int ns = (deg + a.hub_slice - 1) / a.hub_slice;
      int base = 0;
      if (lane == 0) base = atomicAdd(a.q_count, ns);
      base = rfl(base);
      if (base + ns <= a.q_cap) {
        for (int k = lane; k < ns; k += kWave) a.queue[base + k] = make_int2(r0 + i, k);
      } else {
        for (int k = lane; k < ns && base + k < a.q_cap; k += kWave)
          a.queue[base + k] = make_int2(-1, 0);
        if (lane == 0) *a.err = 1;  // > 2^20 slices in one call: result incomplete, reported
      }
    }
    sliced_store<INCAGG>(a, r0 + i, f4(0.f), 0.f, g, live, coloff);  // base value only
  };
  auto has_work = [&](int i) {
    const int deg = row_start(my_rp, i + 1) - row_start(my_rp, i);
    return a.n_cols > 0 && deg > 0 && deg <= a.hub_threshold;
  };
  // first row with work
  int i = 0;
  while (i < nr && !has_work(i)) finish_skipped(i++);
  if (i >= nr) return;
  int e0 = row_start(my_rp, i);
  int my_c, next_c = 0;
  float my_w, next_w = 0.f;
  load_window(a, e0, row_start(my_rp, i + 1), lane, my_c, my_w);
  f4 acc = f4(0.f);
  while (i < nr) {
    const int rs = row_start(my_rp, i), re = row_start(my_rp, i + 1);
    // where the walk goes next
    int ni = i, ne0 = e0 + kWave;
    if (ne0 >= re) {
      ni = i + 1;
      while (ni < nr && !has_work(ni)) finish_skipped(ni++);
      ne0 = ni < nr ? row_start(my_rp, ni) : 0;
    }
    if (ni < nr) load_window(a, ne0, row_start(my_rp, ni + 1), lane, next_c, next_w);
    gather_window<INCAGG, WIDE>(a, my_c, my_w, min(kWave, re - e0), coloff, acc);
    if (ni != i) {  // row i is complete
      const float s = a.scale * (a.mean ? 1.0f / (float)(re - rs) : 1.0f);
      sliced_store<INCAGG>(a, r0 + i, sum_lane_groups(acc), s, g, live, coloff);
      acc = f4(0.f);
    }
    my_c = next_c;
    my_w = next_w;
    i = ni;
    e0 = ne0;
  }
}

template <bool INCAGG, bool WIDE>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_sliced_hub_kernel(const SpmmArgs a,
                                                                                int n_slices) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int t = lane & 7;
  const int n = min(rfl(*a.q_count), a.q_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.q_count_next = 0;  // for the next call
  const int slice = blockIdx.x % n_slices;
  const int col0 = slice * kSliceFloats + t * 4;
  const bool live = col0 < a.width;
  const int coloff = live ? col0 : 0;
  const int stride = (gridDim.x / n_slices) * kWavesPerBlock;
  for (int sidx = (blockIdx.x / n_slices) * kWavesPerBlock + wave; sidx < n; sidx += stride) {
    const int2 ent = a.queue[sidx];
    const int row = rfl(ent.x);
    if (row < 0) continue;
    const int k = rfl(ent.y);
    // an entry is followed only if it names a row and a slice of THIS call's matrix: the
    // queue outlives the call, and an entry that is not ours must be reported, not read
    if (row >= a.n_rows || k < 0) {
      *a.err = 1;
      continue;
    }
    const int rs0 = rfl(a.rowptr[row]);
    const int re0 = rfl(a.rowptr[row + 1]);
    if ((long long)k * a.hub_slice >= (long long)(re0 - rs0)) {
      *a.err = 1;
      continue;
    }
    const int rs = rs0 + k * a.hub_slice;
    const int re = min(re0, rs + a.hub_slice);
    f4 acc = f4(0.f);
    int my_c, next_c = 0;
    float my_w, next_w = 0.f;
    load_window(a, rs, re, lane, my_c, my_w);
    for (int e0 = rs; e0 < re; e0 += kWave) {
      if (e0 + kWave < re) load_window(a, e0 + kWave, re, lane, next_c, next_w);
      gather_window<INCAGG, WIDE>(a, my_c, my_w, min(kWave, re - e0), coloff, acc);
      my_c = next_c;
      my_w = next_w;
    }
    acc = sum_lane_groups(acc);
    const float s = a.scale * (a.mean ? 1.0f / (float)max(re0 - rs0, 1) : 1.0f);
    // lanes 0..31 each add one float: one 128-byte segment per atomic wave-instruction
    const int src = lane >> 2, sel = lane & 3;
    const float v0 = __shfl(acc.x, src), v1 = __shfl(acc.y, src);
    const float v2 = __shfl(acc.z, src), v3 = __shfl(acc.w, src);
    const float v = sel == 0 ? v0 : sel == 1 ? v1 : sel == 2 ? v2 : v3;
    const int c = slice * kSliceFloats + lane;
    if (lane < kSliceFloats && c < a.width) atomicAdd(a.y + (size_t)row * a.ldy + c, v * s);
  }
}

// ---------------------------------------------------------------- grouped feature-sliced variant
// Same slice-per-XCD placement, but every group of 8 lanes owns its own destination row: a
// wave handles 8 rows x one 128-byte slice, each group walks its row's edges 8 at a time
// (one coalesced 32-byte index load per group, broadcast inside the group with ds_swizzle)
// and keeps its own accumulator, so there is no cross-lane reduction and no per-row walk
// logic -- about 1.5x fewer instructions per gathered byte than one row per wave.  The trip
// count of a wave is set by its longest row; `row_order` (optional) lists the rows so that 8
// consecutive entries have similar degrees (built once per batch on the host side).
__device__ __forceinline__ int group_max8(int v) {  // max over the 8 lane groups of a wave
  v = max(v, __shfl_xor(v, 8));
  v = max(v, __shfl_xor(v, 16));
  v = max(v, __shfl_xor(v, 32));
  return v;
}

template <bool INCAGG, bool WIDE>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_grouped_kernel(
    const SpmmArgs a, int n_slices) {
  const int *__restrict__ row_order = a.row_order;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 3, t = lane & 7;
  const int slice = blockIdx.x % n_slices;
  const int p = ((blockIdx.x / n_slices) * kWavesPerBlock + wave) * 8 + g;
  const int col0 = slice * kSliceFloats + t * 4;
  const bool live = col0 < a.width;
  const int coloff = live ? col0 : 0;
  const bool valid = p < a.n_rows;
  int row = 0, rs = 0, re = 0;
  if (valid) {
    row = row_order ? row_order[p] : p;
    if ((unsigned)row >= (unsigned)a.n_rows) {  // bad permutation entry: reported, not followed
      *a.err = 1;
      row = p;
    }
    rs = a.rowptr[row];
    re = a.rowptr[row + 1];
  }
  const int full_deg = a.n_cols > 0 ? re - rs : 0;
  int deg = full_deg;
  if (deg > a.hub_threshold) {
    if (slice == 0) {  // one lane group per row enqueues its slices
      const int ns = (deg + a.hub_slice - 1) / a.hub_slice;
      int base = 0;
      if (t == 0) base = atomicAdd(a.q_count, ns);
      base = group_bcast<0>(base);
      if (base + ns <= a.q_cap) {
        for (int k = t; k < ns; k += 8) a.queue[base + k] = make_int2(row, k);
      } else {
        for (int k = t; k < ns && base + k < a.q_cap; k += 8) a.queue[base + k] = make_int2(-1, 0);
        *a.err = 1;  // > 2^20 slices in one call: result incomplete, reported
      }
    }
    deg = 0;  // only the base value is stored here
  }
  re = rs + deg;
  const int maxdeg = rfl(group_max8(deg));

  // Lane t of a group holds edge e + t; lanes past the row's end re-read its last edge
  // with weight 0, so the gather rounds need no per-lane guards.
  auto load_idx = [&](int e0, int &c, float &w) {
    if (e0 < re) {
      const int slot = e0 + t;
      const int e = min(slot, re - 1);
      c = a.col[e];
      w = a.val ? a.val[e] : 1.0f;
      if (slot >= re) w = 0.f;
      if ((unsigned)c >= (unsigned)a.n_cols) {  // never fault on a bad id
        *a.err = 1;
        c = 0;
        w = 0.f;
      }
    }
  };
  f4 acc = f4(0.f);
  int e = rs;
  int my_c = 0, next_c = 0;
  float my_w = 0.f, next_w = 0.f;
  load_idx(e, my_c, my_w);
  for (int done = 0; done < maxdeg; done += 8) {
    if (done + 8 < maxdeg) load_idx(e + 8, next_c, next_w);
    const int rem = maxdeg - done;  // wave-uniform: rounds that some group still needs
    if (e < re) {
      if (rem > 4) {
        if constexpr (INCAGG) {  // two operands per edge: 4 rounds = 8 gathers in flight
          gather_rounds<INCAGG, WIDE, 0, 4>(a, my_c, my_w, coloff, acc);
          gather_rounds<INCAGG, WIDE, 4, 4>(a, my_c, my_w, coloff, acc);
        } else {
          gather_rounds<INCAGG, WIDE, 0, 8>(a, my_c, my_w, coloff, acc);
        }
      } else if (rem > 2) {
        gather_rounds<INCAGG, WIDE, 0, 4>(a, my_c, my_w, coloff, acc);
      } else if (rem > 1) {
        gather_rounds<INCAGG, WIDE, 0, 2>(a, my_c, my_w, coloff, acc);
      } else {
        gather_rounds<INCAGG, WIDE, 0, 1>(a, my_c, my_w, coloff, acc);
      }
    }
    my_c = next_c;
    my_w = next_w;
    e += 8;
  }
  if (valid && live) {
    const float s = a.scale * (a.mean ? 1.0f / (float)max(full_deg, 1) : 1.0f);
    f4 out = acc * s;
    if (INCAGG) out += a.scale * ldv<4>(a.m_ag + (size_t)row * a.ldag + coloff);
    if (a.add) out += a.add_scale * ldv<4>(a.add + (size_t)row * a.ldadd + coloff);
    stv<4>(a.y + (size_t)row * a.ldy + coloff, out);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void rows_sub_kernel(const float *a, long long lda,
                                                      const float *b, long long ldb,
                                                      long long n_rows, int width, float *d,
                                                      long long ldd) {
  const int per_row = width / VEC;
  const long long total = n_rows * per_row;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / per_row;
    const int c = (int)(i - r * per_row) * VEC;
    stv<VEC>(d + r * ldd + c, ldv<VEC>(a + r * lda + c) - ldv<VEC>(b + r * ldb + c));
  }
}

// ---------------------------------------------------------------- host side
struct HubQueue {
  int2 *entries = nullptr;
  // Two pairs of counter words, each used alternately (the idle word of a pair is zero: every
  // slice kernel clears the word of the next call).  Words 0 / 1 serve eager calls.  Words 2 / 3
  // serve calls that are being recorded into a HIP graph: a replay runs none of this host code,
  // so the alternation cannot carry over from one recording to the next -- the first call of
  // every recording therefore clears both words with one recorded memset and starts at word 2.
  int *counters = nullptr;
  int parity = 0;
  int capture_parity = 0;
  unsigned long long capture_id = ~0ull;
};

static std::mutex s_queue_mutex;
static std::map<std::pair<int, hipStream_t>, HubQueue> s_queues;

// The slice queue of (device, stream); created on first use.  Calls on one stream
// are ordered, so one queue per stream can never be used by two launches at once.
static int hub_queue_for(hipStream_t stream, HubQueue **out) {
  int dev = 0;
  INCAGG_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(s_queue_mutex);
  HubQueue &q = s_queues[{dev, stream}];
  if (!q.entries) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    if (cap != hipStreamCaptureStatusNone) {
      s_queues.erase({dev, stream});
      return fail(INCAGG_E_INVALID,
                  "the slice queue of this stream does not exist yet and cannot be allocated while "
                  "the stream is being captured: run one aggregation on the stream first");
    }
    INCAGG_HIP_TRY(hipMalloc(&q.entries, sizeof(int2) * (size_t)kQueueCap));
    hipError_t e = hipMalloc(&q.counters, 4 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(q.counters, 0, 4 * sizeof(int));
    if (e != hipSuccess) {
      (void)hipFree(q.entries);
      q.entries = nullptr;
      return fail(INCAGG_E_HIP, "hub queue allocation failed: %s", hipGetErrorString(e));
    }
  }
  *out = &q;
  return INCAGG_OK;
}

// Points the launch at the queue's counters.  `advance` is set when the caller has to flip the
// parity after its two launches (eager call); calls that are being captured alternate a pair
// of words of their own, which one recorded memset clears at the start of every recording.
static int bind_queue(SpmmArgs &a, HubQueue *q, hipStream_t stream, bool *advance) {
  a.queue = q->entries;
  a.q_cap = kQueueCap;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  INCAGG_HIP_TRY(hipStreamGetCaptureInfo(stream, &cap, &id));
  if (cap == hipStreamCaptureStatusNone) {
    a.q_count = q->counters + q->parity;
    a.q_count_next = q->counters + (q->parity ^ 1);
    *advance = true;
  } else {
    if (id != q->capture_id) {  // first aggregation of this recording
      INCAGG_HIP_TRY(hipMemsetAsync(q->counters + 2, 0, 2 * sizeof(int), stream));
      q->capture_id = id;
      q->capture_parity = 0;
    }
    a.q_count = q->counters + 2 + q->capture_parity;
    a.q_count_next = q->counters + 2 + (q->capture_parity ^ 1);
    q->capture_parity ^= 1;
    *advance = false;
  }
  return INCAGG_OK;
}

static int env_int(const char *name, int fallback) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : fallback;
}

// Launch tuning.  Defaults are the measured optima (see launch_spmm / launch_sliced);
// INCAGG_* environment variables seed them, incagg_set_tuning() changes them at run time
// (used by the A/B benchmarks and by the tests to force either kernel variant).
struct Tuning {
  int spmm_sliced = env_int("INCAGG_SPMM_SLICED", -1);          // -1 by shape, 0 rows, 1 sliced
  int hub_threshold = env_int("INCAGG_HUB_THRESHOLD", kHubThreshold);
  int hub_slice = env_int("INCAGG_HUB_SLICE", kHubSlice);
  int hub_blocks = env_int("INCAGG_HUB_BLOCKS", 2048);
  int sliced_hub_threshold = env_int("INCAGG_SLICED_HUB_THRESHOLD", 256);
  int sliced_hub_slice = env_int("INCAGG_SLICED_HUB_SLICE", 128);
  int sliced_rows_per_wave = env_int("INCAGG_SLICED_ROWS_PER_WAVE", 4);
  int grouped_hub_threshold = env_int("INCAGG_GROUPED_HUB_THRESHOLD", 128);
  int grouped_hub_slice = env_int("INCAGG_GROUPED_HUB_SLICE", 128);
};
static Tuning &tuning() {
  static Tuning t;
  return t;
}

template <int VEC, int CH, bool INCAGG>
static int launch_spmm(SpmmArgs &a, long long max_row_nnz, hipStream_t stream) {
  const int hub_threshold = std::max(1, tuning().hub_threshold);
  const int hub_slice = std::max(1, tuning().hub_slice);
  const int hub_blocks = std::max(1, tuning().hub_blocks);
  a.hub_slice = hub_slice;
  const bool may_have_hubs = max_row_nnz < 0 || max_row_nnz > hub_threshold;
  HubQueue *q = nullptr;
  bool advance = false;
  if (may_have_hubs) {
    int rc = hub_queue_for(stream, &q);
    if (rc) return rc;
    a.hub_threshold = hub_threshold;
    rc = bind_queue(a, q, stream, &advance);
    if (rc) return rc;
  } else {
    a.hub_threshold = INT_MAX;
    a.queue = nullptr;
    a.q_count = a.q_count_next = nullptr;
    a.q_cap = 0;
  }
  const int blocks = (a.n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  hipLaunchKernelGGL((spmm_rows_kernel<VEC, CH, INCAGG>), dim3(blocks), dim3(kWave * kWavesPerBlock),
                     0, stream, a);
  INCAGG_HIP_TRY(hipGetLastError());
  if (may_have_hubs) {
    hipLaunchKernelGGL((spmm_hub_kernel<VEC, CH, INCAGG>), dim3(hub_blocks),
                       dim3(kWave * kWavesPerBlock), 0, stream, a);
    INCAGG_HIP_TRY(hipGetLastError());
    if (advance) q->parity ^= 1;
  }
  return INCAGG_OK;
}

template <bool INCAGG>
static int launch_sliced(SpmmArgs &a, long long max_row_nnz, bool grouped, hipStream_t stream) {
  // a wave covers up to 64 edges per memory round trip here, so rows are cut later and into
  // longer slices than in the whole-row kernel (measured on an ogbn-products-sized batch:
  // threshold/slice 1024/512 -> 93.7 us, 512/256 -> 89.8 us, 256/128 -> 88.6 us per launch
  // pair at 4 rows per wave; 16 rows per wave -> 97 us)
  // (8 rows per wave: a lane group covers only 8 edges per round trip, so rows are cut
  // earlier; 64..128 / 128 are equally good, 32 / 64 and 512 / 128 lose 20-30 %)
  const int hub_threshold = std::max(1, grouped ? tuning().grouped_hub_threshold
                                                : tuning().sliced_hub_threshold);
  // 2^20 slices of >= 64 edges cover 67M hub edges per call
  const int hub_slice = std::max(64, grouped ? tuning().grouped_hub_slice : tuning().sliced_hub_slice);
  const int hub_blocks = std::max(1, tuning().hub_blocks);
  const int n_slices = (a.width + kSliceFloats - 1) / kSliceFloats;
  a.hub_slice = hub_slice;
  const bool may_have_hubs = max_row_nnz < 0 || max_row_nnz > hub_threshold;
  HubQueue *q = nullptr;
  bool advance = false;
  if (may_have_hubs) {
    int rc = hub_queue_for(stream, &q);
    if (rc) return rc;
    a.hub_threshold = hub_threshold;
    rc = bind_queue(a, q, stream, &advance);
    if (rc) return rc;
  } else {
    a.hub_threshold = INT_MAX;
  }
  const int rows_per_wave = grouped ? 8 : std::min(63, std::max(1, tuning().sliced_rows_per_wave));
  const long long rows_per_block = (long long)kWavesPerBlock * rows_per_wave;
  const long long row_groups = (a.n_rows + rows_per_block - 1) / rows_per_block;
  if (row_groups * n_slices > INT_MAX) return fail(INCAGG_E_INVALID, "grid too large");
  // 32-bit byte offsets on a scalar base cover a single source matrix of < 4 GiB
  const bool wide = a.x2 != nullptr || (long long)a.n_cols * a.ldx * 4 >= (1ll << 32) ||
                    (INCAGG && (long long)a.n_cols * a.ldin * 4 >= (1ll << 32));
  const dim3 grid((unsigned)(row_groups * n_slices)), block(kWave * kWavesPerBlock);
  if (grouped && wide)
    hipLaunchKernelGGL((spmm_grouped_kernel<INCAGG, true>), grid, block, 0, stream, a, n_slices);
  else if (grouped)
    hipLaunchKernelGGL((spmm_grouped_kernel<INCAGG, false>), grid, block, 0, stream, a, n_slices);
  else if (wide)
    hipLaunchKernelGGL((spmm_sliced_kernel<INCAGG, true>), grid, block, 0, stream, a, n_slices,
                       rows_per_wave);
  else
    hipLaunchKernelGGL((spmm_sliced_kernel<INCAGG, false>), grid, block, 0, stream, a, n_slices,
                       rows_per_wave);
  INCAGG_HIP_TRY(hipGetLastError());
  if (may_have_hubs) {
    const dim3 hub_grid(std::max(1, hub_blocks / n_slices) * n_slices);
    if (wide)
      hipLaunchKernelGGL((spmm_sliced_hub_kernel<INCAGG, true>), hub_grid, block, 0, stream, a,
                         n_slices);
    else
      hipLaunchKernelGGL((spmm_sliced_hub_kernel<INCAGG, false>), hub_grid, block, 0, stream, a,
                         n_slices);
    INCAGG_HIP_TRY(hipGetLastError());
    if (advance) q->parity ^= 1;
  }
  return INCAGG_OK;
}

template <bool INCAGG>
static int dispatch_spmm(SpmmArgs &a, bool vec4, long long nnz, long long max_row_nnz,
                         hipStream_t stream) {
  const int w = a.width;
  // Variant ("spmm_sliced" tuning key): 0 = whole rows per wave, 1 = feature-sliced with one
  // row per wave, 2 = feature-sliced with 8 rows per wave, -1 (default) = by shape.  Measured
  // on MI355X, ogbn-products-sized batch at width 256, us per call (rows / sliced / grouped with
  // a degree-sorted row_order): forward [B, M] 98 / 86 / 77; split source 97 / 93 / 90;
  // transposed [M, B] with 7 edges per row 175 / 262 / 88; transposed [B, B] 67 / 65 / 55.
  // Without a row_order the grouped kernel's lane groups idle behind the longest of their 8
  // rows (forward 107 us), so it is only chosen when the caller supplies one.
  const int mode = tuning().spmm_sliced;
  const bool by_shape = w >= 2 * kSliceFloats && a.x2 == nullptr && nnz >= 16ll * a.n_rows;
  const bool grouped = vec4 && (mode == 2 || (mode == -1 && a.row_order != nullptr));
  const bool sliced = vec4 && (mode == 1 || (mode == -1 && by_shape));
  if (grouped || sliced) return launch_sliced<INCAGG>(a, max_row_nnz, grouped, stream);
  if (vec4) {
    if (w <= 256) return launch_spmm<4, 1, INCAGG>(a, max_row_nnz, stream);
    if (w <= 512) return launch_spmm<4, 2, INCAGG>(a, max_row_nnz, stream);
    if (w <= 1024) return launch_spmm<4, 4, INCAGG>(a, max_row_nnz, stream);
  } else {
    if (w <= 64) return launch_spmm<1, 1, INCAGG>(a, max_row_nnz, stream);
    if (w <= 128) return launch_spmm<1, 2, INCAGG>(a, max_row_nnz, stream);
    if (w <= 256) return launch_spmm<1, 4, INCAGG>(a, max_row_nnz, stream);
  }
  return fail(INCAGG_E_INVALID, "width %d unsupported (max 1024 when 16-byte aligned, else 256)", w);
}

static bool rows_vec4(const float *p, long long ld) { return !p || (aligned16(p) && ld % 4 == 0); }

}  // namespace incagg

using namespace incagg;

static int check_common(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx,
                        int64_t n_rows, int64_t n_cols, int64_t width, int reduce,
                        const float *add, int64_t ldadd, float *y, int64_t ldy) {
  INCAGG_REQUIRE(n_rows >= 0 && n_cols >= 0 && width >= 0, "negative size");
  INCAGG_REQUIRE(n_rows < INT_MAX - 8 && n_cols < INT_MAX && width <= 1024,
                 "size out of range (rows %lld cols %lld width %lld)", (long long)n_rows,
                 (long long)n_cols, (long long)width);
  INCAGG_REQUIRE(reduce == INCAGG_REDUCE_SUM || reduce == INCAGG_REDUCE_MEAN, "bad reduce %d", reduce);
  if (n_rows == 0 || width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(rowptr && y, "null rowptr or output");
  INCAGG_REQUIRE(ldy >= width, "ldy %lld < width %lld", (long long)ldy, (long long)width);
  INCAGG_REQUIRE(!x || ldx >= width, "ldx %lld < width %lld", (long long)ldx, (long long)width);
  INCAGG_REQUIRE(!add || ldadd >= width, "ldadd %lld < width %lld", (long long)ldadd, (long long)width);
  // col may be NULL only for an adjacency without edges (nnz is known on the device only)
  INCAGG_REQUIRE(n_cols == 0 || x, "null x with n_cols > 0");
  INCAGG_REQUIRE((const void *)x != (const void *)y, "y must not alias x");
  return INCAGG_OK;
}

extern "C" int incagg_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                   const float *x, int64_t ldx, const float *x_halo,
                                   int64_t ld_halo, int64_t n_split, int64_t n_rows,
                                   int64_t n_cols, int64_t width, int reduce, float scale,
                                   const float *add, int64_t ldadd, float add_scale,
                                   int64_t nnz, int64_t max_row_nnz, const int32_t *row_order,
                                   float *y, int64_t ldy, void *stream) {
  int rc = check_common(rowptr, col, x, ldx, n_rows, n_cols, width, reduce, add, ldadd, y, ldy);
  INCAGG_REQUIRE(nnz >= 0 && nnz < INT_MAX, "nnz %lld out of range", (long long)nnz);
  if (rc) return rc;
  if (n_rows == 0 || width == 0) return INCAGG_OK;
  if (!x_halo) n_split = n_cols;
  INCAGG_REQUIRE(n_split >= 0 && n_split <= n_cols, "n_split %lld outside [0, %lld]",
                 (long long)n_split, (long long)n_cols);
  INCAGG_REQUIRE(!x_halo || ld_halo >= width, "ld_halo %lld < width %lld", (long long)ld_halo,
                 (long long)width);
  INCAGG_REQUIRE((const void *)x_halo != (const void *)y, "y must not alias x_halo");
  SpmmArgs a{};
  a.rowptr = rowptr; a.col = col; a.val = val; a.x = x; a.ldx = ldx;
  a.x2 = x_halo; a.ldx2 = ld_halo; a.split = (int)n_split;
  a.add = add; a.ldadd = ldadd; a.y = y; a.ldy = ldy;
  a.n_rows = (int)n_rows; a.n_cols = (int)n_cols; a.width = (int)width;
  a.mean = reduce == INCAGG_REDUCE_MEAN; a.scale = scale; a.add_scale = add_scale;
  a.row_order = row_order;
  rc = error_flag_address(&a.err);
  if (rc) return rc;
  const bool vec4 = width % 4 == 0 && rows_vec4(x, ldx) && rows_vec4(x_halo, ld_halo) &&
                    rows_vec4(add, ldadd) && rows_vec4(y, ldy);
  return dispatch_spmm<false>(a, vec4, nnz, max_row_nnz, static_cast<hipStream_t>(stream));
}

extern "C" int incagg_spmm_csr_incagg_f32(const int32_t *rowptr, const int32_t *col,
                                          const float *val, const float *x, int64_t ldx,
                                          const float *m_in, int64_t ldin, const float *m_ag,
                                          int64_t ldag, int64_t n_rows, int64_t width, int reduce,
                                          float scale, const float *add, int64_t ldadd,
                                          float add_scale, int64_t nnz, int64_t max_row_nnz,
                                          const int32_t *row_order, float *h, int64_t ldh,
                                          void *stream) {
  int rc = check_common(rowptr, col, x, ldx, n_rows, n_rows, width, reduce, add, ldadd, h, ldh);
  INCAGG_REQUIRE(nnz >= 0 && nnz < INT_MAX, "nnz %lld out of range", (long long)nnz);
  if (rc) return rc;
  if (n_rows == 0 || width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(m_in && m_ag, "null m_in or m_ag");
  INCAGG_REQUIRE(ldin >= width && ldag >= width, "ldin/ldag smaller than width");
  INCAGG_REQUIRE((const void *)m_in != (const void *)h, "h must not alias m_in");
  SpmmArgs a{};
  a.rowptr = rowptr; a.col = col; a.val = val; a.x = x; a.ldx = ldx;
  a.split = (int)n_rows;
  a.m_in = m_in; a.ldin = ldin; a.m_ag = m_ag; a.ldag = ldag;
  a.add = add; a.ldadd = ldadd; a.y = h; a.ldy = ldh;
  a.n_rows = (int)n_rows; a.n_cols = (int)n_rows; a.width = (int)width;
  a.mean = reduce == INCAGG_REDUCE_MEAN; a.scale = scale; a.add_scale = add_scale;
  a.row_order = row_order;
  rc = error_flag_address(&a.err);
  if (rc) return rc;
  const bool vec4 = width % 4 == 0 && rows_vec4(x, ldx) && rows_vec4(m_in, ldin) &&
                    rows_vec4(m_ag, ldag) && rows_vec4(add, ldadd) && rows_vec4(h, ldh);
  return dispatch_spmm<true>(a, vec4, nnz, max_row_nnz, static_cast<hipStream_t>(stream));
}

extern "C" int incagg_set_tuning(const char *key, int value) {
  INCAGG_REQUIRE(key != nullptr, "null key");
  Tuning &t = tuning();
  const std::string k(key);
  if (k == "spmm_sliced") t.spmm_sliced = value;
  else if (k == "hub_threshold") t.hub_threshold = value;
  else if (k == "hub_slice") t.hub_slice = value;
  else if (k == "hub_blocks") t.hub_blocks = value;
  else if (k == "sliced_hub_threshold") t.sliced_hub_threshold = value;
  else if (k == "sliced_hub_slice") t.sliced_hub_slice = value;
  else if (k == "sliced_rows_per_wave") t.sliced_rows_per_wave = value;
  else if (k == "grouped_hub_threshold") t.grouped_hub_threshold = value;
  else if (k == "grouped_hub_slice") t.grouped_hub_slice = value;
  else return fail(INCAGG_E_INVALID, "unknown tuning key '%s'", key);
  return INCAGG_OK;
}

extern "C" int incagg_rows_sub_f32(const float *a, int64_t lda, const float *b, int64_t ldb,
                                   int64_t n_rows, int64_t width, float *d, int64_t ldd,
                                   void *stream) {
  INCAGG_REQUIRE(n_rows >= 0 && width >= 0 && width < INT_MAX, "bad size");
  if (n_rows == 0 || width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(a && b && d, "null pointer");
  INCAGG_REQUIRE(lda >= width && ldb >= width && ldd >= width, "leading dimension < width");
  const bool vec4 = width % 4 == 0 && rows_vec4(a, lda) && rows_vec4(b, ldb) && rows_vec4(d, ldd);
  const long long items = n_rows * (width / (vec4 ? 4 : 1));
  const int blocks = (int)std::min<long long>((items + 255) / 256, 2048);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec4)
    hipLaunchKernelGGL((rows_sub_kernel<4>), dim3(blocks), dim3(256), 0, s, a, lda, b, ldb, n_rows,
                       (int)width, d, ldd);
  else
    hipLaunchKernelGGL((rows_sub_kernel<1>), dim3(blocks), dim3(256), 0, s, a, lda, b, ldb, n_rows,
                       (int)width, d, ldd);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}
