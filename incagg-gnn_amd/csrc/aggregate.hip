// CSR neighbour aggregation for gfx950, one launch per call, no device-side queue.
//
// Work is laid out by a PLAN that is built once per adjacency (spmm_plan.hip): the rows in
// descending order of length, and for the rows longer than the plan's threshold a list of
// (row, segment) items.  The grid is [items | groups of short rows] x feature slices:
//   * an item is one workgroup: its four waves split the segment's edges into 64-edge windows,
//     the partial sums meet in LDS and one wave writes the row -- no atomics, bit-reproducible;
//     only rows cut into SEVERAL segments (longer than the plan's segment length) are zeroed by a
//     small launch in front and combined with fp32 atomics;
//   * short rows run 8 per wave (feature-sliced kernel) or one per wave (whole-row kernel).
// Long rows come first in the grid: they are the longest-running workgroups.
//
// Two kernels share the plan:
//   spmm_plan_kernel   rows 16-byte aligned: the feature dimension is cut into 128-byte slices and
//                      slice s is processed only by workgroups with blockIdx % n_slices == s, i.e.
//                      (round-robin dispatch) by one XCD: every source byte has one home L2 and
//                      the per-XCD working set shrinks by the number of slices.  8 lanes cover the
//                      128 bytes of a gathered piece, so one load instruction of a wave fetches 8
//                      different source rows; indices reach the lane groups by ds_swizzle.
//   spmm_rows_kernel   any width / alignment: the lanes of a wave span the feature width, column
//                      ids are fetched 64 at a time and broadcast with v_readlane.
//
// Nothing here faults on bad index data: row pointers outside [0, nnz], column ids outside
// [0, n_cols), plan entries outside the matrix are skipped and latched in the device error flag.
//
// HBM roofline: a call moves E*(4+4) index/weight bytes, (B+1)*4 row pointers, one read of every
// distinct source row (width*4 B each) and one write of every destination row; flops are
// 2*E*width, i.e. ~0.5 flop per gathered byte.
#include "common.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <string>

namespace incagg {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kSliceFloats = 32;  // 128-byte feature slices
constexpr int kCombineWidth = 1024;                                // widest row the slices cover
constexpr int kCombineSlot = kCombineWidth + kCombineWidth / kSliceFloats;
static_assert(kCombineSlot == INCAGG_SPMM_COMBINE_SLOT_FLOATS, "header and kernel disagree");

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
// max(0, v) per component when `on`, else v untouched (NaNs included)
template <int VEC>
__device__ __forceinline__ typename VT<VEC>::type vrelu(typename VT<VEC>::type v, bool on);
template <>
__device__ __forceinline__ float vrelu<1>(float v, bool on) { return on ? fmaxf(v, 0.f) : v; }
template <>
__device__ __forceinline__ f4 vrelu<4>(f4 v, bool on) {
  f4 r;
  r.x = on ? fmaxf(v.x, 0.f) : v.x; r.y = on ? fmaxf(v.y, 0.f) : v.y;
  r.z = on ? fmaxf(v.z, 0.f) : v.z; r.w = on ? fmaxf(v.w, 0.f) : v.w;
  return r;
}
// max(0, v) as ONE integer max per float: a float with its sign bit set is a negative int, so
// max(bits, 0) gives +0.0 for everything negative and keeps everything else (NaNs with a clear
// sign bit included); with the lower bound INT_MIN instead of 0 the value passes unchanged.
__device__ __forceinline__ float relu_lb(float v, int lb) {
  return __int_as_float(max(__float_as_int(v), lb));
}
__device__ __forceinline__ f4 relu_lb4(f4 v, int lb) {
  f4 r;
  r.x = relu_lb(v.x, lb); r.y = relu_lb(v.y, lb); r.z = relu_lb(v.z, lb); r.w = relu_lb(v.w, lb);
  return r;
}
// v where g > 0, else 0, per component
template <int VEC>
__device__ __forceinline__ typename VT<VEC>::type vgate(typename VT<VEC>::type v,
                                                        typename VT<VEC>::type g);
template <>
__device__ __forceinline__ float vgate<1>(float v, float g) { return g > 0.f ? v : 0.f; }
template <>
__device__ __forceinline__ f4 vgate<4>(f4 v, f4 g) {
  f4 r;
  r.x = g.x > 0.f ? v.x : 0.f; r.y = g.y > 0.f ? v.y : 0.f;
  r.z = g.z > 0.f ? v.z : 0.f; r.w = g.w > 0.f ? v.w : 0.f;
  return r;
}

struct SpmmArgs {
  const int *rowptr;
  const int *col;
  const float *val;
  const int2 *edges;  // optional: (col[e], bits of val[e]) side by side, one 8-byte load per edge
  const float *x;
  long long ldx;
  const float *x2;  // source rows [split, n_cols) live here (row c at x2 + (c - split) * ldx2)
  long long ldx2;
  int split;
  unsigned ldb1, ldb2;  // ldx, ldx2 in bytes
  const float *m_in;
  long long ldin;
  const float *m_ag;
  long long ldag;
  const float *add;
  long long ldadd;
  const float *add2;  // optional second epilogue term: y += add2_scale * add2[i, :]
  long long ldadd2;
  float add2_scale;
  float *y;
  long long ldy;
  int n_rows;
  int n_cols;
  int width;
  int mean;
  float scale;
  float add_scale;
  int nnz;  // entries of col / val: a row pointer outside [0, nnz] is reported, not followed
  int relu_x;           // source rows taken from x (not x2) pass through max(0, .) when gathered
  const float *gate;    // y[i, :] is zeroed where gate[i, :] <= 0 (the backward of a relu)
  long long ldgate;
  int *err;
  // plan (order == nullptr: rows in storage order, no items)
  const int *order;
  const int2 *items;
  int n_hub;    // the first n_hub entries of `order` are covered by `items`
  int n_items;
  int seg;      // edges per item
  // rows cut into several segments: their ids and, when the plan brings one, the combine
  // buffer (kCombineSlot floats per such row: accumulators, then one arrival counter per slice)
  const int *multi_rows;
  int n_multi;
  float *comb;
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Edge range of `row`, never outside the col / val arrays.
__device__ __forceinline__ void row_range(const SpmmArgs &a, int row, int &rs, int &re) {
  rs = a.rowptr[row];
  re = a.rowptr[row + 1];
  if (rs < 0 || re < rs || re > a.nnz) {
    *a.err = 1;
    rs = re = 0;
  }
  if (a.n_cols <= 0) re = rs;
}

// Item `i` of the plan: its row and the edges [es, ee) of its segment; false when the entry
// does not name a row and a segment of this matrix (reported, skipped).  Wave-uniform.
__device__ __forceinline__ bool item_range(const SpmmArgs &a, int i, int &row, int &seg_idx,
                                           int &es, int &ee, int &deg) {
  const int2 ent = a.items[i];
  row = rfl(ent.x);
  seg_idx = rfl(ent.y);
  if ((unsigned)row >= (unsigned)a.n_rows || seg_idx < 0) {
    *a.err = 1;
    return false;
  }
  int rs, re;
  row_range(a, row, rs, re);
  rs = rfl(rs);
  re = rfl(re);
  deg = re - rs;
  if ((long long)seg_idx * a.seg >= (long long)deg) {  // (also: an item for an empty row)
    *a.err = 1;
    return false;
  }
  es = rs + seg_idx * a.seg;
  ee = min(re, es + a.seg);
  return true;
}

// ---------------------------------------------------------------- whole rows per wave
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
    const bool relu = a.relu_x && c < a.split;
#pragma unroll
    for (int ch = 0; ch < CH; ch++) xv[u][ch] = vrelu<VEC>(ldv<VEC>(xr + off[ch]), relu);
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
  // gathers in flight per wave: 8 rows of up to 1 KB; narrow unaligned rows (one float per
  // lane, e.g. the 41 / 47-wide tables of APPNP) are cheap in registers and need more of them
  // to keep the memory pipe busy
  constexpr int UMAX = INCAGG ? 4 : (VEC == 1 && CH == 1 ? 32 : VEC == 1 && CH == 2 ? 16 : 8);
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
    if (UMAX > 16 && j + 16 <= cnt) {
      edge_step<VEC, CH, INCAGG, 16>(a, my_c, my_v, j, off, acc);
      j += 16;
    }
    if (UMAX > 8 && j + 8 <= cnt) {
      edge_step<VEC, CH, INCAGG, 8>(a, my_c, my_v, j, off, acc);
      j += 8;
    }
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

// y[row] = s * acc + epilogue terms; `atomic`: the row was zeroed in front of this launch and
// several segments add onto it (the epilogue terms ride with segment 0: `base`).
template <int VEC, int CH, bool INCAGG>
__device__ __forceinline__ void store_row(const SpmmArgs &a, int row, float s,
                                          const int (&off)[CH], const bool (&live)[CH],
                                          const typename VT<VEC>::type (&acc)[CH],
                                          bool atomic = false, bool base = true) {
  using V = typename VT<VEC>::type;
#pragma unroll
  for (int ch = 0; ch < CH; ch++) {
    if (!live[ch]) continue;
    V out = acc[ch] * s;
    if (base) {
      if (INCAGG) out += a.scale * ldv<VEC>(a.m_ag + (size_t)row * a.ldag + off[ch]);
      if (a.add) out += a.add_scale * ldv<VEC>(a.add + (size_t)row * a.ldadd + off[ch]);
      if (a.add2) out += a.add2_scale * ldv<VEC>(a.add2 + (size_t)row * a.ldadd2 + off[ch]);
    }
    if (a.gate) out = vgate<VEC>(out, ldv<VEC>(a.gate + (size_t)row * a.ldgate + off[ch]));
    float *dst = a.y + (size_t)row * a.ldy + off[ch];
    if (!atomic) {
      stv<VEC>(dst, out);
    } else if constexpr (VEC == 1) {
      atomicAdd(dst, out);
    } else {
      atomicAdd(dst, out.x);
      atomicAdd(dst + 1, out.y);
      atomicAdd(dst + 2, out.z);
      atomicAdd(dst + 3, out.w);
    }
  }
}

template <int VEC, int CH, bool INCAGG>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_rows_kernel(const SpmmArgs a) {
  using V = typename VT<VEC>::type;
  __shared__ V part[kWavesPerBlock][CH][kWave];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  int off[CH];
  bool live[CH];
  column_offsets<VEC, CH>(lane, a.width, off, live);
  V acc[CH];
#pragma unroll
  for (int ch = 0; ch < CH; ch++) acc[ch] = V(0.f);

  if ((int)blockIdx.x < a.n_items) {
    // ---- one segment of a long row: the four waves take a quarter each (whole windows)
    int row, seg_idx, es, ee, deg;
    if (!item_range(a, blockIdx.x, row, seg_idx, es, ee, deg)) return;  // block-uniform
    const int windows = (ee - es + kWave - 1) / kWave;
    const int per_wave = (windows + kWavesPerBlock - 1) / kWavesPerBlock * kWave;
    const int ws = min(ee, es + wave * per_wave), we = min(ee, ws + per_wave);
    accumulate_edges<VEC, CH, INCAGG>(a, ws, we, lane, off, acc);
#pragma unroll
    for (int ch = 0; ch < CH; ch++) part[wave][ch][lane] = acc[ch];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int ch = 0; ch < CH; ch++)
      acc[ch] = (part[0][ch][lane] + part[1][ch][lane]) + (part[2][ch][lane] + part[3][ch][lane]);
    const float s = a.scale * (a.mean ? 1.0f / (float)max(deg, 1) : 1.0f);
    const bool several = deg > a.seg;
    store_row<VEC, CH, INCAGG>(a, row, s, off, live, acc, several, seg_idx == 0);
    return;
  }

  // ---- short rows, one per wave.  Blocks b and b + 8 share an XCD (and its L2): give every
  // XCD one contiguous range of the order, since neighbouring entries share source rows.
  const int nb = gridDim.x - a.n_items, b = blockIdx.x - a.n_items;
  const int q = nb >> 3, r = nb & 7, xcd = b & 7;
  const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int p = a.n_hub + lb * kWavesPerBlock + wave;
  if (p >= a.n_rows) return;
  int row = a.order ? rfl(a.order[p]) : p;
  if ((unsigned)row >= (unsigned)a.n_rows) {  // bad plan entry: reported, not followed
    *a.err = 1;
    return;
  }
  int rs, re;
  row_range(a, row, rs, re);
  rs = rfl(rs);
  re = rfl(re);
  accumulate_edges<VEC, CH, INCAGG>(a, rs, re, lane, off, acc);
  const float s = a.scale * (a.mean ? 1.0f / (float)max(re - rs, 1) : 1.0f);
  store_row<VEC, CH, INCAGG>(a, row, s, off, live, acc);
}

// ---------------------------------------------------------------- feature-sliced kernel
__device__ __forceinline__ f4 shfl_xor4(f4 v, int m) {
  f4 r;
  r.x = __shfl_xor(v.x, m); r.y = __shfl_xor(v.y, m);
  r.z = __shfl_xor(v.z, m); r.w = __shfl_xor(v.w, m);
  return r;
}

// 16 bytes of a row at a 4-byte-aligned address.  Rows start 16-byte aligned and the pieces of a
// row are 16 bytes apart, except the LAST piece of a row whose width is not a multiple of 4: that
// piece is moved back so that it ends with the row (it overlaps its neighbour, see `skip` in the
// kernel), which keeps every lane on one 16-byte access and nothing outside the row touched.
// ODD (the width is not a multiple of 4) selects the 4-byte-aligned access type.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <bool ODD>
__device__ __forceinline__ f4 ld4(const float *p) {
  if constexpr (ODD) return *reinterpret_cast<const f4u *>(p);
  else return *reinterpret_cast<const f4 *>(p);
}
template <bool ODD>
__device__ __forceinline__ void st4(float *p, f4 v) {
  if constexpr (ODD) *reinterpret_cast<f4u *>(p) = v;
  else *reinterpret_cast<f4 *>(p) = v;
}

// Column ids / weights of the 64-edge window starting at e0, one coalesced load each.
// Lane 8 g + j holds window edge 8 j + g: round j of lane group g (see gather_rounds).
// The 64 addresses still fall into the same two cache lines, so the load stays coalesced.
// Lanes beyond `re` re-read the last edge with weight 0, so every lane holds a real
// neighbour and the gather loop needs no per-lane guards.  (re > e0 always.)
__device__ __forceinline__ void load_window(const SpmmArgs &a, int e0, int re, int lane, int &c,
                                            float &w) {
  const int slot = e0 + 8 * (lane & 7) + (lane >> 3);
  const int e = min(slot, re - 1);
  if (a.edges) {  // (kernel argument: a scalar branch)
    const int2 cw = a.edges[e];
    c = cw.x;
    w = __int_as_float(cw.y);
  } else {
    c = a.col[e];
    w = a.val ? a.val[e] : 1.0f;
  }
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

// Rounds [R0, R0 + N) of a lane group's 8 indices: in round r the group (8 lanes = the 128
// bytes of one piece) gathers the row named by its lane r, so one load instruction of the
// wave fetches 8 different source rows; all N gathers are issued before the first FMA.  The
// indices reach the group through the cross-lane network: the vector-memory pipe only sees
// the row gathers.
// WIDE: row offsets need 64 bits or the source is split in two matrices; otherwise a
// 32-bit byte offset per lane on top of a scalar base is enough.
// BSEL >= 0 (lane groups of any size, see spmm_narrow_kernel): the indices come from lanes
// bsel / 4 + R0 ... of the wave through ds_bpermute instead (`bsel`: byte address of the
// group's first lane).
template <bool INCAGG, int WIDE, bool ODD, int R0, int N, bool BPERM = false>
__device__ __forceinline__ void gather_rounds(const SpmmArgs &a, int my_c, float my_w, int coloff,
                                              f4 &acc, int bsel = 0) {
  int c[N];
  float w[N];
  f4 xv[N], mv[N];
#pragma unroll
  for (int r = 0; r < N; r++) {
    if constexpr (BPERM) {  // (the weights follow once the gathers are on their way)
      c[r] = __builtin_amdgcn_ds_bpermute(bsel + 4 * (R0 + r), my_c);
      continue;
    }
    // (the ternary chain keeps the swizzle pattern an immediate)
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
    if constexpr (WIDE != 0) {
      // two source matrices (or one beyond 4 GiB), branch-free: row id, pitch and base are
      // selected per lane, then ONE multiply-add -- a 64-bit one in general (WIDE == 1); when
      // row ids and pitches fit 24 bits and each matrix 4 GiB (WIDE == 2) a full-rate 24-bit
      // one, which is what a products-sized History table read in place needs
      const bool second = c[r] >= a.split;
      const unsigned row = second ? (unsigned)(c[r] - a.split) : (unsigned)c[r];
      const unsigned ldb = second ? a.ldb2 : a.ldb1;
      const char *base = second ? reinterpret_cast<const char *>(a.x2)
                                : reinterpret_cast<const char *>(a.x);
      unsigned long long off;
      if constexpr (WIDE == 2) off = __umul24(row, ldb) + (unsigned)coloff * 4u;
      else off = (unsigned long long)row * ldb + (unsigned)coloff * 4u;
      xv[r] = ld4<ODD>(reinterpret_cast<const float *>(base + off));
      if (INCAGG) mv[r] = ld4<ODD>(a.m_in + (size_t)c[r] * a.ldin + coloff);
    } else {
      // byte offsets computed in 32 bits: "scalar base + 32-bit lane offset" addressing
      const unsigned off = ((unsigned)c[r] * (unsigned)a.ldx + (unsigned)coloff) * 4u;
      xv[r] = ld4<ODD>(reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.x) + off));
      if (INCAGG) {
        const unsigned moff = ((unsigned)c[r] * (unsigned)a.ldin + (unsigned)coloff) * 4u;
        mv[r] = ld4<ODD>(reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.m_in) + moff));
      }
    }
  }
  if constexpr (BPERM) {
#pragma unroll
    for (int r = 0; r < N; r++)
      w[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(bsel + 4 * (R0 + r), __float_as_int(my_w)));
  }
  if (a.relu_x) {  // (kernel argument: a scalar branch, nothing on the other path)
#pragma unroll
    for (int r = 0; r < N; r++)  // rows of the second source matrix are taken as they are
      xv[r] = relu_lb4(xv[r], WIDE && c[r] >= a.split ? INT_MIN : 0);
  }
#pragma unroll
  for (int r = 0; r < N; r++) acc += w[r] * (INCAGG ? xv[r] - mv[r] : xv[r]);
}

// `rounds` (wave-uniform, 1..8) rounds of the indices in (my_c, my_w): short windows only
// pay for the rounds they fill: 1, 2, 4 or 8.
template <bool INCAGG, int WIDE, bool ODD, bool BPERM = false>
__device__ __forceinline__ void gather_some(const SpmmArgs &a, int my_c, float my_w, int rounds,
                                            int coloff, f4 &acc, int bsel = 0) {
  if (rounds > 4) {
    if constexpr (INCAGG) {  // two operands per edge: 4 rounds = 8 gathers in flight
      gather_rounds<INCAGG, WIDE, ODD, 0, 4, BPERM>(a, my_c, my_w, coloff, acc, bsel);
      gather_rounds<INCAGG, WIDE, ODD, 4, 4, BPERM>(a, my_c, my_w, coloff, acc, bsel);
    } else {
      gather_rounds<INCAGG, WIDE, ODD, 0, 8, BPERM>(a, my_c, my_w, coloff, acc, bsel);
    }
  } else if (rounds > 2) {
    gather_rounds<INCAGG, WIDE, ODD, 0, 4, BPERM>(a, my_c, my_w, coloff, acc, bsel);
  } else if (rounds > 1) {
    gather_rounds<INCAGG, WIDE, ODD, 0, 2, BPERM>(a, my_c, my_w, coloff, acc, bsel);
  } else {
    gather_rounds<INCAGG, WIDE, ODD, 0, 1, BPERM>(a, my_c, my_w, coloff, acc, bsel);
  }
}

__device__ __forceinline__ int group_max8(int v) {  // max over the 8 lane groups of a wave
  v = max(v, __shfl_xor(v, 8));
  v = max(v, __shfl_xor(v, 16));
  v = max(v, __shfl_xor(v, 32));
  return v;
}

// Runs of kRun = 128 consecutive positions of [0, n) dealt out to `groups` classes in turn, the class
// of element i being (i + off) % groups: element i takes position regroup(i), so that the
// elements of one class walk through runs j, j + groups, j + 2 groups, ... in ascending order.
// (A bijection of [0, n): positions past the last whole round of groups * kRun stay put.)
// Scalar work.
constexpr int kRun = 128;
__device__ __forceinline__ int regroup(int i, int n, int groups, int off) {
  const int round = groups * kRun;
  if (i >= n - n % round) return i;
  const int j = (i + off) % groups, rank = i / groups;
  return ((rank / kRun) * groups + j) * kRun + rank % kRun;
}

// Position of element i of [0, n) when the elements are laid out class after class instead (the
// class of i being i % groups), in ascending order inside a class: every class takes ONE
// contiguous range.  Scalar work.
__device__ __forceinline__ int regroup_ranges(int i, int n, int groups) {
  const int j = i % groups;
  // classes below j: those below n % groups hold one element more
  return j * (n / groups) + min(j, n % groups) + i / groups;
}

// groups > 1 (narrow matrices: 1, 2 or 4 slices, so that 8 / n_slices workgroups in a row land
// on different XCDs): workgroup v takes the work of position regroup(v), i.e. the workgroups of
// one XCD (pair, quad) take one contiguous range of the plan's items and whole runs of 128 x 32
// consecutive short rows.  With a plan ordered chunk by chunk a run is about four chunks of
// 1,024 rows: their source rows -- mostly rows of the same chunks -- are fetched into ONE
// private L2, and the runs of an XCD alternate between all parts of the matrix (even work per
// XCD also where the rows of one part are much longer than those of another, e.g. the in-batch
// and the out-of-batch rows of a transposed batch adjacency; items are of similar length
// everywhere).  Measured, reddit-sized batch (116 k rows, 11.9 M edges) at width 41, us per
// launch, forward / transposed [M,B] / transposed [B,B]: plan by length only 494 / 446 / 393;
// chunked plan, contiguous ranges for items and rows 258 / 296 / 198; runs of 32 for both
// 282 / 253 / 207; ranges for items + runs of 32 269 / 261 / 200, + runs of 128 257 / 248 / 197.
// (second launch bound: waves per SIMD the register allocation must leave room for)
template <bool INCAGG, int WIDE, bool ODD>
__global__ __launch_bounds__(kWave *kWavesPerBlock, INCAGG ? 4 : 8) void spmm_plan_kernel(
    const SpmmArgs a, int n_slices, int groups) {
  __shared__ f4 part[kWavesPerBlock][8];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 3, t = lane & 7;
  const int slice = blockIdx.x % n_slices;
  int vb = blockIdx.x / n_slices;
  if (groups > 1)
    vb = vb < a.n_items ? regroup_ranges(vb, a.n_items, groups)
                        : a.n_items + regroup(vb - a.n_items, (int)(gridDim.x / n_slices) - a.n_items,
                                              groups, a.n_items);
  const int col0 = slice * kSliceFloats + t * 4;
  const bool live = col0 < a.width;
  // (width >= 4.)  A row's last piece ends with the row: when the width is not a multiple of 4
  // its first `skip` floats are also the last floats of the neighbouring lane's piece -- both
  // lanes compute them from the same inputs in the same order, i.e. bit-identically.
  // Lanes past the row's end (a last slice that the width does not fill) repeat the row's last
  // piece: the same cache line as their live neighbours, result discarded.  (Pointing them at
  // column 0 instead made the XCDs of a part-filled slice fetch a second line of every source
  // row -- one that belongs to another slice's L2s.)
  const int coloff = min(col0, a.width - 4);
  const int skip = live ? col0 - coloff : 0;

  if (vb < a.n_items) {
    // ---- one segment of a long row x one slice: wave w takes the windows w, w + 4, ...; in
    // a window the 8 lane groups take 8 different edges per round, 8 rounds in flight
    int row, seg_idx, es, ee, deg;
    if (!item_range(a, vb, row, seg_idx, es, ee, deg)) return;  // block-uniform
    f4 acc = f4(0.f);
    int e0 = es + wave * kWave;
    if (e0 < ee) {
      int my_c, next_c = 0;
      float my_w, next_w = 0.f;
      load_window(a, e0, ee, lane, my_c, my_w);
      for (; e0 < ee; e0 += kWave * kWavesPerBlock) {
        const int ne0 = e0 + kWave * kWavesPerBlock;
        if (ne0 < ee) load_window(a, ne0, ee, lane, next_c, next_w);
        gather_some<INCAGG, WIDE, ODD>(a, my_c, my_w, (min(kWave, ee - e0) + 7) >> 3, coloff, acc);
        my_c = next_c;
        my_w = next_w;
      }
    }
    acc += shfl_xor4(acc, 8);
    acc += shfl_xor4(acc, 16);
    acc += shfl_xor4(acc, 32);
    if (g == 0) part[wave][t] = acc;
    __syncthreads();
    if (wave != 0) return;
    const bool multi = deg > a.seg;
    int slot = -1;
    if (multi && a.comb) {  // which of the plan's multi-segment rows this is (wave-uniform)
      for (int base = 0; base < a.n_multi && slot < 0; base += kWave) {
        const int idx = base + lane;
        const unsigned long long hit = __ballot(idx < a.n_multi && a.multi_rows[idx] == row);
        if (hit) slot = base + __ffsll((long long)hit) - 1;
      }
      if (slot < 0) {  // not in the list: reported, row left alone
        *a.err = 1;
        return;
      }
    }
    if (g != 0 || !live) return;
    acc = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    const float s = a.scale * (a.mean ? 1.0f / (float)max(deg, 1) : 1.0f);
    f4 out = acc * s;
    if (seg_idx == 0) {
      if (INCAGG) out += a.scale * ld4<ODD>(a.m_ag + (size_t)row * a.ldag + coloff);
      if (a.add) out += a.add_scale * ld4<ODD>(a.add + (size_t)row * a.ldadd + coloff);
      if (a.add2) out += a.add2_scale * ld4<ODD>(a.add2 + (size_t)row * a.ldadd2 + coloff);
    }
    if (a.gate) out = vgate<4>(out, ld4<ODD>(a.gate + (size_t)row * a.ldgate + coloff));
    float *dst = a.y + (size_t)row * a.ldy + coloff;
    if (!multi) {
      st4<ODD>(dst, out);
    } else if (a.comb) {
      // Segments of one row meet in the plan's combine buffer, which is all zero between
      // launches: every segment adds its part, then counts its arrival; the last one to
      // arrive takes the totals out (leaving zeros behind), clears the counter and writes the
      // row.  The adds return their old values: a returned value means the add has been
      // performed at the device-wide coherence point, so waiting for them orders the adds
      // in front of the arrival count without a release fence (which would write the XCD's
      // whole L2 back).
      float *accp = a.comb + (size_t)slot * kCombineSlot + coloff;
      // (floats shared with the neighbouring piece are added by that piece only)
      const float r0 = atomicAdd(accp, skip > 0 ? 0.f : out.x);
      const float r1 = atomicAdd(accp + 1, skip > 1 ? 0.f : out.y);
      const float r2 = atomicAdd(accp + 2, skip > 2 ? 0.f : out.z);
      const float r3 = atomicAdd(accp + 3, out.w);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
      int *cnt = reinterpret_cast<int *>(a.comb + (size_t)slot * kCombineSlot + kCombineWidth) + slice;
      int prev = 0;
      if (t == 0) prev = atomicAdd(cnt, 1);
      prev = rfl(prev);   // (lane 0 is live and active: the first active lane)
      if (prev == (deg + a.seg - 1) / a.seg - 1) {
        // (only the floats this piece owns: the shared ones are the neighbouring piece's, which
        // may sit in another slice with arrivals of its own)
        if (skip < 1) dst[0] = atomicExch(accp, 0.f);
        if (skip < 2) dst[1] = atomicExch(accp + 1, 0.f);
        if (skip < 3) dst[2] = atomicExch(accp + 2, 0.f);
        dst[3] = atomicExch(accp + 3, 0.f);
        if (t == 0) atomicExch(cnt, 0);
      }
    } else {  // several segments add onto the row, zeroed in front of this launch
      if (skip < 1) atomicAdd(dst, out.x);
      if (skip < 2) atomicAdd(dst + 1, out.y);
      if (skip < 3) atomicAdd(dst + 2, out.z);
      atomicAdd(dst + 3, out.w);
    }
    return;
  }

  // ---- short rows: every group of 8 lanes owns a destination row (a wave = 8 rows x one
  // slice), walks its edges 8 at a time (one coalesced 32-byte index load per group) and keeps
  // its own accumulator: no cross-lane reduction.  The trip count of a wave is set by its
  // longest row; the plan's order puts rows of similar length next to each other.
  const int p = a.n_hub + ((vb - a.n_items) * kWavesPerBlock + wave) * 8 + g;
  const bool valid = p < a.n_rows;
  int row = 0, rs = 0, re = 0;
  if (valid) {
    row = a.order ? a.order[p] : p;
    if ((unsigned)row >= (unsigned)a.n_rows) {  // bad plan entry: reported, not followed
      *a.err = 1;
      row = p;
    }
    row_range(a, row, rs, re);
  }
  const int deg = re - rs;
  const int maxdeg = rfl(group_max8(deg));

  // Lane t of a group holds edge e + t; lanes past the row's end re-read its last edge
  // with weight 0, so the gather rounds need no per-lane guards.
  auto load_idx = [&](int e0, int &c, float &w) {
    if (e0 < re) {
      const int slot = e0 + t;
      const int e = min(slot, re - 1);
      if (a.edges) {
        const int2 cw = a.edges[e];
        c = cw.x;
        w = __int_as_float(cw.y);
      } else {
        c = a.col[e];
        w = a.val ? a.val[e] : 1.0f;
      }
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
    // (rounds that SOME group still needs: wave-uniform)
    if (e < re) gather_some<INCAGG, WIDE, ODD>(a, my_c, my_w, maxdeg - done, coloff, acc);
    my_c = next_c;
    my_w = next_w;
    e += 8;
  }
  if (valid && live) {
    const float s = a.scale * (a.mean ? 1.0f / (float)max(deg, 1) : 1.0f);
    f4 out = acc * s;
    if (INCAGG) out += a.scale * ld4<ODD>(a.m_ag + (size_t)row * a.ldag + coloff);
    if (a.add) out += a.add_scale * ld4<ODD>(a.add + (size_t)row * a.ldadd + coloff);
    if (a.add2) out += a.add2_scale * ld4<ODD>(a.add2 + (size_t)row * a.ldadd2 + coloff);
    if (a.gate) out = vgate<4>(out, ld4<ODD>(a.gate + (size_t)row * a.ldgate + coloff));
    st4<ODD>(a.y + (size_t)row * a.ldy + coloff, out);
  }
}

// ---------------------------------------------------------------- narrow rows in one pass
// Rows of 33..64 floats (the 41- and 47-wide tables of APPNP: models/appnp.py:24) are more than
// one 128-byte slice but much less than two: cut into slices, the second slice's XCDs walk the
// whole edge list again for a few floats per row, and a row's cache lines are requested by two
// L2s.  Here a group of L = ceil(width / 4) lanes (9..16) covers a WHOLE row with one 16-byte
// piece per lane, so a load instruction of a wave fetches R = 64 / L different source rows and
// the edge list is read once.  Same plan, same item / short-row split, same epilogue as
// spmm_plan_kernel; the lane groups get their indices through ds_bpermute (a group no longer
// is an aligned power of two of lanes).  The 64 - R L lanes left over shadow lanes of the last
// group (same addresses, results discarded).
struct NarrowLanes {
  int g, t;       // lane group, piece within the row
  bool live;      // owns a piece of an output row
  int bsel;       // byte address (lane id * 4) of the group's first lane, for ds_bpermute
};
__device__ __forceinline__ NarrowLanes narrow_lanes(int lane, int L, int R) {
  NarrowLanes n;
  const bool spare = lane >= R * L;
  n.g = spare ? R - 1 : lane / L;
  n.t = spare ? lane - R * L : lane - n.g * L;
  n.live = !spare;
  n.bsel = n.g * L * 4;
  return n;
}

__device__ __forceinline__ int wave_max(int v) {
  v = max(v, __shfl_xor(v, 1));
  v = max(v, __shfl_xor(v, 2));
  v = max(v, __shfl_xor(v, 4));
  v = max(v, __shfl_xor(v, 8));
  v = max(v, __shfl_xor(v, 16));
  v = max(v, __shfl_xor(v, 32));
  return v;
}

template <int WIDE, bool ODD>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_narrow_kernel(
    const SpmmArgs a, int L, int R, int groups) {
  __shared__ f4 part[kWavesPerBlock][kWave];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const NarrowLanes nl = narrow_lanes(lane, L, R);
  const int g = nl.g, t = nl.t;
  int vb = blockIdx.x;
  if (groups > 1)
    vb = vb < a.n_items ? regroup_ranges(vb, a.n_items, groups)
                        : a.n_items + regroup(vb - a.n_items, (int)gridDim.x - a.n_items, groups,
                                              a.n_items);
  const int col0 = t * 4;                        // (< width: L = ceil(width / 4))
  const int coloff = min(col0, a.width - 4);     // a row's last piece ends with the row
  const int skip = col0 - coloff;

  if (vb < a.n_items) {
    // ---- one segment of a long row: wave w takes the windows w, w + 4, ... of 8 R edges; in a
    // window lane (g, j < 8) holds edge j R + g, i.e. round j of group g
    int row, seg_idx, es, ee, deg;
    if (!item_range(a, vb, row, seg_idx, es, ee, deg)) return;  // block-uniform
    const int win = 8 * R;
    auto load_win = [&](int e0, int &c, float &w) {
      const int slot = e0 + t * R + g;
      const int e = min(slot, ee - 1);
      if (a.edges) {
        const int2 cw = a.edges[e];
        c = cw.x;
        w = __int_as_float(cw.y);
      } else {
        c = a.col[e];
        w = a.val ? a.val[e] : 1.0f;
      }
      if (slot >= ee || t >= 8 || !nl.live) w = 0.f;
      if ((unsigned)c >= (unsigned)a.n_cols) {  // never fault on a bad id
        *a.err = 1;
        c = 0;
        w = 0.f;
      }
    };
    f4 acc = f4(0.f);
    int e0 = es + wave * win;
    if (e0 < ee) {
      int my_c, next_c = 0;
      float my_w, next_w = 0.f;
      load_win(e0, my_c, my_w);
      for (; e0 < ee; e0 += win * kWavesPerBlock) {
        const int ne0 = e0 + win * kWavesPerBlock;
        if (ne0 < ee) load_win(ne0, next_c, next_w);
        gather_some<false, WIDE, ODD, true>(a, my_c, my_w, (min(win, ee - e0) + R - 1) / R, coloff,
                                            acc, nl.bsel);
        my_c = next_c;
        my_w = next_w;
      }
    }
    part[wave][lane] = nl.live ? acc : f4(0.f);
    __syncthreads();
    if (wave != 0) return;
    const bool multi = deg > a.seg;
    int slot = -1;
    if (multi && a.comb) {  // which of the plan's multi-segment rows this is (wave-uniform)
      for (int base = 0; base < a.n_multi && slot < 0; base += kWave) {
        const int idx = base + lane;
        const unsigned long long hit = __ballot(idx < a.n_multi && a.multi_rows[idx] == row);
        if (hit) slot = base + __ffsll((long long)hit) - 1;
      }
      if (slot < 0) {  // not in the list: reported, row left alone
        *a.err = 1;
        return;
      }
    }
    if (g != 0 || !nl.live) return;
    acc = f4(0.f);
    for (int w = 0; w < kWavesPerBlock; w++)
      for (int gg = 0; gg < R; gg++) acc += part[w][gg * L + t];
    const float s = a.scale * (a.mean ? 1.0f / (float)max(deg, 1) : 1.0f);
    f4 out = acc * s;
    if (seg_idx == 0) {
      if (a.add) out += a.add_scale * ld4<ODD>(a.add + (size_t)row * a.ldadd + coloff);
      if (a.add2) out += a.add2_scale * ld4<ODD>(a.add2 + (size_t)row * a.ldadd2 + coloff);
    }
    if (a.gate) out = vgate<4>(out, ld4<ODD>(a.gate + (size_t)row * a.ldgate + coloff));
    float *dst = a.y + (size_t)row * a.ldy + coloff;
    if (!multi) {
      st4<ODD>(dst, out);
    } else if (a.comb) {
      // (the protocol of spmm_plan_kernel's combine buffer, with one slice)
      float *accp = a.comb + (size_t)slot * kCombineSlot + coloff;
      const float r0 = atomicAdd(accp, skip > 0 ? 0.f : out.x);
      const float r1 = atomicAdd(accp + 1, skip > 1 ? 0.f : out.y);
      const float r2 = atomicAdd(accp + 2, skip > 2 ? 0.f : out.z);
      const float r3 = atomicAdd(accp + 3, out.w);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
      int *cnt = reinterpret_cast<int *>(a.comb + (size_t)slot * kCombineSlot + kCombineWidth);
      // every piece counts its own arrival (one counter per piece: the pieces of a segment do
      // not wait for each other)
      const int prev = atomicAdd(cnt + t, 1);
      if (prev == (deg + a.seg - 1) / a.seg - 1) {
        if (skip < 1) dst[0] = atomicExch(accp, 0.f);
        if (skip < 2) dst[1] = atomicExch(accp + 1, 0.f);
        if (skip < 3) dst[2] = atomicExch(accp + 2, 0.f);
        dst[3] = atomicExch(accp + 3, 0.f);
        atomicExch(cnt + t, 0);
      }
    } else {  // several segments add onto the row, zeroed in front of this launch
      if (skip < 1) atomicAdd(dst, out.x);
      if (skip < 2) atomicAdd(dst + 1, out.y);
      if (skip < 3) atomicAdd(dst + 2, out.z);
      atomicAdd(dst + 3, out.w);
    }
    return;
  }

  // ---- short rows: a wave = R rows, every lane group owns one and walks its edges 8 at a time
  const int p = a.n_hub + ((vb - a.n_items) * kWavesPerBlock + wave) * R + g;
  const bool valid = nl.live && p < a.n_rows;
  int row = 0, rs = 0, re = 0;
  if (valid) {
    row = a.order ? a.order[p] : p;
    if ((unsigned)row >= (unsigned)a.n_rows) {  // bad plan entry: reported, not followed
      *a.err = 1;
      row = p;
    }
    row_range(a, row, rs, re);
  }
  const int deg = re - rs;
  const int maxdeg = rfl(wave_max(deg));
  auto load_idx = [&](int e0, int &c, float &w) {
    if (e0 < re) {
      const int slot = e0 + t;
      const int e = min(slot, re - 1);
      if (a.edges) {
        const int2 cw = a.edges[e];
        c = cw.x;
        w = __int_as_float(cw.y);
      } else {
        c = a.col[e];
        w = a.val ? a.val[e] : 1.0f;
      }
      if (slot >= re || t >= 8) w = 0.f;
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
    // (rounds that SOME group still needs: wave-uniform.  A group whose row has ended sits the
    // gathers out; the groups that go on only read indices from their own lanes.)
    if (e < re)
      gather_some<false, WIDE, ODD, true>(a, my_c, my_w, maxdeg - done, coloff, acc, nl.bsel);
    my_c = next_c;
    my_w = next_w;
    e += 8;
  }
  if (valid) {
    const float s = a.scale * (a.mean ? 1.0f / (float)max(deg, 1) : 1.0f);
    f4 out = acc * s;
    if (a.add) out += a.add_scale * ld4<ODD>(a.add + (size_t)row * a.ldadd + coloff);
    if (a.add2) out += a.add2_scale * ld4<ODD>(a.add2 + (size_t)row * a.ldadd2 + coloff);
    if (a.gate) out = vgate<4>(out, ld4<ODD>(a.gate + (size_t)row * a.ldgate + coloff));
    st4<ODD>(a.y + (size_t)row * a.ldy + coloff, out);
  }
}

// Rows that several items add onto start at zero: one wave per row.
__global__ __launch_bounds__(kWave *kWavesPerBlock) void zero_rows_kernel(
    const int *rows, int n, float *y, long long ldy, int width, int n_rows, int *err) {
  const int lane = threadIdx.x & (kWave - 1);
  const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (i >= n) return;
  const int row = rows[i];
  if ((unsigned)row >= (unsigned)n_rows) {
    *err = 1;
    return;
  }
  for (int c = lane; c < width; c += kWave) y[(size_t)row * ldy + c] = 0.f;
}

// d = f(a) - b row by row; f = max(0, .) when relu_a (kernel argument: a scalar branch), else
// the identity.
template <int VEC>
__global__ __launch_bounds__(256) void rows_sub_kernel(const float *a, long long lda,
                                                      const float *b, long long ldb,
                                                      long long n_rows, int width, int relu_a,
                                                      float *d, long long ldd) {
  const int per_row = width / VEC;
  const long long total = n_rows * per_row;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / per_row;
    const int c = (int)(i - r * per_row) * VEC;
    stv<VEC>(d + r * ldd + c,
             vrelu<VEC>(ldv<VEC>(a + r * lda + c), relu_a != 0) - ldv<VEC>(b + r * ldb + c));
  }
}

// ---------------------------------------------------------------- host side
static int env_int(const char *name, int fallback) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : fallback;
}

// Launch tuning.  INCAGG_* environment variables seed it, incagg_set_tuning() changes it at
// run time (used by the A/B benchmarks and by the tests to force either kernel).
// hub_threshold / segment_edges are the defaults of the plan builders (see spmm_plan.hip);
// measured on an ogbn-products-sized batch at width 256 (MI355X), us per forward call:
// threshold/segment 32/1024 94, 64/1024 80, 128/256 89, 128/1024 75, 128/4096 72, 256/1024 79.
struct Tuning {
  int spmm_variant = env_int("INCAGG_SPMM_VARIANT", -1);  // -1 by shape, 0 whole rows, 1 slices
  int hub_threshold = env_int("INCAGG_HUB_THRESHOLD", 128);
  int segment_edges = env_int("INCAGG_SEGMENT_EDGES", 4096);
  int wide_addressing = env_int("INCAGG_WIDE_ADDRESSING", -1);  // 1: never the 24-bit form
  // matrices of more rows than plan_local_rows get a chunk-by-chunk row order (spmm_plan.hip)
  int plan_local_rows = env_int("INCAGG_PLAN_LOCAL_ROWS", 32768);
  int plan_chunk_rows = env_int("INCAGG_PLAN_CHUNK_ROWS", 1024);
  // widest row the one-pass narrow kernel takes (rows wider than one 128-byte slice; 32 = off)
  int narrow_max_width = env_int("INCAGG_NARROW_MAX_WIDTH", 48);
};
static Tuning &tuning() {
  static Tuning t;
  return t;
}

void plan_defaults(int *hub_threshold, int *segment_edges) {
  *hub_threshold = std::max(1, tuning().hub_threshold);
  *segment_edges = std::max(kWave, (tuning().segment_edges + kWave - 1) / kWave * kWave);
}

int plan_chunk_rows(long long n_rows) {
  const Tuning &t = tuning();
  return t.plan_chunk_rows > 0 && n_rows > t.plan_local_rows ? t.plan_chunk_rows : 0;
}

static int bind_plan(SpmmArgs &a, const incagg_spmm_plan *plan) {
  a.order = nullptr;
  a.items = nullptr;
  a.n_hub = a.n_items = 0;
  a.seg = INT_MAX;
  a.multi_rows = nullptr;
  a.n_multi = 0;
  a.comb = nullptr;
  if (!plan) return INCAGG_OK;
  INCAGG_REQUIRE(plan->order != nullptr, "plan without a row order");
  INCAGG_REQUIRE(plan->n_rows == a.n_rows, "plan was built for %lld rows, the call has %d",
                 (long long)plan->n_rows, a.n_rows);
  INCAGG_REQUIRE(plan->n_hub >= 0 && plan->n_hub <= plan->n_rows && plan->n_items >= plan->n_hub &&
                     plan->n_items < INT_MAX && plan->n_multi >= 0 && plan->n_multi <= plan->n_hub,
                 "inconsistent plan counts (hub %lld, items %lld, multi %lld)",
                 (long long)plan->n_hub, (long long)plan->n_items, (long long)plan->n_multi);
  INCAGG_REQUIRE(plan->n_items == 0 || plan->items != nullptr, "plan items missing");
  INCAGG_REQUIRE(plan->n_multi == 0 || plan->multi_rows != nullptr, "plan multi_rows missing");
  INCAGG_REQUIRE(plan->segment_edges >= kWave && plan->segment_edges % kWave == 0,
                 "plan segment length %d is not a positive multiple of 64", plan->segment_edges);
  a.order = plan->order;
  a.items = reinterpret_cast<const int2 *>(plan->items);
  a.n_hub = (int)plan->n_hub;
  a.n_items = (int)plan->n_items;
  a.seg = plan->segment_edges;
  a.multi_rows = plan->multi_rows;
  a.n_multi = (int)plan->n_multi;
  a.edges = reinterpret_cast<const int2 *>(plan->edges);
  INCAGG_REQUIRE(!a.edges || (reinterpret_cast<uintptr_t>(a.edges) & 7) == 0, "plan edges not 8-byte aligned");
  INCAGG_REQUIRE(plan->combine == nullptr || plan->combine_floats >= 0, "negative combine size");
  return INCAGG_OK;
}

static int launch_zero(const SpmmArgs &a, const incagg_spmm_plan *plan, hipStream_t stream) {
  if (!plan || plan->n_multi == 0) return INCAGG_OK;
  const int blocks = (int)((plan->n_multi + kWavesPerBlock - 1) / kWavesPerBlock);
  hipLaunchKernelGGL(zero_rows_kernel, dim3(blocks), dim3(kWave * kWavesPerBlock), 0, stream,
                     plan->multi_rows, (int)plan->n_multi, a.y, a.ldy, a.width, a.n_rows, a.err);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

template <int VEC, int CH, bool INCAGG>
static int launch_rows(SpmmArgs &a, const incagg_spmm_plan *plan, hipStream_t stream) {
  int rc = launch_zero(a, plan, stream);
  if (rc) return rc;
  const long long blocks =
      a.n_items + ((long long)(a.n_rows - a.n_hub) + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks == 0) return INCAGG_OK;
  INCAGG_REQUIRE(blocks < INT_MAX, "grid too large");
  hipLaunchKernelGGL((spmm_rows_kernel<VEC, CH, INCAGG>), dim3((unsigned)blocks),
                     dim3(kWave * kWavesPerBlock), 0, stream, a);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

template <bool INCAGG>
static int launch_sliced(SpmmArgs &a, const incagg_spmm_plan *plan, hipStream_t stream) {
  // with a combine buffer (large enough, 16-byte aligned) the multi-segment rows need no
  // zeroing launch
  if (plan && plan->n_multi > 0 && plan->combine && aligned16(plan->combine) &&
      plan->combine_floats >= plan->n_multi * (int64_t)kCombineSlot && a.width <= kCombineWidth)
    a.comb = plan->combine;
  int rc = a.comb ? INCAGG_OK : launch_zero(a, plan, stream);
  if (rc) return rc;
  const int n_slices = (a.width + kSliceFloats - 1) / kSliceFloats;
  const long long rows_per_block = kWavesPerBlock * 8;
  const long long blocks =
      a.n_items + ((long long)(a.n_rows - a.n_hub) + rows_per_block - 1) / rows_per_block;
  if (blocks == 0) return INCAGG_OK;
  INCAGG_REQUIRE(blocks * n_slices < INT_MAX, "grid too large");
  // 32-bit byte offsets on a scalar base cover a single source matrix of < 4 GiB
  const bool wide = a.x2 != nullptr || (long long)a.n_cols * a.ldx * 4 >= (1ll << 32) ||
                    (INCAGG && (long long)a.n_cols * a.ldin * 4 >= (1ll << 32));
  const dim3 grid((unsigned)(blocks * n_slices)), block(kWave * kWavesPerBlock);
  INCAGG_REQUIRE(a.ldx * 4 < (1ll << 32) && a.ldx2 * 4 < (1ll << 32), "row pitch beyond 4 GiB");
  a.ldb1 = (unsigned)(a.ldx * 4);
  a.ldb2 = (unsigned)(a.ldx2 * 4);
  // 24-bit multiply-add addressing: row ids and pitches below 2^24, each matrix below 4 GiB
  const long long rows2 = a.x2 ? (long long)a.n_cols - a.split : 0;
  const bool fast24 = wide && !INCAGG && tuning().wide_addressing != 1 && a.n_cols < (1 << 24) && a.ldb1 < (1u << 24) &&
                      a.ldb2 < (1u << 24) && (long long)a.split * a.ldb1 < (1ll << 32) &&
                      rows2 * a.ldb2 < (1ll << 32);
  const bool odd = a.width % 4 != 0;
  if constexpr (!INCAGG) {
    // rows of 33..48 floats: whole rows per lane group, one pass over the edge list
    if (a.width > kSliceFloats && a.width <= tuning().narrow_max_width) {
      const int L = (a.width + 3) / 4, R = kWave / L;
      const long long nblocks =
          a.n_items + ((long long)(a.n_rows - a.n_hub) + kWavesPerBlock * R - 1) / (kWavesPerBlock * R);
      INCAGG_REQUIRE(nblocks < INT_MAX, "grid too large");
      const int ngroups = plan && plan_chunk_rows(a.n_rows) > 0 ? 8 : 1;
      const dim3 ngrid((unsigned)nblocks);
#define INCAGG_LAUNCH_NARROW(W, O) \
  hipLaunchKernelGGL((spmm_narrow_kernel<W, O>), ngrid, block, 0, stream, a, L, R, ngroups)
      if (fast24 && odd) INCAGG_LAUNCH_NARROW(2, true);
      else if (fast24) INCAGG_LAUNCH_NARROW(2, false);
      else if (wide && odd) INCAGG_LAUNCH_NARROW(1, true);
      else if (wide) INCAGG_LAUNCH_NARROW(1, false);
      else if (odd) INCAGG_LAUNCH_NARROW(0, true);
      else INCAGG_LAUNCH_NARROW(0, false);
#undef INCAGG_LAUNCH_NARROW
      INCAGG_HIP_TRY(hipGetLastError());
      return INCAGG_OK;
    }
  }
  // (a plan ordered chunk by chunk: see the kernel)
  const int groups = plan && plan_chunk_rows(a.n_rows) > 0 && (n_slices == 1 || n_slices == 2 || n_slices == 4)
                         ? 8 / n_slices : 1;
#define INCAGG_LAUNCH_PLAN(W, O) \
  hipLaunchKernelGGL((spmm_plan_kernel<INCAGG, W, O>), grid, block, 0, stream, a, n_slices, groups)
  if (fast24 && odd) INCAGG_LAUNCH_PLAN(2, true);
  else if (fast24) INCAGG_LAUNCH_PLAN(2, false);
  else if (wide && odd) INCAGG_LAUNCH_PLAN(1, true);
  else if (wide) INCAGG_LAUNCH_PLAN(1, false);
  else if (odd) INCAGG_LAUNCH_PLAN(0, true);
  else INCAGG_LAUNCH_PLAN(0, false);
#undef INCAGG_LAUNCH_PLAN
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}

template <bool INCAGG>
static int dispatch_spmm(SpmmArgs &a, bool vec4, const incagg_spmm_plan *plan,
                         hipStream_t stream) {
  const int w = a.width;
  int rc = bind_plan(a, plan);
  if (rc) return rc;
  // "spmm_variant" tuning key: 0 = whole rows per wave, 1 = feature slices, -1 = slices whenever
  // the rows are 16-byte aligned.  Measured on MI355X, ogbn-products-sized batch, width 256.
  const int mode = tuning().spmm_variant;
  // (forced whole rows: widths beyond the whole-row kernel's reach still take the slices)
  if (vec4 && (mode != 0 || (w % 4 != 0 && w > 256))) return launch_sliced<INCAGG>(a, plan, stream);
  if (vec4 && w % 4 == 0) {
    if (w <= 256) return launch_rows<4, 1, INCAGG>(a, plan, stream);
    if (w <= 512) return launch_rows<4, 2, INCAGG>(a, plan, stream);
    if (w <= 1024) return launch_rows<4, 4, INCAGG>(a, plan, stream);
  } else {
    if (w <= 64) return launch_rows<1, 1, INCAGG>(a, plan, stream);
    if (w <= 128) return launch_rows<1, 2, INCAGG>(a, plan, stream);
    if (w <= 256) return launch_rows<1, 4, INCAGG>(a, plan, stream);
  }
  return fail(INCAGG_E_INVALID, "width %d unsupported (max 1024 when 16-byte aligned, else 256)", w);
}

static bool rows_vec4(const float *p, long long ld) { return !p || (aligned16(p) && ld % 4 == 0); }

}  // namespace incagg

using namespace incagg;

static int check_common(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx,
                        int64_t n_rows, int64_t n_cols, int64_t width, int reduce,
                        const float *add, int64_t ldadd, float *y, int64_t ldy, int64_t nnz) {
  INCAGG_REQUIRE(n_rows >= 0 && n_cols >= 0 && width >= 0, "negative size");
  INCAGG_REQUIRE(n_rows < INT_MAX - 64 && n_cols < INT_MAX && width <= 1024,
                 "size out of range (rows %lld cols %lld width %lld)", (long long)n_rows,
                 (long long)n_cols, (long long)width);
  INCAGG_REQUIRE(reduce == INCAGG_REDUCE_SUM || reduce == INCAGG_REDUCE_MEAN, "bad reduce %d", reduce);
  INCAGG_REQUIRE(nnz >= 0 && nnz < INT_MAX, "nnz %lld out of range", (long long)nnz);
  if (n_rows == 0 || width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(rowptr && y, "null rowptr or output");
  INCAGG_REQUIRE(ldy >= width, "ldy %lld < width %lld", (long long)ldy, (long long)width);
  INCAGG_REQUIRE(!x || ldx >= width, "ldx %lld < width %lld", (long long)ldx, (long long)width);
  INCAGG_REQUIRE(!add || ldadd >= width, "ldadd %lld < width %lld", (long long)ldadd, (long long)width);
  INCAGG_REQUIRE(nnz == 0 || col, "null col with nnz > 0");
  INCAGG_REQUIRE(n_cols == 0 || x, "null x with n_cols > 0");
  INCAGG_REQUIRE((const void *)x != (const void *)y, "y must not alias x");
  INCAGG_REQUIRE(!add || (const void *)add != (const void *)y, "y must not alias add");
  return INCAGG_OK;
}

extern "C" int incagg_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                   const float *x, int64_t ldx, const float *x_halo,
                                   int64_t ld_halo, int64_t n_split, int64_t n_rows,
                                   int64_t n_cols, int64_t width, int reduce, float scale,
                                   const float *add, int64_t ldadd, float add_scale,
                                   int64_t nnz, const incagg_spmm_plan *plan,
                                   const incagg_spmm_fuse *fuse, float *y, int64_t ldy,
                                   void *stream) {
  int rc = check_common(rowptr, col, x, ldx, n_rows, n_cols, width, reduce, add, ldadd, y, ldy, nnz);
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
  a.n_rows = (int)n_rows; a.n_cols = nnz > 0 ? (int)n_cols : 0; a.width = (int)width;
  a.mean = reduce == INCAGG_REDUCE_MEAN; a.scale = scale; a.add_scale = add_scale;
  a.nnz = (int)nnz;
  if (fuse) {
    INCAGG_REQUIRE(!fuse->gate || fuse->ld_gate >= width, "ld_gate %lld < width %lld",
                   (long long)fuse->ld_gate, (long long)width);
    INCAGG_REQUIRE((const void *)fuse->gate != (const void *)y, "y must not alias gate");
    INCAGG_REQUIRE(!fuse->add2 || fuse->ld_add2 >= width, "ld_add2 %lld < width %lld",
                   (long long)fuse->ld_add2, (long long)width);
    INCAGG_REQUIRE(!fuse->add2 || (const void *)fuse->add2 != (const void *)y,
                   "y must not alias add2");
    a.relu_x = fuse->relu_x != 0;
    a.gate = fuse->gate;
    a.ldgate = fuse->ld_gate;
    a.add2 = fuse->add2;
    a.ldadd2 = fuse->ld_add2;
    a.add2_scale = fuse->add2_scale;
  }
  rc = error_flag_address(&a.err);
  if (rc) return rc;
  // (widths >= 4 that are not a multiple of 4 qualify too: only the alignment of the rows counts)
  const bool vec4 = width >= 4 && rows_vec4(x, ldx) && rows_vec4(x_halo, ld_halo) &&
                    rows_vec4(add, ldadd) && rows_vec4(y, ldy) && rows_vec4(a.gate, a.ldgate) &&
                    rows_vec4(a.add2, a.ldadd2);
  return dispatch_spmm<false>(a, vec4, plan, static_cast<hipStream_t>(stream));
}

extern "C" int incagg_spmm_csr_incagg_f32(const int32_t *rowptr, const int32_t *col,
                                          const float *val, const float *x, int64_t ldx,
                                          const float *m_in, int64_t ldin, const float *m_ag,
                                          int64_t ldag, int64_t n_rows, int64_t width, int reduce,
                                          float scale, const float *add, int64_t ldadd,
                                          float add_scale, int64_t nnz,
                                          const incagg_spmm_plan *plan, float *h, int64_t ldh,
                                          void *stream) {
  int rc = check_common(rowptr, col, x, ldx, n_rows, n_rows, width, reduce, add, ldadd, h, ldh, nnz);
  if (rc) return rc;
  if (n_rows == 0 || width == 0) return INCAGG_OK;
  INCAGG_REQUIRE(m_in && m_ag, "null m_in or m_ag");
  INCAGG_REQUIRE(ldin >= width && ldag >= width, "ldin/ldag smaller than width");
  INCAGG_REQUIRE((const void *)m_in != (const void *)h, "h must not alias m_in");
  INCAGG_REQUIRE((const void *)m_ag != (const void *)h, "h must not alias m_ag");
  SpmmArgs a{};
  a.rowptr = rowptr; a.col = col; a.val = val; a.x = x; a.ldx = ldx;
  a.split = (int)n_rows;
  a.m_in = m_in; a.ldin = ldin; a.m_ag = m_ag; a.ldag = ldag;
  a.add = add; a.ldadd = ldadd; a.y = h; a.ldy = ldh;
  a.n_rows = (int)n_rows; a.n_cols = nnz > 0 ? (int)n_rows : 0; a.width = (int)width;
  a.mean = reduce == INCAGG_REDUCE_MEAN; a.scale = scale; a.add_scale = add_scale;
  a.nnz = (int)nnz;
  rc = error_flag_address(&a.err);
  if (rc) return rc;
  const bool vec4 = width >= 4 && rows_vec4(x, ldx) && rows_vec4(m_in, ldin) &&
                    rows_vec4(m_ag, ldag) && rows_vec4(add, ldadd) && rows_vec4(h, ldh);
  return dispatch_spmm<true>(a, vec4, plan, static_cast<hipStream_t>(stream));
}

extern "C" int incagg_set_tuning(const char *key, int value) {
  INCAGG_REQUIRE(key != nullptr, "null key");
  Tuning &t = tuning();
  const std::string k(key);
  if (k == "spmm_variant") t.spmm_variant = value;
  else if (k == "hub_threshold") t.hub_threshold = value;
  else if (k == "segment_edges") t.segment_edges = value;
  else if (k == "wide_addressing") t.wide_addressing = value;
  else if (k == "plan_local_rows") t.plan_local_rows = value;
  else if (k == "plan_chunk_rows") t.plan_chunk_rows = value;
  else if (k == "narrow_max_width") t.narrow_max_width = std::min(std::max(value, kSliceFloats), 64);
  else return fail(INCAGG_E_INVALID, "unknown tuning key '%s'", key);
  return INCAGG_OK;
}

extern "C" int incagg_rows_sub_f32(const float *a, int64_t lda, const float *b, int64_t ldb,
                                   int64_t n_rows, int64_t width, float *d, int64_t ldd,
                                   void *stream) {
  return incagg_rows_sub_relu_f32(a, lda, b, ldb, n_rows, width, 0, d, ldd, stream);
}

extern "C" int incagg_rows_sub_relu_f32(const float *a, int64_t lda, const float *b, int64_t ldb,
                                        int64_t n_rows, int64_t width, int relu_a, float *d,
                                        int64_t ldd, void *stream) {
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
                       (int)width, relu_a, d, ldd);
  else
    hipLaunchKernelGGL((rows_sub_kernel<1>), dim3(blocks), dim3(256), 0, s, a, lda, b, ldb, n_rows,
                       (int)width, relu_a, d, ldd);
  INCAGG_HIP_TRY(hipGetLastError());
  return INCAGG_OK;
}
