// Host-side batch construction: one-hop relabelling of a partition batch and the
// transposed CSR used by the backward aggregation.
//
// The global-id -> local-id map is a dense, epoch-stamped table indexed by node id
// (ids are < n_nodes), kept per thread and reused across calls: a lookup is one
// load and one compare, and starting a new batch costs one increment instead of
// rebuilding a hash table.  Results are defined by the reference's map semantics
// (csrc/cpu/relabel_cpu.cpp:22-36, 56-69): idx[i] -> i with later duplicates
// winning, unseen neighbours -> n_idx + first-seen order.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"

namespace incagg {

struct DenseMap {
  std::vector<int64_t> slot;
  std::vector<uint32_t> stamp;
  uint32_t epoch = 0;

  void begin(int64_t n_nodes) {
    if ((int64_t)slot.size() < n_nodes) {
      slot.resize(n_nodes);
      stamp.assign(n_nodes, 0);
      epoch = 0;
    }
    if (++epoch == 0) {  // wrapped: every stamp is stale again after a clear
      std::fill(stamp.begin(), stamp.end(), 0);
      epoch = 1;
    }
  }
  bool has(int64_t k) const { return stamp[k] == epoch; }
  // returns true when k was not present before
  bool put(int64_t k, int64_t v) {
    const bool fresh = stamp[k] != epoch;
    stamp[k] = epoch;
    slot[k] = v;
    return fresh;
  }
  int64_t get(int64_t k) const { return slot[k]; }
};

static thread_local DenseMap t_map;

static int check_relabel_args(const int64_t *rowptr, int64_t n_nodes, const int64_t *col,
                              const void *value, int64_t value_size, const int64_t *idx,
                              int64_t n_idx, const int64_t *out_rowptr, const int64_t *out_n_id,
                              const int64_t *counts) {
  INCAGG_REQUIRE(rowptr && n_nodes >= 0 && n_idx >= 0, "bad rowptr / sizes");
  INCAGG_REQUIRE(n_idx == 0 || idx, "null idx");
  INCAGG_REQUIRE(out_rowptr && counts, "null output");
  INCAGG_REQUIRE(n_idx == 0 || out_n_id, "null out_n_id");
  INCAGG_REQUIRE(!value || value_size > 0, "value given with value_size %lld", (long long)value_size);
  (void)col;
  return INCAGG_OK;
}

}  // namespace incagg

using namespace incagg;

extern "C" int64_t incagg_relabel_bound(const int64_t *rowptr, int64_t n_nodes, const int64_t *idx,
                                        int64_t n_idx) {
  if (!rowptr || (n_idx > 0 && !idx)) return fail(INCAGG_E_INVALID, "null pointer");
  int64_t ub = 0;
  for (int64_t i = 0; i < n_idx; i++) {
    const int64_t v = idx[i];
    if (v < 0 || v >= n_nodes)
      return fail(INCAGG_E_INDEX, "index %lld at position %lld outside [0, %lld)", (long long)v,
                  (long long)i, (long long)n_nodes);
    ub += rowptr[v + 1] - rowptr[v];
  }
  return ub;
}

extern "C" int incagg_relabel_one_hop(const int64_t *rowptr, int64_t n_nodes, const int64_t *col,
                                      const void *value, int64_t value_size, const int64_t *idx,
                                      int64_t n_idx, int bipartite, int64_t *out_rowptr,
                                      int64_t *out_col, void *out_value, int64_t *out_n_id,
                                      int64_t *counts) {
  int rc = check_relabel_args(rowptr, n_nodes, col, value, value_size, idx, n_idx, out_rowptr,
                              out_n_id, counts);
  if (rc) return rc;
  if (incagg_relabel_bound(rowptr, n_nodes, idx, n_idx) < 0) return INCAGG_E_INDEX;
  DenseMap &map = t_map;
  map.begin(n_nodes);

  out_rowptr[0] = 0;
  int64_t offset = 0;
  for (int64_t i = 0; i < n_idx; i++) {
    const int64_t v = idx[i];
    map.put(v, i);
    offset += rowptr[v + 1] - rowptr[v];
    out_rowptr[i + 1] = offset;
    out_n_id[i] = v;
  }
  INCAGG_REQUIRE(offset == 0 || (col && out_col), "null col with edges present");
  INCAGG_REQUIRE(!value || offset == 0 || out_value, "null out_value");

  int64_t n_new = 0;
  offset = 0;
  const char *vsrc = static_cast<const char *>(value);
  char *vdst = static_cast<char *>(out_value);
  for (int64_t i = 0; i < n_idx; i++) {
    const int64_t v = idx[i];
    const int64_t rs = rowptr[v], re = rowptr[v + 1];
    for (int64_t j = rs; j < re; j++) {
      const int64_t w = col[j];
      if (w < 0 || w >= n_nodes)
        return fail(INCAGG_E_INDEX, "column id %lld of edge %lld outside [0, %lld)", (long long)w,
                    (long long)j, (long long)n_nodes);
      if (!map.has(w)) {
        map.put(w, n_idx + n_new);
        out_n_id[n_idx + n_new] = w;
        n_new++;
      }
      out_col[offset + (j - rs)] = map.get(w);
    }
    if (vsrc && re > rs)  // a row's values are contiguous: copy them in one go
      std::memcpy(vdst + offset * value_size, vsrc + rs * value_size, (size_t)((re - rs) * value_size));
    offset += re - rs;
  }
  int64_t rp_len = n_idx + 1;
  if (!bipartite)
    for (int64_t k = 0; k < n_new; k++) out_rowptr[rp_len++] = offset;
  counts[0] = offset;
  counts[1] = n_idx + n_new;
  counts[2] = rp_len;
  return INCAGG_OK;
}

extern "C" int incagg_relabel_one_hop_within_batch(const int64_t *rowptr, int64_t n_nodes,
                                                   const int64_t *col, const void *value,
                                                   int64_t value_size, const int64_t *idx,
                                                   int64_t n_idx, int bipartite,
                                                   int64_t *out_rowptr, int64_t *out_col,
                                                   void *out_value, int64_t *out_n_id,
                                                   int64_t *counts) {
  int rc = check_relabel_args(rowptr, n_nodes, col, value, value_size, idx, n_idx, out_rowptr,
                              out_n_id, counts);
  if (rc) return rc;
  const int64_t ub = incagg_relabel_bound(rowptr, n_nodes, idx, n_idx);
  if (ub < 0) return INCAGG_E_INDEX;
  INCAGG_REQUIRE(ub == 0 || (col && out_col), "null col with edges present");
  INCAGG_REQUIRE(!value || ub == 0 || out_value, "null out_value");
  DenseMap &map = t_map;
  map.begin(n_nodes);
  int64_t distinct = 0;
  for (int64_t i = 0; i < n_idx; i++) {
    if (map.put(idx[i], i)) distinct++;
    out_n_id[i] = idx[i];
  }
  out_rowptr[0] = 0;
  int64_t offset = 0;
  const char *vsrc = static_cast<const char *>(value);
  char *vdst = static_cast<char *>(out_value);
  for (int64_t i = 0; i < n_idx; i++) {
    const int64_t v = idx[i];
    for (int64_t j = rowptr[v]; j < rowptr[v + 1]; j++) {
      const int64_t w = col[j];
      if (w < 0 || w >= n_nodes)
        return fail(INCAGG_E_INDEX, "column id %lld of edge %lld outside [0, %lld)", (long long)w,
                    (long long)j, (long long)n_nodes);
      if (map.has(w)) {
        out_col[offset] = map.get(w);
        if (vsrc) std::memcpy(vdst + offset * value_size, vsrc + j * value_size, (size_t)value_size);
        offset++;
      }
    }
    out_rowptr[i + 1] = offset;
  }
  int64_t rp_len = n_idx + 1;
  if (!bipartite)
    for (int64_t k = 0; k < distinct; k++) out_rowptr[rp_len++] = offset;
  counts[0] = offset;
  counts[1] = n_idx;
  counts[2] = rp_len;
  return INCAGG_OK;
}

extern "C" int incagg_csr_transpose(const int64_t *rowptr, const int64_t *col, const float *val,
                                    int64_t n_rows, int64_t n_cols, int reduce, int32_t *t_rowptr,
                                    int32_t *t_col, float *t_val) {
  INCAGG_REQUIRE(rowptr && t_rowptr && n_rows >= 0 && n_cols >= 0, "bad arguments");
  INCAGG_REQUIRE(reduce == INCAGG_REDUCE_SUM || reduce == INCAGG_REDUCE_MEAN, "bad reduce %d", reduce);
  const int64_t nnz = rowptr[n_rows];
  INCAGG_REQUIRE(nnz < INT32_MAX && n_rows < INT32_MAX && n_cols < INT32_MAX, "CSR too large for int32");
  INCAGG_REQUIRE(nnz == 0 || (col && t_col && t_val), "null pointer with edges present");
  std::vector<int32_t> fill((size_t)n_cols + 1, 0);
  for (int64_t e = 0; e < nnz; e++) {
    const int64_t c = col[e];
    if (c < 0 || c >= n_cols)
      return fail(INCAGG_E_INDEX, "column id %lld of edge %lld outside [0, %lld)", (long long)c,
                  (long long)e, (long long)n_cols);
    fill[c + 1]++;
  }
  t_rowptr[0] = 0;
  for (int64_t c = 0; c < n_cols; c++) {
    t_rowptr[c + 1] = t_rowptr[c] + fill[c + 1];
    fill[c + 1] = t_rowptr[c];  // next free slot of transposed row c, kept at fill[c + 1]
  }
  for (int64_t r = 0; r < n_rows; r++) {
    const int64_t rs = rowptr[r], re = rowptr[r + 1];
    const float inv = reduce == INCAGG_REDUCE_MEAN ? 1.0f / (float)(re - rs > 1 ? re - rs : 1) : 1.0f;
    for (int64_t e = rs; e < re; e++) {
      const int32_t pos = fill[col[e] + 1]++;
      t_col[pos] = (int32_t)r;
      const float v = val ? val[e] : 1.0f;
      t_val[pos] = reduce == INCAGG_REDUCE_MEAN ? v * inv : v;
    }
  }
  return INCAGG_OK;
}

// ---------------------------------------------------------------- partitioning (host)
// Stand-in for the METIS call of metis.py:31 (the library is not bundled): equal-sized blocks
// grown by breadth-first search, then boundary refinement.  The contract downstream only
// needs a part id per node.

// Fills `num_parts` blocks of ceil(n / num_parts) nodes one after the other by BFS; a new
// search starts from the next unassigned entry of `seeds` (a permutation of the nodes)
// whenever the frontier dries up, and afresh for every block.
extern "C" int incagg_partition_bfs(const int64_t *rowptr, const int64_t *col, int64_t n_nodes,
                                    int64_t num_parts, const int64_t *seeds, int64_t *cluster) {
  INCAGG_REQUIRE(rowptr && cluster && n_nodes >= 0 && num_parts >= 1, "bad arguments");
  INCAGG_REQUIRE(n_nodes == 0 || seeds, "null seeds");
  const int64_t target = (n_nodes + num_parts - 1) / num_parts;
  for (int64_t i = 0; i < n_nodes; i++) cluster[i] = -1;
  std::vector<int64_t> queue;   // FIFO: [head, queue.size())
  queue.reserve((size_t)std::min<int64_t>(n_nodes, 1 << 20));
  size_t head = 0;
  int64_t cursor = 0, part = 0, filled = 0;
  for (int64_t done = 0; done < n_nodes; done++) {
    if (head == queue.size()) {
      queue.clear();
      head = 0;
      while (cursor < n_nodes) {
        const int64_t s = seeds[cursor];
        if (s < 0 || s >= n_nodes) return fail(INCAGG_E_INDEX, "seed %lld out of range", (long long)s);
        if (cluster[s] == -1) break;
        cursor++;
      }
      if (cursor >= n_nodes) return fail(INCAGG_E_INVALID, "seeds is not a permutation of the nodes");
      queue.push_back(seeds[cursor]);
      cluster[seeds[cursor]] = -2;  // queued
    }
    const int64_t v = queue[head++];
    cluster[v] = part;
    filled++;
    if (filled == target && part < num_parts - 1) {
      for (size_t k = head; k < queue.size(); k++) cluster[queue[k]] = -1;  // un-queue
      queue.clear();
      head = 0;
      part++;
      filled = 0;
      continue;
    }
    for (int64_t e = rowptr[v]; e < rowptr[v + 1]; e++) {
      const int64_t w = col[e];
      if (w < 0 || w >= n_nodes) return fail(INCAGG_E_INDEX, "column id %lld out of range", (long long)w);
      if (cluster[w] == -1) {
        cluster[w] = -2;
        queue.push_back(w);
      }
    }
  }
  return INCAGG_OK;
}

// Boundary refinement: in sweeps over the nodes, a node moves to the part that holds most of
// its neighbours when that strictly lowers the number of cut edges and the receiving part
// stays within `max_part_size`.  Returns the number of moves made in `moves` (may be NULL).
extern "C" int incagg_partition_refine(const int64_t *rowptr, const int64_t *col, int64_t n_nodes,
                                       int64_t num_parts, int64_t max_part_size, int64_t sweeps,
                                       int64_t *cluster, int64_t *moves) {
  INCAGG_REQUIRE(rowptr && cluster && n_nodes >= 0 && num_parts >= 1 && sweeps >= 0, "bad arguments");
  std::vector<int64_t> size((size_t)num_parts, 0);
  for (int64_t v = 0; v < n_nodes; v++) {
    if (cluster[v] < 0 || cluster[v] >= num_parts)
      return fail(INCAGG_E_INDEX, "part id %lld of node %lld out of range", (long long)cluster[v], (long long)v);
    size[(size_t)cluster[v]]++;
  }
  std::vector<int64_t> tally((size_t)num_parts, 0), touched;
  int64_t total = 0;
  for (int64_t s = 0; s < sweeps; s++) {
    int64_t moved = 0;
    for (int64_t v = 0; v < n_nodes; v++) {
      touched.clear();
      for (int64_t e = rowptr[v]; e < rowptr[v + 1]; e++) {
        const int64_t w = col[e];
        if (w < 0 || w >= n_nodes) return fail(INCAGG_E_INDEX, "column id %lld out of range", (long long)w);
        if (w == v) continue;
        const int64_t p = cluster[w];
        if (tally[(size_t)p]++ == 0) touched.push_back(p);
      }
      const int64_t home = cluster[v];
      int64_t best = home, best_cnt = tally[(size_t)home];
      for (int64_t p : touched)
        if (p != home && size[(size_t)p] < max_part_size &&
            (tally[(size_t)p] > best_cnt || (tally[(size_t)p] == best_cnt && best != home && p < best))) {
          best = p;
          best_cnt = tally[(size_t)p];
        }
      for (int64_t p : touched) tally[(size_t)p] = 0;
      tally[(size_t)home] = 0;
      if (best != home && size[(size_t)home] > 1) {
        cluster[v] = best;
        size[(size_t)home]--;
        size[(size_t)best]++;
        moved++;
      }
    }
    total += moved;
    if (moved == 0) break;
  }
  if (moves) *moves = total;
  return INCAGG_OK;
}
