/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or called
 * from the shipped library; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker.
 *
 * Plain-C restatement of the reference's one-hop relabelling
 * (csrc/cpu/relabel_cpu.cpp:3-108 and :111-214).  Parity is PINNED: this file is
 * checked against outputs of the reference's own C++ (compiled by
 * oracle/build_ref.py) through the fixtures in tests/golden/relabel_*.npz.
 *
 * The reference keys a std::unordered_map<int64,int64> by global node id.  Only
 * its find / insert-or-assign behaviour matters for the result, so the oracle
 * uses a small open-addressing table with the same observable semantics:
 *   - map[idx[i]] = i for i = 0..n-1 in order, later duplicates overwrite
 *     (relabel_cpu.cpp:31-36, :137-140);
 *   - a neighbour not yet in the map gets id  n + (#new ids so far)
 *     in first-seen edge order (relabel_cpu.cpp:56-69).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t *keys;
  int64_t *vals;
  uint64_t mask;
  int64_t size;
} omap_t;

static uint64_t omap_hash(int64_t k) {
  uint64_t x = (uint64_t)k;
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

static int omap_init(omap_t *m, int64_t expected) {
  uint64_t cap = 16;
  while (cap < (uint64_t)expected * 2 + 2) cap <<= 1;
  m->keys = (int64_t *)malloc(cap * sizeof(int64_t));
  m->vals = (int64_t *)malloc(cap * sizeof(int64_t));
  if (!m->keys || !m->vals) return -1;
  for (uint64_t i = 0; i < cap; i++) m->keys[i] = -1; /* ids are >= 0 */
  m->mask = cap - 1;
  m->size = 0;
  return 0;
}

static void omap_free(omap_t *m) { free(m->keys); free(m->vals); }

static int64_t *omap_find(omap_t *m, int64_t k) {
  uint64_t h = omap_hash(k) & m->mask;
  while (m->keys[h] != -1) {
    if (m->keys[h] == k) return &m->vals[h];
    h = (h + 1) & m->mask;
  }
  return NULL;
}

/* insert-or-assign, i.e. operator[] followed by assignment */
static void omap_put(omap_t *m, int64_t k, int64_t v) {
  uint64_t h = omap_hash(k) & m->mask;
  while (m->keys[h] != -1) {
    if (m->keys[h] == k) { m->vals[h] = v; return; }
    h = (h + 1) & m->mask;
  }
  m->keys[h] = k; m->vals[h] = v; m->size++;
}

/*
 * relabel_one_hop (relabel_cpu.cpp:3-108).
 * Buffers supplied by the caller:
 *   out_rowptr : n_idx + 1 + ub   (ub = sum of degrees of idx; worst case of the
 *                                  non-bipartite tail, relabel_cpu.cpp:99-102)
 *   out_col    : ub,   out_value : ub * vsize (or NULL when value == NULL)
 *   out_n_id   : n_idx + ub
 * counts[0] = nnz, counts[1] = len(n_id), counts[2] = len(out_rowptr)
 */
int oracle_relabel_one_hop(const int64_t *rowptr, const int64_t *col,
                           const void *value, int64_t vsize,
                           const int64_t *idx, int64_t n_idx, int bipartite,
                           int64_t *out_rowptr, int64_t *out_col,
                           void *out_value, int64_t *out_n_id,
                           int64_t *counts) {
  int64_t ub = 0;
  for (int64_t i = 0; i < n_idx; i++) ub += rowptr[idx[i] + 1] - rowptr[idx[i]];
  omap_t map;
  if (omap_init(&map, n_idx + ub)) return -1;

  out_rowptr[0] = 0;
  int64_t offset = 0;
  for (int64_t i = 0; i < n_idx; i++) {          /* :31-36 */
    int64_t v = idx[i];
    omap_put(&map, v, i);
    offset += rowptr[v + 1] - rowptr[v];
    out_rowptr[i + 1] = offset;
    out_n_id[i] = v;
  }
  int64_t n_new = 0;
  offset = 0;
  for (int64_t i = 0; i < n_idx; i++) {          /* :51-74 / :79-96 */
    int64_t v = idx[i];
    for (int64_t j = rowptr[v]; j < rowptr[v + 1]; j++) {
      int64_t w = col[j];
      int64_t *hit = omap_find(&map, w);
      if (!hit) {
        int64_t c = n_idx + n_new;
        omap_put(&map, w, c);
        out_n_id[n_idx + n_new] = w;
        n_new++;
        out_col[offset] = c;
      } else {
        out_col[offset] = *hit;
      }
      if (value)
        memcpy((char *)out_value + offset * vsize, (const char *)value + j * vsize,
               (size_t)vsize);
      offset++;
    }
  }
  int64_t rp_len = n_idx + 1;
  if (!bipartite)                                  /* :99-102 */
    for (int64_t k = 0; k < n_new; k++) out_rowptr[rp_len++] = offset;
  counts[0] = offset;
  counts[1] = n_idx + n_new;                       /* :104-105 */
  counts[2] = rp_len;
  omap_free(&map);
  return 0;
}

/*
 * relabel_one_hop_within_batch (relabel_cpu.cpp:111-214): keep an edge only when
 * its neighbour is itself in idx; n_id is returned unchanged (:213).  The
 * non-bipartite tail repeats nnz once per DISTINCT key of the map (:208-211).
 * out_rowptr must hold 2 * n_idx + 1 entries.
 */
int oracle_relabel_one_hop_within_batch(const int64_t *rowptr, const int64_t *col,
                                        const void *value, int64_t vsize,
                                        const int64_t *idx, int64_t n_idx,
                                        int bipartite, int64_t *out_rowptr,
                                        int64_t *out_col, void *out_value,
                                        int64_t *out_n_id, int64_t *counts) {
  omap_t map;
  if (omap_init(&map, n_idx)) return -1;
  for (int64_t i = 0; i < n_idx; i++) {          /* :137-140 */
    omap_put(&map, idx[i], i);
    out_n_id[i] = idx[i];
  }
  out_rowptr[0] = 0;
  int64_t offset = 0;
  for (int64_t i = 0; i < n_idx; i++) {          /* :143-154 count, :169-205 fill */
    int64_t v = idx[i];
    for (int64_t j = rowptr[v]; j < rowptr[v + 1]; j++) {
      int64_t *hit = omap_find(&map, col[j]);
      if (hit) {
        out_col[offset] = *hit;
        if (value)
          memcpy((char *)out_value + offset * vsize,
                 (const char *)value + j * vsize, (size_t)vsize);
        offset++;
      }
    }
    out_rowptr[i + 1] = offset;
  }
  int64_t rp_len = n_idx + 1;
  if (!bipartite)
    for (int64_t k = 0; k < map.size; k++) out_rowptr[rp_len++] = offset;
  counts[0] = offset;
  counts[1] = n_idx;
  counts[2] = rp_len;
  omap_free(&map);
  return 0;
}
