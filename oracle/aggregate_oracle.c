/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or called
 * from the shipped library; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / the timed CPU baseline.
 *
 * Plain-C restatement of the CSR neighbour aggregation the reference performs
 * at every `adj_t @ x` / `spmm(adj_t, x, reduce)` / conv.propagate call site
 * (models/appnp.py:85,89,152; models/gcn.py:361,386,403; models/gcn2.py:454,477,499;
 * models/graphsage.py:634,683,898,928,952) and of the incremental aggregate
 * `adj_t @ (x - M_in) + M_ag` (models/gcn.py:241,262; models/gcn2.py:255,305;
 * models/appnp.py:122,130; models/graphsage.py:634,683).
 *
 * PARITY UNPINNED at this boundary: the arithmetic itself lives in torch-sparse /
 * torch-scatter / torch-geometric, which are NOT vendored in the reference, have
 * no pinned version there (setup.py:74 lists only ogb and hydra-core) and are
 * not installed here; the reference holds no test or golden vector for it.  The
 * restated algorithm is the published CSR row loop of torch-sparse's CPU spmm:
 * for each row, accumulate value[e] * mat[col[e], :] over the row's edges in
 * storage order in the matrix dtype; 'mean' divides the sum by
 * max(row_end - row_start, 1).
 *
 * All indices are int64 (the reference's layout); data is fp32 row-major with a
 * leading dimension in elements.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { ORACLE_SUM = 0, ORACLE_MEAN = 1 };

void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int oracle_get_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* y[i,:] = reduce_{e in row i} val[e] * x[col[e],:]   (val == NULL -> 1) */
void oracle_spmm_csr_f32(const int64_t *rowptr, const int64_t *col, const float *val,
                         const float *x, int64_t ldx, float *y, int64_t ldy,
                         int64_t n_rows, int64_t width, int reduce) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < n_rows; i++) {
    float *yi = y + i * ldy;
    for (int64_t k = 0; k < width; k++) yi[k] = 0.0f;
    int64_t rs = rowptr[i], re = rowptr[i + 1];
    for (int64_t e = rs; e < re; e++) {
      const float *xr = x + col[e] * ldx;
      if (val) {
        float v = val[e];
        for (int64_t k = 0; k < width; k++) yi[k] += v * xr[k];
      } else {
        for (int64_t k = 0; k < width; k++) yi[k] += xr[k];
      }
    }
    if (reduce == ORACLE_MEAN) {
      int64_t cnt = re - rs;
      float d = (float)(cnt > 1 ? cnt : 1);
      for (int64_t k = 0; k < width; k++) yi[k] /= d;
    }
  }
}

/* Same sum with a double accumulator: the tolerance yardstick for fp32 results. */
void oracle_spmm_csr_f64acc(const int64_t *rowptr, const int64_t *col, const float *val,
                            const float *x, int64_t ldx, double *y, int64_t ldy,
                            int64_t n_rows, int64_t width, int reduce) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < n_rows; i++) {
    double *yi = y + i * ldy;
    for (int64_t k = 0; k < width; k++) yi[k] = 0.0;
    int64_t rs = rowptr[i], re = rowptr[i + 1];
    for (int64_t e = rs; e < re; e++) {
      const float *xr = x + col[e] * ldx;
      double v = val ? (double)val[e] : 1.0;
      for (int64_t k = 0; k < width; k++) yi[k] += v * (double)xr[k];
    }
    if (reduce == ORACLE_MEAN) {
      int64_t cnt = re - rs;
      double d = (double)(cnt > 1 ? cnt : 1);
      for (int64_t k = 0; k < width; k++) yi[k] /= d;
    }
  }
}

/*
 * Incremental aggregate in the reference's operation order
 * (models/gcn2.py:255): d = x - m_in (elementwise, fp32), h = A @ d, h += m_ag.
 * A is n_rows x n_rows (in-batch edges only); scratch holds n_rows*width floats.
 */
void oracle_incagg_f32(const int64_t *rowptr, const int64_t *col, const float *val,
                       const float *x, int64_t ldx, const float *m_in, int64_t ldi,
                       const float *m_ag, int64_t lda, float *h, int64_t ldh,
                       int64_t n_rows, int64_t n_cols, int64_t width, int reduce,
                       float *scratch) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_cols; i++)
    for (int64_t k = 0; k < width; k++)
      scratch[i * width + k] = x[i * ldx + k] - m_in[i * ldi + k];
  oracle_spmm_csr_f32(rowptr, col, val, scratch, width, h, ldh, n_rows, width, reduce);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_rows; i++)
    for (int64_t k = 0; k < width; k++) h[i * ldh + k] += m_ag[i * lda + k];
}

/* History.pull (history.py:33-39): out[k,:] = emb[n_id[k],:] */
void oracle_gather_rows_f32(const float *emb, int64_t lde, const int64_t *n_id,
                            int64_t k_rows, int64_t width, float *out, int64_t ldo) {
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < k_rows; k++)
    memcpy(out + k * ldo, emb + n_id[k] * lde, (size_t)width * sizeof(float));
}

/* History.push by chunks (history.py:60-65): emb[off:off+c] = x[src:src+c] */
void oracle_push_chunks_f32(float *emb, int64_t lde, const float *x, int64_t ldx,
                            const int64_t *offset, const int64_t *count,
                            int64_t n_chunks, int64_t width) {
  int64_t src = 0;
  for (int64_t c = 0; c < n_chunks; c++) {
    for (int64_t r = 0; r < count[c]; r++)
      memcpy(emb + (offset[c] + r) * lde, x + (src + r) * ldx,
             (size_t)width * sizeof(float));
    src += count[c];
  }
}
