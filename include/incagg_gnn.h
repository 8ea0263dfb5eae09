/*
 * incagg_gnn.h -- C ABI of the MI355X (gfx950) hot-path library.
 *
 * One shared object, libincagg_gnn.so, exports everything below with C linkage,
 * plain pointers and sizes; no framework types cross this boundary.  Each entry
 * point names the reference interface it stands in for (paths are relative to the
 * reference tree).
 *
 * Conventions
 *   - Every function returns 0 on success and a negative INCAGG_E_* code on
 *     failure; incagg_last_error() then holds a message for the calling thread.
 *   - "device pointer" = memory the GPU can address (HBM, or host memory pinned
 *     with hipHostMalloc / hipHostRegister, which gfx950 reads over the host link).
 *   - `stream` is a hipStream_t passed as void*; NULL is the default stream.
 *     Kernel entry points only enqueue work; they never synchronise.
 *   - Matrices are fp32, row-major, with a leading dimension given in ELEMENTS.
 *   - Device-side CSR uses int32 row pointers and column ids; the int64 forms of
 *     the reference's Python API are converted when a batch is uploaded.
 */
#ifndef INCAGG_GNN_H
#define INCAGG_GNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INCAGG_ABI_VERSION 8

enum {
  INCAGG_OK = 0,
  INCAGG_E_INVALID = -1,   /* bad argument (shape, size, alignment, null) */
  INCAGG_E_HIP = -2,       /* a HIP runtime call failed */
  INCAGG_E_NOMEM = -3,     /* host allocation failed */
  INCAGG_E_INDEX = -4,     /* an index was outside its table */
  INCAGG_E_NODEVICE = -5   /* no usable GPU */
};

enum { INCAGG_REDUCE_SUM = 0, INCAGG_REDUCE_MEAN = 1 };

/* Message of the last failure on the calling thread ("" if none). */
const char *incagg_last_error(void);
int incagg_abi_version(void);

/* Number of visible GPUs (0 when there is none); never fails. */
int incagg_device_count(void);

/*
 * Kernels skip any row / column id that is outside its table instead of
 * faulting and latch a flag on the device.  This call reads and clears the flag
 * of the current device: returns 0 if clean, INCAGG_E_INDEX if a kernel since
 * the last call saw a bad index.  It synchronises the device.
 */
int incagg_check_device_errors(void);

/* ------------------------------------------------------------------ aggregation
 * y[i,:] = scale * reduce_{e in row i} val[e] * x[col[e],:]  +  add_scale * add[i,:]
 *
 * Stands in for every `adj_t @ x`, `spmm(adj_t, x, reduce)` and conv.propagate of
 * the reference (models/appnp.py:85,89,152; models/gcn.py:143,164,296,361,386,403;
 * models/gcn2.py:130,142,336,454,477,499; models/graphsage.py:187,272,727,898,928,952),
 * with the APPNP / GCNII mix `(1-alpha)*(A x) + alpha*x_0` (models/appnp.py:85)
 * folded into the epilogue.  The backward pass dX = A^T dY is the same call on
 * the transposed CSR (see incagg_csr_transpose).
 *
 *   rowptr[n_rows+1], col[nnz]  device int32;  val[nnz] device fp32 or NULL (= 1)
 *   x[n_cols, width]  ld = ldx;  add (may be NULL) [n_rows, width] ld = ldadd
 *   x_halo (may be NULL): when given, source rows [n_split, n_cols) are read from
 *     x_halo[c - n_split, :] (ld = ld_halo) instead of x.  This is how a layer consumes
 *     "fresh in-batch rows + pulled out-of-batch History rows" without materialising
 *     their concatenation (the torch.cat of models/base.py:426,451).
 *   reduce: INCAGG_REDUCE_SUM | INCAGG_REDUCE_MEAN (divide by max(row length, 1))
 *   nnz: number of entries of col / val.  A window of a longer row-pointer array may be
 *     passed as `rowptr` (the transposed CSR restricted to some rows): its values then do not
 *     start at 0 and nnz is still the length of the whole col array.  Row pointers outside
 *     [0, nnz] are skipped and reported (incagg_check_device_errors), never followed.
 *   plan (may be NULL): the work plan of this rowptr, see incagg_spmm_plan below.  Without a
 *     plan the rows are processed in storage order and every row by one wave (or one group of 8
 *     lanes), however long it is: correct, but slow on power-law graphs.
 *   y[n_rows, width] ld = ldy; must not alias x, x_halo or add.
 *   width <= 1024 when every row starts 16-byte aligned (every pointer 16-byte aligned, every
 *   ld a multiple of 4; the width itself need not be a multiple of 4), else width <= 256.
 *
 * One launch per call.  Rows cut into several segments are combined with fp32 atomics, in the
 * plan's combine buffer or, without one, on the output row, which a small launch in front then
 * zeroes; every other row is written by exactly one wave with a fixed summation order, i.e.
 * bit-reproducibly.  Two kernels, chosen
 * by alignment: 128-byte feature slices pinned to one XCD each so that every source byte has a
 * single home L2, 8 short rows per wave (rows that start 16-byte aligned: give matrices of odd
 * widths such as 41 or 47 a padded pitch, and every gathered matrix a pitch of an odd number of
 * 128-byte lines); or whole rows per wave (any width / alignment, about 2.5x slower).  The
 * call allocates nothing, keeps no state between calls and may be recorded into a HIP graph.
 */
typedef struct incagg_spmm_plan {
  const int32_t *order;      /* device int32[n_rows]: a permutation of the rows, the long ones first */
  const int32_t *items;      /* device int32[2 * n_items]: (row, segment) of the long rows */
  const int32_t *multi_rows; /* device int32[n_multi]: rows with more than one segment */
  int64_t n_rows;
  int64_t n_hub;             /* rows longer than hub_threshold = the first n_hub of `order` */
  int64_t n_items;
  int64_t n_multi;
  int32_t hub_threshold;
  int32_t segment_edges;     /* edges per item, a multiple of 64 */
  /* Optional: where the segments of the multi-segment rows meet.  Device fp32, 16-byte aligned,
   * n_multi * INCAGG_SPMM_COMBINE_SLOT_FLOATS entries, ALL ZERO when handed over; every launch
   * leaves it all zero again.  With it the 16-byte-aligned kernel needs no zeroing launch in
   * front (one launch per call, always).  It is state of the plan: calls that use the same
   * plan must not overlap in time, and a launch that is aborted half-way leaves it dirty
   * (clear it before the plan is used again).  NULL: rows are zeroed by a launch in front. */
  float *combine;
  int64_t combine_floats;
  /* Optional: the call's edge list once more, packed -- device int32[2 * nnz], entry e =
   * (col[e], bit pattern of val[e], or of 1.0f without values), 8-byte aligned.  The 16-byte-
   * aligned kernel then fetches an edge with one 8-byte load instead of two 4-byte ones (index
   * loads are a fifth of its memory instructions).  It must agree with the col / val arrays
   * of every call the plan is passed to (which other kernels still read); NULL: not used. */
  const int32_t *edges;
} incagg_spmm_plan;
#define INCAGG_SPMM_COMBINE_SLOT_FLOATS 1056 /* 1024 accumulators + 32 per-slice arrival counters */

/* Element-wise work of the surrounding layer folded into the launch (may be NULL):
 *   relu_x   != 0: source rows taken from x (not from x_halo) pass through max(0, .) as they
 *            are gathered -- the activation between two layers (models/gcn2.py:131-135,
 *            models/gcn.py:144-148) applied where its output is consumed instead of in a pass
 *            of its own (History rows are stored activated, see incagg_copy_row_chunks_f32);
 *   gate     (device fp32 [n_rows, width], ld = ld_gate, or NULL): y[i, c] = 0 wherever
 *            gate[i, c] <= 0 -- the backward of that activation, applied to dX = A^T dY
 *            (after both epilogue terms). */
typedef struct incagg_spmm_fuse {
  int32_t relu_x;
  const float *gate;
  int64_t ld_gate;
  /* a second epilogue term: y[i,:] += add2_scale * add2[i,:] (device fp32 [n_rows, width],
   * ld = ld_add2, or NULL).  The two-pass incremental aggregate of a model that mixes its
   * aggregate with x_0 (models/appnp.py:122-130: (1 - alpha) * (A_ib (x - M_in) + M_ag) +
   * alpha * x_0) passes M_ag as `add` and x_0 here: one launch, no pass over the rows behind. */
  const float *add2;
  int64_t ld_add2;
  float add2_scale;
} incagg_spmm_fuse;

int incagg_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                        const float *x, int64_t ldx, const float *x_halo, int64_t ld_halo,
                        int64_t n_split, int64_t n_rows, int64_t n_cols, int64_t width,
                        int reduce, float scale, const float *add, int64_t ldadd,
                        float add_scale, int64_t nnz, const incagg_spmm_plan *plan,
                        const incagg_spmm_fuse *fuse, float *y, int64_t ldy, void *stream);

/*
 * Fused incremental aggregate (the fork's contribution):
 *   h[i,:] = scale * ( reduce_{e in row i} val[e] * (x[c,:] - m_in[c,:]) + m_ag[i,:] )
 *            + add_scale * add[i,:],          c = col[e]
 * Stands in for `adj_t @ (x - M_in) + M_ag` (models/gcn.py:241,262;
 * models/gcn2.py:255,305; models/appnp.py:122,130; models/graphsage.py:634,683)
 * including the two `.clone()`s of the pulled tiles that precede it there.
 * The CSR is n_rows x n_rows (in-batch edges only).
 */
int incagg_spmm_csr_incagg_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                               const float *x, int64_t ldx, const float *m_in,
                               int64_t ldin, const float *m_ag, int64_t ldag,
                               int64_t n_rows, int64_t width, int reduce, float scale,
                               const float *add, int64_t ldadd, float add_scale, int64_t nnz,
                               const incagg_spmm_plan *plan, float *h, int64_t ldh,
                               void *stream);

/*
 * Work plan of an aggregation: a pure function of the row pointers, built once per batch next
 * to the relabelling (it replaces the per-step mask / sort work of models/gcn.py:117-135 and
 * has no other counterpart in the reference, whose SpMM kernels live in torch-sparse).
 *   order[n_rows]        the n_hub rows longer than hub_threshold first, then the others; each
 *                        part by descending length (stable: ties in ascending row id).  When
 *                        incagg_spmm_plan_chunk_rows(n_rows) > 0 (matrices of more than 32,768
 *                        rows: 1,024) each part is ordered chunk by chunk instead -- the rows of
 *                        ids [0, chunk) first, then [chunk, 2 chunk), ..., every chunk by
 *                        descending length: rows of one partition of a batch are neighbours and
 *                        gather mostly from each other, so processing them together keeps their
 *                        sources in the L2.  The kernels accept any permutation whose first n_hub
 *                        entries are the rows that `items` covers.
 *   items[2 * n_items]   for the n_hub rows longer than hub_threshold, in that order, one
 *                        (row, segment) pair per started segment of segment_edges edges
 *   multi_rows[n_multi]  the rows with more than one segment
 * hub_threshold / segment_edges <= 0 select the library defaults (incagg_spmm_plan_defaults).
 * Capacities: incagg_spmm_plan_items_bound(n_rows, nnz, ...) entries suffice for `items`,
 * min(n_rows, nnz / segment_edges) for `multi_rows`.
 * counts (HOST int64[5]): n_hub, n_items, n_multi, hub_threshold used, segment_edges used.
 * incagg_spmm_plan_host:   rowptr and all outputs are host arrays (upload them afterwards).
 * incagg_spmm_plan_device: rowptr and all outputs are device arrays; scratch >=
 *   incagg_spmm_plan_device_scratch_bytes(n_rows); enqueues on `stream`, then waits for it (the
 *   counts come back).  Both produce identical arrays.
 */
int incagg_spmm_plan_defaults(int32_t *hub_threshold, int32_t *segment_edges);
int32_t incagg_spmm_plan_chunk_rows(int64_t n_rows);
int64_t incagg_spmm_plan_items_bound(int64_t n_rows, int64_t nnz, int32_t hub_threshold,
                                     int32_t segment_edges);
int incagg_spmm_plan_host(const int32_t *rowptr, int64_t n_rows, int32_t hub_threshold,
                          int32_t segment_edges, int32_t *order, int32_t *items,
                          int64_t items_cap, int32_t *multi_rows, int64_t multi_cap,
                          int64_t *counts);
int64_t incagg_spmm_plan_device_scratch_bytes(int64_t n_rows);
int incagg_spmm_plan_device(const int32_t *rowptr, int64_t n_rows, int32_t hub_threshold,
                            int32_t segment_edges, void *scratch, int64_t scratch_bytes,
                            int32_t *order, int32_t *items, int64_t items_cap,
                            int32_t *multi_rows, int64_t multi_cap, int64_t *counts,
                            void *stream);

/*
 * incagg_spmm_plan_device without any allocation, workspace argument, host copy or wait.  It
 * has no counterpart in the reference; it is for callers that cannot allocate: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  The work arrays live
 * in a fixed 16 MiB array of the library's code object (one copy per GPU, held until the
 * process ends), which bounds the input: 1 <= n_rows <= incagg_spmm_plan_device_static_max_rows()
 * (262,144); larger inputs are rejected with INCAGG_E_INVALID.  Outputs are those of
 * incagg_spmm_plan_device, element for element; the counts stay on the device:
 *   counts_dev (device int32[6]): n_hub, n_items, n_multi, hub_threshold used, segment_edges
 *   used, error (0 ok, 1 row pointers decrease, 2 a capacity was exceeded).
 * Only enqueues on `stream`.  All calls on one GPU share the fixed array: calls that overlap
 * in time (different streams, replays of graphs holding such calls) race.
 */
int64_t incagg_spmm_plan_device_static_max_rows(void);
int incagg_spmm_plan_device_static(const int32_t *rowptr, int64_t n_rows, int32_t hub_threshold,
                                   int32_t segment_edges, int32_t *order, int32_t *items,
                                   int64_t items_cap, int32_t *multi_rows, int64_t multi_cap,
                                   int32_t *counts_dev, void *stream);

/*
 * Launch tuning of the aggregation (no counterpart in the reference; for benchmarks and
 * tests).  Keys: "spmm_variant" (-1 feature slices whenever the rows are 16-byte aligned, 0
 * always whole rows per wave, 1 as -1), "hub_threshold", "segment_edges" (defaults of the plan
 * builders), "wide_addressing" (1: split-source launches always use 64-bit row offsets, -1: the
 * 24-bit multiply-add form whenever row ids and pitches allow), "plan_local_rows" (matrices of
 * more rows get the chunk-by-chunk row order; 32768), "plan_chunk_rows" (rows per chunk; 1024, 0:
 * never chunked).  Not thread-safe against concurrent aggregation calls.
 */
int incagg_set_tuning(const char *key, int value);

/* d[i,:] = a[i,:] - b[i,:] over [n_rows, width]; the unfused first step of the
 * incremental aggregate (`x - M_in`, models/gcn2.py:255). */
int incagg_rows_sub_f32(const float *a, int64_t lda, const float *b, int64_t ldb,
                        int64_t n_rows, int64_t width, float *d, int64_t ldd,
                        void *stream);
/* d[i,:] = max(0, a[i,:]) - b[i,:] when relu_a, else as incagg_rows_sub_f32: `a` then holds the
 * pre-activations of the layer below, whose relu (models/gcn2.py:131-135) is applied where the
 * incremental aggregate reads its input instead of in a pass of its own. */
int incagg_rows_sub_relu_f32(const float *a, int64_t lda, const float *b, int64_t ldb,
                             int64_t n_rows, int64_t width, int relu_a, float *d, int64_t ldd,
                             void *stream);

/* out[i, j] = sum_s parts[s, i, j] + sum_r a_tail[r, i] * b_tail[r, j]   (parts: S contiguous
 * [m, n] matrices; a_tail [R, m] ld = lda, b_tail [R, n] ld = ldb; out [m, n] ld = ldo).
 * Last step of a weight gradient a^T b over 10^4..10^5 rows, whose bulk runs as one batched
 * GEMM over S row slabs: the slab sum and the few remainder rows in one launch.  No
 * counterpart in the reference (its weight gradients are autograd's own matmuls behind
 * main.py:81); dense-side helper of the training step. */
int incagg_tn_finish_f32(const float *parts, int64_t n_parts, const float *a_tail, int64_t lda,
                         const float *b_tail, int64_t ldb, int64_t n_tail, int64_t m, int64_t n,
                         float *out, int64_t ldo, void *stream);
/* The same for n_batch <= 16 products of one shape in ONE launch (the weight gradients of all
 * layers, finished together at the end of the backward pass): out [n_batch, m, n] contiguous;
 * parts / a_tails / b_tails / lda / ldb: HOST arrays with one entry per product (device
 * pointers to its S contiguous [m, n] partials, to its R remainder rows, and their pitches). */
int incagg_tn_finish_batched_f32(const float *const *parts, int64_t n_batch, int64_t n_parts,
                                 const float *const *a_tails, const int64_t *lda,
                                 const float *const *b_tails, const int64_t *ldb, int64_t n_tail,
                                 int64_t m, int64_t n, float *out, void *stream);

/* Dense-side helpers of the training step around the aggregation (the reference calls
 * torch.nn.functional / torch.optim for them, main.py:60-90; none of its own sources is
 * replaced).  All of them only enqueue on `stream`, allocate nothing and can be recorded into a
 * HIP graph.
 *
 * out[c] = sum_r x[r, c] over a tall [rows, width] matrix (bias gradients), two launches, fixed
 * summation order.  ws: device fp32 scratch of >= incagg_column_sums_ws_floats(rows, width). */
int64_t incagg_column_sums_ws_floats(int64_t rows, int64_t width);
int incagg_column_sums_f32(const float *x, int64_t ldx, int64_t rows, int64_t width, float *ws,
                           int64_t ws_floats, float *out, void *stream);

/* torch.nn.BatchNorm1d in training mode on a tall row-major matrix [rows, width] (the layer
 * outputs of a batch: models/gcn.py:145,181,248, models/gcn2.py:132,161,267,
 * models/graphsage.py:191), as streaming passes over rows as they lie in memory:
 *   forward : mean[c], rstd[c] = 1 / sqrt(biased var + eps) of the batch are written out (the
 *             backward pass needs them), y = (x - mean) rstd gamma + beta (gamma / beta may be
 *             NULL: 1 / 0), running_mean / running_var (may be NULL) move by `momentum` towards
 *             the batch mean / the unbiased batch variance, as the module does;
 *   backward: dbeta = sum_r dy, dgamma = sum_r dy xhat, dx = gamma rstd (dy - dbeta / rows -
 *             xhat dgamma / rows) with xhat = (x - mean) rstd; dx may be dy's memory.
 * rows >= 2, width a multiple of 4 (incagg_batch_norm_supported); rows of x, y, dy, dx and all
 * per-column arrays 16-byte aligned; ws: device fp32 scratch of >=
 * incagg_batch_norm_ws_floats(rows, width).  Fixed summation order. */
int incagg_batch_norm_supported(int64_t rows, int64_t width);
int64_t incagg_batch_norm_ws_floats(int64_t rows, int64_t width);
int incagg_batch_norm_forward_f32(const float *x, int64_t ldx, int64_t rows, int64_t width,
                                  const float *gamma, const float *beta, float eps, float momentum,
                                  float *running_mean, float *running_var, float *ws,
                                  int64_t ws_floats, float *mean, float *rstd, float *y,
                                  int64_t ldy, void *stream);
int incagg_batch_norm_backward_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx,
                                   int64_t rows, int64_t width, const float *gamma,
                                   const float *mean, const float *rstd, float *ws,
                                   int64_t ws_floats, float *dgamma, float *dbeta, float *dx,
                                   int64_t lddx, void *stream);

/* Mean softmax cross entropy of logits [rows, classes] (ld) against int64 class labels -- the
 * value of torch.nn.CrossEntropyLoss() with its defaults (main.py:60,78) -- and its gradient:
 *   forward:  *loss = mean_r (log sum_c exp(logits[r, c]) - logits[r, label[r]]);  keeps the
 *             per-row log-sum-exp in ws (>= incagg_softmax_xent_ws_floats(rows) device floats)
 *   backward: dlogits[r, c] = *dloss * (softmax(logits[r])[c] - [c == label[r]]) / rows, from
 *             the same logits, labels and ws; dloss is a device scalar.
 * A label outside [0, classes) is reported (incagg_check_device_errors); its row counts as
 * loss 0 / gradient 0.  No class weights, no ignore_index, no label smoothing. */
int64_t incagg_softmax_xent_ws_floats(int64_t rows);
int incagg_softmax_xent_forward_f32(const float *logits, int64_t ld, const int64_t *labels,
                                    int64_t rows, int64_t classes, float *ws, int64_t ws_floats,
                                    float *loss, void *stream);
int incagg_softmax_xent_backward_f32(const float *logits, int64_t ld, const int64_t *labels,
                                     int64_t rows, int64_t classes, const float *ws,
                                     const float *dloss, float *dlogits, int64_t ldd,
                                     void *stream);

/* One Adam step (torch.optim.Adam's update, main.py:160-165: no amsgrad, weight_decay added to
 * the gradient) over a list of parameter tensors in ONE launch per 24 tensors (+ a one-wave
 * launch that advances their step counts):
 *   g' = g + weight_decay p;  m += (g' - m)(1 - beta1);  v = beta2 v + (1 - beta2) g'^2;
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps),   t = *step + 1.
 * Every tensor has its own step count (device fp32, starts at 0; as in torch, a tensor that is
 * left out of a call does not age), read and advanced on the device, so a recorded launch keeps
 * counting when replayed.  segs: HOST array; entries with n == 0 are skipped.  A step count may
 * be shared by several segments only when they all fall into the same group of 24 consecutive
 * entries (a shared count is advanced once per group): give every segment its own count. */
typedef struct incagg_adam_segment {
  float *p;        /* parameters, updated in place */
  const float *g;  /* gradients */
  float *m;        /* first moments */
  float *v;        /* second moments */
  float *step;     /* steps taken so far */
  int64_t n;
} incagg_adam_segment;
int incagg_adam_step_f32(const incagg_adam_segment *segs, int64_t n_segs, float lr, float beta1,
                         float beta2, float eps, float weight_decay, void *stream);

/* The dense update of a layer on the matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32: a
 * k-ordered chain of fused multiply-adds, one rounding per product):
 *   y[rows, n] = (accumulate ? y : 0) + bias[n] + x[rows, k] @ W,   W = w ([k, n]) or w^T (w: [n, k],
 *   trans_w); bias optional (NULL: none)
 * and, when `table` is given, table[table_row0 + i, :n] = y[i, :] (relu_push: max(0, y[i, :])):
 * what follows the aggregation in a GCNII / GCN / GraphSAGE layer (models/gcn2.py:257,306
 * conv.forward_after_propagate; models/gcn.py:242-244 with GCNConv's bias; models/graphsage.py:634-642) together with
 * the push of the layer's output into the next layer's table (history.py:41-65,
 * models/gcn2.py:129-138), and with trans_w the input-gradient product dX = dY W^T of its backward
 * pass.  table_only (with a table): y is not written -- it still supplies the values accumulated
 * into -- for callers that only need the stored rows (the layer-wise refresh sweep,
 * models/gcn2.py:433-507, reads every layer's input from the tables).
 * k a positive multiple of 32, n a multiple of 4 up to 256 (incagg_rows_gemm_supported);
 * rows of x and w 16-byte aligned; y must not alias x, the table neither. */
int incagg_rows_gemm_supported(int64_t k, int64_t n);
int incagg_rows_gemm_f32(const float *x, int64_t ldx, const float *w, int64_t ldw, int trans_w,
                         const float *bias, int64_t rows, int64_t k, int64_t n, int accumulate,
                         float *y, int64_t ldy,
                         float *table, int64_t ldt, int64_t table_rows, int64_t table_row0,
                         int relu_push, int table_only, void *stream);

/* x = max(x, 0) in place over [rows, width] (pitch ld): the activation in front of a model's
 * classifier (models/gcn2.py:136), at streaming rate. */
int incagg_relu_rows_f32(float *x, int64_t ld, int64_t rows, int64_t width, void *stream);
/* out[i,:] = scale * (srcs[0][i,:] + ... + srcs[n_srcs-1][i,:]) over [rows, width]; srcs / lds:
 * HOST arrays of n_srcs <= 16 device pointers and their row pitches, consumed before the call
 * returns.  APPNP's K propagation steps all add alpha * x_0 (models/appnp.py:85): the K
 * gradients they hand back for that one term are summed and scaled by one launch instead of K
 * multiplications and K - 1 additions; with one source, a scaled copy at the output's pitch
 * (alpha * x_0 itself).  No counterpart in the reference (autograd accumulates there). */
int incagg_rows_sum_f32(const float *const *srcs, const int64_t *lds, int64_t n_srcs, float scale,
                        int64_t rows, int64_t width, float *out, int64_t ldo, void *stream);
/* dst[i, c] += src[i, c] wherever gate[i, c] > 0: a relu's backward added onto a gradient that
 * exists already (the x_0 branch of GCNII joining the first layer's gradient), one launch. */
int incagg_gate_add_rows_f32(float *dst, int64_t ldd, const float *src, int64_t lds,
                             const float *gate, int64_t ldg, int64_t rows, int64_t width,
                             void *stream);
/* dst[i, c] = gate[i, c] > 0 ? src[i, c] : 0 -- the backward of x.relu_() (every model's
 * activation: models/gcn.py:146,182, models/gcn2.py:133,162, models/graphsage.py:192) on matrices
 * with row pitches of their own; dst may be src.  (torch's kernel for pitched operands moves
 * them at half the rate.) */
int incagg_gate_rows_f32(float *dst, int64_t ldd, const float *src, int64_t lds, const float *gate,
                         int64_t ldg, int64_t rows, int64_t width, void *stream);

/* GCNII layers with separate weights (PyG's GCN2Conv as called at models/gcn2.py:46-48):
 *   out_l = (1 - alpha) (A x) ((1 - beta_l) I + beta_l W1_l) + alpha x_0 ((1 - beta_l) I + beta_l W2_l)
 * The identity parts folded into the weights of all L layers (<= 64) by one launch:
 *   w1f[l] = (1 - alpha)(beta_l W1_l + (1 - beta_l) I)               device fp32 [L, C, C]
 *   w2cat[:, l C:(l+1) C] = alpha (beta_l W2_l + (1 - beta_l) I)      device fp32 [C, L C]
 * ws: device fp32 [2 L, C, C] = W1_0 .. W1_{L-1}, W2_0 .. W2_{L-1}; beta: HOST float[L].
 * The backward call maps gradients of w1f / w2cat onto a gradient of ws (same layout). */
int incagg_gcn2_fold_weights_f32(const float *ws, int64_t n_layers, int64_t c, float alpha,
                                 const float *beta, float *w1f, float *w2cat, void *stream);
int incagg_gcn2_fold_weights_backward_f32(const float *dw1f, const float *dw2cat, int64_t n_layers,
                                          int64_t c, float alpha, const float *beta, float *dws,
                                          void *stream);

/* ------------------------------------------------------------------ History rows
 * out[k,:] = emb[n_id[k],:]            (History.pull, history.py:33-39; the
 * gathered half of read_async, csrc/cuda/async_cuda.cu:95-110).
 * n_id: device int64[k_rows].  emb may be pinned host memory (zero-copy pull).
 */
int incagg_gather_rows_f32(const float *emb, int64_t lde, int64_t emb_rows,
                           const int64_t *n_id, int64_t k_rows, int64_t width,
                           float *out, int64_t ldo, void *stream);

/* emb[n_id[k], :width] = x[k, :width]  (History.push by index, history.py:53-58).
 * Indices are expected to be unique; with duplicates one of the rows wins. */
int incagg_scatter_rows_f32(float *emb, int64_t lde, int64_t emb_rows,
                            const int64_t *n_id, int64_t k_rows, int64_t width,
                            const float *x, int64_t ldx, void *stream);

/*
 * Chunked row copy, both directions of the reference's contiguous-chunk traffic:
 *   push: dst[dst_off[c] + r, :] = src[s + r, :]   with s advancing by count[c]
 *         (History.push by chunks, history.py:60-65; write_async,
 *          csrc/cuda/async_cuda.cu:139-163)
 *   pull: dst[s + r, :] = src[src_off[c] + r, :]   (read_async chunks,
 *          csrc/cuda/async_cuda.cu:68-92)
 * `packed_is_src` != 0 selects push (the packed side is the source).
 * offset[], count[] are HOST int64 arrays of n_chunks entries (as in the
 * reference's API); they are consumed before the call returns.
 * Columns [src_width, dst_width) of every written row are zero-filled, which is
 * the zero-padding to the table width done in models/gcn.py:355-365,
 * models/graphsage.py:888-902, models/appnp.py:254-265.
 * relu != 0: the copied values pass through max(0, .) (the layer's activation folded into
 * the push of its output rows, see incagg_spmm_fuse).
 */
int incagg_copy_row_chunks_f32(float *dst, int64_t ldd, int64_t dst_rows, int64_t dst_width,
                               const float *src, int64_t lds, int64_t src_rows,
                               int64_t src_width, const int64_t *offset,
                               const int64_t *count, int64_t n_chunks, int packed_is_src,
                               int relu, void *stream);

/* ------------------------------------------------------------------ host <-> HBM staging
 * One process-wide worker thread runs queued reads in FIFO order
 * (csrc/thread.h:10-69, csrc/cuda/async_cuda.cu:7-12).
 *
 * incagg_stage_read_async  == read_async  (csrc/async.cpp:21-32, async_cuda.cu:14-115)
 *   On the worker: for every chunk, an H2D hipMemcpyAsync of count[c] rows from
 *   src_host + offset[c] to consecutive rows of dst_dev; then, if n_index > 0, the
 *   rows src_host[index[.]] are gathered on the host into `bounce_host` (pinned)
 *   and copied behind the chunks.  All copies go to `stream`.
 *   Preconditions as the reference asserts them (async_cuda.cu:20-59):
 *   sum(count) + n_index <= bounce_rows and <= dst_rows; chunks inside src;
 *   a non-default stream.  offset/count are copied; `index` must stay valid until
 *   the matching incagg_stage_synchronize() returns.
 *   Unlike the reference it does NOT block the caller on the stream, and copy
 *   errors are reported by incagg_stage_synchronize() instead of being printed.
 *
 * incagg_stage_synchronize == synchronize (csrc/async.cpp:13-19, thread.h:64-69)
 *   Joins exactly ONE queued read, oldest first; no-op when none is queued.
 *   Returns that task's status.  The data is on the device once `stream` has
 *   also been synchronised (pool.py:86-87).
 *
 * incagg_stage_write_async == write_async (csrc/async.cpp:34-41, async_cuda.cu:117-165)
 *   On the caller's thread: for every chunk a D2H hipMemcpyAsync of count[c] rows
 *   from consecutive rows of src_dev to dst_host + offset[c].  Does not block.
 */
int incagg_stage_read_async(const void *src_host, int64_t src_rows, int64_t row_bytes,
                            const int64_t *offset, const int64_t *count, int64_t n_chunks,
                            const int64_t *index, int64_t n_index, void *dst_dev,
                            int64_t dst_rows, void *bounce_host, int64_t bounce_rows,
                            void *stream);
int incagg_stage_synchronize(void);
int incagg_stage_write_async(const void *src_dev, int64_t src_rows, int64_t row_bytes,
                             const int64_t *offset, const int64_t *count, int64_t n_chunks,
                             void *dst_host, int64_t dst_rows, void *stream);
/* Number of reads queued and not yet joined (test / debugging aid). */
int64_t incagg_stage_pending(void);

/* ------------------------------------------------------------------ batch construction (host)
 * Sum of the degrees of idx[]: the capacity the relabel outputs need.
 * Returns < 0 if an index is outside [0, n_nodes). */
int64_t incagg_relabel_bound(const int64_t *rowptr, int64_t n_nodes, const int64_t *idx,
                             int64_t n_idx);

/*
 * relabel_one_hop (csrc/relabel.cpp:10-24, csrc/cpu/relabel_cpu.cpp:3-108).
 * From the global CSR and the in-batch ids idx[], builds the local CSR whose rows
 * are idx[] in order and whose columns are local ids: position in idx[] for
 * in-batch neighbours (a duplicated id maps to its LAST position), otherwise
 * n_idx + k with k counting out-of-batch neighbours in first-seen edge order.
 * `value` (may be NULL) holds nnz items of value_size bytes, copied verbatim.
 * Caller-provided outputs, with ub = incagg_relabel_bound(...):
 *   out_rowptr[n_idx + 1 + ub], out_col[ub], out_value[ub * value_size],
 *   out_n_id[n_idx + ub]
 * counts[0] = nnz, counts[1] = len(out_n_id), counts[2] = len(out_rowptr)
 * (out_rowptr gains one trailing `nnz` per out-of-batch id when !bipartite).
 */
int incagg_relabel_one_hop(const int64_t *rowptr, int64_t n_nodes, const int64_t *col,
                           const void *value, int64_t value_size, const int64_t *idx,
                           int64_t n_idx, int bipartite, int64_t *out_rowptr,
                           int64_t *out_col, void *out_value, int64_t *out_n_id,
                           int64_t *counts);

/*
 * relabel_one_hop_within_batch (csrc/relabel.cpp:27-38, relabel_cpu.cpp:111-214):
 * same, keeping only edges whose neighbour is in idx[]; out_n_id = idx.
 * out_rowptr needs 2 * n_idx + 1 entries (when !bipartite it gains one trailing
 * `nnz` per DISTINCT id of idx[]).
 */
int incagg_relabel_one_hop_within_batch(const int64_t *rowptr, int64_t n_nodes,
                                        const int64_t *col, const void *value,
                                        int64_t value_size, const int64_t *idx,
                                        int64_t n_idx, int bipartite, int64_t *out_rowptr,
                                        int64_t *out_col, void *out_value,
                                        int64_t *out_n_id, int64_t *counts);

/*
 * Host-side CSR -> device-format transposed CSR, built once per batch so that the
 * backward pass dX = A^T dY (autograd through every aggregation call site,
 * main.py:81) is a plain row-parallel aggregation without atomics.
 *   in : rowptr[n_rows+1], col[nnz] int64; val[nnz] fp32 or NULL
 *   out: t_rowptr[n_cols+1], t_col[nnz] int32 (t_col = source row id, ascending
 *        inside every transposed row), t_val[nnz] fp32 = val (or 1), divided by
 *        max(len(row), 1) when reduce == INCAGG_REDUCE_MEAN.
 */
int incagg_csr_transpose(const int64_t *rowptr, const int64_t *col, const float *val,
                         int64_t n_rows, int64_t n_cols, int reduce, int32_t *t_rowptr,
                         int32_t *t_col, float *t_val);

/* ------------------------------------------------------------------ partitioning (host)
 * Stand-in for the METIS call of metis.py:31 (`torch.ops.torch_sparse.partition`, a library
 * that is not bundled): a part id per node, from which metis.py:32-33 derives (perm, ptr).
 * incagg_partition_bfs grows num_parts blocks of ceil(n / num_parts) nodes by breadth-first
 * search, restarting from the next unassigned entry of `seeds` (host int64[n_nodes], a
 * permutation); incagg_partition_refine then moves boundary nodes to the part holding most
 * of their neighbours while that lowers the edge cut and no part exceeds max_part_size.
 * All arrays are host int64; the CSR is expected to be symmetric.
 */
int incagg_partition_bfs(const int64_t *rowptr, const int64_t *col, int64_t n_nodes,
                         int64_t num_parts, const int64_t *seeds, int64_t *cluster);
int incagg_partition_refine(const int64_t *rowptr, const int64_t *col, int64_t n_nodes,
                            int64_t num_parts, int64_t max_part_size, int64_t sweeps,
                            int64_t *cluster, int64_t *moves);

/* ------------------------------------------------------------------ batch build on the device
 * The same relabelling computed on the GPU from an HBM-resident global CSR, emitting the
 * device format the aggregation consumes directly (int32 rowptr / col, fp32 values, int64
 * n_id).  n_id order, local column ids, row pointers and values are identical to
 * incagg_relabel_one_hop / _within_batch, i.e. to csrc/cpu/relabel_cpu.cpp:3-108,111-214
 * (first-seen order of a serial scan over the batch's edges, later duplicates of idx win);
 * the row pointers are always the bipartite ones (n_idx + 1 entries): a caller that wants the
 * square form pads them with the last value (counts[1] - n_idx times, or counts[2] - n_idx
 * times for the within-batch variant, as relabel_cpu.cpp:98-106,204-212 does).
 *
 *   rowptr[n_nodes+1], col[nnz] device int64; value device fp32[nnz] or NULL; idx device int64
 *   map_a, map_b: device int32[n_nodes] work tables owned by the caller, every entry -1 on
 *     entry; they are -1 again when the call returns (map_b may be NULL for within_batch)
 *   scratch: device bytes, at least incagg_relabel_device_scratch_bytes(n_idx, edge_capacity)
 *   edge_capacity / node_capacity: sizes of out_col / out_value and of out_n_id;
 *     incagg_relabel_bound_device gives the exact edge count of the full relabelling, which is
 *     the capacity BOTH variants need (within_batch writes fewer edges, but its work arrays hold
 *     one entry per edge of the batch rows); n_idx + that bound always suffices for the nodes
 *   counts (HOST int64[3]): edges written, nodes in n_id, distinct ids in idx
 * Both calls enqueue on `stream` and then wait for it (the sizes come back to the host).
 */
int64_t incagg_relabel_device_scratch_bytes(int64_t n_idx, int64_t edge_capacity);
int incagg_relabel_bound_device(const int64_t *rowptr, int64_t n_nodes, const int64_t *idx,
                                int64_t n_idx, void *scratch, int64_t scratch_bytes,
                                int64_t *bound, void *stream);
int incagg_relabel_one_hop_device(const int64_t *rowptr, int64_t n_nodes, const int64_t *col,
                                  const float *value, const int64_t *idx, int64_t n_idx,
                                  int within_batch, int64_t edge_capacity, int64_t node_capacity,
                                  int32_t *map_a, int32_t *map_b, void *scratch,
                                  int64_t scratch_bytes, int32_t *out_rowptr, int32_t *out_col,
                                  float *out_value, int64_t *out_n_id, int64_t *counts,
                                  void *stream);

/* incagg_csr_transpose on the device (same output, element for element: transposed row c
 * lists its edges in the order of the original rows).  All pointers are device memory;
 * scratch >= incagg_csr_transpose_device_scratch_bytes(n_cols, nnz).  Does not synchronise. */
int64_t incagg_csr_transpose_device_scratch_bytes(int64_t n_cols, int64_t nnz);
int incagg_csr_transpose_device(const int32_t *rowptr, const int32_t *col, const float *val,
                                int64_t n_rows, int64_t n_cols, int64_t nnz, int reduce,
                                void *scratch, int64_t scratch_bytes, int32_t *t_rowptr,
                                int32_t *t_col, float *t_val, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* INCAGG_GNN_H */
