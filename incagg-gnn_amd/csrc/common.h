// Shared helpers of the hot-path library (host side).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/incagg_gnn.h"

namespace incagg {

// Records a message for incagg_last_error() of the calling thread and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define INCAGG_HIP_TRY(expr)                                                          \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess)                                                             \
      return ::incagg::fail(INCAGG_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define INCAGG_REQUIRE(cond, ...)                                  \
  do {                                                             \
    if (!(cond)) return ::incagg::fail(INCAGG_E_INVALID, __VA_ARGS__); \
  } while (0)

// Device address of the per-device "bad index seen" flag (see errors.hip).
int error_flag_address(int **out);

// Defaults of the aggregation plan builders (tunable, see incagg_set_tuning).
void plan_defaults(int *hub_threshold, int *segment_edges);
// Rows per chunk of the plan's row order for a matrix of n_rows rows; 0: one descending-length
// order over all rows (see spmm_plan.hip).
int plan_chunk_rows(long long n_rows);

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace incagg
