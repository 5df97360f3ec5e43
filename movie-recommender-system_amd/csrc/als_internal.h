// Internal interfaces between the ALS translation units.
#pragma once

#include "common.h"

namespace mirec {

// Matrix-free wave-per-row solver for rows with at most short_max entries (als_short.hip).  y_rows = rows of
// the gathered table Y (<= 0: unknown, 64-bit addressing).
// short_order / n_short: the rows this kernel solves, longest first (build_short_order), or nullptr.
int launch_als_short(int nt, int64_t y_rows, const int64_t* indptr, const int32_t* indices, const float* data, int64_t row_begin,
                     int64_t nrows, float* X, const float* Y, const float* G, int32_t short_max, int cg_steps,
                     const int32_t* short_order, int64_t n_short, hipStream_t stream);

// Rows [0, nrows) of the range with at most short_max entries, by descending length (ties: ascending row), into
// order[0 .. n_short); tmp must hold short_order_tmp_bytes(nrows) bytes (plan_sort.hip).
size_t short_order_tmp_bytes(int64_t nrows);
int build_short_order(const int64_t* indptr, int64_t row_begin, int64_t nrows, int32_t short_max, int32_t* order,
                      void* tmp, hipStream_t stream);

// Gathered rows the matrix-free kernel can keep in LDS per CSR row at width 32*nt.
int als_short_cache_rows(int nt);

}  // namespace mirec
