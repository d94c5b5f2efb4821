// join.h -- kNN-join (ivpq_search_in) device index and host loop.  (placeholder: filled in below)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/freddy_gpu.h"

namespace freddy {
struct JoinIndex { int dummy = 0; };
static inline const char* join_error() { return "kNN-join is not built yet"; }
static inline int join_pin(JoinIndex*, const freddy_ivpq_desc*, int64_t*) { return FREDDY_E_LIMIT; }
static inline void join_free(JoinIndex*) {}
static inline int join_run(JoinIndex*, hipStream_t, const float*, int, int, const int32_t*, int64_t, int, int, int, int,
                           float, int, int32_t*, float*, int32_t*) { return FREDDY_E_LIMIT; }
}  // namespace freddy
