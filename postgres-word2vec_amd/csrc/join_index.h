// join_index.h -- the pinned ivpq index (JoinIndex), its workspaces and error buffer: what the handle holds; the kernels and
// the host loop of the kNN-join are in join.h.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/freddy_gpu.h"
#include "wave_topk.h"

namespace freddy {

static constexpr float JOIN_MAX_DIST = 1000.0f;   // ivpq_search_in.c:62
static constexpr int JOIN_CELL_CHUNK = 256;    // cells of a query whose target rows are laid out as ONE index space at a time
static constexpr int JOIN_WG = 256;
static constexpr int JOIN_WAVES = JOIN_WG / 64;

static thread_local char g_join_err[384] = "";
static inline const char* join_error() { return g_join_err; }
static inline int join_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_join_err, sizeof(g_join_err), fmt, ap);
  va_end(ap);
  return code;
}
#define JOIN_HIP(expr)                                                                          \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return join_fail(FREDDY_E_HIP, "%s failed: %s (join.h:%d)", #expr, hipGetErrorString(e_), __LINE__); \
  } while (0)

struct JoinIndex {
  int d = 0, m = 0, K = 0, S = 0, Kc = 0, cells = 0;
  int MP = 0;                 // pitch of a code row in int16: m rounded up to a multiple of 8 (16-byte aligned rows, zero padded)
  int64_t N = 0;
  bool has_vectors = false;
  // device
  float* cbT = nullptr;       // [m][S][K]
  float* coarseT = nullptr;   // [2][d/2][Kc]
  int32_t* ids = nullptr;     // [N]
  int16_t* codes = nullptr;   // [N][MP] -- a lane fetches a row with 16-byte loads
  float* vectors = nullptr;   // [N][d]
  int32_t* cell = nullptr;    // [N] coarse cell of each row
  uint32_t* markbits = nullptr;  // [ceil(N/32)] scratch bitmap of the "id IN (targets)" resolution
  float* d_stats = nullptr;      // [cells+1] the statistics row (device traversal)
  void* h_q = nullptr;           // pinned staging of the query batch (read by a copy kernel: no SDMA ordering hops)
  size_t h_q_cap = 0;
  void* h_sum = nullptr;         // pinned: per-round traversal summaries and result lists come back here
  size_t h_sum_cap = 0;
  // host
  std::vector<int32_t> h_ids, h_cell;
  std::vector<float> h_stats;
  bool ids_affine = false;       // ids[r] == ids[0] + r: O(1) id -> row
  // "fq.id IN (targets)" of the previous call: the same target array (compared word for word) finds its rows resolved,
  // de-duplicated and bucketed by cell already (workspaces 2 and 3 stay as they are); invalidated when rows are appended
  // (kept in ONE pinned block: [cells + 1] bucket offsets, written by the offsets kernel, then the target array, which the mark
  // kernel reads over PCIe -- no SDMA copy in either direction, and the copy the next call is compared with is the staging copy)
  void* h_tl = nullptr;
  size_t h_tl_cap = 0;
  int64_t tl_n = -1;
  int tl_cells = -1;
  bool tl_valid = false;
  // workspaces
  void* w[18] = {nullptr};
  size_t wcap[18] = {0};
  float libm_margin = 1e-5f;     // option join_libm_margin_ppm: device confidences this close to the threshold are re-evaluated by the host's libm
  bool host_traversal = false;   // option join_host_traversal / FREDDY_GPU_JOIN_HOST_TRAVERSAL: every traversal on the host heap
  // stage timers of the last call under the reference's TRACK names (ivpq_search_in.c:234-697)
  freddy_track track;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static inline int join_buf(JoinIndex* j, int slot, size_t bytes, void** out) {
  if (bytes > j->wcap[slot]) {
    if (j->w[slot]) (void)hipFree(j->w[slot]);
    j->w[slot] = nullptr;
    j->wcap[slot] = 0;
    size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&j->w[slot], want) != hipSuccess) return join_fail(FREDDY_E_NOMEM, "workspace allocation of %zu bytes failed", want);
    j->wcap[slot] = want;
  }
  *out = j->w[slot];
  return 0;
}

static inline void join_free(JoinIndex* j) {
  void* ptrs[] = {j->cbT, j->coarseT, j->ids, j->codes, j->vectors, j->cell, j->markbits, j->d_stats};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (j->h_q) (void)hipHostFree(j->h_q);
  if (j->h_sum) (void)hipHostFree(j->h_sum);
  if (j->h_tl) (void)hipHostFree(j->h_tl);
  for (int i = 0; i < 18; ++i) if (j->w[i]) (void)hipFree(j->w[i]);
  if (j->ev0) (void)hipEventDestroy(j->ev0);
  if (j->ev1) (void)hipEventDestroy(j->ev1);
  *j = JoinIndex();
}

static inline std::vector<int16_t> join_pad_codes(const int16_t* codes, int64_t n, int m, int MP) {
  std::vector<int16_t> out((size_t)std::max<int64_t>(n, 1) * MP, 0);
  for (int64_t r = 0; r < n; ++r) memcpy(&out[(size_t)r * MP], codes + (size_t)r * m, sizeof(int16_t) * m);
  return out;
}


}  // namespace freddy
