// bigk.h -- result lists of more than 512 entries (k <= 4096): the reference allocates k entries for any k (freddy.c:66,89,
// 236,258); the selection kernels of this build hold at most 1024 keys per wave.
//
// The tie contract needs a query's 2k smallest (distance, scan position) keys and the guarded insertion
// (updateTopK + "dist < maxDist", index_utils.c:19-33, freddy.c:128-131) replayed over them in scan order on top of the
// carried list.  Here:
//   1. the 2k smallest keys are selected 1024 at a time: pass p re-runs the generic scan (adc_scan_kernel<.., FLOOR>) over the
//      same rows, admitting only keys ABOVE the largest key pass p - 1 selected (keys are unique: the position is part of
//      them), and merge_select_kernel merges the chunks' lists into the pass's 1024 keys;
//   2. bigk_replay_kernel computes what the replay leaves behind in CLOSED FORM instead of performing up to 8192 insertions
//      into a list of 4096 (tests/test_oracle.py test_big_k_closed_form_equals_replay checks the form against the insertion
//      loop on tie-heavy streams, with and without a carried list):
//        * the carried list is a prefix of the stream: feeding its entries last slot first into an empty list rebuilds it
//          (a new entry goes in FRONT of equal ones), so entry s "arrived" at time k - 1 - s, every new row after all of them
//          (arrival = 4096 + position);
//        * S = rows below the sentinel, d* = the k-th smallest distance in S.  Every row with d < d* ends up in the list; a row
//          with d = d* is ACCEPTED iff fewer than k rows of {d <= d*} arrived before it (the list's last entry is still
//          larger); once k such rows are in, every later row with d < d* pushes out the list's last entry = the EARLIEST
//          accepted row at d*.  With A accepted ties and lt rows below d*: the first e = A + lt - k ties are pushed out again;
//        * order: ascending distance, equal distances by DEscending arrival.
// One workgroup per query sorts carried list + selected keys (<= 12288 keys, bitonic in LDS) and writes the list.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "wave_topk.h"

namespace freddy {

static constexpr int BIGK_KMAX = 4096;   // largest k
static constexpr int BIGK_PASS = 1024;   // keys a selection pass adds
static constexpr int BIGK_T = 1024;      // threads of the replay workgroup

struct SelectArgs {
  const u64* part;            // [n_active][parts_per_query][BIGK_PASS] ascending lists of the pass's scan
  const int32_t* active;      // [n_active] or NULL
  u64* sel;                   // [n_active][nsel]: this pass writes [pass * BIGK_PASS, + BIGK_PASS)
  u64* floor;                 // [Q] <- the largest key selected so far (KEY_INF: the query's rows are exhausted)
  int parts_per_query, nsel, pass;
};

// one wave per active query: the BIGK_PASS smallest keys of the query's chunk lists, ascending
static __global__ __launch_bounds__(64) void merge_select_kernel(SelectArgs a) {
  constexpr int V = BIGK_PASS / 64;
  const int x = blockIdx.x, lane = threadIdx.x;
  const int q = a.active ? a.active[x] : x;
  u64 acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = KEY_INF;
  const u64* src = a.part + (size_t)x * a.parts_per_query * BIGK_PASS;
  u64 tau = KEY_INF;
  for (int p = 0; p < a.parts_per_query; ++p) {
    for (int r0 = 0; r0 < BIGK_PASS; r0 += 4 * 64) {   // (four rows of a list requested together)
      u64 rows[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) rows[u] = src[(size_t)p * BIGK_PASS + r0 + u * 64 + lane];
      bool list_done = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        u64 key = rows[u];
        if (!(key < tau)) key = KEY_INF;
        if (list_done || __ballot(key != KEY_INF) == 0ull) { list_done = true; continue; }   // ascending: nothing further down either
        wave_topk_absorb_sorted<V>(acc, key);   // (a prefix of an ascending row, the rest KEY_INF: still ascending)
        tau = wave_topk_at<V>(acc, BIGK_PASS - 1);
      }
      if (list_done) break;
    }
  }
  u64* out = a.sel + (size_t)x * a.nsel + (size_t)a.pass * BIGK_PASS;
#pragma unroll
  for (int v = 0; v < V; ++v) out[v * 64 + lane] = acc[v];
  if (lane == 0) a.floor[q] = tau;   // (= acc[BIGK_PASS - 1]; KEY_INF when the pass found fewer keys: later passes admit nothing)
}

struct BigkArgs {
  const u64* sel;              // [n_active][nsel] the selected keys (any order; KEY_INF = none)
  const int32_t* active;       // [n_active] or NULL
  const int32_t* pos_to_id;    // NULL: position is the id
  const int32_t* round_rows;   // as MergeArgs
  const int32_t* cand_count;
  int32_t* out_ids;            // [Q][k] carried list and result
  float* out_dist;
  int32_t* found;
  int32_t* next_active;
  int32_t* n_next;
  int32_t* status;
  int n_active, nsel, npad, k, found_rule, first_round;
  float sentinel;
};

static inline size_t bigk_lds_bytes(int npad, int k) { return (size_t)npad * sizeof(u64) + (size_t)k * sizeof(int32_t); }

static __global__ __launch_bounds__(BIGK_T) void bigk_replay_kernel(BigkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bigk_smem[];
  u64* keys = reinterpret_cast<u64*>(bigk_smem);                    // [npad] (distance bits, arrival)
  int32_t* old_ids = reinterpret_cast<int32_t*>(keys + a.npad);     // [k] ids of the carried list
  __shared__ int sh_nv, sh_A;
  const int x = blockIdx.x, tid = threadIdx.x;
  const int q = a.active ? a.active[x] : x;
  const int k = a.k, n = a.npad;
  if (tid == 0) { sh_nv = 0; sh_A = 0; }
  for (int i = tid; i < n; i += BIGK_T) {
    u64 key = KEY_INF;
    if (i < k) {
      if (!a.first_round) {
        const float d = a.out_dist[(size_t)q * k + i];
        old_ids[i] = a.out_ids[(size_t)q * k + i];
        if (d < a.sentinel) key = ((u64)__float_as_uint(d) << 32) | (u64)(uint32_t)(k - 1 - i);
      }
    } else if (i - k < a.nsel) {
      const u64 s = a.sel[(size_t)x * a.nsel + (size_t)(i - k)];
      if (s != KEY_INF) key = s + (u64)BIGK_KMAX;   // (positions are below 2^31 - 64: no carry into the distance)
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (n >> 1); t += BIGK_T) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const u64 u = keys[lo], v = keys[hi];
        if ((u > v) == up) { keys[lo] = v; keys[hi] = u; }
      }
      __syncthreads();
    }
  }
  // S = keys[0, nv): ascending (distance, arrival), all below the sentinel
  for (int i = tid; i < n; i += BIGK_T)
    if (keys[i] != KEY_INF && (i + 1 == n || keys[i + 1] == KEY_INF)) sh_nv = i + 1;
  __syncthreads();
  const int nv = sh_nv;
  auto dist_of = [&](int i) { return (uint32_t)(keys[i] >> 32); };
  auto first_at_least = [&](uint32_t db) {   // the first index in [0, nv) whose distance bits are >= db
    int lo = 0, hi = nv;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (dist_of(mid) < db) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  auto first_above = [&](uint32_t db) {
    int lo = 0, hi = nv;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (dist_of(mid) <= db) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  int lt = nv, A = 0;
  if (nv > k) {
    const uint32_t ds = dist_of(k - 1);
    lt = first_at_least(ds);
    const int ntie = first_above(ds) - lt;
    const int nt = ntie < k ? ntie : k;   // (a later tie has at least k rows of S before it)
    for (int j = tid; j < nt; j += BIGK_T) {
      const uint32_t aj = (uint32_t)keys[lt + j];
      int c = j;
      for (int i = 0; i < lt && c < k; ++i) c += ((uint32_t)keys[i] < aj) ? 1 : 0;
      if (c < k) atomicAdd(&sh_A, 1);
    }
    __syncthreads();
    A = sh_A;
  }
  const int e = (nv > k) ? A + lt - k : 0;
  for (int o = tid; o < k; o += BIGK_T) {
    int src = -1;
    if (o < lt) {   // (nv <= k: lt = nv, the slots behind stay empty) the run of equal distances o lies in, reversed
      const uint32_t db = dist_of(o);
      src = first_at_least(db) + first_above(db) - 1 - o;
    } else if (nv > k) {
      src = lt + (A - 1 - (o - lt));   // ties e .. A - 1, the latest first
      if (src < lt + e) src = -1;      // (cannot happen: A - e = k - lt)
    }
    int32_t id = -1;
    float d = a.sentinel;
    if (src >= 0) {
      const u64 key = keys[src];
      const uint32_t arr = (uint32_t)key;
      d = __uint_as_float((uint32_t)(key >> 32));
      if (arr < (uint32_t)BIGK_KMAX) id = old_ids[k - 1 - (int)arr];
      else id = a.pos_to_id ? a.pos_to_id[arr - (uint32_t)BIGK_KMAX] : (int32_t)(arr - (uint32_t)BIGK_KMAX);
    }
    a.out_ids[(size_t)q * k + o] = id;
    a.out_dist[(size_t)q * k + o] = d;
  }
  if (tid == 0) {   // "found" (freddy.c:377 rows rule, :971 accepted rule), as merge_replay_kernel
    int f = (a.first_round || !a.found) ? 0 : a.found[q];
    const int rows = a.round_rows ? a.round_rows[x] : 0;
    f += (a.found_rule == 1 && a.cand_count) ? a.cand_count[q] : (rows > 0 ? rows : 0);
    if (a.found) a.found[q] = f;
    if (a.next_active && f < k && rows >= 0) {
      const int slot = atomicAdd(a.n_next, 1);
      a.next_active[slot] = q;
      if (a.status) a.status[0] = 1;
    }
  }
}

}  // namespace freddy
