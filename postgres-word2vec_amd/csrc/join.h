// join.h -- kNN-join: the body of ivpq_search_in (ivpq_search_in.c:61-699) on gfx950.
//
// Split of work
//   GPU  sub_dist_kernel : the 2 x coarse_codes sub-distances of every query to the
//                          multi-index centroids (index_utils.c:297-305), fp32 sequential.
//   GPU  join_traverse_kernel : multi-index cell selection with statistics (index_utils.c:252-443)
//        for up to 1024 cells, one wave per query.  The reference's heap pops the cells in ascending
//        d0[c0] + d1[c1]; with no two equal keys among the cells a query takes (and the first it leaves)
//        that sequence IS the sorted order, so the wave sorts the 1024 keys, forms the running sum of the
//        cells' statistics in that order (sequential binary32 adds, as :424) and finds the first count n
//        whose getConfidenceHyp (:673-682) reaches the confidence.  The HOST's libm keeps the last word:
//        it evaluates the reference's expression at the stop the device proposes and one step before it
//        (the confidence is non-decreasing in the running sum); a query whose check fails, or whose
//        prefix holds two equal keys (the heap's order among equals is history-dependent), is traversed
//        on the host exactly as before (join_select_cells).  More than 1024 cells: the host path.
//   host "WHERE coarse_id IN (...) AND id IN (...)" (ivpq_search_in.c:352-401): targets are
//        bucketed by cell once per call; a query's candidates are the buckets of its cells.
//   GPU  join_query_kernel : one workgroup per query: LUT (index_utils.c:445-455, pair LUT
//        :457-475) in LDS, ADC / exact distances of the query's candidates, selection by
//        (distance, row) key, post verification (index_utils.c:477-498) and the final
//        insertion replay -- everything that touches a distance.
//   host alpha-doubling retry loop (ivpq_search_in.c:299-684), target-count skip rule
//        (:553-557), re-queue of queries whose list is still at MAX_DIST (:639-669).
//
// Closed forms used on the device (proved in DESIGN.md, checked against the oracle's
// literal restatement in tests/):
//   method 0/1: final list = insertion replay, in ascending id, over the 2k smallest
//               (distance, id) keys of the query's candidates.
//   method 2:   the reference's append-buffer-and-qsort (updateTopKPVFast/reorderTopKPV,
//               ivpq_search_in.c:40-57) keeps exactly the k*pvf smallest (ADC distance,
//               arrival) keys, ascending -- given a stable qsort (glibc <= 2.36) -- and
//               postverify walks them in that order.
#pragma once

#include "join_index.h"

namespace freddy {


template <class T>
static inline int join_upload(T** dst, const T* src, size_t n, int64_t* bytes) {
  size_t sz = sizeof(T) * (n ? n : 1);
  if (hipMalloc((void**)dst, sz) != hipSuccess) return -1;
  if (n && hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice) != hipSuccess) return -1;
  *bytes += (int64_t)sz;
  return 0;
}

static inline int join_pin(JoinIndex* j, const freddy_ivpq_desc* t, int64_t* bytes) {
  j->d = t->d; j->m = t->m; j->K = t->K; j->S = t->d / t->m; j->Kc = t->coarse_codes;
  j->cells = t->coarse_codes * t->coarse_codes;
  j->MP = (t->m + 7) & ~7;
  j->N = t->N;
  j->has_vectors = t->vectors != nullptr;
  if (t->K > 32767) return join_fail(FREDDY_E_LIMIT, "K=%d does not fit an int16 code", t->K);
  for (int64_t r = 0; r < t->N; ++r) {
    if (r && t->ids[r] <= t->ids[r - 1]) return join_fail(FREDDY_E_ARG, "ids must be strictly ascending (row %lld)", (long long)r);
    if (t->coarse_id[r] < 0 || t->coarse_id[r] >= j->cells) return join_fail(FREDDY_E_ARG, "coarse_id %d out of range at row %lld", t->coarse_id[r], (long long)r);
    for (int l = 0; l < t->m; ++l) {
      const int c = t->codes[(size_t)r * t->m + l];
      if (c < 0 || c >= t->K) return join_fail(FREDDY_E_ARG, "code %d out of range at row %lld", c, (long long)r);
    }
  }
  const int m = j->m, K = j->K, S = j->S, half = j->d / 2, Kc = j->Kc;
  std::vector<float> cbT((size_t)m * S * K);
  for (int p = 0; p < m; ++p)
    for (int c = 0; c < K; ++c)
      for (int i = 0; i < S; ++i) cbT[((size_t)p * S + i) * K + c] = t->codebook[((size_t)p * K + c) * S + i];
  std::vector<float> cqT((size_t)2 * half * Kc);
  for (int p = 0; p < 2; ++p)
    for (int c = 0; c < Kc; ++c)
      for (int i = 0; i < half; ++i) cqT[((size_t)p * half + i) * Kc + c] = t->coarse[((size_t)p * Kc + c) * half + i];
  if (join_upload(&j->cbT, cbT.data(), cbT.size(), bytes) || join_upload(&j->coarseT, cqT.data(), cqT.size(), bytes) ||
      join_upload(&j->ids, t->ids, (size_t)t->N, bytes) || join_upload(&j->codes, join_pad_codes(t->codes, t->N, m, j->MP).data(), (size_t)t->N * j->MP, bytes) ||
      join_upload(&j->cell, t->coarse_id, (size_t)t->N, bytes) || join_upload(&j->d_stats, t->stats, (size_t)j->cells + 1, bytes) ||
      (t->vectors && join_upload(&j->vectors, t->vectors, (size_t)t->N * t->d, bytes)))
    return join_fail(FREDDY_E_NOMEM, "device allocation failed while pinning the ivpq tables");
  j->h_ids.assign(t->ids, t->ids + t->N);
  j->h_cell.assign(t->coarse_id, t->coarse_id + t->N);
  j->h_stats.assign(t->stats, t->stats + j->cells + 1);
  if (hipMalloc((void**)&j->markbits, sizeof(uint32_t) * (size_t)((t->N + 31) / 32 + 1)) != hipSuccess)
    return join_fail(FREDDY_E_NOMEM, "device allocation failed while pinning the ivpq tables");
  j->ids_affine = t->N > 0 && (int64_t)t->ids[t->N - 1] - t->ids[0] == t->N - 1;   // strictly ascending => consecutive
  return 0;
}

// pinned host memory -> device (the queries of a call; pinned memory is mapped into the device's address space)
// Cell lists of the (few) queries whose traversal the host had to do (equal keys: the reference's heap order is history-dependent)
// into the rows the device traversal writes for everybody else -- row q of qcells[Q][cells], qcell_cnt[q] -- so that ONE join
// launch serves all queries of a round.  rows: [n][1 + cells] in mapped host memory (count, cells).
__global__ __launch_bounds__(256) void join_fb_rows_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ scan_q,
                                                          int32_t* __restrict__ qcells, int32_t* __restrict__ qcell_cnt, int cells) {
  const int x = blockIdx.x, q = scan_q[x];
  const int32_t* r = rows + (size_t)x * (cells + 1);
  const int n = r[0];
  if (threadIdx.x == 0) qcell_cnt[q] = n;
  for (int i = threadIdx.x; i < n; i += 256) qcells[(size_t)q * cells + i] = r[1 + i];
}
__global__ __launch_bounds__(256) void join_copy_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------
// sub-distances of every query to the multi-index centroids       index_utils.c:297-305
// out[q][pos][code]; lane <-> code (coalesced centroid reads), query half-vector via scalar cache
// ---------------------------------------------------------------------------------------
// copy_out != NULL: `queries` is the pinned staging block (mapped host memory) -- the workgroup pulls its half vector over
// PCIe ONCE (16-byte loads), keeps it in LDS, and also writes it to the device copy the join kernel reads: the query batch
// crosses PCIe inside this kernel, piece by piece behind the host's staging copy (no separate copy kernels, and the
// sub-distances of a piece are done when its bytes have arrived).  q0: first query of the launch.
__global__ __launch_bounds__(64) void sub_dist_kernel(const float* __restrict__ queries,
                                                     const float* __restrict__ coarseT,
                                                     float* __restrict__ out, int d, int Kc,
                                                     float* __restrict__ copy_out = nullptr, int q0 = 0) {
  __shared__ __attribute__((aligned(16))) float qh[512];
  const int q = q0 + blockIdx.x, pos = blockIdx.y;
  const int half = d / 2;
  const float* qv = queries + (size_t)q * d + (size_t)pos * half;
  const bool staged = copy_out != nullptr && half <= 512;
  if (staged) {
    if ((half & 3) == 0 && (((size_t)q * d + (size_t)pos * half) & 3) == 0) {
      const int n4 = half >> 2;
      for (int i = threadIdx.x; i < n4; i += 64) {
        const float4 v = reinterpret_cast<const float4*>(qv)[i];
        reinterpret_cast<float4*>(qh)[i] = v;
        reinterpret_cast<float4*>(copy_out + (size_t)q * d + (size_t)pos * half)[i] = v;
      }
    } else {
      for (int i = threadIdx.x; i < half; i += 64) { const float v = qv[i]; qh[i] = v; copy_out[(size_t)q * d + (size_t)pos * half + i] = v; }
    }
    __syncthreads();
    qv = qh;
  }
  for (int c = threadIdx.x; c < Kc; c += 64) {
    float acc = 0.0f;
    // (the sum is sequential -- squareDistance's order -- but the loads are not: one at a time, each waited for, the kernel was
    //  150 dependent round trips long: 53 us for 5 000 queries; fifteen in flight per batch)
    constexpr int NB = 15;
    int i = 0;
    for (; i + NB <= half; i += NB) {
      float cv[NB], qq[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) { cv[u] = coarseT[((size_t)pos * half + i + u) * Kc + c]; qq[u] = qv[i + u]; }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const float t = qq[u] - cv[u];
        const float p = t * t;
        acc = acc + p;
      }
    }
    for (; i < half; ++i) {
      const float t = qv[i] - coarseT[((size_t)pos * half + i) * Kc + c];
      const float p = t * t;
      acc = acc + p;
    }
    out[((size_t)q * 2 + pos) * Kc + c] = acc;
  }
}

// Stable ascending order of one side's Kc sub-distances (index_utils.c:306-320 sorts each position's
// distances; equal distances keep their code order): key = (distance bits << 32 | code), one wave per
// (query, position).  Kc <= 64 * V.
template <int V>
__global__ __launch_bounds__(64) void side_sort_kernel(const float* __restrict__ sub, u64* __restrict__ sorted, int Kc,
                                                      const int32_t* __restrict__ only = nullptr) {
  // (query * 2 + position); `only`: the queries to sort (a handful that the device traversal handed back)
  const size_t row = only ? (size_t)only[blockIdx.x >> 1] * 2 + (blockIdx.x & 1) : (size_t)blockIdx.x;
  const int lane = threadIdx.x;
  u64 key[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int c = v * 64 + lane;
    key[v] = (c < Kc) ? make_key(sub[row * Kc + c], (uint32_t)c) : KEY_INF;
  }
  wave_sort_full<V>(key);
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int c = v * 64 + lane;
    if (c < Kc) sorted[row * Kc + c] = (key[v] << 32) | (key[v] >> 32);   // memory layout of JoinSide {float dist; int code}
  }
}

// "fq.id IN (targets)" on the device: every target id is resolved to its row (ids ascending: affine
// shortcut or binary search), duplicates and unknown ids drop out through a bitmap, the survivors are
// counted per coarse cell, and a second pass scatters them into per-cell buckets.  (Order inside a
// bucket is arbitrary: the join kernel keys every candidate by (distance, row).)
__global__ __launch_bounds__(256) void join_mark_kernel(const int32_t* __restrict__ tids, int n, const int32_t* __restrict__ ids,
                                                       int64_t N, int affine, const int32_t* __restrict__ cell,
                                                       uint32_t* __restrict__ mark, int32_t* __restrict__ win,
                                                       int32_t* __restrict__ cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t id = tids[i];
  int64_t r = -1;
  if (affine) {
    const int64_t c = (int64_t)id - ids[0];
    if (c >= 0 && c < N) r = c;
  } else {
    int64_t lo = 0, hi = N;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ids[mid] < id) lo = mid + 1; else hi = mid; }
    if (lo < N && ids[lo] == id) r = lo;
  }
  int32_t w = -1;
  if (r >= 0) {
    const uint32_t bit = 1u << (r & 31);
    if (!(atomicOr(mark + (r >> 5), bit) & bit)) { w = (int32_t)r; atomicAdd(cnt + cell[r], 1); }
  }
  win[i] = w;
}
__global__ __launch_bounds__(256) void join_offsets_kernel(const int32_t* __restrict__ cnt, int cells, int32_t* __restrict__ off,
                                                          int32_t* __restrict__ fill, int32_t* __restrict__ off_host) {
  __shared__ int scan[256];
  const int tid = threadIdx.x, per = (cells + 255) / 256;
  const int c0 = tid * per, c1 = (c0 + per < cells) ? c0 + per : cells;
  int sum = 0;
  for (int c = c0; c < c1; ++c) sum += cnt[c];
  scan[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = (tid >= o) ? scan[tid - o] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  int run = scan[tid] - sum;
  for (int c = c0; c < c1; ++c) { off[c] = run; fill[c] = run; off_host[c] = run; run += cnt[c]; }   // (off_host: mapped host memory)
  if (tid == 255) { off[cells] = scan[255]; off_host[cells] = scan[255]; }
}
__global__ __launch_bounds__(256) void join_place_kernel(const int32_t* __restrict__ win, int n, const int32_t* __restrict__ cell,
                                                        int32_t* __restrict__ fill, int32_t* __restrict__ trow) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t r = win[i];
  if (r >= 0) trow[atomicAdd(fill + cell[r], 1)] = r;
}

// ---------------------------------------------------------------------------------------
// one workgroup per scanned query
// ---------------------------------------------------------------------------------------
struct JoinArgs {
  const float* queries;       // [Q][d]
  const int32_t* scan_query;  // [n_scan] query index
  const int32_t* qcell_off;   // [n_scan+1] offsets into qcells (host traversal), or NULL:
  const int32_t* qcell_cnt;   // [Q] cells in row q of qcells[Q][qstride] (device traversal)
  int qstride;
  const int32_t* qcells;      // cells probed by each scanned query
  const int32_t* tcell_off;   // [cells+1] target buckets by cell
  const int32_t* trow;        // target rows, ascending inside a bucket
  const int32_t* ids;         // [N]
  const int16_t* codes;       // [N][MP], rows 16-byte aligned
  int MP;
  const float* vectors;       // [N][d]
  const float* cbT;           // [m][S][K]
  int32_t* out_ids;           // [n_scan][k]
  float* out_dist;            // [n_scan][k]
  int d, m, K, S, k, L, method, double_codes;
  // BIG instantiation (post verification of more than 1024 candidates: k * pvf up to 8192): the candidates and their exact distances
  u64* big_keys = nullptr;    // [n_scan][L]
  float* big_exact = nullptr; // [n_scan][L]
};

__device__ __forceinline__ float sqdist_seq(const float* a, const float* __restrict__ b, int n) {
  float acc = 0.0f;                                   // index_utils.c:500-508
  for (int i = 0; i < n; ++i) {
    const float t = a[i] - b[i];
    const float p = t * t;
    acc = acc + p;
  }
  return acc;
}

// the same chain, the vector read with 16-byte loads (a lane walks its own row: a quarter of the load instructions)
__device__ __forceinline__ float sqdist_seq4(const float* a, const float* __restrict__ b, int n) {
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float acc = 0.0f;
  for (int i = 0; i < n; i += 4) {
    const float4 v = b4[i >> 2];
    float t = a[i] - v.x;     float p = t * t; acc = acc + p;
    t = a[i + 1] - v.y; p = t * t; acc = acc + p;
    t = a[i + 2] - v.z; p = t * t; acc = acc + p;
    t = a[i + 3] - v.w; p = t * t; acc = acc + p;
  }
  return acc;
}

// BIG (method 2 with 1024 < k * pvf <= 8192; V = 16): the k * pvf smallest (ADC distance, row) keys are selected 1024 at a time
// -- pass p walks the query's candidate rows again and admits only keys above the largest key of pass p - 1 (keys are unique:
// the row is part of them) -- into a list in memory; post verification and the replay read it there.
template <int V, bool BIG = false>
__global__ __launch_bounds__(JOIN_WG) void join_query_kernel(JoinArgs a) {
  static_assert(!BIG || V == 16, "selection passes are 1024 keys wide");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int d = a.d, m = a.m, K = a.K, S = a.S, k = a.k, L = a.L;
  const int lutN = m * K;
  const int n_codes = a.double_codes ? m / 2 : m;
  const int range = a.double_codes ? K * K : K;
  // LDS carve (all offsets multiples of 16 bytes)
  size_t off = 0;
  float* qv = reinterpret_cast<float*>(smem + off);            off += ((size_t)d * 4 + 15) & ~(size_t)15;
  float* lut = reinterpret_cast<float*>(smem + off);           off += ((size_t)lutN * 4 + 15) & ~(size_t)15;
  u64* stage = reinterpret_cast<u64*>(smem + off);             off += (size_t)JOIN_WAVES * 64 * 8;
  u64* lists = reinterpret_cast<u64*>(smem + off);             off += (size_t)JOIN_WAVES * 64 * V * 8;
  float* exact = reinterpret_cast<float*>(smem + off);         off += ((size_t)64 * V * 4 + 15) & ~(size_t)15;
  float* s_d = reinterpret_cast<float*>(smem + off);           off += ((size_t)k * 4 + 15) & ~(size_t)15;
  int32_t* s_id = reinterpret_cast<int32_t*>(smem + off);      off += ((size_t)k * 4 + 15) & ~(size_t)15;
  int32_t* c_start = reinterpret_cast<int32_t*>(smem + off);   off += (size_t)JOIN_CELL_CHUNK * 4;        // first target slot of a cell of the chunk
  int32_t* c_pref = reinterpret_cast<int32_t*>(smem + off);    off += (size_t)(JOIN_CELL_CHUNK + 1) * 4;  // [chunk + 1] rows before it
  u64* const s_floor_p = reinterpret_cast<u64*>(smem + ((off + 7) & ~(size_t)7));                          // (BIG; inside the carve's 16 spare bytes)

  const int x = blockIdx.x;
  const int q = a.scan_query[x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

  for (int i = threadIdx.x; i < d; i += JOIN_WG) qv[i] = a.queries[(size_t)q * d + i];
  __syncthreads();
  if (a.method != FREDDY_METHOD_EXACT) {
    // getPrecomputedDistances, index_utils.c:445-455
    for (int e = threadIdx.x; e < lutN; e += JOIN_WG) {
      const int p = e / K, c = e - p * K;
      float acc = 0.0f;
      for (int i = 0; i < S; ++i) {
        const float t = qv[p * S + i] - a.cbT[((size_t)p * S + i) * K + c];
        const float pr = t * t;
        acc = acc + pr;
      }
      lut[e] = acc;
    }
    __syncthreads();
    // (pair codes -- getPrecomputedDistancesDouble, index_utils.c:457-475: a table of the sums of the two rounded
    // sub-distances, n_codes x K^2 entries -- are NOT tabulated: the scan adds the two entries itself, the same binary32
    // addition the table would hold.  15 x 32^2 floats = 61 KB of LDS per workgroup allowed two workgroups per CU.)
  }
  const float* tab = lut;
  const bool vec4 = (d & 3) == 0;   // (rows of d floats are then 16-byte aligned: hipMalloc'd base, pitch 4 d)

  WaveSelect<V> sel;
  const int c_begin = a.qcell_off ? a.qcell_off[x] : q * a.qstride;
  const int c_end = a.qcell_off ? a.qcell_off[x + 1] : c_begin + a.qcell_cnt[q];
  u64 floor_key = 0;
  const int n_pass = BIG ? (L + 64 * V - 1) / (64 * V) : 1;
  for (int pass = 0; pass < n_pass; ++pass) {
  sel.init(stage + wave * 64, (u64)__float_as_uint(JOIN_MAX_DIST) << 32, BIG ? 64 * V : L);
  // The target rows of the query's cells as ONE index space: a query takes ~40 cells of ~16 target rows each, and a loop
  // "cell by cell, 64 rows at a time" left three quarters of the lanes idle and paid three dependent round trips (cell
  // offsets -> row number -> codes) per cell and wave -- 25-33 us of a 48 us workgroup.  Here the cells' offsets are read
  // once (all together), prefix-summed in LDS, and lane t of a pass takes row t of the concatenation (binary search in the
  // prefix sums): every lane busy, two dependent round trips per 256 rows.
  for (int cb = c_begin; cb < c_end; cb += JOIN_CELL_CHUNK) {
    const int nc = c_end - cb < JOIN_CELL_CHUNK ? c_end - cb : JOIN_CELL_CHUNK;
    __syncthreads();
    for (int i = threadIdx.x; i < nc; i += JOIN_WG) {
      const int cell = a.qcells[cb + i];
      const int r0 = a.tcell_off[cell], r1 = a.tcell_off[cell + 1];
      c_start[i] = r0;
      c_pref[i] = r1 - r0;
    }
    __syncthreads();
    if (wave == 0) {   // exclusive prefix sums: a lane sums its stretch, the wave scans the 64 sums
      const int per = (nc + 63) >> 6, lo = lane * per, hi = lo + per < nc ? lo + per : nc;
      int sum = 0;
      for (int i = lo; i < hi; ++i) sum += c_pref[i];
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
      int run = incl - sum;
      for (int i = lo; i < hi; ++i) { const int c = c_pref[i]; c_pref[i] = run; run += c; }
      if (lane == 63) c_pref[nc] = incl;
    }
    __syncthreads();
    const int T = c_pref[nc];
    for (int base = 0; base < T; base += JOIN_WG) {
      const int t = base + (int)threadIdx.x;
      const bool valid = t < T;
      float dist = 0.0f;
      int32_t row = 0;
      if (valid) {
        int lo = 0, hi = nc;   // the last cell whose prefix is <= t (cells without target rows share a prefix with their successor)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c_pref[mid] <= t) lo = mid; else hi = mid; }
        row = a.trow[c_start[lo] + (t - c_pref[lo])];
        if (a.method == FREDDY_METHOD_EXACT) {
          dist = vec4 ? sqdist_seq4(qv, a.vectors + (size_t)row * d, d) : sqdist_seq(qv, a.vectors + (size_t)row * d, d);
        } else {
          // the row's codes: MP / 8 loads of 16 bytes (the first four issued together), eight codes each
          const uint4* cd4 = reinterpret_cast<const uint4*>(a.codes + (size_t)row * a.MP);
          const int nch = a.MP >> 3;
          uint4 w4[4];
#pragma unroll
          for (int c8 = 0; c8 < 4; ++c8) w4[c8] = c8 < nch ? cd4[c8] : uint4{0u, 0u, 0u, 0u};
          for (int c0 = 0; c0 < nch; c0 += 4) {
            if (c0 > 0) {
#pragma unroll
              for (int c8 = 0; c8 < 4; ++c8) w4[c8] = c0 + c8 < nch ? cd4[c0 + c8] : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
              const uint32_t ww[4] = {w4[c8].x, w4[c8].y, w4[c8].z, w4[c8].w};
              if (a.double_codes) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {                     // ivpq_search_in.c:446-451 (int16 pair code)
                  const int l = (c0 + c8) * 4 + u;
                  if (l < n_codes) {   // (K^2 <= 32768: the reference's int16 pair code never wraps, join.h host check)
                    const float pair = tab[(2 * l) * K + (int)(ww[u] & 0xffffu)] + tab[(2 * l + 1) * K + (int)(ww[u] >> 16)];
                    dist = dist + pair;
                  }
                }
              } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {                     // index_utils.c:1126-1133
                  const int l = (c0 + c8) * 8 + u;
                  if (l < m) dist = dist + tab[K * l + (int)((ww[u >> 1] >> ((u & 1) * 16)) & 0xffffu)];
                }
              }
            }
          }
        }
      }
      const u64 key = make_key(dist, (uint32_t)row);
      sel.push(key, valid && (!BIG || pass == 0 || key > floor_key));
    }
  }
  sel.finish();
  // gather the four waves' lists; wave 0 merges them
#pragma unroll
  for (int v = 0; v < V; ++v) lists[(size_t)wave * 64 * V + v * 64 + lane] = sel.acc[v];
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < JOIN_WAVES; ++w) {
      for (int v = 0; v < V; ++v) {
        const u64 key = lists[(size_t)w * 64 * V + v * 64 + lane];
        if (__ballot(key != KEY_INF) == 0ull) break;   // lists are ascending: the rest is empty too
        wave_topk_absorb_sorted<V>(sel.acc, key);
      }
    }
    if constexpr (BIG) {   // this pass's keys behind the earlier ones: (ADC distance, row) ascending over all passes
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int e = pass * 64 * V + v * 64 + lane;
        if (e < L) a.big_keys[(size_t)x * L + e] = sel.acc[v];
      }
      const u64 top = wave_topk_at<V>(sel.acc, 64 * V - 1);   // KEY_INF: fewer keys than a pass holds -- the rows are exhausted
      if (lane == 0) *s_floor_p = top;
    } else
    if (a.method == FREDDY_METHOD_PQ_PV) {
      // survivors stay in (ADC distance, row) order: that is the order postverify walks them
#pragma unroll
      for (int v = 0; v < V; ++v) lists[v * 64 + lane] = (v * 64 + lane < L) ? sel.acc[v] : KEY_INF;
    } else {
      u64 byp[V];
#pragma unroll
      for (int v = 0; v < V; ++v)
        byp[v] = (sel.acc[v] == KEY_INF || v * 64 + lane >= L) ? KEY_INF : ((sel.acc[v] << 32) | (sel.acc[v] >> 32));
      wave_sort_full<V>(byp);
#pragma unroll
      for (int v = 0; v < V; ++v) lists[v * 64 + lane] = byp[v];
    }
  }
  if constexpr (BIG) {
    __syncthreads();
    floor_key = *s_floor_p;
    if (floor_key == KEY_INF) {   // (the slots of the passes that would follow stay empty)
      for (int e = (pass + 1) * 64 * V + (int)threadIdx.x; e < L; e += JOIN_WG) a.big_keys[(size_t)x * L + e] = KEY_INF;
      break;
    }
  }
  }   // passes
  const u64* const cand = BIG ? a.big_keys + (size_t)x * L : lists;
  float* const exact_d = BIG ? a.big_exact + (size_t)x * L : exact;
  for (int i = threadIdx.x; i < k; i += JOIN_WG) { s_d[i] = JOIN_MAX_DIST; s_id[i] = -1; }
  __syncthreads();
  if (a.method == FREDDY_METHOD_PQ_PV) {
    // postverify, index_utils.c:477-498: exact distance of each of the k*pvf survivors
    // (survivor e on lane e / 4 of wave e % 4: the four waves' loads run side by side)
    for (int e0 = 0; e0 < L; e0 += JOIN_WG) {
      const int e = e0 + (int)(threadIdx.x & 63) * JOIN_WAVES + (int)(threadIdx.x >> 6);
      if (e < L) {
        const u64 c = cand[e];
        exact_d[e] = (c == KEY_INF) ? 0.0f
                   : vec4 ? sqdist_seq4(qv, a.vectors + (size_t)key_pos(c) * d, d) : sqdist_seq(qv, a.vectors + (size_t)key_pos(c) * d, d);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 64 && k <= 64) {
    // insertion replay with the list held one slot per lane (wave_topk.h: wave_list_insert)
    float d_slot = JOIN_MAX_DIST;
    int32_t id_slot = -1;
    float maxd = JOIN_MAX_DIST;
    for (int e = 0; e < L; ++e) {
      const u64 c = cand[e];
      if (c == KEY_INF) break;
      float dist;
      uint32_t row;
      if (a.method == FREDDY_METHOD_PQ_PV) { dist = exact_d[e]; row = key_pos(c); }
      else { dist = __uint_as_float((uint32_t)c); row = (uint32_t)(c >> 32); }
      if (dist < maxd) {
        wave_list_insert(d_slot, id_slot, k, dist, a.ids[row]);
        maxd = wave_list_max(d_slot, k);
      }
    }
    if ((int)threadIdx.x < k) { s_d[threadIdx.x] = d_slot; s_id[threadIdx.x] = id_slot; }
  } else if (threadIdx.x == 0 && k > 64) {
    float maxd = JOIN_MAX_DIST;
    for (int e = 0; e < L; ++e) {
      const u64 c = cand[e];
      if (c == KEY_INF) break;
      float dist;
      uint32_t row;
      if (a.method == FREDDY_METHOD_PQ_PV) { dist = exact_d[e]; row = key_pos(c); }
      else { dist = __uint_as_float((uint32_t)c); row = (uint32_t)(c >> 32); }
      if (dist < maxd) {
        int slot = k - 1;                                // updateTopK, index_utils.c:19-33
        while (slot >= 0 && !(s_d[slot] < dist)) --slot;
        ++slot;
        for (int t = k - 2; t >= slot; --t) { s_d[t + 1] = s_d[t]; s_id[t + 1] = s_id[t]; }
        s_d[slot] = dist;
        s_id[slot] = a.ids[row];
        maxd = s_d[k - 1];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < k; i += JOIN_WG) {
    a.out_ids[(size_t)x * k + i] = s_id[i];
    a.out_dist[(size_t)x * k + i] = s_d[i];
  }
}

// ---------------------------------------------------------------------------------------
// device: a10 multi-index traversal for <= 1024 cells             index_utils.c:252-443
// ---------------------------------------------------------------------------------------
// getConfidenceHyp (index_utils.c:673-682) with the reference's types: float variables, double sub-expressions.
// The device's erf is not glibc's bit for bit: the host re-evaluates the expression at the proposed stop.
__host__ __device__ __forceinline__ float join_confidence_expr(int expect, int size, float p, int stat_size) {
  if (expect > size) return 0;
  float mu = size * p;
  float sig = sqrt(size * p * (1.0 - p)) * (((float)stat_size - size) / ((float)stat_size - 1.0));
  return 1.0 - 0.5 * (1.0 + erf((((float)expect) - 0.5 - mu) / (sig * sqrt(2.0))));
}

static constexpr int TRAV_SUM_DW = 8;   // per query: n, cells with targets, target rows, flags (1 tie, 2 exhausted), bits(P_n), bits(P_{n-1})
struct TravArgs {
  const float* sub;          // [Q][2][Kc]
  const int32_t* active;     // [n_active]
  const float* stats;        // [cells+1]
  const int32_t* tcell_off;  // [cells+1]
  int32_t* qcells;           // [Q][cells]: the taken cells that hold targets, in the order they are taken
  int32_t* qcell_cnt;        // [Q]
  int32_t* summary;          // [n_active][TRAV_SUM_DW]
  float* fb_sub;             // [n_active][2 * Kc] or NULL: the sub-distances of a query that is handed to the host (mapped host memory)
  int Kc, cells, n_targets, min_target;
  float confidence;
};

// One wave per query.  Most queries take a few dozen cells: the 64 smallest keys come from a streaming selection
// (WaveSelect) and decide the stop; only a query that needs more than 63 cells sorts all of them -- a bitonic sort in LDS
// with ROLLED loops: the fully unrolled register sort of 1024 keys is ~100 KB of straight-line code that every wave
// streamed through the 64 KB instruction cache once (380 us per launch for 5 000 queries, as long as the join itself).
// SMALL: only the smallest keys are ever held (1.5 instead of 17 KB of LDS for 1024 cells), found with ONE sort + merge (below);
// a query whose stop is not among them (at least its 31 nearest cells) is handed to the host heap like one with equal keys
// (flag 1).  The host picks SMALL when the expected number of cells is far below that.
template <int V, bool SMALL = false>
__global__ __launch_bounds__(64) void join_traverse_kernel(TravArgs a) {
  constexpr int NS = SMALL ? 64 : 64 * V;
  __shared__ u64 s_key[NS];            // (distance bits << 32) | cell, ascending from index 0 as far as they are sorted
  __shared__ float s_stat[NS];
  __shared__ float s_P[NS + 1];
  __shared__ u64 s_stage[64];
  const int lane = threadIdx.x, x = blockIdx.x;
  const int q = a.active[x];
  const int Kc = a.Kc, cells = a.cells;
  const float* d0 = a.sub + ((size_t)q * 2) * Kc;
  const float* d1 = d0 + Kc;
  auto cell_key = [&](int c) -> u64 {
    if (c >= cells) return KEY_INF;
    float acc = 0;            // 0 + D0[c0] + D1[c1], index_utils.c:306-313
    acc += d0[c % Kc];
    acc += d1[c / Kc];
    return make_key(acc, (uint32_t)c);
  };
  const int stat_size = (int)a.stats[cells];
  // ---- the 64 smallest keys, ascending
  int n_valid = 64;   // SMALL: how many of them are known to be the smallest (>= 32)
  if constexpr (SMALL) {
    // The kernel is bound by instruction issue (5 000 lone waves), and a streaming selection that starts without a threshold
    // pays a 64-bit sort + merge for every other batch of 64 keys.  Here: the lane's V keys stay in registers, the 32nd
    // smallest of the 64 lane minima (one 32-bit sort) bounds the 32nd smallest key, only keys up to it are offered (about
    // 40 of 1024): one sort + merge.  Every key below the bound is in the result, so its first n_valid entries are exactly the
    // n_valid smallest keys; a stop beyond them is handed to the host.
    u64 kk[V];
    uint32_t mn = 0xffffffffu;
#pragma unroll
    for (int v = 0; v < V; ++v) { kk[v] = cell_key(v * 64 + lane); mn = min(mn, (uint32_t)(kk[v] >> 32)); }
    const uint32_t dL = (uint32_t)__builtin_amdgcn_readlane((int)wave_sort32(mn), 31);
    WaveSelect<1> sel;
    sel.init(s_stage, ((u64)dL << 32) | 0xffffffffull, 64);
#pragma unroll
    for (int v = 0; v < V; ++v) sel.push(kk[v], kk[v] != KEY_INF);
    sel.finish();
    n_valid = (int)__popcll(__ballot(sel.acc[0] != KEY_INF));
    s_key[lane] = sel.acc[0];
    s_stat[lane] = (sel.acc[0] != KEY_INF) ? a.stats[key_pos(sel.acc[0])] : 0.0f;
  } else {
    WaveSelect<1> sel;
    sel.init(s_stage, KEY_INF, 64);
#pragma unroll 1
    for (int v = 0; v < V; ++v) {
      const u64 kk = cell_key(v * 64 + lane);
      sel.push(kk, kk != KEY_INF);
    }
    sel.finish();
    s_key[lane] = sel.acc[0];
    s_stat[lane] = (sel.acc[0] != KEY_INF) ? a.stats[key_pos(sel.acc[0])] : 0.0f;
  }
  if (lane == 0) s_P[0] = 0.0f;
  __syncthreads();
  // n = the first count whose confidence reaches the threshold ("while (conf(prob) < confidence && emitted < cells)"):
  // lane 0 extends the running sum by a chunk of 64 cells (prob += statistics[cell], :424, sequential binary32 adds),
  // then the 64 lanes test the chunk's 64 counts
  int n = cells;
  bool sorted_all = (V == 1);
  bool beyond = false;   // SMALL: the stop is not among the first 63 cells
  for (int base = 0; base < cells; base += 64) {
    if constexpr (SMALL) { if (base > 0) { beyond = true; n = 0; break; } }
    if (base > 0 && !sorted_all) {
     if constexpr (!SMALL) {
      // more than 63 cells: every key, sorted (rolled bitonic network over LDS; 64 V is a power of two)
#pragma unroll 1
      for (int v = 0; v < V; ++v) s_key[v * 64 + lane] = cell_key(v * 64 + lane);
      __syncthreads();
#pragma unroll 1
      for (int k = 2; k <= 64 * V; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll 1
          for (int t = lane; t < 32 * V; t += 64) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const int l = i | j;
            const u64 lo = s_key[i], hi = s_key[l];
            const bool up = (i & k) == 0;
            if ((lo > hi) == up) { s_key[i] = hi; s_key[l] = lo; }
          }
          __syncthreads();
        }
      }
#pragma unroll 1
      for (int v = 0; v < V; ++v) {
        const u64 kk = s_key[v * 64 + lane];
        s_stat[v * 64 + lane] = (kk != KEY_INF) ? a.stats[key_pos(kk)] : 0.0f;
      }
      sorted_all = true;
      __syncthreads();
     }
    }
    if (lane == 0) {
      float P = s_P[base];
      const int hi = base + 64 < cells ? base + 64 : cells;
      for (int i = base; i < hi; ++i) { P = P + s_stat[i]; s_P[i + 1] = P; }
    }
    __syncthreads();
    const int cnt = base + lane;
    const bool ok = cnt < cells && (!SMALL || cnt + 1 < n_valid) && !(join_confidence_expr(a.min_target, a.n_targets, s_P[cnt < cells ? cnt : 0], stat_size) < a.confidence);
    const u64 m = __ballot(ok);
    if (m != 0ull) { n = base + (int)__builtin_ctzll(m); break; }
  }
  // (n <= 63 when only the 64 smallest keys are sorted; n == cells needs all of them)
  // equal keys among the first n + 1 sorted cells: the heap's order is history-dependent there -> the host decides
  bool tie = false;
  for (int i = lane; i < n && i + 1 < cells; i += 64) tie = tie || ((uint32_t)(s_key[i] >> 32) == (uint32_t)(s_key[i + 1] >> 32));
  const bool any_tie = __ballot(tie) != 0ull;
  // the taken cells that hold targets, compacted in order; their rows
  int n_keep = 0, rows = 0;
  int32_t* dst = a.qcells + (size_t)q * cells;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    int tc = 0, c = 0;
    if (i < n) { c = (int)key_pos(s_key[i]); tc = a.tcell_off[c + 1] - a.tcell_off[c]; }
    const u64 m = __ballot(tc > 0);
    if (tc > 0) dst[n_keep + lanes_below(m)] = c;
    n_keep += (int)__popcll(m);
    rows += tc;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rows += __shfl_xor(rows, o, 64);
  // a query the host heap has to traverse: its 2 x Kc sub-distances go along with the summary (the host sorts the two sides
  // itself: a side-sort kernel, two small copies and a synchronisation per handed-back query cost 0.05 - 0.08 ms per call)
  if ((any_tie || beyond) && a.fb_sub)
    for (int i = lane; i < 2 * Kc; i += 64) a.fb_sub[(size_t)x * 2 * Kc + i] = d0[i];
  if (lane == 0) {
    a.qcell_cnt[q] = n_keep;
    int32_t* sm = a.summary + (size_t)x * TRAV_SUM_DW;
    sm[0] = n; sm[1] = n_keep; sm[2] = rows; sm[3] = ((any_tie || beyond) ? 1 : 0) | (n >= cells ? 2 : 0);
    sm[4] = (int32_t)__float_as_uint(s_P[n]);
    sm[5] = (int32_t)__float_as_uint(n > 0 ? s_P[n - 1] : 0.0f);
    // the device's own values of the expression at the stop and one step before it: the host re-evaluates with its
    // libm only where one of them is within 1e-5 of the confidence (the arguments of erf are IEEE-identical on both
    // sides -- float / double products, correctly rounded sqrt and division -- and the two erf implementations differ
    // by a few units in the last place of a double)
    sm[6] = (int32_t)__float_as_uint(join_confidence_expr(a.min_target, a.n_targets, s_P[n], stat_size));
    sm[7] = (int32_t)__float_as_uint(n > 0 ? join_confidence_expr(a.min_target, a.n_targets, s_P[n - 1], stat_size) : 0.0f);
  }
}

// ---------------------------------------------------------------------------------------
// host: a10 multi-index traversal                               index_utils.c:252-443
// ---------------------------------------------------------------------------------------
static inline float join_confidence_hyp(int expect, int size, float p, int stat_size) {
  // getConfidenceHyp, index_utils.c:673-682 (float variables, double sub-expressions)
  if (expect > size) return 0;
  float mu = size * p;
  float sig = sqrt(size * p * (1.0 - p)) * (((float)stat_size - size) / ((float)stat_size - 1.0));
  return 1.0 - 0.5 * (1.0 + erf((((float)expect) - 0.5 - mu) / (sig * sqrt(2))));
}

struct JoinSide { float dist; int code; };
struct JoinNode { float key; int cell, p0, p1; };

// per-query scratch reused between calls of one worker thread
struct JoinTraversal {
  std::vector<JoinNode> heap;
  std::vector<uint32_t> traversed, queued;
  std::vector<float> cell_dist;
};

// sides: the query's two sub-distance arrays stably sorted ascending (qsort + cmpTopKEntry,
// index_utils.c:104-116,317-319; stable as glibc <= 2.36).  Appends visited cells to `out`.
// Returns true iff the query exhausted every cell.
static inline bool join_select_cells(const JoinSide* s0, const JoinSide* s1, const float* d0, const float* d1, int Kc,
                                     const float* stats, int n_targets, int min_target, float confidence,
                                     JoinTraversal& w, std::vector<int32_t>& out) {
  const int cells = Kc * Kc;
  w.cell_dist.resize(cells);
  for (int c = 0; c < cells; ++c) {        // 0 + D0[c0] + D1[c1], index_utils.c:306-313
    float acc = 0;
    acc += d0[c % Kc];
    acc += d1[c / Kc];
    w.cell_dist[c] = acc;
  }
  w.traversed.assign(cells / 32 + 1, 0u);
  w.queued.assign(cells / 32 + 1, 0u);
  w.heap.resize(cells + 1);
  JoinNode* h = w.heap.data();
  int len = 1;
  h[0].p0 = 0; h[0].p1 = 0;
  h[0].cell = s0[0].code + Kc * s1[0].code;
  h[0].key = w.cell_dist[h[0].cell];
  float prob = 0.0f;
  int emitted = 0;
  const int stat_size = (int)stats[cells];
  auto push = [&](const JoinNode& nd) {     // index_utils.c:118-131
    int i = len, parent = (i - 1) / 2;
    while (i > 0 && h[parent].key > nd.key) { h[i] = h[parent]; i = parent; parent = (parent - 1) / 2; }
    h[i] = nd;
    ++len;
  };
  auto pop = [&]() {                        // index_utils.c:133-155
    JoinNode top = h[0];
    h[0] = h[len - 1];
    --len;
    const int n = len;
    int i = 0;
    while (i != n) {
      int pick = n;
      const int child = 1 + 2 * i;
      if (child <= n - 1 && h[child].key < h[pick].key) pick = child;
      if (child <= n - 1 && h[child + 1].key < h[pick].key) pick = child + 1;
      h[i] = h[pick];
      i = pick;
    }
    return top;
  };
  while (join_confidence_hyp(min_target, n_targets, prob, stat_size) < confidence && emitted < cells) {
    const JoinNode cur = pop();
    const int here = cur.p0 + Kc * cur.p1;
    w.traversed[here / 32] |= 1u << (here % 32);
    int diag = cur.p0 + 1 + Kc * (cur.p1 - 1);
    if (cur.p0 < Kc - 1 && (cur.p1 == 0 || (w.traversed[diag / 32] & (1u << (diag % 32))))) {
      const int np0 = cur.p0 + 1, np1 = cur.p1, npi = np0 + Kc * np1;
      if (!(w.queued[npi / 32] & (1u << (npi % 32)))) {
        JoinNode nd; nd.p0 = np0; nd.p1 = np1; nd.cell = s0[np0].code + Kc * s1[np1].code; nd.key = w.cell_dist[nd.cell];
        push(nd);
        w.queued[npi / 32] |= 1u << (npi % 32);
      }
    }
    diag = cur.p0 - 1 + Kc * (cur.p1 + 1);
    if (cur.p1 < Kc - 1 && (cur.p0 == 0 || (w.traversed[diag / 32] & (1u << (diag % 32))))) {
      const int np0 = cur.p0, np1 = cur.p1 + 1, npi = np0 + Kc * np1;
      if (!(w.queued[npi / 32] & (1u << (npi % 32)))) {
        JoinNode nd; nd.p0 = np0; nd.p1 = np1; nd.cell = s0[np0].code + Kc * s1[np1].code; nd.key = w.cell_dist[nd.cell];
        push(nd);
        w.queued[npi / 32] |= 1u << (npi % 32);
      }
    }
    prob += stats[cur.cell];
    out.push_back(cur.cell);
    ++emitted;
  }
  return emitted >= cells;
}

static inline void join_stable_sort(JoinSide* a, int n, JoinSide* tmp) {
  for (int width = 1; width < n; width *= 2) {
    for (int lo = 0; lo < n; lo += 2 * width) {
      const int mid = std::min(lo + width, n), hi = std::min(lo + 2 * width, n);
      int i = lo, j = mid, o = lo;
      while (i < mid && j < hi) { if (a[j].dist < a[i].dist) tmp[o++] = a[j++]; else tmp[o++] = a[i++]; }
      while (i < mid) tmp[o++] = a[i++];
      while (j < hi) tmp[o++] = a[j++];
    }
    memcpy(a, tmp, sizeof(JoinSide) * (size_t)n);
  }
}

// Host worker pool for the per-query traversals.  The workers are created on first use (never at
// library load, so a forking host stays safe) and parked on a condition variable; spawning threads per
// call cost ~1 ms per join_parallel_for on the GPU box (32 x std::thread), more than the work itself.
class JoinPool {
 public:
  static JoinPool& get() { static JoinPool p; return p; }
  int size() const { return (int)workers_.size(); }
  // runs fn(t) for t = 0..n_chunks-1 (n_chunks <= size()+1; chunk 0 runs on the caller)
  void run(int n_chunks, const std::function<void(int)>& fn) {
    if (n_chunks <= 1) { if (n_chunks == 1) fn(0); return; }
    {
      std::lock_guard<std::mutex> g(mu_);
      fn_ = &fn; chunks_ = n_chunks; pending_ = n_chunks - 1; ++generation_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> g(mu_);
    done_.wait(g, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  JoinPool() {
    static const int cap = getenv("FREDDY_GPU_JOIN_THREADS") ? atoi(getenv("FREDDY_GPU_JOIN_THREADS")) : 32;
    const unsigned hw = std::thread::hardware_concurrency();
    const int n = std::max(0, (int)std::min<unsigned>(hw ? hw : 1, (unsigned)std::max(cap, 1)) - 1);
    for (int i = 0; i < n; ++i) workers_.emplace_back([this, i] { loop(i + 1); });
  }
  ~JoinPool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; ++generation_; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void loop(int id) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)>* fn = nullptr;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        if (id < chunks_) fn = fn_;
      }
      if (fn) {
        (*fn)(id);
        std::lock_guard<std::mutex> g(mu_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int chunks_ = 0, pending_ = 0;
  unsigned long generation_ = 0;
  bool stop_ = false;
};

template <class F>
static inline void join_parallel_for(int n, F&& f) {
  if (n < 256) { f(0, n, 0); return; }
  JoinPool& pool = JoinPool::get();
  const int nt = std::min(pool.size() + 1, (n + 63) / 64);
  if (nt <= 1) { f(0, n, 0); return; }
  const int per = (n + nt - 1) / nt;
  const int chunks = (n + per - 1) / per;
  pool.run(chunks, [&](int t) {
    const int lo = t * per, hi = std::min(n, lo + per);
    if (lo < hi) f(lo, hi, t);
  });
}

static inline int join_pick_V(int L) {
  if (L <= 64) return 1;
  if (L <= 128) return 2;
  if (L <= 256) return 4;
  if (L <= 512) return 8;
  if (L <= 1024) return 16;
  return 0;
}

static inline int join_launch(hipStream_t s, const JoinArgs& a, int n_scan, int V, size_t lds) {
  dim3 grid((unsigned)n_scan), block(JOIN_WG);
  if (a.big_keys) {
    hipLaunchKernelGGL((join_query_kernel<16, true>), grid, block, lds, s, a);
    JOIN_HIP(hipGetLastError());
    return 0;
  }
#define JOIN_CASE(v)                                                                                         \
  case v: hipLaunchKernelGGL((join_query_kernel<v>), grid, block, lds, s, a); break;
  switch (V) {
    JOIN_CASE(1) JOIN_CASE(2) JOIN_CASE(4) JOIN_CASE(8) JOIN_CASE(16)
    default: return join_fail(FREDDY_E_LIMIT, "unsupported selection width");
  }
#undef JOIN_CASE
  JOIN_HIP(hipGetLastError());
  return 0;
}

static inline int join_run(JoinIndex* j, hipStream_t s, const float* queries, int Q, int k, const int32_t* target_ids,
                           int64_t n_targets, int alpha, int pvf, int method, int use_tl, float confidence,
                           int double_threshold, int32_t* out_ids, float* out_dist, int32_t* iterations_out) {
  if (method < 0 || method > 2) return join_fail(FREDDY_E_ARG, "Unknown computation method!");   // ivpq_search_in.c:374-376
  if (method != FREDDY_METHOD_PQ && !j->has_vectors) return join_fail(FREDDY_E_ARG, "methods 1 and 2 need the vectors to be pinned");
  if (n_targets > INT32_MAX) return join_fail(FREDDY_E_LIMIT, "too many targets");
  const int d = j->d, m = j->m, K = j->K, Kc = j->Kc, cells = j->cells;
  const int alpha_original = alpha;
  // stage timers under the names of the reference's elog(INFO, "TRACK <stage> %f") lines (freddy_gpu_last_track)
  j->track = freddy_track();
  auto now = [] { return std::chrono::steady_clock::now(); };
  const auto t_start = now();
  auto t_last = t_start;
#ifdef FREDDY_LAB
  static const bool jtrace = getenv("FREDDY_GPU_JOIN_TRACE") != nullptr;   // host timeline of a call on stderr (lab builds; tools/lab/join_trace_host.py)
  auto mark = [&](const char* what) { if (jtrace) fprintf(stderr, "[join] %7.1f us  %s\n", std::chrono::duration<double, std::micro>(now() - t_start).count(), what); };
#else
  auto mark = [](const char*) {};
#endif
  auto track = [&](double freddy_track::*stage) {
    const auto t = now();
    j->track.*stage += std::chrono::duration<double>(t - t_last).count();
    t_last = t;
  };
  if (pvf < 1) pvf = 1;                                                                       // :207-209
  bool double_codes = false;
  if (method != FREDDY_METHOD_EXACT) double_codes = ((int64_t)alpha * k > double_threshold);  // :262-266
  if (double_codes && (int64_t)K * K > 32768) return join_fail(FREDDY_E_LIMIT, "pair codes of K=%d overflow the reference's int16", K);
  const int64_t Lw = (method == FREDDY_METHOD_PQ_PV) ? (int64_t)k * pvf : 2 * (int64_t)k;
  // (post verification walks its candidates in (ADC distance, row) order whatever their number: up to 8192 of them, selected 1024
  // per pass -- join_query_kernel<16, true>; the replay of methods 0 / 1 holds 2k keys in one wave's registers)
  const bool big = method == FREDDY_METHOD_PQ_PV && Lw > 1024;
  if (Lw > (big ? 8192 : 1024))
    return join_fail(FREDDY_E_LIMIT, big ? "k*pvf=%lld exceeds this build's limit of 8192" : "2k=%lld exceeds this build's limit of 1024", (long long)Lw);
  const int L = (int)Lw;
  const int V = big ? 16 : join_pick_V(L);
  for (int i = 0; i < Q * k; ++i) { out_ids[i] = -1; out_dist[i] = JOIN_MAX_DIST; }            // initTopKs :238
  if (iterations_out) *iterations_out = 0;
  if (Q == 0) return 0;

  const int n_codes = double_codes ? m / 2 : m;
  const int range = double_codes ? K * K : K;
  auto r16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
  size_t lds = r16((size_t)d * 4) + r16((size_t)m * K * 4) +
               (size_t)JOIN_WAVES * 64 * 8 + (size_t)JOIN_WAVES * 64 * V * 8 + r16((size_t)64 * V * 4) + r16((size_t)k * 4) +
               r16((size_t)k * 4) + (size_t)(2 * JOIN_CELL_CHUNK + 1) * 4 + 16;
  if (lds > 160 * 1024) return join_fail(FREDDY_E_LIMIT, "LDS need of %zu bytes exceeds 160 KiB (m=%d K=%d k*pvf=%d)", lds, m, K, L);

  // "fq.id IN (targets)": resolved, de-duplicated and bucketed by cell on the device (see join_mark_kernel)
  void *d_q, *d_sub, *d_tcell, *d_trow, *d_scan, *d_qoff, *d_qcells = nullptr, *d_oi, *d_od, *d_win, *d_cnt, *d_sorted;
  if (join_buf(j, 0, sizeof(float) * (size_t)Q * d, &d_q) || join_buf(j, 1, sizeof(float) * (size_t)Q * 2 * Kc, &d_sub) ||
      join_buf(j, 2, sizeof(int32_t) * (size_t)(cells + 1), &d_tcell) ||
      join_buf(j, 3, sizeof(int32_t) * std::max<size_t>((size_t)n_targets, 1), &d_trow) ||
      join_buf(j, 4, sizeof(int32_t) * (size_t)Q, &d_scan) || join_buf(j, 5, sizeof(int32_t) * (size_t)(Q + 1), &d_qoff) ||
      join_buf(j, 7, sizeof(int32_t) * (size_t)Q * k, &d_oi) || join_buf(j, 8, sizeof(float) * (size_t)Q * k, &d_od) ||
      join_buf(j, 10, sizeof(int32_t) * std::max<size_t>((size_t)n_targets, 1), &d_win) ||
      join_buf(j, 11, sizeof(int32_t) * (size_t)cells * 2, &d_cnt) || join_buf(j, 12, sizeof(u64) * (size_t)Q * 2 * Kc, &d_sorted))
    return FREDDY_E_NOMEM;
  {
    const size_t tl_bytes = sizeof(int32_t) * ((size_t)cells + 1 + (size_t)std::max<int64_t>(n_targets, 1));
    if (tl_bytes > j->h_tl_cap) {
      if (j->h_tl) (void)hipHostFree(j->h_tl);
      j->h_tl = nullptr; j->h_tl_cap = 0; j->tl_valid = false;
      if (hipHostMalloc(&j->h_tl, tl_bytes + tl_bytes / 4 + 256, hipHostMallocDefault) != hipSuccess) { j->h_tl = nullptr; return join_fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
      j->h_tl_cap = tl_bytes + tl_bytes / 4 + 256;
    }
  }
  const int32_t* tcell_off = static_cast<const int32_t*>(j->h_tl);          // [cells + 1]; complete on the host after the first synchronisation
  int32_t* h_tids = static_cast<int32_t*>(j->h_tl) + (size_t)cells + 1;     // the target array as the mark kernel reads it
  const bool tl_hit = j->tl_valid && j->tl_n == n_targets && j->tl_cells == cells &&
                      (n_targets == 0 || memcmp(h_tids, target_ids, sizeof(int32_t) * (size_t)n_targets) == 0);
  if (!tl_hit) {   // (a hit: d_tcell / d_trow and the pinned offsets still hold this target array's buckets)
    j->tl_valid = false;
    int32_t* cnt = (int32_t*)d_cnt;
    int32_t* fill = cnt + cells;
    JOIN_HIP(hipMemsetAsync(j->markbits, 0, sizeof(uint32_t) * (size_t)((j->N + 31) / 32 + 1), s));
    JOIN_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * (size_t)cells * 2, s));
    void* p_tl = nullptr;
    JOIN_HIP(hipHostGetDevicePointer(&p_tl, j->h_tl, 0));
    if (n_targets > 0) {
      memcpy(h_tids, target_ids, sizeof(int32_t) * (size_t)n_targets);
      hipLaunchKernelGGL(join_mark_kernel, dim3((unsigned)((n_targets + 255) / 256)), dim3(256), 0, s, static_cast<const int32_t*>(p_tl) + (size_t)cells + 1,
                         (int)n_targets, (const int32_t*)j->ids, j->N, j->ids_affine ? 1 : 0, (const int32_t*)j->cell, j->markbits,
                         (int32_t*)d_win, cnt);
    }
    hipLaunchKernelGGL(join_offsets_kernel, dim3(1), dim3(256), 0, s, (const int32_t*)cnt, cells, (int32_t*)d_tcell, fill, static_cast<int32_t*>(p_tl));
    if (n_targets > 0)
      hipLaunchKernelGGL(join_place_kernel, dim3((unsigned)((n_targets + 255) / 256)), dim3(256), 0, s, (const int32_t*)d_win,
                         (int)n_targets, (const int32_t*)j->cell, fill, (int32_t*)d_trow);
    JOIN_HIP(hipGetLastError());
    j->tl_n = n_targets; j->tl_cells = cells;   // (valid once the offsets have arrived: first synchronisation below)
  }
  mark("target array enqueued");
  track(&freddy_track::data_retrieval_time);   // "fq.id IN (targets)" (enqueue only: the device work overlaps what follows)
  // a query buffer that is pinned already (freddy_gpu_host_alloc: what pg/freddy_gpu_glue.c's query_buffer() hands over) is read
  // where it is -- the 6 MB staging copy of 5 000 queries is the longest host step of a call
  const float* p_queries = nullptr;
  {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    if (hipPointerGetAttributes(&attr, queries) == hipSuccess && attr.type == hipMemoryTypeHost) p_queries = static_cast<const float*>(attr.devicePointer);
    else (void)hipGetLastError();
    // sub_dist_kernel reads with 16-byte loads: a VIEW into a pinned buffer at an odd offset goes through the staging copy (whose
    // base hipHostMalloc aligns), and so does a buffer another device's context pinned (no device address here)
    if (p_queries && (reinterpret_cast<uintptr_t>(p_queries) & 15u)) p_queries = nullptr;
  }
  const bool fused_front_p = (d & 1) == 0 && d / 2 <= 512;
  if (p_queries && fused_front_p) {
    hipLaunchKernelGGL(sub_dist_kernel, dim3((unsigned)Q, 2), dim3(64), 0, s, p_queries, j->coarseT, (float*)d_sub, d, Kc, (float*)d_q, 0);
    JOIN_HIP(hipGetLastError());
  } else {   // queries: host copy into pinned staging, read by a copy kernel (1.2 KB per query over PCIe)
    const size_t qbytes = sizeof(float) * (size_t)Q * d;
    if (qbytes > j->h_q_cap) {
      if (j->h_q) (void)hipHostFree(j->h_q);
      j->h_q = nullptr; j->h_q_cap = 0;
      if (hipHostMalloc(&j->h_q, qbytes + qbytes / 4 + 256, hipHostMallocDefault) != hipSuccess) { j->h_q = nullptr; return join_fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
      j->h_q_cap = qbytes + qbytes / 4 + 256;
    }
    // (in pieces of whole queries: the host copies piece i + 1 while sub_dist_kernel pulls piece i over PCIe, writes the device
    // copy and computes the piece's sub-distances)
    const int piece_q = std::max((Q + 3) / 4, 64);
    const bool fused_front = (d & 1) == 0 && d / 2 <= 512;   // (the kernel's staging buffer; odd d: the halves do not cover the vector)
    for (int qa = 0; qa < Q; qa += piece_q) {
      const int nq = std::min(piece_q, Q - qa);
      memcpy(static_cast<float*>(j->h_q) + (size_t)qa * d, queries + (size_t)qa * d, sizeof(float) * (size_t)nq * d);
      if (fused_front)
        hipLaunchKernelGGL(sub_dist_kernel, dim3((unsigned)nq, 2), dim3(64), 0, s, (const float*)j->h_q, j->coarseT, (float*)d_sub, d, Kc, (float*)d_q, qa);
      else
        hipLaunchKernelGGL(join_copy_kernel, dim3((unsigned)std::min<size_t>(((size_t)nq * d + 255) / 256, 1024)), dim3(256), 0, s,
                           (const uint32_t*)j->h_q + (size_t)qa * d, (uint32_t*)d_q + (size_t)qa * d, (size_t)nq * d);
    }
    if (!fused_front)
      hipLaunchKernelGGL(sub_dist_kernel, dim3((unsigned)Q, 2), dim3(64), 0, s, (const float*)d_q, j->coarseT, (float*)d_sub, d, Kc, (float*)nullptr, 0);
    JOIN_HIP(hipGetLastError());
  }
  mark("queries staged, sub-distances enqueued");
  const int SV = join_pick_V(Kc);
  if (SV == 0) return join_fail(FREDDY_E_LIMIT, "coarse_codes=%d exceeds this build's limit of 1024", Kc);
  // The multi-index traversal runs on the device for <= 1024 cells (join_traverse_kernel; the host's libm checks every
  // stop); larger multi-indexes, option join_host_traversal and the queries the device hands back use the host heap.
  const bool dev_trav = cells <= 1024 && !j->host_traversal;
  const int TV = join_pick_V(cells);
  std::vector<float> sub;
  std::vector<JoinSide> sides;   // per-query sorted sides (they do not depend on alpha)
  std::vector<int32_t> side_slot;   // query -> its rows in sub / sides (-1: not fetched)
  bool host_sides_all = false;
  auto sort_sides = [&](unsigned rows, const int32_t* only) {
    switch (SV) {
      case 1: hipLaunchKernelGGL((side_sort_kernel<1>), dim3(rows), dim3(64), 0, s, (const float*)d_sub, (u64*)d_sorted, Kc, only); break;
      case 2: hipLaunchKernelGGL((side_sort_kernel<2>), dim3(rows), dim3(64), 0, s, (const float*)d_sub, (u64*)d_sorted, Kc, only); break;
      case 4: hipLaunchKernelGGL((side_sort_kernel<4>), dim3(rows), dim3(64), 0, s, (const float*)d_sub, (u64*)d_sorted, Kc, only); break;
      case 8: hipLaunchKernelGGL((side_sort_kernel<8>), dim3(rows), dim3(64), 0, s, (const float*)d_sub, (u64*)d_sorted, Kc, only); break;
      default: hipLaunchKernelGGL((side_sort_kernel<16>), dim3(rows), dim3(64), 0, s, (const float*)d_sub, (u64*)d_sorted, Kc, only); break;
    }
  };
  // the host heap's inputs -- sub-distances and their stable per-side order -- for the queries of `need`: all queries at
  // once when many are asked for, else just those (the device traversal hands back a query or two per call: sorting and
  // copying 5 000 queries' sides for them cost 0.2 ms)
  auto fetch_sides = [&](const std::vector<int32_t>& need) -> int {
    if (host_sides_all) return 0;
    static_assert(sizeof(JoinSide) == 8, "side_sort_kernel writes JoinSide records");
    std::vector<int32_t> miss;
    for (int32_t q : need) if (side_slot.empty() || side_slot[(size_t)q] < 0) miss.push_back(q);
    if (miss.empty()) return 0;
    const size_t row = (size_t)2 * Kc;
    if (miss.size() * 8 > (size_t)Q) {
      sub.resize((size_t)Q * row);
      sides.resize((size_t)Q * row);
      sort_sides((unsigned)Q * 2, nullptr);
      JOIN_HIP(hipGetLastError());
      JOIN_HIP(hipMemcpyAsync(sub.data(), d_sub, sizeof(float) * sub.size(), hipMemcpyDeviceToHost, s));
      JOIN_HIP(hipMemcpyAsync(sides.data(), d_sorted, sizeof(JoinSide) * sides.size(), hipMemcpyDeviceToHost, s));
      JOIN_HIP(hipStreamSynchronize(s));
      side_slot.resize((size_t)Q);
      for (int q = 0; q < Q; ++q) side_slot[(size_t)q] = q;
      host_sides_all = true;
      return 0;
    }
    if (side_slot.empty()) side_slot.assign((size_t)Q, -1);
    const size_t base = sub.size() / row;
    sub.resize((base + miss.size()) * row);
    sides.resize((base + miss.size()) * row);
    JOIN_HIP(hipMemcpyAsync(d_scan, miss.data(), sizeof(int32_t) * miss.size(), hipMemcpyHostToDevice, s));   // (the scan list goes into this buffer later, in stream order)
    sort_sides((unsigned)miss.size() * 2, (const int32_t*)d_scan);
    JOIN_HIP(hipGetLastError());
    for (size_t i = 0; i < miss.size(); ++i) {
      JOIN_HIP(hipMemcpyAsync(sub.data() + (base + i) * row, (const float*)d_sub + (size_t)miss[i] * row, sizeof(float) * row, hipMemcpyDeviceToHost, s));
      JOIN_HIP(hipMemcpyAsync(sides.data() + (base + i) * row, (const JoinSide*)d_sorted + (size_t)miss[i] * row, sizeof(JoinSide) * row, hipMemcpyDeviceToHost, s));
    }
    JOIN_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < miss.size(); ++i) side_slot[(size_t)miss[i]] = (int32_t)(base + i);
    return 0;
  };
  std::vector<int32_t> all_queries;
  if (!dev_trav) {
    all_queries.resize((size_t)Q);
    for (int i = 0; i < Q; ++i) all_queries[(size_t)i] = i;
    if (int rc = fetch_sides(all_queries)) return rc;
  }
  // pinned landing zone: [Q][TRAV_SUM_DW] traversal summaries, [Q][k] ids, [Q][k] distances
  void *d_active = nullptr, *d_qstrided = nullptr, *d_qcnt = nullptr, *d_summary = nullptr;
  int32_t* h_summary = nullptr; int32_t* h_oi_p = nullptr; float* h_od_p = nullptr; int32_t* h_active = nullptr; int32_t* h_scan = nullptr;
  int32_t* p_summary = nullptr; int32_t* p_oi = nullptr; float* p_od = nullptr; int32_t* p_active = nullptr; int32_t* p_scan = nullptr;   // the same block as the device sees it
  float* h_fbsub = nullptr; float* p_fbsub = nullptr;   // [Q][2 * Kc]: sub-distances of the queries the device traversal hands back
  {
    const size_t need = sizeof(int32_t) * (size_t)Q * (TRAV_SUM_DW + 2 * (size_t)k + 2 + 2 * (size_t)Kc) + 64;   // + the active list, the scan list, the handed-back queries' sub-distances
    if (need > j->h_sum_cap) {
      if (j->h_sum) (void)hipHostFree(j->h_sum);
      j->h_sum = nullptr; j->h_sum_cap = 0;
      if (hipHostMalloc(&j->h_sum, need + need / 4, hipHostMallocDefault) != hipSuccess) { j->h_sum = nullptr; return join_fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
      j->h_sum_cap = need + need / 4;
    }
    h_summary = static_cast<int32_t*>(j->h_sum);
    h_oi_p = h_summary + (size_t)Q * TRAV_SUM_DW;
    h_od_p = reinterpret_cast<float*>(h_oi_p + (size_t)Q * k);
    h_active = reinterpret_cast<int32_t*>(h_od_p + (size_t)Q * k);
    h_scan = h_active + Q;
    h_fbsub = reinterpret_cast<float*>(h_scan + Q);
    // The kernels read the lists and write summaries / results in this pinned block DIRECTLY (it is mapped into the device's
    // address space): every hipMemcpyAsync between two kernels of a stream is an SDMA copy ordered against them by signals,
    // ~12 us per hop, and a round had five of them.
    void* dp = nullptr;
    JOIN_HIP(hipHostGetDevicePointer(&dp, j->h_sum, 0));
    p_summary = static_cast<int32_t*>(dp);
    p_oi = p_summary + (size_t)Q * TRAV_SUM_DW;
    p_od = reinterpret_cast<float*>(p_oi + (size_t)Q * k);
    p_active = reinterpret_cast<int32_t*>(p_od + (size_t)Q * k);
    p_scan = p_active + Q;
    p_fbsub = reinterpret_cast<float*>(p_scan + Q);
  }
  if (dev_trav) {
    if (join_buf(j, 13, sizeof(int32_t) * (size_t)Q, &d_active) || join_buf(j, 14, sizeof(int32_t) * (size_t)Q * cells, &d_qstrided) ||
        join_buf(j, 15, sizeof(int32_t) * (size_t)Q * (1 + TRAV_SUM_DW), &d_qcnt))
      return FREDDY_E_NOMEM;
    d_summary = static_cast<int32_t*>(d_qcnt) + Q;
  }
  if (!dev_trav) JOIN_HIP(hipStreamSynchronize(s));   // (tcell_off is on the host now; the device path waits with its first summaries)

  track(&freddy_track::precomputation_time);   // queries in, sub-distances (+ side sorts and their way back for the host heap)
  std::vector<int32_t> active(Q), target_count(Q, 0);
  for (int i = 0; i < Q; ++i) active[i] = i;
  std::vector<std::vector<int32_t>> qcells(Q);          // host-traversed queries only
  std::vector<int32_t> scan, scan_fb, qoff, flat;
  std::vector<int32_t> q_n(Q, 0), q_rows(Q, 0);         // this round: cells taken, target rows in them
  std::vector<uint8_t> q_host(Q, 0), q_exh(Q, 0);        // this round: traversed on the host / exhausted every cell
  int iterations = 0;
  // Traversal of the n_act queries listed in d_active for `min_target` expected targets; the summaries are on their way to
  // h_summary (row x of the list) when this returns.  Round r + 1's traversal (alpha doubled) is launched right behind
  // round r's join kernel, for every query still active: its summaries arrive with round r's lists in one
  // synchronisation, and the queries that go on find theirs at spec_index[q].
  std::vector<int32_t> spec_index((size_t)Q, 0);
  bool spec_valid = false;
  auto launch_traverse = [&](int n_act, int min_target) -> int {
    TravArgs ta;
    ta.sub = (const float*)d_sub; ta.active = (const int32_t*)d_active; ta.stats = j->d_stats; ta.tcell_off = (const int32_t*)d_tcell;
    ta.qcells = (int32_t*)d_qstrided; ta.qcell_cnt = (int32_t*)d_qcnt; ta.summary = p_summary; ta.fb_sub = p_fbsub;
    ta.Kc = Kc; ta.cells = cells; ta.n_targets = (int)n_targets; ta.min_target = min_target; ta.confidence = confidence;
    // (the 64 smallest keys suffice when the stop is expected far below 63 cells: four times the cells min_target needs at
    // the targets' average density; a query that needs more goes to the host heap)
    const double per_cell = (double)n_targets / (double)std::max(cells, 1);
    const bool small = TV > 1 && per_cell > 0.0 && 3.0 * (double)min_target / per_cell < 31.0;
    if (small) {
      switch (TV) {
        case 2: hipLaunchKernelGGL((join_traverse_kernel<2, true>), dim3((unsigned)n_act), dim3(64), 0, s, ta); break;
        case 4: hipLaunchKernelGGL((join_traverse_kernel<4, true>), dim3((unsigned)n_act), dim3(64), 0, s, ta); break;
        case 8: hipLaunchKernelGGL((join_traverse_kernel<8, true>), dim3((unsigned)n_act), dim3(64), 0, s, ta); break;
        default: hipLaunchKernelGGL((join_traverse_kernel<16, true>), dim3((unsigned)n_act), dim3(64), 0, s, ta); break;
      }
    } else
    switch (TV) {
      case 1: hipLaunchKernelGGL((join_traverse_kernel<1>), dim3((unsigned)n_act), dim3(64), 0, s, ta); break;
      case 2: hipLaunchKernelGGL((join_traverse_kernel<2>), dim3((unsigned)n_act), dim3(64), 0, s, ta); break;
      case 4: hipLaunchKernelGGL((join_traverse_kernel<4>), dim3((unsigned)n_act), dim3(64), 0, s, ta); break;
      case 8: hipLaunchKernelGGL((join_traverse_kernel<8>), dim3((unsigned)n_act), dim3(64), 0, s, ta); break;
      default: hipLaunchKernelGGL((join_traverse_kernel<16>), dim3((unsigned)n_act), dim3(64), 0, s, ta); break;
    }
    JOIN_HIP(hipGetLastError());
    return 0;
  };
  while (!active.empty()) {                                                                 // :299
    ++iterations;
    const int n_active = (int)active.size();
    const int min_target = k * alpha;
    std::vector<int32_t> fb;                            // queries the host heap has to traverse
    if (dev_trav) {
      // (the list on the device is this round's in any case: the traversal launched behind this round's join reads it)
      // (lists go host -> pinned -> a copy kernel: a workgroup that reads its query number over PCIe starts 2 us late, 20 workgroups
      // deep per CU that was +70 us on the join kernel; nothing of the previous round is in flight: it ended with a synchronisation)
      memcpy(h_active, active.data(), sizeof(int32_t) * (size_t)n_active);
      hipLaunchKernelGGL(join_copy_kernel, dim3((unsigned)((n_active + 255) / 256)), dim3(256), 0, s, (const uint32_t*)p_active, (uint32_t*)d_active, (size_t)n_active);
      if (!spec_valid) {
        if (int rc = launch_traverse(n_active, min_target)) return rc;
        mark("traversal enqueued");
        JOIN_HIP(hipStreamSynchronize(s));
        mark("traversal synchronised");
        for (int x = 0; x < n_active; ++x) spec_index[(size_t)active[x]] = x;
      }
      spec_valid = false;
      // the host's libm decides: the reference's expression at the proposed stop and one step before it
      const int stat_size = (int)j->h_stats[(size_t)cells];
      for (int x = 0; x < n_active; ++x) {
        const int q = active[x];
        const int32_t* sm = h_summary + (size_t)spec_index[(size_t)q] * TRAV_SUM_DW;
        const int n = sm[0];
        float Pn, Pm, Cn, Cm;
        memcpy(&Pn, &sm[4], 4); memcpy(&Pm, &sm[5], 4); memcpy(&Cn, &sm[6], 4); memcpy(&Cm, &sm[7], 4);
        bool ok = !(sm[3] & 1) && n >= 0 && n <= cells;
        const float margin = j->libm_margin;
        if (ok && n < cells) {
          ok = !(Cn < confidence);
          if (!(fabsf(Cn - confidence) > margin)) { ok = !(join_confidence_hyp(min_target, (int)n_targets, Pn, stat_size) < confidence); ++j->track.libm_checks; }
        }
        if (ok && n > 0) {
          ok = Cm < confidence;
          if (!(fabsf(Cm - confidence) > margin)) { ok = join_confidence_hyp(min_target, (int)n_targets, Pm, stat_size) < confidence; ++j->track.libm_checks; }
        }
        q_host[q] = ok ? 0 : 1;
        if (ok) { q_n[q] = n; q_rows[q] = sm[2]; q_exh[q] = n >= cells; }
        if (!ok && (sm[3] & 1) && !host_sides_all && (side_slot.empty() || side_slot[(size_t)q] < 0)) {
          // handed back by the device with its sub-distances: the two sides sorted here (stable: equal distances keep their code
          // order, index_utils.c:306-320) -- what side_sort_kernel would have produced
          if (side_slot.empty()) side_slot.assign((size_t)Q, -1);
          const size_t row = (size_t)2 * Kc, base = sub.size() / row;
          sub.resize((base + 1) * row); sides.resize((base + 1) * row);
          const float* fs = h_fbsub + (size_t)spec_index[(size_t)q] * row;
          memcpy(sub.data() + base * row, fs, sizeof(float) * row);
          for (int sd = 0; sd < 2; ++sd) {
            JoinSide* out = sides.data() + base * row + (size_t)sd * Kc;
            for (int c = 0; c < Kc; ++c) { out[c].dist = fs[(size_t)sd * Kc + c]; out[c].code = c; }
            std::stable_sort(out, out + Kc, [](const JoinSide& u, const JoinSide& v) {   // (the kernel's key: the distance's bit pattern, then the code)
              uint32_t ub, vb; memcpy(&ub, &u.dist, 4); memcpy(&vb, &v.dist, 4); return ub < vb; });
          }
          side_slot[(size_t)q] = (int32_t)base;
        }
      }
      for (int q : active) if (q_host[q]) fb.push_back(q);
      mark("summaries checked");
    } else {
      fb = active;
    }
    if (!fb.empty()) {
      if (int rc = fetch_sides(fb)) return rc;
      for (int q : fb) q_host[q] = 1;
      join_parallel_for((int)fb.size(), [&](int lo, int hi, int) {                            // :327-331
        JoinTraversal w;
        for (int x = lo; x < hi; ++x) {
          const int q = fb[x];
          qcells[q].clear();
          const size_t sl = (size_t)side_slot[(size_t)q];
          const bool exhausted = join_select_cells(sides.data() + (sl * 2) * Kc, sides.data() + (sl * 2 + 1) * Kc,
                                                   sub.data() + (sl * 2) * Kc, sub.data() + (sl * 2 + 1) * Kc, Kc,
                                                   j->h_stats.data(), (int)n_targets, min_target, confidence, w, qcells[q]);
          q_exh[q] = exhausted ? 1 : 0;
          int64_t cnt = 0;
          for (int32_t c : qcells[q]) cnt += tcell_off[c + 1] - tcell_off[c];
          q_rows[q] = (int)cnt;
          q_n[q] = (int)qcells[q].size();
        }
      });
    }
    j->track.host_traversals += (int64_t)fb.size();
    bool last = true;
    for (int q : active) if (!q_exh[q]) { last = false; break; }
    track(&freddy_track::determine_coarse_quantization_time);
    // targetCounts (:459) and the target-list skip rule (:553-557)
    scan.clear(); scan_fb.clear(); qoff.assign(1, 0); flat.clear();
    for (int x = 0; x < n_active; ++x) {
      const int q = active[x];
      target_count[q] += q_rows[q];
      if (use_tl && target_count[q] < k * alpha_original && !last) { target_count[q] = 0; continue; }
      j->track.candidate_rows += q_rows[q];
      if (!q_host[q]) { scan.push_back(q); continue; }
      scan_fb.push_back(q);
      for (int32_t c : qcells[q]) if (tcell_off[c + 1] > tcell_off[c]) flat.push_back(c);
      qoff.push_back((int32_t)flat.size());
    }
    // longest first: a query's workgroup is a chain whose length grows with its target rows (a few queries have ten times
    // the average), and the launch ends with whatever was started last -- counting sort on rows / 128, descending
    if (scan.size() > 256) {
      constexpr int NBK = 64;
      int cnt[NBK + 1] = {0};
      auto bucket = [&](int q) { const int b = q_rows[q] >> 7; return NBK - 1 - (b < NBK ? b : NBK - 1); };
      for (int q : scan) ++cnt[bucket(q) + 1];
      for (int b = 0; b < NBK; ++b) cnt[b + 1] += cnt[b];
      std::vector<int32_t> sorted(scan.size());
      for (int q : scan) sorted[(size_t)cnt[bucket(q)]++] = q;
      scan.swap(sorted);
    }
    const int n_dev = (int)scan.size(), n_fb = (int)scan_fb.size(), n_scan = n_dev + n_fb;
    scan.insert(scan.end(), scan_fb.begin(), scan_fb.end());
    mark("scan list built");
    track(&freddy_track::query_construction_time);
    if (n_scan > 0) {
      memcpy(h_scan, scan.data(), sizeof(int32_t) * (size_t)n_scan);
      hipLaunchKernelGGL(join_copy_kernel, dim3((unsigned)((n_scan + 255) / 256)), dim3(256), 0, s, (const uint32_t*)p_scan, (uint32_t*)d_scan, (size_t)n_scan);
      JoinArgs a;
      a.queries = (const float*)d_q; a.tcell_off = (const int32_t*)d_tcell; a.trow = (const int32_t*)d_trow;
      a.ids = j->ids; a.codes = j->codes; a.MP = j->MP; a.vectors = j->vectors; a.cbT = j->cbT;
      a.d = d; a.m = m; a.K = K; a.S = j->S; a.k = k; a.L = L; a.method = method; a.double_codes = double_codes ? 1 : 0;
      if (big) {
        void *bk = nullptr, *be = nullptr;
        if (join_buf(j, 16, sizeof(u64) * (size_t)n_scan * L, &bk) || join_buf(j, 17, sizeof(float) * (size_t)n_scan * L, &be)) return FREDDY_E_NOMEM;
        a.big_keys = (u64*)bk; a.big_exact = (float*)be;
      }
      if (!j->ev0) { JOIN_HIP(hipEventCreate(&j->ev0)); JOIN_HIP(hipEventCreate(&j->ev1)); }
      JOIN_HIP(hipEventRecord(j->ev0, s));
      // (a separate launch for the host-traversed queries ran behind the main one -- a lone workgroup's 45 us -- and its two
      // list uploads were SDMA hops: a query with a tie cost the call 0.1 ms)
      const bool fb_rows = dev_trav && n_fb > 0 && j->h_q && (size_t)n_fb * (size_t)(cells + 1) * sizeof(int32_t) <= j->h_q_cap;
      if (fb_rows) {
        int32_t* hf = static_cast<int32_t*>(j->h_q);   // (the query staging block: its copy kernels finished before the first synchronisation)
        for (int x = 0; x < n_fb; ++x) {
          int32_t* row = hf + (size_t)x * (cells + 1);
          const int cnt = qoff[(size_t)x + 1] - qoff[(size_t)x];
          row[0] = cnt;
          memcpy(row + 1, flat.data() + qoff[(size_t)x], sizeof(int32_t) * (size_t)cnt);
        }
        hipLaunchKernelGGL(join_fb_rows_kernel, dim3((unsigned)n_fb), dim3(256), 0, s, (const int32_t*)j->h_q, (const int32_t*)d_scan + n_dev,
                           (int32_t*)d_qstrided, (int32_t*)d_qcnt, cells);
        JOIN_HIP(hipGetLastError());
      }
      if (n_dev > 0 || fb_rows) {     // cell lists written by the traversal kernel (and join_fb_rows_kernel): row q of [Q][cells]
        a.scan_query = (const int32_t*)d_scan; a.qcell_off = nullptr; a.qcell_cnt = (const int32_t*)d_qcnt; a.qstride = cells;
        a.qcells = (const int32_t*)d_qstrided; a.out_ids = p_oi; a.out_dist = p_od;
        if (int rc = join_launch(s, a, fb_rows ? n_scan : n_dev, V, lds)) return rc;
      }
      if (n_fb > 0 && !fb_rows) {      // host-traversed queries: flat lists with offsets
        if (join_buf(j, 6, sizeof(int32_t) * std::max<size_t>(flat.size(), 1), &d_qcells)) return FREDDY_E_NOMEM;
        JOIN_HIP(hipMemcpyAsync(d_qoff, qoff.data(), sizeof(int32_t) * (n_fb + 1), hipMemcpyHostToDevice, s));
        if (!flat.empty()) JOIN_HIP(hipMemcpyAsync(d_qcells, flat.data(), sizeof(int32_t) * flat.size(), hipMemcpyHostToDevice, s));
        a.scan_query = (const int32_t*)d_scan + n_dev; a.qcell_off = (const int32_t*)d_qoff; a.qcell_cnt = nullptr; a.qstride = 0;
        a.qcells = (const int32_t*)d_qcells; a.out_ids = p_oi + (size_t)n_dev * k; a.out_dist = p_od + (size_t)n_dev * k;
        if (int rc = join_launch(s, a, n_fb, V, lds)) return rc;
      }
      JOIN_HIP(hipEventRecord(j->ev1, s));
      if (dev_trav && !last && (int64_t)k * alpha * 2 < INT32_MAX) {   // the next round's cells for everyone still active (see launch_traverse)
        if (int rc = launch_traverse(n_active, k * (alpha + alpha))) return rc;
        for (int x = 0; x < n_active; ++x) spec_index[(size_t)active[x]] = x;
        spec_valid = true;
      }
      mark("join (+ next traversal) enqueued");
      JOIN_HIP(hipStreamSynchronize(s));
      mark("join synchronised");
      { float ms = 0.0f; if (hipEventElapsedTime(&ms, j->ev0, j->ev1) == hipSuccess) j->track.join_kernel_time += 1e-3 * ms; }
      for (int x = 0; x < n_scan; ++x) {
        memcpy(out_ids + (size_t)scan[x] * k, h_oi_p + (size_t)x * k, sizeof(int32_t) * k);
        memcpy(out_dist + (size_t)scan[x] * k, h_od_p + (size_t)x * k, sizeof(float) * k);
      }
    }
    mark("lists copied out");
    track(&freddy_track::computation_time);   // LUTs, ADC / exact distances, post verification: one kernel
    if (!last) {                                                                            // :639-669
      std::vector<int32_t> next;
      for (int q : active) {
        if (out_dist[(size_t)q * k + k - 1] == JOIN_MAX_DIST) {
          for (int i = 0; i < k; ++i) { out_ids[(size_t)q * k + i] = -1; out_dist[(size_t)q * k + i] = JOIN_MAX_DIST; }
          next.push_back(q);
        }
      }
      active.swap(next);
    } else {
      active.clear();
    }
    alpha += alpha;                                                                         // :680
    track(&freddy_track::recalculate_query_indices_time);
  }
  if (!tl_hit) j->tl_valid = true;   // (the offsets arrived with the first synchronisation)
  j->track.iterations = iterations;
  j->track.total_time = std::chrono::duration<double>(now() - t_start).count();
  if (iterations_out) *iterations_out = iterations;
  return 0;
}

}  // namespace freddy
