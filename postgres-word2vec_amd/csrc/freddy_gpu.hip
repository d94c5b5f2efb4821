// freddy_gpu.hip -- C ABI (include/freddy_gpu.h) over the HIP kernels in kernels.h.
//
// Host-side responsibilities only: lay the pinned tables out for the kernels, size
// workspaces, order the launches of a probing round on one HIP stream, and run the
// (rare) extra rounds of the reference's "while (foundInstances < k)" loop
// (freddy.c:262, :835).  No arithmetic that influences a result happens on the host
// for the PQ / IVFADC calls.
#include "../../include/freddy_gpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"
#include "scan_common.h"
#include "fused3.h"
#include "fused5.h"
#include "one.h"
#include "sparse5.h"
#include "coarse.h"
#include "exact.h"
#include "exact2.h"
#include "join.h"

using namespace freddy;

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* freddy_gpu_last_error(void) { return g_err; }

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(FREDDY_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                  __FILE__, __LINE__);                                                      \
  } while (0)

// ---------------------------------------------------------------------------------------
// index object
// ---------------------------------------------------------------------------------------
enum { KIND_PQ = 1, KIND_IVF = 2, KIND_IVPQ = 3, KIND_VEC = 4 };

// Options.  Read from the environment ONCE, when an index is pinned (never on the search path); freddy_gpu_set_option
// changes them on a pinned index.  None of them changes a result: every setting gives the same lists.  The first group
// is for deployments, the second selects the alternative paths the tests compare with each other (every one of them has a
// GPU test), the third the self-checks.  (INTEGRATION.md lists them; timing experiments of earlier rounds are gone --
// profiles/HISTORY.md has their numbers.)
struct Tuning {
  // -- deployment
  int scan_share = 1;          // FREDDY_GPU_SCAN_SHARE: the batches the CALLER keeps in flight on this handle through the *_dev entry points
                               // (one stream each): a persistent scan takes n_cus / share CUs so that the scans run side by side (DESIGN.md
                               // 5.1).  An explicit contract -- the library does not guess it; the host-buffer calls pass their own lane count
  int reserve_cus = 0;         // FREDDY_GPU_RESERVE_CUS: CUs the persistent scan leaves to the kernels of other streams (RCCL beside the scans)
  int pipeline_batch = 2048;   // FREDDY_GPU_PIPELINE_BATCH: queries per sub-batch of the host-buffer pipeline (freddy_gpu_ivfadc_search)
  int pipeline_lanes = 4;      // FREDDY_GPU_PIPELINE_LANES: sub-batches in flight inside one host-buffer call (1..4)
  int64_t lut_budget_mb = 8192;      // FREDDY_GPU_LUT_BUDGET_MB: per-call workspace cap (queries are chunked to fit); 288 GB of HBM: 8 GiB = 12 800 queries at nprobe 10
  // -- path selection (tests)
  int fused = -1;              // FREDDY_GPU_FUSED: -1 auto (cell-grouped scans for >= 256 items), 0 generic kernels, 1 always
  int scan_kernel = 5;         // FREDDY_GPU_FUSED_KERNEL: 5 filter + refine on int16 slabs (fused5.h), 3 the reference's arithmetic for every row (fused3.h)
  int coarse_approx = 1;       // FREDDY_GPU_COARSE_APPROX: cell selection as filter + refine (coarse.h); 0 = every distance exact
  int one_launch = 1;          // FREDDY_GPU_ONE_LAUNCH: a single query through the host-buffer calls as ONE launch (one.h) instead of a chain
  int pq_fused = -1;           // FREDDY_GPU_PQ_FUSED: batches over the flat PQ table through the cell-grouped filter + refine scan: -1 = from 16 queries on, 0 never, 1 always
  int sparse_items = 2;        // FREDDY_GPU_SPARSE_ITEMS: cells that at most this many queries of a batch probe are scanned item by item (sparse5.h) instead of as cell-grouped work entries (0 = never, < 0 = always for cells of up to that many items)
  int codes_u8 = 1;            // FREDDY_GPU_CODES_U8: K <= 256: the integer-slab scans read one byte per code (packed8, 16 instead of 28 B per row); 0 = the int16 layout
  int exact_filter = -1;       // FREDDY_GPU_EXACT_FILTER: exact kNN as MFMA filter + exact refine (exact2.h): -1 auto (tables of >= 8192 rows, k <= 32), 0 never, 1 always
  // -- self-checks (tests): bit 0 = the scan keeps every row and the merge refines every row (every probed row's bracket is checked),
  //    bit 1 = the cell selection refines every cell, bit 2 = exact kNN refines every row
  int check_brackets = 0;
#ifdef FREDDY_LAB
  int scan_prof = 0;           // FREDDY_GPU_FUSED_PROF (lab builds only): per-phase cycle sums of the scan kernel on stderr
#endif
};
static int64_t env_int(const char* name, int64_t dflt) {
  const char* e = getenv(name);
  return (e && *e) ? (int64_t)strtoll(e, nullptr, 10) : dflt;
}
static Tuning read_tuning() {
  Tuning t;
  t.fused = (int)env_int("FREDDY_GPU_FUSED", t.fused);
  t.scan_kernel = (int)env_int("FREDDY_GPU_FUSED_KERNEL", t.scan_kernel);
  t.reserve_cus = (int)env_int("FREDDY_GPU_RESERVE_CUS", 0);
  t.scan_share = (int)std::max<int64_t>(1, env_int("FREDDY_GPU_SCAN_SHARE", t.scan_share));
  t.pipeline_batch = (int)std::max<int64_t>(16, env_int("FREDDY_GPU_PIPELINE_BATCH", t.pipeline_batch));
  t.pipeline_lanes = (int)std::min<int64_t>(4, std::max<int64_t>(1, env_int("FREDDY_GPU_PIPELINE_LANES", t.pipeline_lanes)));
  t.pq_fused = (int)env_int("FREDDY_GPU_PQ_FUSED", t.pq_fused);
  t.one_launch = (int)env_int("FREDDY_GPU_ONE_LAUNCH", t.one_launch);
  t.coarse_approx = (int)env_int("FREDDY_GPU_COARSE_APPROX", 1);
  t.sparse_items = (int)env_int("FREDDY_GPU_SPARSE_ITEMS", t.sparse_items);
  t.exact_filter = (int)env_int("FREDDY_GPU_EXACT_FILTER", t.exact_filter);
  t.codes_u8 = (int)env_int("FREDDY_GPU_CODES_U8", t.codes_u8);
  t.lut_budget_mb = std::max<int64_t>(1, env_int("FREDDY_GPU_LUT_BUDGET_MB", t.lut_budget_mb));
#ifdef FREDDY_LAB
  t.scan_prof = getenv("FREDDY_GPU_FUSED_PROF") != nullptr;
#endif
  return t;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Everything a search writes besides its outputs.  Keyed by the stream the search is enqueued on, so that two
// batches in flight on two streams (the front end of batch i+1 beside the merge of batch i) never share scratch.
struct Workspace {
  bool used = false;
  hipStream_t owner = nullptr;
  uint64_t last_use = 0;           // claim order (the slot a new stream takes over is the least recently used one)
  DevBuf w_q, w_distT, w_used, w_item_cell, w_item_query, w_rows, w_resid, w_lut,
      w_part, w_cand, w_found, w_act0, w_act1, w_cnt, w_out_ids, w_out_dist, w_sub_rows,
      w_sub_packed, w_sub_pos, w_sub_blk, w_cellcnt, w_sorted, w_groups, w_surv, w_surv_cnt, w_prof, w_qc, w_qn, w_records, w_qn2, w_item_dist, w_tmin, w_one, w_oneb;
  uint64_t one_shape = 0;          // the one-launch kernels' buffer (w_oneb): shape of the call that wrote it last, and that call's epoch (one.h)
  uint32_t one_epoch = 0;
  void release() {
    DevBuf* bufs[] = {&w_q, &w_distT, &w_used, &w_item_cell, &w_item_query, &w_rows, &w_resid, &w_lut, &w_part,
                      &w_cand, &w_found, &w_act0, &w_act1, &w_cnt, &w_out_ids, &w_out_dist, &w_sub_rows, &w_sub_packed,
                      &w_sub_pos, &w_sub_blk, &w_cellcnt, &w_sorted, &w_groups, &w_surv, &w_surv_cnt, &w_prof, &w_qc,
                      &w_qn, &w_records, &w_qn2, &w_item_dist, &w_tmin, &w_one, &w_oneb};
    for (DevBuf* b : bufs) b->release();
    used = false;
    owner = nullptr;
  }
};
static constexpr int FREDDY_MAX_WS = 12;

// One lane of the host-buffer pipeline (freddy_gpu_ivfadc_search): a library-owned stream, pinned staging for the
// queries going in and the lists coming out, device buffers, and the state of the sub-batch it has in flight.
// State of one chunk of queries while its probing rounds are enqueued.
struct IvfRun {
  freddy_gpu_index* ix;
  Workspace* ws;
  hipStream_t s;       // the stream the search is enqueued on
  int share;           // batches in flight on this handle (the scan takes n_cus / share CUs)
  const float* d_q;
  int Q, k, W, L, found_rule, upi;
  float sentinel, cell_limit;
  int32_t *d_out_ids, *d_status;
  float* d_out_dist;
  bool fused;          // cell-grouped scans (fused3.h / fused4.h) instead of lut_build + adc_scan
  int scan_kernel;     // 5: filter + refine, 3: exact fused scan
  bool tiled;          // batch coarse kernels (tiles of queries)
  bool zeroed;         // the coarse kernel has cleared the round-one scratch (ZeroArgs): no memsets in round one
  bool approx;         // cell selection as filter + refine: MFMA distances with a proven bracket, exact ones for the candidates
  bool records_ready;  // a batch over the flat PQ table: the entry records were written by pq_records_kernel (no work-table / record kernels)
  int merge_slices;    // > 0: the merge of such a batch as `merge_slices` partial merges per query + merge_replay_kernel
  // per round
  int n_active, round;
  const int32_t* active;
  int32_t* next;
  bool first() const { return round == 0; }
};

struct LaneSlot {
  hipEvent_t done = nullptr;
  void* h_in = nullptr;  size_t h_in_cap = 0;    // pinned: queries of the sub-batch
  void* h_out = nullptr; size_t h_out_cap = 0;   // pinned: [ids n*k][dist n*k][n_next][unfinished queries n]
  DevBuf d_q, d_ids, d_dist;
  bool busy = false;
  int q0 = 0, n = 0;
};
struct Lane {
  hipStream_t stream = nullptr;
  LaneSlot slot[2];        // two sub-batches queued per lane: the stream never runs dry while the host stages the next one
};
static constexpr int FREDDY_LANES = 4;

struct ProfRec {
  int64_t launches = 0;
  double ms = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> open;
};

struct freddy_gpu_index {
  int kind = 0;
  int device = 0;
  Tuning tune;
  hipStream_t stream = nullptr;
  int d = 0, m = 0, K = 0, C = 0, S = 0, M2 = 0;
  int64_t N = 0;
  int64_t n_blocks = 0;
  int max_list_blocks = 0;
  int64_t bytes = 0;
  int last_Q = 0;
  int n_cus = 256;
  // pinned tables
  float* coarse = nullptr;      // [C][d]
  float* coarseT = nullptr;     // [d][Cpad] for the coarse-distance kernel
  float* coarseP = nullptr;     // centroids in MFMA fragment order [Cpad/32][dp/8][64][4], zero padded (coarse.h)
  void* coarseH = nullptr;      // the centroids split into f16 hi / lo, [Cpad/32][T][2][64][8] (coarse_approx16_body)
  int coarse_ec = 0;            // their power-of-two scale
  float* cn2 = nullptr;         // [Cpad] |c_j|^2
  float cmax = 0.0f;            // max_j |c_j|, rounded up
  int dp = 0;
  int Cpad = 0;
  float* cbT = nullptr;         // [m][S][K]
  float* cbP = nullptr;         // fused kernel layout [m][SP/4][512 slots][4 dims][2 codes] (NULL unless K <= 1024)
  // filter + refine path (fused4.h); NULL unless the shape is the fused one and the table fits the budget
  float* cbR = nullptr;         // [m][K][S] row-major codebook for the exact stage
  float* rterm = nullptr;       // [blocks*64] sum_p (|c|^2 + 2 co_p . c) of every row
  float* pmax = nullptr;        // [m]        max |co_p| + max |c_p|, rounded up
  float* cmaxp = nullptr;       // [m]        max |c_p|, rounded up
  float* cbF = nullptr;         // [m][8 groups][7 steps][64 lanes][8] the codebook in the B-fragment order of the table kernel's matrix instructions (fused5.h query_codebook5_body)
  int32_t* viol = nullptr;      // [4] self-check counters: scan bracket violations / rows checked, coarse bracket violations / cells checked
  int32_t* blk_cell = nullptr;  // [blocks]   list of every row block
  int32_t* list_off = nullptr;  // [lists+1] rows
  int32_t* blk_off = nullptr;   // [lists+1] row blocks
  uint32_t* packed = nullptr;   // [blocks][M2][64]
  uint32_t* packed8 = nullptr;  // K <= 256, m = 12: [blocks][3][64], one BYTE per code -- what the integer-slab scans read (16 B per row with its row term)
  bool packed8_own = false;     // (a PQ handle's view shares its owner's array)
  int32_t* pos = nullptr;       // [blocks*64]
  int32_t* ids = nullptr;       // PQ: [N] position -> id
  std::vector<int32_t> h_ids;   // PQ: ascending ids for "id IN (...)" resolution
  std::vector<int32_t> h_list_off;
  std::vector<float> h_coarse;  // IVF: [C][d], kept for the norm bounds of a replaced codebook
  int32_t max_id = -1;          // largest row id pinned (appended rows must be larger)
  // raw vectors (exact kNN): 64-row blocks [block][d][64]
  float* xb = nullptr;
  // exact kNN as filter + refine (exact2.h): the table's statistics (pin time / append) and the per-call buffers
  bool exf_ok = false;          // every element finite, d % 4 == 0, d <= 512
  float exf_xnorm = 0.0f;       // largest row norm, rounded up
  int exf_ex = 0;               // power-of-two scale of the rows for the f16 split
  DevBuf exf_qfrag, exf_small, exf_sample, exf_cand;
  DevBuf exf_xf;                // the rows in MFMA A-fragment order, scaled and split into f16 hi / lo (exf_layout_kernel)
  int64_t exf_xf_strips = 0;    // 32-row strips laid out (capacity is exf_xf.cap)
  // ivpq extras
  JoinIndex join;
  // flat PQ table through the cell-grouped scan (pq_shadow_build): an IVF-shaped view of this table -- pseudo-lists of
  // 4096 consecutive rows, zero centroids -- that shares packed / codebook tables with its owner
  freddy_gpu_index* pq_shadow = nullptr;
  freddy_gpu_index* pq_sub_view = nullptr; // the same for the rows of an "id IN (...)" subset, refreshed by every such call
  freddy_gpu_index* shadow_of = nullptr;   // set in the shadow: profile records and shared arrays belong to this index
  DevBuf v_coarse, v_list_off, v_blk_off, v_blk_cell, v_pos, v_rterm;   // a shadow's own arrays (grown on demand)
  // workspaces: one per stream the caller searches on (searches on different streams may overlap)
  Workspace ws[FREDDY_MAX_WS];
  Workspace* last_ws = nullptr;   // of the most recent search (freddy_gpu_last_* read its counters)
  uint64_t ws_clock = 0;
  std::mutex mu;                  // guards the workspace slots and the profile map (host threads on different streams)
  // host-buffer pipeline (created by the first host-buffer IVFADC call)
  Lane lanes[FREDDY_LANES];
  // pinned staging of the other synchronous host-buffer calls (pq_search): queries in, lists out -- read / written by
  // kernels, no SDMA copies in the stream
  void* hio_in = nullptr;  size_t hio_in_cap = 0;
  void* hio_out = nullptr; size_t hio_out_cap = 0;
  // replicas of this index on further devices (freddy_gpu_pin_ivf_multi): a host batch is split contiguously over
  // this handle and its replicas; every replica is a complete pinned index of its own
  std::vector<freddy_gpu_index*> replicas;
  // set when a mutation (append_rows / update_codebook / set_option) failed after it had already changed some of the devices
  // behind this handle: the replicas no longer hold the same tables, so every search fails loudly until the handle is unpinned
  bool poisoned = false;
  bool one_launch_failed = false;   // pq_one_kernel once ran out of its bounded polls on this handle: three launches from then on
  // profiling
  bool profiling = false;
  std::map<std::string, ProfRec> prof;
};

template <class F>
static inline void timed_launch(freddy_gpu_index* ix, hipStream_t s, const char* name, F&& f) {
  if (ix->shadow_of) ix = ix->shadow_of;
  if (!ix->profiling) { f(); return; }
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a, s);
  f();
  (void)hipEventRecord(b, s);
  std::lock_guard<std::mutex> lock(ix->mu);
  ProfRec& r = ix->prof[name];
  r.launches++;
  r.open.emplace_back(a, b);
}

// The workspace of the stream a search is enqueued on.  With every slot taken a new stream takes over the least
// recently used one -- after the whole device has drained (rare; no handle of a possibly destroyed caller stream is touched).
static Workspace* workspace_for(freddy_gpu_index* ix, hipStream_t s) {
  std::lock_guard<std::mutex> lock(ix->mu);
  Workspace* w = nullptr;
  for (Workspace& c : ix->ws)
    if (c.used && c.owner == s) { w = &c; break; }
  if (!w)
    for (Workspace& c : ix->ws)
      if (!c.used) { c.used = true; c.owner = s; w = &c; break; }
  if (!w) {
    w = &ix->ws[0];
    for (Workspace& c : ix->ws)
      if (c.last_use < w->last_use) w = &c;
    (void)hipDeviceSynchronize();
    w->owner = s;
  }
  w->last_use = ++ix->ws_clock;
  ix->last_ws = w;
  return w;
}

template <class T>
static int upload(T** dst, const T* src, size_t n, int64_t* bytes) {
  *dst = nullptr;
  size_t sz = sizeof(T) * (n ? n : 1);
  if (hipMalloc((void**)dst, sz) != hipSuccess) return -1;
  if (n && hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice) != hipSuccess) return -2;
  if (bytes) *bytes += (int64_t)sz;
  return 0;
}

static void free_index(freddy_gpu_index* ix) {
  if (!ix) return;
  for (freddy_gpu_index* r : ix->replicas) free_index(r);
  ix->replicas.clear();
  (void)hipSetDevice(ix->device);
  (void)hipDeviceSynchronize();   // (every stream that searched on this handle, without touching a caller's stream handle)
  for (Workspace& w : ix->ws) w.release();
  for (DevBuf* b : {&ix->exf_qfrag, &ix->exf_small, &ix->exf_sample, &ix->exf_cand, &ix->exf_xf}) b->release();
  if (ix->hio_in) { (void)hipHostFree(ix->hio_in); ix->hio_in = nullptr; ix->hio_in_cap = 0; }
  if (ix->hio_out) { (void)hipHostFree(ix->hio_out); ix->hio_out = nullptr; ix->hio_out_cap = 0; }
  for (Lane& l : ix->lanes) {
    if (l.stream) (void)hipStreamDestroy(l.stream);
    for (LaneSlot& c : l.slot) {
      if (c.done) (void)hipEventDestroy(c.done);
      if (c.h_in) (void)hipHostFree(c.h_in);
      if (c.h_out) (void)hipHostFree(c.h_out);
      c.d_q.release(); c.d_ids.release(); c.d_dist.release();
      c = LaneSlot();
    }
    l.stream = nullptr;
  }
  if (ix->shadow_of) {   // a PQ table's IVF-shaped view: its own arrays only (packed, codebook tables and the stream are the owner's)
    DevBuf* own[] = {&ix->v_coarse, &ix->v_list_off, &ix->v_blk_off, &ix->v_blk_cell, &ix->v_pos, &ix->v_rterm};
    for (DevBuf* b : own) b->release();
    if (ix->viol) (void)hipFree(ix->viol);
    delete ix;
    return;
  }
  if (ix->pq_shadow) { free_index(ix->pq_shadow); ix->pq_shadow = nullptr; }
  if (ix->pq_sub_view) { free_index(ix->pq_sub_view); ix->pq_sub_view = nullptr; }
  void* ptrs[] = {ix->xb, ix->coarse, ix->coarseT, ix->coarseP, ix->coarseH, ix->cn2, ix->cbT, ix->cbP, ix->cbR, ix->rterm, ix->pmax, ix->cmaxp, ix->cbF, ix->viol, ix->blk_cell, ix->list_off, ix->blk_off, ix->packed, ix->pos, ix->ids, ix->packed8_own ? ix->packed8 : nullptr};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  join_free(&ix->join);
  for (auto& kv : ix->prof)
    for (auto& ev : kv.second.open) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (ix->stream) (void)hipStreamDestroy(ix->stream);
  delete ix;
}

// transpose codebook [m][K][S] -> [m][S][K]
static std::vector<float> transpose_codebook(const float* cb, int m, int K, int S) {
  std::vector<float> t((size_t)m * S * K);
  for (int p = 0; p < m; ++p)
    for (int c = 0; c < K; ++c)
      for (int j = 0; j < S; ++j) t[((size_t)p * S + j) * K + c] = cb[((size_t)p * K + c) * S + j];
  return t;
}

// Pack rows of `n_lists` inverted lists into 64-row blocks: [block][M2][64] dwords, two
// int16 codes per dword, plus one scan-position dword per row (-1 on padding rows).
// Which rows share a 16-lane group of a 64-row block decides what the scan kernels' LDS gathers cost: a
// wave-level ds_read_b128 of slab rows takes ~2.4 + 4 x (largest number of lanes of a 16-lane group whose
// rows' codes agree modulo 16 = the same LDS bank group) cycles (tools/ubench6: 14.6 cycles for random
// rows, 6.4 without collisions).  The order of the rows inside a list is free (results are ordered by id
// in the merge), so the rows of every group are picked greedily -- each next row from a window of 64
// candidates, the one that raises the per-position maxima least -- which brings the average maximum
// from 3.06 to ~2.1.  order[] = the list's rows in packing order.
static void arrange_list_rows(const int16_t* codes, int m, int64_t lo, int64_t hi, std::vector<int64_t>& order) {
  const int64_t n = hi - lo;
  order.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) order[(size_t)i] = lo + i;
  if (n <= 16 || m > 16) return;
  static const int WINDOW = (int)env_int("FREDDY_GPU_ARRANGE_WINDOW", 1024);   // candidates looked at for every pick (64: scan 103 us, 256: 101.7, 1024: 99.8; pin time 0.2 / 0.4 / 1.3 s for 3 M rows)
  int cnt[16][16], mx[16];
  for (int64_t k = 0; k < n; ++k) {
    if ((k & 15) == 0) { memset(cnt, 0, sizeof(cnt)); memset(mx, 0, sizeof(mx)); }
    const int64_t wend = std::min<int64_t>(n, k + WINDOW);
    int64_t best = k;
    int best_cost = INT32_MAX;
    for (int64_t j = k; j < wend; ++j) {
      const int16_t* row = codes + (size_t)order[(size_t)j] * m;
      int cost = 0;
      for (int p = 0; p < m; ++p) {
        const int c = cnt[p][row[p] & 15];
        cost += c + (c + 1 > mx[p] ? 100 : 0);
      }
      if (cost < best_cost) { best_cost = cost; best = j; }
    }
    std::swap(order[(size_t)k], order[(size_t)best]);
    const int16_t* row = codes + (size_t)order[(size_t)k] * m;
    for (int p = 0; p < m; ++p) {
      const int c = ++cnt[p][row[p] & 15];
      if (c > mx[p]) mx[p] = c;
    }
  }
  // The 16 rows picked together have to sit in the 16 lanes the LDS serves together -- and for ds_read_b128
  // those are NOT 16 consecutive lanes but {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
  // (MI355X_MICROARCH.md, LDS).  (Round 1 placed each group in consecutive lanes: every hardware group then
  // mixed the halves of two picked groups, and the arrangement bought 2 % instead of what tools/ubench6 promised.)
  static const int GROUP_LANES[64] = {0,  1,  2,  3,  12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27, 4,  5,  6,  7,  8,  9,
                                      10, 11, 16, 17, 18, 19, 28, 29, 30, 31, 32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55,
                                      56, 57, 58, 59, 36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63};
  std::vector<int64_t> blk(64);
  for (int64_t b0 = 0; b0 + 64 <= n; b0 += 64) {   // (a partial last block keeps its rows in the first lanes)
    for (int j = 0; j < 64; ++j) blk[(size_t)GROUP_LANES[j]] = order[(size_t)(b0 + j)];
    for (int j = 0; j < 64; ++j) order[(size_t)(b0 + j)] = blk[(size_t)j];
  }
}

static int pack_lists(freddy_gpu_index* ix, int n_lists, const int32_t* list_off, const int16_t* codes,
                      const int32_t* row_pos /*NULL: row index*/) {
  const int m = ix->m, K = ix->K, M2 = ix->M2;
  std::vector<int32_t> blk_off(n_lists + 1, 0);
  int max_blocks = 0;
  for (int c = 0; c < n_lists; ++c) {
    const int64_t len = (int64_t)list_off[c + 1] - list_off[c];
    if (len < 0) return fail(FREDDY_E_ARG, "list_off is not non-decreasing at list %d", c);
    const int nb = (int)((len + 63) / 64);
    blk_off[c + 1] = blk_off[c] + nb;
    max_blocks = std::max(max_blocks, nb);
  }
  const int64_t n_blocks = blk_off[n_lists];
  std::vector<uint32_t> packed((size_t)std::max<int64_t>(n_blocks, 1) * M2 * 64, 0u);
  std::vector<int32_t> pos((size_t)std::max<int64_t>(n_blocks, 1) * 64, -1);
  // (inverted lists only: the flat PQ table is addressed by row index)
  const bool arrange = row_pos != nullptr;
  std::vector<std::vector<int64_t>> orders(arrange ? (size_t)n_lists : 0);
  if (arrange) {
    std::atomic<int> next_list{0};
    auto worker = [&]() {
      for (int c = next_list.fetch_add(1); c < n_lists; c = next_list.fetch_add(1))
        arrange_list_rows(codes, m, list_off[c], list_off[c + 1], orders[(size_t)c]);
    };
    const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt && (int)t < n_lists; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  for (int c = 0; c < n_lists; ++c) {
    for (int64_t i = 0; i < (int64_t)list_off[c + 1] - list_off[c]; ++i) {
      const int64_t r = arrange ? orders[(size_t)c][(size_t)i] : list_off[c] + i;
      const int64_t b = blk_off[c] + i / 64;
      const int lane = (int)(i % 64);
      const int16_t* row = codes + (size_t)r * m;
      for (int l = 0; l < m; ++l) {
        if (row[l] < 0 || row[l] >= K)
          return fail(FREDDY_E_ARG, "code %d at row %lld position %d is outside [0,%d)", (int)row[l],
                      (long long)r, l, K);
      }
      for (int j = 0; j < M2; ++j) {
        const uint32_t lo = (uint16_t)row[2 * j];
        const uint32_t hi = (2 * j + 1 < m) ? (uint16_t)row[2 * j + 1] : 0u;
        packed[((size_t)b * M2 + j) * 64 + lane] = lo | (hi << 16);
      }
      pos[(size_t)b * 64 + lane] = row_pos ? row_pos[r] : (int32_t)r;
    }
  }
  std::vector<int32_t> blk_cell((size_t)std::max<int64_t>(n_blocks, 1), 0);
  for (int c = 0; c < n_lists; ++c)
    for (int b = blk_off[c]; b < blk_off[c + 1]; ++b) blk_cell[(size_t)b] = c;
  if (upload(&ix->blk_cell, blk_cell.data(), blk_cell.size(), &ix->bytes))
    return fail(FREDDY_E_NOMEM, "device allocation/copy failed while pinning the lists");
  ix->n_blocks = n_blocks;
  ix->max_list_blocks = max_blocks;
  ix->h_list_off.assign(list_off, list_off + n_lists + 1);
  if (upload(&ix->blk_off, blk_off.data(), blk_off.size(), &ix->bytes) ||
      upload(&ix->list_off, list_off, (size_t)n_lists + 1, &ix->bytes) ||
      upload(&ix->packed, packed.data(), packed.size(), &ix->bytes) ||
      upload(&ix->pos, pos.data(), pos.size(), &ix->bytes))
    return fail(FREDDY_E_NOMEM, "device allocation/copy failed while pinning the lists");
  return 0;
}

// packed[block][6][64] (two int16 codes per dword) -> packed8[block][3][64] (four one-byte codes per dword)
__global__ __launch_bounds__(256) void pack8_kernel(const uint32_t* __restrict__ packed, uint32_t* __restrict__ packed8, int64_t n_blocks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // (block, t, lane)
  if (i >= n_blocks * 3 * 64) return;
  const int lane = (int)(i & 63);
  const int64_t bt = i >> 6;
  const int t = (int)(bt % 3);
  const int64_t b = bt / 3;
  const uint32_t p0 = packed[((size_t)b * 6 + 2 * t) * 64 + lane], p1 = packed[((size_t)b * 6 + 2 * t + 1) * 64 + lane];
  packed8[i] = (p0 & 0xffu) | (((p0 >> 16) & 0xffu) << 8) | ((p1 & 0xffu) << 16) | (((p1 >> 16) & 0xffu) << 24);
}
// (Re)build the one-byte code array of a handle whose codes fit a byte (K <= 256, m = 12: the cell-grouped scans' shape).
static int build_packed8(freddy_gpu_index* ix) {
  if (ix->packed8 && ix->packed8_own) { (void)hipFree(ix->packed8); }
  ix->packed8 = nullptr; ix->packed8_own = false;
  if (ix->K > 256 || ix->m != 12 || ix->M2 != 6 || !ix->packed || ix->n_blocks <= 0) return 0;
  const size_t bytes = sizeof(uint32_t) * (size_t)ix->n_blocks * 3 * 64;
  if (hipMalloc((void**)&ix->packed8, bytes) != hipSuccess) { ix->packed8 = nullptr; return 0; }   // (no room: the int16 layout serves)
  ix->packed8_own = true;
  ix->bytes += (int64_t)bytes;
  const int64_t n = ix->n_blocks * 3 * 64;
  hipLaunchKernelGGL(pack8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ix->stream, ix->packed, ix->packed8, ix->n_blocks);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return 0;
}

// Kernels that want more than the default 64 KiB of dynamic LDS: the limit is a per-device function
// attribute, so it is raised once for every device an index is pinned on.
static int raise_lds_limits(int device) {
  static std::mutex mu;
  static std::vector<char> done;
  std::lock_guard<std::mutex> g(mu);
  if ((size_t)device < done.size() && done[(size_t)device]) return 0;
  const void* kernels[] = {
      (const void*)&adc_scan_kernel<12, 1>, (const void*)&adc_scan_kernel<12, 2>, (const void*)&adc_scan_kernel<12, 4>,
      (const void*)&adc_scan_kernel<12, 8>, (const void*)&adc_scan_kernel<12, 16>, (const void*)&adc_scan_kernel<0, 1>,
      (const void*)&adc_scan_kernel<0, 2>, (const void*)&adc_scan_kernel<0, 4>, (const void*)&adc_scan_kernel<0, 8>,
      (const void*)&adc_scan_kernel<0, 16>, (const void*)&ivf_spec2_kernel<25, 12, true>,
      (const void*)&ivf_spec2_kernel<25, 12, false>,
      (const void*)&ivf_filter5_kernel<12, true, false>, (const void*)&ivf_filter5_kernel<12, false, false>,
      (const void*)&ivf_filter5_kernel<12, true, true>, (const void*)&ivf_filter5_kernel<12, false, true>,
      (const void*)&ivf_filter5_kernel<12, false, false, false, true>, (const void*)&ivf_filter5_kernel<12, false, true, false, true>,
#ifdef FREDDY_LAB
      (const void*)&ivf_filter5_kernel<12, true, false, true>,
#endif
      (const void*)&grouping_kernel<6>, (const void*)&grouping_kernel<15>,
      (const void*)&grouping_kernel<0>, (const void*)&coarse_approx_kernel, (const void*)&coarse_approx16_kernel, (const void*)&join_query_kernel<1>, (const void*)&join_query_kernel<2>,
      (const void*)&join_query_kernel<4>, (const void*)&join_query_kernel<8>, (const void*)&join_query_kernel<16>,
      (const void*)&exf_filter_kernel<1, false>, (const void*)&exf_filter_kernel<2, false>, (const void*)&exf_filter_kernel<1, true>,
      (const void*)&exf_filter_kernel<2, true>};
  for (const void* k : kernels)
    HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  if (done.size() <= (size_t)device) done.resize((size_t)device + 1, 0);
  done[(size_t)device] = 1;
  return 0;
}

// Everything on the device that is a function of the (residual) codebook: the transposed copy of the generic
// LUT kernel, the paired layout of the exact fused scan, and -- for the filter + refine scan -- the row-major
// copy and the norm bounds.  (Re)built at pin time and by freddy_gpu_update_codebook; the row terms follow
// in refresh_row_terms once the rows are in place.
// Everything that is derived from the codebook.  The new tables are built beside the old ones and swapped in only when
// every upload has succeeded (freddy_gpu_update_codebook on a live handle: a failed call leaves the handle as it was).
static int derive_codebook_tables_into(freddy_gpu_index* ix, const float* codebook);
static int derive_codebook_tables(freddy_gpu_index* ix, const float* codebook) {
  float* const old[] = {ix->cbT, ix->cbP, ix->cbR, ix->pmax, ix->cmaxp, ix->cbF};
  const int64_t bytes_before = ix->bytes;
  ix->cbT = ix->cbP = ix->cbR = ix->pmax = ix->cmaxp = ix->cbF = nullptr;
  const int rc = derive_codebook_tables_into(ix, codebook);
  if (rc) {   // put the old tables back
    float* const fresh[] = {ix->cbT, ix->cbP, ix->cbR, ix->pmax, ix->cmaxp, ix->cbF};
    for (float* p : fresh) if (p) (void)hipFree(p);
    ix->cbT = old[0]; ix->cbP = old[1]; ix->cbR = old[2]; ix->pmax = old[3]; ix->cmaxp = old[4]; ix->cbF = old[5];
    ix->bytes = bytes_before;
    return rc;
  }
  int64_t old_bytes = 0;
  if (old[0]) old_bytes += (int64_t)sizeof(float) * ix->m * ix->S * ix->K;
  if (old[1]) old_bytes += (int64_t)sizeof(float) * ix->m * (((ix->S + 3) & ~3) / 4) * FUSED_T * 8;
  if (old[2]) old_bytes += (int64_t)sizeof(float) * ix->m * ix->K * ix->S;
  if (old[3]) old_bytes += (int64_t)sizeof(float) * ix->m;
  if (old[4]) old_bytes += (int64_t)sizeof(float) * ix->m;
  if (old[5]) old_bytes += (int64_t)sizeof(float) * ix->m * 8 * 7 * 64 * 8;
  ix->bytes -= old_bytes;       // (the footprint changes by the difference, not by a second copy)
  for (float* p : old) if (p) (void)hipFree(p);
  if (ix->kind == KIND_PQ) {    // views of the flat table are rebuilt from the new tables on next use
    if (ix->pq_shadow) { free_index(ix->pq_shadow); ix->pq_shadow = nullptr; }
    if (ix->pq_sub_view) { free_index(ix->pq_sub_view); ix->pq_sub_view = nullptr; }
  }
  return 0;
}
// The codebook in the order the table kernel's v_mfma_f32_16x16x4_f32 B operands are read (m = 12, S = 25, K <= 1024): for
// (position, group g of 16 code slots, step) lane l = (col = l & 15, kq = l >> 4) finds the eight values of dimension
// 4 step + kq for the codes 128 i + 16 g + col + 512 e, (i, e) = (0,0) (0,1) (1,0) ... (3,1), as two 16-byte words: a wave's
// load is 2 KB of consecutive bytes (the transposed copy gave 64-byte pieces of eight different lines).
static int build_fragment_codebook(freddy_gpu_index* ix, const float* codebook) {
  std::vector<float> f((size_t)ix->m * 8 * 7 * 64 * 8, 0.0f);
  for (int p = 0; p < ix->m; ++p)
    for (int g = 0; g < 8; ++g)
      for (int st = 0; st < 7; ++st)
        for (int l = 0; l < 64; ++l)
          for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 2; ++e) {
              const int j = 4 * st + (l >> 4), c = 128 * i + 16 * g + (l & 15) + 512 * e;
              if (j < ix->S && c < ix->K)
                f[((((size_t)p * 8 + g) * 7 + st) * 64 + l) * 8 + i * 2 + e] = codebook[((size_t)p * ix->K + c) * ix->S + j];
            }
  if (upload(&ix->cbF, f.data(), f.size(), &ix->bytes)) return fail(FREDDY_E_NOMEM, "device allocation failed");
  return 0;
}
static int derive_codebook_tables_into(freddy_gpu_index* ix, const float* codebook) {
  std::vector<float> cbT = transpose_codebook(codebook, ix->m, ix->K, ix->S);
  if (upload(&ix->cbT, cbT.data(), cbT.size(), &ix->bytes)) return fail(FREDDY_E_NOMEM, "device allocation failed");
  if (ix->kind == KIND_PQ) {
    // batches over the flat table take the cell-grouped filter + refine scan (pq_shadow_build): its codebook-derived tables,
    // with "centroids" that are zero
    if (ix->m == 12 && ix->S == 25 && ix->K <= FUSED_T * FUSED_E) {
      std::vector<float> cmaxp((size_t)ix->m);
      for (int p = 0; p < ix->m; ++p) {
        double cmax = 0.0;
        for (int c = 0; c < ix->K; ++c) {
          double n2 = 0.0;
          for (int j = 0; j < ix->S; ++j) { const double v = codebook[((size_t)p * ix->K + c) * ix->S + j]; n2 += v * v; }
          cmax = std::max(cmax, std::sqrt(n2));
        }
        cmaxp[p] = (float)(cmax * (1.0 + 1e-6));
      }
      if (upload(&ix->cbR, codebook, (size_t)ix->m * ix->K * ix->S, &ix->bytes) ||
          upload(&ix->pmax, cmaxp.data(), cmaxp.size(), &ix->bytes) ||
          upload(&ix->cmaxp, cmaxp.data(), cmaxp.size(), &ix->bytes))
        return fail(FREDDY_E_NOMEM, "device allocation failed");
      if (int rc = build_fragment_codebook(ix, codebook)) return rc;
    }
    return 0;
  }
  if (ix->kind != KIND_IVF) return 0;
  const int C = ix->C, d = ix->d;
  if (ix->K <= FUSED_T * FUSED_E) {
    // paired layout of the fused kernels: slot t holds codes (t, t+512); 4 dims x 2 codes per 32 bytes.
    // (Splitting the two 16-byte halves of a slot into separate contiguous arrays measured SLOWER: the
    // second load of a slot then no longer hits the lines the first one brought in.)
    const int SP = (ix->S + 3) & ~3, SPq = SP / 4;
    std::vector<float> cbP((size_t)ix->m * SPq * FUSED_T * 8, 0.0f);
    for (int p = 0; p < ix->m; ++p)
      for (int jb = 0; jb < SPq; ++jb)
        for (int tl = 0; tl < FUSED_T; ++tl)
          for (int u = 0; u < 4; ++u)
            for (int e = 0; e < 2; ++e) {
              const int j = jb * 4 + u, c = tl + e * FUSED_T;
              if (j < ix->S && c < ix->K)
                cbP[((((size_t)p * SPq + jb) * FUSED_T + tl) * 4 + u) * 2 + e] = codebook[((size_t)p * ix->K + c) * ix->S + j];
            }
    if (upload(&ix->cbP, cbP.data(), cbP.size(), &ix->bytes)) return fail(FREDDY_E_NOMEM, "device allocation failed");
  }
  // filter + refine tables (fused4.h)
  if (ix->cbP && ix->m == 12 && ix->S == 25) {
    std::vector<float> pmax((size_t)ix->m), cmaxp((size_t)ix->m);
    for (int p = 0; p < ix->m; ++p) {
      double comax = 0.0, cmax = 0.0;
      for (int c = 0; c < C; ++c) {
        double n2 = 0.0;
        for (int j = 0; j < ix->S; ++j) { const double v = ix->h_coarse[(size_t)c * d + p * ix->S + j]; n2 += v * v; }
        comax = std::max(comax, std::sqrt(n2));
      }
      for (int c = 0; c < ix->K; ++c) {
        double n2 = 0.0;
        for (int j = 0; j < ix->S; ++j) { const double v = codebook[((size_t)p * ix->K + c) * ix->S + j]; n2 += v * v; }
        cmax = std::max(cmax, std::sqrt(n2));
      }
      pmax[p] = (float)((comax + cmax) * (1.0 + 1e-6));
      cmaxp[p] = (float)(cmax * (1.0 + 1e-6));
    }
    if (upload(&ix->cbR, codebook, (size_t)ix->m * ix->K * ix->S, &ix->bytes) ||
        upload(&ix->pmax, pmax.data(), pmax.size(), &ix->bytes) ||
        upload(&ix->cmaxp, cmaxp.data(), cmaxp.size(), &ix->bytes))
      return fail(FREDDY_E_NOMEM, "device allocation failed");
    if (int rc = build_fragment_codebook(ix, codebook)) return rc;
  }
  return 0;
}

// rterm[slot] for every row slot of the pinned lists (the (cell, row) part of the filter's cheap distance)
static int refresh_row_terms(freddy_gpu_index* ix) {
  if (ix->rterm) { (void)hipFree(ix->rterm); ix->rterm = nullptr; }
  if (!ix->cbR) return 0;
  const int64_t n_slots = std::max<int64_t>(ix->n_blocks, 1) * 64;
  if (hipMalloc((void**)&ix->rterm, sizeof(float) * (size_t)n_slots) != hipSuccess) return fail(FREDDY_E_NOMEM, "device allocation failed");
  if (ix->n_blocks > 0) {
    hipLaunchKernelGGL(row_term_kernel, dim3((unsigned)((ix->n_blocks * 64 + 255) / 256)), dim3(256), 0, ix->stream, ix->packed,
                       ix->blk_cell, ix->coarse, ix->cbR, ix->rterm, ix->n_blocks * 64, ix->M2, ix->d, ix->m, ix->K, ix->S);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ix->stream) != hipSuccess)
      return fail(FREDDY_E_HIP, "building the row terms failed");
  }
  return 0;
}

static int open_device(freddy_gpu_index* ix, int device) {
  // The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the
  // runtime starts: the pipeline's four lanes want a queue each beside the library's own stream (DESIGN.md 5.2c:
  // 6 queues measured best).  Set here unless the host chose a value; without effect if the runtime is already up.
  setenv("GPU_MAX_HW_QUEUES", "6", 0);
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(FREDDY_E_ARG, "device %d out of range (%d visible)", device, n);
  HIP_TRY(hipSetDevice(device));
  ix->device = device;
  ix->tune = read_tuning();
  if (int rc = raise_lds_limits(device)) return rc;
  HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ix->n_cus = prop.multiProcessorCount;
  return 0;
}

static int check_pq_shape(int d, int m, int K, int64_t N) {
  if (d <= 0 || m <= 0 || K <= 0 || N < 0) return fail(FREDDY_E_ARG, "non-positive dimension");
  if (d % m) return fail(FREDDY_E_ARG, "d=%d is not a multiple of m=%d", d, m);
  if (K > 65536) return fail(FREDDY_E_LIMIT, "K=%d does not fit a 16-bit code", K);
  if ((size_t)m * K * 4 + 4096 > 160 * 1024)
    return fail(FREDDY_E_LIMIT, "LUT of m*K=%d floats does not fit the 160 KiB LDS", m * K);
  if (N > (int64_t)INT32_MAX - 64) return fail(FREDDY_E_LIMIT, "N too large for 32-bit row positions");
  return 0;
}

extern "C" int freddy_gpu_pin_pq(const freddy_pq_desc* t, int device, freddy_gpu_index_t** out) {
  if (!t || !out || !t->codebook || (t->N && (!t->ids || !t->codes))) return fail(FREDDY_E_ARG, "NULL argument");
  if (int rc = check_pq_shape(t->d, t->m, t->K, t->N)) return rc;
  for (int64_t r = 1; r < t->N; ++r)
    if (t->ids[r] <= t->ids[r - 1])
      return fail(FREDDY_E_ARG, "ids must be strictly ascending (canonical scan order); violated at row %lld", (long long)r);
  freddy_gpu_index* ix = new freddy_gpu_index();
  ix->kind = KIND_PQ;
  ix->d = t->d; ix->m = t->m; ix->K = t->K; ix->S = t->d / t->m; ix->M2 = (t->m + 1) / 2; ix->N = t->N;
  int rc = open_device(ix, device);
  if (!rc) rc = derive_codebook_tables(ix, t->codebook);
  if (!rc && upload(&ix->ids, t->ids, (size_t)t->N, &ix->bytes)) rc = fail(FREDDY_E_NOMEM, "device allocation failed");
  if (!rc) {
    const int32_t off[2] = {0, (int32_t)t->N};
    rc = pack_lists(ix, 1, off, t->codes, nullptr);
    if (!rc) rc = build_packed8(ix);
  }
  if (!rc) { ix->h_ids.assign(t->ids, t->ids + t->N); ix->max_id = t->N ? t->ids[t->N - 1] : -1; }
  if (rc) { free_index(ix); return rc; }
  *out = ix;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_pin_ivf(const freddy_ivf_desc* t, int device, freddy_gpu_index_t** out) {
  if (!t || !out || !t->codebook || !t->coarse || !t->list_off || (t->N && (!t->ids || !t->codes)))
    return fail(FREDDY_E_ARG, "NULL argument");
  if (int rc = check_pq_shape(t->d, t->m, t->K, t->N)) return rc;
  if (t->C <= 0) return fail(FREDDY_E_ARG, "C must be positive");
  if (t->list_off[0] != 0 || t->list_off[t->C] != t->N) return fail(FREDDY_E_ARG, "list_off must span [0, N]");
  for (int c = 0; c < t->C; ++c)   // every offset is checked BEFORE any row is touched through it
    if (t->list_off[c] < 0 || t->list_off[c] > t->list_off[c + 1] || (int64_t)t->list_off[c + 1] > t->N)
      return fail(FREDDY_E_ARG, "list_off is not non-decreasing inside [0, N] at list %d", c);
  for (int c = 0; c < t->C; ++c)
    for (int64_t r = t->list_off[c]; r < t->list_off[c + 1]; ++r) {
      if (t->ids[r] < 0) return fail(FREDDY_E_ARG, "negative id at row %lld", (long long)r);
      if (r > t->list_off[c] && t->ids[r] <= t->ids[r - 1])
        return fail(FREDDY_E_ARG, "ids must be strictly ascending inside list %d (row %lld)", c, (long long)r);
    }
  {   // "unique overall": a row id may sit in one list only (the merge orders a query's candidates by id)
    std::vector<int32_t> sorted_ids(t->ids, t->ids + t->N);
    std::sort(sorted_ids.begin(), sorted_ids.end());
    for (int64_t r = 1; r < t->N; ++r)
      if (sorted_ids[(size_t)r] == sorted_ids[(size_t)r - 1])
        return fail(FREDDY_E_ARG, "id %d occurs in more than one list", (int)sorted_ids[(size_t)r]);
  }
  freddy_gpu_index* ix = new freddy_gpu_index();
  ix->kind = KIND_IVF;
  ix->d = t->d; ix->m = t->m; ix->K = t->K; ix->S = t->d / t->m; ix->M2 = (t->m + 1) / 2; ix->N = t->N; ix->C = t->C;
  int rc = open_device(ix, device);
  if (!rc) {
    ix->Cpad = (t->C + WG - 1) / WG * WG;
    std::vector<float> cT((size_t)t->d * ix->Cpad, 0.0f);
    for (int c = 0; c < t->C; ++c)
      for (int i = 0; i < t->d; ++i) cT[(size_t)i * ix->Cpad + c] = t->coarse[(size_t)c * t->d + i];
    if (upload(&ix->coarse, t->coarse, (size_t)t->C * t->d, &ix->bytes) ||
        upload(&ix->coarseT, cT.data(), cT.size(), &ix->bytes))
      rc = fail(FREDDY_E_NOMEM, "device allocation failed");
    if (!rc) {   // MFMA coarse kernel (coarse.h): zero-padded rows, squared norms (fp64, rounded once), largest norm
      ix->dp = (t->d + COARSE_DP_ALIGN - 1) / COARSE_DP_ALIGN * COARSE_DP_ALIGN;
      // fragment order [Cpad / 32][dp / 8][lane = 32 h + r][4]: element t = c[32 g + r][8 i + 4 h + t]
      std::vector<float> cP((size_t)ix->Cpad * ix->dp, 0.0f), cn2((size_t)ix->Cpad, 0.0f);
      const int nit = ix->dp / 8;
      double cmax2 = 0.0;
      for (int c = 0; c < t->C; ++c) {
        double n2 = 0.0;
        for (int i = 0; i < t->d; ++i) {
          const float v = t->coarse[(size_t)c * t->d + i];
          const int it = i >> 3, hh = (i >> 2) & 1, tt = i & 3;
          cP[((((size_t)(c >> 5) * nit + it) * 64) + (size_t)hh * 32 + (c & 31)) * 4 + tt] = v;
          n2 += (double)v * (double)v;
        }
        cn2[(size_t)c] = (float)n2;
        cmax2 = std::max(cmax2, n2);
      }
      ix->cmax = (float)(std::sqrt(cmax2) * (1.0 + 1e-6));
      // the f16-split copy of the centroids for the matrix cores (coarse.h coarse_approx16_body; FREDDY_GPU_COARSE_H16=0: the fp32 tiles).
      // Many cells: the fp32 tiles are bound by the matrix pipe (134 -> 102 us at 13 000 cells); 1000 cells: 32 -> 29 us alone,
      // 54 -> 46 us with four batches in flight (fewer matrix-pipe cycles beside the other batches' kernels)
      if (env_int("FREDDY_GPU_COARSE_H16", 1) != 0 && t->d % 4 == 0) {
        float amax = 0.0f;
        for (size_t i = 0; i < (size_t)t->C * t->d; ++i) amax = std::max(amax, std::fabs(t->coarse[i]));
        int e = 0;
        if (amax > 0.0f && amax < 3e38f) { (void)frexpf(amax, &e); e = 14 - e; }
        ix->coarse_ec = e;
        const int T = (t->d + 15) / 16;
        std::vector<_Float16> cH((size_t)ix->Cpad * T * 2 * 8 * 2, (_Float16)0.0f);   // [Cpad/32][T][2][64][8]
        for (int c = 0; c < t->C; ++c)
          for (int i = 0; i < t->d; ++i) {
            const float v = ldexpf(t->coarse[(size_t)c * t->d + i], e);
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            const int tt = i >> 4, g = (i >> 3) & 1, u = i & 7;
            const size_t base = (((size_t)(c >> 5) * T + tt) * 2) * 64;
            cH[(base + (size_t)g * 32 + (c & 31)) * 8 + u] = hi;
            cH[(base + 64 + (size_t)g * 32 + (c & 31)) * 8 + u] = lo;
          }
        _Float16* dH = nullptr;
        if (upload(&dH, cH.data(), cH.size(), &ix->bytes)) rc = fail(FREDDY_E_NOMEM, "device allocation failed");
        ix->coarseH = dH;
      }
      if (rc) {} else
      if (upload(&ix->coarseP, cP.data(), cP.size(), &ix->bytes) || upload(&ix->cn2, cn2.data(), cn2.size(), &ix->bytes) ||
          hipMalloc((void**)&ix->viol, 4 * sizeof(int32_t)) != hipSuccess || hipMemset(ix->viol, 0, 4 * sizeof(int32_t)) != hipSuccess)
        rc = fail(FREDDY_E_NOMEM, "device allocation failed");
    }
    if (!rc) { ix->h_coarse.assign(t->coarse, t->coarse + (size_t)t->C * t->d); rc = derive_codebook_tables(ix, t->codebook); }
  }
  if (!rc) rc = pack_lists(ix, t->C, t->list_off, t->codes, t->ids);
  if (!rc) rc = build_packed8(ix);
  if (!rc) rc = refresh_row_terms(ix);   // one float per row slot: the (cell, row) part of the filter's cheap distance
  if (!rc) {
    if (ix->rterm) ix->bytes += (int64_t)sizeof(float) * std::max<int64_t>(ix->n_blocks, 1) * 64;
    for (int64_t r = 0; r < t->N; ++r) ix->max_id = std::max(ix->max_id, t->ids[r]);
  }
  if (rc) { free_index(ix); return rc; }
  *out = ix;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_pin_ivpq(const freddy_ivpq_desc* t, int device, freddy_gpu_index_t** out) {
  if (!t || !out || !t->codebook || !t->coarse || !t->stats || (t->N && (!t->ids || !t->codes || !t->coarse_id)))
    return fail(FREDDY_E_ARG, "NULL argument");
  if (t->d <= 0 || t->m <= 0 || t->K <= 0 || t->d % t->m) return fail(FREDDY_E_ARG, "bad d/m/K");
  if (t->coarse_positions != 2) return fail(FREDDY_E_LIMIT, "only 2 coarse positions are supported (as in the reference, index_utils.c:322)");
  if (t->coarse_codes <= 0 || t->d % 2) return fail(FREDDY_E_ARG, "bad coarse multi-index shape");
  freddy_gpu_index* ix = new freddy_gpu_index();
  ix->kind = KIND_IVPQ;
  ix->d = t->d; ix->m = t->m; ix->K = t->K; ix->S = t->d / t->m; ix->N = t->N;
  int rc = open_device(ix, device);
  if (!rc) {
    rc = join_pin(&ix->join, t, &ix->bytes);
    ix->join.host_traversal = env_int("FREDDY_GPU_JOIN_HOST_TRAVERSAL", 0) != 0;
    if (rc) rc = fail(rc, "%s", join_error());
  }
  if (rc) { free_index(ix); return rc; }
  *out = ix;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_unpin(freddy_gpu_index_t* ix) {
  free_index(ix);
  return FREDDY_OK;
}

extern "C" int64_t freddy_gpu_index_bytes(const freddy_gpu_index_t* ix) { return ix ? ix->bytes : 0; }
extern "C" int64_t freddy_gpu_last_scanned_rows(const freddy_gpu_index_t* ix) {
  // rows retrieved in the most recent probing round (last query chunk): read back on demand
  const Workspace* ws = ix ? (ix->last_ws ? ix->last_ws : &ix->ws[0]) : nullptr;
  if (!ix || ix->kind != KIND_IVF || ix->last_Q <= 0 || !ws->w_rows.p) return 0;
  if (hipSetDevice(ix->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
  std::vector<int32_t> rows((size_t)ix->last_Q);
  if (hipMemcpy(rows.data(), ws->w_rows.p, sizeof(int32_t) * rows.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  int64_t sum = 0;
  for (int32_t r : rows) if (r > 0) sum += r;
  return sum;
}

extern "C" int freddy_gpu_last_probed_cells(const freddy_gpu_index_t* ix, int64_t* n_cells, int64_t* rows) {
  // distinct cells the most recent probing round (cell-grouped scans only) touched, and the rows of their
  // lists: what a scan that reads every probed list ONCE per batch has to move (freddy.c:939-974)
  if (!ix || !n_cells || !rows) return fail(FREDDY_E_ARG, "NULL argument");
  *n_cells = 0; *rows = 0;
  if (ix->kind != KIND_IVF) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  const Workspace* ws = ix->last_ws ? ix->last_ws : &ix->ws[0];
  if (ix->last_Q <= 0 || !ws->w_cellcnt.p) return FREDDY_OK;
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<int32_t> cnt((size_t)ix->C);
  HIP_TRY(hipMemcpy(cnt.data(), ws->w_cellcnt.p, sizeof(int32_t) * cnt.size(), hipMemcpyDeviceToHost));
  for (int c = 0; c < ix->C; ++c)
    if (cnt[(size_t)c] > 0) { *n_cells += 1; *rows += ix->h_list_off[(size_t)c + 1] - ix->h_list_off[(size_t)c]; }
  return FREDDY_OK;
}

// one device's own counters (a PQ handle: those of its two views)
static int64_t read_viol_one(const freddy_gpu_index* ix, int which) {
  if (ix && (ix->pq_shadow || ix->pq_sub_view)) {
    const int64_t a = ix->pq_shadow ? read_viol_one(ix->pq_shadow, which) : 0, b = ix->pq_sub_view ? read_viol_one(ix->pq_sub_view, which) : 0;
    return (a < 0 || b < 0) ? -1 : a + b;
  }
  if (!ix || !ix->viol) return 0;
  int32_t h[4] = {0, 0, 0, 0};
  if (hipSetDevice(ix->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(h, ix->viol, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return which == 0 ? (int64_t)h[0] + h[2] : h[which];
}
static int64_t read_viol(const freddy_gpu_index_t* ix, int which) {
  if (!ix) return 0;
  int64_t sum = 0;
  for (const freddy_gpu_index* r : ix->replicas) { const int64_t v = read_viol_one(r, which); if (v < 0) return -1; sum += v; }
  const int64_t v = read_viol_one(ix, which);   // (last: the calling thread is left on the primary's device)
  return v < 0 ? -1 : sum + v;
}
extern "C" int64_t freddy_gpu_filter_bound_violations(const freddy_gpu_index_t* ix) { return read_viol(ix, 0); }
extern "C" int64_t freddy_gpu_filter_bound_checked(const freddy_gpu_index_t* ix) { return read_viol(ix, 1); }
extern "C" int64_t freddy_gpu_coarse_bound_checked(const freddy_gpu_index_t* ix) { return read_viol(ix, 3); }

extern "C" int freddy_gpu_profile_enable(freddy_gpu_index_t* ix, int32_t enable) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  for (auto& kv : ix->prof)
    for (auto& ev : kv.second.open) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  ix->prof.clear();
  ix->profiling = enable != 0;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_profile_read(freddy_gpu_index_t* ix, int32_t cap, char (*names)[64],
                                       int64_t* launches, double* total_ms) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipDeviceSynchronize());
  int n = 0;
  for (auto& kv : ix->prof) {
    ProfRec& r = kv.second;
    for (auto& ev : r.open) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) r.ms += ms;
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    r.open.clear();
    if (n < cap) {
      if (names) { strncpy(names[n], kv.first.c_str(), 63); names[n][63] = 0; }
      if (launches) launches[n] = r.launches;
      if (total_ms) total_ms[n] = r.ms;
    }
    ++n;
  }
  return n;
}

extern "C" int freddy_gpu_set_option(freddy_gpu_index_t* ix, const char* name, int64_t value) {
  if (!ix || !name) return fail(FREDDY_E_ARG, "NULL argument");
  if (!ix->replicas.empty()) {   // the primary first: an unknown name is rejected before any device has changed
    std::vector<freddy_gpu_index*> reps;
    reps.swap(ix->replicas);
    int rc = freddy_gpu_set_option(ix, name, value);
    reps.swap(ix->replicas);
    if (rc) return rc;
    for (freddy_gpu_index* r : ix->replicas)
      if ((rc = freddy_gpu_set_option(r, name, value))) { ix->poisoned = true; return rc; }
    return FREDDY_OK;
  }
  Tuning& t = ix->tune;
  const std::string n(name);
  if (n == "fused") t.fused = (int)value;
  else if (n == "fused_kernel") t.scan_kernel = (int)value;
  else if (n == "reserve_cus") t.reserve_cus = (int)value;
  else if (n == "scan_share") t.scan_share = (int)std::max<int64_t>(1, value);
  else if (n == "join_host_traversal") ix->join.host_traversal = value != 0;
  else if (n == "join_libm_margin_ppm") ix->join.libm_margin = (float)value * 1e-6f;
  else if (n == "sparse_items") t.sparse_items = std::max(-16, std::min(16, (int)value));
  else if (n == "pipeline_batch") t.pipeline_batch = (int)std::max<int64_t>(16, value);
  else if (n == "pipeline_lanes") t.pipeline_lanes = (int)std::min<int64_t>(FREDDY_LANES, std::max<int64_t>(1, value));
  else if (n == "pq_fused") t.pq_fused = (int)value;
  else if (n == "one_launch") t.one_launch = (int)value;
  else if (n == "coarse_approx") t.coarse_approx = (int)value;
  else if (n == "check_brackets") t.check_brackets = (int)value;
  else if (n == "lut_budget_mb") t.lut_budget_mb = std::max<int64_t>(1, value);
  else if (n == "exact_filter") t.exact_filter = (int)value;
  else if (n == "codes_u8") t.codes_u8 = (int)value;
#ifdef FREDDY_LAB
  else if (n == "fused_prof") t.scan_prof = (int)value;
#endif
  else return fail(FREDDY_E_ARG, "unknown option '%s'", name);
  return FREDDY_OK;
}

// ---------------------------------------------------------------------------------------
// kernel dispatch helpers
// ---------------------------------------------------------------------------------------
static int pick_V(int L) {
  if (L <= 64) return 1;
  if (L <= 128) return 2;
  if (L <= 256) return 4;
  if (L <= 512) return 8;
  if (L <= 1024) return 16;
  return 0;
}

template <int M, int V>
static int launch_scan_mv(freddy_gpu_index* ix, hipStream_t s, const ScanArgs& a, dim3 grid, size_t lds) {
  timed_launch(ix, s, "adc_scan", [&] { hipLaunchKernelGGL((adc_scan_kernel<M, V>), grid, dim3(SCAN_WG), lds, s, a); });
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int M>
static int launch_scan_m(freddy_gpu_index* ix, hipStream_t s, const ScanArgs& a, dim3 grid, size_t lds, int V) {
  switch (V) {
    case 1: return launch_scan_mv<M, 1>(ix, s, a, grid, lds);
    case 2: return launch_scan_mv<M, 2>(ix, s, a, grid, lds);
    case 4: return launch_scan_mv<M, 4>(ix, s, a, grid, lds);
    case 8: return launch_scan_mv<M, 8>(ix, s, a, grid, lds);
    case 16: return launch_scan_mv<M, 16>(ix, s, a, grid, lds);
  }
  return fail(FREDDY_E_LIMIT, "unsupported selection width");
}

static int launch_scan(freddy_gpu_index* ix, hipStream_t s, const ScanArgs& a, int n_items) {
  if (n_items <= 0 || a.nchunk <= 0) return 0;
  const int Vl = pick_V(a.L);
  const size_t lds = std::max((((size_t)a.m * a.K * 4 + 15) & ~(size_t)15) + (size_t)SCAN_WAVES * 64 * sizeof(u64),
                              (size_t)SCAN_WAVES * 64 * Vl * sizeof(u64));
  dim3 grid((unsigned)a.nchunk, (unsigned)n_items);
  const int V = pick_V(a.L);
  if (a.m == 12) return launch_scan_m<12>(ix, s, a, grid, lds, V);
  return launch_scan_m<0>(ix, s, a, grid, lds, V);
}

static int launch_merge(freddy_gpu_index* ix, hipStream_t s, const MergeArgs& a) {
  if (a.n_active <= 0) return 0;
  const int V = pick_V(a.L);
  dim3 grid((unsigned)a.n_active), block(64);
  timed_launch(ix, s, "merge_replay", [&] {
    switch (V) {
      case 1: hipLaunchKernelGGL((merge_replay_kernel<1>), grid, block, 0, s, a); break;
      case 2: hipLaunchKernelGGL((merge_replay_kernel<2>), grid, block, 0, s, a); break;
      case 4: hipLaunchKernelGGL((merge_replay_kernel<4>), grid, block, 0, s, a); break;
      case 8: hipLaunchKernelGGL((merge_replay_kernel<8>), grid, block, 0, s, a); break;
      case 16: hipLaunchKernelGGL((merge_replay_kernel<16>), grid, block, 0, s, a); break;
    }
  });
  HIP_TRY(hipGetLastError());
  return 0;
}

// coarse != NULL: `vecs` are the queries and the kernel forms the residual q - coarse[cell] itself (no residual_kernel launch)
static int launch_lut(freddy_gpu_index* ix, hipStream_t s, const float* vecs, const int32_t* item_cell,
                      float* lut, int n_items, const float* coarse = nullptr, const int32_t* item_query = nullptr) {
  if (n_items <= 0) return 0;
  // enough workgroups to fill 256 CUs several times over, while amortising the register
  // fill of the codebook slice over as many items as possible
  int ipw = (int)std::max<int64_t>(1, ((int64_t)ix->m * n_items + 4095) / 4096);
  ipw = std::min(ipw, 64);
  dim3 grid((unsigned)ix->m, (unsigned)((n_items + ipw - 1) / ipw));
  const int m = ix->m, K = ix->K, d = ix->d, S = ix->S;
  timed_launch(ix, s, "lut_build", [&] {
    if (S == 25) hipLaunchKernelGGL((lut_build_kernel<25, 4>), grid, dim3(WG), 0, s, vecs, item_cell, ix->cbT, lut, n_items, ipw, m, K, d, coarse, item_query);
    else if (S == 10) hipLaunchKernelGGL((lut_build_kernel<10, 4>), grid, dim3(WG), 0, s, vecs, item_cell, ix->cbT, lut, n_items, ipw, m, K, d, coarse, item_query);
    else if (S == 20) hipLaunchKernelGGL((lut_build_kernel<20, 4>), grid, dim3(WG), 0, s, vecs, item_cell, ix->cbT, lut, n_items, ipw, m, K, d, coarse, item_query);
    else hipLaunchKernelGGL(lut_build_generic_kernel, grid, dim3(WG), 0, s, vecs, item_cell, ix->cbT, lut, n_items, ipw, m, K, d, S, coarse, item_query);
  });
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// IVFADC
// ---------------------------------------------------------------------------------------

// coarse distances (a6/a7) of every query of the chunk, and -- for the filter + refine scan -- the
// per-batch query x codebook table beside them on the side stream
static int ivf_coarse(IvfRun& r) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int Q = r.Q, d = ix->d, C = ix->C, m = ix->m, K = ix->K, Cpad = ix->Cpad;
  const int used_words = (C + 31) / 32;
  const size_t items = (size_t)Q * r.W;
  // round-one scratch that must start at zero: the probe bitmaps, the counters (n_next, n_groups, work
  // counter), the per-cell item counts and the accepted-candidate counts: every coarse kernel clears them itself
  // (ZeroArgs) -- except the small-batch kernel for vectors of more than 1024 dimensions, which gets memsets.
  const bool small_zero = !r.tiled && d <= 1024;
  if (!r.tiled && !small_zero) {
    HIP_TRY(hipMemsetAsync(ws->w_used.p, 0, sizeof(uint32_t) * (size_t)Q * used_words, s));
    HIP_TRY(hipMemsetAsync(ws->w_cnt.p, 0, sizeof(int32_t) * 8, s));
  }
  ZeroArgs za;
  za.p[0] = ws->w_used.as<uint32_t>(); za.n[0] = Q * used_words;
  za.p[1] = ws->w_cnt.as<uint32_t>(); za.n[1] = 8;
  za.p[2] = r.fused ? ws->w_cellcnt.as<uint32_t>() : nullptr; za.n[2] = r.fused ? C * 2 : 0;
  za.p[3] = ws->w_cand.as<uint32_t>(); za.n[3] = Q;
  // survivor counts: regions of chunks a list does not have, or of items without a cell, stay at zero
  za.p[4] = r.fused ? ws->w_surv_cnt.as<uint32_t>() : nullptr; za.n[4] = r.fused ? (int)(items * r.upi * FUSED_NW) : 0;

  // more than 1024 cells: the (query, 128-cell tile) minima for the plan's two-level selection (round one: no cell is used yet)
  float* tile_min = nullptr;
  if (r.approx && Cpad > COARSE_MAX_CPAD) {
    if (ws->w_tmin.ensure(sizeof(float) * (size_t)Q * (Cpad / 128))) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
    tile_min = ws->w_tmin.as<float>();
  }
  auto launch_coarse = [&]() -> int {
    timed_launch(ix, s, "coarse_dist", [&] {
      if (r.approx && ix->coarseH)
        hipLaunchKernelGGL(coarse_approx16_kernel, dim3(Cpad / 128, (Q + COARSE_TQ - 1) / COARSE_TQ), dim3(256), coarse_approx16_lds(d), s, r.d_q,
                           (const ch8v*)ix->coarseH, ix->coarse_ec, ix->cn2, ws->w_distT.as<float>(), ws->w_qn2.as<float>(), Q, Cpad, d, za, tile_min, C);
      else if (r.approx)
        hipLaunchKernelGGL(coarse_approx_kernel, dim3(Cpad / 128, (Q + COARSE_TQ - 1) / COARSE_TQ), dim3(256),
                           (size_t)(COARSE_TQ * (ix->dp + 4) + 128) * sizeof(float), s, r.d_q, ix->coarseP, ix->cn2,
                           ws->w_distT.as<float>(), ws->w_qn2.as<float>(), Q, Cpad, d, ix->dp, za, tile_min, C);
      else if (r.tiled)
        hipLaunchKernelGGL((coarse_tile_kernel<2, 16>), dim3(Cpad / 32, (Q + 63) / 64), dim3(256), 0, s, r.d_q, ix->coarseT,
                           ws->w_distT.as<float>(), Q, Cpad, d, za);
      else if (small_zero)
        hipLaunchKernelGGL((coarse_small_kernel<50>), dim3(Cpad / 64, Q), dim3(64), 0, s, r.d_q, ix->coarseT, ws->w_distT.as<float>(), Q, Cpad, d, za);
      else
        hipLaunchKernelGGL((coarse_dist_kernel<16>), dim3(Cpad / WG, (Q + 15) / 16), dim3(WG), (size_t)d * 16 * sizeof(float), s, r.d_q,
                           ix->coarseT, ws->w_distT.as<float>(), Q, Cpad, d);
    });
    HIP_TRY(hipGetLastError());
    return 0;
  };
  // The query x codebook table is independent of the coarse distances: with the MFMA cell selection the coarse tiles and
  // the table units are the workgroups of ONE launch (fused5.h coarse_table5_kernel); otherwise the table kernel runs in
  // line before the coarse kernel.
  if (r.approx && r.fused && r.scan_kernel == 5) {
    CoarseTableArgs ct;
    ct.queries = r.d_q; ct.coarseF = ix->coarseP; ct.cn2 = ix->cn2; ct.dist = ws->w_distT.as<float>(); ct.qn2 = ws->w_qn2.as<float>();
    ct.Q = Q; ct.Cpad = Cpad; ct.d = d; ct.dp = ix->dp; ct.z = za; ct.coarse_gx = Cpad / 128; ct.coarse_gy = (Q + COARSE_TQ - 1) / COARSE_TQ;
    ct.cbT = ix->cbF; ct.cmax = ix->cmaxp; ct.qn = ws->w_qn.as<float>(); ct.qscale = ws->w_qn.as<float>() + (size_t)Q * m;
    ct.qc = ws->w_qc.as<uint32_t>(); ct.m = m; ct.K = K; ct.tmin = tile_min; ct.C = C;
    ct.coarseH = (const ch8v*)ix->coarseH; ct.ec = ix->coarse_ec;
    const size_t lds = std::max<size_t>(ix->coarseH ? coarse_approx16_lds(d) : (size_t)(COARSE_TQ * (ix->dp + 4) + 128) * sizeof(float), (size_t)query_codebook5_lds<25, 16>());
    const unsigned grid = (unsigned)(ct.coarse_gx * ct.coarse_gy + m * ((Q + 15) / 16));
    timed_launch(ix, s, "coarse_table", [&] {
      if (ix->coarseH) hipLaunchKernelGGL((coarse_table5_kernel<25, 16, true>), dim3(grid), dim3(256), lds, s, ct);
      else hipLaunchKernelGGL((coarse_table5_kernel<25, 16>), dim3(grid), dim3(256), lds, s, ct);
    });
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (r.fused && r.scan_kernel == 5) {
    timed_launch(ix, s, "query_codebook", [&] {
      hipLaunchKernelGGL((query_codebook5_kernel<25, 16>), dim3(m, (Q + 15) / 16), dim3(256), 0, s, r.d_q, ix->cbF, ix->cmaxp,
                         ws->w_qn.as<float>(), ws->w_qn.as<float>() + (size_t)Q * m, ws->w_qc.as<uint32_t>(), Q, d, m, K);
    });
    HIP_TRY(hipGetLastError());
  }
  return launch_coarse();
}

// a7: the W nearest not-yet-used cells of every active query (+ their items appended to the cells' buckets)
static int ivf_plan(IvfRun& r, PlanArgs& pa) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int C = ix->C, W = r.W;
  pa.dist = ws->w_distT.as<float>(); pa.active = r.active; pa.list_off = ix->list_off;
  pa.used = ws->w_used.as<uint32_t>();
  pa.item_cell = ws->w_item_cell.as<int32_t>(); pa.item_query = ws->w_item_query.as<int32_t>();
  pa.item_dist = ws->w_item_dist.as<float>();
  pa.round_rows = ws->w_rows.as<int32_t>();
  pa.n_active = r.n_active; pa.Cpad = ix->Cpad; pa.C = C; pa.W = W; pa.used_words = (C + 31) / 32;
  pa.cell_count = r.fused ? ws->w_cellcnt.as<int32_t>() : nullptr;
  pa.cell_items = r.fused ? ws->w_sorted.as<int32_t>() : nullptr; pa.cell_cap = r.n_active;
  pa.cell_limit = r.cell_limit;
  const int n_items = r.n_active * W;
  if (r.fused && !(r.zeroed && r.first())) {
    HIP_TRY(hipMemsetAsync(ws->w_cellcnt.p, 0, sizeof(int32_t) * (size_t)C * 2, s));   // counts + fill cursors
    HIP_TRY(hipMemsetAsync(ws->w_surv_cnt.p, 0, sizeof(int32_t) * (size_t)n_items * r.upi * FUSED_NW, s));
  }
  const int PV = pick_V(2 * W);
  const size_t plan_lds = (size_t)(64 + 64 * PV) * sizeof(u64) + (size_t)W * 8;
  if (r.approx) {
    Plan2Args g;
    g.p = pa; g.queries = r.d_q; g.coarse = ix->coarse; g.qn2 = ws->w_qn2.as<float>(); g.item_dist = pa.item_dist;
    g.violations = ix->viol; g.cmax = ix->cmax; g.d = ix->d; g.refine_all = (ix->tune.check_brackets & 2) ? 1 : 0; g.prof = nullptr;
    g.tmin = (r.first() && ix->Cpad > COARSE_MAX_CPAD && ws->w_tmin.p) ? ws->w_tmin.as<float>() : nullptr;
    timed_launch(ix, s, "probe_plan", [&] {
      // (one batch at a time: four waves per query, the shortest latency; batches in flight: one wave per query, the smallest footprint)
      if (ix->Cpad <= COARSE_MAX_CPAD && r.share > 1) hipLaunchKernelGGL((probe_plan2_kernel<0, false, 1>), dim3(r.n_active), dim3(64), 0, s, g);
      else if (ix->Cpad <= COARSE_MAX_CPAD) hipLaunchKernelGGL((probe_plan2_kernel<0, false>), dim3(r.n_active), dim3(64 * PLAN2_NW), 0, s, g);
      else hipLaunchKernelGGL((probe_plan2_kernel<0, true>), dim3(r.n_active), dim3(64 * PLAN2_NW), 0, s, g);   // (more than 1024 cells: streamed)
    });
  } else
  timed_launch(ix, s, "probe_plan", [&] {
    switch (PV) {
      case 1: hipLaunchKernelGGL((probe_plan_kernel<1>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
      case 2: hipLaunchKernelGGL((probe_plan_kernel<2>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
      case 4: hipLaunchKernelGGL((probe_plan_kernel<4>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
      case 8: hipLaunchKernelGGL((probe_plan_kernel<8>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
      default: hipLaunchKernelGGL((probe_plan_kernel<16>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
    }
  });
  HIP_TRY(hipGetLastError());
  if (!(r.zeroed && r.first())) HIP_TRY(hipMemsetAsync(ws->w_cand.p, 0, sizeof(int32_t) * r.Q, s));
  return 0;
}

// The work table shared by both cell-grouped scans: per-cell item counts -> (<= 12 items of a cell, 4096-row
// chunk) entries, largest first.
struct WorkTable {
  size_t max_groups;
  int32_t *group_cell, *group_first, *group_cnt, *n_groups, *work_counter;
  // (item, chunk) units of the cells that few queries probe (sparse5.h); sp_cap = 0: none
  size_t sp_cap;
  int32_t *sp_cell, *sp_first, *sp_chunk, *n_sparse, *sp_counter;
  bool sp_pairs = false;   // units of up to two items (sparse5.h NI = 2)
};
static int ivf_work_table(IvfRun& r, WorkTable& wt) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int n_items = r.n_active * r.W;
  wt.max_groups = ((size_t)n_items / SPEC2_G + (size_t)ix->C + 1) * r.upi;   // (group, chunk) work entries
  int32_t* base = ws->w_groups.as<int32_t>();
  wt.group_cell = base; wt.group_first = base + wt.max_groups; wt.group_cnt = base + 2 * wt.max_groups;
  wt.n_groups = ws->w_cnt.as<int32_t>() + 1;
  wt.work_counter = ws->w_cnt.as<int32_t>() + 2;
  wt.sp_counter = ws->w_cnt.as<int32_t>() + 3;
  wt.n_sparse = ws->w_cnt.as<int32_t>() + 4;
  // cells that one or two queries probe are scanned item by item -- where such cells are the rule (fewer than four items per
  // cell on average: a corpus with more cells than the batch has probes) and there are enough of them to fill the chip's
  // workgroup slots several times (the item-wise scan is built for throughput: a 256-query batch on 1000 cells took 0.187
  // instead of 0.155 ms with it); a dense batch does not pay the extra launch for its handful of thin cells
  // (a negative option value forces the item-wise scan for cells of up to that many items whatever the batch: tests)
  const int sparse_max = r.scan_kernel != 5 ? 0
                         : ix->tune.sparse_items < 0 ? -ix->tune.sparse_items
                         : ((size_t)n_items < 4 * (size_t)ix->C && n_items >= 16 * ix->n_cus) ? ix->tune.sparse_items : 0;
  wt.sp_cap = sparse_max > 0 ? (size_t)n_items * r.upi : 0;
  wt.sp_cell = base + 3 * wt.max_groups; wt.sp_first = wt.sp_cell + wt.sp_cap; wt.sp_chunk = wt.sp_first + wt.sp_cap;
  timed_launch(ix, s, "work_table", [&] {
    hipLaunchKernelGGL(work_table_kernel, dim3(1), dim3(1024), 0, s, ws->w_cellcnt.as<int32_t>(), ix->C, r.n_active, r.scan_kernel == 5 ? SCAN5_G : SPEC2_G, ix->blk_off,
                       wt.group_cell, wt.group_first, wt.group_cnt, wt.n_groups, r.scan_kernel == 5 ? 2 : 0,
                       sparse_max, wt.sp_cell, wt.sp_first, wt.sp_chunk, wt.n_sparse, sparse_max >= 2 ? 1 : 0);
  });
  wt.sp_pairs = sparse_max >= 2;
  HIP_TRY(hipGetLastError());
  if (!(r.zeroed && r.first())) HIP_TRY(hipMemsetAsync(wt.work_counter, 0, 2 * sizeof(int32_t), s));   // (both work counters)
  return 0;
}

static int scan_prof_buffer(freddy_gpu_index* ix, Workspace* ws, long long** prof) {
  *prof = nullptr;
#ifndef FREDDY_LAB
  (void)ix; (void)ws;
  return 0;
#else
  if (!ix->tune.scan_prof) return 0;
  if (ws->w_prof.ensure(sizeof(long long) * 8 * 1024)) return fail(FREDDY_E_NOMEM, "profile buffer");
  *prof = ws->w_prof.as<long long>();
  return 0;
#endif
}

#ifdef FREDDY_LAB
// debugging aid (option fused_prof): per-phase shader-clock sums of every persistent workgroup's builder wave 0
static int scan_prof_print(freddy_gpu_index* ix, hipStream_t s, const long long* prof, unsigned n_persist) {
  std::vector<long long> h(8 * (size_t)n_persist);
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMemcpy(h.data(), prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
  double sum[8] = {0}; long long mx_end = 0, mn_end = -1; double ent = 0;
  for (unsigned b = 0; b < n_persist; ++b) {
    for (int i = 0; i < 6; ++i) sum[i] += (double)h[b * 8 + i];
    ent += (double)h[b * 8 + 7];
    mx_end = std::max(mx_end, h[b * 8 + 6]);
    mn_end = mn_end < 0 ? h[b * 8 + 6] : std::min(mn_end, h[b * 8 + 6]);
  }
  fprintf(stderr, "[scan prof] wgs=%u entries=%.0f  builder cycles/entry: builds=%.0f barrier-wait=%.0f tail=%.0f | gatherer wave 0 (fused5.h): main=%.0f colmin=%.0f S2=%.0f | workgroups ran dry over %.1f us\n",
          n_persist, ent, sum[0] / ent, sum[1] / ent, sum[3] / ent, sum[2] / ent, sum[4] / ent, sum[5] / ent, (mx_end - mn_end) / 100.0);
  (void)ix;
  return 0;
}
#endif

// Default scan: filter + refine.  entry records -> ivf_filter5_kernel (+ the item-wise scan of thin cells) -> merge_refine_kernel.
static int ivf_scan_filter(IvfRun& r, const PlanArgs& pa, const WorkTable& wt) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int Q = r.Q, m = ix->m, K = ix->K;
  if (!r.records_ready && ws->w_records.ensure(sizeof(int32_t) * REC_DW * wt.max_groups)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  RecordArgs ra;
  ra.group_cell = wt.group_cell; ra.group_first = wt.group_first; ra.group_cnt = wt.group_cnt; ra.n_groups = wt.n_groups;
  ra.sorted_item = ws->w_sorted.as<int32_t>(); ra.item_query = pa.item_query; ra.blk_off = ix->blk_off; ra.list_off = ix->list_off;
  ra.item_dist = pa.item_dist; ra.qn = ws->w_qn.as<float>(); ra.qscale = ws->w_qn.as<float>() + (size_t)Q * m; ra.pmax = ix->pmax;
  ra.records = ws->w_records.as<int32_t>(); ra.sentinel = r.sentinel;
  if (!r.records_ready) {
    timed_launch(ix, s, "entry_records", [&] {
      hipLaunchKernelGGL((entry_record5_kernel<12>), dim3((unsigned)((wt.max_groups + 3) / 4)), dim3(256), 0, s, ra);
    });
    HIP_TRY(hipGetLastError());
  }
  FilterArgs fl;
  fl.qc = ws->w_qc.as<uint32_t>(); fl.rterm = ix->rterm; fl.records = ws->w_records.as<int32_t>(); fl.n_groups = wt.n_groups;
  fl.work_counter = wt.work_counter; fl.packed = ix->packed; fl.surv = ws->w_surv.as<u64>(); fl.surv_count = ws->w_surv_cnt.as<int32_t>();
  fl.cand_count = (r.found_rule == 1) ? ws->w_cand.as<int32_t>() : nullptr;
  fl.K = K; fl.L = r.L; fl.upi = r.upi; fl.sentinel = r.sentinel; fl.keep_all = (ix->tune.check_brackets & 1) ? 1 : 0;
  if (int rc = scan_prof_buffer(ix, ws, &fl.prof)) return rc;
  // LDS: slabs [2 buffers][2 positions][K][16 items] int16, then column minima / thresholds, two entry records, row terms
  const size_t desc_off = (size_t)4 * SCAN5_G * 2 * K;
  const size_t flds = desc_off + 4096 + 64 + (2 * REC_DW + 4) * sizeof(int32_t) + 4096 * sizeof(float);
  fl.desc_offset = (uint32_t)desc_off;
  // One persistent workgroup per CU (LDS admits exactly one), never more than there is work.  Batches in flight share the
  // chip: a persistent scan that took every CU would hold up the small kernels of the other batches until it drains, and
  // their scans behind them; with n_cus / share workgroups each, the scans of `share` batches run side by side, the small
  // kernels fit in between, and a scan's workgroups pull more entries each (a shorter tail).
  const int scan_cus = std::max(ix->n_cus / std::max(1, r.share), std::min(ix->n_cus, 32)) - ix->tune.reserve_cus;
  const unsigned n_persist = (unsigned)std::min<size_t>(wt.max_groups, (size_t)std::max(1, scan_cus));
  // K <= 256: one byte per code (packed8); the profiling instantiation stays with the int16 layout
  const bool u8 = ix->packed8 && ix->tune.codes_u8 != 0 && K <= 256 && !fl.prof;
  fl.packed8 = u8 ? ix->packed8 : nullptr;
  timed_launch(ix, s, "ivf_filter", [&] {
    // (instantiations: the rule that counts accepted rows doubles the selection code, and the kernel is larger than the
    // instruction cache as it is)
    if (u8) {
      if (fl.cand_count) hipLaunchKernelGGL((ivf_filter5_kernel<12, false, true, false, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
      else hipLaunchKernelGGL((ivf_filter5_kernel<12, false, false, false, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
    } else if (fl.cand_count) {
      if (K == 1024) hipLaunchKernelGGL((ivf_filter5_kernel<12, true, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
      else hipLaunchKernelGGL((ivf_filter5_kernel<12, false, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
    }
#ifdef FREDDY_LAB
    else if (K == 1024 && fl.prof) hipLaunchKernelGGL((ivf_filter5_kernel<12, true, false, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
#endif
    else if (K == 1024) hipLaunchKernelGGL((ivf_filter5_kernel<12, true, false>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
    else hipLaunchKernelGGL((ivf_filter5_kernel<12, false, false>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
  });
  HIP_TRY(hipGetLastError());
  if (wt.sp_cap > 0) {
    // cells that one or two queries of the batch probe: item by item (sparse5.h), six workgroups of four waves per CU
    SparseArgs sp;
    sp.qc = fl.qc; sp.qscale = ra.qscale; sp.qn = ra.qn; sp.pmax = ix->pmax; sp.rterm = ix->rterm; sp.packed = ix->packed;
    sp.blk_off = ix->blk_off; sp.list_off = ix->list_off; sp.sorted_item = ra.sorted_item; sp.item_query = pa.item_query;
    sp.item_dist = pa.item_dist; sp.sp_cell = wt.sp_cell; sp.sp_first = wt.sp_first; sp.sp_chunk = wt.sp_chunk;
    sp.n_units = wt.n_sparse; sp.work_counter = wt.sp_counter; sp.surv = fl.surv; sp.surv_count = fl.surv_count; sp.packed8 = fl.packed8;
    sp.cand_count = fl.cand_count; sp.K = K; sp.L = r.L; sp.upi = r.upi; sp.sentinel = r.sentinel; sp.keep_all = fl.keep_all;
    const unsigned sp_grid = (unsigned)std::min<size_t>(wt.sp_cap, (size_t)std::max(1, scan_cus) * (wt.sp_pairs ? 3 : 6));
    timed_launch(ix, s, "sparse_items", [&] {
      if (wt.sp_pairs) {   // (cell, chunk) units of one or two items: the rows of a two-item cell are read once
        if (u8) {
          if (fl.cand_count) hipLaunchKernelGGL((sparse_pair5_kernel<12, true, true>), dim3(sp_grid), dim3(256), 0, s, sp);
          else hipLaunchKernelGGL((sparse_pair5_kernel<12, false, true>), dim3(sp_grid), dim3(256), 0, s, sp);
        } else if (fl.cand_count) hipLaunchKernelGGL((sparse_pair5_kernel<12, true, false>), dim3(sp_grid), dim3(256), 0, s, sp);
        else hipLaunchKernelGGL((sparse_pair5_kernel<12, false, false>), dim3(sp_grid), dim3(256), 0, s, sp);
      } else if (u8) {
        if (fl.cand_count) hipLaunchKernelGGL((sparse_item5_kernel<12, true, true>), dim3(sp_grid), dim3(256), 0, s, sp);
        else hipLaunchKernelGGL((sparse_item5_kernel<12, false, true>), dim3(sp_grid), dim3(256), 0, s, sp);
      } else if (fl.cand_count) hipLaunchKernelGGL((sparse_item5_kernel<12, true>), dim3(sp_grid), dim3(256), 0, s, sp);
      else hipLaunchKernelGGL((sparse_item5_kernel<12, false>), dim3(sp_grid), dim3(256), 0, s, sp);
    });
    HIP_TRY(hipGetLastError());
  }
#ifdef FREDDY_LAB
  if (fl.prof && K == 1024 && !fl.cand_count) if (int rc = scan_prof_print(ix, s, fl.prof, n_persist)) return rc;   // (the counters live in one instantiation)
#endif

  MergeRefineArgs mr;
  mr.surv = fl.surv; mr.surv_count = fl.surv_count; mr.active = r.active; mr.round_rows = pa.round_rows;
  mr.item_cell = pa.item_cell; mr.queries = r.d_q; mr.coarse = ix->coarse; mr.cbR = ix->cbR;
  mr.qn = ws->w_qn.as<float>(); mr.pmax = ix->pmax; mr.qscale5 = ws->w_qn.as<float>() + (size_t)Q * m; mr.packed = ix->packed; mr.pos = ix->pos; mr.blk_cell = ix->blk_cell;
  mr.cand_count = fl.cand_count; mr.violations = ix->viol; mr.out_ids = r.d_out_ids; mr.out_dist = r.d_out_dist;
  mr.found = ws->w_found.as<int32_t>(); mr.next_active = r.next; mr.n_next = ws->w_cnt.as<int32_t>();
  mr.status = r.d_status;
  mr.n_active = r.n_active; mr.W = r.W; mr.upi = r.upi; mr.L = r.L; mr.k = r.k; mr.found_rule = r.found_rule;
  mr.first_round = r.first() ? 1 : 0; mr.K = K; mr.d = ix->d; mr.sentinel = r.sentinel;
  mr.refine_all = (ix->tune.check_brackets & 1) ? 1 : 0;
  mr.slices = 0; mr.part = nullptr;
  if (r.merge_slices > 0) {
    // a batch over the flat PQ table: `merge_slices` workgroups per query, each over its share of the pseudo-lists (r.W is the
    // padded item count per query, a multiple of the slices), then merge_replay_kernel over the slices' keys
    const int SL = r.merge_slices;
    if (ws->w_part.ensure(sizeof(u64) * (size_t)r.n_active * SL * r.L)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
    mr.slices = SL; mr.part = ws->w_part.as<u64>(); mr.W = r.W / SL; mr.n_active = r.n_active * SL;
    timed_launch(ix, s, "merge_refine", [&] {
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 12, true, true>), dim3(r.n_active * SL), dim3(768), 0, s, mr);
    });
    HIP_TRY(hipGetLastError());
    MergeArgs ma;
    ma.part = ws->w_part.as<u64>(); ma.active = nullptr; ma.pos_to_id = nullptr; ma.round_rows = nullptr; ma.cand_count = nullptr;
    ma.out_ids = r.d_out_ids; ma.out_dist = r.d_out_dist; ma.found = nullptr; ma.next_active = nullptr; ma.n_next = nullptr; ma.status = nullptr;
    ma.n_active = r.n_active; ma.parts_per_query = SL; ma.L = r.L; ma.k = r.k; ma.found_rule = 0; ma.first_round = 1; ma.sentinel = r.sentinel;
    return launch_merge(ix, s, ma);
  }
  timed_launch(ix, s, "merge_refine", [&] {
    // (one batch at a time: four waves per query, the shortest latency; batches in flight: one wave per query, the smallest footprint)
    // (hundreds of survivor regions per query -- a batch over the flat PQ table: four waves, which split the selection)
    const bool many_regions = (size_t)r.W * r.upi * FUSED_NW > 256;
    if (!many_regions && r.share > 1)
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 1>), dim3(r.n_active), dim3(64), 0, s, mr);
    else if (many_regions && r.n_active <= 256)   // (a few queries with many qualifying rows each: twelve waves, 64 rows per round of the exact stage)
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 12, true>), dim3(r.n_active), dim3(768), 0, s, mr);
    else if (many_regions)                         // (a large batch: the workgroups' footprint decides, 95 against 52 us at 1024 queries)
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 4, true>), dim3(r.n_active), dim3(256), 0, s, mr);
    else
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 4>), dim3(r.n_active), dim3(256), 0, s, mr);
  });
  HIP_TRY(hipGetLastError());
  return 0;
}

// The yardstick: the reference's arithmetic for every probed row (fused3.h).  ivf_spec2_kernel -> merge_surv_kernel.
static int ivf_scan_exact(IvfRun& r, const PlanArgs& pa, const WorkTable& wt) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int K = ix->K;
  FusedArgs fa;
  fa.resid = nullptr; fa.item_query = pa.item_query; fa.queries = r.d_q; fa.coarse = ix->coarse;
  fa.sorted_item = ws->w_sorted.as<int32_t>(); fa.group_cell = wt.group_cell; fa.group_first = wt.group_first;
  fa.group_cnt = wt.group_cnt; fa.n_groups = wt.n_groups; fa.work_counter = wt.work_counter;
  fa.cbP = ix->cbP; fa.blk_off = ix->blk_off; fa.packed = ix->packed; fa.pos = ix->pos;
  fa.surv = ws->w_surv.as<u64>(); fa.surv_count = ws->w_surv_cnt.as<int32_t>();
  fa.cand_count = (r.found_rule == 1) ? ws->w_cand.as<int32_t>() : nullptr;
  fa.d = ix->d; fa.K = K; fa.L = r.L; fa.upi = r.upi;
  memcpy(&fa.sentinel_bits, &r.sentinel, 4);
  if (int rc = scan_prof_buffer(ix, ws, &fa.prof)) return rc;
  const size_t desc_off = ((size_t)2 * SPEC2_G * K * sizeof(float) + 15) & ~(size_t)15;
  const size_t flds = desc_off + 4096 + 64 + 512 + (size_t)SPEC2_G * 12 * 28 * sizeof(float);
  fa.desc_offset = (uint32_t)desc_off;
  const unsigned n_persist = (unsigned)std::min<size_t>(wt.max_groups, (size_t)ix->n_cus);
  timed_launch(ix, s, "ivf_exact_scan", [&] {
    if (K == 1024) hipLaunchKernelGGL((ivf_spec2_kernel<25, 12, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fa);
    else hipLaunchKernelGGL((ivf_spec2_kernel<25, 12, false>), dim3(n_persist), dim3(SPEC2_T), flds, s, fa);
  });
  HIP_TRY(hipGetLastError());
#ifdef FREDDY_LAB
  if (fa.prof) if (int rc = scan_prof_print(ix, s, fa.prof, n_persist)) return rc;
#endif
  MergeSurvArgs ms;
  ms.surv = fa.surv; ms.surv_count = fa.surv_count; ms.active = r.active; ms.round_rows = pa.round_rows;
  ms.cand_count = fa.cand_count; ms.out_ids = r.d_out_ids; ms.out_dist = r.d_out_dist;
  ms.found = ws->w_found.as<int32_t>(); ms.next_active = r.next; ms.n_next = ws->w_cnt.as<int32_t>();
  ms.status = r.d_status;
  ms.n_active = r.n_active; ms.W = r.W; ms.upi = r.upi; ms.L = r.L; ms.k = r.k; ms.found_rule = r.found_rule;
  ms.first_round = r.first() ? 1 : 0; ms.sentinel = r.sentinel;
  timed_launch(ix, s, "merge_surv", [&] { hipLaunchKernelGGL(merge_surv_kernel, dim3(r.n_active), dim3(64), 0, s, ms); });
  HIP_TRY(hipGetLastError());
  return 0;
}

// row blocks per workgroup of the generic scan: one workgroup per (query, probed cell) unless the list is huge -- but a
// handful of items (the reference's single-query ivfadc_search: W of them) would leave the chip to W workgroups: 32-block
// chunks then (one query over 10 lists of 3 000 rows: 30 instead of 10 workgroups)
static int generic_chunk_blocks(int n_items) { return n_items <= 64 ? 32 : 256; }

// Generic path (small batches, other m / S / K, k > 32): lut_build (residual inline) -> adc_scan -> merge_replay;
// the LUTs round-trip through memory.
static int ivf_scan_generic(IvfRun& r, const PlanArgs& pa) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int n_items = r.n_active * r.W;
  const int chunk_blocks = generic_chunk_blocks(n_items);
  const int nchunk = std::max(1, (ix->max_list_blocks + chunk_blocks - 1) / chunk_blocks);
  // (the residual r = q - coarse[cell] is formed by the LUT kernel: one launch less in a single query's chain)
  if (int rc = launch_lut(ix, s, r.d_q, pa.item_cell, ws->w_lut.as<float>(), n_items, ix->coarse, pa.item_query)) return rc;
  ScanArgs sa;
  sa.lut = ws->w_lut.as<float>(); sa.item_list = pa.item_cell; sa.item_query = pa.item_query;
  sa.blk_off = ix->blk_off; sa.packed = ix->packed; sa.pos = ix->pos; sa.part = ws->w_part.as<u64>();
  sa.cand_count = ws->w_cand.as<int32_t>();
  sa.m = ix->m; sa.K = ix->K; sa.chunk_blocks = chunk_blocks; sa.nchunk = nchunk; sa.L = r.L;
  memcpy(&sa.sentinel_bits, &r.sentinel, 4);
  if (int rc = launch_scan(ix, s, sa, n_items)) return rc;
  MergeArgs ma;
  ma.part = sa.part; ma.active = r.active; ma.pos_to_id = nullptr; ma.round_rows = pa.round_rows;
  ma.cand_count = sa.cand_count; ma.out_ids = r.d_out_ids; ma.out_dist = r.d_out_dist;
  ma.found = ws->w_found.as<int32_t>(); ma.next_active = r.next; ma.n_next = ws->w_cnt.as<int32_t>();
  ma.status = r.d_status;
  ma.n_active = r.n_active; ma.parts_per_query = r.W * nchunk; ma.L = r.L; ma.k = r.k;
  ma.found_rule = r.found_rule; ma.first_round = r.first() ? 1 : 0; ma.sentinel = r.sentinel;
  return launch_merge(ix, s, ma);
}

// One probing round of a chunk: cell selection, then the scan + merge of the path the chunk takes.
static int ivfadc_round(IvfRun& r) {
  PlanArgs pa;
  if (int rc = ivf_plan(r, pa)) return rc;
  if (r.fused) {
    WorkTable wt;
    if (int rc = ivf_work_table(r, wt)) return rc;
    return (r.scan_kernel == 5) ? ivf_scan_filter(r, pa, wt) : ivf_scan_exact(r, pa, wt);
  }
  return ivf_scan_generic(r, pa);
}

// One chunk of queries (device pointers): workspace, coarse distances and round one are enqueued on s, nothing is
// synchronised.  `share` = the batches in flight on this handle (the persistent scan takes n_cus / share CUs).  The
// state for further rounds stays in r (and in the stream's workspace): ivfadc_finish() runs them.
static int ivfadc_begin(freddy_gpu_index* ix, hipStream_t s, int share, const float* d_q, int Q, int k, int W,
                        float sentinel, int found_rule, int32_t* d_out_ids, float* d_out_dist,
                        int32_t* d_status, IvfRun& r) {
  Workspace* ws = workspace_for(ix, s);
  const int C = ix->C, m = ix->m, K = ix->K;
  if (2 * W > 1024) return fail(FREDDY_E_LIMIT, "W=%d exceeds this build's limit of 512 probes per round", W);
  r.ix = ix; r.ws = ws; r.s = s; r.d_q = d_q; r.Q = Q; r.k = k; r.W = W; r.L = std::min(2 * k, 64 * 16);
  r.sentinel = sentinel; r.d_out_ids = d_out_ids; r.d_out_dist = d_out_dist; r.d_status = d_status;
  r.share = std::max(1, share);
  // FREDDY_FOUND_BATCH_UDF = the accepted-rows rule + the batch UDF's cell limit (argmin from minDist = 1000,
  // freddy.c:853-866); ivfadc_search's cell list starts at 100.0 (freddy.c:266-283)
  r.found_rule = found_rule == FREDDY_FOUND_ROWS ? 0 : 1;
  r.cell_limit = found_rule == FREDDY_FOUND_BATCH_UDF ? 1000.0f : 100.0f;
  const size_t items = (size_t)Q * W;
  // Cell-grouped scans: residual PQ with m=12, S=25, K<=1024 and a selection width that one wave holds
  // (2k <= 64); lists longer than 8 chunks of 4096 rows would need survivor buffers out of proportion.  They
  // pay off once several (query, cell) items share a cell, i.e. for batches; option fused = 1 / 0 forces
  // them / the generic lut_build + adc_scan kernels (the tests run both).
  r.upi = std::max(1, (ix->max_list_blocks + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS);
  r.fused = ix->tune.fused != 0 && m == 12 && ix->S == 25 && K <= 1024 && ix->cbP && r.L <= 64 && r.upi <= 8 &&
            (ix->tune.fused == 1 || items >= 256);
  r.scan_kernel = (ix->tune.scan_kernel == 3 || !ix->rterm) ? 3 : 5;
  r.tiled = Q >= 32;
  r.zeroed = r.tiled || ix->d <= 1024;
  r.records_ready = false; r.merge_slices = 0;
  // (the MFMA tile is 64 queries wide and the plan keeps a query's distances in registers: batches, <= 1024 cells)
  r.approx = ix->tune.coarse_approx != 0 && r.tiled && ix->Cpad <= COARSE_STREAM_MAX_CPAD && 2 * W <= 64 && ix->d <= 300 && ix->d % 4 == 0 && ix->coarseP;
  const int Cpad = ix->Cpad, used_words = (C + 31) / 32;
  if (ws->w_distT.ensure(sizeof(float) * (size_t)Q * Cpad) ||
      ws->w_used.ensure(sizeof(uint32_t) * (size_t)Q * used_words) ||
      ws->w_item_cell.ensure(sizeof(int32_t) * items) || ws->w_item_query.ensure(sizeof(int32_t) * items) ||
      ws->w_rows.ensure(sizeof(int32_t) * Q) || ws->w_cand.ensure(sizeof(int32_t) * Q) ||
      ws->w_qn2.ensure(sizeof(float) * Q) || ws->w_item_dist.ensure(sizeof(float) * items) ||
      ws->w_found.ensure(sizeof(int32_t) * Q) || ws->w_act0.ensure(sizeof(int32_t) * Q) ||
      ws->w_act1.ensure(sizeof(int32_t) * Q) || ws->w_cnt.ensure(sizeof(int32_t) * 8))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d, W=%d)", Q, W);
  if (r.fused) {
    // cell_count[C] + cursors; cell_items[C][Q]; work table: 3 arrays of (items/G + C + 1) * upi entries
    if (ws->w_cellcnt.ensure(sizeof(int32_t) * (size_t)C * 3) || ws->w_sorted.ensure(sizeof(int32_t) * (size_t)C * Q) ||
        ws->w_groups.ensure(sizeof(int32_t) * 3 * ((items / SPEC2_G + (size_t)C + 1) * r.upi + items * r.upi)) ||   // + the (item, chunk) units of sparse cells
        ws->w_surv.ensure(sizeof(u64) * items * r.upi * FUSED_NW * FUSED_RMAX * 64) ||
        ws->w_surv_cnt.ensure(sizeof(int32_t) * items * r.upi * FUSED_NW))
      return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d, W=%d)", Q, W);
    if (r.scan_kernel == 5 &&
        (ws->w_qc.ensure(sizeof(uint32_t) * (size_t)Q * m * 512) || ws->w_qn.ensure(sizeof(float) * (size_t)Q * m * 2)))
      return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d, W=%d)", Q, W);
  } else {
    // (a later probing round has fewer items and may take the finer chunks: room for either)
    const size_t nchunk_big = (size_t)std::max(1, (ix->max_list_blocks + 255) / 256), nchunk_small = (size_t)std::max(1, (ix->max_list_blocks + 31) / 32);
    const size_t parts = std::max(items * nchunk_big, std::min<size_t>(items, 64) * nchunk_small);
    if (ws->w_resid.ensure(sizeof(float) * items * (size_t)ix->d) || ws->w_lut.ensure(sizeof(float) * items * (size_t)m * K) ||
        ws->w_part.ensure(sizeof(u64) * parts * SCAN_WAVES * r.L))
      return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d, W=%d)", Q, W);
  }

  if (int rc = ivf_coarse(r)) return rc;
  ix->last_Q = Q;
  r.n_active = Q; r.active = nullptr; r.next = ws->w_act0.as<int32_t>();
  r.round = 0;
  return ivfadc_round(r);
}

// The extra rounds of the reference's "while (foundInstances < k)" loop (freddy.c:262, :835), one host sync per
// round.  n_next: the number of queries round one left unfinished if the caller has already read it back
// (ws->w_cnt[0], after the stream drained), -1 = read it here.
static int ivfadc_finish(IvfRun& r, int n_next) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int max_rounds = (ix->C + r.W - 1) / r.W + 1;
  for (;;) {
    if (n_next < 0) {
      int32_t h = 0;
      HIP_TRY(hipMemcpyAsync(&h, ws->w_cnt.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      n_next = h;
    }
    if (n_next <= 0 || ++r.round >= max_rounds) break;
    HIP_TRY(hipMemsetAsync(ws->w_cnt.p, 0, sizeof(int32_t), s));
    r.active = r.next;
    r.next = (r.next == ws->w_act0.as<int32_t>()) ? ws->w_act1.as<int32_t>() : ws->w_act0.as<int32_t>();
    r.n_active = n_next;
    if (int rc = ivfadc_round(r)) return rc;
    n_next = -1;
  }
  return 0;
}

static int check_search_args(const freddy_gpu_index* ix, int kind, const void* q, int Q, int k, const void* oi,
                             const void* od) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (ix->kind != kind) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  if (ix->poisoned) return fail(FREDDY_E_HIP, "this handle's devices hold different tables (an append / codebook update failed part-way): unpin it and pin again");
  if (Q < 0 || k <= 0) return fail(FREDDY_E_ARG, "Q must be >= 0 and k > 0");
  if (Q > 0 && (!q || !oi || !od)) return fail(FREDDY_E_ARG, "NULL buffer");
  if (2 * k > 1024) return fail(FREDDY_E_LIMIT, "k=%d exceeds this build's limit of 512", k);
  return 0;
}

static int max_queries_per_chunk(const freddy_gpu_index* ix, int W) {
  // workspace per query: the LUTs of its W items (generic path) or their survivor regions (fused path)
  const size_t upi = (size_t)std::max(1, (ix->max_list_blocks + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS);
  const size_t lut_bytes = sizeof(float) * (size_t)ix->m * ix->K * (size_t)W;
  const size_t surv_bytes = upi <= 8 ? sizeof(u64) * (size_t)W * upi * FUSED_NW * FUSED_RMAX * 64 : 0;
  const size_t per_query = std::max(lut_bytes, surv_bytes);
  size_t n = ((size_t)ix->tune.lut_budget_mb << 20) / std::max<size_t>(per_query, 1);
  // the fused path's per-cell item buckets are [C][queries of the chunk]: keep them within 256 MiB
  if (surv_bytes) n = std::min<size_t>(n, ((size_t)256 << 20) / (sizeof(int32_t) * (size_t)std::max(ix->C, 1)));
  if (n < 1) n = 1;
  if (n > (1u << 20)) n = 1u << 20;
  return (int)n;
}

extern "C" int freddy_gpu_ivfadc_search_dev(freddy_gpu_index_t* ix, const float* d_queries, int32_t Q, int32_t k,
                                            int32_t W, float sentinel, int32_t found_rule, int32_t* d_out_ids,
                                            float* d_out_dist, int32_t* d_status, void* hip_stream) {
  if (int rc = check_search_args(ix, KIND_IVF, d_queries, Q, k, d_out_ids, d_out_dist)) return rc;
  if (W <= 0) return fail(FREDDY_E_ARG, "W must be positive");
  if (found_rule < 0 || found_rule > 2 || (found_rule == FREDDY_FOUND_BATCH_UDF && W != 1))
    return fail(FREDDY_E_ARG, "bad found_rule (FREDDY_FOUND_BATCH_UDF needs W == 1)");
  if (W > ix->C) W = ix->C;
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ix->stream;
  const int qc = max_queries_per_chunk(ix, W);
  for (int q0 = 0; q0 < Q; q0 += qc) {
    const int n = std::min(qc, Q - q0);
    IvfRun r;
    if (int rc = ivfadc_begin(ix, s, ix->tune.scan_share, d_queries + (size_t)q0 * ix->d, n, k, W, sentinel, found_rule,
                              d_out_ids + (size_t)q0 * k, d_out_dist + (size_t)q0 * k, d_status, r))
      return rc;
  }
  return FREDDY_OK;
}

// ---------------------------------------------------------------------------------------
// The host-buffer call (what the PostgreSQL hosts make: one synchronous call per batch, freddy.c:679-999) as a
// pipeline.  The batch is cut into sub-batches of <= pipeline_batch queries; sub-batch j goes to lane j mod L
// (L <= 4 library-owned streams, each with its own workspace, pinned staging and device buffers):
//   host memcpy of its queries into the lane's pinned buffer (skipped when the caller's buffer is pinned itself:
//   freddy_gpu_host_alloc) -> asynchronous H2D -> round one of the search with an explicit scan share of L -> asynchronous
//   D2H of the lists and of the round's straggler count into pinned memory -> event.
// The host only waits when it needs a lane again (or at the end), and that is where a sub-batch's rare extra probing
// rounds run and its lists are copied out: the transfers and the latency-bound ends of one sub-batch hide under the
// scans of its neighbours, and ONE stream synchronisation per lane ends the call.
// ---------------------------------------------------------------------------------------
// The lanes move their data with KERNELS, not with hipMemcpyAsync: pinned host memory is mapped into the device's address
// space, so a grid-stride copy reads the staged queries over PCIe (1.2 MB per 1024 queries: ~25 us) and a second one
// writes the lists, the straggler count and the stragglers' numbers back -- ordinary launches in the lane's stream.
// Measured (tools/pipe_trace.py): with hipMemcpyAsync (SDMA copies ordered against kernels by signals) the four lanes'
// chains ran in pairs one after the other, 1.4 ms per 4096 queries; with copy kernels they overlap like the
// device-resident batches of bench.py: 0.68 ms.
__global__ __launch_bounds__(256) void lane_copy_in_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void lane_copy_out_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dist,
                                                           const int32_t* __restrict__ n_next, const int32_t* __restrict__ unfinished,
                                                           int32_t* __restrict__ h_out, int n_out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_out) { h_out[i] = ids[i]; h_out[n_out + i] = __float_as_int(dist[i]); }
  const int nn = n_next[0];
  if (i == 0) h_out[2 * n_out] = nn;
  if (i < nn && i < n) h_out[2 * n_out + 1 + i] = unfinished[i];
}

// The same by ONE workgroup, followed by a completion word the host polls (lane_retire): the lists are in (mapped host)
// memory before the word.  n_out * 2 + n + 1 words: a few tens of KB.
__global__ __launch_bounds__(1024) void lane_copy_out_flag_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dist,
                                                                 const int32_t* __restrict__ n_next, const int32_t* __restrict__ unfinished,
                                                                 int32_t* __restrict__ h_out, int n_out, int n, int32_t* __restrict__ flag) {
  const int nn = n_next[0];
  for (int i = threadIdx.x; i < n_out; i += 1024) { h_out[i] = ids[i]; h_out[n_out + i] = __float_as_int(dist[i]); }
  if (threadIdx.x == 0) h_out[2 * n_out] = nn;
  for (int i = threadIdx.x; i < nn && i < n; i += 1024) h_out[2 * n_out + 1 + i] = unfinished[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void host_io_out_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dist, int32_t* __restrict__ h_out, int n_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_out) { h_out[i] = ids[i]; h_out[n_out + i] = __float_as_int(dist[i]); }
}

static int lane_open(Lane& l, LaneSlot& c, size_t in_bytes, size_t n, size_t n_out) {
  if (!l.stream) HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
  if (!c.done) HIP_TRY(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
  if (in_bytes > c.h_in_cap) {
    if (c.h_in) (void)hipHostFree(c.h_in);
    c.h_in = nullptr; c.h_in_cap = 0;
    const size_t want = in_bytes + in_bytes / 8 + 256;
    if (hipHostMalloc(&c.h_in, want, hipHostMallocDefault) != hipSuccess) { c.h_in = nullptr; return fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
    c.h_in_cap = want;
  }
  const size_t out_bytes = (n_out * 2 + 1 + n + 1) * 4;   // (+ the completion word)
  if (out_bytes > c.h_out_cap) {
    if (c.h_out) (void)hipHostFree(c.h_out);
    c.h_out = nullptr; c.h_out_cap = 0;
    const size_t want = out_bytes + out_bytes / 8 + 256;
    if (hipHostMalloc(&c.h_out, want, hipHostMallocDefault) != hipSuccess) { c.h_out = nullptr; return fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
    c.h_out_cap = want;
  }
  if (c.d_q.ensure(in_bytes + 16) || c.d_ids.ensure(n_out * 4) || c.d_dist.ensure(n_out * 4))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  return 0;
}

// The old shape of the call, kept for what the pipeline hands back: a (small) batch searched to the end on the library's
// own stream -- round one, then the extra rounds of the reference's "while (foundInstances < k)" loop with a host sync each.
static int ivfadc_sync_search(freddy_gpu_index* ix, const float* queries, int Q, int k, int W, float sentinel, int found_rule,
                              int32_t* out_ids, float* out_dist) {
  Workspace* ws = workspace_for(ix, ix->stream);
  hipStream_t s = ix->stream;
  if (ws->w_q.ensure(sizeof(float) * (size_t)Q * ix->d) || ws->w_out_ids.ensure(sizeof(int32_t) * (size_t)Q * k) ||
      ws->w_out_dist.ensure(sizeof(float) * (size_t)Q * k))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  HIP_TRY(hipMemcpyAsync(ws->w_q.p, queries, sizeof(float) * (size_t)Q * ix->d, hipMemcpyHostToDevice, s));
  const int qc = max_queries_per_chunk(ix, W);
  for (int q0 = 0; q0 < Q; q0 += qc) {
    const int n = std::min(qc, Q - q0);
    IvfRun r;
    if (int rc = ivfadc_begin(ix, s, 1, ws->w_q.as<float>() + (size_t)q0 * ix->d, n, k, W, sentinel, found_rule,
                              ws->w_out_ids.as<int32_t>() + (size_t)q0 * k, ws->w_out_dist.as<float>() + (size_t)q0 * k, nullptr, r))
      return rc;
    if (int rc = ivfadc_finish(r, -1)) return rc;
  }
  HIP_TRY(hipMemcpyAsync(out_ids, ws->w_out_ids.p, sizeof(int32_t) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(out_dist, ws->w_out_dist.p, sizeof(float) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

struct PipeCall {   // the arguments of one host-buffer call, for the lanes' retire step
  freddy_gpu_index* ix;
  const float* queries;
  int k, W, found_rule;
  float sentinel;
  int32_t* out_ids;
  float* out_dist;
};

// Wait for a slot's sub-batch and hand its lists to the caller.  Queries that round one left unfinished (their first W
// cells hold fewer than k rows -- rare) are searched again from the start, synchronously, with all their rounds: the
// search is deterministic, so that is the list the round-by-round continuation would have produced, and no lane has to
// keep per-round state while its stream already runs the next sub-batch.
static int lane_retire(LaneSlot& c, const PipeCall& pc) {
  if (!c.busy) return 0;
  c.busy = false;
  const int k = pc.k;
  const size_t n_out = (size_t)c.n * k;
  const int32_t* ho = static_cast<const int32_t*>(c.h_out);
  {
    // the copy-out kernel's last store is a completion word behind the lists: polled for up to a millisecond (a few
    // microseconds sooner than the event), then the event is waited for the usual way -- which is also where a fault in one
    // of the lane's kernels surfaces, before its output is trusted
    volatile const int32_t* flag = ho + 2 * n_out + 1 + (size_t)c.n;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(1000);
    int spins = 0;
    while (*flag == 0) {
      __builtin_ia32_pause();
      if ((++spins & 255) == 0 && std::chrono::steady_clock::now() > t_end) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*flag == 0) {
      HIP_TRY(hipEventSynchronize(c.done));
      HIP_TRY(hipGetLastError());
    }
  }
  memcpy(pc.out_ids + (size_t)c.q0 * k, ho, n_out * 4);
  memcpy(pc.out_dist + (size_t)c.q0 * k, ho + n_out, n_out * 4);
  const int n_next = std::min(ho[2 * n_out], c.n);
  if (n_next <= 0) return 0;
  const int d = pc.ix->d;
  std::vector<int32_t> who(ho + 2 * n_out + 1, ho + 2 * n_out + 1 + n_next);
  // (device-written numbers index the caller's buffers: a value outside the sub-batch -- e.g. after a kernel fault whose
  // error has not surfaced yet -- must never become a host read or write out of bounds)
  for (int i = 0; i < n_next; ++i)
    if (who[(size_t)i] < 0 || who[(size_t)i] >= c.n) return fail(FREDDY_E_HIP, "sub-batch returned a straggler index %d outside [0, %d)", who[(size_t)i], c.n);
  std::vector<float> q((size_t)n_next * d);
  std::vector<int32_t> ri((size_t)n_next * k);
  std::vector<float> rd((size_t)n_next * k);
  for (int i = 0; i < n_next; ++i) memcpy(&q[(size_t)i * d], pc.queries + ((size_t)c.q0 + who[(size_t)i]) * d, sizeof(float) * d);
  if (int rc = ivfadc_sync_search(pc.ix, q.data(), n_next, k, pc.W, pc.sentinel, pc.found_rule, ri.data(), rd.data())) return rc;
  for (int i = 0; i < n_next; ++i) {
    memcpy(pc.out_ids + ((size_t)c.q0 + who[(size_t)i]) * k, &ri[(size_t)i * k], sizeof(int32_t) * k);
    memcpy(pc.out_dist + ((size_t)c.q0 + who[(size_t)i]) * k, &rd[(size_t)i * k], sizeof(float) * k);
  }
  return 0;
}

// the device-side address of a pinned (hipHostMalloc / freddy_gpu_host_alloc) host buffer, or NULL for ordinary memory
static const void* pinned_device_pointer(const void* p) {
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
}

// The one-launch kernels' hand-off buffer: every published word carries the call's epoch in its top bit (one.h).  Calls of
// one shape write exactly the same words, so the epoch just flips; a different shape (or a new allocation) clears the
// buffer to epoch 0 and starts with epoch 1.
static int one_buffer(Workspace* ws, hipStream_t s, uint64_t shape, size_t bytes, uint32_t* epoch) {
  void* before = ws->w_oneb.p;
  if (ws->w_oneb.ensure(bytes)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  if (ws->w_oneb.p != before || ws->one_shape != shape) {
    HIP_TRY(hipMemsetAsync(ws->w_oneb.p, 0, ws->w_oneb.cap, s));
    ws->one_shape = shape;
    ws->one_epoch = 1;
  } else {
    ws->one_epoch ^= 1u;
  }
  *epoch = ws->one_epoch;
  return 0;
}

// ONE ivfadc_search query as one launch (one.h ivf_one_kernel).  Returns through *verdict: 2 = the list is in out_ids /
// out_dist; anything else = not answered here (shape not covered, the reference would probe a second time, or the grid
// never met at a barrier): the caller takes the multi-round path.
static bool ivf_one_shape(const freddy_gpu_index* ix, int Q, int k, int W, int found_rule) {
  return ix->tune.one_launch && !ix->one_launch_failed && Q == 1 && ix->kind == KIND_IVF && ix->m == 12 && ix->S == 25 && ix->d == 300 &&
         ix->K <= 1024 && (ix->K & 3) == 0 && W <= 32 && 2 * k <= 64 && ix->C <= 4096 && ix->coarse && ix->cbT &&
         found_rule != FREDDY_FOUND_BATCH_UDF && ix->replicas.empty();
}
static int ivf_one(freddy_gpu_index* ix, const float* queries, int k, int W, float sentinel, int found_rule, int32_t* out_ids,
                   float* out_dist, int* verdict) {
  *verdict = 0;
  hipStream_t s = ix->stream;
  Workspace* ws = workspace_for(ix, s);
  const int K = ix->K, C = ix->C, L = 2 * k;
  const size_t lutN = (size_t)12 * K, n_out = (size_t)k;
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)ix->n_cus, (int64_t)256, (int64_t)(56 * 1024) / (8 * L)}));
  if (G < W) return 0;   // (an item per workgroup at least)
  const size_t need_out = n_out * 8 + 16;
  if (need_out > ix->hio_out_cap) {
    if (ix->hio_out) (void)hipHostFree(ix->hio_out);
    ix->hio_out = nullptr; ix->hio_out_cap = 0;
    if (hipHostMalloc(&ix->hio_out, need_out + 256, hipHostMallocDefault) != hipSuccess) { ix->hio_out = nullptr; return fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
    ix->hio_out_cap = need_out + 256;
  }
  // the hand-off buffer: coarse distances | the W tables | the workgroups' lists | their accepted-row counts
  const size_t lut_off = (sizeof(float) * ((size_t)C + 8) + 255) & ~(size_t)255;
  const size_t part_off = (lut_off + sizeof(float) * (size_t)W * lutN + 255) & ~(size_t)255;
  const size_t cnt_off = (part_off + sizeof(u64) * (size_t)G * L + 255) & ~(size_t)255;
  uint32_t epoch = 0;
  if (int rc = one_buffer(ws, s, (2ull << 60) | ((uint64_t)C << 44) | ((uint64_t)K << 32) | ((uint64_t)W << 24) | ((uint64_t)G << 12) | (uint64_t)L,
                          cnt_off + sizeof(uint32_t) * (size_t)G, &epoch)) return rc;
  static const bool one_prof = getenv("FREDDY_GPU_ONE_PROF") != nullptr;
  if (one_prof && ws->w_one.ensure(256)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  int32_t* h_ids = static_cast<int32_t*>(ix->hio_out);
  float* h_dist = reinterpret_cast<float*>(h_ids + n_out);
  int32_t* err = reinterpret_cast<int32_t*>(static_cast<char*>(ix->hio_out) + n_out * 8);
  *err = 0;
  IvfOneArgs a;
  memcpy(a.qv, queries, sizeof(a.qv));
  a.coarse = ix->coarse; a.cbT = ix->cbT; a.list_off = ix->list_off; a.blk_off = ix->blk_off; a.packed = ix->packed; a.pos = ix->pos;
  char* ob = ws->w_oneb.as<char>();
  a.dist_g = reinterpret_cast<float*>(ob); a.lut_g = reinterpret_cast<float*>(ob + lut_off); a.part = reinterpret_cast<u64*>(ob + part_off);
  a.cnt_g = reinterpret_cast<uint32_t*>(ob + cnt_off);
  a.out_ids = h_ids; a.out_dist = h_dist; a.epoch = epoch; a.err = err;
  a.C = C; a.K = K; a.W = W; a.L = L; a.k = k; a.found_rule = found_rule == FREDDY_FOUND_ROWS ? 0 : 1;
  a.cell_limit = 100.0f; a.sentinel = sentinel;
  a.prof = one_prof ? ws->w_one.as<unsigned long long>() + 8 : nullptr;
  memcpy(&a.sentinel_bits, &sentinel, 4);
  const size_t n_mine = ((size_t)C + G - 1) / G;
  const size_t lds = std::max({(n_mine + 1) * 300 * sizeof(float), (size_t)C * 4 + 64 + 64 * sizeof(u64) + 64,
                               ((lutN * 4 + 15) & ~(size_t)15) + (size_t)ONE_WAVES * 64 * sizeof(u64),
                               (size_t)ONE_WAVES * 64 * sizeof(u64) + (size_t)G * L * sizeof(u64)});
  if (lds > 60 * 1024) return 0;
  timed_launch(ix, s, "ivf_one", [&] { hipLaunchKernelGGL((ivf_one_kernel<25>), dim3((unsigned)G), dim3(ONE_WG), lds, s, a); });
  HIP_TRY(hipGetLastError());
  {   // (the kernel's last store is this word: polled for up to a millisecond, then the stream is waited for the usual way)
    volatile int32_t* flag = err;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(1000);
    int spins = 0;
    while (*flag == 0) {
      __builtin_ia32_pause();
      if ((++spins & 255) == 0 && std::chrono::steady_clock::now() > t_end) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*flag != 2) HIP_TRY(hipStreamSynchronize(s));
  }
  if (one_prof) {
    HIP_TRY(hipStreamSynchronize(s));
    unsigned long long st[16];
    (void)hipMemcpy(st, ws->w_one.as<unsigned long long>() + 8, sizeof(st), hipMemcpyDeviceToHost);
    fprintf(stderr, "[ivf_one] wg0: coarse %.2f barrier %.2f plan %.2f tables %.2f barrier %.2f stage %.2f scan %.2f publish %.2f | last: since wg0 start %.2f load %.2f merge+list %.2f us\n",
            (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01, (st[6] - st[5]) * 0.01,
            (st[7] - st[6]) * 0.01, (st[8] - st[7]) * 0.01, (st[10] - st[0]) * 0.01, (st[11] - st[10]) * 0.01, (st[12] - st[11]) * 0.01);
  }
  if (*err == 2) {
    memcpy(out_ids, h_ids, n_out * 4);
    memcpy(out_dist, h_dist, n_out * 4);
    *verdict = 2;
    return 0;
  }
  if (*err != 3) {   // a poll ran out: counters re-armed, this handle keeps to the multi-launch paths
    ix->one_launch_failed = true;
    ws->one_shape = 0;
    HIP_TRY(hipStreamSynchronize(s));
  }
  return 0;
}

// the batch [0, Q) of one device's handle
static int ivfadc_host_search(freddy_gpu_index* ix, const float* queries, int Q, int k, int W, float sentinel, int found_rule,
                              int32_t* out_ids, float* out_dist) {
  HIP_TRY(hipSetDevice(ix->device));
  if (ivf_one_shape(ix, Q, k, W, found_rule)) {
    int verdict = 0;
    if (int rc = ivf_one(ix, queries, k, W, sentinel, found_rule, out_ids, out_dist, &verdict)) return rc;
    if (verdict == 2) return FREDDY_OK;
  }
  const int cap = std::max(1, std::min(max_queries_per_chunk(ix, W), ix->tune.pipeline_batch));
  const int n_sub = (Q + cap - 1) / cap;
  const int per = (Q + n_sub - 1) / n_sub;               // equal sub-batches rather than full ones and a remainder
  const int n_lanes = std::min(n_sub, std::min(ix->tune.pipeline_lanes, FREDDY_LANES));
  const float* pinned_in = static_cast<const float*>(pinned_device_pointer(queries));
  const size_t row = sizeof(float) * (size_t)ix->d;
  const PipeCall pc{ix, queries, k, W, found_rule, sentinel, out_ids, out_dist};
  int rc = 0;
#ifdef FREDDY_LAB
  static const bool trace = getenv("FREDDY_GPU_PIPE_TRACE") != nullptr;   // host timestamps of the pipeline's steps on stderr (lab builds)
#else
  constexpr bool trace = false;
#endif
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = trace ? now_us() : 0.0;
  auto slot_of = [&](int j) -> LaneSlot& { return ix->lanes[j % n_lanes].slot[(j / n_lanes) & 1]; };
  for (int j = 0; j < n_sub && !rc; ++j) {
    Lane& l = ix->lanes[j % n_lanes];
    LaneSlot& c = slot_of(j);
    double t0 = trace ? now_us() : 0.0, t1 = 0, t2 = 0, t3 = 0;
    if ((rc = lane_retire(c, pc))) break;
    if (trace) t1 = now_us();
    const int q0 = j * per, n = std::min(per, Q - q0);
    if ((rc = lane_open(l, c, row * n, (size_t)n, (size_t)n * k))) break;
    c.q0 = q0; c.n = n;
    const float* src = pinned_in ? pinned_in + (size_t)q0 * ix->d : nullptr;
    const bool stage = !src || reinterpret_cast<uintptr_t>(src) % 16 || (row * n) % 16;   // (the copy kernel moves whole 16-byte words)
    const float* d_queries = c.d_q.as<float>();
    if (n <= 8) {
      // a handful of queries: the kernels read them where they are staged (pinned, mapped) -- one launch less
      if (stage) { memcpy(c.h_in, queries + (size_t)q0 * ix->d, row * n); src = static_cast<const float*>(c.h_in); }
      d_queries = src;
      if (trace) t2 = now_us();
    } else {
      // pageable queries cross in pieces: the copy kernel of a piece reads it over PCIe while the host stages the next one
      // (1.2 MB per 1024 queries: 28 us of memcpy + 25 us of PCIe, back to back until round 4)
      const size_t total = row * n, n16_all = (total + 15) / 16;
      const int pieces = stage && total >= (size_t)512 * 1024 ? 4 : 1;
      const size_t per16 = (n16_all + pieces - 1) / pieces;
      for (int pi = 0; pi < pieces; ++pi) {
        const size_t w0 = (size_t)pi * per16, w1 = std::min(n16_all, w0 + per16);
        if (w0 >= w1) break;
        const size_t b0 = w0 * 16, b1 = std::min(total, w1 * 16);
        if (stage) memcpy(static_cast<char*>(c.h_in) + b0, reinterpret_cast<const char*>(queries + (size_t)q0 * ix->d) + b0, b1 - b0);
        const char* from = stage ? static_cast<const char*>(c.h_in) : reinterpret_cast<const char*>(src);
        hipLaunchKernelGGL(lane_copy_in_kernel, dim3((unsigned)std::min<size_t>((w1 - w0 + 255) / 256, 512)), dim3(256), 0, l.stream,
                           reinterpret_cast<const uint4*>(from + b0), reinterpret_cast<uint4*>(c.d_q.as<char>() + b0), w1 - w0);
        if (hipGetLastError() != hipSuccess) { rc = fail(FREDDY_E_HIP, "launch of the query copy failed"); break; }
      }
      if (rc) break;
      if (trace) t2 = now_us();
    }
    IvfRun r;
    if ((rc = ivfadc_begin(ix, l.stream, n_lanes, d_queries, n, k, W, sentinel, found_rule, c.d_ids.as<int32_t>(),
                           c.d_dist.as<float>(), nullptr, r)))
      break;
    if (trace) t3 = now_us();
    const int n_out = n * k;
    int32_t* h_flag = static_cast<int32_t*>(c.h_out) + 2 * (size_t)n_out + 1 + (size_t)n;
    *h_flag = 0;
    hipLaunchKernelGGL(lane_copy_out_flag_kernel, dim3(1), dim3(1024), 0, l.stream, c.d_ids.as<int32_t>(),
                       c.d_dist.as<float>(), r.ws->w_cnt.as<int32_t>(), r.next, static_cast<int32_t*>(c.h_out), n_out, n, h_flag);
    if (hipGetLastError() != hipSuccess || hipEventRecord(c.done, l.stream) != hipSuccess) { rc = fail(FREDDY_E_HIP, "launch of the result copy failed"); break; }
    c.busy = true;
    if (trace)
      fprintf(stderr, "[pipe] sub %d lane %d n=%d  t=%.0f us: retire %.0f, stage %.0f, launches %.0f, copy-out + event %.0f\n", j, j % n_lanes, n,
              t0 - t_start, t1 - t0, t2 - t1, t3 - t2, now_us() - t3);
  }
  // drain in submission order (oldest first)
  for (int j = std::max(0, n_sub - 2 * n_lanes); j < n_sub && !rc; ++j) {
    const double t0 = trace ? now_us() : 0.0;
    rc = lane_retire(slot_of(j), pc);
    if (trace) fprintf(stderr, "[pipe] drain sub %d  t=%.0f us: %.0f\n", j, t0 - t_start, now_us() - t0);
  }
  if (rc)   // a failed call: nothing of it may still be in flight when the caller gets its buffers back
    for (Lane& l : ix->lanes) {
      if (l.stream) (void)hipStreamSynchronize(l.stream);
      for (LaneSlot& c : l.slot) c.busy = false;
    }
  return rc;
}

// Q queries split contiguously over a handle and its replicas (freddy_gpu_pin_ivf_multi): part g of G gets
// [lo, hi) with sizes differing by at most one.  fn(part, index of that part, lo, hi) runs on its own host thread
// for every part but the first; the first failure's code and message are returned on the caller's thread.
template <class F>
static int over_replicas(freddy_gpu_index* ix, int Q, F&& fn) {
  const int G = 1 + (int)ix->replicas.size();
  if (G == 1 || Q < 2 * G) return fn(ix, 0, Q);
  std::vector<int> rcs((size_t)G, 0);
  std::vector<std::string> msgs((size_t)G);
  std::vector<std::thread> th;
  const int base = Q / G, rem = Q % G;
  auto bounds = [&](int g, int* lo, int* hi) { *lo = g * base + std::min(g, rem); *hi = *lo + base + (g < rem ? 1 : 0); };
  for (int g = 1; g < G; ++g)
    th.emplace_back([&, g] {
      int lo, hi;
      bounds(g, &lo, &hi);
      rcs[(size_t)g] = fn(ix->replicas[(size_t)g - 1], lo, hi);
      if (rcs[(size_t)g]) msgs[(size_t)g] = g_err;
    });
  int lo, hi;
  bounds(0, &lo, &hi);
  rcs[0] = fn(ix, lo, hi);
  if (rcs[0]) msgs[0] = g_err;
  for (std::thread& t : th) t.join();
  for (int g = 0; g < G; ++g)
    if (rcs[(size_t)g]) return fail(rcs[(size_t)g], "device %d: %s", g == 0 ? ix->device : ix->replicas[(size_t)g - 1]->device, msgs[(size_t)g].c_str());
  return 0;
}

extern "C" int freddy_gpu_ivfadc_search(freddy_gpu_index_t* ix, const float* queries, int32_t Q, int32_t k, int32_t W,
                                        float sentinel, int32_t found_rule, int32_t* out_ids, float* out_dist) {
  if (int rc = check_search_args(ix, KIND_IVF, queries, Q, k, out_ids, out_dist)) return rc;
  if (W <= 0) return fail(FREDDY_E_ARG, "W must be positive");
  if (found_rule < 0 || found_rule > 2 || (found_rule == FREDDY_FOUND_BATCH_UDF && W != 1))
    return fail(FREDDY_E_ARG, "bad found_rule (FREDDY_FOUND_BATCH_UDF needs W == 1)");
  if (W > ix->C) W = ix->C;
  if (Q == 0) return FREDDY_OK;
  return over_replicas(ix, Q, [&](freddy_gpu_index* part, int lo, int hi) {
    return ivfadc_host_search(part, queries + (size_t)lo * ix->d, hi - lo, k, W, sentinel, found_rule, out_ids + (size_t)lo * k,
                              out_dist + (size_t)lo * k);
  });
}

extern "C" int freddy_gpu_host_alloc(void** out, size_t bytes) {
  setenv("GPU_MAX_HW_QUEUES", "6", 0);   // (as open_device: this call may be the process's first HIP call; a deployment sets it in the environment)
  if (!out) return fail(FREDDY_E_ARG, "NULL argument");
  *out = nullptr;
  if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { *out = nullptr; return fail(FREDDY_E_NOMEM, "pinned host allocation of %zu bytes failed", bytes); }
  return FREDDY_OK;
}
extern "C" int freddy_gpu_host_free(void* p) {
  if (p) HIP_TRY(hipHostFree(p));
  return FREDDY_OK;
}

extern "C" int freddy_gpu_pin_ivf_multi(const freddy_ivf_desc* t, const int* devices, int n_devices, freddy_gpu_index_t** out) {
  if (!devices || n_devices < 1 || !out) return fail(FREDDY_E_ARG, "bad device list");
  freddy_gpu_index* first = nullptr;
  if (int rc = freddy_gpu_pin_ivf(t, devices[0], &first)) return rc;
  for (int g = 1; g < n_devices; ++g) {
    freddy_gpu_index* rep = nullptr;
    if (int rc = freddy_gpu_pin_ivf(t, devices[g], &rep)) { free_index(first); return rc; }
    first->replicas.push_back(rep);
  }
  (void)hipSetDevice(devices[0]);
  *out = first;
  return FREDDY_OK;
}
extern "C" int freddy_gpu_replica_count(const freddy_gpu_index_t* ix) { return ix ? 1 + (int)ix->replicas.size() : 0; }

// ---------------------------------------------------------------------------------------
// exhaustive / subset PQ
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// Batches over the flat PQ table through the cell-grouped filter + refine scan (fused5.h).
//
// adc_scan_kernel runs one workgroup per (query, chunk): every query re-reads the code table from the caches and gathers
// 4-byte LUT entries one (query, row, position) at a time.  The IVFADC scan shares a chunk's rows among 16 queries and
// gathers eight 16-bit table values per LDS access -- and pq_search's distance is ivfadc_search's with a residual
// r = q - 0: the flat table is pinned a second time only as METADATA -- pseudo-lists of 4096 consecutive rows
// (FUSED_UNIT_BLOCKS blocks; the packed codes are shared), a zero centroid per list, the row terms sum_p |c|^2, the rows'
// ids -- in a shadow index of kind IVF, and a batch "probes" every list: items (query, list) for all pairs, no coarse
// distances, no plan.  The exact stage then evaluates (q_i - 0) - c_i: x - 0 = x exactly, so its squares, their
// order of summation (index_utils.c:500-508, 1126-1133) and the guarded insertion in ascending id order are pq_search's
// (freddy.c:28-152).  The item's bound on |r|^2 is squareDistance(q, 0) evaluated the reference's way.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_shadow_meta_kernel(const int32_t* __restrict__ pos, const int32_t* __restrict__ ids,
                                                            int64_t n_blocks, int64_t n_rows, int lists, int32_t* __restrict__ list_off,
                                                            int32_t* __restrict__ blk_off, int32_t* __restrict__ blk_cell,
                                                            int32_t* __restrict__ pos_ids) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i <= lists) {
    const int64_t r = i * (FUSED_UNIT_BLOCKS * 64), b = i * FUSED_UNIT_BLOCKS;
    list_off[i] = (int32_t)(r < n_rows ? r : n_rows);
    blk_off[i] = (int32_t)(b < n_blocks ? b : n_blocks);
  }
  if (i < n_blocks) blk_cell[i] = (int32_t)(i / FUSED_UNIT_BLOCKS);
  if (i < n_blocks * 64) { const int32_t r = pos[i]; pos_ids[i] = r >= 0 ? ids[r] : -1; }
}

// One workgroup per query: A = squareDistance(q, 0) (sequential binary32, index_utils.c:500-508), the query's items --
// one per pseudo-list -- and its slot in every list's bucket.
__global__ __launch_bounds__(256) void pq_items_kernel(const float* __restrict__ queries, int Q, int d, int lists, int64_t n_rows,
                                                      int32_t* __restrict__ item_cell, int32_t* __restrict__ item_query,
                                                      float* __restrict__ item_dist, int32_t* __restrict__ cell_items,
                                                      int32_t* __restrict__ cell_count, int32_t* __restrict__ round_rows) {
  __shared__ float A_s;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) {
    float acc = 0.0f;
    for (int i = 0; i < d; ++i) { const float t = queries[(size_t)q * d + i] - 0.0f; acc = acc + t * t; }
    A_s = acc;
    round_rows[q] = (int32_t)n_rows;
  }
  __syncthreads();
  const float A = A_s;
  for (int c = threadIdx.x; c < lists; c += 256) {
    const int it = q * lists + c;
    item_cell[it] = c; item_query[it] = q; item_dist[it] = A;
    cell_items[(size_t)c * Q + q] = it;
    if (q == 0) cell_count[c] = Q;
  }
}

static bool pq_fused_shape(const freddy_gpu_index* ix) {
  return ix->kind == KIND_PQ && ix->cbR && ix->m == 12 && ix->S == 25 && ix->K <= FUSED_T * FUSED_E && ix->n_blocks > 0 && ix->N > 0;
}

static bool pq_use_fused(const freddy_gpu_index* ix, int Q, int k) {
  if (ix->tune.pq_fused == 0 || !pq_fused_shape(ix) || 2 * k > 64) return false;
  return ix->tune.pq_fused > 0 || Q >= 16;
}
// A batch over the flat PQ table needs no probe plan and no work table: every query "probes" every pseudo-list, so the
// work entries are (group of 16 queries, pseudo-list) and their records follow from the query's table scale alone.  One
// workgroup per query: |q|^2 in the reference's order (squareDistance(q, 0): the coarse distance of the zero centroid, the
// bound item_bounds builds on), then the query's lane of every record of its group; the first query of a group also writes
// the records' headers.  Replaces pq_items + work_table + entry_record kernels (27 us of three dependent launches).
// item index = q * W + list (W >= lists: padded to a multiple of the merge's slices; the padding items have no entry, their
// survivor regions stay zero).
struct PqFrontArgs {
  const float* queries; int Q, d, lists, W; int64_t n_rows;
  const int32_t* blk_off; const int32_t* list_off;
  const float* cbT; const float* cmax; const float* pmax;
  float* qn; float* qscale; uint32_t* qc; int m, K;
  float sentinel;
  int32_t* item_cell; int32_t* item_query; float* item_dist; int32_t* round_rows; int32_t* records; int32_t* n_groups;
};
__device__ __forceinline__ void pq_records_body(const PqFrontArgs& a, int q, unsigned char* smem) {
  const float* __restrict__ queries = a.queries;
  const int Q = a.Q, d = a.d, lists = a.lists, W = a.W;
  const int64_t n_rows = a.n_rows;
  const int32_t* __restrict__ blk_off = a.blk_off; const int32_t* __restrict__ list_off = a.list_off;
  const float* __restrict__ pmax = a.pmax;
  const float sentinel = a.sentinel;
  int32_t* __restrict__ item_cell = a.item_cell; int32_t* __restrict__ item_query = a.item_query; float* __restrict__ item_dist = a.item_dist;
  int32_t* __restrict__ round_rows = a.round_rows; int32_t* __restrict__ records = a.records; int32_t* __restrict__ n_groups = a.n_groups;
  float* sqs = reinterpret_cast<float*>(smem);          // [1024]
  float* qn_s = sqs + 1024;                              // [16] |q_p| rounded up, as query_codebook5_body forms it
  float* fs = qn_s + 16;                                 // [0] A, [1] scale
  const int tid = threadIdx.x;
  for (int i = tid; i < d; i += 256) { const float t = queries[(size_t)q * d + i] - 0.0f; sqs[i] = t * t; }
  // the query's per-position norms and its table scale: the very operations of query_codebook5_body (same order, same roundings),
  // so that this workgroup needs nothing from the table units of the same launch
  if (tid < 16) {
    const int pp = tid, S = d / a.m;
    float best = 0.0f;
    if (pp < a.m) {
      float n2 = 0.0f;
      for (int j = 0; j < S; ++j) { const float v = queries[(size_t)q * d + pp * S + j]; n2 = __builtin_fmaf(v, v, n2); }
      const float nrm = __builtin_sqrtf(n2) * (1.0f + 1e-5f);
      qn_s[pp] = nrm;
      best = 2.0f * nrm * a.cmax[pp];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (pp == 0) fs[1] = best * (1.0f / (float)FILT5_VMAX) * (1.0f + 1e-6f);
  }
  __syncthreads();
  if (tid == 0) {
    float acc = 0.0f;
    for (int i = 0; i < d; ++i) acc = acc + sqs[i];     // index_utils.c:500-508, i ascending
    fs[0] = acc;
    round_rows[q] = (int32_t)n_rows;
    if (q == 0) n_groups[0] = ((Q + SCAN5_G - 1) / SCAN5_G) * lists;
  }
  __syncthreads();
  const float A = fs[0];
  const float sc = fs[1];
  const ItemBounds ib = item_bounds(A, filter_width5<12>(qn_s, pmax, sc), sentinel);
  const int g = q / SCAN5_G, slot = q % SCAN5_G;
  const int cnt = (Q - g * SCAN5_G < SCAN5_G) ? Q - g * SCAN5_G : SCAN5_G;
  for (int c = tid; c < lists; c += 256) {
    const int it = q * W + c;
    item_cell[it] = c; item_query[it] = q; item_dist[it] = A;
    int32_t* rec = records + ((size_t)g * lists + c) * REC_DW;
    rec[8 + slot] = it;
    rec[24 + slot] = q;
    rec[40 + slot] = (int32_t)__float_as_uint(ib.off);
    rec[56 + slot] = (int32_t)__float_as_uint(ib.e);
    rec[72 + slot] = (int32_t)__float_as_uint(ib.shift);
    rec[88 + slot] = (int32_t)ib.lo_bits;
    rec[104 + slot] = (int32_t)ib.hi_bits;
    rec[128 + slot] = (int32_t)__float_as_uint(sc < 1e30f ? sc : 0.0f);
    if (slot == 0) {
      const int b0 = blk_off[c];
      rec[0] = c; rec[1] = cnt; rec[2] = 0; rec[3] = b0; rec[4] = blk_off[c + 1] - b0; rec[5] = list_off[c + 1] - list_off[c];
      // the slots beyond the group's queries: no item, the first query's number (a valid table), no bounds (entry_record5_kernel)
      const ItemBounds none = item_bounds(0.0f, 0.0f, sentinel);
      for (int u = cnt; u < SCAN5_G; ++u) {
        rec[8 + u] = -1; rec[24 + u] = q;
        rec[40 + u] = (int32_t)__float_as_uint(none.off); rec[56 + u] = (int32_t)__float_as_uint(none.e); rec[72 + u] = (int32_t)__float_as_uint(none.shift);
        rec[88 + u] = (int32_t)none.lo_bits; rec[104 + u] = (int32_t)none.hi_bits; rec[128 + u] = 0;
      }
    }
  }
}

// The table units of query_codebook5_kernel and the record workgroups above as ONE launch (neither needs the other: the record
// workgroups form the query's scale themselves): a dependent launch less in a PQ batch's chain.
__global__ __launch_bounds__(256) void pq_front_kernel(PqFrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n_table = a.m * ((a.Q + 15) / 16);
  const int b = blockIdx.x;
  if (b < n_table) query_codebook5_body<25, 16>(a.queries, a.cbT, a.cmax, a.qn, a.qscale, a.qc, a.Q, a.d, a.m, a.K, b % a.m, b / a.m, smem);
  else pq_records_body(a, b - n_table, smem);
}

// survivor regions: 32 KiB per (query, pseudo-list) within the workspace budget; the buckets [lists][queries] within 256 MiB
static int pq_fused_queries_per_chunk(const freddy_gpu_index* ix, int64_t n_blocks) {
  const size_t lists = (size_t)((n_blocks + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS);
  size_t n = ((size_t)ix->tune.lut_budget_mb << 20) / (sizeof(u64) * lists * FUSED_NW * FUSED_RMAX * 64);
  n = std::min<size_t>(n, ((size_t)256 << 20) / (sizeof(int32_t) * lists));
  return (int)std::max<size_t>(16, std::min<size_t>(n, 1u << 16));
}

// An IVF-shaped view (*view; created on first use) of `n_rows` rows in `n_blocks` packed blocks: pseudo-lists, zero centroids,
// row terms, ids.  Everything is enqueued on s; nothing is synchronised.
static int pq_view_refresh(freddy_gpu_index* ix, freddy_gpu_index** view, hipStream_t s, const uint32_t* packed, const int32_t* pos,
                           int64_t n_blocks, int64_t n_rows) {
  freddy_gpu_index* fx = *view;
  if (!fx) {
    fx = new freddy_gpu_index();
    fx->shadow_of = ix;
    fx->kind = KIND_IVF; fx->device = ix->device; fx->stream = ix->stream; fx->n_cus = ix->n_cus;
    fx->d = ix->d; fx->m = ix->m; fx->K = ix->K; fx->S = ix->S; fx->M2 = ix->M2;
    if (hipMalloc((void**)&fx->viol, 4 * sizeof(int32_t)) != hipSuccess || hipMemset(fx->viol, 0, 4 * sizeof(int32_t)) != hipSuccess) {
      free_index(fx);
      return fail(FREDDY_E_NOMEM, "device allocation failed (PQ table as pseudo-lists)");
    }
    *view = fx;
  }
  fx->tune = ix->tune;
  fx->cbT = ix->cbT; fx->cbR = ix->cbR; fx->pmax = ix->pmax; fx->cmaxp = ix->cmaxp; fx->cbF = ix->cbF;   // shared with the owner
  fx->packed = const_cast<uint32_t*>(packed);
  fx->packed8 = (packed == ix->packed) ? ix->packed8 : nullptr; fx->packed8_own = false;   // (a subset's gathered rows: the int16 layout)
  fx->N = n_rows; fx->n_blocks = n_blocks; fx->max_list_blocks = FUSED_UNIT_BLOCKS;
  const int lists = (int)((n_blocks + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS);
  fx->C = lists;
  const size_t slots = (size_t)n_blocks * 64;
  if (fx->v_coarse.ensure(sizeof(float) * (size_t)lists * ix->d) || fx->v_list_off.ensure(sizeof(int32_t) * ((size_t)lists + 1)) ||
      fx->v_blk_off.ensure(sizeof(int32_t) * ((size_t)lists + 1)) || fx->v_blk_cell.ensure(sizeof(int32_t) * (size_t)n_blocks) ||
      fx->v_pos.ensure(sizeof(int32_t) * slots) || fx->v_rterm.ensure(sizeof(float) * slots))
    return fail(FREDDY_E_NOMEM, "device allocation failed (PQ table as pseudo-lists)");
  fx->coarse = fx->v_coarse.as<float>(); fx->list_off = fx->v_list_off.as<int32_t>(); fx->blk_off = fx->v_blk_off.as<int32_t>();
  fx->blk_cell = fx->v_blk_cell.as<int32_t>(); fx->pos = fx->v_pos.as<int32_t>(); fx->rterm = fx->v_rterm.as<float>();
  HIP_TRY(hipMemsetAsync(fx->coarse, 0, sizeof(float) * (size_t)lists * ix->d, s));
  hipLaunchKernelGGL(pq_shadow_meta_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, pos, ix->ids, n_blocks, n_rows, lists,
                     fx->list_off, fx->blk_off, fx->blk_cell, fx->pos);
  hipLaunchKernelGGL(row_term_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, fx->packed, fx->blk_cell, fx->coarse, fx->cbR,
                     fx->rterm, (int64_t)slots, fx->M2, fx->d, fx->m, fx->K, fx->S);
  HIP_TRY(hipGetLastError());
  return 0;
}

// the whole table's view: built once (and again after rows were appended or the codebook was replaced)
static int pq_shadow_build(freddy_gpu_index* ix) {
  if (ix->pq_shadow) return 0;
  if (int rc = pq_view_refresh(ix, &ix->pq_shadow, ix->stream, ix->packed, ix->pos, ix->n_blocks, ix->N)) {
    if (ix->pq_shadow) { free_index(ix->pq_shadow); ix->pq_shadow = nullptr; }
    return rc;
  }
  HIP_TRY(hipStreamSynchronize(ix->stream));   // (searches may come in on other streams)
  return 0;
}

static int ivf_work_table(IvfRun& r, WorkTable& wt);
static int ivf_scan_filter(IvfRun& r, const PlanArgs& pa, const WorkTable& wt);

static int pq_fused_chunk(freddy_gpu_index* ix, freddy_gpu_index* fx, hipStream_t s, const float* d_q, int Q, int k, float sentinel,
                          int32_t* d_out_ids, float* d_out_dist) {
  fx->tune = ix->tune;
  Workspace* ws = workspace_for(fx, s);
  const int lists = fx->C, m = fx->m, K = fx->K;
  // the merge of a small batch over many pseudo-lists: four workgroups per query, each over a quarter of the lists (64 queries
  // x 1 960 survivor regions on 64 workgroups took 62 us on a quarter of the chip); the item space of a query is padded to
  // a multiple of the slices
  const int SL = (lists >= 32 && Q <= 256 && (size_t)lists * FUSED_NW > 256) ? 4 : 0;
  const int W = SL ? ((lists + SL - 1) / SL) * SL : lists;
  IvfRun r;
  r.ix = fx; r.ws = ws; r.s = s; r.d_q = d_q; r.Q = Q; r.k = k; r.W = W; r.L = 2 * k;
  r.sentinel = sentinel; r.cell_limit = 0.0f; r.d_out_ids = d_out_ids; r.d_out_dist = d_out_dist; r.d_status = nullptr;
  r.found_rule = 0; r.upi = 1; r.fused = true; r.scan_kernel = 5; r.tiled = false; r.zeroed = false; r.approx = false;
  r.records_ready = true; r.merge_slices = SL;
  r.n_active = Q; r.round = 0; r.active = nullptr;
  r.share = std::max(1, ix->tune.scan_share);   // (the caller's contract: its batches in flight on this handle)
  const size_t items = (size_t)Q * W;
  const size_t n_entries = (size_t)((Q + SCAN5_G - 1) / SCAN5_G) * lists;
  if (ws->w_item_cell.ensure(sizeof(int32_t) * items) || ws->w_item_query.ensure(sizeof(int32_t) * items) ||
      ws->w_item_dist.ensure(sizeof(float) * items) || ws->w_rows.ensure(sizeof(int32_t) * Q) || ws->w_cand.ensure(sizeof(int32_t) * Q) ||
      ws->w_found.ensure(sizeof(int32_t) * Q) || ws->w_act0.ensure(sizeof(int32_t) * Q) || ws->w_act1.ensure(sizeof(int32_t) * Q) ||
      ws->w_cnt.ensure(sizeof(int32_t) * 8) || ws->w_records.ensure(sizeof(int32_t) * REC_DW * n_entries) ||
      ws->w_surv.ensure(sizeof(u64) * items * r.upi * FUSED_NW * FUSED_RMAX * 64) ||
      ws->w_surv_cnt.ensure(sizeof(int32_t) * items * r.upi * FUSED_NW) ||
      ws->w_qc.ensure(sizeof(uint32_t) * (size_t)Q * m * 512) || ws->w_qn.ensure(sizeof(float) * (size_t)Q * m * 2))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d over %d pseudo-lists)", Q, lists);
  r.next = ws->w_act0.as<int32_t>();
  HIP_TRY(hipMemsetAsync(ws->w_cnt.p, 0, sizeof(int32_t) * 8, s));
  HIP_TRY(hipMemsetAsync(ws->w_surv_cnt.p, 0, sizeof(int32_t) * items * r.upi * FUSED_NW, s));
  PlanArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.item_cell = ws->w_item_cell.as<int32_t>(); pa.item_query = ws->w_item_query.as<int32_t>(); pa.item_dist = ws->w_item_dist.as<float>();
  pa.round_rows = ws->w_rows.as<int32_t>(); pa.n_active = Q; pa.C = lists; pa.W = W;
  WorkTable wt;
  wt.max_groups = n_entries; wt.group_cell = wt.group_first = wt.group_cnt = nullptr;
  wt.n_groups = ws->w_cnt.as<int32_t>() + 1; wt.work_counter = ws->w_cnt.as<int32_t>() + 2;
  wt.sp_cap = 0; wt.sp_cell = wt.sp_first = wt.sp_chunk = nullptr; wt.sp_counter = ws->w_cnt.as<int32_t>() + 3; wt.n_sparse = ws->w_cnt.as<int32_t>() + 4;
  // ONE launch: the table units (query x codebook, int16) and the record workgroups -- the entry records straight from the
  // queries' table scales: no item / work-table / record kernels (pq_front_kernel)
  PqFrontArgs fa;
  fa.queries = d_q; fa.Q = Q; fa.d = fx->d; fa.lists = lists; fa.W = W; fa.n_rows = fx->N; fa.blk_off = fx->blk_off; fa.list_off = fx->list_off;
  fa.cbT = fx->cbF; fa.cmax = fx->cmaxp; fa.pmax = fx->pmax; fa.qn = ws->w_qn.as<float>(); fa.qscale = ws->w_qn.as<float>() + (size_t)Q * m;
  fa.qc = ws->w_qc.as<uint32_t>(); fa.m = m; fa.K = K; fa.sentinel = sentinel; fa.item_cell = pa.item_cell; fa.item_query = pa.item_query;
  fa.item_dist = pa.item_dist; fa.round_rows = pa.round_rows; fa.records = ws->w_records.as<int32_t>(); fa.n_groups = wt.n_groups;
  const size_t front_lds = std::max<size_t>((size_t)query_codebook5_lds<25, 16>(), (size_t)(1024 + 16 + 2) * sizeof(float));
  timed_launch(fx, s, "pq_front", [&] {
    hipLaunchKernelGGL(pq_front_kernel, dim3((unsigned)(m * ((Q + 15) / 16) + Q)), dim3(256), front_lds, s, fa);
  });
  HIP_TRY(hipGetLastError());
  return ivf_scan_filter(r, pa, wt);
}

// ONE query over the flat table as one launch (one.h): table slices, grid barrier, scan, last-arriver merge.  `err` is a
// word of mapped host memory the kernel sets when one of its bounded polls ran out (the grid was not co-resident): the
// caller then re-arms the counters and takes the three-launch path.
static bool pq_one_shape(const freddy_gpu_index* ix, int Q, int k, int64_t n_blocks) {
  return ix->tune.one_launch && !ix->one_launch_failed && Q == 1 && ix->m == 12 && ix->S == 25 && ix->K <= 1024 && (ix->K & 3) == 0 && ix->d == 300 &&
         2 * k <= 64 && n_blocks >= 64 && (int64_t)ix->h_ids.size() == ix->N;
}
static int pq_one(freddy_gpu_index* ix, hipStream_t s, const float* h_q, int k, float sentinel, const int32_t* blk_off,
                  const uint32_t* packed, const int32_t* pos, int64_t n_blocks, int32_t* d_out_ids, float* d_out_dist, int32_t* err) {
  Workspace* ws = workspace_for(ix, s);
  const int K = ix->K, L = 2 * k;
  const size_t lutN = (size_t)12 * K;
  // (one workgroup per CU at most: all co-resident; the last arriver stages every list in LDS: G * L keys within 56 KB)
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)ix->n_cus, (int64_t)256, (n_blocks + ONE_WAVES - 1) / ONE_WAVES, (int64_t)(56 * 1024) / (8 * L)}));
  const int chunk_blocks = (int)((n_blocks + G - 1) / G);
  const size_t part_off = (lutN * sizeof(float) + 255) & ~(size_t)255;
  uint32_t epoch = 0;
  if (int rc = one_buffer(ws, s, (1ull << 60) | ((uint64_t)K << 32) | ((uint64_t)G << 16) | (uint64_t)L, part_off + sizeof(u64) * (size_t)G * L, &epoch)) return rc;
  static const bool one_prof = getenv("FREDDY_GPU_ONE_PROF") != nullptr;
  if (one_prof && ws->w_one.ensure(256)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  OneArgs a;
  memcpy(a.qv, h_q, sizeof(a.qv)); a.cbT = ix->cbT; a.lut_g = ws->w_oneb.as<float>(); a.blk_off = blk_off; a.packed = packed; a.pos = pos;
  a.pos_to_id = nullptr;   // (positions out: the caller maps them through its host copy of the ids -- no dependent gather at the kernel's end)
  a.part = reinterpret_cast<u64*>(ws->w_oneb.as<char>() + part_off); a.out_ids = d_out_ids; a.out_dist = d_out_dist;
  a.epoch = epoch; a.err = err;
  a.prof = one_prof ? ws->w_one.as<unsigned long long>() + 8 : nullptr;
  a.K = K; a.L = L; a.k = k; a.chunk_blocks = chunk_blocks; a.sentinel = sentinel;
  memcpy(&a.sentinel_bits, &sentinel, 4);
  const size_t lds = std::max(((lutN * 4 + 15) & ~(size_t)15) + (size_t)ONE_WAVES * 64 * sizeof(u64),
                              (size_t)ONE_WAVES * 64 * sizeof(u64) + (size_t)G * L * sizeof(u64));
  timed_launch(ix, s, "pq_one", [&] { hipLaunchKernelGGL((pq_one_kernel<25>), dim3((unsigned)G), dim3(ONE_WG), lds, s, a); });
  HIP_TRY(hipGetLastError());
  return 0;
}

static int pq_chunk(freddy_gpu_index* ix, hipStream_t s, const float* d_q, int Q, int k, float sentinel,
                    const int32_t* blk_off, const uint32_t* packed, const int32_t* pos, int64_t n_blocks,
                    int32_t* d_out_ids, float* d_out_dist) {
  Workspace* ws = workspace_for(ix, s);
  const int m = ix->m, K = ix->K;
  const int L = std::min(2 * k, 64 * 16);
  const size_t lutN = (size_t)m * K;
  // enough (query, chunk) workgroups to fill the chip, but chunks long enough to amortise
  // the 48 KiB LUT staging
  int chunk_blocks = 64;   // 4096 rows; longer chunks once there are enough (query, chunk) workgroups
  while ((n_blocks + chunk_blocks - 1) / chunk_blocks * (int64_t)Q > 4096 && chunk_blocks < 8192) chunk_blocks *= 2;
  const int nchunk = (int)std::max<int64_t>(1, (n_blocks + chunk_blocks - 1) / chunk_blocks);
  if (ws->w_lut.ensure(sizeof(float) * (size_t)Q * lutN) ||
      ws->w_part.ensure(sizeof(u64) * (size_t)Q * nchunk * SCAN_WAVES * L))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  if (int rc = launch_lut(ix, s, d_q, nullptr, ws->w_lut.as<float>(), Q)) return rc;
  ScanArgs sa;
  sa.lut = ws->w_lut.as<float>(); sa.item_list = nullptr; sa.item_query = nullptr;
  sa.blk_off = blk_off; sa.packed = packed; sa.pos = pos; sa.part = ws->w_part.as<u64>();
  sa.cand_count = nullptr;
  sa.m = m; sa.K = K; sa.chunk_blocks = chunk_blocks; sa.nchunk = nchunk; sa.L = L;
  memcpy(&sa.sentinel_bits, &sentinel, 4);
  if (int rc = launch_scan(ix, s, sa, Q)) return rc;
  MergeArgs ma;
  ma.part = sa.part; ma.active = nullptr; ma.pos_to_id = ix->ids; ma.round_rows = nullptr; ma.cand_count = nullptr;
  ma.out_ids = d_out_ids; ma.out_dist = d_out_dist; ma.found = nullptr; ma.next_active = nullptr; ma.n_next = nullptr;
  ma.status = nullptr;
  ma.n_active = Q; ma.parts_per_query = nchunk; ma.L = L; ma.k = k; ma.found_rule = 0; ma.first_round = 1;
  ma.sentinel = sentinel;
  return launch_merge(ix, s, ma);
}

extern "C" int freddy_gpu_pq_search_dev(freddy_gpu_index_t* ix, const float* d_queries, int32_t Q, int32_t k,
                                        float sentinel, int32_t* d_out_ids, float* d_out_dist, void* hip_stream) {
  if (int rc = check_search_args(ix, KIND_PQ, d_queries, Q, k, d_out_ids, d_out_dist)) return rc;
  if (Q == 0) return FREDDY_OK;
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ix->stream;
  if (pq_use_fused(ix, Q, k)) {
    if (int rc = pq_shadow_build(ix)) return rc;
    const int qf = pq_fused_queries_per_chunk(ix, ix->n_blocks);
    for (int q0 = 0; q0 < Q; q0 += qf) {
      const int n = std::min(qf, Q - q0);
      if (int rc = pq_fused_chunk(ix, ix->pq_shadow, s, d_queries + (size_t)q0 * ix->d, n, k, sentinel, d_out_ids + (size_t)q0 * k, d_out_dist + (size_t)q0 * k))
        return rc;
    }
    return FREDDY_OK;
  }
  const int qc = max_queries_per_chunk(ix, 1);
  for (int q0 = 0; q0 < Q; q0 += qc) {
    const int n = std::min(qc, Q - q0);
    if (int rc = pq_chunk(ix, s, d_queries + (size_t)q0 * ix->d, n, k, sentinel, ix->blk_off, ix->packed, ix->pos,
                          ix->n_blocks, d_out_ids + (size_t)q0 * k, d_out_dist + (size_t)q0 * k))
      return rc;
  }
  return FREDDY_OK;
}

// "WHERE id IN (...)" over the flat PQ table: unknown ids vanish, duplicates collapse, order = table
// order; the rows' packed codes are gathered into a temporary one-list table (synchronises the stream).
static int pq_subset(freddy_gpu_index* ix, hipStream_t s, const int32_t* subset_ids, int64_t n_subset, const int32_t** blk_off,
                     const uint32_t** packed, const int32_t** pos, int64_t* n_blocks, int64_t* n_rows_out = nullptr) {
  Workspace* ws = workspace_for(ix, s);
  std::vector<int32_t> rows;
  rows.reserve((size_t)n_subset);
  for (int64_t i = 0; i < n_subset; ++i) {
    auto it = std::lower_bound(ix->h_ids.begin(), ix->h_ids.end(), subset_ids[i]);
    if (it != ix->h_ids.end() && *it == subset_ids[i]) rows.push_back((int32_t)(it - ix->h_ids.begin()));
  }
  std::sort(rows.begin(), rows.end());
  rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
  const int n_rows = (int)rows.size();
  const int nb = (n_rows + 63) / 64;
  const int n_pad = nb * 64;
  const int32_t h_blk[2] = {0, nb};
  if (ws->w_sub_rows.ensure(sizeof(int32_t) * std::max(n_rows, 1)) ||
      ws->w_sub_packed.ensure(sizeof(uint32_t) * (size_t)std::max(nb, 1) * ix->M2 * 64) ||
      ws->w_sub_pos.ensure(sizeof(int32_t) * (size_t)std::max(n_pad, 1)) || ws->w_sub_blk.ensure(sizeof(int32_t) * 2))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  if (n_rows) HIP_TRY(hipMemcpyAsync(ws->w_sub_rows.p, rows.data(), sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ws->w_sub_blk.p, h_blk, sizeof(h_blk), hipMemcpyHostToDevice, s));
  if (n_pad) {
    timed_launch(ix, s, "gather_rows", [&] {
      hipLaunchKernelGGL(gather_rows_kernel, dim3((n_pad + WG - 1) / WG), dim3(WG), 0, s, ws->w_sub_rows.as<int32_t>(),
                         n_rows, ix->packed, ws->w_sub_packed.as<uint32_t>(), ws->w_sub_pos.as<int32_t>(), ix->M2, n_pad);
    });
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(s));  // rows / h_blk are stack/heap temporaries
  *blk_off = ws->w_sub_blk.as<int32_t>();
  *packed = ws->w_sub_packed.as<uint32_t>();
  *pos = ws->w_sub_pos.as<int32_t>();
  *n_blocks = nb;
  if (n_rows_out) *n_rows_out = n_rows;
  return 0;
}

extern "C" int freddy_gpu_pq_search(freddy_gpu_index_t* ix, const float* queries, int32_t Q, int32_t k, float sentinel,
                                    const int32_t* subset_ids, int64_t n_subset, int32_t* out_ids, float* out_dist) {
  if (int rc = check_search_args(ix, KIND_PQ, queries, Q, k, out_ids, out_dist)) return rc;
  if (n_subset < 0 || (n_subset > 0 && !subset_ids)) return fail(FREDDY_E_ARG, "bad subset");
  if (Q == 0) return FREDDY_OK;
  HIP_TRY(hipSetDevice(ix->device));
  Workspace* ws = workspace_for(ix, ix->stream);
  hipStream_t s = ix->stream;
  // queries in, lists out through pinned staging that the kernels read and write themselves (mapped host memory): a
  // handful of queries are read where they are staged and their lists written straight back; larger batches cross PCIe
  // once, by a copy kernel each way.  No hipMemcpyAsync in the stream (each one is an SDMA hop with its own latency).
  const size_t q_bytes = sizeof(float) * (size_t)Q * ix->d, n_out = (size_t)Q * k;
  auto pinned_fit = [](void** p, size_t* cap, size_t need) -> int {
    if (need <= *cap) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    if (hipHostMalloc(p, need + need / 4 + 256, hipHostMallocDefault) != hipSuccess) { *p = nullptr; return -1; }
    *cap = need + need / 4 + 256;
    return 0;
  };
  if (pinned_fit(&ix->hio_in, &ix->hio_in_cap, q_bytes + 16) || pinned_fit(&ix->hio_out, &ix->hio_out_cap, n_out * 8 + 16))
    return fail(FREDDY_E_NOMEM, "pinned staging allocation failed");
  memcpy(ix->hio_in, queries, q_bytes);
  const bool direct = Q <= 8;
  if (!direct && (ws->w_q.ensure(q_bytes + 16) || ws->w_out_ids.ensure(sizeof(int32_t) * n_out) || ws->w_out_dist.ensure(sizeof(float) * n_out)))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  const float* d_q = static_cast<const float*>(ix->hio_in);
  int32_t* d_oi = static_cast<int32_t*>(ix->hio_out);
  float* d_od = reinterpret_cast<float*>(d_oi + n_out);
  if (!direct) {
    const size_t n16 = (q_bytes + 15) / 16;
    hipLaunchKernelGGL(lane_copy_in_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 512)), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(ix->hio_in), ws->w_q.as<uint4>(), n16);
    HIP_TRY(hipGetLastError());
    d_q = ws->w_q.as<float>(); d_oi = ws->w_out_ids.as<int32_t>(); d_od = ws->w_out_dist.as<float>();
  }

  const int32_t* blk_off = ix->blk_off;
  const uint32_t* packed = ix->packed;
  const int32_t* pos = ix->pos;
  int64_t n_blocks = ix->n_blocks, n_rows = ix->N;
  if (subset_ids)
    if (int rc = pq_subset(ix, s, subset_ids, n_subset, &blk_off, &packed, &pos, &n_blocks, &n_rows)) return rc;
  // (a subset of at least one full pseudo-list: its gathered rows get a view of their own, refreshed on this stream)
  const bool fused_path = pq_use_fused(ix, Q, k) && (!subset_ids || n_blocks >= FUSED_UNIT_BLOCKS);
  freddy_gpu_index* view = nullptr;
  if (fused_path && !subset_ids) { if (int rc = pq_shadow_build(ix)) return rc; view = ix->pq_shadow; }
  if (fused_path && subset_ids) {
    if (int rc = pq_view_refresh(ix, &ix->pq_sub_view, s, packed, pos, n_blocks, n_rows)) return rc;
    view = ix->pq_sub_view;
  }
  const int qc = fused_path ? pq_fused_queries_per_chunk(ix, n_blocks) : max_queries_per_chunk(ix, 1);
  if (!fused_path && direct && pq_one_shape(ix, Q, k, n_blocks)) {
    int32_t* err = reinterpret_cast<int32_t*>(static_cast<char*>(ix->hio_out) + n_out * 8);   // (the staging block's spare 16 bytes)
    *err = 0;
    if (int rc = pq_one(ix, s, queries, k, sentinel, blk_off, packed, pos, n_blocks, d_oi, d_od, err)) return rc;
    // the kernel's last store is this word (2 = list written, 1 = a bounded poll ran out): polled here for up to a millisecond
    // -- a few microseconds sooner than the runtime's completion signal -- then the stream is waited for the usual way
    {
      volatile int32_t* flag = err;
      const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(1000);
      int spins = 0;
      while (*flag == 0) {
        __builtin_ia32_pause();
        if ((++spins & 255) == 0 && std::chrono::steady_clock::now() > t_end) break;
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (*flag != 2) HIP_TRY(hipStreamSynchronize(s));
    }
    if (getenv("FREDDY_GPU_ONE_PROF")) {
      HIP_TRY(hipStreamSynchronize(s));
      unsigned long long st[16];
      (void)hipMemcpy(st, ws->w_one.as<unsigned long long>() + 8, sizeof(st), hipMemcpyDeviceToHost);
      fprintf(stderr, "[pq_one] wg0: slice %.2f barrier %.2f stage %.2f scan %.2f publish %.2f | last: since wg0 start %.2f merge %.2f replay %.2f us\n",
              (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01,
              (st[8] - st[0]) * 0.01, (st[9] - st[8]) * 0.01, (st[10] - st[9]) * 0.01);
    }
    if (*err == 2) {
      const int32_t* h_pos = static_cast<const int32_t*>(ix->hio_out);
      for (size_t i = 0; i < n_out; ++i) out_ids[i] = h_pos[i] >= 0 ? ix->h_ids[(size_t)h_pos[i]] : -1;
      memcpy(out_dist, h_pos + n_out, n_out * 4);
      return FREDDY_OK;
    }
    // the grid never met at its barrier (not co-resident): counters re-armed, this handle keeps to the three-launch path
    ix->one_launch_failed = true;
    ws->one_shape = 0;
  }
  for (int q0 = 0; q0 < Q; q0 += qc) {
    const int n = std::min(qc, Q - q0);
    if (fused_path) {
      if (int rc = pq_fused_chunk(ix, view, s, d_q + (size_t)q0 * ix->d, n, k, sentinel, d_oi + (size_t)q0 * k, d_od + (size_t)q0 * k))
        return rc;
      continue;
    }
    if (int rc = pq_chunk(ix, s, d_q + (size_t)q0 * ix->d, n, k, sentinel, blk_off, packed, pos, n_blocks, d_oi + (size_t)q0 * k,
                          d_od + (size_t)q0 * k))
      return rc;
  }
  if (!direct) {
    hipLaunchKernelGGL(host_io_out_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, d_oi, d_od, static_cast<int32_t*>(ix->hio_out), (int)n_out);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(out_ids, ix->hio_out, n_out * 4);
  memcpy(out_dist, static_cast<const int32_t*>(ix->hio_out) + n_out, n_out * 4);
  return FREDDY_OK;
}

// ---------------------------------------------------------------------------------------
// grouping_pq
// ---------------------------------------------------------------------------------------
extern "C" int freddy_gpu_grouping_pq(freddy_gpu_index_t* ix, const float* group_vectors, int32_t G, const int32_t* subset_ids,
                                      int64_t n_subset, int32_t* out_ids, int32_t* out_group, int64_t* n_out) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (ix->kind != KIND_PQ) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  if (G <= 0 || !group_vectors || !out_ids || !out_group || !n_out) return fail(FREDDY_E_ARG, "bad argument");
  if (n_subset < 0 || (n_subset > 0 && !subset_ids)) return fail(FREDDY_E_ARG, "bad subset");
  *n_out = 0;
  HIP_TRY(hipSetDevice(ix->device));
  Workspace* ws = workspace_for(ix, ix->stream);
  hipStream_t s = ix->stream;
  const int m = ix->m, K = ix->K, d = ix->d;
  const size_t lutN = (size_t)m * K;
  if (lutN * sizeof(float) > 160 * 1024) return fail(FREDDY_E_LIMIT, "m*K=%zu LUT entries exceed the 160 KiB of LDS", lutN);
  const int32_t* blk_off = ix->blk_off;
  const uint32_t* packed = ix->packed;
  const int32_t* pos = ix->pos;
  int64_t n_blocks = ix->n_blocks;
  if (subset_ids)
    if (int rc = pq_subset(ix, s, subset_ids, n_subset, &blk_off, &packed, &pos, &n_blocks)) return rc;
  (void)blk_off;
  if (n_blocks == 0) return FREDDY_OK;
  if (ws->w_q.ensure(sizeof(float) * (size_t)G * d) || ws->w_lut.ensure(sizeof(float) * (size_t)G * lutN) ||
      ws->w_out_ids.ensure(sizeof(int32_t) * (size_t)n_blocks * 64))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  HIP_TRY(hipMemcpyAsync(ws->w_q.p, group_vectors, sizeof(float) * (size_t)G * d, hipMemcpyHostToDevice, s));
  if (int rc = launch_lut(ix, s, ws->w_q.as<float>(), nullptr, ws->w_lut.as<float>(), G)) return rc;   // freddy.c:1288-1299
  const dim3 grid((unsigned)((n_blocks + GROUP_BLOCKS - 1) / GROUP_BLOCKS));
  timed_launch(ix, s, "grouping", [&] {
    if (ix->M2 == 6)
      hipLaunchKernelGGL((grouping_kernel<6>), grid, dim3(WG), lutN * sizeof(float), s, ws->w_lut.as<float>(), G, m, K, packed, (int)n_blocks, ws->w_out_ids.as<int32_t>());
    else if (ix->M2 == 15)
      hipLaunchKernelGGL((grouping_kernel<15>), grid, dim3(WG), lutN * sizeof(float), s, ws->w_lut.as<float>(), G, m, K, packed, (int)n_blocks, ws->w_out_ids.as<int32_t>());
    else
      hipLaunchKernelGGL((grouping_kernel<0>), grid, dim3(WG), lutN * sizeof(float), s, ws->w_lut.as<float>(), G, m, K, packed, (int)n_blocks, ws->w_out_ids.as<int32_t>());
  });
  HIP_TRY(hipGetLastError());
  std::vector<int32_t> h_grp((size_t)n_blocks * 64), h_pos((size_t)n_blocks * 64);
  HIP_TRY(hipMemcpyAsync(h_grp.data(), ws->w_out_ids.p, sizeof(int32_t) * h_grp.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_pos.data(), pos, sizeof(int32_t) * h_pos.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  int64_t n = 0;
  for (size_t i = 0; i < h_pos.size(); ++i)
    if (h_pos[i] >= 0) { out_ids[n] = ix->h_ids[(size_t)h_pos[i]]; out_group[n] = h_grp[i]; ++n; }
  *n_out = n;
  return FREDDY_OK;
}

// ---------------------------------------------------------------------------------------
// index build: encoding
// ---------------------------------------------------------------------------------------
// limit_coarse / limit_code < +inf: insert_batch's searches start from that distance (strict "<"); *n_too_far
// counts the (vector[, position]) pairs with no centroid nearer than the limit.
static int encode_impl(const freddy_encode_desc* t, int device, const float* vectors, int64_t N, int32_t* out_cell,
                       int16_t* out_codes, float limit_coarse, float limit_code, int32_t* n_too_far) {
  if (!t || !t->codebook || !out_codes || N < 0 || (N > 0 && !vectors)) return fail(FREDDY_E_ARG, "NULL argument");
  if (t->d <= 0 || t->m <= 0 || t->K <= 0 || t->d % t->m) return fail(FREDDY_E_ARG, "bad shape d=%d m=%d K=%d", t->d, t->m, t->K);
  if (t->K > 32767) return fail(FREDDY_E_LIMIT, "K=%d does not fit an int16 code", t->K);
  if ((t->C > 0) != (t->coarse != nullptr)) return fail(FREDDY_E_ARG, "coarse and C must be given together");
  if (t->C > 0 && !out_cell) return fail(FREDDY_E_ARG, "out_cell is required with a coarse quantizer");
  if (N == 0) return FREDDY_OK;
  HIP_TRY(hipSetDevice(device));
  const int d = t->d, m = t->m, K = t->K, S = d / m, C = t->C;
  const int Cpad = ((C + 63) / 64) * 64;
  std::vector<float> cbT((size_t)m * S * K);
  for (int p = 0; p < m; ++p)
    for (int c = 0; c < K; ++c)
      for (int i = 0; i < S; ++i) cbT[((size_t)p * S + i) * K + c] = t->codebook[((size_t)p * K + c) * S + i];
  std::vector<float> cT;
  if (C) {
    cT.assign((size_t)d * Cpad, 0.0f);
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < d; ++i) cT[(size_t)i * Cpad + c] = t->coarse[(size_t)c * d + i];
  }
  const int64_t chunk = std::min<int64_t>(N, 1 << 16);
  float *d_cbT = nullptr, *d_cT = nullptr, *d_coarse = nullptr, *d_vec = nullptr, *d_res = nullptr;
  int32_t *d_cell = nullptr, *d_far = nullptr;
  int16_t* d_codes = nullptr;
  int rc = FREDDY_OK;
  hipStream_t s = nullptr;
  auto cleanup = [&] {
    void* ptrs[] = {d_cbT, d_cT, d_coarse, d_vec, d_res, d_cell, d_codes, d_far};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (s) (void)hipStreamDestroy(s);
  };
#define ENC_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) { cleanup(); return fail(FREDDY_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } \
  } while (0)
  ENC_TRY(hipStreamCreate(&s));
  ENC_TRY(hipMalloc((void**)&d_cbT, sizeof(float) * cbT.size()));
  ENC_TRY(hipMalloc((void**)&d_vec, sizeof(float) * (size_t)chunk * d));
  ENC_TRY(hipMalloc((void**)&d_codes, sizeof(int16_t) * (size_t)chunk * m));
  ENC_TRY(hipMemcpyAsync(d_cbT, cbT.data(), sizeof(float) * cbT.size(), hipMemcpyHostToDevice, s));
  if (n_too_far) {
    ENC_TRY(hipMalloc((void**)&d_far, sizeof(int32_t)));
    ENC_TRY(hipMemsetAsync(d_far, 0, sizeof(int32_t), s));
  }
  if (C) {
    ENC_TRY(hipMalloc((void**)&d_cT, sizeof(float) * cT.size()));
    ENC_TRY(hipMalloc((void**)&d_coarse, sizeof(float) * (size_t)C * d));
    ENC_TRY(hipMalloc((void**)&d_res, sizeof(float) * (size_t)chunk * d));
    ENC_TRY(hipMalloc((void**)&d_cell, sizeof(int32_t) * (size_t)chunk));
    ENC_TRY(hipMemcpyAsync(d_cT, cT.data(), sizeof(float) * cT.size(), hipMemcpyHostToDevice, s));
    ENC_TRY(hipMemcpyAsync(d_coarse, t->coarse, sizeof(float) * (size_t)C * d, hipMemcpyHostToDevice, s));
  }
  for (int64_t i0 = 0; i0 < N; i0 += chunk) {
    const int n = (int)std::min<int64_t>(chunk, N - i0);
    ENC_TRY(hipMemcpyAsync(d_vec, vectors + (size_t)i0 * d, sizeof(float) * (size_t)n * d, hipMemcpyHostToDevice, s));
    const float* src = d_vec;
    if (C) {
      hipLaunchKernelGGL(assign_coarse_kernel, dim3((unsigned)n), dim3(64), 0, s, (const float*)d_vec, (const float*)d_cT, d_cell, n, C, Cpad, d, limit_coarse, d_far);
      hipLaunchKernelGGL(residual_kernel, dim3((unsigned)n), dim3(WG), 0, s, (const float*)d_vec, (const float*)d_coarse,
                         (const int32_t*)d_cell, (const int32_t*)nullptr, d_res, d, S, S);
      src = d_res;
    }
    const int ipw = 64;
    const dim3 grid((unsigned)m, (unsigned)((n + ipw - 1) / ipw));
    if (S == 25) hipLaunchKernelGGL((encode_pq_kernel<25, 4>), grid, dim3(WG), 0, s, src, (const float*)d_cbT, d_codes, n, ipw, m, K, d, S, limit_code, d_far);
    else if (S == 10) hipLaunchKernelGGL((encode_pq_kernel<10, 4>), grid, dim3(WG), 0, s, src, (const float*)d_cbT, d_codes, n, ipw, m, K, d, S, limit_code, d_far);
    else hipLaunchKernelGGL((encode_pq_kernel<0, 4>), grid, dim3(WG), 0, s, src, (const float*)d_cbT, d_codes, n, ipw, m, K, d, S, limit_code, d_far);
    ENC_TRY(hipGetLastError());
    ENC_TRY(hipMemcpyAsync(out_codes + (size_t)i0 * m, d_codes, sizeof(int16_t) * (size_t)n * m, hipMemcpyDeviceToHost, s));
    if (C) ENC_TRY(hipMemcpyAsync(out_cell + i0, d_cell, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    ENC_TRY(hipStreamSynchronize(s));
  }
  if (n_too_far) ENC_TRY(hipMemcpy(n_too_far, d_far, sizeof(int32_t), hipMemcpyDeviceToHost));
#undef ENC_TRY
  cleanup();
  return rc;
}

extern "C" int freddy_gpu_encode(const freddy_encode_desc* t, int device, const float* vectors, int64_t N, int32_t* out_cell,
                                 int16_t* out_codes) {
  const float inf = std::numeric_limits<float>::infinity();
  return encode_impl(t, device, vectors, N, out_cell, out_codes, inf, inf, nullptr);
}

// insert_batch, quantisation of the new vectors (freddy.c:1557-1623): codes against the PQ codebook, coarse
// cell (from minDistCoarse = 100) + codes of the residual against the residual codebook, codes against the ivpq
// codebook, and the two coarse codes of the multi index (from MAX_DIST = 1000).  Every search is the exact
// 1-NN by squareDistance with the first entry winning ties, as updateCodebook's strict "<" scan.
extern "C" int freddy_gpu_insert_quantize(const freddy_insert_desc* t, int device, const float* vectors, int64_t n,
                                          int16_t* pq_codes, int32_t* coarse_id, int16_t* residual_codes, int16_t* ivpq_codes,
                                          int16_t* coarse_multi_codes) {
  if (!t || n < 0 || (n > 0 && !vectors)) return fail(FREDDY_E_ARG, "NULL argument");
  const float inf = std::numeric_limits<float>::infinity();
  int32_t far = 0, far_total = 0;
  if (t->pq_codebook) {
    if (!pq_codes) return fail(FREDDY_E_ARG, "pq_codes is required with a PQ codebook");
    freddy_encode_desc e = {t->d, t->pq_m, t->pq_K, t->pq_codebook, 0, nullptr};
    if (int rc = encode_impl(&e, device, vectors, n, nullptr, pq_codes, inf, 100.0f, &far)) return rc;
    far_total += far;
  }
  if (t->residual_codebook) {
    if (!t->coarse || !coarse_id || !residual_codes) return fail(FREDDY_E_ARG, "the residual codebook needs the coarse quantizer and both outputs");
    freddy_encode_desc e = {t->d, t->res_m, t->res_K, t->residual_codebook, t->C, t->coarse};
    if (int rc = encode_impl(&e, device, vectors, n, coarse_id, residual_codes, 100.0f, 100.0f, &far)) return rc;
    far_total += far;
  }
  if (t->ivpq_codebook) {
    if (!ivpq_codes) return fail(FREDDY_E_ARG, "ivpq_codes is required with an ivpq codebook");
    freddy_encode_desc e = {t->d, t->ivpq_m, t->ivpq_K, t->ivpq_codebook, 0, nullptr};
    if (int rc = encode_impl(&e, device, vectors, n, nullptr, ivpq_codes, inf, 100.0f, &far)) return rc;
    far_total += far;
  }
  if (t->coarse_multi) {
    if (!coarse_multi_codes) return fail(FREDDY_E_ARG, "coarse_multi_codes is required with a multi-index coarse quantizer");
    freddy_encode_desc e = {t->d, t->multi_positions, t->multi_codes, t->coarse_multi, 0, nullptr};
    if (int rc = encode_impl(&e, device, vectors, n, nullptr, coarse_multi_codes, inf, inf, nullptr)) return rc;
  }
  if (far_total)
    return fail(FREDDY_E_ARG, "%d (vector, position) pairs are 100 or farther from every centroid: insert_batch is undefined for them "
                "(index_utils.c:925-939 leaves the code uninitialised)", far_total);
  return FREDDY_OK;
}

// ---------------------------------------------------------------------------------------
// index build: quantizer training
// ---------------------------------------------------------------------------------------
extern "C" int freddy_gpu_kmeans(int device, const float* vectors, int64_t n, int32_t d, int32_t k, int32_t iters,
                                 const int32_t* init_rows, float* centroids, int32_t* assign_out) {
  if (!vectors || !centroids || n <= 0 || d <= 0 || k <= 0 || iters < 0) return fail(FREDDY_E_ARG, "bad argument");
  if (d > 1024) return fail(FREDDY_E_LIMIT, "d=%d exceeds this build's limit of 1024 dimensions", d);
  if (n > INT32_MAX) return fail(FREDDY_E_LIMIT, "too many training vectors");
  HIP_TRY(hipSetDevice(device));
  const int kpad = (k + 63) / 64 * 64;
  std::vector<float> init((size_t)k * d);
  for (int c = 0; c < k; ++c) {
    const int64_t r = init_rows ? init_rows[c] : c % n;
    if (r < 0 || r >= n) return fail(FREDDY_E_ARG, "init_rows[%d] = %lld is not a training row", c, (long long)r);
    memcpy(&init[(size_t)c * d], vectors + (size_t)r * d, sizeof(float) * (size_t)d);
  }
  float *d_vec = nullptr, *d_cent = nullptr, *d_centT = nullptr;
  int32_t* d_assign = nullptr;
  hipStream_t s = nullptr;
  int rc = FREDDY_OK;
  auto cleanup = [&] {
    void* ptrs[] = {d_vec, d_cent, d_centT, d_assign};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (s) (void)hipStreamDestroy(s);
  };
#define KM_TRY(expr)                                                                           \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) { cleanup(); return fail(FREDDY_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } \
  } while (0)
  KM_TRY(hipStreamCreate(&s));
  KM_TRY(hipMalloc((void**)&d_vec, sizeof(float) * (size_t)n * d));
  KM_TRY(hipMalloc((void**)&d_cent, sizeof(float) * (size_t)k * d));
  KM_TRY(hipMalloc((void**)&d_centT, sizeof(float) * (size_t)kpad * d));
  KM_TRY(hipMalloc((void**)&d_assign, sizeof(int32_t) * (size_t)n));
  KM_TRY(hipMemcpyAsync(d_vec, vectors, sizeof(float) * (size_t)n * d, hipMemcpyHostToDevice, s));
  KM_TRY(hipMemcpyAsync(d_cent, init.data(), sizeof(float) * init.size(), hipMemcpyHostToDevice, s));
  const float inf = std::numeric_limits<float>::infinity();
  for (int it = 0; it <= iters; ++it) {
    hipLaunchKernelGGL(kmeans_transpose_kernel, dim3((unsigned)(((size_t)d * kpad + 255) / 256)), dim3(256), 0, s, (const float*)d_cent, d_centT, k, kpad, d);
    hipLaunchKernelGGL(assign_coarse_kernel, dim3((unsigned)n), dim3(64), 0, s, (const float*)d_vec, (const float*)d_centT, d_assign, (int)n, k, kpad, d,
                       inf, (int32_t*)nullptr);
    if (it == iters) break;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)k), dim3(256), 0, s, (const float*)d_vec, (const int32_t*)d_assign, n, d, d_cent);
    KM_TRY(hipGetLastError());
  }
  KM_TRY(hipGetLastError());
  KM_TRY(hipMemcpyAsync(centroids, d_cent, sizeof(float) * (size_t)k * d, hipMemcpyDeviceToHost, s));
  if (assign_out) KM_TRY(hipMemcpyAsync(assign_out, d_assign, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
  KM_TRY(hipStreamSynchronize(s));
#undef KM_TRY
  cleanup();
  return rc;
}

// ---------------------------------------------------------------------------------------
// insert_batch: HBM index mutation
// ---------------------------------------------------------------------------------------
// new block j of list blk_cell[b] <- old block j of that list (or empty)
__global__ __launch_bounds__(256) void repack_blocks_kernel(const uint32_t* __restrict__ old_packed, const int32_t* __restrict__ old_pos,
                                                           const int32_t* __restrict__ old_blk_off, const int32_t* __restrict__ new_blk_off,
                                                           const int32_t* __restrict__ new_blk_cell, uint32_t* __restrict__ packed,
                                                           int32_t* __restrict__ pos, int64_t n_new_blocks, int M2) {
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= n_new_blocks) return;
  const int c = new_blk_cell[b];
  const int j = (int)(b - new_blk_off[c]);
  const bool have = j < old_blk_off[c + 1] - old_blk_off[c];
  const int64_t ob = (int64_t)old_blk_off[c] + j;
  for (int w = 0; w < M2; ++w) packed[((size_t)b * M2 + w) * 64 + lane] = have ? old_packed[((size_t)ob * M2 + w) * 64 + lane] : 0u;
  pos[(size_t)b * 64 + lane] = have ? old_pos[(size_t)ob * 64 + lane] : -1;
}
// new rows into their slots: slot[i] = row slot (block * 64 + lane) of new row i
__global__ __launch_bounds__(256) void place_rows_kernel(const int64_t* __restrict__ slot, const int32_t* __restrict__ row_pos,
                                                        const int16_t* __restrict__ codes, int64_t n, uint32_t* __restrict__ packed,
                                                        int32_t* __restrict__ pos, int m, int M2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t sl = slot[i], b = sl >> 6;
  const int lane = (int)(sl & 63);
  for (int w = 0; w < M2; ++w) {
    const uint32_t lo = (uint16_t)codes[(size_t)i * m + 2 * w];
    const uint32_t hi = (2 * w + 1 < m) ? (uint16_t)codes[(size_t)i * m + 2 * w + 1] : 0u;
    packed[((size_t)b * M2 + w) * 64 + lane] = lo | (hi << 16);
  }
  pos[(size_t)sl] = row_pos[i];
}
// raw vectors into the 64-row blocked layout: row r -> xb[r / 64][dim][r % 64]
__global__ __launch_bounds__(256) void place_vectors_kernel(const float* __restrict__ src, int64_t first_row, int64_t n, float* __restrict__ xb, int d) {
  const int64_t i = (int64_t)blockIdx.x;
  if (i >= n) return;
  const int64_t r = first_row + i;
  for (int dim = threadIdx.x; dim < d; dim += 256) xb[((r >> 6) * d + dim) * 64 + (r & 63)] = src[(size_t)i * d + dim];
}

template <class T>
static int grow_device_array(T** arr, size_t old_n, size_t new_n, const T* append_host, size_t append_n) {
  T* fresh = nullptr;
  if (hipMalloc((void**)&fresh, sizeof(T) * std::max<size_t>(new_n, 1)) != hipSuccess) return -1;
  if (old_n && hipMemcpy(fresh, *arr, sizeof(T) * old_n, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(fresh); return -2; }
  if (append_n && hipMemcpy(fresh + old_n, append_host, sizeof(T) * append_n, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(fresh); return -2; }
  if (*arr) (void)hipFree(*arr);
  *arr = fresh;
  return 0;
}

// rows of a pq / ivf index: each new row goes to the end of its list; the 64-row block layout is rebuilt on
// the device (old blocks copied to their new places, new rows written into the free slots behind them)
static int append_packed_rows(freddy_gpu_index* ix, int n_lists, int64_t n, const int32_t* cell, const int32_t* row_pos, const int16_t* codes) {
  const int m = ix->m, M2 = ix->M2;
  std::vector<int32_t> new_list_off((size_t)n_lists + 1, 0), add((size_t)n_lists, 0);
  for (int64_t i = 0; i < n; ++i) {
    const int c = cell ? cell[i] : 0;
    if (c < 0 || c >= n_lists) return fail(FREDDY_E_ARG, "coarse_id %d of new row %lld is outside [0, %d)", c, (long long)i, n_lists);
    for (int l = 0; l < m; ++l)
      if (codes[(size_t)i * m + l] < 0 || codes[(size_t)i * m + l] >= ix->K)
        return fail(FREDDY_E_ARG, "code %d of new row %lld is outside [0, %d)", (int)codes[(size_t)i * m + l], (long long)i, ix->K);
    add[(size_t)c]++;
  }
  std::vector<int32_t> old_blk((size_t)n_lists + 1, 0), new_blk((size_t)n_lists + 1, 0);
  int max_blocks = 0;
  for (int c = 0; c < n_lists; ++c) {
    const int64_t old_len = ix->h_list_off[(size_t)c + 1] - ix->h_list_off[(size_t)c];
    old_blk[(size_t)c + 1] = old_blk[(size_t)c] + (int32_t)((old_len + 63) / 64);
    const int64_t len = old_len + add[(size_t)c];
    if ((int64_t)new_list_off[(size_t)c] + len > INT32_MAX - 64) return fail(FREDDY_E_LIMIT, "N too large for 32-bit row positions");
    new_list_off[(size_t)c + 1] = new_list_off[(size_t)c] + (int32_t)len;
    const int nb = (int)((len + 63) / 64);
    new_blk[(size_t)c + 1] = new_blk[(size_t)c] + nb;
    max_blocks = std::max(max_blocks, nb);
  }
  const int64_t n_new_blocks = new_blk[(size_t)n_lists];
  std::vector<int32_t> blk_cell((size_t)std::max<int64_t>(n_new_blocks, 1), 0);
  for (int c = 0; c < n_lists; ++c)
    for (int b = new_blk[(size_t)c]; b < new_blk[(size_t)c + 1]; ++b) blk_cell[(size_t)b] = c;
  std::vector<int64_t> slot((size_t)n);
  std::vector<int32_t> cursor((size_t)n_lists, 0);
  for (int64_t i = 0; i < n; ++i) {
    const int c = cell ? cell[i] : 0;
    const int64_t old_len = ix->h_list_off[(size_t)c + 1] - ix->h_list_off[(size_t)c];
    slot[(size_t)i] = (int64_t)new_blk[(size_t)c] * 64 + old_len + cursor[(size_t)c]++;
  }
  uint32_t* packed = nullptr;
  int32_t *pos = nullptr, *d_blk_cell = nullptr, *d_new_blk = nullptr, *d_list_off = nullptr, *d_row_pos = nullptr;
  int64_t* d_slot = nullptr;
  int16_t* d_codes = nullptr;
  int64_t junk = 0;
  int rc = 0;
  if (hipMalloc((void**)&packed, sizeof(uint32_t) * (size_t)std::max<int64_t>(n_new_blocks, 1) * M2 * 64) != hipSuccess ||
      hipMalloc((void**)&pos, sizeof(int32_t) * (size_t)std::max<int64_t>(n_new_blocks, 1) * 64) != hipSuccess ||
      upload(&d_blk_cell, blk_cell.data(), blk_cell.size(), &junk) || upload(&d_new_blk, new_blk.data(), new_blk.size(), &junk) ||
      upload(&d_list_off, new_list_off.data(), new_list_off.size(), &junk) || upload(&d_slot, slot.data(), slot.size(), &junk) ||
      upload(&d_row_pos, row_pos, (size_t)n, &junk) || upload(&d_codes, codes, (size_t)n * m, &junk))
    rc = fail(FREDDY_E_NOMEM, "device allocation failed while appending rows");
  if (!rc && n_new_blocks > 0) {
    hipLaunchKernelGGL(repack_blocks_kernel, dim3((unsigned)((n_new_blocks + 3) / 4)), dim3(256), 0, ix->stream, ix->packed, ix->pos, ix->blk_off,
                       d_new_blk, d_blk_cell, packed, pos, n_new_blocks, M2);
    if (n > 0)
      hipLaunchKernelGGL(place_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ix->stream, d_slot, d_row_pos, d_codes, n, packed, pos, m, M2);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ix->stream) != hipSuccess) rc = fail(FREDDY_E_HIP, "re-blocking the lists failed");
  }
  void* tmp[] = {d_slot, d_row_pos, d_codes};
  for (void* p : tmp) if (p) (void)hipFree(p);
  if (rc) {
    void* fresh[] = {packed, pos, d_blk_cell, d_new_blk, d_list_off};
    for (void* p : fresh) if (p) (void)hipFree(p);
    return rc;
  }
  void* old[] = {ix->packed, ix->pos, ix->blk_cell, ix->blk_off, ix->list_off};
  for (void* p : old) if (p) (void)hipFree(p);
  ix->packed = packed; ix->pos = pos; ix->blk_cell = d_blk_cell; ix->blk_off = d_new_blk; ix->list_off = d_list_off;
  ix->n_blocks = n_new_blocks;
  ix->max_list_blocks = max_blocks;
  ix->h_list_off = new_list_off;
  ix->N += n;
  if (!ix->shadow_of) { if (int rc = build_packed8(ix)) return rc; }
  return 0;
}

static int exf_table_stats(freddy_gpu_index* ix, int64_t r0, int64_t n);
extern "C" int freddy_gpu_append_rows(freddy_gpu_index_t* ix, int64_t n, const int32_t* ids, const int32_t* coarse_id,
                                      const int16_t* codes, const float* vectors) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (n < 0 || (n > 0 && !ids)) return fail(FREDDY_E_ARG, "bad argument");
  if (n == 0) return FREDDY_OK;
  if (!ix->replicas.empty()) {
    // every replica holds the same tables: the primary goes first (argument errors are found there before anything has
    // changed anywhere); a failure after that leaves the devices with different tables -> the handle is poisoned and
    // every search on it fails loudly until it is unpinned
    std::vector<freddy_gpu_index*> reps;
    reps.swap(ix->replicas);
    int rc = freddy_gpu_append_rows(ix, n, ids, coarse_id, codes, vectors);
    reps.swap(ix->replicas);
    if (rc) { if (rc == FREDDY_E_HIP || rc == FREDDY_E_NOMEM) ix->poisoned = true; return rc; }
    for (freddy_gpu_index* r : ix->replicas)
      if ((rc = freddy_gpu_append_rows(r, n, ids, coarse_id, codes, vectors))) { ix->poisoned = true; return rc; }
    return FREDDY_OK;
  }
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  if (ix->kind == KIND_IVPQ) ix->join.tl_valid = false;   // (the cached "id IN (targets)" resolution refers to the rows as they were)
  const int32_t last_id = ix->kind == KIND_IVPQ ? (ix->join.h_ids.empty() ? -1 : ix->join.h_ids.back())
                          : ix->kind == KIND_IVF ? ix->max_id : (ix->h_ids.empty() ? -1 : ix->h_ids.back());
  for (int64_t i = 0; i < n; ++i)
    if (ids[i] <= (i ? ids[i - 1] : last_id))
      return fail(FREDDY_E_ARG, "appended ids must ascend beyond the largest pinned id %d (row %lld has %d)", last_id, (long long)i, ids[i]);
  if (ix->N + n > (int64_t)INT32_MAX - 64) return fail(FREDDY_E_LIMIT, "N too large for 32-bit row positions");
  switch (ix->kind) {
    case KIND_PQ: {
      if (!codes) return fail(FREDDY_E_ARG, "codes are required");
      std::vector<int32_t> row_pos((size_t)n);
      for (int64_t i = 0; i < n; ++i) row_pos[(size_t)i] = (int32_t)(ix->N + i);   // flat table: position = row index
      const int64_t old_n = ix->N;
      if (ix->pq_shadow) { free_index(ix->pq_shadow); ix->pq_shadow = nullptr; }   // (rebuilt by the next batch search)
      if (int rc = append_packed_rows(ix, 1, n, nullptr, row_pos.data(), codes)) return rc;
      if (grow_device_array(&ix->ids, (size_t)old_n, (size_t)(old_n + n), ids, (size_t)n)) return fail(FREDDY_E_NOMEM, "device allocation failed");
      ix->h_ids.insert(ix->h_ids.end(), ids, ids + n);
      ix->max_id = ids[n - 1];
      return FREDDY_OK;
    }
    case KIND_IVF: {
      if (!codes || !coarse_id) return fail(FREDDY_E_ARG, "coarse_id and codes are required");
      if (int rc = append_packed_rows(ix, ix->C, n, coarse_id, ids, codes)) return rc;
      ix->max_id = ids[n - 1];
      return refresh_row_terms(ix);
    }
    case KIND_IVPQ: {
      JoinIndex& j = ix->join;
      if (!codes || !coarse_id || (j.has_vectors && !vectors)) return fail(FREDDY_E_ARG, "coarse_id, codes (and vectors, if pinned) are required");
      for (int64_t i = 0; i < n; ++i) {
        if (coarse_id[i] < 0 || coarse_id[i] >= j.cells) return fail(FREDDY_E_ARG, "coarse_id %d out of range", coarse_id[i]);
        for (int l = 0; l < j.m; ++l)
          if (codes[(size_t)i * j.m + l] < 0 || codes[(size_t)i * j.m + l] >= j.K) return fail(FREDDY_E_ARG, "code out of range at new row %lld", (long long)i);
      }
      const size_t o = (size_t)j.N, nn = (size_t)(j.N + n);
      if (grow_device_array(&j.ids, o, nn, ids, (size_t)n) || grow_device_array(&j.cell, o, nn, coarse_id, (size_t)n) ||
          grow_device_array(&j.codes, o * j.MP, nn * j.MP, join_pad_codes(codes, n, j.m, j.MP).data(), (size_t)n * j.MP) ||
          (j.has_vectors && grow_device_array(&j.vectors, o * j.d, nn * j.d, vectors, (size_t)n * j.d)))
        return fail(FREDDY_E_NOMEM, "device allocation failed");
      if (j.markbits) (void)hipFree(j.markbits);
      j.markbits = nullptr;
      HIP_TRY(hipMalloc((void**)&j.markbits, sizeof(uint32_t) * ((nn + 31) / 32 + 1)));
      j.h_ids.insert(j.h_ids.end(), ids, ids + n);
      j.h_cell.insert(j.h_cell.end(), coarse_id, coarse_id + n);
      j.N += n; ix->N = j.N;
      j.ids_affine = (int64_t)j.h_ids.back() - j.h_ids.front() == j.N - 1;
      return FREDDY_OK;
    }
    case KIND_VEC: {
      if (!vectors) return fail(FREDDY_E_ARG, "vectors are required");
      const int d = ix->d;
      const size_t o = (size_t)ix->N, nn = (size_t)(ix->N + n);
      const int64_t new_blocks = (int64_t)((nn + 63) / 64);
      float* xb = nullptr;
      HIP_TRY(hipMalloc((void**)&xb, sizeof(float) * (size_t)new_blocks * d * 64));
      HIP_TRY(hipMemset(xb, 0, sizeof(float) * (size_t)new_blocks * d * 64));
      if (ix->n_blocks) HIP_TRY(hipMemcpy(xb, ix->xb, sizeof(float) * (size_t)ix->n_blocks * d * 64, hipMemcpyDeviceToDevice));
      if (grow_device_array(&ix->coarse, o * d, nn * d, vectors, (size_t)n * d) || grow_device_array(&ix->ids, o, nn, ids, (size_t)n)) {
        (void)hipFree(xb);
        return fail(FREDDY_E_NOMEM, "device allocation failed");
      }
      hipLaunchKernelGGL(place_vectors_kernel, dim3((unsigned)n), dim3(256), 0, ix->stream, ix->coarse + o * d, (int64_t)o, n, xb, d);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(ix->stream));
      if (ix->xb) (void)hipFree(ix->xb);
      ix->xb = xb; ix->n_blocks = new_blocks; ix->N += n;
      ix->h_ids.insert(ix->h_ids.end(), ids, ids + n);
      return exf_table_stats(ix, (int64_t)o, n);   // (the filter's scale and norm bound cover the new rows)
    }
  }
  return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
}

extern "C" int freddy_gpu_update_codebook(freddy_gpu_index_t* ix, const float* codebook) {
  if (!ix || !codebook) return fail(FREDDY_E_ARG, "NULL argument");
  if (!ix->replicas.empty()) {   // every device or none: a failure after the first device has changed poisons the handle
    size_t done = 0;
    int rc = 0;
    for (freddy_gpu_index* r : ix->replicas) { if ((rc = freddy_gpu_update_codebook(r, codebook))) break; ++done; }
    if (!rc) {
      std::vector<freddy_gpu_index*> none;
      none.swap(ix->replicas);
      rc = freddy_gpu_update_codebook(ix, codebook);
      none.swap(ix->replicas);
      if (!rc) return FREDDY_OK;
      done = ix->replicas.size();
    }
    if (done > 0 || rc == FREDDY_E_HIP || rc == FREDDY_E_NOMEM) ix->poisoned = true;
    return rc;
  }
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipDeviceSynchronize());   // (searches of every stream and lane have drained before the tables change)
  if (ix->kind == KIND_PQ) return derive_codebook_tables(ix, codebook);
  if (ix->kind == KIND_IVF) {
    if (int rc = derive_codebook_tables(ix, codebook)) return rc;
    return refresh_row_terms(ix);
  }
  if (ix->kind == KIND_IVPQ) {
    JoinIndex& j = ix->join;
    std::vector<float> cbT((size_t)j.m * j.S * j.K);
    for (int p = 0; p < j.m; ++p)
      for (int c = 0; c < j.K; ++c)
        for (int i = 0; i < j.S; ++i) cbT[((size_t)p * j.S + i) * j.K + c] = codebook[((size_t)p * j.K + c) * j.S + i];
    HIP_TRY(hipMemcpy(j.cbT, cbT.data(), sizeof(float) * cbT.size(), hipMemcpyHostToDevice));
    return FREDDY_OK;
  }
  return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
}

// ---------------------------------------------------------------------------------------
// kNN-join (ivpq_search_in): host loop in join.h
// ---------------------------------------------------------------------------------------
extern "C" int freddy_gpu_knn_join(freddy_gpu_index_t* ix, const float* queries, int32_t Q, int32_t k,
                                   const int32_t* target_ids, int64_t n_targets, int32_t alpha, int32_t pvf,
                                   int32_t method, int32_t use_target_lists, float confidence, int32_t double_threshold,
                                   int32_t* out_ids, float* out_dist, int32_t* iterations_out) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (ix->kind != KIND_IVPQ) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  if (Q < 0 || k <= 0 || n_targets < 0) return fail(FREDDY_E_ARG, "bad sizes");
  if (Q > 0 && (!queries || !out_ids || !out_dist)) return fail(FREDDY_E_ARG, "NULL buffer");
  if (n_targets > 0 && !target_ids) return fail(FREDDY_E_ARG, "NULL target ids");
  HIP_TRY(hipSetDevice(ix->device));
  int rc = join_run(&ix->join, ix->stream, queries, Q, k, target_ids, n_targets, alpha, pvf, method,
                    use_target_lists, confidence, double_threshold, out_ids, out_dist, iterations_out);
  if (rc) return fail(rc, "%s", join_error());
  return FREDDY_OK;
}

extern "C" int freddy_gpu_last_track(const freddy_gpu_index_t* ix, freddy_track* out) {
  if (!ix || !out) return fail(FREDDY_E_ARG, "NULL argument");
  if (ix->kind != KIND_IVPQ) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  *out = ix->join.track;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_last_track_sized(const freddy_gpu_index_t* ix, void* out, size_t out_size) {
  if (!ix || !out) return fail(FREDDY_E_ARG, "NULL argument");
  if (ix->kind != KIND_IVPQ) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  const size_t n = std::min(out_size, sizeof(freddy_track));
  memcpy(out, &ix->join.track, n);
  return (int)n;
}
extern "C" int freddy_gpu_abi_version(void) { return FREDDY_GPU_ABI_VERSION; }

// ---------------------------------------------------------------------------------------
// exact brute-force kNN
// ---------------------------------------------------------------------------------------
// ---- exact kNN as filter + refine (exact2.h) ----------------------------------------------------------------------
// The table's largest |element| / largest row norm over rows [r0, r0 + n) of the row-major copy, folded into the handle's.
static int exf_table_stats(freddy_gpu_index* ix, int64_t r0, int64_t n) {
  const bool shape_ok = ix->d % 4 == 0 && ix->d <= 512 && ix->d >= 16;
  if (!shape_ok) { ix->exf_ok = false; return 0; }
  if (n <= 0) return 0;
  if (ix->exf_small.ensure(4096)) return fail(FREDDY_E_NOMEM, "device allocation failed");
  uint32_t* st = ix->exf_small.as<uint32_t>() + 512;   // (the upper part of the small buffer; the lower one is per-call state)
  HIP_TRY(hipMemsetAsync(st, 0, 16, ix->stream));
  const unsigned grid = (unsigned)std::min<int64_t>((n + 3) / 4, (int64_t)ix->n_cus * 8);
  hipLaunchKernelGGL(exf_table_stats_kernel, dim3(grid), dim3(256), 0, ix->stream, ix->coarse + (size_t)r0 * ix->d, n, ix->d, st);
  HIP_TRY(hipGetLastError());
  uint32_t h[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h, st, 16, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  float amax, n2;
  memcpy(&amax, &h[0], 4); memcpy(&n2, &h[1], 4);
  const bool first = r0 == 0;
  if (h[2] || !(n2 < 1e30f)) { ix->exf_ok = false; return 0; }
  const float xn = std::sqrt(n2) * (1.0f + 1e-5f);
  int64_t relayout_from = r0;
  if (first) { ix->exf_ok = true; ix->exf_xnorm = xn; ix->exf_ex = exf_scale_exp(amax); }
  else if (ix->exf_ok) {
    ix->exf_xnorm = std::max(ix->exf_xnorm, xn);
    const int ex_new = std::min(ix->exf_ex, exf_scale_exp(amax));   // (a larger element: a smaller scale -> everything is laid out again)
    if (ex_new != ix->exf_ex) relayout_from = 0;
    ix->exf_ex = ex_new;
  } else return 0;
  // the fragment-order copy: rows [relayout_from, r0 + n) (whole strips; the strip the old last row sat in is rewritten)
  const int T = (ix->d + 15) / 16;
  const int64_t n_total = r0 + n, strips = (n_total + 31) / 32, strip0 = relayout_from / 32;
  const size_t need = (size_t)strips * T * 2 * 64 * 16;
  if (need > ix->exf_xf.cap) {
    DevBuf bigger;
    if (bigger.ensure(need)) { ix->exf_ok = false; return 0; }   // (no room for the copy: the all-exact kernels stay)
    if (ix->exf_xf.p && strip0 > 0) HIP_TRY(hipMemcpy(bigger.p, ix->exf_xf.p, (size_t)strip0 * T * 2 * 64 * 16, hipMemcpyDeviceToDevice));
    ix->bytes += (int64_t)bigger.cap - (int64_t)ix->exf_xf.cap;
    ix->exf_xf.release();
    ix->exf_xf = bigger;
  }
  const int64_t threads = (strips - strip0) * T * 64;
  hipLaunchKernelGGL(exf_layout_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ix->stream, ix->coarse, n_total, ix->d, T, ix->exf_ex,
                     strip0, strips - strip0, ix->exf_xf.as<h8v>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->exf_xf_strips = strips;
  return 0;
}

// The filter + refine path for all rows of the table.  *fell_back = 1: a candidate buffer overflowed or a query was not
// finite -- nothing was written, the caller runs the all-exact kernels.
static int exact_filter_search(freddy_gpu_index* ix, Workspace* ws, hipStream_t s, const float* d_queries, int Q, int k, int* fell_back) {
  *fell_back = 0;
  const int d = ix->d, T = (d + 15) / 16, L = k, V = pick_V(L);
  const int64_t N = ix->N;
  const bool all = (ix->tune.check_brackets & 4) != 0;
  const int64_t cap64 = all ? N : std::min<int64_t>(N, 8192);
  const int cap = (int)cap64;
  const int n_sample = (int)std::min<int64_t>(N, EXF_SAMPLE);
  // small per-call state: [0..63] thr, [64..127] qeps, [128..191] qunscale, [192..255] cand_cnt, [256] qbad
  if (ix->exf_small.ensure(4096) || ix->exf_qfrag.ensure((size_t)2 * T * 2 * 64 * 16) ||
      ix->exf_sample.ensure(sizeof(float) * (size_t)EXF_QT * n_sample) || ix->exf_cand.ensure(sizeof(uint2) * (size_t)EXF_QT * cap) ||
      ws->w_part.ensure(sizeof(u64) * (size_t)Q * EXF_TW * L) || ws->w_out_ids.ensure(sizeof(int32_t) * (size_t)Q * k) ||
      ws->w_out_dist.ensure(sizeof(float) * (size_t)Q * k))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  if (!ix->viol) {
    HIP_TRY(hipMalloc((void**)&ix->viol, 4 * sizeof(int32_t)));
    HIP_TRY(hipMemset(ix->viol, 0, 4 * sizeof(int32_t)));
  }
  float* sm = ix->exf_small.as<float>();
  float* thr = sm; float* qeps = sm + 64; float* qunscale = sm + 128;
  int32_t* cand_cnt = reinterpret_cast<int32_t*>(sm + 192);
  int32_t* qbad = reinterpret_cast<int32_t*>(sm + 256);
  HIP_TRY(hipMemsetAsync(qbad, 0, 4, s));
  HIP_TRY(hipMemsetAsync(ix->viol + 3, 0, 4, s));
  const size_t lds1 = (size_t)1 * T * 2 * 64 * 16, lds2 = 2 * lds1;
  for (int q0 = 0; q0 < Q; q0 += EXF_QT) {
    const int nq = std::min(EXF_QT, Q - q0);
    const int NT = nq <= 32 ? 1 : 2;
    ExfPrepArgs pa;
    pa.queries = d_queries + (size_t)q0 * d; pa.nq = nq; pa.d = d; pa.T = T; pa.xmax_norm = ix->exf_xnorm; pa.ex = ix->exf_ex;
    pa.eps_factor = exf_eps_factor(d); pa.qfrag = ix->exf_qfrag.as<h8v>(); pa.qeps = qeps; pa.qunscale = qunscale; pa.qbad = qbad;
    timed_launch(ix, s, "exact_prep", [&] { hipLaunchKernelGGL(exf_prep_kernel, dim3(EXF_QT), dim3(256), 0, s, pa); });
    HIP_TRY(hipGetLastError());
    ExfArgs fa;
    fa.xf = ix->exf_xf.as<h8v>(); fa.n_rows = n_sample; fa.T = T; fa.qfrag = ix->exf_qfrag.as<h8v>();
    fa.qunscale = qunscale; fa.sample_out = ix->exf_sample.as<float>(); fa.thr = thr; fa.cand_cnt = cand_cnt; fa.cand = ix->exf_cand.as<uint2>(); fa.cap = cap;
    auto grid_for = [&](int64_t rows) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((rows + 255) / 256, (int64_t)ix->n_cus * 2)); };
    timed_launch(ix, s, "exact_sample", [&] {
      if (NT == 1) hipLaunchKernelGGL((exf_filter_kernel<1, true>), dim3(grid_for(n_sample)), dim3(EXF_WG), lds1, s, fa);
      else hipLaunchKernelGGL((exf_filter_kernel<2, true>), dim3(grid_for(n_sample)), dim3(EXF_WG), lds2, s, fa);
    });
    HIP_TRY(hipGetLastError());
    ExfThrArgs ta;
    ta.sample = fa.sample_out; ta.n_sample = n_sample; ta.nq = nq; ta.k = k; ta.qeps = qeps; ta.qunscale = qunscale; ta.thr = thr; ta.refine_all = all ? 1 : 0;
    timed_launch(ix, s, "exact_threshold", [&] { hipLaunchKernelGGL(exf_threshold_kernel, dim3(EXF_QT), dim3(64 * EXF_TW), 0, s, ta); });
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(cand_cnt, 0, sizeof(int32_t) * EXF_QT, s));
    fa.n_rows = N; fa.sample_out = nullptr;
    timed_launch(ix, s, "exact_filter", [&] {
      if (NT == 1) hipLaunchKernelGGL((exf_filter_kernel<1, false>), dim3(grid_for(N)), dim3(EXF_WG), lds1, s, fa);
      else hipLaunchKernelGGL((exf_filter_kernel<2, false>), dim3(grid_for(N)), dim3(EXF_WG), lds2, s, fa);
    });
    HIP_TRY(hipGetLastError());
    ExfRefineArgs ra;
    ra.rows = ix->coarse; ra.queries = d_queries + (size_t)q0 * d; ra.cand = fa.cand; ra.cand_cnt = cand_cnt; ra.qeps = qeps;
    ra.part = ws->w_part.as<u64>() + (size_t)q0 * EXF_TW * L; ra.viol = ix->viol; ra.cap = cap; ra.d = d; ra.L = L; ra.count_checked = all ? 1 : 0;
    const size_t rlds = (((size_t)d * 4 + 15) & ~(size_t)15) + (size_t)EXF_TW * 64 * sizeof(u64);
    timed_launch(ix, s, "exact_refine", [&] {
      switch (V) {
        case 1: hipLaunchKernelGGL((exf_refine_kernel<1>), dim3(nq), dim3(64 * EXF_TW), rlds, s, ra); break;
        default: hipLaunchKernelGGL((exf_refine_kernel<2>), dim3(nq), dim3(64 * EXF_TW), rlds, s, ra); break;
      }
    });
    HIP_TRY(hipGetLastError());
  }
  timed_launch(ix, s, "exact_merge", [&] {
    switch (V) {
      case 1: hipLaunchKernelGGL((exact_merge_kernel<1>), dim3(Q), dim3(4 * 64), (size_t)4 * 64 * (1 + 1) * sizeof(u64), s, ws->w_part.as<u64>(), EXF_TW, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
      default: hipLaunchKernelGGL((exact_merge_kernel<2>), dim3(Q), dim3(4 * 64), (size_t)4 * 64 * (2 + 1) * sizeof(u64), s, ws->w_part.as<u64>(), EXF_TW, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
    }
  });
  HIP_TRY(hipGetLastError());
  int32_t flags[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(&flags[0], ix->viol + 3, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&flags[1], qbad, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (flags[0] || flags[1]) *fell_back = 1;
  return 0;
}

extern "C" int freddy_gpu_pin_vectors(const freddy_vec_desc* t, int device, freddy_gpu_index_t** out) {
  if (!t || !out || t->d <= 0 || t->N < 0 || (t->N && (!t->ids || !t->vectors))) return fail(FREDDY_E_ARG, "bad argument");
  if (t->N > (int64_t)INT32_MAX - 64) return fail(FREDDY_E_LIMIT, "N too large for 32-bit row positions");
  for (int64_t r = 1; r < t->N; ++r)
    if (t->ids[r] <= t->ids[r - 1]) return fail(FREDDY_E_ARG, "ids must be strictly ascending (row %lld)", (long long)r);
  freddy_gpu_index* ix = new freddy_gpu_index();
  ix->kind = KIND_VEC;
  ix->d = t->d; ix->N = t->N;
  int rc = open_device(ix, device);
  if (!rc) {
    ix->n_blocks = (t->N + 63) / 64;
    const size_t xb_bytes = sizeof(float) * (size_t)std::max<int64_t>(ix->n_blocks, 1) * t->d * 64;
    if (hipMalloc((void**)&ix->xb, xb_bytes) != hipSuccess) rc = fail(FREDDY_E_NOMEM, "device allocation of %zu bytes failed", xb_bytes);
    else ix->bytes += (int64_t)xb_bytes;
    if (!rc && upload(&ix->ids, t->ids, (size_t)t->N, &ix->bytes)) rc = fail(FREDDY_E_NOMEM, "device allocation failed");
    // row-major rows go up in slices and are re-blocked on the device
    const int64_t slice = 1 << 16;
    DevBuf tmp;
    for (int64_t r0 = 0; !rc && r0 < t->N; r0 += slice) {
      const int64_t n = std::min(slice, t->N - r0);
      if (tmp.ensure(sizeof(float) * (size_t)n * t->d)) { rc = fail(FREDDY_E_NOMEM, "device allocation failed"); break; }
      if (hipMemcpy(tmp.p, t->vectors + (size_t)r0 * t->d, sizeof(float) * (size_t)n * t->d, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(FREDDY_E_HIP, "hipMemcpy failed"); break;
      }
      hipLaunchKernelGGL(block_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, ix->stream, tmp.as<float>(), nullptr, n,
                         ix->xb + (size_t)(r0 / 64) * t->d * 64, nullptr, t->d);
      if (hipStreamSynchronize(ix->stream) != hipSuccess) { rc = fail(FREDDY_E_HIP, "re-blocking kernel failed"); break; }
    }
    tmp.release();
  }
  if (!rc) {
    ix->h_ids.assign(t->ids, t->ids + t->N);
    // the source rows are only needed again for "id = ANY(...)" subsets: keep them row-major too
    if (t->N && upload(&ix->coarse, t->vectors, (size_t)t->N * t->d, &ix->bytes)) rc = fail(FREDDY_E_NOMEM, "device allocation failed");
    if (!rc && t->N) rc = exf_table_stats(ix, 0, t->N);
  }
  if (rc) { free_index(ix); return rc; }
  *out = ix;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_exact_search(freddy_gpu_index_t* ix, const float* queries, int32_t Q, int32_t k,
                                       const int32_t* subset_ids, int64_t n_subset, int32_t* out_ids, float* out_sim) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (ix->kind != KIND_VEC) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  if (Q < 0 || k <= 0 || n_subset < 0 || (n_subset > 0 && !subset_ids)) return fail(FREDDY_E_ARG, "bad sizes");
  if (Q > 0 && (!queries || !out_ids || !out_sim)) return fail(FREDDY_E_ARG, "NULL buffer");
  if (k > 1024) return fail(FREDDY_E_LIMIT, "k=%d exceeds this build's limit of 1024", k);
  if (Q == 0) return FREDDY_OK;
  HIP_TRY(hipSetDevice(ix->device));
  Workspace* ws = workspace_for(ix, ix->stream);
  hipStream_t s = ix->stream;
  const int d = ix->d, L = k, V = pick_V(L);
  const float* xb = ix->xb;
  const int32_t* pos = nullptr;
  int64_t n_rows = ix->N, n_blocks = ix->n_blocks;
  if (subset_ids) {
    std::vector<int32_t> rows;
    rows.reserve((size_t)n_subset);
    for (int64_t i = 0; i < n_subset; ++i) {
      auto it = std::lower_bound(ix->h_ids.begin(), ix->h_ids.end(), subset_ids[i]);
      if (it != ix->h_ids.end() && *it == subset_ids[i]) rows.push_back((int32_t)(it - ix->h_ids.begin()));
    }
    std::sort(rows.begin(), rows.end());
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    n_rows = (int64_t)rows.size();
    n_blocks = (n_rows + 63) / 64;
    if (ws->w_sub_rows.ensure(sizeof(int32_t) * std::max<size_t>(rows.size(), 1)) ||
        ws->w_sub_pos.ensure(sizeof(int32_t) * (size_t)std::max<int64_t>(n_blocks, 1) * 64) ||
        ws->w_resid.ensure(sizeof(float) * (size_t)std::max<int64_t>(n_blocks, 1) * d * 64))
      return fail(FREDDY_E_NOMEM, "workspace allocation failed");
    if (n_rows) {
      HIP_TRY(hipMemcpyAsync(ws->w_sub_rows.p, rows.data(), sizeof(int32_t) * rows.size(), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(block_rows_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, ix->coarse, ws->w_sub_rows.as<int32_t>(), n_rows,
                         ws->w_resid.as<float>(), ws->w_sub_pos.as<int32_t>(), d);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(s));   // `rows` is a host temporary
    }
    xb = ws->w_resid.as<float>();
    pos = ws->w_sub_pos.as<int32_t>();
  }
  // Filter + refine (exact2.h): the whole table, k <= 32, finite rows of a supported shape; identical lists.
  const bool want_filter = !subset_ids && ix->exf_ok && k <= 32 && ix->tune.exact_filter != 0 &&
                           (ix->tune.exact_filter == 1 || n_rows >= 8192) && n_rows >= 1;
  if (want_filter) {
    if (ws->w_q.ensure(sizeof(float) * (size_t)Q * d)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
    HIP_TRY(hipMemcpyAsync(ws->w_q.p, queries, sizeof(float) * (size_t)Q * d, hipMemcpyHostToDevice, s));
    int fell_back = 0;
    if (int rc = exact_filter_search(ix, ws, s, ws->w_q.as<float>(), Q, k, &fell_back)) return rc;
    if (!fell_back) {
      HIP_TRY(hipMemcpyAsync(out_ids, ws->w_out_ids.p, sizeof(int32_t) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(out_sim, ws->w_out_dist.p, sizeof(float) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      return FREDDY_OK;
    }
  }
  int chunk_blocks = 8;   // 512 rows per workgroup-chunk; longer chunks once the grid is large enough
  const int EX_QT = ex_qt(V, Q);
  const int qgroups = (Q + EX_QT - 1) / EX_QT;
  while ((n_blocks + chunk_blocks - 1) / chunk_blocks * (int64_t)qgroups > 8192 && chunk_blocks < 1024) chunk_blocks *= 2;
  const int nchunk = (int)std::max<int64_t>(1, (n_blocks + chunk_blocks - 1) / chunk_blocks);
  if (ws->w_q.ensure(sizeof(float) * (size_t)Q * d) || ws->w_out_ids.ensure(sizeof(int32_t) * (size_t)Q * k) ||
      ws->w_out_dist.ensure(sizeof(float) * (size_t)Q * k) ||
      ws->w_part.ensure(sizeof(u64) * (size_t)Q * nchunk * EX_WAVES * L))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  HIP_TRY(hipMemcpyAsync(ws->w_q.p, queries, sizeof(float) * (size_t)Q * d, hipMemcpyHostToDevice, s));
  ExactArgs ea;
  ea.xb = xb; ea.pos = pos; ea.queries = ws->w_q.as<float>(); ea.part = ws->w_part.as<u64>();
  ea.n_rows = n_rows; ea.n_blocks = (int)n_blocks; ea.chunk_blocks = chunk_blocks; ea.nchunk = nchunk; ea.Q = Q; ea.d = d; ea.L = L;
  const size_t lds = (((size_t)d * EX_QT * 4 + 15) & ~(size_t)15) + (size_t)EX_WAVES * EX_QT * 64 * sizeof(u64);
  dim3 grid((unsigned)nchunk, (unsigned)qgroups);
  timed_launch(ix, s, "exact_scan", [&] {
    switch (V) {
      case 1: if (EX_QT == 16) hipLaunchKernelGGL((exact_scan_kernel<1, 16>), grid, dim3(EX_WG), lds, s, ea);
              else hipLaunchKernelGGL((exact_scan_kernel<1, 8>), grid, dim3(EX_WG), lds, s, ea);
              break;
      case 2: if (EX_QT == 16) hipLaunchKernelGGL((exact_scan_kernel<2, 16>), grid, dim3(EX_WG), lds, s, ea);
              else hipLaunchKernelGGL((exact_scan_kernel<2, 8>), grid, dim3(EX_WG), lds, s, ea);
              break;
      case 4: if (EX_QT == 16) hipLaunchKernelGGL((exact_scan_kernel<4, 16>), grid, dim3(EX_WG), lds, s, ea);
              else hipLaunchKernelGGL((exact_scan_kernel<4, 8>), grid, dim3(EX_WG), lds, s, ea);
              break;
      case 8: hipLaunchKernelGGL((exact_scan_kernel<8, 8>), grid, dim3(EX_WG), lds, s, ea); break;
      default: hipLaunchKernelGGL((exact_scan_kernel<16, 8>), grid, dim3(EX_WG), lds, s, ea); break;
    }
  });
  HIP_TRY(hipGetLastError());
  const int ppq = nchunk * EX_WAVES;
  timed_launch(ix, s, "exact_merge", [&] {
    switch (V) {
      case 1: hipLaunchKernelGGL((exact_merge_kernel<1>), dim3(Q), dim3(16 * 64), (size_t)16 * 64 * (1 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
      case 2: hipLaunchKernelGGL((exact_merge_kernel<2>), dim3(Q), dim3(16 * 64), (size_t)16 * 64 * (2 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
      case 4: hipLaunchKernelGGL((exact_merge_kernel<4>), dim3(Q), dim3(16 * 64), (size_t)16 * 64 * (4 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
      case 8: hipLaunchKernelGGL((exact_merge_kernel<8>), dim3(Q), dim3(8 * 64), (size_t)8 * 64 * (8 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
      default: hipLaunchKernelGGL((exact_merge_kernel<16>), dim3(Q), dim3(4 * 64), (size_t)4 * 64 * (16 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
    }
  });
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out_ids, ws->w_out_ids.p, sizeof(int32_t) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(out_sim, ws->w_out_dist.p, sizeof(float) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return FREDDY_OK;
}
