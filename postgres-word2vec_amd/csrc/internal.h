// internal.h -- what the translation units of libfreddy_gpu.so share on the HOST side: the handle, its workspaces and
// options, error reporting, and the functions one unit calls in another.  The C ABI is include/freddy_gpu.h; the units:
//   core.hip    errors, options, workspaces, profiling records, unpin, counters
//   pin.hip     pin_pq / pin_ivf(_multi): table layouts; append_rows / update_codebook (HBM index mutation)
//   ivfadc.hip  the IVFADC search: cell selection, work table, scans, merge; *_dev entry, host-buffer pipeline, one-query launch
//   pq.hip      pq_search (+ subsets, pseudo-list batches, one-query launch), grouping_pq
//   join.hip    pin_ivpq, knn_join
//   exact.hip   pin_vectors, exact kNN
//   build.hip   encode, insert_quantize, k-means
// Kernel headers are included by the unit that launches them (kernels shared by two units are static or templates).
#pragma once
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

#include "join_index.h"

using namespace freddy;

// ---- errors (core.hip) ----
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(FREDDY_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                  __FILE__, __LINE__);                                                      \
  } while (0)

enum { KIND_PQ = 1, KIND_IVF = 2, KIND_IVPQ = 3, KIND_VEC = 4 };

// Options.  Read from the environment ONCE, when an index is pinned (never on the search path); freddy_gpu_set_option
// changes them on a pinned index.  None of them changes a result: every setting gives the same lists.  The first group
// is for deployments, the second selects the alternative paths the tests compare with each other (every one of them has a
// GPU test), the third the self-checks.  (INTEGRATION.md lists them; timing experiments of earlier rounds are gone --
// profiles/HISTORY.md has their numbers.)
struct Tuning {
  // -- deployment
  int scan_share = 0;          // FREDDY_GPU_SCAN_SHARE (0 = auto: 1, or 2 for a host-buffer search that starts while another BACKEND is searching -- core.hip registry): the batches that share the chip with one of this handle's -- the batches the CALLER keeps
                               // in flight through the *_dev entry points (one stream each), or the other BACKENDS (processes) searching at
                               // the same time: a persistent scan takes n_cus / share CUs so that the scans run side by side.
                               // An explicit contract -- the library does not guess it; the host-buffer calls multiply it by their lane count
  int reserve_cus = 0;         // FREDDY_GPU_RESERVE_CUS: CUs the persistent scan leaves to the kernels of other streams (RCCL beside the scans)
  int pipeline_batch = 2048;   // FREDDY_GPU_PIPELINE_BATCH: queries per sub-batch of the host-buffer pipeline (freddy_gpu_ivfadc_search)
  int pipeline_lanes = 4;      // FREDDY_GPU_PIPELINE_LANES: sub-batches in flight inside one host-buffer call (1..4)
  int lane0_own = 0;           // FREDDY_GPU_LANE0_OWN: 1 = the first pipeline lane on a stream of its own (until round 6) instead of the handle's stream
  int coarse_pieces = 1;       // FREDDY_GPU_COARSE_PIECES: a host-buffer call of ONE sub-batch launches its cell selection / table kernel per staged piece of the queries (0 = once, behind the last piece)
  int64_t lut_budget_mb = 8192;      // FREDDY_GPU_LUT_BUDGET_MB: per-call workspace cap (queries are chunked to fit); 288 GB of HBM: 8 GiB = 12 800 queries at nprobe 10
  // -- path selection (tests)
  int fused = -1;              // FREDDY_GPU_FUSED: -1 auto (cell-grouped scans for >= 256 items), 0 generic kernels, 1 always
  int scan_kernel = 5;         // FREDDY_GPU_FUSED_KERNEL: 5 filter + refine on int16 slabs (fused5.h), 3 the reference's arithmetic for every row (fused3.h)
  int coarse_approx = 1;       // FREDDY_GPU_COARSE_APPROX: cell selection as filter + refine (coarse.h); 0 = every distance exact
  int one_launch = 1;          // FREDDY_GPU_ONE_LAUNCH: a single query through the host-buffer calls as ONE launch (one.h) instead of a chain
  int pq_fused = -1;           // FREDDY_GPU_PQ_FUSED: batches over the flat PQ table through the cell-grouped filter + refine scan: -1 = from 16 queries on, 0 never, 1 always
  int sparse_items = 2;        // FREDDY_GPU_SPARSE_ITEMS: cells that at most this many queries of a batch probe are scanned item by item (sparse5.h) instead of as cell-grouped work entries (0 = never, < 0 = always for cells of up to that many items)
  int running_bound = 1;       // FREDDY_GPU_RUNNING_BOUND: the scan's entries share a per-query running bound of the L-th smallest cheap distance
                               // (FilterArgs::tau_run): fewer survivors for the merge; 0 = every (item, chunk) cuts at its own threshold
  int codes_u8 = 1;            // FREDDY_GPU_CODES_U8: K <= 256: the integer-slab scans read one byte per code (packed8, 16 instead of 28 B per row) -- 1: the scan with the whole entry's slab in LDS (fused8.h), 2: the six-phase scan (fused5.h); 0 = the int16 layout
  int exact_filter = -1;       // FREDDY_GPU_EXACT_FILTER: exact kNN as MFMA filter + exact refine (exact2.h): -1 auto (tables of >= 8192 rows, k <= 32), 0 never, 1 always
  // -- self-checks (tests): bit 0 = the scan keeps every row and the merge refines every row (every probed row's bracket is checked),
  //    bit 1 = the cell selection refines every cell, bit 2 = exact kNN refines every row
  int check_brackets = 0;
#ifdef FREDDY_LAB
  int scan_prof = 0;           // FREDDY_GPU_FUSED_PROF (lab builds only): per-phase cycle sums of the scan kernel on stderr
  int scan_fence = 0;          // option scan_fence (lab builds only; tools/lab/ablate.py): FilterArgs::fence -- parts of the scan kernel switched OFF (results wrong, time only)
#endif
};
int64_t env_int(const char* name, int64_t dflt);
int backends_other(bool searching, int device);   // core.hip: live backends (processes) besides this one with handles on the same PHYSICAL GPU as HIP device `device` (-1: on any) -- registered / inside a host-buffer search
void backend_handles(int delta, int device);   // +1 per pinned index of this process (on HIP device `device`), -1 when it is freed: registered while > 0
void backend_busy(int delta);         // this process enters (+1) / leaves (-1) a host-buffer search
void choose_hw_queues(int device);    // GPU_MAX_HW_QUEUES before the first HIP call (never overrides the environment)
struct BackendBusy { BackendBusy() { backend_busy(1); } ~BackendBusy() { backend_busy(-1); } };
// option scan_share as a number: the explicit value, or (0 = auto) 2 while another backend is searching, else 1
static inline int scan_share_now(int option, bool host_call, int device) { return option > 0 ? option : (host_call && backends_other(true, device) > 0 ? 2 : 1); }
Tuning read_tuning();

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
      w_sub_packed, w_sub_pos, w_sub_blk, w_cellcnt, w_sorted, w_groups, w_surv, w_surv_cnt, w_prof, w_qc, w_qn, w_records, w_qn2, w_item_dist, w_tmin, w_one, w_oneb, w_bigsel, w_floor;
  uint64_t one_shape = 0;          // the one-launch kernels' buffer (w_oneb): shape of the call that wrote it last, and that call's epoch (one.h)
  uint32_t one_epoch = 0;
  bool one_pending = false;        // one_buffer() flipped the epoch and the kernel was not (yet) launched
  void release() {
    DevBuf* bufs[] = {&w_q, &w_distT, &w_used, &w_item_cell, &w_item_query, &w_rows, &w_resid, &w_lut, &w_part,
                      &w_cand, &w_found, &w_act0, &w_act1, &w_cnt, &w_out_ids, &w_out_dist, &w_sub_rows, &w_sub_packed,
                      &w_sub_pos, &w_sub_blk, &w_cellcnt, &w_sorted, &w_groups, &w_surv, &w_surv_cnt, &w_prof, &w_qc,
                      &w_qn, &w_records, &w_qn2, &w_item_dist, &w_tmin, &w_one, &w_oneb, &w_bigsel, &w_floor};
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
  int64_t packed8_bytes = 0;    // its share of `bytes`
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
  bool exf_dirty = true;        // no filter + refine call has completed yet, or the last one failed part-way: its device-side words are cleared before the next
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
  bool registered = false;        // counted in the registry of backends (core.hip backend_handles)
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

// The workspace of the stream a search is enqueued on (core.hip).
Workspace* workspace_for(freddy_gpu_index* ix, hipStream_t s);

template <class T>
static int upload(T** dst, const T* src, size_t n, int64_t* bytes) {
  *dst = nullptr;
  size_t sz = sizeof(T) * (n ? n : 1);
  if (hipMalloc((void**)dst, sz) != hipSuccess) return -1;
  if (n && hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice) != hipSuccess) return -2;
  if (bytes) *bytes += (int64_t)sz;
  return 0;
}

void free_index(freddy_gpu_index* ix);
int check_search_args(const freddy_gpu_index* ix, int kind, const void* q, int Q, int k, const void* oi, const void* od);

// ---- pin.hip ----
int open_device(freddy_gpu_index* ix, int device);
// (every unit raises the dynamic-LDS limit of its own kernels; open_device calls them all)
int raise_lds_limits_ivfadc(int device);
int raise_lds_limits_pq(int device);
int raise_lds_limits_join(int device);
int raise_lds_limits_exact(int device);

// ---- ivfadc.hip ----
namespace freddy { struct PlanArgs; struct ScanArgs; struct MergeArgs; }
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
int pick_V(int L);
int launch_scan(freddy_gpu_index* ix, hipStream_t s, const ScanArgs& a, int n_items);
int launch_merge(freddy_gpu_index* ix, hipStream_t s, const MergeArgs& a);
int bigk_select_replay(freddy_gpu_index* ix, hipStream_t s, Workspace* ws, ScanArgs sa, int n_items, const MergeArgs& ma, int Q);
int launch_lut(freddy_gpu_index* ix, hipStream_t s, const float* vecs, const int32_t* item_cell, float* lut, int n_items,
               const float* coarse = nullptr, const int32_t* item_query = nullptr);
int ivf_work_table(IvfRun& r, WorkTable& wt);
int ivf_scan_filter(IvfRun& r, const PlanArgs& pa, const WorkTable& wt);
int max_queries_per_chunk(const freddy_gpu_index* ix, int W, int k);
int one_buffer(Workspace* ws, hipStream_t s, uint64_t shape, size_t bytes, uint32_t* epoch);
const void* pinned_device_pointer(const void* p);

// ---- pq.hip ----
int pq_shadow_build(freddy_gpu_index* ix);

// ---- exact.hip ----
int exf_table_stats(freddy_gpu_index* ix, int64_t r0, int64_t n);

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
      if (rcs[(size_t)g]) msgs[(size_t)g] = freddy_gpu_last_error();
    });
  int lo, hi;
  bounds(0, &lo, &hi);
  rcs[0] = fn(ix, lo, hi);
  if (rcs[0]) msgs[0] = freddy_gpu_last_error();
  for (std::thread& t : th) t.join();
  for (int g = 0; g < G; ++g)
    if (rcs[(size_t)g]) return fail(rcs[(size_t)g], "device %d: %s", g == 0 ? ix->device : ix->replicas[(size_t)g - 1]->device, msgs[(size_t)g].c_str());
  return 0;
}

