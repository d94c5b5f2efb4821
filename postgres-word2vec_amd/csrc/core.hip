// core.hip -- errors, options, workspaces, profiling records, unpin and the handle's counters (see internal.h).
#include "internal.h"

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* freddy_gpu_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------
// Backends on one GPU: the registry of live backends per physical GPU is registry.h (plain C++, tested on its own)
// ---------------------------------------------------------------------------------------
#include "registry.h"

void backend_handles(int delta, int device) { freddy::registry::handles(delta, device); }
int backends_other(bool searching, int device) { return freddy::registry::others(searching, device); }
void backend_busy(int delta) { freddy::registry::busy(delta); }
// Before the process's first HIP call: the hardware queues the runtime may create (never overrides the environment).
void choose_hw_queues(int device) {
  static std::once_flag once;
  std::call_once(once, [device] {
    const char* policy = getenv("FREDDY_GPU_HWQ_POLICY");   // "6" / "2": fixed; default: by the registry
    const bool alone = backends_other(false, device) == 0;
    const char* q = (policy && *policy && strcmp(policy, "registry") != 0) ? policy : (alone ? "6" : "2");
    setenv("GPU_MAX_HW_QUEUES", q, 0);
  });
}

int64_t env_int(const char* name, int64_t dflt) {
  const char* e = getenv(name);
  return (e && *e) ? (int64_t)strtoll(e, nullptr, 10) : dflt;
}
Tuning read_tuning() {
  Tuning t;
  t.fused = (int)env_int("FREDDY_GPU_FUSED", t.fused);
  t.scan_kernel = (int)env_int("FREDDY_GPU_FUSED_KERNEL", t.scan_kernel);
  t.reserve_cus = (int)env_int("FREDDY_GPU_RESERVE_CUS", 0);
  t.scan_share = (int)std::max<int64_t>(0, env_int("FREDDY_GPU_SCAN_SHARE", t.scan_share));
  t.pipeline_batch = (int)std::max<int64_t>(16, env_int("FREDDY_GPU_PIPELINE_BATCH", t.pipeline_batch));
  t.pipeline_lanes = (int)std::min<int64_t>(4, std::max<int64_t>(1, env_int("FREDDY_GPU_PIPELINE_LANES", t.pipeline_lanes)));
  t.pq_fused = (int)env_int("FREDDY_GPU_PQ_FUSED", t.pq_fused);
  t.one_launch = (int)env_int("FREDDY_GPU_ONE_LAUNCH", t.one_launch);
  t.coarse_approx = (int)env_int("FREDDY_GPU_COARSE_APPROX", 1);
  t.sparse_items = (int)env_int("FREDDY_GPU_SPARSE_ITEMS", t.sparse_items);
  t.exact_filter = (int)env_int("FREDDY_GPU_EXACT_FILTER", t.exact_filter);
  t.codes_u8 = (int)env_int("FREDDY_GPU_CODES_U8", t.codes_u8);
  t.running_bound = (int)env_int("FREDDY_GPU_RUNNING_BOUND", t.running_bound);
  t.lut_budget_mb = std::max<int64_t>(1, env_int("FREDDY_GPU_LUT_BUDGET_MB", t.lut_budget_mb));
  t.lane0_own = (int)env_int("FREDDY_GPU_LANE0_OWN", t.lane0_own);
  t.coarse_pieces = (int)env_int("FREDDY_GPU_COARSE_PIECES", t.coarse_pieces);
#ifdef FREDDY_LAB
  t.scan_prof = getenv("FREDDY_GPU_FUSED_PROF") != nullptr;
#endif
  return t;
}

// The workspace of the stream a search is enqueued on.  With every slot taken a new stream takes over the least
// recently used one -- after the whole device has drained (rare; no handle of a possibly destroyed caller stream is touched).
Workspace* workspace_for(freddy_gpu_index* ix, hipStream_t s) {
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

void free_index(freddy_gpu_index* ix) {
  if (!ix) return;
  if (ix->registered) { ix->registered = false; backend_handles(-1, ix->device); }
  for (freddy_gpu_index* r : ix->replicas) free_index(r);
  ix->replicas.clear();
  (void)hipSetDevice(ix->device);
  (void)hipDeviceSynchronize();   // (every stream that searched on this handle, without touching a caller's stream handle)
  for (Workspace& w : ix->ws) w.release();
  for (DevBuf* b : {&ix->exf_qfrag, &ix->exf_small, &ix->exf_sample, &ix->exf_cand, &ix->exf_xf}) b->release();
  if (ix->hio_in) { (void)hipHostFree(ix->hio_in); ix->hio_in = nullptr; ix->hio_in_cap = 0; }
  if (ix->hio_out) { (void)hipHostFree(ix->hio_out); ix->hio_out = nullptr; ix->hio_out_cap = 0; }
  for (Lane& l : ix->lanes) {
    if (l.stream && l.stream != ix->stream) (void)hipStreamDestroy(l.stream);
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
  else if (n == "scan_share") t.scan_share = (int)std::max<int64_t>(0, value);
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
  else if (n == "running_bound") t.running_bound = (int)value;
  else if (n == "coarse_pieces") t.coarse_pieces = (int)value;
#ifdef FREDDY_LAB
  else if (n == "fused_prof") t.scan_prof = (int)value;
  else if (n == "scan_fence") t.scan_fence = (int)value;
#endif
  else return fail(FREDDY_E_ARG, "unknown option '%s'", name);
  return FREDDY_OK;
}

int check_search_args(const freddy_gpu_index* ix, int kind, const void* q, int Q, int k, const void* oi,
                             const void* od) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (ix->kind != kind) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  if (ix->poisoned) return fail(FREDDY_E_HIP, "this handle's devices hold different tables (an append / codebook update failed part-way): unpin it and pin again");
  if (Q < 0 || k <= 0) return fail(FREDDY_E_ARG, "Q must be >= 0 and k > 0");
  if (Q > 0 && (!q || !oi || !od)) return fail(FREDDY_E_ARG, "NULL buffer");
  if (k > 4096) return fail(FREDDY_E_LIMIT, "k=%d exceeds this build's limit of 4096", k);
  return 0;
}

extern "C" int freddy_gpu_host_alloc(void** out, size_t bytes) {
  choose_hw_queues(-1);   // (as open_device: this call may be the process's first HIP call; -1: the device is not known here -- every registered backend counts)
  if (!out) return fail(FREDDY_E_ARG, "NULL argument");
  *out = nullptr;
  if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { *out = nullptr; return fail(FREDDY_E_NOMEM, "pinned host allocation of %zu bytes failed", bytes); }
  return FREDDY_OK;
}
extern "C" int freddy_gpu_host_free(void* p) {
  if (p) HIP_TRY(hipHostFree(p));
  return FREDDY_OK;
}

extern "C" int freddy_gpu_replica_count(const freddy_gpu_index_t* ix) { return ix ? 1 + (int)ix->replicas.size() : 0; }
extern "C" int freddy_gpu_abi_version(void) { return FREDDY_GPU_ABI_VERSION; }
