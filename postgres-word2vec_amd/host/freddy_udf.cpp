// freddy_udf.cpp -- host-side mirror of the FREDDY search UDFs (include/freddy_udf.h).
//
// Stands where freddy_extension/freddy.c and ivpq_search_in.c stand in a PostgreSQL backend:
// keeps the tables and the config functions, resolves ids, pins the tables into HBM once
// (the reference re-reads them through SPI on every call: freddy.c:69,239-241,746-749;
// ivpq_search_in.c:218-232) and forwards every search through the device C ABI.  No distance
// is computed here.
#include "../../include/freddy_udf.h"

#include <algorithm>
#include <initializer_list>
#include <string>
#include <map>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/freddy_gpu.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int gpu_fail(int rc) { return fail(rc, "%s", freddy_gpu_last_error()); }

// CodebookCompound from an entry list: slot from each entry's own (pos, code); positions and
// codeSize are max+1 as in getCodebook (index_utils.c:577-630)
struct Codebook {
  int m = 0, K = 0, s = 0;
  std::vector<float> dense;   // [m][K][s]
  int build(const int32_t* pos, const int32_t* code, const float* vec, int n, int sub) {
    if (n <= 0 || sub <= 0 || !pos || !code || !vec) return fail(-1, "empty codebook");
    m = 0; K = 0; s = sub;
    for (int i = 0; i < n; ++i) {
      if (pos[i] < 0 || code[i] < 0) return fail(-1, "negative pos/code in codebook");
      m = std::max(m, pos[i] + 1);
      K = std::max(K, code[i] + 1);
    }
    if ((int64_t)m * K != n) return fail(-1, "codebook has %d entries, expected positions*codes = %d*%d", n, m, K);
    dense.assign((size_t)m * K * s, 0.0f);
    for (int i = 0; i < n; ++i)
      memcpy(&dense[((size_t)pos[i] * K + code[i]) * s], vec + (size_t)i * s, sizeof(float) * s);
    return 0;
  }
};

// permutation that sorts row ids ascending (table scan order = canonical order)
std::vector<int64_t> order_by_id(const int32_t* ids, int64_t N) {
  std::vector<int64_t> ord((size_t)N);
  std::iota(ord.begin(), ord.end(), 0);
  std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return ids[a] < ids[b]; });
  return ord;
}

}  // namespace

struct freddy_session {
  int device = 0;
  // config functions, defaults from freddy--0.0.1.sql:188-194
  int w = 3, pvf = 20, alpha = 3, long_codes_threshold = 10000000, method_flag = 0, use_targetlist = 1;
  float confidence = 0.8f;
  // google_vecs_norm
  int d = 0;
  std::vector<int32_t> norm_ids;      // ascending
  std::vector<float> norm_vecs;       // [N][d] in that order
  // pinned tables
  freddy_gpu_index_t* pq = nullptr;
  freddy_gpu_index_t* ivf = nullptr;
  freddy_gpu_index_t* ivpq = nullptr;
  freddy_gpu_index_t* vecs = nullptr;  // google_vecs_norm pinned for the exact kNN functions (lazily)
  int pq_d = 0, ivf_d = 0, ivpq_d = 0;
  // what insert_batch reads and writes besides the rows (freddy.c:1545-1560): the codebooks with their
  // count column, the coarse quantizers, and the largest id of each table ("SELECT max(id) + 1")
  Codebook pq_cb, res_cb, ivpq_cb, cq_multi;
  std::vector<int32_t> pq_counts, res_counts, ivpq_counts;   // count column; 1 unless freddy_set_codebook_counts was called
  std::vector<float> coarse;                                 // [C][d]
  int C = 0;
  int32_t pq_max_id = 0, fine_max_id = 0, ivpq_max_id = 0;
};

extern "C" {

const char* freddy_udf_last_error(void) { return g_err; }

int freddy_session_open(int device, freddy_session_t** out) {
  if (!out) return fail(-1, "NULL argument");
  freddy_session* s = new freddy_session();
  s->device = device;
  *out = s;
  return 0;
}

int freddy_session_close(freddy_session_t* s) {
  if (!s) return 0;
  if (s->pq) freddy_gpu_unpin(s->pq);
  if (s->ivf) freddy_gpu_unpin(s->ivf);
  if (s->ivpq) freddy_gpu_unpin(s->ivpq);
  if (s->vecs) freddy_gpu_unpin(s->vecs);
  delete s;
  return 0;
}

int freddy_load_vecs_norm(freddy_session_t* s, const int32_t* ids, const float* vectors, int64_t N, int32_t d) {
  if (!s || !ids || !vectors || N < 0 || d <= 0) return fail(-1, "bad argument");
  std::vector<int64_t> ord = order_by_id(ids, N);
  s->d = d;
  s->norm_ids.resize((size_t)N);
  s->norm_vecs.resize((size_t)N * d);
  for (int64_t i = 0; i < N; ++i) {
    s->norm_ids[i] = ids[ord[i]];
    memcpy(&s->norm_vecs[(size_t)i * d], vectors + (size_t)ord[i] * d, sizeof(float) * d);
  }
  if (s->vecs) { freddy_gpu_unpin(s->vecs); s->vecs = nullptr; }
  return 0;
}

int freddy_load_pq(freddy_session_t* s, const int32_t* cb_pos, const int32_t* cb_code, const float* cb_vectors,
                   int32_t n_entries, int32_t sub_dim, const int32_t* ids, const int16_t* codes, int64_t N) {
  if (!s || !ids || !codes) return fail(-1, "bad argument");
  Codebook cb;
  if (int rc = cb.build(cb_pos, cb_code, cb_vectors, n_entries, sub_dim)) return rc;
  std::vector<int64_t> ord = order_by_id(ids, N);
  std::vector<int32_t> sid((size_t)N);
  std::vector<int16_t> scodes((size_t)N * cb.m);
  for (int64_t i = 0; i < N; ++i) {
    sid[i] = ids[ord[i]];
    memcpy(&scodes[(size_t)i * cb.m], codes + (size_t)ord[i] * cb.m, sizeof(int16_t) * cb.m);
  }
  freddy_pq_desc desc = {cb.m * cb.s, cb.m, cb.K, N, cb.dense.data(), sid.data(), scodes.data()};
  if (s->pq) { freddy_gpu_unpin(s->pq); s->pq = nullptr; }
  if (int rc = freddy_gpu_pin_pq(&desc, s->device, &s->pq)) return gpu_fail(rc);
  s->pq_d = cb.m * cb.s;
  s->pq_cb = cb;
  s->pq_counts.assign((size_t)cb.m * cb.K, 1);
  s->pq_max_id = N ? sid[(size_t)N - 1] : 0;
  return 0;
}

int freddy_load_ivfadc(freddy_session_t* s, const int32_t* coarse_ids_tbl, const float* coarse_vectors, int32_t C,
                       const int32_t* cb_pos, const int32_t* cb_code, const float* cb_vectors, int32_t n_entries,
                       int32_t sub_dim, const int32_t* ids, const int32_t* coarse_id, const int16_t* codes, int64_t N) {
  if (!s || !coarse_ids_tbl || !coarse_vectors || C <= 0 || !ids || !coarse_id || !codes) return fail(-1, "bad argument");
  Codebook cb;
  if (int rc = cb.build(cb_pos, cb_code, cb_vectors, n_entries, sub_dim)) return rc;
  const int d = cb.m * cb.s;
  // coarse ids must be 0..C-1: the C code indexes the array by id (freddy.c:309,367,873,908)
  std::vector<float> coarse((size_t)C * d);
  std::vector<char> seen((size_t)C, 0);
  for (int i = 0; i < C; ++i) {
    const int id = coarse_ids_tbl[i];
    if (id < 0 || id >= C || seen[id]) return fail(-1, "coarse_quantization ids must be exactly 0..%d", C - 1);
    seen[id] = 1;
    memcpy(&coarse[(size_t)id * d], coarse_vectors + (size_t)i * d, sizeof(float) * d);
  }
  // "ORDER BY coarse_id, id": inverted lists with ascending ids inside
  std::vector<int64_t> ord((size_t)N);
  std::iota(ord.begin(), ord.end(), 0);
  for (int64_t i = 0; i < N; ++i)
    if (coarse_id[i] < 0 || coarse_id[i] >= C) return fail(-1, "coarse_id %d out of range at row %lld", coarse_id[i], (long long)i);
  std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) {
    return coarse_id[a] != coarse_id[b] ? coarse_id[a] < coarse_id[b] : ids[a] < ids[b];
  });
  std::vector<int32_t> list_off((size_t)C + 1, 0), sid((size_t)N);
  std::vector<int16_t> scodes((size_t)N * cb.m);
  for (int64_t i = 0; i < N; ++i) {
    list_off[coarse_id[ord[i]] + 1]++;
    sid[i] = ids[ord[i]];
    memcpy(&scodes[(size_t)i * cb.m], codes + (size_t)ord[i] * cb.m, sizeof(int16_t) * cb.m);
  }
  for (int c = 0; c < C; ++c) list_off[c + 1] += list_off[c];
  freddy_ivf_desc desc = {d, cb.m, cb.K, C, N, coarse.data(), cb.dense.data(), list_off.data(), sid.data(), scodes.data()};
  if (s->ivf) { freddy_gpu_unpin(s->ivf); s->ivf = nullptr; }
  if (int rc = freddy_gpu_pin_ivf(&desc, s->device, &s->ivf)) return gpu_fail(rc);
  s->ivf_d = d;
  s->res_cb = cb;
  s->res_counts.assign((size_t)cb.m * cb.K, 1);
  s->coarse = coarse;
  s->C = C;
  s->fine_max_id = 0;
  for (int64_t i = 0; i < N; ++i) s->fine_max_id = std::max(s->fine_max_id, ids[i]);
  return 0;
}

int freddy_load_ivpq(freddy_session_t* s, const int32_t* cb_pos, const int32_t* cb_code, const float* cb_vectors,
                     int32_t n_entries, int32_t sub_dim, const int32_t* cq_pos, const int32_t* cq_code,
                     const float* cq_vectors, int32_t n_cq_entries, const int32_t* ids, const int32_t* coarse_id,
                     const int16_t* codes, int64_t N, const int32_t* stat_coarse_id, const float* stat_freq,
                     int32_t n_stat) {
  if (!s || !ids || !coarse_id || !codes || !stat_coarse_id || !stat_freq) return fail(-1, "bad argument");
  Codebook cb;
  if (int rc = cb.build(cb_pos, cb_code, cb_vectors, n_entries, sub_dim)) return rc;
  const int d = cb.m * cb.s;
  int cpos = 0;
  for (int i = 0; i < n_cq_entries; ++i) cpos = std::max(cpos, cq_pos[i] + 1);
  if (cpos != 2) return fail(-1, "the multi-index coarse quantizer must have 2 positions (index_utils.c:322)");
  Codebook cq;
  if (int rc = cq.build(cq_pos, cq_code, cq_vectors, n_cq_entries, d / 2)) return rc;
  const int cells = cq.K * cq.K;
  // getStatistics (index_utils.c:632-665): result[coarse_id] = coarse_freq
  if (n_stat != cells + 1) return fail(-1, "statistics table has %d rows, expected %d", n_stat, cells + 1);
  std::vector<float> stats((size_t)cells + 1, 0.0f);
  for (int i = 0; i < n_stat; ++i) {
    if (stat_coarse_id[i] < 0 || stat_coarse_id[i] > cells) return fail(-1, "statistics coarse_id out of range");
    stats[stat_coarse_id[i]] = stat_freq[i];
  }
  std::vector<int64_t> ord = order_by_id(ids, N);
  std::vector<int32_t> sid((size_t)N), scell((size_t)N);
  std::vector<int16_t> scodes((size_t)N * cb.m);
  std::vector<float> svec;
  const bool have_vecs = !s->norm_ids.empty() && s->d == d;
  if (have_vecs) svec.resize((size_t)N * d);
  for (int64_t i = 0; i < N; ++i) {
    const int64_t r = ord[i];
    sid[i] = ids[r];
    scell[i] = coarse_id[r];
    memcpy(&scodes[(size_t)i * cb.m], codes + (size_t)r * cb.m, sizeof(int16_t) * cb.m);
    if (have_vecs) {   // "INNER JOIN vecs ON fq.id = vecs.id" (ivpq_search_in.c:361-371)
      auto it = std::lower_bound(s->norm_ids.begin(), s->norm_ids.end(), ids[r]);
      if (it == s->norm_ids.end() || *it != ids[r]) return fail(-1, "id %d of fine_quantization_ivpq has no vector", ids[r]);
      memcpy(&svec[(size_t)i * d], &s->norm_vecs[(size_t)(it - s->norm_ids.begin()) * d], sizeof(float) * d);
    }
  }
  freddy_ivpq_desc desc = {d, cb.m, cb.K, 2, cq.K, N, cb.dense.data(), cq.dense.data(), sid.data(), scell.data(),
                           scodes.data(), have_vecs ? svec.data() : nullptr, stats.data()};
  if (s->ivpq) { freddy_gpu_unpin(s->ivpq); s->ivpq = nullptr; }
  if (int rc = freddy_gpu_pin_ivpq(&desc, s->device, &s->ivpq)) return gpu_fail(rc);
  s->ivpq_d = d;
  s->ivpq_cb = cb;
  s->ivpq_counts.assign((size_t)cb.m * cb.K, 1);
  s->cq_multi = cq;
  s->ivpq_max_id = N ? sid[(size_t)N - 1] : 0;
  return 0;
}

// ---- config functions ------------------------------------------------------------------------
#define SETTER(name, field, type) \
  int freddy_set_##name(freddy_session_t* s, type v) { if (!s) return fail(-1, "NULL session"); s->field = v; return 0; }
SETTER(w, w, int32_t)
SETTER(pvf, pvf, int32_t)
SETTER(alpha, alpha, int32_t)
SETTER(confidence_value, confidence, float)
SETTER(long_codes_threshold, long_codes_threshold, int32_t)
SETTER(method_flag, method_flag, int32_t)
SETTER(use_targetlist, use_targetlist, int32_t)
#undef SETTER
int32_t freddy_get_w(const freddy_session_t* s) { return s->w; }
int32_t freddy_get_pvf(const freddy_session_t* s) { return s->pvf; }
int32_t freddy_get_alpha(const freddy_session_t* s) { return s->alpha; }
float freddy_get_confidence_value(const freddy_session_t* s) { return s->confidence; }
int32_t freddy_get_long_codes_threshold(const freddy_session_t* s) { return s->long_codes_threshold; }
int32_t freddy_get_method_flag(const freddy_session_t* s) { return s->method_flag; }
int32_t freddy_get_use_targetlist(const freddy_session_t* s) { return s->use_targetlist; }

// ---- UDFs ------------------------------------------------------------------------------------
static int emit2(const std::vector<int32_t>& ids, const std::vector<float>& dist, int k, freddy_row2* out, int32_t* n_rows) {
  for (int i = 0; i < k; ++i) { out[i].id = ids[i]; out[i].distance = dist[i]; }
  if (n_rows) *n_rows = k;   // the SRF returns k rows, unfilled ones are (-1, sentinel)  freddy.c:154-169
  return 0;
}

int pq_search(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows) {
  if (!s || !s->pq) return fail(-1, "pq_quantization / pq_codebook are not loaded");
  if (dim != s->pq_d) return fail(-1, "query has %d dimensions, index has %d", dim, s->pq_d);
  if (k <= 0 || !query || !out) return fail(-1, "bad argument");
  std::vector<int32_t> ids((size_t)k); std::vector<float> dist((size_t)k);
  if (int rc = freddy_gpu_pq_search(s->pq, query, 1, k, 100.0f, nullptr, 0, ids.data(), dist.data())) return gpu_fail(rc);
  return emit2(ids, dist, k, out, n_rows);
}

int pq_search_in(freddy_session_t* s, const float* query, int32_t dim, int32_t k, const int32_t* input_ids,
                 int32_t n_ids, freddy_row2* out, int32_t* n_rows) {
  if (!s || !s->pq) return fail(-1, "pq_quantization / pq_codebook are not loaded");
  if (dim != s->pq_d) return fail(-1, "query has %d dimensions, index has %d", dim, s->pq_d);
  if (k <= 0 || !query || !out || n_ids < 0) return fail(-1, "bad argument");
  std::vector<int32_t> ids((size_t)k); std::vector<float> dist((size_t)k);
  const int32_t none = -1;   // empty IN-list: nothing matches
  if (int rc = freddy_gpu_pq_search(s->pq, query, 1, k, 1000.0f, n_ids ? input_ids : &none, n_ids ? n_ids : 1, ids.data(),
                                    dist.data()))
    return gpu_fail(rc);
  return emit2(ids, dist, k, out, n_rows);
}

int ivfadc_search(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows) {
  if (!s || !s->ivf) return fail(-1, "coarse_quantization / residual_codebook / fine_quantization are not loaded");
  if (dim != s->ivf_d) return fail(-1, "query has %d dimensions, index has %d", dim, s->ivf_d);
  if (k <= 0 || !query || !out) return fail(-1, "bad argument");
  std::vector<int32_t> ids((size_t)k); std::vector<float> dist((size_t)k);
  if (int rc = freddy_gpu_ivfadc_search(s->ivf, query, 1, k, s->w, 1000.0f, FREDDY_FOUND_ROWS, ids.data(), dist.data()))
    return gpu_fail(rc);
  return emit2(ids, dist, k, out, n_rows);
}

static int emit3(const int32_t* qids, int Q, int k, const std::vector<int32_t>& ids, const std::vector<float>& dist,
                 freddy_row3* out, int32_t* n_rows) {
  // iter / k -> query, iter % k -> rank   (freddy.c:654-674, 1001-1023; ivpq_search_in.c:700-720)
  for (int i = 0; i < Q * k; ++i) { out[i].query_id = qids[i / k]; out[i].id = ids[i]; out[i].distance = dist[i]; }
  if (n_rows) *n_rows = Q * k;
  return 0;
}

int pq_search_in_batch(freddy_session_t* s, const float* queries, int32_t n_queries, int32_t dim,
                       const int32_t* query_ids, int32_t n_query_ids, int32_t k, const int32_t* input_ids,
                       int32_t n_ids, int32_t use_target_lists, freddy_row3* out, int32_t* n_rows) {
  (void)use_target_lists;   // both branches of the reference give identical lists (freddy.c:596-628)
  if (!s || !s->pq) return fail(-1, "pq_quantization / pq_codebook are not loaded");
  if (n_query_ids != n_queries) return fail(-1, "Number of query vectors and query vector ids differs!");   // freddy.c:495
  if (dim != s->pq_d) return fail(-1, "query has %d dimensions, index has %d", dim, s->pq_d);
  if (k <= 0 || n_queries < 0 || n_ids < 0) return fail(-1, "bad argument");
  std::vector<int32_t> ids((size_t)n_queries * k); std::vector<float> dist((size_t)n_queries * k);
  const int32_t none = -1;
  if (int rc = freddy_gpu_pq_search(s->pq, queries, n_queries, k, 1000.0f, n_ids ? input_ids : &none, n_ids ? n_ids : 1,
                                    ids.data(), dist.data()))
    return gpu_fail(rc);
  return emit3(query_ids, n_queries, k, ids, dist, out, n_rows);
}

int ivfadc_batch_search(freddy_session_t* s, const int32_t* query_ids, int32_t n_query_ids, int32_t k, freddy_row3* out,
                        int32_t* n_rows) {
  if (!s || !s->ivf) return fail(-1, "coarse_quantization / residual_codebook / fine_quantization are not loaded");
  if (s->norm_ids.empty() || s->d != s->ivf_d) return fail(-1, "google_vecs_norm is not loaded");
  if (k <= 0 || n_query_ids < 0 || !out) return fail(-1, "bad argument");
  // "SELECT id, vector FROM <norm> WHERE id IN (...)": table order, duplicates collapse,
  // unknown ids vanish (freddy.c:767-804)
  std::vector<int32_t> rows;
  for (int i = 0; i < n_query_ids; ++i) {
    auto it = std::lower_bound(s->norm_ids.begin(), s->norm_ids.end(), query_ids[i]);
    if (it != s->norm_ids.end() && *it == query_ids[i]) rows.push_back((int32_t)(it - s->norm_ids.begin()));
  }
  std::sort(rows.begin(), rows.end());
  rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
  const int Q = (int)rows.size();
  const int d = s->d;
  std::vector<float> qv((size_t)Q * d);
  std::vector<int32_t> qid((size_t)Q);
  for (int i = 0; i < Q; ++i) {
    qid[i] = s->norm_ids[rows[i]];
    memcpy(&qv[(size_t)i * d], &s->norm_vecs[(size_t)rows[i] * d], sizeof(float) * d);
  }
  std::vector<int32_t> ids((size_t)Q * k); std::vector<float> dist((size_t)Q * k);
  if (Q > 0)
    if (int rc = freddy_gpu_ivfadc_search(s->ivf, qv.data(), Q, k, 1, 100.0f, FREDDY_FOUND_BATCH_UDF, ids.data(), dist.data()))
      return gpu_fail(rc);
  return emit3(qid.data(), Q, k, ids, dist, out, n_rows);
}

int ivpq_search_in(freddy_session_t* s, const float* queries, int32_t n_queries, int32_t dim, const int32_t* query_ids,
                   int32_t n_query_ids, int32_t k, const int32_t* input_ids, int32_t n_ids, int32_t alpha, int32_t pvf,
                   int32_t method, int32_t use_target_lists, float confidence, int32_t double_threshold,
                   freddy_row3* out, int32_t* n_rows) {
  if (!s || !s->ivpq) return fail(-1, "the ivpq tables are not loaded");
  if (n_query_ids != n_queries)   // ivpq_search_in.c:180
    return fail(-1, "Number of query vectors and query vector ids differs! ( %d, %d)", n_query_ids, n_queries);
  if (dim != s->ivpq_d) return fail(-1, "query has %d dimensions, index has %d", dim, s->ivpq_d);
  if (k <= 0 || n_queries < 0 || n_ids < 0) return fail(-1, "bad argument");
  std::vector<int32_t> ids((size_t)n_queries * k); std::vector<float> dist((size_t)n_queries * k);
  if (int rc = freddy_gpu_knn_join(s->ivpq, queries, n_queries, k, input_ids, n_ids, alpha, pvf, method, use_target_lists,
                                   confidence, double_threshold, ids.data(), dist.data(), nullptr))
    return gpu_fail(rc);
  return emit3(query_ids, n_queries, k, ids, dist, out, n_rows);
}

int knn_join(freddy_session_t* s, const float* queries, int32_t n_queries, int32_t dim, const int32_t* query_ids,
             int32_t k, const int32_t* input_ids, int32_t n_ids, freddy_row3* out, int32_t* n_rows) {
  if (!s) return fail(-1, "NULL session");
  return ivpq_search_in(s, queries, n_queries, dim, query_ids, n_queries, k, input_ids, n_ids, s->alpha, s->pvf,
                        s->method_flag, s->use_targetlist, s->confidence, s->long_codes_threshold, out, n_rows);
}

// ---- exact brute force -------------------------------------------------------------
// google_vecs_norm pinned as raw vectors, on first use
static int ensure_vecs(freddy_session_t* s) {
  if (s->vecs) return 0;
  freddy_vec_desc desc = {s->d, (int64_t)s->norm_ids.size(), s->norm_ids.data(), s->norm_vecs.data()};
  if (int rc = freddy_gpu_pin_vectors(&desc, s->device, &s->vecs)) return gpu_fail(rc);
  return 0;
}

static int exact_common(freddy_session_t* s, const float* query, int32_t dim, int32_t k, const int32_t* input_ids,
                        int32_t n_ids, bool subset, freddy_row2* out, int32_t* n_rows) {
  if (!s || s->norm_ids.empty()) return fail(-1, "google_vecs_norm is not loaded");
  if (dim != s->d) return fail(-1, "query has %d dimensions, table has %d", dim, s->d);
  if (k <= 0 || !query || !out || n_ids < 0) return fail(-1, "bad argument");
  if (int rc = ensure_vecs(s)) return rc;
  std::vector<int32_t> ids((size_t)k); std::vector<float> sim((size_t)k);
  const int32_t none = -1;
  const int32_t* sub = subset ? (n_ids ? input_ids : &none) : nullptr;
  if (int rc = freddy_gpu_exact_search(s->vecs, query, 1, k, sub, subset ? (n_ids ? n_ids : 1) : 0, ids.data(), sim.data()))
    return gpu_fail(rc);
  int n = 0;   // FETCH FIRST k ROWS ONLY returns fewer rows when fewer exist
  while (n < k && ids[n] >= 0) { out[n].id = ids[n]; out[n].distance = sim[n]; ++n; }
  if (n_rows) *n_rows = n;
  return 0;
}

int k_nearest_neighbour(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows) {
  return exact_common(s, query, dim, k, nullptr, 0, false, out, n_rows);
}

int knn_in_exact(freddy_session_t* s, const float* query, int32_t dim, int32_t k, const int32_t* input_ids, int32_t n_ids,
                 freddy_row2* out, int32_t* n_rows) {
  return exact_common(s, query, dim, k, input_ids, n_ids, true, out, n_rows);
}

// ---- index files -----------------------------------------------------------------------------
namespace {
struct FileArray {
  int dtype = 0, ndim = 0;
  int64_t dims[2] = {0, 0};
  std::vector<unsigned char> data;
  int64_t rows() const { return dims[0]; }
  int64_t cols() const { return ndim == 2 ? dims[1] : 1; }
};
const size_t kElem[3] = {4, 4, 2};

int read_index_file(const char* path, std::map<std::string, FileArray>& out) {
  FILE* f = fopen(path, "rb");
  if (!f) return fail(-1, "cannot open index file %s", path);
  auto bad = [&](const char* what) { fclose(f); return fail(-1, "index file %s: %s", path, what); };
  char magic[8];
  uint32_t n = 0;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "FRDYIDX1", 8) != 0) return bad("bad magic");
  if (fread(&n, 4, 1, f) != 1 || n > 4096) return bad("bad array count");
  size_t off = 12;
  auto skip_pad = [&]() { while (off % 8) { if (fgetc(f) == EOF) return false; ++off; } return true; };
  for (uint32_t i = 0; i < n; ++i) {
    uint16_t nl = 0;
    if (fread(&nl, 2, 1, f) != 1 || nl == 0 || nl > 255) return bad("bad name length");
    std::string name(nl, '\0');
    if (fread(&name[0], 1, nl, f) != nl) return bad("truncated name");
    unsigned char dt = 0, nd = 0;
    if (fread(&dt, 1, 1, f) != 1 || fread(&nd, 1, 1, f) != 1 || dt > 2 || nd < 1 || nd > 2) return bad("bad array header");
    FileArray a;
    a.dtype = dt; a.ndim = nd;
    uint64_t dims[2] = {0, 0};
    if (fread(dims, 8, nd, f) != nd) return bad("truncated dims");
    off += 2 + nl + 2 + 8 * nd;
    if (!skip_pad()) return bad("truncated padding");
    a.dims[0] = (int64_t)dims[0]; a.dims[1] = (int64_t)dims[1];
    const uint64_t count = dims[0] * (nd == 2 ? dims[1] : 1);
    if (count > (1ull << 40)) return bad("array too large");
    a.data.resize((size_t)count * kElem[dt]);
    if (!a.data.empty() && fread(a.data.data(), 1, a.data.size(), f) != a.data.size()) return bad("truncated data");
    off += a.data.size();
    if (i + 1 < n && !skip_pad()) return bad("truncated padding");
    out[name] = std::move(a);
  }
  fclose(f);
  return 0;
}
}  // namespace

int freddy_index_file_write(const char* path, const freddy_file_array* arrays, int32_t n_arrays) {
  if (!path || n_arrays < 0 || (n_arrays > 0 && !arrays)) return fail(-1, "bad argument");
  FILE* f = fopen(path, "wb");
  if (!f) return fail(-1, "cannot create index file %s", path);
  size_t off = 0;
  auto put = [&](const void* p, size_t n) { off += n; return n == 0 || fwrite(p, 1, n, f) == n; };
  auto pad = [&]() { static const char z[8] = {0}; const size_t n = (8 - off % 8) % 8; return put(z, n); };
  const uint32_t n = (uint32_t)n_arrays;
  bool ok = put("FRDYIDX1", 8) && put(&n, 4);
  for (int i = 0; ok && i < n_arrays; ++i) {
    const freddy_file_array& a = arrays[i];
    const size_t nl = a.name ? strlen(a.name) : 0;
    if (nl == 0 || nl > 255 || a.dtype < 0 || a.dtype > 2 || a.ndim < 1 || a.ndim > 2 || a.dims[0] < 0 || (a.ndim == 2 && a.dims[1] < 0)) {
      fclose(f);
      return fail(-1, "bad array descriptor %d", i);
    }
    const uint16_t nl16 = (uint16_t)nl;
    const unsigned char dt = (unsigned char)a.dtype, nd = (unsigned char)a.ndim;
    const uint64_t dims[2] = {(uint64_t)a.dims[0], (uint64_t)a.dims[1]};
    const size_t bytes = (size_t)dims[0] * (nd == 2 ? dims[1] : 1) * kElem[dt];
    ok = put(&nl16, 2) && put(a.name, nl) && put(&dt, 1) && put(&nd, 1) && put(dims, 8 * nd) && pad() && put(a.data, bytes) &&
         (i + 1 == n_arrays || pad());
  }
  if (fclose(f) != 0) ok = false;
  return ok ? 0 : fail(-1, "write to index file %s failed", path);
}

int freddy_import_index(freddy_session_t* s, const char* path) {
  if (!s || !path) return fail(-1, "bad argument");
  std::map<std::string, FileArray> t;
  if (int rc = read_index_file(path, t)) return rc;
  auto has = [&](std::initializer_list<const char*> names) {
    for (const char* n : names) if (!t.count(n)) return false;
    return true;
  };
  auto want = [&](const char* name, int dtype, int ndim) -> const FileArray* {
    const FileArray& a = t[name];
    if (a.dtype != dtype || a.ndim != ndim) { fail(-1, "index file %s: array %s has the wrong type or rank", path, name); return nullptr; }
    return &a;
  };
#define ARR(var, name, dtype, ndim) const FileArray* var = want(name, dtype, ndim); if (!var) return -1;
#define I32(a) reinterpret_cast<const int32_t*>((a)->data.data())
#define F32(a) reinterpret_cast<const float*>((a)->data.data())
#define I16(a) reinterpret_cast<const int16_t*>((a)->data.data())
  int loaded = 0;
  if (has({"google_vecs_norm.id", "google_vecs_norm.vector"})) {
    ARR(id, "google_vecs_norm.id", 1, 1) ARR(v, "google_vecs_norm.vector", 0, 2)
    if (v->rows() != id->rows()) return fail(-1, "index file %s: google_vecs_norm columns differ in length", path);
    if (int rc = freddy_load_vecs_norm(s, I32(id), F32(v), id->rows(), (int32_t)v->cols())) return rc;
    ++loaded;
  }
  if (has({"pq_codebook.pos", "pq_codebook.code", "pq_codebook.vector", "pq_quantization.id", "pq_quantization.vector"})) {
    ARR(pos, "pq_codebook.pos", 1, 1) ARR(code, "pq_codebook.code", 1, 1) ARR(vec, "pq_codebook.vector", 0, 2)
    ARR(id, "pq_quantization.id", 1, 1) ARR(q, "pq_quantization.vector", 2, 2)
    if (pos->rows() != code->rows() || pos->rows() != vec->rows() || id->rows() != q->rows())
      return fail(-1, "index file %s: pq tables have columns of different lengths", path);
    if (int rc = freddy_load_pq(s, I32(pos), I32(code), F32(vec), (int32_t)pos->rows(), (int32_t)vec->cols(), I32(id), I16(q), id->rows()))
      return rc;
    if (has({"pq_codebook.count"})) {   // the count column insert_batch updates (index_utils.c:949-956)
      ARR(cnt, "pq_codebook.count", 1, 1)
      if (cnt->rows() == pos->rows()) if (int rc = freddy_set_codebook_counts(s, 0, I32(pos), I32(code), I32(cnt), (int32_t)pos->rows())) return rc;
    }
    ++loaded;
  }
  if (has({"coarse_quantization.id", "coarse_quantization.vector", "residual_codebook.pos", "residual_codebook.code",
           "residual_codebook.vector", "fine_quantization.id", "fine_quantization.coarse_id", "fine_quantization.vector"})) {
    ARR(cid, "coarse_quantization.id", 1, 1) ARR(cv, "coarse_quantization.vector", 0, 2)
    ARR(pos, "residual_codebook.pos", 1, 1) ARR(code, "residual_codebook.code", 1, 1) ARR(vec, "residual_codebook.vector", 0, 2)
    ARR(id, "fine_quantization.id", 1, 1) ARR(co, "fine_quantization.coarse_id", 1, 1) ARR(q, "fine_quantization.vector", 2, 2)
    if (cid->rows() != cv->rows() || pos->rows() != code->rows() || pos->rows() != vec->rows() || id->rows() != co->rows() ||
        id->rows() != q->rows())
      return fail(-1, "index file %s: ivfadc tables have columns of different lengths", path);
    if (int rc = freddy_load_ivfadc(s, I32(cid), F32(cv), (int32_t)cid->rows(), I32(pos), I32(code), F32(vec), (int32_t)pos->rows(),
                                    (int32_t)vec->cols(), I32(id), I32(co), I16(q), id->rows()))
      return rc;
    if (has({"residual_codebook.count"})) {
      ARR(cnt, "residual_codebook.count", 1, 1)
      if (cnt->rows() == pos->rows()) if (int rc = freddy_set_codebook_counts(s, 1, I32(pos), I32(code), I32(cnt), (int32_t)pos->rows())) return rc;
    }
    ++loaded;
  }
  if (has({"codebook_ivpq.pos", "codebook_ivpq.code", "codebook_ivpq.vector", "coarse_quantization_ivpq.pos",
           "coarse_quantization_ivpq.code", "coarse_quantization_ivpq.vector", "fine_quantization_ivpq.id",
           "fine_quantization_ivpq.coarse_id", "fine_quantization_ivpq.vector", "stat.coarse_id", "stat.coarse_freq"})) {
    ARR(pos, "codebook_ivpq.pos", 1, 1) ARR(code, "codebook_ivpq.code", 1, 1) ARR(vec, "codebook_ivpq.vector", 0, 2)
    ARR(qpos, "coarse_quantization_ivpq.pos", 1, 1) ARR(qcode, "coarse_quantization_ivpq.code", 1, 1)
    ARR(qvec, "coarse_quantization_ivpq.vector", 0, 2)
    ARR(id, "fine_quantization_ivpq.id", 1, 1) ARR(co, "fine_quantization_ivpq.coarse_id", 1, 1)
    ARR(q, "fine_quantization_ivpq.vector", 2, 2) ARR(sid, "stat.coarse_id", 1, 1) ARR(sf, "stat.coarse_freq", 0, 1)
    if (pos->rows() != code->rows() || pos->rows() != vec->rows() || qpos->rows() != qcode->rows() || qpos->rows() != qvec->rows() ||
        id->rows() != co->rows() || id->rows() != q->rows() || sid->rows() != sf->rows())
      return fail(-1, "index file %s: ivpq tables have columns of different lengths", path);
    if (int rc = freddy_load_ivpq(s, I32(pos), I32(code), F32(vec), (int32_t)pos->rows(), (int32_t)vec->cols(), I32(qpos), I32(qcode),
                                  F32(qvec), (int32_t)qpos->rows(), I32(id), I32(co), I16(q), id->rows(), I32(sid), F32(sf),
                                  (int32_t)sid->rows()))
      return rc;
    if (has({"codebook_ivpq.count"})) {
      ARR(cnt, "codebook_ivpq.count", 1, 1)
      if (cnt->rows() == pos->rows()) if (int rc = freddy_set_codebook_counts(s, 2, I32(pos), I32(code), I32(cnt), (int32_t)pos->rows())) return rc;
    }
    ++loaded;
  }
#undef ARR
#undef I32
#undef F32
#undef I16
  if (!loaded) return fail(-1, "index file %s holds no complete table group", path);
  return 0;
}

// ---- grouping and analogy on the PQ / IVFADC indexes ---------------------------------------
static const float* norm_vec_of(const freddy_session* s, int32_t id) {
  auto it = std::lower_bound(s->norm_ids.begin(), s->norm_ids.end(), id);
  if (it == s->norm_ids.end() || *it != id) return nullptr;
  return s->norm_vecs.data() + (size_t)(it - s->norm_ids.begin()) * s->d;
}

int grouping_pq(freddy_session_t* s, const int32_t* input_ids, int32_t n_ids, const int32_t* group_ids, int32_t n_groups,
                freddy_group_row* out, int32_t* n_rows) {
  if (!s || !s->pq) return fail(-1, "pq_quantization / pq_codebook are not loaded");
  if (s->norm_ids.empty() || s->d != s->pq_d) return fail(-1, "google_vecs_norm is not loaded");
  if (n_ids < 0 || n_groups <= 0 || !group_ids || !out || (n_ids > 0 && !input_ids)) return fail(-1, "bad argument");
  std::vector<int32_t> groups(group_ids, group_ids + n_groups);
  std::sort(groups.begin(), groups.end());                                       // freddy.c:1241
  std::vector<float> gvec((size_t)n_groups * s->d);
  for (int g = 0; g < n_groups; ++g) {
    // "SELECT id, vector ... WHERE id IN (groups) ORDER BY id ASC" must return one row per group id (:1262-1265)
    const float* v = (g > 0 && groups[g] == groups[g - 1]) ? nullptr : norm_vec_of(s, groups[g]);
    if (!v) return fail(-1, "Group ids do not exist");
    memcpy(&gvec[(size_t)g * s->d], v, sizeof(float) * s->d);
  }
  if (n_rows) *n_rows = 0;
  if (n_ids == 0) return 0;
  std::vector<int32_t> ids((size_t)n_ids), grp((size_t)n_ids);
  int64_t n = 0;
  if (int rc = freddy_gpu_grouping_pq(s->pq, gvec.data(), n_groups, input_ids, n_ids, ids.data(), grp.data(), &n)) return gpu_fail(rc);
  for (int64_t i = 0; i < n; ++i) { out[i].id = ids[i]; out[i].group_id = grp[i] >= 0 ? groups[grp[i]] : -1; }
  if (n_rows) *n_rows = (int32_t)n;
  return 0;
}

// vec_minus_bytea / vec_plus_bytea / vec_normalize_bytea / cosine_similarity_bytea   core_functions.c:118-136,177-195,241-266,67-81
static void vec3cosadd(const float* v1, const float* v2, const float* v3, int d, std::vector<float>& raw, std::vector<float>& unit) {
  raw.resize(d); unit.resize(d);
  for (int i = 0; i < d; ++i) { const float t = v3[i] - v1[i]; raw[i] = t + v2[i]; }
  float sq = 0;
  for (int i = 0; i < d; ++i) { const float p = raw[i] * raw[i]; sq = sq + p; }
  const float length = (float)sqrt((double)sq);
  for (int i = 0; i < d; ++i) unit[i] = raw[i] / length;
}
static float cos_sim_bytea(const float* a, const float* b, int d) {
  float scalar = 0;
  for (int i = 0; i < d; ++i) { const float p = a[i] * b[i]; scalar = scalar + p; }
  return scalar;
}

// analogy_3cosadd_pq / analogy_3cosadd_ivfadc, by row id   freddy--0.0.1.sql:1317-1346, 1428-1460
static int analogy_common(freddy_session_t* s, bool ivf, int32_t id1, int32_t id2, int32_t id3, int32_t* result) {
  if (!s || !result) return fail(-1, "bad argument");
  if (ivf ? !s->ivf : !s->pq) return fail(-1, ivf ? "coarse_quantization / residual_codebook / fine_quantization are not loaded"
                                                  : "pq_quantization / pq_codebook are not loaded");
  if (s->norm_ids.empty() || s->d != (ivf ? s->ivf_d : s->pq_d)) return fail(-1, "google_vecs_norm is not loaded");
  *result = -1;
  const float *v1 = norm_vec_of(s, id1), *v2 = norm_vec_of(s, id2), *v3 = norm_vec_of(s, id3);
  if (!v1 || !v2 || !v3) return 0;            // the cross join over the three words is empty: NULL
  std::vector<float> raw, unit;
  vec3cosadd(v1, v2, v3, s->d, raw, unit);
  const int k = s->pvf + 3;
  std::vector<int32_t> ids((size_t)k); std::vector<float> dist((size_t)k);
  const int rc = ivf ? freddy_gpu_ivfadc_search(s->ivf, unit.data(), 1, k, s->w, 1000.0f, FREDDY_FOUND_ROWS, ids.data(), dist.data())
                     : freddy_gpu_pq_search(s->pq, unit.data(), 1, k, 100.0f, nullptr, 0, ids.data(), dist.data());
  if (rc) return gpu_fail(rc);
  // ... ORDER BY cosine_similarity_bytea(v3 - v1 + v2, v4.vector) DESC FETCH FIRST 1 ROWS ONLY (ties: lowest id)
  bool have = false;
  float best = 0;
  for (int i = 0; i < k; ++i) {
    const int32_t id = ids[i];
    if (id < 0 || id == id1 || id == id2 || id == id3) continue;
    const float* v4 = norm_vec_of(s, id);
    if (!v4) continue;
    const float sim = cos_sim_bytea(raw.data(), v4, s->d);
    if (!have || sim > best || (sim == best && id < *result)) { have = true; best = sim; *result = id; }
  }
  return 0;
}
int analogy_3cosadd_pq(freddy_session_t* s, int32_t id1, int32_t id2, int32_t id3, int32_t* result) {
  return analogy_common(s, false, id1, id2, id3, result);
}
int analogy_3cosadd_ivfadc(freddy_session_t* s, int32_t id1, int32_t id2, int32_t id3, int32_t* result) {
  return analogy_common(s, true, id1, id2, id3, result);
}

// ---- the plpgsql callers of pq_search / ivfadc_search, keyed by row id ------------------
// SRF text round trip of a distance: snprintf("%f") into the tuple, float4in on the way out (freddy.c:164)
static float emitted(float distance) {
  char buf[16];
  snprintf(buf, sizeof buf, "%f", distance);
  return strtof(buf, nullptr);
}
// (1.0 - (distance / 2.0))::float4 -- float4 / numeric is evaluated in float8   freddy--0.0.1.sql:527,617
static float similarity_of(float distance) { return (float)(1.0 - (double)emitted(distance) / 2.0); }

static int knn_plain(freddy_session_t* s, bool ivf, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows) {
  if (!s || (ivf ? !s->ivf : !s->pq)) return fail(-1, ivf ? "coarse_quantization / residual_codebook / fine_quantization are not loaded"
                                                          : "pq_quantization / pq_codebook are not loaded");
  if (dim != (ivf ? s->ivf_d : s->pq_d)) return fail(-1, "query has %d dimensions, index has %d", dim, ivf ? s->ivf_d : s->pq_d);
  if (k <= 0 || !query || !out) return fail(-1, "bad argument");
  std::vector<int32_t> ids((size_t)k); std::vector<float> dist((size_t)k);
  const int rc = ivf ? freddy_gpu_ivfadc_search(s->ivf, query, 1, k, s->w, 1000.0f, FREDDY_FOUND_ROWS, ids.data(), dist.data())
                     : freddy_gpu_pq_search(s->pq, query, 1, k, 100.0f, nullptr, 0, ids.data(), dist.data());
  if (rc) return gpu_fail(rc);
  int n = 0;   // INNER JOIN ... ON idx = id drops the (-1, sentinel) rows
  for (int i = 0; i < k; ++i)
    if (ids[i] >= 0) { out[n].id = ids[i]; out[n].distance = similarity_of(dist[i]); ++n; }
  if (n_rows) *n_rows = n;
  return 0;
}
int k_nearest_neighbour_pq(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows) {
  return knn_plain(s, false, query, dim, k, out, n_rows);
}
int k_nearest_neighbour_ivfadc(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows) {
  return knn_plain(s, true, query, dim, k, out, n_rows);
}

// ... with post verification: get_pvf() * k candidates, ORDER BY cosine_similarity_bytea(q, v.vector) DESC
// FETCH FIRST k ROWS ONLY (equal similarities: ascending id)            freddy--0.0.1.sql:575-591, 625-641
static int knn_pv(freddy_session_t* s, bool ivf, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows) {
  if (!s || (ivf ? !s->ivf : !s->pq)) return fail(-1, ivf ? "coarse_quantization / residual_codebook / fine_quantization are not loaded"
                                                          : "pq_quantization / pq_codebook are not loaded");
  if (s->norm_ids.empty() || s->d != dim) return fail(-1, "google_vecs_norm is not loaded");
  if (dim != (ivf ? s->ivf_d : s->pq_d)) return fail(-1, "query has %d dimensions, index has %d", dim, ivf ? s->ivf_d : s->pq_d);
  if (k <= 0 || !query || !out) return fail(-1, "bad argument");
  const int64_t kc = (int64_t)k * std::max(s->pvf, 1);
  if (kc > 4096) return fail(-1, "pvf * k = %lld exceeds this build's limit of 4096 candidates", (long long)kc);
  std::vector<int32_t> ids((size_t)kc); std::vector<float> dist((size_t)kc);
  const int rc = ivf ? freddy_gpu_ivfadc_search(s->ivf, query, 1, (int)kc, s->w, 1000.0f, FREDDY_FOUND_ROWS, ids.data(), dist.data())
                     : freddy_gpu_pq_search(s->pq, query, 1, (int)kc, 100.0f, nullptr, 0, ids.data(), dist.data());
  if (rc) return gpu_fail(rc);
  std::vector<freddy_row2> cand;
  for (int64_t i = 0; i < kc; ++i) {
    if (ids[i] < 0) continue;
    const float* v = norm_vec_of(s, ids[i]);
    if (!v) continue;
    cand.push_back({ids[i], cos_sim_bytea(query, v, dim)});
  }
  std::sort(cand.begin(), cand.end(), [](const freddy_row2& a, const freddy_row2& b) {
    return a.distance != b.distance ? a.distance > b.distance : a.id < b.id;
  });
  const int n = (int)std::min<size_t>(cand.size(), (size_t)k);
  for (int i = 0; i < n; ++i) out[i] = cand[i];
  if (n_rows) *n_rows = n;
  return 0;
}
int k_nearest_neighbour_pq_pv(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows) {
  return knn_pv(s, false, query, dim, k, out, n_rows);
}
int k_nearest_neighbour_ivfadc_pv(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows) {
  return knn_pv(s, true, query, dim, k, out, n_rows);
}

// knn_in_pq(anyarray, int, int[])                                          freddy--0.0.1.sql:830-843
int knn_in_pq(freddy_session_t* s, const float* query, int32_t dim, int32_t k, const int32_t* input_ids, int32_t n_ids,
              freddy_row2* out, int32_t* n_rows) {
  if (k <= 0 || !out) return fail(-1, "bad argument");
  std::vector<freddy_row2> rows((size_t)k);
  int32_t n = 0;
  if (int rc = pq_search_in(s, query, dim, k, input_ids, n_ids, rows.data(), &n)) return rc;
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (rows[i].id >= 0) { out[m].id = rows[i].id; out[m].distance = similarity_of(rows[i].distance); ++m; }
  if (n_rows) *n_rows = m;
  return 0;
}

// k_nearest_neighbour_ivfadc_batch(varchar[], int), by query ids            freddy--0.0.1.sql:535-553
int k_nearest_neighbour_ivfadc_batch(freddy_session_t* s, const int32_t* query_ids, int32_t n_query_ids, int32_t k,
                                     freddy_row3* out, int32_t* n_rows) {
  if (k <= 0 || !out || n_query_ids < 0) return fail(-1, "bad argument");
  std::vector<freddy_row3> rows((size_t)std::max(n_query_ids, 1) * k);
  int32_t n = 0;
  if (int rc = ivfadc_batch_search(s, query_ids, n_query_ids, k, rows.data(), &n)) return rc;
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (rows[i].id >= 0) { out[m] = rows[i]; out[m].distance = similarity_of(rows[i].distance); ++m; }
  if (n_rows) *n_rows = m;
  return 0;
}

// ---- insert_batch --------------------------------------------------------------------
int freddy_set_codebook_counts(freddy_session_t* s, int32_t table, const int32_t* pos, const int32_t* code, const int32_t* count, int32_t n) {
  if (!s || !pos || !code || !count) return fail(-1, "bad argument");
  Codebook* cb = table == 0 ? &s->pq_cb : table == 1 ? &s->res_cb : table == 2 ? &s->ivpq_cb : nullptr;
  std::vector<int32_t>* cnt = table == 0 ? &s->pq_counts : table == 1 ? &s->res_counts : table == 2 ? &s->ivpq_counts : nullptr;
  if (!cb || cb->m == 0) return fail(-1, "that codebook is not loaded");
  for (int i = 0; i < n; ++i) {
    if (pos[i] < 0 || pos[i] >= cb->m || code[i] < 0 || code[i] >= cb->K) return fail(-1, "(pos, code) out of range");
    (*cnt)[(size_t)pos[i] * cb->K + code[i]] = count[i];
  }
  return 0;
}

// sprintf("%f") -> '{...}'::float4[]: how every float reaches the tables (index_utils.c:976, :1053)
static float text_roundtrip(float v) {
  char buf[64];
  snprintf(buf, sizeof buf, "%f", v);
  return strtof(buf, nullptr);
}

// updateCodebook's bookkeeping + updateCodebookRelation (index_utils.c:940-991) for codes found on the device.
// Statement by statement, slips included: `nearestCentroidRaw` is ONE pointer for all positions -- after the
// scan in table order (position-major) it is the nearest entry of the LAST position --, that vector is what
// every position of the row adds to its bucket, the recalculation reads bucket [pos + code] and adds
// (1.0 / count) * bucket in double, only entries with an increment are written back, as "%f" text.
static void update_codebook_host(Codebook& cb, std::vector<int32_t>& counts, const int16_t* codes, int n) {
  const int m = cb.m, K = cb.K, sdim = cb.s, E = m * K;
  std::vector<float> differences((size_t)E * sdim, 0.0f), work(cb.dense);
  std::vector<int32_t> incs((size_t)E, 0), wcount(counts);
  for (int i = 0; i < n; ++i) {
    const float* nearest_raw = &work[((size_t)(m - 1) * K + codes[(size_t)i * m + (m - 1)]) * sdim];
    for (int j = 0; j < m; ++j) {
      const int code = codes[(size_t)i * m + j];
      incs[(size_t)j * K + code] += 1;
      for (int k = 0; k < sdim; ++k) differences[((size_t)j * K + code) * sdim + k] += nearest_raw[k];
    }
  }
  for (int i = 0; i < E; ++i) {
    const int pos = i / K, code = i % K;
    wcount[(size_t)i] += incs[(size_t)pos * K + code];
    for (int j = 0; j < sdim; ++j)
      work[(size_t)i * sdim + j] += (1.0 / wcount[(size_t)i]) * differences[(size_t)(pos + code) * sdim + j];
  }
  for (int i = 0; i < E; ++i)
    if (incs[(size_t)i] > 0) {
      for (int j = 0; j < sdim; ++j) cb.dense[(size_t)i * sdim + j] = text_roundtrip(work[(size_t)i * sdim + j]);
      counts[(size_t)i] = wcount[(size_t)i];
    }
}

int insert_batch(freddy_session_t* s, const float* norm_vectors, int32_t n, int32_t dim, int32_t* new_ids) {
  if (!s || n < 0 || (n > 0 && !norm_vectors)) return fail(-1, "bad argument");
  if (!s->pq || !s->ivf || !s->ivpq || s->norm_ids.empty())
    return fail(-1, "insert_batch needs google_vecs_norm, the pq, ivfadc and ivpq tables (freddy.c:1467-1481)");
  if (dim != s->d || dim != s->pq_d || dim != s->ivf_d || dim != s->ivpq_d) return fail(-1, "vectors have %d dimensions, the tables %d", dim, s->d);
  if (n == 0) return 0;
  // quantisation of the new vectors on the device (freddy.c:1557-1623)
  freddy_insert_desc desc = {dim, s->pq_cb.m, s->pq_cb.K, s->pq_cb.dense.data(), s->res_cb.m, s->res_cb.K, s->res_cb.dense.data(),
                             s->C, s->coarse.data(), s->ivpq_cb.m, s->ivpq_cb.K, s->ivpq_cb.dense.data(),
                             s->cq_multi.m, s->cq_multi.K, s->cq_multi.dense.data()};
  std::vector<int16_t> pq_codes((size_t)n * s->pq_cb.m), res_codes((size_t)n * s->res_cb.m), iv_codes((size_t)n * s->ivpq_cb.m),
      multi((size_t)n * s->cq_multi.m);
  std::vector<int32_t> cq((size_t)n);
  if (int rc = freddy_gpu_insert_quantize(&desc, s->device, norm_vectors, n, pq_codes.data(), cq.data(), res_codes.data(),
                                          iv_codes.data(), multi.data()))
    return gpu_fail(rc);
  // multi-index cell: factor *= POSITIONS (freddy.c:1599, sic)
  std::vector<int32_t> cq_multi_id((size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    int factor = 1;
    for (int p = 0; p < s->cq_multi.m; ++p) { cq_multi_id[(size_t)i] += factor * multi[(size_t)i * s->cq_multi.m + p]; factor *= s->cq_multi.m; }
  }
  // the three codebooks (index_utils.c:940-991) -- on COPIES: the session's tables change only after every device
  // mutation below has succeeded; a failure midway leaves the host tables as they were and drops the pinned handles
  // (they may hold part of the batch): searches then fail loudly with "not loaded" until the tables are loaded again
  Codebook pq_cb = s->pq_cb, res_cb = s->res_cb, ivpq_cb = s->ivpq_cb;
  std::vector<int32_t> pq_counts = s->pq_counts, res_counts = s->res_counts, ivpq_counts = s->ivpq_counts;
  update_codebook_host(pq_cb, pq_counts, pq_codes.data(), n);
  update_codebook_host(res_cb, res_counts, res_codes.data(), n);
  update_codebook_host(ivpq_cb, ivpq_counts, iv_codes.data(), n);
  // the rows: every INSERT takes (SELECT max(id) + 1 FROM <its table>)   index_utils.c:1003-1021, 1046-1058
  std::vector<int32_t> id_pq((size_t)n), id_fine((size_t)n), id_iv((size_t)n), id_norm((size_t)n);
  std::vector<float> stored((size_t)n * dim);
  for (size_t i = 0; i < stored.size(); ++i) stored[i] = text_roundtrip(norm_vectors[i]);
  for (int i = 0; i < n; ++i) {
    id_pq[(size_t)i] = s->pq_max_id + 1 + i;
    id_fine[(size_t)i] = s->fine_max_id + 1 + i;
    id_iv[(size_t)i] = s->ivpq_max_id + 1 + i;
    id_norm[(size_t)i] = s->norm_ids.back() + 1 + i;
  }
  auto device_side = [&]() -> int {
    if (int rc = freddy_gpu_append_rows(s->pq, n, id_pq.data(), nullptr, pq_codes.data(), nullptr)) return rc;
    if (int rc = freddy_gpu_append_rows(s->ivf, n, id_fine.data(), cq.data(), res_codes.data(), nullptr)) return rc;
    // (a cell id built with factor = positions can exceed codes^2 only if positions > codes; it is stored as the reference stores it)
    if (int rc = freddy_gpu_append_rows(s->ivpq, n, id_iv.data(), cq_multi_id.data(), iv_codes.data(), stored.data())) return rc;
    if (int rc = freddy_gpu_update_codebook(s->pq, pq_cb.dense.data())) return rc;
    if (int rc = freddy_gpu_update_codebook(s->ivf, res_cb.dense.data())) return rc;
    if (int rc = freddy_gpu_update_codebook(s->ivpq, ivpq_cb.dense.data())) return rc;
    if (s->vecs) if (int rc = freddy_gpu_append_rows(s->vecs, n, id_norm.data(), nullptr, nullptr, stored.data())) return rc;
    return 0;
  };
  if (int rc = device_side()) {
    const int code = gpu_fail(rc);     // (the message first: unpinning below may overwrite the library's)
    // the handles may hold part of the batch: drop them, the host tables of the session are unchanged
    if (s->pq) { freddy_gpu_unpin(s->pq); s->pq = nullptr; }
    if (s->ivf) { freddy_gpu_unpin(s->ivf); s->ivf = nullptr; }
    if (s->ivpq) { freddy_gpu_unpin(s->ivpq); s->ivpq = nullptr; }
    if (s->vecs) { freddy_gpu_unpin(s->vecs); s->vecs = nullptr; }
    return code;
  }
  // commit
  s->pq_cb = std::move(pq_cb); s->res_cb = std::move(res_cb); s->ivpq_cb = std::move(ivpq_cb);
  s->pq_counts = std::move(pq_counts); s->res_counts = std::move(res_counts); s->ivpq_counts = std::move(ivpq_counts);
  s->pq_max_id += n; s->fine_max_id += n; s->ivpq_max_id += n;
  s->norm_ids.insert(s->norm_ids.end(), id_norm.begin(), id_norm.end());
  s->norm_vecs.insert(s->norm_vecs.end(), stored.begin(), stored.end());
  if (new_ids) memcpy(new_ids, id_norm.data(), sizeof(int32_t) * (size_t)n);
  return 0;
}

// ---- analogy_3cosadd_in_pq / analogy_3cosadd_in_ivpq      freddy--0.0.1.sql:1348-1426 --------------------
static int analogy_in_common(freddy_session_t* s, bool ivpq, int32_t id1, int32_t id2, int32_t id3, const int32_t* input_ids,
                             int32_t n_ids, int32_t* result) {
  if (!s || !result || n_ids < 0 || (n_ids > 0 && !input_ids)) return fail(-1, "bad argument");
  if (ivpq ? !s->ivpq : !s->pq) return fail(-1, ivpq ? "the ivpq tables are not loaded" : "pq_quantization / pq_codebook are not loaded");
  if (s->norm_ids.empty() || s->d != (ivpq ? s->ivpq_d : s->pq_d)) return fail(-1, "google_vecs_norm is not loaded");
  *result = -1;
  const float *v1 = norm_vec_of(s, id1), *v2 = norm_vec_of(s, id2), *v3 = norm_vec_of(s, id3);
  if (!v1 || !v2 || !v3) return 0;
  std::vector<float> raw, unit;
  vec3cosadd(v1, v2, v3, s->d, raw, unit);
  // pq: pq_search_in(q, get_pvf() + 3, ids of the input set); ivpq: ivpq_search_in(ARRAY[q], '{0}', 4, ids, alpha, pvf,
  // method_flag, use_targetlist, confidence, long_codes_threshold) -- k is the literal 4 there (:1414)
  const int k = ivpq ? 4 : s->pvf + 3;
  std::vector<int32_t> ids((size_t)k); std::vector<float> dist((size_t)k);
  const int32_t none = -1;
  const int rc = ivpq ? freddy_gpu_knn_join(s->ivpq, unit.data(), 1, k, input_ids, n_ids, s->alpha, s->pvf, s->method_flag,
                                            s->use_targetlist, s->confidence, s->long_codes_threshold, ids.data(), dist.data(), nullptr)
                      : freddy_gpu_pq_search(s->pq, unit.data(), 1, k, 1000.0f, n_ids ? input_ids : &none, n_ids ? n_ids : 1,
                                             ids.data(), dist.data());
  if (rc) return gpu_fail(rc);
  bool have = false;
  float best = 0;
  for (int i = 0; i < k; ++i) {
    const int32_t id = ids[i];
    if (id < 0 || id == id1 || id == id2 || id == id3) continue;
    const float* v4 = norm_vec_of(s, id);
    if (!v4) continue;
    const float sim = cos_sim_bytea(raw.data(), v4, s->d);
    if (!have || sim > best || (sim == best && id < *result)) { have = true; best = sim; *result = id; }
  }
  return 0;
}
int analogy_3cosadd_in_pq(freddy_session_t* s, int32_t id1, int32_t id2, int32_t id3, const int32_t* input_ids, int32_t n_ids, int32_t* result) {
  return analogy_in_common(s, false, id1, id2, id3, input_ids, n_ids, result);
}
int analogy_3cosadd_in_ivpq(freddy_session_t* s, int32_t id1, int32_t id2, int32_t id3, const int32_t* input_ids, int32_t n_ids, int32_t* result) {
  return analogy_in_common(s, true, id1, id2, id3, input_ids, n_ids, result);
}

// ---- cluster_exact / cluster_pq / cluster_ivpq = generic_cluster      freddy--0.0.1.sql:1086-1209 --------------
namespace {
struct SimRow { float sim; int qid; int tid; };
}

// (query, target, similarity) rows of knn_search_in_batch / knn_in_pq_batch / knn_in_ivpq_batch (bytea[] overloads:
// query = 1-based centroid index) for k = all tokens; target as 1-based token index
static int cluster_knn(freddy_session_t* s, int method, const std::vector<float>& centroids, int kc, const int32_t* token_ids, int n,
                       std::vector<SimRow>& rows) {
  rows.clear();
  std::vector<int32_t> ids((size_t)kc * n); std::vector<float> val((size_t)kc * n);
  if (method == 0) {          // knn_in_exact per centroid: cosine_similarity_bytea DESC (freddy--0.0.1.sql:456-476, 1041-1054)
    if (int rc = ensure_vecs(s)) return rc;
    if (int rc = freddy_gpu_exact_search(s->vecs, centroids.data(), kc, n, token_ids, n, ids.data(), val.data())) return gpu_fail(rc);
  } else if (method == 1) {   // pq_search_in_batch (:880-902)
    if (!s->pq) return fail(-1, "pq_quantization / pq_codebook are not loaded");
    if (int rc = freddy_gpu_pq_search(s->pq, centroids.data(), kc, n, 1000.0f, token_ids, n, ids.data(), val.data())) return gpu_fail(rc);
  } else {                    // ivpq_search_in through knn_in_iv_batch (:754-795)
    if (!s->ivpq) return fail(-1, "the ivpq tables are not loaded");
    if (int rc = freddy_gpu_knn_join(s->ivpq, centroids.data(), kc, n, token_ids, n, s->alpha, s->pvf, s->method_flag, s->use_targetlist,
                                     s->confidence, s->long_codes_threshold, ids.data(), val.data(), nullptr))
      return gpu_fail(rc);
  }
  // token id -> 1-based token index (INNER JOIN unnest(token_ids, tokens) ON token = target; duplicates: every index)
  std::vector<std::pair<int32_t, int>> by_id((size_t)n);
  for (int i = 0; i < n; ++i) by_id[(size_t)i] = {token_ids[i], i + 1};
  std::sort(by_id.begin(), by_id.end());
  for (int qi = 0; qi < kc; ++qi)
    for (int r = 0; r < n; ++r) {
      const int32_t id = ids[(size_t)qi * n + r];
      if (id < 0) continue;   // the joins drop the (-1, sentinel) filler rows
      const float sim = method == 0 ? val[(size_t)qi * n + r] : similarity_of(val[(size_t)qi * n + r]);
      auto it = std::lower_bound(by_id.begin(), by_id.end(), std::make_pair(id, 0));
      for (; it != by_id.end() && it->first == id; ++it) rows.push_back({sim, qi + 1, it->second});
    }
  // ORDER BY similarity DESC (ties are unspecified in SQL: query, then token position)
  std::stable_sort(rows.begin(), rows.end(), [](const SimRow& a, const SimRow& b) {
    if (a.sim != b.sim) return a.sim > b.sim;
    if (a.qid != b.qid) return a.qid < b.qid;
    return a.tid < b.tid;
  });
  return 0;
}

static int generic_cluster(freddy_session_t* s, int method, const int32_t* token_ids, int32_t n, int32_t k, const double* draws,
                           int32_t n_draws, int32_t* cluster_out) {
  if (!s || !token_ids || n <= 0 || k <= 0 || !cluster_out) return fail(-1, "bad argument");
  if (s->norm_ids.empty()) return fail(-1, "google_vecs_norm is not loaded");
  const int d = s->d;
  int used = 0;
  uint64_t state = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() -> double {   // random(): the caller's sequence first (reproducible runs), then an xorshift
    if (draws && used < n_draws) return draws[used++];
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;
    return (double)(state >> 11) / 9007199254740992.0;
  };
  auto pick = [&](int upper) -> int {   // round(random() * upper + 0.5): 1..upper
    const int v = (int)std::nearbyint(rnd() * upper + 0.5);   // float8 round() is rint()
    return v < 1 ? 1 : (v > upper ? upper : v);
  };
  std::vector<const float*> tok((size_t)n);
  for (int i = 0; i < n; ++i) {
    tok[(size_t)i] = norm_vec_of(s, token_ids[i]);
    if (!tok[(size_t)i]) return fail(-1, "token id %d has no vector", token_ids[i]);
  }
  std::vector<float> centroids((size_t)k * d);
  std::vector<int> clusters((size_t)n, 0), lens((size_t)k, 0);
  std::vector<char> processed((size_t)n, 0);
  for (int I = 0; I < k; ++I) memcpy(&centroids[(size_t)I * d], tok[(size_t)pick(n) - 1], sizeof(float) * d);   // :1107-1112
  std::vector<SimRow> rows;
  for (int J = 1; J <= 10; ++J) {                                                                                  // :1114
    if (int rc = cluster_knn(s, method, centroids, k, token_ids, n, rows)) return rc;
    for (const SimRow& r : rows)                                                                                   // :1122-1128
      if (!processed[(size_t)r.tid - 1]) {
        clusters[(size_t)r.tid - 1] = r.qid;
        lens[(size_t)r.qid - 1] += 1;
        processed[(size_t)r.tid - 1] = 1;
      }
    if (J < 10) {
      for (int I = 1; I <= k; ++I) {                                                                               // :1131-1156
        if (lens[(size_t)I - 1] == 0) {
          for (int t = 0; t < 10; ++t) (void)pick(n);   // ten samples are drawn and thrown away: the centroid keeps its value (:1133-1142)
        } else {
          std::vector<int> members;   // tokens of cluster I, in token order
          for (int i = 0; i < n; ++i) if (clusters[(size_t)i] == I) members.push_back(i);
          const int len = lens[(size_t)I - 1];
          std::vector<const float*> samples;
          for (int t = 0; t < 10; ++t) {   // r.x INNER JOIN x.id: ten draws with replacement (draws beyond the members vanish)
            const int x = pick(len);
            if (x <= (int)members.size()) samples.push_back(tok[(size_t)members[(size_t)x - 1]]);
          }
          if (!samples.empty()) {   // centroid_bytea: output[j] += data[i][j] / (float) n   core_functions.c:371-379
            float* c = &centroids[(size_t)(I - 1) * d];
            for (int j = 0; j < d; ++j) c[j] = 0;
            for (size_t i = 0; i < samples.size(); ++i)
              for (int j = 0; j < d; ++j) c[j] += samples[i][j] / (float)samples.size();
          }
          lens[(size_t)I - 1] = 0;
        }
      }
      std::fill(processed.begin(), processed.end(), 0);                                                            // :1158-1160
    }
  }
  for (int i = 0; i < n; ++i) cluster_out[i] = clusters[(size_t)i];
  return 0;
}
int cluster_exact(freddy_session_t* s, const int32_t* token_ids, int32_t n, int32_t k, const double* draws, int32_t n_draws, int32_t* cluster_out) {
  return generic_cluster(s, 0, token_ids, n, k, draws, n_draws, cluster_out);
}
int cluster_pq(freddy_session_t* s, const int32_t* token_ids, int32_t n, int32_t k, const double* draws, int32_t n_draws, int32_t* cluster_out) {
  return generic_cluster(s, 1, token_ids, n, k, draws, n_draws, cluster_out);
}
int cluster_ivpq(freddy_session_t* s, const int32_t* token_ids, int32_t n, int32_t k, const double* draws, int32_t n_draws, int32_t* cluster_out) {
  return generic_cluster(s, 2, token_ids, n, k, draws, n_draws, cluster_out);
}

void freddy_emit_row2(const freddy_row2* row, char values[2][16]) {
  snprintf(values[0], 16, "%d", row->id);
  snprintf(values[1], 16, "%f", row->distance);
}

void freddy_emit_row3(const freddy_row3* row, char values[3][16]) {
  snprintf(values[0], 16, "%d", row->query_id);
  snprintf(values[1], 16, "%d", row->id);
  snprintf(values[2], 16, "%f", row->distance);
}

}  // extern "C"
