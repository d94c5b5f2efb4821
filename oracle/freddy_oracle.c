/*
 * freddy_oracle.c -- CPU ORACLE (test infrastructure only; see freddy_oracle.h).
 *
 * PARITY UNPINNED: restated from reading the reference, never checked against a run
 * of the reference itself (its sources need PostgreSQL headers that are absent here)
 * nor against upstream golden vectors (upstream has none).
 *
 * Every function cites the reference lines it restates; paths are relative to
 * /root/reference/freddy_extension/.  Compile with:  gcc -O2 -ffp-contract=off -fopenmp
 * (no -march=native, no -ffast-math: mirrors the PGXS default flags of the
 * reference's Makefile, so there is no FMA contraction and no reassociation).
 */
#include "freddy_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FO_MAX_DIST 1000.0f       /* MAX_DIST: freddy.c:184,415 ; ivpq_search_in.c:62 */
#define FO_BATCH_SENTINEL 100.0f  /* freddy.c:90-93,823-827 */
#define FO_TOPK_BATCH_SIZE 200    /* ivpq_search_in.c:63 */

/* ------------------------------------------------------------------------------------
 * a1  squareDistance                                              index_utils.c:500-508
 * result starts at +0, each step: t = a-b (rounded), p = t*t (rounded), r = r+p (rounded)
 * ------------------------------------------------------------------------------------ */
float fo_sqdist(const float* a, const float* b, int n) {
  float acc = 0.0f;
  for (int i = 0; i < n; ++i) {
    float t = a[i] - b[i];
    float p = t * t;
    acc = acc + p;
  }
  return acc;
}

/* a2  getPrecomputedDistances                                     index_utils.c:445-455 */
void fo_lut(float* lut, int m, int K, int s, const float* q, const float* codebook) {
  for (int pos = 0; pos < m; ++pos)
    for (int code = 0; code < K; ++code)
      lut[pos * K + code] = fo_sqdist(q + pos * s, codebook + ((size_t)pos * K + code) * s, s);
}

/* a2 over the reference's entry-list form: the slot comes from the entry's own
 * (pos, code), so the order of the entries is irrelevant      index_utils.c:448-453 */
void fo_lut_entries(float* lut, int n_entries, int K, int s, const float* q,
                    const int32_t* pos, const int32_t* code, const float* vectors) {
  for (int e = 0; e < n_entries; ++e)
    lut[pos[e] * K + code[e]] = fo_sqdist(q + pos[e] * s, vectors + (size_t)e * s, s);
}

/* a3  getPrecomputedDistancesDouble                               index_utils.c:457-475
 * pair table i covers positions (2i, 2i+1); slot = code(2i) + K*code(2i+1); the value
 * is the fp32 sum of the two already-rounded sub-distances.  An odd last position is
 * ignored, as in the reference (cbPositions / 2). */
void fo_lut_double(float* lut2, int m, int K, int s, const float* q, const float* codebook) {
  for (int i = 0; i < m / 2; ++i) {
    int p0 = 2 * i, p1 = 2 * i + 1;
    float* dst = lut2 + (size_t)K * K * i;
    for (int j = 0; j < K * K; ++j) {
      int c0 = j % K, c1 = j / K;
      float d0 = fo_sqdist(q + p0 * s, codebook + ((size_t)p0 * K + c0) * s, s);
      float d1 = fo_sqdist(q + p1 * s, codebook + ((size_t)p1 * K + c1) * s, s);
      dst[c0 + K * c1] = d0 + d1;
    }
  }
}

/* a4  computePQDistanceInt16                                     index_utils.c:1126-1133 */
float fo_adc(const float* lut, const int16_t* codes, int m, int K) {
  float acc = 0.0f;
  for (int l = 0; l < m; ++l) acc = acc + lut[K * l + codes[l]];
  return acc;
}

/* a5  initTopK                                                    index_utils.c:66-72 */
void fo_topk_init(fo_entry* tk, int k, float sentinel) {
  for (int i = 0; i < k; ++i) {
    tk[i].id = -1;
    tk[i].dist = sentinel;
  }
}

/* a5  updateTopK                                                  index_utils.c:19-33
 * Walk from the tail to the first entry that is strictly smaller, insert behind it.
 * A new entry therefore lands BEFORE existing entries of equal distance.
 * (If no guard was applied and dist is larger than every entry the reference writes
 * tk[k]; we return instead -- callers always guard with dist < maxDist.) */
void fo_topk_insert(fo_entry* tk, int k, float dist, int32_t id) {
  int slot = k - 1;
  while (slot >= 0 && !(tk[slot].dist < dist)) --slot;
  ++slot;
  if (slot >= k) return;
  for (int j = k - 2; j >= slot; --j) tk[j + 1] = tk[j];
  tk[slot].id = id;
  tk[slot].dist = dist;
}

/* the caller-side guard used at every call site, e.g.               freddy.c:128-131 */
int fo_offer(fo_entry* tk, int k, float* maxd, float dist, int32_t id) {
  if (dist < *maxd) {
    fo_topk_insert(tk, k, dist, id);
    *maxd = tk[k - 1].dist;
    return 1;
  }
  return 0;
}

/* a10 getConfidenceHyp                                            index_utils.c:673-682
 * mu and sig are float variables; the expressions feeding them are evaluated in
 * double exactly as C's usual arithmetic conversions dictate. */
float fo_confidence_hyp(int expect, int size, float p, int stat_size) {
  if (expect > size) return 0;
  float mu = size * p;
  float sig = sqrt(size * p * (1.0 - p)) * (((float)stat_size - size) / ((float)stat_size - 1.0));
  return 1.0 - 0.5 * (1.0 + erf((((float)expect) - 0.5 - mu) / (sig * sqrt(2))));
}

/* a16 "%f" text round trip of the emitted distance                  freddy.c:164,1016 */
float fo_emit_roundtrip(float dist) {
  char buf[16];
  snprintf(buf, 16, "%f", dist);
  return strtof(buf, NULL);
}

/* ------------------------------------------------------------------------------------ */
/* helpers                                                                              */
/* ------------------------------------------------------------------------------------ */

static void* xmalloc(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) {
    fprintf(stderr, "freddy_oracle: out of memory (%zu bytes)\n", n);
    abort();
  }
  return p;
}

/* lower-bound binary search of id in an ascending id column; -1 if absent */
static int64_t find_row(const int32_t* ids, int64_t n, int32_t id) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if (ids[mid] < id) lo = mid + 1; else hi = mid;
  }
  return (lo < n && ids[lo] == id) ? lo : -1;
}

/* rows of a table selected by "WHERE id IN (...)": ascending row order, de-duplicated,
 * missing ids dropped.  Returns the number of rows, writes a malloc'd array. */
static int64_t rows_for_ids(const int32_t* ids, int64_t n, const int32_t* wanted, int n_wanted,
                            int64_t** rows_out) {
  unsigned char* mark = (unsigned char*)calloc((size_t)(n ? n : 1), 1);
  int64_t cnt = 0;
  for (int i = 0; i < n_wanted; ++i) {
    int64_t r = find_row(ids, n, wanted[i]);
    if (r >= 0 && !mark[r]) { mark[r] = 1; ++cnt; }
  }
  int64_t* rows = (int64_t*)xmalloc(sizeof(int64_t) * (size_t)cnt);
  int64_t w = 0;
  for (int64_t r = 0; r < n; ++r) if (mark[r]) rows[w++] = r;
  free(mark);
  *rows_out = rows;
  return cnt;
}

/* Stable merge sort of (dist,payload) records by dist with the reference comparators
 * cmpTopKEntry / cmpTopKPVEntry (index_utils.c:104-116).  The reference calls qsort();
 * glibc <= 2.36 implements it as a stable merge sort, which is the behaviour pinned
 * here (newer glibc uses introsort and leaves the order of equal keys unspecified). */
typedef struct fo_rec {
  int32_t id;
  float dist;
  int64_t aux;
} fo_rec;

static void stable_sort_recs(fo_rec* a, int n) {
  if (n < 2) return;
  fo_rec* tmp = (fo_rec*)xmalloc(sizeof(fo_rec) * (size_t)n);
  for (int width = 1; width < n; width *= 2) {
    for (int lo = 0; lo < n; lo += 2 * width) {
      int mid = lo + width < n ? lo + width : n;
      int hi = lo + 2 * width < n ? lo + 2 * width : n;
      int i = lo, j = mid, o = lo;
      while (i < mid && j < hi) {
        /* take left unless right is strictly smaller: keeps equal keys in order */
        if (a[j].dist < a[i].dist) tmp[o++] = a[j++]; else tmp[o++] = a[i++];
      }
      while (i < mid) tmp[o++] = a[i++];
      while (j < hi) tmp[o++] = a[j++];
    }
    memcpy(a, tmp, sizeof(fo_rec) * (size_t)n);
  }
  free(tmp);
}

/* ------------------------------------------------------------------------------------
 * a9  pq_search                                                      freddy.c:28-152
 * LUT once (:81-83), every row of pq_quantization in scan order (:112-132),
 * sentinel 100.0 (:88-93).
 * ------------------------------------------------------------------------------------ */
int fo_pq_search(const fo_pq_table* t, const float* q, int k, fo_entry* out) {
  if (!t || !q || !out || k <= 0 || t->m <= 0 || t->d % t->m) return -1;
  int s = t->d / t->m;
  float* lut = (float*)xmalloc(sizeof(float) * (size_t)t->m * t->K);
  fo_lut(lut, t->m, t->K, s, q, t->codebook);
  float maxd = FO_BATCH_SENTINEL;
  fo_topk_init(out, k, FO_BATCH_SENTINEL);
  for (int64_t r = 0; r < t->N; ++r) {
    float dist = fo_adc(lut, t->codes + r * t->m, t->m, t->K);
    fo_offer(out, k, &maxd, dist, t->ids[r]);
  }
  free(lut);
  return 0;
}

/* a9  pq_search_in                                                freddy.c:1028-1157
 * identical, restricted to "id IN (inputIds)" (:1100-1114), sentinel 1000.0 (:1094) */
int fo_pq_search_in(const fo_pq_table* t, const float* q, int k, const int32_t* input_ids,
                    int n_ids, fo_entry* out) {
  if (!t || !q || !out || k <= 0 || t->m <= 0 || t->d % t->m) return -1;
  int s = t->d / t->m;
  float* lut = (float*)xmalloc(sizeof(float) * (size_t)t->m * t->K);
  fo_lut(lut, t->m, t->K, s, q, t->codebook);
  int64_t* rows;
  int64_t n_rows = rows_for_ids(t->ids, t->N, input_ids, n_ids, &rows);
  float maxd = FO_MAX_DIST;
  fo_topk_init(out, k, FO_MAX_DIST);
  for (int64_t i = 0; i < n_rows; ++i) {
    int64_t r = rows[i];
    float dist = fo_adc(lut, t->codes + r * t->m, t->m, t->K);
    fo_offer(out, k, &maxd, dist, t->ids[r]);
  }
  free(rows);
  free(lut);
  return 0;
}

/* a9  pq_search_in_batch                                           freddy.c:414-653
 * one LUT per query (:518-524); without target lists every fetched row is offered to
 * every query as it arrives (:596-605); with target lists the rows are first chained
 * into chunks of TARGET_LISTS_SIZE and each query walks the chain (:607-628).  Both
 * give each query the same candidate order. */
int fo_pq_search_in_batch(const fo_pq_table* t, const float* queries, int Q, int k,
                          const int32_t* input_ids, int n_ids, int use_target_lists,
                          fo_entry* out) {
  if (!t || !queries || !out || k <= 0 || Q < 0 || t->m <= 0 || t->d % t->m) return -1;
  int s = t->d / t->m;
  size_t lut_n = (size_t)t->m * t->K;
  float* luts = (float*)xmalloc(sizeof(float) * lut_n * (size_t)(Q ? Q : 1));
  float* maxd = (float*)xmalloc(sizeof(float) * (size_t)(Q ? Q : 1));
  for (int i = 0; i < Q; ++i) {
    fo_lut(luts + lut_n * i, t->m, t->K, s, queries + (size_t)i * t->d, t->codebook);
    fo_topk_init(out + (size_t)i * k, k, FO_MAX_DIST);
    maxd[i] = FO_MAX_DIST;
  }
  int64_t* rows;
  int64_t n_rows = rows_for_ids(t->ids, t->N, input_ids, n_ids, &rows);
  if (!use_target_lists) {
    for (int64_t x = 0; x < n_rows; ++x) {
      int64_t r = rows[x];
      for (int j = 0; j < Q; ++j) {
        float dist = fo_adc(luts + lut_n * j, t->codes + r * t->m, t->m, t->K);
        fo_offer(out + (size_t)j * k, k, &maxd[j], dist, t->ids[r]);
      }
    }
  } else {
    /* the chained chunks only delay the work; the walk order is the fetch order */
    for (int i = 0; i < Q; ++i) {
      for (int64_t x = 0; x < n_rows; ++x) {
        int64_t r = rows[x];
        float dist = 0.0f;
        for (int l = 0; l < t->m; ++l) dist = dist + luts[lut_n * i + (size_t)t->K * l + t->codes[r * t->m + l]];
        fo_offer(out + (size_t)i * k, k, &maxd[i], dist, t->ids[r]);
      }
    }
  }
  free(rows);
  free(maxd);
  free(luts);
  return 0;
}

/* ------------------------------------------------------------------------------------
 * a7/a8  ivfadc_search                                              freddy.c:174-393
 * ------------------------------------------------------------------------------------ */
int fo_ivfadc_search(const fo_ivf_table* t, const float* q, int k, int W, float sentinel,
                     int found_rule, fo_entry* out) {
  if (!t || !q || !out || k <= 0 || W <= 0 || t->m <= 0 || t->d % t->m) return -1;
  const int d = t->d, m = t->m, K = t->K, C = t->C, s = d / m;
  const size_t lut_n = (size_t)m * K;
  unsigned char* black = (unsigned char*)calloc((size_t)(C ? C : 1), 1);
  fo_entry* sel = (fo_entry*)xmalloc(sizeof(fo_entry) * (size_t)W);
  float* resid = (float*)xmalloc(sizeof(float) * (size_t)d);
  float* luts = (float*)xmalloc(sizeof(float) * lut_n * (size_t)W);
  int32_t* cursor = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)W);
  int32_t* cell = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)W);

  fo_topk_init(out, k, sentinel);                         /* :258-260 */
  float maxd = sentinel;
  long found = 0;
  while (found < k) {                                     /* :262 */
    /* W best not-yet-used cells via updateTopK on cqSelection   :266-283
     * (list sentinel 100.0, running threshold starts at 1000.0) */
    float mind = 1000.0f;
    for (int i = 0; i < W; ++i) { sel[i].id = -1; sel[i].dist = 100.0f; }
    for (int j = 0; j < C; ++j) {
      if (black[j]) continue;
      float dist = fo_sqdist(q, t->coarse + (size_t)j * d, d);
      if (dist < mind) {
        fo_topk_insert(sel, W, dist, j);
        mind = sel[W - 1].dist;
      }
    }
    /* blacklist them :289-293.  The reference also blacklists / dereferences id -1 when
     * fewer than W cells are left (undefined behaviour); we use the valid ones only and
     * stop when none is left. */
    int n_sel = 0;
    for (int i = 0; i < W; ++i)
      if (sel[i].id >= 0) { cell[n_sel++] = sel[i].id; black[sel[i].id] = 1; }
    if (n_sel == 0) break;
    /* residuals and one LUT per probed cell                      :296-314 */
    for (int i = 0; i < n_sel; ++i) {
      const float* c = t->coarse + (size_t)cell[i] * d;
      for (int j = 0; j < d; ++j) resid[j] = q[j] - c[j];
      fo_lut(luts + lut_n * i, m, K, s, resid, t->codebook);
      cursor[i] = t->list_off[cell[i]];
    }
    /* SELECT ... WHERE coarse_id IN (sel)  :324-342 ; canonical order = ascending id
     * over the union of the lists -> n_sel-way merge by id */
    long rows = 0, accepted = 0;
    for (;;) {
      int best = -1;
      for (int i = 0; i < n_sel; ++i) {
        if (cursor[i] >= t->list_off[cell[i] + 1]) continue;
        if (best < 0 || t->ids[cursor[i]] < t->ids[cursor[best]]) best = i;
      }
      if (best < 0) break;
      int32_t r = cursor[best]++;
      ++rows;
      float dist = fo_adc(luts + lut_n * best, t->codes + (size_t)r * m, m, K);   /* :363-368 */
      accepted += fo_offer(out, k, &maxd, dist, t->ids[r]);                     /* :369-372 */
    }
    found += found_rule ? accepted : rows;                                      /* :377 / :971 */
  }
  free(cell); free(cursor); free(luts); free(resid); free(sel); free(black);
  return 0;
}

int fo_ivfadc_search_many(const fo_ivf_table* t, const float* queries, int Q, int k, int W,
                          float sentinel, int found_rule, int n_threads, fo_entry* out) {
  if (!t || !queries || !out || Q < 0) return -1;
  int rc = 0;
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(static) num_threads(n_threads) if (n_threads > 1)
  for (int i = 0; i < Q; ++i) {
    int r = fo_ivfadc_search(t, queries + (size_t)i * t->d, k, W, sentinel, found_rule,
                             out + (size_t)i * k);
    if (r) {
#pragma omp atomic write
      rc = r;
    }
  }
  return rc;
}

/* ------------------------------------------------------------------------------------
 * a6/a8  ivfadc_batch_search                                         freddy.c:679-999
 * ------------------------------------------------------------------------------------ */
int fo_ivfadc_batch_search(const fo_ivf_table* t, const float* queries, int Q, int k,
                           fo_entry* out) {
  if (!t || !queries || !out || k <= 0 || Q < 0 || t->m <= 0 || t->d % t->m) return -1;
  const int d = t->d, m = t->m, K = t->K, C = t->C, s = d / m;
  const size_t lut_n = (size_t)m * K;
  const size_t Qn = (size_t)(Q ? Q : 1);
  int* found = (int*)calloc(Qn, sizeof(int));
  int* cq = (int*)xmalloc(sizeof(int) * Qn);
  float* maxd = (float*)xmalloc(sizeof(float) * Qn);
  unsigned char* black = (unsigned char*)calloc(Qn * (size_t)(C ? C : 1), 1);
  unsigned char* stuck = (unsigned char*)calloc(Qn, 1);
  float* luts = (float*)xmalloc(sizeof(float) * lut_n * Qn);
  float* resid = (float*)xmalloc(sizeof(float) * (size_t)d);
  int* count = (int*)xmalloc(sizeof(int) * (size_t)(C ? C : 1));
  int** table = (int**)xmalloc(sizeof(int*) * (size_t)(C ? C : 1));

  for (int i = 0; i < Q; ++i) {                          /* :812-828 */
    fo_topk_init(out + (size_t)i * k, k, FO_BATCH_SENTINEL);
    cq[i] = -1;
    maxd[i] = 100;
  }
  int finished = 0;
  while (!finished) {                                     /* :835 */
    for (int c = 0; c < C; ++c) { count[c] = 0; table[c] = NULL; }
    for (int i = 0; i < Q; ++i) {                         /* :845-888 */
      if (found[i] >= k || stuck[i]) continue;
      float mind = 1000;
      int pick = -1;
      for (int j = 0; j < C; ++j) {                       /* argmin, strict <  :855-866 */
        if (black[(size_t)i * C + j]) continue;
        float dist = fo_sqdist(queries + (size_t)i * d, t->coarse + (size_t)j * d, d);
        if (dist < mind) { mind = dist; pick = j; }
      }
      if (pick < 0) {
        /* every cell already used: the reference would spin forever re-using its last
         * cell; we retire the query with what it has. */
        stuck[i] = 1;
        continue;
      }
      cq[i] = pick;
      black[(size_t)i * C + pick] = 1;                    /* :868-872 */
      count[pick] += 1;                                   /* :873 */
      const float* c = t->coarse + (size_t)pick * d;      /* residual :876-879 */
      for (int j = 0; j < d; ++j) resid[j] = queries[(size_t)i * d + j] - c[j];
      fo_lut(luts + lut_n * i, m, K, s, resid, t->codebook);   /* :882-886 */
    }
    /* cell -> queries table, filled through the reference's "first zero slot" walk
     * (:895-912): query index 0 is indistinguishable from an empty slot, so it ends up
     * in the last slot of its cell; every query of the cell is still visited once. */
    for (int c = 0; c < C; ++c)
      if (count[c] > 0) table[c] = (int*)calloc((size_t)count[c], sizeof(int));
    for (int i = 0; i < Q; ++i) {
      if (found[i] >= k || stuck[i]) continue;
      int j = 0;
      while (table[cq[i]][j]) ++j;
      table[cq[i]][j] = i;
    }
    /* rows of all probed cells (:915-935); a query sits in exactly one cell per round,
     * so its candidate order is ascending id inside that cell however the cells
     * interleave -- we walk cell by cell. */
    for (int c = 0; c < C; ++c) {
      if (count[c] == 0) continue;
      for (int32_t r = t->list_off[c]; r < t->list_off[c + 1]; ++r) {
        const int16_t* codes = t->codes + (size_t)r * m;
        for (int j = 0; j < count[c]; ++j) {              /* :955-973 */
          int qi = table[c][j];
          float dist = 0;
          for (int l = 0; l < m; ++l) dist += luts[lut_n * qi + (size_t)l * K + codes[l]];
          if (dist < maxd[qi]) {
            fo_topk_insert(out + (size_t)qi * k, k, dist, t->ids[r]);
            maxd[qi] = out[(size_t)qi * k + k - 1].dist;
            found[qi]++;
          }
        }
      }
      free(table[c]);
    }
    finished = 1;                                         /* :977-981 */
    for (int i = 0; i < Q; ++i)
      if (found[i] < k && !stuck[i]) finished = 0;
  }
  free(table); free(count); free(resid); free(luts); free(stuck); free(black);
  free(maxd); free(cq); free(found);
  return 0;
}

/* ------------------------------------------------------------------------------------
 * a10  determineCoarseIdsMultiWithStatisticsMulti                 index_utils.c:252-443
 * (USE_PROPERTY_QUEUE branch, two positions only, as the reference states at :322)
 * ------------------------------------------------------------------------------------ */
typedef struct fo_qnode {
  float key;
  int cell;
  int p0, p1;
} fo_qnode;

/* push                                                            index_utils.c:118-131 */
static void heap_push(fo_qnode* h, int* len, fo_qnode nd) {
  int i = *len;
  int parent = (i - 1) / 2;
  while (i > 0 && h[parent].key > nd.key) {
    h[i] = h[parent];
    i = parent;
    parent = (parent - 1) / 2;
  }
  h[i] = nd;
  ++*len;
}

/* pop                                                             index_utils.c:133-155
 * The former last node is re-seated from the root downwards; it stays readable at
 * h[len] during the walk, which is what the child comparisons are made against. */
static fo_qnode heap_pop(fo_qnode* h, int* len) {
  fo_qnode top = h[0];
  h[0] = h[*len - 1];
  --*len;
  int n = *len;
  int i = 0;
  while (i != n) {
    int pick = n;
    int child = 1 + 2 * i;
    if (child <= n - 1 && h[child].key < h[pick].key) pick = child;
    if (child <= n - 1 && h[child + 1].key < h[pick].key) pick = child + 1;
    h[i] = h[pick];
    i = pick;
  }
  return top;
}

int fo_multi_index_select(const fo_ivpq_table* t, const float* queries, const int32_t* active,
                          int n_active, int n_targets, int min_target_count, float confidence,
                          int32_t* cells_out, int32_t* counts_out) {
  if (!t || t->cpos != 2) return -1;
  const int Kc = t->ccodes, cells = Kc * Kc, sub = t->d / t->cpos;
  fo_rec* side[2];
  side[0] = (fo_rec*)xmalloc(sizeof(fo_rec) * (size_t)Kc);
  side[1] = (fo_rec*)xmalloc(sizeof(fo_rec) * (size_t)Kc);
  float* cell_dist = (float*)xmalloc(sizeof(float) * (size_t)cells);
  uint32_t* traversed = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)(cells / 32 + 1));
  uint32_t* queued = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)(cells / 32 + 1));
  fo_qnode* heap = (fo_qnode*)xmalloc(sizeof(fo_qnode) * (size_t)(cells + 1));
  int last_iteration = 1;

  for (int x = 0; x < n_active; ++x) {
    const float* q = queries + (size_t)active[x] * t->d;
    int32_t* emitted = cells_out + (size_t)x * cells;
    int n_emitted = 0;
    float prob = 0.0f;
    /* sub-distances per position                                   :297-305 */
    for (int pos = 0; pos < 2; ++pos)
      for (int j = 0; j < Kc; ++j) {
        side[pos][j].id = j;
        side[pos][j].aux = 0;
        side[pos][j].dist = fo_sqdist(q + pos * sub, t->coarse + ((size_t)pos * Kc + j) * sub, sub);
      }
    /* distance of every cell = 0 + D0[c0] + D1[c1]                 :306-313 */
    for (int c = 0; c < cells; ++c) {
      float acc = 0;
      acc += side[0][c % Kc].dist;
      acc += side[1][c / Kc].dist;
      cell_dist[c] = acc;
    }
    stable_sort_recs(side[0], Kc);                                /* :317-319 */
    stable_sort_recs(side[1], Kc);
    memset(traversed, 0, sizeof(uint32_t) * (size_t)(cells / 32 + 1));   /* :333-338 */
    memset(queued, 0, sizeof(uint32_t) * (size_t)(cells / 32 + 1));
    int len = 0;
    {                                                             /* :343-349 */
      fo_qnode first;
      first.p0 = 0; first.p1 = 0;
      first.cell = side[0][0].id + Kc * side[1][0].id;
      first.key = cell_dist[first.cell];
      heap[0] = first;
      len = 1;
    }
    while (fo_confidence_hyp(min_target_count, n_targets, prob, (int)t->stats[cells]) < confidence &&
           n_emitted < cells) {                                   /* :350-352 */
      fo_qnode cur = heap_pop(heap, &len);
      int here = cur.p0 + Kc * cur.p1;
      traversed[here / 32] |= 1u << (here % 32);
      /* neighbour (p0+1, p1): allowed once (p0+1, p1-1) was traversed  :357-374 */
      int diag = cur.p0 + 1 + Kc * (cur.p1 - 1);
      if (cur.p0 < Kc - 1 && (cur.p1 == 0 || (traversed[diag / 32] & (1u << (diag % 32))))) {
        int np0 = cur.p0 + 1, np1 = cur.p1, npi = np0 + Kc * np1;
        if (!(queued[npi / 32] & (1u << (npi % 32)))) {
          fo_qnode nd;
          nd.p0 = np0; nd.p1 = np1;
          nd.cell = side[0][np0].id + Kc * side[1][np1].id;
          nd.key = cell_dist[nd.cell];
          heap_push(heap, &len, nd);
          queued[npi / 32] |= 1u << (npi % 32);
        }
      }
      /* neighbour (p0, p1+1): allowed once (p0-1, p1+1) was traversed  :375-393 */
      diag = cur.p0 - 1 + Kc * (cur.p1 + 1);
      if (cur.p1 < Kc - 1 && (cur.p0 == 0 || (traversed[diag / 32] & (1u << (diag % 32))))) {
        int np0 = cur.p0, np1 = cur.p1 + 1, npi = np0 + Kc * np1;
        if (!(queued[npi / 32] & (1u << (npi % 32)))) {
          fo_qnode nd;
          nd.p0 = np0; nd.p1 = np1;
          nd.cell = side[0][np0].id + Kc * side[1][np1].id;
          nd.key = cell_dist[nd.cell];
          heap_push(heap, &len, nd);
          queued[npi / 32] |= 1u << (npi % 32);
        }
      }
      prob += t->stats[cur.cell];                                 /* :395 */
      emitted[n_emitted++] = cur.cell;                            /* :397-398 */
    }
    if (n_emitted < cells) last_iteration = 0;                    /* :405-407 */
    counts_out[x] = n_emitted;
  }
  free(heap); free(queued); free(traversed); free(cell_dist); free(side[1]); free(side[0]);
  return last_iteration;
}

/* a13 postverify, one query                                       index_utils.c:477-498 */
void fo_postverify(const float* q, int d, int k, int n_cand, const int32_t* cand_ids,
                   const float* const* cand_vecs, float sentinel, fo_entry* tk) {
  float maxd = sentinel;
  for (int j = 0; j < n_cand; ++j) {
    if (cand_ids[j] == -1) continue;
    float dist = fo_sqdist(q, cand_vecs[j], d);
    if (dist < maxd) {
      fo_topk_insert(tk, k, dist, cand_ids[j]);
      maxd = tk[k - 1].dist;
    }
  }
}

/* ------------------------------------------------------------------------------------
 * a11-a14  ivpq_search_in                                        ivpq_search_in.c:61-699
 * ------------------------------------------------------------------------------------ */
typedef struct fo_ilist {
  int64_t* v;
  int n, cap;
} fo_ilist;

static void ilist_push(fo_ilist* l, int64_t x) {
  if (l->n == l->cap) {
    l->cap = l->cap ? 2 * l->cap : 64;
    l->v = (int64_t*)realloc(l->v, sizeof(int64_t) * (size_t)l->cap);
    if (!l->v) abort();
  }
  l->v[l->n++] = x;
}

/* reorderTopKPV                                                   ivpq_search_in.c:40-45 */
static void pv_reorder(fo_rec* buf, int keep, int* fill, float* maxd) {
  stable_sort_recs(buf, *fill);
  *fill = keep;
  *maxd = buf[keep - 1].dist;
}

/* updateTopKPVFast                                                ivpq_search_in.c:47-57 */
static void pv_append(fo_rec* buf, int batch, int keep, int* fill, float* maxd, int32_t id,
                      float dist, int64_t row) {
  buf[*fill].id = id;
  buf[*fill].dist = dist;
  buf[*fill].aux = row;
  ++*fill;
  if (*fill == batch - 1) pv_reorder(buf, keep, fill, maxd);
}

static void pv_init(fo_rec* buf, int n) {                         /* initTopKPV :84-92 */
  for (int i = 0; i < n; ++i) { buf[i].id = -1; buf[i].dist = FO_MAX_DIST; buf[i].aux = -1; }
}

int fo_ivpq_search_in(const fo_ivpq_table* t, const float* queries, int Q, int k,
                      const int32_t* target_ids, int n_targets, int alpha, int pvf, int method,
                      int use_target_lists, float confidence, int double_threshold,
                      fo_entry* out, int* iterations_out) {
  if (!t || !queries || !out || k <= 0 || Q < 0 || t->cpos != 2) return -1;
  if (method < 0 || method > 2) return -2;                         /* :374-376 */
  if ((method == 1 || method == 2) && !t->vectors) return -3;
  const int d = t->d, m = t->m, K = t->K, s = d / m;
  const int cells = t->ccodes * t->ccodes;                          /* :225 */
  const int alpha_original = alpha;                                 /* :206 */
  if (pvf < 1) pvf = 1;                                             /* :207-209 */
  const int keep = k * pvf;
  const int batch = FO_TOPK_BATCH_SIZE + keep;                      /* :243 */
  const size_t Qn = (size_t)(Q ? Q : 1);

  float* maxd = (float*)xmalloc(sizeof(float) * Qn);
  int* target_count = (int*)calloc(Qn, sizeof(int));               /* :239-242 */
  int* fill = (int*)calloc(Qn, sizeof(int));
  fo_rec* pv = NULL;
  for (int i = 0; i < Q; ++i) { fo_topk_init(out + (size_t)i * k, k, FO_MAX_DIST); maxd[i] = FO_MAX_DIST; }
  if (method == 2) {                                                /* :243-259 */
    pv = (fo_rec*)xmalloc(sizeof(fo_rec) * (size_t)batch * Qn);
    pv_init(pv, (int)((size_t)batch * Qn));
  }

  /* LUTs (:261-291): pair tables when alpha*k > double_threshold */
  int double_codes = 0, n_codes = m, code_range = K;
  float* luts = NULL;
  size_t lut_n = 0;
  if (method == 0 || method == 2) {
    double_codes = (alpha * k > double_threshold);
    if (double_codes) {
      if ((long)K * K > 32768) return -4;   /* the reference's int16 pair code would overflow */
      n_codes = m / 2; code_range = K * K;
      lut_n = (size_t)n_codes * code_range;
      luts = (float*)xmalloc(sizeof(float) * lut_n * Qn);
      for (int i = 0; i < Q; ++i) fo_lut_double(luts + lut_n * i, m, K, s, queries + (size_t)i * d, t->codebook);
    } else {
      lut_n = (size_t)m * K;
      luts = (float*)xmalloc(sizeof(float) * lut_n * Qn);
      for (int i = 0; i < Q; ++i) fo_lut(luts + lut_n * i, m, K, s, queries + (size_t)i * d, t->codebook);
    }
  }

  /* "fq.id IN (targets)" resolved once: ascending rows, de-duplicated */
  int64_t* trows;
  int64_t n_trows = rows_for_ids(t->ids, t->N, target_ids, n_targets, &trows);

  int32_t* active = (int32_t*)xmalloc(sizeof(int32_t) * Qn);
  int n_active = Q;
  for (int i = 0; i < Q; ++i) active[i] = i;
  int32_t* sel_cells = (int32_t*)xmalloc(sizeof(int32_t) * Qn * (size_t)cells);
  int32_t* sel_counts = (int32_t*)xmalloc(sizeof(int32_t) * Qn);
  fo_ilist* table = (fo_ilist*)calloc((size_t)cells, sizeof(fo_ilist));
  fo_ilist* tlists = (fo_ilist*)calloc(Qn, sizeof(fo_ilist));
  int16_t* pair_codes = (int16_t*)xmalloc(sizeof(int16_t) * (size_t)(m ? m : 1));
  int iterations = 0;

  while (n_active > 0) {                                            /* :299 */
    ++iterations;
    for (int c = 0; c < cells; ++c) table[c].n = 0;
    if (use_target_lists) for (int i = 0; i < Q; ++i) tlists[i].n = 0;     /* :309-316 */
    int last = fo_multi_index_select(t, queries, active, n_active, n_targets, k * alpha,
                                     confidence, sel_cells, sel_counts);   /* :327-331 */
    for (int x = 0; x < n_active; ++x)
      for (int e = 0; e < sel_counts[x]; ++e)
        ilist_push(&table[sel_cells[(size_t)x * cells + e]], active[x]);     /* :399-403 */

    /* rows = targets whose cell is probed by someone, canonical order  :352-401 */
    for (int64_t x = 0; x < n_trows; ++x) {
      int64_t r = trows[x];
      int cell = t->coarse_id[r];
      if (cell < 0 || cell >= cells || table[cell].n == 0) continue;
      const int16_t* codes = t->codes + (size_t)r * m;
      const int16_t* use_codes = codes;
      if (double_codes) {                                           /* :446-451 */
        for (int l = 0; l < n_codes; ++l) pair_codes[l] = (int16_t)(codes[2 * l] + codes[2 * l + 1] * K);
        use_codes = pair_codes;
      }
      for (int j = 0; j < table[cell].n; ++j) {                     /* :458-543 */
        int qi = (int)table[cell].v[j];
        target_count[qi] += 1;
        if (use_target_lists) { ilist_push(&tlists[qi], r); continue; }
        float dist;
        if (method == 1) {
          dist = fo_sqdist(queries + (size_t)qi * d, t->vectors + (size_t)r * d, d);
          fo_offer(out + (size_t)qi * k, k, &maxd[qi], dist, t->ids[r]);
        } else {
          dist = fo_adc(luts + lut_n * qi, use_codes, n_codes, code_range);
          if (method == 2) {
            if (dist < maxd[qi]) pv_append(pv + (size_t)batch * qi, batch, keep, &fill[qi], &maxd[qi], t->ids[r], dist, r);
          } else {
            fo_offer(out + (size_t)qi * k, k, &maxd[qi], dist, t->ids[r]);
          }
        }
      }
    }
    if (use_target_lists) {                                         /* :546-608 */
      for (int x = 0; x < n_active; ++x) {
        int qi = active[x];
        if (target_count[qi] < k * alpha_original && !last) {       /* :553-557 */
          target_count[qi] = 0;
          continue;
        }
        for (int e = 0; e < tlists[qi].n; ++e) {
          int64_t r = tlists[qi].v[e];
          float dist;
          if (method == 1) {
            dist = fo_sqdist(queries + (size_t)qi * d, t->vectors + (size_t)r * d, d);
            fo_offer(out + (size_t)qi * k, k, &maxd[qi], dist, t->ids[r]);
          } else {
            const int16_t* codes = t->codes + (size_t)r * m;
            const int16_t* use_codes = codes;
            if (double_codes) {
              for (int l = 0; l < n_codes; ++l) pair_codes[l] = (int16_t)(codes[2 * l] + codes[2 * l + 1] * K);
              use_codes = pair_codes;
            }
            dist = fo_adc(luts + lut_n * qi, use_codes, n_codes, code_range);
            if (method == 2) {
              if (dist < maxd[qi]) pv_append(pv + (size_t)batch * qi, batch, keep, &fill[qi], &maxd[qi], t->ids[r], dist, r);
            } else {
              fo_offer(out + (size_t)qi * k, k, &maxd[qi], dist, t->ids[r]);
            }
          }
        }
      }
    }
    if (method == 2) {                                              /* :611-629 */
      for (int x = 0; x < n_active; ++x) {
        int qi = active[x];
        pv_reorder(pv + (size_t)batch * qi, keep, &fill[qi], &maxd[qi]);
      }
      for (int x = 0; x < n_active; ++x) {                          /* postverify */
        int qi = active[x];
        const fo_rec* buf = pv + (size_t)batch * qi;
        float local = FO_MAX_DIST;
        for (int j = 0; j < keep; ++j) {
          if (buf[j].id == -1) continue;
          float dist = fo_sqdist(queries + (size_t)qi * d, t->vectors + (size_t)buf[j].aux * d, d);
          if (dist < local) {
            fo_topk_insert(out + (size_t)qi * k, k, dist, buf[j].id);
            local = out[(size_t)qi * k + k - 1].dist;
          }
        }
      }
    }
    if (!last) {                                                    /* :639-669 */
      int n_next = 0;
      for (int x = 0; x < n_active; ++x) {
        int qi = active[x];
        if (out[(size_t)qi * k + k - 1].dist == FO_MAX_DIST) {
          fo_topk_init(out + (size_t)qi * k, k, FO_MAX_DIST);
          maxd[qi] = FO_MAX_DIST;
          if (method == 2) {
            pv_init(pv + (size_t)batch * qi, batch);
            fill[x] = 0;   /* sic: indexed by the loop position, not by qi  (:654) */
          }
          active[n_next++] = qi;
        }
      }
      n_active = n_next;
    } else {
      n_active = 0;
    }
    alpha += alpha;                                                 /* :680 */
  }
  if (iterations_out) *iterations_out = iterations;

  free(pair_codes);
  for (int i = 0; i < Q; ++i) free(tlists[i].v);
  free(tlists);
  for (int c = 0; c < cells; ++c) free(table[c].v);
  free(table);
  free(sel_counts); free(sel_cells); free(active); free(trows);
  free(luts); free(pv); free(fill); free(target_count); free(maxd);
  return 0;
}

/* ------------------------------------------------------------------------------------
 * next row 8f-1: exact kNN
 * cosine_similarity_bytea                                        core_functions.c:67-81
 * ------------------------------------------------------------------------------------ */
float fo_cosine_similarity_bytea(const float* v1, const float* v2, int n) {
  float scalar = 0;
  for (int i = 0; i < n; ++i) {
    float p = v1[i] * v2[i];
    scalar = scalar + p;
  }
  return scalar;
}

/* ORDER BY similarity DESC (ties: ascending id) FETCH FIRST k   freddy--0.0.1.sql:426-454,991-1084 */
int fo_exact_knn(const float* vectors, const int32_t* ids, int64_t N, int d, const float* q, int k,
                 const int32_t* input_ids, int n_ids, fo_entry* out) {
  int64_t* rows = NULL;
  int64_t n_rows = N;
  if (input_ids) n_rows = rows_for_ids(ids, N, input_ids, n_ids, &rows);
  int n_out = 0;
  for (int64_t x = 0; x < n_rows; ++x) {
    const int64_t r = rows ? rows[x] : x;
    const float sim = fo_cosine_similarity_bytea(q, vectors + (size_t)r * d, d);
    /* insertion into a descending list; rows arrive in ascending id, an equal similarity stays behind */
    int slot = n_out;
    while (slot > 0 && out[slot - 1].dist < sim) --slot;
    if (slot >= k) continue;
    const int last = (n_out < k) ? n_out : k - 1;
    for (int j = last; j > slot; --j) out[j] = out[j - 1];
    out[slot].id = ids[r];
    out[slot].dist = sim;
    if (n_out < k) ++n_out;
  }
  free(rows);
  return n_out;
}

/* ---- next row ------------------------------------------------------------------ */
void fo_vec_minus(const float* a, const float* b, int n, float* out) {   /* core_functions.c:118-136 */
  for (int i = 0; i < n; ++i) out[i] = a[i] - b[i];
}
void fo_vec_plus(const float* a, const float* b, int n, float* out) {    /* core_functions.c:177-195 */
  for (int i = 0; i < n; ++i) out[i] = a[i] + b[i];
}
void fo_vec_normalize(const float* v, int n, float* out) {               /* core_functions.c:241-266 */
  float sq_length = 0;
  float length = 0;
  for (int i = 0; i < n; ++i) {
    const float p = v[i] * v[i];
    sq_length = sq_length + p;
  }
  length = (float)sqrt((double)sq_length);
  for (int i = 0; i < n; ++i) out[i] = v[i] / length;
}

/* grouping_pq   freddy.c:1176-1401 */
int fo_grouping_pq(const fo_pq_table* t, const float* group_vecs, int n_groups, const int32_t* input_ids,
                   int n_ids, int32_t* out_ids, int32_t* out_group) {
  const int m = t->m, K = t->K, s = t->d / t->m;
  float* luts = (float*)malloc(sizeof(float) * (size_t)n_groups * m * K);
  if (!luts && n_groups) return -1;
  for (int g = 0; g < n_groups; ++g)                                     /* :1288-1299 */
    fo_lut(luts + (size_t)g * m * K, m, K, s, group_vecs + (size_t)g * t->d, t->codebook);
  int64_t* rows = NULL;
  const int64_t n_rows = rows_for_ids(t->ids, t->N, input_ids, n_ids, &rows);
  for (int64_t x = 0; x < n_rows; ++x) {                                 /* :1325-1358 */
    const int64_t r = rows[x];
    float minDist = 100;
    int nearest = -1;
    for (int g = 0; g < n_groups; ++g) {
      float distance = 0;
      for (int j = 0; j < m; ++j) distance += luts[(size_t)g * m * K + (size_t)j * K + t->codes[(size_t)r * m + j]];
      if (distance < minDist) { minDist = distance; nearest = g; }
    }
    out_ids[x] = t->ids[r];
    out_group[x] = nearest;
  }
  free(rows);
  free(luts);
  return (int)n_rows;
}

/* ---- next row ------------------------------------------------------------------ */
void fo_encode_pq(const float* codebook, int m, int K, int s, const float* vecs, int64_t n, int16_t* codes) {
  const int d = m * s;
  for (int64_t i = 0; i < n; ++i)
    for (int p = 0; p < m; ++p) {
      float min_dist = 0;
      int code = -1;
      for (int j = 0; j < K; ++j) {                                   /* pq_index.py:78-86 */
        const float dist = fo_sqdist(vecs + (size_t)i * d + (size_t)p * s, codebook + ((size_t)p * K + j) * s, s);
        if (code < 0 || dist < min_dist) { min_dist = dist; code = j; }
      }
      codes[(size_t)i * m + p] = (int16_t)code;
    }
}
void fo_assign_coarse(const float* coarse, int C, int d, const float* vecs, int64_t n, int32_t* cell) {
  for (int64_t i = 0; i < n; ++i) {
    float min_dist = 0;
    int best = -1;
    for (int c = 0; c < C; ++c) {
      const float dist = fo_sqdist(vecs + (size_t)i * d, coarse + (size_t)c * d, d);
      if (best < 0 || dist < min_dist) { min_dist = dist; best = c; }
    }
    cell[i] = best;
  }
}

/* ------------------------------------------------------------------------------------
 * f4  insert_batch                                   freddy.c:1403-1658, index_utils.c:908-1074
 * The reference quantises the NEW vectors against the coarse quantizer, the multi-index coarse
 * quantizer and the three codebooks, nudges the codebook entries that received vectors, and INSERTs
 * the rows; every float goes to the tables as "%f" text.  Restated statement by statement, slips
 * included (they decide what ends up in the tables):
 *   - the nearest centroid of a position is searched from minDist = 100 by strict "<" over the entries in
 *     table order (index_utils.c:925-939); a sub-vector farther than 100 from every entry leaves the code
 *     uninitialised in the reference -> here: return -2;
 *   - `nearestCentroidRaw` is ONE pointer for all positions: after the scan it is the vector of the LAST entry
 *     (in table order) that improved its position's minimum, and that vector -- not the difference to the new
 *     sub-vector -- is what every position of this row adds to its bucket (:940-946);
 *   - the recalculation reads the bucket `differences[pos + code]` (not pos * codes + code) and adds
 *     (1.0 / count) * bucket in double (:949-956);
 *   - only entries with a count increment are written back (updateCodebookRelation :959-991), through
 *     sprintf("%f") and float4 input (:972-983);
 *   - the multi-index cell id is built with factor *= POSITIONS (freddy.c:1599), not codes.
 * Table order = position-major, code-minor (what index_creation/database_export.py inserts).
 * ------------------------------------------------------------------------------------ */
float fo_text_roundtrip(float v) {                      /* sprintf("%f") -> '{...}'::float4[] : index_utils.c:976, :1053 */
  char buf[64];
  snprintf(buf, sizeof buf, "%f", v);
  return strtof(buf, NULL);
}

/* updateCodebook + updateCodebookRelation.  codebook [m][K][s] and counts [m*K] are updated in place (what the
 * table holds afterwards); codes [n][m] = nearestCentroids; count_incs [m*K]. */
/* `order` (NULL: position-major, code-minor): order[j] = pos * K + code of the j-th tuple the SPI scan returns -- the order
 * decides which entry wins a tie and which vector `nearestCentroidRaw` ends up pointing at (:928-939), and insert_batch's own
 * UPDATEs move codebook tuples, so from the second call on the heap order is no longer the export's. */
int fo_update_codebook_ordered(float* codebook, int32_t* counts, int m, int K, int s, const float* vecs, int n,
                               int16_t* codes, int32_t* count_incs, const int32_t* order);
int fo_update_codebook(float* codebook, int32_t* counts, int m, int K, int s, const float* vecs, int n,
                       int16_t* codes, int32_t* count_incs) {
  return fo_update_codebook_ordered(codebook, counts, m, K, s, vecs, n, codes, count_incs, NULL);
}
int fo_update_codebook_ordered(float* codebook, int32_t* counts, int m, int K, int s, const float* vecs, int n,
                               int16_t* codes, int32_t* count_incs, const int32_t* order) {
  const int d = m * s, E = m * K;
  float* differences = (float*)calloc((size_t)E * s, sizeof(float));        /* :913-921 */
  float* min_dist = (float*)xmalloc(sizeof(float) * (size_t)m);
  float* work = (float*)xmalloc(sizeof(float) * (size_t)E * s);             /* the in-memory CodebookWithCounts */
  int32_t* wcount = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)E);
  int* nearest = (int*)xmalloc(sizeof(int) * (size_t)m);
  memcpy(work, codebook, sizeof(float) * (size_t)E * s);
  memcpy(wcount, counts, sizeof(int32_t) * (size_t)E);
  for (int i = 0; i < E; ++i) count_incs[i] = 0;
  int rc = 0;
  for (int i = 0; i < n && !rc; ++i) {                                      /* :923 */
    const float* nearest_raw = NULL;
    for (int j = 0; j < m; ++j) { min_dist[j] = 100; nearest[j] = -1; }     /* :925-927 */
    for (int jj = 0; jj < E; ++jj) {                                         /* :928-939 */
      const int j = order ? order[jj] : jj;                                 /* cb[jj] = the jj-th tuple; its slot */
      const int pos = j / K, code = j % K;
      const float dist = fo_sqdist(vecs + (size_t)i * d + (size_t)pos * s, work + (size_t)j * s, s);
      if (dist < min_dist[pos]) {
        nearest[pos] = code;
        min_dist[pos] = dist;
        nearest_raw = work + (size_t)j * s;
      }
    }
    for (int j = 0; j < m; ++j) {                                           /* :940-946 */
      if (nearest[j] < 0) { rc = -2; break; }
      codes[(size_t)i * m + j] = (int16_t)nearest[j];
      count_incs[j * K + nearest[j]] += 1;
      for (int k = 0; k < s; ++k) differences[((size_t)j * K + nearest[j]) * s + k] += nearest_raw[k];
    }
  }
  if (!rc) {
    for (int i = 0; i < E; ++i) {                                           /* :949-956 */
      const int pos = i / K, code = i % K;
      wcount[i] += count_incs[pos * K + code];
      for (int j = 0; j < s; ++j)
        work[(size_t)i * s + j] += (1.0 / wcount[i]) * differences[(size_t)(pos + code) * s + j];
    }
    for (int i = 0; i < E; ++i) {                                           /* :959-991 */
      if (count_incs[i] > 0) {
        for (int j = 0; j < s; ++j) codebook[(size_t)i * s + j] = fo_text_roundtrip(work[(size_t)i * s + j]);
        counts[i] = wcount[i];
      }
    }
  }
  free(nearest); free(wcount); free(work); free(min_dist); free(differences);
  return rc;
}

/* coarse quantisation + residuals                                          freddy.c:1566-1582 */
int fo_insert_coarse(const float* coarse, int C, int d, const float* vecs, int n, int32_t* cq, float* residuals) {
  for (int i = 0; i < n; ++i) {
    float min_dist = 100;
    int best = -1;
    for (int j = 0; j < C; ++j) {
      const float dist = fo_sqdist(vecs + (size_t)i * d, coarse + (size_t)j * d, d);
      if (dist < min_dist) { best = j; min_dist = dist; }
    }
    if (best < 0) return -2;   /* the reference would use a stale / NULL centroid pointer here */
    cq[i] = best;
    for (int j = 0; j < d; ++j) residuals[(size_t)i * d + j] = vecs[(size_t)i * d + j] - coarse[(size_t)best * d + j];
  }
  return 0;
}

/* multi-index coarse id                                                    freddy.c:1584-1604 */
int fo_insert_coarse_multi(const float* cq_multi, int P, int Kc, int d, const float* vecs, int n, int32_t* out) {
  const int subdim = d / P;
  for (int i = 0; i < n; ++i) {
    int factor = 1;
    out[i] = 0;
    for (int pos = 0; pos < P; ++pos) {
      float min_dist = FO_MAX_DIST;
      int coarse_id = 0;
      for (int j = 0; j < Kc; ++j) {
        const float dist = fo_sqdist(vecs + (size_t)i * d + (size_t)pos * subdim, cq_multi + ((size_t)pos * Kc + j) * subdim, subdim);
        if (dist < min_dist) { coarse_id = j; min_dist = dist; }
      }
      out[i] += factor * coarse_id;
      factor *= P;   /* sic: freddy.c:1599 multiplies by the number of POSITIONS */
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------
 * f2  quantizer training: Lloyd's k-means                  index_creation/quantizer_creation.py:13-52
 * The reference calls scipy.cluster.vq.kmeans (random initial centroids, no seed): there is no result of
 * the reference to be identical to.  What is pinned here is the algorithm the device runs, so that the HIP
 * path and this restatement agree bit for bit: centroids start at the given vectors; per iteration every
 * vector goes to its nearest centroid by squareDistance (strict "<": lowest index on ties), a centroid
 * becomes sum(members in index order) / count in binary32, an empty cluster keeps its centroid.
 * ------------------------------------------------------------------------------------ */
int fo_kmeans(const float* vecs, int64_t n, int d, int k, int iters, const int32_t* init_rows, float* centroids,
              int32_t* assign_out) {
  if (!vecs || !centroids || n <= 0 || d <= 0 || k <= 0 || iters < 0) return -1;
  for (int c = 0; c < k; ++c) memcpy(centroids + (size_t)c * d, vecs + (size_t)(init_rows ? init_rows[c] : c % n) * d, sizeof(float) * (size_t)d);
  int32_t* a = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)n);
  float* sum = (float*)xmalloc(sizeof(float) * (size_t)d);
  for (int it = 0; it <= iters; ++it) {
    fo_assign_coarse(centroids, k, d, vecs, n, a);
    if (it == iters) break;
    for (int c = 0; c < k; ++c) {
      int64_t cnt = 0;
      for (int j = 0; j < d; ++j) sum[j] = 0.0f;
      for (int64_t i = 0; i < n; ++i)
        if (a[i] == c) {
          ++cnt;
          for (int j = 0; j < d; ++j) sum[j] = sum[j] + vecs[(size_t)i * d + j];
        }
      if (cnt > 0)
        for (int j = 0; j < d; ++j) centroids[(size_t)c * d + j] = sum[j] / (float)cnt;
    }
  }
  if (assign_out) memcpy(assign_out, a, sizeof(int32_t) * (size_t)n);
  free(sum); free(a);
  return 0;
}
