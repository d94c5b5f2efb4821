/*
 * freddy_oracle.h -- CPU ORACLE for the FREDDY PQ / IVFADC / kNN-join hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libfreddy_gpu.so)
 * never links, loads or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (guenthermi/postgres-word2vec) ships no tests, no
 * golden vectors and no fixtures for this path, and its C sources cannot be built in
 * this image (every file includes PostgreSQL server headers -- postgres.h, fmgr.h,
 * funcapi.h, executor/spi.h, utils/array.h -- that are absent, and writing stand-ins
 * for them is not allowed).  This oracle is therefore a from-scratch restatement of
 * the reference algorithm, each function citing the reference file:line it follows
 * (paths relative to /root/reference/freddy_extension/), checked by hand-derived
 * known-answer tests and brute-force property tests only.
 *
 * Arithmetic contract (reference is built by PGXS with default flags, x86-64 SSE2,
 * no -ffast-math, no FMA): every float operation below is a separately rounded
 * IEEE-754 binary32 operation, evaluated in exactly the written order.  This file
 * must be compiled with -O2 -ffp-contract=off and without -march=native.
 */
#ifndef FREDDY_ORACLE_H
#define FREDDY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* TopKEntry{int id; float distance}   index_utils.h:17-20 */
typedef struct fo_entry {
  int32_t id;
  float dist;
} fo_entry;

/* ---- primitives (a1-a5, a13) ------------------------------------------ */

/* squareDistance                         index_utils.c:500-508 */
float fo_sqdist(const float* a, const float* b, int n);

/* getPrecomputedDistances                index_utils.c:445-455
 * codebook is dense [m][K][s] (entry (pos,code) at ((pos*K)+code)*s); lut is [m*K]. */
void fo_lut(float* lut, int m, int K, int s, const float* q, const float* codebook);

/* Same, but over an explicit entry list in arbitrary order, exactly like the
 * reference's Codebook array of {pos, code, vector}  (index_utils.h:52-56). */
void fo_lut_entries(float* lut, int n_entries, int K, int s, const float* q,
                    const int32_t* pos, const int32_t* code, const float* vectors);

/* getPrecomputedDistancesDouble          index_utils.c:457-475 ; lut2 is [(m/2)*K*K] */
void fo_lut_double(float* lut2, int m, int K, int s, const float* q, const float* codebook);

/* computePQDistanceInt16                 index_utils.c:1126-1133 */
float fo_adc(const float* lut, const int16_t* codes, int m, int K);

/* initTopK                               index_utils.c:66-72 */
void fo_topk_init(fo_entry* tk, int k, float sentinel);

/* updateTopK                             index_utils.c:19-33 */
void fo_topk_insert(fo_entry* tk, int k, float dist, int32_t id);

/* guard + updateTopK + maxDist refresh   e.g. freddy.c:128-131 ; returns 1 if inserted */
int fo_offer(fo_entry* tk, int k, float* maxd, float dist, int32_t id);

/* getConfidenceHyp                       index_utils.c:673-682 */
float fo_confidence_hyp(int expect, int size, float p, int stat_size);

/* ---- tables ("what SPI returns", flattened) ----------------------------------------- */

/* pq_quantization + pq_codebook          (pq_index.py:25-26) */
typedef struct fo_pq_table {
  int32_t d, m, K;
  int64_t N;
  const float* codebook;   /* [m][K][d/m] */
  const int32_t* ids;      /* [N] rows in canonical scan order (ascending id) */
  const int16_t* codes;    /* [N][m] */
} fo_pq_table;

/* coarse_quantization + residual_codebook + fine_quantization (ivfadc.py:28-30),
 * fine_quantization held as inverted lists; inside a list rows are ascending id. */
typedef struct fo_ivf_table {
  int32_t d, m, K, C;
  int64_t N;
  const float* coarse;     /* [C][d]; array index == coarse id (freddy.c:309,873) */
  const float* codebook;   /* [m][K][d/m] residual codebook */
  const int32_t* list_off; /* [C+1] */
  const int32_t* ids;      /* [N] */
  const int16_t* codes;    /* [N][m] */
} fo_ivf_table;

/* codebook_ivpq + coarse_quantization_ivpq (multi index) + fine_quantization_ivpq
 * (+ normalised vectors joined by id) + stat table   (ivpq.py:18-38; sql:158-168) */
typedef struct fo_ivpq_table {
  int32_t d, m, K;          /* fine PQ */
  int32_t cpos, ccodes;     /* coarse multi-index: cpos positions (must be 2) x ccodes */
  int64_t N;
  const float* codebook;    /* [m][K][d/m] */
  const float* coarse;      /* [cpos][ccodes][d/cpos] */
  const int32_t* ids;       /* [N] ascending */
  const int32_t* coarse_id; /* [N]  cell = code0 + ccodes*code1 */
  const int16_t* codes;     /* [N][m] */
  const float* vectors;     /* [N][d] row-aligned with ids (the JOIN vecs), may be NULL */
  const float* stats;       /* [cells+1]; last entry = total count (sql:150-168) */
} fo_ivpq_table;

/* ---- drivers (a6-a9, a14) -------------------------------------------------
 * All return 0 on success, <0 on argument error.  Outputs are caller-allocated.
 * Unfilled result slots keep the reference's sentinel (id=-1, dist=sentinel).        */

/* pq_search                              freddy.c:28-152   (sentinel 100.0) */
int fo_pq_search(const fo_pq_table* t, const float* q, int k, fo_entry* out);

/* pq_search_in                           freddy.c:1028-1157 (sentinel 1000.0)
 * rows = rows whose id is in input_ids, canonical order. */
int fo_pq_search_in(const fo_pq_table* t, const float* q, int k, const int32_t* input_ids,
                    int n_ids, fo_entry* out);

/* pq_search_in_batch                     freddy.c:414-653  (sentinel 1000.0)
 * out is [Q][k]; both use_target_lists branches are restated. */
int fo_pq_search_in_batch(const fo_pq_table* t, const float* queries, int Q, int k,
                          const int32_t* input_ids, int n_ids, int use_target_lists,
                          fo_entry* out);

/* ivfadc_search                          freddy.c:174-393  (W = get_w(), sentinel 1000.0)
 * found_rule: 0 = rows retrieved (single-query UDF, freddy.c:377)
 *             1 = accepted insertions (batch UDF, freddy.c:971)                         */
int fo_ivfadc_search(const fo_ivf_table* t, const float* q, int k, int W, float sentinel,
                     int found_rule, fo_entry* out);

/* The same per-query routine over Q queries (the build's nprobe generalisation of the
 * batch UDF); n_threads>1 splits queries statically with OpenMP
 * ("one backend per core" for the CPU baseline). */
int fo_ivfadc_search_many(const fo_ivf_table* t, const float* queries, int Q, int k, int W,
                          float sentinel, int found_rule, int n_threads, fo_entry* out);

/* ivfadc_batch_search                    freddy.c:679-999  (1 probe / round, sentinel 100.0)
 * Restated with the reference's own loop structure (rounds over all unfinished
 * queries; rows outer, queries-of-cell inner).  queries are the already-fetched
 * normalised vectors in fetch order; out is [Q][k]. */
int fo_ivfadc_batch_search(const fo_ivf_table* t, const float* queries, int Q, int k,
                           fo_entry* out);

/* determineCoarseIdsMultiWithStatisticsMulti   index_utils.c:252-443 (cpos == 2 only)
 * For every active query emits the visited cells in order.
 * cells_out: [n_active][cells] (row x holds the cells of active[x]); counts_out[n_active].
 * Returns lastIteration (1 iff every active query exhausted all cells). */
int fo_multi_index_select(const fo_ivpq_table* t, const float* queries, const int32_t* active,
                          int n_active, int n_targets, int min_target_count, float confidence,
                          int32_t* cells_out, int32_t* counts_out);

/* ivpq_search_in                         ivpq_search_in.c:61-699
 * method 0 PQ, 1 EXACT, 2 PQ + post verification.  out is [Q][k], sentinel 1000.0.
 * iterations_out (may be NULL) receives the number of alpha rounds executed. */
int fo_ivpq_search_in(const fo_ivpq_table* t, const float* queries, int Q, int k,
                      const int32_t* target_ids, int n_targets, int alpha, int pvf, int method,
                      int use_target_lists, float confidence, int double_threshold,
                      fo_entry* out, int* iterations_out);

/* postverify                             index_utils.c:477-498 (one query)
 * cand_ids/cand_vecs: k*pvf candidates in buffer order (id -1 = hole). */
void fo_postverify(const float* q, int d, int k, int n_cand, const int32_t* cand_ids,
                   const float* const* cand_vecs, float sentinel, fo_entry* tk);

/* ---- next row: exact brute-force kNN ------------------------------------------
 * cosine_similarity_bytea                 core_functions.c:67-81
 * float scalar = 0; scalar += v1[i] * v2[i]  (binary32 mul, then binary32 add, i ascending) */
float fo_cosine_similarity_bytea(const float* v1, const float* v2, int n);

/* k_nearest_neighbour / knn_in_exact      freddy--0.0.1.sql:426-454, 991-1084
 * "ORDER BY cosine_similarity_bytea(q, v.vector) DESC FETCH FIRST k ROWS ONLY" over all rows
 * (input_ids == NULL) or "WHERE v.id = ANY(input_ids)".  PostgreSQL leaves the order of equal
 * similarities unspecified; pinned here to ascending id.  ids ascending; out[i].dist holds
 * the SIMILARITY.  Returns the number of rows produced (<= k). */
int fo_exact_knn(const float* vectors, const int32_t* ids, int64_t N, int d, const float* q, int k,
                 const int32_t* input_ids, int n_ids, fo_entry* out);

/* ---- next row: grouping and analogy on the PQ index ------------------------------
 * vec_minus_bytea / vec_plus_bytea        core_functions.c:118-136, 177-195   out[i] = a[i] -/+ b[i]
 * vec_normalize_bytea                     core_functions.c:241-266
 *   float sq = 0; sq += v[i]*v[i]; length = (float)sqrt((double)sq); out[i] = v[i] / length           */
void fo_vec_minus(const float* a, const float* b, int n, float* out);
void fo_vec_plus(const float* a, const float* b, int n, float* out);
void fo_vec_normalize(const float* v, int n, float* out);

/* grouping_pq                             freddy.c:1176-1401
 * group_vecs: the groups' vectors in ascending GROUP ID order (the reference sorts the group ids and
 * fetches "ORDER BY id ASC").  For every row of pq_quantization with id IN input_ids (canonical order,
 * duplicates / unknown ids dropped): ADC distance to each group's LUT, strict "<" from minDist = 100, so
 * the first of equally near groups wins.  out_group[i] = index into group_vecs, or -1 when no group is
 * nearer than 100 (the reference leaves nearestGroup[i] uninitialised there).  Returns the row count. */
int fo_grouping_pq(const fo_pq_table* t, const float* group_vecs, int n_groups, const int32_t* input_ids,
                   int n_ids, int32_t* out_ids, int32_t* out_group);

/* ---- next row: index build ------------------------------------------------------------
 * PQ encoding as index_creation/pq_index.py:65-92 (create_index): per position the code whose codeword
 * is nearest to the sub-vector, strict "<" over the codes in order = lowest code on ties.  The
 * reference measures with np.linalg.norm; this restatement uses squareDistance (same argmin up to
 * rounding; bit-exactness is defined against THIS arithmetic).  codes: [n][m].                          */
void fo_encode_pq(const float* codebook, int m, int K, int s, const float* vecs, int64_t n, int16_t* codes);
/* coarse assignment (ivfadc.py / quantizer_creation.py:41-49: faiss IndexFlatL2, k = 1): nearest centroid
 * by squareDistance over all d dimensions, lowest index on ties. */
void fo_assign_coarse(const float* coarse, int C, int d, const float* vecs, int64_t n, int32_t* cell);

/* ---- f2: quantizer training (quantizer_creation.py:13-52): Lloyd's k-means as the device runs it.  Initial
 * centroids = vecs[init_rows[c]] (NULL: vecs[c mod n]); assign_out [n] (may be NULL) = the final assignment. */
int fo_kmeans(const float* vecs, int64_t n, int d, int k, int iters, const int32_t* init_rows, float* centroids,
              int32_t* assign_out);

/* ---- f4: insert_batch (freddy.c:1403-1658, index_utils.c:908-1074) ------------------------------------- */
float fo_text_roundtrip(float v);   /* sprintf("%f") -> float4 input, as every float the reference INSERTs / UPDATEs */
/* updateCodebook + updateCodebookRelation: codebook [m][K][s] / counts [m*K] updated in place; codes [n][m];
 * count_incs [m*K].  -2: a sub-vector is >= 100 away from every entry (undefined in the reference). */
int fo_update_codebook(float* codebook, int32_t* counts, int m, int K, int s, const float* vecs, int n,
                       int16_t* codes, int32_t* count_incs);
int fo_insert_coarse(const float* coarse, int C, int d, const float* vecs, int n, int32_t* cq, float* residuals);
int fo_insert_coarse_multi(const float* cq_multi, int P, int Kc, int d, const float* vecs, int n, int32_t* out);

/* SRF emit text round trip               freddy.c:164 ("%f" into a 16-byte buffer) */
float fo_emit_roundtrip(float dist);

#ifdef __cplusplus
}
#endif
#endif /* FREDDY_ORACLE_H */
