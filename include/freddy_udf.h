/*
 * freddy_udf.h -- host-side mirror of the FREDDY search UDFs (C ABI, libfreddy_host.so).
 *
 * PostgreSQL is not available in this image, so this library stands where the extension's
 * SRF hosts (freddy_extension/freddy.c, ivpq_search_in.c) stand: it owns what the SQL layer
 * owns -- the tables, the set_*()/get_*() "config functions" (freddy--0.0.1.sql:5-132,188-194),
 * id -> vector lookup, "WHERE id IN (...)" semantics, row emission -- and forwards the search
 * itself through the device C ABI (include/freddy_gpu.h).  Function names, argument order and
 * meaning, result row shapes and error messages follow the reference UDFs, so a test written
 * against this header reads like a SQL call of the extension.
 *
 * Every function returns 0 or a negative code; freddy_udf_last_error() gives the message the
 * reference would have raised with elog(ERROR, ...).
 */
#ifndef FREDDY_UDF_H
#define FREDDY_UDF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct freddy_session freddy_session_t;

/* result rows: (Id, Distance)  freddy.c:142-146 ; (QueryId, TargetId|Id, Distance)  freddy.c:643-649,988-994 */
typedef struct freddy_row2 { int32_t id; float distance; } freddy_row2;
typedef struct freddy_row3 { int32_t query_id; int32_t id; float distance; } freddy_row3;
typedef struct freddy_group_row { int32_t id; int32_t group_id; } freddy_group_row;   /* grouping_pq: (Ids, GroupIds) */

int freddy_session_open(int device, freddy_session_t** out);
int freddy_session_close(freddy_session_t* s);
const char* freddy_udf_last_error(void);

/* ---- tables (what the index_creation scripts insert; rows may come in any order) -------- */
/* google_vecs_norm (id, vector)                                   vec2database.py:47-58 */
int freddy_load_vecs_norm(freddy_session_t* s, const int32_t* ids, const float* vectors, int64_t N, int32_t d);
/* pq_codebook (pos, code, vector) + pq_quantization (id, vector)   pq_index.py:25-26 */
int freddy_load_pq(freddy_session_t* s, const int32_t* cb_pos, const int32_t* cb_code, const float* cb_vectors,
                   int32_t n_entries, int32_t sub_dim, const int32_t* ids, const int16_t* codes, int64_t N);
/* coarse_quantization (id, vector) + residual_codebook + fine_quantization (id, coarse_id, vector)  ivfadc.py:28-30 */
int freddy_load_ivfadc(freddy_session_t* s, const int32_t* coarse_ids_tbl, const float* coarse_vectors, int32_t C,
                       const int32_t* cb_pos, const int32_t* cb_code, const float* cb_vectors, int32_t n_entries,
                       int32_t sub_dim, const int32_t* ids, const int32_t* coarse_id, const int16_t* codes, int64_t N);
/* codebook_ivpq + coarse_quantization_ivpq (pos, code, vector) + fine_quantization_ivpq (id, coarse_id, vector)
 * + stat table (coarse_id, coarse_freq); vectors for methods 1/2 come from google_vecs_norm.   ivpq.py:18-38 */
int freddy_load_ivpq(freddy_session_t* s, const int32_t* cb_pos, const int32_t* cb_code, const float* cb_vectors,
                     int32_t n_entries, int32_t sub_dim, const int32_t* cq_pos, const int32_t* cq_code,
                     const float* cq_vectors, int32_t n_cq_entries, const int32_t* ids, const int32_t* coarse_id,
                     const int16_t* codes, int64_t N, const int32_t* stat_coarse_id, const float* stat_freq,
                     int32_t n_stat);

/* ---- index files ("Postgres tables -> flat binary -> HBM loader") ------------------------
 * One little-endian container of named arrays, the tables exactly as the index_creation scripts insert
 * them (the reference exports a Python pickle, index_manager.py:10-18; this is the language-neutral
 * counterpart):
 *   "FRDYIDX1" | uint32 n_arrays | n_arrays x { uint16 name_len | name | uint8 dtype (0 float32, 1 int32,
 *   2 int16) | uint8 ndim | uint64 dims[ndim] | padding to 8 | data | padding to 8 }
 * Array names are "<table>.<column>":
 *   google_vecs_norm.id/.vector | pq_codebook.pos/.code/.vector, pq_quantization.id/.vector |
 *   coarse_quantization.id/.vector, residual_codebook.pos/.code/.vector, fine_quantization.id/.coarse_id/.vector |
 *   codebook_ivpq.pos/.code/.vector, coarse_quantization_ivpq.pos/.code/.vector,
 *   fine_quantization_ivpq.id/.coarse_id/.vector, stat.coarse_id/.coarse_freq
 * freddy_index_file_write writes the arrays it is given; freddy_import_index reads a file and loads every
 * table group that is complete in it (same effect as the freddy_load_* calls; google_vecs_norm first). */
typedef struct freddy_file_array {
  const char* name;
  int32_t dtype;        /* 0 float32, 1 int32, 2 int16 */
  int32_t ndim;         /* 1 or 2 */
  int64_t dims[2];
  const void* data;
} freddy_file_array;
int freddy_index_file_write(const char* path, const freddy_file_array* arrays, int32_t n_arrays);
int freddy_import_index(freddy_session_t* s, const char* path);

/* ---- config functions                                    freddy--0.0.1.sql:21-132, 188-194 */
int freddy_set_w(freddy_session_t* s, int32_t w);                       /* default 3 */
int freddy_set_pvf(freddy_session_t* s, int32_t pvf);                   /* default 20 */
int freddy_set_alpha(freddy_session_t* s, int32_t alpha);               /* default 3 */
int freddy_set_confidence_value(freddy_session_t* s, float c);          /* default 0.8 */
int freddy_set_long_codes_threshold(freddy_session_t* s, int32_t t);    /* default 10000000 */
int freddy_set_method_flag(freddy_session_t* s, int32_t m);             /* default 0 */
int freddy_set_use_targetlist(freddy_session_t* s, int32_t flag);       /* default true */
int32_t freddy_get_w(const freddy_session_t* s);
int32_t freddy_get_pvf(const freddy_session_t* s);
int32_t freddy_get_alpha(const freddy_session_t* s);
float freddy_get_confidence_value(const freddy_session_t* s);
int32_t freddy_get_long_codes_threshold(const freddy_session_t* s);
int32_t freddy_get_method_flag(const freddy_session_t* s);
int32_t freddy_get_use_targetlist(const freddy_session_t* s);

/* ---- the UDFs (out arrays are caller-allocated; *n_rows receives the number of rows) ----- */
/* pq_search(bytea, int) -> SETOF (Id, Distance)                              freddy.c:28-171 */
int pq_search(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows);
/* ivfadc_search(bytea, int) -> SETOF (Id, Distance); W = get_w()            freddy.c:174-410 */
int ivfadc_search(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows);
/* pq_search_in(bytea, int, int[]) -> SETOF (Id, Distance)                  freddy.c:1028-1174 */
int pq_search_in(freddy_session_t* s, const float* query, int32_t dim, int32_t k, const int32_t* input_ids,
                 int32_t n_ids, freddy_row2* out, int32_t* n_rows);
/* pq_search_in_batch(bytea[], int[], int, int[], bool) -> SETOF (QueryId, TargetId, Distance)  freddy.c:412-676
 * out holds n_queries*k rows, query-major, rank-minor. */
int pq_search_in_batch(freddy_session_t* s, const float* queries, int32_t n_queries, int32_t dim,
                       const int32_t* query_ids, int32_t n_query_ids, int32_t k, const int32_t* input_ids,
                       int32_t n_ids, int32_t use_target_lists, freddy_row3* out, int32_t* n_rows);
/* ivfadc_batch_search(int[], int) -> SETOF (QueryId, Id, Distance)          freddy.c:677-1025
 * queries = rows of google_vecs_norm whose id is in query_ids, in table order (ascending id),
 * duplicates and unknown ids dropped; out must hold n_query_ids*k rows. */
int ivfadc_batch_search(freddy_session_t* s, const int32_t* query_ids, int32_t n_query_ids, int32_t k,
                        freddy_row3* out, int32_t* n_rows);
/* ivpq_search_in(bytea[], int[], int, int[], int, int, int, bool, float4, int)   ivpq_search_in.c:59-721 */
int ivpq_search_in(freddy_session_t* s, const float* queries, int32_t n_queries, int32_t dim, const int32_t* query_ids,
                   int32_t n_query_ids, int32_t k, const int32_t* input_ids, int32_t n_ids, int32_t alpha, int32_t pvf,
                   int32_t method, int32_t use_target_lists, float confidence, int32_t double_threshold,
                   freddy_row3* out, int32_t* n_rows);
/* knn_in_ivpq_batch's parameter plumbing (freddy--0.0.1.sql:720-828): ivpq_search_in with
 * alpha/pvf/method/use_targetlist/confidence/long_codes_threshold taken from the getters. */
int knn_join(freddy_session_t* s, const float* queries, int32_t n_queries, int32_t dim, const int32_t* query_ids,
             int32_t k, const int32_t* input_ids, int32_t n_ids, freddy_row3* out, int32_t* n_rows);

/* Next row: the exact brute-force functions, by row id instead of word.
 * k_nearest_neighbour(bytea, int)             freddy--0.0.1.sql:426-439
 * knn_in_exact(bytea, int, integer[])         freddy--0.0.1.sql:1041-1054
 * ORDER BY cosine_similarity_bytea(q, vector) DESC FETCH FIRST k ROWS ONLY over google_vecs_norm;
 * row.distance carries the SIMILARITY; *n_rows <= k (fewer when fewer rows qualify). */
int k_nearest_neighbour(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows);
int knn_in_exact(freddy_session_t* s, const float* query, int32_t dim, int32_t k, const int32_t* input_ids, int32_t n_ids,
                 freddy_row2* out, int32_t* n_rows);

/* Next row: grouping and analogy on the same kernels, keyed by row id instead of word.
 * grouping_pq(integer[], integer[]) -> SETOF (Ids, GroupIds)                 freddy.c:1176-1401
 *   rows of pq_quantization with id IN input_ids (table order), each with the nearest of the groups
 *   (ADC distance to the group's normalised vector; groups tried in ascending id, first nearest wins;
 *   group_id -1 if none is nearer than 100).  Error "Group ids do not exist" as the reference.
 * analogy_3cosadd_pq / analogy_3cosadd_ivfadc                               freddy--0.0.1.sql:1317-1346, 1428-1460
 *   q = vec_normalize(v3 - v1 + v2); candidates = pq_search / ivfadc_search(q, get_pvf() + 3) minus the
 *   three inputs; result = the candidate with the largest cosine_similarity_bytea(v3 - v1 + v2, v4)
 *   (-1: none, the SQL returns NULL). */
int grouping_pq(freddy_session_t* s, const int32_t* input_ids, int32_t n_ids, const int32_t* group_ids, int32_t n_groups,
                freddy_group_row* out, int32_t* n_rows);
int analogy_3cosadd_pq(freddy_session_t* s, int32_t id1, int32_t id2, int32_t id3, int32_t* result);
int analogy_3cosadd_ivfadc(freddy_session_t* s, int32_t id1, int32_t id2, int32_t id3, int32_t* result);

/* The plpgsql callers of the two single-query SRFs, keyed by row id; row.distance
 * carries the SIMILARITY, *n_rows <= k (the joins drop the (-1, sentinel) filler rows).
 * k_nearest_neighbour_pq / k_nearest_neighbour_ivfadc (bytea, int)        freddy--0.0.1.sql:610-622, 520-531
 *   similarity = (1.0 - (distance / 2.0))::float4 of the distance as the SRF emits it ("%f")
 * k_nearest_neighbour_pq_pv / k_nearest_neighbour_ivfadc_pv (bytea, int)  freddy--0.0.1.sql:625-641, 575-591
 *   get_pvf() * k candidates, re-ranked by cosine_similarity_bytea(q, vector) DESC, first k */
int k_nearest_neighbour_pq(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows);
int k_nearest_neighbour_ivfadc(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows);
int k_nearest_neighbour_pq_pv(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows);
int k_nearest_neighbour_ivfadc_pv(freddy_session_t* s, const float* query, int32_t dim, int32_t k, freddy_row2* out, int32_t* n_rows);

/* knn_in_pq(anyarray, int, int[]) and k_nearest_neighbour_ivfadc_batch(varchar[], int) (by query ids):
 * pq_search_in / ivfadc_batch_search with the same similarity mapping     freddy--0.0.1.sql:830-843, 535-553 */
int knn_in_pq(freddy_session_t* s, const float* query, int32_t dim, int32_t k, const int32_t* input_ids, int32_t n_ids,
              freddy_row2* out, int32_t* n_rows);
int k_nearest_neighbour_ivfadc_batch(freddy_session_t* s, const int32_t* query_ids, int32_t n_query_ids, int32_t k,
                                     freddy_row3* out, int32_t* n_rows);

/* Next row (remainder): analogy over an input set and the clustering functions, by row id.
 * analogy_3cosadd_in_pq / analogy_3cosadd_in_ivpq                        freddy--0.0.1.sql:1348-1426
 *   as analogy_3cosadd_pq, candidates from pq_search_in(q, get_pvf() + 3, input ids) resp.
 *   ivpq_search_in(ARRAY[q], '{0}', 4, input ids, get_alpha(), get_pvf(), get_method_flag(), ...) (k is the literal 4 there).
 * cluster_exact / cluster_pq / cluster_ivpq = generic_cluster             freddy--0.0.1.sql:1086-1209
 *   k-means in the reference's plpgsql: k random tokens as initial centroids, 10 rounds of "every token goes to the
 *   centroid that lists it with the highest similarity" (rows of knn_search_in_batch / knn_in_pq_batch /
 *   knn_in_ivpq_batch with k = all tokens, ORDER BY similarity DESC; ties: centroid, then token position), centroids
 *   re-estimated by centroid_bytea over 10 random members -- empty clusters draw their samples and keep their
 *   centroid, as the reference does.  `draws` are the values random() returns, in call order (k, then 10 per
 *   cluster and round): a caller-supplied sequence makes a run reproducible; once it is used up (or NULL) an
 *   internal generator continues.  cluster_out[i] = 1..k for token i (0: no centroid listed it). */
int analogy_3cosadd_in_pq(freddy_session_t* s, int32_t id1, int32_t id2, int32_t id3, const int32_t* input_ids, int32_t n_ids, int32_t* result);
int analogy_3cosadd_in_ivpq(freddy_session_t* s, int32_t id1, int32_t id2, int32_t id3, const int32_t* input_ids, int32_t n_ids, int32_t* result);
int cluster_exact(freddy_session_t* s, const int32_t* token_ids, int32_t n, int32_t k, const double* draws, int32_t n_draws, int32_t* cluster_out);
int cluster_pq(freddy_session_t* s, const int32_t* token_ids, int32_t n, int32_t k, const double* draws, int32_t n_draws, int32_t* cluster_out);
int cluster_ivpq(freddy_session_t* s, const int32_t* token_ids, int32_t n, int32_t k, const double* draws, int32_t n_draws, int32_t* cluster_out);

/* Next row: insert_batch(varchar[]) -> int4                  freddy.c:1403-1658
 * The tokenisation sub-query (:1503-1519: tokenize(term) for the terms NOT yet in the vocabulary) stays with SQL;
 * the caller passes its result, the normalised vectors of the new terms.  Per vector: PQ code, coarse cell +
 * residual code, ivpq code + multi-index cell (device); the three codebooks' running update exactly as
 * updateCodebook / updateCodebookRelation compute and store it (index_utils.c:908-991, "%f" text included);
 * one row per table with id = max(id) + 1 of THAT table (index_utils.c:993-1074); every pinned index is extended in
 * HBM.  new_ids (may be NULL) receives the ids given in google_vecs_norm.  The count column of a codebook is 1
 * unless freddy_set_codebook_counts (table: 0 pq_codebook, 1 residual_codebook, 2 codebook_ivpq) has set it. */
int freddy_set_codebook_counts(freddy_session_t* s, int32_t table, const int32_t* pos, const int32_t* code, const int32_t* count, int32_t n);
int insert_batch(freddy_session_t* s, const float* norm_vectors, int32_t n, int32_t dim, int32_t* new_ids);

/* per-call row emit: snprintf("%d") / snprintf("%f") into 16-byte buffers   freddy.c:154-169,1001-1023 */
void freddy_emit_row2(const freddy_row2* row, char values[2][16]);
void freddy_emit_row3(const freddy_row3* row, char values[3][16]);

#ifdef __cplusplus
}
#endif
#endif /* FREDDY_UDF_H */
