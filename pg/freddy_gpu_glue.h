/*
 * freddy_gpu_glue.h -- what the thin PostgreSQL hosts (pg/freddy_srf.c, pg/ivpq_search_in.c) share:
 * the per-backend pinned handles, the table flattening, bytea / array argument codecs, row emission.
 *
 * NOT compiled in the build image of this repository (no PostgreSQL headers there); pg/Makefile builds
 * it wherever pg_config exists, against the untouched reference sources for everything that is not the
 * search path (index_utils.c: getCodebook, getCoarseQuantizer, getStatistics, getTableName, getParameter).
 */
#ifndef FREDDY_GPU_GLUE_H
#define FREDDY_GPU_GLUE_H

#include "postgres.h"
#include "fmgr.h"
#include "funcapi.h"
#include "utils/array.h"

#include "freddy_gpu.h"

/* Handles live as long as the backend (TopMemoryContext is not involved: the library owns HBM and its own
 * host state) but are CHECKED against the tables before every search: each of the three functions below compares the
 * handle's stamp (table OIDs, relfilenodes, generation counters of pg/freddy_gpu_watch.sql or a weaker size / max(id)
 * / sum(count) stamp) with the tables as the caller's snapshot sees them, and appends new rows in HBM, reloads the
 * codebook, or unpins and pins again -- the reference re-reads the tables on every call (freddy.c:69, :239-241), so a
 * pinned copy must never answer from older rows.  freddy_glue_unpin_all() runs from an on_proc_exit hook.  HIP is
 * initialised lazily by the first pin, i.e. after the fork.  One GPU context and one pinned copy PER BACKEND: N
 * backends hold N copies of the tables in HBM (100 MB each for the 3 M-row index of the benchmark; 288 GB of HBM)
 * and their persistent scans each ask for every CU -- a deployment that wants several searches in flight on one pinned
 * copy sends larger batches through one backend (the library pipelines a batch internally, include/freddy_gpu.h). */
freddy_gpu_index_t *freddy_glue_pq(void);     /* pq_codebook + pq_quantization                         freddy.c:69,96-100   */
freddy_gpu_index_t *freddy_glue_ivf(void);    /* coarse_quantization + residual_codebook + fine_quant.  freddy.c:239-241     */
freddy_gpu_index_t *freddy_glue_ivpq(void);   /* codebook_ivpq + coarse multi index + fine_quant._ivpq  ivpq_search_in.c:218-232 */
void freddy_glue_unpin_all(void);
void freddy_glue_refresh_pinned(void);        /* run the staleness check of every handle this backend has pinned (after insert_batch) */

/* argument codecs (index_utils.c:1078-1106, :797-808) */
float *freddy_glue_bytea_f32(bytea *b, int *n);                    /* palloc'd copy of a float4 bytea */
int32 *freddy_glue_int_array(ArrayType *a, int *n);                /* int[] -> palloc'd int32[] */
/* bytea[] -> flat [rows][dim] in the backend's reusable PINNED buffer (freddy_gpu_host_alloc: the library reads it
 * without a staging copy); valid until the next call of this function, never pfree'd by the caller */
float *freddy_glue_bytea_array_f32(ArrayType *a, int *rows, int *dim);
/* elog(ERROR) unless a query's dimensionality is the pinned index's (the library would read past a shorter vector) */
void freddy_glue_check_dim(int query_dim, int index_dim);
int  freddy_glue_dim(freddy_gpu_index_t *h);

/* elog(ERROR) with the library's message if rc != 0 */
void freddy_glue_check(int rc);

/* Value-per-call emission of (Id, Distance) / (QueryId, Id, Distance) rows: snprintf("%d") / snprintf("%f")
 * into C strings, BuildTupleFromCStrings -- the text round trip is part of the observable output
 * (freddy.c:154-169, :1001-1023; ivpq_search_in.c:700-720). */
typedef struct FreddyRows {
    int32 *query_ids;   /* NULL: two-column rows */
    int32 *ids;
    float *dist;
    int    n_rows, k, iter;
} FreddyRows;
Datum freddy_glue_emit(FunctionCallInfo fcinfo, FuncCallContext *funcctx);

#endif
