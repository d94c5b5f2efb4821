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
 * host state); freddy_glue_unpin_all() runs from an on_proc_exit hook and may be called by set_*() wrappers
 * after the tables changed.  HIP is initialised lazily by the first pin, i.e. after the fork. */
freddy_gpu_index_t *freddy_glue_pq(void);     /* pq_codebook + pq_quantization                         freddy.c:69,96-100   */
freddy_gpu_index_t *freddy_glue_ivf(void);    /* coarse_quantization + residual_codebook + fine_quant.  freddy.c:239-241     */
freddy_gpu_index_t *freddy_glue_ivpq(void);   /* codebook_ivpq + coarse multi index + fine_quant._ivpq  ivpq_search_in.c:218-232 */
void freddy_glue_unpin_all(void);

/* argument codecs (index_utils.c:1078-1106, :797-808) */
float *freddy_glue_bytea_f32(bytea *b, int *n);                    /* palloc'd copy of a float4 bytea */
int32 *freddy_glue_int_array(ArrayType *a, int *n);                /* int[] -> palloc'd int32[] */
float *freddy_glue_bytea_array_f32(ArrayType *a, int *rows, int *dim);   /* bytea[] -> flat [rows][dim] */

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
