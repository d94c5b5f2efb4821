/*
 * freddy_srf.c -- the search SRFs of freddy_extension/freddy.c as thin hosts over libfreddy_gpu.so.
 *
 * Same PG_FUNCTION_INFO_V1 symbols, same SQL signatures (freddy--0.0.1.sql:334-424 stays untouched), same
 * record shapes and the same "%d" / "%f" row text; what used to be "re-read the tables through SPI, build
 * LUTs, scan tuples, updateTopK" inside SRF_IS_FIRSTCALL() is one call through the C ABI on tables that were
 * pinned into HBM once per backend (pg/freddy_gpu_glue.c).
 *
 *   pq_search(bytea, int)                               freddy.c:28-171
 *   ivfadc_search(bytea, int)                           freddy.c:174-410
 *   pq_search_in(bytea, int, int[])                     freddy.c:1028-1174
 *   pq_search_in_batch(bytea[], int[], int, int[], bool) freddy.c:412-676
 *   ivfadc_batch_search(int[], int)                     freddy.c:677-1025
 *
 * The remaining symbols of freddy.c (grouping_pq, insert_batch, read_bytea*, vec_to_bytea) are compiled from
 * the reference file itself, whose five search functions step aside by -D renames (pg/Makefile).
 * NOT compiled in this repository's image (no PostgreSQL headers).
 */
#include "freddy_gpu_glue.h"

#include "executor/spi.h"
#include "utils/builtins.h"
#include "utils/memutils.h"

#include "index_utils.h"

/* first-call frame shared by the five functions: result arrays in the multi-call context */
static FreddyRows *begin_rows(FunctionCallInfo fcinfo, FuncCallContext **pctx, int n_queries, int k, bool with_query)
{
    FuncCallContext *funcctx = SRF_FIRSTCALL_INIT();
    MemoryContext    old = MemoryContextSwitchTo(funcctx->multi_call_memory_ctx);
    TupleDesc        tupdesc;
    FreddyRows      *r = palloc0(sizeof(FreddyRows));
    if (get_call_result_type(fcinfo, NULL, &tupdesc) != TYPEFUNC_COMPOSITE)
        ereport(ERROR, (errcode(ERRCODE_FEATURE_NOT_SUPPORTED),
                        errmsg("function returning record called in context that cannot accept type record")));
    funcctx->attinmeta = TupleDescGetAttInMetadata(tupdesc);
    r->k = k; r->n_rows = n_queries * k; r->iter = 0;
    r->ids = palloc(sizeof(int32) * (r->n_rows > 0 ? r->n_rows : 1));
    r->dist = palloc(sizeof(float) * (r->n_rows > 0 ? r->n_rows : 1));
    r->query_ids = with_query ? palloc(sizeof(int32) * (n_queries > 0 ? n_queries : 1)) : NULL;
    funcctx->user_fctx = r;
    MemoryContextSwitchTo(old);
    *pctx = funcctx;
    return r;
}

PG_FUNCTION_INFO_V1(pq_search);
Datum pq_search(PG_FUNCTION_ARGS)
{
    if (SRF_IS_FIRSTCALL()) {
        FuncCallContext *funcctx;
        int    k = PG_GETARG_INT32(1), n;
        float *q = freddy_glue_bytea_f32(PG_GETARG_BYTEA_P(0), &n);
        freddy_gpu_index_t *h = freddy_glue_pq();
        FreddyRows *r = begin_rows(fcinfo, &funcctx, 1, k, false);
        freddy_glue_check_dim(n, freddy_glue_dim(h));
        /* replaces freddy.c:69-132: all rows, list sentinel 100.0 */
        freddy_glue_check(freddy_gpu_pq_search(h, q, 1, k, 100.0f, NULL, 0, r->ids, r->dist));
    }
    return freddy_glue_emit(fcinfo, SRF_PERCALL_SETUP());
}

PG_FUNCTION_INFO_V1(pq_search_in);
Datum pq_search_in(PG_FUNCTION_ARGS)
{
    if (SRF_IS_FIRSTCALL()) {
        FuncCallContext *funcctx;
        int    k = PG_GETARG_INT32(1), n, n_ids;
        float *q = freddy_glue_bytea_f32(PG_GETARG_BYTEA_P(0), &n);
        int32 *ids = freddy_glue_int_array(PG_GETARG_ARRAYTYPE_P(2), &n_ids);
        int32  none = -1;
        freddy_gpu_index_t *h = freddy_glue_pq();
        FreddyRows *r = begin_rows(fcinfo, &funcctx, 1, k, false);
        freddy_glue_check_dim(n, freddy_glue_dim(h));
        /* replaces freddy.c:1086-1143: "WHERE id IN (...)" -- unknown ids vanish, duplicates collapse; sentinel 1000.0 */
        freddy_glue_check(freddy_gpu_pq_search(h, q, 1, k, 1000.0f, n_ids ? ids : &none, n_ids ? n_ids : 1, r->ids, r->dist));
    }
    return freddy_glue_emit(fcinfo, SRF_PERCALL_SETUP());
}

PG_FUNCTION_INFO_V1(pq_search_in_batch);
Datum pq_search_in_batch(PG_FUNCTION_ARGS)
{
    if (SRF_IS_FIRSTCALL()) {
        FuncCallContext *funcctx;
        int    Q, dim, n_qids, n_ids, k = PG_GETARG_INT32(2);
        float *qs = freddy_glue_bytea_array_f32(PG_GETARG_ARRAYTYPE_P(0), &Q, &dim);
        int32 *qids = freddy_glue_int_array(PG_GETARG_ARRAYTYPE_P(1), &n_qids);
        int32 *ids = freddy_glue_int_array(PG_GETARG_ARRAYTYPE_P(3), &n_ids);
        int32  none = -1;
        FreddyRows *r;
        freddy_gpu_index_t *h = freddy_glue_pq();
        if (n_qids != Q) elog(ERROR, "Number of query vectors and query vector ids differs!");   /* freddy.c:495 */
        if (Q > 0) freddy_glue_check_dim(dim, freddy_glue_dim(h));
        r = begin_rows(fcinfo, &funcctx, Q, k, true);
        memcpy(r->query_ids, qids, sizeof(int32) * Q);
        /* replaces freddy.c:518-631; both useTargetLists branches (arg 4) give the same lists */
        freddy_glue_check(freddy_gpu_pq_search(h, qs, Q, k, 1000.0f, n_ids ? ids : &none, n_ids ? n_ids : 1, r->ids, r->dist));
    }
    return freddy_glue_emit(fcinfo, SRF_PERCALL_SETUP());
}

PG_FUNCTION_INFO_V1(ivfadc_search);
Datum ivfadc_search(PG_FUNCTION_ARGS)
{
    if (SRF_IS_FIRSTCALL()) {
        FuncCallContext *funcctx;
        int    k = PG_GETARG_INT32(1), n, w = 0;
        float *q = freddy_glue_bytea_f32(PG_GETARG_BYTEA_P(0), &n);
        freddy_gpu_index_t *h = freddy_glue_ivf();
        FreddyRows *r = begin_rows(fcinfo, &funcctx, 1, k, false);
        freddy_glue_check_dim(n, freddy_glue_dim(h));
        getParameter(PARAM_W, &w);                                                      /* freddy.c:229 */
        /* replaces freddy.c:239-378: W cells per round, rounds while fewer than k rows were retrieved (:377) */
        freddy_glue_check(freddy_gpu_ivfadc_search(h, q, 1, k, w, 1000.0f, FREDDY_FOUND_ROWS, r->ids, r->dist));
    }
    return freddy_glue_emit(fcinfo, SRF_PERCALL_SETUP());
}

PG_FUNCTION_INFO_V1(ivfadc_batch_search);
Datum ivfadc_batch_search(PG_FUNCTION_ARGS)
{
    if (SRF_IS_FIRSTCALL()) {
        FuncCallContext *funcctx;
        int    k = PG_GETARG_INT32(1), n_qids, Q = 0, d = 0;
        int32 *qids = freddy_glue_int_array(PG_GETARG_ARRAYTYPE_P(0), &n_qids);
        char   vecName[100], *sql, *cur;
        float *qs = NULL;
        int32 *found_ids = NULL;
        FreddyRows *r;
        /* the query vectors are the rows of the normalised table with id IN (...), duplicates collapsed, unknown ids dropped
         * (freddy.c:767-804).  ONE deliberate difference: ORDER BY id is appended.  The reference emits the queries in the
         * order SPI happens to return them (freddy.c:767-804, :986) -- a heap / index scan order that depends on the planner
         * and on where UPDATEs have moved tuples; the SRF's rows carry the query id, so the SET of rows is the same and only
         * the unspecified order of the result set is made canonical (ascending query id). */
        getTableName(NORMALIZED, vecName, 100);
        sql = palloc(200 + 12 * (n_qids > 0 ? n_qids : 1));
        cur = sql + sprintf(sql, "SELECT id, vector FROM %s WHERE id IN (", vecName);
        for (int i = 0; i < n_qids; i++) cur += sprintf(cur, i + 1 < n_qids ? "%d," : "%d", qids[i]);
        if (n_qids == 0) cur += sprintf(cur, "NULL");
        sprintf(cur, ") ORDER BY id");
        {
            MemoryContext caller = CurrentMemoryContext, old;
            SPI_connect();
            if (SPI_exec(sql, 0) > 0 && SPI_tuptable != NULL) {
                Q = SPI_processed;
                old = MemoryContextSwitchTo(caller);
                found_ids = palloc(sizeof(int32) * (Q > 0 ? Q : 1));
                for (int i = 0; i < Q; i++) {
                    bool   isnull, isnull2;
                    bytea *b = DatumGetByteaPP(SPI_getbinval(SPI_tuptable->vals[i], SPI_tuptable->tupdesc, 2, &isnull));
                    int    len = isnull ? -1 : (int) (VARSIZE_ANY_EXHDR(b) / sizeof(float4));
                    if (i == 0) { d = len; qs = MemoryContextAllocHuge(caller, sizeof(float) * (size_t) Q * (d > 0 ? d : 1)); }
                    found_ids[i] = DatumGetInt32(SPI_getbinval(SPI_tuptable->vals[i], SPI_tuptable->tupdesc, 1, &isnull2));
                    if (isnull || isnull2 || len != d || d <= 0) elog(ERROR, "freddy_gpu: NULL or ragged vector in %s", vecName);
                    memcpy(qs + (size_t) i * d, VARDATA_ANY(b), sizeof(float) * d);
                }
                MemoryContextSwitchTo(old);
            }
            SPI_finish();
        }
        r = begin_rows(fcinfo, &funcctx, Q, k, true);
        if (Q > 0) {
            freddy_gpu_index_t *h = freddy_glue_ivf();
            freddy_glue_check_dim(d, freddy_glue_dim(h));
            memcpy(r->query_ids, found_ids, sizeof(int32) * Q);
            /* replaces the whole while (!finished) loop, freddy.c:835-982: one cell per query and round (argmin from
             * minDist = 1000), rounds until k candidates were ACCEPTED (:971), list sentinel 100.0 */
            freddy_glue_check(freddy_gpu_ivfadc_search(h, qs, Q, k, 1, 100.0f, FREDDY_FOUND_BATCH_UDF, r->ids, r->dist));
        }
    }
    return freddy_glue_emit(fcinfo, SRF_PERCALL_SETUP());
}
