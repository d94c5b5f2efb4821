/*
 * ivpq_search_in.c -- the kNN-join SRF (freddy_extension/ivpq_search_in.c:59-721) as a thin host.
 *
 *   ivpq_search_in(bytea[] query_vectors, int[] query_ids, int k, int[] target_ids, int alpha, int pvf,
 *                  int method, bool use_target_lists, float4 confidence, int double_threshold)
 *     -> SETOF (QueryId int4, TargetId int4, Distance float4)
 *
 * Arguments are read as today (:168-208); everything between the table loads and the emit block (:211-684:
 * LUTs, multi-index cell selection with statistics, the per-iteration SELECT, target lists, ADC / exact
 * distances, post verification, the alpha-doubling loop) is one call of freddy_gpu_knn_join on the tables
 * pinned once per backend.  The stage timers the evaluation scripts scrape (elog(INFO, "TRACK <stage> %f"),
 * :234-697, evaluation/tracking.py) are re-emitted from freddy_gpu_last_track under the same names.
 * NOT compiled in this repository's image (no PostgreSQL headers).
 */
#include "freddy_gpu_glue.h"

#include "utils/builtins.h"

PG_FUNCTION_INFO_V1(ivpq_search_in);
Datum ivpq_search_in(PG_FUNCTION_ARGS)
{
    if (SRF_IS_FIRSTCALL()) {
        FuncCallContext *funcctx;
        int    Q, dim, n_qids, n_targets, iterations = 0;
        float *qs = freddy_glue_bytea_array_f32(PG_GETARG_ARRAYTYPE_P(0), &Q, &dim);
        int32 *qids = freddy_glue_int_array(PG_GETARG_ARRAYTYPE_P(1), &n_qids);
        int    k = PG_GETARG_INT32(2);
        int32 *targets = freddy_glue_int_array(PG_GETARG_ARRAYTYPE_P(3), &n_targets);
        int    alpha = PG_GETARG_INT32(4), pvf = PG_GETARG_INT32(5), method = PG_GETARG_INT32(6);
        bool   use_target_lists = PG_GETARG_BOOL(7);
        float4 confidence = PG_GETARG_FLOAT4(8);
        int    double_threshold = PG_GETARG_INT32(9);
        freddy_track t;
        freddy_gpu_index_t *h;
        FreddyRows *r;
        if (n_qids != Q)                                                                   /* :180 */
            elog(ERROR, "Number of query vectors and query vector ids differs! ( %d, %d)", n_qids, Q);
        {
            MemoryContext old;
            TupleDesc tupdesc;
            funcctx = SRF_FIRSTCALL_INIT();
            old = MemoryContextSwitchTo(funcctx->multi_call_memory_ctx);
            if (get_call_result_type(fcinfo, NULL, &tupdesc) != TYPEFUNC_COMPOSITE)
                ereport(ERROR, (errcode(ERRCODE_FEATURE_NOT_SUPPORTED),
                                errmsg("function returning record called in context that cannot accept type record")));
            funcctx->attinmeta = TupleDescGetAttInMetadata(tupdesc);
            r = palloc0(sizeof(FreddyRows));
            r->k = k; r->n_rows = Q * k;
            r->ids = palloc(sizeof(int32) * (r->n_rows > 0 ? r->n_rows : 1));
            r->dist = palloc(sizeof(float) * (r->n_rows > 0 ? r->n_rows : 1));
            r->query_ids = palloc(sizeof(int32) * (Q > 0 ? Q : 1));
            memcpy(r->query_ids, qids, sizeof(int32) * Q);
            funcctx->user_fctx = r;
            MemoryContextSwitchTo(old);
        }
        h = freddy_glue_ivpq();
        if (Q > 0) freddy_glue_check_dim(dim, freddy_glue_dim(h));
        freddy_glue_check(freddy_gpu_knn_join(h, qs, Q, k, targets, n_targets, alpha, pvf, method,
                                              use_target_lists ? 1 : 0, confidence, double_threshold, r->ids, r->dist, &iterations));
        memset(&t, 0, sizeof t);
        if (freddy_gpu_last_track_sized(h, &t, sizeof t) > 0) {   /* (sized: a library newer than this host's header cannot overrun t) */
            elog(INFO, "TRACK precomputation_time %f", t.precomputation_time);                                   /* :294 */
            elog(INFO, "TRACK determine_coarse_quantization_time %f", t.determine_coarse_quantization_time);     /* :341 */
            elog(INFO, "TRACK query_construction_time %f", t.query_construction_time);                           /* :397 */
            elog(INFO, "TRACK data_retrieval_time %f", t.data_retrieval_time);                                   /* :403 */
            elog(INFO, "TRACK computation_time %f", t.computation_time);                                         /* :632 */
            elog(INFO, "TRACK pv_computation_time %f", t.pv_computation_time);                                   /* :627 */
            elog(INFO, "TRACK recalculate_query_indices_time %f", t.recalculate_query_indices_time);             /* :671 */
            elog(INFO, "TRACK total_time %f", t.total_time);                                                     /* :697 */
        }
    }
    return freddy_glue_emit(fcinfo, SRF_PERCALL_SETUP());
}
