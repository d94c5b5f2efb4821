/*
 * freddy_insert.c -- insert_batch(varchar[]) and grouping_pq(integer[], integer[]) of freddy_extension/freddy.c as
 * hosts over libfreddy_gpu.so.  Same PG_FUNCTION_INFO_V1 symbols and SQL signatures (freddy--0.0.1.sql:380-424);
 * the reference's own copies in freddy.c step aside by -D renames (pg/Makefile).
 *
 *   insert_batch(varchar[])                     freddy.c:1403-1658
 *     unchanged (SQL / SPI): the tokenisation query for the terms not yet in the vocabulary (:1503-1544), the
 *       codebooks with their counts and the coarse quantizers (:1545-1560), and the reference's own writers
 *       updateProductQuantizationRelation / updateCodebookRelation / updateWordVectorsRelation (index_utils.c:959-1074)
 *       -- row ids, "%f" / "%d" text and statement order are theirs;
 *     on the device: the quantisation of the new vectors -- PQ codes, coarse cell + residual codes, ivpq codes,
 *       multi-index cell (:1562-1623; updateCodebook's 1-NN search, index_utils.c:925-939) -- through
 *       freddy_gpu_insert_quantize: n x (3 codebooks x m x K + C) squareDistance evaluations;
 *     on the host: updateCodebook's bookkeeping for codes that are already known (index_utils.c:940-956: count
 *       increments and the running-mean update, a few float / double operations per new vector and position), restated
 *       statement by statement -- including its use of ONE nearestCentroidRaw pointer for all positions and of bucket
 *       [pos + code] -- so that the codebook rows written are the rows the reference writes;
 *     afterwards the pinned handles of THIS backend follow the tables at once (the glue's staleness check sees the
 *       generation bump and appends the new rows / reloads the codebooks in HBM: freddy_glue_pq/_ivf/_ivpq); other
 *       backends do the same at their next search.
 *
 *   grouping_pq(integer[], integer[])           freddy.c:1176-1401
 *     unchanged: argument parsing, the group vectors ("Group ids do not exist"), the (Ids, GroupIds) rows;
 *     on the device: one LUT per group and the argmin over groups for every requested row (:1288-1360) =
 *       freddy_gpu_grouping_pq on the pinned pq_quantization table.
 *
 * NOT compiled in the build image of this repository (no PostgreSQL headers there).
 */
#include "freddy_gpu_glue.h"

#include "catalog/pg_type.h"
#include "executor/spi.h"
#include "utils/builtins.h"

#include "index_utils.h"
#include "freddy_pure.h"

/* dense [m][K][s] copy of a codebook-with-counts (entries carry their own pos / code) */
static float *dense_of(CodebookWithCounts cb, int m, int K, int s)
{
    float *out = palloc0(sizeof(float) * (size_t) m * K * s);
    for (int i = 0; i < m * K; i++) {
        if (cb[i].pos < 0 || cb[i].pos >= m || cb[i].code < 0 || cb[i].code >= K) elog(ERROR, "freddy_gpu: codebook entry out of range");
        memcpy(out + ((size_t) cb[i].pos * K + cb[i].code) * s, cb[i].vector, sizeof(float) * s);
    }
    return out;
}

/* updateCodebook's bookkeeping for codes found on the device: freddy_update_codebook_known_codes (freddy_pure.h -- the
 * PostgreSQL-free part, compiled and compared with the oracle's literal restatement in this repository's tests).  It works
 * on the reference's own entries: the two struct layouts are the same. */
StaticAssertDecl(sizeof(FreddyCbEntry) == sizeof(CodebookEntryComplete) && offsetof(FreddyCbEntry, vector) == offsetof(CodebookEntryComplete, vector) &&
                 offsetof(FreddyCbEntry, count) == offsetof(CodebookEntryComplete, count), "FreddyCbEntry mirrors CodebookEntryComplete");
static void update_codebook_known_codes(int rawVectorsSize, int subvectorSize, CodebookWithCounts cb, int cbPositions, int cbCodes,
                                        const int16 *codes, int **nearestCentroids, int *countIncs)
{
    /* the row writer (updateProductQuantizationRelation) takes one int pointer per row: rows of one flat array */
    int *row_codes = palloc(sizeof(int) * (size_t) Max(rawVectorsSize, 1) * cbPositions);
    freddy_update_codebook_known_codes(rawVectorsSize, subvectorSize, (FreddyCbEntry *) cb, cbPositions, cbCodes, codes, row_codes, countIncs);
    for (int i = 0; i < rawVectorsSize; i++) nearestCentroids[i] = row_codes + (size_t) i * cbPositions;
}

PG_FUNCTION_INFO_V1(insert_batch);
Datum insert_batch(PG_FUNCTION_ARGS)
{
    Datum *termsData; int n = 0;
    char **inputTerms; int inputTermsPlaneSize = 0;
    char  *command, *cur;
    float4 **rawVectors, **rawVectorsUnnormalized; char **tokens;
    int    rawVectorsSize = 0, vectorSize = 0;
    char   tCodebook[100], tPq[100], tResCodebook[100], tFine[100], tNorm[100], tOrig[100], tIvpq[100], tIvpqCodebook[100], tCQMulti[100];

    getTableName(CODEBOOK, tCodebook, 100);            getTableName(PQ_QUANTIZATION, tPq, 100);
    getTableName(RESIDUAL_CODEBOOK, tResCodebook, 100); getTableName(RESIDUAL_QUANTIZATION, tFine, 100);
    getTableName(NORMALIZED, tNorm, 100);              getTableName(ORIGINAL, tOrig, 100);
    getTableName(IVPQ_QUANTIZATION, tIvpq, 100);       getTableName(IVPQ_CODEBOOK, tIvpqCodebook, 100);
    getTableName(COARSE_QUANTIZATION_MULTI, tCQMulti, 100);

    /* terms from the argument, and their tokenisation -- the reference's own query (freddy.c:1483-1544) */
    getArray(PG_GETARG_ARRAYTYPE_P(0), &termsData, &n);
    inputTerms = palloc(sizeof(char *) * (n > 0 ? n : 1));
    for (int j = 0; j < n; j++) {
        inputTerms[j] = text_to_cstring((text *) DatumGetPointer(termsData[j]));
        inputTermsPlaneSize += strlen(inputTerms[j]);
    }
    command = palloc(inputTermsPlaneSize * 3 + 2 * n + 400);
    cur = command + sprintf(command, "SELECT replace(term, ' ', '_') AS token, tokenize(term), tokenize_raw(term) FROM unnest('{");
    for (int i = 0; i < n; i++) cur += sprintf(cur, i + 1 < n ? "%s, " : "%s", inputTerms[i]);
    sprintf(cur, "}'::varchar(100)[]) AS term WHERE NOT replace(term, ' ', '_') IN (SELECT word FROM %s)", tNorm);
    {
        MemoryContext caller = CurrentMemoryContext, old;
        if (SPI_connect() != SPI_OK_CONNECT) elog(ERROR, "freddy_gpu: SPI_connect failed");
        if (SPI_exec(command, 0) <= 0 || SPI_tuptable == NULL) { SPI_finish(); elog(ERROR, "freddy_gpu: the tokenisation query failed"); }
        rawVectorsSize = (int) SPI_processed;
        old = MemoryContextSwitchTo(caller);
        rawVectors = palloc(sizeof(float4 *) * (rawVectorsSize > 0 ? rawVectorsSize : 1));
        rawVectorsUnnormalized = palloc(sizeof(float4 *) * (rawVectorsSize > 0 ? rawVectorsSize : 1));
        tokens = palloc(sizeof(char *) * (rawVectorsSize > 0 ? rawVectorsSize : 1));
        for (int i = 0; i < rawVectorsSize; i++) {
            bool   null1, null2;
            HeapTuple tuple = SPI_tuptable->vals[i];
            char  *token = SPI_getvalue(tuple, SPI_tuptable->tupdesc, 1);
            bytea *v = DatumGetByteaPP(SPI_getbinval(tuple, SPI_tuptable->tupdesc, 2, &null1));
            bytea *u = DatumGetByteaPP(SPI_getbinval(tuple, SPI_tuptable->tupdesc, 3, &null2));
            int    len;
            if (token == NULL || null1 || null2) elog(ERROR, "freddy_gpu: a term could not be tokenised");
            len = (int) (VARSIZE_ANY_EXHDR(v) / sizeof(float4));
            if (i == 0) vectorSize = len;
            if (len != vectorSize || (int) (VARSIZE_ANY_EXHDR(u) / sizeof(float4)) != vectorSize) elog(ERROR, "freddy_gpu: vectors of different dimensionality");
            tokens[i] = pstrdup(token);
            rawVectors[i] = palloc(sizeof(float4) * vectorSize);             memcpy(rawVectors[i], VARDATA_ANY(v), sizeof(float4) * vectorSize);
            rawVectorsUnnormalized[i] = palloc(sizeof(float4) * vectorSize); memcpy(rawVectorsUnnormalized[i], VARDATA_ANY(u), sizeof(float4) * vectorSize);
        }
        MemoryContextSwitchTo(old);
        SPI_finish();
    }
    if (rawVectorsSize == 0) PG_RETURN_INT32(0);

    {
        CodebookWithCounts cb, residualCb, ivpqCb;
        CodebookCompound   cqMulti;
        CoarseQuantizer    cq;
        int cbPositions = 0, cbCodes = 0, cbrPositions = 0, cbrCodes = 0, cbIvPositions = 0, cbIvCodes = 0, cqSize = 0;
        int subvectorSize, residualSubvectorSize, ivSubvectorSize;
        float *flat, *coarse;
        int16 *pq_codes, *res_codes, *iv_codes, *multi_codes;
        int32 *cqQuantizations; int *cqQuantizationMulti;
        int  **nearestCentroids, **nearestResidualCentroids, **nearestCentroidsIvpq;
        int   *countIncs, *residualCountIncs, *ivCountIncs;
        freddy_insert_desc desc;

        cb = getCodebookWithCounts(&cbPositions, &cbCodes, tCodebook);
        residualCb = getCodebookWithCounts(&cbrPositions, &cbrCodes, tResCodebook);
        ivpqCb = getCodebookWithCounts(&cbIvPositions, &cbIvCodes, tIvpqCodebook);
        cqMulti = getCodebook(tCQMulti);
        cq = getCoarseQuantizer(&cqSize);
        /* (getCodebookWithCounts returns sizes: largest pos / code + 1, index_utils.c:716-733) */
        if (vectorSize % cbPositions || vectorSize % cbrPositions || vectorSize % cbIvPositions || cqMulti.positions <= 0 ||
            vectorSize % cqMulti.positions || cqSize <= 0)
            elog(ERROR, "freddy_gpu: the codebooks do not divide %d dimensions", vectorSize);
        subvectorSize = vectorSize / cbPositions; residualSubvectorSize = vectorSize / cbrPositions; ivSubvectorSize = vectorSize / cbIvPositions;

        /* flat inputs for the device */
        flat = palloc(sizeof(float) * (size_t) rawVectorsSize * vectorSize);
        for (int i = 0; i < rawVectorsSize; i++) memcpy(flat + (size_t) i * vectorSize, rawVectors[i], sizeof(float) * vectorSize);
        coarse = palloc0(sizeof(float) * (size_t) cqSize * vectorSize);
        for (int i = 0; i < cqSize; i++) {
            if (cq[i].id < 0 || cq[i].id >= cqSize) elog(ERROR, "freddy_gpu: coarse id %d outside [0, %d)", cq[i].id, cqSize);
            memcpy(coarse + (size_t) cq[i].id * vectorSize, cq[i].vector, sizeof(float) * vectorSize);
        }
        memset(&desc, 0, sizeof desc);
        desc.d = vectorSize;
        desc.pq_m = cbPositions;     desc.pq_K = cbCodes;     desc.pq_codebook = dense_of(cb, cbPositions, cbCodes, subvectorSize);
        desc.res_m = cbrPositions;   desc.res_K = cbrCodes;   desc.residual_codebook = dense_of(residualCb, cbrPositions, cbrCodes, residualSubvectorSize);
        desc.C = cqSize;             desc.coarse = coarse;
        desc.ivpq_m = cbIvPositions; desc.ivpq_K = cbIvCodes; desc.ivpq_codebook = dense_of(ivpqCb, cbIvPositions, cbIvCodes, ivSubvectorSize);
        desc.multi_positions = cqMulti.positions; desc.multi_codes = cqMulti.codeSize;
        {
            int    ms = vectorSize / cqMulti.positions;
            float *mc = palloc0(sizeof(float) * (size_t) cqMulti.positions * cqMulti.codeSize * ms);
            for (int i = 0; i < cqMulti.positions * cqMulti.codeSize; i++)
                memcpy(mc + ((size_t) cqMulti.codebook[i].pos * cqMulti.codeSize + cqMulti.codebook[i].code) * ms, cqMulti.codebook[i].vector, sizeof(float) * ms);
            desc.coarse_multi = mc;
        }
        pq_codes = palloc(sizeof(int16) * (size_t) rawVectorsSize * cbPositions);
        res_codes = palloc(sizeof(int16) * (size_t) rawVectorsSize * cbrPositions);
        iv_codes = palloc(sizeof(int16) * (size_t) rawVectorsSize * cbIvPositions);
        multi_codes = palloc(sizeof(int16) * (size_t) rawVectorsSize * cqMulti.positions);
        cqQuantizations = palloc(sizeof(int32) * rawVectorsSize);
        /* freddy.c:1562-1623 and the 1-NN part of updateCodebook, for all new vectors at once */
        freddy_glue_check(freddy_gpu_insert_quantize(&desc, 0, flat, rawVectorsSize, pq_codes, cqQuantizations, res_codes, iv_codes, multi_codes));
        /* multi-index cell id: factor *= POSITIONS, as the reference has it (freddy.c:1599) */
        cqQuantizationMulti = palloc(sizeof(int) * rawVectorsSize);
        for (int i = 0; i < rawVectorsSize; i++) {
            int factor = 1;
            cqQuantizationMulti[i] = 0;
            for (int pos = 0; pos < cqMulti.positions; pos++) { cqQuantizationMulti[i] += factor * multi_codes[(size_t) i * cqMulti.positions + pos]; factor *= cqMulti.positions; }
        }

        nearestCentroids = palloc(sizeof(int *) * rawVectorsSize);
        countIncs = palloc(cbPositions * cbCodes * sizeof(int));
        update_codebook_known_codes(rawVectorsSize, subvectorSize, cb, cbPositions, cbCodes, pq_codes, nearestCentroids, countIncs);
        nearestResidualCentroids = palloc(sizeof(int *) * rawVectorsSize);
        residualCountIncs = palloc(cbrPositions * cbrCodes * sizeof(int));
        update_codebook_known_codes(rawVectorsSize, residualSubvectorSize, residualCb, cbrPositions, cbrCodes, res_codes, nearestResidualCentroids, residualCountIncs);
        nearestCentroidsIvpq = palloc(sizeof(int *) * rawVectorsSize);
        ivCountIncs = palloc(cbIvPositions * cbIvCodes * sizeof(int));
        update_codebook_known_codes(rawVectorsSize, ivSubvectorSize, ivpqCb, cbIvPositions, cbIvCodes, iv_codes, nearestCentroidsIvpq, ivCountIncs);

        /* the reference's writers, in the reference's order (freddy.c:1625-1655) */
        updateProductQuantizationRelation(nearestCentroids, tokens, cbPositions, cb, tPq, rawVectorsSize, NULL);
        updateProductQuantizationRelation(nearestResidualCentroids, tokens, cbrPositions, residualCb, tFine, rawVectorsSize, (int *) cqQuantizations);
        updateProductQuantizationRelation(nearestCentroidsIvpq, NULL, cbIvPositions, ivpqCb, tIvpq, rawVectorsSize, cqQuantizationMulti);
        updateCodebookRelation(cb, cbPositions, cbCodes, tCodebook, countIncs, subvectorSize);
        updateCodebookRelation(residualCb, cbrPositions, cbrCodes, tResCodebook, residualCountIncs, residualSubvectorSize);
        updateCodebookRelation(ivpqCb, cbIvPositions, cbIvCodes, tIvpqCodebook, ivCountIncs, ivSubvectorSize);
        updateWordVectorsRelation(tNorm, tokens, rawVectors, rawVectorsSize, vectorSize);
        updateWordVectorsRelation(tOrig, tokens, rawVectorsUnnormalized, rawVectorsSize, vectorSize);
    }
    /* HBM follows the tables now: the staleness check of each handle this backend has pinned sees the new generation
     * and appends the rows / reloads the codebook (a handle that is not pinned yet is left alone) */
    CommandCounterIncrement();
    freddy_glue_refresh_pinned();
    PG_RETURN_INT32(0);
}

/* ---- grouping_pq ---------------------------------------------------------------------------------------- */
static int cmp_int(const void *a, const void *b) { return (*(const int *) a > *(const int *) b) - (*(const int *) a < *(const int *) b); }

typedef struct GroupRows { int32 *ids, *group; int64 n, iter; } GroupRows;

PG_FUNCTION_INFO_V1(grouping_pq);
Datum grouping_pq(PG_FUNCTION_ARGS)
{
    FuncCallContext *funcctx;
    GroupRows *g;
    if (SRF_IS_FIRSTCALL()) {
        MemoryContext old;
        int    n_ids, n_groups, dim = 0;
        int32 *ids, *groups, *out_ids, *out_group; int64 n_out = 0;
        float *gv = NULL;
        char   tNorm[100], *sql, *cur;
        TupleDesc tupdesc;
        freddy_gpu_index_t *h;
        funcctx = SRF_FIRSTCALL_INIT();
        old = MemoryContextSwitchTo(funcctx->multi_call_memory_ctx);
        ids = freddy_glue_int_array(PG_GETARG_ARRAYTYPE_P(0), &n_ids);
        groups = freddy_glue_int_array(PG_GETARG_ARRAYTYPE_P(1), &n_groups);
        qsort(groups, n_groups, sizeof(int32), cmp_int);                                   /* freddy.c:1238 */
        getTableName(NORMALIZED, tNorm, 100);
        sql = palloc(200 + 12 * (n_groups > 0 ? n_groups : 1));
        cur = sql + sprintf(sql, "SELECT id, vector FROM %s WHERE id IN (", tNorm);
        for (int i = 0; i < n_groups; i++) cur += sprintf(cur, i + 1 < n_groups ? "%d, " : "%d", groups[i]);
        if (n_groups == 0) cur += sprintf(cur, "NULL");
        sprintf(cur, ") ORDER BY id ASC");                                                   /* :1254 */
        if (SPI_connect() != SPI_OK_CONNECT) elog(ERROR, "freddy_gpu: SPI_connect failed");
        if (SPI_exec(sql, 0) <= 0 || (int) SPI_processed != n_groups) { SPI_finish(); elog(ERROR, "Group ids do not exist"); }   /* :1261 */
        for (int i = 0; i < n_groups; i++) {
            bool isnull;
            bytea *b = DatumGetByteaPP(SPI_getbinval(SPI_tuptable->vals[i], SPI_tuptable->tupdesc, 2, &isnull));
            int len = isnull ? -1 : (int) (VARSIZE_ANY_EXHDR(b) / sizeof(float4));
            if (i == 0) { dim = len; gv = MemoryContextAlloc(funcctx->multi_call_memory_ctx, sizeof(float) * (size_t) n_groups * (dim > 0 ? dim : 1)); }
            if (len != dim || dim <= 0) elog(ERROR, "freddy_gpu: NULL or ragged group vector");
            memcpy(gv + (size_t) i * dim, VARDATA_ANY(b), sizeof(float) * dim);
        }
        SPI_finish();
        h = freddy_glue_pq();
        if (n_groups > 0) freddy_glue_check_dim(dim, freddy_glue_dim(h));
        out_ids = palloc(sizeof(int32) * (n_ids > 0 ? n_ids : 1));
        out_group = palloc(sizeof(int32) * (n_ids > 0 ? n_ids : 1));
        /* replaces freddy.c:1288-1360: LUT per group, ADC of every requested row against every group, first nearest wins */
        if (n_ids > 0 && n_groups > 0)
            freddy_glue_check(freddy_gpu_grouping_pq(h, gv, n_groups, ids, n_ids, out_ids, out_group, &n_out));
        g = palloc0(sizeof(GroupRows));
        g->ids = out_ids; g->group = out_group; g->n = n_out; g->iter = 0;
        for (int64 i = 0; i < n_out; i++) g->group[i] = g->group[i] >= 0 ? groups[g->group[i]] : -1;   /* index -> group id (:1386) */
        funcctx->user_fctx = g;
        tupdesc = CreateTemplateTupleDesc(2);
        TupleDescInitEntry(tupdesc, 1, "Ids", INT4OID, -1, 0);
        TupleDescInitEntry(tupdesc, 2, "GroupIds", INT4OID, -1, 0);
        funcctx->attinmeta = TupleDescGetAttInMetadata(tupdesc);
        MemoryContextSwitchTo(old);
    }
    funcctx = SRF_PERCALL_SETUP();
    g = (GroupRows *) funcctx->user_fctx;
    if (g->iter < g->n) {
        char  b0[16], b1[16];
        char *values[2] = {b0, b1};
        HeapTuple t;
        snprintf(b0, 16, "%d", g->ids[g->iter]);
        snprintf(b1, 16, "%d", g->group[g->iter]);
        g->iter++;
        t = BuildTupleFromCStrings(funcctx->attinmeta, values);
        SRF_RETURN_NEXT(funcctx, HeapTupleGetDatum(t));
    }
    SRF_RETURN_DONE(funcctx);
}
