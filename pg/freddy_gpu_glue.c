/*
 * freddy_gpu_glue.c -- per-backend pin cache and marshalling for the GPU-backed FREDDY SRFs.
 * See freddy_gpu_glue.h.  The SPI loaders are the reference's own (index_utils.c), called ONCE per backend
 * instead of once per UDF call (freddy.c:69, :239-241, :746-749; ivpq_search_in.c:218-232).
 */
#include "freddy_gpu_glue.h"

#include "catalog/pg_type.h"
#include "executor/spi.h"
#include "storage/ipc.h"
#include "utils/builtins.h"
#include "utils/lsyscache.h"

#include "index_utils.h"

static freddy_gpu_index_t *h_pq = NULL, *h_ivf = NULL, *h_ivpq = NULL;
static bool exit_hook_set = false;

void freddy_glue_check(int rc)
{
    if (rc != FREDDY_OK)
        elog(ERROR, "%s", freddy_gpu_last_error());   /* e.g. "Unknown computation method!" (ivpq_search_in.c:376) */
}

void freddy_glue_unpin_all(void)
{
    if (h_pq)   { freddy_gpu_unpin(h_pq);   h_pq = NULL; }
    if (h_ivf)  { freddy_gpu_unpin(h_ivf);  h_ivf = NULL; }
    if (h_ivpq) { freddy_gpu_unpin(h_ivpq); h_ivpq = NULL; }
}

static void on_exit_unpin(int code, Datum arg) { freddy_glue_unpin_all(); }

static void ensure_exit_hook(void)
{
    if (!exit_hook_set) { on_proc_exit(on_exit_unpin, (Datum) 0); exit_hook_set = true; }
}

/* ---- argument codecs ---------------------------------------------------------------------------- */
float *freddy_glue_bytea_f32(bytea *b, int *n)
{
    int    len = (VARSIZE_ANY_EXHDR(b)) / sizeof(float4);
    float *out = palloc(sizeof(float) * (len > 0 ? len : 1));
    memcpy(out, VARDATA_ANY(b), sizeof(float) * len);
    *n = len;
    return out;
}

int32 *freddy_glue_int_array(ArrayType *a, int *n)
{
    Datum *elems; bool *nulls; int16 typlen; bool typbyval; char typalign;
    int32 *out;
    get_typlenbyvalalign(ARR_ELEMTYPE(a), &typlen, &typbyval, &typalign);
    deconstruct_array(a, ARR_ELEMTYPE(a), typlen, typbyval, typalign, &elems, &nulls, n);
    out = palloc(sizeof(int32) * (*n > 0 ? *n : 1));
    for (int i = 0; i < *n; i++) out[i] = DatumGetInt32(elems[i]);
    return out;
}

float *freddy_glue_bytea_array_f32(ArrayType *a, int *rows, int *dim)
{
    Datum *elems; bool *nulls; int16 typlen; bool typbyval; char typalign;
    float *out = NULL;
    get_typlenbyvalalign(ARR_ELEMTYPE(a), &typlen, &typbyval, &typalign);
    deconstruct_array(a, ARR_ELEMTYPE(a), typlen, typbyval, typalign, &elems, &nulls, rows);
    *dim = 0;
    for (int i = 0; i < *rows; i++) {
        bytea *b = DatumGetByteaP(elems[i]);
        int    len = VARSIZE_ANY_EXHDR(b) / sizeof(float4);
        if (i == 0) { *dim = len; out = palloc(sizeof(float) * (size_t) (*rows) * (len > 0 ? len : 1)); }
        if (len != *dim) elog(ERROR, "query vectors of different dimensionality");
        memcpy(out + (size_t) i * len, VARDATA_ANY(b), sizeof(float) * len);
    }
    return out ? out : palloc(sizeof(float));
}

/* ---- table flattening ---------------------------------------------------------------------------- */
/* CodebookCompound (entries carrying their own pos / code, index_utils.h:52-56) -> dense [m][K][s] */
static float *dense_codebook(CodebookCompound cb, int s)
{
    float *out = palloc0(sizeof(float) * (size_t) cb.positions * cb.codeSize * s);
    for (int i = 0; i < cb.positions * cb.codeSize; i++)
        memcpy(out + ((size_t) cb.codebook[i].pos * cb.codeSize + cb.codebook[i].code) * s, cb.codebook[i].vector, sizeof(float) * s);
    return out;
}

/* "SELECT id, [coarse_id,] vector FROM <table> ORDER BY ..." -> ids, [cells,] int16 codes, all palloc'd in the
 * caller's context.  Returns the number of rows; *m = codes per row. */
static int64 fetch_code_rows(const char *sql, bool with_cell, int32 **ids, int32 **cells, int16 **codes, int *m)
{
    int64 n;
    MemoryContext caller = CurrentMemoryContext, old;
    SPI_connect();
    if (SPI_exec(sql, 0) <= 0 || SPI_tuptable == NULL) { SPI_finish(); elog(ERROR, "freddy_gpu: cannot read the quantization table"); }
    n = SPI_processed;
    old = MemoryContextSwitchTo(caller);
    *ids = palloc(sizeof(int32) * (n > 0 ? n : 1));
    if (with_cell) *cells = palloc(sizeof(int32) * (n > 0 ? n : 1));
    *codes = NULL; *m = 0;
    for (int64 i = 0; i < n; i++) {
        bool      isnull;
        HeapTuple t = SPI_tuptable->vals[i];
        bytea    *b;
        int       len;
        (*ids)[i] = DatumGetInt32(SPI_getbinval(t, SPI_tuptable->tupdesc, 1, &isnull));
        if (with_cell) (*cells)[i] = DatumGetInt32(SPI_getbinval(t, SPI_tuptable->tupdesc, 2, &isnull));
        b = DatumGetByteaP(SPI_getbinval(t, SPI_tuptable->tupdesc, with_cell ? 3 : 2, &isnull));
        len = VARSIZE_ANY_EXHDR(b) / sizeof(int16);
        if (i == 0) { *m = len; *codes = palloc(sizeof(int16) * (size_t) n * len); }
        if (len != *m) elog(ERROR, "freddy_gpu: code rows of different lengths");
        memcpy(*codes + (size_t) i * len, VARDATA_ANY(b), sizeof(int16) * len);
    }
    MemoryContextSwitchTo(old);
    SPI_finish();
    if (*codes == NULL) *codes = palloc(sizeof(int16));
    return n;
}

freddy_gpu_index_t *freddy_glue_pq(void)
{
    if (h_pq == NULL) {
        char  cbName[100], qName[100], sql[256];
        CodebookCompound cb;
        int32 *ids; int16 *codes; int m, s; int64 n;
        freddy_pq_desc desc;
        getTableName(CODEBOOK, cbName, 100);
        getTableName(PQ_QUANTIZATION, qName, 100);
        cb = getCodebook(cbName);
        snprintf(sql, sizeof sql, "SELECT id, vector FROM %s ORDER BY id", qName);   /* canonical scan order */
        n = fetch_code_rows(sql, false, &ids, NULL, &codes, &m);
        {   /* sub-vector size from the first entry: d = positions * s */
            char q[256]; bool isnull; SPI_connect();
            snprintf(q, sizeof q, "SELECT octet_length(vector) / 4 FROM %s LIMIT 1", cbName);
            SPI_exec(q, 1);
            s = DatumGetInt32(SPI_getbinval(SPI_tuptable->vals[0], SPI_tuptable->tupdesc, 1, &isnull));
            SPI_finish();
        }
        desc.d = cb.positions * s; desc.m = cb.positions; desc.K = cb.codeSize; desc.N = n;
        desc.codebook = dense_codebook(cb, s); desc.ids = ids; desc.codes = codes;
        freddy_glue_check(freddy_gpu_pin_pq(&desc, 0, &h_pq));
        ensure_exit_hook();
    }
    return h_pq;
}

freddy_gpu_index_t *freddy_glue_ivf(void)
{
    if (h_ivf == NULL) {
        char  cbName[100], fqName[100], sql[300];
        CodebookCompound cb; CoarseQuantizer cq; int C;
        int32 *ids, *cells, *list_off; int16 *codes; int m, s, d; int64 n;
        float *coarse;
        freddy_ivf_desc desc;
        getTableName(RESIDUAL_CODEBOOK, cbName, 100);
        getTableName(RESIDUAL_QUANTIZATION, fqName, 100);
        cb = getCodebook(cbName);
        cq = getCoarseQuantizer(&C);
        /* inverted lists: rows grouped by coarse id, ascending id inside (the canonical order of freddy.c:324-342) */
        snprintf(sql, sizeof sql, "SELECT id, coarse_id, vector FROM %s ORDER BY coarse_id, id", fqName);
        n = fetch_code_rows(sql, true, &ids, &cells, &codes, &m);
        {
            char q[256]; bool isnull; SPI_connect();
            snprintf(q, sizeof q, "SELECT octet_length(vector) / 4 FROM %s LIMIT 1", cbName);
            SPI_exec(q, 1);
            s = DatumGetInt32(SPI_getbinval(SPI_tuptable->vals[0], SPI_tuptable->tupdesc, 1, &isnull));
            SPI_finish();
        }
        d = cb.positions * s;
        coarse = palloc0(sizeof(float) * (size_t) C * d);      /* array index == coarse id (freddy.c:309,873) */
        for (int i = 0; i < C; i++) memcpy(coarse + (size_t) cq[i].id * d, cq[i].vector, sizeof(float) * d);
        list_off = palloc0(sizeof(int32) * (C + 1));
        for (int64 i = 0; i < n; i++) list_off[cells[i] + 1]++;
        for (int c = 0; c < C; c++) list_off[c + 1] += list_off[c];
        desc.d = d; desc.m = cb.positions; desc.K = cb.codeSize; desc.C = C; desc.N = n;
        desc.coarse = coarse; desc.codebook = dense_codebook(cb, s); desc.list_off = list_off; desc.ids = ids; desc.codes = codes;
        freddy_glue_check(freddy_gpu_pin_ivf(&desc, 0, &h_ivf));
        ensure_exit_hook();
    }
    return h_ivf;
}

freddy_gpu_index_t *freddy_glue_ivpq(void)
{
    if (h_ivpq == NULL) {
        char  cbName[100], cqName[100], fqName[100], vecName[100], sql[400];
        CodebookCompound cb, cq;
        int32 *ids, *cells; int16 *codes; int m, s, d; int64 n;
        float *vectors, *stats;
        freddy_ivpq_desc desc;
        getTableName(IVPQ_CODEBOOK, cbName, 100);
        getTableName(COARSE_QUANTIZATION_MULTI, cqName, 100);
        getTableName(IVPQ_QUANTIZATION, fqName, 100);
        getTableName(NORMALIZED, vecName, 100);
        cb = getCodebook(cbName);
        cq = getCodebook(cqName);
        stats = getStatistics();                               /* [cells + 1], last = total count (index_utils.c:632-665) */
        snprintf(sql, sizeof sql, "SELECT id, coarse_id, vector FROM %s ORDER BY id", fqName);
        n = fetch_code_rows(sql, true, &ids, &cells, &codes, &m);
        {
            char q[256]; bool isnull; SPI_connect();
            snprintf(q, sizeof q, "SELECT octet_length(vector) / 4 FROM %s LIMIT 1", cbName);
            SPI_exec(q, 1);
            s = DatumGetInt32(SPI_getbinval(SPI_tuptable->vals[0], SPI_tuptable->tupdesc, 1, &isnull));
            SPI_finish();
        }
        d = cb.positions * s;
        {   /* the vectors the reference JOINs in for methods 1 and 2 (ivpq_search_in.c:361-371), row-aligned with ids */
            MemoryContext caller = CurrentMemoryContext, old;
            SPI_connect();
            snprintf(sql, sizeof sql, "SELECT v.vector FROM %s AS fq INNER JOIN %s AS v ON fq.id = v.id ORDER BY fq.id", fqName, vecName);
            if (SPI_exec(sql, 0) <= 0 || (int64) SPI_processed != n) { SPI_finish(); elog(ERROR, "freddy_gpu: every ivpq row needs its vector"); }
            old = MemoryContextSwitchTo(caller);
            vectors = palloc(sizeof(float) * (size_t) (n > 0 ? n : 1) * d);
            for (int64 i = 0; i < n; i++) {
                bool isnull;
                bytea *b = DatumGetByteaP(SPI_getbinval(SPI_tuptable->vals[i], SPI_tuptable->tupdesc, 1, &isnull));
                memcpy(vectors + (size_t) i * d, VARDATA_ANY(b), sizeof(float) * d);
            }
            MemoryContextSwitchTo(old);
            SPI_finish();
        }
        desc.d = d; desc.m = cb.positions; desc.K = cb.codeSize;
        desc.coarse_positions = cq.positions; desc.coarse_codes = cq.codeSize; desc.N = n;
        desc.codebook = dense_codebook(cb, s); desc.coarse = dense_codebook(cq, d / cq.positions);
        desc.ids = ids; desc.coarse_id = cells; desc.codes = codes; desc.vectors = vectors; desc.stats = stats;
        freddy_glue_check(freddy_gpu_pin_ivpq(&desc, 0, &h_ivpq));
        ensure_exit_hook();
    }
    return h_ivpq;
}

/* ---- value-per-call emission --------------------------------------------------------------------- */
Datum freddy_glue_emit(FunctionCallInfo fcinfo, FuncCallContext *funcctx)
{
    FreddyRows *r = (FreddyRows *) funcctx->user_fctx;
    if (r->iter < r->n_rows) {
        char  buf[3][16];
        char *values[3];
        int   c = 0;
        HeapTuple tuple;
        if (r->query_ids) { snprintf(buf[c], 16, "%d", r->query_ids[r->iter / r->k]); values[c] = buf[c]; c++; }
        snprintf(buf[c], 16, "%d", r->ids[r->iter]);  values[c] = buf[c]; c++;
        snprintf(buf[c], 16, "%f", r->dist[r->iter]); values[c] = buf[c]; c++;
        r->iter++;
        tuple = BuildTupleFromCStrings(funcctx->attinmeta, values);
        SRF_RETURN_NEXT(funcctx, HeapTupleGetDatum(tuple));
    }
    SRF_RETURN_DONE(funcctx);
}
