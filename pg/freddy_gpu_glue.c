/*
 * freddy_gpu_glue.c -- per-backend pin cache, staleness tracking and marshalling for the GPU-backed FREDDY SRFs.
 * See freddy_gpu_glue.h.  The SPI loaders are the reference's own (index_utils.c), called when a handle is
 * (re)built instead of once per UDF call (freddy.c:69, :239-241, :746-749; ivpq_search_in.c:218-232).
 *
 * Staleness.  The reference sees the current tables on every call; a pinned copy must be checked.  Every handle
 * remembers what it was built from (FreddyStamp: the tables' OIDs and relfilenodes, their generation counters from
 * pg/freddy_gpu_watch.sql or -- without that script -- size / max(id) / sum(count), and the largest pinned id).
 * freddy_glue_pq / _ivf / _ivpq compare the stamp before EVERY search:
 *   nothing changed                          -> the pinned handle
 *   only INSERTs into the row table          -> the rows with id > max pinned id are fetched and appended in HBM
 *   the codebook table was UPDATEd           -> the codebook is re-read and its derived tables rebuilt in HBM
 *   anything else (other OID / relfilenode, DELETE / UPDATE / TRUNCATE of rows, coarse quantizer changed)
 *                                            -> unpin, pin again
 * NOT compiled in the build image of this repository (no PostgreSQL headers there).
 */
#include "freddy_gpu_glue.h"

#include "access/htup_details.h"
#include "catalog/namespace.h"
#include "catalog/pg_type.h"
#include "executor/spi.h"
#include "storage/ipc.h"
#include "utils/builtins.h"
#include "utils/lsyscache.h"
#include "utils/memutils.h"
#include "utils/rel.h"

#include "index_utils.h"

#include "access/xact.h"   /* RegisterXactCallback, GetCurrentTransactionNestLevel */
#include "freddy_pure.h"   /* FreddyStamp, freddy_compare_stamp, payload codecs: the PostgreSQL-free parts (compiled and tested in this repository) */

#define MAX_TABS FREDDY_MAX_TABS

typedef struct FreddyPin {
    freddy_gpu_index_t *h;
    FreddyStamp         st;
    int                 mut_level;   /* nesting level of the open (sub)transaction that last changed the pinned copy; 0 = none (freddy_pure.h) */
} FreddyPin;

static FreddyPin pin_pq, pin_ivf, pin_ivpq;
static bool exit_hook_set = false;
/* One pinned (freddy_gpu_host_alloc) buffer per backend for decoded query batches, grown on demand and reused: the
 * library reads a batch from it without its staging copy.  A search finishes inside the SRF's first call, so the
 * buffer is free again before any other call can want it; an elog(ERROR) in between leaks nothing. */
static float *pinned_q = NULL;
static Size   pinned_q_cap = 0;

void freddy_glue_check(int rc)
{
    if (rc != FREDDY_OK)
        elog(ERROR, "%s", freddy_gpu_last_error());   /* e.g. "Unknown computation method!" (ivpq_search_in.c:376) */
}

static void drop_pin(FreddyPin *p)
{
    if (p->h) { freddy_gpu_unpin(p->h); p->h = NULL; }
    memset(&p->st, 0, sizeof p->st);
    p->mut_level = 0;
}

/* A handle that was pinned / appended to / given a new codebook inside a transaction holds what that transaction saw.
 * If the (sub)transaction aborts, those rows never existed: the handle goes (the next search pins again); on commit the
 * change belongs to the parent level.  Decisions: freddy_pure.h (tested in tests/test_pg_pure.py). */
static void mark_mutated(FreddyPin *p) { p->mut_level = GetCurrentTransactionNestLevel(); }
static void pins_after_abort(int level)
{
    FreddyPin *all[3] = {&pin_pq, &pin_ivf, &pin_ivpq};
    for (int i = 0; i < 3; i++)
        if (all[i]->h && !freddy_pin_survives_abort(all[i]->mut_level, level)) drop_pin(all[i]);
}
static void pins_after_commit(int level)
{
    FreddyPin *all[3] = {&pin_pq, &pin_ivf, &pin_ivpq};
    for (int i = 0; i < 3; i++) all[i]->mut_level = freddy_pin_level_after_commit(all[i]->mut_level, level);
}
static void freddy_xact_cb(XactEvent event, void *arg)
{
    if (event == XACT_EVENT_ABORT || event == XACT_EVENT_PARALLEL_ABORT) pins_after_abort(1);
    else if (event == XACT_EVENT_COMMIT || event == XACT_EVENT_PARALLEL_COMMIT) pins_after_commit(1);
    /* PREPARE TRANSACTION: the rows may still be rolled back (ROLLBACK PREPARED) -- a pin that holds rows of this transaction is
     * dropped like after an abort and pinned again from whatever has committed by then */
    else if (event == XACT_EVENT_PREPARE) pins_after_abort(1);
}
static void freddy_subxact_cb(SubXactEvent event, SubTransactionId mySubid, SubTransactionId parentSubid, void *arg)
{
    /* (callbacks run while the aborting / committing subtransaction is still current: its own nesting level) */
    if (event == SUBXACT_EVENT_ABORT_SUB) pins_after_abort(GetCurrentTransactionNestLevel());
    else if (event == SUBXACT_EVENT_COMMIT_SUB) pins_after_commit(GetCurrentTransactionNestLevel());
}

void freddy_glue_unpin_all(void)
{
    drop_pin(&pin_pq);
    drop_pin(&pin_ivf);
    drop_pin(&pin_ivpq);
}

static void on_exit_unpin(int code, Datum arg);
static void ensure_exit_hook(void);

static void on_exit_unpin(int code, Datum arg)
{
    freddy_glue_unpin_all();
    if (pinned_q) { freddy_gpu_host_free(pinned_q); pinned_q = NULL; pinned_q_cap = 0; }
}

static float *query_buffer(Size bytes)
{
    if (bytes > pinned_q_cap) {
        if (pinned_q) { freddy_gpu_host_free(pinned_q); pinned_q = NULL; pinned_q_cap = 0; }
        freddy_glue_check(freddy_gpu_host_alloc((void **) &pinned_q, bytes + bytes / 4 + 4096));
        pinned_q_cap = bytes + bytes / 4 + 4096;
        ensure_exit_hook();
    }
    return pinned_q;
}

static void ensure_exit_hook(void)
{
    /* (first use of the library in this backend) the library this extension was linked against may have been replaced */
    if (freddy_gpu_abi_version() != FREDDY_GPU_ABI_VERSION)
        elog(ERROR, "freddy_gpu: libfreddy_gpu.so has ABI version %d, this extension was built against %d", freddy_gpu_abi_version(), FREDDY_GPU_ABI_VERSION);
    if (!exit_hook_set) {
        on_proc_exit(on_exit_unpin, (Datum) 0);
        RegisterXactCallback(freddy_xact_cb, NULL);
        RegisterSubXactCallback(freddy_subxact_cb, NULL);
        exit_hook_set = true;
    }
}

/* arrays of a 3 M x 300 table exceed MaxAllocSize (1 GB): huge allocations in the caller's context */
static void *big_alloc(Size bytes)
{
    return MemoryContextAllocHuge(CurrentMemoryContext, bytes > 0 ? bytes : 1);
}

/* ---- argument codecs ---------------------------------------------------------------------------- */
float *freddy_glue_bytea_f32(bytea *b, int *n)
{
    Size   bytes = VARSIZE_ANY_EXHDR(b);
    float *out;
    *n = freddy_payload_count(bytes, sizeof(float4), -1);
    if (*n < 0) elog(ERROR, "freddy_gpu: a float4 vector of %zu bytes", (size_t) bytes);
    out = palloc(sizeof(float) * (*n > 0 ? *n : 1));
    (void) freddy_payload_f32(VARDATA_ANY(b), bytes, *n, out);
    return out;
}

int32 *freddy_glue_int_array(ArrayType *a, int *n)
{
    Datum *elems; bool *nulls; int16 typlen; bool typbyval; char typalign;
    int32 *out;
    if (ARR_ELEMTYPE(a) != INT4OID) elog(ERROR, "freddy_gpu: integer[] expected");
    get_typlenbyvalalign(ARR_ELEMTYPE(a), &typlen, &typbyval, &typalign);
    deconstruct_array(a, ARR_ELEMTYPE(a), typlen, typbyval, typalign, &elems, &nulls, n);
    out = palloc(sizeof(int32) * (*n > 0 ? *n : 1));
    for (int i = 0; i < *n; i++) {
        if (nulls[i]) elog(ERROR, "freddy_gpu: NULL element in an id array");
        out[i] = DatumGetInt32(elems[i]);
    }
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
        bytea *b;
        Size   bytes;
        if (nulls[i]) elog(ERROR, "freddy_gpu: NULL query vector");
        b = DatumGetByteaPP(elems[i]);
        bytes = VARSIZE_ANY_EXHDR(b);
        if (freddy_payload_count(bytes, sizeof(float4), -1) < 0) elog(ERROR, "freddy_gpu: a float4 vector of %zu bytes", (size_t) bytes);
        if (i == 0) {
            *dim = (int) (bytes / sizeof(float4));
            /* pinned host memory: the library reads the batch from here without a staging copy (not palloc'd: the
             * backend's reusable buffer, valid until the next call of this function) */
            out = query_buffer(sizeof(float) * (Size) (*rows) * (*dim > 0 ? *dim : 1));
        }
        if (freddy_payload_f32(VARDATA_ANY(b), bytes, *dim, out + (size_t) i * (*dim)) < 0) elog(ERROR, "query vectors of different dimensionality");
    }
    return out ? out : query_buffer(sizeof(float));
}

void freddy_glue_check_dim(int query_dim, int index_dim)
{
    if (query_dim != index_dim)
        elog(ERROR, "freddy_gpu: query vector has %d dimensions, the index %d", query_dim, index_dim);
}

/* ---- SPI helpers --------------------------------------------------------------------------------- */
/* up to max_rows rows of up to 3 int8 columns (NULL -> -1); returns the number of rows */
static int spi_int64_rows(const char *sql, int n_cols, int64 out[][3], int max_rows)
{
    int n = 0;
    if (SPI_connect() != SPI_OK_CONNECT) elog(ERROR, "freddy_gpu: SPI_connect failed");
    if (SPI_exec(sql, max_rows) > 0 && SPI_tuptable != NULL) {
        n = (int) SPI_processed;
        for (int i = 0; i < n; i++)
            for (int c = 0; c < n_cols; c++) {
                bool isnull;
                Datum dv = SPI_getbinval(SPI_tuptable->vals[i], SPI_tuptable->tupdesc, c + 1, &isnull);
                out[i][c] = isnull ? -1 : DatumGetInt64(dv);
            }
    }
    SPI_finish();
    return n;
}

static int64 spi_int64(const char *sql, int64 dflt)
{
    int64 v[1][3];
    return (spi_int64_rows(sql, 1, v, 1) == 1 && v[0][0] != -1) ? v[0][0] : dflt;
}

static Oid table_oid(const char *name)
{
    /* the getters print a regclass: possibly schema-qualified, possibly quoted */
    Oid oid = DatumGetObjectId(DirectFunctionCall1(regclassin, CStringGetDatum(name)));
    if (!OidIsValid(oid)) elog(ERROR, "freddy_gpu: table %s does not exist", name);
    return oid;
}

/* what the tables look like NOW (max_id / d / m are filled by the loaders): ONE catalog query per search for the
 * relfilenodes and generation counters of all tables of the handle (two more without the watch script) */
static void take_stamp(FreddyStamp *st, int n_tabs, char names[][100])
{
    char  sql[900], *cur = sql;
    int64 rows[MAX_TABS][3];
    bool  watched = spi_int64("SELECT (to_regclass('freddy_gpu_generation') IS NOT NULL)::int::bigint", 0) == 1;
    memset(st, 0, sizeof *st);
    st->n_tabs = n_tabs;
    for (int i = 0; i < n_tabs; i++) st->rel[i] = table_oid(names[i]);
    /* relfilenode changes on TRUNCATE, VACUUM FULL, CLUSTER and other rewrites */
    cur += sprintf(cur, "SELECT pg_relation_filenode(t.o)::bigint, %s FROM unnest(ARRAY[",
                   watched ? "g.appends, g.rewrites" : "pg_relation_size(t.o)::bigint, NULL::bigint");
    for (int i = 0; i < n_tabs; i++) cur += sprintf(cur, i ? ",%u" : "%u", st->rel[i]);
    cur += sprintf(cur, "]::oid[]) WITH ORDINALITY AS t(o, n)%s ORDER BY t.n", watched ? " LEFT JOIN freddy_gpu_generation g ON g.tab = t.o" : "");
    if (spi_int64_rows(sql, 3, rows, n_tabs) != n_tabs) elog(ERROR, "freddy_gpu: cannot read the catalog state of the index tables");
    for (int i = 0; i < n_tabs; i++) {
        st->filenode[i] = (Oid) rows[i][0];
        if (watched && rows[i][1] >= 0) { st->appends[i] = rows[i][1]; st->rewrites[i] = rows[i][2]; st->weak[i] = 0; }
        else {   /* this table carries no watch trigger: the weaker stamp */
            st->appends[i] = st->rewrites[i] = -1;
            st->weak[i] = watched ? spi_int64(psprintf("SELECT pg_relation_size(%u)::bigint", st->rel[i]), 0) : rows[i][1];
            if (i == 1) st->weak[i] = spi_int64(psprintf("SELECT sum(count)::bigint FROM %s", names[i]), 0);   /* insert_batch bumps counts (index_utils.c:949) */
        }
    }
}

typedef enum { PIN_CURRENT, PIN_APPENDED, PIN_CODEBOOK, PIN_STALE } PinState;

/* compare a pinned handle's stamp with the tables now (the decision itself: freddy_compare_stamp, freddy_pure.h).
 * appended / codebook may both be set (insert_batch does both) */
static PinState compare_stamp(const FreddyStamp *old, const FreddyStamp *now, const char *row_table, unsigned ignore_inserts_mask,
                              bool *appended, bool *codebook)
{
    int app = 0, cbk = 0;
    int64 row_max = -1;
    FreddyPinState ps;
    if (now->appends[0] < 0)   /* no trigger on the row table: an append shows as a larger max(id) */
        row_max = spi_int64(psprintf("SELECT max(id)::bigint FROM %s", row_table), -1);
    ps = freddy_compare_stamp(old, now, row_max, ignore_inserts_mask, &app, &cbk);
    *appended = app != 0; *codebook = cbk != 0;
    return ps == FREDDY_PIN_STALE ? PIN_STALE : ps == FREDDY_PIN_CATCH_UP ? PIN_APPENDED : PIN_CURRENT;
}

/* ---- table flattening ---------------------------------------------------------------------------- */
/* CodebookCompound (entries carrying their own pos / code, index_utils.h:52-56) -> dense [m][K][s]; every (pos, code)
 * slot must be present exactly once */
static float *dense_codebook(CodebookCompound cb, int s)
{
    int    n = cb.positions * cb.codeSize;
    float *out = palloc0(sizeof(float) * (size_t) n * s);
    bool  *seen = palloc0(sizeof(bool) * (n > 0 ? n : 1));
    if (n <= 0 || s <= 0) elog(ERROR, "freddy_gpu: empty codebook");
    for (int i = 0; i < n; i++) {
        int pos = cb.codebook[i].pos, code = cb.codebook[i].code;
        if (pos < 0 || pos >= cb.positions || code < 0 || code >= cb.codeSize || seen[pos * cb.codeSize + code])
            elog(ERROR, "freddy_gpu: codebook entry (pos %d, code %d) out of range or duplicated", pos, code);
        seen[pos * cb.codeSize + code] = true;
        memcpy(out + ((size_t) pos * cb.codeSize + code) * s, cb.codebook[i].vector, sizeof(float) * s);
    }
    return out;
}

/* sub-vector length of a codebook table (its vectors are float4 bytea) */
static int codebook_subdim(const char *cbName)
{
    int64 s = spi_int64(psprintf("SELECT (octet_length(vector) / 4)::bigint FROM %s LIMIT 1", cbName), 0);
    if (s <= 0) elog(ERROR, "freddy_gpu: codebook table %s is empty or unreadable", cbName);
    return (int) s;
}

/* "SELECT id, [coarse_id,] vector FROM <table> [WHERE id > x] ORDER BY ..." -> ids, [cells,] int16 codes, allocated
 * (huge-capable) in the caller's context, streamed through a cursor so that SPI never materialises 3 M tuples at once.
 * Returns the number of rows; *m = codes per row (must equal m_expected unless that is 0). */
static int64 fetch_code_rows(const char *sql, bool with_cell, int m_expected, int32 **ids, int32 **cells, int16 **codes, int *m)
{
    int64  n = 0, cap = 0;
    MemoryContext caller = CurrentMemoryContext, old;
    Portal portal;
    SPIPlanPtr plan;
    *ids = NULL; *codes = NULL; *m = m_expected;
    if (with_cell) *cells = NULL;
    if (SPI_connect() != SPI_OK_CONNECT) elog(ERROR, "freddy_gpu: SPI_connect failed");
    plan = SPI_prepare(sql, 0, NULL);
    if (plan == NULL) elog(ERROR, "freddy_gpu: cannot read the quantization table (%s)", sql);
    portal = SPI_cursor_open(NULL, plan, NULL, NULL, true);
    for (;;) {
        SPI_cursor_fetch(portal, true, 65536);
        if (SPI_processed == 0 || SPI_tuptable == NULL) break;
        old = MemoryContextSwitchTo(caller);
        if (n + (int64) SPI_processed > cap) {
            int64 ncap = cap ? cap * 2 : 1 << 20;
            while (ncap < n + (int64) SPI_processed) ncap *= 2;
            if (*m == 0) {   /* codes per row from the first row */
                bool isnull;
                bytea *b = DatumGetByteaPP(SPI_getbinval(SPI_tuptable->vals[0], SPI_tuptable->tupdesc, with_cell ? 3 : 2, &isnull));
                if (isnull) elog(ERROR, "freddy_gpu: NULL code vector");
                *m = (int) (VARSIZE_ANY_EXHDR(b) / sizeof(int16));
                if (*m <= 0) elog(ERROR, "freddy_gpu: empty code vector");
            }
            *ids = *ids ? repalloc_huge(*ids, sizeof(int32) * ncap) : big_alloc(sizeof(int32) * ncap);
            if (with_cell) *cells = *cells ? repalloc_huge(*cells, sizeof(int32) * ncap) : big_alloc(sizeof(int32) * ncap);
            *codes = *codes ? repalloc_huge(*codes, sizeof(int16) * (Size) ncap * *m) : big_alloc(sizeof(int16) * (Size) ncap * *m);
            cap = ncap;
        }
        MemoryContextSwitchTo(old);
        for (uint64 i = 0; i < SPI_processed; i++, n++) {
            bool      isnull, null2 = false, null3;
            HeapTuple t = SPI_tuptable->vals[i];
            bytea    *b;
            (*ids)[n] = DatumGetInt32(SPI_getbinval(t, SPI_tuptable->tupdesc, 1, &isnull));
            if (with_cell) (*cells)[n] = DatumGetInt32(SPI_getbinval(t, SPI_tuptable->tupdesc, 2, &null2));
            b = DatumGetByteaPP(SPI_getbinval(t, SPI_tuptable->tupdesc, with_cell ? 3 : 2, &null3));
            if (isnull || null2 || null3) elog(ERROR, "freddy_gpu: NULL in a quantization row");
            if (freddy_payload_i16(VARDATA_ANY(b), VARSIZE_ANY_EXHDR(b), *m, *codes + (Size) n * *m) < 0)
                elog(ERROR, "freddy_gpu: code rows of different lengths");
        }
        SPI_freetuptable(SPI_tuptable);
    }
    SPI_cursor_close(portal);
    SPI_finish();
    old = MemoryContextSwitchTo(caller);
    if (*ids == NULL) { *ids = palloc(sizeof(int32)); *codes = palloc(sizeof(int16)); if (with_cell) *cells = palloc(sizeof(int32)); }
    MemoryContextSwitchTo(old);
    return n;
}

/* rows appended since the handle was pinned: "id > max pinned id", in id order (freddy_gpu_append_rows wants that) */
static void append_new_rows(FreddyPin *p, const char *table, bool with_cell, int n_cells, bool with_vectors, const char *vecTable)
{
    char   sql[400];
    int32 *ids, *cells = NULL; int16 *codes; int m; int64 n;
    float *vectors = NULL;
    snprintf(sql, sizeof sql, with_cell ? "SELECT id, coarse_id, vector FROM %s WHERE id > %d ORDER BY id"
                                        : "SELECT id, vector FROM %s WHERE id > %d ORDER BY id", table, p->st.max_id);
    n = fetch_code_rows(sql, with_cell, p->st.m, &ids, &cells, &codes, &m);
    /* the rows above max_id must be the continuation of what is pinned (max_id + 1, + 2, ...): an append that was flagged
     * but shows no row, a gap, or ids this handle already holds from a rolled-back transaction cannot be repaired by
     * appending -- pin again (freddy_pure.h freddy_catch_up_rows_ok) */
    if (!freddy_catch_up_rows_ok(p->st.max_id, n, ids)) { drop_pin(p); return; }
    if (with_cell)
        for (int64 i = 0; i < n; i++)
            if (cells[i] < 0 || cells[i] >= n_cells) elog(ERROR, "freddy_gpu: coarse_id %d outside [0, %d)", cells[i], n_cells);
    if (with_vectors) {   /* ivpq: the vectors the reference JOINs in for methods 1 and 2 (ivpq_search_in.c:361-371) */
        vectors = big_alloc(sizeof(float) * (Size) n * p->st.d);
        if (SPI_connect() != SPI_OK_CONNECT) elog(ERROR, "freddy_gpu: SPI_connect failed");
        snprintf(sql, sizeof sql, "SELECT v.vector FROM %s AS fq INNER JOIN %s AS v ON fq.id = v.id WHERE fq.id > %d ORDER BY fq.id",
                 table, vecTable, p->st.max_id);
        if (SPI_exec(sql, 0) <= 0 || (int64) SPI_processed != n) elog(ERROR, "freddy_gpu: every ivpq row needs its vector");
        for (int64 i = 0; i < n; i++) {
            bool isnull;
            bytea *b = DatumGetByteaPP(SPI_getbinval(SPI_tuptable->vals[i], SPI_tuptable->tupdesc, 1, &isnull));
            if (isnull || VARSIZE_ANY_EXHDR(b) != sizeof(float) * (Size) p->st.d) elog(ERROR, "freddy_gpu: vector of the wrong length");
            memcpy(vectors + (Size) i * p->st.d, VARDATA_ANY(b), sizeof(float) * p->st.d);
        }
        SPI_finish();
    }
    freddy_glue_check(freddy_gpu_append_rows(p->h, n, ids, cells, codes, vectors));
    p->st.max_id = ids[n - 1];
    mark_mutated(p);
}

static void reload_codebook(FreddyPin *p, char *cbName)
{
    CodebookCompound cb = getCodebook(cbName);
    int s = codebook_subdim(cbName);
    if (cb.positions != p->st.m || cb.positions * s != p->st.d) { drop_pin(p); return; }   /* another shape: pin again */
    freddy_glue_check(freddy_gpu_update_codebook(p->h, dense_codebook(cb, s)));
    mark_mutated(p);
}

static int32 last_id(const int32 *ids, int64 n, bool ascending)
{
    int32 mx = -1;
    if (ascending) return n > 0 ? ids[n - 1] : -1;
    for (int64 i = 0; i < n; i++) if (ids[i] > mx) mx = ids[i];
    return mx;
}

/* ---- the three handles ----------------------------------------------------------------------------- */
freddy_gpu_index_t *freddy_glue_pq(void)
{
    char names[MAX_TABS][100];
    FreddyStamp now;
    bool appended, codebook;
    ensure_exit_hook();   /* the ABI check and the transaction callbacks BEFORE the first library call of this backend */
    getTableName(PQ_QUANTIZATION, names[0], 100);
    getTableName(CODEBOOK, names[1], 100);
    take_stamp(&now, 2, names);
    if (pin_pq.h) {
        PinState ps = compare_stamp(&pin_pq.st, &now, names[0], 0u, &appended, &codebook);
        if (ps == PIN_STALE) drop_pin(&pin_pq);
        else if (ps != PIN_CURRENT) {
            now.max_id = pin_pq.st.max_id; now.d = pin_pq.st.d; now.m = pin_pq.st.m;
            if (codebook) reload_codebook(&pin_pq, names[1]);
            if (pin_pq.h && appended) append_new_rows(&pin_pq, names[0], false, 0, false, NULL);
            if (pin_pq.h) { now.max_id = pin_pq.st.max_id; pin_pq.st = now; }
        }
    }
    if (pin_pq.h == NULL) {
        char  sql[256];
        CodebookCompound cb;
        int32 *ids; int16 *codes; int m, s; int64 n;
        freddy_pq_desc desc;
        cb = getCodebook(names[1]);
        s = codebook_subdim(names[1]);
        snprintf(sql, sizeof sql, "SELECT id, vector FROM %s ORDER BY id", names[0]);   /* canonical scan order */
        n = fetch_code_rows(sql, false, cb.positions, &ids, NULL, &codes, &m);
        desc.d = cb.positions * s; desc.m = cb.positions; desc.K = cb.codeSize; desc.N = n;
        desc.codebook = dense_codebook(cb, s); desc.ids = ids; desc.codes = codes;
        freddy_glue_check(freddy_gpu_pin_pq(&desc, 0, &pin_pq.h));
        pin_pq.st = now; pin_pq.st.max_id = last_id(ids, n, true); pin_pq.st.d = desc.d; pin_pq.st.m = desc.m;
        mark_mutated(&pin_pq);   /* (pinned from this transaction's snapshot, which may include its own uncommitted rows) */
        ensure_exit_hook();
    }
    return pin_pq.h;
}

freddy_gpu_index_t *freddy_glue_ivf(void)
{
    char names[MAX_TABS][100];
    FreddyStamp now;
    bool appended, codebook;
    static int pinned_C = 0;
    ensure_exit_hook();   /* the ABI check and the transaction callbacks BEFORE the first library call of this backend */
    getTableName(RESIDUAL_QUANTIZATION, names[0], 100);
    getTableName(RESIDUAL_CODEBOOK, names[1], 100);
    getTableName(COARSE_QUANTIZATION, names[2], 100);
    take_stamp(&now, 3, names);
    if (pin_ivf.h) {
        PinState ps = compare_stamp(&pin_ivf.st, &now, names[0], 0u, &appended, &codebook);
        if (ps == PIN_STALE) drop_pin(&pin_ivf);
        else if (ps != PIN_CURRENT) {
            now.d = pin_ivf.st.d; now.m = pin_ivf.st.m;
            if (codebook) reload_codebook(&pin_ivf, names[1]);
            if (pin_ivf.h && appended) append_new_rows(&pin_ivf, names[0], true, pinned_C, false, NULL);
            if (pin_ivf.h) { now.max_id = pin_ivf.st.max_id; pin_ivf.st = now; }
        }
    }
    if (pin_ivf.h == NULL) {
        char  sql[300];
        CodebookCompound cb; CoarseQuantizer cq; int C = 0;
        int32 *ids, *cells, *list_off; int16 *codes; int m, s, d; int64 n;
        float *coarse; bool *seen;
        freddy_ivf_desc desc;
        cb = getCodebook(names[1]);
        s = codebook_subdim(names[1]);
        cq = getCoarseQuantizer(&C);
        if (C <= 0) elog(ERROR, "freddy_gpu: empty coarse quantizer");
        d = cb.positions * s;
        /* inverted lists: rows grouped by coarse id, ascending id inside (the canonical order of freddy.c:324-342) */
        snprintf(sql, sizeof sql, "SELECT id, coarse_id, vector FROM %s ORDER BY coarse_id, id", names[0]);
        n = fetch_code_rows(sql, true, cb.positions, &ids, &cells, &codes, &m);
        coarse = palloc0(sizeof(float) * (size_t) C * d);      /* array index == coarse id (freddy.c:309,873) */
        seen = palloc0(sizeof(bool) * C);
        for (int i = 0; i < C; i++) {
            if (cq[i].id < 0 || cq[i].id >= C || seen[cq[i].id]) elog(ERROR, "freddy_gpu: coarse ids must be 0..%d, each once (found %d)", C - 1, cq[i].id);
            seen[cq[i].id] = true;
            memcpy(coarse + (size_t) cq[i].id * d, cq[i].vector, sizeof(float) * d);
        }
        list_off = palloc0(sizeof(int32) * (C + 1));
        for (int64 i = 0; i < n; i++) {
            if (cells[i] < 0 || cells[i] >= C) elog(ERROR, "freddy_gpu: coarse_id %d of row %d outside [0, %d)", cells[i], ids[i], C);
            list_off[cells[i] + 1]++;
        }
        for (int c = 0; c < C; c++) list_off[c + 1] += list_off[c];
        desc.d = d; desc.m = cb.positions; desc.K = cb.codeSize; desc.C = C; desc.N = n;
        desc.coarse = coarse; desc.codebook = dense_codebook(cb, s); desc.list_off = list_off; desc.ids = ids; desc.codes = codes;
        freddy_glue_check(freddy_gpu_pin_ivf(&desc, 0, &pin_ivf.h));
        pin_ivf.st = now; pin_ivf.st.max_id = last_id(ids, n, false); pin_ivf.st.d = d; pin_ivf.st.m = desc.m;
        pinned_C = C;
        mark_mutated(&pin_ivf);
        ensure_exit_hook();
    }
    return pin_ivf.h;
}

freddy_gpu_index_t *freddy_glue_ivpq(void)
{
    char names[MAX_TABS][100];
    FreddyStamp now;
    bool appended, codebook;
    static int pinned_cells = 0;
    ensure_exit_hook();   /* the ABI check and the transaction callbacks BEFORE the first library call of this backend */
    getTableName(IVPQ_QUANTIZATION, names[0], 100);
    getTableName(IVPQ_CODEBOOK, names[1], 100);
    getTableName(COARSE_QUANTIZATION_MULTI, names[2], 100);
    getTableName(NORMALIZED, names[3], 100);
    getTableName(STATISTICS, names[4], 100);
    take_stamp(&now, 5, names);
    if (pin_ivpq.h) {
        /* an INSERT into the vector table [3] accompanies every appended ivpq row (updateWordVectorsRelation): expected */
        PinState ps = compare_stamp(&pin_ivpq.st, &now, names[0], 1u << 3, &appended, &codebook);
        if (ps == PIN_STALE) drop_pin(&pin_ivpq);
        else if (ps != PIN_CURRENT) {
            now.d = pin_ivpq.st.d; now.m = pin_ivpq.st.m;
            if (codebook) reload_codebook(&pin_ivpq, names[1]);
            if (pin_ivpq.h && appended) append_new_rows(&pin_ivpq, names[0], true, pinned_cells, true, names[3]);
            if (pin_ivpq.h) { now.max_id = pin_ivpq.st.max_id; pin_ivpq.st = now; }
        }
    }
    if (pin_ivpq.h == NULL) {
        char  sql[400];
        CodebookCompound cb, cq;
        int32 *ids, *cells; int16 *codes; int m, s, d, n_cells; int64 n;
        float *vectors, *stats;
        freddy_ivpq_desc desc;
        cb = getCodebook(names[1]);
        cq = getCodebook(names[2]);
        s = codebook_subdim(names[1]);
        d = cb.positions * s;
        if (cq.positions != 2 || cq.codeSize <= 0) elog(ERROR, "freddy_gpu: the multi index needs 2 positions (index_utils.c:322)");
        n_cells = cq.codeSize * cq.codeSize;
        stats = getStatistics();                               /* [cells + 1], last = total count (index_utils.c:632-665) */
        snprintf(sql, sizeof sql, "SELECT id, coarse_id, vector FROM %s ORDER BY id", names[0]);
        n = fetch_code_rows(sql, true, cb.positions, &ids, &cells, &codes, &m);
        for (int64 i = 0; i < n; i++)
            if (cells[i] < 0 || cells[i] >= n_cells) elog(ERROR, "freddy_gpu: coarse_id %d of row %d outside [0, %d)", cells[i], ids[i], n_cells);
        {   /* the vectors the reference JOINs in for methods 1 and 2 (ivpq_search_in.c:361-371), row-aligned with ids:
             * 3 M x 300 floats = 3.6 GB, beyond MaxAllocSize -- a huge allocation, filled through a cursor */
            MemoryContext caller = CurrentMemoryContext, old;
            Portal portal; SPIPlanPtr plan; int64 got = 0;
            vectors = big_alloc(sizeof(float) * (Size) (n > 0 ? n : 1) * d);
            if (SPI_connect() != SPI_OK_CONNECT) elog(ERROR, "freddy_gpu: SPI_connect failed");
            snprintf(sql, sizeof sql, "SELECT v.vector FROM %s AS fq INNER JOIN %s AS v ON fq.id = v.id ORDER BY fq.id", names[0], names[3]);
            plan = SPI_prepare(sql, 0, NULL);
            if (plan == NULL) elog(ERROR, "freddy_gpu: cannot read the vectors of the ivpq rows");
            portal = SPI_cursor_open(NULL, plan, NULL, NULL, true);
            for (;;) {
                SPI_cursor_fetch(portal, true, 16384);
                if (SPI_processed == 0 || SPI_tuptable == NULL) break;
                if (got + (int64) SPI_processed > n) elog(ERROR, "freddy_gpu: more vectors than ivpq rows");
                for (uint64 i = 0; i < SPI_processed; i++, got++) {
                    bool isnull;
                    bytea *b = DatumGetByteaPP(SPI_getbinval(SPI_tuptable->vals[i], SPI_tuptable->tupdesc, 1, &isnull));
                    if (isnull || VARSIZE_ANY_EXHDR(b) != sizeof(float) * (Size) d) elog(ERROR, "freddy_gpu: vector of the wrong length");
                    memcpy(vectors + (Size) got * d, VARDATA_ANY(b), sizeof(float) * d);
                }
                SPI_freetuptable(SPI_tuptable);
            }
            SPI_cursor_close(portal);
            SPI_finish();
            old = MemoryContextSwitchTo(caller); MemoryContextSwitchTo(old);
            if (got != n) elog(ERROR, "freddy_gpu: every ivpq row needs its vector (%lld of %lld found)", (long long) got, (long long) n);
        }
        desc.d = d; desc.m = cb.positions; desc.K = cb.codeSize;
        desc.coarse_positions = cq.positions; desc.coarse_codes = cq.codeSize; desc.N = n;
        desc.codebook = dense_codebook(cb, s); desc.coarse = dense_codebook(cq, d / cq.positions);
        desc.ids = ids; desc.coarse_id = cells; desc.codes = codes; desc.vectors = vectors; desc.stats = stats;
        freddy_glue_check(freddy_gpu_pin_ivpq(&desc, 0, &pin_ivpq.h));
        pin_ivpq.st = now; pin_ivpq.st.max_id = last_id(ids, n, true); pin_ivpq.st.d = d; pin_ivpq.st.m = desc.m;
        pinned_cells = n_cells;
        pfree(vectors);
        mark_mutated(&pin_ivpq);
        ensure_exit_hook();
    }
    return pin_ivpq.h;
}

/* after a write in THIS backend (insert_batch): bring every handle that is already pinned up to date now */
void freddy_glue_refresh_pinned(void)
{
    if (pin_pq.h) (void) freddy_glue_pq();
    if (pin_ivf.h) (void) freddy_glue_ivf();
    if (pin_ivpq.h) (void) freddy_glue_ivpq();
}

int freddy_glue_dim(freddy_gpu_index_t *h)
{
    if (h == pin_pq.h) return pin_pq.st.d;
    if (h == pin_ivf.h) return pin_ivf.st.d;
    if (h == pin_ivpq.h) return pin_ivpq.st.d;
    return 0;
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
