/*
 * freddy_pure.h -- the parts of the PostgreSQL hosts that do not need PostgreSQL: decisions and arithmetic on plain
 * memory.  pg/freddy_gpu_glue.c and pg/freddy_insert.c include it (with palloc as the allocator); so does the test
 * driver that tests/test_pg_pure.py compiles with gcc in this repository's image (with malloc) -- the PostgreSQL side
 * itself (SPI, varlena / array macros, fmgr) cannot be compiled here, this can.
 *
 *   freddy_catch_up_rows_ok, freddy_pin_survives_abort   transaction safety of a handle that was brought up to date
 *                            inside a transaction that may still roll back            
 *   freddy_compare_stamp     what a changed table means for a pinned handle: nothing / append rows / reload the codebook /
 *                            pin again                                                     
 *   freddy_payload_f32/_i16  a bytea's payload (what VARDATA_ANY / VARSIZE_ANY_EXHDR give) -> typed array, with the
 *                            length checks                         (index_utils.c:1078-1106 convert_bytea_*)
 *   freddy_update_codebook_known_codes   updateCodebook's bookkeeping after the 1-NN search   (index_utils.c:940-956)
 */
#ifndef FREDDY_PURE_H
#define FREDDY_PURE_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifndef FREDDY_ALLOC
#define FREDDY_ALLOC palloc
#endif
#ifndef FREDDY_FREE
#define FREDDY_FREE pfree
#endif

/* ---- staleness ----------------------------------------------------------------------------------------- */
#define FREDDY_MAX_TABS 5

typedef struct FreddyStamp {
    int      n_tabs;
    uint32_t rel[FREDDY_MAX_TABS];       /* [0] = the row table (ids + codes), [1] = its codebook, others: coarse quantizer, vectors, statistics */
    uint32_t filenode[FREDDY_MAX_TABS];
    int64_t  appends[FREDDY_MAX_TABS];   /* freddy_gpu_generation.appends, or -1: the table carries no watch trigger */
    int64_t  rewrites[FREDDY_MAX_TABS];
    int64_t  weak[FREDDY_MAX_TABS];      /* fallback stamp: pg_relation_size; for [1] sum(count) of the codebook */
    int32_t  max_id;                     /* largest id of the row table that is pinned */
    int      d, m;
} FreddyStamp;

typedef enum { FREDDY_PIN_CURRENT = 0, FREDDY_PIN_CATCH_UP = 1, FREDDY_PIN_STALE = 2 } FreddyPinState;

/* Compare what a handle was built from with the tables now.  row_max_id_now: max(id) of the row table as seen now, only
 * consulted when the row table carries no trigger (-1 = unknown).  ignore_inserts_mask: bit i set = INSERTs into table i
 * are expected and harmless (the vector table of the ivpq handle gains a row with every appended ivpq row).
 * FREDDY_PIN_CATCH_UP: *appended (rows with id > max_id are to be fetched and appended) and / or *codebook (the codebook
 * table is to be re-read) say what to do. */
static inline FreddyPinState freddy_compare_stamp(const FreddyStamp *old, const FreddyStamp *now, int64_t row_max_id_now,
                                                  unsigned ignore_inserts_mask, int *appended, int *codebook)
{
    *appended = *codebook = 0;
    if (old->n_tabs != now->n_tabs) return FREDDY_PIN_STALE;
    for (int i = 0; i < now->n_tabs; i++) {
        if (old->rel[i] != now->rel[i] || old->filenode[i] != now->filenode[i]) return FREDDY_PIN_STALE;   /* set_*() / TRUNCATE / rewrite */
        if ((old->appends[i] < 0) != (now->appends[i] < 0)) return FREDDY_PIN_STALE;                       /* the watch script came or went */
        if (now->appends[i] >= 0) {
            const int ins = now->appends[i] != old->appends[i], rew = now->rewrites[i] != old->rewrites[i];
            /* a counter that went BACKWARDS: the statements the handle was brought up to date with were rolled back
             * (insert_batch refreshes the handles inside its own transaction) -- the pinned copy holds rows no snapshot
             * will ever see, and ids it has seen will be handed out again */
            if (now->appends[i] < old->appends[i] || now->rewrites[i] < old->rewrites[i]) return FREDDY_PIN_STALE;
            if (i == 0) { if (rew) return FREDDY_PIN_STALE; if (ins) *appended = 1; }
            else if (i == 1) { if (ins) return FREDDY_PIN_STALE; if (rew) *codebook = 1; }
            else if (rew || (ins && !((ignore_inserts_mask >> i) & 1u))) return FREDDY_PIN_STALE;           /* coarse quantizer, vectors, statistics */
        } else if (now->weak[i] != old->weak[i]) {
            if (i == 0) *appended = 1;               /* the file grew: rows with a larger id are looked for */
            else if (i == 1) *codebook = 1;
            else if (!((ignore_inserts_mask >> i) & 1u)) return FREDDY_PIN_STALE;
        }
    }
    if (now->appends[0] < 0 && row_max_id_now >= 0) {   /* fallback: an append shows as a larger max(id) even if the file did not grow */
        if (row_max_id_now > old->max_id) *appended = 1;
        if (row_max_id_now < old->max_id) return FREDDY_PIN_STALE;
    }
    return (*appended || *codebook) ? FREDDY_PIN_CATCH_UP : FREDDY_PIN_CURRENT;
}

/* The catch-up fetch "id > max_id ORDER BY id" after an append was flagged: what it returned must be the continuation of
 * what is pinned.  insert_batch numbers its rows max(id) + 1, + 2, ... (freddy.c:1521-1545), so the first fetched id is
 * max_id + 1 and the ids are consecutive; anything else -- no row at all although the append counter moved (the rows
 * were rolled back, or committed below ids this handle already holds from an aborted transaction), a gap, a row that
 * committed out of id order in another backend -- cannot be repaired by appending: pin again.  1 = append, 0 = stale. */
static inline int freddy_catch_up_rows_ok(int32_t pinned_max_id, int64_t n_fetched, const int32_t *fetched_ids)
{
    if (n_fetched <= 0) return 0;
    for (int64_t i = 0; i < n_fetched; i++)
        if ((int64_t) fetched_ids[i] != (int64_t) pinned_max_id + 1 + i) return 0;
    return 1;
}

/* Handles that were changed inside a transaction (rows appended, codebook reloaded, freshly pinned from uncommitted
 * tables) carry that transaction's nesting level; when a (sub)transaction of that level or deeper aborts, the handle
 * must go (PostgreSQL: RegisterXactCallback / RegisterSubXactCallback).  0 = never touched inside an open transaction.
 * freddy_pin_survives_abort: 1 if a handle last mutated at `mutated_level` survives the abort of a (sub)transaction at
 * `aborted_level` (1 = the top-level transaction). */
static inline int freddy_pin_survives_abort(int mutated_level, int aborted_level)
{
    return mutated_level == 0 || mutated_level < aborted_level;
}
/* after a commit of (sub)transaction `level` its changes belong to the parent: the new level of a mark */
static inline int freddy_pin_level_after_commit(int mutated_level, int committed_level)
{
    if (mutated_level < committed_level) return mutated_level;
    return committed_level <= 1 ? 0 : committed_level - 1;
}

/* ---- bytea payloads ------------------------------------------------------------------------------------- */
/* number of elements, or -1 if the payload is not a whole number of them / differs from `expect` (expect < 0: any) */
static inline int freddy_payload_count(size_t bytes, size_t elem, int expect)
{
    if (elem == 0 || bytes % elem != 0) return -1;
    if (bytes / elem > (size_t) INT32_MAX) return -1;
    if (expect >= 0 && (size_t) expect != bytes / elem) return -1;
    return (int) (bytes / elem);
}
static inline int freddy_payload_f32(const void *data, size_t bytes, int expect, float *out)
{
    const int n = freddy_payload_count(bytes, sizeof(float), expect);
    if (n > 0) memcpy(out, data, bytes);
    return n;
}
static inline int freddy_payload_i16(const void *data, size_t bytes, int expect, int16_t *out)
{
    const int n = freddy_payload_count(bytes, sizeof(int16_t), expect);
    if (n > 0) memcpy(out, data, bytes);
    return n;
}

/* ---- insert_batch: the codebook bookkeeping that follows the device's 1-NN search ------------------------------- */
/* One codebook entry as the hosts hold it (same layout as the reference's CodebookEntryComplete, index_utils.h:58-63,
 * so pg/freddy_insert.c can hand the reference's own array over). */
typedef struct FreddyCbEntry { int pos; int code; float *vector; int count; } FreddyCbEntry;

/* What updateCodebook (index_utils.c:940-956) leaves behind, given the codes the device found (codes[r*m + p] = code of
 * new row r at position p).  Flat working arrays, three passes:
 *   1. slot (p, c) = p*K + c.  Every new row adds ONE vector to the bucket of each of its m slots and bumps the slot's
 *      increment.  Documented behaviour of the reference kept on purpose (the rows written must equal its rows): the
 *      vector added is the same for all positions of a row -- its 1-NN loop keeps a single pointer `nearestCentroidRaw`
 *      across positions (index_utils.c:925-939), which ends up at the entry that improved some position's minimum LAST in
 *      the order the entries are GIVEN (SPI order).  A position's last improvement is at its final nearest entry, so that
 *      entry is the one with the largest array index among the row's m nearest entries.  (Position-major tables: the
 *      nearest entry of the last position -- but insert_batch's own UPDATEs move codebook tuples, so the heap order is the
 *      export's only until the first call.)  The device's 1-NN takes the lowest code where two entries of a position are
 *      EQUALLY near (bit-equal distances); the reference takes the first in SPI order: the two differ only for such ties
 *      in a table that is no longer position-major.
 *   2. an entry's count grows by its slot's increment.
 *   3. an entry's vector moves by bucket[pos + code] / count, the quotient formed as (1.0 / count) in double -- the
 *      bucket index is the SUM pos + code, not the slot (documented behaviour, kept).
 * entries: m*K of them in any table order, updated in place; incs[m*K] by slot (out); row_codes[n*m] (out): the codes
 * widened to int, row-major (what the reference's row writer takes, one pointer per row). */
static inline void freddy_update_codebook_known_codes(int n, int s, FreddyCbEntry *entries, int m, int K,
                                                      const int16_t *codes, int *row_codes, int *incs)
{
    const size_t slots = (size_t) m * (size_t) K;
    float *bucket = (float *) FREDDY_ALLOC(sizeof(float) * slots * (size_t) s);
    size_t *entry_of_slot = (size_t *) FREDDY_ALLOC(sizeof(size_t) * slots);
    memset(bucket, 0, sizeof(float) * slots * (size_t) s);
    memset(incs, 0, sizeof(int) * slots);
    for (size_t e = 0; e < slots; e++) entry_of_slot[(size_t) entries[e].pos * K + entries[e].code] = e;
    for (int r = 0; r < n; r++) {
        const int16_t *rc = codes + (size_t) r * m;
        size_t last = 0;
        const float *added;
        for (int p = 0; p < m; p++) {
            const size_t e = entry_of_slot[(size_t) p * K + rc[p]];
            if (e > last) last = e;
        }
        added = entries[last].vector;
        for (int p = 0; p < m; p++) {
            const size_t slot = (size_t) p * K + rc[p];
            float *b = bucket + slot * (size_t) s;
            row_codes[(size_t) r * m + p] = rc[p];
            incs[slot] += 1;
            for (int t = 0; t < s; t++) b[t] += added[t];
        }
    }
    for (size_t e = 0; e < slots; e++) {
        FreddyCbEntry *en = &entries[e];
        const float *b = bucket + (size_t) (en->pos + en->code) * (size_t) s;
        en->count += incs[(size_t) en->pos * K + en->code];
        for (int t = 0; t < s; t++) en->vector[t] += (1.0 / en->count) * b[t];
    }
    FREDDY_FREE(bucket);
    FREDDY_FREE((void *) entry_of_slot);
}

#endif /* FREDDY_PURE_H */
