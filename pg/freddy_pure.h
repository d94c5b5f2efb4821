/*
 * freddy_pure.h -- the parts of the PostgreSQL hosts that do not need PostgreSQL: decisions and arithmetic on plain
 * memory.  pg/freddy_gpu_glue.c and pg/freddy_insert.c include it (with palloc as the allocator); so does the test
 * driver that tests/test_pg_pure.py compiles with gcc in this repository's image (with malloc) -- the PostgreSQL side
 * itself (SPI, varlena / array macros, fmgr) cannot be compiled here, this can.
 *
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

/* ---- insert_batch: updateCodebook after the 1-NN search ---------------------------------------------------- */
/* layout of the reference's CodebookEntryComplete (index_utils.h:58-63) */
typedef struct FreddyCbEntry { int pos; int code; float *vector; int count; } FreddyCbEntry;

/* updateCodebook (index_utils.c:908-957) for codes found on the device: nearestCentroids[i][pos] = codes[i*m + pos];
 * everything after the 1-NN search is the reference's statement sequence, slips included:
 *   - ONE nearestCentroidRaw across positions: after the reference's scan over the table (position-major in every table
 *     the index scripts write) it is the nearest entry of the LAST position (:931-938) -- that vector is what every
 *     position of the row adds to its bucket (:944-946);
 *   - the recalculation reads bucket [pos + code] (:954), adds (1.0 / count) * bucket in double (:953-955).
 * cb: cbPositions * cbCodes entries in table order, updated in place (vector, count); countIncs [cbPositions*cbCodes]. */
static inline void freddy_update_codebook_known_codes(int rawVectorsSize, int subvectorSize, FreddyCbEntry *cb, int cbPositions, int cbCodes,
                                                      const int16_t *codes, int **nearestCentroids, int *countIncs)
{
    const int E = cbPositions * cbCodes;
    float **differences = (float **) FREDDY_ALLOC(sizeof(float *) * (size_t) E);
    float **entry_of = (float **) FREDDY_ALLOC(sizeof(float *) * (size_t) E);   /* (pos, code) -> the entry's vector */
    for (int i = 0; i < E; i++) {
        differences[i] = (float *) FREDDY_ALLOC(sizeof(float) * (size_t) subvectorSize);
        for (int j = 0; j < subvectorSize; j++) differences[i][j] = 0;
        countIncs[i] = 0;
        entry_of[cb[i].pos * cbCodes + cb[i].code] = cb[i].vector;
    }
    for (int i = 0; i < rawVectorsSize; i++) {
        float *nearestCentroidRaw = entry_of[(cbPositions - 1) * cbCodes + codes[(size_t) i * cbPositions + cbPositions - 1]];
        nearestCentroids[i] = (int *) FREDDY_ALLOC(sizeof(int) * (size_t) cbPositions);
        for (int j = 0; j < cbPositions; j++) nearestCentroids[i][j] = codes[(size_t) i * cbPositions + j];
        for (int j = 0; j < cbPositions; j++) {
            const int code = nearestCentroids[i][j];
            countIncs[j * cbCodes + code] += 1;
            for (int k = 0; k < subvectorSize; k++) differences[j * cbCodes + code][k] += nearestCentroidRaw[k];
        }
    }
    for (int i = 0; i < E; i++) {   /* recalculate codebook (index_utils.c:949-956) */
        cb[i].count += countIncs[cb[i].pos * cbCodes + cb[i].code];
        for (int j = 0; j < subvectorSize; j++)
            cb[i].vector[j] += (1.0 / cb[i].count) * differences[cb[i].pos + cb[i].code][j];
    }
}

#endif /* FREDDY_PURE_H */
