/* Test driver for pg/freddy_pure.h: the PostgreSQL-free parts of the hosts, compiled with gcc (allocator = malloc) into a
 * small shared object that tests/test_pg_pure.py loads with ctypes. */
#include <stdlib.h>
#define FREDDY_ALLOC malloc
#define FREDDY_FREE free
#include "../../pg/freddy_pure.h"

int drv_compare(const FreddyStamp *old, const FreddyStamp *now, long long row_max, unsigned mask, int *appended, int *codebook)
{
    return (int) freddy_compare_stamp(old, now, row_max, mask, appended, codebook);
}
int drv_catch_up_ok(int pinned_max, long long n, const int32_t *ids) { return freddy_catch_up_rows_ok(pinned_max, n, ids); }
int drv_survives_abort(int mutated, int aborted) { return freddy_pin_survives_abort(mutated, aborted); }
int drv_level_after_commit(int mutated, int committed) { return freddy_pin_level_after_commit(mutated, committed); }
size_t drv_stamp_size(void) { return sizeof(FreddyStamp); }
int drv_payload_f32(const void *data, size_t bytes, int expect, float *out) { return freddy_payload_f32(data, bytes, expect, out); }
int drv_payload_i16(const void *data, size_t bytes, int expect, int16_t *out) { return freddy_payload_i16(data, bytes, expect, out); }

/* codebook [m][K][s] + counts [m*K] as flat arrays, entries handed over in the permuted table order `order` */
void drv_update_codebook(float *codebook, int *counts, int m, int K, int s, const int16_t *codes, int n, const int *order, int *count_incs)
{
    const int E = m * K;
    FreddyCbEntry *cb = (FreddyCbEntry *) malloc(sizeof(FreddyCbEntry) * (size_t) E);
    int *row_codes = (int *) malloc(sizeof(int) * (size_t) (n > 0 ? n : 1) * (size_t) m);
    for (int i = 0; i < E; i++) {
        const int e = order[i];
        cb[i].pos = e / K; cb[i].code = e % K; cb[i].vector = codebook + (size_t) e * s; cb[i].count = counts[e];
    }
    freddy_update_codebook_known_codes(n, s, cb, m, K, codes, row_codes, count_incs);
    for (int i = 0; i < n * m; i++) if (row_codes[i] != codes[i]) abort();
    for (int i = 0; i < E; i++) counts[order[i]] = cb[i].count;
    free(cb);
    free(row_codes);
}
