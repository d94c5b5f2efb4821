/*
 * freddy_gpu.h -- C ABI of the MI355X (gfx950) search library behind the FREDDY UDFs.
 *
 * This is the drop-in boundary for the reference's hot path: the bodies of the
 * PostgreSQL SRFs in freddy_extension/freddy.c and freddy_extension/ivpq_search_in.c
 * stop scanning SPI tuples and call these entry points instead (INTEGRATION.md shows
 * the binding).  Plain pointers and sizes only; caller allocates every host buffer;
 * the library never retains a host pointer after a call returns, never throws and
 * never longjmps.  Every function returns 0 on success and a negative FREDDY_E_* code
 * on failure, with a message available from freddy_gpu_last_error().
 *
 * Threading: one caller thread per process (a PostgreSQL backend is single threaded,
 * SURVEY 8b).  HIP is initialised lazily on the first pin call -- never at library
 * load -- so a postmaster can dlopen() the extension and fork() safely.
 *
 * Numerics: all distances are IEEE binary32, computed with separately rounded
 * sub/mul/add in the reference's summation order (index_utils.c:500-508, :1126-1133);
 * result lists follow the reference's insertion rule including its tie behaviour
 * (index_utils.c:19-33), with the canonical scan order "ascending id".
 */
#ifndef FREDDY_GPU_H
#define FREDDY_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FREDDY_OK 0
#define FREDDY_E_ARG (-1)      /* bad argument (NULL, shape mismatch, unsupported size) */
#define FREDDY_E_HIP (-2)      /* HIP runtime / device failure */
#define FREDDY_E_NOMEM (-3)    /* host or device allocation failed */
#define FREDDY_E_KIND (-4)     /* index handle of the wrong kind for this call */
#define FREDDY_E_LIMIT (-5)    /* parameter beyond what this build supports (see message): k > 4096, W > 512 probes per round,
                                * k * pvf > 8192 (or k > 512) in the kNN-join, K > 32767, d > 1024 for training */

/* found_rule for freddy_gpu_ivfadc_search */
#define FREDDY_FOUND_ROWS 0      /* ivfadc_search:       found += rows retrieved  (freddy.c:377) */
#define FREDDY_FOUND_ACCEPTED 1  /* found += insertions (freddy.c:971) over ivfadc_search's cell choice */
#define FREDDY_FOUND_BATCH_UDF 2 /* ivfadc_batch_search itself: found += insertions (freddy.c:971) AND its cell choice --
                                  * argmin by strict "<" from minDist = 1000 (freddy.c:853-866), where ivfadc_search's cell
                                  * list never admits a cell at distance >= 100 (freddy.c:266-283).  W must be 1. */

/* calculationMethod, index_utils.h:111 */
#define FREDDY_METHOD_PQ 0
#define FREDDY_METHOD_EXACT 1
#define FREDDY_METHOD_PQ_PV 2

typedef struct freddy_gpu_index freddy_gpu_index_t; /* opaque, library-owned, HBM-resident */

/* ---- what gets pinned: the tables the reference re-reads through SPI on EVERY call ----- */

/* pq_codebook + pq_quantization.  Replaces getCodebook(CODEBOOK) (freddy.c:69,
 * index_utils.c:577-630) and "SELECT id, vector FROM pq_quantization" (freddy.c:96-100). */
typedef struct freddy_pq_desc {
  int32_t d;             /* vector dimensionality (300) */
  int32_t m;             /* sub-quantizers = codebook positions */
  int32_t K;             /* codes per sub-quantizer */
  int64_t N;             /* rows */
  const float* codebook; /* [m][K][d/m]  entry (pos, code) -> its centroid */
  const int32_t* ids;    /* [N] row ids, rows in canonical scan order (ascending id, unique) */
  const int16_t* codes;  /* [N][m] as stored in the bytea column (int16, index_utils.c:1088) */
} freddy_pq_desc;

/* coarse_quantization + residual_codebook + fine_quantization.  Replaces getCodebook +
 * getCoarseQuantizer (freddy.c:239-241,746-749; index_utils.c:531-575) and the per-round
 * "SELECT id, vector, coarse_id FROM fine_quantization WHERE coarse_id IN (...)"
 * (freddy.c:324-342, 915-935).  Rows are grouped into inverted lists by coarse id. */
typedef struct freddy_ivf_desc {
  int32_t d, m, K;
  int32_t C;               /* coarse cells; coarse id == array index (freddy.c:309,873) */
  int64_t N;
  const float* coarse;     /* [C][d] */
  const float* codebook;   /* [m][K][d/m] residual codebook */
  const int32_t* list_off; /* [C+1] CSR offsets into ids/codes */
  const int32_t* ids;      /* [N] ascending and unique inside each list, unique overall, >= 0 */
  const int16_t* codes;    /* [N][m] */
} freddy_ivf_desc;

/* codebook_ivpq + coarse_quantization_ivpq (2-position multi index) +
 * fine_quantization_ivpq (+ the normalised vectors the reference JOINs in for methods
 * 1 and 2) + the stat_* table.  Replaces getCodebook x2, getStatistics
 * (ivpq_search_in.c:218-232) and the per-iteration SELECT (ivpq_search_in.c:352-401). */
typedef struct freddy_ivpq_desc {
  int32_t d, m, K;
  int32_t coarse_positions; /* must be 2 (index_utils.c:322) */
  int32_t coarse_codes;     /* cells = coarse_codes^2, cell = code0 + coarse_codes*code1 */
  int64_t N;
  const float* codebook;    /* [m][K][d/m] */
  const float* coarse;      /* [2][coarse_codes][d/2] */
  const int32_t* ids;       /* [N] ascending, unique */
  const int32_t* coarse_id; /* [N] */
  const int16_t* codes;     /* [N][m] */
  const float* vectors;     /* [N][d] row-aligned with ids; NULL if methods 1/2 are never used */
  const float* stats;       /* [cells+1] coarse_freq; last = total count (freddy--0.0.1.sql:158-168) */
} freddy_ivpq_desc;

/* Copy the tables into HBM once (layouts re-organised for the kernels).  The host
 * arrays may be freed as soon as the call returns. */
int freddy_gpu_pin_pq(const freddy_pq_desc* desc, int device, freddy_gpu_index_t** out);
int freddy_gpu_pin_ivf(const freddy_ivf_desc* desc, int device, freddy_gpu_index_t** out);
int freddy_gpu_pin_ivpq(const freddy_ivpq_desc* desc, int device, freddy_gpu_index_t** out);
int freddy_gpu_unpin(freddy_gpu_index_t* index);

/* ---- searches (host buffers in, host buffers out, synchronous) --------------------------- */

/* Body of pq_search (freddy.c:28-152), pq_search_in (freddy.c:1028-1157) and
 * pq_search_in_batch (freddy.c:414-653): exhaustive ADC over all rows
 * (subset_ids == NULL) or over the rows whose id is in subset_ids ("WHERE id IN (...)":
 * duplicates and unknown ids are ignored), for Q query vectors at once.
 * sentinel: 100.0 for pq_search, 1000.0 for the _in variants.
 * out_ids/out_dist: [Q][k]; unfilled slots hold (-1, sentinel). */
int freddy_gpu_pq_search(freddy_gpu_index_t* pq, const float* queries, int32_t Q, int32_t k,
                         float sentinel, const int32_t* subset_ids, int64_t n_subset,
                         int32_t* out_ids, float* out_dist);

/* Body of ivfadc_search (freddy.c:174-393) for Q independent queries; with W == 1,
 * sentinel 100.0 and FREDDY_FOUND_BATCH_UDF it is the body of ivfadc_batch_search
 * (freddy.c:679-999).  Each round probes the W nearest not-yet-used cells; rounds repeat
 * while found < k (found_rule).  out_*: [Q][k]. */
int freddy_gpu_ivfadc_search(freddy_gpu_index_t* ivf, const float* queries, int32_t Q, int32_t k,
                             int32_t W, float sentinel, int32_t found_rule, int32_t* out_ids,
                             float* out_dist);

/* How the host-buffer IVFADC call runs (the reference makes one synchronous call per batch, freddy.c:679-999): a batch
 * larger than `pipeline_batch` (2048) queries is cut into equal sub-batches that go round-robin to up to four
 * library-owned streams ("lanes", two sub-batches queued on each), with pinned staging buffers: host copy of the
 * sub-batch's queries into pinned memory -> a copy kernel reads them over PCIe -> round one of the search, the lanes'
 * persistent scans sharing the CUs -> a copy kernel writes the lists into pinned memory.  The host waits only when it
 * needs a staging slot again and once per slot at the end.  Queries that round one leaves unfinished (their first W
 * cells hold fewer than k rows: rare) are searched again from the start with all their rounds where the host waits for
 * their sub-batch -- the search is deterministic, so that is the list the round-by-round continuation gives.  Results
 * do not depend on how a batch is cut.
 *
 * Query buffers obtained from freddy_gpu_host_alloc (pinned host memory) skip the staging copy: a host that decodes
 * its bytea / array arguments can write the floats straight into such a buffer.  freddy_gpu_host_free releases it. */
int freddy_gpu_host_alloc(void** out, size_t bytes);
int freddy_gpu_host_free(void* p);

/* Product-level multi-GPU (north_star: "partition queries across the 8 MI355X with a replicated index"): the ivf tables
 * pinned on EVERY device of devices[0 .. n_devices) behind one handle.  freddy_gpu_ivfadc_search on such a handle splits
 * the host batch contiguously over the devices (sizes differ by at most one; one host thread per device inside the
 * call) -- queries are independent (freddy.c:835-982 keeps no cross-query state) and the lists land in the caller's
 * host buffers, so no collective is needed.  The same device may be listed more than once (tests).  append_rows /
 * update_codebook / set_option / the self-check counters act on every replica; the *_dev entry point, the profile and
 * the freddy_gpu_last_* diagnostics act on devices[0] only.  freddy_gpu_replica_count: number of devices behind a handle. */
int freddy_gpu_pin_ivf_multi(const freddy_ivf_desc* desc, const int* devices, int n_devices, freddy_gpu_index_t** out);
int freddy_gpu_replica_count(const freddy_gpu_index_t* index);

/* Body of ivpq_search_in (ivpq_search_in.c:61-699), the kNN-join.  Arguments are the
 * SRF's own (ivpq_search_in.c:168-208).  iterations_out (may be NULL) receives the
 * number of alpha-doubling rounds.  out_*: [Q][k], sentinel 1000.0. */
int freddy_gpu_knn_join(freddy_gpu_index_t* ivpq, const float* queries, int32_t Q, int32_t k,
                        const int32_t* target_ids, int64_t n_targets, int32_t alpha, int32_t pvf,
                        int32_t method, int32_t use_target_lists, float confidence,
                        int32_t double_threshold, int32_t* out_ids, float* out_dist,
                        int32_t* iterations_out);

/* ---- next row after the PQ / IVFADC / kNN-join path: exact brute-force kNN --------
 * google_vecs_norm pinned as raw vectors.  Replaces the SQL of k_nearest_neighbour
 * (freddy--0.0.1.sql:426-454) and knn_in_exact (:991-1084):
 *   ORDER BY cosine_similarity_bytea(q, v.vector) DESC FETCH FIRST k ROWS ONLY
 * with cosine_similarity_bytea = the binary32 chain "scalar += v1[i] * v2[i]"
 * (core_functions.c:67-81), reproduced bit for bit.  Equal similarities are returned in
 * ascending id (PostgreSQL leaves their order unspecified). */
typedef struct freddy_vec_desc {
  int32_t d;
  int64_t N;
  const int32_t* ids;     /* [N] strictly ascending */
  const float* vectors;   /* [N][d] */
} freddy_vec_desc;
int freddy_gpu_pin_vectors(const freddy_vec_desc* desc, int device, freddy_gpu_index_t** out);
/* subset_ids == NULL: all rows; else "id = ANY(subset_ids)" (duplicates / unknown ids ignored).
 * out_ids/out_sim: [Q][k]; slots beyond the number of rows hold (-1, -inf). */
int freddy_gpu_exact_search(freddy_gpu_index_t* vecs, const float* queries, int32_t Q, int32_t k,
                            const int32_t* subset_ids, int64_t n_subset, int32_t* out_ids, float* out_sim);

/* ---- next row: grouping_pq ---------------------------------------------------------
 * Body of grouping_pq (freddy.c:1176-1401): for every row of the PQ table (subset_ids == NULL) or of
 * "id IN (subset_ids)" the nearest of G group vectors by ADC distance -- one LUT per group from the PQ
 * codebook (:1288-1299), positions summed in order (:1346-1351), strict "<" from minDist = 100 so the
 * first of equally near groups wins (:1337,1353-1356).  group_vectors: [G][d] in the order the caller
 * wants the ties broken (the reference: ascending group id).  out_ids / out_group: caller-allocated,
 * one slot per requested row (n_subset, or N); rows come back in table order, out_group[i] is the
 * group's index or -1 if no group is nearer than 100 (the reference leaves that case undefined).
 * *n_out receives the number of rows. */
int freddy_gpu_grouping_pq(freddy_gpu_index_t* pq, const float* group_vectors, int32_t G, const int32_t* subset_ids,
                           int64_t n_subset, int32_t* out_ids, int32_t* out_group, int64_t* n_out);

/* ---- next row: index build, encoding step --------------------------------------------
 * What index_creation/pq_index.py:65-92 (create_index / create_index_with_faiss) and ivfadc.py do once
 * the quantizers are trained: every vector's coarse cell (nearest of C centroids, all d dimensions) and
 * its PQ code (per position the nearest codeword of the -- residual, if coarse != NULL -- sub-vector),
 * by squareDistance with the lowest index on ties.  Standalone: no index handle, tables and vectors are
 * host arrays, results are host arrays.  out_cell may be NULL when coarse is NULL. */
typedef struct freddy_encode_desc {
  int32_t d, m, K;
  const float* codebook;   /* [m][K][d/m] */
  int32_t C;               /* 0: flat PQ (no coarse quantizer) */
  const float* coarse;     /* [C][d] or NULL */
} freddy_encode_desc;
int freddy_gpu_encode(const freddy_encode_desc* desc, int device, const float* vectors, int64_t N, int32_t* out_cell,
                      int16_t* out_codes);

/* Quantizer training (index_creation/quantizer_creation.py:13-52: scipy k-means for the coarse quantizer, per
 * sub-vector position for the PQ codebooks): Lloyd's algorithm on the device.  Initial centroids =
 * vectors[init_rows[c]] (NULL: vectors[c mod n]); every iteration assigns each vector to its nearest centroid by
 * squareDistance (lowest index on ties) and replaces each centroid by the binary32 mean of its members, summed
 * in index order; an empty cluster keeps its centroid.  centroids [k][d]; assign_out [n] (may be NULL) = the
 * assignment under the final centroids.  Deterministic: equals oracle/fo_kmeans bit for bit.  (The reference
 * seeds scipy randomly, so its own output is not reproducible; d <= 1024.) */
int freddy_gpu_kmeans(int device, const float* vectors, int64_t n, int32_t d, int32_t k, int32_t iters,
                      const int32_t* init_rows, float* centroids, int32_t* assign_out);

/* ---- next row: insert_batch --------------------------------------------------------------
 * Quantisation of NEW vectors as insert_batch does it (freddy.c:1557-1623): per vector the PQ code, the coarse
 * cell (argmin from minDistCoarse = 100, :1568-1575) with the code of the residual, the ivpq code, and the
 * two multi-index coarse codes (argmin from MAX_DIST = 1000, :1588-1597); codes = exact 1-NN by squareDistance
 * from minDist = 100 with the first entry winning ties (updateCodebook, index_utils.c:925-939).  A codebook
 * pointer that is NULL skips its part.  FREDDY_E_ARG if some (vector, position) has no centroid nearer than
 * 100 (undefined behaviour in the reference).  The codebook update itself (index_utils.c:940-956: a few
 * sequential float / double operations per new vector) and the table rows stay with the host. */
typedef struct freddy_insert_desc {
  int32_t d;
  int32_t pq_m, pq_K;       const float* pq_codebook;         /* [pq_m][pq_K][d/pq_m]   pq_codebook */
  int32_t res_m, res_K;     const float* residual_codebook;   /* residual_codebook */
  int32_t C;                const float* coarse;              /* [C][d] coarse_quantization */
  int32_t ivpq_m, ivpq_K;   const float* ivpq_codebook;       /* codebook_ivpq */
  int32_t multi_positions, multi_codes; const float* coarse_multi;   /* [positions][codes][d/positions] coarse_quantization_ivpq */
} freddy_insert_desc;
int freddy_gpu_insert_quantize(const freddy_insert_desc* desc, int device, const float* vectors, int64_t n,
                               int16_t* pq_codes /*[n][pq_m]*/, int32_t* coarse_id /*[n]*/, int16_t* residual_codes /*[n][res_m]*/,
                               int16_t* ivpq_codes /*[n][ivpq_m]*/, int16_t* coarse_multi_codes /*[n][multi_positions]*/);
/* Append rows to a pinned index in HBM (the INSERTs of updateProductQuantizationRelation /
 * updateWordVectorsRelation, index_utils.c:993-1074): ids must be larger than every id already pinned and
 * ascending.  pq: (ids, codes); ivf: (ids, coarse_id, codes) -- each row joins the end of its cell's inverted
 * list, the block layout is rebuilt on the device; ivpq: (ids, coarse_id, codes[, vectors]); vectors: (ids, vectors). */
int freddy_gpu_append_rows(freddy_gpu_index_t* index, int64_t n, const int32_t* ids, const int32_t* coarse_id,
                           const int16_t* codes, const float* vectors);
/* Replace the codebook of a pinned pq / ivf / ivpq index (updateCodebookRelation, index_utils.c:959-991) and
 * re-derive everything on the device that depends on it. */
int freddy_gpu_update_codebook(freddy_gpu_index_t* index, const float* codebook /*[m][K][d/m]*/);

/* ---- device-resident variant used for throughput measurement ----------------------------
 * Same as freddy_gpu_ivfadc_search, but queries / outputs are DEVICE pointers on the
 * index's device and all work is enqueued on `hip_stream` (a hipStream_t; NULL = the
 * library's own stream) without synchronising.  d_status[0] is set non-zero by the
 * device if some query needs a further probing round (rare: its first W cells hold
 * fewer than k rows); such queries keep partial results and the caller should re-run
 * them through freddy_gpu_ivfadc_search.  Single round only.
 *
 * Concurrency: everything a search writes besides its outputs lives in a workspace that belongs to the stream
 * the search is enqueued on (twelve slots per handle; with all taken a new stream takes over the least recently
 * used one after the device has drained).  Searches enqueued on DIFFERENT streams may therefore be in flight
 * together on one handle -- bench.py keeps four batches going that way.  The caller states how many with option
 * "scan_share" (an explicit contract; the library does not guess it): a batch's persistent scan then takes
 * n_cus / scan_share CUs so that the scans run side by side and the small kernels fit in between.  Searches on the
 * same stream are ordered by it.  Host threads may call concurrently as long as they do not share a stream (the
 * slot table and the profile map are locked).  The synchronous calls above use the library's own streams.
 * freddy_gpu_last_* report on the most recent call. */
int freddy_gpu_ivfadc_search_dev(freddy_gpu_index_t* ivf, const float* d_queries, int32_t Q,
                                 int32_t k, int32_t W, float sentinel, int32_t found_rule,
                                 int32_t* d_out_ids, float* d_out_dist, int32_t* d_status,
                                 void* hip_stream);
int freddy_gpu_pq_search_dev(freddy_gpu_index_t* pq, const float* d_queries, int32_t Q, int32_t k,
                             float sentinel, int32_t* d_out_ids, float* d_out_dist,
                             void* hip_stream);

/* ---- diagnostics ------------------------------------------------------------------------ */

/* Stage timers of the most recent freddy_gpu_knn_join call on this handle, in seconds, under the names the
 * reference reports with elog(INFO, "TRACK <stage> %f") (ivpq_search_in.c:234-697; scraped off the
 * connection by evaluation/tracking.py).  The PostgreSQL host re-emits them with the same elog lines. */
typedef struct freddy_track {
  double precomputation_time;                /* :294  here: coarse sub-distances + side sorts (the LUTs are built inside the join kernel) */
  double determine_coarse_quantization_time; /* :341  multi-index traversal, summed over the alpha rounds */
  double query_construction_time;            /* :397  per-query cell lists */
  double data_retrieval_time;                /* :403  "fq.id IN (targets)": ids resolved and bucketed by cell on the device */
  double computation_time;                   /* :632  the join kernel: ADC / exact distances, selection, replay */
  double pv_computation_time;                /* :627  0: post verification happens inside the join kernel */
  double recalculate_query_indices_time;     /* :671 */
  double total_time;                         /* :697 */
  double join_kernel_time;                   /* HIP-event time of the join kernel launches alone (inside computation_time) */
  int64_t candidate_rows;                    /* sum over queries of the target rows in their selected cells (what the kernel scans) */
  int32_t iterations;                        /* alpha-doubling rounds */
  int32_t reserved;
  int64_t host_traversals;                   /* (query, round) pairs whose multi-index traversal ran on the host heap: every one with more than
                                              * 1024 cells; with the device traversal only those it hands back (equal keys in the taken prefix,
                                              * or the host's libm disagreeing with the proposed stop) */
  int64_t libm_checks;                       /* evaluations of getConfidenceHyp by the host's libm on device-proposed stops (only where the device's
                                              * own value lies within 1e-5 of the confidence) */
} freddy_track;
int freddy_gpu_last_track(const freddy_gpu_index_t* ivpq, freddy_track* out);
/* The same with the caller's idea of the struct's size: at most `out_size` bytes are written (a host built against an older
 * header, whose freddy_track ends earlier, gets the fields it knows); returns the number of bytes written, or < 0.  Hosts
 * that are built separately from the library (the PostgreSQL extension) call this one. */
int freddy_gpu_last_track_sized(const freddy_gpu_index_t* ivpq, void* out, size_t out_size);

/* ABI version of the library: bumped whenever a struct of this header grows or an entry point changes meaning.  A host
 * compares it with the FREDDY_GPU_ABI_VERSION it was compiled against before its first other call into the library
 * (pg/freddy_gpu_glue.c: ensure_exit_hook(), at the top of every freddy_glue_* entry) and refuses a mismatch instead of
 * overrunning a stack variable. */
#define FREDDY_GPU_ABI_VERSION 4
int freddy_gpu_abi_version(void);

/* Options of a pinned index (the FREDDY_GPU_* environment variables of the same names are read once, at pin time).  No setting
 * changes a result.
 *   deployment:  "scan_share" (the batches that share the chip with one of this handle's: the batches the caller keeps in flight
 *                through the *_dev entry points, one stream each, or the other BACKENDS searching at the same time -- a persistent
 *                scan takes n_cus / scan_share CUs; default 0 = auto: the whole chip, or half of it for a host-buffer call that
 *                starts while another backend (process) of the same GPU is searching: the library keeps a registry of live backends per
 *                physical GPU in /dev/shm and also picks GPU_MAX_HW_QUEUES from it before its first HIP call -- 6 alone, 2 beside
 *                others),
 *                "reserve_cus" (CUs a persistent scan leaves free), "pipeline_batch" / "pipeline_lanes" (host-buffer IVFADC calls:
 *                queries per sub-batch, 2048; sub-batches in flight, 1..4), "coarse_pieces" (1: a call of one sub-batch launches its
 *                cell selection per staged piece of the queries), "lut_budget_mb" (workspace cap per call)
 *   paths (each has GPU tests of its own):  "fused" (-1 auto, 0 generic kernels, 1 cell-grouped scans always), "fused_kernel"
 *                (5 filter + refine on int16 slabs, 3 the reference's arithmetic for every row), "coarse_approx" (1: cell selection
 *                as filter + refine, 0: every coarse distance exact), "one_launch" (1: a host-buffer call with ONE query -- the
 *                reference's own call shape -- is a single launch, pq_one_kernel / ivf_one_kernel, the host polls the kernel's
 *                completion word; 0: the multi-launch path.  A handle whose grid once failed to meet within the kernel's bounded
 *                polls switches itself to 0), "pq_fused" (batches over the flat PQ table through the cell-grouped scan over
 *                pseudo-lists of 4096 rows: -1 = from 16 queries on, 0 never, 1 always), "sparse_items" (cells that at most this
 *                many queries of a batch probe are scanned item by item; default 2, 0 = never, a negative value forces it for
 *                cells of up to that many items whatever the batch), "running_bound" (1: the scan's work entries share a
 *                per-query bound of the L-th smallest cheap distance), "codes_u8" (1: indexes with K <= 256 are scanned from one
 *                byte per code by the kernel that keeps a whole work entry's slab in LDS; 2: one byte per code, the six-phase kernel;
 *                0: the int16 layout), "exact_filter" (exact brute-force kNN as f16-split MFMA filter + exact
 *                refine: -1 = tables of >= 8192 rows and k <= 32, 0 never -- and no fragment copy of a table pinned with it --,
 *                1 always)
 *   self-checks (tests):  "check_brackets" (bit 0: the scan keeps and the merge refines EVERY probed row, bit 1: the cell
 *                selection refines every cell, bit 2: exact kNN refines every row -- each with its proven bracket compared with
 *                the reference's value: freddy_gpu_filter_bound_violations / _checked), "join_host_traversal",
 *                "join_libm_margin_ppm" */
int freddy_gpu_set_option(freddy_gpu_index_t* index, const char* name, int64_t value);

/* Thread-local message of the last failing call; valid until the next call. */
const char* freddy_gpu_last_error(void);

/* Per-kernel timing with HIP events on the launch stream.  enable=1 starts recording
 * (and clears earlier records); freddy_gpu_profile_read() synchronises, then reports up
 * to `cap` kernels: name, number of launches, total milliseconds.  Returns the number
 * of distinct kernels (or <0). */
int freddy_gpu_profile_enable(freddy_gpu_index_t* index, int32_t enable);
int freddy_gpu_profile_read(freddy_gpu_index_t* index, int32_t cap, char (*names)[64],
                            int64_t* launches, double* total_ms);

/* Sizes the caller may want for roofline arithmetic. */
int64_t freddy_gpu_index_bytes(const freddy_gpu_index_t* index);   /* HBM footprint of the pinned index */
/* Sum of list lengths the last ivfadc call scanned (all queries, all probes). */
int64_t freddy_gpu_last_scanned_rows(const freddy_gpu_index_t* index);
/* Distinct cells the last probing round touched (cell-grouped scans) and the rows of their lists: the bytes a
 * scan that reads every probed list once per batch must move, as the reference's loop does (freddy.c:939-974). */
int freddy_gpu_last_probed_cells(const freddy_gpu_index_t* index, int64_t* n_cells, int64_t* rows);
/* Self-check of the filter + refine IVFADC scan: every row that reaches the exact stage
 * has both its proven bracket [d_lo, d_lo + E] and the reference's distance d in hand; this returns how
 * many such rows had d outside the bracket since the index was pinned (0 unless the error analysis is
 * wrong for some input; <0 on a HIP error).  Synchronises the device. */
int64_t freddy_gpu_filter_bound_violations(const freddy_gpu_index_t* index);
/* How many rows that check has seen -- counted only in the tests' refine-every-row mode
 * (option check_brackets, bit 0), where it is the number of probed rows; 0 otherwise. */
int64_t freddy_gpu_filter_bound_checked(const freddy_gpu_index_t* index);
/* The coarse-cell selection is a filter + refine too (MFMA distances with a proven bracket, the reference's
 * squareDistance for the candidate cells): refined cells whose distance left the bracket are
 * INCLUDED in freddy_gpu_filter_bound_violations; this returns how many cells the tests' refine-every-cell mode
 * (option check_brackets, bit 1) has checked. */
int64_t freddy_gpu_coarse_bound_checked(const freddy_gpu_index_t* index);

#ifdef __cplusplus
}
#endif
#endif /* FREDDY_GPU_H */
