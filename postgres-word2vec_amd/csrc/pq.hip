// pq.hip -- pq_search / pq_search_in(_batch) (freddy.c:28-152, :1028-1157, :414-653) and grouping_pq (freddy.c:1176-1401).
#include "internal.h"

#include "kernels.h"
#include "scan_common.h"
#include "fused5.h"   // query_codebook5_body: the table units of pq_front_kernel
#include "one.h"
#include "io_kernels.h"

// ---------------------------------------------------------------------------------------
// exhaustive / subset PQ
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// Batches over the flat PQ table through the cell-grouped filter + refine scan (fused5.h).
//
// adc_scan_kernel runs one workgroup per (query, chunk): every query re-reads the code table from the caches and gathers
// 4-byte LUT entries one (query, row, position) at a time.  The IVFADC scan shares a chunk's rows among 16 queries and
// gathers eight 16-bit table values per LDS access -- and pq_search's distance is ivfadc_search's with a residual
// r = q - 0: the flat table is pinned a second time only as METADATA -- pseudo-lists of 4096 consecutive rows
// (FUSED_UNIT_BLOCKS blocks; the packed codes are shared), a zero centroid per list, the row terms sum_p |c|^2, the rows'
// ids -- in a shadow index of kind IVF, and a batch "probes" every list: items (query, list) for all pairs, no coarse
// distances, no plan.  The exact stage then evaluates (q_i - 0) - c_i: x - 0 = x exactly, so its squares, their
// order of summation (index_utils.c:500-508, 1126-1133) and the guarded insertion in ascending id order are pq_search's
// (freddy.c:28-152).  The item's bound on |r|^2 is squareDistance(q, 0) evaluated the reference's way.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_shadow_meta_kernel(const int32_t* __restrict__ pos, const int32_t* __restrict__ ids,
                                                            int64_t n_blocks, int64_t n_rows, int lists, int32_t* __restrict__ list_off,
                                                            int32_t* __restrict__ blk_off, int32_t* __restrict__ blk_cell,
                                                            int32_t* __restrict__ pos_ids) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i <= lists) {
    const int64_t r = i * (FUSED_UNIT_BLOCKS * 64), b = i * FUSED_UNIT_BLOCKS;
    list_off[i] = (int32_t)(r < n_rows ? r : n_rows);
    blk_off[i] = (int32_t)(b < n_blocks ? b : n_blocks);
  }
  if (i < n_blocks) blk_cell[i] = (int32_t)(i / FUSED_UNIT_BLOCKS);
  if (i < n_blocks * 64) { const int32_t r = pos[i]; pos_ids[i] = r >= 0 ? ids[r] : -1; }
}

// One workgroup per query: A = squareDistance(q, 0) (sequential binary32, index_utils.c:500-508), the query's items --
// one per pseudo-list -- and its slot in every list's bucket.
__global__ __launch_bounds__(256) void pq_items_kernel(const float* __restrict__ queries, int Q, int d, int lists, int64_t n_rows,
                                                      int32_t* __restrict__ item_cell, int32_t* __restrict__ item_query,
                                                      float* __restrict__ item_dist, int32_t* __restrict__ cell_items,
                                                      int32_t* __restrict__ cell_count, int32_t* __restrict__ round_rows) {
  __shared__ float A_s;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) {
    float acc = 0.0f;
    for (int i = 0; i < d; ++i) { const float t = queries[(size_t)q * d + i] - 0.0f; acc = acc + t * t; }
    A_s = acc;
    round_rows[q] = (int32_t)n_rows;
  }
  __syncthreads();
  const float A = A_s;
  for (int c = threadIdx.x; c < lists; c += 256) {
    const int it = q * lists + c;
    item_cell[it] = c; item_query[it] = q; item_dist[it] = A;
    cell_items[(size_t)c * Q + q] = it;
    if (q == 0) cell_count[c] = Q;
  }
}

static bool pq_fused_shape(const freddy_gpu_index* ix) {
  return ix->kind == KIND_PQ && ix->cbR && ix->m == 12 && ix->S == 25 && ix->K <= FUSED_T * FUSED_E && ix->n_blocks > 0 && ix->N > 0;
}

static bool pq_use_fused(const freddy_gpu_index* ix, int Q, int k) {
  if (ix->tune.pq_fused == 0 || !pq_fused_shape(ix) || 2 * k > 64) return false;
  return ix->tune.pq_fused > 0 || Q >= 16;
}
// A batch over the flat PQ table needs no probe plan and no work table: every query "probes" every pseudo-list, so the
// work entries are (group of 16 queries, pseudo-list) and their records follow from the query's table scale alone.  One
// workgroup per query: |q|^2 in the reference's order (squareDistance(q, 0): the coarse distance of the zero centroid, the
// bound item_bounds builds on), then the query's lane of every record of its group; the first query of a group also writes
// the records' headers.  Replaces pq_items + work_table + entry_record kernels (27 us of three dependent launches).
// item index = q * W + list (W >= lists: padded to a multiple of the merge's slices; the padding items have no entry, their
// survivor regions stay zero).
struct PqFrontArgs {
  const float* queries; int Q, d, lists, W; int64_t n_rows;
  const int32_t* blk_off; const int32_t* list_off;
  const float* cbT; const float* cmax; const float* pmax;
  float* qn; float* qscale; uint32_t* qc; int m, K;
  uint32_t* qc8;   // K <= 256: the compact copy of the table (fused8.h); NULL: not wanted
  float sentinel;
  int32_t* item_cell; int32_t* item_query; float* item_dist; int32_t* round_rows; int32_t* records; int32_t* n_groups;
  ZeroArgs z;   // the call's scratch that must start at zero (counters, running bounds, survivor counts): no memset launches in front
};
__device__ __forceinline__ void pq_records_body(const PqFrontArgs& a, int q, unsigned char* smem) {
  const float* __restrict__ queries = a.queries;
  const int Q = a.Q, d = a.d, lists = a.lists, W = a.W;
  const int64_t n_rows = a.n_rows;
  const int32_t* __restrict__ blk_off = a.blk_off; const int32_t* __restrict__ list_off = a.list_off;
  const float* __restrict__ pmax = a.pmax;
  const float sentinel = a.sentinel;
  int32_t* __restrict__ item_cell = a.item_cell; int32_t* __restrict__ item_query = a.item_query; float* __restrict__ item_dist = a.item_dist;
  int32_t* __restrict__ round_rows = a.round_rows; int32_t* __restrict__ records = a.records; int32_t* __restrict__ n_groups = a.n_groups;
  float* sqs = reinterpret_cast<float*>(smem);          // [1024]
  float* qn_s = sqs + 1024;                              // [16] |q_p| rounded up, as query_codebook5_body forms it
  float* fs = qn_s + 16;                                 // [0] A, [1] scale
  const int tid = threadIdx.x;
  for (int i = tid; i < d; i += 256) { const float t = queries[(size_t)q * d + i] - 0.0f; sqs[i] = t * t; }
  // the query's per-position norms and its table scale: the very operations of query_codebook5_body (same order, same roundings),
  // so that this workgroup needs nothing from the table units of the same launch
  if (tid < 16) {
    const int pp = tid, S = d / a.m;
    float best = 0.0f;
    if (pp < a.m) {
      float n2 = 0.0f;
      for (int j = 0; j < S; ++j) { const float v = queries[(size_t)q * d + pp * S + j]; n2 = __builtin_fmaf(v, v, n2); }
      const float nrm = __builtin_sqrtf(n2) * (1.0f + 1e-5f);
      qn_s[pp] = nrm;
      best = 2.0f * nrm * a.cmax[pp];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (pp == 0) fs[1] = best * (1.0f / (float)FILT5_VMAX) * (1.0f + 1e-6f);
  }
  __syncthreads();
  if (tid == 0) {
    float acc = 0.0f;
    for (int i = 0; i < d; ++i) acc = acc + sqs[i];     // index_utils.c:500-508, i ascending
    fs[0] = acc;
    round_rows[q] = (int32_t)n_rows;
    if (q == 0) n_groups[0] = ((Q + SCAN5_G - 1) / SCAN5_G) * lists;
  }
  __syncthreads();
  const float A = fs[0];
  const float sc = fs[1];
  const ItemBounds ib = item_bounds(A, filter_width5<12>(qn_s, pmax, sc), sentinel);
  const int g = q / SCAN5_G, slot = q % SCAN5_G;
  const int cnt = (Q - g * SCAN5_G < SCAN5_G) ? Q - g * SCAN5_G : SCAN5_G;
  for (int c = tid; c < lists; c += 256) {
    const int it = q * W + c;
    item_cell[it] = c; item_query[it] = q; item_dist[it] = A;
    int32_t* rec = records + ((size_t)g * lists + c) * REC_DW;
    rec[8 + slot] = it;
    rec[24 + slot] = q;
    rec[40 + slot] = (int32_t)__float_as_uint(ib.off);
    rec[56 + slot] = (int32_t)__float_as_uint(ib.e);
    rec[72 + slot] = (int32_t)__float_as_uint(ib.shift);
    rec[88 + slot] = (int32_t)ib.lo_bits;
    rec[104 + slot] = (int32_t)ib.hi_bits;
    rec[128 + slot] = (int32_t)__float_as_uint(sc < 1e30f ? sc : 0.0f);
    {   // (entry_record5_kernel: the item's coarse distance -- here |q|^2, the zero centroid's -- as an interval)
      const bool fin = ib.e < 1e30f && A >= 0.0f && A < 1e30f;
      rec[144 + slot] = (int32_t)__float_as_uint(fin ? A * (1.0f + 2e-5f) : __uint_as_float(0x7f800000u));
      rec[160 + slot] = (int32_t)__float_as_uint(fin ? A * (1.0f - 2e-5f) : 0.0f);
    }
    if (slot == 0) {
      const int b0 = blk_off[c];
      rec[0] = c; rec[1] = cnt; rec[2] = 0; rec[3] = b0; rec[4] = blk_off[c + 1] - b0; rec[5] = list_off[c + 1] - list_off[c];
      // the slots beyond the group's queries: no item, the first query's number (a valid table), no bounds (entry_record5_kernel)
      const ItemBounds none = item_bounds(0.0f, 0.0f, sentinel);
      for (int u = cnt; u < SCAN5_G; ++u) {
        rec[8 + u] = -1; rec[24 + u] = q;
        rec[40 + u] = (int32_t)__float_as_uint(none.off); rec[56 + u] = (int32_t)__float_as_uint(none.e); rec[72 + u] = (int32_t)__float_as_uint(none.shift);
        rec[88 + u] = (int32_t)none.lo_bits; rec[104 + u] = (int32_t)none.hi_bits; rec[128 + u] = 0;
        rec[144 + u] = (int32_t)0x7f800000; rec[160 + u] = 0;
      }
    }
  }
}

// The table units of query_codebook5_kernel and the record workgroups above as ONE launch (neither needs the other: the record
// workgroups form the query's scale themselves): a dependent launch less in a PQ batch's chain.
__global__ __launch_bounds__(256) void pq_front_kernel(PqFrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n_table = a.m * ((a.Q + 15) / 16);
  const int b = blockIdx.x;
  {
    const int gtid = b * 256 + (int)threadIdx.x, gsz = (int)gridDim.x * 256;
#pragma unroll
    for (int r = 0; r < 5; ++r)
      for (int i = gtid; i < a.z.n[r]; i += gsz) a.z.p[r][i] = 0u;
  }
  if (b < n_table) query_codebook5_body<25, 16>(a.queries, a.cbT, a.cmax, a.qn, a.qscale, a.qc, a.Q, a.d, a.m, a.K, b % a.m, b / a.m, smem, a.qc8);
  else pq_records_body(a, b - n_table, smem);
}

// survivor regions: 32 KiB per (query, pseudo-list) within the workspace budget; the buckets [lists][queries] within 256 MiB
static int pq_fused_queries_per_chunk(const freddy_gpu_index* ix, int64_t n_blocks) {
  const size_t lists = (size_t)((n_blocks + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS);
  size_t n = ((size_t)ix->tune.lut_budget_mb << 20) / (sizeof(u64) * lists * FUSED_NW * FUSED_RMAX * 64);
  n = std::min<size_t>(n, ((size_t)256 << 20) / (sizeof(int32_t) * lists));
  return (int)std::max<size_t>(16, std::min<size_t>(n, 1u << 16));
}

// An IVF-shaped view (*view; created on first use) of `n_rows` rows in `n_blocks` packed blocks: pseudo-lists, zero centroids,
// row terms, ids.  Everything is enqueued on s; nothing is synchronised.
static int pq_view_refresh(freddy_gpu_index* ix, freddy_gpu_index** view, hipStream_t s, const uint32_t* packed, const int32_t* pos,
                           int64_t n_blocks, int64_t n_rows) {
  freddy_gpu_index* fx = *view;
  if (!fx) {
    fx = new freddy_gpu_index();
    fx->shadow_of = ix;
    fx->kind = KIND_IVF; fx->device = ix->device; fx->stream = ix->stream; fx->n_cus = ix->n_cus;
    fx->d = ix->d; fx->m = ix->m; fx->K = ix->K; fx->S = ix->S; fx->M2 = ix->M2;
    if (hipMalloc((void**)&fx->viol, 4 * sizeof(int32_t)) != hipSuccess || hipMemset(fx->viol, 0, 4 * sizeof(int32_t)) != hipSuccess) {
      free_index(fx);
      return fail(FREDDY_E_NOMEM, "device allocation failed (PQ table as pseudo-lists)");
    }
    *view = fx;
  }
  fx->tune = ix->tune;
  fx->cbT = ix->cbT; fx->cbR = ix->cbR; fx->pmax = ix->pmax; fx->cmaxp = ix->cmaxp; fx->cbF = ix->cbF;   // shared with the owner
  fx->packed = const_cast<uint32_t*>(packed);
  fx->packed8 = (packed == ix->packed) ? ix->packed8 : nullptr; fx->packed8_own = false;   // (a subset's gathered rows: the int16 layout)
  fx->N = n_rows; fx->n_blocks = n_blocks; fx->max_list_blocks = FUSED_UNIT_BLOCKS;
  const int lists = (int)((n_blocks + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS);
  fx->C = lists;
  const size_t slots = (size_t)n_blocks * 64;
  if (fx->v_coarse.ensure(sizeof(float) * (size_t)lists * ix->d) || fx->v_list_off.ensure(sizeof(int32_t) * ((size_t)lists + 1)) ||
      fx->v_blk_off.ensure(sizeof(int32_t) * ((size_t)lists + 1)) || fx->v_blk_cell.ensure(sizeof(int32_t) * (size_t)n_blocks) ||
      fx->v_pos.ensure(sizeof(int32_t) * slots) || fx->v_rterm.ensure(sizeof(float) * slots))
    return fail(FREDDY_E_NOMEM, "device allocation failed (PQ table as pseudo-lists)");
  fx->coarse = fx->v_coarse.as<float>(); fx->list_off = fx->v_list_off.as<int32_t>(); fx->blk_off = fx->v_blk_off.as<int32_t>();
  fx->blk_cell = fx->v_blk_cell.as<int32_t>(); fx->pos = fx->v_pos.as<int32_t>(); fx->rterm = fx->v_rterm.as<float>();
  HIP_TRY(hipMemsetAsync(fx->coarse, 0, sizeof(float) * (size_t)lists * ix->d, s));
  hipLaunchKernelGGL(pq_shadow_meta_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, pos, ix->ids, n_blocks, n_rows, lists,
                     fx->list_off, fx->blk_off, fx->blk_cell, fx->pos);
  hipLaunchKernelGGL(row_term_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, fx->packed, fx->blk_cell, fx->coarse, fx->cbR,
                     fx->rterm, (int64_t)slots, fx->M2, fx->d, fx->m, fx->K, fx->S);
  HIP_TRY(hipGetLastError());
  return 0;
}

// the whole table's view: built once (and again after rows were appended or the codebook was replaced)
int pq_shadow_build(freddy_gpu_index* ix) {
  if (ix->pq_shadow) return 0;
  if (int rc = pq_view_refresh(ix, &ix->pq_shadow, ix->stream, ix->packed, ix->pos, ix->n_blocks, ix->N)) {
    if (ix->pq_shadow) { free_index(ix->pq_shadow); ix->pq_shadow = nullptr; }
    return rc;
  }
  HIP_TRY(hipStreamSynchronize(ix->stream));   // (searches may come in on other streams)
  return 0;
}


static int pq_fused_chunk(freddy_gpu_index* ix, freddy_gpu_index* fx, hipStream_t s, const float* d_q, int Q, int k, float sentinel,
                          int32_t* d_out_ids, float* d_out_dist) {
  fx->tune = ix->tune;
  Workspace* ws = workspace_for(fx, s);
  const int lists = fx->C, m = fx->m, K = fx->K;
  // the merge of a small batch over many pseudo-lists: four workgroups per query, each over a quarter of the lists (64 queries
  // x 1 960 survivor regions on 64 workgroups took 62 us on a quarter of the chip); the item space of a query is padded to
  // a multiple of the slices
  const int SL = (lists >= 32 && Q <= 256 && (size_t)lists * FUSED_NW > 256) ? 4 : 0;
  const int W = SL ? ((lists + SL - 1) / SL) * SL : lists;
  IvfRun r;
  r.ix = fx; r.ws = ws; r.s = s; r.d_q = d_q; r.Q = Q; r.k = k; r.W = W; r.L = 2 * k;
  r.sentinel = sentinel; r.cell_limit = 0.0f; r.d_out_ids = d_out_ids; r.d_out_dist = d_out_dist; r.d_status = nullptr;
  r.found_rule = 0; r.upi = 1; r.fused = true; r.scan_kernel = 5; r.tiled = false; r.zeroed = false; r.approx = false;
  r.records_ready = true; r.merge_slices = SL;
  r.n_active = Q; r.round = 0; r.active = nullptr;
  r.share = scan_share_now(ix->tune.scan_share, false, ix->device);   // (the caller's contract: its batches in flight on this handle)
  const size_t items = (size_t)Q * W;
  const size_t n_entries = (size_t)((Q + SCAN5_G - 1) / SCAN5_G) * lists;
  if (ws->w_item_cell.ensure(sizeof(int32_t) * items) || ws->w_item_query.ensure(sizeof(int32_t) * items) ||
      ws->w_item_dist.ensure(sizeof(float) * items) || ws->w_rows.ensure(sizeof(int32_t) * Q) || ws->w_cand.ensure(sizeof(int32_t) * 2 * Q) ||
      ws->w_found.ensure(sizeof(int32_t) * Q) || ws->w_act0.ensure(sizeof(int32_t) * Q) || ws->w_act1.ensure(sizeof(int32_t) * Q) ||
      ws->w_cnt.ensure(sizeof(int32_t) * 8) || ws->w_records.ensure(sizeof(int32_t) * REC_DW * n_entries) ||
      ws->w_surv.ensure(sizeof(u64) * items * r.upi * FUSED_NW * FUSED_RMAX * 64) ||
      ws->w_surv_cnt.ensure(sizeof(int32_t) * items * r.upi * FUSED_NW) ||
      ws->w_qc.ensure(sizeof(uint32_t) * (size_t)Q * m * 640) || ws->w_qn.ensure(sizeof(float) * (size_t)Q * m * 2))   // (512 + 128: the compact copy for K <= 256, fused8.h)
    return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d over %d pseudo-lists)", Q, lists);
  r.next = ws->w_act0.as<int32_t>();
  PlanArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.item_cell = ws->w_item_cell.as<int32_t>(); pa.item_query = ws->w_item_query.as<int32_t>(); pa.item_dist = ws->w_item_dist.as<float>();
  pa.round_rows = ws->w_rows.as<int32_t>(); pa.n_active = Q; pa.C = lists; pa.W = W;
  WorkTable wt;
  wt.max_groups = n_entries; wt.group_cell = wt.group_first = wt.group_cnt = nullptr;
  wt.n_groups = ws->w_cnt.as<int32_t>() + 1; wt.work_counter = ws->w_cnt.as<int32_t>() + 2;
  wt.sp_cap = 0; wt.sp_cell = wt.sp_first = wt.sp_chunk = nullptr; wt.sp_counter = ws->w_cnt.as<int32_t>() + 3; wt.n_sparse = ws->w_cnt.as<int32_t>() + 4;
  // ONE launch: the table units (query x codebook, int16) and the record workgroups -- the entry records straight from the
  // queries' table scales: no item / work-table / record kernels (pq_front_kernel)
  PqFrontArgs fa;
  fa.queries = d_q; fa.Q = Q; fa.d = fx->d; fa.lists = lists; fa.W = W; fa.n_rows = fx->N; fa.blk_off = fx->blk_off; fa.list_off = fx->list_off;
  fa.cbT = fx->cbF; fa.cmax = fx->cmaxp; fa.pmax = fx->pmax; fa.qn = ws->w_qn.as<float>(); fa.qscale = ws->w_qn.as<float>() + (size_t)Q * m;
  fa.qc = ws->w_qc.as<uint32_t>(); fa.m = m; fa.K = K; fa.sentinel = sentinel;
  fa.qc8 = (fx->packed8 && fx->tune.codes_u8 == 1 && K <= 256 && m == 12) ? ws->w_qc.as<uint32_t>() + (size_t)Q * m * 512 : nullptr; fa.item_cell = pa.item_cell; fa.item_query = pa.item_query;
  fa.item_dist = pa.item_dist; fa.round_rows = pa.round_rows; fa.records = ws->w_records.as<int32_t>(); fa.n_groups = wt.n_groups;
  // (w_cnt[1] = n_groups is WRITTEN by this launch's record workgroups: not among the words it clears)
  fa.z.p[0] = ws->w_cnt.as<uint32_t>(); fa.z.n[0] = 1;
  fa.z.p[1] = ws->w_cnt.as<uint32_t>() + 2; fa.z.n[1] = 6;
  fa.z.p[2] = ws->w_cand.as<uint32_t>() + Q; fa.z.n[2] = Q;   // the queries' running bounds (FilterArgs::tau_run)
  fa.z.p[3] = ws->w_surv_cnt.as<uint32_t>(); fa.z.n[3] = (int)(items * r.upi * FUSED_NW);
  fa.z.p[4] = nullptr; fa.z.n[4] = 0;
  const size_t front_lds = std::max<size_t>((size_t)query_codebook5_lds<25, 16>(), (size_t)(1024 + 16 + 2) * sizeof(float));
  timed_launch(fx, s, "pq_front", [&] {
    hipLaunchKernelGGL(pq_front_kernel, dim3((unsigned)(m * ((Q + 15) / 16) + Q)), dim3(256), front_lds, s, fa);
  });
  HIP_TRY(hipGetLastError());
  return ivf_scan_filter(r, pa, wt);
}

// ONE query over the flat table as one launch (one.h): table slices, grid barrier, scan, last-arriver merge.  `err` is a
// word of mapped host memory the kernel sets when one of its bounded polls ran out (the grid was not co-resident): the
// caller then re-arms the counters and takes the three-launch path.
static bool pq_one_shape(const freddy_gpu_index* ix, int Q, int k, int64_t n_blocks) {
  return ix->tune.one_launch && !ix->one_launch_failed && Q == 1 && ix->m == 12 && ix->S == 25 && ix->K <= 1024 && (ix->K & 3) == 0 && ix->d == 300 &&
         2 * k <= 64 && n_blocks >= 64 && (int64_t)ix->h_ids.size() == ix->N;
}
static int pq_one(freddy_gpu_index* ix, hipStream_t s, const float* h_q, int k, float sentinel, const int32_t* blk_off,
                  const uint32_t* packed, const int32_t* pos, int64_t n_blocks, int32_t* d_out_ids, float* d_out_dist, int32_t* err) {
  Workspace* ws = workspace_for(ix, s);
  const int K = ix->K, L = 2 * k;
  const size_t lutN = (size_t)12 * K;
  // (one workgroup per CU at most: all co-resident; the last arriver stages every list in LDS: G * L keys within 56 KB)
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)ix->n_cus, (int64_t)256, (n_blocks + ONE_WAVES - 1) / ONE_WAVES, (int64_t)(56 * 1024) / (8 * L)}));
  const int chunk_blocks = (int)((n_blocks + G - 1) / G);
  const size_t part_off = (lutN * sizeof(float) + 255) & ~(size_t)255;
  uint32_t epoch = 0;
  if (int rc = one_buffer(ws, s, (1ull << 60) | ((uint64_t)K << 32) | ((uint64_t)G << 16) | (uint64_t)L, part_off + sizeof(u64) * (size_t)G * L, &epoch)) return rc;
  static const bool one_prof = getenv("FREDDY_GPU_ONE_PROF") != nullptr;
  if (one_prof && ws->w_one.ensure(256)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  OneArgs a;
  memcpy(a.qv, h_q, sizeof(a.qv)); a.cbT = ix->cbT; a.lut_g = ws->w_oneb.as<float>(); a.blk_off = blk_off; a.packed = packed; a.pos = pos;
  a.pos_to_id = nullptr;   // (positions out: the caller maps them through its host copy of the ids -- no dependent gather at the kernel's end)
  a.part = reinterpret_cast<u64*>(ws->w_oneb.as<char>() + part_off); a.out_ids = d_out_ids; a.out_dist = d_out_dist;
  a.epoch = epoch; a.err = err;
  a.prof = one_prof ? ws->w_one.as<unsigned long long>() + 8 : nullptr;
  a.K = K; a.L = L; a.k = k; a.chunk_blocks = chunk_blocks; a.sentinel = sentinel;
  memcpy(&a.sentinel_bits, &sentinel, 4);
  const size_t lds = std::max(((lutN * 4 + 15) & ~(size_t)15) + (size_t)ONE_WAVES * 64 * sizeof(u64),
                              (size_t)ONE_WAVES * 64 * sizeof(u64) + (size_t)G * L * sizeof(u64));
  timed_launch(ix, s, "pq_one", [&] { hipLaunchKernelGGL((pq_one_kernel<25>), dim3((unsigned)G), dim3(ONE_WG), lds, s, a); });
  HIP_TRY(hipGetLastError());
  ws->one_pending = false;
  return 0;
}

// Queries per chunk of the generic scan over `n_blocks` row blocks: a LUT per query, one list of L keys per (query, 4096-row
// chunk) -- the finest chunks pq_chunk may choose -- and, from k = 513 on, the keys the selection passes keep (bigk.h).
static int pq_queries_per_chunk(const freddy_gpu_index* ix, int64_t n_blocks, int k) {
  const size_t L = (size_t)std::min(2 * k, 1024), nchunk = (size_t)std::max<int64_t>(1, (n_blocks + 63) / 64);
  size_t per_query = sizeof(float) * (size_t)ix->m * ix->K + sizeof(u64) * nchunk * L;
  if (2 * k > 1024) per_query += sizeof(u64) * (size_t)((2 * k + 1024 - 1) / 1024) * 1024;
  const size_t n = ((size_t)ix->tune.lut_budget_mb << 20) / per_query;
  return (int)std::min<size_t>(std::max<size_t>(n, 1), (size_t)1 << 20);
}

static int pq_chunk(freddy_gpu_index* ix, hipStream_t s, const float* d_q, int Q, int k, float sentinel,
                    const int32_t* blk_off, const uint32_t* packed, const int32_t* pos, int64_t n_blocks,
                    int32_t* d_out_ids, float* d_out_dist) {
  Workspace* ws = workspace_for(ix, s);
  const int m = ix->m, K = ix->K;
  const int L = std::min(2 * k, 64 * 16);
  const size_t lutN = (size_t)m * K;
  // enough (query, chunk) workgroups to fill the chip, but chunks long enough to amortise
  // the 48 KiB LUT staging
  int chunk_blocks = 64;   // 4096 rows; longer chunks once there are enough (query, chunk) workgroups
  while ((n_blocks + chunk_blocks - 1) / chunk_blocks * (int64_t)Q > 4096 && chunk_blocks < 8192) chunk_blocks *= 2;
  const int nchunk = (int)std::max<int64_t>(1, (n_blocks + chunk_blocks - 1) / chunk_blocks);
  if (ws->w_lut.ensure(sizeof(float) * (size_t)Q * lutN) ||
      ws->w_part.ensure(sizeof(u64) * (size_t)Q * nchunk * L))   // (one list of L keys per (query, chunk): kernels.h adc_scan_kernel)
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  if (int rc = launch_lut(ix, s, d_q, nullptr, ws->w_lut.as<float>(), Q)) return rc;
  ScanArgs sa;
  sa.lut = ws->w_lut.as<float>(); sa.item_list = nullptr; sa.item_query = nullptr;
  sa.blk_off = blk_off; sa.packed = packed; sa.pos = pos; sa.part = ws->w_part.as<u64>();
  sa.cand_count = nullptr;
  sa.m = m; sa.K = K; sa.chunk_blocks = chunk_blocks; sa.nchunk = nchunk; sa.L = L;
  memcpy(&sa.sentinel_bits, &sentinel, 4);
  MergeArgs ma;
  ma.part = sa.part; ma.active = nullptr; ma.pos_to_id = ix->ids; ma.round_rows = nullptr; ma.cand_count = nullptr;
  ma.out_ids = d_out_ids; ma.out_dist = d_out_dist; ma.found = nullptr; ma.next_active = nullptr; ma.n_next = nullptr;
  ma.status = nullptr;
  ma.n_active = Q; ma.parts_per_query = nchunk; ma.L = L; ma.k = k; ma.found_rule = 0; ma.first_round = 1;
  ma.sentinel = sentinel;
  if (2 * k > 1024) return bigk_select_replay(ix, s, ws, sa, Q, ma, Q);   // (k > 512: bigk.h)
  if (int rc = launch_scan(ix, s, sa, Q)) return rc;
  return launch_merge(ix, s, ma);
}

extern "C" int freddy_gpu_pq_search_dev(freddy_gpu_index_t* ix, const float* d_queries, int32_t Q, int32_t k,
                                        float sentinel, int32_t* d_out_ids, float* d_out_dist, void* hip_stream) {
  if (int rc = check_search_args(ix, KIND_PQ, d_queries, Q, k, d_out_ids, d_out_dist)) return rc;
  if (Q == 0) return FREDDY_OK;
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ix->stream;
  if (pq_use_fused(ix, Q, k)) {
    if (int rc = pq_shadow_build(ix)) return rc;
    const int qf = pq_fused_queries_per_chunk(ix, ix->n_blocks);
    for (int q0 = 0; q0 < Q; q0 += qf) {
      const int n = std::min(qf, Q - q0);
      if (int rc = pq_fused_chunk(ix, ix->pq_shadow, s, d_queries + (size_t)q0 * ix->d, n, k, sentinel, d_out_ids + (size_t)q0 * k, d_out_dist + (size_t)q0 * k))
        return rc;
    }
    return FREDDY_OK;
  }
  const int qc = pq_queries_per_chunk(ix, ix->n_blocks, k);
  for (int q0 = 0; q0 < Q; q0 += qc) {
    const int n = std::min(qc, Q - q0);
    if (int rc = pq_chunk(ix, s, d_queries + (size_t)q0 * ix->d, n, k, sentinel, ix->blk_off, ix->packed, ix->pos,
                          ix->n_blocks, d_out_ids + (size_t)q0 * k, d_out_dist + (size_t)q0 * k))
      return rc;
  }
  return FREDDY_OK;
}

// "WHERE id IN (...)" over the flat PQ table: unknown ids vanish, duplicates collapse, order = table
// order; the rows' packed codes are gathered into a temporary one-list table (synchronises the stream).
static int pq_subset(freddy_gpu_index* ix, hipStream_t s, const int32_t* subset_ids, int64_t n_subset, const int32_t** blk_off,
                     const uint32_t** packed, const int32_t** pos, int64_t* n_blocks, int64_t* n_rows_out = nullptr) {
  Workspace* ws = workspace_for(ix, s);
  std::vector<int32_t> rows;
  rows.reserve((size_t)n_subset);
  for (int64_t i = 0; i < n_subset; ++i) {
    auto it = std::lower_bound(ix->h_ids.begin(), ix->h_ids.end(), subset_ids[i]);
    if (it != ix->h_ids.end() && *it == subset_ids[i]) rows.push_back((int32_t)(it - ix->h_ids.begin()));
  }
  std::sort(rows.begin(), rows.end());
  rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
  const int n_rows = (int)rows.size();
  const int nb = (n_rows + 63) / 64;
  const int n_pad = nb * 64;
  const int32_t h_blk[2] = {0, nb};
  if (ws->w_sub_rows.ensure(sizeof(int32_t) * std::max(n_rows, 1)) ||
      ws->w_sub_packed.ensure(sizeof(uint32_t) * (size_t)std::max(nb, 1) * ix->M2 * 64) ||
      ws->w_sub_pos.ensure(sizeof(int32_t) * (size_t)std::max(n_pad, 1)) || ws->w_sub_blk.ensure(sizeof(int32_t) * 2))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  if (n_rows) HIP_TRY(hipMemcpyAsync(ws->w_sub_rows.p, rows.data(), sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ws->w_sub_blk.p, h_blk, sizeof(h_blk), hipMemcpyHostToDevice, s));
  if (n_pad) {
    timed_launch(ix, s, "gather_rows", [&] {
      hipLaunchKernelGGL(gather_rows_kernel, dim3((n_pad + WG - 1) / WG), dim3(WG), 0, s, ws->w_sub_rows.as<int32_t>(),
                         n_rows, ix->packed, ws->w_sub_packed.as<uint32_t>(), ws->w_sub_pos.as<int32_t>(), ix->M2, n_pad);
    });
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(s));  // rows / h_blk are stack/heap temporaries
  *blk_off = ws->w_sub_blk.as<int32_t>();
  *packed = ws->w_sub_packed.as<uint32_t>();
  *pos = ws->w_sub_pos.as<int32_t>();
  *n_blocks = nb;
  if (n_rows_out) *n_rows_out = n_rows;
  return 0;
}

extern "C" int freddy_gpu_pq_search(freddy_gpu_index_t* ix, const float* queries, int32_t Q, int32_t k, float sentinel,
                                    const int32_t* subset_ids, int64_t n_subset, int32_t* out_ids, float* out_dist) {
  if (int rc = check_search_args(ix, KIND_PQ, queries, Q, k, out_ids, out_dist)) return rc;
  if (n_subset < 0 || (n_subset > 0 && !subset_ids)) return fail(FREDDY_E_ARG, "bad subset");
  if (Q == 0) return FREDDY_OK;
  HIP_TRY(hipSetDevice(ix->device));
  Workspace* ws = workspace_for(ix, ix->stream);
  hipStream_t s = ix->stream;
  // queries in, lists out through pinned staging that the kernels read and write themselves (mapped host memory): a
  // handful of queries are read where they are staged and their lists written straight back; larger batches cross PCIe
  // once, by a copy kernel each way.  No hipMemcpyAsync in the stream (each one is an SDMA hop with its own latency).
  const size_t q_bytes = sizeof(float) * (size_t)Q * ix->d, n_out = (size_t)Q * k;
  auto pinned_fit = [](void** p, size_t* cap, size_t need) -> int {
    if (need <= *cap) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    if (hipHostMalloc(p, need + need / 4 + 256, hipHostMallocDefault) != hipSuccess) { *p = nullptr; return -1; }
    *cap = need + need / 4 + 256;
    return 0;
  };
  if (pinned_fit(&ix->hio_in, &ix->hio_in_cap, q_bytes + 16) || pinned_fit(&ix->hio_out, &ix->hio_out_cap, n_out * 8 + 16))
    return fail(FREDDY_E_NOMEM, "pinned staging allocation failed");
  memcpy(ix->hio_in, queries, q_bytes);
  const bool direct = Q <= 8;
  if (!direct && (ws->w_q.ensure(q_bytes + 16) || ws->w_out_ids.ensure(sizeof(int32_t) * n_out) || ws->w_out_dist.ensure(sizeof(float) * n_out)))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  const float* d_q = static_cast<const float*>(ix->hio_in);
  int32_t* d_oi = static_cast<int32_t*>(ix->hio_out);
  float* d_od = reinterpret_cast<float*>(d_oi + n_out);
  if (!direct) {
    const size_t n16 = (q_bytes + 15) / 16;
    hipLaunchKernelGGL(lane_copy_in_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 512)), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(ix->hio_in), ws->w_q.as<uint4>(), n16);
    HIP_TRY(hipGetLastError());
    d_q = ws->w_q.as<float>(); d_oi = ws->w_out_ids.as<int32_t>(); d_od = ws->w_out_dist.as<float>();
  }

  const int32_t* blk_off = ix->blk_off;
  const uint32_t* packed = ix->packed;
  const int32_t* pos = ix->pos;
  int64_t n_blocks = ix->n_blocks, n_rows = ix->N;
  if (subset_ids)
    if (int rc = pq_subset(ix, s, subset_ids, n_subset, &blk_off, &packed, &pos, &n_blocks, &n_rows)) return rc;
  // (a subset of at least one full pseudo-list: its gathered rows get a view of their own, refreshed on this stream)
  const bool fused_path = pq_use_fused(ix, Q, k) && (!subset_ids || n_blocks >= FUSED_UNIT_BLOCKS);
  freddy_gpu_index* view = nullptr;
  if (fused_path && !subset_ids) { if (int rc = pq_shadow_build(ix)) return rc; view = ix->pq_shadow; }
  if (fused_path && subset_ids) {
    if (int rc = pq_view_refresh(ix, &ix->pq_sub_view, s, packed, pos, n_blocks, n_rows)) return rc;
    view = ix->pq_sub_view;
  }
  const int qc = fused_path ? pq_fused_queries_per_chunk(ix, n_blocks) : pq_queries_per_chunk(ix, n_blocks, k);
  if (!fused_path && direct && pq_one_shape(ix, Q, k, n_blocks)) {
    int32_t* err = reinterpret_cast<int32_t*>(static_cast<char*>(ix->hio_out) + n_out * 8);   // (the staging block's spare 16 bytes)
    *err = 0;
    if (int rc = pq_one(ix, s, queries, k, sentinel, blk_off, packed, pos, n_blocks, d_oi, d_od, err)) return rc;
    // the kernel's last store is this word (2 = list written, 1 = a bounded poll ran out): polled here for up to a millisecond
    // -- a few microseconds sooner than the runtime's completion signal -- then the stream is waited for the usual way
    {
      volatile int32_t* flag = err;
      const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(1000);
      int spins = 0;
      while (*flag == 0) {
        __builtin_ia32_pause();
        if ((++spins & 255) == 0 && std::chrono::steady_clock::now() > t_end) break;
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (*flag != 2) HIP_TRY(hipStreamSynchronize(s));
    }
    if (getenv("FREDDY_GPU_ONE_PROF")) {
      HIP_TRY(hipStreamSynchronize(s));
      unsigned long long st[16];
      (void)hipMemcpy(st, ws->w_one.as<unsigned long long>() + 8, sizeof(st), hipMemcpyDeviceToHost);
      fprintf(stderr, "[pq_one] wg0: slice %.2f barrier %.2f stage %.2f scan %.2f publish %.2f | last: since wg0 start %.2f merge %.2f replay %.2f us\n",
              (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01,
              (st[8] - st[0]) * 0.01, (st[9] - st[8]) * 0.01, (st[10] - st[9]) * 0.01);
    }
    if (*err == 2) {
      const int32_t* h_pos = static_cast<const int32_t*>(ix->hio_out);
      for (size_t i = 0; i < n_out; ++i) out_ids[i] = h_pos[i] >= 0 ? ix->h_ids[(size_t)h_pos[i]] : -1;
      memcpy(out_dist, h_pos + n_out, n_out * 4);
      return FREDDY_OK;
    }
    // the grid never met at its barrier (not co-resident): counters re-armed, this handle keeps to the three-launch path
    ix->one_launch_failed = true;
    ws->one_shape = 0;
  }
  for (int q0 = 0; q0 < Q; q0 += qc) {
    const int n = std::min(qc, Q - q0);
    if (fused_path) {
      if (int rc = pq_fused_chunk(ix, view, s, d_q + (size_t)q0 * ix->d, n, k, sentinel, d_oi + (size_t)q0 * k, d_od + (size_t)q0 * k))
        return rc;
      continue;
    }
    if (int rc = pq_chunk(ix, s, d_q + (size_t)q0 * ix->d, n, k, sentinel, blk_off, packed, pos, n_blocks, d_oi + (size_t)q0 * k,
                          d_od + (size_t)q0 * k))
      return rc;
  }
  if (!direct) {
    hipLaunchKernelGGL(host_io_out_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, d_oi, d_od, static_cast<int32_t*>(ix->hio_out), (int)n_out);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(out_ids, ix->hio_out, n_out * 4);
  memcpy(out_dist, static_cast<const int32_t*>(ix->hio_out) + n_out, n_out * 4);
  return FREDDY_OK;
}

// ---------------------------------------------------------------------------------------
// grouping_pq
// ---------------------------------------------------------------------------------------
extern "C" int freddy_gpu_grouping_pq(freddy_gpu_index_t* ix, const float* group_vectors, int32_t G, const int32_t* subset_ids,
                                      int64_t n_subset, int32_t* out_ids, int32_t* out_group, int64_t* n_out) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (ix->kind != KIND_PQ) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  if (G <= 0 || !group_vectors || !out_ids || !out_group || !n_out) return fail(FREDDY_E_ARG, "bad argument");
  if (n_subset < 0 || (n_subset > 0 && !subset_ids)) return fail(FREDDY_E_ARG, "bad subset");
  *n_out = 0;
  HIP_TRY(hipSetDevice(ix->device));
  Workspace* ws = workspace_for(ix, ix->stream);
  hipStream_t s = ix->stream;
  const int m = ix->m, K = ix->K, d = ix->d;
  const size_t lutN = (size_t)m * K;
  if (lutN * sizeof(float) > 160 * 1024) return fail(FREDDY_E_LIMIT, "m*K=%zu LUT entries exceed the 160 KiB of LDS", lutN);
  const int32_t* blk_off = ix->blk_off;
  const uint32_t* packed = ix->packed;
  const int32_t* pos = ix->pos;
  int64_t n_blocks = ix->n_blocks;
  if (subset_ids)
    if (int rc = pq_subset(ix, s, subset_ids, n_subset, &blk_off, &packed, &pos, &n_blocks)) return rc;
  (void)blk_off;
  if (n_blocks == 0) return FREDDY_OK;
  if (ws->w_q.ensure(sizeof(float) * (size_t)G * d) || ws->w_lut.ensure(sizeof(float) * (size_t)G * lutN) ||
      ws->w_out_ids.ensure(sizeof(int32_t) * (size_t)n_blocks * 64))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  HIP_TRY(hipMemcpyAsync(ws->w_q.p, group_vectors, sizeof(float) * (size_t)G * d, hipMemcpyHostToDevice, s));
  if (int rc = launch_lut(ix, s, ws->w_q.as<float>(), nullptr, ws->w_lut.as<float>(), G)) return rc;   // freddy.c:1288-1299
  const dim3 grid((unsigned)((n_blocks + GROUP_BLOCKS - 1) / GROUP_BLOCKS));
  timed_launch(ix, s, "grouping", [&] {
    if (ix->M2 == 6)
      hipLaunchKernelGGL((grouping_kernel<6>), grid, dim3(WG), lutN * sizeof(float), s, ws->w_lut.as<float>(), G, m, K, packed, (int)n_blocks, ws->w_out_ids.as<int32_t>());
    else if (ix->M2 == 15)
      hipLaunchKernelGGL((grouping_kernel<15>), grid, dim3(WG), lutN * sizeof(float), s, ws->w_lut.as<float>(), G, m, K, packed, (int)n_blocks, ws->w_out_ids.as<int32_t>());
    else
      hipLaunchKernelGGL((grouping_kernel<0>), grid, dim3(WG), lutN * sizeof(float), s, ws->w_lut.as<float>(), G, m, K, packed, (int)n_blocks, ws->w_out_ids.as<int32_t>());
  });
  HIP_TRY(hipGetLastError());
  std::vector<int32_t> h_grp((size_t)n_blocks * 64), h_pos((size_t)n_blocks * 64);
  HIP_TRY(hipMemcpyAsync(h_grp.data(), ws->w_out_ids.p, sizeof(int32_t) * h_grp.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_pos.data(), pos, sizeof(int32_t) * h_pos.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  int64_t n = 0;
  for (size_t i = 0; i < h_pos.size(); ++i)
    if (h_pos[i] >= 0) { out_ids[n] = ix->h_ids[(size_t)h_pos[i]]; out_group[n] = h_grp[i]; ++n; }
  *n_out = n;
  return FREDDY_OK;
}

// The kernels of this unit that want more than the default 64 KiB of dynamic LDS (a per-device function attribute).
int raise_lds_limits_pq(int device) {
  static std::mutex mu;
  static std::vector<char> done;
  std::lock_guard<std::mutex> g(mu);
  if ((size_t)device < done.size() && done[(size_t)device]) return 0;
  const void* kernels[] = {
      (const void*)&grouping_kernel<6>, (const void*)&grouping_kernel<15>, (const void*)&grouping_kernel<0>};
  for (const void* k : kernels)
    HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  if (done.size() <= (size_t)device) done.resize((size_t)device + 1, 0);
  done[(size_t)device] = 1;
  return 0;
}
