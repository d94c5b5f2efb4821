// ivfadc.hip -- the IVFADC search (freddy.c:174-393, :679-999): cell selection, work table, scans, merge; the *_dev entry
// point, the host-buffer pipeline and the one-query launch.  Host-side responsibilities only: size workspaces, order the
// launches of a probing round on one HIP stream, run the (rare) extra rounds of the reference's "while (foundInstances < k)"
// loop (freddy.c:262, :835).  No arithmetic that influences a result happens on the host.
#include "internal.h"

#include <functional>

#include "kernels.h"
#include "scan_common.h"
#include "fused3.h"
#include "fused5.h"
#include "fused8.h"
#include "one.h"
#include "sparse5.h"
#include "coarse.h"
#include "multi.h"
#include "io_kernels.h"
#include "bigk.h"

// ---------------------------------------------------------------------------------------
// kernel dispatch helpers
// ---------------------------------------------------------------------------------------
int pick_V(int L) {
  if (L <= 64) return 1;
  if (L <= 128) return 2;
  if (L <= 256) return 4;
  if (L <= 512) return 8;
  if (L <= 1024) return 16;
  return 0;
}

template <int M, int V>
static int launch_scan_mv(freddy_gpu_index* ix, hipStream_t s, const ScanArgs& a, dim3 grid, size_t lds) {
  timed_launch(ix, s, "adc_scan", [&] { hipLaunchKernelGGL((adc_scan_kernel<M, V>), grid, dim3(SCAN_WG), lds, s, a); });
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int M>
static int launch_scan_m(freddy_gpu_index* ix, hipStream_t s, const ScanArgs& a, dim3 grid, size_t lds, int V) {
  switch (V) {
    case 1: return launch_scan_mv<M, 1>(ix, s, a, grid, lds);
    case 2: return launch_scan_mv<M, 2>(ix, s, a, grid, lds);
    case 4: return launch_scan_mv<M, 4>(ix, s, a, grid, lds);
    case 8: return launch_scan_mv<M, 8>(ix, s, a, grid, lds);
    case 16: return launch_scan_mv<M, 16>(ix, s, a, grid, lds);
  }
  return fail(FREDDY_E_LIMIT, "unsupported selection width");
}

int launch_scan(freddy_gpu_index* ix, hipStream_t s, const ScanArgs& a, int n_items) {
  if (n_items <= 0 || a.nchunk <= 0) return 0;
  const int Vl = pick_V(a.L);
  const size_t lds = std::max((((size_t)a.m * a.K * 4 + 15) & ~(size_t)15) + (size_t)SCAN_WAVES * 64 * sizeof(u64),
                              (size_t)SCAN_WAVES * 64 * Vl * sizeof(u64));
  dim3 grid((unsigned)a.nchunk, (unsigned)n_items);
  const int V = pick_V(a.L);
  if (a.floor) {   // a later selection pass of a list of more than 512 entries (bigk.h)
    if (V != 16) return fail(FREDDY_E_ARG, "selection passes are 1024 keys wide");
    timed_launch(ix, s, "adc_scan", [&] {
      if (a.m == 12) hipLaunchKernelGGL((adc_scan_kernel<12, 16, true>), grid, dim3(SCAN_WG), lds, s, a);
      else hipLaunchKernelGGL((adc_scan_kernel<0, 16, true>), grid, dim3(SCAN_WG), lds, s, a);
    });
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (a.m == 12) return launch_scan_m<12>(ix, s, a, grid, lds, V);
  return launch_scan_m<0>(ix, s, a, grid, lds, V);
}

int launch_merge(freddy_gpu_index* ix, hipStream_t s, const MergeArgs& a) {
  if (a.n_active <= 0) return 0;
  const int V = pick_V(a.L);
  dim3 grid((unsigned)a.n_active), block(64);
  timed_launch(ix, s, "merge_replay", [&] {
    switch (V) {
      case 1: hipLaunchKernelGGL((merge_replay_kernel<1>), grid, block, 0, s, a); break;
      case 2: hipLaunchKernelGGL((merge_replay_kernel<2>), grid, block, 0, s, a); break;
      case 4: hipLaunchKernelGGL((merge_replay_kernel<4>), grid, block, 0, s, a); break;
      case 8: hipLaunchKernelGGL((merge_replay_kernel<8>), grid, block, 0, s, a); break;
      case 16: hipLaunchKernelGGL((merge_replay_kernel<16>), grid, block, 0, s, a); break;
    }
  });
  HIP_TRY(hipGetLastError());
  return 0;
}

// k > 512 (bigk.h): the 2k smallest keys of every active query, 1024 per pass over the same rows, then the list in closed form.
// sa: the generic scan's arguments with L = 1024; ma: what merge_replay_kernel would get (its L is not used); Q: queries of the chunk.
int bigk_select_replay(freddy_gpu_index* ix, hipStream_t s, Workspace* ws, ScanArgs sa, int n_items, const MergeArgs& ma, int Q) {
  if (ma.n_active <= 0) return 0;
  const int k = ma.k;
  if (k > BIGK_KMAX) return fail(FREDDY_E_LIMIT, "k=%d exceeds this build's limit of %d", k, BIGK_KMAX);
  if (sa.L != BIGK_PASS) return fail(FREDDY_E_ARG, "selection passes are %d keys wide (got L=%d)", BIGK_PASS, sa.L);
  const int passes = (2 * k + BIGK_PASS - 1) / BIGK_PASS;
  const int nsel = passes * BIGK_PASS;
  int npad = 2048;
  while (npad < k + nsel) npad *= 2;
  if (ws->w_bigsel.ensure(sizeof(u64) * (size_t)ma.n_active * nsel) || ws->w_floor.ensure(sizeof(u64) * (size_t)std::max(Q, 1)))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed (k=%d)", k);
  SelectArgs se;
  se.part = sa.part; se.active = ma.active; se.sel = ws->w_bigsel.as<u64>(); se.floor = ws->w_floor.as<u64>();
  se.parts_per_query = ma.parts_per_query; se.nsel = nsel;
  int32_t* const cand_count = sa.cand_count;
  for (int p = 0; p < passes; ++p) {
    sa.floor = p ? ws->w_floor.as<u64>() : nullptr;
    sa.cand_count = p ? nullptr : cand_count;   // (the accepted-rows rule counts every row once)
    if (int rc = launch_scan(ix, s, sa, n_items)) return rc;
    se.pass = p;
    timed_launch(ix, s, "merge_select", [&] { hipLaunchKernelGGL(merge_select_kernel, dim3(ma.n_active), dim3(64), 0, s, se); });
    HIP_TRY(hipGetLastError());
  }
  BigkArgs b;
  b.sel = se.sel; b.active = ma.active; b.pos_to_id = ma.pos_to_id; b.round_rows = ma.round_rows; b.cand_count = ma.cand_count;
  b.out_ids = ma.out_ids; b.out_dist = ma.out_dist; b.found = ma.found; b.next_active = ma.next_active; b.n_next = ma.n_next;
  b.status = ma.status; b.n_active = ma.n_active; b.nsel = nsel; b.npad = npad; b.k = k; b.found_rule = ma.found_rule;
  b.first_round = ma.first_round; b.sentinel = ma.sentinel;
  timed_launch(ix, s, "bigk_replay", [&] {
    hipLaunchKernelGGL(bigk_replay_kernel, dim3(ma.n_active), dim3(BIGK_T), bigk_lds_bytes(npad, k), s, b);
  });
  HIP_TRY(hipGetLastError());
  return 0;
}

// coarse != NULL: `vecs` are the queries and the kernel forms the residual q - coarse[cell] itself (no residual_kernel launch)
int launch_lut(freddy_gpu_index* ix, hipStream_t s, const float* vecs, const int32_t* item_cell,
                      float* lut, int n_items, const float* coarse, const int32_t* item_query) {
  if (n_items <= 0) return 0;
  // enough workgroups to fill 256 CUs several times over, while amortising the register
  // fill of the codebook slice over as many items as possible
  int ipw = (int)std::max<int64_t>(1, ((int64_t)ix->m * n_items + 4095) / 4096);
  ipw = std::min(ipw, 64);
  dim3 grid((unsigned)ix->m, (unsigned)((n_items + ipw - 1) / ipw));
  const int m = ix->m, K = ix->K, d = ix->d, S = ix->S;
  timed_launch(ix, s, "lut_build", [&] {
    if (S == 25) hipLaunchKernelGGL((lut_build_kernel<25, 4>), grid, dim3(WG), 0, s, vecs, item_cell, ix->cbT, lut, n_items, ipw, m, K, d, coarse, item_query);
    else if (S == 10) hipLaunchKernelGGL((lut_build_kernel<10, 4>), grid, dim3(WG), 0, s, vecs, item_cell, ix->cbT, lut, n_items, ipw, m, K, d, coarse, item_query);
    else if (S == 20) hipLaunchKernelGGL((lut_build_kernel<20, 4>), grid, dim3(WG), 0, s, vecs, item_cell, ix->cbT, lut, n_items, ipw, m, K, d, coarse, item_query);
    else hipLaunchKernelGGL(lut_build_generic_kernel, grid, dim3(WG), 0, s, vecs, item_cell, ix->cbT, lut, n_items, ipw, m, K, d, S, coarse, item_query);
  });
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// IVFADC
// ---------------------------------------------------------------------------------------

// coarse distances (a6/a7) of every query of the chunk, and -- for the filter + refine scan -- the
// per-batch query x codebook table beside them on the side stream
// (q_lo, q_n): the combined MFMA cell-selection + table launch for the queries [q_lo, q_lo + q_n) of the chunk only -- the host-buffer
// pipeline launches it piece by piece behind the pieces of the staged queries (q_lo a multiple of 32); q_n < 0: the whole chunk
// K <= 256 with one-byte codes: the scan that keeps the whole entry's slab in LDS (fused8.h) reads a compact copy of the query table,
// written by the table kernel behind the general one (w_qc: [Q][m][512] + [Q][m][128] dwords)
static bool scan_whole_slab(const freddy_gpu_index* ix) { return ix->packed8 && ix->tune.codes_u8 == 1 && ix->K <= 256 && ix->m == 12; }
static bool ivf_coarse_by_pieces(const IvfRun& r) { return r.approx && r.fused && r.scan_kernel == 5; }
static int ivf_coarse(IvfRun& r, int q_lo = 0, int q_n = -1) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int Q = r.Q, d = ix->d, C = ix->C, m = ix->m, K = ix->K, Cpad = ix->Cpad;
  if (q_n < 0) { q_lo = 0; q_n = Q; }
  if ((q_lo != 0 || q_n != Q) && !ivf_coarse_by_pieces(r)) return fail(FREDDY_E_ARG, "internal: this path launches its coarse kernel once");
  const int used_words = (C + 31) / 32;
  const size_t items = (size_t)Q * r.W;
  // round-one scratch that must start at zero: the probe bitmaps, the counters (n_next, n_groups, work
  // counter), the per-cell item counts and the accepted-candidate counts: every coarse kernel clears them itself
  // (ZeroArgs) -- except the small-batch kernel for vectors of more than 1024 dimensions, which gets memsets.
  const bool small_zero = !r.tiled && d <= 1024;
  if (!r.tiled && !small_zero) {
    HIP_TRY(hipMemsetAsync(ws->w_used.p, 0, sizeof(uint32_t) * (size_t)Q * used_words, s));
    HIP_TRY(hipMemsetAsync(ws->w_cnt.p, 0, sizeof(int32_t) * 8, s));
  }
  ZeroArgs za;
  za.p[0] = ws->w_used.as<uint32_t>(); za.n[0] = Q * used_words;
  za.p[1] = ws->w_cnt.as<uint32_t>(); za.n[1] = 8;
  za.p[2] = r.fused ? ws->w_cellcnt.as<uint32_t>() : nullptr; za.n[2] = r.fused ? C * 2 : 0;
  za.p[3] = ws->w_cand.as<uint32_t>(); za.n[3] = 2 * Q;   // accepted-row counts, then the queries' running bounds (FilterArgs::tau_run)
  // survivor counts: regions of chunks a list does not have, or of items without a cell, stay at zero
  za.p[4] = r.fused ? ws->w_surv_cnt.as<uint32_t>() : nullptr; za.n[4] = r.fused ? (int)(items * r.upi * FUSED_NW) : 0;

  // more than 1024 cells: the (query, 128-cell tile) minima for the plan's two-level selection (round one: no cell is used yet)
  float* tile_min = nullptr;
  if (r.approx && Cpad > COARSE_MAX_CPAD) {
    if (ws->w_tmin.ensure(sizeof(float) * (size_t)Q * (Cpad / 128))) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
    tile_min = ws->w_tmin.as<float>();
  }
  auto launch_coarse = [&]() -> int {
    timed_launch(ix, s, "coarse_dist", [&] {
      if (r.approx && ix->coarseH)
        hipLaunchKernelGGL(coarse_approx16_kernel, dim3(Cpad / 128, (Q + COARSE_TQ - 1) / COARSE_TQ), dim3(256), coarse_approx16_lds(d), s, r.d_q,
                           (const ch8v*)ix->coarseH, ix->coarse_ec, ix->cn2, ws->w_distT.as<float>(), ws->w_qn2.as<float>(), Q, Cpad, d, za, tile_min, C);
      else if (r.approx)
        hipLaunchKernelGGL(coarse_approx_kernel, dim3(Cpad / 128, (Q + COARSE_TQ - 1) / COARSE_TQ), dim3(256),
                           (size_t)(COARSE_TQ * (ix->dp + 4) + 128) * sizeof(float), s, r.d_q, ix->coarseP, ix->cn2,
                           ws->w_distT.as<float>(), ws->w_qn2.as<float>(), Q, Cpad, d, ix->dp, za, tile_min, C);
      else if (r.tiled)
        hipLaunchKernelGGL((coarse_tile_kernel<2, 16>), dim3(Cpad / 32, (Q + 63) / 64), dim3(256), 0, s, r.d_q, ix->coarseT,
                           ws->w_distT.as<float>(), Q, Cpad, d, za);
      else if (small_zero)
        hipLaunchKernelGGL((coarse_small_kernel<50>), dim3(Cpad / 64, Q), dim3(64), 0, s, r.d_q, ix->coarseT, ws->w_distT.as<float>(), Q, Cpad, d, za);
      else
        hipLaunchKernelGGL((coarse_dist_kernel<16>), dim3(Cpad / WG, (Q + 15) / 16), dim3(WG), (size_t)d * 16 * sizeof(float), s, r.d_q,
                           ix->coarseT, ws->w_distT.as<float>(), Q, Cpad, d);
    });
    HIP_TRY(hipGetLastError());
    return 0;
  };
  // The query x codebook table is independent of the coarse distances: with the MFMA cell selection the coarse tiles and
  // the table units are the workgroups of ONE launch (fused5.h coarse_table5_kernel); otherwise the table kernel runs in
  // line before the coarse kernel.
  if (r.approx && r.fused && r.scan_kernel == 5) {
    CoarseTableArgs ct;   // (every array is query-major: a piece is the same launch on offset pointers)
    ct.queries = r.d_q + (size_t)q_lo * d; ct.coarseF = ix->coarseP; ct.cn2 = ix->cn2; ct.dist = ws->w_distT.as<float>() + (size_t)q_lo * Cpad;
    ct.qn2 = ws->w_qn2.as<float>() + q_lo;
    ct.Q = q_n; ct.Cpad = Cpad; ct.d = d; ct.dp = ix->dp; ct.z = za; ct.coarse_gx = Cpad / 128; ct.coarse_gy = (q_n + COARSE_TQ - 1) / COARSE_TQ;
    ct.cbT = ix->cbF; ct.cmax = ix->cmaxp; ct.qn = ws->w_qn.as<float>() + (size_t)q_lo * m; ct.qscale = ws->w_qn.as<float>() + (size_t)Q * m + q_lo;
    ct.qc = ws->w_qc.as<uint32_t>() + (size_t)q_lo * m * 512; ct.m = m; ct.K = K; ct.tmin = tile_min ? tile_min + (size_t)q_lo * (Cpad / 128) : nullptr; ct.C = C;
    ct.coarseH = (const ch8v*)ix->coarseH; ct.ec = ix->coarse_ec;
    ct.qc8 = scan_whole_slab(ix) ? ws->w_qc.as<uint32_t>() + (size_t)Q * m * 512 + (size_t)q_lo * m * 128 : nullptr;
    const size_t lds = std::max<size_t>(ix->coarseH ? coarse_approx16_lds(d) : (size_t)(COARSE_TQ * (ix->dp + 4) + 128) * sizeof(float), (size_t)query_codebook5_lds<25, 16>());
    const unsigned grid = (unsigned)(ct.coarse_gx * ct.coarse_gy + m * ((q_n + 15) / 16));
    timed_launch(ix, s, "coarse_table", [&] {
      if (ix->coarseH) hipLaunchKernelGGL((coarse_table5_kernel<25, 16, true>), dim3(grid), dim3(256), lds, s, ct);
      else hipLaunchKernelGGL((coarse_table5_kernel<25, 16>), dim3(grid), dim3(256), lds, s, ct);
    });
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (r.fused && r.scan_kernel == 5) {
    timed_launch(ix, s, "query_codebook", [&] {
      hipLaunchKernelGGL((query_codebook5_kernel<25, 16>), dim3(m, (Q + 15) / 16), dim3(256), 0, s, r.d_q, ix->cbF, ix->cmaxp,
                         ws->w_qn.as<float>(), ws->w_qn.as<float>() + (size_t)Q * m, ws->w_qc.as<uint32_t>(), Q, d, m, K,
                         scan_whole_slab(ix) ? ws->w_qc.as<uint32_t>() + (size_t)Q * m * 512 : nullptr);
    });
    HIP_TRY(hipGetLastError());
  }
  return launch_coarse();
}

// a7: the W nearest not-yet-used cells of every active query (+ their items appended to the cells' buckets)
static int ivf_plan(IvfRun& r, PlanArgs& pa) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int C = ix->C, W = r.W;
  pa.dist = ws->w_distT.as<float>(); pa.active = r.active; pa.list_off = ix->list_off;
  pa.used = ws->w_used.as<uint32_t>();
  pa.item_cell = ws->w_item_cell.as<int32_t>(); pa.item_query = ws->w_item_query.as<int32_t>();
  pa.item_dist = ws->w_item_dist.as<float>();
  pa.round_rows = ws->w_rows.as<int32_t>();
  pa.n_active = r.n_active; pa.Cpad = ix->Cpad; pa.C = C; pa.W = W; pa.used_words = (C + 31) / 32;
  pa.cell_count = r.fused ? ws->w_cellcnt.as<int32_t>() : nullptr;
  pa.cell_items = r.fused ? ws->w_sorted.as<int32_t>() : nullptr; pa.cell_cap = r.n_active;
  pa.cell_limit = r.cell_limit;
  const int n_items = r.n_active * W;
  if (r.fused && !(r.zeroed && r.first())) {
    HIP_TRY(hipMemsetAsync(ws->w_cellcnt.p, 0, sizeof(int32_t) * (size_t)C * 2, s));   // counts + fill cursors
    HIP_TRY(hipMemsetAsync(ws->w_surv_cnt.p, 0, sizeof(int32_t) * (size_t)n_items * r.upi * FUSED_NW, s));
  }
  const int PV = pick_V(2 * W);
  const size_t plan_lds = (size_t)(64 + 64 * PV) * sizeof(u64) + (size_t)W * 8;
  if (r.approx) {
    Plan2Args g;
    g.p = pa; g.queries = r.d_q; g.coarse = ix->coarse; g.qn2 = ws->w_qn2.as<float>(); g.item_dist = pa.item_dist;
    g.violations = ix->viol; g.cmax = ix->cmax; g.d = ix->d; g.refine_all = (ix->tune.check_brackets & 2) ? 1 : 0; g.prof = nullptr;
    g.tmin = (r.first() && ix->Cpad > COARSE_MAX_CPAD && ws->w_tmin.p) ? ws->w_tmin.as<float>() : nullptr;
    timed_launch(ix, s, "probe_plan", [&] {
      // (one batch at a time: four waves per query, the shortest latency; batches in flight: one wave per query, the smallest footprint)
      if (ix->Cpad <= COARSE_MAX_CPAD && r.share > 1) hipLaunchKernelGGL((probe_plan2_kernel<0, false, 1>), dim3(r.n_active), dim3(64), 0, s, g);
      else if (ix->Cpad <= COARSE_MAX_CPAD) hipLaunchKernelGGL((probe_plan2_kernel<0, false>), dim3(r.n_active), dim3(64 * PLAN2_NW), 0, s, g);
      else hipLaunchKernelGGL((probe_plan2_kernel<0, true>), dim3(r.n_active), dim3(64 * PLAN2_NW), 0, s, g);   // (more than 1024 cells: streamed)
    });
  } else
  timed_launch(ix, s, "probe_plan", [&] {
    switch (PV) {
      case 1: hipLaunchKernelGGL((probe_plan_kernel<1>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
      case 2: hipLaunchKernelGGL((probe_plan_kernel<2>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
      case 4: hipLaunchKernelGGL((probe_plan_kernel<4>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
      case 8: hipLaunchKernelGGL((probe_plan_kernel<8>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
      default: hipLaunchKernelGGL((probe_plan_kernel<16>), dim3(r.n_active), dim3(64), plan_lds, s, pa); break;
    }
  });
  HIP_TRY(hipGetLastError());
  if (!(r.zeroed && r.first())) HIP_TRY(hipMemsetAsync(ws->w_cand.p, 0, sizeof(int32_t) * 2 * r.Q, s));   // (every round starts without bounds)
  return 0;
}

int ivf_work_table(IvfRun& r, WorkTable& wt) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int n_items = r.n_active * r.W;
  wt.max_groups = ((size_t)n_items / MULTI_G + (size_t)ix->C + 1) * r.upi;   // (group, chunk) work entries (the smallest group size: a bound for every scan)
  int32_t* base = ws->w_groups.as<int32_t>();
  wt.group_cell = base; wt.group_first = base + wt.max_groups; wt.group_cnt = base + 2 * wt.max_groups;
  wt.n_groups = ws->w_cnt.as<int32_t>() + 1;
  wt.work_counter = ws->w_cnt.as<int32_t>() + 2;
  wt.sp_counter = ws->w_cnt.as<int32_t>() + 3;
  wt.n_sparse = ws->w_cnt.as<int32_t>() + 4;
  // cells that one or two queries probe are scanned item by item -- where such cells are the rule (fewer than four items per
  // cell on average: a corpus with more cells than the batch has probes) and there are enough of them to fill the chip's
  // workgroup slots several times (the item-wise scan is built for throughput: a 256-query batch on 1000 cells took 0.187
  // instead of 0.155 ms with it); a dense batch does not pay the extra launch for its handful of thin cells
  // (a negative option value forces the item-wise scan for cells of up to that many items whatever the batch: tests)
  const int sparse_max = r.scan_kernel != 5 ? 0
                         : ix->tune.sparse_items < 0 ? -ix->tune.sparse_items
                         : ((size_t)n_items < 4 * (size_t)ix->C && n_items >= 16 * ix->n_cus) ? ix->tune.sparse_items : 0;
  wt.sp_cap = sparse_max > 0 ? (size_t)n_items * r.upi : 0;
  wt.sp_cell = base + 3 * wt.max_groups; wt.sp_first = wt.sp_cell + wt.sp_cap; wt.sp_chunk = wt.sp_first + wt.sp_cap;
  timed_launch(ix, s, "work_table", [&] {
    hipLaunchKernelGGL(work_table_kernel, dim3(1), dim3(1024), 0, s, ws->w_cellcnt.as<int32_t>(), ix->C, r.n_active, r.scan_kernel == 5 ? SCAN5_G : r.scan_kernel == 2 ? MULTI_G : SPEC2_G, ix->blk_off,
                       wt.group_cell, wt.group_first, wt.group_cnt, wt.n_groups, r.scan_kernel == 5 ? 2 : 0,
                       sparse_max, wt.sp_cell, wt.sp_first, wt.sp_chunk, wt.n_sparse, sparse_max >= 2 ? 1 : 0);
  });
  wt.sp_pairs = sparse_max >= 2;
  HIP_TRY(hipGetLastError());
  if (!(r.zeroed && r.first())) HIP_TRY(hipMemsetAsync(wt.work_counter, 0, 2 * sizeof(int32_t), s));   // (both work counters)
  return 0;
}

static int scan_prof_buffer(freddy_gpu_index* ix, Workspace* ws, long long** prof) {
  *prof = nullptr;
#ifndef FREDDY_LAB
  (void)ix; (void)ws;
  return 0;
#else
  if (!ix->tune.scan_prof) return 0;
  if (ws->w_prof.ensure(sizeof(long long) * 8 * 1024)) return fail(FREDDY_E_NOMEM, "profile buffer");
  *prof = ws->w_prof.as<long long>();
  return 0;
#endif
}

#ifdef FREDDY_LAB
// debugging aid (option fused_prof): per-phase shader-clock sums of every persistent workgroup's builder wave 0
static int scan_prof_print(freddy_gpu_index* ix, hipStream_t s, const long long* prof, unsigned n_persist) {
  std::vector<long long> h(8 * (size_t)n_persist);
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMemcpy(h.data(), prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
  double sum[8] = {0}; long long mx_end = 0, mn_end = -1; double ent = 0;
  for (unsigned b = 0; b < n_persist; ++b) {
    for (int i = 0; i < 6; ++i) sum[i] += (double)h[b * 8 + i];
    ent += (double)h[b * 8 + 7];
    mx_end = std::max(mx_end, h[b * 8 + 6]);
    mn_end = mn_end < 0 ? h[b * 8 + 6] : std::min(mn_end, h[b * 8 + 6]);
  }
  fprintf(stderr, "[scan prof] wgs=%u entries=%.0f  builder cycles/entry: builds=%.0f barrier-wait=%.0f tail=%.0f | gatherer wave 0 (fused5.h): main=%.0f colmin=%.0f S2=%.0f | workgroups ran dry over %.1f us\n",
          n_persist, ent, sum[0] / ent, sum[1] / ent, sum[3] / ent, sum[2] / ent, sum[4] / ent, sum[5] / ent, (mx_end - mn_end) / 100.0);
  (void)ix;
  return 0;
}
#endif

// Default scan: filter + refine.  entry records -> ivf_filter5_kernel (+ the item-wise scan of thin cells) -> merge_refine_kernel.
int ivf_scan_filter(IvfRun& r, const PlanArgs& pa, const WorkTable& wt) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int Q = r.Q, m = ix->m, K = ix->K;
  if (!r.records_ready && ws->w_records.ensure(sizeof(int32_t) * REC_DW * wt.max_groups)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  RecordArgs ra;
  ra.group_cell = wt.group_cell; ra.group_first = wt.group_first; ra.group_cnt = wt.group_cnt; ra.n_groups = wt.n_groups;
  ra.sorted_item = ws->w_sorted.as<int32_t>(); ra.item_query = pa.item_query; ra.blk_off = ix->blk_off; ra.list_off = ix->list_off;
  ra.item_dist = pa.item_dist; ra.qn = ws->w_qn.as<float>(); ra.qscale = ws->w_qn.as<float>() + (size_t)Q * m; ra.pmax = ix->pmax;
  ra.records = ws->w_records.as<int32_t>(); ra.sentinel = r.sentinel;
  if (!r.records_ready) {
    timed_launch(ix, s, "entry_records", [&] {
      hipLaunchKernelGGL((entry_record5_kernel<12>), dim3((unsigned)((wt.max_groups + 3) / 4)), dim3(256), 0, s, ra);
    });
    HIP_TRY(hipGetLastError());
  }
  FilterArgs fl;
  fl.qc = ws->w_qc.as<uint32_t>(); fl.qc8 = ws->w_qc.as<uint32_t>() + (size_t)Q * m * 512; fl.rterm = ix->rterm; fl.records = ws->w_records.as<int32_t>(); fl.n_groups = wt.n_groups;
  fl.work_counter = wt.work_counter; fl.packed = ix->packed; fl.surv = ws->w_surv.as<u64>(); fl.surv_count = ws->w_surv_cnt.as<int32_t>();
  fl.cand_count = (r.found_rule == 1) ? ws->w_cand.as<int32_t>() : nullptr;
  fl.K = K; fl.L = r.L; fl.upi = r.upi; fl.sentinel = r.sentinel; fl.keep_all = (ix->tune.check_brackets & 1) ? 1 : 0; fl.fence = 0;
#ifdef FREDDY_LAB
  fl.fence = (uint32_t)ix->tune.scan_fence;
#endif
  // (not for a batch over the flat PQ table: a few dozen queries x a thousand entries read and update the same two cache lines --
  // 96 -> 128 us -- and its merge gains nothing; an IVFADC batch: +2.6 % queries/s with four batches in flight)
  fl.tau_run = (ix->tune.running_bound && !r.records_ready) ? ws->w_cand.as<uint32_t>() + Q : nullptr;
  if (int rc = scan_prof_buffer(ix, ws, &fl.prof)) return rc;
  // K <= 256: one byte per code (packed8); the profiling instantiation stays with the int16 layout
  const bool u8 = ix->packed8 && ix->tune.codes_u8 != 0 && K <= 256;
  // ... and by default the kernel that keeps the WHOLE entry's slab in LDS (fused8.h; option codes_u8 = 2: fused5.h's one-byte instantiation)
  const bool whole = u8 && scan_whole_slab(ix);
  // LDS: slabs [2 buffers][2 positions][K][16 items] int16 (whole: [12 positions][2 halves][256][8 items]), then column minima /
  // thresholds, two entry records, row terms
  const size_t desc_off = whole ? (size_t)scan8_slab_bytes(12) : (size_t)4 * SCAN5_G * 2 * K;
  const size_t flds = desc_off + 4096 + 64 + (2 * REC_DW + 4) * sizeof(int32_t) + 4096 * sizeof(float);
  fl.desc_offset = (uint32_t)desc_off;
  // One persistent workgroup per CU (LDS admits exactly one), never more than there is work.  Batches in flight share the
  // chip: a persistent scan that took every CU would hold up the small kernels of the other batches until it drains, and
  // their scans behind them; with n_cus / share workgroups each, the scans of `share` batches run side by side, the small
  // kernels fit in between, and a scan's workgroups pull more entries each (a shorter tail).
  const int scan_cus = std::max(ix->n_cus / std::max(1, r.share), std::min(ix->n_cus, 32)) - ix->tune.reserve_cus;
  const unsigned n_persist = (unsigned)std::min<size_t>(wt.max_groups, (size_t)std::max(1, scan_cus));
  fl.packed8 = u8 ? ix->packed8 : nullptr;
  timed_launch(ix, s, "ivf_filter", [&] {
    // (instantiations: the rule that counts accepted rows doubles the selection code, and the kernel is larger than the
    // instruction cache as it is)
    if (whole) {
#ifdef FREDDY_LAB
      if (fl.prof && !fl.cand_count) hipLaunchKernelGGL((ivf_filter8_kernel<12, false, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
      else
#endif
      if (fl.cand_count) hipLaunchKernelGGL((ivf_filter8_kernel<12, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
      else hipLaunchKernelGGL((ivf_filter8_kernel<12, false>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
    } else if (u8) {
      if (fl.cand_count) hipLaunchKernelGGL((ivf_filter5_kernel<12, false, true, false, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
      else hipLaunchKernelGGL((ivf_filter5_kernel<12, false, false, false, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
    } else if (fl.cand_count) {
      if (K == 1024) hipLaunchKernelGGL((ivf_filter5_kernel<12, true, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
      else hipLaunchKernelGGL((ivf_filter5_kernel<12, false, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
    }
#ifdef FREDDY_LAB
    else if (K == 1024 && fl.prof) hipLaunchKernelGGL((ivf_filter5_kernel<12, true, false, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
#endif
    else if (K == 1024) hipLaunchKernelGGL((ivf_filter5_kernel<12, true, false>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
    else hipLaunchKernelGGL((ivf_filter5_kernel<12, false, false>), dim3(n_persist), dim3(SPEC2_T), flds, s, fl);
  });
  HIP_TRY(hipGetLastError());
  if (wt.sp_cap > 0) {
    // cells that one or two queries of the batch probe: item by item (sparse5.h), six workgroups of four waves per CU
    SparseArgs sp;
    sp.qc = fl.qc; sp.qscale = ra.qscale; sp.qn = ra.qn; sp.pmax = ix->pmax; sp.rterm = ix->rterm; sp.packed = ix->packed;
    sp.blk_off = ix->blk_off; sp.list_off = ix->list_off; sp.sorted_item = ra.sorted_item; sp.item_query = pa.item_query;
    sp.item_dist = pa.item_dist; sp.sp_cell = wt.sp_cell; sp.sp_first = wt.sp_first; sp.sp_chunk = wt.sp_chunk;
    sp.n_units = wt.n_sparse; sp.work_counter = wt.sp_counter; sp.surv = fl.surv; sp.surv_count = fl.surv_count; sp.packed8 = fl.packed8;
    sp.cand_count = fl.cand_count; sp.K = K; sp.L = r.L; sp.upi = r.upi; sp.sentinel = r.sentinel; sp.keep_all = fl.keep_all; sp.tau_run = fl.tau_run;
    const unsigned sp_grid = (unsigned)std::min<size_t>(wt.sp_cap, (size_t)std::max(1, scan_cus) * (wt.sp_pairs ? 3 : 6));
    timed_launch(ix, s, "sparse_items", [&] {
      if (wt.sp_pairs) {   // (cell, chunk) units of one or two items: the rows of a two-item cell are read once
        if (u8) {
          if (fl.cand_count) hipLaunchKernelGGL((sparse_pair5_kernel<12, true, true>), dim3(sp_grid), dim3(256), 0, s, sp);
          else hipLaunchKernelGGL((sparse_pair5_kernel<12, false, true>), dim3(sp_grid), dim3(256), 0, s, sp);
        } else if (fl.cand_count) hipLaunchKernelGGL((sparse_pair5_kernel<12, true, false>), dim3(sp_grid), dim3(256), 0, s, sp);
        else hipLaunchKernelGGL((sparse_pair5_kernel<12, false, false>), dim3(sp_grid), dim3(256), 0, s, sp);
      } else if (u8) {
        if (fl.cand_count) hipLaunchKernelGGL((sparse_item5_kernel<12, true, true>), dim3(sp_grid), dim3(256), 0, s, sp);
        else hipLaunchKernelGGL((sparse_item5_kernel<12, false, true>), dim3(sp_grid), dim3(256), 0, s, sp);
      } else if (fl.cand_count) hipLaunchKernelGGL((sparse_item5_kernel<12, true>), dim3(sp_grid), dim3(256), 0, s, sp);
      else hipLaunchKernelGGL((sparse_item5_kernel<12, false>), dim3(sp_grid), dim3(256), 0, s, sp);
    });
    HIP_TRY(hipGetLastError());
  }
#ifdef FREDDY_LAB
  if (fl.prof && K == 1024 && !fl.cand_count) if (int rc = scan_prof_print(ix, s, fl.prof, n_persist)) return rc;   // (the counters live in one instantiation)
  if (fl.prof && whole && !fl.cand_count) {   // fused8.h: gatherer wave 0's stages
    std::vector<long long> h(8 * (size_t)n_persist);
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(h.data(), fl.prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    double sum[8] = {0}, life_max = 0, life_min = 1e30;
    for (unsigned b = 0; b < n_persist; ++b) {
      for (int i = 0; i < 8; ++i) sum[i] += (double)h[b * 8 + i];
      life_max = std::max(life_max, (double)h[b * 8 + 7]); life_min = std::min(life_min, (double)h[b * 8 + 7]);
    }
    int32_t ng = 0;
    HIP_TRY(hipMemcpy(&ng, wt.n_groups, 4, hipMemcpyDeviceToHost));
    const double e = std::max(1, ng);
    fprintf(stderr, "[scan8 prof] wgs=%u entries=%d  gatherer wave 0 cycles/entry: gather=%.0f B1=%.0f colmin=%.0f B2=%.0f S1=%.0f B3+S2+B4=%.0f | per workgroup: prologue %.0f, stages %.0f, life mean %.0f min %.0f max %.0f\n",
            n_persist, ng, sum[0] / e, sum[1] / e, sum[2] / e, sum[3] / e, sum[4] / e, sum[5] / e, sum[6] / n_persist,
            (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5]) / n_persist, sum[7] / n_persist, life_min, life_max);
  }
#endif

  MergeRefineArgs mr;
  mr.surv = fl.surv; mr.surv_count = fl.surv_count; mr.active = r.active; mr.round_rows = pa.round_rows;
  mr.item_cell = pa.item_cell; mr.queries = r.d_q; mr.coarse = ix->coarse; mr.cbR = ix->cbR;
  mr.qn = ws->w_qn.as<float>(); mr.pmax = ix->pmax; mr.qscale5 = ws->w_qn.as<float>() + (size_t)Q * m; mr.packed = ix->packed; mr.pos = ix->pos; mr.blk_cell = ix->blk_cell;
  mr.cand_count = fl.cand_count; mr.violations = ix->viol; mr.out_ids = r.d_out_ids; mr.out_dist = r.d_out_dist;
  mr.found = ws->w_found.as<int32_t>(); mr.next_active = r.next; mr.n_next = ws->w_cnt.as<int32_t>();
  mr.status = r.d_status;
  mr.n_active = r.n_active; mr.W = r.W; mr.upi = r.upi; mr.L = r.L; mr.k = r.k; mr.found_rule = r.found_rule;
  mr.first_round = r.first() ? 1 : 0; mr.K = K; mr.d = ix->d; mr.sentinel = r.sentinel;
  mr.refine_all = (ix->tune.check_brackets & 1) ? 1 : 0; mr.fence = 0;
  mr.slices = 0; mr.part = nullptr;
  if (r.merge_slices > 0) {
    // a batch over the flat PQ table: `merge_slices` workgroups per query, each over its share of the pseudo-lists (r.W is the
    // padded item count per query, a multiple of the slices), then merge_replay_kernel over the slices' keys
    const int SL = r.merge_slices;
    if (ws->w_part.ensure(sizeof(u64) * (size_t)r.n_active * SL * r.L)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
    mr.slices = SL; mr.part = ws->w_part.as<u64>(); mr.W = r.W / SL; mr.n_active = r.n_active * SL;
    timed_launch(ix, s, "merge_refine", [&] {
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 12, true, true>), dim3(r.n_active * SL), dim3(768), 0, s, mr);
    });
    HIP_TRY(hipGetLastError());
    MergeArgs ma;
    ma.part = ws->w_part.as<u64>(); ma.active = nullptr; ma.pos_to_id = nullptr; ma.round_rows = nullptr; ma.cand_count = nullptr;
    ma.out_ids = r.d_out_ids; ma.out_dist = r.d_out_dist; ma.found = nullptr; ma.next_active = nullptr; ma.n_next = nullptr; ma.status = nullptr;
    ma.n_active = r.n_active; ma.parts_per_query = SL; ma.L = r.L; ma.k = r.k; ma.found_rule = 0; ma.first_round = 1; ma.sentinel = r.sentinel;
    return launch_merge(ix, s, ma);
  }
  timed_launch(ix, s, "merge_refine", [&] {
    // (one batch at a time: four waves per query, the shortest latency; batches in flight: one wave per query, the smallest footprint)
    // (hundreds of survivor regions per query -- a batch over the flat PQ table: four waves, which split the selection)
    const bool many_regions = (size_t)r.W * r.upi * FUSED_NW > 256;
    if (!many_regions && r.share > 1)
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 1>), dim3(r.n_active), dim3(64), 0, s, mr);
    else if (many_regions && r.n_active <= 256)   // (a few queries with many qualifying rows each: twelve waves, 64 rows per round of the exact stage)
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 12, true>), dim3(r.n_active), dim3(768), 0, s, mr);
    else if (many_regions)                         // (a large batch: the workgroups' footprint decides, 95 against 52 us at 1024 queries)
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 4, true>), dim3(r.n_active), dim3(256), 0, s, mr);
    else
      hipLaunchKernelGGL((merge_refine_kernel<25, 12, 4>), dim3(r.n_active), dim3(256), 0, s, mr);
  });
  HIP_TRY(hipGetLastError());
  return 0;
}

// The yardstick: the reference's arithmetic for every probed row (fused3.h).  ivf_spec2_kernel -> merge_surv_kernel.
static int ivf_scan_exact(IvfRun& r, const PlanArgs& pa, const WorkTable& wt) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int K = ix->K;
  FusedArgs fa;
  fa.resid = nullptr; fa.item_query = pa.item_query; fa.queries = r.d_q; fa.coarse = ix->coarse;
  fa.sorted_item = ws->w_sorted.as<int32_t>(); fa.group_cell = wt.group_cell; fa.group_first = wt.group_first;
  fa.group_cnt = wt.group_cnt; fa.n_groups = wt.n_groups; fa.work_counter = wt.work_counter;
  fa.cbP = ix->cbP; fa.blk_off = ix->blk_off; fa.packed = ix->packed; fa.pos = ix->pos;
  fa.surv = ws->w_surv.as<u64>(); fa.surv_count = ws->w_surv_cnt.as<int32_t>();
  fa.cand_count = (r.found_rule == 1) ? ws->w_cand.as<int32_t>() : nullptr;
  fa.d = ix->d; fa.K = K; fa.L = r.L; fa.upi = r.upi;
  memcpy(&fa.sentinel_bits, &r.sentinel, 4);
  if (int rc = scan_prof_buffer(ix, ws, &fa.prof)) return rc;
  const size_t desc_off = ((size_t)2 * SPEC2_G * K * sizeof(float) + 15) & ~(size_t)15;
  const size_t flds = desc_off + 4096 + 64 + 512 + (size_t)SPEC2_G * 12 * 28 * sizeof(float);
  fa.desc_offset = (uint32_t)desc_off;
  const unsigned n_persist = (unsigned)std::min<size_t>(wt.max_groups, (size_t)ix->n_cus);
  timed_launch(ix, s, "ivf_exact_scan", [&] {
    if (K == 1024) hipLaunchKernelGGL((ivf_spec2_kernel<25, 12, true>), dim3(n_persist), dim3(SPEC2_T), flds, s, fa);
    else hipLaunchKernelGGL((ivf_spec2_kernel<25, 12, false>), dim3(n_persist), dim3(SPEC2_T), flds, s, fa);
  });
  HIP_TRY(hipGetLastError());
#ifdef FREDDY_LAB
  if (fa.prof) if (int rc = scan_prof_print(ix, s, fa.prof, n_persist)) return rc;
#endif
  MergeSurvArgs ms;
  ms.surv = fa.surv; ms.surv_count = fa.surv_count; ms.active = r.active; ms.round_rows = pa.round_rows;
  ms.cand_count = fa.cand_count; ms.out_ids = r.d_out_ids; ms.out_dist = r.d_out_dist;
  ms.found = ws->w_found.as<int32_t>(); ms.next_active = r.next; ms.n_next = ws->w_cnt.as<int32_t>();
  ms.status = r.d_status;
  ms.n_active = r.n_active; ms.W = r.W; ms.upi = r.upi; ms.L = r.L; ms.k = r.k; ms.found_rule = r.found_rule;
  ms.first_round = r.first() ? 1 : 0; ms.sentinel = r.sentinel;
  timed_launch(ix, s, "merge_surv", [&] { hipLaunchKernelGGL(merge_surv_kernel, dim3(r.n_active), dim3(64), 0, s, ms); });
  HIP_TRY(hipGetLastError());
  return 0;
}

// Every other index shape (multi.h): exact LUTs of all items -> ivf_multi_kernel -> merge_surv_kernel.
static int ivf_scan_multi(IvfRun& r, const PlanArgs& pa, const WorkTable& wt) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int n_items = r.n_active * r.W;
  if (int rc = launch_lut(ix, s, r.d_q, pa.item_cell, ws->w_lut.as<float>(), n_items, ix->coarse, pa.item_query)) return rc;
  MultiArgs ma;
  ma.lut = ws->w_lut.as<float>(); ma.item_query = pa.item_query; ma.sorted_item = ws->w_sorted.as<int32_t>();
  ma.group_cell = wt.group_cell; ma.group_first = wt.group_first; ma.group_cnt = wt.group_cnt; ma.n_groups = wt.n_groups;
  ma.work_counter = wt.work_counter; ma.blk_off = ix->blk_off; ma.packed = ix->packed; ma.pos = ix->pos;
  ma.surv = ws->w_surv.as<u64>(); ma.surv_count = ws->w_surv_cnt.as<int32_t>();
  ma.cand_count = (r.found_rule == 1) ? ws->w_cand.as<int32_t>() : nullptr;
  ma.m = ix->m; ma.M2 = ix->M2; ma.K = ix->K; ma.L = r.L; ma.upi = r.upi;
  memcpy(&ma.sentinel_bits, &r.sentinel, 4);
  const size_t lds = multi_lds_bytes(ix->m, ix->K);
  // persistent workgroups: as many as fit (LDS, 8 waves of ~100 registers), never more than there is work
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (size_t)(150 * 1024) / lds));
  const unsigned grid = (unsigned)std::min<size_t>(wt.max_groups, (size_t)ix->n_cus * per_cu);
  timed_launch(ix, s, "ivf_multi_scan", [&] { hipLaunchKernelGGL(ivf_multi_kernel, dim3(grid), dim3(MULTI_T), lds, s, ma); });
  HIP_TRY(hipGetLastError());
  MergeSurvArgs ms;
  ms.surv = ma.surv; ms.surv_count = ma.surv_count; ms.active = r.active; ms.round_rows = pa.round_rows;
  ms.cand_count = ma.cand_count; ms.out_ids = r.d_out_ids; ms.out_dist = r.d_out_dist;
  ms.found = ws->w_found.as<int32_t>(); ms.next_active = r.next; ms.n_next = ws->w_cnt.as<int32_t>();
  ms.status = r.d_status;
  ms.n_active = r.n_active; ms.W = r.W; ms.upi = r.upi; ms.L = r.L; ms.k = r.k; ms.found_rule = r.found_rule;
  ms.first_round = r.first() ? 1 : 0; ms.sentinel = r.sentinel;
  timed_launch(ix, s, "merge_surv", [&] { hipLaunchKernelGGL(merge_surv_kernel, dim3(r.n_active), dim3(64), 0, s, ms); });
  HIP_TRY(hipGetLastError());
  return 0;
}

// row blocks per workgroup of the generic scan: one workgroup per (query, probed cell) unless the list is huge -- but a
// handful of items (the reference's single-query ivfadc_search: W of them) would leave the chip to W workgroups: 32-block
// chunks then (one query over 10 lists of 3 000 rows: 30 instead of 10 workgroups)
static int generic_chunk_blocks(int n_items) { return n_items <= 64 ? 32 : 256; }

// Generic path (small batches, other m / S / K, k > 32): lut_build (residual inline) -> adc_scan -> merge_replay;
// the LUTs round-trip through memory.
static int ivf_scan_generic(IvfRun& r, const PlanArgs& pa) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int n_items = r.n_active * r.W;
  const int chunk_blocks = generic_chunk_blocks(n_items);
  const int nchunk = std::max(1, (ix->max_list_blocks + chunk_blocks - 1) / chunk_blocks);
  // (the residual r = q - coarse[cell] is formed by the LUT kernel: one launch less in a single query's chain)
  if (int rc = launch_lut(ix, s, r.d_q, pa.item_cell, ws->w_lut.as<float>(), n_items, ix->coarse, pa.item_query)) return rc;
  ScanArgs sa;
  sa.lut = ws->w_lut.as<float>(); sa.item_list = pa.item_cell; sa.item_query = pa.item_query;
  sa.blk_off = ix->blk_off; sa.packed = ix->packed; sa.pos = ix->pos; sa.part = ws->w_part.as<u64>();
  sa.cand_count = ws->w_cand.as<int32_t>();
  sa.m = ix->m; sa.K = ix->K; sa.chunk_blocks = chunk_blocks; sa.nchunk = nchunk; sa.L = r.L;
  memcpy(&sa.sentinel_bits, &r.sentinel, 4);
  MergeArgs ma;
  ma.part = sa.part; ma.active = r.active; ma.pos_to_id = nullptr; ma.round_rows = pa.round_rows;
  ma.cand_count = sa.cand_count; ma.out_ids = r.d_out_ids; ma.out_dist = r.d_out_dist;
  ma.found = ws->w_found.as<int32_t>(); ma.next_active = r.next; ma.n_next = ws->w_cnt.as<int32_t>();
  ma.status = r.d_status;
  ma.n_active = r.n_active; ma.parts_per_query = r.W * nchunk; ma.L = r.L; ma.k = r.k;
  ma.found_rule = r.found_rule; ma.first_round = r.first() ? 1 : 0; ma.sentinel = r.sentinel;
  if (2 * r.k > 1024) return bigk_select_replay(ix, s, ws, sa, n_items, ma, r.Q);
  if (int rc = launch_scan(ix, s, sa, n_items)) return rc;
  return launch_merge(ix, s, ma);
}

// One probing round of a chunk: cell selection, then the scan + merge of the path the chunk takes.
static int ivfadc_round(IvfRun& r) {
  PlanArgs pa;
  if (int rc = ivf_plan(r, pa)) return rc;
  if (r.fused) {
    WorkTable wt;
    if (int rc = ivf_work_table(r, wt)) return rc;
    return (r.scan_kernel == 5) ? ivf_scan_filter(r, pa, wt) : (r.scan_kernel == 2) ? ivf_scan_multi(r, pa, wt) : ivf_scan_exact(r, pa, wt);
  }
  return ivf_scan_generic(r, pa);
}

// One chunk of queries (device pointers): workspace, coarse distances and round one are enqueued on s, nothing is
// synchronised.  `share` = the batches in flight on this handle (the persistent scan takes n_cus / share CUs).  The
// state for further rounds stays in r (and in the stream's workspace): ivfadc_finish() runs them.
static int ivfadc_begin(freddy_gpu_index* ix, hipStream_t s, int share, const float* d_q, int Q, int k, int W,
                        float sentinel, int found_rule, int32_t* d_out_ids, float* d_out_dist,
                        int32_t* d_status, IvfRun& r, const std::function<int(int, int)>* stage = nullptr, bool by_pieces = false) {
  // stage(q_lo, q_hi) (host-buffer pipeline): brings the queries [q_lo, q_hi) of this chunk into d_q on stream s -- called once for
  // the whole chunk, or piece by piece with the piece's cell-selection / table launch right behind it
  Workspace* ws = workspace_for(ix, s);
  const int C = ix->C, m = ix->m, K = ix->K;
  if (2 * W > 1024) return fail(FREDDY_E_LIMIT, "W=%d exceeds this build's limit of 512 probes per round", W);
  r.ix = ix; r.ws = ws; r.s = s; r.d_q = d_q; r.Q = Q; r.k = k; r.W = W; r.L = std::min(2 * k, 64 * 16);
  r.sentinel = sentinel; r.d_out_ids = d_out_ids; r.d_out_dist = d_out_dist; r.d_status = d_status;
  r.share = std::max(1, share);
  // FREDDY_FOUND_BATCH_UDF = the accepted-rows rule + the batch UDF's cell limit (argmin from minDist = 1000,
  // freddy.c:853-866); ivfadc_search's cell list starts at 100.0 (freddy.c:266-283)
  r.found_rule = found_rule == FREDDY_FOUND_ROWS ? 0 : 1;
  r.cell_limit = found_rule == FREDDY_FOUND_BATCH_UDF ? 1000.0f : 100.0f;
  const size_t items = (size_t)Q * W;
  // Cell-grouped scans: residual PQ with m=12, S=25, K<=1024 and a selection width that one wave holds
  // (2k <= 64); lists longer than 8 chunks of 4096 rows would need survivor buffers out of proportion.  They
  // pay off once several (query, cell) items share a cell, i.e. for batches; option fused = 1 / 0 forces
  // them / the generic lut_build + adc_scan kernels (the tests run both).
  r.upi = std::max(1, (ix->max_list_blocks + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS);
  r.fused = ix->tune.fused != 0 && m == 12 && ix->S == 25 && K <= 1024 && ix->cbP && r.L <= 64 && r.upi <= 8 &&
            (ix->tune.fused == 1 || items >= 256);
  r.scan_kernel = (ix->tune.scan_kernel == 3 || !ix->rterm) ? 3 : 5;
  if (!r.fused) {
    // every other shape whose interleaved LUT slab fits the LDS: the cell-grouped exact scan of multi.h (option fused = 0
    // keeps the generic kernels, fused = 1 takes it for small batches too)
    const bool special = m == 12 && ix->S == 25 && K <= 1024 && ix->cbP;
    // (lists of up to 32 chunks: the reference's shipped 32-cell configuration has 37 000 rows per list)
    r.fused = ix->tune.fused != 0 && !special && multi_lds_bytes(m, K) <= (size_t)150 * 1024 && r.L <= 64 && r.upi <= 32 &&
              (ix->tune.fused == 1 || items >= 256);
    if (r.fused) r.scan_kernel = 2;
  }
  r.tiled = Q >= 32;
  r.zeroed = r.tiled || ix->d <= 1024;
  r.records_ready = false; r.merge_slices = 0;
  // (the MFMA tile is 64 queries wide and the plan keeps a query's distances in registers: batches, <= 1024 cells)
  r.approx = ix->tune.coarse_approx != 0 && r.tiled && ix->Cpad <= COARSE_STREAM_MAX_CPAD && 2 * W <= 64 && ix->d <= 300 && ix->d % 4 == 0 && ix->coarseP;
  const int Cpad = ix->Cpad, used_words = (C + 31) / 32;
  if (ws->w_distT.ensure(sizeof(float) * (size_t)Q * Cpad) ||
      ws->w_used.ensure(sizeof(uint32_t) * (size_t)Q * used_words) ||
      ws->w_item_cell.ensure(sizeof(int32_t) * items) || ws->w_item_query.ensure(sizeof(int32_t) * items) ||
      ws->w_rows.ensure(sizeof(int32_t) * Q) || ws->w_cand.ensure(sizeof(int32_t) * 2 * Q) ||
      ws->w_qn2.ensure(sizeof(float) * Q) || ws->w_item_dist.ensure(sizeof(float) * items) ||
      ws->w_found.ensure(sizeof(int32_t) * Q) || ws->w_act0.ensure(sizeof(int32_t) * Q) ||
      ws->w_act1.ensure(sizeof(int32_t) * Q) || ws->w_cnt.ensure(sizeof(int32_t) * 8))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d, W=%d)", Q, W);
  if (r.fused) {
    // cell_count[C] + cursors; cell_items[C][Q]; work table: 3 arrays of (items/G + C + 1) * upi entries
    if (ws->w_cellcnt.ensure(sizeof(int32_t) * (size_t)C * 3) || ws->w_sorted.ensure(sizeof(int32_t) * (size_t)C * Q) ||
        ws->w_groups.ensure(sizeof(int32_t) * 3 * ((items / MULTI_G + (size_t)C + 1) * r.upi + items * r.upi)) ||   // + the (item, chunk) units of sparse cells
        ws->w_surv.ensure(sizeof(u64) * items * r.upi * FUSED_NW * FUSED_RMAX * 64) ||
        ws->w_surv_cnt.ensure(sizeof(int32_t) * items * r.upi * FUSED_NW))
      return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d, W=%d)", Q, W);
    if (r.scan_kernel == 2 && ws->w_lut.ensure(sizeof(float) * items * (size_t)m * K))
      return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d, W=%d)", Q, W);
    if (r.scan_kernel == 5 &&
        (ws->w_qc.ensure(sizeof(uint32_t) * (size_t)Q * m * 640) || ws->w_qn.ensure(sizeof(float) * (size_t)Q * m * 2)))   // (512 + 128: the compact copy, fused8.h)
      return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d, W=%d)", Q, W);
  } else {
    // (a later probing round has fewer items and may take the finer chunks: room for either)
    const size_t nchunk_big = (size_t)std::max(1, (ix->max_list_blocks + 255) / 256), nchunk_small = (size_t)std::max(1, (ix->max_list_blocks + 31) / 32);
    const size_t parts = std::max(items * nchunk_big, std::min<size_t>(items, 64) * nchunk_small);
    if (ws->w_resid.ensure(sizeof(float) * items * (size_t)ix->d) || ws->w_lut.ensure(sizeof(float) * items * (size_t)m * K) ||
        ws->w_part.ensure(sizeof(u64) * parts * r.L))   // (adc_scan_kernel leaves ONE list of L keys per (item, chunk): kernels.h)
      return fail(FREDDY_E_NOMEM, "workspace allocation failed (Q=%d, W=%d)", Q, W);
  }

  // pieces of whole 32-query tiles; a piece of fewer than 128 queries is not worth a launch of its own
  const int pieces = (stage && by_pieces && ix->tune.coarse_pieces && ivf_coarse_by_pieces(r) && Q >= 512) ? 4 : 1;
  const int per = ((Q + pieces - 1) / pieces + 31) & ~31;
  for (int q_lo = 0; q_lo < Q; q_lo += (pieces > 1 ? per : Q)) {
    const int q_n = pieces > 1 ? std::min(per, Q - q_lo) : Q;
    if (stage) if (int rc = (*stage)(q_lo, q_lo + q_n)) return rc;
    if (int rc = ivf_coarse(r, q_lo, q_n)) return rc;
  }
  ix->last_Q = Q;
  r.n_active = Q; r.active = nullptr; r.next = ws->w_act0.as<int32_t>();
  r.round = 0;
  return ivfadc_round(r);
}

// The extra rounds of the reference's "while (foundInstances < k)" loop (freddy.c:262, :835), one host sync per
// round.  n_next: the number of queries round one left unfinished if the caller has already read it back
// (ws->w_cnt[0], after the stream drained), -1 = read it here.
static int ivfadc_finish(IvfRun& r, int n_next) {
  Workspace* ws = r.ws;
  freddy_gpu_index* ix = r.ix;
  hipStream_t s = r.s;
  const int max_rounds = (ix->C + r.W - 1) / r.W + 1;
  for (;;) {
    if (n_next < 0) {
      int32_t h = 0;
      HIP_TRY(hipMemcpyAsync(&h, ws->w_cnt.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      n_next = h;
    }
    if (n_next <= 0 || ++r.round >= max_rounds) break;
    HIP_TRY(hipMemsetAsync(ws->w_cnt.p, 0, sizeof(int32_t), s));
    r.active = r.next;
    r.next = (r.next == ws->w_act0.as<int32_t>()) ? ws->w_act1.as<int32_t>() : ws->w_act0.as<int32_t>();
    r.n_active = n_next;
    if (int rc = ivfadc_round(r)) return rc;
    n_next = -1;
  }
  return 0;
}

int max_queries_per_chunk(const freddy_gpu_index* ix, int W, int k) {
  // workspace per query: the LUTs of its W items (generic path) or their survivor regions (fused path)
  const size_t upi = (size_t)std::max(1, (ix->max_list_blocks + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS);
  size_t lut_bytes = sizeof(float) * (size_t)ix->m * ix->K * (size_t)W;
  if (2 * k > 64) {
    // lists beyond the cell-grouped scans' selection width take the generic kernels: one list of L keys per (item, chunk) beside
    // the LUTs, and from k = 513 on the passes' selected keys (bigk.h: ceil(2k / 1024) x 1024 per query)
    const size_t L = (size_t)std::min(2 * k, 1024), nchunk = (size_t)std::max(1, (ix->max_list_blocks + 255) / 256);
    lut_bytes += sizeof(u64) * (size_t)W * nchunk * L;
    if (2 * k > 1024) lut_bytes += sizeof(u64) * (size_t)((2 * k + BIGK_PASS - 1) / BIGK_PASS) * BIGK_PASS;
  }
  const bool special = ix->m == 12 && ix->S == 25 && ix->K <= 1024 && ix->cbP;
  const size_t surv_bytes = upi <= (special ? 8u : 32u) ? sizeof(u64) * (size_t)W * upi * FUSED_NW * FUSED_RMAX * 64 : 0;
  const size_t per_query = special ? std::max(lut_bytes, surv_bytes) : lut_bytes + surv_bytes;   // (multi.h: the LUTs of all items AND their survivor regions)
  size_t n = ((size_t)ix->tune.lut_budget_mb << 20) / std::max<size_t>(per_query, 1);
  // the fused path's per-cell item buckets are [C][queries of the chunk]: keep them within 256 MiB
  if (surv_bytes) n = std::min<size_t>(n, ((size_t)256 << 20) / (sizeof(int32_t) * (size_t)std::max(ix->C, 1)));
  if (n < 1) n = 1;
  if (n > (1u << 20)) n = 1u << 20;
  return (int)n;
}

extern "C" int freddy_gpu_ivfadc_search_dev(freddy_gpu_index_t* ix, const float* d_queries, int32_t Q, int32_t k,
                                            int32_t W, float sentinel, int32_t found_rule, int32_t* d_out_ids,
                                            float* d_out_dist, int32_t* d_status, void* hip_stream) {
  if (int rc = check_search_args(ix, KIND_IVF, d_queries, Q, k, d_out_ids, d_out_dist)) return rc;
  if (W <= 0) return fail(FREDDY_E_ARG, "W must be positive");
  if (found_rule < 0 || found_rule > 2 || (found_rule == FREDDY_FOUND_BATCH_UDF && W != 1))
    return fail(FREDDY_E_ARG, "bad found_rule (FREDDY_FOUND_BATCH_UDF needs W == 1)");
  if (W > ix->C) W = ix->C;
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ix->stream;
  const int qc = max_queries_per_chunk(ix, W, k);
  for (int q0 = 0; q0 < Q; q0 += qc) {
    const int n = std::min(qc, Q - q0);
    IvfRun r;
    if (int rc = ivfadc_begin(ix, s, scan_share_now(ix->tune.scan_share, false, ix->device), d_queries + (size_t)q0 * ix->d, n, k, W, sentinel, found_rule,
                              d_out_ids + (size_t)q0 * k, d_out_dist + (size_t)q0 * k, d_status, r))
      return rc;
  }
  return FREDDY_OK;
}

// ---------------------------------------------------------------------------------------
// The host-buffer call (what the PostgreSQL hosts make: one synchronous call per batch, freddy.c:679-999) as a
// pipeline.  The batch is cut into sub-batches of <= pipeline_batch queries; sub-batch j goes to lane j mod L
// (L <= 4 library-owned streams, each with its own workspace, pinned staging and device buffers):
//   host memcpy of its queries into the lane's pinned buffer (skipped when the caller's buffer is pinned itself:
//   freddy_gpu_host_alloc) -> asynchronous H2D -> round one of the search with an explicit scan share of L -> asynchronous
//   D2H of the lists and of the round's straggler count into pinned memory -> event.
// The host only waits when it needs a lane again (or at the end), and that is where a sub-batch's rare extra probing
// rounds run and its lists are copied out: the transfers and the latency-bound ends of one sub-batch hide under the
// scans of its neighbours, and ONE stream synchronisation per lane ends the call.
// ---------------------------------------------------------------------------------------
static int lane_open(freddy_gpu_index* ix, Lane& l, LaneSlot& c, size_t in_bytes, size_t n, size_t n_out) {
  // lane 0 is the handle's own stream: a call of one sub-batch (<= pipeline_batch queries: what a PostgreSQL backend makes) uses
  // ONE stream per process, and only larger calls create further streams -- every stream a process creates takes a hardware
  // queue, and the queues of several backends on one GPU are what decides their aggregate rate (profiles/r06_backends.txt)
  if (!l.stream && &l == &ix->lanes[0] && !ix->tune.lane0_own) l.stream = ix->stream;
  if (!l.stream) HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
  if (!c.done) HIP_TRY(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
  if (in_bytes > c.h_in_cap) {
    if (c.h_in) (void)hipHostFree(c.h_in);
    c.h_in = nullptr; c.h_in_cap = 0;
    const size_t want = in_bytes + in_bytes / 8 + 256;
    if (hipHostMalloc(&c.h_in, want, hipHostMallocDefault) != hipSuccess) { c.h_in = nullptr; return fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
    c.h_in_cap = want;
  }
  const size_t out_bytes = (n_out * 2 + 1 + n + 1) * 4;   // (+ the completion word)
  if (out_bytes > c.h_out_cap) {
    if (c.h_out) (void)hipHostFree(c.h_out);
    c.h_out = nullptr; c.h_out_cap = 0;
    const size_t want = out_bytes + out_bytes / 8 + 256;
    if (hipHostMalloc(&c.h_out, want, hipHostMallocDefault) != hipSuccess) { c.h_out = nullptr; return fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
    c.h_out_cap = want;
  }
  if (c.d_q.ensure(in_bytes + 16) || c.d_ids.ensure(n_out * 4) || c.d_dist.ensure(n_out * 4))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  return 0;
}

// The old shape of the call, kept for what the pipeline hands back: a (small) batch searched to the end on the library's
// own stream -- round one, then the extra rounds of the reference's "while (foundInstances < k)" loop with a host sync each.
static int ivfadc_sync_search(freddy_gpu_index* ix, const float* queries, int Q, int k, int W, float sentinel, int found_rule,
                              int32_t* out_ids, float* out_dist) {
  Workspace* ws = workspace_for(ix, ix->stream);
  hipStream_t s = ix->stream;
  if (ws->w_q.ensure(sizeof(float) * (size_t)Q * ix->d) || ws->w_out_ids.ensure(sizeof(int32_t) * (size_t)Q * k) ||
      ws->w_out_dist.ensure(sizeof(float) * (size_t)Q * k))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  HIP_TRY(hipMemcpyAsync(ws->w_q.p, queries, sizeof(float) * (size_t)Q * ix->d, hipMemcpyHostToDevice, s));
  const int qc = max_queries_per_chunk(ix, W, k);
  for (int q0 = 0; q0 < Q; q0 += qc) {
    const int n = std::min(qc, Q - q0);
    IvfRun r;
    if (int rc = ivfadc_begin(ix, s, scan_share_now(ix->tune.scan_share, true, ix->device), ws->w_q.as<float>() + (size_t)q0 * ix->d, n, k, W, sentinel, found_rule,
                              ws->w_out_ids.as<int32_t>() + (size_t)q0 * k, ws->w_out_dist.as<float>() + (size_t)q0 * k, nullptr, r))
      return rc;
    if (int rc = ivfadc_finish(r, -1)) return rc;
  }
  HIP_TRY(hipMemcpyAsync(out_ids, ws->w_out_ids.p, sizeof(int32_t) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(out_dist, ws->w_out_dist.p, sizeof(float) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

struct PipeCall {   // the arguments of one host-buffer call, for the lanes' retire step
  freddy_gpu_index* ix;
  const float* queries;
  int k, W, found_rule;
  float sentinel;
  int32_t* out_ids;
  float* out_dist;
};

// Wait for a slot's sub-batch and hand its lists to the caller.  Queries that round one left unfinished (their first W
// cells hold fewer than k rows -- rare) are searched again from the start, synchronously, with all their rounds: the
// search is deterministic, so that is the list the round-by-round continuation would have produced, and no lane has to
// keep per-round state while its stream already runs the next sub-batch.
static int lane_retire(LaneSlot& c, const PipeCall& pc) {
  if (!c.busy) return 0;
  c.busy = false;
  const int k = pc.k;
  const size_t n_out = (size_t)c.n * k;
  const int32_t* ho = static_cast<const int32_t*>(c.h_out);
  {
    // the copy-out kernel's last store is a completion word behind the lists: polled for up to a millisecond (a few
    // microseconds sooner than the event), then the event is waited for the usual way -- which is also where a fault in one
    // of the lane's kernels surfaces, before its output is trusted
    volatile const int32_t* flag = ho + 2 * n_out + 1 + (size_t)c.n;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(1000);
    int spins = 0;
    while (*flag == 0) {
      __builtin_ia32_pause();
      if ((++spins & 255) == 0 && std::chrono::steady_clock::now() > t_end) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*flag == 0) {
      HIP_TRY(hipEventSynchronize(c.done));
      HIP_TRY(hipGetLastError());
    }
  }
  memcpy(pc.out_ids + (size_t)c.q0 * k, ho, n_out * 4);
  memcpy(pc.out_dist + (size_t)c.q0 * k, ho + n_out, n_out * 4);
  const int n_next = std::min(ho[2 * n_out], c.n);
  if (n_next <= 0) return 0;
  const int d = pc.ix->d;
  std::vector<int32_t> who(ho + 2 * n_out + 1, ho + 2 * n_out + 1 + n_next);
  // (device-written numbers index the caller's buffers: a value outside the sub-batch -- e.g. after a kernel fault whose
  // error has not surfaced yet -- must never become a host read or write out of bounds)
  for (int i = 0; i < n_next; ++i)
    if (who[(size_t)i] < 0 || who[(size_t)i] >= c.n) return fail(FREDDY_E_HIP, "sub-batch returned a straggler index %d outside [0, %d)", who[(size_t)i], c.n);
  std::vector<float> q((size_t)n_next * d);
  std::vector<int32_t> ri((size_t)n_next * k);
  std::vector<float> rd((size_t)n_next * k);
  for (int i = 0; i < n_next; ++i) memcpy(&q[(size_t)i * d], pc.queries + ((size_t)c.q0 + who[(size_t)i]) * d, sizeof(float) * d);
  if (int rc = ivfadc_sync_search(pc.ix, q.data(), n_next, k, pc.W, pc.sentinel, pc.found_rule, ri.data(), rd.data())) return rc;
  for (int i = 0; i < n_next; ++i) {
    memcpy(pc.out_ids + ((size_t)c.q0 + who[(size_t)i]) * k, &ri[(size_t)i * k], sizeof(int32_t) * k);
    memcpy(pc.out_dist + ((size_t)c.q0 + who[(size_t)i]) * k, &rd[(size_t)i * k], sizeof(float) * k);
  }
  return 0;
}

// the device-side address of a pinned (hipHostMalloc / freddy_gpu_host_alloc) host buffer, or NULL for ordinary memory
const void* pinned_device_pointer(const void* p) {
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
}

// The one-launch kernels' hand-off buffer: every published word carries the call's epoch in its top bit (one.h).  Calls of
// one shape write exactly the same words, so the epoch just flips; a different shape (or a new allocation) clears the
// buffer to epoch 0 and starts with epoch 1.
int one_buffer(Workspace* ws, hipStream_t s, uint64_t shape, size_t bytes, uint32_t* epoch) {
  void* before = ws->w_oneb.p;
  if (ws->w_oneb.ensure(bytes)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  // (one_pending: the previous call flipped the epoch and then did not launch -- an allocation or launch error in between --
  // so the words still carry the epoch of the call before it, which is this call's: cleared like a change of shape)
  if (ws->w_oneb.p != before || ws->one_shape != shape || ws->one_pending) {
    HIP_TRY(hipMemsetAsync(ws->w_oneb.p, 0, ws->w_oneb.cap, s));
    ws->one_shape = shape;
    ws->one_epoch = 1;
  } else {
    ws->one_epoch ^= 1u;
  }
  *epoch = ws->one_epoch;
  ws->one_pending = true;   // until the caller has seen its kernel launched (one_buffer_launched)
  return 0;
}

// ONE ivfadc_search query as one launch (one.h ivf_one_kernel).  Returns through *verdict: 2 = the list is in out_ids /
// out_dist; anything else = not answered here (shape not covered, the reference would probe a second time, or the grid
// never met at a barrier): the caller takes the multi-round path.
static bool ivf_one_shape(const freddy_gpu_index* ix, int Q, int k, int W, int found_rule) {
  return ix->tune.one_launch && !ix->one_launch_failed && Q == 1 && ix->kind == KIND_IVF && ix->m == 12 && ix->S == 25 && ix->d == 300 &&
         ix->K <= 1024 && (ix->K & 3) == 0 && W <= 32 && 2 * k <= 64 && ix->C <= 4096 && ix->coarse && ix->cbT &&
         found_rule != FREDDY_FOUND_BATCH_UDF && ix->replicas.empty();
}
static int ivf_one(freddy_gpu_index* ix, const float* queries, int k, int W, float sentinel, int found_rule, int32_t* out_ids,
                   float* out_dist, int* verdict) {
  *verdict = 0;
  hipStream_t s = ix->stream;
  Workspace* ws = workspace_for(ix, s);
  const int K = ix->K, C = ix->C, L = 2 * k;
  const size_t lutN = (size_t)12 * K, n_out = (size_t)k;
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)ix->n_cus, (int64_t)256, (int64_t)(56 * 1024) / (8 * L)}));
  if (G < W) return 0;   // (an item per workgroup at least)
  const size_t need_out = n_out * 8 + 16;
  if (need_out > ix->hio_out_cap) {
    if (ix->hio_out) (void)hipHostFree(ix->hio_out);
    ix->hio_out = nullptr; ix->hio_out_cap = 0;
    if (hipHostMalloc(&ix->hio_out, need_out + 256, hipHostMallocDefault) != hipSuccess) { ix->hio_out = nullptr; return fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
    ix->hio_out_cap = need_out + 256;
  }
  // the hand-off buffer: coarse distances | the W tables | the workgroups' lists | their accepted-row counts
  const size_t lut_off = (sizeof(float) * ((size_t)C + 8) + 255) & ~(size_t)255;
  const size_t part_off = (lut_off + sizeof(float) * (size_t)W * lutN + 255) & ~(size_t)255;
  const size_t cnt_off = (part_off + sizeof(u64) * (size_t)G * L + 255) & ~(size_t)255;
  uint32_t epoch = 0;
  if (int rc = one_buffer(ws, s, (2ull << 60) | ((uint64_t)C << 44) | ((uint64_t)K << 32) | ((uint64_t)W << 24) | ((uint64_t)G << 12) | (uint64_t)L,
                          cnt_off + sizeof(uint32_t) * (size_t)G, &epoch)) return rc;
  static const bool one_prof = getenv("FREDDY_GPU_ONE_PROF") != nullptr;
  if (one_prof && ws->w_one.ensure(256)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  int32_t* h_ids = static_cast<int32_t*>(ix->hio_out);
  float* h_dist = reinterpret_cast<float*>(h_ids + n_out);
  int32_t* err = reinterpret_cast<int32_t*>(static_cast<char*>(ix->hio_out) + n_out * 8);
  *err = 0;
  IvfOneArgs a;
  memcpy(a.qv, queries, sizeof(a.qv));
  a.coarse = ix->coarse; a.cbT = ix->cbT; a.list_off = ix->list_off; a.blk_off = ix->blk_off; a.packed = ix->packed; a.pos = ix->pos;
  char* ob = ws->w_oneb.as<char>();
  a.dist_g = reinterpret_cast<float*>(ob); a.lut_g = reinterpret_cast<float*>(ob + lut_off); a.part = reinterpret_cast<u64*>(ob + part_off);
  a.cnt_g = reinterpret_cast<uint32_t*>(ob + cnt_off);
  a.out_ids = h_ids; a.out_dist = h_dist; a.epoch = epoch; a.err = err;
  a.C = C; a.K = K; a.W = W; a.L = L; a.k = k; a.found_rule = found_rule == FREDDY_FOUND_ROWS ? 0 : 1;
  a.cell_limit = 100.0f; a.sentinel = sentinel;
  a.prof = one_prof ? ws->w_one.as<unsigned long long>() + 8 : nullptr;
  memcpy(&a.sentinel_bits, &sentinel, 4);
  const size_t n_mine = ((size_t)C + G - 1) / G;
  const size_t lds = std::max({(n_mine + 1) * 300 * sizeof(float), (size_t)C * 4 + 64 + 64 * sizeof(u64) + 64,
                               ((lutN * 4 + 15) & ~(size_t)15) + (size_t)ONE_WAVES * 64 * sizeof(u64),
                               (size_t)ONE_WAVES * 64 * sizeof(u64) + (size_t)G * L * sizeof(u64)});
  if (lds > 60 * 1024) return 0;
  timed_launch(ix, s, "ivf_one", [&] { hipLaunchKernelGGL((ivf_one_kernel<25>), dim3((unsigned)G), dim3(ONE_WG), lds, s, a); });
  HIP_TRY(hipGetLastError());
  ws->one_pending = false;
  {   // (the kernel's last store is this word: polled for up to a millisecond, then the stream is waited for the usual way)
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
  if (one_prof) {
    HIP_TRY(hipStreamSynchronize(s));
    unsigned long long st[16];
    (void)hipMemcpy(st, ws->w_one.as<unsigned long long>() + 8, sizeof(st), hipMemcpyDeviceToHost);
    fprintf(stderr, "[ivf_one] wg0: coarse %.2f barrier %.2f plan %.2f tables %.2f barrier %.2f stage %.2f scan %.2f publish %.2f | last: since wg0 start %.2f load %.2f merge+list %.2f us\n",
            (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01, (st[6] - st[5]) * 0.01,
            (st[7] - st[6]) * 0.01, (st[8] - st[7]) * 0.01, (st[10] - st[0]) * 0.01, (st[11] - st[10]) * 0.01, (st[12] - st[11]) * 0.01);
  }
  if (*err == 2) {
    memcpy(out_ids, h_ids, n_out * 4);
    memcpy(out_dist, h_dist, n_out * 4);
    *verdict = 2;
    return 0;
  }
  if (*err != 3) {   // a poll ran out: counters re-armed, this handle keeps to the multi-launch paths
    ix->one_launch_failed = true;
    ws->one_shape = 0;
    HIP_TRY(hipStreamSynchronize(s));
  }
  return 0;
}

// the batch [0, Q) of one device's handle
static int ivfadc_host_search(freddy_gpu_index* ix, const float* queries, int Q, int k, int W, float sentinel, int found_rule,
                              int32_t* out_ids, float* out_dist) {
  HIP_TRY(hipSetDevice(ix->device));
  if (ivf_one_shape(ix, Q, k, W, found_rule)) {
    int verdict = 0;
    if (int rc = ivf_one(ix, queries, k, W, sentinel, found_rule, out_ids, out_dist, &verdict)) return rc;
    if (verdict == 2) return FREDDY_OK;
  }
  const int cap = std::max(1, std::min(max_queries_per_chunk(ix, W, k), ix->tune.pipeline_batch));
  const int n_sub = (Q + cap - 1) / cap;
  const int per = (Q + n_sub - 1) / n_sub;               // equal sub-batches rather than full ones and a remainder
  const int n_lanes = std::min(n_sub, std::min(ix->tune.pipeline_lanes, FREDDY_LANES));
  const float* pinned_in = static_cast<const float*>(pinned_device_pointer(queries));
  const size_t row = sizeof(float) * (size_t)ix->d;
  const PipeCall pc{ix, queries, k, W, found_rule, sentinel, out_ids, out_dist};
  const BackendBusy busy;   // (the registry of backends on this GPU: core.hip)
  const int share_call = scan_share_now(ix->tune.scan_share, true, ix->device);
  int rc = 0;
#ifdef FREDDY_LAB
  static const bool trace = getenv("FREDDY_GPU_PIPE_TRACE") != nullptr;   // host timestamps of the pipeline's steps on stderr (lab builds)
#else
  constexpr bool trace = false;
#endif
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = trace ? now_us() : 0.0;
  auto slot_of = [&](int j) -> LaneSlot& { return ix->lanes[j % n_lanes].slot[(j / n_lanes) & 1]; };
  for (int j = 0; j < n_sub && !rc; ++j) {
    Lane& l = ix->lanes[j % n_lanes];
    LaneSlot& c = slot_of(j);
    double t0 = trace ? now_us() : 0.0, t1 = 0, t2 = 0, t3 = 0;
    if ((rc = lane_retire(c, pc))) break;
    if (trace) t1 = now_us();
    const int q0 = j * per, n = std::min(per, Q - q0);
    if ((rc = lane_open(ix, l, c, row * n, (size_t)n, (size_t)n * k))) break;
    c.q0 = q0; c.n = n;
    const float* src = pinned_in ? pinned_in + (size_t)q0 * ix->d : nullptr;
    const bool stage = !src || reinterpret_cast<uintptr_t>(src) % 16 || (row * n) % 16;   // (the copy kernel moves whole 16-byte words)
    const float* d_queries = c.d_q.as<float>();
    if (n <= 8) {
      // a handful of queries: the kernels read them where they are staged (pinned, mapped) -- one launch less
      if (stage) { memcpy(c.h_in, queries + (size_t)q0 * ix->d, row * n); src = static_cast<const float*>(c.h_in); }
      d_queries = src;
    }
    // pageable queries cross in pieces: the copy kernel of a piece reads it over PCIe while the host stages the next one (1.2 MB per
    // 1024 queries: 28 us of memcpy + 25 us of PCIe, back to back until round 4) -- and, since round 6, the piece's cell-selection /
    // table launch runs behind its copy while the next piece is staged (ivfadc_begin calls this once per piece)
    const std::function<int(int, int)> stage_piece = [&](int q_lo, int q_hi) -> int {
      if (n <= 8) return 0;
      const size_t b_lo = row * (size_t)q_lo, b_hi = row * (size_t)q_hi;
      // (a range of whole queries; whole 16-byte words except at the end of the sub-batch, whose buffers are padded)
      const size_t w0 = b_lo / 16, w1 = (b_hi + 15) / 16;
      if (b_lo % 16) return fail(FREDDY_E_ARG, "internal: a staged piece must start at a 16-byte word");
      if (stage) memcpy(static_cast<char*>(c.h_in) + b_lo, reinterpret_cast<const char*>(queries + (size_t)q0 * ix->d) + b_lo, b_hi - b_lo);
      const char* from = stage ? static_cast<const char*>(c.h_in) : reinterpret_cast<const char*>(src);
      // without per-piece coarse launches (every other path) the whole range still crosses as four copy launches
      const int cuts = (q_lo == 0 && q_hi == n && stage && b_hi >= (size_t)512 * 1024) ? 4 : 1;
      const size_t per16 = (w1 - w0 + cuts - 1) / cuts;
      for (int ci = 0; ci < cuts; ++ci) {
        const size_t a0 = w0 + (size_t)ci * per16, a1 = std::min(w1, a0 + per16);
        if (a0 >= a1) break;
        hipLaunchKernelGGL(lane_copy_in_kernel, dim3((unsigned)std::min<size_t>((a1 - a0 + 255) / 256, 512)), dim3(256), 0, l.stream,
                           reinterpret_cast<const uint4*>(from) + a0, reinterpret_cast<uint4*>(c.d_q.as<char>()) + a0, a1 - a0);
      }
      if (hipGetLastError() != hipSuccess) return fail(FREDDY_E_HIP, "launch of the query copy failed");
      return 0;
    };
    if (trace) t2 = now_us();
    IvfRun r;
    const int n_out = n * k;
    int32_t* h_flag = static_cast<int32_t*>(c.h_out) + 2 * (size_t)n_out + 1 + (size_t)n;
    *h_flag = 0;
    // (a call of ONE sub-batch launches its cell selection piece by piece behind the staged pieces: 0.398 -> 0.374 ms at 2048 queries;
    // with several lanes the extra launches only get in the way of the other lanes' chains: 0.613 -> 0.666 ms at 4096)
    if ((rc = ivfadc_begin(ix, l.stream, n_lanes * share_call, d_queries, n, k, W, sentinel, found_rule, c.d_ids.as<int32_t>(),
                           c.d_dist.as<float>(), nullptr, r, &stage_piece, n_sub == 1)))
      break;
    if (trace) t3 = now_us();
    hipLaunchKernelGGL(lane_copy_out_flag_kernel, dim3(1), dim3(1024), 0, l.stream, c.d_ids.as<int32_t>(),
                       c.d_dist.as<float>(), r.ws->w_cnt.as<int32_t>(), r.next, static_cast<int32_t*>(c.h_out), n_out, n, h_flag);
    if (hipGetLastError() != hipSuccess || hipEventRecord(c.done, l.stream) != hipSuccess) { rc = fail(FREDDY_E_HIP, "launch of the result copy failed"); break; }
    c.busy = true;
    if (trace)
      fprintf(stderr, "[pipe] sub %d lane %d n=%d  t=%.0f us: retire %.0f, stage %.0f, launches %.0f, copy-out + event %.0f\n", j, j % n_lanes, n,
              t0 - t_start, t1 - t0, t2 - t1, t3 - t2, now_us() - t3);
  }
  // drain in submission order (oldest first)
  for (int j = std::max(0, n_sub - 2 * n_lanes); j < n_sub && !rc; ++j) {
    const double t0 = trace ? now_us() : 0.0;
    rc = lane_retire(slot_of(j), pc);
    if (trace) fprintf(stderr, "[pipe] drain sub %d  t=%.0f us: %.0f\n", j, t0 - t_start, now_us() - t0);
  }
  if (rc)   // a failed call: nothing of it may still be in flight when the caller gets its buffers back
    for (Lane& l : ix->lanes) {
      if (l.stream) (void)hipStreamSynchronize(l.stream);
      for (LaneSlot& c : l.slot) c.busy = false;
    }
  return rc;
}

// Q queries split contiguously over a handle and its replicas (freddy_gpu_pin_ivf_multi): part g of G gets
// [lo, hi) with sizes differing by at most one.  fn(part, index of that part, lo, hi) runs on its own host thread
// for every part but the first; the first failure's code and message are returned on the caller's thread.
extern "C" int freddy_gpu_ivfadc_search(freddy_gpu_index_t* ix, const float* queries, int32_t Q, int32_t k, int32_t W,
                                        float sentinel, int32_t found_rule, int32_t* out_ids, float* out_dist) {
  if (int rc = check_search_args(ix, KIND_IVF, queries, Q, k, out_ids, out_dist)) return rc;
  if (W <= 0) return fail(FREDDY_E_ARG, "W must be positive");
  if (found_rule < 0 || found_rule > 2 || (found_rule == FREDDY_FOUND_BATCH_UDF && W != 1))
    return fail(FREDDY_E_ARG, "bad found_rule (FREDDY_FOUND_BATCH_UDF needs W == 1)");
  if (W > ix->C) W = ix->C;
  if (Q == 0) return FREDDY_OK;
  return over_replicas(ix, Q, [&](freddy_gpu_index* part, int lo, int hi) {
    return ivfadc_host_search(part, queries + (size_t)lo * ix->d, hi - lo, k, W, sentinel, found_rule, out_ids + (size_t)lo * k,
                              out_dist + (size_t)lo * k);
  });
}

// The kernels of this unit that want more than the default 64 KiB of dynamic LDS (a per-device function attribute).
int raise_lds_limits_ivfadc(int device) {
  static std::mutex mu;
  static std::vector<char> done;
  std::lock_guard<std::mutex> g(mu);
  if ((size_t)device < done.size() && done[(size_t)device]) return 0;
  const void* kernels[] = {
      (const void*)&adc_scan_kernel<12, 1>, (const void*)&adc_scan_kernel<12, 2>, (const void*)&adc_scan_kernel<12, 4>,
      (const void*)&adc_scan_kernel<12, 8>, (const void*)&adc_scan_kernel<12, 16>, (const void*)&adc_scan_kernel<0, 1>,
      (const void*)&adc_scan_kernel<0, 2>, (const void*)&adc_scan_kernel<0, 4>, (const void*)&adc_scan_kernel<0, 8>,
      (const void*)&adc_scan_kernel<0, 16>, (const void*)&adc_scan_kernel<12, 16, true>, (const void*)&adc_scan_kernel<0, 16, true>,
      (const void*)&ivf_spec2_kernel<25, 12, true>,
      (const void*)&ivf_spec2_kernel<25, 12, false>,
      (const void*)&ivf_filter5_kernel<12, true, false>, (const void*)&ivf_filter5_kernel<12, false, false>,
      (const void*)&ivf_filter5_kernel<12, true, true>, (const void*)&ivf_filter5_kernel<12, false, true>,
      (const void*)&ivf_filter5_kernel<12, false, false, false, true>, (const void*)&ivf_filter5_kernel<12, false, true, false, true>,
      (const void*)&ivf_filter8_kernel<12, false>, (const void*)&ivf_filter8_kernel<12, true>,
#ifdef FREDDY_LAB
      (const void*)&ivf_filter5_kernel<12, true, false, true>, (const void*)&ivf_filter8_kernel<12, false, true>,
#endif
      (const void*)&coarse_approx_kernel, (const void*)&coarse_approx16_kernel, (const void*)&ivf_multi_kernel};
  for (const void* k : kernels)
    HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // (static LDS beside the dynamic: 16384 keys + 4096 carried ids = 144 KB)
  HIP_TRY(hipFuncSetAttribute((const void*)&bigk_replay_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bigk_lds_bytes(16384, BIGK_KMAX)));
  if (done.size() <= (size_t)device) done.resize((size_t)device + 1, 0);
  done[(size_t)device] = 1;
  return 0;
}
