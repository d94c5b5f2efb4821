// exact.hip -- pin_vectors and exact brute-force kNN (core_functions.c:67-81, freddy--0.0.1.sql:426-454).
#include "internal.h"

#include "kernels.h"
#include "exact.h"
#include "exact2.h"

// ---------------------------------------------------------------------------------------
// exact brute-force kNN
// ---------------------------------------------------------------------------------------
// ---- exact kNN as filter + refine (exact2.h) ----------------------------------------------------------------------
// The table's largest |element| / largest row norm over rows [r0, r0 + n) of the row-major copy, folded into the handle's.
int exf_table_stats(freddy_gpu_index* ix, int64_t r0, int64_t n) {
  const bool shape_ok = ix->d % 4 == 0 && ix->d <= 512 && ix->d >= 16;
  if (!shape_ok) { ix->exf_ok = false; return 0; }
  if (ix->tune.exact_filter == 0) { ix->exf_ok = false; return 0; }   // (option exact_filter = 0 when the table is pinned: no third copy of it)
  if (n <= 0) return 0;
  if (ix->exf_small.ensure(4096)) return fail(FREDDY_E_NOMEM, "device allocation failed");
  uint32_t* st = ix->exf_small.as<uint32_t>() + 512;   // (the upper part of the small buffer; the lower one is per-call state)
  HIP_TRY(hipMemsetAsync(st, 0, 16, ix->stream));
  const unsigned grid = (unsigned)std::min<int64_t>((n + 3) / 4, (int64_t)ix->n_cus * 8);
  hipLaunchKernelGGL(exf_table_stats_kernel, dim3(grid), dim3(256), 0, ix->stream, ix->coarse + (size_t)r0 * ix->d, n, ix->d, st);
  HIP_TRY(hipGetLastError());
  uint32_t h[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h, st, 16, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  float amax, n2;
  memcpy(&amax, &h[0], 4); memcpy(&n2, &h[1], 4);
  const bool first = r0 == 0;
  if (h[2] || !(n2 < 1e30f)) { ix->exf_ok = false; return 0; }
  const float xn = std::sqrt(n2) * (1.0f + 1e-5f);
  int64_t relayout_from = r0;
  if (first) { ix->exf_ok = true; ix->exf_xnorm = xn; ix->exf_ex = exf_scale_exp(amax); }
  else if (ix->exf_ok) {
    ix->exf_xnorm = std::max(ix->exf_xnorm, xn);
    const int ex_new = std::min(ix->exf_ex, exf_scale_exp(amax));   // (a larger element: a smaller scale -> everything is laid out again)
    if (ex_new != ix->exf_ex) relayout_from = 0;
    ix->exf_ex = ex_new;
  } else return 0;
  // the fragment-order copy: rows [relayout_from, r0 + n) (whole strips; the strip the old last row sat in is rewritten)
  const int T = (ix->d + 15) / 16;
  const int64_t n_total = r0 + n, strips = (n_total + 31) / 32, strip0 = relayout_from / 32;
  const size_t need = (size_t)strips * T * 2 * 64 * 16;
  if (need > ix->exf_xf.cap) {
    DevBuf bigger;
    if (bigger.ensure(need)) { ix->exf_ok = false; return 0; }   // (no room for the copy: the all-exact kernels stay)
    if (ix->exf_xf.p && strip0 > 0) HIP_TRY(hipMemcpy(bigger.p, ix->exf_xf.p, (size_t)strip0 * T * 2 * 64 * 16, hipMemcpyDeviceToDevice));
    ix->bytes += (int64_t)bigger.cap - (int64_t)ix->exf_xf.cap;
    ix->exf_xf.release();
    ix->exf_xf = bigger;
  }
  const int64_t threads = (strips - strip0) * T * 64;
  hipLaunchKernelGGL(exf_layout_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ix->stream, ix->coarse, n_total, ix->d, T, ix->exf_ex,
                     strip0, strips - strip0, ix->exf_xf.as<h8v>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->exf_xf_strips = strips;
  return 0;
}

// The filter + refine path for all rows of the table.  *fell_back = 1: a candidate buffer overflowed or a query was not
// finite -- nothing was written, the caller runs the all-exact kernels.
// q_copy: NULL, or device memory for the queries when d_queries is mapped host memory (exf_prep_kernel copies them).
// h_out: mapped host memory [Q*k ids][Q*k similarities][2 verdict words], p_out the same block as the device sees it: the merge
// writes there, ONE synchronisation ends the call (four 12-us copies and a second synchronisation before).
static int exact_filter_search(freddy_gpu_index* ix, Workspace* ws, hipStream_t s, const float* d_queries, int Q, int k, int* fell_back,
                               int32_t* h_out, int32_t* p_out, float* q_copy) {
  *fell_back = 0;
  const int d = ix->d, T = (d + 15) / 16, L = k, V = pick_V(L);
  const int64_t N = ix->N;
  const bool all = (ix->tune.check_brackets & 4) != 0;
  const int64_t cap64 = all ? N : std::min<int64_t>(N, 8192);
  const int cap = (int)cap64;
  // the threshold's sample: whole 32-row strips of REAL rows (a zero-padded row would be a similarity of 0 that no row has),
  // spread evenly over the table
  const int64_t full_strips = N / 32;
  const int n_sample = (int)(std::min<int64_t>(full_strips, EXF_SAMPLE / 32) * 32);
  const int64_t sample_stride = n_sample > 0 ? std::max<int64_t>(1, full_strips / (n_sample / 32)) : 1;
  // small per-call state: [0..63] thr, [64..127] qeps, [128..191] qunscale, [192..255] cand_cnt, [256] qbad
  if (ix->exf_small.ensure(4096) || ix->exf_qfrag.ensure((size_t)2 * T * 2 * 64 * 16) ||
      ix->exf_sample.ensure(sizeof(float) * (size_t)EXF_QT * n_sample) || ix->exf_cand.ensure(sizeof(uint2) * (size_t)EXF_QT * cap) ||
      false)
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  if (!ix->viol) {
    HIP_TRY(hipMalloc((void**)&ix->viol, 4 * sizeof(int32_t)));
    HIP_TRY(hipMemset(ix->viol, 0, 4 * sizeof(int32_t)));
  }
  if (ix->exf_dirty) {   // (the handle's first call, or the one after a call that failed part-way: the verdict words and the arrival counter may be anything; a call's last workgroup leaves them at zero)
    HIP_TRY(hipMemsetAsync(ix->exf_small.as<float>() + 256, 0, 8, s));
    HIP_TRY(hipMemsetAsync(ix->viol + 3, 0, 4, s));
  }
  ix->exf_dirty = true;
  int32_t* const flags = h_out + 2 * (size_t)Q * k;
  flags[0] = flags[1] = -1;   // (before anything is enqueued: the call's last workgroup overwrites both)
  float* sm = ix->exf_small.as<float>();
  float* thr = sm; float* qeps = sm + 64; float* qunscale = sm + 128;
  int32_t* cand_cnt = reinterpret_cast<int32_t*>(sm + 192);
  int32_t* qbad = reinterpret_cast<int32_t*>(sm + 256);
  int32_t* arrived = reinterpret_cast<int32_t*>(sm + 257);
  int32_t* const o_ids = p_out;
  float* const o_sim = reinterpret_cast<float*>(p_out + (size_t)Q * k);
  int32_t* const o_flags = p_out + 2 * (size_t)Q * k;
  const size_t lds1 = (size_t)1 * T * 2 * 64 * 16, lds2 = 2 * lds1;
  for (int q0 = 0; q0 < Q; q0 += EXF_QT) {
    const int nq = std::min(EXF_QT, Q - q0);
    const int NT = nq <= 32 ? 1 : 2;
    ExfPrepArgs pa;
    pa.queries = d_queries + (size_t)q0 * d; pa.nq = nq; pa.d = d; pa.T = T; pa.xmax_norm = ix->exf_xnorm; pa.ex = ix->exf_ex;
    pa.eps_factor = exf_eps_factor(d); pa.qfrag = ix->exf_qfrag.as<h8v>(); pa.qeps = qeps; pa.qunscale = qunscale; pa.qbad = qbad;
    pa.copy_out = q_copy ? q_copy + (size_t)q0 * d : nullptr;
    timed_launch(ix, s, "exact_prep", [&] { hipLaunchKernelGGL(exf_prep_kernel, dim3(EXF_QT), dim3(256), 0, s, pa); });
    HIP_TRY(hipGetLastError());
    ExfArgs fa;
    fa.xf = ix->exf_xf.as<h8v>(); fa.n_rows = n_sample; fa.strip_stride = sample_stride; fa.T = T; fa.qfrag = ix->exf_qfrag.as<h8v>();
    fa.qunscale = qunscale; fa.sample_out = ix->exf_sample.as<float>(); fa.thr = thr; fa.cand_cnt = cand_cnt; fa.cand = ix->exf_cand.as<uint2>(); fa.cap = cap;
    auto grid_for = [&](int64_t rows) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((rows + 255) / 256, (int64_t)ix->n_cus * 2)); };
    timed_launch(ix, s, "exact_sample", [&] {
      if (NT == 1) hipLaunchKernelGGL((exf_filter_kernel<1, true>), dim3(grid_for(n_sample)), dim3(EXF_WG), lds1, s, fa);
      else hipLaunchKernelGGL((exf_filter_kernel<2, true>), dim3(grid_for(n_sample)), dim3(EXF_WG), lds2, s, fa);
    });
    HIP_TRY(hipGetLastError());
    ExfThrArgs ta;
    ta.sample = fa.sample_out; ta.n_sample = n_sample; ta.nq = nq; ta.k = k; ta.qeps = qeps; ta.qunscale = qunscale; ta.thr = thr; ta.refine_all = all ? 1 : 0; ta.cand_cnt = cand_cnt;
    timed_launch(ix, s, "exact_threshold", [&] { hipLaunchKernelGGL(exf_threshold_kernel, dim3(EXF_QT), dim3(64 * EXF_TW), 0, s, ta); });
    HIP_TRY(hipGetLastError());
    fa.n_rows = N; fa.strip_stride = 1; fa.sample_out = nullptr;
    timed_launch(ix, s, "exact_filter", [&] {
      if (NT == 1) hipLaunchKernelGGL((exf_filter_kernel<1, false>), dim3(grid_for(N)), dim3(EXF_WG), lds1, s, fa);
      else hipLaunchKernelGGL((exf_filter_kernel<2, false>), dim3(grid_for(N)), dim3(EXF_WG), lds2, s, fa);
    });
    HIP_TRY(hipGetLastError());
    ExfRefineArgs ra;
    ra.rows = ix->coarse; ra.queries = (q_copy ? q_copy : d_queries) + (size_t)q0 * d; ra.cand = fa.cand; ra.cand_cnt = cand_cnt; ra.qeps = qeps;
    ra.viol = ix->viol; ra.cap = cap; ra.d = d; ra.L = L; ra.count_checked = all ? 1 : 0;
    ra.ids = ix->ids; ra.out_ids = o_ids + (size_t)q0 * k; ra.out_sim = o_sim + (size_t)q0 * k; ra.k = k; ra.arrived = arrived; ra.total_wgs = Q;
    ra.qbad = qbad; ra.flags_out = o_flags;
    const size_t rlds = exf_refine_lds(d, V == 1 ? 1 : 2);
    timed_launch(ix, s, "exact_refine", [&] {
      switch (V) {
        case 1: hipLaunchKernelGGL((exf_refine_kernel<1>), dim3(nq), dim3(64 * EXF_TW), rlds, s, ra); break;
        default: hipLaunchKernelGGL((exf_refine_kernel<2>), dim3(nq), dim3(64 * EXF_TW), rlds, s, ra); break;
      }
    });
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (flags[0] == -1 || flags[1] == -1) return fail(FREDDY_E_HIP, "exact search: the verdict words did not arrive");
  ix->exf_dirty = false;
  if (flags[0] || flags[1]) *fell_back = 1;
  return 0;
}

extern "C" int freddy_gpu_pin_vectors(const freddy_vec_desc* t, int device, freddy_gpu_index_t** out) {
  if (!t || !out || t->d <= 0 || t->N < 0 || (t->N && (!t->ids || !t->vectors))) return fail(FREDDY_E_ARG, "bad argument");
  if (t->N > (int64_t)INT32_MAX - 64) return fail(FREDDY_E_LIMIT, "N too large for 32-bit row positions");
  for (int64_t r = 1; r < t->N; ++r)
    if (t->ids[r] <= t->ids[r - 1]) return fail(FREDDY_E_ARG, "ids must be strictly ascending (row %lld)", (long long)r);
  freddy_gpu_index* ix = new freddy_gpu_index();
  ix->kind = KIND_VEC;
  ix->d = t->d; ix->N = t->N;
  int rc = open_device(ix, device);
  if (!rc) {
    ix->n_blocks = (t->N + 63) / 64;
    const size_t xb_bytes = sizeof(float) * (size_t)std::max<int64_t>(ix->n_blocks, 1) * t->d * 64;
    if (hipMalloc((void**)&ix->xb, xb_bytes) != hipSuccess) rc = fail(FREDDY_E_NOMEM, "device allocation of %zu bytes failed", xb_bytes);
    else ix->bytes += (int64_t)xb_bytes;
    if (!rc && upload(&ix->ids, t->ids, (size_t)t->N, &ix->bytes)) rc = fail(FREDDY_E_NOMEM, "device allocation failed");
    // row-major rows go up in slices and are re-blocked on the device
    const int64_t slice = 1 << 16;
    DevBuf tmp;
    for (int64_t r0 = 0; !rc && r0 < t->N; r0 += slice) {
      const int64_t n = std::min(slice, t->N - r0);
      if (tmp.ensure(sizeof(float) * (size_t)n * t->d)) { rc = fail(FREDDY_E_NOMEM, "device allocation failed"); break; }
      if (hipMemcpy(tmp.p, t->vectors + (size_t)r0 * t->d, sizeof(float) * (size_t)n * t->d, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(FREDDY_E_HIP, "hipMemcpy failed"); break;
      }
      hipLaunchKernelGGL(block_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, ix->stream, tmp.as<float>(), nullptr, n,
                         ix->xb + (size_t)(r0 / 64) * t->d * 64, nullptr, t->d);
      if (hipStreamSynchronize(ix->stream) != hipSuccess) { rc = fail(FREDDY_E_HIP, "re-blocking kernel failed"); break; }
    }
    tmp.release();
  }
  if (!rc) {
    ix->h_ids.assign(t->ids, t->ids + t->N);
    // the source rows are only needed again for "id = ANY(...)" subsets: keep them row-major too
    if (t->N && upload(&ix->coarse, t->vectors, (size_t)t->N * t->d, &ix->bytes)) rc = fail(FREDDY_E_NOMEM, "device allocation failed");
    if (!rc && t->N) rc = exf_table_stats(ix, 0, t->N);
  }
  if (rc) { free_index(ix); return rc; }
  *out = ix;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_exact_search(freddy_gpu_index_t* ix, const float* queries, int32_t Q, int32_t k,
                                       const int32_t* subset_ids, int64_t n_subset, int32_t* out_ids, float* out_sim) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (ix->kind != KIND_VEC) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  if (Q < 0 || k <= 0 || n_subset < 0 || (n_subset > 0 && !subset_ids)) return fail(FREDDY_E_ARG, "bad sizes");
  if (Q > 0 && (!queries || !out_ids || !out_sim)) return fail(FREDDY_E_ARG, "NULL buffer");
  if (k > 4096) return fail(FREDDY_E_LIMIT, "k=%d exceeds this build's limit of 4096", k);
  if (Q == 0) return FREDDY_OK;
  HIP_TRY(hipSetDevice(ix->device));
  Workspace* ws = workspace_for(ix, ix->stream);
  hipStream_t s = ix->stream;
  const int d = ix->d, L = std::min(k, 1024), V = pick_V(L);   // (k > 1024: passes of 1024 keys, below)
  const float* xb = ix->xb;
  const int32_t* pos = nullptr;
  int64_t n_rows = ix->N, n_blocks = ix->n_blocks;
  if (subset_ids) {
    std::vector<int32_t> rows;
    rows.reserve((size_t)n_subset);
    for (int64_t i = 0; i < n_subset; ++i) {
      auto it = std::lower_bound(ix->h_ids.begin(), ix->h_ids.end(), subset_ids[i]);
      if (it != ix->h_ids.end() && *it == subset_ids[i]) rows.push_back((int32_t)(it - ix->h_ids.begin()));
    }
    std::sort(rows.begin(), rows.end());
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    n_rows = (int64_t)rows.size();
    n_blocks = (n_rows + 63) / 64;
    if (ws->w_sub_rows.ensure(sizeof(int32_t) * std::max<size_t>(rows.size(), 1)) ||
        ws->w_sub_pos.ensure(sizeof(int32_t) * (size_t)std::max<int64_t>(n_blocks, 1) * 64) ||
        ws->w_resid.ensure(sizeof(float) * (size_t)std::max<int64_t>(n_blocks, 1) * d * 64))
      return fail(FREDDY_E_NOMEM, "workspace allocation failed");
    if (n_rows) {
      HIP_TRY(hipMemcpyAsync(ws->w_sub_rows.p, rows.data(), sizeof(int32_t) * rows.size(), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(block_rows_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, ix->coarse, ws->w_sub_rows.as<int32_t>(), n_rows,
                         ws->w_resid.as<float>(), ws->w_sub_pos.as<int32_t>(), d);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(s));   // `rows` is a host temporary
    }
    xb = ws->w_resid.as<float>();
    pos = ws->w_sub_pos.as<int32_t>();
  }
  // Filter + refine (exact2.h): the whole table, k <= 32, finite rows of a supported shape; identical lists.
  const bool want_filter = !subset_ids && ix->exf_ok && k <= 32 && ix->tune.exact_filter != 0 &&
                           (ix->tune.exact_filter == 1 || n_rows >= 8192) && n_rows >= 1;
  if (want_filter) {
    // one block of mapped host memory: [lists][verdict words][the queries, when they are few]: the kernels read a handful of
    // queries where the host put them (1.2 KB each over PCIe) and write the lists where the host reads them
    const size_t n_out = (size_t)Q * k, q_bytes = sizeof(float) * (size_t)Q * d;
    const bool q_pinned = q_bytes <= (256u << 10);
    const size_t out_bytes = (n_out * 8 + 8 + 255) & ~(size_t)255, need = out_bytes + (q_pinned ? q_bytes : 0);
    if (need > ix->hio_out_cap) {
      if (ix->hio_out) (void)hipHostFree(ix->hio_out);
      ix->hio_out = nullptr; ix->hio_out_cap = 0;
      if (hipHostMalloc(&ix->hio_out, need + need / 4 + 256, hipHostMallocDefault) != hipSuccess) { ix->hio_out = nullptr; return fail(FREDDY_E_NOMEM, "pinned staging allocation failed"); }
      ix->hio_out_cap = need + need / 4 + 256;
    }
    void* dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, ix->hio_out, 0));
    int32_t* const h_out = static_cast<int32_t*>(ix->hio_out);
    const float* d_q = nullptr;
    if (ws->w_q.ensure(q_bytes)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
    if (q_pinned) {
      memcpy(static_cast<char*>(ix->hio_out) + out_bytes, queries, q_bytes);
      d_q = reinterpret_cast<const float*>(static_cast<char*>(dp) + out_bytes);
    } else {
      HIP_TRY(hipMemcpyAsync(ws->w_q.p, queries, q_bytes, hipMemcpyHostToDevice, s));
      d_q = ws->w_q.as<float>();
    }
    int fell_back = 0;
    if (int rc = exact_filter_search(ix, ws, s, d_q, Q, k, &fell_back, h_out, static_cast<int32_t*>(dp), q_pinned ? ws->w_q.as<float>() : nullptr)) return rc;
    if (!fell_back) {
      memcpy(out_ids, h_out, n_out * 4);
      memcpy(out_sim, h_out + n_out, n_out * 4);
      return FREDDY_OK;
    }
  }
  int chunk_blocks = 8;   // 512 rows per workgroup-chunk; longer chunks once the grid is large enough
  const int EX_QT = ex_qt(V, Q);
  const int qgroups = (Q + EX_QT - 1) / EX_QT;
  while ((n_blocks + chunk_blocks - 1) / chunk_blocks * (int64_t)qgroups > 8192 && chunk_blocks < 1024) chunk_blocks *= 2;
  const int nchunk = (int)std::max<int64_t>(1, (n_blocks + chunk_blocks - 1) / chunk_blocks);
  if (ws->w_q.ensure(sizeof(float) * (size_t)Q * d) || ws->w_out_ids.ensure(sizeof(int32_t) * (size_t)Q * k) ||
      ws->w_out_dist.ensure(sizeof(float) * (size_t)Q * k) ||
      ws->w_part.ensure(sizeof(u64) * (size_t)Q * nchunk * EX_WAVES * L))
    return fail(FREDDY_E_NOMEM, "workspace allocation failed");
  HIP_TRY(hipMemcpyAsync(ws->w_q.p, queries, sizeof(float) * (size_t)Q * d, hipMemcpyHostToDevice, s));
  ExactArgs ea;
  ea.xb = xb; ea.pos = pos; ea.queries = ws->w_q.as<float>(); ea.part = ws->w_part.as<u64>();
  ea.n_rows = n_rows; ea.n_blocks = (int)n_blocks; ea.chunk_blocks = chunk_blocks; ea.nchunk = nchunk; ea.Q = Q; ea.d = d; ea.L = L; ea.floor = nullptr;
  const size_t lds = (((size_t)d * EX_QT * 4 + 15) & ~(size_t)15) + (size_t)EX_WAVES * EX_QT * 64 * sizeof(u64);
  dim3 grid((unsigned)nchunk, (unsigned)qgroups);
  const int ppq = nchunk * EX_WAVES;
  if (k > 1024) {
    // lists of 1025 .. 4096 entries: 1024 keys per pass over the same rows, each pass above the last key of the one before
    if (ws->w_floor.ensure(sizeof(u64) * (size_t)Q)) return fail(FREDDY_E_NOMEM, "workspace allocation failed");
    for (int p0 = 0; p0 < k; p0 += 1024) {
      const int Lp = std::min(1024, k - p0);
      ea.L = Lp; ea.floor = p0 ? ws->w_floor.as<u64>() : nullptr;
      timed_launch(ix, s, "exact_scan", [&] {
        if (p0) hipLaunchKernelGGL((exact_scan_kernel<16, 8, true>), grid, dim3(EX_WG), lds, s, ea);
        else hipLaunchKernelGGL((exact_scan_kernel<16, 8>), grid, dim3(EX_WG), lds, s, ea);
      });
      HIP_TRY(hipGetLastError());
      timed_launch(ix, s, "exact_merge", [&] {
        hipLaunchKernelGGL((exact_merge_kernel<16>), dim3(Q), dim3(4 * 64), (size_t)4 * 64 * (16 + 1) * sizeof(u64), s, ea.part, ppq, Lp, k, ix->ids,
                           ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>(), p0, Lp, ws->w_floor.as<u64>());
      });
      HIP_TRY(hipGetLastError());
    }
  } else {
    timed_launch(ix, s, "exact_scan", [&] {
      switch (V) {
        case 1: if (EX_QT == 16) hipLaunchKernelGGL((exact_scan_kernel<1, 16>), grid, dim3(EX_WG), lds, s, ea);
                else hipLaunchKernelGGL((exact_scan_kernel<1, 8>), grid, dim3(EX_WG), lds, s, ea);
                break;
        case 2: if (EX_QT == 16) hipLaunchKernelGGL((exact_scan_kernel<2, 16>), grid, dim3(EX_WG), lds, s, ea);
                else hipLaunchKernelGGL((exact_scan_kernel<2, 8>), grid, dim3(EX_WG), lds, s, ea);
                break;
        case 4: if (EX_QT == 16) hipLaunchKernelGGL((exact_scan_kernel<4, 16>), grid, dim3(EX_WG), lds, s, ea);
                else hipLaunchKernelGGL((exact_scan_kernel<4, 8>), grid, dim3(EX_WG), lds, s, ea);
                break;
        case 8: hipLaunchKernelGGL((exact_scan_kernel<8, 8>), grid, dim3(EX_WG), lds, s, ea); break;
        default: hipLaunchKernelGGL((exact_scan_kernel<16, 8>), grid, dim3(EX_WG), lds, s, ea); break;
      }
    });
    HIP_TRY(hipGetLastError());
    timed_launch(ix, s, "exact_merge", [&] {
      switch (V) {
        case 1: hipLaunchKernelGGL((exact_merge_kernel<1>), dim3(Q), dim3(16 * 64), (size_t)16 * 64 * (1 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
        case 2: hipLaunchKernelGGL((exact_merge_kernel<2>), dim3(Q), dim3(16 * 64), (size_t)16 * 64 * (2 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
        case 4: hipLaunchKernelGGL((exact_merge_kernel<4>), dim3(Q), dim3(16 * 64), (size_t)16 * 64 * (4 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
        case 8: hipLaunchKernelGGL((exact_merge_kernel<8>), dim3(Q), dim3(8 * 64), (size_t)8 * 64 * (8 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
        default: hipLaunchKernelGGL((exact_merge_kernel<16>), dim3(Q), dim3(4 * 64), (size_t)4 * 64 * (16 + 1) * sizeof(u64), s, ea.part, ppq, L, k, ix->ids, ws->w_out_ids.as<int32_t>(), ws->w_out_dist.as<float>()); break;
      }
    });
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(out_ids, ws->w_out_ids.p, sizeof(int32_t) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(out_sim, ws->w_out_dist.p, sizeof(float) * (size_t)Q * k, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return FREDDY_OK;
}

// The kernels of this unit that want more than the default 64 KiB of dynamic LDS (a per-device function attribute).
int raise_lds_limits_exact(int device) {
  static std::mutex mu;
  static std::vector<char> done;
  std::lock_guard<std::mutex> g(mu);
  if ((size_t)device < done.size() && done[(size_t)device]) return 0;
  const void* kernels[] = {
      (const void*)&exf_filter_kernel<1, false>, (const void*)&exf_filter_kernel<2, false>, (const void*)&exf_filter_kernel<1, true>,
      (const void*)&exf_filter_kernel<2, true>};
  for (const void* k : kernels)
    HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  if (done.size() <= (size_t)device) done.resize((size_t)device + 1, 0);
  done[(size_t)device] = 1;
  return 0;
}
