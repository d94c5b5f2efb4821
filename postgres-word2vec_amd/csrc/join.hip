// join.hip -- pin_ivpq and knn_join (ivpq_search_in.c:61-699); the kernels and the host loop are in join.h.
#include "internal.h"

#include "join.h"

extern "C" int freddy_gpu_pin_ivpq(const freddy_ivpq_desc* t, int device, freddy_gpu_index_t** out) {
  if (!t || !out || !t->codebook || !t->coarse || !t->stats || (t->N && (!t->ids || !t->codes || !t->coarse_id)))
    return fail(FREDDY_E_ARG, "NULL argument");
  if (t->d <= 0 || t->m <= 0 || t->K <= 0 || t->d % t->m) return fail(FREDDY_E_ARG, "bad d/m/K");
  if (t->coarse_positions != 2) return fail(FREDDY_E_LIMIT, "only 2 coarse positions are supported (as in the reference, index_utils.c:322)");
  if (t->coarse_codes <= 0 || t->d % 2) return fail(FREDDY_E_ARG, "bad coarse multi-index shape");
  freddy_gpu_index* ix = new freddy_gpu_index();
  ix->kind = KIND_IVPQ;
  ix->d = t->d; ix->m = t->m; ix->K = t->K; ix->S = t->d / t->m; ix->N = t->N;
  int rc = open_device(ix, device);
  if (!rc) {
    rc = join_pin(&ix->join, t, &ix->bytes);
    ix->join.host_traversal = env_int("FREDDY_GPU_JOIN_HOST_TRAVERSAL", 0) != 0;
    if (rc) rc = fail(rc, "%s", join_error());
  }
  if (rc) { free_index(ix); return rc; }
  *out = ix;
  return FREDDY_OK;
}

// ---------------------------------------------------------------------------------------
// kNN-join (ivpq_search_in): host loop in join.h
// ---------------------------------------------------------------------------------------
extern "C" int freddy_gpu_knn_join(freddy_gpu_index_t* ix, const float* queries, int32_t Q, int32_t k,
                                   const int32_t* target_ids, int64_t n_targets, int32_t alpha, int32_t pvf,
                                   int32_t method, int32_t use_target_lists, float confidence, int32_t double_threshold,
                                   int32_t* out_ids, float* out_dist, int32_t* iterations_out) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (ix->kind != KIND_IVPQ) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  if (Q < 0 || k <= 0 || n_targets < 0) return fail(FREDDY_E_ARG, "bad sizes");
  if (Q > 0 && (!queries || !out_ids || !out_dist)) return fail(FREDDY_E_ARG, "NULL buffer");
  if (n_targets > 0 && !target_ids) return fail(FREDDY_E_ARG, "NULL target ids");
  HIP_TRY(hipSetDevice(ix->device));
  int rc = join_run(&ix->join, ix->stream, queries, Q, k, target_ids, n_targets, alpha, pvf, method,
                    use_target_lists, confidence, double_threshold, out_ids, out_dist, iterations_out);
  if (rc) return fail(rc, "%s", join_error());
  return FREDDY_OK;
}

extern "C" int freddy_gpu_last_track(const freddy_gpu_index_t* ix, freddy_track* out) {
  if (!ix || !out) return fail(FREDDY_E_ARG, "NULL argument");
  if (ix->kind != KIND_IVPQ) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  *out = ix->join.track;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_last_track_sized(const freddy_gpu_index_t* ix, void* out, size_t out_size) {
  if (!ix || !out) return fail(FREDDY_E_ARG, "NULL argument");
  if (ix->kind != KIND_IVPQ) return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
  const size_t n = std::min(out_size, sizeof(freddy_track));
  memcpy(out, &ix->join.track, n);
  return (int)n;
}

// The kernels of this unit that want more than the default 64 KiB of dynamic LDS (a per-device function attribute).
int raise_lds_limits_join(int device) {
  static std::mutex mu;
  static std::vector<char> done;
  std::lock_guard<std::mutex> g(mu);
  if ((size_t)device < done.size() && done[(size_t)device]) return 0;
  const void* kernels[] = {
      (const void*)&join_query_kernel<1>, (const void*)&join_query_kernel<2>,
      (const void*)&join_query_kernel<4>, (const void*)&join_query_kernel<8>, (const void*)&join_query_kernel<16>, (const void*)&join_query_kernel<16, true>};
  for (const void* k : kernels)
    HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  if (done.size() <= (size_t)device) done.resize((size_t)device + 1, 0);
  done[(size_t)device] = 1;
  return 0;
}
