// build.hip -- index build: encoding, insert_batch's quantisation, Lloyd k-means.
#include "internal.h"

#include "kernels.h"

// ---------------------------------------------------------------------------------------
// index build: encoding
// ---------------------------------------------------------------------------------------
// limit_coarse / limit_code < +inf: insert_batch's searches start from that distance (strict "<"); *n_too_far
// counts the (vector[, position]) pairs with no centroid nearer than the limit.
static int encode_impl(const freddy_encode_desc* t, int device, const float* vectors, int64_t N, int32_t* out_cell,
                       int16_t* out_codes, float limit_coarse, float limit_code, int32_t* n_too_far) {
  if (!t || !t->codebook || !out_codes || N < 0 || (N > 0 && !vectors)) return fail(FREDDY_E_ARG, "NULL argument");
  if (t->d <= 0 || t->m <= 0 || t->K <= 0 || t->d % t->m) return fail(FREDDY_E_ARG, "bad shape d=%d m=%d K=%d", t->d, t->m, t->K);
  if (t->K > 32767) return fail(FREDDY_E_LIMIT, "K=%d does not fit an int16 code", t->K);
  if ((t->C > 0) != (t->coarse != nullptr)) return fail(FREDDY_E_ARG, "coarse and C must be given together");
  if (t->C > 0 && !out_cell) return fail(FREDDY_E_ARG, "out_cell is required with a coarse quantizer");
  if (N == 0) return FREDDY_OK;
  HIP_TRY(hipSetDevice(device));
  const int d = t->d, m = t->m, K = t->K, S = d / m, C = t->C;
  const int Cpad = ((C + 63) / 64) * 64;
  std::vector<float> cbT((size_t)m * S * K);
  for (int p = 0; p < m; ++p)
    for (int c = 0; c < K; ++c)
      for (int i = 0; i < S; ++i) cbT[((size_t)p * S + i) * K + c] = t->codebook[((size_t)p * K + c) * S + i];
  std::vector<float> cT;
  if (C) {
    cT.assign((size_t)d * Cpad, 0.0f);
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < d; ++i) cT[(size_t)i * Cpad + c] = t->coarse[(size_t)c * d + i];
  }
  const int64_t chunk = std::min<int64_t>(N, 1 << 16);
  float *d_cbT = nullptr, *d_cT = nullptr, *d_coarse = nullptr, *d_vec = nullptr, *d_res = nullptr;
  int32_t *d_cell = nullptr, *d_far = nullptr;
  int16_t* d_codes = nullptr;
  int rc = FREDDY_OK;
  hipStream_t s = nullptr;
  auto cleanup = [&] {
    void* ptrs[] = {d_cbT, d_cT, d_coarse, d_vec, d_res, d_cell, d_codes, d_far};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (s) (void)hipStreamDestroy(s);
  };
#define ENC_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) { cleanup(); return fail(FREDDY_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } \
  } while (0)
  ENC_TRY(hipStreamCreate(&s));
  ENC_TRY(hipMalloc((void**)&d_cbT, sizeof(float) * cbT.size()));
  ENC_TRY(hipMalloc((void**)&d_vec, sizeof(float) * (size_t)chunk * d));
  ENC_TRY(hipMalloc((void**)&d_codes, sizeof(int16_t) * (size_t)chunk * m));
  ENC_TRY(hipMemcpyAsync(d_cbT, cbT.data(), sizeof(float) * cbT.size(), hipMemcpyHostToDevice, s));
  if (n_too_far) {
    ENC_TRY(hipMalloc((void**)&d_far, sizeof(int32_t)));
    ENC_TRY(hipMemsetAsync(d_far, 0, sizeof(int32_t), s));
  }
  if (C) {
    ENC_TRY(hipMalloc((void**)&d_cT, sizeof(float) * cT.size()));
    ENC_TRY(hipMalloc((void**)&d_coarse, sizeof(float) * (size_t)C * d));
    ENC_TRY(hipMalloc((void**)&d_res, sizeof(float) * (size_t)chunk * d));
    ENC_TRY(hipMalloc((void**)&d_cell, sizeof(int32_t) * (size_t)chunk));
    ENC_TRY(hipMemcpyAsync(d_cT, cT.data(), sizeof(float) * cT.size(), hipMemcpyHostToDevice, s));
    ENC_TRY(hipMemcpyAsync(d_coarse, t->coarse, sizeof(float) * (size_t)C * d, hipMemcpyHostToDevice, s));
  }
  for (int64_t i0 = 0; i0 < N; i0 += chunk) {
    const int n = (int)std::min<int64_t>(chunk, N - i0);
    ENC_TRY(hipMemcpyAsync(d_vec, vectors + (size_t)i0 * d, sizeof(float) * (size_t)n * d, hipMemcpyHostToDevice, s));
    const float* src = d_vec;
    if (C) {
      hipLaunchKernelGGL(assign_coarse_kernel, dim3((unsigned)n), dim3(64), 0, s, (const float*)d_vec, (const float*)d_cT, d_cell, n, C, Cpad, d, limit_coarse, d_far);
      hipLaunchKernelGGL(residual_kernel, dim3((unsigned)n), dim3(WG), 0, s, (const float*)d_vec, (const float*)d_coarse,
                         (const int32_t*)d_cell, (const int32_t*)nullptr, d_res, d, S, S);
      src = d_res;
    }
    const int ipw = 64;
    const dim3 grid((unsigned)m, (unsigned)((n + ipw - 1) / ipw));
    if (S == 25) hipLaunchKernelGGL((encode_pq_kernel<25, 4>), grid, dim3(WG), 0, s, src, (const float*)d_cbT, d_codes, n, ipw, m, K, d, S, limit_code, d_far);
    else if (S == 10) hipLaunchKernelGGL((encode_pq_kernel<10, 4>), grid, dim3(WG), 0, s, src, (const float*)d_cbT, d_codes, n, ipw, m, K, d, S, limit_code, d_far);
    else hipLaunchKernelGGL((encode_pq_kernel<0, 4>), grid, dim3(WG), 0, s, src, (const float*)d_cbT, d_codes, n, ipw, m, K, d, S, limit_code, d_far);
    ENC_TRY(hipGetLastError());
    ENC_TRY(hipMemcpyAsync(out_codes + (size_t)i0 * m, d_codes, sizeof(int16_t) * (size_t)n * m, hipMemcpyDeviceToHost, s));
    if (C) ENC_TRY(hipMemcpyAsync(out_cell + i0, d_cell, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    ENC_TRY(hipStreamSynchronize(s));
  }
  if (n_too_far) ENC_TRY(hipMemcpy(n_too_far, d_far, sizeof(int32_t), hipMemcpyDeviceToHost));
#undef ENC_TRY
  cleanup();
  return rc;
}

extern "C" int freddy_gpu_encode(const freddy_encode_desc* t, int device, const float* vectors, int64_t N, int32_t* out_cell,
                                 int16_t* out_codes) {
  const float inf = std::numeric_limits<float>::infinity();
  return encode_impl(t, device, vectors, N, out_cell, out_codes, inf, inf, nullptr);
}

// insert_batch, quantisation of the new vectors (freddy.c:1557-1623): codes against the PQ codebook, coarse
// cell (from minDistCoarse = 100) + codes of the residual against the residual codebook, codes against the ivpq
// codebook, and the two coarse codes of the multi index (from MAX_DIST = 1000).  Every search is the exact
// 1-NN by squareDistance with the first entry winning ties, as updateCodebook's strict "<" scan.
extern "C" int freddy_gpu_insert_quantize(const freddy_insert_desc* t, int device, const float* vectors, int64_t n,
                                          int16_t* pq_codes, int32_t* coarse_id, int16_t* residual_codes, int16_t* ivpq_codes,
                                          int16_t* coarse_multi_codes) {
  if (!t || n < 0 || (n > 0 && !vectors)) return fail(FREDDY_E_ARG, "NULL argument");
  const float inf = std::numeric_limits<float>::infinity();
  int32_t far = 0, far_total = 0;
  if (t->pq_codebook) {
    if (!pq_codes) return fail(FREDDY_E_ARG, "pq_codes is required with a PQ codebook");
    freddy_encode_desc e = {t->d, t->pq_m, t->pq_K, t->pq_codebook, 0, nullptr};
    if (int rc = encode_impl(&e, device, vectors, n, nullptr, pq_codes, inf, 100.0f, &far)) return rc;
    far_total += far;
  }
  if (t->residual_codebook) {
    if (!t->coarse || !coarse_id || !residual_codes) return fail(FREDDY_E_ARG, "the residual codebook needs the coarse quantizer and both outputs");
    freddy_encode_desc e = {t->d, t->res_m, t->res_K, t->residual_codebook, t->C, t->coarse};
    if (int rc = encode_impl(&e, device, vectors, n, coarse_id, residual_codes, 100.0f, 100.0f, &far)) return rc;
    far_total += far;
  }
  if (t->ivpq_codebook) {
    if (!ivpq_codes) return fail(FREDDY_E_ARG, "ivpq_codes is required with an ivpq codebook");
    freddy_encode_desc e = {t->d, t->ivpq_m, t->ivpq_K, t->ivpq_codebook, 0, nullptr};
    if (int rc = encode_impl(&e, device, vectors, n, nullptr, ivpq_codes, inf, 100.0f, &far)) return rc;
    far_total += far;
  }
  if (t->coarse_multi) {
    if (!coarse_multi_codes) return fail(FREDDY_E_ARG, "coarse_multi_codes is required with a multi-index coarse quantizer");
    freddy_encode_desc e = {t->d, t->multi_positions, t->multi_codes, t->coarse_multi, 0, nullptr};
    if (int rc = encode_impl(&e, device, vectors, n, nullptr, coarse_multi_codes, inf, inf, nullptr)) return rc;
  }
  if (far_total)
    return fail(FREDDY_E_ARG, "%d (vector, position) pairs are 100 or farther from every centroid: insert_batch is undefined for them "
                "(index_utils.c:925-939 leaves the code uninitialised)", far_total);
  return FREDDY_OK;
}

// ---------------------------------------------------------------------------------------
// index build: quantizer training
// ---------------------------------------------------------------------------------------
extern "C" int freddy_gpu_kmeans(int device, const float* vectors, int64_t n, int32_t d, int32_t k, int32_t iters,
                                 const int32_t* init_rows, float* centroids, int32_t* assign_out) {
  if (!vectors || !centroids || n <= 0 || d <= 0 || k <= 0 || iters < 0) return fail(FREDDY_E_ARG, "bad argument");
  if (d > 1024) return fail(FREDDY_E_LIMIT, "d=%d exceeds this build's limit of 1024 dimensions", d);
  if (n > INT32_MAX) return fail(FREDDY_E_LIMIT, "too many training vectors");
  HIP_TRY(hipSetDevice(device));
  const int kpad = (k + 63) / 64 * 64;
  std::vector<float> init((size_t)k * d);
  for (int c = 0; c < k; ++c) {
    const int64_t r = init_rows ? init_rows[c] : c % n;
    if (r < 0 || r >= n) return fail(FREDDY_E_ARG, "init_rows[%d] = %lld is not a training row", c, (long long)r);
    memcpy(&init[(size_t)c * d], vectors + (size_t)r * d, sizeof(float) * (size_t)d);
  }
  float *d_vec = nullptr, *d_cent = nullptr, *d_centT = nullptr;
  int32_t* d_assign = nullptr;
  hipStream_t s = nullptr;
  int rc = FREDDY_OK;
  auto cleanup = [&] {
    void* ptrs[] = {d_vec, d_cent, d_centT, d_assign};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (s) (void)hipStreamDestroy(s);
  };
#define KM_TRY(expr)                                                                           \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) { cleanup(); return fail(FREDDY_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } \
  } while (0)
  KM_TRY(hipStreamCreate(&s));
  KM_TRY(hipMalloc((void**)&d_vec, sizeof(float) * (size_t)n * d));
  KM_TRY(hipMalloc((void**)&d_cent, sizeof(float) * (size_t)k * d));
  KM_TRY(hipMalloc((void**)&d_centT, sizeof(float) * (size_t)kpad * d));
  KM_TRY(hipMalloc((void**)&d_assign, sizeof(int32_t) * (size_t)n));
  KM_TRY(hipMemcpyAsync(d_vec, vectors, sizeof(float) * (size_t)n * d, hipMemcpyHostToDevice, s));
  KM_TRY(hipMemcpyAsync(d_cent, init.data(), sizeof(float) * init.size(), hipMemcpyHostToDevice, s));
  const float inf = std::numeric_limits<float>::infinity();
  for (int it = 0; it <= iters; ++it) {
    hipLaunchKernelGGL(kmeans_transpose_kernel, dim3((unsigned)(((size_t)d * kpad + 255) / 256)), dim3(256), 0, s, (const float*)d_cent, d_centT, k, kpad, d);
    hipLaunchKernelGGL(assign_coarse_kernel, dim3((unsigned)n), dim3(64), 0, s, (const float*)d_vec, (const float*)d_centT, d_assign, (int)n, k, kpad, d,
                       inf, (int32_t*)nullptr);
    if (it == iters) break;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)k), dim3(256), 0, s, (const float*)d_vec, (const int32_t*)d_assign, n, d, d_cent);
    KM_TRY(hipGetLastError());
  }
  KM_TRY(hipGetLastError());
  KM_TRY(hipMemcpyAsync(centroids, d_cent, sizeof(float) * (size_t)k * d, hipMemcpyDeviceToHost, s));
  if (assign_out) KM_TRY(hipMemcpyAsync(assign_out, d_assign, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
  KM_TRY(hipStreamSynchronize(s));
#undef KM_TRY
  cleanup();
  return rc;
}

