// GPU-box micro-benchmark: coarse_approx_kernel + probe_plan2_kernel (csrc/coarse.h) in isolation on clustered data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/lab/ubench_plan.hip -o tools/lab/ubench_plan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../postgres-word2vec_amd/csrc/coarse.h"
using namespace freddy;
static float rnd() { return (float)rand() / RAND_MAX - 0.5f; }
int main(int argc, char** argv) {
  const int Q = argc > 4 ? atoi(argv[4]) : 1024, C = 1000, Cpad = 1024, d = 300, dp = 320, W = argc > 1 ? atoi(argv[1]) : 10;
  const int nit = dp / 8;
  std::vector<float> q((size_t)Q * d), c((size_t)C * d), cf((size_t)Cpad * dp, 0.f), cn(Cpad, 0.f);
  // unit-norm clustered data: centroids = normalised random, queries = centroid + noise, normalised
  double cmax2 = 0;
  for (int j = 0; j < C; ++j) {
    double n2 = 0; for (int i = 0; i < d; ++i) { c[(size_t)j * d + i] = rnd(); n2 += c[(size_t)j * d + i] * c[(size_t)j * d + i]; }
    const float s = 0.8f / std::sqrt(n2); n2 = 0;
    for (int i = 0; i < d; ++i) { float& v = c[(size_t)j * d + i]; v *= s; n2 += (double)v * v;
      cf[((((size_t)(j >> 5) * nit + (i >> 3)) * 64) + (size_t)((i >> 2) & 1) * 32 + (j & 31)) * 4 + (i & 3)] = v; }
    cn[j] = (float)n2; cmax2 = std::max(cmax2, n2);
  }
  for (int x = 0; x < Q; ++x) {
    const int j = rand() % C; double n2 = 0;
    for (int i = 0; i < d; ++i) { float v = c[(size_t)j * d + i] + 0.08f * rnd(); q[(size_t)x * d + i] = v; n2 += (double)v * v; }
    const float s = 1.0f / std::sqrt(n2);
    for (int i = 0; i < d; ++i) q[(size_t)x * d + i] *= s;
  }
  std::vector<int32_t> list_off(C + 1); for (int j = 0; j <= C; ++j) list_off[j] = j * 3000;
  float *dq, *dc, *dcf, *dn, *dist, *qn, *idist; int32_t *lo, *icell, *iquery, *rows, *ccount, *citems, *viol; uint32_t* used;
  (void)hipMalloc(&dq, q.size() * 4); (void)hipMalloc(&dc, c.size() * 4); (void)hipMalloc(&dcf, cf.size() * 4); (void)hipMalloc(&dn, Cpad * 4);
  (void)hipMalloc(&dist, (size_t)Q * Cpad * 4); (void)hipMalloc(&qn, Q * 4); (void)hipMalloc(&idist, (size_t)Q * W * 4);
  (void)hipMalloc(&lo, (C + 1) * 4); (void)hipMalloc(&icell, (size_t)Q * W * 4); (void)hipMalloc(&iquery, (size_t)Q * W * 4);
  (void)hipMalloc(&rows, Q * 4); (void)hipMalloc(&ccount, C * 2 * 4); (void)hipMalloc(&citems, (size_t)C * Q * 4); (void)hipMalloc(&viol, 16);
  (void)hipMalloc(&used, (size_t)Q * 32 * 4);
  (void)hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dcf, cf.data(), cf.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dn, cn.data(), Cpad * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(lo, list_off.data(), (C + 1) * 4, hipMemcpyHostToDevice);
  (void)hipMemset(viol, 0, 16);
  ZeroArgs z; for (int i = 0; i < 5; ++i) { z.p[i] = nullptr; z.n[i] = 0; }
  z.p[0] = used; z.n[0] = Q * 32; z.p[1] = (uint32_t*)ccount; z.n[1] = C * 2;
  (void)hipFuncSetAttribute((const void*)&coarse_approx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = (size_t)(64 * (dp + 4) + 128) * 4;
  Plan2Args g;
  g.p.dist = dist; g.p.active = nullptr; g.p.list_off = lo; g.p.used = used; g.p.item_cell = icell; g.p.item_query = iquery;
  g.p.item_dist = idist; g.p.round_rows = rows; g.p.cell_count = ccount; g.p.cell_items = citems; g.p.cell_cap = Q;
  g.p.n_active = Q; g.p.Cpad = Cpad; g.p.C = C; g.p.W = W; g.p.used_words = 32; g.p.cell_limit = 100.0f;
  g.queries = dq; g.coarse = dc; g.qn2 = qn; g.item_dist = idist; g.violations = (int32_t*)viol; g.cmax = (float)(std::sqrt(cmax2) * 1.000001);
  g.d = d; g.refine_all = 0;
  long long* dprof; (void)hipMalloc(&dprof, (size_t)Q * 16 * 8); (void)hipMemset(dprof, 0, (size_t)Q * 16 * 8); g.prof = argc > 3 ? dprof : nullptr;
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  const int abl = argc > 2 ? atoi(argv[2]) : 0;
  float t_c = 0, t_p = 0;
  for (int it = 0; it < 40; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(coarse_approx_kernel, dim3(Cpad / 128, (Q + COARSE_TQ - 1) / COARSE_TQ), dim3(256), (size_t)(COARSE_TQ * (dp + 4) + 128) * 4, 0, dq, dcf, dn, dist, qn, Q, Cpad, d, dp, z);
    hipEventRecord(e1, 0);
    if (abl == 0) hipLaunchKernelGGL(probe_plan2_kernel<0>, dim3(Q), dim3(64 * PLAN2_NW), 0, 0, g);
    else if (abl == 1) hipLaunchKernelGGL(probe_plan2_kernel<1>, dim3(Q), dim3(64 * PLAN2_NW), 0, 0, g);
    else if (abl == 2) hipLaunchKernelGGL(probe_plan2_kernel<2>, dim3(Q), dim3(64 * PLAN2_NW), 0, 0, g);
    else if (abl == 3) hipLaunchKernelGGL(probe_plan2_kernel<3>, dim3(Q), dim3(64 * PLAN2_NW), 0, 0, g);
    else if (abl == 4) hipLaunchKernelGGL(probe_plan2_kernel<4>, dim3(Q), dim3(64 * PLAN2_NW), 0, 0, g);
    else hipLaunchKernelGGL(probe_plan2_kernel<5>, dim3(Q), dim3(64 * PLAN2_NW), 0, 0, g);
    hipEventRecord(e2, 0); hipEventSynchronize(e2);
    float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
    if (it >= 10) { t_c += a; t_p += b; }
  }
  printf("abl=%d ", abl);
  if (g.prof) { std::vector<long long> hq((size_t)Q * 16); (void)hipMemcpy(hq.data(), dprof, hq.size() * 8, hipMemcpyDeviceToHost);
    double hp[16] = {0}; for (int x = 0; x < Q; ++x) for (int i = 0; i < 16; ++i) hp[i] += (double)hq[(size_t)x * 16 + i];
    const char* nm[] = {"A loads+min", "B threshold", "C candidates", "D rows+squares", "E sums", "F list", "G atomics+stores"};
    for (int i = 0; i < 7; ++i) printf("  %-22s %8.0f cycles per query\n", nm[i], hp[i] / 40.0 / Q); }
  int32_t hv[4]; (void)hipMemcpy(hv, viol, 16, hipMemcpyDeviceToHost);
  printf("W=%d coarse %.2f us, plan %.2f us (%s); violations %d; debug counter (candidates) %d -> %.1f per query-launch\n", W, t_c * 1000 / 30, t_p * 1000 / 30,
         hipGetErrorString(hipGetLastError()), hv[2], hv[3], hv[3] / 40.0 / Q);
  return 0;
}
