// Dependent-chain latency / throughput of v_pk_{add,mul}_f32 and scalar v_{sub,mul,add}_f32 on gfx950:
// NCH independent accumulate chains per wave, WPS waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int NCH, int PK>
__global__ __launch_bounds__(64) void k(float* out, float a0, int iters) {
  const float t = threadIdx.x * 1e-3f;
  if (PK) {
    v2f acc[NCH], cb[NCH];
    for (int i = 0; i < NCH; ++i) { acc[i] = v2f{0.f, 0.f}; cb[i] = v2f{t + i, t - i}; }
    for (int it = 0; it < iters; ++it) {
      const float r = a0 + it;
#pragma unroll
      for (int i = 0; i < NCH; ++i) { v2f d = v2f{r, r} - cb[i]; v2f p = d * d; acc[i] = acc[i] + p; }
    }
    float s = 0; for (int i = 0; i < NCH; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
  } else {
    float acc[NCH], cb[NCH];
    for (int i = 0; i < NCH; ++i) { acc[i] = 0.f; cb[i] = t + i; }
    for (int it = 0; it < iters; ++it) {
      const float r = a0 + it;
#pragma unroll
      for (int i = 0; i < NCH; ++i) { float d = r - cb[i]; float p = d * d; acc[i] = acc[i] + p; }
    }
    float s = 0; for (int i = 0; i < NCH; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
  }
}
template <int NCH, int PK>
void run(float* out, int wps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 20000, grid = 256 * 4 * wps;
  hipLaunchKernelGGL((k<NCH, PK>), dim3(grid), dim3(64), 0, 0, out, 1.0f, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<NCH, PK>), dim3(grid), dim3(64), 0, 0, out, 1.0f, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double instr_per_wave = (double)iters * NCH * 3;
  const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * wps);
  printf("pk=%d chains=%d waves/SIMD=%d: %.3f ms  -> %.2f ns per wave-instr per SIMD (%.1f cycles @2.4GHz)\n", PK, NCH, wps, ms,
         ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
}
int main() {
  float* out; (void)hipMalloc(&out, 4 * 64 * 256 * 4 * 8);
  for (int wps : {1, 2, 4}) {
    run<1, 1>(out, wps); run<2, 1>(out, wps); run<4, 1>(out, wps); run<8, 1>(out, wps);
    run<1, 0>(out, wps); run<2, 0>(out, wps); run<4, 0>(out, wps); run<8, 0>(out, wps);
  }
  return 0;
}
