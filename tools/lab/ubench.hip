// VALU micro-benchmark: scalar v_sub/v_mul/v_add chains vs packed v_pk_* on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, float a0, int iters) {
  const float t = threadIdx.x * 1e-3f;
  if (PK) {
    float2v acc[8], cb[8];
    for (int i = 0; i < 8; ++i) { acc[i] = 0.f; cb[i] = float2v{t + i, t - i}; }
    for (int it = 0; it < iters; ++it) {
      const float r = a0 + it;
#pragma unroll
      for (int i = 0; i < 8; ++i) { float2v d = float2v{r, r} - cb[i]; float2v p = d * d; acc[i] = acc[i] + p; }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    float acc[16], cb[16];
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; cb[i] = t + i; }
    for (int it = 0; it < iters; ++it) {
      const float r = a0 + it;
#pragma unroll
      for (int i = 0; i < 16; ++i) { float d = r - cb[i]; float p = d * d; acc[i] = acc[i] + p; }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 8192);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4096, grid = 8192;
  for (int pk = 0; pk < 2; ++pk) for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    if (pk) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, 1.0f, iters);
    else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, 1.0f, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)grid * 256 * iters * 16 * 3;
    printf("pk=%d: %.3f ms  %.2f T lane-ops/s (sub+mul+add counted 1 each)\n", pk, ms, ops / ms / 1e9);
  }
  return 0;
}
