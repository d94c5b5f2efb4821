// LDS read cost microbenchmark (gfx950): cycles per ds_read_b128 per CU when every lane reads the SAME
// 16 bytes (broadcast) vs a lane-private 16 bytes, 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float4 acc = {0, 0, 0, 0};
  int base = (MODE == 0) ? 0 : lane * 4;                       // 0: broadcast, 1: contiguous per lane
  if (MODE == 2) base = lane * 16;                             // 2: 64-byte stride (the slab-row pattern)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float4 v = *reinterpret_cast<const float4*>(&lds[(base + ((it * 16 + u) & 63) * 256) & 16380]);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE>
void run(float* out, const char* name) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double reads_per_cu = (double)iters * 16 * 8;   // wave-level ds_read_b128 per CU
  printf("%-28s %.3f ms -> %.2f ns per wave-level ds_read_b128 per CU (%.1f cycles @2.1GHz)\n", name, ms, ms * 1e6 / reads_per_cu,
         ms * 1e6 / reads_per_cu * 2.1);
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  run<0>(out, "broadcast (same address)");
  run<1>(out, "contiguous 16 B per lane");
  run<2>(out, "64-byte stride per lane");
  return 0;
}
