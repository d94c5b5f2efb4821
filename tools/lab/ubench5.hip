// v_permlane16_swap / v_permlane32_swap (gfx950): what each lane holds afterwards, to derive lane^16 / lane^32
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned v = threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  auto r2 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  out[threadIdx.x * 4 + 0] = r[0]; out[threadIdx.x * 4 + 1] = r[1];
  out[threadIdx.x * 4 + 2] = r2[0]; out[threadIdx.x * 4 + 3] = r2[1];
}
int main() {
  unsigned *d, h[256];
  (void)hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63})
    printf("lane %2d: swap16 -> (%2u, %2u)   swap32 -> (%2u, %2u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
