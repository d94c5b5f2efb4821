// tools/lab/ubench_cumask.hip -- which CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) get on this chip?
// Every workgroup records (XCC_ID, SE, SH, CU) from the hardware registers; the host prints the distinct CUs per XCC
// for a few mask patterns, and whether a kernel on the complement mask runs BESIDE a long kernel on the mask.
//   hipcc --offload-arch=gfx950 -O2 -o tools/lab/ubench_cumask tools/lab/ubench_cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>
#include <map>
#include <chrono>

__global__ void where_kernel(uint32_t* out, int spin) {
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
  long long t0 = clock64();
  while (clock64() - t0 < spin) { }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
__global__ void spin_kernel(long long* out, long long cycles) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (threadIdx.x == 0) out[blockIdx.x] = wall_clock64();
}

static void report(const char* name, const std::vector<uint32_t>& h, int n) {
  std::map<int, std::set<int>> per;
  for (int i = 0; i < n; ++i) {
    const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per[(int)xcc].insert(se * 32 + sh * 16 + cu);
  }
  int tot = 0;
  printf("%-28s", name);
  for (auto& kv : per) { printf(" x%d:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
  printf("  total %d CUs\n", tot);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d\n", p.multiProcessorCount);
  const int n = 4096;
  uint32_t* d; hipMalloc(&d, n * 8);
  std::vector<uint32_t> h(2 * n);
  auto run = [&](const char* name, hipStream_t s) {
    hipMemsetAsync(d, 0, n * 8, s);
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(1024), 0, s, d, 20000);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    report(name, h, n);
  };
  run("unmasked", nullptr);
  struct Pat { const char* name; uint32_t w[8]; };
  Pat pats[] = {
      {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 0..63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}},
      {"bits 32..255", {0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
      {"every 8th bit", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
      {"bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 224..255", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu}},
  };
  for (auto& pt : pats) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, pt.w);
    if (e != hipSuccess) { printf("%s: create failed: %s\n", pt.name, hipGetErrorString(e)); continue; }
    run(pt.name, s);
    hipStreamDestroy(s);
  }
  // concurrency: a long spin kernel on mask A (one workgroup per CU of A, persistent style) and a short kernel on mask B
  {
    uint32_t a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0xffffffffu; b[i] = 0; }
    a[0] = 0; b[0] = 0xffffffffu;
    hipStream_t sa, sb; hipExtStreamCreateWithCUMask(&sa, 8, a); hipExtStreamCreateWithCUMask(&sb, 8, b);
    long long* dl; hipMalloc(&dl, 8 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(spin_kernel, dim3(224), dim3(1024), 65536, sa, dl, 100000000LL / 10 * 2);   // ~200 us at 100 MHz wall clock
      auto t0 = std::chrono::steady_clock::now();
      hipEventRecord(e0, sb);
      hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, sb, dl + 1024, 1000);   // 10 us
      hipEventRecord(e1, sb);
      hipStreamSynchronize(sb);
      auto t1 = std::chrono::steady_clock::now();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      hipStreamSynchronize(sa);
      auto t2 = std::chrono::steady_clock::now();
      printf("short kernel on the complement mask while a 2 ms kernel holds the other CUs: events %.3f ms, host wait %.3f ms (long kernel done after %.3f ms)\n",
             ms, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t0).count());
    }
  }
  return 0;
}
