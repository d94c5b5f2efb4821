// LDS gather cost: 48-byte fp32 rows (3 x ds_read_b128 + 6 packed adds) vs 24-byte int16 rows
// (3 x ds_read_b64 + integer adds), 8 rows per lane and phase, random rows, 8 gathering waves of 16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(1024) void gather_kernel(const uint32_t* codes, float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 24576; i += 1024) reinterpret_cast<float*>(smem)[i] = (float)(i & 1023);
  __syncthreads();
  uint32_t cw[8];
  for (int r = 0; r < 8; ++r) cw[r] = codes[(blockIdx.x * 1024 + tid) * 8 + r];
  long long t0 = 0;
  if (wave >= 8) {
    if (MODE == 0) {
      v2f acc[6][8];
      for (int h = 0; h < 6; ++h) for (int r = 0; r < 8; ++r) acc[h][r] = v2f{0.f, 0.f};
      t0 = clock64();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int code = (cw[r] >> ((it & 1) * 16)) & 1023;
          const float* row = reinterpret_cast<const float*>(smem) + code * 12;
          float4 v[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) v[q] = *reinterpret_cast<const float4*>(row + q * 4);
#pragma unroll
          for (int q = 0; q < 3; ++q) { acc[2*q][r] += v2f{v[q].x, v[q].y}; acc[2*q+1][r] += v2f{v[q].z, v[q].w}; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      t0 = clock64() - t0;
      float s = 0; for (int h = 0; h < 6; ++h) for (int r = 0; r < 8; ++r) s += acc[h][r].x + acc[h][r].y;
      out[blockIdx.x * 1024 + tid] = s;
    } else {
      int acc[12][8];
      for (int h = 0; h < 12; ++h) for (int r = 0; r < 8; ++r) acc[h][r] = 0;
      t0 = clock64();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int code = (cw[r] >> ((it & 1) * 16)) & 1023;
          const uint32_t* row = reinterpret_cast<const uint32_t*>(smem) + code * 6;
          uint2 v[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) v[q] = *reinterpret_cast<const uint2*>(row + q * 2);
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            acc[4*q+0][r] += (int)(short)(v[q].x & 0xffff); acc[4*q+1][r] += (int)v[q].x >> 16;
            acc[4*q+2][r] += (int)(short)(v[q].y & 0xffff); acc[4*q+3][r] += (int)v[q].y >> 16;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      t0 = clock64() - t0;
      int s = 0; for (int h = 0; h < 12; ++h) for (int r = 0; r < 8; ++r) s += acc[h][r];
      out[blockIdx.x * 1024 + tid] = (float)s;
    }
    if (tid == 512) cyc[blockIdx.x] = t0;
  } else {
    for (int it = 0; it < iters; ++it) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}
int main() {
  const int nb = 256, iters = 1200;
  uint32_t* h = (uint32_t*)malloc(nb * 1024 * 8 * 4);
  const int arrange = getenv("ARRANGE") ? atoi(getenv("ARRANGE")) : 0;   // g: lanes of every g-lane group get distinct (code mod 16)... residues
  for (int i = 0; i < nb * 1024 * 8; ++i) {
    uint32_t lo = rand() & 1023, hi = rand() & 1023;
    if (arrange) {
      const int lane = (i / 8) & 63;
      const int res = (lane % arrange) % 16;
      lo = (lo & ~15u) | res; hi = (hi & ~15u) | res;
    }
    h[i] = lo | (hi << 16);
  }
  uint32_t* d; float* o; long long* c;
  hipMalloc(&d, nb * 1024 * 8 * 4); hipMalloc(&o, nb * 1024 * 4); hipMalloc(&c, nb * 8);
  hipMemcpy(d, h, nb * 1024 * 8 * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)gather_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)gather_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
    if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(nb), dim3(1024), 98304 + 16384, 0, d, o, c, iters);
    else hipLaunchKernelGGL(gather_kernel<1>, dim3(nb), dim3(1024), 98304 + 16384, 0, d, o, c, iters);
    hipDeviceSynchronize();
    long long hc[256]; hipMemcpy(hc, c, nb * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nb; ++i) s += hc[i];
    printf("mode %d (%s): %.0f cycles per phase (8 rows x 12 items per lane, 8 waves)\n", mode, mode ? "24-byte int16 rows" : "48-byte fp32 rows", s / nb / iters);
  }
  return 0;
}
