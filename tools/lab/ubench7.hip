// tools/lab/ubench7.hip -- the scan's gather stage in two structures, K = 256, 16 items x 8 rows per lane, 8 gathering + 8 building waves:
//   mode 0: as ivf_filter5_kernel -- six phases of two positions, a barrier per phase, the builders storing the next phase's slab
//           (8 ds_write_b128 per lane and phase) while the gatherers read this one's; two rows (8 reads) in flight per wave
//   mode 1: WHOLE-ENTRY slab (12 positions x 2 halves x 256 codes x 16 B = 96 KB, built once per entry): the gatherers read all twelve
//           positions of a row back to back, no barrier inside the entry, nobody stores meanwhile; 8 reads in flight per wave
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench7 tools/lab/ubench7.hip && /tmp/ubench7
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
static constexpr uint32_t HALFB = 256 * 16, POSB = 2 * HALFB;

template <int MODE>
__global__ __launch_bounds__(1024) void gather_kernel(const uint32_t* codes, uint32_t* out, long long* cyc, int entries, int arrange) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (int)(12 * POSB / 4); i += 1024) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)(i * 2654435761u) & 0x0fff0fffu;
  __syncthreads();
  const uint32_t seed = codes[blockIdx.x * 1024 + tid];
  auto code_word = [&](int r, int t, int e) -> uint32_t {   // 4 one-byte codes of row r, position quad t (pseudo-random per lane, row, entry)
    uint32_t x = seed + (uint32_t)(r * 3 + t) * 0x9E3779B9u + (uint32_t)e * 0x85EBCA6Bu;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    if (arrange) {   // conflict-free: the 16 lanes the LDS serves together read 16 different bank quads (code mod 16 = rank in the group)
      const int l = lane & 31;
      const int g0 = (l < 4) ? l : (l >= 12 && l < 16) ? l - 8 : (l >= 20 && l < 28) ? l - 12 : -1;
      const int rank = g0 >= 0 ? g0 : ((l >= 4 && l < 12) ? l - 4 : (l >= 16 && l < 20) ? l - 8 : l - 16);
      x = (x & 0xf0f0f0f0u) | (uint32_t)rank * 0x01010101u;
    }
    return x;
  };
  long long t0 = 0;
  if (wave >= 8) {
    uint32_t acc[8][8];
    for (int h = 0; h < 8; ++h) for (int r = 0; r < 8; ++r) acc[h][r] = 0u;
    t0 = clock64();
    for (int e = 0; e < entries; ++e) {
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const uint32_t bb = (uint32_t)(j & 1) * 2u * POSB;     // (two buffers of two positions each at the front of the slab area)
          u4 va[2][2][2];
          auto issue = [&](int r) {
            const uint32_t w = code_word(r, j >> 1, e);
            const uint32_t a0 = (((j & 1) ? (w >> 12) : (w << 4)) & 0xff0u) + bb;
            const uint32_t a1 = (((j & 1) ? (w >> 20) : (w >> 4)) & 0xff0u) + bb;
#pragma unroll
            for (int q = 0; q < 2; ++q) va[r & 1][0][q] = *reinterpret_cast<const u4*>(smem + a0 + q * HALFB);
#pragma unroll
            for (int q = 0; q < 2; ++q) va[r & 1][1][q] = *reinterpret_cast<const u4*>(smem + a1 + POSB + q * HALFB);
          };
          issue(0);
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            if (r + 1 < 8) issue(r + 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const u4 x = va[r & 1][0][q], y = va[r & 1][1][q];
              acc[q * 4 + 0][r] += x.x + y.x; acc[q * 4 + 1][r] += x.y + y.y; acc[q * 4 + 2][r] += x.z + y.z; acc[q * 4 + 3][r] += x.w + y.w;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
      } else {
        // twelve positions of a row back to back: steps of two positions (4 reads), two steps in flight (32 landing registers, as mode 0)
        u4 va[2][2][2];
        auto issue = [&](int i) {     // step i = (row i / 6, position pair i % 6)
          const int r = i / 6, pp = i % 6;
          const uint32_t w = code_word(r, pp >> 1, e) >> (16 * (pp & 1));
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const uint32_t a = (((w >> (8 * p)) << 4) & 0xff0u) + (uint32_t)(2 * pp + p) * POSB;
#pragma unroll
            for (int q = 0; q < 2; ++q) va[i & 1][p][q] = *reinterpret_cast<const u4*>(smem + a + q * HALFB);
          }
        };
        issue(0);
#pragma unroll
        for (int i = 0; i < 48; ++i) {
          const int r = i / 6;
          if (i + 1 < 48) issue(i + 1);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const u4 x = va[i & 1][0][q], y = va[i & 1][1][q];
            acc[q * 4 + 0][r] += x.x + y.x; acc[q * 4 + 1][r] += x.y + y.y; acc[q * 4 + 2][r] += x.z + y.z; acc[q * 4 + 3][r] += x.w + y.w;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (the entry's single barrier: slab consumed)
      }
    }
    t0 = clock64() - t0;
    uint32_t s = 0; for (int h = 0; h < 8; ++h) for (int r = 0; r < 8; ++r) s += acc[h][r];
    out[blockIdx.x * 1024 + tid] = s;
    if (tid == 512) cyc[blockIdx.x] = t0;
  } else {
    // builders: mode 0 stores the other buffer's 32 KB every phase (8 x 16 B per lane of 512 lanes = 64 KB... half rows of 2 positions x 2 halves x 256 codes = 16 KB at K = 256: 2 stores per lane)
    const u4 v = u4{(uint32_t)tid, 1u, 2u, 3u};
    for (int e = 0; e < entries; ++e) {
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const uint32_t bb = (uint32_t)((j + 1) & 1) * 2u * POSB;
#pragma unroll
          for (int k = 0; k < 2; ++k) *reinterpret_cast<u4*>(smem + bb + (uint32_t)(tid + 512 * k) * 16u) = v;
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
  }
}
int main() {
  const int nb = 256, entries = 200;
  const int arrange = getenv("ARRANGE") ? atoi(getenv("ARRANGE")) : 0;
  const size_t n = (size_t)nb * 1024;
  uint32_t* h = (uint32_t*)malloc(n * 4);
  for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
  uint32_t *d, *o; long long* c;
  hipMalloc(&d, n * 4); hipMalloc(&o, nb * 1024 * 4); hipMalloc(&c, nb * 8);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)gather_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)gather_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) for (int mode = 0; mode < 2; ++mode) {
    if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(nb), dim3(1024), 12 * POSB + 1024, 0, d, o, c, entries, arrange);
    else hipLaunchKernelGGL(gather_kernel<1>, dim3(nb), dim3(1024), 12 * POSB + 1024, 0, d, o, c, entries, arrange);
    hipDeviceSynchronize();
    long long hc[256]; hipMemcpy(hc, c, nb * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nb; ++i) s += hc[i];
    printf("arrange %d mode %d (%s): %.0f cycles per entry (16 items x 8 rows per lane x 12 positions, 8 gathering waves)\n", arrange, mode,
           mode ? "whole-entry slab, no barriers inside" : "six phases, a barrier each, builders storing", s / nb / entries);
  }
  return 0;
}
