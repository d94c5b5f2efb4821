// Build-loop microbenchmark (gfx950): the fused kernel's slab arithmetic in isolation.
// 512-thread workgroups (2 waves per SIMD), 25 codebook pairs in registers, residuals broadcast from
// LDS with ds_read_b128, CH items built together as CH interleaved packed chains.  Reports cycles per
// packed instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int S = 25, SP = 28, M = 12, G = 16, T = 512;

template <int CH, int WRITE>
__global__ __launch_bounds__(T) void k(float* out, const float* resid, int iters, int cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);              // [G][1024]
  float* res = slab + G * 1024;                              // [G][M][SP]
  const int tid = threadIdx.x;
  for (int i = tid; i < G * M * SP; i += T) res[i] = resid[i];
  v2f cb[S];
  for (int j = 0; j < S; ++j) cb[j] = v2f{tid * 1e-3f + j, tid * 2e-3f - j};
  __syncthreads();
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int p = it % M;
#pragma unroll 1
    for (int g = 0; g < cnt; g += CH) {
      const float4* R[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) R[c] = reinterpret_cast<const float4*>(res + ((size_t)(g + c < cnt ? g + c : g) * M + p) * SP);
      v2f s[CH];
      float4 n[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) { s[c] = v2f{0.f, 0.f}; n[c] = R[c][0]; }
#pragma unroll
      for (int jb = 0; jb < SP / 4; ++jb) {
        float4 c4[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) c4[c] = n[c];
        if (jb + 1 < SP / 4) {
#pragma unroll
          for (int c = 0; c < CH; ++c) n[c] = R[c][jb + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = jb * 4 + u;
          if (j < S) {
            if (CH == 2) {
              v2f t0, t1;
              const v2f a0 = (u < 2) ? v2f{c4[0].x, c4[0].y} : v2f{c4[0].z, c4[0].w};
              const v2f a1 = (u < 2) ? v2f{c4[1].x, c4[1].y} : v2f{c4[1].z, c4[1].w};
              if ((u & 1) == 0)
                asm volatile("v_pk_add_f32 %2, %4, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %3, %5, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_mul_f32 %2, %2, %2\n\tv_pk_mul_f32 %3, %3, %3\n\t"
                             "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
                             : "+v"(s[0]), "+v"(s[1]), "=&v"(t0), "=&v"(t1) : "v"(a0), "v"(a1), "v"(cb[j]));
              else
                asm volatile("v_pk_add_f32 %2, %4, %6 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %3, %5, %6 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_mul_f32 %2, %2, %2\n\tv_pk_mul_f32 %3, %3, %3\n\t"
                             "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
                             : "+v"(s[0]), "+v"(s[1]), "=&v"(t0), "=&v"(t1) : "v"(a0), "v"(a1), "v"(cb[j]));
            } else {
              v2f t0, t1, t2, t3;
              const v2f a0 = (u < 2) ? v2f{c4[0].x, c4[0].y} : v2f{c4[0].z, c4[0].w};
              const v2f a1 = (u < 2) ? v2f{c4[1].x, c4[1].y} : v2f{c4[1].z, c4[1].w};
              const v2f a2 = (u < 2) ? v2f{c4[2 % CH].x, c4[2 % CH].y} : v2f{c4[2 % CH].z, c4[2 % CH].w};
              const v2f a3 = (u < 2) ? v2f{c4[3 % CH].x, c4[3 % CH].y} : v2f{c4[3 % CH].z, c4[3 % CH].w};
              if ((u & 1) == 0)
                asm volatile("v_pk_add_f32 %4, %8, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %5, %9, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %6, %10, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %7, %11, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_mul_f32 %4, %4, %4\n\tv_pk_mul_f32 %5, %5, %5\n\tv_pk_mul_f32 %6, %6, %6\n\tv_pk_mul_f32 %7, %7, %7\n\t"
                             "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                             : "+v"(s[0]), "+v"(s[1]), "+v"(s[2 % CH]), "+v"(s[3 % CH]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(cb[j]));
              else
                asm volatile("v_pk_add_f32 %4, %8, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %5, %9, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %6, %10, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %7, %11, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_mul_f32 %4, %4, %4\n\tv_pk_mul_f32 %5, %5, %5\n\tv_pk_mul_f32 %6, %6, %6\n\tv_pk_mul_f32 %7, %7, %7\n\t"
                             "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                             : "+v"(s[0]), "+v"(s[1]), "+v"(s[2 % CH]), "+v"(s[3 % CH]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(cb[j]));
            }
          }
        }
      }
      if (WRITE) {
#pragma unroll
        for (int c = 0; c < CH; c += 2) {
          *reinterpret_cast<v2f*>(slab + tid * G + ((g + c) & 15)) = v2f{s[c].x, s[c + 1].x};
          *reinterpret_cast<v2f*>(slab + (tid + T) * G + ((g + c) & 15)) = v2f{s[c].y, s[c + 1].y};
        }
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) sink += s[c].x + s[c].y;
      }
    }
  }
  if (!WRITE) out[blockIdx.x * T + tid] = sink;
  else out[blockIdx.x * T + tid] = slab[tid];
}

template <int CH, int WRITE>
__global__ __launch_bounds__(256) void k1(float* out, const float* resid, int iters, int cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);              // [G][1024]
  float* res = slab + G * 1024;                              // [G][M][SP]
  const int tid = threadIdx.x;
  for (int i = tid; i < G * M * SP; i += 256) res[i] = resid[i];
  v2f cb[S];
  for (int j = 0; j < S; ++j) cb[j] = v2f{tid * 1e-3f + j, tid * 2e-3f - j};
  __syncthreads();
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int p = it % M;
#pragma unroll 1
    for (int g = 0; g < cnt; g += CH) {
      const float4* R[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) R[c] = reinterpret_cast<const float4*>(res + ((size_t)(g + c < cnt ? g + c : g) * M + p) * SP);
      v2f s[CH];
      float4 n[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) { s[c] = v2f{0.f, 0.f}; n[c] = R[c][0]; }
#pragma unroll
      for (int jb = 0; jb < SP / 4; ++jb) {
        float4 c4[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) c4[c] = n[c];
        if (jb + 1 < SP / 4) {
#pragma unroll
          for (int c = 0; c < CH; ++c) n[c] = R[c][jb + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = jb * 4 + u;
          if (j < S) {
            if (CH == 2) {
              v2f t0, t1;
              const v2f a0 = (u < 2) ? v2f{c4[0].x, c4[0].y} : v2f{c4[0].z, c4[0].w};
              const v2f a1 = (u < 2) ? v2f{c4[1].x, c4[1].y} : v2f{c4[1].z, c4[1].w};
              if ((u & 1) == 0)
                asm volatile("v_pk_add_f32 %2, %4, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %3, %5, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_mul_f32 %2, %2, %2\n\tv_pk_mul_f32 %3, %3, %3\n\t"
                             "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
                             : "+v"(s[0]), "+v"(s[1]), "=&v"(t0), "=&v"(t1) : "v"(a0), "v"(a1), "v"(cb[j]));
              else
                asm volatile("v_pk_add_f32 %2, %4, %6 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %3, %5, %6 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_mul_f32 %2, %2, %2\n\tv_pk_mul_f32 %3, %3, %3\n\t"
                             "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
                             : "+v"(s[0]), "+v"(s[1]), "=&v"(t0), "=&v"(t1) : "v"(a0), "v"(a1), "v"(cb[j]));
            } else {
              v2f t0, t1, t2, t3;
              const v2f a0 = (u < 2) ? v2f{c4[0].x, c4[0].y} : v2f{c4[0].z, c4[0].w};
              const v2f a1 = (u < 2) ? v2f{c4[1].x, c4[1].y} : v2f{c4[1].z, c4[1].w};
              const v2f a2 = (u < 2) ? v2f{c4[2 % CH].x, c4[2 % CH].y} : v2f{c4[2 % CH].z, c4[2 % CH].w};
              const v2f a3 = (u < 2) ? v2f{c4[3 % CH].x, c4[3 % CH].y} : v2f{c4[3 % CH].z, c4[3 % CH].w};
              if ((u & 1) == 0)
                asm volatile("v_pk_add_f32 %4, %8, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %5, %9, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %6, %10, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %7, %11, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_mul_f32 %4, %4, %4\n\tv_pk_mul_f32 %5, %5, %5\n\tv_pk_mul_f32 %6, %6, %6\n\tv_pk_mul_f32 %7, %7, %7\n\t"
                             "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                             : "+v"(s[0]), "+v"(s[1]), "+v"(s[2 % CH]), "+v"(s[3 % CH]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(cb[j]));
              else
                asm volatile("v_pk_add_f32 %4, %8, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %5, %9, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %6, %10, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %7, %11, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_mul_f32 %4, %4, %4\n\tv_pk_mul_f32 %5, %5, %5\n\tv_pk_mul_f32 %6, %6, %6\n\tv_pk_mul_f32 %7, %7, %7\n\t"
                             "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                             : "+v"(s[0]), "+v"(s[1]), "+v"(s[2 % CH]), "+v"(s[3 % CH]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(cb[j]));
            }
          }
        }
      }
      if (WRITE) {
#pragma unroll
        for (int c = 0; c < CH; c += 2) {
          *reinterpret_cast<v2f*>(slab + tid * G + ((g + c) & 15)) = v2f{s[c].x, s[c + 1].x};
          *reinterpret_cast<v2f*>(slab + (tid + 256) * G + ((g + c) & 15)) = v2f{s[c].y, s[c + 1].y};
        }
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) sink += s[c].x + s[c].y;
      }
    }
  }
  if (!WRITE) out[blockIdx.x * 256 + tid] = sink;
  else out[blockIdx.x * 256 + tid] = slab[tid];
}

template <int CH, int WRITE>
void run(float* out, const float* resid, int cnt) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 600, grid = 256;
  const size_t lds = (size_t)(G * 1024 + G * M * SP) * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<CH, WRITE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k<CH, WRITE>), dim3(grid), dim3(T), lds, 0, out, resid, 10, cnt);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<CH, WRITE>), dim3(grid), dim3(T), lds, 0, out, resid, iters, cnt);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const int groups = (cnt + CH - 1) / CH;
  const double pk_per_wave = (double)iters * groups * CH * S * 3;
  const double cyc = ms * 1e-3 * 2.4e9 / (pk_per_wave * 2);   // 2 waves per SIMD
  printf("chains/wave=%d write=%d items=%d: %.3f ms -> %.2f cycles per pk instr per SIMD; %.1f us per (entry position)\n", CH, WRITE, cnt, ms,
         cyc, ms * 1e3 / iters);
}
template <int CH, int WRITE>
void run1(float* out, const float* resid, int cnt) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 600, grid = 256;
  const size_t lds = (size_t)(G * 1024 + G * M * SP) * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1<CH, WRITE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k1<CH, WRITE>), dim3(grid), dim3(256), lds, 0, out, resid, 10, cnt);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k1<CH, WRITE>), dim3(grid), dim3(256), lds, 0, out, resid, iters, cnt);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const int groups = (cnt + CH - 1) / CH;
  const double pk_per_wave = (double)iters * groups * CH * S * 3;
  const double cyc = ms * 1e-3 * 2.4e9 / (pk_per_wave * 1);   // 1 wave per SIMD
  printf("ONE wave/SIMD chains/wave=%d write=%d items=%d: %.3f ms -> %.2f cycles per pk instr per SIMD; %.1f us per (entry position)\n", CH, WRITE, cnt, ms,
         cyc, ms * 1e3 / iters);
}
int main() {
  float *out, *resid; (void)hipMalloc(&out, 256 * T * 4); (void)hipMalloc(&resid, G * M * SP * 4);
  (void)hipMemset(resid, 0, G * M * SP * 4);
  for (int cnt : {16, 10, 8}) {
    run<2, 0>(out, resid, cnt); run<2, 1>(out, resid, cnt);
    run<4, 0>(out, resid, cnt); run<4, 1>(out, resid, cnt);
  }
  for (int cnt : {16, 10}) { run1<2, 1>(out, resid, cnt); run1<4, 1>(out, resid, cnt); }
  return 0;
}
