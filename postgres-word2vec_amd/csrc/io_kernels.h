// io_kernels.h -- the copy kernels of the host-buffer calls (ivfadc.hip, pq.hip): pinned host memory is read and written by
// kernels in the search's own stream, not by SDMA copies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The lanes move their data with KERNELS, not with hipMemcpyAsync: pinned host memory is mapped into the device's address
// space, so a grid-stride copy reads the staged queries over PCIe (1.2 MB per 1024 queries: ~25 us) and a second one
// writes the lists, the straggler count and the stragglers' numbers back -- ordinary launches in the lane's stream.
// Measured (tools/pipe_trace.py): with hipMemcpyAsync (SDMA copies ordered against kernels by signals) the four lanes'
// chains ran in pairs one after the other, 1.4 ms per 4096 queries; with copy kernels they overlap like the
// device-resident batches of bench.py: 0.68 ms.
static __global__ __launch_bounds__(256) void lane_copy_in_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static __global__ __launch_bounds__(256) void lane_copy_out_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dist,
                                                           const int32_t* __restrict__ n_next, const int32_t* __restrict__ unfinished,
                                                           int32_t* __restrict__ h_out, int n_out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_out) { h_out[i] = ids[i]; h_out[n_out + i] = __float_as_int(dist[i]); }
  const int nn = n_next[0];
  if (i == 0) h_out[2 * n_out] = nn;
  if (i < nn && i < n) h_out[2 * n_out + 1 + i] = unfinished[i];
}

// The same by ONE workgroup, followed by a completion word the host polls (lane_retire): the lists are in (mapped host)
// memory before the word.  n_out * 2 + n + 1 words: a few tens of KB.
static __global__ __launch_bounds__(1024) void lane_copy_out_flag_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dist,
                                                                 const int32_t* __restrict__ n_next, const int32_t* __restrict__ unfinished,
                                                                 int32_t* __restrict__ h_out, int n_out, int n, int32_t* __restrict__ flag) {
  const int nn = n_next[0];
  for (int i = threadIdx.x; i < n_out; i += 1024) { h_out[i] = ids[i]; h_out[n_out + i] = __float_as_int(dist[i]); }
  if (threadIdx.x == 0) h_out[2 * n_out] = nn;
  for (int i = threadIdx.x; i < nn && i < n; i += 1024) h_out[2 * n_out + 1 + i] = unfinished[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static __global__ __launch_bounds__(256) void host_io_out_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dist, int32_t* __restrict__ h_out, int n_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_out) { h_out[i] = ids[i]; h_out[n_out + i] = __float_as_int(dist[i]); }
}

