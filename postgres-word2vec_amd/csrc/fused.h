// fused.h -- IVFADC: LUT build + ADC scan + candidate selection in ONE kernel (gfx950).
//
// This file holds what the three fused kernels share (constants, the work table, the argument block,
// the survivor merge) and the first, SYMMETRIC kernel ivf_fused_kernel (FREDDY_GPU_FUSED_KERNEL=1);
// fused2.h and fused3.h (default) are the role-specialised successors.
//
// Why: with separate kernels the per-(query, cell) LUTs (48 KiB each, 503 MB per 1024-query
// batch at nprobe=10) are written to memory by lut_build and read back by adc_scan -- the
// first rocprofv3 pass showed that round trip to be 2/3 of the scan kernel's traffic.  Here a
// LUT never exists as a whole: the workgroup walks the m positions, builds the 64 KiB LUT slab of
// one position for its G items in LDS, and every lane immediately adds it to the running ADC sums
// of its rows, which live in registers.  The sum still runs over positions 0..m-1 in order
// (index_utils.c:1126-1133), each slab entry is still the sequential squareDistance over the
// sub-vector (index_utils.c:445-455, :500-508).
//
//   work item  = (query, probed cell); items are grouped by cell, <= 16 per workgroup
//   workgroup  = 512 threads (8 waves), one 4096-row chunk of the cell's list
//   LDS        = 2 buffers x 16 items x K floats = 128 KiB (K = 1024)
//
// Selection (replaces the per-wave streaming top-L of adc_scan) works on the distance bits: each
// lane's smallest distance per item goes through LDS; one wave per item takes the column minima over
// the 8 waves, sorts those 64 values and uses the L-th as threshold tau.  L rows have a distance
// <= tau, so {distance <= tau} contains the item's L smallest (distance, row id) keys (typically
// L + a few rows).  Every wave appends its survivors to its own region of the item's buffer;
// merge_surv_kernel picks the query's 2k smallest keys and replays the reference's insertion.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "wave_topk.h"

namespace freddy {

static constexpr int FUSED_T = 512;
static constexpr int FUSED_NW = FUSED_T / 64;
static constexpr int FUSED_G = 16;                                 // (query, cell) items per workgroup
static constexpr int FUSED_RMAX = 8;
static constexpr int FUSED_E = 2;                                  // codes per lane: K <= 1024
static constexpr int FUSED_UNIT_BLOCKS = FUSED_RMAX * FUSED_NW;   // 64 row blocks = 4096 rows per chunk

// ---------------------------------------------------------------------------------------
// Cell-major grouping of the round's (query, cell) items: the probe plan appends every item to its
// cell's bucket (cell_items[cell][0..count), kernels.h); work_table_kernel turns the per-cell counts
// into work entries and orders them for the persistent workgroups.  (Order inside a cell is
// irrelevant: every item is selected and merged on its own.)
// ---------------------------------------------------------------------------------------
// One workgroup of 256 threads: per-cell counts -> work entries (group of <= gsz <= FUSED_G items of a
// cell x 4096-row chunk; first = index into cell_items), so the fused kernels' grids have no holes, in
// longest-processing-time-first order: entries with more items (more slab arithmetic) are pulled first,
// the tail of the launch is made of small entries (counting sort on (items, rows) classes).
__global__ __launch_bounds__(1024) void work_table_kernel(const int32_t* __restrict__ cell_count, int C, int cell_cap, int gsz,
                                                         const int32_t* __restrict__ blk_off, int32_t* __restrict__ out_cell,
                                                         int32_t* __restrict__ out_first, int32_t* __restrict__ out_cnt,
                                                         int32_t* __restrict__ n_groups, int cost_mode) {
  constexpr int NB = 128;   // cost classes, descending (cost_mode 0 uses FUSED_G * 4 + 4 of them: (items, quarter of a full chunk))
  constexpr int T = 1024, CPT = 4;   // the first T * CPT cells are read once and kept in registers for both sweeps
  __shared__ int hist[NB];
  __shared__ int start[NB];
  const int tid = threadIdx.x;
  for (int i = tid; i < NB; i += T) hist[i] = 0;
  int cn[CPT], cb[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + i * T;
    cn[i] = c < C ? cell_count[c] : 0;
    cb[i] = c < C ? blk_off[c + 1] - blk_off[c] : 0;
  }
  __syncthreads();
  // Two sweeps over this thread's cells: count the entries per class, then emit them into their class's
  // range.  Order inside a class is irrelevant.
  auto cell = [&](bool emit, int c, int n, int nblk) {
    if (n == 0) return;
    const int chunks = (nblk + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS;
    for (int f = 0; f < n; f += gsz) {
      const int cnt = (n - f < gsz) ? n - f : gsz;
      for (int ch = 0; ch < chunks; ++ch) {
        int nb = nblk - ch * FUSED_UNIT_BLOCKS;
        nb = nb > FUSED_UNIT_BLOCKS ? FUSED_UNIT_BLOCKS : nb;
        const int rq = (nb * 4 - 1) / FUSED_UNIT_BLOCKS;          // 0..3
        // small class index = big entry.  cost_mode 0 (exact kernels): slab arithmetic grows with the items;
        // cost_mode 1 (filter kernel, LDS-bound): measured model in units of 100 cycles, selection tail +
        // 12 x max(builder phase, gather phase)
        int k = (FUSED_G - cnt) * 4 + (3 - rq);
        if (cost_mode) {
          const int gp = 4 + (26 * (rq + 1) * ((cnt + 3) >> 2) + 5) / 10;
          const int cost = 50 + 7 * cnt + 12 * (gp > 15 ? gp : 15);   // 237 .. 554
          k = (560 - cost) / 3;
        }
        if (!emit) {
          atomicAdd(&hist[k], 1);
        } else {
          const int slot = atomicAdd(&start[k], 1);
          out_cell[slot] = c;
          out_first[slot] = c * cell_cap + f;
          out_cnt[slot] = cnt | (ch << 8);
        }
      }
    }
  };
  auto sweep = [&](bool emit) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) cell(emit, tid + i * T, cn[i], cb[i]);
    for (int c = tid + CPT * T; c < C; c += T) cell(emit, c, cell_count[c], blk_off[c + 1] - blk_off[c]);
  };
  sweep(false);
  __syncthreads();
  if (tid < 64) {   // exclusive prefix over the NB = 128 classes, two per lane
    const int h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
    int inc = h0 + h1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (tid >= o) inc += up;
    }
    start[2 * tid] = inc - h0 - h1;
    start[2 * tid + 1] = inc - h1;
    if (tid == 63) n_groups[0] = inc;
  }
  __syncthreads();
  sweep(true);
}

struct FusedArgs {
  const float* resid;          // [items][m][SP] residuals, each position padded to SP floats (freddy.c:296-303); symmetric kernel
  const float* queries;        // [Q][d]   the role-specialised kernel forms r = q - coarse[cell] itself while staging
  const float* coarse;         // [C][d]
  const int32_t* item_query;   // [items]
  const int32_t* sorted_item;  // items in cell order
  const int32_t* group_cell;   // [groups]
  const int32_t* group_first;
  const int32_t* group_cnt;
  const int32_t* n_groups;     // [1] number of (group, chunk) work entries
  int32_t* work_counter;       // [1] zeroed before the launch
  const float* cbP;            // [m][SP/4][512 slots][4 dims][2 codes] (see load_cb)
  const int32_t* blk_off;      // [C+1]
  const uint32_t* packed;      // [blocks][M2][64]
  const int32_t* pos;          // [blocks*64]
  u64* surv;                   // [items][upi][8 waves][512] survivor keys, one region per (item, chunk, wave)
  int32_t* surv_count;         // [items][upi][8 waves] written by the kernel for every region of a live item
  int32_t* cand_count;         // [Q] or NULL
  int d, K, L, upi;            // upi: chunks per item the buffers are laid out for
  uint32_t sentinel_bits;
  uint32_t desc_offset;        // byte offset of the item-descriptor scratch inside dynamic LDS
  uint32_t ablate;             // timing experiments only (FREDDY_GPU_FUSED_ABLATE)
  long long* prof;             // NULL, or [gridDim.x][8] cycle sums per phase (FREDDY_GPU_FUSED_PROF)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc drain vmcnt too,
// which would expose the latency of every prefetch that is meant to fly across the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One workgroup (8 waves, 256-VGPR budget) = up to 16 items that probe the SAME cell, and one
// 4096-row chunk of that cell's list.  Per position p:
//     build slab(p+1)   lane <-> code; this lane's 2 codebook entries (50 VGPRs, packed fp32
//                       math) serve all items; an item's residual sub-vector is wave-uniform
//                       (scalar loads, double-buffered)
//     prefetch the codebook entries of position p+2
//     gather slab(p)    lane <-> row; the code dword of a row is loaded ONCE for all 16 items
//     barrier (LDS only)
// so per workgroup the codebook (1.2 MB) and the list's codes (<= 96 KiB) cross the L2 once for
// 16 (query, cell) pairs.
template <int S, int M, bool FULLK>   // FULLK: K == T*E, no per-lane code guards (keeps both chains in one block)
__global__ __launch_bounds__(FUSED_T) void ivf_fused_kernel(FusedArgs a) {
  constexpr int G = FUSED_G, RMAX = FUSED_RMAX, NW = FUSED_NW, T = FUSED_T, E = FUSED_E;
  constexpr int M2 = M / 2;
  static_assert(M % 2 == 0, "two int16 codes per dword");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);   // [2][G][K]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int K = a.K;
  // Persistent workgroups: one per CU, work entries are pulled from a device-wide counter.  The
  // descriptor of the NEXT entry (work index -> cell, items, row blocks) is fetched underneath the
  // selection phase of the current one -- three dependent global round trips that used to sit in
  // front of every entry -- into the other half of a double-buffered LDS descriptor:
  //   dsc[b*16 + g]      item id of slot g (-1: unused)          b = 0/1
  //   dsc[32 + b*8 + ..] {work index or -1, items, first block, blocks, chunk}
  const int n_work = a.n_groups[0];
  int32_t* dsc = reinterpret_cast<int32_t*>(smem + a.desc_offset);
  float* res = reinterpret_cast<float*>(smem + a.desc_offset + 256);  // [G][M][SP] residuals
  constexpr int ROW4 = M * ((S + 3) & ~3) / 4;                        // float4 per item
  // opaque: stops LICM from hoisting (and then spilling) one 64-bit codebook address per position
  const float* cbp = a.cbP;
  asm volatile("" : "+s"(cbp));

  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f cb[S];   // .x: code tid, .y: code tid + T  (packed so the two chains run as v_pk_* ops)
  static_assert(E == 2, "two codes per lane");
  // cbP layout [m][SP/4][T slots][4 dims][2 codes]: slot tid = codes (tid, tid+T) interleaved, so two
  // 16-byte loads bring four dimensions of both codes already in (x, y) pair order (codes >= K are
  // zero-padded by the host).  14 wide loads per position instead of 50 dword loads.
  auto load_cb = [&](int p) {
    constexpr int SPq = ((S + 3) & ~3) / 4;
    // explicitly GLOBAL pointers: after the asm above the compiler only knows a generic one and
    // would emit flat_load, which also counts in lgkmcnt -- every LDS-only barrier would then wait
    // for the prefetch it is supposed to let fly
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const f4 __attribute__((address_space(1))) * gptr4;
    typedef const char __attribute__((address_space(1))) * gptrc;
    const gptrc base = (gptrc)(uintptr_t)cbp + (size_t)(uint32_t)p * (uint32_t)(SPq * T * 32);
    const uint32_t voff = (uint32_t)tid * 32u;
#pragma unroll
    for (int jb = 0; jb < SPq; ++jb) {
      const gptrc bj = base + (uint32_t)jb * (uint32_t)(T * 32);
      const f4 lo = *(gptr4)(bj + voff), hi = *(gptr4)(bj + voff + 16u);
      if (jb * 4 + 0 < S) cb[jb * 4 + 0] = v2f{lo.x, lo.y};
      if (jb * 4 + 1 < S) cb[jb * 4 + 1] = v2f{lo.z, lo.w};
      if (jb * 4 + 2 < S) cb[jb * 4 + 2] = v2f{hi.x, hi.y};
      if (jb * 4 + 3 < S) cb[jb * 4 + 3] = v2f{hi.z, hi.w};
    }
  };

  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc = 0;
  auto tick = [&](int slot) { if (a.prof) { const long long t = clock64(); pt[slot] += t - pc; pc = t; } };
  if (a.prof) pc = clock64();
  // ---- first entry: fetched serially ----
  int cur = 0;
  if (tid == 0) dsc[32] = atomicAdd(a.work_counter, 1);
  __syncthreads();
  {
    const int gid0 = dsc[32];
    if (gid0 >= n_work) return;
    if (wave == 0) {
      const int cell = a.group_cell[gid0], first = a.group_first[gid0], gc = a.group_cnt[gid0];
      const int cnt0 = gc & 0xff, chunk0 = gc >> 8;
      const int b0 = a.blk_off[cell] + chunk0 * FUSED_UNIT_BLOCKS;
      int nb0 = a.blk_off[cell + 1] - b0;
      if (nb0 > FUSED_UNIT_BLOCKS) nb0 = FUSED_UNIT_BLOCKS;   // (>= 1 by construction of the work table)
      if (lane < G) dsc[lane] = (lane < cnt0) ? a.sorted_item[first + lane] : -1;
      if (lane == 0) { dsc[33] = cnt0; dsc[34] = b0; dsc[35] = nb0; dsc[36] = chunk0; }
    }
  }
  __syncthreads();
  load_cb(0);
  for (;;) {   // ---- one work entry per iteration; dsc[cur] is complete and visible here ----
  const int32_t* desc = dsc + cur * 16;
  const int cnt = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 1]);
  const int blk0 = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 2]);
  const int nblk = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 3]);
  const int chunk = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 4]);
  for (int i = tid; i < cnt * ROW4; i += T) {
    const int g = i / ROW4, o = i - g * ROW4;
    reinterpret_cast<float4*>(res)[i] = reinterpret_cast<const float4*>(a.resid)[(size_t)desc[g] * ROW4 + o];
  }
  int ngid = 0;   // thread 0: work index of the next entry, requested now, published at the end
  if (tid == 0) ngid = atomicAdd(a.work_counter, 1);
  __syncthreads();
  tick(0);   // descriptor + residual staging

  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f acc[G / 2][RMAX];   // ADC sums: acc[h][r] = items (2h, 2h+1) of this lane's row r
  uint32_t cw[RMAX];
#pragma unroll
  for (int h = 0; h < G / 2; ++h)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[h][r] = v2f{0.0f, 0.0f};
  // rows past the end of the chunk re-read its last block (always in bounds); masked at the end
  auto row_block = [&](int r) {
    const int b = r * NW + wave;
    return (uint32_t)(blk0 + (b < nblk - 1 ? b : nblk - 1));
  };
  auto load_codes = [&](int pair) {
#pragma unroll
    for (int r = 0; r < RMAX; ++r) cw[r] = a.packed[(row_block(r) * M2 + (uint32_t)pair) * 64u + (uint32_t)lane];
  };
  // Slab layout: [code][G items] floats (64 bytes per code).  A row's four 16-byte chunks are
  // stored at chunk index (q ^ ((code >> 1) & 3)): readers (one random row per lane) are not
  // affected, writers (64 consecutive rows per wave, same item) spread over 8 bank groups
  // instead of 2.
  static_assert(G == 16, "slab rows hold 16 items");
  auto slab_at = [&](int code, int g) { return code * G + ((((g >> 2) ^ ((code >> 1) & 3))) << 2) + (g & 3); };

  // Residual sub-vectors of the group's items live in LDS, padded to SP floats per position so
  // that a lane fetches four dimensions with one aligned ds_read_b128 (all lanes read the same
  // address: a broadcast, no bank conflict).  Two items are built together: their two packed
  // chains are independent, which is what keeps the VALU pipe full with only 2 waves per SIMD.
  constexpr int SP = (S + 3) & ~3;
  auto build_slab = [&](int p, float* dst) {
#pragma unroll 1
    for (int g = 0; g < cnt; g += 2) {
      const float4* R0 = reinterpret_cast<const float4*>(res + ((size_t)g * M + p) * SP);
      const float4* R1 = reinterpret_cast<const float4*>(res + ((size_t)(g + 1 < cnt ? g + 1 : g) * M + p) * SP);
      // One dimension of both items per step, written out as six packed instructions in a fixed
      // order (sub, sub, mul, mul, add, add): the two chains alternate, so every instruction's
      // operands were produced two issues earlier (covers the 1-wait-state VALU->v_pk hazard
      // without s_nop) and the pipe always has an independent instruction to issue.  hipcc's
      // scheduler otherwise serialises one chain after the other under this register pressure.
      // a + (-b) with the neg modifier is the IEEE subtraction; each half rounds like the scalar op.
      v2f s0 = {0.0f, 0.0f}, s1 = {0.0f, 0.0f};
      float4 n0 = R0[0], n1 = R1[0];
#pragma unroll
      for (int jb = 0; jb < SP / 4; ++jb) {
        const float4 c0 = n0, c1 = n1;
        if (jb + 1 < SP / 4) { n0 = R0[jb + 1]; n1 = R1[jb + 1]; }
        const v2f a0[2] = {{c0.x, c0.y}, {c0.z, c0.w}};
        const v2f a1[2] = {{c1.x, c1.y}, {c1.z, c1.w}};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = jb * 4 + u;
          if (j < S) {
            v2f t0, t1;
            if ((u & 1) == 0) {
              asm volatile(
                  "v_pk_add_f32 %2, %4, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                  "v_pk_add_f32 %3, %5, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                  "v_pk_mul_f32 %2, %2, %2\n\t"
                  "v_pk_mul_f32 %3, %3, %3\n\t"
                  "v_pk_add_f32 %0, %0, %2\n\t"
                  "v_pk_add_f32 %1, %1, %3"
                  : "+v"(s0), "+v"(s1), "=&v"(t0), "=&v"(t1)
                  : "v"(a0[u >> 1]), "v"(a1[u >> 1]), "v"(cb[j]));
            } else {
              asm volatile(
                  "v_pk_add_f32 %2, %4, %6 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                  "v_pk_add_f32 %3, %5, %6 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                  "v_pk_mul_f32 %2, %2, %2\n\t"
                  "v_pk_mul_f32 %3, %3, %3\n\t"
                  "v_pk_add_f32 %0, %0, %2\n\t"
                  "v_pk_add_f32 %1, %1, %3"
                  : "+v"(s0), "+v"(s1), "=&v"(t0), "=&v"(t1)
                  : "v"(a0[u >> 1]), "v"(a1[u >> 1]), "v"(cb[j]));
            }
          }
        }
      }
      // slab layout [code][16 items], 16-byte item chunks XOR-swizzled by the code row (see
      // slab_at): the pair (g, g+1) of one code is one aligned 8-byte store.  An unused odd slot
      // (g+1 == cnt) receives a value nobody reads.
      if (FULLK || tid < K) *reinterpret_cast<v2f*>(dst + slab_at(tid, g)) = v2f{s0.x, s1.x};
      if (FULLK || tid + T < K) *reinterpret_cast<v2f*>(dst + slab_at(tid + T, g)) = v2f{s0.y, s1.y};
    }
  };
  // gather slab(p): lane <-> row.  NQ = number of 4-item chunks in use (workgroup-uniform, chosen
  // once per entry): inside one instantiation there is no branch, so the 2*NQ ds_read_b128 of two
  // rows are in flight together and the adds of one row pair overlap the reads of the next.  (With
  // a per-chunk `if` every read sat in its own basic block behind an lgkmcnt(0).)
  // hipcc's waitcnt pass merges the "load pending" state of the code registers over the loop's
  // paths and puts a vmcnt(0) in front of the gather -- behind the codebook prefetch that was just
  // issued.  Re-defining cw with a VALU move at a point where its loads have certainly landed
  // (build-first waves: where the build waits for its codebook; gather-first waves: after the
  // build) retires that state, and the gather starts without waiting for the prefetch.
  auto settle_codes = [&]() {
#pragma unroll
    for (int r = 0; r < RMAX; ++r) asm volatile("v_mov_b32 %0, %0" : "+v"(cw[r]));
  };
  auto gather_n = [&](auto nq_tag, int p, const float* cur) {
    constexpr int NQ = decltype(nq_tag)::value;
    const int sh = (p & 1) * 16;
#pragma unroll
    for (int r0 = 0; r0 < RMAX; r0 += 2) {
      float4 v[2][NQ];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int code = (int)((cw[r0 + rr] >> sh) & 0xffffu);
        const int sw = (code >> 1) & 3;
        const float* row = cur + code * G;
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[rr][q] = *reinterpret_cast<const float4*>(row + ((q ^ sw) << 2));
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          acc[q * 2 + 0][r0 + rr] = acc[q * 2 + 0][r0 + rr] + v2f{v[rr][q].x, v[rr][q].y};
          acc[q * 2 + 1][r0 + rr] = acc[q * 2 + 1][r0 + rr] + v2f{v[rr][q].z, v[rr][q].w};
        }
      __builtin_amdgcn_sched_barrier(0);   // keep it at two rows in flight (register budget)
    }
  };
  // All four 4-item chunks are always fetched (unused item slots hold stale finite-or-not values
  // nobody reads): choosing between instantiations per entry makes the register allocator copy the
  // 128 sums around at the joins.
  auto gather = [&](int p, const float* cur) {
    if (a.ablate & 2) return;
    gather_n(std::integral_constant<int, 4>{}, p, cur);
  };

  load_codes(0);   // (the codebook registers of position 0 are already in flight)
  build_slab(0, slab);
  if (M > 1) load_cb(1);
  lds_barrier();
  tick(1);   // first slab
  // Waves 0-3 build first and gather second, waves 4-7 the other way round (each SIMD hosts one
  // wave of each kind): while one half keeps the VALU busy with slab(p+1), the other half keeps
  // the LDS busy with the gathers of slab(p).  Both orders only read buffer p&1 and write the
  // other one, so one barrier per position still suffices.
  // The two roles are two separate loops, not one loop with role branches: hipcc's waitcnt pass
  // merges the "load pending" state of all paths at every join, and with shared code it put a
  // vmcnt(0) in front of each gather -- behind the codebook prefetch that had just been issued.
  const bool gather_first = (wave >> 2) & 1;
  if (gather_first) {
    for (int p = 0; p + 1 < M; ++p) {
      float* nxt = slab + (size_t)((p + 1) & 1) * G * K;
      const float* cur = slab + (size_t)(p & 1) * G * K;
      gather(p, cur);
      __builtin_amdgcn_sched_barrier(0);
      if ((p & 1) && !(a.ablate & 8)) load_codes((p + 1) >> 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(a.ablate & 1)) build_slab(p + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      settle_codes();
      if (p + 2 < M && !(a.ablate & 8)) load_cb(p + 2);
      lds_barrier();
    }
  } else {
    for (int p = 0; p + 1 < M; ++p) {
      float* nxt = slab + (size_t)((p + 1) & 1) * G * K;
      const float* cur = slab + (size_t)(p & 1) * G * K;
      settle_codes();
      if (!(a.ablate & 1)) build_slab(p + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      if (p + 2 < M && !(a.ablate & 8)) load_cb(p + 2);
      __builtin_amdgcn_sched_barrier(0);
      gather(p, cur);
      __builtin_amdgcn_sched_barrier(0);
      if ((p & 1) && !(a.ablate & 8)) load_codes((p + 1) >> 1);
      lds_barrier();
    }
  }
  tick(2);   // main loop
  // last position: nothing left to build, the codebook registers are free -> fetch the row ids
  // (scan positions) the selection needs, and position 0 of the codebook for the next entry (it does
  // not depend on the entry), underneath the last gather
  int32_t pid[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) pid[r] = a.pos[row_block(r) * 64u + (uint32_t)lane];
  if (!(a.ablate & 8)) load_cb(0);
  gather(M - 1, slab + (size_t)((M - 1) & 1) * G * K);
  const int nb = cur ^ 1;
  if (tid == 0) dsc[32 + nb * 8] = (ngid < n_work) ? ngid : -1;
  lds_barrier();   // every wave is done reading the slabs: the selection scratch aliases them
  tick(3);   // last gather

  // ---- selection -------------------------------------------------------------------------
  // Works on the distance bits only (distances are >= +0, so the bit patterns order like the
  // floats).  Per item: the L-th smallest of the 64 column minima (column = lane index over the 8
  // waves x 8 row slots) is an upper bound tau of the L-th smallest distance of the chunk, so
  // {distance <= tau} contains the item's L smallest (distance, row id) keys; typically L + a few
  // rows survive, with many equal distances possibly more, never more than the chunk has rows.
  // Each wave appends its survivors to its own region of the item's buffer (no atomics), the merge
  // kernel picks the 2k smallest keys of the query and replays.
  uint32_t* exch = reinterpret_cast<uint32_t*>(smem);   // [G][T], aliases the slabs
  uint32_t* tau_s = exch + (size_t)G * T;               // [G]
  auto bits = [&](int g, int r) { return __float_as_uint((g & 1) ? acc[g >> 1][r].y : acc[g >> 1][r].x); };
  {
    bool dead[RMAX];
    bool some = false;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) { dead[r] = !(((r * NW + wave) < nblk) && pid[r] >= 0); some |= dead[r]; }
    if (__ballot(some) != 0ull) {   // only the last chunk of a list has padding rows: park them above everything
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
#pragma unroll
        for (int h = 0; h < G / 2; ++h)
          if (dead[r]) acc[h][r] = v2f{__uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu)};
    }
  }
  if (!(a.ablate & 4)) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < cnt) {
        uint32_t best = bits(g, 0);
#pragma unroll
        for (int r = 1; r < RMAX; ++r) best = min(best, bits(g, r));
        exch[(size_t)g * T + tid] = best;
      }
    }
  }
  // next entry, level 2 of 3: its work-table row (wave 0; every lane reads the same words)
  const int ngid_l = dsc[32 + nb * 8];
  int n_cell = 0, n_first = 0, n_gc = 0;
  if (wave == 0 && ngid_l >= 0) {
    n_cell = a.group_cell[ngid_l];
    n_first = a.group_first[ngid_l];
    n_gc = a.group_cnt[ngid_l];
  }
  __syncthreads();
  if (!(a.ablate & 4)) {   // each wave finds the thresholds of its two items, the two sorts interleaved
    static_assert(G == 2 * NW, "two items per wave");
    const int g0 = wave, g1 = wave + NW;
    uint32_t c0 = 0xffffffffu, c1 = 0xffffffffu;
    if (g0 < cnt) {
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) c0 = min(c0, exch[(size_t)g0 * T + w2 * 64 + lane]);
    }
    if (g1 < cnt) {
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) c1 = min(c1, exch[(size_t)g1 * T + w2 * 64 + lane]);
    }
    wave_sort32_x2(c0, c1);
    const uint32_t t0 = __shfl(c0, a.L - 1, 64), t1 = __shfl(c1, a.L - 1, 64);
    // survivors are {bits <= tau and bits < sentinel}: fold both into one bound
    if (lane == 0) {
      tau_s[g0] = min(t0, a.sentinel_bits - 1u);
      tau_s[g1] = min(t1, a.sentinel_bits - 1u);
    }
  }
  // next entry, level 3 of 3: its item ids and row-block range
  int n_item = -1, n_b0 = 0, n_b1 = 0;
  if (wave == 0 && ngid_l >= 0) {
    if (lane < (n_gc & 0xff)) n_item = a.sorted_item[n_first + lane];
    n_b0 = a.blk_off[n_cell];
    n_b1 = a.blk_off[n_cell + 1];
  }
  __syncthreads();
  if (!(a.ablate & 4)) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < cnt) {
        const uint32_t tau = (uint32_t)__builtin_amdgcn_readfirstlane((int)tau_s[g]);
        const int it = __builtin_amdgcn_readfirstlane(desc[g]);
        const size_t region = ((size_t)it * a.upi + chunk) * NW + wave;
        u64* dst = a.surv + region * (size_t)(RMAX * 64);
        if (a.cand_count) {   // freddy.c:971 counts the rows that pass the sentinel guard
          int accepted = 0;
#pragma unroll
          for (int r = 0; r < RMAX; ++r) accepted += __popcll(__ballot(bits(g, r) < a.sentinel_bits));
          if (lane == 0 && accepted) atomicAdd(a.cand_count + a.item_query[it], accepted);
        }
        int run = 0;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          const bool pass = bits(g, r) <= tau;
          const u64 mask = __ballot(pass);
          if (mask != 0ull) {
            if (pass) dst[run + lanes_below(mask)] = ((u64)bits(g, r) << 32) | (u64)(uint32_t)pid[r];
            run += __popcll(mask);
          }
        }
        if (lane == 0) a.surv_count[region] = run;
      }
    }
  }
  // publish the next entry's descriptor
  if (wave == 0 && ngid_l >= 0) {
    const int cntn = n_gc & 0xff, chn = n_gc >> 8;
    const int b0 = n_b0 + chn * FUSED_UNIT_BLOCKS;
    int nbn = n_b1 - b0;
    if (nbn > FUSED_UNIT_BLOCKS) nbn = FUSED_UNIT_BLOCKS;
    if (lane < G) dsc[nb * 16 + lane] = n_item;
    if (lane == 0) { dsc[32 + nb * 8 + 1] = cntn; dsc[32 + nb * 8 + 2] = b0; dsc[32 + nb * 8 + 3] = nbn; dsc[32 + nb * 8 + 4] = chn; }
  }
  __syncthreads();   // the LDS regions are reused by the next work entry; its descriptor is visible
  tick(4);   // selection
  pt[7] += 1;
  if (ngid_l < 0) break;
  cur = nb;
  }  // persistent loop
  if (a.prof && tid == 0) {
    for (int i = 0; i < 8; ++i) a.prof[(size_t)blockIdx.x * 8 + i] = pt[i];
    a.prof[(size_t)blockIdx.x * 8 + 6] = clock64();
  }
}

// ---------------------------------------------------------------------------------------
// merge + replay over survivor buffers (same contract as merge_replay_kernel)
// ---------------------------------------------------------------------------------------
struct MergeSurvArgs {
  const u64* surv;             // [n_active*W][upi][8][512]
  const int32_t* surv_count;   // [n_active*W][upi][8], zero for the regions no workgroup wrote
  const int32_t* active;
  const int32_t* round_rows;
  const int32_t* cand_count;
  int32_t* out_ids;
  float* out_dist;
  int32_t* found;
  int32_t* next_active;
  int32_t* n_next;
  int32_t* status;
  int n_active, W, upi, L, k, found_rule, first_round;
  float sentinel;
};

// One wave per query.  Lane <-> survivor region: the query's W items x upi chunks x 8 waves regions
// mostly hold one or two keys each, so the lanes walk their own regions in lock step and feed the
// streaming selection one key per lane and step.
__global__ __launch_bounds__(64) void merge_surv_kernel(MergeSurvArgs a) {
  __shared__ u64 stage[64];
  const int x = blockIdx.x, lane = threadIdx.x;
  const int q = a.active ? a.active[x] : x;
  const int k = a.k;

  WaveSelect<1> sel;
  sel.init(stage, KEY_INF, a.L);
  const int per_item = a.upi * FUSED_NW;
  const int R = a.W * per_item;
  constexpr int NBATCH = 4;   // region rounds whose (dependent) descriptor loads are issued together
  for (int jb = 0; jb < R; jb += 64 * NBATCH) {
    int c[NBATCH];
    size_t region[NBATCH];
#pragma unroll
    for (int u = 0; u < NBATCH; ++u) {
      const int j = jb + u * 64 + lane;
      region[u] = (size_t)x * R + (size_t)(j < R ? j : 0);
      c[u] = (j < R) ? a.surv_count[region[u]] : 0;
    }
    // a region typically holds 0-2 keys: fetch the first two of every region up front (independent
    // loads), only longer regions go back to memory inside the loop
    u64 k0[NBATCH], k1[NBATCH];
#pragma unroll
    for (int u = 0; u < NBATCH; ++u) {
      const u64* src = a.surv + region[u] * (size_t)(FUSED_RMAX * 64);
      k0[u] = (c[u] > 0) ? src[0] : KEY_INF;
      k1[u] = (c[u] > 1) ? src[1] : KEY_INF;
    }
#pragma unroll
    for (int u = 0; u < NBATCH; ++u) {
      const u64* src = a.surv + region[u] * (size_t)(FUSED_RMAX * 64);
      int maxc = c[u];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o, 64));
      if (maxc > 0) sel.push(k0[u], c[u] > 0);
      if (maxc > 1) sel.push(k1[u], c[u] > 1);
      for (int t = 2; t < maxc; ++t) {
        const bool valid = t < c[u];
        sel.push(valid ? src[t] : KEY_INF, valid);
      }
    }
  }
  sel.finish();
  u64 byp = (sel.acc[0] == KEY_INF || lane >= a.L) ? KEY_INF : ((sel.acc[0] << 32) | (sel.acc[0] >> 32));
  byp = wave_sort64(byp);
  // lane i = slot i of the carried list (k <= 32 on this path); candidates replayed in scan order
  float d_slot = (a.first_round || lane >= k) ? a.sentinel : a.out_dist[(size_t)q * k + lane];
  int32_t id_slot = (a.first_round || lane >= k) ? -1 : a.out_ids[(size_t)q * k + lane];
  wave_list_replay(d_slot, id_slot, k, byp, a.L, [](uint32_t hi) { return (int32_t)hi; });
  if (lane < k) {
    a.out_ids[(size_t)q * k + lane] = id_slot;
    a.out_dist[(size_t)q * k + lane] = d_slot;
  }
  if (lane == 0) {
    int f = a.first_round ? 0 : a.found[q];
    const int rows = a.round_rows[x];
    f += (a.found_rule == 1 && a.cand_count) ? a.cand_count[q] : (rows > 0 ? rows : 0);
    a.found[q] = f;
    if (f < k && rows >= 0) {
      const int slot = atomicAdd(a.n_next, 1);
      a.next_active[slot] = q;
      if (a.status) a.status[0] = 1;
    }
  }
}

}  // namespace freddy
