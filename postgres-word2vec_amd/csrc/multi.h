// multi.h -- the cell-grouped scan for EVERY OTHER index shape: the reference's arithmetic for every probed row, the rows of a cell
// read once for up to eight queries.
//
// The reference's primitives are shape-generic (getPrecomputedDistances index_utils.c:445-455, computePQDistanceInt16 :1126-1133)
// and it ships an index configuration with m = 5, K = 256, 25 dimensions, 32 cells (index_creation/config/
// ivfadc_complete_config.json:2-4); the filter + refine scan (fused5.h) is built for m = 12, S = 25.  Until round 5 every other
// shape took lut_build_kernel + adc_scan_kernel: one workgroup per (query, cell chunk), the cell's rows read again for every
// query that probes it and a streaming selection per row -- 466 us per 1024 queries on that shipped shape.  Here a work entry is
// (<= 8 items of ONE cell, one 4096-row chunk), as in the other cell-grouped scans:
//   * the items' exact LUTs (lut_build_kernel: squareDistance per (position, code), residual formed in the kernel) are staged in
//     LDS interleaved, slab[position][code][8 items] -- a row's eight values of a position are two ds_read_b128;
//   * a lane holds 8 rows x 8 items of sums, position by position in the reference's order (dist = 0.0f; dist += preDists[...]):
//     the sums ARE the reference's distances, bit for bit;
//   * selection as in fused3.h: column minima -> the L-th smallest -> rows <= it (and below the sentinel) go to the wave's
//     survivor region of the item; merge_surv_kernel selects and replays.
// Shapes: m * K * 32 B of slab within the LDS budget (m = 5 / K = 256: 40 KB; m = 12 / K = 256: 96 KB; the ivpq shape m = 30 /
// K = 32: 30 KB), any sub-vector length, 2k <= 64, lists of <= 8 chunks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_common.h"

namespace freddy {

static constexpr int MULTI_G = 8;       // items per work entry
static constexpr int MULTI_T = 512;     // eight waves: wave w <-> row blocks w, w + 8, ... of the chunk

struct MultiArgs {
  const float* lut;            // [items][m][K] exact LUTs (lut_build_kernel)
  const int32_t* item_query;   // [items]
  const int32_t* sorted_item;  // items in cell order
  const int32_t* group_cell;   // [entries]
  const int32_t* group_first;
  const int32_t* group_cnt;    // items | chunk << 8
  const int32_t* n_groups;     // [1]
  int32_t* work_counter;       // [1] zeroed before the launch
  const int32_t* blk_off;      // [C+1]
  const uint32_t* packed;      // [blocks][M2][64]
  const int32_t* pos;          // [blocks*64]
  u64* surv;                   // [items][upi][8 waves][512]
  int32_t* surv_count;
  int32_t* cand_count;         // [Q] or NULL
  int m, M2, K, L, upi;
  uint32_t sentinel_bits;
};

static inline size_t multi_lds_bytes(int m, int K) { return (size_t)m * K * MULTI_G * sizeof(float) + MULTI_G * 64 * 4 + MULTI_G * 4 + 64; }

__global__ __launch_bounds__(MULTI_T) void ivf_multi_kernel(MultiArgs a) {
  constexpr int G = MULTI_G, RMAX = FUSED_RMAX, NW = FUSED_NW;
  static_assert(NW * 64 == MULTI_T && G == 8, "one wave per item in the threshold step, two 16-byte reads per slab row");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);                                           // [m][K][G]
  const int mK = a.m * a.K;
  uint32_t* colmin = reinterpret_cast<uint32_t*>(smem + (size_t)mK * G * sizeof(float));  // [G][64]
  uint32_t* tau_s = colmin + G * 64;                                                      // [G]
  int32_t* hdr = reinterpret_cast<int32_t*>(tau_s + G);                                   // [0] the entry's number
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_work = a.n_groups[0];
  for (int i = tid; i < G * 64; i += MULTI_T) colmin[i] = 0xffffffffu;
  for (;;) {
    __syncthreads();   // (the previous entry's survivors are out: slab, thresholds and the header may change)
    if (tid == 0) hdr[0] = atomicAdd(a.work_counter, 1);
    __syncthreads();
    const int e = hdr[0];
    if (e >= n_work) return;
    const int cell = a.group_cell[e], first = a.group_first[e], gc = a.group_cnt[e];
    const int cnt = gc & 0xff, chunk = gc >> 8;
    const int blk0 = a.blk_off[cell] + chunk * FUSED_UNIT_BLOCKS;
    int nblk = a.blk_off[cell + 1] - blk0;
    nblk = nblk > FUSED_UNIT_BLOCKS ? FUSED_UNIT_BLOCKS : nblk;
    // the items' LUTs, interleaved: slab[j][g] = lut[item g][j], j = position * K + code (slots beyond the entry's items keep
    // whatever they held: their sums are never looked at)
    int item[G];
#pragma unroll
    for (int g = 0; g < G; ++g) item[g] = a.sorted_item[first + (g < cnt ? g : 0)];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < cnt) {
        const float* src = a.lut + (size_t)item[g] * mK;
        for (int j = tid; j < mK; j += MULTI_T) slab[(size_t)j * G + g] = src[j];
      }
    }
    __syncthreads();
    // the rows: wave w <-> blocks r * 8 + w of the chunk (past its end: the last block again, masked below)
    auto row_block = [&](int r) {
      const int bl = r * NW + wave;
      return (uint32_t)(blk0 + (bl < nblk - 1 ? bl : nblk - 1));
    };
    float acc[G][RMAX];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < RMAX; ++r) acc[g][r] = 0.0f;
    for (int pair = 0; pair < a.M2; ++pair) {
      uint32_t cw[RMAX];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) cw[r] = a.packed[(row_block(r) * (uint32_t)a.M2 + (uint32_t)pair) * 64u + (uint32_t)lane];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int p = 2 * pair + half;
        if (p < a.m) {
          const float* sp = slab + (size_t)p * a.K * G;
#pragma unroll
          for (int r = 0; r < RMAX; ++r) {
            const uint32_t code = (cw[r] >> (16 * half)) & 0xffffu;
            const float4* row = reinterpret_cast<const float4*>(sp + (size_t)code * G);
            const float4 v0 = row[0], v1 = row[1];
            acc[0][r] = acc[0][r] + v0.x; acc[1][r] = acc[1][r] + v0.y; acc[2][r] = acc[2][r] + v0.z; acc[3][r] = acc[3][r] + v0.w;   // index_utils.c:1126-1133
            acc[4][r] = acc[4][r] + v1.x; acc[5][r] = acc[5][r] + v1.y; acc[6][r] = acc[6][r] + v1.z; acc[7][r] = acc[7][r] + v1.w;
          }
        }
      }
    }
    int32_t pid[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) pid[r] = a.pos[row_block(r) * 64u + (uint32_t)lane];
    // Selection on the distance bits (sums of squares: >= +0, so the bits order like the floats).  Padding rows and the slots
    // past the chunk's last block are parked above everything.
    auto bits = [&](int g, int r) { return __float_as_uint(acc[g][r]); };
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const bool dead = !(((r * NW + wave) < nblk) && pid[r] >= 0);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g][r] = dead ? __uint_as_float(0xffffffffu) : acc[g][r];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < cnt) {
        uint32_t best = bits(g, 0);
#pragma unroll
        for (int r = 1; r < RMAX; ++r) best = min(best, bits(g, r));
        atomicMin(colmin + g * 64 + lane, best);
      }
    }
    __syncthreads();
    {   // thresholds: wave g <-> item g.  Survivors are {bits <= tau and bits < sentinel}: one bound
      uint32_t c = colmin[wave * 64 + lane];
      c = wave_sort32(c);
      const uint32_t t = __shfl(c, a.L - 1, 64);
      if (lane == 0) tau_s[wave] = min(t, a.sentinel_bits - 1u);
      colmin[wave * 64 + lane] = 0xffffffffu;   // ready for the next entry
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < cnt) {
        const uint32_t tau = tau_s[g];
        const size_t region = ((size_t)item[g] * a.upi + chunk) * NW + wave;
        u64* dst = a.surv + region * (size_t)(RMAX * 64);
        if (a.cand_count) {   // freddy.c:971 counts the rows that pass the sentinel guard
          int accepted = 0;
#pragma unroll
          for (int r = 0; r < RMAX; ++r) accepted += __popcll(__ballot(bits(g, r) < a.sentinel_bits));
          if (lane == 0 && accepted) atomicAdd(a.cand_count + a.item_query[item[g]], accepted);
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
}

}  // namespace freddy
