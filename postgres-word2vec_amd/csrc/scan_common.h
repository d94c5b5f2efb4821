// scan_common.h -- what the fused IVFADC scan kernels share: chunk geometry, the cell-major work table,
// the argument block of the exact scan (fused3.h), the LDS-only barrier and the survivor merge.
//
// Both scans (fused3.h: the reference's arithmetic for every row; fused5.h: filter + refine, default) walk
// work entries = (<= 12 items of ONE cell, one 4096-row chunk of its list) with persistent workgroups;
// every gatherer wave appends the rows that pass the item's threshold to its own survivor region, and a
// merge kernel picks the query's 2k smallest keys and replays the reference's insertion.
// (The first two generations of the exact scan -- symmetric 8-wave kernel, one builder wave per SIMD --
// are in the history of this repository: fused.h / fused2.h up to round 1.)
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
static __global__ __launch_bounds__(1024) void work_table_kernel(const int32_t* __restrict__ cell_count, int C, int cell_cap, int gsz,
                                                         const int32_t* __restrict__ blk_off, int32_t* __restrict__ out_cell,
                                                         int32_t* __restrict__ out_first, int32_t* __restrict__ out_cnt,
                                                         int32_t* __restrict__ n_groups, int cost_mode,
                                                         int sparse_max = 0, int32_t* __restrict__ sp_cell = nullptr,
                                                         int32_t* __restrict__ sp_first = nullptr, int32_t* __restrict__ sp_chunk = nullptr,
                                                         int32_t* __restrict__ n_sparse = nullptr, int sp_pairs = 0) {
  constexpr int NB = 128;   // cost classes, descending (cost_mode 0 uses FUSED_G * 4 + 4 of them: (items, quarter of a full chunk))
  constexpr int T = 1024, CPT = 4;   // the first T * CPT cells are read once and kept in registers for both sweeps
  __shared__ int hist[NB];
  __shared__ int start[NB];
  __shared__ int sp_n;   // (item, chunk) units of the cells with <= sparse_max items: sparse5.h takes them one by one
  const int tid = threadIdx.x;
  for (int i = tid; i < NB; i += T) hist[i] = 0;
  if (tid == 0) sp_n = 0;
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
  // (j, J): a cell's groups of items are dealt to J threads -- few cells with many items each (the reference's shipped 32-cell
  // configuration: 64 items per cell at 1024 queries x 2 probes, ten chunks per list) left the table to 32 of 1024 threads: 55 us
  auto cell = [&](bool emit, int c, int n, int nblk, int j = 0, int J = 1) {
    if (n == 0) return;
    const int chunks = (nblk + FUSED_UNIT_BLOCKS - 1) / FUSED_UNIT_BLOCKS;
    if (n <= sparse_max) {
      if (j != 0) return;
      // sp_pairs: units of two items (and a last one of one: sparse_pair5_kernel reads a unit's rows once for both);
      // bits 8.. of sp_chunk = items of the unit
      const int per = sp_pairs ? 2 : 1;
      if (!emit)
        for (int f = 0; f < n; f += per)
          for (int ch = 0; ch < chunks; ++ch) {
            const int slot = atomicAdd(&sp_n, 1);
            sp_cell[slot] = c;
            sp_first[slot] = c * cell_cap + f;
            sp_chunk[slot] = ch | ((n - f < per ? n - f : per) << 8);
          }
      return;
    }
    for (int f = j * gsz; f < n; f += J * gsz) {
      const int cnt = (n - f < gsz) ? n - f : gsz;
      for (int ch = 0; ch < chunks; ++ch) {
        int nb = nblk - ch * FUSED_UNIT_BLOCKS;
        nb = nb > FUSED_UNIT_BLOCKS ? FUSED_UNIT_BLOCKS : nb;
        const int rq = (nb * 4 - 1) / FUSED_UNIT_BLOCKS;          // 0..3
        // small class index = big entry.  cost_mode 0 (exact kernels): slab arithmetic grows with the items;
        // cost_mode 1 (filter kernel, LDS-bound): measured model in units of 100 cycles, selection tail +
        // 12 x max(builder phase, gather phase)
        int k = (FUSED_G - cnt) * 4 + (3 - rq);
        if (cost_mode == 1) {
          const int gp = 4 + (26 * (rq + 1) * ((cnt + 3) >> 2) + 5) / 10;
          const int cost = 50 + 7 * cnt + 12 * (gp > 15 ? gp : 15);   // 237 .. 554
          k = (560 - cost) / 3;
        } else if (cost_mode == 2) {   // integer-slab scan (fused5.h): 6 double phases, gathers per 8 items, <= 16 items
          // measured (units of 100 cycles): tail 15 + 11 per item; a double phase = max(builders 25, gathers 4 + 6.5 per
          // (quarter of a full chunk, 8 items))
          const int gp = 4 + (13 * (rq + 1) * ((cnt + 7) >> 3)) / 2;
          const int cost = 15 + 11 * cnt + 6 * (gp > 25 ? gp : 25);   // 176 .. 527
          k = (530 - cost) / 3;
        }
        k = k < 0 ? 0 : (k > NB - 1 ? NB - 1 : k);
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
  const int J = (C > 0 && C <= T / 2) ? T / C : 1;
  auto sweep = [&](bool emit) {
    if (J > 1) {   // thread <-> (cell tid % C, share tid / C of its groups)
      const int c = tid % C, j = tid / C;
      if (j < J) cell(emit, c, cell_count[c], blk_off[c + 1] - blk_off[c], j, J);
      return;
    }
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
    if (tid == 0 && n_sparse) n_sparse[0] = sp_n;
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
  long long* prof;             // NULL, or [gridDim.x][8] cycle sums per phase (FREDDY_GPU_FUSED_PROF)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc drain vmcnt too,
// which would expose the latency of every prefetch that is meant to fly across the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
static __global__ __launch_bounds__(64) void merge_surv_kernel(MergeSurvArgs a) {
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
