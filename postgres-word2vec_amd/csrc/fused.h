// fused.h -- IVFADC: LUT build + ADC scan + candidate selection in ONE kernel (gfx950).
//
// Why: with separate kernels the per-(query, cell) LUTs (48 KiB each, 503 MB per 1024-query
// batch at nprobe=10) are written to memory by lut_build and read back by adc_scan -- the
// first rocprofv3 pass showed that round trip to be 2/3 of the scan kernel's traffic.  Here a
// LUT never exists as a whole: the workgroup walks the m positions two at a time, builds
// the two 4 KiB LUT slabs of G work units in LDS, and every lane immediately adds them to
// the running ADC sums of its rows, which live in registers.  The sum still runs over
// positions 0..m-1 in order (index_utils.c:1126-1133), each slab entry is still the
// sequential squareDistance over the sub-vector (index_utils.c:445-455, :500-508).
//
//   work item  = (query, probed cell); items are grouped by cell, <= 16 per workgroup
//   workgroup  = 512 threads (8 waves), one 4096-row chunk of the cell's list
//   LDS        = 2 buffers x 16 items x K floats = 128 KiB (K = 1024)
//
// Selection (replaces the per-wave streaming top-L of adc_scan): each lane's best key per unit
// goes through LDS; one wave per unit takes the column minima over the 16 waves, sorts those
// 64 keys once and uses the L-th as threshold tau.  The L smallest of 64 distinct candidates
// bound the L-th smallest of all from above, so {key <= tau} is a superset of the unit's L
// smallest keys (typically L + a few, never more than 64*L).  Survivors are appended to the
// item's buffer in memory; merge_replay picks the query's 2k smallest and replays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wave_topk.h"

namespace freddy {

static constexpr int FUSED_T = 512;
static constexpr int FUSED_NW = FUSED_T / 64;
static constexpr int FUSED_G = 16;                                 // (query, cell) items per workgroup
static constexpr int FUSED_RMAX = 8;
static constexpr int FUSED_E = 2;                                  // codes per lane: K <= 1024
static constexpr int FUSED_UNIT_BLOCKS = FUSED_RMAX * FUSED_NW;   // 64 row blocks = 4096 rows per chunk

// ---------------------------------------------------------------------------------------
// Cell-major grouping of the round's (query, cell) items: the probe plan counts items per
// cell; group_table turns the counts into offsets and into groups of <= FUSED_G items of
// one cell; bucket_items scatters the items into cell order.  (Order inside a cell is
// irrelevant: every item is selected and merged on its own.)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void group_table_kernel(const int32_t* __restrict__ cell_count, int C,
                                                        int32_t* __restrict__ cell_start,   // [C]
                                                        int32_t* __restrict__ group_cell,   // [max groups]
                                                        int32_t* __restrict__ group_first,  // index into sorted items
                                                        int32_t* __restrict__ group_cnt,
                                                        int32_t* __restrict__ n_groups) {
  const int lane = threadIdx.x;
  const int per = (C + 63) / 64;
  const int c0 = lane * per, c1 = (c0 + per < C) ? c0 + per : C;
  int items = 0, groups = 0;
  for (int c = c0; c < c1; ++c) {
    const int n = cell_count[c];
    items += n;
    groups += (n + FUSED_G - 1) / FUSED_G;
  }
  int it_off = items, gr_off = groups;   // inclusive wave scan
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int a = __shfl_up(it_off, d, 64), b = __shfl_up(gr_off, d, 64);
    if (lane >= d) { it_off += a; gr_off += b; }
  }
  if (lane == 63) n_groups[0] = gr_off;
  it_off -= items;
  gr_off -= groups;
  for (int c = c0; c < c1; ++c) {
    const int n = cell_count[c];
    cell_start[c] = it_off;
    for (int f = 0; f < n; f += FUSED_G) {
      group_cell[gr_off] = c;
      group_first[gr_off] = it_off + f;
      group_cnt[gr_off] = (n - f < FUSED_G) ? n - f : FUSED_G;
      ++gr_off;
    }
    it_off += n;
  }
}

__global__ __launch_bounds__(256) void bucket_items_kernel(const int32_t* __restrict__ item_cell, int n_items,
                                                          const int32_t* __restrict__ cell_start,
                                                          int32_t* __restrict__ cell_fill,
                                                          int32_t* __restrict__ sorted_item) {
  const int it = blockIdx.x * 256 + threadIdx.x;
  if (it >= n_items) return;
  const int c = item_cell[it];
  if (c < 0) return;
  sorted_item[cell_start[c] + atomicAdd(cell_fill + c, 1)] = it;
}

struct FusedArgs {
  const float* resid;          // [items][d] residuals (freddy.c:296-303)
  const int32_t* item_query;   // [items]
  const int32_t* sorted_item;  // items in cell order
  const int32_t* group_cell;   // [groups]
  const int32_t* group_first;
  const int32_t* group_cnt;
  const int32_t* n_groups;     // [1]
  const float* cbT;            // [m][S][K]
  const int32_t* blk_off;      // [C+1]
  const uint32_t* packed;      // [blocks][M2][64]
  const int32_t* pos;          // [blocks*64]
  u64* surv;                   // [items][cap]
  int32_t* surv_count;         // [items]
  int32_t* cand_count;         // [Q] or NULL
  int d, K, L, cap;
  uint32_t sentinel_bits;
  uint32_t desc_offset;        // byte offset of the item-descriptor scratch inside dynamic LDS
  uint32_t ablate;             // timing experiments only (FREDDY_GPU_FUSED_ABLATE)
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
  const int gid = blockIdx.x, chunk = blockIdx.y;
  if (gid >= a.n_groups[0]) return;
  const int cell = a.group_cell[gid];
  const int first = a.group_first[gid];
  const int cnt = a.group_cnt[gid];
  const int blk0 = a.blk_off[cell] + chunk * FUSED_UNIT_BLOCKS;
  int nblk = a.blk_off[cell + 1] - blk0;
  if (nblk <= 0) return;
  if (nblk > FUSED_UNIT_BLOCKS) nblk = FUSED_UNIT_BLOCKS;

  int32_t* desc = reinterpret_cast<int32_t*>(smem + a.desc_offset);   // [G] item ids
  if (tid < G) desc[tid] = (tid < cnt) ? a.sorted_item[first + tid] : -1;
  __syncthreads();

  typedef float v2f __attribute__((ext_vector_type(2)));
  float acc[G][RMAX];
  uint32_t cw[RMAX];
  v2f cb[S];   // .x: code tid, .y: code tid + T  (packed so the two chains run as v_pk_* ops)
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[g][r] = 0.0f;

  // rows past the end of the chunk re-read its last block (always in bounds); masked at the end
  auto row_block = [&](int r) {
    const int b = r * NW + wave;
    return (uint32_t)(blk0 + (b < nblk - 1 ? b : nblk - 1));
  };
  static_assert(E == 2, "two codes per lane");
  auto load_cb = [&](int p) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const uint32_t base = ((uint32_t)p * S + j) * (uint32_t)K;
      cb[j].x = (FULLK || tid < K) ? a.cbT[base + tid] : 0.0f;
      cb[j].y = (FULLK || tid + T < K) ? a.cbT[base + tid + T] : 0.0f;
    }
  };
  auto load_codes = [&](int pair) {
#pragma unroll
    for (int r = 0; r < RMAX; ++r) cw[r] = a.packed[(row_block(r) * M2 + (uint32_t)pair) * 64u + (uint32_t)lane];
  };
  auto slab_entry = [&](const float (&rc)[S], int g, float* dst) {
    // IEEE binary32 per component: v_pk_add/v_pk_mul round each half exactly like the scalar ops
    // blocks of 5 dimensions: the subs and muls of a block are independent, only the adds chain
    v2f sum = {0.0f, 0.0f};
    constexpr int JB = 5;
#pragma unroll
    for (int j0 = 0; j0 < S; j0 += JB) {
      v2f pr[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) {
        if (j0 + u < S) {
          const v2f rj = {rc[j0 + u], rc[j0 + u]};
          const v2f t = rj - cb[j0 + u];
          pr[u] = t * t;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < JB; ++u)
        if (j0 + u < S) sum = sum + pr[u];
    }
    if (FULLK || tid < K) dst[g * K + tid] = sum.x;
    if (FULLK || tid + T < K) dst[g * K + tid + T] = sum.y;
  };
  auto load_resid = [&](float (&rr)[S], int g, int p) {
    const int it = __builtin_amdgcn_readfirstlane(desc[g < cnt ? g : cnt - 1]);
    const float* r = a.resid + (size_t)it * a.d + (size_t)p * S;   // wave-uniform -> scalar loads
#pragma unroll
    for (int j = 0; j < S; ++j) rr[j] = r[j];
  };
  auto build_slab = [&](int p, float* dst) {
    // two residual buffers, alternating: item g+1's slice is fetched while item g is computed
    float ra[S], rb[S];
    load_resid(ra, 0, p);
#pragma unroll 1
    for (int g = 0; g < cnt; g += 2) {
      load_resid(rb, g + 1, p);
      slab_entry(ra, g, dst);
      load_resid(ra, g + 2, p);
      if (g + 1 < cnt) slab_entry(rb, g + 1, dst);
    }
  };
  auto gather = [&](int p, const float* cur) {
    const int sh = (p & 1) * 16;
    uint32_t code[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) code[r] = (cw[r] >> sh) & 0xffffu;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < cnt) {   // workgroup-uniform; only LDS reads inside
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[g][r] = acc[g][r] + cur[g * K + (int)code[r]];
      }
    }
  };

  load_cb(0);
  load_codes(0);
  build_slab(0, slab);
  if (M > 1) load_cb(1);
  lds_barrier();
  // Waves 0-3 build first and gather second, waves 4-7 the other way round (each SIMD hosts one
  // wave of each kind): while one half keeps the VALU busy with slab(p+1), the other half keeps
  // the LDS busy with the gathers of slab(p).  Both orders only read buffer p&1 and write the
  // other one, so one barrier per position still suffices.
  const bool gather_first = (wave >> 2) & 1;
  for (int p = 0; p < M; ++p) {
    float* nxt = slab + (size_t)((p + 1) & 1) * G * K;
    const float* cur = slab + (size_t)(p & 1) * G * K;
    if (gather_first) {
      if (!(a.ablate & 2)) gather(p, cur);
      __builtin_amdgcn_sched_barrier(0);
      if ((p & 1) && p + 1 < M && !(a.ablate & 8)) load_codes((p + 1) >> 1);
      __builtin_amdgcn_sched_barrier(0);
      if (p + 1 < M && !(a.ablate & 1)) build_slab(p + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      if (p + 2 < M && !(a.ablate & 8)) load_cb(p + 2);
    } else {
      if (p + 1 < M && !(a.ablate & 1)) build_slab(p + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      if (p + 2 < M && !(a.ablate & 8)) load_cb(p + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (!(a.ablate & 2)) gather(p, cur);
      __builtin_amdgcn_sched_barrier(0);
      if ((p & 1) && p + 1 < M && !(a.ablate & 8)) load_codes((p + 1) >> 1);
    }
    lds_barrier();
  }
  if (a.ablate & 4) return;

  // ---- selection -------------------------------------------------------------------------
  u64* exch = reinterpret_cast<u64*>(smem);          // [G][T], aliases the slabs (all reads done)
  u64* tau_s = exch + (size_t)G * T;                 // [G]
  const u64 sentinel_key = (u64)a.sentinel_bits << 32;
  int32_t pid[RMAX];
  bool live[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    pid[r] = a.pos[row_block(r) * 64u + (uint32_t)lane];
    live[r] = ((r * NW + wave) < nblk) && pid[r] >= 0;
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    u64 mn = KEY_INF;
    if (g < cnt) {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        const u64 key = make_key(acc[g][r], (uint32_t)pid[r]);
        if (live[r] && key < mn) mn = key;
      }
    }
    exch[(size_t)g * T + tid] = mn;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < G / NW; ++h) {   // each wave finds the threshold of G/NW items
    const int g = wave + h * NW;
    u64 col = KEY_INF;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) col = umin64(col, exch[(size_t)g * T + w2 * 64 + lane]);
    col = wave_sort64(col);
    const u64 t = __shfl(col, a.L - 1, 64);
    if (lane == 0) tau_s[g] = t;
  }
  __syncthreads();
  const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g < cnt) {
      const u64 tau = tau_s[g];
      const int it = __builtin_amdgcn_readfirstlane(desc[g]);
      int accepted = 0;
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        const u64 key = make_key(acc[g][r], (uint32_t)pid[r]);
        const bool ok = live[r] && key < sentinel_key;
        accepted += __popcll(__ballot(ok));
        const bool pass = ok && key <= tau;
        const u64 mask = __ballot(pass);
        const int n = __popcll(mask);
        if (n) {
          int base = 0;
          if (lane == 0) base = atomicAdd(a.surv_count + it, n);
          base = __shfl(base, 0, 64);
          const int idx = base + __popcll(mask & lt);
          if (pass && idx < a.cap) a.surv[(size_t)it * a.cap + idx] = key;
        }
      }
      if (a.cand_count && lane == 0 && accepted) atomicAdd(a.cand_count + a.item_query[it], accepted);
    }
  }
}

// ---------------------------------------------------------------------------------------
// merge + replay over survivor buffers (same contract as merge_replay_kernel)
// ---------------------------------------------------------------------------------------
struct MergeSurvArgs {
  const u64* surv;             // [n_active*W][cap]
  const int32_t* surv_count;   // [n_active*W]
  const int32_t* active;
  const int32_t* round_rows;
  const int32_t* cand_count;
  int32_t* out_ids;
  float* out_dist;
  int32_t* found;
  int32_t* next_active;
  int32_t* n_next;
  int32_t* status;
  int n_active, W, cap, L, k, found_rule, first_round;
  float sentinel;
};

__global__ __launch_bounds__(64) void merge_surv_kernel(MergeSurvArgs a) {
  __shared__ u64 stage[64];
  __shared__ u64 cand[64];
  __shared__ int32_t s_id[32];
  __shared__ float s_d[32];
  const int x = blockIdx.x, lane = threadIdx.x;
  const int q = a.active ? a.active[x] : x;
  const int k = a.k;

  WaveSelect<1> sel;
  sel.init(stage, KEY_INF, a.L);
  for (int i = 0; i < a.W; ++i) {
    const int it = x * a.W + i;
    int cnt = a.surv_count[it];
    cnt = cnt > a.cap ? a.cap : cnt;
    const u64* src = a.surv + (size_t)it * a.cap;
    for (int base = 0; base < cnt; base += 64) {
      const bool valid = base + lane < cnt;
      const u64 key = valid ? src[base + lane] : KEY_INF;
      sel.push(key, valid);
    }
  }
  sel.finish();
  u64 byp = (sel.acc[0] == KEY_INF || lane >= a.L) ? KEY_INF : ((sel.acc[0] << 32) | (sel.acc[0] >> 32));
  byp = wave_sort64(byp);
  cand[lane] = byp;
  for (int i = lane; i < k; i += 64) {
    s_id[i] = a.first_round ? -1 : a.out_ids[(size_t)q * k + i];
    s_d[i] = a.first_round ? a.sentinel : a.out_dist[(size_t)q * k + i];
  }
  __syncthreads();
  if (lane == 0) {
    float maxd = s_d[k - 1];
    for (int e = 0; e < a.L; ++e) {
      const u64 c = cand[e];
      if (c == KEY_INF) break;
      const float dist = __uint_as_float((uint32_t)c);
      if (dist < maxd) {
        int slot = k - 1;                                // updateTopK, index_utils.c:19-33
        while (slot >= 0 && !(s_d[slot] < dist)) --slot;
        ++slot;
        for (int t = k - 2; t >= slot; --t) { s_d[t + 1] = s_d[t]; s_id[t + 1] = s_id[t]; }
        s_d[slot] = dist;
        s_id[slot] = (int32_t)(uint32_t)(c >> 32);
        maxd = s_d[k - 1];
      }
    }
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
  __syncthreads();
  for (int i = lane; i < k; i += 64) {
    a.out_ids[(size_t)q * k + i] = s_id[i];
    a.out_dist[(size_t)q * k + i] = s_d[i];
  }
}

}  // namespace freddy
