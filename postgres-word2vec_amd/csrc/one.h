// one.h -- ONE query as a single launch (gfx950): pq_one_kernel, ivf_one_kernel.
//
// The generic path answers a single pq_search query with three dependent launches -- lut_build_kernel, adc_scan_kernel,
// merge_replay_kernel: 14 + 22 + 14 us under HIP events for 28 MB of codes that sit in the caches -- i.e. mostly launch
// gaps.  pq_one_kernel is the same three stages in one grid of at most one workgroup per CU (all co-resident):
//   1. every workgroup computes its slice of the query's table lut[pos*K + code] = squareDistance(q_pos, cb[pos][code])
//      (index_utils.c:445-455; the sequential binary32 chain of lut_build_kernel) and publishes it;
//   2. every workgroup waits for the table, stages it in LDS and scans its chunk of row blocks exactly as adc_scan_kernel
//      does (position-order sums, index_utils.c:1126-1133; WaveSelect of the L = 2k smallest (distance, position) keys),
//      then publishes its L keys;
//   3. workgroup 0 collects the lists, merges them and forms the reference's list (merge_replay_kernel's tail: updateTopK +
//      "dist < maxDist", index_utils.c:19-33, freddy.c:128-131), writes it to mapped host memory and then the completion
//      word the host polls.
// Hand-offs carry no counters: every published word carries the call's epoch in its top bit (below, "hand-offs without
// counters").  All waits are bounded: a grid that cannot become co-resident reports through `err` instead of hanging (the
// host then takes the three-launch path).  ivf_one_kernel: further down.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "wave_topk.h"

namespace freddy {

struct OneArgs {
  float qv[300];             // the query, by value: it travels with the dispatch (no PCIe read at kernel start)
  const float* cbT;          // [m][S][K]
  float* lut_g;              // [m*K] workspace
  const int32_t* blk_off;    // [2] row blocks of the (one) list
  const uint32_t* packed;    // [blocks][M/2][64]
  const int32_t* pos;        // [blocks*64] scan position, -1 = padding
  const int32_t* pos_to_id;  // NULL: the position is the id
  u64* part;                 // [grid][L] workspace
  int32_t* out_ids;          // [k]
  float* out_dist;           // [k]
  uint32_t epoch;            // 0 / 1: the tag of this call's published words
  int32_t* err;              // [1] mapped host memory: 1 = a poll ran out; 2 = result written (the host may poll this word)
  unsigned long long* prof;  // debugging: phase stamps (100 MHz) of workgroup 0 [0..7] and of the last arriver [8..15]
  int K, L, k, chunk_blocks;
  float sentinel;
  uint32_t sentinel_bits;
};

static constexpr int ONE_WG = 512;
static constexpr int ONE_WAVES = ONE_WG / 64;
// ---- hand-offs without counters: every published word carries the call's epoch ----
// Distances are sums of squares: >= +0, so the sign bit of a published float -- and bit 63 of a published (distance,
// position) key -- is free.  A producer stores (value | epoch << 31) with a write-through store and moves on: no drain, no
// arrival counter.  A consumer loads with sc1 loads and accepts a word when its top bit equals the epoch, else loads again
// (bounded).  Every word validates itself, so nothing needs ordering.  The host flips the epoch from call to call as long
// as the calls' shapes are equal (then this call writes exactly the words the previous one wrote, all of which still carry
// the other epoch) and clears the buffer to epoch 0 when the shape changes (freddy_gpu.hip one_buffer()).
__device__ __forceinline__ uint32_t one_tag(float v, uint32_t ep) { return (__float_as_uint(v) & 0x7fffffffu) | (ep << 31); }
__device__ __forceinline__ u64 one_tag_key(u64 key, uint32_t ep) {   // KEY_INF travels as 0x7fff...f
  return ((key == KEY_INF ? 0x7fffffffffffffffull : key) & 0x7fffffffffffffffull) | ((u64)ep << 63);
}
__device__ __forceinline__ u64 one_untag_key(u64 w) {
  const u64 k = w & 0x7fffffffffffffffull;
  return k == 0x7fffffffffffffffull ? KEY_INF : k;
}
static constexpr int ONE_RETRY_LIMIT = 20000;   // rounds of >= 1 us each

// A table of tagged floats -> LDS (tags stripped) by six 16-byte sc1 loads per lane in ONE asm statement that ends with the
// wait (the compiler cannot see that an asm load's result arrives later); words [0, n_valid) are checked, the rest (padding up to a multiple of 4)
// is not.  Returns false when the table did not become valid within the bounded number of rounds.  All threads call.
__device__ __forceinline__ bool one_stage_tagged(const float* g, float* lds, int n_floats, int n_valid, uint32_t ep, int tid, int* retry_sh) {
  typedef uint32_t w4 __attribute__((ext_vector_type(4)));
  const w4* s16 = reinterpret_cast<const w4*>(g);
  w4* d16 = reinterpret_cast<w4*>(lds);
  const int n16 = n_floats >> 2;
  for (int round = 0; round < ONE_RETRY_LIMIT; ++round) {
    if (tid == 0) *retry_sh = 0;
    __syncthreads();
    bool bad = false;
    for (int i0 = 0; i0 < n16; i0 += ONE_WG * 6) {
      w4 t[6];
      const int i = i0 + tid;
      const w4* p0 = s16 + (i < n16 ? i : n16 - 1);
      const w4* p1 = s16 + (i + ONE_WG < n16 ? i + ONE_WG : n16 - 1);
      const w4* p2 = s16 + (i + 2 * ONE_WG < n16 ? i + 2 * ONE_WG : n16 - 1);
      const w4* p3 = s16 + (i + 3 * ONE_WG < n16 ? i + 3 * ONE_WG : n16 - 1);
      const w4* p4 = s16 + (i + 4 * ONE_WG < n16 ? i + 4 * ONE_WG : n16 - 1);
      const w4* p5 = s16 + (i + 5 * ONE_WG < n16 ? i + 5 * ONE_WG : n16 - 1);
      asm volatile(
          "global_load_dwordx4 %0, %6, off sc1\n\t"
          "global_load_dwordx4 %1, %7, off sc1\n\t"
          "global_load_dwordx4 %2, %8, off sc1\n\t"
          "global_load_dwordx4 %3, %9, off sc1\n\t"
          "global_load_dwordx4 %4, %10, off sc1\n\t"
          "global_load_dwordx4 %5, %11, off sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5])
          : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5)
          : "memory");
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int iu = i + u * ONE_WG;
        if (iu < n16) {
          const int f = iu * 4;
          bad |= (f < n_valid && (t[u].x >> 31) != ep) || (f + 1 < n_valid && (t[u].y >> 31) != ep) ||
                 (f + 2 < n_valid && (t[u].z >> 31) != ep) || (f + 3 < n_valid && (t[u].w >> 31) != ep);
          w4 v = t[u];
          v.x &= 0x7fffffffu; v.y &= 0x7fffffffu; v.z &= 0x7fffffffu; v.w &= 0x7fffffffu;
          d16[iu] = v;
        }
      }
    }
    if (bad) *retry_sh = 1;
    __syncthreads();
    const int again = *retry_sh;
    __syncthreads();   // (everyone has read the flag before the next round clears it)
    if (!again) return true;
    __builtin_amdgcn_s_sleep(16);
  }
  return false;
}

// Smallest key over the wave's lanes [0, 1 << STEPS) (the other lanes are ignored), in every lane.  The minimum of the
// distance words decides almost every time (32-bit exchanges); only lanes that tie on it compare their position words.
template <int STEPS>
__device__ __forceinline__ u64 one_wave_min(u64 key) {
  const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
  uint32_t m = hi;
#pragma unroll
  for (int st = 0; st < STEPS; ++st) m = min(m, (uint32_t)lane_xor((int)m, 1 << st));
  m = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
  const u64 in_range = STEPS == 6 ? ~0ull : ((1ull << (1 << (STEPS < 6 ? STEPS : 0))) - 1ull);
  const u64 tied = __ballot(hi == m) & in_range;
  uint32_t ml;
  if (__popcll(tied) == 1) {
    ml = (uint32_t)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(tied));
  } else {
    ml = hi == m ? lo : 0xffffffffu;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) ml = min(ml, (uint32_t)lane_xor((int)ml, 1 << st));
    ml = (uint32_t)__builtin_amdgcn_readfirstlane((int)ml);
  }
  return ((u64)m << 32) | (u64)ml;
}

// The L smallest keys of n <= 8 ascending lists held in LDS (list x: rows[r * stride + x], r = 0.., KEY_INF where a list
// ends), ascending, lane r <- rank r (KEY_INF beyond L): lane x walks list x, each round the smallest head is taken and
// its lane advances.  Keys are unique (the position is part of the key).
__device__ __forceinline__ u64 one_multiway(const u64* rows, int stride, int n, int L, int lane) {
  int h = 0;
  u64 head = lane < n ? rows[lane] : KEY_INF;
  u64 out = KEY_INF;
  for (int r = 0; r < L; ++r) {
    const u64 mn = one_wave_min<3>(head);
    if (mn == KEY_INF) break;   // (uniform)
    if (lane == r) out = mn;
    if (head == mn) {
      ++h;
      head = h < L ? rows[(size_t)h * stride + lane] : KEY_INF;
    }
  }
  return out;
}

// The same over up to 256 lists with ONE wave: lane x walks the lists x, x + 64, x + 128, x + 192 (rows[r * n + list]).
__device__ __forceinline__ u64 one_multiway4(const u64* rows, int n, int L, int lane) {
  int h[4] = {0, 0, 0, 0};
  u64 head[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) head[t] = lane + 64 * t < n ? rows[lane + 64 * t] : KEY_INF;
  u64 out = KEY_INF;
  for (int r = 0; r < L; ++r) {
    const u64 m01 = umin64(head[0], head[1]), m23 = umin64(head[2], head[3]);
    const u64 mine = umin64(m01, m23);
    const u64 mn = one_wave_min<6>(mine);
    if (mn == KEY_INF) break;   // (uniform)
    if (lane == r) out = mn;
    if (mine == mn) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (head[t] == mn) {
          ++h[t];
          head[t] = h[t] < L ? rows[(size_t)h[t] * n + lane + 64 * t] : KEY_INF;
        }
      }
    }
  }
  return out;
}

template <int S>
__global__ __launch_bounds__(ONE_WG) void pq_one_kernel(OneArgs a) {
  constexpr int M = 12, M2T = 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int flag_sh, retry_sh;
  const int K = a.K, lutN = M * K;
  float* lut = reinterpret_cast<float*>(smem);
  u64* stage = reinterpret_cast<u64*>(smem + (((size_t)lutN * 4 + 15) & ~(size_t)15));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int w = blockIdx.x, G = gridDim.x;
#define ONE_STAMP(i) do { if (a.prof && tid == 0 && w == 0) a.prof[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define ONE_STAMP_LAST(i) do { if (a.prof && tid == 0) a.prof[8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  ONE_STAMP(0);

  // (the first row blocks of every wave are requested between the drain of the table slice and the barrier's poll)
  const int b0 = a.blk_off[0] + w * a.chunk_blocks;
  int b1 = a.blk_off[1];
  if (b0 + a.chunk_blocks < b1) b1 = b0 + a.chunk_blocks;
  constexpr int PF = 8;
  RowBlock<M2T> ring[PF];
  auto fetch = [&](RowBlock<M2T>& rb, int blk) {
    const int bc = blk < b1 ? blk : b1 - 1;           // (past the end: a repeat of the last block, never used)
    const uint32_t* pk = a.packed + (size_t)bc * M2T * 64 + lane;
#pragma unroll
    for (int j = 0; j < M2T; ++j) rb.w[j] = pk[j * 64];
    rb.p = a.pos[(size_t)bc * 64 + lane];
  };
  int b = b0 + wave;
  // ---- 1. this workgroup's slice of the table: unit = (position, range of codes) ----
  {
    static_assert(M * S == 300, "the query travels as 300 floats");
    const int n_slot = G >= M ? G / M : 1, cw = (K + n_slot - 1) / n_slot;
    for (int unit = w; unit < M * n_slot; unit += G) {
      const int p = unit / n_slot, c0 = (unit - p * n_slot) * cw;   // (p: workgroup-uniform)
      const int c1 = c0 + cw < K ? c0 + cw : K;
      for (int c = c0 + tid; c < c1; c += ONE_WG) {
        float cb[S];
#pragma unroll
        for (int j = 0; j < S; ++j) cb[j] = a.cbT[((size_t)p * S + j) * K + c];
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < S; ++j) {
          const float t = a.qv[p * S + j] - cb[j];
          const float pr = t * t;
          acc = acc + pr;
        }
        __hip_atomic_store(reinterpret_cast<uint32_t*>(a.lut_g) + p * K + c, one_tag(acc, a.epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    ONE_STAMP(1);
    if (b0 < b1) {   // (the first row blocks of every wave: in flight while the table arrives)
#pragma unroll
      for (int u = 0; u < PF; ++u) fetch(ring[u], b + u * ONE_WAVES);
    }
    // the first word of every slice is polled (a few hundred bytes) until all carry this call's epoch; the full table is
    // then staged and verified word by word (and staged again in the rare case that a slice was still in flight)
    if (wave == 0) {
      bool ok = false;
      for (int round = 0; round < ONE_RETRY_LIMIT && !ok; ++round) {
        bool bad = false;
        for (int unit = lane; unit < M * n_slot; unit += 64) {
          const int p = unit / n_slot, c0 = (unit - p * n_slot) * cw;
          if (c0 < K) bad |= (__hip_atomic_load(reinterpret_cast<uint32_t*>(a.lut_g) + p * K + c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 31) != a.epoch;
        }
        ok = __ballot(bad) == 0ull;
        if (!ok) __builtin_amdgcn_s_sleep(8);
      }
      if (lane == 0) flag_sh = ok ? 1 : 0;
    }
    __syncthreads();
    if (!flag_sh) { if (tid == 0) a.err[0] = 1; return; }
    ONE_STAMP(2);
  }

  // ---- 2. the table -> LDS, this workgroup's chunk of row blocks ----
  WaveSelect<1> sel;
  const u64 sentinel_key = (u64)a.sentinel_bits << 32;   // key < this  <=>  dist < sentinel
  sel.init(stage + wave * 64, sentinel_key, a.L);
  if (b0 < b1) {   // workgroup-uniform
    if (!one_stage_tagged(a.lut_g, lut, lutN, lutN, a.epoch, tid, &retry_sh)) { if (tid == 0) a.err[0] = 1; return; }
    ONE_STAMP(3);
    for (; b < b1; b += PF * ONE_WAVES) {
      u64 keys[PF];
      uint32_t mn = 0xffffffffu;   // this lane's smallest distance (bits) among the group's valid rows
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const RowBlock<M2T> cur = ring[u];
        const int bu = b + u * ONE_WAVES;
        if (bu + PF * ONE_WAVES < b1) fetch(ring[u], bu + PF * ONE_WAVES);   // (wave-uniform)
        keys[u] = KEY_INF;
        if (bu < b1) {   // wave-uniform
          float dist = 0.0f;
#pragma unroll
          for (int l = 0; l < M; ++l) {
            const uint32_t code = (l & 1) ? (cur.w[l >> 1] >> 16) : (cur.w[l >> 1] & 0xffffu);
            dist = dist + lut[l * K + code];
          }
          if (cur.p >= 0) {
            keys[u] = make_key(dist, (uint32_t)cur.p);
            mn = min(mn, __float_as_uint(dist));
          }
        }
      }
      // L rows of the group are at most as far as the L-th smallest lane minimum: nothing farther can be among the L
      // smallest keys (one 32-bit sort instead of a 64-bit sort + merge per 64 passing keys while the threshold is loose)
      const uint32_t dL = (uint32_t)__builtin_amdgcn_readlane((int)wave_sort32(mn), a.L - 1);
      const u64 bound = ((u64)dL << 32) | 0xffffffffull;
      if (bound < sel.tau) sel.tau = bound;
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (b + u * ONE_WAVES < b1) sel.push(keys[u], keys[u] != KEY_INF);
    }
    sel.finish();
  }
  ONE_STAMP(4);
  // the eight waves' lists meet in LDS (the table is dead by now), wave 0 merges them and publishes the workgroup's list
  // rank-major: part[r * G + w]
  __syncthreads();
  u64* lists = reinterpret_cast<u64*>(smem);   // [64 ranks][ONE_WAVES]
  lists[lane * ONE_WAVES + wave] = sel.acc[0];
  __syncthreads();
  if (wave == 0) {
    const u64 mine = one_multiway(lists, ONE_WAVES, ONE_WAVES, a.L, lane);
    if (lane < a.L) __hip_atomic_store(a.part + (size_t)lane * G + w, one_tag_key(mine, a.epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  ONE_STAMP(5);
  if (w != 0) return;   // (workgroup 0 collects the lists)
  ONE_STAMP_LAST(0);

  // ---- 3. the last workgroup: multiway merge of the G lists (lane <-> list, 64 lists per wave), replay ----
  {
    u64* all = reinterpret_cast<u64*>(smem) + ONE_WAVES * 64;   // [L][G] as published; behind the waves' result rows
    const int total = G * a.L;
    {
      bool ok = false;
      for (int round = 0; round < ONE_RETRY_LIMIT && !ok; ++round) {
        if (tid == 0) retry_sh = 0;
        __syncthreads();
        bool bad = false;
        for (int i = tid; i < total; i += ONE_WG) {
          const u64 word = __hip_atomic_load(a.part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad |= (uint32_t)(word >> 63) != a.epoch;
          all[i] = one_untag_key(word);
        }
        if (bad) retry_sh = 1;
        __syncthreads();
        ok = retry_sh == 0;
        __syncthreads();   // (everyone has read the flag before the next round clears it)
        if (!ok) __builtin_amdgcn_s_sleep(8);
      }
      if (!ok) { if (tid == 0) a.err[0] = 1; return; }
    }
    // one wave, lane <-> the lists x = lane, lane + 64, ... (<= 4 for 256 CUs): each round takes the smallest head
    if (wave != 0) return;
    ONE_STAMP_LAST(1);
    const u64 top = one_multiway4(all, G, a.L, lane);
    // The reference's pass inserts in scan order behind a guard "dist < maxDist" (freddy.c:128-131): when no two of the 2k
    // candidates have the same distance, the order of insertion does not matter and the list is the k smallest in ascending
    // order (those below the sentinel).  Equal distances among them: re-key as (position, distance bits), sort, replay.
    const uint32_t db = (uint32_t)(top >> 32);
    const uint32_t db_next = (uint32_t)__shfl_down((int)db, 1, 64);
    const bool tie = lane + 1 < a.L && top != KEY_INF && db == db_next;   // (top is ascending: equal distances are neighbours; KEY_INF's word is not a distance)
    float d_slot = a.sentinel;
    int32_t id_slot = -1;
    if (__ballot(tie) == 0ull) {
      if (lane < a.k && top != KEY_INF && __uint_as_float(db) < a.sentinel) {
        d_slot = __uint_as_float(db);
        id_slot = a.pos_to_id ? a.pos_to_id[(uint32_t)top] : (int32_t)(uint32_t)top;
      }
    } else {
      u64 byp[1];
      byp[0] = (top == KEY_INF || lane >= a.L) ? KEY_INF : ((top << 32) | (top >> 32));
      wave_sort_full<1>(byp);
      if (a.pos_to_id && byp[0] != KEY_INF) byp[0] = ((u64)(uint32_t)a.pos_to_id[(uint32_t)(byp[0] >> 32)] << 32) | (u64)(uint32_t)byp[0];
      wave_list_replay(d_slot, id_slot, a.k, byp[0], a.L < 64 ? a.L : 64, [](uint32_t id) { return (int32_t)id; });
    }
    if (lane < a.k) {
      a.out_ids[lane] = id_slot;
      a.out_dist[lane] = d_slot;
    }
    ONE_STAMP_LAST(2);
    // the list is in (mapped host) memory before the word the host polls
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#undef ONE_STAMP
#undef ONE_STAMP_LAST

// ---------------------------------------------------------------------------------------
// ivfadc_search for ONE query as a single launch: the reference's own call shape (freddy.c:174-393).  The generic path is
// five dependent launches (coarse_small -> probe_plan -> lut_build -> adc_scan -> merge_replay).  Here:
//   A. every workgroup: the coarse distances of its <= ceil(C / G) cells (the centroid rows staged in LDS, one lane per
//      cell walks the 300 dimensions in order: squareDistance, index_utils.c:500-508) -> published -> grid barrier;
//   B. every workgroup, redundantly: the W nearest cells below the cell limit by the reference's guarded insertion in
//      cell order (probe_plan_kernel's selection: freddy.c:266-283) -- every workgroup arrives at the same list;
//   C. the W tables lut[i][pos*K + code] = squareDistance((q - coarse[cell_i])_pos, cb[pos][code]) (freddy.c:296-303,
//      index_utils.c:445-455) in slices -> published -> grid barrier;
//   D. workgroup (item, part): the item's table -> LDS, its share of the cell's row blocks (adc_scan_kernel's loop),
//      its L smallest keys -> published; the accepted-row count (freddy.c:971) by atomics;
//   E. the last arriver: merge, guarded insertion in scan (= id) order, the list; the query's "found" decides whether the
//      reference would probe again (freddy.c:377): then the host runs the multi-round path instead (flag 3).
// ---------------------------------------------------------------------------------------
struct IvfOneArgs {
  float qv[300];
  const float* coarse;       // [C][d]
  const float* cbT;          // [m][S][K]
  const int32_t* list_off;   // [C+1]
  const int32_t* blk_off;    // [C+1]
  const uint32_t* packed;
  const int32_t* pos;        // row ids (the scan position of an IVF table), -1 = padding
  float* dist_g;             // [C] workspace
  float* lut_g;              // [W][m*K] workspace
  u64* part;                 // [L][grid] workspace
  int32_t* out_ids;          // [k] mapped host memory
  float* out_dist;           // [k]
  uint32_t* cnt_g;           // [grid] workspace: every workgroup's accepted-row count (tagged)
  uint32_t epoch;            // 0 / 1: the tag of this call's published words (pq_one_kernel)
  int32_t* err;              // mapped host word: 1 = a poll ran out, 2 = list written, 3 = list written, the reference would probe again
  unsigned long long* prof;  // debugging: phase stamps (100 MHz) of workgroup 0 [0..9] and of the last arriver [10..12]
  int C, K, W, L, k, found_rule;
  float cell_limit, sentinel;
  uint32_t sentinel_bits;
};

template <int S>
__global__ __launch_bounds__(ONE_WG) void ivf_one_kernel(IvfOneArgs a) {
  constexpr int M = 12, M2T = 6, D = M * S;
  static_assert(D == 300, "the query travels as 300 floats");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int flag_sh, retry_sh, acc_sh;
  __shared__ int32_t cells_sh[32];
  const int K = a.K, lutN = M * K, C = a.C, W = a.W;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int w = blockIdx.x, G = gridDim.x;
#define ONE_STAMP(i) do { if (a.prof && tid == 0 && w == 0) a.prof[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define ONE_STAMP_LAST(i) do { if (a.prof && tid == 0) a.prof[10 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  ONE_STAMP(0);

  // ---- A. coarse distances of the cells w, w + G, ... ----
  {
    float* crow = reinterpret_cast<float*>(smem);   // [cells of this workgroup][D], then the query
    const int n_mine = w < C ? (C - w + G - 1) / G : 0;
    float* qs = crow + n_mine * D;
    for (int i = tid; i < n_mine * D; i += ONE_WG) {
      const int ci = i / D, j = i - ci * D;
      crow[i] = a.coarse[(size_t)(w + ci * G) * D + j];
    }
    if (tid < D) qs[tid] = a.qv[tid];
    __syncthreads();
    // one lane per cell, the cells of a workgroup on different waves; 16-byte LDS reads, the chain in dimension order
    for (int ci = wave; ci < n_mine; ci += ONE_WAVES) {
      if (lane == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* c4 = reinterpret_cast<const f4*>(crow + ci * D);
        const f4* q4 = reinterpret_cast<const f4*>(qs);
        float acc = 0.0f;
#pragma unroll 5
        for (int i = 0; i < D / 4; ++i) {
          const f4 cv = c4[i], qv = q4[i];
          { const float t = qv.x - cv.x; const float pr = t * t; acc = acc + pr; }
          { const float t = qv.y - cv.y; const float pr = t * t; acc = acc + pr; }
          { const float t = qv.z - cv.z; const float pr = t * t; acc = acc + pr; }
          { const float t = qv.w - cv.w; const float pr = t * t; acc = acc + pr; }
        }
        __hip_atomic_store(reinterpret_cast<uint32_t*>(a.dist_g) + (w + ci * G), one_tag(acc, a.epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (tid == 0) acc_sh = 0;
    __syncthreads();   // (the centroid rows in LDS are dead)
    ONE_STAMP(1);
  }

  // ---- B. the W nearest cells (every workgroup for itself) ----
  int my_rows = 0;
  bool any_cell = false;
  {
    float* dl = reinterpret_cast<float*>(smem);                       // [C rounded up to 4]
    u64* prow = reinterpret_cast<u64*>(smem + (((size_t)C * 4 + 63) & ~(size_t)15));   // [64] wave 0's staging row
    if (!one_stage_tagged(a.dist_g, dl, (C + 3) & ~3, C, a.epoch, tid, &retry_sh)) { if (tid == 0) a.err[0] = 1; return; }
    ONE_STAMP(2);
    if (wave == 0) {
      const int L2 = 2 * W;
      const u64 limit = (u64)__float_as_uint(a.cell_limit) << 32;
      // the L2-th smallest lane minimum bounds the L2-th smallest distance: the selection starts with a tight threshold
      uint32_t mn = 0xffffffffu;
      for (int j = lane; j < C; j += 64) mn = min(mn, __float_as_uint(dl[j]));
      const uint32_t dL = (uint32_t)__builtin_amdgcn_readlane((int)wave_sort32(mn), L2 - 1);
      const u64 bound = ((u64)dL << 32) | 0xffffffffull;
      WaveSelect<1> sel;
      sel.init(prow, bound < limit ? bound : limit, L2);
      for (int base = 0; base < C; base += 64) {
        const int j = base + lane;
        sel.push(make_key(dl[j < C ? j : C - 1], (uint32_t)j), j < C);
      }
      sel.finish();
      // the reference keeps its W nearest cells by guarded insertion in cell order (freddy.c:266-283): with no two of the
      // 2W candidates equally far that is the W smallest below the limit in ascending order; otherwise the replay
      const u64 top = lane < L2 ? sel.acc[0] : KEY_INF;
      const uint32_t db = (uint32_t)(top >> 32);
      const uint32_t db_next = (uint32_t)__shfl_down((int)db, 1, 64);
      const bool tie = lane + 1 < L2 && top != KEY_INF && db == db_next;
      float d_slot = a.cell_limit;
      int32_t c_slot = -1;
      if (__ballot(tie) == 0ull) {
        if (lane < W && top != KEY_INF && top < limit) { d_slot = __uint_as_float(db); c_slot = (int32_t)(uint32_t)top; }
      } else {
        u64 byp[1];
        byp[0] = top == KEY_INF ? KEY_INF : ((top << 32) | (top >> 32));
        wave_sort_full<1>(byp);
        wave_list_replay(d_slot, c_slot, W, byp[0], L2, [](uint32_t hi) { return (int32_t)hi; });
      }
      const bool have = lane < W && c_slot >= 0;
      my_rows = have ? (a.list_off[c_slot + 1] - a.list_off[c_slot]) : 0;   // (summed by the last arriver's wave 0)
      any_cell = __ballot(have) != 0ull;
      if (lane < 32) cells_sh[lane] = lane < W ? c_slot : -1;
    }
    __syncthreads();
  }

  ONE_STAMP(3);
  // this workgroup's share of the scan: item = w / P, part = w % P of the item's row blocks
  const int P = G >= W ? G / W : 1;
  const int my_item = w / P;
  const int my_cell = my_item < W ? cells_sh[my_item] : -1;
  int b0 = 0, b1 = 0;
  if (my_cell >= 0) {
    const int c0 = a.blk_off[my_cell], c1 = a.blk_off[my_cell + 1];
    const int chunk = (c1 - c0 + P - 1) / P;
    b0 = c0 + (w - my_item * P) * chunk;
    b1 = b0 + chunk < c1 ? b0 + chunk : c1;
  }
  constexpr int PF = 4;
  RowBlock<M2T> ring[PF];
  auto fetch = [&](RowBlock<M2T>& rb, int blk) {
    const int bc = blk < b1 ? blk : b1 - 1;
    const uint32_t* pk = a.packed + (size_t)bc * M2T * 64 + lane;
#pragma unroll
    for (int j = 0; j < M2T; ++j) rb.w[j] = pk[j * 64];
    rb.p = a.pos[(size_t)bc * 64 + lane];
  };
  int b = b0 + wave;

  // ---- C. the items' tables in slices: unit = (item, position, range of codes) ----
  {
    const int n_slot = G >= W * M ? G / (W * M) : 1, cw = (K + n_slot - 1) / n_slot;
    for (int unit = w; unit < W * M * n_slot; unit += G) {
      const int i = unit / (M * n_slot), rem = unit - i * (M * n_slot);
      const int p = rem / n_slot, c0 = (rem - p * n_slot) * cw;   // (i, p: workgroup-uniform)
      const int cell = __builtin_amdgcn_readfirstlane(cells_sh[i]);
      const int c1 = c0 + cw < K ? c0 + cw : K;
      if (cell < 0) {   // no such item this time: its words still get this call's epoch (the next call may read them)
        for (int c = c0 + tid; c < c1; c += ONE_WG)
          __hip_atomic_store(reinterpret_cast<uint32_t*>(a.lut_g) + (size_t)i * lutN + p * K + c, a.epoch << 31, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      const float* co = a.coarse + (size_t)cell * D + (size_t)p * S;
      for (int c = c0 + tid; c < c1; c += ONE_WG) {
        float cb[S];
#pragma unroll
        for (int j = 0; j < S; ++j) cb[j] = a.cbT[((size_t)p * S + j) * K + c];
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < S; ++j) {
          const float rj = a.qv[p * S + j] - co[j];   // (the residual: one binary32 subtraction, freddy.c:296-303)
          const float t = rj - cb[j];
          const float pr = t * t;
          acc = acc + pr;
        }
        __hip_atomic_store(reinterpret_cast<uint32_t*>(a.lut_g) + (size_t)i * lutN + p * K + c, one_tag(acc, a.epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    ONE_STAMP(4);
    if (b0 < b1) {   // (the first row blocks in flight while the item's table arrives)
#pragma unroll
      for (int u = 0; u < PF; ++u) fetch(ring[u], b + u * ONE_WAVES);
      // the first word of every slice of this workgroup's item is polled until all carry this call's epoch
      if (wave == 0) {
        bool ok = false;
        for (int round = 0; round < ONE_RETRY_LIMIT && !ok; ++round) {
          bool bad = false;
          for (int unit = lane; unit < M * n_slot; unit += 64) {
            const int p = unit / n_slot, c0 = (unit - p * n_slot) * cw;
            if (c0 < K) bad |= (__hip_atomic_load(reinterpret_cast<uint32_t*>(a.lut_g) + (size_t)my_item * lutN + p * K + c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 31) != a.epoch;
          }
          ok = __ballot(bad) == 0ull;
          if (!ok) __builtin_amdgcn_s_sleep(8);
        }
        if (lane == 0) flag_sh = ok ? 1 : 0;
      }
      __syncthreads();
      if (!flag_sh) { if (tid == 0) a.err[0] = 1; return; }
    }
    ONE_STAMP(5);
  }

  // ---- D. the item's table -> LDS, this workgroup's row blocks ----
  float* lut = reinterpret_cast<float*>(smem);
  u64* stage = reinterpret_cast<u64*>(smem + (((size_t)lutN * 4 + 15) & ~(size_t)15));
  WaveSelect<1> sel;
  const u64 sentinel_key = (u64)a.sentinel_bits << 32;
  sel.init(stage + wave * 64, sentinel_key, a.L);
  int accepted = 0;
  if (b0 < b1) {   // workgroup-uniform
    if (!one_stage_tagged(a.lut_g + (size_t)my_item * lutN, lut, lutN, lutN, a.epoch, tid, &retry_sh)) { if (tid == 0) a.err[0] = 1; return; }
    ONE_STAMP(6);
    for (; b < b1; b += PF * ONE_WAVES) {
      u64 keys[PF];
      uint32_t mn = 0xffffffffu;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const RowBlock<M2T> cur = ring[u];
        const int bu = b + u * ONE_WAVES;
        if (bu + PF * ONE_WAVES < b1) fetch(ring[u], bu + PF * ONE_WAVES);   // (wave-uniform)
        keys[u] = KEY_INF;
        if (bu < b1) {   // wave-uniform
          float dist = 0.0f;
#pragma unroll
          for (int l = 0; l < M; ++l) {
            const uint32_t code = (l & 1) ? (cur.w[l >> 1] >> 16) : (cur.w[l >> 1] & 0xffffu);
            dist = dist + lut[l * K + code];
          }
          if (cur.p >= 0) {
            keys[u] = make_key(dist, (uint32_t)cur.p);
            mn = min(mn, __float_as_uint(dist));
          }
          accepted += __popcll(__ballot(cur.p >= 0 && keys[u] < sentinel_key));
        }
      }
      const uint32_t dL = (uint32_t)__builtin_amdgcn_readlane((int)wave_sort32(mn), a.L - 1);
      const u64 bound = ((u64)dL << 32) | 0xffffffffull;
      if (bound < sel.tau) sel.tau = bound;
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (b + u * ONE_WAVES < b1) sel.push(keys[u], keys[u] != KEY_INF);
    }
    sel.finish();
  }
  ONE_STAMP(7);
  if (lane == 0 && accepted) atomicAdd(&acc_sh, accepted);
  __syncthreads();
  u64* lists = reinterpret_cast<u64*>(smem);   // [64 ranks][ONE_WAVES]
  lists[lane * ONE_WAVES + wave] = sel.acc[0];
  __syncthreads();
  if (wave == 0) {
    const u64 mine = one_multiway(lists, ONE_WAVES, ONE_WAVES, a.L, lane);
    if (lane < a.L) __hip_atomic_store(a.part + (size_t)lane * G + w, one_tag_key(mine, a.epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) __hip_atomic_store(a.cnt_g + w, (uint32_t)acc_sh | (a.epoch << 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  ONE_STAMP(8);
  if (w != 0) return;   // (workgroup 0 collects the lists)
  ONE_STAMP_LAST(0);

  // ---- E. the last workgroup: merge, replay, the list ----
  {
    u64* all = reinterpret_cast<u64*>(smem) + ONE_WAVES * 64;
    const int total = G * a.L;
    {
      bool ok = false;
      for (int round = 0; round < ONE_RETRY_LIMIT && !ok; ++round) {
        if (tid == 0) { retry_sh = 0; acc_sh = 0; }
        __syncthreads();
        bool bad = false;
        for (int i = tid; i < total; i += ONE_WG) {
          const u64 word = __hip_atomic_load(a.part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad |= (uint32_t)(word >> 63) != a.epoch;
          all[i] = one_untag_key(word);
        }
        int cnt = 0;
        for (int i = tid; i < G; i += ONE_WG) {
          const uint32_t word = __hip_atomic_load(a.cnt_g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad |= (word >> 31) != a.epoch;
          cnt += (int)(word & 0x7fffffffu);
        }
        if (cnt) atomicAdd(&acc_sh, cnt);
        if (bad) retry_sh = 1;
        __syncthreads();
        ok = retry_sh == 0;
        __syncthreads();   // (everyone has read the flag before the next round clears it)
        if (!ok) __builtin_amdgcn_s_sleep(8);
      }
      if (!ok) { if (tid == 0) a.err[0] = 1; return; }
    }
    if (wave != 0) return;
    ONE_STAMP_LAST(1);
    const u64 top = one_multiway4(all, G, a.L, lane);
    const uint32_t db = (uint32_t)(top >> 32);
    const uint32_t db_next = (uint32_t)__shfl_down((int)db, 1, 64);
    const bool tie = lane + 1 < a.L && top != KEY_INF && db == db_next;
    float d_slot = a.sentinel;
    int32_t id_slot = -1;
    if (__ballot(tie) == 0ull) {   // (no two candidates equally far: the order of insertion does not matter, pq_one_kernel)
      if (lane < a.k && top != KEY_INF && __uint_as_float(db) < a.sentinel) {
        d_slot = __uint_as_float(db);
        id_slot = (int32_t)(uint32_t)top;
      }
    } else {
      u64 byp[1];
      byp[0] = (top == KEY_INF || lane >= a.L) ? KEY_INF : ((top << 32) | (top >> 32));
      wave_sort_full<1>(byp);
      wave_list_replay(d_slot, id_slot, a.k, byp[0], a.L < 64 ? a.L : 64, [](uint32_t id) { return (int32_t)id; });
    }
    if (lane < a.k) {
      a.out_ids[lane] = id_slot;
      a.out_dist[lane] = d_slot;
    }
    // "found" of the first round (freddy.c:377 rows rule, :971 accepted rule): fewer than k and a cell was probed -> the
    // reference probes the next W cells; the host runs that path
    int rows = my_rows;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rows += __shfl_xor(rows, o, 64);
    if (!any_cell) rows = -1;   // no cell below the limit: the query retires
    const int found = a.found_rule == 1 ? acc_sh : (rows > 0 ? rows : 0);
    const int verdict = (found < a.k && rows >= 0) ? 3 : 2;
    ONE_STAMP_LAST(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(a.err, verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#undef ONE_STAMP
#undef ONE_STAMP_LAST

}  // namespace freddy
