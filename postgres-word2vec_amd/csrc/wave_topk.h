// wave_topk.h -- wavefront-wide (64 lanes) selection of the smallest 64-bit keys.
//
// A candidate is one 64-bit key  (float_bits(distance) << 32) | scan_position.
// Distances on this path are sums of squares (>= +0, never -0), so the IEEE bit pattern
// orders exactly like the float; the scan position (row id / row index) makes every key
// unique, i.e. the order is total.  That is what lets a parallel selection reproduce the
// reference's order-dependent insertion list (index_utils.c:19-33): the device selects
// the 2k smallest keys, a single lane then replays the reference's insertion over them
// in scan order (merge_replay kernel).
//
// The accumulator holds 64*V keys, ascending over index (v*64 + lane).  gfx950 only:
// wave size is 64, cross-lane moves are ds_bpermute/DPP via __shfl.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace freddy {

typedef unsigned long long u64;
static constexpr u64 KEY_INF = ~0ull;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ u64 make_key(float dist, uint32_t pos) {
  return ((u64)__float_as_uint(dist) << 32) | (u64)pos;
}
__device__ __forceinline__ float key_dist(u64 key) { return __uint_as_float((uint32_t)(key >> 32)); }
__device__ __forceinline__ uint32_t key_pos(u64 key) { return (uint32_t)key; }

// Value of lane (lane ^ j).  Strides 1, 2, 4 and 8 stay inside a row of 16 lanes and are DPP moves
// (quad_perm, row_half_mirror o quad_perm, row_ror:8) -- a few cycles instead of a trip through the LDS
// crossbar (ds_bpermute, ~100 cycles); strides 16 and 32 are gfx950's v_permlane16/32_swap.  So none of
// the 21 steps of a 64-key bitonic sort touches the LDS.  j must be a compile-time constant after unrolling.
__device__ __forceinline__ int lane_xor(int v, int j) {
  if (j == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  if (j == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  if (j == 4) {                                                                  // (i ^ 7) then (i ^ 3)
    const int t = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);      // row_half_mirror
    return __builtin_amdgcn_update_dpp(0, t, 0x1B, 0xF, 0xF, true);              // quad_perm [3,2,1,0]
  }
  if (j == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);   // row_ror:8
  // gfx950's row swaps (checked on the hardware with tools/lab/ubench5): v_permlane16_swap(v, v) leaves the
  // even row's value of each row pair in [0] and the odd row's in [1]; v_permlane32_swap the lower half's
  // in [0] and the upper half's in [1]
  if (j == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane_id() & 16) ? r[0] : r[1]);
  }
  if (j == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane_id() & 32) ? r[0] : r[1]);
  }
  return __shfl_xor(v, j, 64);
}
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long v, int j) {
  const uint32_t lo = (uint32_t)lane_xor((int)(uint32_t)v, j), hi = (uint32_t)lane_xor((int)(uint32_t)(v >> 32), j);
  return ((unsigned long long)hi << 32) | lo;
}

// number of set bits of a ballot mask below this lane (v_mbcnt: no per-lane mask register to keep alive)
__device__ __forceinline__ int lanes_below(u64 mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ u64 umin64(u64 a, u64 b) { return a < b ? a : b; }
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a < b ? b : a; }

// Full ascending bitonic sort of 64 keys, one per lane.
__device__ __forceinline__ u64 wave_sort64(u64 key) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const u64 other = lane_xor64(key, j);
      const bool up = ((lane & k) == 0);
      const bool lower = ((lane & j) == 0);
      key = (lower == up) ? umin64(key, other) : umax64(key, other);
    }
  }
  return key;
}

// Ascending bitonic sort of 64 32-bit keys, one per lane.
__device__ __forceinline__ uint32_t wave_sort32(uint32_t x) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t ox = (uint32_t)lane_xor((int)x, j);
      const bool take_min = (((lane & k) == 0) == ((lane & j) == 0));
      x = take_min ? min(x, ox) : max(x, ox);
    }
  }
  return x;
}

// Two independent ascending bitonic sorts of 64 32-bit keys (one key of each per lane), interleaved
// so that the cross-lane moves of one hide behind the other's.
__device__ __forceinline__ void wave_sort32_x2(uint32_t& x, uint32_t& y) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t ox = (uint32_t)lane_xor((int)x, j);
      const uint32_t oy = (uint32_t)lane_xor((int)y, j);
      const bool take_min = (((lane & k) == 0) == ((lane & j) == 0));
      x = take_min ? min(x, ox) : max(x, ox);
      y = take_min ? min(y, oy) : max(y, oy);
    }
  }
}

// Sort a bitonic sequence of 64*V keys (index v*64+lane) into ascending order.
template <int V>
__device__ __forceinline__ void wave_bitonic_merge(u64 (&a)[V]) {
  const int lane = lane_id();
  // strides >= 64 pair registers of the same lane (static indices after unrolling)
#pragma unroll
  for (int jv = V >> 1; jv > 0; jv >>= 1) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if ((v & jv) == 0) {
        const u64 lo = umin64(a[v], a[v | jv]);
        const u64 hi = umax64(a[v], a[v | jv]);
        a[v] = lo;
        a[v | jv] = hi;
      }
    }
  }
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    const bool lower = ((lane & j) == 0);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const u64 other = lane_xor64(a[v], j);
      a[v] = lower ? umin64(a[v], other) : umax64(a[v], other);
    }
  }
}

// acc: 64*V smallest keys so far, ascending.  batch: up to 64 new keys, one per lane
// (KEY_INF where empty), in any order.  Afterwards acc holds the 64*V smallest keys of
// the union, ascending.  (min(acc_i, reversed(batch)_i) keeps exactly the smallest
// 64*V of the union and is bitonic; only the top register can meet a real batch key.)
template <int V>
__device__ __forceinline__ void wave_topk_absorb(u64 (&acc)[V], u64 batch) {
  batch = wave_sort64(batch);
  const u64 rev = __shfl(batch, 63 - lane_id(), 64);
  acc[V - 1] = umin64(acc[V - 1], rev);
  wave_bitonic_merge<V>(acc);
}

// The same for a batch that is already ascending in lane order (another wave's accumulator row): no sort of the batch --
// 6 + log2(V) exchange stages instead of 21 + 6 + log2(V).
template <int V>
__device__ __forceinline__ void wave_topk_absorb_sorted(u64 (&acc)[V], u64 batch_ascending) {
  const u64 rev = __shfl(batch_ascending, 63 - lane_id(), 64);
  acc[V - 1] = umin64(acc[V - 1], rev);
  wave_bitonic_merge<V>(acc);
}

// Key at global rank r (0-based) of the accumulator, broadcast to all lanes.
template <int V>
__device__ __forceinline__ u64 wave_topk_at(const u64 (&acc)[V], int r) {
  u64 out = KEY_INF;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const u64 x = __shfl(acc[v], r & 63, 64);
    if ((r >> 6) == v) out = x;
  }
  return out;
}

// Full ascending bitonic sort of 64*V keys (index v*64 + lane).
template <int V>
__device__ __forceinline__ void wave_sort_full(u64 (&a)[V]) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 2; k <= 64 * V; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
        const int jv = j >> 6;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if ((v & jv) == 0) {
            const bool up = ((v & (k >> 6)) == 0);
            const u64 lo = umin64(a[v], a[v | jv]);
            const u64 hi = umax64(a[v], a[v | jv]);
            a[v] = up ? lo : hi;
            a[v | jv] = up ? hi : lo;
          }
        }
      } else {
        const bool lower = ((lane & j) == 0);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const u64 other = lane_xor64(a[v], j);
          const bool up = (((v * 64 + lane) & k) == 0);
          a[v] = (lower == up) ? umin64(a[v], other) : umax64(a[v], other);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// The reference's result list (TopK, index_utils.h:17-20) held one entry per lane: lane i owns
// slot i of a list of n <= 64 entries, ascending.  wave_list_insert is updateTopK
// (index_utils.c:19-33): the new entry goes in front of the first entry that is not strictly
// smaller -- the strictly smaller entries are a prefix, so that slot is their count -- everything
// behind moves one slot down and the last entry falls off.  The caller applies the guard
// "dist < maxDist" (freddy.c:128-131) with maxDist = wave_list_max().  Lanes >= n carry junk.
// One insertion is a ballot, a popcount and two DPP wave shifts instead of a serial walk over LDS.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_list_max(float d_slot, int n) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d_slot), n - 1));
}
__device__ __forceinline__ void wave_list_insert(float& d_slot, int32_t& id_slot, int n, float dist, int32_t id) {
  const int lane = lane_id();
  const int slot = __popcll(__ballot(lane < n && d_slot < dist));
  // wave_shr:1 -- lane i receives lane i-1 (lane 0 keeps its own value)
  const int up_d = __builtin_amdgcn_update_dpp((int)__float_as_uint(d_slot), (int)__float_as_uint(d_slot), 0x138, 0xf, 0xf, false);
  const int up_id = __builtin_amdgcn_update_dpp(id_slot, id_slot, 0x138, 0xf, 0xf, false);
  if (lane > slot) {
    d_slot = __uint_as_float((uint32_t)up_d);
    id_slot = up_id;
  } else if (lane == slot) {
    d_slot = dist;
    id_slot = id;
  }
}
// Replays the guarded insertion over `count` candidates held one per lane (64-bit value: low word =
// distance bits, high word = payload), in lane order; KEY_INF ends the sequence.  `payload(hi)` maps
// the high word to the id stored in the list.
template <typename F>
__device__ __forceinline__ void wave_list_replay(float& d_slot, int32_t& id_slot, int n, u64 cand, int count, F payload) {
  float maxd = wave_list_max(d_slot, n);
  const int lo = (int)(uint32_t)cand, hi = (int)(uint32_t)(cand >> 32);
  for (int e = 0; e < count; ++e) {
    const uint32_t clo = (uint32_t)__builtin_amdgcn_readlane(lo, e);
    const uint32_t chi = (uint32_t)__builtin_amdgcn_readlane(hi, e);
    if (clo == 0xffffffffu && chi == 0xffffffffu) break;
    const float dist = __uint_as_float(clo);
    if (dist < maxd) {
      wave_list_insert(d_slot, id_slot, n, dist, payload(chi));
      maxd = wave_list_max(d_slot, n);
    }
  }
}

// Streaming selection of the L smallest keys seen by one wave.  Keys below the running
// threshold are compacted into a 64-entry LDS staging row (ballot + prefix popcount) and
// absorbed by a bitonic merge only when the row fills up, which becomes rare once the
// threshold has tightened (expected number of passing keys ~ L * ln(n / L)).
template <int V>
struct WaveSelect {
  u64 acc[V];
  u64 tau;      // keys >= tau can no longer enter the result
  u64* stage;   // this wave's 64-entry LDS row
  int pending;  // wave-uniform fill level of the row
  int L;

  __device__ __forceinline__ void init(u64* stage_row, u64 limit_key, int L_) {
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = KEY_INF;
    tau = limit_key;
    stage = stage_row;
    pending = 0;
    L = L_;
  }
  __device__ __forceinline__ void flush() {
    const u64 batch = (lane_id() < pending) ? stage[lane_id()] : KEY_INF;
    wave_topk_absorb<V>(acc, batch);
    const u64 t = wave_topk_at<V>(acc, L - 1);
    tau = (t < tau) ? t : tau;
    pending = 0;
    __builtin_amdgcn_wave_barrier();
  }
  // every lane of the wave must call this (valid=false for idle lanes)
  __device__ __forceinline__ void push(u64 key, bool valid) {
    const bool pass = valid && key < tau;
    const u64 mask = __ballot(pass);
    const int n = __popcll(mask);
    if (n) {
      if (pending + n > 64) flush();
      const int lane = lane_id();
      // after a flush tau may have dropped; a key that no longer passes is still harmless
      if (pass) stage[pending + lanes_below(mask)] = key;
      pending += n;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __device__ __forceinline__ void finish() {
    if (pending) flush();
  }
};

}  // namespace freddy
