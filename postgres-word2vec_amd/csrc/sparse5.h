// sparse5.h -- the integer filter scan for cells that only ONE or TWO queries of a batch probe.
//
// The cell-grouped scan (fused5.h) pays a whole work entry -- 128 KB of slab writes, a CU for ~30 k cycles -- whatever the
// number of items in it: right for the benchmark's ten items per cell, wrong for a corpus with more cells than (query,
// probe) items (40 M rows, 13 000 cells, 1 024 x 10 items: 1.5 items per probed cell) and for small batches.  Here one
// workgroup of four waves takes ONE (item, 4096-row chunk) unit:
//   * the query's table (ivf_filter5_kernel's: [12][1024] biased int16, query_codebook5_kernel) goes to LDS once, 24 KB
//     in plain [position][code] order -- six workgroups per CU;
//   * a lane holds 16 rows: per row six code dwords, twelve ds_read_u16 lookups, the sum (unsigned, see filt5_bias),
//     s' = fma(scale, V, rterm[row]) -- the SAME value, bit for bit, as the cell-grouped scan forms, kept in registers;
//   * threshold and survivors exactly as there: tau' = the L-th smallest of the 64 column minima, rows with
//     s' <= tau' + E (and the sentinel rule's ambiguous rows) -> the item's survivor regions, (bits(d_lo) << 32) | location.
//     Wave w's row slots r = 0 .. 15 are the blocks 4 r + w of the chunk, i.e. the cell-grouped scan's gatherer waves
//     w (even r) and w + 4 (odd r): the regions, their capacity and merge_refine_kernel are unchanged.
// Which cells go here: work_table_kernel (sparse_max items or fewer); the units are pulled from a queue like the scan's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fused5.h"

namespace freddy {

struct SparseArgs {
  const uint32_t* qc;          // [Q][12][512] the integer table (codes b, b + 512 per dword)
  const float* qscale;         // [Q]
  const float* qn;             // [Q][12]
  const float* pmax;           // [12]
  const float* rterm;          // [blocks * 64]
  const uint32_t* packed;      // [blocks][6][64]
  const int32_t* blk_off;      // [C + 1]
  const int32_t* list_off;     // [C + 1]
  const int32_t* sorted_item;  // cell-major item lists (probe plan)
  const int32_t* item_query;   // [items]
  const float* item_dist;      // [items]
  const int32_t* sp_cell;      // [units] cell, index into sorted_item, chunk of every unit (work_table_kernel)
  const int32_t* sp_first;
  const int32_t* sp_chunk;
  const int32_t* n_units;      // [1]
  int32_t* work_counter;       // [1] zeroed before the launch
  const uint32_t* packed8;     // U8 instantiation: [blocks][3][64], one byte per code
  u64* surv;
  int32_t* surv_count;
  int32_t* cand_count;         // [Q] or NULL
  int K, L, upi;
  float sentinel;
  int keep_all;                // option filter_keep_all (tests): every row survives (as ivf_filter5_kernel)
  uint32_t* tau_run;           // [Q] the queries' running bounds (FilterArgs::tau_run), or NULL
};

template <int M, bool CAND, bool U8 = false>   // U8: one byte per code (K <= 256), packed8[block][3][64]
__global__ __launch_bounds__(256, 6) void sparse_item5_kernel(SparseArgs a) {
  static_assert(M == 12, "table layout");
  constexpr int RS = 16;       // row slots per lane: 4 waves x 16 x 64 = a chunk of 4096 rows
  constexpr int NGV = 8;       // survivor regions per (item, chunk): the cell-grouped scan's gatherer waves
  constexpr int RB = 4;        // rows whose loads are in flight together
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) uint16_t lut[M][1024];
  __shared__ uint32_t colmin[64];
  __shared__ uint32_t thr_sh;
  __shared__ int unit_sh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_units = a.n_units[0];
  for (;;) {
    if (tid == 0) unit_sh = atomicAdd(a.work_counter, 1);
    if (tid < 64) colmin[tid] = 0xffffffffu;
    __syncthreads();
    const int unit = unit_sh;
    if (unit >= n_units) return;
    const int cell = a.sp_cell[unit], chunk = a.sp_chunk[unit] & 0xff;   // (bits 8..: items of the unit, always 1 here)
    const int it = a.sorted_item[a.sp_first[unit]];
    const int q = a.item_query[it];
    const float sc0 = a.qscale[q];
    const float sc = sc0 < 1e30f ? sc0 : 0.0f;
    const float A = a.item_dist[it];
    const ItemBounds ib = item_bounds(A, filter_width5<M>(a.qn + (size_t)q * M, a.pmax, sc0), a.sentinel);
    const uint32_t run_inv = a.tau_run ? a.tau_run[q] : 0u;   // the query's running bound as it is now (used after the scan of the rows)
    const int b0 = a.blk_off[cell] + chunk * FUSED_UNIT_BLOCKS;
    int nb = a.blk_off[cell + 1] - b0;
    nb = nb > FUSED_UNIT_BLOCKS ? FUSED_UNIT_BLOCKS : nb;
    int rows = a.list_off[cell + 1] - a.list_off[cell] - chunk * (FUSED_UNIT_BLOCKS * 64);
    rows = rows > FUSED_UNIT_BLOCKS * 64 ? FUSED_UNIT_BLOCKS * 64 : rows;
    // the query's table -> LDS, [position][code]
    {
      const u4* src = reinterpret_cast<const u4*>(a.qc + (size_t)q * (M * 512));
#pragma unroll
      for (int i = 0; i < (M * 128) / 256; ++i) {
        const int u = tid + 256 * i;
        const int p = u >> 7, li = u & 127;
        const u4 v = src[u];
        uint16_t* row = lut[p];
        row[li] = (uint16_t)v.x; row[li + 512] = (uint16_t)(v.x >> 16);
        row[li + 128] = (uint16_t)v.y; row[li + 640] = (uint16_t)(v.y >> 16);
        row[li + 256] = (uint16_t)v.z; row[li + 768] = (uint16_t)(v.z >> 16);
        row[li + 384] = (uint16_t)v.w; row[li + 896] = (uint16_t)(v.w >> 16);
      }
    }
    __syncthreads();
    // s' of this lane's 16 rows (+inf: no such row)
    float sv[RS];
    const int last_blk = rows > 0 ? (rows - 1) >> 6 : -1;
    const int tail_rows = rows & 63;
    const unsigned char* lut_b = reinterpret_cast<const unsigned char*>(&lut[0][0]);
#pragma unroll
    for (int r0 = 0; r0 < RS; r0 += RB) {
      constexpr int NCW = U8 ? M / 4 : M / 2;
      uint32_t cw[RB][NCW];
      float rt[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int bl = (r0 + u) * 4 + wave;
        const uint32_t blk = (uint32_t)(b0 + (bl < nb - 1 ? bl : (nb > 0 ? nb - 1 : 0)));
#pragma unroll
        for (int pr = 0; pr < NCW; ++pr)
          cw[u][pr] = U8 ? a.packed8[((size_t)blk * NCW + pr) * 64u + (uint32_t)lane] : a.packed[((size_t)blk * NCW + pr) * 64u + (uint32_t)lane];
        rt[u] = a.rterm[(size_t)blk * 64u + (uint32_t)lane];
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int bl = (r0 + u) * 4 + wave;
        uint32_t sum = 0u;
#pragma unroll
        for (int pr = 0; pr < M / 2; ++pr) {
          const uint32_t w = U8 ? cw[u][pr >> 1] : cw[u][pr];
          const uint32_t a0 = U8 ? ((pr & 1) ? ((w >> 15) & 0x1feu) : ((w << 1) & 0x1feu)) : ((w << 1) & 0x7feu);
          const uint32_t a1 = U8 ? ((pr & 1) ? ((w >> 23) & 0x1feu) : ((w >> 7) & 0x1feu)) : ((w >> 15) & 0x7feu);
          const uint32_t v0 = *reinterpret_cast<const uint16_t*>(lut_b + a0 + (uint32_t)(2 * pr) * 2048u);
          const uint32_t v1 = *reinterpret_cast<const uint16_t*>(lut_b + a1 + (uint32_t)(2 * pr + 1) * 2048u);
          sum = sum + v0 + v1;
        }
        const int v = (int)sum - 32768;   // (the biases add up to 2^15: filt5_bias)
        const bool live = bl < nb && !(bl == last_blk && tail_rows != 0 && lane >= tail_rows);
        sv[r0 + u] = live ? __builtin_fmaf(sc, (float)v, rt[u]) : __uint_as_float(0x7f800000u);
      }
    }
    // column minima over the 4 waves x 16 rows of a lane index -> tau' = the L-th smallest -> threshold
    {
      float mn = sv[0];
#pragma unroll
      for (int r = 1; r < RS; ++r) mn = fminf(mn, sv[r]);
      if (wave < nb) atomicMin(&colmin[lane], float_key(mn));
    }
    __syncthreads();
    if (wave == 0) {
      const uint32_t c0 = wave_sort32(colmin[lane]);
      uint32_t t0 = __shfl(c0, a.L - 1, 64);
      if (lane == 0) {
        if (a.tau_run && ib.e < 1e30f && A >= 0.0f && A < 1e30f) t0 = running_bound5(a.tau_run, (uint32_t)q, t0, A * (1.0f + 2e-5f), A * (1.0f - 2e-5f), run_inv);
        thr_sh = a.keep_all ? 0x7f800000u : widen_threshold5(t0, ib.e);
      }
    }
    __syncthreads();
    const float thr = __uint_as_float(thr_sh);
    // survivors: even row slots -> region `wave`, odd ones -> region `wave + 4`
    int run[2] = {0, 0};
    int accepted = 0;
    u64* dst[2];
    size_t region[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      region[h] = ((size_t)it * a.upi + chunk) * NGV + (size_t)(wave + 4 * h);
      dst[h] = a.surv + region[h] * (size_t)(FUSED_RMAX * 64);
    }
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int bl = r * 4 + wave;
      if (bl >= nb) break;   // (uniform)
      const bool live = !(bl == last_blk && tail_rows != 0 && lane >= tail_rows);
      bool pass;
      uint32_t sb = 0u;
      bool amb = false;
      if constexpr (CAND) {   // rows below the sentinel are counted (freddy.c:971): bounds on the bits of s = s' + OFF > 0
        sb = __float_as_uint(sv[r] + ib.off);
        accepted += __popcll(__ballot(live && sb < ib.lo_bits));
        amb = sb >= ib.lo_bits && sb < ib.hi_bits;
        pass = live && (!(sv[r] > thr) || amb);
      } else {
        pass = live && !(sv[r] > thr);   // (a NaN passes: exact stage)
      }
      const u64 mask = __ballot(pass);
      if (mask != 0ull) {
        const int h = r & 1;
        if (pass) {
          const float dlo = CAND ? fmaxf(0.0f, __uint_as_float(sb) - ib.shift) : fmaxf(0.0f, (sv[r] + ib.off) - ib.shift);
          const uint32_t loc = ((uint32_t)(b0 + bl) * 64u + (uint32_t)lane) | ((CAND && amb) ? 0x80000000u : 0u);
          dst[h][run[h] + lanes_below(mask)] = ((u64)__float_as_uint(dlo) << 32) | (u64)loc;
        }
        run[h] += __popcll(mask);
      }
    }
    if (lane == 0) {
      a.surv_count[region[0]] = run[0];
      a.surv_count[region[1]] = run[1];
      if (CAND && accepted) atomicAdd(a.cand_count + q, accepted);
    }
    __syncthreads();   // (colmin / thr_sh / unit_sh are rewritten by the next unit)
  }
}

// ---------------------------------------------------------------------------------------
// The same scan with units of ONE OR TWO items: a (cell, chunk) that exactly two queries of the batch probe is one unit
// (work_table_kernel, sp_pairs: the count sits in bits 8.. of sp_chunk), its rows are read once and looked up in both
// queries' tables (48 KB of LDS, three workgroups per CU).  On the 40 M-row corpus a fifth of the item-wise scan's bytes are
// cells with exactly two items, read twice by sparse_item5_kernel.
// ---------------------------------------------------------------------------------------
template <int M, bool CAND, bool U8, int N>   // N = items of the unit
__device__ __forceinline__ void sparse_rows5(const SparseArgs& a, const unsigned char* lut_b, int b0, int nb, int last_blk, int tail_rows,
                                            int wave, int lane, const float (&sc)[2], float (&sv)[2][16]) {
  constexpr int RS = 16, RB = 8;   // (eight rows' loads in flight: half the waves per CU of the one-item kernel)
  constexpr int NCW = U8 ? M / 4 : M / 2;
#pragma unroll
  for (int r0 = 0; r0 < RS; r0 += RB) {
    uint32_t cw[RB][NCW];
    float rt[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int bl = (r0 + u) * 4 + wave;
      const uint32_t blk = (uint32_t)(b0 + (bl < nb - 1 ? bl : (nb > 0 ? nb - 1 : 0)));
#pragma unroll
      for (int pr = 0; pr < NCW; ++pr)
        cw[u][pr] = U8 ? a.packed8[((size_t)blk * NCW + pr) * 64u + (uint32_t)lane] : a.packed[((size_t)blk * NCW + pr) * 64u + (uint32_t)lane];
      rt[u] = a.rterm[(size_t)blk * 64u + (uint32_t)lane];
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int bl = (r0 + u) * 4 + wave;
      uint32_t sum[2] = {0u, 0u};
#pragma unroll
      for (int pr = 0; pr < M / 2; ++pr) {
        const uint32_t w = U8 ? cw[u][pr >> 1] : cw[u][pr];
        const uint32_t a0 = U8 ? ((pr & 1) ? ((w >> 15) & 0x1feu) : ((w << 1) & 0x1feu)) : ((w << 1) & 0x7feu);
        const uint32_t a1 = U8 ? ((pr & 1) ? ((w >> 23) & 0x1feu) : ((w >> 7) & 0x1feu)) : ((w >> 15) & 0x7feu);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const uint32_t v0 = *reinterpret_cast<const uint16_t*>(lut_b + (uint32_t)i * (M * 2048u) + a0 + (uint32_t)(2 * pr) * 2048u);
          const uint32_t v1 = *reinterpret_cast<const uint16_t*>(lut_b + (uint32_t)i * (M * 2048u) + a1 + (uint32_t)(2 * pr + 1) * 2048u);
          sum[i] = sum[i] + v0 + v1;
        }
      }
      const bool live = bl < nb && !(bl == last_blk && tail_rows != 0 && lane >= tail_rows);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = (int)sum[i] - 32768;   // (the biases add up to 2^15: filt5_bias)
        sv[i][r0 + u] = (live && i < N) ? __builtin_fmaf(sc[i], (float)v, rt[u]) : __uint_as_float(0x7f800000u);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // (one group of RB rows in flight at a time: the scheduler otherwise hoists every load and spills)
  }
}

template <int M, bool CAND, bool U8 = false>
__global__ __launch_bounds__(256, 3) void sparse_pair5_kernel(SparseArgs a) {
  static_assert(M == 12, "table layout");
  constexpr int NI = 2;
  constexpr int RS = 16;       // row slots per lane: 4 waves x 16 x 64 = a chunk of 4096 rows
  constexpr int NGV = 8;       // survivor regions per (item, chunk): the cell-grouped scan's gatherer waves
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) uint16_t lut[NI][M][1024];
  __shared__ uint32_t colmin[NI][64];
  __shared__ uint32_t thr_sh[NI];
  __shared__ int unit_sh;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int n_units = a.n_units[0];
  for (;;) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));   // (opaque per unit: lane-derived addresses are not hoisted out of the unit loop)
    const int lane = tid & 63;
    if (tid == 0) unit_sh = atomicAdd(a.work_counter, 1);
    if (tid < 64) { colmin[0][tid] = 0xffffffffu; colmin[1][tid] = 0xffffffffu; }
    __syncthreads();
    const int unit = __builtin_amdgcn_readfirstlane(unit_sh);
    if (unit >= n_units) return;
    const int cell = a.sp_cell[unit];
    const int chw = a.sp_chunk[unit];
    const int chunk = chw & 0xff;
    const int cnt = (chw >> 8) >= 2 ? 2 : 1;
    int it[NI], q[NI];
    float sc[NI], Ad[NI];
    uint32_t run_inv[NI];
    ItemBounds ib[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      it[i] = a.sorted_item[a.sp_first[unit] + (i < cnt ? i : 0)];
      q[i] = a.item_query[it[i]];
      const float sc0 = a.qscale[q[i]];
      sc[i] = sc0 < 1e30f ? sc0 : 0.0f;
      Ad[i] = a.item_dist[it[i]];
      ib[i] = item_bounds(Ad[i], filter_width5<M>(a.qn + (size_t)q[i] * M, a.pmax, sc0), a.sentinel);
      run_inv[i] = a.tau_run ? a.tau_run[q[i]] : 0u;
    }
    const int b0 = a.blk_off[cell] + chunk * FUSED_UNIT_BLOCKS;
    int nb = a.blk_off[cell + 1] - b0;
    nb = nb > FUSED_UNIT_BLOCKS ? FUSED_UNIT_BLOCKS : nb;
    int rows = a.list_off[cell + 1] - a.list_off[cell] - chunk * (FUSED_UNIT_BLOCKS * 64);
    rows = rows > FUSED_UNIT_BLOCKS * 64 ? FUSED_UNIT_BLOCKS * 64 : rows;
    // the queries' tables -> LDS, [item][position][code]
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (i < cnt) {
        const u4* src = reinterpret_cast<const u4*>(a.qc + (size_t)q[i] * (M * 512));
#pragma unroll
        for (int k = 0; k < (M * 128) / 256; ++k) {
          const int u = tid + 256 * k;
          const int p = u >> 7, li = u & 127;
          const u4 v = src[u];
          uint16_t* row = lut[i][p];
          row[li] = (uint16_t)v.x; row[li + 512] = (uint16_t)(v.x >> 16);
          row[li + 128] = (uint16_t)v.y; row[li + 640] = (uint16_t)(v.y >> 16);
          row[li + 256] = (uint16_t)v.z; row[li + 768] = (uint16_t)(v.z >> 16);
          row[li + 384] = (uint16_t)v.w; row[li + 896] = (uint16_t)(v.w >> 16);
        }
      }
    }
    __syncthreads();
    // s' of this lane's 16 rows (+inf: no such row), per item
    float sv[NI][RS];
    const int last_blk = rows > 0 ? (rows - 1) >> 6 : -1;
    const int tail_rows = rows & 63;
    const unsigned char* lut_b = reinterpret_cast<const unsigned char*>(&lut[0][0][0]);
    if (cnt == 1) sparse_rows5<M, CAND, U8, 1>(a, lut_b, b0, nb, last_blk, tail_rows, wave, lane, sc, sv);
    else sparse_rows5<M, CAND, U8, 2>(a, lut_b, b0, nb, last_blk, tail_rows, wave, lane, sc, sv);
    // column minima over the 4 waves x 16 rows of a lane index -> tau' = the L-th smallest -> threshold
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (i < cnt) {
        float mn = sv[i][0];
#pragma unroll
        for (int r = 1; r < RS; ++r) mn = fminf(mn, sv[i][r]);
        if (wave < nb) atomicMin(&colmin[i][lane], float_key(mn));
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (wave == i && i < cnt) {     // (wave i: item i's threshold)
        const uint32_t c0 = wave_sort32(colmin[i][lane]);
        uint32_t t0 = __shfl(c0, a.L - 1, 64);
        if (lane == 0) {
          if (a.tau_run && ib[i].e < 1e30f && Ad[i] >= 0.0f && Ad[i] < 1e30f)
            t0 = running_bound5(a.tau_run, (uint32_t)q[i], t0, Ad[i] * (1.0f + 2e-5f), Ad[i] * (1.0f - 2e-5f), run_inv[i]);
          thr_sh[i] = a.keep_all ? 0x7f800000u : widen_threshold5(t0, ib[i].e);
        }
      }
    }
    __syncthreads();
    // survivors: even row slots -> region `wave`, odd ones -> region `wave + 4`
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (i < cnt) {
        const float thr = __uint_as_float(thr_sh[i]);
        int run[2] = {0, 0};
        int accepted = 0;
        u64* dst[2];
        size_t region[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          region[h] = ((size_t)it[i] * a.upi + chunk) * NGV + (size_t)(wave + 4 * h);
          dst[h] = a.surv + region[h] * (size_t)(FUSED_RMAX * 64);
        }
#pragma unroll
        for (int r = 0; r < RS; ++r) {
          const int bl = r * 4 + wave;
          if (bl >= nb) break;   // (uniform)
          const bool live = !(bl == last_blk && tail_rows != 0 && lane >= tail_rows);
          bool pass;
          uint32_t sb = 0u;
          bool amb = false;
          if constexpr (CAND) {   // rows below the sentinel are counted (freddy.c:971): bounds on the bits of s = s' + OFF > 0
            sb = __float_as_uint(sv[i][r] + ib[i].off);
            accepted += __popcll(__ballot(live && sb < ib[i].lo_bits));
            amb = sb >= ib[i].lo_bits && sb < ib[i].hi_bits;
            pass = live && (!(sv[i][r] > thr) || amb);
          } else {
            pass = live && !(sv[i][r] > thr);   // (a NaN passes: exact stage)
          }
          const u64 mask = __ballot(pass);
          if (mask != 0ull) {
            const int h = r & 1;
            if (pass) {
              const float dlo = CAND ? fmaxf(0.0f, __uint_as_float(sb) - ib[i].shift) : fmaxf(0.0f, (sv[i][r] + ib[i].off) - ib[i].shift);
              const uint32_t loc = ((uint32_t)(b0 + bl) * 64u + (uint32_t)lane) | ((CAND && amb) ? 0x80000000u : 0u);
              dst[h][run[h] + lanes_below(mask)] = ((u64)__float_as_uint(dlo) << 32) | (u64)loc;
            }
            run[h] += __popcll(mask);
          }
        }
        if (lane == 0) {
          a.surv_count[region[0]] = run[0];
          a.surv_count[region[1]] = run[1];
          if (CAND && accepted) atomicAdd(a.cand_count + q[i], accepted);
        }
      }
    }
    __syncthreads();   // (colmin / thr_sh / unit_sh are rewritten by the next unit)
  }
}

}  // namespace freddy
