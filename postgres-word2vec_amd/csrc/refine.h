// refine.h -- IVFADC as FILTER + REFINE: what the filter scans (fused5.h cell-grouped, sparse5.h item-wise) and the merge share,
// and the merge itself.
//
// The exact path (fused3.h) spends 9.4 G separately rounded lane-operations per 1024-query batch on
// the residual LUTs although only ~2k rows per query ever reach the result list.  The
// scans run on a CHEAP distance with a PROVEN error bound, and the reference's arithmetic is
// replayed only for the rows that can still matter:
//
//   |r - c|^2 = |r|^2 + (|c|^2 + 2 co.c) - 2 q.c          r = q - co  (co = coarse centroid, c = codeword)
//                       `----- summed over the row's codewords: rterm[row], pinned once (fp64 -> fp32)
//                                        `---- qc[query][p][code] = -2 q_p.c   int16 fixed point, one table per batch (fused5.h)
//
// Selection keeps every row whose cheap value is within E of the item's L-th column minimum: that
// set contains every row whose exact distance is <= the L-th smallest exact distance of the item,
// ties included.  Survivors carry (d_lo, row location); merge_refine_kernel (four waves per query)
// finds T = (L-th smallest d_lo) + E, recomputes the reference's distance -- sequential binary32
// squareDistance per position, positions added in order (index_utils.c:500-508, :1126-1133) -- for
// the rows with d_lo <= T only (typically L + 1 of ~130 survivors), and runs the same 2k-smallest
// selection and updateTopK replay as merge_surv_kernel on those exact keys.  Rows whose bound
// straddles the sentinel guard (freddy.c:971 counts them) are flagged and decided exactly as well.
// Non-finite inputs make E non-finite, which sends every row to the exact stage (slow, still exact).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_common.h"
#include "fused3.h"

namespace freddy {

struct FilterArgs {
  const uint32_t* qc;          // [Q][M][512] -2 q_p . c as int16 pairs (codes b, b+512)   (query_codebook5_body)
  const float* rterm;          // [blocks*64] sum_p (|c|^2 + 2 co_p . c) of every row (pinned)
  const int32_t* records;      // [entries][REC_DW] (entry_record5_kernel)
  const int32_t* n_groups;     // [1] number of work entries
  int32_t* work_counter;       // [1] zeroed before the launch
  const uint32_t* packed;      // [blocks][M2][64]
  u64* surv;                   // [items][upi][8 waves][512]: (bits(d_lo) << 32) | flag << 31 | row location
  int32_t* surv_count;
  int32_t* cand_count;         // [Q] or NULL: rows certainly below the sentinel (the flagged ones are added by the merge)
  int K, L, upi;
  float sentinel;
  uint32_t desc_offset;
  uint32_t fence;              // always 0: conditions the compiler cannot see through keep the code of ivf_filter5_kernel's phases in basic blocks of their own -- without them the register allocator spills in the main loop (86.5 instead of 82 us; sched_barriers do not have the same effect)
  int keep_all;                // option filter_keep_all (tests): every row survives the filter -- with refine_all, every bracket is checked
  long long* prof;
  const uint32_t* packed8;     // U8 instantiation: [blocks][3][64], one byte per code (K <= 256)
  // The query's RUNNING bound (ivf_filter5_kernel, S1): tau_run[q] = ~key of the smallest (tau' + A_up) any finished (item, chunk)
  // of query q has reported this round -- an upper bound of the query's L-th smallest cheap distance over everything it probes;
  // 0 = none yet.  A later entry cuts at min(tau', bound - A_lo) + E instead of tau' + E.  Purely opportunistic: a workgroup
  // reads whatever is there (no waiting, any stale value is a valid bound), so the survivors differ from run to run and the
  // lists never do.  NULL: off.
  uint32_t* tau_run;
  const uint32_t* qc8;         // fused8.h (K <= 256): [Q][M][128] the compact copy of the table: dword s = code s | code s + 128 << 16
};

// The integer-slab scan (fused5.h) quantises the table with one scale per query; its margin (derivation there):
static constexpr int FILT5_VMAX = 2730;   // 12 positions x 2730 = 32760 < 2^15
static constexpr float FILT5_EPS = 512.0f * 5.9604644775390625e-8f * 1.0001f;
template <int M>
__device__ __forceinline__ float filter_width5(const float* __restrict__ qn, const float* __restrict__ pmax, float scale) {
  float sb = 0.0f;
#pragma unroll
  for (int p = 0; p < M; ++p) {
    const float t = qn[p] + pmax[p];
    sb = __builtin_fmaf(t, t, sb);
  }
  return __builtin_fmaf(sb, FILT5_EPS, __builtin_fmaf(28.0f, scale, 1e-30f));
}

struct ItemBounds {
  float off;          // initial value of the running sums
  float e;            // selection margin E (+inf: keep every row)
  float shift;        // d_lo = max(0, s - shift)
  uint32_t lo_bits;   // s <  lo : certainly below the sentinel
  uint32_t hi_bits;   // s >= hi : certainly not below it;  in between: decided exactly
};
// A = the reference's coarse distance (sequential binary32 over d <= 300 dimensions: relative error
// < 2e-5 against the exact |r|^2 of the rounded residual).
__device__ __forceinline__ ItemBounds item_bounds(float A, float E, float sentinel) {
  ItemBounds b;
  if (E < 1e30f && A < 1e30f && A >= 0.0f) {
    const float a_up = A * (1.0f + 2e-5f), a_lo = A * (1.0f - 2e-5f);
    b.off = a_up + E;
    b.e = E;
    b.shift = ((b.off - a_lo) + 0.25f * E) * (1.0f + 1e-6f);
    const float hi = (sentinel + b.shift) * (1.0f + 1e-6f);
    const float lo = ((sentinel + b.shift) - E) * (1.0f - 1e-6f);
    b.hi_bits = hi < 3e38f ? __float_as_uint(hi) : 0xfffffffeu;
    b.lo_bits = lo > 0.0f ? __float_as_uint(lo < 3e38f ? lo : 3e38f) : 0u;
  } else {
    b.off = 0.0f;
    b.e = __uint_as_float(0x7f800000u);
    b.shift = __uint_as_float(0x7f800000u);
    b.lo_bits = 0u;
    b.hi_bits = 0xfffffffeu;
  }
  return b;
}
// selection threshold on the stored bits: everything <= tau + E (rounded up)
__device__ __forceinline__ uint32_t widen_threshold(uint32_t tau_bits, float E) {
  if (tau_bits >= 0x7f800000u || !(E < 1e30f)) return 0xfffffffeu;
  const float t = (__uint_as_float(tau_bits) + E) * (1.0f + 2.4e-7f);
  return __float_as_uint(t);
}

// rterm[row] = sum_p (|c|^2 + 2 co_p . c) over the row's 12 codewords c and its cell's centroid co, in fp64,
// rounded once (pin time).  The part of the cheap distance that depends on (cell, row) only: it is the
// initial value of the row's running sum, so the scan streams nothing per cell.
static __global__ __launch_bounds__(256) void row_term_kernel(const uint32_t* __restrict__ packed, const int32_t* __restrict__ blk_cell,
                                                      const float* __restrict__ coarse, const float* __restrict__ cbR,
                                                      float* __restrict__ rterm, int64_t n_slots, int M2, int d, int m, int K, int S) {
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= n_slots) return;
  const int64_t block = slot >> 6;
  const int lane = (int)(slot & 63);
  const int cell = blk_cell[block];
  double acc = 0.0;
  for (int p = 0; p < m; ++p) {
    const uint32_t word = packed[((size_t)block * M2 + (p >> 1)) * 64 + lane];
    const int code = (int)((word >> ((p & 1) * 16)) & 0xffffu);
    const float* cv = cbR + ((size_t)p * K + (code < K ? code : 0)) * S;
    const float* co = coarse + (size_t)cell * d + (size_t)p * S;
    for (int j = 0; j < S; ++j) acc += (double)cv[j] * (double)cv[j] + 2.0 * (double)co[j] * (double)cv[j];
  }
  rterm[slot] = (float)acc;
}

// ---------------------------------------------------------------------------------------
// Work-entry records: everything the scan needs to know about an entry in one 512-byte row, so that the
// persistent workgroups fetch the next entry with ONE load instead of a chain of dependent ones
// (cell -> items -> queries -> bounds), which the short phases of this kernel can no longer hide.
//   [0] cell  [1] items  [2] chunk  [3] first row block  [4] row blocks  [5] rows
//   [8+g] item  [24+g] query (slots past the last item repeat item 0: always loadable)
//   [40+g] OFF  [56+g] E  [72+g] SHIFT  [88+g] lo bits  [104+g] hi bits        (item_bounds)
// ---------------------------------------------------------------------------------------
static constexpr int REC_DW = 176;   // + [128 + g] the table scale of item g's query, [144 + g] / [160 + g] A_up / A_lo of the item (entry_record5_kernel)

struct RecordArgs {
  const int32_t* group_cell;
  const int32_t* group_first;
  const int32_t* group_cnt;
  const int32_t* n_groups;
  const int32_t* sorted_item;
  const int32_t* item_query;
  const int32_t* blk_off;
  const int32_t* list_off;
  const float* item_dist;   // [items] exact coarse distance of every item (probe plan)
  const float* qn;
  const float* qscale;
  const float* pmax;
  int32_t* records;
  float sentinel;
};

// ---------------------------------------------------------------------------------------
// merge + exact refine + replay (one wave per query); see the header comment.
// ---------------------------------------------------------------------------------------
struct MergeRefineArgs {
  const u64* surv;             // [n_active*W][upi][8][512]
  const int32_t* surv_count;
  const int32_t* active;
  const int32_t* round_rows;
  const int32_t* item_cell;    // [n_active*W]
  const float* queries;        // [Q][d]
  const float* coarse;         // [C][d]
  const float* cbR;            // [m][K][S]
  const float* qn;             // [Q][M]
  const float* pmax;           // [M]
  const float* qscale5;        // [Q] the table scale of every query (fused5.h): part of the margin E
  const uint32_t* packed;
  const int32_t* pos;
  const int32_t* blk_cell;     // [blocks] list (cell) of every row block
  int32_t* cand_count;
  int32_t* violations;         // [2] rows of the exact stage whose distance left the bracket [d_lo, d_lo + E] / rows checked
  int32_t* out_ids;
  float* out_dist;
  int32_t* found;
  int32_t* next_active;
  int32_t* n_next;
  int32_t* status;
  int n_active, W, upi, L, k, found_rule, first_round, K, d;
  float sentinel;
  int refine_all;    // option refine_all (tests): every survivor goes through the exact stage and is counted in violations[1]
  uint32_t fence;    // always 0 (as FilterArgs::fence: never-true conditions that keep the stages' code in blocks of their own: 25.2 instead of 26.6 us)
  // PARTIAL instantiation (a batch over the flat PQ table): workgroup x = (query x / slices, slice x % slices) merges the
  // survivors of ITS W items (the query's items are slices * W wide) and leaves its L smallest exact keys in part[x][L];
  // merge_replay_kernel selects among the slices' keys and replays.  (Each slice's 2k smallest exact keys contain the
  // query's 2k smallest that lie in the slice: selection-then-replay as before, on 4 x as many workgroups.)
  int slices;
  u64* part;
};

// MANY = true (with NWV = 12): the instantiation for queries with hundreds of survivor regions (a batch over the flat PQ
// table: 245 pseudo-lists x 8 waves) -- the selection of the lower bounds split over the four waves, dense neighbourhoods
// collected by all of them.  It needs 145 registers (three workgroups per CU); the IVFADC instantiation stays at 128.
template <int S, int M, int NWV, bool MANY = false, bool PARTIAL = false>
__global__ __launch_bounds__(64 * NWV, 4) void merge_refine_kernel(MergeRefineArgs a) {
  static_assert(!MANY || NWV > 1, "the split selection needs the four waves");
  static_assert(!PARTIAL || MANY, "slices of a query: the flat PQ table's instantiation");
  // NWV = 4: four waves per query.  Wave 0 selects and replays; the exact stage of the normal case (<= NC rows)
  // is spread over all four -- one tile of 64 (row, position) chains each -- because a wave spends it
  // waiting for two dependent round trips per tile: the shortest latency for ONE batch.
  // NWV = 1: one wave per query does everything, tile after tile -- a quarter of the wave slots and 12 instead of
  // 30 KB of LDS per query: with several batches in flight, when this kernel has to fit into the CUs the scans of the
  // other batches leave, the smaller footprint is worth more than the latency.
  constexpr int NT = NWV > 4 ? NWV : 4;     // tiles of 64 chains refined together (one per wave in the multi-wave rounds)
  static_assert(NT * 64 / M <= 64, "a round's rows are finalised by one wave");
  constexpr int NC = NT * 64 / M;           // = 21 candidates
  constexpr int SQ = S + 1;                 // row pitch of the squared differences
  constexpr int M2 = M / 2;
  __shared__ u64 stage_all[MANY ? NWV : 1][64];
  u64* const stage = stage_all[0];
  __shared__ u64 part_key[MANY ? NWV : 1][MANY ? 64 : 1];        // pass 1 split over the waves (many survivor regions: the flat PQ table)
  __shared__ uint32_t part_flag[MANY ? NWV : 1][MANY ? 64 : 1];
  __shared__ float qs[M * S];
  __shared__ float sq[NWV == 1 ? 32 * SQ : NT * 64 * SQ];
  __shared__ float lutv[NT * 64];
  __shared__ int32_t cbo[NT * 64], coo[NT * 64];
  __shared__ u64 cq_key[64 + NC];
  __shared__ int32_t cq_cell[64 + NC];
  __shared__ int sh_n;
  constexpr int BQ = MANY ? 512 : 1;     // rows with d_lo <= T beyond the kept keys, collected by all waves (dense neighbourhoods)
  __shared__ u64 bq_key[BQ];
  constexpr int PB = 256;                // survivor regions per block of the dense sweep
  __shared__ int pref_all[MANY ? NWV : 1][PB + 1];
  __shared__ int bq_n;
  __shared__ uint32_t sh_T;
  const int x = blockIdx.x, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int q = PARTIAL ? x / a.slices : (a.active ? a.active[x] : x);
  const int k = a.k;
  const int per_item = a.upi * FUSED_NW;
  const int R = a.W * per_item;
  constexpr int NBATCH = 4;

  if (!(a.fence & 2)) for (int j = threadIdx.x; j < M * S; j += 64 * NWV) qs[j] = a.queries[(size_t)q * a.d + j];
  const float E = (a.fence & 2) ? 0.0f : filter_width5<M>(a.qn + (size_t)q * M, a.pmax, a.qscale5[q]);

  // one tile of the exact stage: chains [t*64, t*64+64) of the first n queue entries -> lutv
  auto tile_work = [&](int t, float* sqb, int n) {
    const int chains = n * M;
    const int ch = t * 64 + lane;
    if (ch < chains) {   // round trip 1: the chain's code
      const int c = ch / M, p = ch - c * M;
      const uint32_t loc = (uint32_t)cq_key[c] & 0x7fffffffu;
      const uint32_t word = a.packed[((size_t)(loc >> 6) * M2 + (uint32_t)(p >> 1)) * 64u + (loc & 63u)];
      const int code = (int)((word >> ((p & 1) * 16)) & 0xffffu);
      cbo[ch] = (p * a.K + code) * S;
      coo[ch] = cq_cell[c] * a.d + p * S;
    }
    __builtin_amdgcn_wave_barrier();
    // round trips 2 and 3: every element (chain, dimension) -- consecutive lanes read consecutive floats
    // of a codeword / centroid; batches of H steps each, the loads of a batch go out together (all S
    // at once would need more than the 128 registers four resident workgroups per CU leave a wave).
    // CH chains are staged at a time: all 64 of the tile, or -- one wave per query (NWV = 1), where the kernel's LDS decides how
    // many queries are resident beside the other batches' scans -- 32 and 32 (3.3 instead of 6.7 KB of squared differences:
    // 9.2 KB per query, sixteen per CU instead of eleven).
    constexpr int CH = (NWV == 1) ? 32 : 64;
    constexpr int NSTEP = (CH * S + 63) / 64;
    constexpr int H = (NWV == 1) ? (NSTEP + 1) / 2 : (S + 1) / 2;   // (one wave per query: 7 + 6 loads in flight -- with 13 the kernel spilled at 128 registers)
#pragma unroll 1
    for (int hh = 0; hh < 64 / CH; ++hh) {
    if constexpr (NWV == 1) {
      // One wave per query: ONE round trip per staged half instead of two.  A chain's codeword / centroid slice (S = 25 floats, 4-byte
      // aligned) is read as UN = 7 units of 16 bytes -- the last one overlaps its neighbour (floats S - 4 .. S - 1), so no unit leaves
      // the slice --, a lane <-> a unit: 4 + 4 loads of 16 bytes in flight per lane instead of 7 + 6 dwords twice.  With batches in
      // flight the memory system is loaded and a round trip of this latency chain is 3 - 4 us: three tiles x two halves save six.
      constexpr int UN = (S + 3) / 4, W0 = UN * 4 - S;   // units per chain; the last unit's first NEW float
      constexpr int NU = (CH * UN + 63) / 64;
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
      f4u cv4[NU], cov4[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int unit = u * 64 + lane;
        const int cl = unit / UN, part = unit - cl * UN;
        const int chn = t * 64 + hh * CH + cl;
        const bool live = unit < CH * UN && chn < chains;
        const uint32_t j0 = part == UN - 1 ? (uint32_t)(S - 4) : (uint32_t)(part * 4);
        const uint32_t off = ((uint32_t)cbo[live ? chn : t * 64] + (live ? j0 : 0u)) * 4u;
        const uint32_t offc = ((uint32_t)coo[live ? chn : t * 64] + (live ? j0 : 0u)) * 4u;   // (C*d*4 < 2^32)
        cv4[u] = *reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(a.cbR) + off);
        cov4[u] = *reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(a.coarse) + offc);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int unit = u * 64 + lane;
        const int cl = unit / UN, part = unit - cl * UN;
        const int chn = t * 64 + hh * CH + cl;
        if (unit < CH * UN && chn < chains) {
          const int p = chn % M;
          const int j0 = part == UN - 1 ? S - 4 : part * 4;
          const int w0 = part == UN - 1 ? W0 : 0;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            if (w >= w0) {
              const float r = qs[p * S + j0 + w] - cov4[u][w];               // freddy.c:296-303
              const float tt = r - cv4[u][w];
              sqb[cl * SQ + j0 + w] = tt * tt;                               // index_utils.c:500-508
            }
          }
        }
      }
    } else
#pragma unroll 1
    for (int h0 = 0; h0 < NSTEP; h0 += H) {
      float cv[H], cov[H];
#pragma unroll
      for (int u = 0; u < H; ++u) {
        const int e = (h0 + u) * 64 + lane;
        const int cl = t * 64 + hh * CH + e / S, j = e % S;
        const bool live = (h0 + u < NSTEP) && e < CH * S && cl < chains;
        const uint32_t off = ((uint32_t)cbo[live ? cl : t * 64] + (uint32_t)j) * 4u;
        const uint32_t offc = ((uint32_t)coo[live ? cl : t * 64] + (uint32_t)j) * 4u;   // (C*d*4 < 2^32)
        cv[u] = live ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.cbR) + off) : 0.0f;
        cov[u] = live ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.coarse) + offc) : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < H; ++u) {
        const int e = (h0 + u) * 64 + lane;
        const int cl = e / S, j = e % S;                                 // (chain within the staged CH)
        const int p = (t * 64 + hh * CH + cl) % M;
        if ((h0 + u < NSTEP) && e < CH * S && t * 64 + hh * CH + cl < chains) {
          const float r = qs[p * S + j] - cov[u];                        // freddy.c:296-303
          const float tt = r - cv[u];
          sqb[cl * SQ + j] = tt * tt;                                    // index_utils.c:500-508
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int mych = t * 64 + hh * CH + lane;
    if (lane < CH && mych < chains) {
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < S; ++j) acc = acc + sqb[lane * SQ + j];
      lutv[mych] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    }
  };

  // ---- pass 1: the L smallest lower bounds ----
  // (a few more than L are kept: the rows to refine are normally all among them)
  const int LW = (a.fence & 4) ? a.L : ((a.L + 22 < 64 && R <= 64 * NBATCH) ? a.L + 22 : 64);
  // Dense sweep.  The survivors of a query sit in R regions (item x chunk x gatherer wave) of a few keys each; walking them
  // region by region (a lane per region) is a chain of dependent round trips as long as the fullest region.  Instead: the
  // counts of a block of <= PB regions (one round trip) -> exclusive prefix in LDS -> lane s of the wave takes the s-th key
  // of the block (binary search in the prefix), 64 x NBATCH keys per round trip.  Every key goes to sink(key, valid,
  // region), called by the whole wave.
  int* const pref = pref_all[MANY ? wave : 0];
  // pre(kk, s0, total): called once per round trip of up to 64 x NBATCH keys, before they go to the sink (pass 1 tightens its
  // selection threshold from the keys' lane minima there)
  auto dense_block = [&](auto&& sink, int jb0, int nreg, auto&& pre) {
    int carry = 0;
    for (int i0 = 0; i0 < nreg; i0 += 64 * NBATCH) {
      int c[NBATCH];
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) {
        const int j = i0 + u * 64 + lane;
        c[u] = (j < nreg) ? a.surv_count[(size_t)x * R + (size_t)(jb0 + j)] : 0;
      }
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) {
        int inc = c[u];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int up = __shfl_up(inc, o, 64);
          if (lane >= o) inc += up;
        }
        const int j = i0 + u * 64 + lane;
        if (j < nreg) pref[j] = carry + inc - c[u];
        carry += __shfl(inc, 63, 64);
      }
    }
    const int total = carry;
    if (lane == 0) pref[nreg] = total;
    __builtin_amdgcn_wave_barrier();
    for (int s0 = 0; s0 < total; s0 += 64 * NBATCH) {
      u64 kk[NBATCH];
      int jj[NBATCH];
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) {
        const int sidx = s0 + u * 64 + lane;
        int lo = 0;   // the largest region index with pref <= sidx: the (non-empty) region holding key sidx
#pragma unroll
        for (int step = PB / 2; step >= 1; step >>= 1) {
          const int mid = lo + step;
          if (mid < nreg && pref[mid] <= sidx) lo = mid;
        }
        jj[u] = lo;
        kk[u] = (sidx < total) ? a.surv[((size_t)x * R + (size_t)(jb0 + lo)) * (size_t)(FUSED_RMAX * 64) + (size_t)(sidx - pref[lo])] : KEY_INF;
      }
      pre(kk, s0, total);
#pragma unroll
      for (int u = 0; u < NBATCH; ++u)
        if (s0 + u * 64 < total) sink(kk[u], s0 + u * 64 + lane < total, jb0 + jj[u]);
    }
    __builtin_amdgcn_wave_barrier();
  };
  // blocks jb0, jb0 + jstep, ... of the query's regions
  auto no_pre = [](const u64 (&)[NBATCH], int, int) {};
  auto sweep_pre = [&](auto&& sink, int jb0, int jstep, auto&& pre) {
    for (int jb = jb0; jb < R; jb += jstep) dense_block(sink, jb, R - jb < PB ? R - jb : PB, pre);
  };
  auto sweep = [&](auto&& sink, int jb0, int jstep) { sweep_pre(sink, jb0, jstep, no_pre); };
  // More regions than one sweep of a wave covers (W x chunks x 8 > 256: a batch over the flat PQ table "probes" hundreds of
  // pseudo-lists): every wave selects from a quarter of them -- the sweeps are chains of dependent round trips -- and
  // wave 0 merges the four selections.
  const bool split1 = MANY && R > PB;
  if (MANY && split1) {
    WaveSelect<1> sp;
    sp.init(stage_all[MANY ? wave : 0], KEY_INF, LW);
    uint32_t fs = 0u;
    // (every wave takes 1 / NWV of the regions: in blocks of PB = 256, a slice of 490 regions kept two of the twelve waves busy
    // for 12 us and the other ten idle)
    {
      auto sink1 = [&](u64 kk, bool v, int) { if (v) fs |= (uint32_t)kk; sp.push(kk, v); };
      const int chunk = (R + NWV - 1) / NWV;
      if (chunk <= PB) {
        const int j0 = wave * chunk;
        if (j0 < R) dense_block(sink1, j0, R - j0 < chunk ? R - j0 : chunk, no_pre);
      } else {
        sweep(sink1, wave * PB, NWV * PB);
      }
    }
    sp.finish();
    part_key[MANY ? wave : 0][lane] = sp.acc[0];
    part_flag[MANY ? wave : 0][lane] = fs;
    __syncthreads();
  }

  // every row with d_lo <= T (or a pending sentinel decision) of this wave's share of the regions -> bq_key
  auto collect = [&]() {
    const uint32_t Tb = sh_T;
    sweep([&](u64 kk, bool valid, int) {
      const bool need = valid && (((uint32_t)(kk >> 32) <= Tb) || ((uint32_t)kk & 0x80000000u));
      const u64 mask = __ballot(need);
      if (mask != 0ull) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&bq_n, (int)__popcll(mask));
        base = __shfl(base, 0, 64);
        const int slot = base + (int)lanes_below(mask);
        if (need && slot < BQ) bq_key[slot] = kk;
      }
    }, wave * PB, NWV * PB);
  };

  if (wave != 0) {
    for (;;) {                               // rounds of <= NC rows until wave 0 announces none
      __syncthreads();                       // wave 0 has queued the rows to refine
      const int n1 = sh_n;
      if (MANY && n1 < 0) { collect(); __syncthreads(); continue; }
      if (n1 <= 0) break;
      if (wave * 64 < n1 * M) tile_work(wave, sq + wave * 64 * SQ, n1);
      __syncthreads();
    }
    return;
  }
  WaveSelect<1> sel;
  sel.init(stage, KEY_INF, LW);
  uint32_t flag_seen = 0u;
  if (MANY && split1) {
    // (the waves' selections are ascending lists: merged without re-sorting them)
    sel.acc[0] = part_key[0][lane];
    flag_seen |= part_flag[0][lane];
#pragma unroll
    for (int w = 1; w < (MANY ? NWV : 1); ++w) {
      const u64 kk = part_key[w][lane];
      flag_seen |= part_flag[w][lane];
      if (__ballot(kk != KEY_INF) != 0ull) wave_topk_absorb_sorted<1>(sel.acc, kk);
    }
  } else {
    // (a lone wave is bound by instruction issue, and a streaming selection that starts without a threshold pays a 64-bit
    // sort + merge for nearly every batch of 64 keys: the LW-th smallest lane minimum of a round trip's keys -- one 32-bit
    // sort -- bounds the LW-th smallest key, nothing farther can be among the LW smallest)
    if constexpr (NWV != 1)   // (the four-wave instantiation has no register to spare: 56 B of scratch with the hook)
      sweep([&](u64 kk, bool v, int) { if (v) flag_seen |= (uint32_t)kk; sel.push(kk, v); }, 0, PB);
    else
    sweep_pre([&](u64 kk, bool v, int) { if (v) flag_seen |= (uint32_t)kk; sel.push(kk, v); }, 0, PB,
              [&](const u64 (&kk)[NBATCH], int s0, int total) {
                uint32_t mn = 0xffffffffu;
#pragma unroll
                for (int u = 0; u < NBATCH; ++u)
                  if (s0 + u * 64 + lane < total) mn = min(mn, (uint32_t)(kk[u] >> 32));
                const uint32_t dL = (uint32_t)__builtin_amdgcn_readlane((int)wave_sort32(mn), LW - 1);
                const u64 bound = ((u64)dL << 32) | 0xffffffffull;
                if (bound < sel.tau) sel.tau = bound;
              });
  }
  sel.finish();
  if (a.fence & 8) {
    if (lane < k) a.out_ids[(size_t)q * k + lane] = (int32_t)sel.acc[0];
    if (lane == 0) sh_n = 0;
    __syncthreads();
    return;
  }
  // T = (L-th smallest d_lo) + E, rounded up; every key of the query if there are fewer than L or E is not finite
  uint32_t T_bits;
  {
    const u64 kth = wave_topk_at<1>(sel.acc, a.L - 1);
    T_bits = (kth == KEY_INF || a.refine_all) ? 0xfffffffeu : widen_threshold((uint32_t)(kth >> 32), E);   // (refine_all: tests, every row)
  }
  __builtin_amdgcn_wave_barrier();
  // ---- pass 2: exact distances of the rows with d_lo <= T (and of the flagged ones) ----
  WaveSelect<1> sel2;
  sel2.init(stage, KEY_INF, a.L);
  int queued = 0;        // wave-uniform
  int amb_accepted = 0;  // lane 0..NC-1 partial counts
  auto finalize = [&](int n) {   // chain sums of the first n queue entries -> exact keys into sel2; drops them
    u64 out_key = KEY_INF;
    if (lane < n) {
      float dsum = 0.0f;
#pragma unroll
      for (int p = 0; p < M; ++p) dsum = dsum + lutv[lane * M + p];       // index_utils.c:1126-1133
      const uint32_t lo = (uint32_t)cq_key[lane];
      const int32_t pid = a.pos[lo & 0x7fffffffu];
      {   // self-check of the bound (freddy_gpu_filter_bound_violations)
        const float dlo = __uint_as_float((uint32_t)(cq_key[lane] >> 32));
        if (E < 1e20f && (dsum < dlo || dsum > dlo + E)) atomicAdd(a.violations, 1);
        if (a.refine_all) atomicAdd(a.violations + 1, 1);
      }
      if (dsum < a.sentinel) {
        out_key = ((u64)__float_as_uint(dsum) << 32) | (u64)(uint32_t)pid;
        if (lo & 0x80000000u) amb_accepted += 1;
      }
    }
    __builtin_amdgcn_wave_barrier();
    sel2.push(out_key, out_key != KEY_INF);
    // drop the refined entries from the queue
    const u64 mk = (lane + n < queued) ? cq_key[lane + n] : 0ull;
    const int32_t mc = (lane + n < queued) ? cq_cell[lane + n] : 0;
    __builtin_amdgcn_wave_barrier();
    if (lane + n < queued) { cq_key[lane] = mk; cq_cell[lane] = mc; }
    queued -= n;
    __builtin_amdgcn_wave_barrier();
  };
  auto refine = [&](int n) {     // (wave 0 alone: the rare cases)
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t * 64 < n * M; ++t) tile_work(t, sq, n);
    finalize(n);
  };
  auto offer = [&](u64 key, bool valid, int cell) {
    const bool need = valid && (((uint32_t)(key >> 32) <= T_bits) || ((uint32_t)key & 0x80000000u));
    const u64 mask = __ballot(need);
    if (mask != 0ull) {
      while (queued >= NC) refine(NC);   // (the queue holds < NC entries afterwards: room for 64 more)
      if (need) {
        const int slot = queued + lanes_below(mask);
        cq_key[slot] = key;
        cq_cell[slot] = cell;
      }
      queued += __popcll(mask);
      __builtin_amdgcn_wave_barrier();
    }
  };
  // Normal case: the rows to refine (d_lo <= T) are a proper prefix of the LW keys pass 1 kept and no
  // flagged row exists -- they go to the queue straight from the registers.  Otherwise (more such rows
  // than were kept, e.g. many duplicates of one vector, or a sentinel decision pending) every key is revisited.
  bool revisit = true;
  {
    const u64 mine = sel.acc[0];
    const bool in = lane < LW && mine != KEY_INF && (uint32_t)(mine >> 32) <= T_bits;
    const u64 in_mask = __ballot(in);
    const bool any_flag = __ballot((flag_seen & 0x80000000u) != 0u) != 0ull;
    const bool all_in = __popcll(in_mask) >= LW;   // every kept key qualifies: there may be more outside
    if (!all_in && !any_flag) {
      revisit = false;
      if (in) {
        cq_key[lanes_below(in_mask)] = mine;
        cq_cell[lanes_below(in_mask)] = a.blk_cell[((uint32_t)mine & 0x7fffffffu) >> 6];
      }
      queued = __popcll(in_mask);
      __builtin_amdgcn_wave_barrier();
    }
  }
  // More qualifying rows than were kept (a dense neighbourhood: hundreds of rows within E of the 2k-th smallest bound --
  // common for batches over the flat PQ table): all waves collect them, then the usual rounds of NC rows.  Only if
  // even that queue overflows does wave 0 walk the regions alone (below).
  auto round4 = [&](int n1) {     // the first n1 <= NC queue entries, refined by the four waves together
    if (lane == 0) sh_n = n1;
    __syncthreads();
    tile_work(0, sq, n1);
    __syncthreads();
    finalize(n1);
  };
  if (MANY && revisit) {
    if (lane == 0) { sh_T = T_bits; bq_n = 0; sh_n = -1; }
    __syncthreads();
    collect();
    __syncthreads();
    const int cnt = bq_n;
    if (cnt <= BQ) {
      revisit = false;
      for (int base = 0; base < cnt; base += NC) {
        const int n = cnt - base < NC ? cnt - base : NC;
        if (lane < n) {
          const u64 kk = bq_key[base + lane];
          cq_key[lane] = kk;
          cq_cell[lane] = a.blk_cell[((uint32_t)kk & 0x7fffffffu) >> 6];
        }
        queued = n;
        __builtin_amdgcn_wave_barrier();
        if (!(a.fence & 1)) round4(n); else queued = 0;
      }
    }
  }
  if (revisit) sweep([&](u64 kk, bool valid, int j) { offer(kk, valid, valid ? a.item_cell[(size_t)x * a.W + j / per_item] : 0); }, 0, PB);
  if (a.fence & 1) queued = 0;
  // the queued rows are refined by the four waves together, NC per round (normally one round: <= NC rows; a batch over
  // the flat PQ table has 20 .. 60 rows within E of its 2k-th smallest bound)
  if (NWV > 1) {
    while (queued > 0) round4(queued < NC ? queued : NC);
    if (lane == 0) sh_n = 0;
    __syncthreads();                         // (the other waves leave)
  }
  while (queued > 0) refine(queued < NC ? queued : NC);
  sel2.finish();
  if constexpr (PARTIAL) {   // this slice's L smallest exact keys (distance bits, id); the rest is merge_replay_kernel's
    if (lane < a.L) a.part[(size_t)x * a.L + lane] = sel2.acc[0];
    return;
  }

  u64 byp = (sel2.acc[0] == KEY_INF || lane >= a.L) ? KEY_INF : ((sel2.acc[0] << 32) | (sel2.acc[0] >> 32));
  byp = wave_sort64(byp);
  float d_slot = (a.first_round || lane >= k) ? a.sentinel : a.out_dist[(size_t)q * k + lane];
  int32_t id_slot = (a.first_round || lane >= k) ? -1 : a.out_ids[(size_t)q * k + lane];
  wave_list_replay(d_slot, id_slot, k, byp, a.L, [](uint32_t hi) { return (int32_t)hi; });
  if (lane < k) {
    a.out_ids[(size_t)q * k + lane] = id_slot;
    a.out_dist[(size_t)q * k + lane] = d_slot;
  }
  int amb_total = amb_accepted;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amb_total += __shfl_xor(amb_total, o, 64);
  if (lane == 0) {
    int f = a.first_round ? 0 : a.found[q];
    const int rows = a.round_rows[x];
    f += (a.found_rule == 1 && a.cand_count) ? a.cand_count[q] + amb_total : (rows > 0 ? rows : 0);
    a.found[q] = f;
    if (f < k && rows >= 0) {
      const int slot = atomicAdd(a.n_next, 1);
      a.next_active[slot] = q;
      if (a.status) a.status[0] = 1;
    }
  }
}

}  // namespace freddy
