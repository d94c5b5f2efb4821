// fused8.h -- the filter + refine scan for K <= 256 (one byte per code: the reference's shipped default index shape,
// index_creation/config/ivfadc_config.json) with the WHOLE work entry's slab resident in LDS.
//
// ivf_filter5_kernel (fused5.h) walks an entry in six phases of two positions because a K = 1024 slab of twelve positions would be
// 384 KB: every phase is a barrier, the builders' stores share the LDS with the gatherers' reads, and the phase lasts as long as the
// slower of the two roles.  With K <= 256 twelve positions x two halves x 256 codes x 16 B are 96 KB -- the whole entry fits:
//
//   * the builders interleave the table words of ALL twelve positions for the entry's 16 items once, and they do it while the
//     gatherers are in the PREVIOUS entry's tail (column minima, thresholds, survivor pass: VALU work that needs no slab);
//   * the gatherers read a row's twelve positions back to back -- no barrier inside the main loop, nobody stores meanwhile
//     (tools/lab/ubench7.hip: 18.2 k instead of 20.0 k cycles per entry with random codes, 7.9 k instead of 10.4 k conflict-free);
//   * the thresholds (the L-th smallest column minimum per item) are taken by the GATHERER waves, so the builders are free
//     for the whole tail; four barriers per entry instead of eight;
//   * the table kernel writes a COMPACT copy of the table beside the general one (qc8: 512 B per (query, position), dword s =
//     code s | code s + 128 << 16): a builder wave's load is 256 consecutive bytes, an entry's table words are 96 KB -- the
//     general layout's 2 KB rows, of which K = 256 uses a quarter, cost the builders as much as K = 1024 (measured: no gain).
//
// Everything else is fused5.h's: entry queue and records, row terms, the (row, item) sums as unsigned 16-bit fields with the
// biased table, s' = fma(scale, V, rterm) in the tail, thresholds tau' + E with the query's running bound, survivor regions,
// merge_refine_kernel.  The sums are the same integers, so the survivors and the lists are the same (tests: every K <= 256 index
// of the GPU suite takes this kernel by default; option codes_u8 = 2 selects fused5.h's one-byte instantiation, 0 the int16 layout).
#pragma once
#include "fused5.h"

namespace freddy {

static constexpr uint32_t SCAN8_HALFB = 256u * 16u;         // one plane: the values of 8 items for every code of a position
static constexpr uint32_t SCAN8_POSB = 2u * SCAN8_HALFB;    // one position: two planes (items 0-7, 8-15)
static constexpr uint32_t scan8_slab_bytes(int m) { return (uint32_t)m * SCAN8_POSB; }

// S1 for ONE item (every wave of the workgroup takes one: builders items 0-7, gatherers 8-15): tau' = the L-th smallest of the
// item's 64 column minima, lowered by the query's running bound, widened by E -> thr_s[i]; the column is re-armed.  (The same split
// in ivf_filter5_kernel, where the builders take two items each and the gatherers wait: measured, no gain -- 82.3 against 82.0 us.)
__device__ __forceinline__ void scan8_threshold(const FilterArgs& a, const int32_t* rec, uint32_t* colmin, uint32_t* thr_s, int i, int lane, uint32_t run) {
  uint32_t c = colmin[i * 64 + lane];
  c = wave_sort32(c);   // (order-preserving keys of the float column minima)
  uint32_t t = __shfl(c, a.L - 1, 64);
  if (lane == 0) {
    if (a.tau_run) t = running_bound5(a.tau_run, (uint32_t)rec[24 + i], t, __int_as_float(rec[144 + i]), __int_as_float(rec[160 + i]), run);
    thr_s[i] = a.keep_all ? 0x7f800000u : widen_threshold5(t, __int_as_float(rec[56 + i]));
  }
  colmin[i * 64 + lane] = 0xffffffffu;
}

template <int M, bool CAND, bool PROF = false>   // PROF (lab builds): cycle sums of gatherer wave 0 per stage
__global__ __launch_bounds__(SPEC2_T) void ivf_filter8_kernel(FilterArgs a) {
  constexpr int G = SCAN5_G, RMAX = FUSED_RMAX, NG = SPEC2_NG;
  constexpr int NP = M / 2;             // position pairs
  constexpr uint32_t HROWB = 16, HALFB = SCAN8_HALFB, POSB = SCAN8_POSB;
  static_assert(M == 12 && G == 16 && SPEC2_NB == 8 && SPEC2_NG == 8, "layout");
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* slab = smem;                                                      // [M][2 halves][256][16 B]
  uint32_t* colmin = reinterpret_cast<uint32_t*>(smem + a.desc_offset);           // [16][64]
  uint32_t* thr_s = colmin + 16 * 64;                                             // [16]
  int32_t* dsc = reinterpret_cast<int32_t*>(smem + a.desc_offset + 4096 + 64);    // [2][REC_DW] entry records
  int32_t* gidq = dsc + 2 * REC_DW;
  float* rt_s = reinterpret_cast<float*>(gidq + 4);                                // [4096] row terms of the current entry (from B2 on: of the next)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool builder = wave < SPEC2_NB;
  const long long t_kernel = PROF ? clock64() : 0;
  const int K = a.K;
  const int n_work = a.n_groups[0];

  int cur = 0, ei = 0;
  // The first entry of a workgroup is its own number (the entries are ordered largest first: the grid takes the first gridDim.x of
  // them, one each); the work counter deals out the entries BEYOND those.  A workgroup so starts with ONE round trip -- its record
  // and the claim of its second entry together -- instead of two dependent atomics and then the record.
  if (tid == 0) { gidq[0] = (int)blockIdx.x; gidq[1] = (int)gridDim.x + atomicAdd(a.work_counter, 1); }
  if (tid < REC_DW) dsc[tid] = a.records[(size_t)blockIdx.x * REC_DW + tid];   // (before n_work is known: the grid never exceeds the records' capacity)
  for (int i = tid; i < 16 * 64; i += SPEC2_T) colmin[i] = 0xffffffffu;
  __syncthreads();
  if ((int)blockIdx.x >= n_work) return;

  if (builder) {
    // =====================================================================================
    // BUILDERS: a pair of waves per (position of the pair, half = 8 items), lane li of the pair <-> codes li and li + 128; six
    // passes (position pairs) build the whole entry's slab.  Two register sets of one pass each.
    // =====================================================================================
    const int hpos = (wave >> 1) & 1;
    const int half = wave >> 2;
    const int li = (wave & 1) * 64 + lane;
    const uint32_t qoff = (uint32_t)hpos * POSB + (uint32_t)half * HALFB + (uint32_t)li * HROWB;
    const uint32_t vq = (uint32_t)li * 4u + (uint32_t)hpos * 512u;
    typedef const char __attribute__((address_space(1))) * gptrc;
    typedef const uint32_t __attribute__((address_space(1))) * gptr1u;
    // ALL six passes' table words of an entry are requested at once (48 registers) -- for the next entry that is while the
    // gatherers read this one's slab, so that between the barriers only the LDS stores remain.  The words come from the COMPACT
    // copy of the table (query_codebook5_body, qc8): dword li of row (query, position) = code li | code li + 128 << 16 -- a
    // wave's load is 256 consecutive bytes
    uint32_t qw[NP][8];
#pragma unroll
    for (int s = 0; s < NP; ++s)
#pragma unroll
      for (int g = 0; g < 8; ++g) qw[s][g] = 0u;
    int qid[8];
    auto issue_all = [&](int nh) {   // position 2 pass + hpos of the half's 8 items, pass = 0 .. 5
      if (half >= nh) return;
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) {
        uint32_t voff = vq + (uint32_t)pass * 1024u;
        asm volatile("" : "+v"(voff));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const gptrc qb = (gptrc)(uintptr_t)a.qc8 + (size_t)(uint32_t)qid[u] * (size_t)(M * 512);
          qw[pass][u] = *(gptr1u)(qb + voff);
        }
      }
    };
    auto emit_all = [&](int nh) {
      if (half >= nh) return;
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) {
        unsigned char* dp = slab + (uint32_t)pass * (2u * POSB) + qoff;
        const uint32_t (&w)[8] = qw[pass];
        // v_perm_b32: bytes 0-3 come from the second operand, 4-7 from the first
        const u4 lo = u4{__builtin_amdgcn_perm(w[1], w[0], 0x05040100u), __builtin_amdgcn_perm(w[3], w[2], 0x05040100u),
                         __builtin_amdgcn_perm(w[5], w[4], 0x05040100u), __builtin_amdgcn_perm(w[7], w[6], 0x05040100u)};
        const u4 hi = u4{__builtin_amdgcn_perm(w[1], w[0], 0x07060302u), __builtin_amdgcn_perm(w[3], w[2], 0x07060302u),
                         __builtin_amdgcn_perm(w[5], w[4], 0x07060302u), __builtin_amdgcn_perm(w[7], w[6], 0x07060302u)};
        if (li < K) *reinterpret_cast<u4*>(dp) = lo;
        if (li + 128 < K) *reinterpret_cast<u4*>(dp + 128u * HROWB) = hi;
      }
    };
    // (rc: the record in LDS, or in global memory -- the next entry's, which the other builder waves cannot see in LDS before B1)
    auto entry_queries = [&](const int32_t* rc) -> int {
#pragma unroll
      for (int u = 0; u < 8; ++u) qid[u] = __builtin_amdgcn_readfirstlane(rc[24 + half * 8 + u]);
      return (__builtin_amdgcn_readfirstlane(rc[1]) + 7) >> 3;
    };
    float rtv[RMAX];
    auto fetch_row_terms = [&](const int32_t* rc) {
      const int b0 = __builtin_amdgcn_readfirstlane(rc[3]), nbk = __builtin_amdgcn_readfirstlane(rc[4]);
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        const int bl = r * NG + wave;
        rtv[r] = a.rterm[(size_t)(uint32_t)(b0 + (bl < nbk - 1 ? bl : nbk - 1)) * 64u + (uint32_t)lane];
      }
    };
    auto stash_row_terms = [&]() {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) rt_s[(r * NG + wave) * 64 + lane] = rtv[r];
    };
    {
      fetch_row_terms(dsc);
      const int nh0 = entry_queries(dsc);
      issue_all(nh0);
      emit_all(nh0);
      stash_row_terms();
    }
    lds_barrier();   // B0: slab and row terms of the first entry
    for (;;) {
      const int nb = cur ^ 1;
      const int ngid = __builtin_amdgcn_readfirstlane(gidq[(ei + 1) & 1]);
      const bool have_next = ngid < n_work;
      // while the gatherers read this entry's slab: the next entry's record into LDS, the entry after next claimed, and the next
      // entry's table words and row terms into registers
      int next_nh = 0;
      {
        int32_t rr0 = 0;
        int gid2 = 0;
        if (tid < REC_DW && have_next) rr0 = a.records[(size_t)ngid * REC_DW + tid];
        if (tid == 0) gid2 = (int)gridDim.x + atomicAdd(a.work_counter, 1);
        if (have_next) {
          const int32_t* grec = a.records + (size_t)ngid * REC_DW;
          next_nh = entry_queries(grec);
          issue_all(next_nh);
          fetch_row_terms(grec);
        }
        if (tid < REC_DW) {
          if (tid == 0) gidq[ei & 1] = gid2;
          dsc[nb * REC_DW + tid] = (tid == 6) ? (have_next ? 1 : -1) : rr0;
        }
      }
      lds_barrier();   // B1: this entry's slab is consumed; the next record is visible
      if (have_next) emit_all(next_nh);     // (LDS stores only: the words arrived during the gather)
      // the running bound of this wave's item (S1 below): requested before the barrier
      uint32_t run_b = 0u;
      const int cnt_b = __builtin_amdgcn_readfirstlane(dsc[cur * REC_DW + 1]);
      if (a.tau_run && wave < cnt_b) run_b = a.tau_run[(uint32_t)dsc[cur * REC_DW + 24 + wave]];
      lds_barrier();   // B2: the gatherers have this entry's row terms in registers, every column minimum is in
      if (wave < cnt_b) scan8_threshold(a, dsc + cur * REC_DW, colmin, thr_s, wave, lane, run_b);   // S1, item `wave`
      if (have_next) stash_row_terms();
      lds_barrier();   // B3
      lds_barrier();   // B4: the next entry's slab and row terms are complete
      if (!have_next) break;
      cur = nb;
      ++ei;
    }
  } else {
    // =====================================================================================
    // GATHERERS: lane <-> 8 rows x 16 items; a row's twelve positions back to back
    // =====================================================================================
    const int gw = wave - SPEC2_NB;
    uint32_t acc[G / 2][RMAX];
    uint32_t cwn[2][3];     // the code dwords of the coming entry's first two rows of this wave: requested an entry ahead
    auto prefetch_codes = [&](const int32_t* rc) {
      const int b0 = __builtin_amdgcn_readfirstlane(rc[3]), nbk = __builtin_amdgcn_readfirstlane(rc[4]);
      const uint32_t l4 = lane_byte4();
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int bl = r * NG + gw;
        const char* rowp = reinterpret_cast<const char*>(a.packed8) + (size_t)((uint32_t)(b0 + (bl < nbk - 1 ? bl : nbk - 1)) * (uint32_t)(M / 4)) * 256u;
#pragma unroll
        for (int t = 0; t < 3; ++t) cwn[r][t] = *reinterpret_cast<const uint32_t*>(rowp + t * 256 + l4);
      }
    };
    prefetch_codes(dsc);
    long long gt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gc = 0;
    auto gtick = [&](int slot) { if constexpr (PROF) { const long long t = clock64(); if (slot >= 0) gt[slot] += t - gc; gc = t; } };
    lds_barrier();   // B0
    if constexpr (PROF) { gt[7] = clock64() - t_kernel; }   // (the prologue: kernel start .. first slab ready)
    for (;;) {
      gtick(-1);
      const int32_t* rec = dsc + cur * REC_DW;
      const int cnt = __builtin_amdgcn_readfirstlane(rec[1]);
      const int chunk = __builtin_amdgcn_readfirstlane(rec[2]);
      const int blk0 = __builtin_amdgcn_readfirstlane(rec[3]);
      const int nblk = __builtin_amdgcn_readfirstlane(rec[4]);
      const int nrows = __builtin_amdgcn_readfirstlane(rec[5]);
      const int nq = (cnt + 7) >> 3;
      const int nb = cur ^ 1;
      const int rl_wave = (nblk - gw + NG - 1) / NG < 0 ? 0 : (nblk - gw + NG - 1) / NG;
      auto row_block = [&](int r) {
        const int bl = r * NG + gw;
        return (uint32_t)(blk0 + (bl < nblk - 1 ? bl : nblk - 1));
      };
      auto main_loop = [&](auto nqc, auto rlc) {
        constexpr int NQ = decltype(nqc)::value, RL = decltype(rlc)::value;
        constexpr int DEPTH = NQ == 1 ? 4 : 2;    // steps of one position pair in flight: 8 reads, 32 registers
        constexpr int NS = RL * NP;               // steps of the entry: (row, position pair)
        u4 va[DEPTH][2][NQ];
        uint32_t cw[3][3];                        // the code dwords of three rows in turn (a row's are requested two rows ahead)
        auto load_codes = [&](int r) {
          const uint32_t l4 = lane_byte4();
          const char* rowp = reinterpret_cast<const char*>(a.packed8) + (size_t)(row_block(r) * (uint32_t)(M / 4)) * 256u;
#pragma unroll
          for (int t = 0; t < 3; ++t) cw[r % 3][t] = *reinterpret_cast<const uint32_t*>(rowp + t * 256 + l4);
        };
        auto issue_step = [&](int i) {
          const int r = i / NP, pp = i % NP;
          const uint32_t w = cw[r % 3][pp >> 1];
          const uint32_t a0 = ((pp & 1) ? ((w >> 12) & 0xff0u) : ((w << 4) & 0xff0u)) + (uint32_t)(2 * pp) * POSB;
          const uint32_t a1 = ((pp & 1) ? ((w >> 20) & 0xff0u) : ((w >> 4) & 0xff0u)) + (uint32_t)(2 * pp + 1) * POSB;
#pragma unroll
          for (int q = 0; q < NQ; ++q) va[i % DEPTH][0][q] = *reinterpret_cast<const u4*>(slab + a0 + (uint32_t)q * HALFB);
#pragma unroll
          for (int q = 0; q < NQ; ++q) va[i % DEPTH][1][q] = *reinterpret_cast<const u4*>(slab + a1 + (uint32_t)q * HALFB);
        };
#pragma unroll
        for (int h = 0; h < G / 2; ++h)
#pragma unroll
          for (int r = 0; r < RMAX; ++r) acc[h][r] = 0u;
#pragma unroll
        for (int t = 0; t < 3; ++t) { cw[0][t] = cwn[0][t]; cw[1][t] = cwn[1][t]; }
#pragma unroll
        for (int i = 0; i < DEPTH - 1; ++i) if (i < NS) issue_step(i);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int r = i / NP;
          if (i % NP == 0 && r + 2 < RL) load_codes(r + 2);
          if (i + DEPTH - 1 < NS) issue_step(i + DEPTH - 1);
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const u4 x = va[i % DEPTH][0][q], y = va[i % DEPTH][1][q];
            acc[q * 4 + 0][r] = acc[q * 4 + 0][r] + x.x + y.x;   // (v_add3_u32: unsigned fields, fused5.h filt5_bias)
            acc[q * 4 + 1][r] = acc[q * 4 + 1][r] + x.y + y.y;
            acc[q * 4 + 2][r] = acc[q * 4 + 2][r] + x.z + y.z;
            acc[q * 4 + 3][r] = acc[q * 4 + 3][r] + x.w + y.w;
          }
          if (i + 1 < NS) __builtin_amdgcn_sched_barrier(0);
        }
      };
      {
        int rl = rl_wave;
        rl = rl < 1 ? 1 : rl;
        const int rc = (rl + 1) >> 1;
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;
        using I6 = std::integral_constant<int, 6>; using I8 = std::integral_constant<int, 8>;
        switch ((nq < 1 ? 1 : nq) * 4 + rc) {
          case 1 * 4 + 1: main_loop(I1{}, I2{}); break;
          case 1 * 4 + 2: main_loop(I1{}, I4{}); break;
          case 1 * 4 + 3: main_loop(I1{}, I6{}); break;
          case 1 * 4 + 4: main_loop(I1{}, I8{}); break;
          case 2 * 4 + 1: main_loop(I2{}, I2{}); break;
          case 2 * 4 + 2: main_loop(I2{}, I4{}); break;
          case 2 * 4 + 3: main_loop(I2{}, I6{}); break;
          default: main_loop(I2{}, I8{}); break;
        }
      }
      // the running bound of this wave's item (S1 below: gatherer wave gw takes item gw + 8): on its way during the column minima
      uint32_t run1 = 0u;
      if (a.tau_run) run1 = gw + NG < cnt ? a.tau_run[(uint32_t)rec[24 + gw + NG]] : 0u;
      gtick(0);
      lds_barrier();   // B1: the slab is free for the next entry
      gtick(1);
      // ---- tail (fused5.h): s' = fma(scale[item], V, rterm[row]) compared as floats; OFF is added for the survivors only
      float base[RMAX];
      const int last_blk = nrows > 0 ? (nrows - 1) >> 6 : -1;
      const int rs2 = (last_blk >= 0 && (last_blk % NG) == gw && (nrows & 63)) ? last_blk / NG : -1;
      const bool live_lane = lane < (nrows & 63);
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        base[r] = rt_s[(r * NG + gw) * 64 + lane];
        if (r >= rl_wave || (r == rs2 && !live_lane)) base[r] = __uint_as_float(0x7f800000u);
      }
#pragma unroll
      for (int h = 0; h < G / 2; ++h)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[h][r] ^= 0x80008000u;   // biased unsigned fields -> signed sums
      auto sval = [&](int g, int r, float sc) -> float {
        const uint32_t w = acc[g >> 1][r];
        const int v = (g & 1) ? ((int32_t)w >> 16) : ((int32_t)(w << 16) >> 16);
        return __builtin_fmaf(sc, (float)v, base[r]);
      };
      const int gi = lane & 15;
      const float p_sc = __int_as_float(rec[128 + gi]);
      uint32_t live8 = 0u;
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < rl_wave && !(r == rs2 && !live_lane)) live8 |= 1u << r;
      float best[G];
      uint32_t sec16[G / 2];
      uint32_t apack[2] = {0u, 0u};
#pragma unroll
      for (int g = 0; g < G; ++g) best[g] = __uint_as_float(0x7f800000u);
#pragma unroll
      for (int i = 0; i < G / 2; ++i) sec16[i] = 0x7f807f80u;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g < cnt) {
          const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_sc), g));
          float b1 = __uint_as_float(0x7f800000u), b2 = __uint_as_float(0x7f800000u);
          uint32_t ar = 0u;
#pragma unroll
          for (int r = 0; r < RMAX; ++r) {
            const float sv = sval(g, r, sc);
            b2 = __builtin_amdgcn_fmed3f(b1, b2, sv);
            ar = sv < b1 ? (uint32_t)r : ar;
            b1 = fminf(b1, sv);
          }
          best[g] = b1;
          {
            const uint32_t bb = __float_as_uint(b2);
            const uint32_t dn = ((bb >> 31) ? bb + 0xffffu : bb) >> 16;
            sec16[g >> 1] = (g & 1) ? ((sec16[g >> 1] & 0x0000ffffu) | (dn << 16)) : ((sec16[g >> 1] & 0xffff0000u) | dn);
          }
          // (opaque: the compiler otherwise folds the shift into the eight selects above, whose constants 128, 192, ... are no inline
          // operands -- a v_mov per row and item)
          asm volatile("" : "+v"(ar));
          apack[g >> 3] |= ar << (3 * (g & 7));
          if (rl_wave > 0) atomicMin(colmin + g * 64 + lane, float_key(b1));
        }
      }
      gtick(2);
      lds_barrier();   // B2: every wave's column minima are in
      gtick(3);
      // S1: the threshold tau' + E of item gw + 8 (the builder waves take items 0-7 at the same time)
      if (gw + NG < cnt) scan8_threshold(a, rec, colmin, thr_s, gw + NG, lane, run1);
      gtick(4);
      lds_barrier();   // B3: thresholds
      gtick(5);
      // S2: survivors -> this wave's region of each item's buffer (fused5.h)
      {
        const float p_thr = __uint_as_float(thr_s[gi]);
        const int p_it = rec[8 + gi];
        const float p_shift = __int_as_float(rec[72 + gi]);
        const float p_off = __int_as_float(rec[40 + gi]);
        const uint32_t p_lo = (uint32_t)rec[88 + gi], p_hi = (uint32_t)rec[104 + gi];
        const int p_q = rec[24 + gi];
        // lane g: item g's survivor region of this wave, and (collected below) its count -- ONE store of the counts per entry
        const int p_reg = (p_it * a.upi + chunk) * NG + gw;
        int cntv = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g < cnt) {
            const float thr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_thr), g));
            const float off = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_off), g));
            const float shift = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_shift), g));
            const uint32_t region = (uint32_t)__builtin_amdgcn_readlane(p_reg, g);
            u64* dst = a.surv + (size_t)region * (size_t)(RMAX * 64);
            int run = 0;
            if constexpr (!CAND) {
              const float second = __uint_as_float((g & 1) ? (sec16[g >> 1] & 0xffff0000u) : (sec16[g >> 1] << 16));
              const u64 multi = __ballot(!(second > thr));
              if (__builtin_expect(multi == 0ull, 1)) {
                // (no uniform branch around the emission: nearly every (item, wave) has a survivor, the exec mask does the rest)
                const bool pass = !(best[g] > thr);
                const u64 mask = __ballot(pass);
                if (pass) {
                  const uint32_t r = (apack[g >> 3] >> (3 * (g & 7))) & 7u;
                  const float dlo = fmaxf(0.0f, (best[g] + off) - shift);
                  const uint32_t loc = ((uint32_t)(blk0 + gw) + r * (uint32_t)NG) * 64u + (uint32_t)lane;
                  dst[lanes_below(mask)] = ((u64)__float_as_uint(dlo) << 32) | (u64)loc;
                }
                run = __popcll(mask);
              } else {
                const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_sc), g));   // (this path only)
              uint32_t m8 = 0u;
#pragma unroll
                for (int r = RMAX - 1; r >= 0; --r) m8 = m8 + m8 + (!(sval(g, r, sc) > thr) ? 1u : 0u);
                m8 &= live8;
                if (__ballot(m8 != 0u) != 0ull) {
#pragma unroll
                  for (int r = 0; r < RMAX; ++r) {
                    const bool pass = (m8 >> r) & 1u;
                    const u64 mask = __ballot(pass);
                    if (mask != 0ull) {
                      if (pass) {
                        const float dlo = fmaxf(0.0f, (sval(g, r, sc) + off) - shift);
                        const uint32_t loc = (uint32_t)(blk0 + r * NG + gw) * 64u + (uint32_t)lane;
                        dst[run + lanes_below(mask)] = ((u64)__float_as_uint(dlo) << 32) | (u64)loc;
                      }
                      run += __popcll(mask);
                    }
                  }
                }
              }
            } else {
              const uint32_t lo_b = (uint32_t)__builtin_amdgcn_readlane((int)p_lo, g);
              const uint32_t hi_b = (uint32_t)__builtin_amdgcn_readlane((int)p_hi, g);
              const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_sc), g));
              int accepted = 0;
#pragma unroll
              for (int r = 0; r < RMAX; ++r) {
                if (r >= rl_wave) break;
                const float sv = sval(g, r, sc);
                const uint32_t sb = __float_as_uint(sv + off);
                const bool live = (live8 >> r) & 1u;
                accepted += __popcll(__ballot(live && sb < lo_b));
                const bool amb = sb >= lo_b && sb < hi_b;
                const bool pass = live && (!(sv > thr) || amb);
                const u64 mask = __ballot(pass);
                if (mask != 0ull) {
                  if (pass) {
                    const float dlo = fmaxf(0.0f, __uint_as_float(sb) - shift);
                    const uint32_t loc = ((uint32_t)(blk0 + r * NG + gw) * 64u + (uint32_t)lane) | (amb ? 0x80000000u : 0u);
                    dst[run + lanes_below(mask)] = ((u64)__float_as_uint(dlo) << 32) | (u64)loc;
                  }
                  run += __popcll(mask);
                }
              }
              if (lane == 0 && accepted) atomicAdd(a.cand_count + __builtin_amdgcn_readlane(p_q, g), accepted);
            }
            // (v_writelane: the compiler's own select read its sixteen lane masks back from spilled scalar registers, five instructions per item)
            asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(run), "i"(g));
          }
        }
        if (lane < cnt) a.surv_count[(uint32_t)p_reg] = cntv;
      }
      const int next_ok = __builtin_amdgcn_readfirstlane(dsc[nb * REC_DW + 6]);
      if (next_ok > 0) prefetch_codes(dsc + nb * REC_DW);
      gtick(5);
      lds_barrier();   // B4
      gtick(5);
      if (next_ok < 0) break;
      cur = nb;
    }
    if (PROF && a.prof && gw == 0 && lane == 0) {
      for (int i = 0; i < 6; ++i) a.prof[(size_t)blockIdx.x * 8 + i] = gt[i];
      a.prof[(size_t)blockIdx.x * 8 + 6] = gt[6] + (clock64() - gc) * 0;                                   // S2 (the wait at B4 is in the life, not here)
      a.prof[(size_t)blockIdx.x * 8 + 6] = gt[7];                                                          // slot 6: the prologue
      a.prof[(size_t)blockIdx.x * 8 + 7] = clock64() - t_kernel;                                           // the workgroup's whole life
    }
  }
}

}  // namespace freddy
