// fused5.h -- the filter + refine scan of fused4.h with INTEGER slabs: one LDS access serves eight items, six
// phases instead of twelve.  (DESIGN.md 5.3c has the measurements.)
//
// What bounds ivf_filter_kernel (fused4.h) in its main loop is the LDS pipe: per position 48 KB of fp32 slab are
// written (48 wave-level ds_write_b128) and gathered (8 waves x 8 rows x <= 3 ds_read_b128, a third of the read
// cycles being bank conflicts), 87 % of the phase (rocprofv3 SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT) -- and the
// cost of a gather is per ACCESS, not per byte (tools/lab/ubench6.hip).  Here a slab entry is the table's own 16-bit
// integer:
//
//   * the query x codebook table is quantised with ONE scale per query (not per (query, position)) to
//     |v| <= 2730, so that the sum over the 12 positions fits 16 bits and is EXACT: the gatherers add rows
//     with v_pk_add_u16 and a (row, item) sum is one 16-bit half of a register -- 64 sum registers for
//     16 items x 8 rows;
//   * a slab row is 16 items x 2 B = 32 B = two 16-byte halves of 8 items: ONE ds_read_b128 per (row, position)
//     for an entry of <= 8 items, two for 9..16 (fp32: one per 4 items).  Half h of row c sits at byte
//     32 c + 16 (h ^ bit 3 of c), so that first halves do not all share 8 of the 16 bank groups; the builders only
//     interleave the items' table words (v_perm_b32) -- no conversion, no multiplication;
//   * two positions fit one buffer (2 x 32 KB): SIX phases -- barriers -- per entry instead of twelve, and the
//     rows' code dword of a phase is exactly the two codes it needs;
//   * work entries hold up to 16 items (scan_common.h work_table_kernel, cost_mode 2);
//   * the floating-point value s' = fma(scale[item], V, rterm[row]) is formed in the tail, where the selection
//     needs it -- WITHOUT the item's constant OFF: the selection compares floats (a constant shift changes neither
//     the order nor tau' + E); OFF is added for the survivors only (s = s' + OFF > 0, the bits the merge expects).
//
// The bound (u = 2^-24, B and the reference's error as in fused4.h; D exact, d the reference's binary32 value):
//   reference                                   |d - D|        <= 39 u B
//   rterm (fp64, rounded once)                                  <=  1 u B
//   per position: dot product (25 terms; since round 4 on the matrix cores, seven v_mfma_f32_16x16x4_f32 steps -- any
//                 order of binary32 products and sums of 25 terms stays below 56 u, the fmaf chain of rounds 2-3 below 26 u)
//                 and quotient                                <= 58 u 2|q_p||c| ; rint: <= 0.5 scale
//                 no clamping: 2 |q_p| max|c_p| <= 2730 scale by construction
//       summed over 12 positions                                <= 58 u B + 6 scale
//   the sum V itself: exact (integers below 2^15)
//   s' = fma(scale, V, rterm), s = s' + OFF: two roundings of values <= 3 B + E      <=  7 u B
//   residual's own rounding (as fused4.h)                                    <=  2 u B
//   =>  |(s - OFF + |r|^2) - d| <= e = 107 u B + 6 scale;  the construction needs e <= E / 4.2 = 121.9 u B + 6.67 scale:
//   E = 512 u B + 28 scale   (typical: 2.0e-3 against 1.2e-3 of fused4.h -- 13.7 instead of 11.8 survivors per item).
// Everything downstream (survivor regions, merge_refine_kernel with the same E, the self-check of the bracket on
// every refined row) is fused4.h's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "refine.h"
#include "coarse.h"

namespace freddy {

// (FILT5_VMAX, filter_width5: fused4.h, next to the merge that shares them)
static constexpr int SCAN5_G = 16;   // items per work entry

// order-preserving 32-bit key of a float (NaNs sort above +inf or below -inf: only met with non-finite inputs)
__device__ __forceinline__ uint32_t float_key(float x) {
  const uint32_t b = __float_as_uint(x);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
// selection threshold of the integer scan from the KEY of tau': tau' + E rounded up, as a float; +inf = keep every row
__device__ __forceinline__ uint32_t widen_threshold5(uint32_t tau_key, float E) {
  const uint32_t b = (tau_key & 0x80000000u) ? (tau_key ^ 0x80000000u) : ~tau_key;
  const float tau = __uint_as_float(b);
  if (!(tau < 3e38f) || !(tau > -3e38f) || !(E < 1e30f)) return 0x7f800000u;
  const float t = tau + E;
  return __float_as_uint(t + __builtin_fabsf(t) * 2.4e-7f + 1e-37f);
}

// The table stores v + bias(p) >= 0 with sum_p bias(p) = 2^15 (m = 12: 2731 for the first eight positions, 2730 for the
// others): a (row, item) sum is then an UNSIGNED 16-bit field in [8, 65528] at every step, so two fields per register are
// added with plain 32-bit adds (v_add3_u32: two positions at once, no carry ever crosses the halves), and field ^ 0x8000 is
// the signed sum V in two's complement.
__host__ __device__ __forceinline__ constexpr int filt5_bias(int p, int m) { return FILT5_VMAX + (p < 32768 - FILT5_VMAX * m ? 1 : 0); }
// 4 x lane id, never hoisted or spilled (two instructions where it is used)
__device__ __forceinline__ uint32_t lane_byte4() {
  uint32_t x;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0" : "=v"(x));
  return x;
}
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add_i16(uint32_t x, uint32_t y) {   // v_pk_add_u16 (wrap-around: never reached)
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2v, x) + __builtin_bit_cast(s2v, y));
}

// The table: values rint(-2 q_p . c / scale[q]) clamped to +-2730, scale[q] = max_p 2 |q_p| max|c_p| / 2730 -- ONE per query;
// row (query, position) = 128 uint4, word j of uint4 s = codes 128 j + s (low half) and 128 j + s + 512 (high half).  Every workgroup (position p, 16
// queries) derives the scales of its queries itself (16 lanes per query, lane <-> position: the same fmaf chain and
// the same maximum in every workgroup); the workgroups of position 0 also write qn[q][p] = |q_p| (rounded up) and
// scale[q] for the record and merge kernels.
template <int S, int QT>
__device__ __forceinline__ void query_codebook5_body(const float* __restrict__ queries, const float* __restrict__ cbT,
                                                     const float* __restrict__ cmax, float* __restrict__ qn,
                                                     float* __restrict__ qscale, uint32_t* __restrict__ qc,
                                                     int Q, int d, int m, int K, int bx, int by, unsigned char* smem,
                                                     uint32_t* __restrict__ qc8 = nullptr) {
  // qc8 (K <= 256, fused8.h): a COMPACT copy of the table beside the general one -- row (query, position) = 128 dwords, dword s =
  // code s (low half) | code s + 128 (high half): 512 B instead of the 2 KB row of which K = 256 uses a quarter
  static_assert(QT == 16, "one 16-lane group per query");
  constexpr int SP = (S + 3) & ~3;
  // LDS from the caller: qs [QT][SP] floats, inv_s [QT]
  float (*qs)[SP] = reinterpret_cast<float (*)[SP]>(smem);
  float* inv_s = reinterpret_cast<float*>(smem + QT * SP * 4);
  const int tid = threadIdx.x, p = bx, q0 = by * QT;
  // MFMA path: the B operands of the wave's FIRST group of code slots are requested before anything else -- the codebook
  // round trip runs under the prologue (the workgroup is a chain of latencies: all 768 are resident at once); the second
  // group's are requested into the same registers right after the first group's matrix instructions, under its conversion
  constexpr int STEPS_B = (S + 3) / 4;
  float bv[STEPS_B][4][2];
  auto load_b = [&](int g) {   // (cbT here = the fragment-order copy, freddy_gpu.hip build_fragment_codebook)
    typedef float f4b __attribute__((ext_vector_type(4)));
    const f4b* src = reinterpret_cast<const f4b*>(cbT + ((((size_t)p * 8 + g) * STEPS_B) * 64 + (tid & 63)) * 8);
#pragma unroll
    for (int st = 0; st < STEPS_B; ++st) {
      const f4b lo = src[(size_t)st * 128], hi = src[(size_t)st * 128 + 1];
      bv[st][0][0] = lo.x; bv[st][0][1] = lo.y; bv[st][1][0] = lo.z; bv[st][1][1] = lo.w;
      bv[st][2][0] = hi.x; bv[st][2][1] = hi.y; bv[st][3][0] = hi.z; bv[st][3][1] = hi.w;
    }
  };
  load_b((tid >> 6) * 2);
  for (int i = tid; i < QT * SP; i += 256) {
    const int qi = i / SP, j = i - qi * SP;
    qs[qi][j] = (j < S && q0 + qi < Q) ? queries[(size_t)(q0 + qi) * d + p * S + j] : 0.0f;
  }
  {
    const int qi = tid >> 4, pp = tid & 15, q = q0 + qi;
    float best = 0.0f;
    if (pp < m && q < Q) {
      float n2 = 0.0f;
      for (int j = 0; j < S; ++j) { const float v = queries[(size_t)q * d + pp * S + j]; n2 = __builtin_fmaf(v, v, n2); }
      const float nrm = __builtin_sqrtf(n2) * (1.0f + 1e-5f);
      if (p == 0) qn[(size_t)q * m + pp] = nrm;
      best = 2.0f * nrm * cmax[pp];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (pp == 0) {
      const float sc = q < Q ? best * (1.0f / (float)FILT5_VMAX) * (1.0f + 1e-6f) : 0.0f;
      inv_s[qi] = (sc > 0.0f && sc < 1e30f) ? 1.0f / sc : 0.0f;
      if (p == 0 && q < Q) qscale[q] = sc;
    }
  }
  {
    // The dot products on the matrix cores (v_mfma_f32_16x16x4_f32: A = 16 queries x 4 dimensions, B = 4 dimensions x 16
    // codes, seven steps for S = 25): a wave takes two groups of 16 code slots and, per group, the eight tiles whose codes
    // share a slot's uint4 (code 128 i + slot, i = 0..3, and its partner + 512) -- so a lane ends up with exactly the four
    // words of one 16-byte store per query.  D[row = 4 (lane >> 4) + reg][col = lane & 15].  The sum's rounding differs from
    // the fmaf chain's (order, product rounding): within the 28 u 2|q_p||c| -> 56 u the bracket's derivation (top of this
    // file) leaves room for (e = 105 u B + 6 scale <= E / 4.2 = 122 u B + 6.67 scale).
    typedef float f4v __attribute__((ext_vector_type(4)));
    __syncthreads();   // (qs, inv_s)
    const int wave = tid >> 6, lane = tid & 63, col = lane & 15, kq = lane >> 4;
    const int nq = (Q - q0 < QT) ? Q - q0 : QT;
    constexpr int STEPS = (S + 3) / 4;
    float av[STEPS];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) av[st] = 4 * st + kq < S ? qs[col][4 * st + kq] : 0.0f;   // (A: row = lane & 15 = the query)
    const float vmax = (float)FILT5_VMAX;
    const int bias = filt5_bias(p, m);
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      const int g = wave * 2 + gg;
      f4v acc[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int st = 0; st < STEPS; ++st)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) acc[i][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], bv[st][i][e], acc[i][e], 0, 0, 0);
      if (gg == 0) load_b(g + 1);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int qi = 4 * kq + reg;
        if (qi < nq) {
          const float inv = inv_s[qi];
          uint32_t wd[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int i0 = (int)fminf(fmaxf(__builtin_rintf(-2.0f * acc[i][0][reg] * inv), -vmax), vmax) + bias;
            const int i1 = (int)fminf(fmaxf(__builtin_rintf(-2.0f * acc[i][1][reg] * inv), -vmax), vmax) + bias;
            wd[i] = (uint32_t)i0 | ((uint32_t)i1 << 16);
          }
          *reinterpret_cast<uint4*>(qc + ((size_t)(q0 + qi) * m + p) * 512 + 4 * (16 * g + col)) = uint4{wd[0], wd[1], wd[2], wd[3]};
          if (qc8) qc8[((size_t)(q0 + qi) * m + p) * 128 + (16 * g + col)] = (wd[0] & 0xffffu) | (wd[1] << 16);
        }
      }
    }
  }
}
template <int S, int QT>
static constexpr int query_codebook5_lds() { return QT * ((S + 3) & ~3) * 4 + QT * 4; }
template <int S, int QT>
__global__ __launch_bounds__(256) void query_codebook5_kernel(const float* __restrict__ queries, const float* __restrict__ cbT,
                                                             const float* __restrict__ cmax, float* __restrict__ qn,
                                                             float* __restrict__ qscale, uint32_t* __restrict__ qc,
                                                             int Q, int d, int m, int K, uint32_t* __restrict__ qc8) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[query_codebook5_lds<S, QT>()];
  query_codebook5_body<S, QT>(queries, cbT, cmax, qn, qscale, qc, Q, d, m, K, blockIdx.x, blockIdx.y, smem, qc8);
}

// The MFMA cell-selection distances (coarse.h) and the query x codebook table in ONE launch: the first `n_coarse`
// workgroups are coarse tiles, the others table units (position, 16 queries).  Neither depends on the other and both are a
// few tens of microseconds of small workgroups; as two kernels of a batch's chain they cost two launches, two
// dependencies (with batches in flight the table kernel runs in line: a side stream per batch collides with the other
// batches' streams in the hardware queues).  The workgroups keep their own shape and lifetime -- a version in which every
// workgroup did both jobs was 40 % slower with batches in flight (big workgroups that live long keep the scans of the
// other batches from finding CUs).
struct CoarseTableArgs {
  const float* queries;
  const float* coarseF; const float* cn2; float* dist; float* qn2; int Q, Cpad, d, dp; ZeroArgs z; int coarse_gx, coarse_gy;
  const float* cbT; const float* cmax; float* qn; float* qscale; uint32_t* qc; int m, K;
  float* tmin; int C;   // many cells: the (query, 128-cell tile) minima for the plan's two-level selection (NULL: not wanted)
  const ch8v* coarseH; int ec;   // many cells: the centroids split into f16 hi / lo (coarse_approx16_body); NULL: the fp32 tile
  uint32_t* qc8;                 // K <= 256: the compact copy of the table (query_codebook5_body); NULL: not wanted
};
template <int S, int QT, bool H16 = false>   // H16: the coarse tiles on f16-split operands (many cells); an instantiation of its own, so
__global__ __launch_bounds__(256) void coarse_table5_kernel(CoarseTableArgs a) {   // that the <= 1024-cell kernel compiles as before
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n_coarse = a.coarse_gx * a.coarse_gy;
  const int b = blockIdx.x;
  if (b < n_coarse) {
    if constexpr (H16)
      coarse_approx16_body(a.queries, a.coarseH, a.ec, a.cn2, a.dist, a.qn2, a.Q, a.Cpad, a.d, a.z, b % a.coarse_gx, b / a.coarse_gx,
                           a.coarse_gx, a.coarse_gy, smem, a.tmin, a.C);
    else
    coarse_approx_body(a.queries, a.coarseF, a.cn2, a.dist, a.qn2, a.Q, a.Cpad, a.d, a.dp, a.z, b % a.coarse_gx, b / a.coarse_gx,
                       a.coarse_gx, a.coarse_gy, smem, a.tmin, a.C);
  } else {
    const int t = b - n_coarse;
    query_codebook5_body<S, QT>(a.queries, a.cbT, a.cmax, a.qn, a.qscale, a.qc, a.Q, a.d, a.m, a.K, t % a.m, t / a.m, smem, a.qc8);
  }
}

// The query's running bound (FilterArgs::tau_run), one (item, chunk)'s part: t = key of its own tau', [a_lo, a_up] its coarse
// distance, inv = the bound as read earlier (0: none).  Reports tau' + a_up if that improves on what was read (one atomic at
// most), returns the key the item cuts at: min(tau', bound - a_lo).  (Derivation: ivf_filter5_kernel, S1.)
__device__ __forceinline__ uint32_t running_bound5(uint32_t* __restrict__ tau_run, uint32_t q, uint32_t t, float a_up, float a_lo, uint32_t inv) {
  const uint32_t tb = (t & 0x80000000u) ? (t ^ 0x80000000u) : ~t;   // key -> bits
  const float tau = __uint_as_float(tb);
  if (tau < 3e38f && tau > -3e38f && a_up < 3e38f) {
    const uint32_t mine = ~float_key(tau + a_up);
    if (mine > inv) atomicMax(tau_run + q, mine);
    if (inv != 0u) {
      const uint32_t bk = ~inv;
      const float alt = __uint_as_float((bk & 0x80000000u) ? (bk ^ 0x80000000u) : ~bk) - a_lo;
      if (alt < tau) return float_key(alt);
    }
  }
  return t;
}

// Entry records as entry_record_kernel; [128 + g] = the table scale of item g's query.
template <int M>
__global__ __launch_bounds__(256) void entry_record5_kernel(RecordArgs a) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= a.n_groups[0]) return;
  const int cell = a.group_cell[e], first = a.group_first[e], gc = a.group_cnt[e];
  const int cnt = gc & 0xff, chunk = gc >> 8;
  int32_t* rec = a.records + (size_t)e * REC_DW;
  if (lane == 0) {
    const int b0 = a.blk_off[cell] + chunk * FUSED_UNIT_BLOCKS;
    int nb = a.blk_off[cell + 1] - b0;
    if (nb > FUSED_UNIT_BLOCKS) nb = FUSED_UNIT_BLOCKS;
    int rows = a.list_off[cell + 1] - a.list_off[cell] - chunk * (FUSED_UNIT_BLOCKS * 64);
    if (rows > FUSED_UNIT_BLOCKS * 64) rows = FUSED_UNIT_BLOCKS * 64;
    rec[0] = cell; rec[1] = cnt; rec[2] = chunk; rec[3] = b0; rec[4] = nb; rec[5] = rows;
  }
  if (lane < 16) {
    const int it = lane < cnt ? a.sorted_item[first + lane] : -1;
    const int q = a.item_query[it >= 0 ? it : a.sorted_item[first]];
    const float sc = a.qscale[q];
    ItemBounds ib = item_bounds(0.0f, 0.0f, a.sentinel);
    if (it >= 0) ib = item_bounds(a.item_dist[it], filter_width5<M>(a.qn + (size_t)q * M, a.pmax, sc), a.sentinel);
    rec[8 + lane] = it;
    rec[24 + lane] = q;
    rec[40 + lane] = (int32_t)__float_as_uint(ib.off);
    rec[56 + lane] = (int32_t)__float_as_uint(ib.e);
    rec[72 + lane] = (int32_t)__float_as_uint(ib.shift);
    rec[88 + lane] = (int32_t)ib.lo_bits;
    rec[104 + lane] = (int32_t)ib.hi_bits;
    rec[128 + lane] = (int32_t)__float_as_uint((it >= 0 && sc < 1e30f) ? sc : 0.0f);
    // the item's coarse distance as an interval (item_bounds: relative error < 2e-5): what makes the cheap distances of
    // different cells comparable (running bound of the query, ivf_filter5_kernel S1); +inf / 0: no part in it
    const float A = it >= 0 ? a.item_dist[it] : -1.0f;
    const bool fin = it >= 0 && ib.e < 1e30f && A >= 0.0f && A < 1e30f;
    rec[144 + lane] = (int32_t)__float_as_uint(fin ? A * (1.0f + 2e-5f) : __uint_as_float(0x7f800000u));
    rec[160 + lane] = (int32_t)__float_as_uint(fin ? A * (1.0f - 2e-5f) : 0.0f);
  }
}

// ---------------------------------------------------------------------------------------
// The scan.  Roles, entry queue, records, row terms, thresholds and survivor regions as ivf_filter_kernel;
// phases j = 0 .. M/2 - 1 cover positions 2j and 2j + 1.
//   LDS: slab[2 buffers][2 positions][K][12] int16 = 96 KB, then colmin / thresholds / records / row terms.
// ---------------------------------------------------------------------------------------
// PROF: the per-phase cycle counters of option fused_prof (18 registers of a builder wave, 8 of a gatherer wave: an
// instantiation of its own, the production kernel does not carry them)
// U8: the rows' codes as one byte each (K <= 256: what the reference's shipped default indexes use) -- packed8[block][3][64],
// dword t of a row = its codes 4 t .. 4 t + 3: 12 instead of 24 bytes of codes per row, three code loads per row and entry
// instead of six (the dword of phases 2 t and 2 t + 1 is the same)
template <int M, bool FULLK, bool CAND, bool PROF = false, bool U8 = false>
__global__ __launch_bounds__(SPEC2_T) void ivf_filter5_kernel(FilterArgs a) {
  static_assert(!(U8 && FULLK), "one byte per code: K <= 256");
  constexpr int G = SCAN5_G, RMAX = FUSED_RMAX, NG = SPEC2_NG;
  constexpr int NP = M / 2;             // phases per entry
  constexpr int ROWB = G * 2;           // bytes of a slab row: two 16-byte halves of 8 items
  constexpr int HROWB = 16;             // a half row; the halves of a position live in two planes of K half rows each
  static_assert(M == 12 && G == 16 && SPEC2_NB == 8, "layout");
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* slab = smem;                                                      // [2][2][K][ROWB]
  uint32_t* colmin = reinterpret_cast<uint32_t*>(smem + a.desc_offset);           // [16][64]
  uint32_t* thr_s = colmin + 16 * 64;                                             // [16]
  int32_t* dsc = reinterpret_cast<int32_t*>(smem + a.desc_offset + 4096 + 64);    // [2][REC_DW] entry records
  int32_t* gidq = dsc + 2 * REC_DW;
  float* rt_s = reinterpret_cast<float*>(gidq + 4);                                // [4096] row terms of the entry about to start

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool builder = wave < SPEC2_NB;
  const int K = FULLK ? 1024 : a.K;
  const uint32_t HALFB = (uint32_t)K * HROWB;        // bytes of one plane: the values of 8 items for every code of a position
  const uint32_t POSB = 2u * HALFB;                  // bytes of one position's slab
  const uint32_t BUFB = 2u * POSB;                   // bytes of one buffer
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
    // BUILDERS: a pair of waves per (position of the phase, half = 8 items); lane li of the pair <-> code pairs
    // li + 128 k (k < 4), one 16-byte load per item.  A lane holds the values of ALL eight items of its half for its
    // eight codes, so a half row leaves as ONE 16-byte store and consecutive lanes store consecutive half rows: no bank
    // conflicts (8-byte stores per item quad were 4-way conflicts: half of the kernel's conflict cycles).
    // Two register sets of one PHASE each: the set that phase j + 1 is written from at the start of phase j is
    // refilled at once with phase j + 3.
    // =====================================================================================
    const int hpos = (wave >> 1) & 1;    // position 2 j + hpos of phase j
    const int half = wave >> 2;          // items 8 half .. 8 half + 7
    const int li = (wave & 1) * 64 + lane;
    const uint32_t qoff = (uint32_t)hpos * POSB + (uint32_t)half * HALFB + (uint32_t)li * HROWB;
    const uint32_t vq = (uint32_t)li * 16u + (uint32_t)hpos * 2048u;
    u4 qw[2][8];   // [set][item of the half]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int g = 0; g < 8; ++g) qw[s][g] = u4{0u, 0u, 0u, 0u};
    typedef const char __attribute__((address_space(1))) * gptrc;
    typedef const u4 __attribute__((address_space(1))) * gptr4u;
    auto issue = [&](int set, int phase, const int (&qids)[8], int nh) {   // position 2 phase + hpos of the half's 8 items
      if (half >= nh || (a.fence & 32)) return;
      uint32_t voff = vq + (uint32_t)phase * 4096u;
      asm volatile("" : "+v"(voff));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const gptrc qb = (gptrc)(uintptr_t)a.qc + (size_t)(uint32_t)qids[u] * (size_t)(M * 2048);
        qw[set][u] = *(gptr4u)(qb + voff);
      }
    };
    // the eight table words of code pair b = li + 128 k interleaved: low halves -> half row b, high halves -> b + 512
    auto emit = [&](int set, unsigned char* dst, int nh) {
      if (half >= nh || (a.fence & 16)) return;
      unsigned char* dp = dst + qoff;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          w[u] = k == 0 ? qw[set][u].x : k == 1 ? qw[set][u].y : k == 2 ? qw[set][u].z : qw[set][u].w;
        // v_perm_b32: bytes 0-3 come from the second operand, 4-7 from the first
        const u4 lo = u4{__builtin_amdgcn_perm(w[1], w[0], 0x05040100u), __builtin_amdgcn_perm(w[3], w[2], 0x05040100u),
                         __builtin_amdgcn_perm(w[5], w[4], 0x05040100u), __builtin_amdgcn_perm(w[7], w[6], 0x05040100u)};
        const u4 hi = u4{__builtin_amdgcn_perm(w[1], w[0], 0x07060302u), __builtin_amdgcn_perm(w[3], w[2], 0x07060302u),
                         __builtin_amdgcn_perm(w[5], w[4], 0x07060302u), __builtin_amdgcn_perm(w[7], w[6], 0x07060302u)};
        const int b = li + 128 * k;
        if (FULLK || b < K) *reinterpret_cast<u4*>(dp + (uint32_t)(128 * k) * HROWB) = lo;
        if (FULLK || b + 512 < K) *reinterpret_cast<u4*>(dp + (uint32_t)(128 * k + 512) * HROWB) = hi;
      }
    };

    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc = 0;
    auto tick = [&](int slot) { if constexpr (PROF) { const long long t = clock64(); pt[slot] += t - pc; pc = t; } };
    if constexpr (PROF) pc = clock64();
    int nq = (__builtin_amdgcn_readfirstlane(dsc[1]) + 7) >> 3;   // halves in use
    const int g0 = half * 8;
    int qid[8], nqid[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      qid[u] = __builtin_amdgcn_readfirstlane(dsc[24 + g0 + u]);
      nqid[u] = qid[u];
    }
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
    fetch_row_terms(dsc);
    stash_row_terms();
    issue(0, 0, qid, nq);
    issue(1, 1, qid, nq);
    emit(0, slab, nq);          // phase 0 -> buffer 0 (waits for set 0)
    issue(0, 2, qid, nq);
    lds_barrier();
    for (;;) {
      const int nb = cur ^ 1;
      const int ngid = __builtin_amdgcn_readfirstlane(gidq[(ei + 1) & 1]);
      const bool have_next = ngid < n_work;
      int gid2 = 0;
      int next_nq = 0;
      int32_t rr0 = 0;
      uint32_t run0 = 0u, run1 = 0u;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        // the next entry's record: requested in phase 0, stored in phase 1, first read in phase 2
        if (j == 1 && tid < REC_DW) {
          if (tid == 0) gidq[ei & 1] = gid2;
          dsc[nb * REC_DW + tid] = rr0;
          if (wave == 0 && lane == 6) dsc[nb * REC_DW + 6] = have_next ? 1 : -1;
        }
        if (j == NP - 3) {   // (the next entry's record is in LDS since the barrier of phase 1)
          const int ncnt = have_next ? __builtin_amdgcn_readfirstlane(dsc[nb * REC_DW + 1]) : 0;
          next_nq = (ncnt + 7) >> 3;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            nqid[u] = !have_next ? qid[u] : __builtin_amdgcn_readfirstlane(dsc[nb * REC_DW + 24 + g0 + u]);
        }
        // slab of phase j + 1 of this entry -- or phase 0 of the next one -- from the set requested two phases ago,
        // then the same set again for phase j + 3
        const int set = (j + 1) & 1;
        if (j + 1 < NP) emit(set, slab + (uint32_t)((j + 1) & 1) * BUFB, nq);
        else emit(set, slab, next_nq);
        if (j + 3 < NP) issue(set, j + 3, qid, nq);
        else issue(set, j + 3 - NP, nqid, next_nq);   // (nqid is the next entry's from phase NP - 3 on: j + 3 >= NP <=> j >= NP - 3)
        if (j == 2) fetch_row_terms(dsc + (have_next ? nb : cur) * REC_DW);
        if (j == NP - 1 && a.tau_run) {   // the running bounds of this wave's two items (S1): on their way while the gatherers finish
          const int32_t* rc = dsc + cur * REC_DW;
          const int cn = __builtin_amdgcn_readfirstlane(rc[1]);
          run0 = wave < cn ? a.tau_run[(uint32_t)rc[24 + wave]] : 0u;
          run1 = wave + NG < cn ? a.tau_run[(uint32_t)rc[24 + wave + NG]] : 0u;
        }
        if (j == 0 && tid < REC_DW && have_next) rr0 = a.records[(size_t)ngid * REC_DW + tid];
        if (j == 0 && tid == 0) gid2 = (int)gridDim.x + atomicAdd(a.work_counter, 1);
        tick(0);
        lds_barrier();
        tick(1);
      }
      // S1: thresholds tau + E (builder wave w: items w and w + 8), while the gatherers are in their tail
      if (!(a.fence & 4)) {
        const int32_t* rec = dsc + cur * REC_DW;
        const int cnt = __builtin_amdgcn_readfirstlane(rec[1]);
        const int i0 = wave, i1 = wave + NG;
        if (i0 < cnt) {
          uint32_t c0 = colmin[i0 * 64 + lane], c1 = colmin[i1 * 64 + lane];
          wave_sort32_x2(c0, c1);   // (order-preserving keys of the float column minima)
          uint32_t t0 = __shfl(c0, a.L - 1, 64), t1 = __shfl(c1, a.L - 1, 64);
          if (lane == 0) {
            if (a.tau_run) {
              // The query's running bound (FilterArgs::tau_run).  tau' + A_up of this (item, chunk) is reported (no answer is
              // waited for); the bound read at the start of the last phase -- whatever other workgroups had reported by
              // then -- lowers this item's cut to bound - A_lo.  Why any such value is valid: tau' is at least the L-th smallest s'
              // of the chunk, s' + A is the cheap distance up to the item-independent part of its error, and the L-th smallest
              // over MORE rows is never larger -- so every reported value is an upper bound of the query's L-th smallest cheap
              // distance D_L, and the rows that can matter have s' + A <= D_L + 2 e.  (The three float roundings here are
              // below 10 u B of the 2.2 e = 230 u B + 13 T that E leaves over 2 e.)
              auto lower = [&](uint32_t t, int i, uint32_t inv) -> uint32_t {
                // (only a value that improves on what was read goes out: see running_bound5)
                return running_bound5(a.tau_run, (uint32_t)rec[24 + i], t, __int_as_float(rec[144 + i]), __int_as_float(rec[160 + i]), inv);
              };
              t0 = lower(t0, i0, run0);
              if (i1 < cnt) t1 = lower(t1, i1, run1);
            }
            thr_s[i0] = a.keep_all ? 0x7f800000u : widen_threshold5(t0, __int_as_float(rec[56 + i0]));
            thr_s[i1] = a.keep_all ? 0x7f800000u : widen_threshold5(t1, __int_as_float(rec[56 + i1]));
          }
          colmin[i0 * 64 + lane] = 0xffffffffu;
          colmin[i1 * 64 + lane] = 0xffffffffu;
        }
      }
      lds_barrier();   // S1
      stash_row_terms();   // (the gatherers took the current entry's into registers before their S1 barrier)
      lds_barrier();   // S2
      tick(3);
      if constexpr (PROF) pt[7] += 1;
      if (!have_next) break;
      cur = nb;
      ++ei;
      nq = next_nq;
#pragma unroll
      for (int u = 0; u < 8; ++u) qid[u] = nqid[u];
    }
    if (PROF && a.prof && tid == 0) {
      for (int i = 0; i < 8; ++i) if (i != 2 && i != 4 && i != 5) a.prof[(size_t)blockIdx.x * 8 + i] = pt[i];   // (2, 4, 5: gatherer wave 0)
      a.prof[(size_t)blockIdx.x * 8 + 6] = wall_clock64();
    }
  } else {
    // =====================================================================================
    // GATHERERS: lane <-> 8 rows x 12 items, a (row, item) sum = one 16-bit half of a register
    // =====================================================================================
    const int gw = wave - SPEC2_NB;
    uint32_t acc[G / 2][RMAX];     // [item pair][row]: low half = item 2i, high half = item 2i + 1
    uint32_t cwa[RMAX], cwb[RMAX]; // code dwords of the even / odd phases (double buffered: requested a phase ahead)
    // the first two rows' code dwords of phase 0 of the COMING entry, requested an entry ahead (at the end of the survivor pass): a
    // gatherer otherwise starts every entry with an exposed round trip for row 0's codes (the other rows' arrive behind it)
    constexpr int NPRE = 2;
    uint32_t cpre[NPRE] = {};
    auto prefetch_first_codes = [&](const int32_t* rc) {
      const int b0 = __builtin_amdgcn_readfirstlane(rc[3]), nbk = __builtin_amdgcn_readfirstlane(rc[4]);
      const uint32_t l4 = lane_byte4();
#pragma unroll
      for (int r = 0; r < NPRE; ++r) {
        const int bl = r * NG + gw;
        const uint32_t blk = (uint32_t)(b0 + (bl < nbk - 1 ? bl : nbk - 1));
        const char* rowp = U8 ? reinterpret_cast<const char*>(a.packed8) + (size_t)(blk * (uint32_t)(M / 4)) * 256u
                              : reinterpret_cast<const char*>(a.packed) + (size_t)(blk * (uint32_t)(M / 2)) * 256u;
        cpre[r] = *reinterpret_cast<const uint32_t*>(rowp + l4);
      }
    };
    prefetch_first_codes(dsc);
    lds_barrier();   // (pairs with the builders' barrier after the first slab)
    long long gt[3] = {0, 0, 0}, gc = 0;
    auto gtick = [&](int slot) { if constexpr (PROF) { const long long t = clock64(); if (slot >= 0) gt[slot] += t - gc; gc = t; } };
    for (;;) {
      gtick(-1);
      const int32_t* rec = dsc + cur * REC_DW;
      const int cnt = __builtin_amdgcn_readfirstlane(rec[1]);
      const int chunk = __builtin_amdgcn_readfirstlane(rec[2]);
      const int blk0 = __builtin_amdgcn_readfirstlane(rec[3]);
      const int nblk = __builtin_amdgcn_readfirstlane(rec[4]);
      const int nrows = __builtin_amdgcn_readfirstlane(rec[5]);
      const int nq = (cnt + 7) >> 3;     // 16-byte halves of a slab row in use
      const int nb = cur ^ 1;
      const int rl_wave = (nblk - gw + NG - 1) / NG < 0 ? 0 : (nblk - gw + NG - 1) / NG;
      auto row_block = [&](int r) {
        const int bl = r * NG + gw;
        return (uint32_t)(blk0 + (bl < nblk - 1 ? bl : nblk - 1));
      };
      auto main_loop = [&](auto nqc, auto rlc) {
        constexpr int NQ = decltype(nqc)::value, RL = decltype(rlc)::value;
        // (uniform 64-bit row address + this lane's byte offset, recomputed where it is needed: hoisted out of the entry
        // loop it was spilled, and every phase waited for the reload)
        auto load_codes = [&](uint32_t (&cw)[RMAX], int pair) {
          const uint32_t l4 = lane_byte4();
#pragma unroll
          for (int r = 0; r < RL; ++r) {
            const char* rowp = U8 ? reinterpret_cast<const char*>(a.packed8) + (size_t)((row_block(r) * (uint32_t)(M / 4) + (uint32_t)(pair >> 1)) * 256u)
                                  : reinterpret_cast<const char*>(a.packed) + (size_t)((row_block(r) * (uint32_t)(M / 2) + (uint32_t)pair) * 256u);
            cw[r] = *reinterpret_cast<const uint32_t*>(rowp + l4);
          }
        };
        // both positions of a phase, one row at a time: NQ ds_read_b128 per position fetch the values of 8 items each.
        // Half h of row c sits at byte 32 c + 16 (h ^ bit 3 of c): with the plain layout the first halves of all rows
        // would share 8 of the 16 bank groups.
        // Software pipelined: the reads of row r + 1 are issued before the sums of row r -- with one row's 2 NQ reads in
        // flight per wave the LDS pipe idled while the waves added (a closed loop: 8 waves x 4 reads, ~15 cycles of
        // service each, then ~100 cycles of adds before the next batch).
        auto gather = [&](const uint32_t (&cw)[RMAX], int j) {
          constexpr int DEPTH = NQ == 1 ? 4 : 2;   // rows in flight (32 registers either way; 3 rows of two halves spill)
          u4 va[DEPTH][2][NQ];   // [row slot][position of the phase][half]
          // byte offsets of the two codes' half rows: 16 c (| the buffer); halves and positions are constant offsets
          uint32_t bb = (uint32_t)(j & 1) * BUFB;
          asm volatile("" : "+s"(bb));   // (kept out of the constant folder: the halves / positions stay immediate offsets)
          auto issue_row = [&](int r) {
            // (U8: phase j's two codes are bytes 2 (j & 1) and 2 (j & 1) + 1 of the dword)
            const uint32_t a0 = (U8 ? ((j & 1) ? ((cw[r] >> 12) & 0xff0u) : ((cw[r] << 4) & 0xff0u)) : ((cw[r] << 4) & 0x3ff0u)) + bb;
            const uint32_t a1 = (U8 ? ((j & 1) ? ((cw[r] >> 20) & 0xff0u) : ((cw[r] >> 4) & 0xff0u)) : ((cw[r] >> 12) & 0x3ff0u)) + bb;
#pragma unroll
            for (int q = 0; q < NQ; ++q) va[r % DEPTH][0][q] = *reinterpret_cast<const u4*>(slab + a0 + (uint32_t)q * HALFB);
#pragma unroll
            for (int q = 0; q < NQ; ++q) va[r % DEPTH][1][q] = *reinterpret_cast<const u4*>(slab + a1 + POSB + (uint32_t)q * HALFB);
          };
#pragma unroll
          for (int r = 0; r < DEPTH - 1; ++r) if (r < RL) issue_row(r);
#pragma unroll
          for (int r = 0; r < RL; ++r) {
            if (r + DEPTH - 1 < RL) issue_row(r + DEPTH - 1);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              const u4 x = va[r % DEPTH][0][q], y = va[r % DEPTH][1][q];
              acc[q * 4 + 0][r] = acc[q * 4 + 0][r] + x.x + y.x;   // (v_add3_u32: unsigned fields, see filt5_bias)
              acc[q * 4 + 1][r] = acc[q * 4 + 1][r] + x.y + y.y;
              acc[q * 4 + 2][r] = acc[q * 4 + 2][r] + x.z + y.z;
              acc[q * 4 + 3][r] = acc[q * 4 + 3][r] + x.w + y.w;
            }
            if (r + 1 < RL) __builtin_amdgcn_sched_barrier(0);
          }
        };
#pragma unroll
        for (int h = 0; h < G / 2; ++h)
#pragma unroll
          for (int r = 0; r < RMAX; ++r) acc[h][r] = 0u;
        load_codes(cwa, 0);
#pragma unroll
        for (int r = 0; r < NPRE; ++r) if (r < RL) cwa[r] = cpre[r];   // (requested an entry ahead; the loads above for these rows are dropped by the compiler)
        load_codes(cwb, 1);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          if (!(a.fence & 2)) {
            if (j & 1) gather(cwb, j); else gather(cwa, j);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (j + 2 < NP && !(a.fence & 64)) {   // the codes of phase j + 2 into the set phase j has just used
            if (j & 1) load_codes(cwb, j + 2); else load_codes(cwa, j + 2);
          }
          if (j + 1 < NP) lds_barrier();
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
      // ---- tail.  The selection works on s' = fma(scale[item], V, rterm[row]) -- the stored sum WITHOUT the item's
      // constant OFF -- compared as floats: a constant shift changes neither the order nor tau' + E.  OFF (which keeps
      // the stored bits positive for the merge) is added for the survivors only: s = s' + OFF.
      // base[r] = the row's own term (staged by the builders in the previous entry's tail, replaced in this one's after the S1 barrier); +inf for the slots of this wave beyond its last block and for the lanes
      // past the end of the list (s = +inf: above every finite threshold; S2 skips the former and masks the latter)
      float base[RMAX];
      const int last_blk = nrows > 0 ? (nrows - 1) >> 6 : -1;     // chunk-relative block holding the last row
      const int rs2 = (last_blk >= 0 && (last_blk % NG) == gw && (nrows & 63)) ? last_blk / NG : -1;
      const bool live_lane = lane < (nrows & 63);
      {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          base[r] = rt_s[(r * NG + gw) * 64 + lane];
          if (r >= rl_wave || (r == rs2 && !live_lane)) base[r] = __uint_as_float(0x7f800000u);
        }
      }
#pragma unroll
      for (int h = 0; h < G / 2; ++h)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[h][r] ^= 0x80008000u;   // biased unsigned fields -> signed sums
      gtick(0);
      auto sval = [&](int g, int r, float sc) -> float {
        const uint32_t w = acc[g >> 1][r];
        const int v = (g & 1) ? ((int32_t)w >> 16) : ((int32_t)(w << 16) >> 16);
        return __builtin_fmaf(sc, (float)v, base[r]);
      };
      // Per-item parameters: lane g holds item g's (one LDS read each, fetched with v_readlane below -- a chain of
      // dependent LDS round trips per item was a quarter of the entry's time).
      const int gi = lane & 15;
      const float p_sc = __int_as_float(rec[128 + gi]);
      // rows this lane really holds: bit r of live8
      uint32_t live8 = 0u;
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < rl_wave && !(r == rs2 && !live_lane)) live8 |= 1u << r;
      float best[G];
      uint32_t sec16[G / 2];           // second smallest, rounded DOWN to 16 bits (sign, exponent, 7 bits): two items per register
      uint32_t apack[2] = {0u, 0u};
#pragma unroll
      for (int g = 0; g < G; ++g) best[g] = __uint_as_float(0x7f800000u);
#pragma unroll
      for (int i = 0; i < G / 2; ++i) sec16[i] = 0x7f807f80u;
      if (!(a.fence & 4)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g < cnt) {
            // the lane's smallest and second smallest s' of this item and the row of the smallest: a lane hardly ever
            // holds two survivors, so S2 can emit (best, its row) without looking at the sums again
            const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p_sc), g));
            float b1 = __uint_as_float(0x7f800000u), b2 = __uint_as_float(0x7f800000u);
            uint32_t ar = 0u;
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
              const float sv = sval(g, r, sc);
              b2 = __builtin_amdgcn_fmed3f(b1, b2, sv);   // (b1 <= b2: the median is the new second smallest)
              ar = sv < b1 ? (uint32_t)r : ar;
              b1 = fminf(b1, sv);
            }
            best[g] = b1;
            {
              const uint32_t bb = __float_as_uint(b2);
              const uint32_t dn = ((bb >> 31) ? bb + 0xffffu : bb) >> 16;   // toward -inf: the test below errs to the slow path
              sec16[g >> 1] = (g & 1) ? ((sec16[g >> 1] & 0x0000ffffu) | (dn << 16)) : ((sec16[g >> 1] & 0xffff0000u) | dn);
            }
            // (opaque: the compiler otherwise folds the shift into the eight selects above, whose constants 128, 192, ... are no inline
            // operands -- a v_mov per row and item)
            asm volatile("" : "+v"(ar));
            apack[g >> 3] |= ar << (3 * (g & 7));
            if (rl_wave > 0) atomicMin(colmin + g * 64 + lane, float_key(b1));
          }
        }
      }
      gtick(1);
      lds_barrier();
      // (S1, the thresholds tau' + E, is computed by the builder waves between these two barriers)
      lds_barrier();
      gtick(-1);
      // S2: survivors -> this wave's region of each item's buffer.  Normally every lane has at most one (its smallest
      // sum, kept from the pass above); otherwise the pass bits of the lane's 8 rows, branch free, then per-row ballots.
      if (!(a.fence & 4)) {
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
            if constexpr (!CAND) {   // the common case (freddy.c:366 counts retrieved rows): nothing but the threshold test
             const float second = __uint_as_float((g & 1) ? (sec16[g >> 1] & 0xffff0000u) : (sec16[g >> 1] << 16));
             const u64 multi = __ballot(!(second > thr));   // lanes with two survivors (or: keep every row, NaNs)
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
              for (int r = RMAX - 1; r >= 0; --r) m8 = m8 + m8 + (!(sval(g, r, sc) > thr) ? 1u : 0u);   // (a NaN passes: exact stage)
              m8 &= live8;
              if ((a.fence & 128) == 0 && __ballot(m8 != 0u) != 0ull) {
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
            } else {   // rows below the sentinel are counted (freddy.c:971): bounds on the bits of s = s' + OFF > 0
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
      gtick(2);
      const int next_ok = __builtin_amdgcn_readfirstlane(dsc[nb * REC_DW + 6]);
      if (next_ok > 0) prefetch_first_codes(dsc + nb * REC_DW);
      lds_barrier();
      if (next_ok < 0) break;
      cur = nb;
    }
    if (PROF && a.prof && gw == 0 && lane == 0) {
      a.prof[(size_t)blockIdx.x * 8 + 2] = gt[0];
      a.prof[(size_t)blockIdx.x * 8 + 4] = gt[1];
      a.prof[(size_t)blockIdx.x * 8 + 5] = gt[2];
    }
  }
}

}  // namespace freddy
