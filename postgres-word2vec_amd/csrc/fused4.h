// fused4.h -- IVFADC as FILTER + REFINE: the bit-exact result without building exact LUTs for every row.
//
// The exact path (fused3.h) spends 9.4 G separately rounded lane-operations per 1024-query batch on
// the residual LUTs although only ~2k rows per query ever reach the result list.  Here
// the scan runs on a CHEAP distance with a PROVEN error bound, and the reference's arithmetic is
// replayed only for the rows that can still matter:
//
//   |r - c|^2 = |r|^2 + (|c|^2 + 2 co.c) - 2 q.c          r = q - co  (co = coarse centroid, c = codeword)
//                       `----- summed over the row's codewords: rterm[row], pinned once (fp64 -> fp32)
//                                        `---- qc[query][p][code] = -2 q_p.c   one small kernel per batch
//
// so a row's running sum starts at rterm[row] and the slab of a position is ONE multiplication per
// (code, item): slab = scale * qc16.  The |r|^2 term is the same for all rows of an item and is only
// needed as a bound (the coarse distance the probe plan
// already has).  With u = 2^-24, B = sum_p (|q_p| + max|co_p| + max|c_p|)^2 and E = 2048 u B:
//   * stored sum  s = OFF + rterm + sum_p slab_p  (OFF = A_up + E >= what keeps s positive)
//   * | (s - OFF + |r|^2) - d | <= 264 u B  for the reference's binary32 result d  (derivation: DESIGN.md 5.3b)
//   * d_lo = max(0, s - SHIFT) <= d <= d_lo + E
// Selection keeps every row with s <= tau + E (tau = the L-th column minimum, as in fused3.h): that
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
  const uint32_t* qc;          // [Q][M][512] -2 q_p . c as int16 pairs (codes b, b+512)   (query_codebook_kernel)
  const float* rterm;          // [blocks*64] sum_p (|c|^2 + 2 co_p . c) of every row (pinned)
  const int32_t* records;      // [entries][REC_DW] (entry_record_kernel)
  const int32_t* n_groups;     // [1] number of work entries
  int32_t* work_counter;       // [1] zeroed before the launch
  const uint32_t* packed;      // [blocks][M2][64]
  u64* surv;                   // [items][upi][8 waves][512]: (bits(d_lo) << 32) | flag << 31 | row location
  int32_t* surv_count;
  int32_t* cand_count;         // [Q] or NULL: rows certainly below the sentinel (the flagged ones are added by the merge)
  int K, L, upi;
  float sentinel;
  uint32_t desc_offset;
  uint32_t ablate;
  long long* prof;
  // Quota-limited workgroups (fused5.h): the first `quota_wgs` workgroups of the grid stop after `quota` work entries;
  // the rest are persistent and drain the table.  quota_wgs = 0: every workgroup persistent.
  int quota, quota_wgs;
  // DIRECT mode of the integer-slab scan (fused5.h): static (cell, chunk) units, the cells' item counts of this batch, the
  // static record slots the probe plan filled, record slots per cell
  const int32_t* units;        // [n_units][4]: cell | chunk << 24, first row block, row blocks | rows << 8, 0
  const int32_t* cell_count;   // [C]
  const int32_t* drecs;        // [C][submax][DREC_DW]
  int n_units, submax;
  const uint32_t* packed8;     // U8 instantiation of the integer-slab scan: [blocks][3][64], one byte per code (K <= 256)
};

static constexpr float FILT_EPS = 2048.0f * 5.9604644775390625e-8f * 1.0001f;   // E = FILT_EPS * B

// E of a query: the same expression in the scan and in the merge (it must be the same number).
template <int M>
__device__ __forceinline__ float filter_width(const float* __restrict__ qn, const float* __restrict__ pmax) {
  float sb = 0.0f;
#pragma unroll
  for (int p = 0; p < M; ++p) {
    const float t = qn[p] + pmax[p];
    sb = __builtin_fmaf(t, t, sb);
  }
  return __builtin_fmaf(sb, FILT_EPS, 1e-30f);
}

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

// ---------------------------------------------------------------------------------------
// qc[q][p][code] = -2 q_p . c_{p,code}, stored as 16-bit fixed point with one scale per (query,
// position): |value| <= 2 |q_p| max|c_p| = 32767 * scale, so the quantisation error is <= scale / 2
// -- an ABSOLUTE bound that fits the budget E (a 16-bit float's relative error would not).  The table
// is what the scan streams per (query, cell) item, so its width is the scan's memory traffic.
// Layout [q][p][512] dwords: low half = code b, high half = code b + 512, pair b at dword 4 (b mod 128) + b / 128
// (a builder lane's four code pairs are one 16-byte load).
// Also writes qn[q][p] = |q_p| (rounded up) and qscale[q][p].  Thread <-> code pair, QT queries per
// workgroup through LDS.
// ---------------------------------------------------------------------------------------
template <int S, int QT>
__global__ __launch_bounds__(256) void query_codebook_kernel(const float* __restrict__ queries, const float* __restrict__ cbT,
                                                            const float* __restrict__ cmax, uint32_t* __restrict__ qc,
                                                            float* __restrict__ qn, float* __restrict__ qscale, int Q, int d, int m, int K) {
  // One workgroup = QT queries x ALL 512 code pairs of one position: thread t owns pairs t and t + 256, so the
  // workgroup produces whole 2 KB rows of the table.  The values of four queries at a time are laid out in LDS
  // exactly as the rows will sit in memory and leave as 16 bytes per lane, 1 KB per wave-level store (the
  // first version stored one dword per lane at a stride of 16 bytes, every cache line being completed by four
  // waves of two workgroups: 27 us alone, bound by those partial-line writes).
  constexpr int SP = (S + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float qs[QT][SP];
  __shared__ float inv_s[QT];
  __shared__ __attribute__((aligned(16))) uint32_t ob[2][4][512];   // [buffer][query of the group][dword of the row]
  const int tid = threadIdx.x, p = blockIdx.x, q0 = blockIdx.y * QT;
  for (int i = tid; i < QT * SP; i += 256) {
    const int qi = i / SP, j = i - qi * SP;
    qs[qi][j] = (j < S && q0 + qi < Q) ? queries[(size_t)(q0 + qi) * d + p * S + j] : 0.0f;
  }
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f cb[2][S];   // [pair t / t + 256][dimension] = (code b, code b + 512): one packed fma per dimension
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int b = tid + 256 * e;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      cb[e][j].x = b < K ? cbT[((size_t)p * S + j) * K + b] : 0.0f;
      cb[e][j].y = b + 512 < K ? cbT[((size_t)p * S + j) * K + b + 512] : 0.0f;
    }
  }
  __syncthreads();
  if (tid < QT) {
    float n2 = 0.0f;
#pragma unroll
    for (int j = 0; j < S; ++j) n2 = __builtin_fmaf(qs[tid][j], qs[tid][j], n2);
    const float nrm = __builtin_sqrtf(n2) * (1.0f + 1e-5f);
    const float sc = 2.0f * nrm * cmax[p] * (1.0f / 32767.0f) * (1.0f + 1e-6f);
    inv_s[tid] = (sc > 0.0f && sc < 1e30f) ? 1.0f / sc : 0.0f;
    if (q0 + tid < Q) {
      qn[(size_t)(q0 + tid) * m + p] = nrm;
      qscale[(size_t)(q0 + tid) * m + p] = sc;
    }
  }
  __syncthreads();
  const int nq = (Q - q0 < QT) ? Q - q0 : QT;
  static_assert(QT % 4 == 0, "four queries per step");
  // four queries per step = eight independent fma chains
  for (int qi = 0; qi < nq; qi += 4) {
    v2f acc[4][2];
#pragma unroll
    for (int w = 0; w < 4; ++w) acc[w][0] = acc[w][1] = v2f{0.0f, 0.0f};
#pragma unroll
    for (int jb = 0; jb < SP / 4; ++jb) {
      float vv[4][4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(&qs[qi + w][jb * 4]);
        vv[w][0] = v.x; vv[w][1] = v.y; vv[w][2] = v.z; vv[w][3] = v.w;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (jb * 4 + u < S) {
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const v2f qq = v2f{vv[w][u], vv[w][u]};
            acc[w][0] = __builtin_elementwise_fma(qq, cb[0][jb * 4 + u < S ? jb * 4 + u : 0], acc[w][0]);
            acc[w][1] = __builtin_elementwise_fma(qq, cb[1][jb * 4 + u < S ? jb * 4 + u : 0], acc[w][1]);
          }
        }
    }
    const int buf = (qi >> 2) & 1;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float inv = inv_s[qi + w];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int b = tid + 256 * e;
        const int i0 = (int)fminf(fmaxf(__builtin_rintf(-2.0f * acc[w][e].x * inv), -32767.0f), 32767.0f);
        const int i1 = (int)fminf(fmaxf(__builtin_rintf(-2.0f * acc[w][e].y * inv), -32767.0f), 32767.0f);
        // (dword 4 * (b mod 128) + b / 128: the scan's builder lane li loads pairs li, li+128, li+256, li+384 at once)
        ob[buf][w][4 * (b & 127) + (b >> 7)] = ((uint32_t)i0 & 0xffffu) | ((uint32_t)i1 << 16);
      }
    }
    __syncthreads();   // (two buffers: the next group's stores go to the other one, so one barrier per group is enough)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = tid + 256 * h;   // 16-byte element of the group's 4 x 2 KB
      const int w = i >> 7, e4 = i & 127;
      if (qi + w < nq)
        *reinterpret_cast<uint4*>(qc + ((size_t)(q0 + qi + w) * m + p) * 512 + 4 * e4) = *reinterpret_cast<const uint4*>(&ob[buf][w][4 * e4]);
    }
  }
}

// rterm[row] = sum_p (|c|^2 + 2 co_p . c) over the row's 12 codewords c and its cell's centroid co, in fp64,
// rounded once (pin time).  The part of the cheap distance that depends on (cell, row) only: it is the
// initial value of the row's running sum, so the scan streams nothing per cell.
__global__ __launch_bounds__(256) void row_term_kernel(const uint32_t* __restrict__ packed, const int32_t* __restrict__ blk_cell,
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
static constexpr int REC_DW = 272;   // + [128 + p*12 + g] fixed-point scale of item g at position p

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

template <int M>
__global__ __launch_bounds__(256) void entry_record_kernel(RecordArgs a) {
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
  int q = 0;
  if (lane < 16) {
    const int it = lane < cnt ? a.sorted_item[first + lane] : -1;
    q = a.item_query[it >= 0 ? it : a.sorted_item[first]];
    ItemBounds ib = item_bounds(0.0f, 0.0f, a.sentinel);
    if (it >= 0) ib = item_bounds(a.item_dist[it], filter_width<M>(a.qn + (size_t)q * M, a.pmax), a.sentinel);
    rec[8 + lane] = it;
    rec[24 + lane] = q;
    rec[40 + lane] = (int32_t)__float_as_uint(ib.off);
    rec[56 + lane] = (int32_t)__float_as_uint(ib.e);
    rec[72 + lane] = (int32_t)__float_as_uint(ib.shift);
    rec[88 + lane] = (int32_t)ib.lo_bits;
    rec[104 + lane] = (int32_t)ib.hi_bits;
  }
  for (int i0 = 0; i0 < M * 12; i0 += 64) {
    const int i = i0 + lane;
    const int p = i / 12, g = i - p * 12;
    const int qg = __shfl(q, g, 64);   // (every lane takes part; g < 12 always names a lane that holds a query)
    if (i < M * 12) rec[128 + i] = (int32_t)__float_as_uint(a.qscale[(size_t)qg * M + p]);
  }
}

// ---------------------------------------------------------------------------------------
// The scan.  Same roles, slab layout, barrier schedule and survivor regions as ivf_spec2_kernel.  The
// builders only add two streamed tables (requested one phase ahead, consumed at the start of a phase so
// that the wait is for loads that are a whole phase old); entries are assigned statically (entry i of
// workgroup w = the w-th of the i-th stripe of the largest-first table, stripes alternating direction).
// ---------------------------------------------------------------------------------------
template <int M, bool FULLK>
__global__ __launch_bounds__(SPEC2_T) void ivf_filter_kernel(FilterArgs a) {
  constexpr int G = SPEC2_G, RMAX = FUSED_RMAX, NG = SPEC2_NG;
  constexpr int M2 = M / 2;
  static_assert(M % 2 == 0 && G == 12 && M >= 6, "layout");
  typedef float v2f __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);                                   // [2][K][G]
  uint32_t* colmin = reinterpret_cast<uint32_t*>(smem + a.desc_offset);           // [16][64]
  uint32_t* thr_s = colmin + 16 * 64;                                             // [16]
  int32_t* dsc = reinterpret_cast<int32_t*>(smem + a.desc_offset + 4096 + 64);    // [2][REC_DW] entry records
  int32_t* gidq = dsc + 2 * REC_DW;                                                // [2] entry numbers: slot i & 1 = the workgroup's i-th
  float* rt_s = reinterpret_cast<float*>(gidq + 4);                                // [4096] row terms of the entry about to start

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool builder = wave < SPEC2_NB;
  const int K = FULLK ? 1024 : a.K;
  const int n_work = a.n_groups[0];

  // Entries are pulled from a device counter in largest-first order, TWO ahead: the number of the entry
  // after the next is requested while the current one runs, so neither the atomic nor the record load
  // that depends on it is ever waited for.
  int cur = 0, ei = 0;
  if (tid == 0) { gidq[0] = atomicAdd(a.work_counter, 1); gidq[1] = atomicAdd(a.work_counter, 1); }
  for (int i = tid; i < 16 * 64; i += SPEC2_T) colmin[i] = 0xffffffffu;
  __syncthreads();
  if (gidq[0] >= n_work) return;
  if (tid < REC_DW) dsc[tid] = a.records[(size_t)gidq[0] * REC_DW + tid];
  static_assert(REC_DW <= 5 * 64, "record prefetch by builder waves 0-4");
  __syncthreads();

  if (builder) {
    // =====================================================================================
    // BUILDERS: a pair of waves per item quad (waves 6-7 only stage records and row terms); lane li of
    // the pair <-> the code pairs li, li+128, li+256, li+384 (a pair = codes b and b+512) -- the qc table
    // is laid out so that these are the four dwords of ONE 16-byte load: a dword load moves 256 bytes per
    // wave-level instruction and the CU's address unit takes about as long for it as for 1 KB, which is
    // what bounded the builders (12 dword loads per lane and phase: ~1.4 k cycles per phase for the CU)
    // =====================================================================================
    const int grp = wave >> 1;                          // item quad of this wave pair
    const bool has_quad = grp < G / 4;
    const int li = (wave & 1) * 64 + lane;              // 0..127
    const uint32_t vq = (uint32_t)li * 16u;             // this lane's 16 bytes of a [512]-dword row of qc
    // three register sets: position p+3 is requested while p+1 is written -- with one phase of lead the
    // bytes a CU has in flight bound the table stream to ~3.4 TB/s chip-wide (latency ~2 us)
    static_assert(M % 3 == 0, "register sets rotate with the position, also across entries");
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 qw[3][4];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) qw[u][g] = u4{0u, 0u, 0u, 0u};
    // request the query-table values of position p for the quad's 4 item slots (their query ids): scalar
    // base of the item's [M][512] block + one 32-bit lane offset per position.  Always all 4 slots (unused
    // ones repeat item 0 and hit the L1): a STATIC number of loads lets the compiler wait for the set that
    // is two phases old and no younger one.
    typedef const char __attribute__((address_space(1))) * gptrc;
    typedef const u4 __attribute__((address_space(1))) * gptr4u;
    auto issue = [&](int buf, int p, const int (&qids)[4]) {
      if (!has_quad || (a.ablate & 32)) return;
      uint32_t voff = vq + (uint32_t)p * 2048u;
      asm volatile("" : "+v"(voff));   // opaque: keeps per-load 64-bit addresses from being materialised
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const gptrc qb = (gptrc)(uintptr_t)a.qc + (size_t)(uint32_t)qids[u] * (size_t)(M * 2048);
        qw[buf][u] = *(gptr4u)(qb + voff);
      }
    };
    // slab rows are [code][12 items]: one aligned 16-byte store per item quad and code; value =
    // scale * fixed-point qc.  sc = the entry record's scales of this position.  Store k of a lane goes to
    // code li + 128 k: consecutive lanes, consecutive 48-byte rows (conflict-free).
    auto emit = [&](int buf, float* dst, int nq, const int32_t* sc) {
      if (!has_quad || grp >= nq || (a.ablate & 16)) return;
      const float4 s4 = *reinterpret_cast<const float4*>(sc + grp * 4);
      const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t w = k == 0 ? qw[buf][u].x : k == 1 ? qw[buf][u].y : k == 2 ? qw[buf][u].z : qw[buf][u].w;
          lo[u] = (float)((int32_t)(w << 16) >> 16) * ss[u];
          hi[u] = (float)((int32_t)w >> 16) * ss[u];
        }
        const int b = li + 128 * k;
        if (FULLK || b < K) *reinterpret_cast<float4*>(dst + b * G + grp * 4) = float4{lo[0], lo[1], lo[2], lo[3]};
        if (FULLK || b + 512 < K) *reinterpret_cast<float4*>(dst + (b + 512) * G + grp * 4) = float4{hi[0], hi[1], hi[2], hi[3]};
      }
    };

    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc = 0;
    auto tick = [&](int slot) { if (a.prof) { const long long t = clock64(); pt[slot] += t - pc; pc = t; } };
    if (a.prof) pc = clock64();
    int nq = (__builtin_amdgcn_readfirstlane(dsc[1]) + 3) >> 2;
    const int g0 = has_quad ? grp * 4 : 0;   // first item slot of this wave pair's quad
    int qid[4], nqid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { qid[u] = __builtin_amdgcn_readfirstlane(dsc[24 + g0 + u]); nqid[u] = qid[u]; }
    // row terms of an entry (record rc): builder wave w fetches what gatherer wave w's lanes start from
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
    issue(0, 0, qid);
    issue(1, 1, qid);
    issue(2, 2, qid);
    emit(0, slab, nq, dsc + 128);
    lds_barrier();
    for (;;) {
      const int nb = cur ^ 1;
      const int ngid = __builtin_amdgcn_readfirstlane(gidq[(ei + 1) & 1]);
      const bool have_next = ngid < n_work;
      int gid2 = 0;   // (thread 0) the entry after the next
      int next_nq = 0;
      int32_t rr0 = 0;   // (threads < REC_DW) the next entry's record on its way to LDS
#pragma unroll
      for (int p = 0; p < M; ++p) {
        // the next entry's record: requested in P(0), stored in P(2), first read in P(M-3)
        if (p == 2 && tid < REC_DW) {
          if (tid == 0) gidq[ei & 1] = gid2;   // (slot of the current entry: read by everybody before P(0))
          dsc[nb * REC_DW + tid] = rr0;
          if (wave == 0 && lane == 6) dsc[nb * REC_DW + 6] = have_next ? 1 : -1;
        }
        // slab(p+1) of this entry -- or slab(0) of the next one -- from the registers requested two phases ago
        if (p + 1 < M) emit((p + 1) % 3, slab + (size_t)((p + 1) & 1) * G * K, nq, dsc + cur * REC_DW + 128 + (p + 1) * 12);
        else emit(0, slab, next_nq, dsc + nb * REC_DW + 128);
        // request position p+3
        if (p + 3 < M) {
          issue(p % 3, p + 3, qid);
        } else {
          if (p + 3 == M) {
            next_nq = have_next ? (__builtin_amdgcn_readfirstlane(dsc[nb * REC_DW + 1]) + 3) >> 2 : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) nqid[u] = have_next ? __builtin_amdgcn_readfirstlane(dsc[nb * REC_DW + 24 + g0 + u]) : qid[u];
          }
          issue(p % 3, p + 3 - M, nqid);
        }
        // the next entry's row terms: requested in P(4) (its record is in LDS since P(2)), staged in P(7) --
        // the gatherers read the current entry's before P(0)
        if (p == 4) fetch_row_terms(dsc + (have_next ? nb : cur) * REC_DW);
        if (p == 7) stash_row_terms();
        if (p == 0 && tid < REC_DW && have_next) rr0 = a.records[(size_t)ngid * REC_DW + tid];
        if (p == 0 && tid == 0) gid2 = atomicAdd(a.work_counter, 1);
        tick(0);
        lds_barrier();
        tick(1);
      }
      // S1 (thresholds tau + E) is done HERE, by the builder waves, which have nothing else to do in the tail
      // phases: the gatherers' tail (column minima, survivor passes) is what the SIMDs' issue slots go to, and
      // the two 64-key sorts per wave were a fifth of it.  Builder wave w takes items w and w + 8.
      if (!(a.ablate & 4)) {
        const int32_t* rec = dsc + cur * REC_DW;
        const int cnt = __builtin_amdgcn_readfirstlane(rec[1]);
        const int g0 = wave, g1 = wave + NG;
        if (g0 < cnt) {
          uint32_t c0 = colmin[g0 * 64 + lane], c1 = colmin[g1 * 64 + lane];
          wave_sort32_x2(c0, c1);
          const uint32_t t0 = __shfl(c0, a.L - 1, 64), t1 = __shfl(c1, a.L - 1, 64);
          if (lane == 0) {
            // (ablate & 8, tests only: keep EVERY row, so that the exact stage -- and its self-check of the
            // bracket -- sees all of them)
            thr_s[g0] = (a.ablate & 8) ? 0xfffffffeu : widen_threshold(t0, __int_as_float(rec[56 + g0]));
            thr_s[g1] = (a.ablate & 8) ? 0xfffffffeu : widen_threshold(t1, __int_as_float(rec[56 + g1]));
          }
          colmin[g0 * 64 + lane] = 0xffffffffu;
          colmin[g1 * 64 + lane] = 0xffffffffu;
        }
      }
      lds_barrier();   // S1
      lds_barrier();   // S2
      tick(3);
      pt[7] += 1;
      if (!have_next) break;
      cur = nb;
      ++ei;
      nq = next_nq;
#pragma unroll
      for (int u = 0; u < 4; ++u) qid[u] = nqid[u];
    }
    if (a.prof && tid == 0) {
      for (int i = 0; i < 8; ++i) a.prof[(size_t)blockIdx.x * 8 + i] = pt[i];
      a.prof[(size_t)blockIdx.x * 8 + 6] = wall_clock64();
    }
  } else {
    // =====================================================================================
    // GATHERERS: lane <-> 8 rows x 12 items
    // =====================================================================================
    const int gw = wave - SPEC2_NB;
    v2f acc[G / 2][RMAX];
    uint32_t cw[RMAX];
    auto bits = [&](int g, int r) { return __float_as_uint((g & 1) ? acc[g >> 1][r].y : acc[g >> 1][r].x); };
    lds_barrier();   // (pairs with the builders' barrier after the first slab)
    for (;;) {
      const int32_t* rec = dsc + cur * REC_DW;
      const int cnt = __builtin_amdgcn_readfirstlane(rec[1]);
      const int chunk = __builtin_amdgcn_readfirstlane(rec[2]);
      const int blk0 = __builtin_amdgcn_readfirstlane(rec[3]);
      const int nblk = __builtin_amdgcn_readfirstlane(rec[4]);
      const int nrows = __builtin_amdgcn_readfirstlane(rec[5]);
      const int nq = (cnt + 3) >> 2;
      const int nb = cur ^ 1;
      const int rl_wave = (nblk - gw + NG - 1) / NG < 0 ? 0 : (nblk - gw + NG - 1) / NG;   // row slots of this wave that hold rows
      auto row_block = [&](int r) {
        const int bl = r * NG + gw;
        return (uint32_t)(blk0 + (bl < nblk - 1 ? bl : nblk - 1));
      };
      // The main loop, instantiated per (item quads NQ the entry has, row slots RL this wave has rows in):
      // LDS bandwidth bounds it, so the reads of unused item slots and of row slots past the end of the
      // list are not issued.  (Selected OUTSIDE the position loop: conditional updates inside it make
      // hipcc copy the whole register tile.)  Every variant passes the same barriers.
      auto main_loop = [&](auto nqc, auto rlc) {
        constexpr int NQ = decltype(nqc)::value, RL = decltype(rlc)::value;
        auto load_codes = [&](int pair) {
#pragma unroll
          for (int r = 0; r < RL; ++r) cw[r] = a.packed[(row_block(r) * M2 + (uint32_t)pair) * 64u + (uint32_t)lane];
        };
        // one row at a time: NQ ds_read_b128 fetch a row's item values
        auto gather = [&](int p) {
          const float* curs = slab + (size_t)(p & 1) * G * K;
          const int sh = (p & 1) * 16;
#pragma unroll
          for (int r = 0; r < RL; ++r) {
            const int code = (int)((cw[r] >> sh) & 0xffffu);
            const float* row = curs + code * G;
            float4 v[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) v[q] = *reinterpret_cast<const float4*>(row + q * 4);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              acc[q * 2 + 0][r] = acc[q * 2 + 0][r] + v2f{v[q].x, v[q].y};
              acc[q * 2 + 1][r] = acc[q * 2 + 1][r] + v2f{v[q].z, v[q].w};
            }
          }
        };
        // sums start at OFF + the row's own term (staged in LDS by the builders during the previous entry).
        // Row slots >= RL hold no rows: parked.
#pragma unroll
        for (int r = 0; r < RL; ++r) cw[r] = __float_as_uint(rt_s[(r * NG + gw) * 64 + lane]);
#pragma unroll
        for (int h = 0; h < G / 2; ++h) {
          const v2f o = v2f{__int_as_float(rec[40 + 2 * h]), __int_as_float(rec[40 + 2 * h + 1])};
#pragma unroll
          for (int r = 0; r < RL; ++r) acc[h][r] = o + v2f{__uint_as_float(cw[r]), __uint_as_float(cw[r])};
#pragma unroll
          for (int r = RL; r < RMAX; ++r) acc[h][r] = v2f{__uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu)};
        }
        load_codes(0);
        for (int p = 0; p + 1 < M; ++p) {
          if (!(a.ablate & 2)) gather(p);
          __builtin_amdgcn_sched_barrier(0);
          if ((p & 1) && !(a.ablate & 64)) load_codes((p + 1) >> 1);   // (64: timing experiment, the first pair's codes for all positions)
          lds_barrier();
        }
        if (!(a.ablate & 2)) gather(M - 1);
      };
      {
        int rl = rl_wave;
        rl = rl < 1 ? 1 : rl;
        const int rc = (rl + 1) >> 1;         // 1..4 -> RL = 2, 4, 6, 8
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
        using I6 = std::integral_constant<int, 6>; using I8 = std::integral_constant<int, 8>;
        switch (nq * 4 + rc) {
          case 1 * 4 + 1: main_loop(I1{}, I2{}); break;
          case 1 * 4 + 2: main_loop(I1{}, I4{}); break;
          case 1 * 4 + 3: main_loop(I1{}, I6{}); break;
          case 1 * 4 + 4: main_loop(I1{}, I8{}); break;
          case 2 * 4 + 1: main_loop(I2{}, I2{}); break;
          case 2 * 4 + 2: main_loop(I2{}, I4{}); break;
          case 2 * 4 + 3: main_loop(I2{}, I6{}); break;
          case 2 * 4 + 4: main_loop(I2{}, I8{}); break;
          case 3 * 4 + 1: main_loop(I3{}, I2{}); break;
          case 3 * 4 + 2: main_loop(I3{}, I4{}); break;
          case 3 * 4 + 3: main_loop(I3{}, I6{}); break;
          default: main_loop(I3{}, I8{}); break;
        }
      }
      // Rows past the end of the list are parked above everything.  Row slots of this wave beyond its last block
      // (the gather variants round the slot count up to an even number and re-read the last block there) are
      // parked whole -- a uniform test per slot --, and only the LAST row block of the chunk can be partial: one
      // slot of one wave gets the per-lane test.  (Testing every slot of every wave per lane was a tenth of the tail.)
      {
        const int last_blk = nrows > 0 ? (nrows - 1) >> 6 : -1;     // chunk-relative block holding the last row
        const int rs = (last_blk >= 0 && (last_blk % NG) == gw && (nrows & 63)) ? last_blk / NG : -1;
        const bool dead = lane >= (nrows & 63);
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          if (r >= rl_wave) {
#pragma unroll
            for (int h = 0; h < G / 2; ++h) acc[h][r] = v2f{__uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu)};
          } else if (r == rs) {
#pragma unroll
            for (int h = 0; h < G / 2; ++h)
              if (dead) acc[h][r] = v2f{__uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu)};
          }
        }
      }
      if (!(a.ablate & 4)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g < cnt) {
            uint32_t best = bits(g, 0);
#pragma unroll
            for (int r = 1; r < RMAX; ++r) best = min(best, bits(g, r));
            if (rl_wave > 0) atomicMin(colmin + g * 64 + lane, best);
          }
        }
      }
      lds_barrier();
      // (S1, the thresholds tau + E, is computed by the builder waves between these two barriers)
      lds_barrier();
      // S2: survivors -> this wave's region of each item's buffer
      if (!(a.ablate & 4)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g < cnt) {
            const uint32_t thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)thr_s[g]);
            const int it = __builtin_amdgcn_readfirstlane(rec[8 + g]);
            const float shift = __int_as_float(__builtin_amdgcn_readfirstlane(rec[72 + g]));
            const size_t region = ((size_t)it * a.upi + chunk) * NG + gw;
            u64* dst = a.surv + region * (size_t)(RMAX * 64);
            uint32_t lo_b = 0xffffffffu, hi_b = 0u;   // (no flagged rows unless the accepted rows are counted)
            if (a.cand_count) {
              lo_b = (uint32_t)__builtin_amdgcn_readfirstlane(rec[88 + g]);
              hi_b = (uint32_t)__builtin_amdgcn_readfirstlane(rec[104 + g]);
              int accepted = 0;
#pragma unroll
              for (int r = 0; r < RMAX; ++r) accepted += __popcll(__ballot(bits(g, r) < lo_b));
              if (lane == 0 && accepted) atomicAdd(a.cand_count + __builtin_amdgcn_readfirstlane(rec[24 + g]), accepted);
            }
            int run = 0;
            if (!a.cand_count) {   // the common case (freddy.c:366 counts retrieved rows): nothing but the threshold test
#pragma unroll
              for (int r = 0; r < RMAX; ++r) {
                if (r >= rl_wave) break;   // (uniform: the wave's remaining row slots are past the end of the list)
                const uint32_t sb = bits(g, r);
                const u64 mask = __ballot(sb <= thr);
                if (mask != 0ull) {
                  if (sb <= thr) {
                    const float dlo = fmaxf(0.0f, __uint_as_float(sb) - shift);
                    const uint32_t loc = (uint32_t)(blk0 + r * NG + gw) * 64u + (uint32_t)lane;
                    dst[run + lanes_below(mask)] = ((u64)__float_as_uint(dlo) << 32) | (u64)loc;
                  }
                  run += __popcll(mask);
                }
              }
            } else
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
              if (r >= rl_wave) break;
              const uint32_t sb = bits(g, r);
              const bool amb = sb >= lo_b && sb < hi_b;
              const bool pass = sb <= thr || amb;
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
            if (lane == 0) a.surv_count[region] = run;
          }
        }
      }
      const int next_ok = __builtin_amdgcn_readfirstlane(dsc[nb * REC_DW + 6]);
      lds_barrier();
      if (next_ok < 0) break;
      cur = nb;
    }
  }
}

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
  const float* qscale5;        // [Q] or NULL: the integer-slab scan's per-query table scale (selects its margin E)
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
  uint32_t ablate;   // timing experiments only (FREDDY_GPU_MERGE_ABLATE): 1 = skip the exact stage
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

  if (!(a.ablate & 2)) for (int j = threadIdx.x; j < M * S; j += 64 * NWV) qs[j] = a.queries[(size_t)q * a.d + j];
  const float E = (a.ablate & 2) ? 0.0f
                  : a.qscale5 ? filter_width5<M>(a.qn + (size_t)q * M, a.pmax, a.qscale5[q])
                              : filter_width<M>(a.qn + (size_t)q * M, a.pmax);

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
  const int LW = (a.ablate & 4) ? a.L : ((a.L + 22 < 64 && R <= 64 * NBATCH) ? a.L + 22 : 64);
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
  if (a.ablate & 8) {
    if (lane < k) a.out_ids[(size_t)q * k + lane] = (int32_t)sel.acc[0];
    if (lane == 0) sh_n = 0;
    __syncthreads();
    return;
  }
  // T = (L-th smallest d_lo) + E, rounded up; every key of the query if there are fewer than L or E is not finite
  uint32_t T_bits;
  {
    const u64 kth = wave_topk_at<1>(sel.acc, a.L - 1);
    T_bits = (kth == KEY_INF || (a.ablate & 32)) ? 0xfffffffeu : widen_threshold((uint32_t)(kth >> 32), E);   // (32: tests, every row)
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
        if (a.ablate & 32) atomicAdd(a.violations + 1, 1);
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
    if (a.ablate & 16) {   // debugging aid: what the exact stage would be asked to do
      if (lane == 0) {
        float* o = a.out_dist + (size_t)q * k;
        o[0] = (float)__popcll(in_mask); o[1] = all_in ? 1.0f : 0.0f; o[2] = E; o[3] = __uint_as_float(T_bits);
        o[4] = __uint_as_float((uint32_t)(wave_topk_at<1>(sel.acc, a.L - 1) >> 32));
        sh_n = 0;
      }
      __syncthreads();
      return;
    }
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
        if (!(a.ablate & 1)) round4(n); else queued = 0;
      }
    }
  }
  if (revisit) sweep([&](u64 kk, bool valid, int j) { offer(kk, valid, valid ? a.item_cell[(size_t)x * a.W + j / per_item] : 0); }, 0, PB);
  if (a.ablate & 1) queued = 0;
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
