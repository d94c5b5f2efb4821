// fused4.h -- IVFADC as FILTER + REFINE: the bit-exact result without building exact LUTs for every row.
//
// The exact path (fused3.h) spends 9.4 G separately rounded lane-operations per 1024-query batch on
// the residual LUTs although only ~2k rows per query ever reach the result list.  Here
// the scan runs on a CHEAP distance with a PROVEN error bound, and the reference's arithmetic is
// replayed only for the rows that can still matter:
//
//   |r - c|^2 = |r|^2 + (|c|^2 + 2 co.c) - 2 q.c          r = q - co  (co = coarse centroid, c = codeword)
//                       `----- dt[cell][p][code]           pinned once (fp64 -> fp32)
//                                        `---- qc[query][p][code] = -2 q_p.c   one small kernel per batch
//
// so the slab of a position is ONE addition per (code, item): slab = dt + qc.  The |r|^2 term is the
// same for all rows of an item and is only needed as a bound (the coarse distance the probe plan
// already has).  With u = 2^-24, B = sum_p (|q_p| + max|co_p| + max|c_p|)^2 and E = 2048 u B:
//   * stored sum  s = OFF + sum_p slab_p  (OFF = A_up + E >= what keeps s positive)
//   * | (s - OFF + |r|^2) - d | <= 117 u B  for the reference's binary32 result d  (derivation: DESIGN.md 5.3b)
//   * d_lo = max(0, s - SHIFT) <= d <= d_lo + E
// Selection keeps every row with s <= tau + E (tau = the L-th column minimum, as in fused3.h): that
// set contains every row whose exact distance is <= the L-th smallest exact distance of the item,
// ties included.  Survivors carry (d_lo, row location); merge_refine_kernel (one wave per query)
// finds T = (L-th smallest d_lo) + E, recomputes the reference's distance -- sequential binary32
// squareDistance per position, positions added in order (index_utils.c:500-508, :1126-1133) -- for
// the rows with d_lo <= T only (typically L + 1 of ~130 survivors), and runs the same 2k-smallest
// selection and updateTopK replay as merge_surv_kernel on those exact keys.  Rows whose bound
// straddles the sentinel guard (freddy.c:971 counts them) are flagged and decided exactly as well.
// Non-finite inputs make E non-finite, which sends every row to the exact stage (slow, still exact).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fused.h"
#include "fused3.h"

namespace freddy {

struct FilterArgs {
  const float* qc;             // [Q][M][K]   -2 q_p . c            (query_codebook_kernel)
  const float* qn;             // [Q][M]      |q_p|, rounded up
  const float* dt;             // [C][M][K]   |c|^2 + 2 co_p . c    (pinned)
  const float* pmax;           // [M]         max_cell |co_p| + max_code |c_p|, rounded up (pinned)
  const float* dist;           // [Q][Cpad]   coarse distances |q - co|^2 of the probe plan
  int Cpad;
  const int32_t* item_query;   // [items]
  const int32_t* sorted_item;  // items in cell order
  const int32_t* group_cell;   // [groups]
  const int32_t* group_first;
  const int32_t* group_cnt;
  const int32_t* n_groups;
  int32_t* work_counter;
  const int32_t* blk_off;      // [C+1]
  const uint32_t* packed;      // [blocks][M2][64]
  const int32_t* pos;          // [blocks*64]
  u64* surv;                   // [items][upi][8 waves][512]: (bits(d_lo) << 32) | flag << 31 | row location
  int32_t* surv_count;
  int32_t* cand_count;         // [Q] or NULL: rows certainly below the sentinel (the flagged ones are added by the merge)
  int K, L, upi;
  float sentinel;
  uint32_t desc_offset;
  uint32_t ablate;
  long long* prof;
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
// qc[q][p][code] = -2 q_p . c_{p,code} and qn[q][p] = |q_p| (rounded up).  Thread <-> code (its S
// codebook values in registers), QT queries per workgroup through LDS.
// ---------------------------------------------------------------------------------------
template <int S, int QT>
__global__ __launch_bounds__(256) void query_codebook_kernel(const float* __restrict__ queries, const float* __restrict__ cbT,
                                                            float* __restrict__ qc, float* __restrict__ qn, int Q, int d, int m, int K) {
  constexpr int SP = (S + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float qs[QT][SP];
  const int tid = threadIdx.x, p = blockIdx.y, q0 = blockIdx.z * QT;
  const int c = blockIdx.x * 256 + tid;
  for (int i = tid; i < QT * SP; i += 256) {
    const int qi = i / SP, j = i - qi * SP;
    qs[qi][j] = (j < S && q0 + qi < Q) ? queries[(size_t)(q0 + qi) * d + p * S + j] : 0.0f;
  }
  float cb[S];
#pragma unroll
  for (int j = 0; j < S; ++j) cb[j] = c < K ? cbT[((size_t)p * S + j) * K + c] : 0.0f;
  __syncthreads();
  if (blockIdx.x == 0 && tid < QT && q0 + tid < Q) {
    float n2 = 0.0f;
#pragma unroll
    for (int j = 0; j < S; ++j) n2 = __builtin_fmaf(qs[tid][j], qs[tid][j], n2);
    qn[(size_t)(q0 + tid) * m + p] = __builtin_sqrtf(n2) * (1.0f + 1e-5f);
  }
  const int nq = (Q - q0 < QT) ? Q - q0 : QT;
  for (int qi = 0; qi < nq; ++qi) {
    float acc = 0.0f;
#pragma unroll
    for (int jb = 0; jb < SP / 4; ++jb) {
      const float4 v = *reinterpret_cast<const float4*>(&qs[qi][jb * 4]);
      if (jb * 4 + 0 < S) acc = __builtin_fmaf(v.x, cb[jb * 4 + 0 < S ? jb * 4 + 0 : 0], acc);
      if (jb * 4 + 1 < S) acc = __builtin_fmaf(v.y, cb[jb * 4 + 1 < S ? jb * 4 + 1 : 0], acc);
      if (jb * 4 + 2 < S) acc = __builtin_fmaf(v.z, cb[jb * 4 + 2 < S ? jb * 4 + 2 : 0], acc);
      if (jb * 4 + 3 < S) acc = __builtin_fmaf(v.w, cb[jb * 4 + 3 < S ? jb * 4 + 3 : 0], acc);
    }
    if (c < K) qc[((size_t)(q0 + qi) * m + p) * K + c] = -2.0f * acc;
  }
}

// dt[cell][p][code] = |c|^2 + 2 co_p . c in fp64, rounded once (pin time)
__global__ __launch_bounds__(256) void cell_codebook_kernel(const float* __restrict__ coarse, const float* __restrict__ cbT,
                                                           float* __restrict__ dt, int d, int m, int K, int S) {
  const int cell = blockIdx.x, p = blockIdx.y;
  for (int c = threadIdx.x; c < K; c += 256) {
    double acc = 0.0;
    for (int j = 0; j < S; ++j) {
      const double cv = (double)cbT[((size_t)p * S + j) * K + c];
      const double co = (double)coarse[(size_t)cell * d + p * S + j];
      acc += cv * cv + 2.0 * co * cv;
    }
    dt[((size_t)cell * m + p) * K + c] = (float)acc;
  }
}

// ---------------------------------------------------------------------------------------
// The scan.  Same roles, slab layout, barrier schedule, work entries and survivor regions as
// ivf_spec2_kernel; the builders only add two streamed tables now (prefetched two positions ahead).
// ---------------------------------------------------------------------------------------
template <int M, bool FULLK>
__global__ __launch_bounds__(SPEC2_T) void ivf_filter_kernel(FilterArgs a) {
  constexpr int G = SPEC2_G, RMAX = FUSED_RMAX, NG = SPEC2_NG;
  constexpr int M2 = M / 2;
  static_assert(M % 2 == 0 && G % 4 == 0 && G <= 16 && M >= 10, "layout");
  typedef float v2f __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);                                   // [2][K][G]
  uint32_t* colmin = reinterpret_cast<uint32_t*>(smem + a.desc_offset);           // [16][64]
  uint32_t* thr_s = colmin + 16 * 64;                                             // [16]
  int32_t* dsc = reinterpret_cast<int32_t*>(smem + a.desc_offset + 4096 + 64);    // as in fused3.h
  float* bnd = reinterpret_cast<float*>(smem + a.desc_offset + 4096 + 64 + 512);  // [2][5][16] item bounds
  float* pmax_s = bnd + 2 * 5 * 16;                                               // [16]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool builder = wave < SPEC2_NB;
  const int K = a.K;
  const int n_work = a.n_groups[0];

  // bounds of the items of descriptor buffer b (lane g < cnt of one wave)
  auto stage_bounds = [&](int b, int g, float A, float E) {
    const ItemBounds ib = item_bounds(A, E, a.sentinel);
    float* o = bnd + b * 80;
    o[g] = ib.off; o[16 + g] = ib.e; o[32 + g] = ib.shift;
    o[48 + g] = __uint_as_float(ib.lo_bits); o[64 + g] = __uint_as_float(ib.hi_bits);
  };

  // ---- first entry: fetched serially by everybody ----
  int cur = 0;
  if (tid == 0) dsc[32] = atomicAdd(a.work_counter, 1);
  for (int i = tid; i < 16 * 64; i += SPEC2_T) colmin[i] = 0xffffffffu;
  if (tid < 16) pmax_s[tid] = tid < M ? a.pmax[tid] : 0.0f;
  __syncthreads();
  {
    const int gid0 = dsc[32];
    if (gid0 >= n_work) return;
    if (wave == 0) {
      const int cell = a.group_cell[gid0], first = a.group_first[gid0], gc = a.group_cnt[gid0];
      const int cnt0 = gc & 0xff, chunk0 = gc >> 8;
      const int b0 = a.blk_off[cell] + chunk0 * FUSED_UNIT_BLOCKS;
      int nb0 = a.blk_off[cell + 1] - b0;
      if (nb0 > FUSED_UNIT_BLOCKS) nb0 = FUSED_UNIT_BLOCKS;
      if (lane < G) {
        const int it = (lane < cnt0) ? a.sorted_item[first + lane] : -1;
        const int q = it >= 0 ? a.item_query[it] : 0;
        dsc[lane] = it;
        dsc[64 + lane] = q;
        if (it >= 0) stage_bounds(0, lane, a.dist[(size_t)q * a.Cpad + cell], filter_width<M>(a.qn + (size_t)q * M, pmax_s));
      }
      if (lane == 0) { dsc[33] = cnt0; dsc[34] = b0; dsc[35] = nb0; dsc[36] = chunk0; dsc[37] = cell; }
    }
  }
  __syncthreads();

  if (builder) {
    // =====================================================================================
    // BUILDERS: lane <-> codes b and b+512
    // =====================================================================================
    const int b = tid;
    float dtv[2][2], qv[2][G][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int g = 0; g < G; ++g) qv[u][g][0] = qv[u][g][1] = 0.0f;
    const bool ok0 = FULLK || b < K, ok1 = FULLK || b + 512 < K;
    // request the table values of position p for an entry (cell, cnt items, their queries in LDS)
    auto issue = [&](int buf, int p, int cell, int cnt, const int (&qids)[G]) {
      if (cnt > 0) {
        const float* dp = a.dt + ((size_t)cell * M + p) * K;
        dtv[buf][0] = ok0 ? dp[b] : 0.0f;
        dtv[buf][1] = ok1 ? dp[b + 512] : 0.0f;
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g < cnt) {
          const float* qp = a.qc + ((size_t)qids[g] * M + p) * K;
          qv[buf][g][0] = ok0 ? qp[b] : 0.0f;
          qv[buf][g][1] = ok1 ? qp[b + 512] : 0.0f;
        }
      }
    };
    // slab rows are [code][12 items]: one aligned 16-byte store per item quad and code
    auto emit = [&](int buf, float* dst, int cnt) {
#pragma unroll
      for (int gq = 0; gq < G / 4; ++gq) {
        if (gq * 4 < cnt) {
          if (ok0)
            *reinterpret_cast<float4*>(dst + b * G + gq * 4) =
                float4{dtv[buf][0] + qv[buf][gq * 4 + 0][0], dtv[buf][0] + qv[buf][gq * 4 + 1][0],
                       dtv[buf][0] + qv[buf][gq * 4 + 2][0], dtv[buf][0] + qv[buf][gq * 4 + 3][0]};
          if (ok1)
            *reinterpret_cast<float4*>(dst + (b + 512) * G + gq * 4) =
                float4{dtv[buf][1] + qv[buf][gq * 4 + 0][1], dtv[buf][1] + qv[buf][gq * 4 + 1][1],
                       dtv[buf][1] + qv[buf][gq * 4 + 2][1], dtv[buf][1] + qv[buf][gq * 4 + 3][1]};
        }
      }
    };

    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc = 0;
    auto tick = [&](int slot) { if (a.prof) { const long long t = clock64(); pt[slot] += t - pc; pc = t; } };
    if (a.prof) pc = clock64();
    int cnt = __builtin_amdgcn_readfirstlane(dsc[33]);
    int cell = __builtin_amdgcn_readfirstlane(dsc[37]);
    int qid[G], nqid[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { qid[g] = __builtin_amdgcn_readfirstlane(dsc[64 + g]); nqid[g] = 0; }
    issue(0, 0, cell, cnt, qid);
    issue(1, 1, cell, cnt, qid);
    emit(0, slab, cnt);
    lds_barrier();
    for (;;) {
      int ngid = 0;
      int n_cell = 0, n_first = 0, n_gc = 0, n_item = -1, n_b0 = 0, n_b1 = 0, n_q = 0;
      float n_A = 0.0f, n_E = 0.0f;
      int next_gid = -1, next_cnt = 0, next_cell = 0;
      const int nb = cur ^ 1;
      if (tid == 0) ngid = atomicAdd(a.work_counter, 1);
#pragma unroll
      for (int p = 0; p < M; ++p) {
        tick(5);
        // slab(p+1) of this entry -- or slab(0) of the next one -- from the registers filled two phases ago
        if (p + 1 < M) emit((p + 1) & 1, slab + (size_t)((p + 1) & 1) * G * K, cnt);
        else emit(0, slab, next_cnt);
        // request position p+2
        if (p + 2 < M) {
          issue(p & 1, p + 2, cell, cnt, qid);
        } else {
          if (p + 2 == M) {
            next_gid = __builtin_amdgcn_readfirstlane(dsc[32 + nb * 8]);
            next_cnt = next_gid >= 0 ? __builtin_amdgcn_readfirstlane(dsc[32 + nb * 8 + 1]) : 0;
            next_cell = __builtin_amdgcn_readfirstlane(dsc[32 + nb * 8 + 5]);
#pragma unroll
            for (int g = 0; g < G; ++g) nqid[g] = __builtin_amdgcn_readfirstlane(dsc[64 + nb * 16 + g]);
          }
          issue(p & 1, p + 2 - M, next_cell, next_cnt, nqid);
        }
        tick(0);
        // next entry's descriptor and bounds, one dependent global round trip per position (wave 0 only)
        if (wave == 0) {
          if (p == 1) {
            ngid = __builtin_amdgcn_readfirstlane(ngid);
            if (ngid < n_work) {
              n_cell = a.group_cell[ngid];
              n_first = a.group_first[ngid];
              n_gc = a.group_cnt[ngid];
            }
          } else if (p == 3) {
            if (ngid < n_work) {
              if (lane < (n_gc & 0xff)) n_item = a.sorted_item[n_first + lane];
              n_b0 = a.blk_off[n_cell];
              n_b1 = a.blk_off[n_cell + 1];
            }
          } else if (p == 5) {
            if (ngid < n_work) {
              const int cntn = n_gc & 0xff, chn = n_gc >> 8;
              const int b0 = n_b0 + chn * FUSED_UNIT_BLOCKS;
              int nbn = n_b1 - b0;
              if (nbn > FUSED_UNIT_BLOCKS) nbn = FUSED_UNIT_BLOCKS;
              if (n_item >= 0) n_q = a.item_query[n_item];
              if (lane < G) dsc[nb * 16 + lane] = n_item;
              if (lane == 0) {
                dsc[32 + nb * 8 + 1] = cntn; dsc[32 + nb * 8 + 2] = b0;
                dsc[32 + nb * 8 + 3] = nbn; dsc[32 + nb * 8 + 4] = chn; dsc[32 + nb * 8 + 5] = n_cell;
              }
            }
          } else if (p == 6) {
            if (ngid < n_work && n_item >= 0) {
              n_A = a.dist[(size_t)n_q * a.Cpad + n_cell];
              n_E = filter_width<M>(a.qn + (size_t)n_q * M, pmax_s);
            }
          } else if (p == 7) {
            if (ngid < n_work && n_item >= 0) stage_bounds(nb, lane, n_A, n_E);
            if (lane < G) dsc[64 + nb * 16 + lane] = n_q;
            if (lane == 0) dsc[32 + nb * 8 + 0] = (ngid < n_work) ? ngid : -1;
          }
        }
        tick(5);
        lds_barrier();
        tick(1);
      }
      lds_barrier();   // S1
      lds_barrier();   // S2
      tick(3);
      pt[7] += 1;
      if (next_gid < 0) break;
      cur = nb;
      cnt = next_cnt;
      cell = next_cell;
#pragma unroll
      for (int g = 0; g < G; ++g) qid[g] = nqid[g];
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
      const int32_t* desc = dsc + cur * 16;
      const float* bn = bnd + cur * 80;
      const int cnt = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 1]);
      const int blk0 = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 2]);
      const int nblk = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 3]);
      const int chunk = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 4]);
      const int nb = cur ^ 1;
      auto row_block = [&](int r) {
        const int bl = r * NG + gw;
        return (uint32_t)(blk0 + (bl < nblk - 1 ? bl : nblk - 1));
      };
      auto load_codes = [&](int pair) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) cw[r] = a.packed[(row_block(r) * M2 + (uint32_t)pair) * 64u + (uint32_t)lane];
      };
      auto gather = [&](int p, const float* curs) {
        const int sh = (p & 1) * 16;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          const int code = (int)((cw[r] >> sh) & 0xffffu);
          const float* row = curs + code * G;
          float4 v[G / 4];
#pragma unroll
          for (int q = 0; q < G / 4; ++q) v[q] = *reinterpret_cast<const float4*>(row + q * 4);
#pragma unroll
          for (int q = 0; q < G / 4; ++q) {
            acc[q * 2 + 0][r] = acc[q * 2 + 0][r] + v2f{v[q].x, v[q].y};
            acc[q * 2 + 1][r] = acc[q * 2 + 1][r] + v2f{v[q].z, v[q].w};
          }
        }
      };
#pragma unroll
      for (int h = 0; h < G / 2; ++h) {
        const v2f o = v2f{bn[2 * h], bn[2 * h + 1]};
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[h][r] = o;
      }
      load_codes(0);
      for (int p = 0; p + 1 < M; ++p) {
        if (!(a.ablate & 2)) gather(p, slab + (size_t)(p & 1) * G * K);
        __builtin_amdgcn_sched_barrier(0);
        if (p & 1) load_codes((p + 1) >> 1);
        lds_barrier();
      }
      if (!(a.ablate & 2)) gather(M - 1, slab + (size_t)((M - 1) & 1) * G * K);
      int32_t pid[RMAX];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) pid[r] = a.pos[row_block(r) * 64u + (uint32_t)lane];
      {
        bool dead[RMAX];
        bool some = false;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) { dead[r] = !(((r * NG + gw) < nblk) && pid[r] >= 0); some |= dead[r]; }
        if (__ballot(some) != 0ull) {
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
#pragma unroll
            for (int h = 0; h < G / 2; ++h)
              if (dead[r]) acc[h][r] = v2f{__uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu)};
        }
      }
      if (!(a.ablate & 4)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g < cnt) {
            uint32_t best = bits(g, 0);
#pragma unroll
            for (int r = 1; r < RMAX; ++r) best = min(best, bits(g, r));
            atomicMin(colmin + g * 64 + lane, best);
          }
        }
      }
      lds_barrier();
      // S1: thresholds tau + E, two items per gatherer wave
      if (!(a.ablate & 4)) {
        static_assert(G <= 2 * NG, "at most two items per gatherer wave");
        const int g0 = gw, g1 = gw + NG;
        uint32_t c0 = colmin[g0 * 64 + lane], c1 = colmin[g1 * 64 + lane];
        wave_sort32_x2(c0, c1);
        const uint32_t t0 = __shfl(c0, a.L - 1, 64), t1 = __shfl(c1, a.L - 1, 64);
        if (lane == 0) {
          thr_s[g0] = widen_threshold(t0, bn[16 + g0]);
          thr_s[g1] = widen_threshold(t1, bn[16 + g1]);
        }
        colmin[g0 * 64 + lane] = 0xffffffffu;
        colmin[g1 * 64 + lane] = 0xffffffffu;
      }
      lds_barrier();
      // S2: survivors -> this wave's region of each item's buffer
      if (!(a.ablate & 4)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g < cnt) {
            const uint32_t thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)thr_s[g]);
            const int it = __builtin_amdgcn_readfirstlane(desc[g]);
            const float shift = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bn[32 + g])));
            const size_t region = ((size_t)it * a.upi + chunk) * NG + gw;
            u64* dst = a.surv + region * (size_t)(RMAX * 64);
            uint32_t lo_b = 0xffffffffu, hi_b = 0u;   // (no flagged rows unless the accepted rows are counted)
            if (a.cand_count) {
              lo_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bn[48 + g]));
              hi_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bn[64 + g]));
              int accepted = 0;
#pragma unroll
              for (int r = 0; r < RMAX; ++r) accepted += __popcll(__ballot(bits(g, r) < lo_b));
              if (lane == 0 && accepted) atomicAdd(a.cand_count + a.item_query[it], accepted);
            }
            int run = 0;
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
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
      const int next_gid = __builtin_amdgcn_readfirstlane(dsc[32 + nb * 8]);
      lds_barrier();
      if (next_gid < 0) break;
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
  const uint32_t* packed;
  const int32_t* pos;
  int32_t* cand_count;
  int32_t* out_ids;
  float* out_dist;
  int32_t* found;
  int32_t* next_active;
  int32_t* n_next;
  int32_t* status;
  int n_active, W, upi, L, k, found_rule, first_round, K, d;
  float sentinel;
};

template <int S, int M>
__global__ __launch_bounds__(64) void merge_refine_kernel(MergeRefineArgs a) {
  constexpr int NC = 16;            // candidates refined together
  constexpr int SQ = S + 1;         // row pitch of the squared differences
  constexpr int M2 = M / 2;
  __shared__ u64 stage[64];
  __shared__ float qs[M * S];
  __shared__ float sq[64 * SQ];
  __shared__ float lutv[NC * M];
  __shared__ int32_t cbo[64], coo[64];
  __shared__ u64 cq_key[64 + NC];
  __shared__ int32_t cq_cell[64 + NC];
  const int x = blockIdx.x, lane = threadIdx.x;
  const int q = a.active ? a.active[x] : x;
  const int k = a.k;
  const int per_item = a.upi * FUSED_NW;
  const int R = a.W * per_item;
  constexpr int NBATCH = 4;

  for (int j = lane; j < M * S; j += 64) qs[j] = a.queries[(size_t)q * a.d + j];
  const float E = filter_width<M>(a.qn + (size_t)q * M, a.pmax);

  // ---- pass 1: the L smallest lower bounds ----
  WaveSelect<1> sel;
  sel.init(stage, KEY_INF, a.L);
  for (int jb = 0; jb < R; jb += 64 * NBATCH) {
    int c[NBATCH];
    size_t region[NBATCH];
#pragma unroll
    for (int u = 0; u < NBATCH; ++u) {
      const int j = jb + u * 64 + lane;
      region[u] = (size_t)x * R + (size_t)(j < R ? j : 0);
      c[u] = (j < R) ? a.surv_count[region[u]] : 0;
    }
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
  // T = (L-th smallest d_lo) + E, rounded up; every key of the query if there are fewer than L or E is not finite
  uint32_t T_bits;
  {
    const u64 kth = wave_topk_at<1>(sel.acc, a.L - 1);
    T_bits = (kth == KEY_INF) ? 0xfffffffeu : widen_threshold((uint32_t)(kth >> 32), E);
  }
  __builtin_amdgcn_wave_barrier();

  // ---- pass 2: exact distances of the rows with d_lo <= T (and of the flagged ones) ----
  WaveSelect<1> sel2;
  sel2.init(stage, KEY_INF, a.L);
  int queued = 0;        // wave-uniform
  int amb_accepted = 0;  // lane 0..NC-1 partial counts
  auto refine = [&](int n) {   // the first n <= NC queue entries -> exact keys into sel2
    __builtin_amdgcn_wave_barrier();
    u64 out_key = KEY_INF;
    const int chains = n * M;
    for (int t0 = 0; t0 < chains; t0 += 64) {
      const int ch = t0 + lane;
      if (ch < chains) {
        const int c = ch / M, p = ch - c * M;
        const uint32_t loc = (uint32_t)cq_key[c] & 0x7fffffffu;
        const uint32_t word = a.packed[((size_t)(loc >> 6) * M2 + (uint32_t)(p >> 1)) * 64u + (loc & 63u)];
        const int code = (int)((word >> ((p & 1) * 16)) & 0xffffu);
        cbo[lane] = (p * a.K + code) * S;
        coo[lane] = cq_cell[c] * a.d + p * S;
      }
      __builtin_amdgcn_wave_barrier();
      const int nch = (chains - t0 < 64) ? chains - t0 : 64;
      // element e = (chain, dimension): consecutive lanes read consecutive floats of a codeword
      for (int e = lane; e < nch * S; e += 64) {
        const int cl = e / S, j = e - cl * S;
        const int p = (t0 + cl) % M;
        const float cv = a.cbR[(size_t)cbo[cl] + j];
        const float r = qs[p * S + j] - a.coarse[(size_t)coo[cl] + j];   // freddy.c:296-303
        const float t = r - cv;
        sq[cl * SQ + j] = t * t;                                          // index_utils.c:500-508
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < nch) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < S; ++j) acc = acc + sq[lane * SQ + j];
        lutv[t0 + lane] = acc;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane < n) {
      float dsum = 0.0f;
#pragma unroll
      for (int p = 0; p < M; ++p) dsum = dsum + lutv[lane * M + p];       // index_utils.c:1126-1133
      const uint32_t lo = (uint32_t)cq_key[lane];
      const int32_t pid = a.pos[lo & 0x7fffffffu];
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
  for (int jb = 0; jb < R; jb += 64) {
    const int j = jb + lane;
    const size_t region = (size_t)x * R + (size_t)(j < R ? j : 0);
    const int c = (j < R) ? a.surv_count[region] : 0;
    const int cell = (j < R) ? a.item_cell[(size_t)x * a.W + j / per_item] : 0;
    const u64* src = a.surv + region * (size_t)(FUSED_RMAX * 64);
    int maxc = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o, 64));
    for (int t = 0; t < maxc; ++t) {
      const bool valid = t < c;
      offer(valid ? src[t] : KEY_INF, valid, cell);
    }
  }
  while (queued > 0) refine(queued < NC ? queued : NC);
  sel2.finish();

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
