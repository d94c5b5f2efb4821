// coarse.h -- coarse-cell selection (a6 / a7) as FILTER + REFINE, like the scan of fused5.h.
//
// The reference computes squareDistance(q, cq[j], d) for EVERY (query, cell) pair (freddy.c:272-283, :855-866):
// Q*C*d separately rounded sub / mul / add triples (0.92 G lane-operations per 1024-query batch, 28-34 us as
// coarse_tile_kernel) of which only the W nearest cells of a query ever matter.  Here
//
//   a[q][j] = |q|^2 + |c_j|^2 - 2 q.c_j          one fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32: an fmaf chain)
//
// ranks the cells, with a proven bracket |a - d| <= eps(q) around the reference's binary32 result d, and the
// reference's chain -- (q_i - c_i) rounded, squared, added in order i = 0..d-1 from +0 (index_utils.c:500-508) --
// is evaluated only for the cells whose bracket reaches the 2W-th smallest upper bound: a superset of the 2W
// smallest exact (distance, cell) keys, which is all the reference's cell list can depend on (DESIGN.md 3,
// "selection-then-replay"; same argument as probe_plan_kernel in kernels.h).  The approximate value never
// reaches a result: the probe plan, the item bounds of the scan and the tie behaviour all use the exact d.
//
// The bracket (u = 2^-24, D = exact real |q - c|^2 <= (|q| + |c|)^2):
//   reference      |d - D| <= ((1+u)^(d+3) - 1) D                      <= 1.82e-5 (|q| + |c|)^2   for d <= 300
//   dot product    fmaf chain of <= 320 terms (any order):  <= 320 u |q||c|, doubled by the factor 2,
//                  |q||c| <= (|q| + |c|)^2 / 4                          <= 0.91e-5 (|q| + |c|)^2
//   |q|^2, |c|^2   |q|^2: two fp32 fma chains of 152 terms, added; |c|^2: fp64, rounded once  <= 0.92e-5 (|q| + |c|)^2
//   final add/fma  2 u (|q| + |c|)^2
//   sum < 3.7e-5 (|q| + |c|)^2;   eps(q) = COARSE_EPS (|q| + max_j |c_j|)^2 with COARSE_EPS = 1.2e-4 (3x margin,
//   and it would still hold if the matrix core TRUNCATED every accumulation step instead of rounding it).
// Every refined cell has both numbers in hand: the kernel counts the cells whose d left [a - eps, a + eps]
// (freddy_gpu_filter_bound_violations; the tests also run with EVERY cell refined).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "wave_topk.h"
#include "refine.h"

namespace freddy {

static constexpr float COARSE_EPS = 1.2e-4f;
static constexpr int COARSE_DP_ALIGN = 64;    // padded dimension count: whole blocks of 8 iterations x 8 dimensions (zeros)
static constexpr int COARSE_TQ = 32;          // queries per MFMA tile (a workgroup = 32 queries x 128 cells)
static constexpr int COARSE_MAX_CPAD = 1024;  // the plan keeps a query's approximate distances in registers: 16 per lane
static constexpr int COARSE_STREAM_MAX_CPAD = 16384;   // beyond 1024 cells the plan streams them twice (minima, then candidates as a bitmap)

// ---------------------------------------------------------------------------------------
// a[q][j] for a 32-query x 128-cell tile per workgroup; wave w owns the 32 x 32 block of cells [32 w, 32 w + 32).
// (Until round 2 the tile was 64 x 64 with 83 KB of LDS: one workgroup per CU, and with several batches in flight a
// kernel of such workgroups waits for CUs the other batches' scans do not hold; 32 query rows are 41 KB.)
// The k index of an MFMA step may be any permutation as long as A and B agree: step t of iteration i pairs
// elements 8 i + t (lanes 0-31) and 8 i + 4 + t (lanes 32-63), so a lane's operands for FOUR consecutive MFMAs
// are 16 contiguous bytes of its row.
//   B (centroids): pinned in FRAGMENT order coarseF[Cpad / 32][dp / 8][64 lanes][4] -- a wave's operand load is
//     one fully coalesced 1 KB read, several iterations in flight (a first version read row-major centroids
//     and queries directly, 64 different cache lines per wave-level load: 28 us, latency-bound);
//   A (queries, row-major from the caller): the tile's 32 rows are staged ONCE in LDS by coalesced 16-byte
//     loads, all in flight together (row pitch dp + 4 floats: the ds_read_b128 of 16 consecutive rows at one
//     column hit 16 different 4-bank groups).
//   cn2 [Cpad] = |c_j|^2 (fp64 sum rounded once, pin time), out [Q][Cpad], qn2 [Q] = |q|^2.
// Also clears the round-one scratch (ZeroArgs), as coarse_tile_kernel does.
// ---------------------------------------------------------------------------------------
// (the body takes its block coordinates and its LDS from the caller: coarse_table5_kernel, fused5.h, runs it beside the
// query x codebook table in one launch)
__device__ __forceinline__ void coarse_approx_body(const float* __restrict__ queries, const float* __restrict__ coarseF,
                                                   const float* __restrict__ cn2, float* __restrict__ out,
                                                   float* __restrict__ qn2, int Q, int Cpad, int d, int dp, const ZeroArgs& z,
                                                   int bx, int by, int gx, int gy, unsigned char* smem,
                                                   float* __restrict__ tmin = nullptr, int C = 0) {
  // (host guarantees: d % 4 == 0, dp % 64 == 0 -- whole blocks of UN = 8 iterations, zero padded on both sides,
  // so that the loops below carry no guards: guarded loads made hipcc emit a branch per element)
  typedef float f16v __attribute__((ext_vector_type(16)));
  {
    const int gtid = (by * gx + bx) * 256 + threadIdx.x, gsz = gx * gy * 256;
#pragma unroll
    for (int a = 0; a < 5; ++a)
      for (int i = gtid; i < z.n[a]; i += gsz) z.p[a][i] = 0u;
  }
  constexpr int TQ = COARSE_TQ;                       // query rows of a tile
  const int PA = dp + 4;
  float* As = reinterpret_cast<float*>(smem);        // [TQ][PA]
  float* rown = As + TQ * PA;                         // [4][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q0w = 0;
  const int q0 = by * TQ, c0 = bx * 128 + wave * 32;
  const int nit = dp >> 3;
  const float4* bp = reinterpret_cast<const float4*>(coarseF) + ((size_t)(c0 >> 5) * nit) * 64 + lane;
  constexpr int UN = 8;   // B operands in flight per wave
  float4 bv[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) bv[u] = bp[(size_t)u * 64];
  // Stage the query tile: its rows are ONE contiguous span of the row-major query matrix, copied flat with
  // every load of a thread in flight together (a per-row formulation with clamped indices measured 13 us for
  // this step alone, the flat copy 2); columns d..dp-1 are zeroed, rows beyond Q hold junk that is never stored.
  {
    const int d4n = d >> 2;
    const int rows = Q - q0 < TQ ? Q - q0 : TQ;
    const int n4 = rows * d4n;
    const float4* src = reinterpret_cast<const float4*>(queries + (size_t)q0 * d);
    constexpr int SB = 10;   // 32 rows x 300 floats = 2400 float4 = 9.4 per thread
    for (int base = 0; base < TQ * d4n; base += 256 * SB) {
      float4 v[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int i = base + u * 256 + tid;
        v[u] = src[i < n4 ? i : n4 - 1];
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int i = base + u * 256 + tid;
        if (i < TQ * d4n) {
          const int row = i / d4n, c4 = i - row * d4n;
          *reinterpret_cast<float4*>(As + row * PA + c4 * 4) = v[u];
        }
      }
    }
    const int p4n = (dp - d) >> 2;   // zero columns
    for (int i = tid; i < TQ * p4n; i += 256) {
      const int row = i / p4n, c4 = i - row * p4n;
      *reinterpret_cast<float4*>(As + row * PA + d + c4 * 4) = float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  f16v acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
  float nrm = 0.0f;
  const float* arow = As + (q0w + r) * PA + 4 * h;
  for (int i0 = 0; i0 < nit; i0 += UN) {
    float4 cur[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) cur[u] = bv[u];
    const int inb = i0 + UN < nit ? i0 + UN : i0;   // (last block: re-request the current one, unused)
#pragma unroll
    for (int u = 0; u < UN; ++u) bv[u] = bp[(size_t)(inb + u) * 64];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float4 av = *reinterpret_cast<const float4*>(arow + (i0 + u) * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, cur[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, cur[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, cur[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, cur[u].w, acc, 0, 0, 0);
      nrm = __builtin_fmaf(av.x, av.x, nrm);
      nrm = __builtin_fmaf(av.y, av.y, nrm);
      nrm = __builtin_fmaf(av.z, av.z, nrm);
      nrm = __builtin_fmaf(av.w, av.w, nrm);
    }
  }
  nrm += __shfl_xor(nrm, 32, 64);
  if (h == 0) {
    rown[wave * 32 + r] = nrm;
    if (wave == 0 && bx == 0 && q0 + q0w + r < Q) qn2[q0 + q0w + r] = nrm;
  }
  __syncthreads();
  const float cn = cn2[c0 + r];
  // C layout of the 32x32 MFMA: register v of lane l holds row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32
  float tv[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int i = 8 * (v >> 2) + 4 * h + (v & 3);
    tv[v] = __builtin_fmaf(-2.0f, acc[v], rown[wave * 32 + i] + cn);
    if (q0 + q0w + i < Q) out[(size_t)(q0 + q0w + i) * Cpad + c0 + r] = tv[v];
  }
  // Two-level selection for many cells (probe_plan2_kernel<.., STREAM>): the minimum of every (query, 128-cell tile) -- the plan
  // takes its threshold from these and reads only the tiles that can hold a candidate.  Padding cells do not count; a NaN
  // makes its tile one that is always read (-inf).
  if (tmin) {
    const float INF = __uint_as_float(0x7f800000u);
    __syncthreads();                       // (rown is reused below)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      float m = (c0 + r < C) ? (tv[v] == tv[v] ? tv[v] : -INF) : INF;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));   // (lanes of one half: the 32 cells of the block)
      if (r == 0) rown[wave * 32 + 8 * (v >> 2) + 4 * h + (v & 3)] = m;
    }
    __syncthreads();
    if (tid < TQ && q0 + tid < Q)
      tmin[(size_t)(q0 + tid) * gx + bx] = fminf(fminf(rown[tid], rown[32 + tid]), fminf(rown[64 + tid], rown[96 + tid]));
  }
}
// ---------------------------------------------------------------------------------------
// The same tile on f16-SPLIT operands (v_mfma_f32_32x32x16_f16, fp32 accumulate): every fp32 input is hi + lo with
// hi = f16(v), lo = f16(v - hi) after a power-of-two scaling into f16's range, and q.c ~ hi.hi + hi.lo + lo.hi -- 57 matrix
// instructions of 32 cycles per 32 x 32 block instead of 160 of 64 cycles.  For MANY cells (the 40 M-row corpus: 13 000) the
// fp32 version is bound by the matrix pipe (8.6 GFLOP at the fp32 MFMA rate = 55 us of the kernel's 100); with 1000 cells it saves
// 3 us alone and 8 us beside the other batches' kernels (the default since round 4; the fp32 tile stays for d % 4 != 0).  Error of the dot product: dropped lo.lo and the f16 rounding of lo
// <= 3 * 2^-22 |q||c|, fp32 accumulation (<= 64 roundings, doubled in case the matrix core truncates) <= 128 u |q||c|:
// together < 0.9e-5 |q||c| -- below the 0.91e-5 (|q| + |c|)^2 the fp32 chain is given in COARSE_EPS's budget (|q||c| <= (|q| + |c|)^2 / 4).
//   coarseH [Cpad / 32][T][hi / lo][64 lanes][8 halves]: lane = cell r + 32 g of the group, its 8 values = dimensions
//   16 t + 8 g .. + 7, scaled by 2^ec (pin time).  The queries are split per tile: LDS [T][hi / lo][64 lanes] with lane = row + 32 g.
// ---------------------------------------------------------------------------------------
typedef _Float16 ch8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void coarse_approx16_body(const float* __restrict__ queries, const ch8v* __restrict__ coarseH, int ec,
                                                     const float* __restrict__ cn2, float* __restrict__ out,
                                                     float* __restrict__ qn2, int Q, int Cpad, int d, const ZeroArgs& z,
                                                     int bx, int by, int gx, int gy, unsigned char* smem,
                                                     float* __restrict__ tmin, int C) {
  typedef float f16v __attribute__((ext_vector_type(16)));
  {
    const int gtid = (by * gx + bx) * 256 + threadIdx.x, gsz = gx * gy * 256;
#pragma unroll
    for (int a = 0; a < 5; ++a)
      for (int i = gtid; i < z.n[a]; i += gsz) z.p[a][i] = 0u;
  }
  constexpr int TQ = COARSE_TQ;
  const int T = (d + 15) >> 4;
  ch8v* Ah = reinterpret_cast<ch8v*>(smem);                     // [T][2][64]
  float* rown = reinterpret_cast<float*>(Ah + (size_t)T * 128);  // [4][32]
  float* qsc = rown + 128;                                       // [32] 2^-(eq + ec) of the tile's queries
  int* qe = reinterpret_cast<int*>(qsc + 32);                    // [32] eq
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q0 = by * TQ, c0 = bx * 128 + wave * 32;
  const ch8v* bp = coarseH + ((size_t)(c0 >> 5) * T) * 128 + lane;
  constexpr int UN = 4;   // k-steps of B operands in flight per wave
  ch8v bh[UN], bl[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) { const int t = u < T ? u : T - 1; bh[u] = bp[(size_t)t * 128]; bl[u] = bp[(size_t)t * 128 + 64]; }
  // norms and scales of the tile's queries: eight threads per row, every load of a thread in flight together (one thread per
  // row, one dependent load after the other, made this prologue 30 us of every workgroup: 134 -> 214 us for the launch)
  {
    const int row = tid >> 3, part = tid & 7;
    const int q = q0 + row < Q ? q0 + row : Q - 1;
    const float4* qv = reinterpret_cast<const float4*>(queries + (size_t)q * d);
    const int d4n = d >> 2;
    constexpr int NL = 10;                 // float4 per thread: 8 x 10 x 4 = 320 dimensions
    float4 v[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) { const int i = part + 8 * u; v[u] = qv[i < d4n ? i : d4n - 1]; }
    float n0 = 0.0f, am = 0.0f;
#pragma unroll
    for (int u = 0; u < NL; ++u)
      if (part + 8 * u < d4n) {
        n0 = __builtin_fmaf(v[u].x, v[u].x, n0); n0 = __builtin_fmaf(v[u].y, v[u].y, n0);
        n0 = __builtin_fmaf(v[u].z, v[u].z, n0); n0 = __builtin_fmaf(v[u].w, v[u].w, n0);
        am = fmaxf(am, fmaxf(fmaxf(__builtin_fabsf(v[u].x), __builtin_fabsf(v[u].y)), fmaxf(__builtin_fabsf(v[u].z), __builtin_fabsf(v[u].w))));
      }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { n0 += __shfl_xor(n0, o, 64); am = fmaxf(am, __shfl_xor(am, o, 64)); }
    // (every thread of the row has the row's sum and maximum: xor-shuffles)
    int e = 0;
    if (am > 0.0f && am < 3e38f) { (void)__builtin_frexpf(am, &e); e = 14 - e; }
    const float sc = __builtin_ldexpf(1.0f, e);     // the scale itself (a power of two: v * 2^e is exact)
    if (part == 0) {
      rown[row] = n0; rown[32 + row] = n0; rown[64 + row] = n0; rown[96 + row] = n0;
      qsc[row] = __builtin_ldexpf(1.0f, -(e + ec));
      reinterpret_cast<float*>(qe)[row] = sc;
      if (bx == 0 && q0 + row < Q) qn2[q0 + row] = n0;
    }
    // the A fragments straight from the registers the norms were computed from (a second pass over the queries -- five more
    // round trips per thread and a barrier -- was a third of the tile's time): float4 i of the row = dimensions 4 i .. 4 i + 3
    // = half (i & 1) of the eight-dimension group g = (i & 3) >> 1 of k-step t = i >> 2
    typedef _Float16 h4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int i = part + 8 * u;
      if (i < T * 4) {
        float4 x = v[u];
        if (i >= d4n) x = float4{0.f, 0.f, 0.f, 0.f};
        const float f[4] = {x.x * sc, x.y * sc, x.z * sc, x.w * sc};
        h4v hi, lo;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const _Float16 hh = (_Float16)f[w];
          hi[w] = hh;
          lo[w] = (_Float16)(f[w] - (float)hh);
        }
        const int t = i >> 2, g = (i & 3) >> 1, l = row + 32 * g;
        reinterpret_cast<h4v*>(&Ah[(size_t)(t * 2 + 0) * 64 + l])[i & 1] = hi;
        reinterpret_cast<h4v*>(&Ah[(size_t)(t * 2 + 1) * 64 + l])[i & 1] = lo;
      }
    }
  }
  __syncthreads();
  f16v acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
  for (int t0 = 0; t0 < T; t0 += UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int t = t0 + u;
      if (t < T) {
        const ch8v ch = bh[u], cl = bl[u];
        const int tn = t + UN < T ? t + UN : T - 1;
        bh[u] = bp[(size_t)tn * 128]; bl[u] = bp[(size_t)tn * 128 + 64];
        const ch8v ah = Ah[(size_t)(t * 2 + 0) * 64 + lane], al = Ah[(size_t)(t * 2 + 1) * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ch, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ch, acc, 0, 0, 0);
      }
    }
  }
  const float cn = cn2[c0 + r];
  float tv[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int i = 8 * (v >> 2) + 4 * h + (v & 3);
    tv[v] = __builtin_fmaf(-2.0f * qsc[i], acc[v], rown[wave * 32 + i] + cn);   // (the scales are powers of two: -2 q.c exactly as accumulated)
    if (q0 + i < Q) out[(size_t)(q0 + i) * Cpad + c0 + r] = tv[v];
  }
  if (tmin) {
    const float INF = __uint_as_float(0x7f800000u);
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      float m = (c0 + r < C) ? (tv[v] == tv[v] ? tv[v] : -INF) : INF;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
      if (r == 0) rown[wave * 32 + 8 * (v >> 2) + 4 * h + (v & 3)] = m;
    }
    __syncthreads();
    if (tid < TQ && q0 + tid < Q)
      tmin[(size_t)(q0 + tid) * gx + bx] = fminf(fminf(rown[tid], rown[32 + tid]), fminf(rown[64 + tid], rown[96 + tid]));
  }
}
// LDS of the f16-split tile: [T][2][64] fragments + row norms + scales
__host__ __device__ inline size_t coarse_approx16_lds(int d) { return (size_t)((d + 15) >> 4) * 128 * 16 + 128 * 4 + 32 * 4 + 32 * 4; }
static __global__ __launch_bounds__(256) void coarse_approx16_kernel(const float* __restrict__ queries, const ch8v* __restrict__ coarseH, int ec,
                                                             const float* __restrict__ cn2, float* __restrict__ out,
                                                             float* __restrict__ qn2, int Q, int Cpad, int d, ZeroArgs z,
                                                             float* __restrict__ tmin, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  coarse_approx16_body(queries, coarseH, ec, cn2, out, qn2, Q, Cpad, d, z, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem, tmin, C);
}

static __global__ __launch_bounds__(256) void coarse_approx_kernel(const float* __restrict__ queries, const float* __restrict__ coarseF,
                                                           const float* __restrict__ cn2, float* __restrict__ out,
                                                           float* __restrict__ qn2, int Q, int Cpad, int d, int dp, ZeroArgs z,
                                                           float* __restrict__ tmin = nullptr, int C = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  coarse_approx_body(queries, coarseF, cn2, out, qn2, Q, Cpad, d, dp, z, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem, tmin, C);
}

// ---------------------------------------------------------------------------------------
// a7 probe plan on the approximate distances: FOUR waves per active query (a single wave per query spends its
// time waiting on its own dependent instructions: 41 us for the batch, tools/lab/ubench_plan).
//   1. wave w takes cells [256 w, 256 w + 256): approximate distances into registers (4 per lane), cells
//      already probed masked;
//   2. tau = the 2W-th smallest of the 64 per-lane minima over all four waves (>= the 2W-th smallest
//      overall); candidates = cells with a <= tau + 2 eps: every cell among the 2W smallest EXACT keys is one
//      of them (at least 2W cells have d <= a + eps <= tau + eps, so such a cell has a - eps <= d <= tau + eps);
//   3. every wave refines ITS candidates with the reference's squareDistance, PLAN2_NCB at a time: lanes <->
//      dimensions for the separately rounded (q_i - c_i)^2 (coalesced centroid rows, all loads of a batch in
//      flight), staged in LDS, then lane <-> candidate for the sequential sum;
//   4. wave 0 keeps the 2W smallest exact keys, orders them by cell id and replays updateTopK: identical to
//      probe_plan_kernel from here on (same outputs, plus the exact distance of every item).
// ---------------------------------------------------------------------------------------
struct Plan2Args {
  PlanArgs p;               // p.dist = the APPROXIMATE distances [Q][Cpad]
  const float* queries;     // [Q][d]
  const float* coarse;      // [C][d]
  const float* qn2;         // [Q] |q|^2 (coarse_approx_kernel)
  float* item_dist;         // [n_active*W] exact coarse distance of every item (the scan's bound on |r|^2)
  int32_t* violations;      // [4]: [2] += refined cells whose d left [a - eps, a + eps], [3] += cells checked (refine_all only)
  float cmax;               // max_j |c_j|, rounded up
  int d;
  int refine_all;           // tests: every unused cell is refined (exhaustive check of the bracket)
  long long* prof;          // NULL, or [queries][16] cycle sums per phase of wave 0 (tools/lab/ubench_plan)
  // STREAM, first round: the minima of the (query, 128-cell tile) blocks (coarse_approx_body) -- NULL: every cell is read twice
  const float* tmin;        // [Q][Cpad / 128]
};

static constexpr int PLAN2_NW = 4;       // waves per query
static constexpr int PLAN2_NCB = 7;      // candidates a wave loads per round: 28 per round and query
static constexpr int PLAN2_PITCH = 308;  // floats per candidate row in LDS (16-byte aligned rows)
static constexpr int PLAN2_PASS = 128;   // candidates per pass (more than one pass only in degenerate cases / the tests' refine-all mode)
// LDS per workgroup: 4 x 7 x 308 x 4 = 34.5 KB + lists: 4 workgroups per CU = a whole 1024-query batch resident.
// The kernel is bound by instruction ISSUE (16 waves per CU, 4 per SIMD, mostly serial work): everything that
// one wave can do for the query is done by wave 0 alone, the others only fetch and square.

__device__ __forceinline__ uint32_t float_order_bits(float f) {   // monotone map float -> u32 (negative values included)
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// STREAM = false: up to COARSE_MAX_CPAD cells, a query's approximate distances stay in registers between the threshold and
// the candidate pass.  STREAM = true (up to COARSE_STREAM_MAX_CPAD cells): the distance row is read twice -- per-lane
// minima first, then the candidates, kept as a BITMAP in LDS (2 KB; the i-th candidate = the i-th set bit, found through
// per-word prefix counts) -- so any number of cells and of candidates fits the same footprint.
// NWP = 4 waves per query: the shortest latency for ONE batch (34.5 KB of LDS, four queries per CU).  NWP = 1: one wave does
// everything, seven candidates per round -- a quarter of the wave slots and of the LDS per query: with several batches in
// flight, when this kernel has to fit into the CUs the other batches' scans leave, a whole batch is resident on a quarter of
// the chip (as merge_refine_kernel's one-wave instantiation).
template <int ABL, bool STREAM = false, int NWP = PLAN2_NW>   // ABL: 0 in production; > 0: timing experiments of tools/lab/ubench_plan (results are wrong)
__global__ __launch_bounds__(64 * NWP, NWP == 1 ? 4 : 4) void probe_plan2_kernel(Plan2Args g) {   // (<= 128 registers: four waves per SIMD)
  const PlanArgs& a = g.p;
  constexpr int NW = NWP, NCB = PLAN2_NCB, RC = NW * NCB;   // RC candidates per round
  constexpr int NV = COARSE_MAX_CPAD / 64 / NW;                  // cells per lane
  __shared__ u64 stage[64];
  __shared__ float mins[NW][64];
  __shared__ int nws[NW];
  __shared__ float thr_s;
  __shared__ uint16_t clw[STREAM ? 1 : NW][STREAM ? 1 : 64 * NV];   // candidate cells of each wave's cell range, ascending
  constexpr int SW = STREAM ? COARSE_STREAM_MAX_CPAD / 32 : 1;
  __shared__ uint32_t cbits[SW];           // STREAM: candidate bitmap over the cells
  __shared__ uint16_t cpre[SW + 1];        // STREAM: candidates before word w
  __shared__ float tms[STREAM ? COARSE_STREAM_MAX_CPAD / 128 : 1];   // STREAM: the query's tile minima
  __shared__ float cdist[PLAN2_PASS];     // exact distances of the pass's candidates
  __shared__ __attribute__((aligned(16))) float sq[RC * PLAN2_PITCH];   // row i % RC: candidate i of the round
  __shared__ __attribute__((aligned(16))) float qs[320];
  const int x = blockIdx.x, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int q = a.active ? a.active[x] : x;
  const int W = a.W, L = 2 * W, d = g.d;
  uint32_t* used = a.used + (size_t)q * a.used_words;
  const float* drow = a.dist + (size_t)q * a.Cpad;
  const float INF = __uint_as_float(0x7f800000u);
  long long pc = g.prof ? clock64() : 0;
  auto tick = [&](int slot) {
    if (g.prof && threadIdx.x == 0) { const long long t = clock64(); g.prof[(size_t)blockIdx.x * 16 + slot] += t - pc; pc = t; }
  };

  // ---- A (all waves): this wave's 256 cells, masked; per-lane minimum; the query into LDS ----
  float av[NV];
  uint32_t uw[NV];
  const int nvt = STREAM ? a.Cpad / (64 * NW) : NV;   // cells per lane of this wave (Cpad is a multiple of 256)
  if constexpr (!STREAM) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int j = (wave * NV + u) * 64 + lane;
      const int jc = j < a.C ? j : a.C - 1;
      av[u] = drow[jc];
      uw[u] = used[jc >> 5];
    }
  } else {
    for (int i = threadIdx.x; i < SW; i += 64 * NW) cbits[i] = 0u;
  }
  const int d4n = d >> 2;   // (d % 4 == 0, d <= 320)
  if constexpr (NW >= 2) {   // (d4n <= 80 <= 128 threads)
    const int tq = (int)threadIdx.x < d4n ? (int)threadIdx.x : d4n - 1;
    const float4 qreg = *reinterpret_cast<const float4*>(g.queries + (size_t)q * d + 4 * tq);
    if ((int)threadIdx.x < d4n) *reinterpret_cast<float4*>(qs + 4 * threadIdx.x) = qreg;
  } else {                   // one wave: two float4 per lane
    const int t0 = lane < d4n ? lane : d4n - 1, t1 = lane + 64 < d4n ? lane + 64 : d4n - 1;
    const float4 q0r = *reinterpret_cast<const float4*>(g.queries + (size_t)q * d + 4 * t0);
    const float4 q1r = *reinterpret_cast<const float4*>(g.queries + (size_t)q * d + 4 * t1);
    if (lane < d4n) *reinterpret_cast<float4*>(qs + 4 * lane) = q0r;
    if (lane + 64 < d4n) *reinterpret_cast<float4*>(qs + 4 * (lane + 64)) = q1r;
  }
  float mn = INF;
  // the masked value of cell j: INF = not a candidate (past the end / already probed), -INF = NaN (always one)
  auto masked = [&](int j) -> float {
    const int jc = j < a.C ? j : a.C - 1;
    const float v = drow[jc];
    const uint32_t w = used[jc >> 5];
    const bool valid = j < a.C && !((w >> (j & 31)) & 1u);
    return !valid ? INF : (v == v ? v : -INF);
  };
  if constexpr (!STREAM) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int j = (wave * NV + u) * 64 + lane;
      const bool valid = j < a.C && !((uw[u] >> (j & 31)) & 1u);
      if (!valid) av[u] = INF;
      if (av[u] == av[u]) mn = fminf(mn, av[u]);
      else av[u] = -INF;                          // NaN (non-finite table entries): always a candidate, never a threshold
    }
  } else if (g.tmin) {
    // two-level: the threshold from the tiles' minima (every one of them is some cell's distance: the 2W-th smallest of them
    // is >= the 2W-th smallest distance), the candidates from the tiles whose minimum can reach it
    const int gx = a.Cpad >> 7;
    for (int i = threadIdx.x; i < gx; i += 64 * NW) {
      const float tm = g.tmin[(size_t)q * gx + i];
      tms[i] = tm;
      if (tm > -INF) mn = fminf(mn, tm);
    }
  } else {
    for (int u0 = 0; u0 < nvt; u0 += 4) {         // (four loads in flight)
      float v4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) v4[t] = u0 + t < nvt ? masked((wave * nvt + u0 + t) * 64 + lane) : INF;
#pragma unroll
      for (int t = 0; t < 4; ++t) if (v4[t] > -INF) mn = fminf(mn, v4[t]);
    }
  }
  mins[wave][lane] = mn;
  tick(0);
  __syncthreads();
  if (ABL == 2) { if (mn == 12345.0f) g.item_dist[0] = mn; return; }
  // ---- B (wave 0): tau = the 2W-th smallest of the 64 per-lane minima; thr = tau + 2 eps ----
  float eps = 0.0f;
  bool finite = true;
  if (wave == 0) {
    const float s = __builtin_sqrtf(g.qn2[q]) * (1.0f + 1e-6f) + g.cmax;
    eps = s * s * COARSE_EPS;
    finite = eps < 1e30f;   // (false for NaN too)
    float thr = INF;
    if (finite && !g.refine_all) {
      float m4 = mins[0][lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) m4 = fminf(m4, mins[w][lane]);
      // (a lone wave issues about one instruction per 10 cycles whether it depends on the previous one or not:
      // what counts in wave 0 is the NUMBER of instructions -- counting ranks with 64 independent compares
      // measured twice the time of this 21-stage sort)
      const uint32_t k0 = wave_sort32(float_order_bits(m4));
      const uint32_t tb = (uint32_t)__shfl((int)k0, L - 1 < 63 ? L - 1 : 63, 64);
      const float tau = __uint_as_float((tb & 0x80000000u) ? (tb & 0x7fffffffu) : ~tb);
      if (tau < 1e30f) thr = (tau + 2.0f * eps) * (1.0f + 1e-6f) + 1e-37f;
    }
    if (lane == 0) thr_s = thr;
  }
  tick(1);
  __syncthreads();
  // ---- C (all waves): this wave's candidates, ascending, into its own list ----
  const float thr = thr_s;
  int n_mine = 0;
  if constexpr (!STREAM) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const bool c = av[u] <= thr && av[u] < INF;
      const u64 mask = __ballot(c);
      if (mask != 0ull) {
        if (c) clw[wave][n_mine + lanes_below(mask)] = (uint16_t)((wave * NV + u) * 64 + lane);
        n_mine += __popcll(mask);
      }
    }
  } else {
    for (int u0 = 0; u0 < nvt; u0 += 4) {
      float v4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // (64 cells = half a tile: read only if the tile's minimum can reach the threshold; -inf / NaN: always)
        const bool rd = u0 + t < nvt && (!g.tmin || !(tms[(wave * nvt + u0 + t) >> 1] > thr));
        v4[t] = rd ? masked((wave * nvt + u0 + t) * 64 + lane) : INF;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool c = v4[t] <= thr && v4[t] < INF;
        const u64 mask = __ballot(c);
        if (mask != 0ull) {
          if (lane == 0) {
            const int w0 = ((wave * nvt + u0 + t) * 64) >> 5;
            cbits[w0] = (uint32_t)mask;
            cbits[w0 + 1] = (uint32_t)(mask >> 32);
          }
          n_mine += __popcll(mask);
        }
      }
    }
  }
  if (lane == 0) nws[wave] = n_mine;
  tick(2);
  __syncthreads();
  if constexpr (STREAM) {   // candidates before every word of the bitmap: wave w scans words [w * wpw, (w + 1) * wpw), two per lane and step
    const int nwords = a.Cpad >> 5, wpw = nwords / NW;
    int before = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) before += (w < wave) ? nws[w] : 0;
    for (int b0 = 0; b0 < wpw; b0 += 128) {
      const int w0 = wave * wpw + b0 + 2 * lane;
      const bool in = b0 + 2 * lane < wpw;
      const int c0 = in ? __popc(cbits[w0]) : 0, c1 = in ? __popc(cbits[w0 + 1]) : 0;
      int inc = c0 + c1;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
      }
      if (in) {
        cpre[w0] = (uint16_t)(before + inc - c0 - c1);
        cpre[w0 + 1] = (uint16_t)(before + inc - c1);
      }
      before += __shfl(inc, 63, 64);
    }
    if (threadIdx.x == 0) { int tot = 0; for (int w = 0; w < NW; ++w) tot += nws[w]; cpre[nwords] = (uint16_t)tot; }
    __syncthreads();
  }
  // the query's candidate list = the four lists one after the other: candidate i lives in list wi(i) at i - off[wi]
  int off[NW + 1];
  off[0] = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) off[w + 1] = off[w] + nws[w];
  const int n_all = off[NW];
  auto cell_of = [&](int i) -> int {
    if constexpr (STREAM) {   // the i-th set bit of the bitmap: the word by bisection over the prefix counts, then the bit
      int lo = 0, hi = a.Cpad >> 5;          // invariant: cpre[lo] <= i < cpre[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)cpre[mid] <= i) lo = mid; else hi = mid;
      }
      uint32_t x = cbits[lo];
      int r = i - (int)cpre[lo], pos = 0;
#pragma unroll
      for (int sh = 16; sh > 0; sh >>= 1) {
        const uint32_t lowmask = (1u << sh) - 1u;
        const int c = __popc(x & lowmask);
        if (r >= c) { r -= c; x >>= sh; pos += sh; } else x &= lowmask;
      }
      return (lo << 5) + pos;
    }
    int w = 0;
#pragma unroll
    for (int t = 1; t < NW; ++t) w += i >= off[t] ? 1 : 0;
    int o = off[0];
#pragma unroll
    for (int t = 1; t < NW; ++t) o = i >= off[t] ? off[t] : o;
    return (int)clw[w][i - o];
  };
  if (ABL == 3) { if (n_all == 12345) g.item_dist[0] = thr; return; }

  // flat view of a wave's NCB candidate rows as 16-byte elements: element 64 k + lane = float4 fc[k] of slot ft[k]
  constexpr int NK = (NCB * 80 + 63) / 64;
  int ft[NK], fc[NK];
  {
    int t = 0, c = lane;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      while (c >= d4n) { c -= d4n; ++t; }
      ft[k] = t; fc[k] = c;
      c += 64;
    }
  }
  const u64 limit = (u64)__float_as_uint(a.cell_limit) << 32;
  WaveSelect<1> sel;   // (more candidates than lanes only)
  if (wave == 0 && n_all > 64) sel.init(stage, limit, L);
  int viol = 0;
  for (int p0 = 0; p0 < n_all; p0 += PLAN2_PASS) {
    const int np = n_all - p0 < PLAN2_PASS ? n_all - p0 : PLAN2_PASS;
    for (int r0 = 0; r0 < np; r0 += RC) {
      // ---- D (all waves): candidates r0 + wave * NCB + t (t < NCB) of the pass: rows fetched, (q_i - c_i)^2 into LDS ----
      int nb = np - r0 - wave * NCB;
      nb = nb < 0 ? 0 : (nb > NCB ? NCB : nb);
      if (nb > 0) {
        // (NO conditions around the loads: a guarded load becomes a basic block of its own with a full wait at
        // its end; elements past the wave's last candidate re-read its last element and are not stored)
        const int nf = nb * d4n;
        const int mycell = cell_of(p0 + r0 + wave * NCB + (lane < nb ? lane : nb - 1));   // lane t: candidate slot t
        float4 cv[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const bool in = 64 * k + lane < nf;
          const int t = in ? ft[k] : nb - 1, c4 = in ? fc[k] : d4n - 1;
          const int cell = __shfl(mycell, t, 64);
          cv[k] = *reinterpret_cast<const float4*>(g.coarse + (size_t)cell * d + 4 * c4);
        }
        float* sqw = sq + wave * NCB * PLAN2_PITCH;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const float4 qq = *reinterpret_cast<const float4*>(qs + 4 * fc[k]);
          float4 pr;                                      // index_utils.c:500-508: sub, mul, add rounded one by one
          pr.x = qq.x - cv[k].x; pr.y = qq.y - cv[k].y; pr.z = qq.z - cv[k].z; pr.w = qq.w - cv[k].w;
          pr.x = pr.x * pr.x; pr.y = pr.y * pr.y; pr.z = pr.z * pr.z; pr.w = pr.w * pr.w;
          if (64 * k + lane < nf) *reinterpret_cast<float4*>(sqw + ft[k] * PLAN2_PITCH + 4 * fc[k]) = pr;
        }
      }
      tick(3);
      __syncthreads();
      // ---- E (wave 0): lane <-> candidate, the sequential sum ----
      if (wave == 0) {
        const int nr = np - r0 < RC ? np - r0 : RC;
        if (lane < nr) {
          const int cell = cell_of(p0 + r0 + lane);
          const float ap = drow[cell];   // the approximate value again (bracket check; L1 / L2 hit, used after the sum)
          float acc = 0.0f;
          const float* row = sq + lane * PLAN2_PITCH;
          int e = 0;
          // (15 LDS reads in flight per block: a read per step would expose its latency 75 times; 25 per block cost
          // the kernel 171 registers, i.e. two resident workgroups per CU instead of four)
          for (; e + 60 <= d; e += 60) {
            float4 v[15];
#pragma unroll
            for (int t = 0; t < 15; ++t) v[t] = *reinterpret_cast<const float4*>(row + e + 4 * t);
#pragma unroll
            for (int t = 0; t < 15; ++t) { acc = acc + v[t].x; acc = acc + v[t].y; acc = acc + v[t].z; acc = acc + v[t].w; }
          }
          for (; e < d; ++e) acc = acc + row[e];
          cdist[r0 + lane] = acc;
          if (finite && ap == ap && !(__builtin_fabsf(acc - ap) <= eps)) ++viol;
        }
      }
      tick(4);
      if (r0 + RC < np || p0 + PLAN2_PASS < n_all) __syncthreads();   // (sq is rewritten by the next round)
    }
    if (wave == 0 && n_all > 64) {   // streaming selection of the 2W smallest keys
      __builtin_amdgcn_wave_barrier();
      for (int c0 = 0; c0 < np; c0 += 64) {
        const bool v = c0 + lane < np;
        const u64 key = v ? make_key(cdist[c0 + lane], (uint32_t)cell_of(p0 + c0 + (v ? lane : 0))) : KEY_INF;
        sel.push(key, v);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (wave != 0) return;
  if (g.violations && viol) atomicAdd(g.violations + 2, viol);
  if (ABL == 4) { if (n_all == 12345) g.item_dist[0] = 1.0f; return; }
  if (g.violations) {
#ifdef FREDDY_PLAN2_STATS   // (tools/lab/ubench_plan: candidates per query)
    if (lane == 0) atomicAdd(g.violations + 3, n_all);
#else
    if (g.refine_all && lane == 0) atomicAdd(g.violations + 3, n_all);
#endif
  }
  // ---- F (wave 0): lane i = slot i of the W-entry list (freddy.c:266-283: cells offered in ascending id, updateTopK) ----
  float d_slot = a.cell_limit;
  int32_t c_slot = -1;
  bool done = false;
  u64 byc = KEY_INF;   // candidates in cell order: (cell << 32) | distance bits
  if (n_all <= 64) {
    // The candidates ARE in ascending cell order already (lane i = candidate i), and replaying a superset of the
    // 2W smallest keys gives the same list as replaying exactly those.  Without equal distances among the W + 1
    // smallest the list is simply the W smallest below the limit, ascending: one sort instead of the replay.
    __builtin_amdgcn_wave_barrier();
    const bool v = lane < n_all;
    const uint32_t db = v ? __float_as_uint(cdist[lane]) : 0xffffffffu;
    const uint32_t cb = v ? (uint32_t)cell_of(v ? lane : 0) : 0xffffffffu;
    byc = v ? (((u64)cb << 32) | (u64)db) : KEY_INF;
    const u64 byd = wave_sort64(v ? (((u64)db << 32) | (u64)cb) : KEY_INF);
    const uint32_t dn = (uint32_t)(__shfl_down(byd, 1, 64) >> 32);
    const bool tie = lane < W && lane + 1 < n_all && (uint32_t)(byd >> 32) == dn;
    if (__ballot(tie) == 0ull) {
      if (lane < W && byd < limit) { d_slot = __uint_as_float((uint32_t)(byd >> 32)); c_slot = (int32_t)(uint32_t)byd; }
      done = true;
    }
  } else {
    sel.finish();
    byc = (sel.acc[0] == KEY_INF || lane >= L) ? KEY_INF : ((sel.acc[0] << 32) | (sel.acc[0] >> 32));
    byc = wave_sort64(byc);
  }
  if (!done) wave_list_replay(d_slot, c_slot, W, byc, n_all <= 64 ? n_all : L, [](uint32_t hi) { return (int32_t)hi; });
  if (ABL == 5) { if (c_slot == 12345) g.item_dist[0] = d_slot; return; }
  tick(5);
  const bool have = lane < W && c_slot >= 0;
  int rows = have ? (a.list_off[c_slot + 1] - a.list_off[c_slot]) : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rows += __shfl_xor(rows, o, 64);
  if (have) {
    atomicOr(used + (c_slot >> 5), 1u << (c_slot & 31));
    if (a.cell_count) {
      const int at = atomicAdd(a.cell_count + c_slot, 1);
      a.cell_items[(size_t)c_slot * a.cell_cap + at] = x * W + lane;
    }
  }
  const bool any_cell = __ballot(have) != 0ull;
  if (lane == 0) a.round_rows[x] = any_cell ? rows : -1;   // -1: every cell already used, the query retires
  if (lane < W) {
    a.item_cell[(size_t)x * W + lane] = c_slot;
    a.item_query[(size_t)x * W + lane] = q;
    g.item_dist[(size_t)x * W + lane] = d_slot;
  }
  tick(6);
}

}  // namespace freddy
