// exact2.h -- exact brute-force kNN as FILTER + REFINE on the matrix cores.
//
// exact.h evaluates the reference's similarity chain (core_functions.c:67-81: scalar += v1[i] * v2[i], one rounded
// multiply and one rounded add per dimension) for EVERY (row, query) pair: 2 N d Q separately rounded VALU operations
// -- 4.3 ms for 64 queries over 3 M x 300 rows, while the table itself (3.6 GB) crosses HBM in 0.45 ms.  Only the k
// largest similarities of a query matter.  Here
//
//   a[r][q] ~ x_r . q      f16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate) on SPLIT operands: every fp32 input is
//                          hi + lo with hi = f16(v), lo = f16(v - hi) (22 significand bits); a = hi.hi + hi.lo + lo.hi
//
// ranks the rows, with a proven bracket |a - s| <= eps(q) around the reference's binary32 similarity s, and the
// reference's chain is evaluated only for the rows whose bracket can reach the k-th largest similarity:
//
//   1. SAMPLE: a for the first S rows (S = 32 768) -> tau = the k-th largest a of the sample.  The k-th largest EXACT
//      similarity of the whole table is >= the k-th largest exact one of the sample >= tau - eps.
//   2. FILTER: a for every row; candidates = rows with a >= tau - 2 eps -- a superset of every row whose exact
//      similarity reaches the k-th largest one (such a row has s >= tau - eps, hence a >= tau - 2 eps), ties included.
//   3. REFINE: the reference's chain (exact.h's arithmetic, bit for bit) for the candidates, ordering by
//      (similarity DESC, id ASC), the k first: the list exact.h produces.
//
// The approximate value never reaches a result.  Operands are scaled by powers of two (exact) so that the largest
// |element| of the table / of a query lands near 2^14: f16 holds hi with 11 bits and lo with 11 more (subnormal lo:
// absolute error 2^-25 against a maximum of 2^14).
//
// The bracket (u = 2^-24; |.| Euclidean norms; |x_r| <= X = the largest row norm of the table, rounded up):
//   reference chain      |s - x.q| <= ((1+u)^(d+1) - 1) sum|x_i q_i|  <= 1.80e-5 |x||q|      (d <= 300)
//   dropped lo.lo        <= 2^-22 sum|x_i q_i|                        <= 0.03e-5 |x||q|
//   lo rounded to f16    <= 2 * 2^-22 sum|x_i q_i| (+ 2^-39-relative subnormal terms)  <= 0.05e-5 |x||q|
//   MFMA accumulation    fp32, <= 64 roundings (19 steps x 3 products + in-step adds), doubled in case the matrix core
//                        TRUNCATES instead of rounding: 128 u sum|terms|            <= 0.77e-5 |x||q|
//   sum < 2.7e-5 |x||q|;   eps(q) = EXF_EPS X |q| with EXF_EPS = 4e-5 (d <= 512: the chain term grows to 3.1e-5 -> 6e-5).
// Every refined row has both numbers in hand: the refine kernel counts rows whose similarity left [a - eps, a + eps]
// (freddy_gpu_filter_bound_violations on the vector handle; the tests refine EVERY row, option exact_refine_all).
//
// Roofline: the table once per pass of <= 64 queries: N d 4 bytes from HBM (3.6 GB: beyond the 256 MiB Infinity Cache).
// MFMA work per pass: 3 x 2 N d 64 flops = 0.35 PFLOP-equivalents at ~2 PF/s = 0.17 ms -- below the 0.45 ms of HBM time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exact.h"
#include "wave_topk.h"

namespace freddy {

static constexpr int EXF_QT = 64;          // queries per pass (two 32-column MFMA tiles)
static constexpr int EXF_WG = 512;         // 8 waves share one LDS image of the query fragments
static constexpr int EXF_SAMPLE = 32768;   // rows of the threshold sample
static constexpr int EXF_PF = 4;           // k-steps of row operands a wave keeps in flight (8 x 1 KB)
static constexpr int EXF_TARGET_EXP = 14;  // scaled maxima land in [2^13, 2^14]

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16acc __attribute__((ext_vector_type(16)));

__host__ __device__ __forceinline__ float exf_eps_factor(int d) { return d <= 300 ? 4e-5f : 6e-5f; }

// power-of-two exponent e with |v| * 2^e in [2^(T-1), 2^T] (v > 0, finite); 0 for v == 0
__host__ __device__ __forceinline__ int exf_scale_exp(float vmax) {
  if (!(vmax > 0.0f) || !(vmax < 3e38f)) return 0;
  int e;
#if defined(__HIP_DEVICE_COMPILE__)
  (void)__builtin_frexpf(vmax, &e);
#else
  (void)frexpf(vmax, &e);
#endif
  return EXF_TARGET_EXP - e;   // vmax = f * 2^e, f in [0.5, 1)
}

// ---- pin time: the table's largest |element| and largest row norm (both rounded up) ------------------------
// out[0] = max |x_i| (bits, atomicMax on non-negative floats), out[1] = max row norm^2 (fp32 fma chain, bits); out[2] |= 1 if non-finite
__global__ __launch_bounds__(256) void exf_table_stats_kernel(const float* __restrict__ rows, int64_t n_rows, int d, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float amax = 0.0f, nmax = 0.0f;
  bool bad = false;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n_rows; r += (int64_t)gridDim.x * 4) {
    const float* x = rows + (size_t)r * d;
    float n2 = 0.0f;
    for (int i = lane; i < d; i += 64) {
      const float v = x[i];
      if (!(__builtin_fabsf(v) < 3e38f)) bad = true;
      amax = fmaxf(amax, __builtin_fabsf(v));
      n2 = __builtin_fmaf(v, v, n2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o, 64);
    nmax = fmaxf(nmax, n2);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if (lane == 0) {
    atomicMax(out + 0, __float_as_uint(amax));
    atomicMax(out + 1, __float_as_uint(nmax));
  }
  if (__ballot(bad) != 0ull && lane == 0) atomicOr(out + 2, 1u);
}

// ---- pin time: the table in A-fragment order, split and scaled once ------------------------------------------------
// xf[strip][k-step t][hi / lo][lane] (8 halves = 16 bytes each): lane = row i + 32 g of the strip, its 8 values = dimensions
// 16 t + 8 g .. + 7 of row 32 strip + i, scaled by 2^ex, hi = f16(v), lo = f16(v - hi); zero beyond d and beyond the last
// row.  A wave's operand load for one k-step is two fully coalesced 1 KB reads (row-major rows, 1200 bytes apart, cost
// 64 cache-line lookups per load instruction: 4.7 instead of 6.0 TB/s -- measured with lane-linear addresses), and the
// conversion leaves the kernel.  Same bytes as the fp32 rows.
__global__ __launch_bounds__(256) void exf_layout_kernel(const float* __restrict__ rows, int64_t n_rows, int d, int T, int ex,
                                                        int64_t strip0, int64_t n_strips, h8v* __restrict__ xf) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (strip - strip0, t, lane)
  if (i >= n_strips * T * 64) return;
  const int lane = (int)(i & 63);
  const int64_t st = i >> 6;
  const int t = (int)(st % T);
  const int64_t strip = strip0 + st / T;
  const int64_t row = strip * 32 + (lane & 31);
  const int g = lane >> 5;
  h8v hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int dim = 16 * t + 8 * g + e;
    const float v = (row < n_rows && dim < d) ? __builtin_ldexpf(rows[(size_t)row * d + dim], ex) : 0.0f;
    const _Float16 h = (_Float16)v;
    hi[e] = h;
    lo[e] = (_Float16)(v - (float)h);
  }
  xf[((size_t)(strip * T + t) * 2 + 0) * 64 + lane] = hi;
  xf[((size_t)(strip * T + t) * 2 + 1) * 64 + lane] = lo;
}

// ---- per pass: the queries' norms, scales and f16 fragments ---------------------------------------------------
// Fragment order for v_mfma_f32_32x32x16_f16's B operand: qfrag[tile n][k-step t][hi / lo][lane][8 halves], lane =
// column j + 32 g, the lane's 8 values = dimensions 16 t + 8 g .. + 7 of query 32 n + j (zero beyond d / beyond the pass's
// queries).  The A operand (rows) uses the same (g, element) -> dimension map, which is all the instruction requires.
struct ExfPrepArgs {
  const float* queries;   // [nq][d] of this pass
  int nq, d, T;           // T = k-steps = ceil(d / 16)
  float xmax_norm;        // X
  int ex;                 // the table's scale exponent
  float eps_factor;
  h8v* qfrag;             // [2][T][2][64]
  float* qeps;            // [EXF_QT] eps(q)
  float* qunscale;        // [EXF_QT] 2^-(eq + ex)
  int32_t* qbad;          // [1] |= 1 if a query is not finite
  float* copy_out;        // NULL, or [nq][d]: `queries` is mapped host memory -- read once here, the refine kernel reads this copy
};
__global__ __launch_bounds__(256) void exf_prep_kernel(ExfPrepArgs a) {
  const int q = blockIdx.x;        // 0 .. EXF_QT-1
  const int tid = threadIdx.x;
  __shared__ float red[8];
  __shared__ int eq_s;
  const bool live = q < a.nq;
  const float* qv = a.queries + (size_t)(live ? q : 0) * a.d;
  float amax = 0.0f, n2 = 0.0f;
  bool bad = false;
  if (live)
    for (int i = tid; i < a.d; i += 256) {
      const float v = qv[i];
      if (a.copy_out) a.copy_out[(size_t)q * a.d + i] = v;
      if (!(__builtin_fabsf(v) < 3e38f)) bad = true;
      amax = fmaxf(amax, __builtin_fabsf(v));
      n2 = __builtin_fmaf(v, v, n2);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, o, 64)); n2 += __shfl_xor(n2, o, 64); }
  if ((tid & 63) == 0) { red[tid >> 6] = amax; red[4 + (tid >> 6)] = n2; }
  if (__ballot(bad) != 0ull && (tid & 63) == 0) atomicOr(a.qbad, 1);
  __syncthreads();
  if (tid == 0) {
    const float am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float nn = (red[4] + red[5]) + (red[6] + red[7]);
    const int eq = exf_scale_exp(am);
    eq_s = eq;
    const float qn = __builtin_sqrtf(nn) * (1.0f + 1e-5f);
    a.qeps[q] = live ? a.eps_factor * a.xmax_norm * qn * (1.0f + 1e-6f) + 1e-37f : 0.0f;
    a.qunscale[q] = __builtin_ldexpf(1.0f, -(eq + a.ex));
  }
  __syncthreads();
  const int eq = eq_s;
  if (a.copy_out && live) qv = a.copy_out + (size_t)q * a.d;   // (this workgroup's own stores, behind the barrier)
  const int n = q >> 5, j = q & 31;
  for (int i = tid; i < a.T * 2; i += 256) {   // (k-step t, lane group g)
    const int t = i >> 1, g = i & 1;
    h8v hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int dim = 16 * t + 8 * g + e;
      const float v = (live && dim < a.d) ? __builtin_ldexpf(qv[dim], eq) : 0.0f;
      const _Float16 h = (_Float16)v;
      hi[e] = h;
      lo[e] = (_Float16)(v - (float)h);
    }
    a.qfrag[((size_t)(n * a.T + t) * 2 + 0) * 64 + j + 32 * g] = hi;
    a.qfrag[((size_t)(n * a.T + t) * 2 + 1) * 64 + j + 32 * g] = lo;
  }
}

// ---- the MFMA pass -----------------------------------------------------------------------------------------------
struct ExfArgs {
  const h8v* xf;            // the table in fragment order (exf_layout_kernel)
  int64_t n_rows;           // rows [0, n_rows) of this launch
  int64_t strip_stride;     // SAMPLE mode: strip i of the launch is strip i * strip_stride of the table (a strided sample: on an id-ordered
                            // or clustered table the first rows are no sample of it); 1 otherwise
  int T;
  const h8v* qfrag;         // [NT][T][2][64]
  const float* qunscale;    // [EXF_QT]
  // SAMPLE mode: approximate similarities of every row of the launch
  float* sample_out;        // [EXF_QT][n_rows] or NULL
  // FILTER mode
  const float* thr;         // [EXF_QT] scaled thresholds: candidates have acc >= thr
  int32_t* cand_cnt;        // [EXF_QT]
  uint2* cand;              // [EXF_QT][cap] (row, bits of the approximate similarity)
  int cap;
};

template <int NT, bool SAMPLE>
__global__ __launch_bounds__(EXF_WG, 2) void exf_filter_kernel(ExfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h8v* qf = reinterpret_cast<h8v*>(smem);                  // [NT][T][2][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T;
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.qfrag);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const int n16 = NT * T * 2 * 64;
    for (int i = tid; i < n16; i += EXF_WG) dst[i] = src[i];
  }
  __syncthreads();
  const int i_row = lane & 31, g = lane >> 5;
  float thr[NT], unsc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    thr[n] = SAMPLE ? 0.0f : a.thr[32 * n + i_row];
    unsc[n] = a.qunscale[32 * n + i_row];
  }
  const int64_t n_strips = (a.n_rows + 31) >> 5;
  for (int64_t strip = (int64_t)blockIdx.x * (EXF_WG / 64) + wave; strip < n_strips; strip += (int64_t)gridDim.x * (EXF_WG / 64)) {
    const h8v* xs = a.xf + (size_t)(SAMPLE ? strip * a.strip_stride : strip) * T * 128 + lane;
    f16acc acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[n][v] = 0.0f;
    // the strip's operands: [t][hi / lo][64 lanes], EXF_PF k-steps in flight
    h8v ring[EXF_PF][2];
    auto issue = [&](int slot, int t) {
      ring[slot][0] = xs[(size_t)(2 * t) * 64];
      ring[slot][1] = xs[(size_t)(2 * t + 1) * 64];
    };
#pragma unroll
    for (int p = 0; p < EXF_PF; ++p) if (p < T) issue(p, p);
    for (int t0 = 0; t0 < T; t0 += EXF_PF) {
#pragma unroll
      for (int p = 0; p < EXF_PF; ++p) {
        const int t = t0 + p;
        if (t < T) {
          const h8v hi = ring[p][0], lo = ring[p][1];
          if (t + EXF_PF < T) issue(p, t + EXF_PF);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const h8v bh = qf[((size_t)(n * T + t) * 2 + 0) * 64 + lane];
            const h8v bl = qf[((size_t)(n * T + t) * 2 + 1) * 64 + lane];
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, bh, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, bl, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, bh, acc[n], 0, 0, 0);
          }
        }
      }
    }
    // C layout: register v of lane l = row (v & 3) + 8 (v >> 2) + 4 (l >> 5) of the strip, column l & 31
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int q = 32 * n + i_row;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int64_t r = strip * 32 + (v & 3) + 8 * (v >> 2) + 4 * g;
        if constexpr (SAMPLE) {
          if (r < a.n_rows) a.sample_out[(size_t)q * a.n_rows + r] = acc[n][v] * unsc[n];
        } else {
          if (!(acc[n][v] < thr[n]) && r < a.n_rows) {     // (a NaN passes: the refine stage decides)
            const int slot = atomicAdd(a.cand_cnt + q, 1);
            if (slot < a.cap) a.cand[(size_t)q * a.cap + slot] = uint2{(uint32_t)r, __float_as_uint(acc[n][v] * unsc[n])};
          }
        }
      }
    }
  }
}

// ---- tau = the k-th largest approximate similarity of the sample; thr = (tau - 2 eps) in the scaled domain --------
struct ExfThrArgs {
  const float* sample;     // [EXF_QT][n_sample]
  int n_sample, nq, k;
  const float* qeps;
  const float* qunscale;
  float* thr;              // [EXF_QT]
  int refine_all;          // tests: every row is a candidate
  int32_t* cand_cnt;       // [EXF_QT] <- 0 (the filter pass behind this launch counts from there)
};
static constexpr int EXF_TW = 16;   // waves per query of the threshold and refine kernels (one query: the launch is one workgroup)
__global__ __launch_bounds__(64 * EXF_TW) void exf_threshold_kernel(ExfThrArgs a) {
  __shared__ u64 stage[EXF_TW][64];
  __shared__ u64 lists[EXF_TW][64];
  __shared__ int nan_s[EXF_TW];
  const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float NEG_INF = -__builtin_huge_valf();
  if (threadIdx.x == 0) a.cand_cnt[q] = 0;
  if (q >= a.nq) { if (threadIdx.x == 0) a.thr[q] = __builtin_huge_valf(); return; }   // padding columns never pass
  // the k-th best of the 1024 lanes' best sample rows: every one of them is some row's approximate similarity, so their k-th
  // best is not above the sample's k-th best -- a valid threshold, and within a row or two of it (two of a query's k best rows
  // in one lane's 32: k^2 / 2048) -- for one selection step per wave instead of 32 (30 -> 9 us, a twentieth of a one-query call)
  const float* s = a.sample + (size_t)q * a.n_sample;
  bool nan = false;
  u64 best = KEY_INF;
  for (int base0 = wave * 64; base0 < a.n_sample; base0 += 64 * EXF_TW * 8) {   // (eight loads in flight)
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = base0 + u * 64 * EXF_TW + lane; x[u] = i < a.n_sample ? s[i] : NEG_INF; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base0 + u * 64 * EXF_TW + lane;
      if (i < a.n_sample) {
        if (!(x[u] == x[u])) nan = true;
        else { const u64 key = sim_key(x[u], (uint32_t)i); best = key < best ? key : best; }
      }
    }
  }
  WaveSelect<1> sel;
  sel.init(stage[wave], KEY_INF, a.k < 64 ? a.k : 64);
  sel.push(best, best != KEY_INF);
  sel.finish();
  lists[wave][lane] = sel.acc[0];
  const bool wave_nan = __ballot(nan) != 0ull;
  if (lane == 0) nan_s[wave] = wave_nan ? 1 : 0;
  __syncthreads();
  if (wave != 0) return;
  bool any_nan = false;
#pragma unroll
  for (int w = 0; w < EXF_TW; ++w) any_nan = any_nan || nan_s[w] != 0;
  WaveSelect<1> fin;
  fin.init(stage[0], KEY_INF, a.k < 64 ? a.k : 64);
  for (int w = 0; w < EXF_TW; ++w) { const u64 key = lists[w][lane]; fin.push(key, key != KEY_INF); }
  fin.finish();
  const u64 kth = wave_topk_at<1>(fin.acc, a.k - 1);
  if (lane == 0) {
    float thr = NEG_INF;   // fewer than k sample rows / NaNs around / refine-all: every row is a candidate
    if (kth != KEY_INF && !any_nan && !a.refine_all) {
      const float tau = key_sim(kth);
      const float t = tau - 2.0f * a.qeps[q];
      const float tt = t - __builtin_fabsf(t) * 2.4e-7f - 1e-37f;    // (rounded down)
      thr = tt / a.qunscale[q];                                       // exact: a power of two
      thr = thr - __builtin_fabsf(thr) * 2.4e-7f;                     // (scaled products are compared: once more down)
      if (!(thr == thr)) thr = NEG_INF;
    }
    a.thr[q] = thr;
  }
}

// ---- the reference's chain for the candidates, and the query's list -----------------------------------------------
struct ExfRefineArgs {
  const float* rows;        // [N][d]
  const float* queries;     // [nq][d]
  const uint2* cand;        // [EXF_QT][cap]
  const int32_t* cand_cnt;  // [EXF_QT]
  const float* qeps;
  int32_t* viol;            // [4]: [0] += rows whose similarity left the bracket, [1] += rows refined (refine_all only), [3] |= overflow (a query had more candidates than the buffer holds)
  int cap, d, L, count_checked;
  // the query's list straight from this workgroup (its sixteen waves' lists meet in LDS: no merge launch), and -- from the call's
  // LAST workgroup to finish, nobody waits -- the call's two verdict words behind the lists; both are cleared for the next call
  const int32_t* ids;       // [N]
  int32_t* out_ids;         // [Q][k] (mapped host memory), this pass's slice
  float* out_sim;
  int k;
  int32_t* arrived;         // [1] workgroups of this CALL that have finished (zero between calls)
  int total_wgs;            // of the call
  int32_t* qbad;            // the other verdict word (exf_prep_kernel)
  int32_t* flags_out;       // [2]
};
static inline size_t exf_refine_lds(int d, int V) {
  return (((size_t)d * 4 + 15) & ~(size_t)15) + (size_t)EXF_TW * 64 * sizeof(u64) + (size_t)EXF_TW * 64 * V * sizeof(u64);
}
template <int V>
__global__ __launch_bounds__(64 * EXF_TW) void exf_refine_kernel(ExfRefineArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* qs = reinterpret_cast<float*>(smem);                                                     // [d]
  u64* stage = reinterpret_cast<u64*>(smem + (((size_t)a.d * 4 + 15) & ~(size_t)15));             // [EXF_TW][64]
  u64* lists = stage + EXF_TW * 64;                                                               // [EXF_TW][64 * V]
  const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = a.d;
  for (int i = threadIdx.x; i < d; i += 64 * EXF_TW) qs[i] = a.queries[(size_t)q * d + i];
  __syncthreads();
  int cnt = a.cand_cnt[q];
  if (cnt > a.cap) { if (threadIdx.x == 0) atomicOr(a.viol + 3, 1); cnt = a.cap; }
  const float eps = a.qeps[q];
  WaveSelect<V> sel;
  sel.init(stage + wave * 64, KEY_INF, a.L);
  int viol = 0;
  for (int base = wave * 64; base < cnt; base += 64 * EXF_TW) {
    const bool v = base + lane < cnt;
    const uint2 c = a.cand[(size_t)q * a.cap + (v ? base + lane : 0)];
    const float4* x = reinterpret_cast<const float4*>(a.rows + (size_t)c.x * d);
    float acc = 0.0f;
    int i = 0;
    for (; i + 32 <= d; i += 32) {      // core_functions.c:77: scalar += v1[i] * v2[i], i ascending, each operation rounded
      float4 xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = x[(i >> 2) + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 qv = *reinterpret_cast<const float4*>(qs + i + 4 * u);
        acc = acc + qv.x * xv[u].x; acc = acc + qv.y * xv[u].y; acc = acc + qv.z * xv[u].z; acc = acc + qv.w * xv[u].w;
      }
    }
    for (; i < d; ++i) acc = acc + qs[i] * reinterpret_cast<const float*>(x)[i];
    const float ap = __uint_as_float(c.y);
    if (v && ap == ap && acc == acc && !(__builtin_fabsf(acc - ap) <= eps)) ++viol;
    sel.push(sim_key(acc, c.x), v);
  }
  sel.finish();
#pragma unroll
  for (int v = 0; v < V; ++v) lists[(size_t)wave * 64 * V + v * 64 + lane] = sel.acc[v];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) viol += __shfl_xor(viol, o, 64);
  if (lane == 0 && viol) atomicAdd(a.viol + 0, viol);
  if (a.count_checked && threadIdx.x == 0) atomicAdd(a.viol + 1, cnt);
  __syncthreads();
  if (wave != 0) return;
  for (int w = 1; w < EXF_TW; ++w)
    for (int v = 0; v < V; ++v) {
      const u64 key = lists[(size_t)w * 64 * V + v * 64 + lane];
      if (__ballot(key != KEY_INF) == 0ull) break;   // ascending: the rest of this list is empty too
      wave_topk_absorb_sorted<V>(sel.acc, key);
    }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int r = v * 64 + lane;
    if (r < a.k) {
      const u64 key = sel.acc[v];
      a.out_ids[(size_t)q * a.k + r] = (key == KEY_INF) ? -1 : a.ids[key_pos(key)];
      a.out_sim[(size_t)q * a.k + r] = (key == KEY_INF) ? -__builtin_huge_valf() : key_sim(key);
    }
  }
  // the call's last workgroup hands the verdict over (every workgroup's overflow flag is in memory before its ticket)
  __threadfence();
  int ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(a.arrived, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  if (ticket == a.total_wgs - 1 && lane == 0) {
    a.flags_out[0] = __hip_atomic_load(a.viol + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.flags_out[1] = __hip_atomic_load(a.qbad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.viol[3] = 0; *a.qbad = 0;
    __hip_atomic_store(a.arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace freddy
