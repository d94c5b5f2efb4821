// coarse.h -- coarse-cell selection (a6 / a7) as FILTER + REFINE, like the scan of fused4.h.
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
//   dot product    fmaf chain of <= 304 terms (any order):  <= 304 u |q||c|, doubled by the factor 2,
//                  |q||c| <= (|q| + |c|)^2 / 4                          <= 0.91e-5 (|q| + |c|)^2
//   |q|^2, |c|^2   fp32 fma sums of 8 / 38 partial terms, one rounding  <= 0.3e-5  (|q| + |c|)^2
//   final add/fma  2 u (|q| + |c|)^2
//   sum < 3.1e-5 (|q| + |c|)^2;   eps(q) = COARSE_EPS (|q| + max_j |c_j|)^2 with COARSE_EPS = 1.2e-4 (4x margin,
//   and it would still hold if the matrix core TRUNCATED every accumulation step instead of rounding it).
// Every refined cell has both numbers in hand: the kernel counts the cells whose d left [a - eps, a + eps]
// (freddy_gpu_filter_bound_violations; the tests also run with EVERY cell refined).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "wave_topk.h"

namespace freddy {

static constexpr float COARSE_EPS = 1.2e-4f;
static constexpr int COARSE_DP_ALIGN = 8;     // the padded dimension count is a multiple of 8 (two float4 per MFMA quad)
static constexpr int COARSE_MAX_CPAD = 1024;  // the plan keeps a query's approximate distances in registers: 16 per lane

// ---------------------------------------------------------------------------------------
// a[q][j] for a 64-query x 64-cell tile per workgroup; wave w owns the 32 x 32 quadrant (w >> 1, w & 1).
// No LDS staging: a lane's operands for FOUR consecutive MFMAs are one 16-byte load from its query row and
// one from its centroid row (the k index of an MFMA step may be any permutation as long as A and B agree:
// step t of iteration i pairs elements 8 i + t (lanes 0-31) and 8 i + 4 + t (lanes 32-63)).
//   queries [Q][d] (d need not be a multiple of 8: the tail is guarded), coarseP [Cpad][dp] zero padded,
//   cn2 [Cpad] = |c_j|^2 (fp64 sum rounded once, pin time), out [Q][Cpad], qn2 [Q] = |q|^2.
// Also clears the round-one scratch (ZeroArgs), as coarse_tile_kernel does.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coarse_approx_kernel(const float* __restrict__ queries, const float* __restrict__ coarseP,
                                                           const float* __restrict__ cn2, float* __restrict__ out,
                                                           float* __restrict__ qn2, int Q, int Cpad, int d, int dp, ZeroArgs z) {
  typedef float f16v __attribute__((ext_vector_type(16)));
  {
    const int gtid = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x, gsz = gridDim.x * gridDim.y * 256;
#pragma unroll
    for (int a = 0; a < 5; ++a)
      for (int i = gtid; i < z.n[a]; i += gsz) z.p[a][i] = 0u;
  }
  __shared__ float rown[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.y * 64 + (wave >> 1) * 32, c0 = blockIdx.x * 64 + (wave & 1) * 32;
  const int qrow = (q0 + r < Q) ? q0 + r : Q - 1;
  const float* ap = queries + (size_t)qrow * d + 4 * h;
  const float* bp = coarseP + (size_t)(c0 + r) * dp + 4 * h;
  const bool d4 = (d & 3) == 0;   // 16-byte alignment of the query rows
  f16v acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
  float nrm = 0.0f;
  auto load_a = [&](int k) -> float4 {
    const int kk = k + 4 * h;
    if (d4 && kk + 4 <= d) return *reinterpret_cast<const float4*>(ap + k);
    float4 v;
    v.x = kk + 0 < d ? ap[k + 0] : 0.0f;
    v.y = kk + 1 < d ? ap[k + 1] : 0.0f;
    v.z = kk + 2 < d ? ap[k + 2] : 0.0f;
    v.w = kk + 3 < d ? ap[k + 3] : 0.0f;
    return v;
  };
  constexpr int UN = 4;   // iterations whose loads are in flight together
  for (int k0 = 0; k0 < dp; k0 += 8 * UN) {
    float4 av[UN], bv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = k0 + 8 * u;
      if (k < dp) {
        av[u] = load_a(k);
        bv[u] = *reinterpret_cast<const float4*>(bp + k);
      } else {
        av[u] = float4{0.f, 0.f, 0.f, 0.f};
        bv[u] = float4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
      nrm = __builtin_fmaf(av[u].x, av[u].x, nrm);
      nrm = __builtin_fmaf(av[u].y, av[u].y, nrm);
      nrm = __builtin_fmaf(av[u].z, av[u].z, nrm);
      nrm = __builtin_fmaf(av[u].w, av[u].w, nrm);
    }
  }
  nrm += __shfl_xor(nrm, 32, 64);
  if (h == 0) {
    rown[wave][r] = nrm;
    if ((wave & 1) == 0 && blockIdx.x == 0 && q0 + r < Q) qn2[q0 + r] = nrm;
  }
  __syncthreads();
  const float cn = cn2[c0 + r];
  // C layout of the 32x32 MFMA: register v of lane l holds row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int i = 8 * (v >> 2) + 4 * h + (v & 3);
    if (q0 + i < Q) out[(size_t)(q0 + i) * Cpad + c0 + r] = __builtin_fmaf(-2.0f, acc[v], rown[wave][i] + cn);
  }
}

// ---------------------------------------------------------------------------------------
// a7 probe plan on the approximate distances: one wave per active query.
//   1. the query's <= 1024 approximate distances into registers (16 per lane), cells already probed masked;
//   2. tau = the 2W-th smallest of the 64 lane minima (>= the 2W-th smallest overall); candidates =
//      cells with a <= tau + 2 eps: every cell among the 2W smallest EXACT keys is one of them
//      (at least 2W cells have d <= a + eps <= tau + eps, so such a cell has a - eps <= d <= tau + eps);
//   3. the reference's squareDistance for the candidates: lanes <-> dimensions for the separately rounded
//      (q_i - c_i)^2 (coalesced centroid rows), staged in LDS, then lane <-> candidate for the sequential sum;
//   4. the 2W smallest exact keys, ordered by cell id, replayed through updateTopK: identical to
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
};

static constexpr int PLAN2_NCB = 24;     // candidates refined per batch
static constexpr int PLAN2_PITCH = 301;  // floats per candidate row in LDS (odd: the per-candidate sums read conflict-free)

__device__ __forceinline__ uint32_t float_order_bits(float f) {   // monotone map float -> u32 (negative values included)
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(64) void probe_plan2_kernel(Plan2Args g) {
  const PlanArgs& a = g.p;
  __shared__ u64 stage[64];
  __shared__ int32_t cl[COARSE_MAX_CPAD];   // candidate cells
  __shared__ float ca[COARSE_MAX_CPAD];     // their approximate distances
  __shared__ float sq[PLAN2_NCB * PLAN2_PITCH];
  const int x = blockIdx.x, lane = threadIdx.x;
  const int q = a.active ? a.active[x] : x;
  const int W = a.W, L = 2 * W, d = g.d;
  uint32_t* used = a.used + (size_t)q * a.used_words;
  const float* drow = a.dist + (size_t)q * a.Cpad;
  constexpr int NV = COARSE_MAX_CPAD / 64;
  const float INF = __uint_as_float(0x7f800000u);

  float av[NV];
  uint32_t uw[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int j = u * 64 + lane;
    const int jc = j < a.C ? j : a.C - 1;
    av[u] = drow[jc];
    uw[u] = used[jc >> 5];
  }
  // the query itself: dimension i = lane + 64 u
  constexpr int QV = 5;   // d <= 320
  float qv[QV];
#pragma unroll
  for (int u = 0; u < QV; ++u) qv[u] = (lane + 64 * u < d) ? g.queries[(size_t)q * d + lane + 64 * u] : 0.0f;
  float eps;
  {
    const float s = __builtin_sqrtf(g.qn2[q]) * (1.0f + 1e-6f) + g.cmax;
    eps = s * s * COARSE_EPS;
  }
  const bool finite = eps < 1e30f;   // (false for NaN too)
  float mn = INF;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int j = u * 64 + lane;
    const bool valid = j < a.C && !((uw[u] >> (j & 31)) & 1u);
    if (!valid) av[u] = INF;
    if (av[u] == av[u]) mn = fminf(mn, av[u]);
    else av[u] = -INF;                          // NaN (non-finite table entries): always a candidate, never a threshold
  }
  float thr = INF;
  if (finite && !g.refine_all) {
    const u64 sorted = wave_sort64((u64)float_order_bits(mn));
    const uint32_t tb = (uint32_t)__shfl(sorted, L - 1 < 63 ? L - 1 : 63, 64);
    const float tau = __uint_as_float((tb & 0x80000000u) ? (tb & 0x7fffffffu) : ~tb);
    if (tau < 1e30f) thr = (tau + 2.0f * eps) * (1.0f + 1e-6f) + 1e-37f;
  }
  // candidates -> LDS
  int n_cand = 0;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const bool c = av[u] <= thr && av[u] < INF;
    const u64 mask = __ballot(c);
    if (mask != 0ull) {
      if (c) {
        const int slot = n_cand + lanes_below(mask);
        cl[slot] = u * 64 + lane;
        ca[slot] = av[u];
      }
      n_cand += __popcll(mask);
    }
  }
  __builtin_amdgcn_wave_barrier();

  const u64 limit = (u64)__float_as_uint(a.cell_limit) << 32;
  WaveSelect<1> sel;
  sel.init(stage, limit, L);
  int viol = 0;
  for (int b0 = 0; b0 < n_cand; b0 += PLAN2_NCB) {
    const int nb = n_cand - b0 < PLAN2_NCB ? n_cand - b0 : PLAN2_NCB;
    // (q_i - c_i)^2, separately rounded: four candidates' rows in flight at a time
    for (int b = 0; b < nb; b += 4) {
      float cv[4][QV];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int cell = cl[b0 + (b + t < nb ? b + t : nb - 1)];
        const float* crow = g.coarse + (size_t)cell * d;
#pragma unroll
        for (int u = 0; u < QV; ++u) cv[t][u] = (lane + 64 * u < d) ? crow[lane + 64 * u] : 0.0f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (b + t < nb) {
#pragma unroll
          for (int u = 0; u < QV; ++u) {
            const float df = qv[u] - cv[t][u];          // index_utils.c:500-508: sub, mul, add rounded one by one
            const float pr = df * df;
            if (lane + 64 * u < d) sq[(b + t) * PLAN2_PITCH + lane + 64 * u] = pr;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    u64 key = KEY_INF;
    if (lane < nb) {
      float acc = 0.0f;
      const float* row = sq + lane * PLAN2_PITCH;
#pragma unroll 10
      for (int i = 0; i < d; ++i) acc = acc + row[i];
      key = make_key(acc, (uint32_t)cl[b0 + lane]);
      const float ap = ca[b0 + lane];
      if (finite && !(__builtin_fabsf(acc - ap) <= eps) && ap > -INF) ++viol;
    }
    __builtin_amdgcn_wave_barrier();
    sel.push(key, lane < nb);
  }
  sel.finish();
  if (g.violations) {
    if (viol) atomicAdd(g.violations + 2, viol);
    if (g.refine_all && lane == 0) atomicAdd(g.violations + 3, n_cand);
  }
  u64 byp = (sel.acc[0] == KEY_INF || lane >= L) ? KEY_INF : ((sel.acc[0] << 32) | (sel.acc[0] >> 32));
  byp = wave_sort64(byp);
  // lane i = slot i of the W-entry list; candidates replayed in cell order (freddy.c:266-283)
  float d_slot = a.cell_limit;
  int32_t c_slot = -1;
  wave_list_replay(d_slot, c_slot, W, byp, L, [](uint32_t hi) { return (int32_t)hi; });
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
}

}  // namespace freddy
