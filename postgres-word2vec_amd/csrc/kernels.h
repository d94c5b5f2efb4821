// kernels.h -- hand-written HIP kernels (gfx950 / CDNA4) for the PQ / IVFADC hot path.
//
// Every floating-point distance is produced by separately rounded binary32 sub, mul and
// add in the reference's order (index_utils.c:500-508 squareDistance, :445-455
// getPrecomputedDistances, :1126-1133 computePQDistanceInt16).  This translation unit
// must be compiled with -ffp-contract=off; there is deliberately no fmaf and no MFMA
// here (an FMA rounds once where the reference rounds twice).
//
// Pipeline of one probing round (host side: freddy_gpu.hip):
//   transpose_queries -> coarse_dist -> probe_plan -> residual -> lut_build
//   -> adc_scan (per-wave top-L by (distance, scan position)) -> merge_replay
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wave_topk.h"

namespace freddy {

static constexpr int WG = 256;           // 4 waves per workgroup
static constexpr int ROWS_PER_BLOCK = 64;  // one row block = one wave-wide coalesced load

// ---------------------------------------------------------------------------------------
// a6/a7 distances: dist[q][j] = squareDistance(query q, coarse centroid j)
//   freddy.c:272-283, :855-866 ; index_utils.c:500-508
// lane <-> cell (coarseT is [d][Cpad], so reads and the dist row writes are coalesced);
// the query row is workgroup-uniform and comes through the scalar cache.  QJ queries per
// thread share every centroid load and give QJ independent add chains per lane.
// ---------------------------------------------------------------------------------------
template <int QJ>
__global__ __launch_bounds__(WG) void coarse_dist_kernel(const float* __restrict__ queries,
                                                        const float* __restrict__ coarseT,
                                                        float* __restrict__ dist, int Q, int Cpad,
                                                        int d) {
  // QJ query rows are staged in LDS transposed to [dim][QJ]: per dimension every lane then reads
  // the QJ query values with QJ/4 broadcast ds_read_b128 (same address in all lanes) instead of
  // QJ dependent scalar loads, and runs QJ independent add chains.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* qs = reinterpret_cast<float*>(smem);   // [d][QJ]
  const int j = blockIdx.x * WG + threadIdx.x;
  const int q0 = blockIdx.y * QJ;
  for (int i = threadIdx.x; i < d * QJ; i += WG) {
    const int t = i / d, dim = i - t * d;       // coalesced over dim inside a query row
    const int q = (q0 + t < Q) ? (q0 + t) : (Q - 1);
    qs[dim * QJ + t] = queries[(size_t)q * d + dim];
  }
  __syncthreads();
  // packed fp32 math: two queries per instruction (each half is an IEEE binary32 op, so the
  // sub/mul/add chain of every (query, cell) pair rounds exactly like squareDistance)
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f acc2[QJ / 2];
#pragma unroll
  for (int t = 0; t < QJ / 2; ++t) acc2[t] = v2f{0.0f, 0.0f};
  // centroid values are fetched DB dimensions ahead of their use (one wave per SIMD here, so
  // nothing else would hide the L2 latency)
  constexpr int DB = 8;
  float cn[DB];
#pragma unroll
  for (int u = 0; u < DB; ++u) cn[u] = (u < d) ? coarseT[(size_t)u * Cpad + j] : 0.0f;
  for (int i0 = 0; i0 < d; i0 += DB) {
    float cc[DB];
#pragma unroll
    for (int u = 0; u < DB; ++u) cc[u] = cn[u];
#pragma unroll
    for (int u = 0; u < DB; ++u) {
      const int in = i0 + DB + u;
      cn[u] = (in < d) ? coarseT[(size_t)in * Cpad + j] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < DB; ++u) {
      const int i = i0 + u;
      if (i < d) {
        const float4* qrow = reinterpret_cast<const float4*>(qs + i * QJ);
        const v2f c2 = {cc[u], cc[u]};
#pragma unroll
        for (int t4 = 0; t4 < QJ / 4; ++t4) {
          const float4 qv = qrow[t4];
          const v2f qa = {qv.x, qv.y}, qb = {qv.z, qv.w};
          const v2f da = qa - c2, db = qb - c2;
          const v2f pa = da * da, pb = db * db;
          acc2[t4 * 2 + 0] = acc2[t4 * 2 + 0] + pa;
          acc2[t4 * 2 + 1] = acc2[t4 * 2 + 1] + pb;
        }
      }
    }
  }
  float acc[QJ];
#pragma unroll
  for (int t = 0; t < QJ / 2; ++t) { acc[2 * t] = acc2[t].x; acc[2 * t + 1] = acc2[t].y; }
#pragma unroll
  for (int t = 0; t < QJ; ++t)
    if (q0 + t < Q) dist[(size_t)(q0 + t) * Cpad + j] = acc[t];
}

// Tiled form of the same computation for batches: a workgroup produces a 64-query x 64-cell tile,
// every thread a 4x4 block of it.  Queries and centroids are staged through LDS DK dimensions
// at a time (double buffered, one barrier per chunk of DK dimensions), so per dimension a thread issues two
// ds_read_b128 for 24 packed VALU instructions (8 independent sub/mul/add chains) and the tables
// cross the L2 once per tile.  Dimensions past d are staged as zeros: (0-0)^2 adds +0, which
// leaves every partial sum bit-identical.  Order per (query, cell): dimensions ascending, as
// squareDistance (index_utils.c:500-508).
// Scratch words the round's later kernels expect zeroed (probe bitmaps, counters): cleared by the
// coarse kernel's threads on their way in instead of by four or five separate memset launches.
struct ZeroArgs {
  uint32_t* p[5];
  int n[5];
};

// Small batches (below 32 queries -- the reference's ivfadc_search takes ONE, freddy.c:174-393): one wave per (64 cells,
// query).  The chain r = r + (q_i - c_i)^2 is sequential in i (index_utils.c:500-508), but its loads are not: DB centroid
// values per lane are requested together, then summed in order -- coarse_dist_kernel's four workgroups walked the 300
// dimensions with 8 loads in flight and took 39 us for one query; this takes the round trips of d / DB batches.
// Also clears the round-one scratch (ZeroArgs), as the batch kernels do: no memset launches in front of a small call.
template <int DB>
__global__ __launch_bounds__(64) void coarse_small_kernel(const float* __restrict__ queries, const float* __restrict__ coarseT,
                                                         float* __restrict__ dist, int Q, int Cpad, int d, ZeroArgs z) {
  {
    const int gtid = (blockIdx.y * gridDim.x + blockIdx.x) * 64 + threadIdx.x, gsz = gridDim.x * gridDim.y * 64;
#pragma unroll
    for (int a = 0; a < 5; ++a)
      for (int i = gtid; i < z.n[a]; i += gsz) z.p[a][i] = 0u;
  }
  __shared__ float qs[1024];
  const int j = blockIdx.x * 64 + threadIdx.x, q = blockIdx.y;
  for (int i = threadIdx.x; i < d; i += 64) qs[i] = queries[(size_t)q * d + i];
  __syncthreads();
  float acc = 0.0f;
  // (no guards around the loads: hipcc turns a guarded load into a basic block of its own with a full memory wait)
  const int nfull = d / DB * DB;
  for (int i0 = 0; i0 < nfull; i0 += DB) {
    float cc[DB];
#pragma unroll
    for (int u = 0; u < DB; ++u) cc[u] = coarseT[(size_t)(i0 + u) * Cpad + j];
#pragma unroll
    for (int u = 0; u < DB; ++u) {
      const float t = qs[i0 + u] - cc[u];
      const float pr = t * t;
      acc = acc + pr;
    }
  }
  for (int i = nfull; i < d; ++i) {
    const float t = qs[i] - coarseT[(size_t)i * Cpad + j];
    const float pr = t * t;
    acc = acc + pr;
  }
  dist[(size_t)q * Cpad + j] = acc;
}
// TCW = cells per thread (4: 64x64 tile, one workgroup per CU for Q = C = 1024; 2: 64x32 tile, twice
// the workgroups -- two waves per SIMD issue packed ops ~25 % faster than one)
template <int TCW, int DK>
__global__ __launch_bounds__(256) void coarse_tile_kernel(const float* __restrict__ queries,
                                                         const float* __restrict__ coarseT,
                                                         float* __restrict__ dist, int Q, int Cpad, int d, ZeroArgs z) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  {
    const int gtid = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x, gsz = gridDim.x * gridDim.y * 256;
#pragma unroll
    for (int a = 0; a < 5; ++a)
      for (int i = gtid; i < z.n[a]; i += gsz) z.p[a][i] = 0u;
  }
  constexpr int TCELLS = 16 * TCW;   // cells per tile
  constexpr int APT = DK / 4;        // dimensions of one query each thread stages per chunk
  constexpr int BPT = DK / 16;       // dimension rows (of TCW cells) each thread stages per chunk
  static_assert(DK % 16 == 0, "staging roles");
  __shared__ __attribute__((aligned(16))) float As[2][DK][64];
  __shared__ __attribute__((aligned(16))) float Bs[2][DK][TCELLS];
  const int tid = threadIdx.x;
  const int tc = tid & 15, tq = tid >> 4;
  const int c0 = blockIdx.x * TCELLS, q0 = blockIdx.y * 64;
  // staging roles
  const int aq = tid >> 2, adim = (tid & 3) * APT;    // query aq, dims adim..adim+APT-1 of the chunk
  const int bdim = tid >> 4, bc = (tid & 15) * TCW;   // dims bdim, bdim+16, ..., cells bc..bc+TCW-1
  const int aqg = (q0 + aq < Q) ? q0 + aq : Q - 1;
  const float* arow = queries + (size_t)aqg * d;
  float ra[APT];
  float rb[BPT][TCW];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < APT; ++u) ra[u] = (k0 + adim + u < d) ? arow[k0 + adim + u] : 0.0f;
#pragma unroll
    for (int w = 0; w < BPT; ++w) {
      const int dim = k0 + bdim + w * 16;
      const float* brow = coarseT + (size_t)dim * Cpad + c0 + bc;
      if (dim < d) {
        if (TCW == 4) { const float4 v = *reinterpret_cast<const float4*>(brow); rb[w][0] = v.x; rb[w][1] = v.y; rb[w][2 % TCW] = v.z; rb[w][3 % TCW] = v.w; }
        else { const float2 v = *reinterpret_cast<const float2*>(brow); rb[w][0] = v.x; rb[w][1] = v.y; }
      } else {
#pragma unroll
        for (int i = 0; i < TCW; ++i) rb[w][i] = 0.0f;
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < APT; ++u) As[buf][adim + u][aq] = ra[u];
#pragma unroll
    for (int w = 0; w < BPT; ++w)
#pragma unroll
      for (int i = 0; i < TCW; ++i) Bs[buf][bdim + w * 16][bc + i] = rb[w][i];
  };
  v2f acc[TCW][2];   // [cell i][query pair j]
#pragma unroll
  for (int i = 0; i < TCW; ++i) acc[i][0] = acc[i][1] = v2f{0.0f, 0.0f};
  fetch(0);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < d; k0 += DK) {
    const bool more = k0 + DK < d;
    if (more) fetch(k0 + DK);
#pragma unroll
    for (int dd = 0; dd < DK; ++dd) {
      const float4 a4 = *reinterpret_cast<const float4*>(&As[buf][dd][tq * 4]);
      float bv[TCW];
      if (TCW == 4) {
        const float4 b4 = *reinterpret_cast<const float4*>(&Bs[buf][dd][tc * TCW]);
        bv[0] = b4.x; bv[1] = b4.y; bv[2 % TCW] = b4.z; bv[3 % TCW] = b4.w;
      } else {
        const float2 b2 = *reinterpret_cast<const float2*>(&Bs[buf][dd][tc * TCW]);
        bv[0] = b2.x; bv[1] = b2.y;
      }
      const v2f qa = {a4.x, a4.y}, qb = {a4.z, a4.w};
#pragma unroll
      for (int i = 0; i < TCW; ++i) {
        const v2f c2 = {bv[i], bv[i]};
        const v2f da = qa - c2, db = qb - c2;
        const v2f pa = da * da, pb = db * db;
        acc[i][0] = acc[i][0] + pa;
        acc[i][1] = acc[i][1] + pb;
      }
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = q0 + tq * 4 + j;
    if (q < Q) {
#pragma unroll
      for (int i = 0; i < TCW; ++i)
        dist[(size_t)q * Cpad + c0 + tc * TCW + i] = (j & 1) ? acc[i][j >> 1].y : acc[i][j >> 1].x;
    }
  }
}

// ---------------------------------------------------------------------------------------
// a7 probe plan: one wave per active query reproduces the reference's cell selection
//   (freddy.c:266-293): cells are offered in ascending id, used ones skipped, and the W
//   best kept through updateTopK's insertion rule.  (The reference starts its threshold
//   at 1000.0 over a list of 100.0 sentinels; a cell at distance >= 100 can never be
//   inserted, so this is the guarded insertion with sentinel 100.0.)  W == 1 is the batch
//   UDF's strict-< argmin (freddy.c:853-866).
//   Parallel form: the 2W smallest (distance, cell id) keys are selected by the wave,
//   ordered by cell id and replayed by lane 0 -- same argument as merge_replay below.
//   Emits W work items per query (cell -1 = no cell left) and marks the cells used.
// ---------------------------------------------------------------------------------------
struct PlanArgs {
  const float* dist;        // [Q][Cpad]
  const int32_t* active;    // [n_active] query indices (NULL = identity)
  const int32_t* list_off;  // [C+1]
  uint32_t* used;           // [Q][used_words] bitmap of cells already probed
  int32_t* item_cell;       // [n_active*W]
  int32_t* item_query;      // [n_active*W]
  float* item_dist;         // [n_active*W] or NULL: the item's coarse distance (the scan's bound on |r|^2)
  int32_t* round_rows;      // [n_active] rows retrieved this round, -1 = no cell was left
  int32_t* cell_count;      // [C] fused path only (NULL otherwise): += items probing each cell
  int32_t* cell_items;      // [C][cell_cap] fused path: the items of each cell, in arrival order (any order is fine)
  int cell_cap;             // >= number of active queries (a query probes a cell at most once)
  int n_active, Cpad, C, W, used_words;
  float cell_limit;         // a cell at this distance or beyond is never probed: 100.0 (ivfadc_search, the cell list's
                            // sentinel, freddy.c:266-283), 1000.0 (ivfadc_batch_search, minDist of its argmin, freddy.c:855)
};

template <int V>
__global__ __launch_bounds__(64) void probe_plan_kernel(PlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* stage = reinterpret_cast<u64*>(smem);                 // [64]
  u64* cand = stage + 64;                                    // [64*V]
  float* sd = reinterpret_cast<float*>(cand + 64 * V);       // [W]
  int32_t* sc = reinterpret_cast<int32_t*>(sd + a.W);        // [W]
  const int x = blockIdx.x;
  const int lane = threadIdx.x;
  const int q = a.active ? a.active[x] : x;
  const int W = a.W;
  const int L = 2 * W;
  uint32_t* used = a.used + (size_t)q * a.used_words;
  const float* drow = a.dist + (size_t)q * a.Cpad;

  const u64 limit = (u64)__float_as_uint(a.cell_limit) << 32;
  // Cells are read in batches of PB x 64: distance and used-bitmap word of every slot are independent
  // loads issued together (one wave per query: nothing else would hide their latency).
  constexpr int PB = 8;
  auto load_batch = [&](int base, u64 (&key)[PB], bool (&valid)[PB]) {
    float dv[PB];
    uint32_t uw[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int j = base + u * 64 + lane;
      const int jc = j < a.C ? j : a.C - 1;
      dv[u] = drow[jc];
      uw[u] = used[jc >> 5];
    }
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int j = base + u * 64 + lane;
      valid[u] = j < a.C && !((uw[u] >> (j & 31)) & 1u);
      key[u] = make_key(dv[u], (uint32_t)j);
    }
  };
  // Pre-pass: the L-th smallest of the 64 per-lane minima bounds the L-th smallest key from above,
  // so the selection below starts with a tight threshold and (almost always) a single merge.
  u64 tau0 = limit;
  if (L <= 64) {
    u64 mn = KEY_INF;
    for (int base = 0; base < a.C; base += 64 * PB) {
      u64 key[PB];
      bool valid[PB];
      load_batch(base, key, valid);
#pragma unroll
      for (int u = 0; u < PB; ++u)
        if (valid[u] && key[u] < mn) mn = key[u];
    }
    mn = wave_sort64(mn);
    const u64 t = __shfl(mn, L - 1, 64);
    if (t != KEY_INF && t + 1 < tau0) tau0 = t + 1;   // keys are unique: "< t+1" keeps t itself
  }
  WaveSelect<V> sel;
  sel.init(stage, tau0, L);
  for (int base = 0; base < a.C; base += 64 * PB) {
    u64 key[PB];
    bool valid[PB];
    load_batch(base, key, valid);
#pragma unroll
    for (int u = 0; u < PB; ++u)
      if (base + u * 64 < a.C) sel.push(key[u], valid[u]);
  }
  sel.finish();
  u64 byp[V];
#pragma unroll
  for (int v = 0; v < V; ++v)
    byp[v] = (sel.acc[v] == KEY_INF || v * 64 + lane >= L) ? KEY_INF : ((sel.acc[v] << 32) | (sel.acc[v] >> 32));
  wave_sort_full<V>(byp);
  if (V == 1 && W <= 64) {
    // lane i = slot i of the W-entry list; candidates replayed in cell order
    float d_slot = a.cell_limit;
    int32_t c_slot = -1;
    wave_list_replay(d_slot, c_slot, W, byp[0], L, [](uint32_t hi) { return (int32_t)hi; });
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
      if (a.item_dist) a.item_dist[(size_t)x * W + lane] = d_slot;
    }
    return;
  }
#pragma unroll
  for (int v = 0; v < V; ++v) cand[v * 64 + lane] = byp[v];
  for (int i = lane; i < W; i += 64) { sd[i] = a.cell_limit; sc[i] = -1; }
  __syncthreads();
  if (lane == 0) {
    float maxd = a.cell_limit;
    for (int e = 0; e < L; ++e) {
      const u64 c = cand[e];
      if (c == KEY_INF) break;
      const float dv = __uint_as_float((uint32_t)c);
      if (dv < maxd) {
        int slot = W - 1;                                  // updateTopK, index_utils.c:19-33
        while (slot >= 0 && !(sd[slot] < dv)) --slot;
        ++slot;
        for (int t = W - 2; t >= slot; --t) { sd[t + 1] = sd[t]; sc[t + 1] = sc[t]; }
        sd[slot] = dv;
        sc[slot] = (int32_t)(c >> 32);
        maxd = sd[W - 1];
      }
    }
    int rows = 0, n_cells = 0;
    for (int i = 0; i < W; ++i) {
      const int c = sc[i];
      if (c >= 0) {
        used[c >> 5] |= 1u << (c & 31);
        rows += a.list_off[c + 1] - a.list_off[c];
        ++n_cells;
        if (a.cell_count) {
          const int at = atomicAdd(a.cell_count + c, 1);
          a.cell_items[(size_t)c * a.cell_cap + at] = x * W + i;
        }
      }
    }
    a.round_rows[x] = n_cells ? rows : -1;   // -1: every cell already used, the query retires
  }
  __syncthreads();
  for (int i = lane; i < W; i += 64) {
    a.item_cell[(size_t)x * W + i] = sc[i];
    a.item_query[(size_t)x * W + i] = q;
    if (a.item_dist) a.item_dist[(size_t)x * W + i] = sd[i];
  }
}

// ---------------------------------------------------------------------------------------
// a8 residual r = q - cq[cell], elementwise binary32 (freddy.c:296-303, :876-879)
// ---------------------------------------------------------------------------------------
// Output row layout: [m][SP] with each position's S values padded to SP (SP == S: dense [d]).
static __global__ __launch_bounds__(WG) void residual_kernel(const float* __restrict__ queries,
                                                     const float* __restrict__ coarse,
                                                     const int32_t* __restrict__ item_cell,
                                                     const int32_t* __restrict__ item_query,
                                                     float* __restrict__ resid, int d, int S, int SP) {
  const int item = blockIdx.x;
  const int cell = item_cell[item];
  if (cell < 0) return;
  const float* q = queries + (size_t)(item_query ? item_query[item] : item) * d;   // (NULL: item i is query i)
  const float* c = coarse + (size_t)cell * d;
  const int row = (d / S) * SP;
  for (int o = threadIdx.x; o < row; o += WG) {
    const int p = o / SP, j = o - p * SP;
    resid[(size_t)item * row + o] = (j < S) ? q[p * S + j] - c[p * S + j] : 0.0f;
  }
}

// ---------------------------------------------------------------------------------------
// a2 LUT build: lut[item][pos*K + code] = squareDistance(r_item[pos*S..], cb[pos][code], S)
//   index_utils.c:445-455.  This is the arithmetic-heavy phase (m*K*S*3 separately rounded
//   ops per item, no FMA allowed), so the codebook slice of E codes per thread is pulled
//   into registers ONCE per workgroup and reused for `items_per_wg` residuals; the
//   residual sub-vector is wave-uniform and arrives through the scalar cache.
//   cbT layout: [m][S][K] (code contiguous -> coalesced register fill, coalesced LUT store).
// ---------------------------------------------------------------------------------------
template <int S, int E>
__global__ __launch_bounds__(WG) void lut_build_kernel(const float* __restrict__ vecs,   // [items][d] residuals (or queries)
                                                      const int32_t* __restrict__ item_cell,  // NULL: every item valid
                                                      const float* __restrict__ cbT,
                                                      float* __restrict__ lut, int n_items,
                                                      int items_per_wg, int m, int K, int d,
                                                      const float* __restrict__ coarse = nullptr,       // non-NULL: vecs are QUERIES and the
                                                      const int32_t* __restrict__ item_query = nullptr) { // residual q - coarse[cell] (freddy.c:296-303) is formed here
  const int p = blockIdx.x;
  const int it0 = blockIdx.y * items_per_wg;
  const int it1 = (it0 + items_per_wg < n_items) ? it0 + items_per_wg : n_items;
  const size_t lutN = (size_t)m * K;
  for (int c0 = 0; c0 < K; c0 += WG * E) {
    float cb[E][S];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int c = c0 + e * WG + (int)threadIdx.x;
#pragma unroll
      for (int j = 0; j < S; ++j) cb[e][j] = (c < K) ? cbT[((size_t)p * S + j) * K + c] : 0.0f;
    }
    for (int it = it0; it < it1; ++it) {
      if (item_cell && item_cell[it] < 0) continue;
      const float* r = vecs + (size_t)(coarse && item_query ? item_query[it] : it) * d + (size_t)p * S;
      const float* co = coarse ? coarse + (size_t)item_cell[it] * d + (size_t)p * S : nullptr;
      float acc[E];
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = 0.0f;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const float rj = co ? r[j] - co[j] : r[j];   // (the residual_kernel's value: one binary32 subtraction)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float t = rj - cb[e][j];
          const float pr = t * t;
          acc[e] = acc[e] + pr;
        }
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int c = c0 + e * WG + (int)threadIdx.x;
        if (c < K) lut[(size_t)it * lutN + (size_t)p * K + c] = acc[e];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Index build: PQ encoding = exact 1-NN of every sub-vector among its position's K
// codewords by squareDistance, lowest code on ties (index_creation/pq_index.py:65-92 "create_index":
// strict "<" over the codes in order).  Same register-cached codebook slice and the same sequential
// fp32 chain as lut_build_kernel, but the K distances of an item are reduced to their argmin instead of
// being stored: key = (distance bits << 32 | code), minimum over the lane's codes, the wave, the
// workgroup.  S == 0: runtime sub-vector size, codebook streamed from L2.
//   grid (m, ceil(n_items / items_per_wg)); codes[item][p] int16.
// ---------------------------------------------------------------------------------------
template <int S, int E>
__global__ __launch_bounds__(WG) void encode_pq_kernel(const float* __restrict__ vecs, const float* __restrict__ cbT,
                                                      int16_t* __restrict__ codes, int n_items, int items_per_wg, int m,
                                                      int K, int d, int S_rt, float limit, int32_t* __restrict__ too_far) {
  __shared__ u64 wmin[WG / 64];
  const int p = blockIdx.x;
  const int it0 = blockIdx.y * items_per_wg;
  const int it1 = (it0 + items_per_wg < n_items) ? it0 + items_per_wg : n_items;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Sr = S ? S : S_rt;
  for (int it = it0; it < it1; ++it) {
    const float* r = vecs + (size_t)it * d + (size_t)p * Sr;
    u64 best = KEY_INF;
    for (int c0 = 0; c0 < K; c0 += WG * E) {
      // (for K <= WG*E, i.e. every configuration of the reference, this loop runs once and the compiler
      // keeps the codebook slice in registers across the items of the workgroup)
      float acc[E];
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = 0.0f;
      if (S) {
#pragma unroll
        for (int j = 0; j < (S ? S : 1); ++j) {
          const float rj = r[j];
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const int c = c0 + e * WG + (int)threadIdx.x;
            const float cv = (c < K) ? cbT[((size_t)p * S + j) * K + c] : 0.0f;
            const float t = rj - cv;
            const float pr = t * t;
            acc[e] = acc[e] + pr;
          }
        }
      } else {
        for (int j = 0; j < Sr; ++j) {
          const float rj = r[j];
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const int c = c0 + e * WG + (int)threadIdx.x;
            const float cv = (c < K) ? cbT[((size_t)p * Sr + j) * K + c] : 0.0f;
            const float t = rj - cv;
            const float pr = t * t;
            acc[e] = acc[e] + pr;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int c = c0 + e * WG + (int)threadIdx.x;
        if (c < K) best = umin64(best, make_key(acc[e], (uint32_t)c));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = umin64(best, __shfl_xor(best, o, 64));
    __syncthreads();
    if (lane == 0) wmin[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
      u64 b = wmin[0];
#pragma unroll
      for (int w = 1; w < WG / 64; ++w) b = umin64(b, wmin[w]);
      codes[(size_t)it * m + p] = (int16_t)key_pos(b);
      // insert_batch searches from minDist = 100 by strict "<" (index_utils.c:925-939): the same code unless NO
      // entry is nearer than the limit, which the reference leaves undefined -- reported
      if (too_far && !(key_dist(b) < limit)) atomicAdd(too_far, 1);
    }
  }
}

// Coarse assignment: nearest of C centroids by squareDistance over all d dimensions, lowest index on
// ties (faiss IndexFlatL2 search k=1 / ivfadc.py); one wave per vector, lane <-> centroid.
static __global__ __launch_bounds__(64) void assign_coarse_kernel(const float* __restrict__ vecs, const float* __restrict__ coarseT,
                                                          int32_t* __restrict__ cell, int n, int C, int Cpad, int d,
                                                          float limit, int32_t* __restrict__ too_far) {
  const int it = blockIdx.x, lane = threadIdx.x;
  if (it >= n) return;
  const float* v = vecs + (size_t)it * d;
  u64 best = KEY_INF;
  for (int c = lane; c < C; c += 64) {
    float acc = 0.0f;
    for (int j = 0; j < d; ++j) {
      const float t = v[j] - coarseT[(size_t)j * Cpad + c];
      const float pr = t * t;
      acc = acc + pr;
    }
    best = umin64(best, make_key(acc, (uint32_t)c));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = umin64(best, __shfl_xor(best, o, 64));
  if (lane == 0) {
    cell[it] = (int32_t)key_pos(best);
    if (too_far && !(key_dist(best) < limit)) atomicAdd(too_far, 1);   // (freddy.c:1568-1575: minDistCoarse = 100)
  }
}

// ---------------------------------------------------------------------------------------
// Quantizer training (quantizer_creation.py:13-52): Lloyd's k-means.  Assignment =
// assign_coarse_kernel on the transposed centroids; update = kmeans_update_kernel: one workgroup per
// cluster walks the assignment array in index order, compacts its members chunk by chunk into LDS
// (ballot + prefix counts: the order of the members is their index order) and adds their vectors
// dimension-parallel, member-sequential -- a binary32 sum in a fixed order, so the result does not depend on
// the launch and equals the restatement in oracle/ bit for bit.  An empty cluster keeps its centroid.
// ---------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void kmeans_transpose_kernel(const float* __restrict__ cent, float* __restrict__ centT, int k, int kpad, int d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d * kpad) return;
  const int dim = i / kpad, c = i - dim * kpad;
  centT[i] = c < k ? cent[(size_t)c * d + dim] : 0.0f;
}
static __global__ __launch_bounds__(256) void kmeans_update_kernel(const float* __restrict__ vecs, const int32_t* __restrict__ assign, int64_t n,
                                                           int d, float* __restrict__ cent) {
  __shared__ int32_t members[256];
  __shared__ int wcount[4];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int DV = 4;   // dimensions per thread: d <= 1024
  float sum[DV];
#pragma unroll
  for (int u = 0; u < DV; ++u) sum[u] = 0.0f;
  int64_t cnt = 0;
  for (int64_t base = 0; base < n; base += 256) {
    const int64_t i = base + tid;
    const bool mine = i < n && assign[i] == c;
    const u64 mask = __ballot(mine);
    if (lane == 0) wcount[wave] = __popcll(mask);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) off += wcount[w]; total += wcount[w]; }
    if (mine) members[off + lanes_below(mask)] = (int32_t)(i - base);
    __syncthreads();
    for (int t = 0; t < total; ++t) {
      const float* v = vecs + (size_t)(base + members[t]) * d;
#pragma unroll
      for (int u = 0; u < DV; ++u)
        if (tid + 256 * u < d) sum[u] = sum[u] + v[tid + 256 * u];
    }
    cnt += total;
    __syncthreads();
  }
  if (cnt > 0) {
#pragma unroll
    for (int u = 0; u < DV; ++u)
      if (tid + 256 * u < d) cent[(size_t)c * d + tid + 256 * u] = sum[u] / (float)cnt;
  }
}

// generic sub-vector size (runtime S): no register cache, codebook streamed from L2
static __global__ __launch_bounds__(WG) void lut_build_generic_kernel(const float* __restrict__ vecs,
                                                              const int32_t* __restrict__ item_cell,
                                                              const float* __restrict__ cbT,
                                                              float* __restrict__ lut, int n_items,
                                                              int items_per_wg, int m, int K, int d,
                                                              int S, const float* __restrict__ coarse = nullptr,
                                                              const int32_t* __restrict__ item_query = nullptr) {
  const int p = blockIdx.x;
  const int it0 = blockIdx.y * items_per_wg;
  const int it1 = (it0 + items_per_wg < n_items) ? it0 + items_per_wg : n_items;
  const size_t lutN = (size_t)m * K;
  for (int it = it0; it < it1; ++it) {
    if (item_cell && item_cell[it] < 0) continue;
    const float* r = vecs + (size_t)(coarse && item_query ? item_query[it] : it) * d + (size_t)p * S;
    const float* co = coarse ? coarse + (size_t)item_cell[it] * d + (size_t)p * S : nullptr;
    for (int c = threadIdx.x; c < K; c += WG) {
      float acc = 0.0f;
      for (int j = 0; j < S; ++j) {
        const float rj = co ? r[j] - co[j] : r[j];
        const float t = rj - cbT[((size_t)p * S + j) * K + c];
        const float pr = t * t;
        acc = acc + pr;
      }
      lut[(size_t)it * lutN + (size_t)p * K + c] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------
// a4 + a5 ADC scan with fused selection.
//   One workgroup (8 waves) = one (item, chunk of row blocks).  The item's LUT (m*K floats,
//   48 KiB for m=12, K=1024) is staged in LDS with 16-byte loads issued back to back; each
//   wave then walks 64-row blocks with the next block's loads in flight: per row M2
//   coalesced dwords of packed int16 codes + one dword of scan position, m LDS gathers
//   summed in position order (index_utils.c:1126-1133), then WaveSelect on the 64-bit
//   (distance, position) key.
//   Output: the workgroup's L smallest keys -> part[(item*nchunk + chunk)*L + r].
// ---------------------------------------------------------------------------------------
static constexpr int SCAN_WG = 512;
static constexpr int SCAN_WAVES = SCAN_WG / 64;

struct ScanArgs {
  const float* lut;           // [items][m*K]
  const int32_t* item_list;   // [items] list (cell) of the item; NULL = list 0; -1 = skip
  const int32_t* item_query;  // [items] query of the item; NULL = item
  const int32_t* blk_off;     // [n_lists+1] first row block of each list
  const uint32_t* packed;     // [blocks][M2][64] two int16 codes per dword
  const int32_t* pos;         // [blocks*64] scan position (row id / row index), -1 = padding
  u64* part;                  // [items][nchunk][L]
  int32_t* cand_count;        // [Q] += candidates with dist < sentinel (FOUND_ACCEPTED rule)
  int m, K, chunk_blocks, nchunk, L;
  uint32_t sentinel_bits;
  // FLOOR instantiations (k > 512, bigk.h: the 2k smallest keys are selected 1024 at a time): only keys ABOVE the query's floor
  const u64* floor = nullptr;   // [Q] the largest key the passes so far have selected
};

template <int M2T>
struct RowBlock {
  uint32_t w[M2T];
  int32_t p;
};

template <int M, int V, bool FLOOR = false>
__global__ __launch_bounds__(SCAN_WG) void adc_scan_kernel(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int M2T = (M > 0) ? (M + 1) / 2 : 1;
  const int m = (M > 0) ? M : a.m;
  const int K = a.K;
  const int lutN = m * K;
  float* lut = reinterpret_cast<float*>(smem);
  u64* stage = reinterpret_cast<u64*>(smem + (((size_t)lutN * 4 + 15) & ~(size_t)15));

  const int item = blockIdx.y, chunk = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int list = a.item_list ? a.item_list[item] : 0;

  WaveSelect<V> sel;
  const u64 sentinel_key = (u64)a.sentinel_bits << 32;   // key < this  <=>  dist < sentinel
  sel.init(stage + wave * 64, sentinel_key, a.L);
  int accepted = 0;
  u64 flo = 0;
  if constexpr (FLOOR) flo = a.floor[a.item_query ? a.item_query[item] : item];

  int b0 = 0, b1 = 0;
  if (list >= 0) {
    b0 = a.blk_off[list] + chunk * a.chunk_blocks;
    b1 = a.blk_off[list + 1];
    if (b0 + a.chunk_blocks < b1) b1 = b0 + a.chunk_blocks;
  }
  if (b0 < b1) {  // workgroup-uniform
    const float* src = a.lut + (size_t)item * lutN;
    if ((lutN & 3) == 0) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
      float4* d4 = reinterpret_cast<float4*>(lut);
      const int n4 = lutN >> 2;
      for (int i0 = 0; i0 < n4; i0 += SCAN_WG * 6) {   // six 16-byte loads in flight per lane
        float4 t[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int i = i0 + u * SCAN_WG + (int)threadIdx.x;
          if (i < n4) t[u] = s4[i];
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int i = i0 + u * SCAN_WG + (int)threadIdx.x;
          if (i < n4) d4[i] = t[u];
        }
      }
    } else {
      for (int i = threadIdx.x; i < lutN; i += SCAN_WG) lut[i] = src[i];
    }
    __syncthreads();

    if (M > 0) {
      // PF row blocks of this wave in flight: with one block ahead a wave spent a global round trip per block (a single
      // query over 1 M rows: 245 workgroups x 8 waves x 8 blocks, 24 us for 28 MB that sit in the caches)
      constexpr int PF = 4;
      RowBlock<M2T> ring[PF];
      auto fetch = [&](RowBlock<M2T>& rb, int blk) {
        const int bc = blk < b1 ? blk : b1 - 1;           // (past the end: a repeat of the last block, never used)
        const uint32_t* pk = a.packed + (size_t)bc * M2T * 64 + lane;
#pragma unroll
        for (int j = 0; j < M2T; ++j) rb.w[j] = pk[j * 64];
        rb.p = a.pos[(size_t)bc * 64 + lane];
      };
      int b = b0 + wave;
#pragma unroll
      for (int u = 0; u < PF; ++u) fetch(ring[u], b + u * SCAN_WAVES);
      for (; b < b1; b += PF * SCAN_WAVES) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const RowBlock<M2T> cur = ring[u];
          const int bu = b + u * SCAN_WAVES;
          fetch(ring[u], bu + PF * SCAN_WAVES);            // the block this slot serves in the next trip of the loop
          if (bu < b1) {   // wave-uniform
            float dist = 0.0f;
#pragma unroll
            for (int l = 0; l < M; ++l) {
              const uint32_t code = (l & 1) ? (cur.w[l >> 1] >> 16) : (cur.w[l >> 1] & 0xffffu);
              dist = dist + lut[l * K + code];
            }
            const u64 key = make_key(dist, (uint32_t)cur.p);
            const bool valid = (cur.p >= 0) && (!FLOOR || key > flo);
            accepted += __popcll(__ballot(valid && key < sentinel_key));
            sel.push(key, valid);
          }
        }
      }
    } else {
      const int M2 = (m + 1) >> 1;
      for (int b = b0 + wave; b < b1; b += SCAN_WAVES) {
        const uint32_t* pk = a.packed + (size_t)b * M2 * 64 + lane;
        const int32_t p = a.pos[(size_t)b * 64 + lane];
        float dist = 0.0f;
        for (int l = 0; l < m; l += 2) {
          const uint32_t w = pk[(l >> 1) * 64];
          dist = dist + lut[l * K + (w & 0xffffu)];
          if (l + 1 < m) dist = dist + lut[(l + 1) * K + (w >> 16)];
        }
        const u64 key = make_key(dist, (uint32_t)p);
        const bool valid = (p >= 0) && (!FLOOR || key > flo);
        accepted += __popcll(__ballot(valid && key < sentinel_key));
        sel.push(key, valid);
      }
    }
    sel.finish();
  }
  // the 8 waves' lists meet in LDS (the LUT is dead by now) and wave 0 merges them, so the
  // workgroup emits ONE list of L keys
  __syncthreads();
  u64* lists = reinterpret_cast<u64*>(smem);   // [SCAN_WAVES][64*V]
#pragma unroll
  for (int v = 0; v < V; ++v) lists[((size_t)wave * V + v) * 64 + lane] = sel.acc[v];
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < SCAN_WAVES; ++w) {
      for (int v = 0; v < V; ++v) {
        const u64 key = lists[((size_t)w * V + v) * 64 + lane];
        if (__ballot(key != KEY_INF) == 0ull) break;   // ascending: the rest of this list is empty too
        wave_topk_absorb_sorted<V>(sel.acc, key);      // (a row of another wave's accumulator: already ascending)
      }
    }
    u64* out = a.part + ((size_t)item * a.nchunk + chunk) * a.L;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int r = v * 64 + lane;
      if (r < a.L) out[r] = sel.acc[v];
    }
  }
  if (a.cand_count && lane == 0 && accepted)
    atomicAdd(a.cand_count + (a.item_query ? a.item_query[item] : item), accepted);
}

// ---------------------------------------------------------------------------------------
// a5 / a-T merge + replay: one wave per active query.
//   Merges the query's partial lists down to its 2k smallest (distance, position) keys,
//   orders those by scan position and lets lane 0 replay the reference's guarded
//   insertion (updateTopK + "dist < maxDist", index_utils.c:19-33, freddy.c:128-131)
//   on top of the query's carried list.  Candidates outside the 2k smallest keys can
//   never influence the final list ("tie contract"), so this equals the
//   reference's sequential pass over every scanned row.
// ---------------------------------------------------------------------------------------
struct MergeArgs {
  const u64* part;             // [n_active][parts_per_query][L]
  const int32_t* active;       // [n_active] or NULL
  const int32_t* pos_to_id;    // NULL: position is the id; else id = pos_to_id[position]
  const int32_t* round_rows;   // [n_active] rows retrieved this round
  const int32_t* cand_count;   // [Q] accepted-candidate count this round
  int32_t* out_ids;            // [Q][k]  carried list (state) and final result
  float* out_dist;             // [Q][k]
  int32_t* found;              // [Q] accumulated over rounds
  int32_t* next_active;        // [Q] queries needing another round
  int32_t* n_next;             // [1]
  int32_t* status;             // [1] optional flag for the async API
  int n_active, parts_per_query, L, k, found_rule, first_round;
  float sentinel;
};

template <int V>
__global__ __launch_bounds__(64) void merge_replay_kernel(MergeArgs a) {
  __shared__ u64 cand[64 * V];
  __shared__ int32_t s_id[64 * V > 32 ? 64 * V : 32];
  __shared__ float s_d[64 * V > 32 ? 64 * V : 32];
  const int x = blockIdx.x;
  const int lane = threadIdx.x;
  const int q = a.active ? a.active[x] : x;
  const int k = a.k;

  u64 acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = KEY_INF;
  const u64* src = a.part + (size_t)x * a.parts_per_query * a.L;
  const int total = a.parts_per_query * a.L;
  u64 tau = KEY_INF;
  // (NB batches of 64 keys requested together: with one batch per trip the single wave of a one-query call paid a global
  // round trip for each of its 39 batches -- 245 parts of a 1 M-row table: 17 us)
  constexpr int NB = 8;
  for (int base0 = 0; base0 < total; base0 += 64 * NB) {
    u64 keys[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int i = base0 + u * 64 + lane;
      keys[u] = src[i < total ? i : total - 1];
      if (i >= total) keys[u] = KEY_INF;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      u64 key = keys[u];
      if (!(key < tau)) key = KEY_INF;
      if (__ballot(key != KEY_INF) == 0ull) continue;
      wave_topk_absorb<V>(acc, key);
      tau = wave_topk_at<V>(acc, a.L - 1);
    }
  }
  if (V == 1 && k <= 64 && a.first_round) {
    // First round, nothing carried: when no two of the L candidates are equally far, the order of the guarded insertions
    // (freddy.c:128-131) does not matter -- the list is the k nearest below the sentinel in ascending order, which is how
    // the accumulator already holds them.  (Equal distances: the replay in scan order below.)
    const u64 top = lane < a.L ? acc[0] : KEY_INF;
    const uint32_t db = (uint32_t)(top >> 32);
    const uint32_t db_next = (uint32_t)__shfl_down((int)db, 1, 64);
    const bool tie = lane + 1 < a.L && top != KEY_INF && db == db_next;
    if (__ballot(tie) == 0ull) {
      float d_slot = a.sentinel;
      int32_t id_slot = -1;
      if (lane < k && top != KEY_INF && __uint_as_float(db) < a.sentinel) {
        d_slot = __uint_as_float(db);
        id_slot = a.pos_to_id ? a.pos_to_id[(uint32_t)top] : (int32_t)(uint32_t)top;
      }
      if (lane < k) {
        a.out_ids[(size_t)q * k + lane] = id_slot;
        a.out_dist[(size_t)q * k + lane] = d_slot;
      }
      if (lane == 0) {
        int f = 0;
        const int rows = a.round_rows ? a.round_rows[x] : 0;
        f += (a.found_rule == 1 && a.cand_count) ? a.cand_count[q] : (rows > 0 ? rows : 0);
        if (a.found) a.found[q] = f;
        if (a.next_active && f < k && rows >= 0) {
          const int slot = atomicAdd(a.n_next, 1);
          a.next_active[slot] = q;
          if (a.status) a.status[0] = 1;
        }
      }
      return;
    }
  }
  // order the survivors by scan position: re-key as (position, distance bits) and sort
  u64 byp[V];
#pragma unroll
  for (int v = 0; v < V; ++v)
    byp[v] = (acc[v] == KEY_INF || v * 64 + lane >= a.L) ? KEY_INF : ((acc[v] << 32) | (acc[v] >> 32));
  wave_sort_full<V>(byp);
  auto bookkeeping = [&]() {   // "found" (freddy.c:377 rows rule, :971 accepted rule); one lane
    int f = a.first_round ? 0 : a.found[q];
    const int rows = a.round_rows ? a.round_rows[x] : 0;
    f += (a.found_rule == 1 && a.cand_count) ? a.cand_count[q] : (rows > 0 ? rows : 0);
    if (a.found) a.found[q] = f;
    if (a.next_active && f < k && rows >= 0) {   // rows < 0: no cell left, the query retires
      const int slot = atomicAdd(a.n_next, 1);
      a.next_active[slot] = q;
      if (a.status) a.status[0] = 1;
    }
  };
  if (V == 1 && k <= 64) {
    // lane i = slot i of the carried list; candidates replayed in scan order
    float d_slot = (a.first_round || lane >= k) ? a.sentinel : a.out_dist[(size_t)q * k + lane];
    int32_t id_slot = (a.first_round || lane >= k) ? -1 : a.out_ids[(size_t)q * k + lane];
    // (the candidates' ids gathered by all lanes at once: a load inside the replay is a round trip per insertion)
    u64 cand_id = byp[0];
    if (a.pos_to_id && cand_id != KEY_INF) cand_id = ((u64)(uint32_t)a.pos_to_id[(uint32_t)(cand_id >> 32)] << 32) | (u64)(uint32_t)cand_id;
    wave_list_replay(d_slot, id_slot, k, cand_id, a.L < 64 ? a.L : 64, [](uint32_t id) { return (int32_t)id; });
    if (lane < k) {
      a.out_ids[(size_t)q * k + lane] = id_slot;
      a.out_dist[(size_t)q * k + lane] = d_slot;
    }
    if (lane == 0) bookkeeping();
    return;
  }
#pragma unroll
  for (int v = 0; v < V; ++v) cand[v * 64 + lane] = byp[v];
  // carried list
  for (int i = lane; i < k; i += 64) {
    s_id[i] = a.first_round ? -1 : a.out_ids[(size_t)q * k + i];
    s_d[i] = a.first_round ? a.sentinel : a.out_dist[(size_t)q * k + i];
  }
  __syncthreads();
  if (lane == 0) {
    float maxd = s_d[k - 1];
    for (int e = 0; e < a.L; ++e) {
      const u64 c = cand[e];
      if (c == KEY_INF) break;
      const float dist = __uint_as_float((uint32_t)c);
      const uint32_t p = (uint32_t)(c >> 32);
      if (dist < maxd) {
        int slot = k - 1;
        while (slot >= 0 && !(s_d[slot] < dist)) --slot;
        ++slot;
        for (int t = k - 2; t >= slot; --t) { s_d[t + 1] = s_d[t]; s_id[t + 1] = s_id[t]; }
        s_d[slot] = dist;
        s_id[slot] = a.pos_to_id ? a.pos_to_id[p] : (int32_t)p;
        maxd = s_d[k - 1];
      }
    }
    bookkeeping();
  }
  __syncthreads();
  for (int i = lane; i < k; i += 64) {
    a.out_ids[(size_t)q * k + i] = s_id[i];
    a.out_dist[(size_t)q * k + i] = s_d[i];
  }
}

// ---------------------------------------------------------------------------------------
// grouping_pq (freddy.c:1176-1401): nearest of G group LUTs for every row.
//   Per row: for g = 0..G-1 the ADC sum over positions 0..m-1 in order (:1346-1351), nearest by strict
//   "<" starting from minDist = 100 (:1337,1353-1356) -> the first of equally near groups, -1 if none is
//   nearer than 100.  A workgroup owns GROUP_BLOCKS row blocks (lane <-> row, 4 rows per thread, their
//   code dwords stay in registers) and walks the groups, staging one LUT at a time in LDS.
// ---------------------------------------------------------------------------------------
static constexpr int GROUP_BLOCKS = 16;
template <int M2>   // dwords of codes per row; 0: read the codes from memory for every group (any m)
__global__ __launch_bounds__(WG) void grouping_kernel(const float* __restrict__ lut, int G, int m, int K,
                                                     const uint32_t* __restrict__ packed, int n_blocks,
                                                     int32_t* __restrict__ out_group) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sl = reinterpret_cast<float*>(smem);   // [m][K]
  constexpr int RPT = GROUP_BLOCKS * 64 / WG;   // rows per thread
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m2 = M2 ? M2 : m / 2;
  uint32_t cw[RPT][M2 ? M2 : 1];
  float best[RPT];
  int bi[RPT];
  int blk[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int b = blockIdx.x * GROUP_BLOCKS + r * (WG / 64) + wave;
    blk[r] = b < n_blocks ? b : n_blocks - 1;
    best[r] = 100.0f;
    bi[r] = -1;
    if (M2) {
#pragma unroll
      for (int j = 0; j < (M2 ? M2 : 1); ++j) cw[r][j] = packed[((size_t)blk[r] * m2 + j) * 64 + lane];
    }
  }
  for (int g = 0; g < G; ++g) {
    __syncthreads();
    for (int i = threadIdx.x; i < m * K; i += WG) sl[i] = lut[(size_t)g * m * K + i];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      float dist = 0.0f;
      if (M2) {
#pragma unroll
        for (int j = 0; j < (M2 ? M2 : 1); ++j) {
          const uint32_t w = cw[r][j];
          dist = dist + sl[(2 * j) * K + (int)(w & 0xffffu)];
          dist = dist + sl[(2 * j + 1) * K + (int)(w >> 16)];
        }
      } else {
        for (int j = 0; j < m2; ++j) {
          const uint32_t w = packed[((size_t)blk[r] * m2 + j) * 64 + lane];
          dist = dist + sl[(2 * j) * K + (int)(w & 0xffffu)];
          dist = dist + sl[(2 * j + 1) * K + (int)(w >> 16)];
        }
      }
      if (dist < best[r]) { best[r] = dist; bi[r] = g; }
    }
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int b = blockIdx.x * GROUP_BLOCKS + r * (WG / 64) + wave;
    if (b < n_blocks) out_group[(size_t)b * 64 + lane] = bi[r];
  }
}

// ---------------------------------------------------------------------------------------
// pq_search_in: gather the packed codes of a row subset into a temporary list
// ("SELECT id, vector FROM pq_quantization WHERE id IN (...)", freddy.c:1100-1114)
// ---------------------------------------------------------------------------------------
static __global__ __launch_bounds__(WG) void gather_rows_kernel(const int32_t* __restrict__ rows, int n_rows,
                                                        const uint32_t* __restrict__ packed,
                                                        uint32_t* __restrict__ packed_out,
                                                        int32_t* __restrict__ pos_out, int M2,
                                                        int n_out_padded) {
  const int i = blockIdx.x * WG + threadIdx.x;
  if (i >= n_out_padded) return;
  if (i >= n_rows) { pos_out[i] = -1; for (int j = 0; j < M2; ++j) packed_out[((size_t)(i >> 6) * M2 + j) * 64 + (i & 63)] = 0u; return; }
  const int r = rows[i];
  pos_out[i] = r;
  for (int j = 0; j < M2; ++j)
    packed_out[((size_t)(i >> 6) * M2 + j) * 64 + (i & 63)] = packed[((size_t)(r >> 6) * M2 + j) * 64 + (r & 63)];
}

}  // namespace freddy
