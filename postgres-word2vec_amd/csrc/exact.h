// exact.h -- exact brute-force kNN (the first "next row" after the PQ / IVFADC
// / kNN-join path): ORDER BY cosine_similarity_bytea(q, v) DESC FETCH FIRST k over all rows
// (k_nearest_neighbour, freddy--0.0.1.sql:426-454) or over "id = ANY(input_set)" (knn_in_exact,
// :991-1084).
//
// cosine_similarity_bytea (core_functions.c:67-81) is "scalar += v1[i] * v2[i]" in binary32 with
// PGXS default flags: a separately rounded multiply and add per dimension, i ascending.  The
// kernel keeps exactly that chain (v_pk_mul_f32 + v_pk_add_f32 over query pairs, each half an
// IEEE op), so similarities are bit-identical to the reference arithmetic.  (An fp32 MFMA would
// be ~5x faster but is an fma chain -- one rounding per term -- and would only match to ~1e-6;
// left as the documented alternative.)
//
// Layout: vectors are pinned in 64-row blocks xb[block][dim][64] so that lane <-> row reads are
// coalesced; the QT queries of a workgroup sit in LDS as [dim][QT] (broadcast ds_read_b128).
// Ordering: key = (~ordered(sim) << 32) | row, smallest key first = largest similarity first,
// equal similarities by ascending row (= ascending id).  PostgreSQL leaves that tie order
// unspecified; this is the pinned choice (oracle: fo_exact_knn).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wave_topk.h"

namespace freddy {

static constexpr int EX_WG = 256;
static constexpr int EX_WAVES = EX_WG / 64;
// queries per workgroup (every tile of queries streams the whole table once): 16 for batches while the
// selection state fits the registers (k <= 256), else 8
__host__ __device__ constexpr int ex_qt(int V, int Q) { return (V <= 4 && Q > 8) ? 16 : 8; }

__device__ __forceinline__ u64 sim_key(float sim, uint32_t row) {
  const uint32_t b = __float_as_uint(sim);
  const uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // ascending with the float
  return ((u64)(~ord) << 32) | (u64)row;                              // descending similarity
}
__device__ __forceinline__ float key_sim(u64 key) {
  const uint32_t ord = ~(uint32_t)(key >> 32);
  return __uint_as_float((ord & 0x80000000u) ? (ord ^ 0x80000000u) : ~ord);
}

// rows [N][d] -> xb[block][d][64] (zero padded); pos_out[i] = source row or -1
__global__ __launch_bounds__(256) void block_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                                        int64_t n_rows, float* __restrict__ xb, int32_t* __restrict__ pos_out,
                                                        int d) {
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);   // output row
  const int part = threadIdx.x >> 6;                                   // 4 dimension slices
  const int64_t r = (i < n_rows) ? (rows ? rows[i] : i) : -1;
  if (part == 0 && pos_out) pos_out[i] = (int32_t)r;
  for (int dim = part; dim < d; dim += 4)
    xb[((int64_t)blockIdx.x * d + dim) * 64 + (threadIdx.x & 63)] = (r >= 0) ? src[r * d + dim] : 0.0f;
}

struct ExactArgs {
  const float* xb;          // [blocks][d][64]
  const int32_t* pos;       // [blocks*64] row of each slot (-1 padding) or NULL: slot index, valid below n_rows
  const float* queries;     // [Q][d]
  u64* part;                // [Q][nchunk][EX_WAVES][L]
  int64_t n_rows;
  int n_blocks, chunk_blocks, nchunk, Q, d, L;
  const u64* floor;         // FLOOR instantiation (k > 1024: a later pass of 1024 keys): [Q] only keys ABOVE the query's floor are offered
};

// FLOOR: lists of more than 1024 entries are selected 1024 keys per pass over the same rows (the reference's ORDER BY ... FETCH FIRST k
// takes any k, freddy--0.0.1.sql:426-454): pass p admits only keys above the last key pass p - 1 selected.  Keys are unique (the row
// is part of them) and totally ordered (similarity DESC, row ASC), so the passes' lists simply follow each other.
template <int V, int EX_QT, bool FLOOR = false>
__global__ __launch_bounds__(EX_WG) void exact_scan_kernel(ExactArgs a) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  constexpr int QT = EX_QT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* qs = reinterpret_cast<float*>(smem);                                              // [d][QT]
  u64* stage = reinterpret_cast<u64*>(smem + (((size_t)a.d * QT * 4 + 15) & ~(size_t)15)); // [waves][QT][64]
  const int chunk = blockIdx.x, q0 = blockIdx.y * QT;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d = a.d;
  for (int i = threadIdx.x; i < d * QT; i += EX_WG) {
    const int t = i / d, dim = i - t * d;
    const int q = (q0 + t < a.Q) ? (q0 + t) : (a.Q - 1);
    qs[dim * QT + t] = a.queries[(size_t)q * d + dim];
  }
  __syncthreads();
  WaveSelect<V> sel[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) sel[t].init(stage + ((size_t)wave * QT + t) * 64, KEY_INF, a.L);

  const int b0 = chunk * a.chunk_blocks;
  const int b1 = (b0 + a.chunk_blocks < a.n_blocks) ? b0 + a.chunk_blocks : a.n_blocks;
  for (int b = b0 + wave; b < b1; b += EX_WAVES) {
    const float* xrow = a.xb + (size_t)b * d * 64 + lane;
    v2f acc[QT / 2];
#pragma unroll
    for (int t = 0; t < QT / 2; ++t) acc[t] = v2f{0.0f, 0.0f};
    constexpr int DB = 8;   // row values fetched DB dimensions ahead
    float xn[DB];
#pragma unroll
    for (int u = 0; u < DB; ++u) xn[u] = (u < d) ? xrow[(size_t)u * 64] : 0.0f;
    for (int i0 = 0; i0 < d; i0 += DB) {
      float xc[DB];
#pragma unroll
      for (int u = 0; u < DB; ++u) xc[u] = xn[u];
#pragma unroll
      for (int u = 0; u < DB; ++u) xn[u] = (i0 + DB + u < d) ? xrow[(size_t)(i0 + DB + u) * 64] : 0.0f;
#pragma unroll
      for (int u = 0; u < DB; ++u) {
        const int i = i0 + u;
        if (i < d) {
          const float4* qrow = reinterpret_cast<const float4*>(qs + i * QT);
          const v2f x2 = {xc[u], xc[u]};
#pragma unroll
          for (int t4 = 0; t4 < QT / 4; ++t4) {
            const float4 qv = qrow[t4];
            const v2f qa = {qv.x, qv.y}, qb = {qv.z, qv.w};
            const v2f pa = qa * x2, pb = qb * x2;          // core_functions.c:77: v1[i] * v2[i]
            acc[t4 * 2 + 0] = acc[t4 * 2 + 0] + pa;        //                     scalar += ...
            acc[t4 * 2 + 1] = acc[t4 * 2 + 1] + pb;
          }
        }
      }
    }
    const int64_t slot = (int64_t)b * 64 + lane;
    const int32_t row = a.pos ? a.pos[slot] : (slot < a.n_rows ? (int32_t)slot : -1);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const float sim = (t & 1) ? acc[t >> 1].y : acc[t >> 1].x;
      const u64 key = sim_key(sim, (uint32_t)row);
      bool ok = row >= 0;
      if constexpr (FLOOR) ok = ok && key > a.floor[(q0 + t < a.Q) ? (q0 + t) : (a.Q - 1)];
      sel[t].push(key, ok);
    }
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    sel[t].finish();
    if (q0 + t < a.Q) {
      u64* out = a.part + (((size_t)(q0 + t) * a.nchunk + chunk) * EX_WAVES + wave) * a.L;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int r = v * 64 + lane;
        if (r < a.L) out[r] = sel[t].acc[v];
      }
    }
  }
}

// One workgroup of EX_MW waves per query: every wave selects the L best keys of its slice of the
// partial lists, wave 0 merges the EX_MW lists and emits (id, similarity) in order.  (A single wave
// walking all lists took 0.8 ms for one query over 3 M rows: 23 440 lists.)
static constexpr int EX_MW = 16;   // at most; the launch uses fewer waves for wide lists (LDS)
template <int V>
__global__ __launch_bounds__(EX_MW * 64) void exact_merge_kernel(const u64* __restrict__ part, int parts_per_query, int L, int k,
                                                                const int32_t* __restrict__ ids, int32_t* __restrict__ out_ids,
                                                                float* __restrict__ out_sim, int out_off = 0, int out_n = -1,
                                                                u64* __restrict__ floor_out = nullptr) {
  // (out_off, out_n, floor_out: a pass of a list of more than 1024 entries -- ranks [0, out_n) go to places out_off.. of the
  // query's k, and the last selected key becomes the next pass's floor; KEY_INF when the rows ran out: nothing is above it)
  if (out_n < 0) out_n = k;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int MW = blockDim.x >> 6;
  u64* stage = reinterpret_cast<u64*>(smem);        // [MW][64]
  u64* lists = stage + MW * 64;                      // [MW][64*V]
  const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveSelect<V> sel;
  sel.init(stage + wave * 64, KEY_INF, L);
  const u64* src = part + (size_t)q * parts_per_query * L;
  const long long total = (long long)parts_per_query * L;
  const long long per = ((total + MW - 1) / MW + 63) / 64 * 64;
  const long long lo = per * wave, hi = (lo + per < total) ? lo + per : total;
  for (long long base = lo; base < hi; base += 64) {
    const bool valid = base + lane < hi;
    const u64 key = valid ? src[base + lane] : KEY_INF;
    sel.push(key, valid && key != KEY_INF);
  }
  sel.finish();
#pragma unroll
  for (int v = 0; v < V; ++v) lists[(size_t)wave * 64 * V + v * 64 + lane] = sel.acc[v];
  __syncthreads();
  if (wave != 0) return;
  WaveSelect<V> fin;
  fin.init(stage, KEY_INF, L);
  for (int w = 0; w < MW; ++w)
    for (int base = 0; base < L; base += 64) {
      const bool valid = base + lane < L;
      const u64 key = valid ? lists[(size_t)w * 64 * V + base + lane] : KEY_INF;
      fin.push(key, valid && key != KEY_INF);
    }
  fin.finish();
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int r = v * 64 + lane;
    if (r < out_n) {
      const u64 key = fin.acc[v];
      out_ids[(size_t)q * k + out_off + r] = (key == KEY_INF) ? -1 : ids[key_pos(key)];
      out_sim[(size_t)q * k + out_off + r] = (key == KEY_INF) ? -__builtin_huge_valf() : key_sim(key);
    }
    if (floor_out && r == L - 1) floor_out[q] = fin.acc[v];
  }
}

}  // namespace freddy
