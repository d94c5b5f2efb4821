// fused3.h -- IVFADC LUT build + ADC scan + selection, role-specialised with TWO builder waves per SIMD.
//
// Same scheme as fused2.h (ivf_spec2_kernel: builders write slab(p+1) while gatherers read slab(p), one
// LDS-only barrier per position, next entry's descriptor / residuals / slab 0 prepared in the shadow of
// the current one; same slab arithmetic, selection and outputs).  What changes is the wave budget.  A
// SIMD issues packed fp32 ~27 % faster from two waves than from one (2.31 vs 3.15 ns per
// instruction), and while one builder wave is blocked issuing its codebook loads the other one
// computes.  16 waves per workgroup, four per SIMD, 128 VGPRs each:
//   waves 0-7   BUILDERS   two per SIMD; lane <-> 2 codes (one packed pair, 50 codebook VGPRs); four
//                          items at a time = four independent packed chains (two for a last item pair)
//   waves 8-15  GATHERERS  lane <-> 8 rows x 12 items ADC sums (96 VGPRs)
// 128 registers only hold the sums of TWELVE items, so a work entry here is <= 12 items of one cell
// (the work table is built with that group size); slab rows are [code][12 items] = 48 bytes, three
// ds_read_b128 per row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_common.h"

namespace freddy {

static constexpr int SPEC2_T = 1024;
static constexpr int SPEC2_NB = 8;    // builder waves
static constexpr int SPEC2_NG = 8;    // gatherer waves (== FUSED_NW: survivor regions are per gatherer wave)
static constexpr int SPEC2_G = 12;    // items per work entry
static_assert(SPEC2_NG == FUSED_NW, "survivor region layout");

template <int S, int M, bool FULLK>
__global__ __launch_bounds__(SPEC2_T) void ivf_spec2_kernel(FusedArgs a) {
  constexpr int G = SPEC2_G, RMAX = FUSED_RMAX, NG = SPEC2_NG;
  constexpr int M2 = M / 2;
  constexpr int SP = (S + 3) & ~3;
  constexpr int SPq = SP / 4;
  static_assert(M % 2 == 0 && G % 4 == 0 && G <= 16, "layout");
  typedef float v2f __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);                                   // [2][K][G]
  uint32_t* colmin = reinterpret_cast<uint32_t*>(smem + a.desc_offset);           // [16][64]
  uint32_t* tau_s = colmin + 16 * 64;                                             // [16]
  int32_t* dsc = reinterpret_cast<int32_t*>(smem + a.desc_offset + 4096 + 64);    // see fused.h / fused2.h
  float* res = reinterpret_cast<float*>(smem + a.desc_offset + 4096 + 64 + 512);  // [G][M][SP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool builder = wave < SPEC2_NB;
  const int K = a.K;
  const int n_work = a.n_groups[0];
  auto slab_at = [&](int code, int g) { return code * G + g; };

  // Residuals r = q - coarse[cell] (freddy.c:296-303: one binary32 subtraction per dimension) are formed
  // while staging: thread <-> (position, dimension) slot of the padded [M][SP] row, the cell's centroid
  // value is loaded once and serves all items of the entry.  nthr threads take part (t = 0..nthr-1).
  auto stage_residuals = [&](int b, int cnt, int t, int nthr, int p_lo, int p_hi) {   // positions [p_lo, p_hi)
    const int cell = dsc[32 + b * 8 + 5];
    const float* crow = a.coarse + (size_t)cell * a.d;
    for (int o = p_lo * SP + t; o < p_hi * SP; o += nthr) {
      const int p = o / SP, j = o - p * SP;
      if (j < S) {
        const float cv = crow[p * S + j];
        for (int g = 0; g < cnt; ++g) res[(size_t)g * (M * SP) + o] = a.queries[(size_t)dsc[64 + b * 16 + g] * a.d + p * S + j] - cv;
      } else {
        for (int g = 0; g < cnt; ++g) res[(size_t)g * (M * SP) + o] = 0.0f;
      }
    }
  };
  // The NEXT entry's residual rows are staged by the gatherers, which have time to spare inside the
  // main loop: the rows of position q are dead as soon as slab(q) of the current entry is built (in
  // P(q-1)), so from P(STAGE_P0) on -- the next descriptor is complete by then -- the gatherers fill
  // positions <= p during P(p); only the last position is left to the builders in P(M-1).
  constexpr int STAGE_P0 = 8;
  static_assert(M - 2 >= STAGE_P0, "descriptor prefetch schedule assumes M >= 10");

  // ---- first entry: fetched serially by everybody ----
  int cur = 0;
  if (tid == 0) dsc[32] = atomicAdd(a.work_counter, 1);
  for (int i = tid; i < 16 * 64; i += SPEC2_T) colmin[i] = 0xffffffffu;
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
        dsc[lane] = it;
        dsc[64 + lane] = it >= 0 ? a.item_query[it] : 0;
      }
      if (lane == 0) { dsc[33] = cnt0; dsc[34] = b0; dsc[35] = nb0; dsc[36] = chunk0; dsc[37] = cell; }
    }
  }
  __syncthreads();
  stage_residuals(0, dsc[33], tid, SPEC2_T, 0, M);
  __syncthreads();

  if (builder) {
    // =====================================================================================
    // BUILDERS
    // =====================================================================================
    const int b = tid;   // 0..511: codes b and b+512 (one packed pair)
    v2f cb[S];
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const f4 __attribute__((address_space(1))) * gptr4;
    typedef const char __attribute__((address_space(1))) * gptrc;
    // cbP layout [m][SP/4][512 slots][4 dims][2 codes]: slot i holds codes (i, i+512) interleaved
    auto load_cb = [&](int p) {
      const gptrc base = (gptrc)(uintptr_t)a.cbP + (size_t)(uint32_t)p * (uint32_t)(SPq * 512 * 32);
      uint32_t voff = (uint32_t)b * 32u;
      asm volatile("" : "+v"(voff));   // opaque: keeps hoisted 64-bit addresses out of the register budget
#pragma unroll
      for (int jb = 0; jb < SPq; ++jb) {
        const gptrc bj = base + (uint32_t)jb * (uint32_t)(512 * 32);
        const f4 lo = *(gptr4)(bj + voff), hi = *(gptr4)(bj + voff + 16u);
        if (jb * 4 + 0 < S) cb[jb * 4 + 0] = v2f{lo.x, lo.y};
        if (jb * 4 + 1 < S) cb[jb * 4 + 1] = v2f{lo.z, lo.w};
        if (jb * 4 + 2 < S) cb[jb * 4 + 2] = v2f{hi.x, hi.y};
        if (jb * 4 + 3 < S) cb[jb * 4 + 3] = v2f{hi.z, hi.w};
      }
    };
    // slab(p) of the items [g_lo, g_hi) (g_lo a multiple of 4).  Four items per step as four interleaved
    // chains -- twelve packed instructions per dimension in the order sub x4, mul x4, add x4, operands
    // four issues apart -- and two chains for a last pair.  a + (-b) with the neg modifier is the IEEE
    // subtraction; each half rounds like the scalar op; dimensions in order (index_utils.c:500-508).
    auto build_slab = [&](int p, float* dst, int g_lo, int g_hi, int cnt) {
      int g = g_lo;
#pragma unroll 1
      for (; g + 2 < g_hi; g += 4) {   // at least three items left: a block of four (a missing 4th repeats the 3rd)
        const int gl = cnt - 1;
        const float4* R0 = reinterpret_cast<const float4*>(res + ((size_t)g * M + p) * SP);
        const float4* R1 = reinterpret_cast<const float4*>(res + ((size_t)(g + 1) * M + p) * SP);
        const float4* R2 = reinterpret_cast<const float4*>(res + ((size_t)(g + 2) * M + p) * SP);
        const float4* R3 = reinterpret_cast<const float4*>(res + ((size_t)(g + 3 < gl ? g + 3 : gl) * M + p) * SP);
        v2f s0 = {0.0f, 0.0f}, s1 = {0.0f, 0.0f}, s2 = {0.0f, 0.0f}, s3 = {0.0f, 0.0f};
        float4 n0 = R0[0], n1 = R1[0], n2 = R2[0], n3 = R3[0];
#pragma unroll
        for (int jb = 0; jb < SPq; ++jb) {
          const float4 c0 = n0, c1 = n1, c2 = n2, c3 = n3;
          if (jb + 1 < SPq) { n0 = R0[jb + 1]; n1 = R1[jb + 1]; n2 = R2[jb + 1]; n3 = R3[jb + 1]; }
          const v2f a0[2] = {{c0.x, c0.y}, {c0.z, c0.w}};
          const v2f a1[2] = {{c1.x, c1.y}, {c1.z, c1.w}};
          const v2f a2[2] = {{c2.x, c2.y}, {c2.z, c2.w}};
          const v2f a3[2] = {{c3.x, c3.y}, {c3.z, c3.w}};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = jb * 4 + u;
            if (j < S) {
              v2f t0, t1, t2, t3;
              if ((u & 1) == 0) {
                asm volatile(
                    "v_pk_add_f32 %4, %8, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %5, %9, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %6, %10, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %7, %11, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %4, %4, %4\n\t"
                    "v_pk_mul_f32 %5, %5, %5\n\t"
                    "v_pk_mul_f32 %6, %6, %6\n\t"
                    "v_pk_mul_f32 %7, %7, %7\n\t"
                    "v_pk_add_f32 %0, %0, %4\n\t"
                    "v_pk_add_f32 %1, %1, %5\n\t"
                    "v_pk_add_f32 %2, %2, %6\n\t"
                    "v_pk_add_f32 %3, %3, %7"
                    : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                    : "v"(a0[u >> 1]), "v"(a1[u >> 1]), "v"(a2[u >> 1]), "v"(a3[u >> 1]), "v"(cb[j]));
              } else {
                asm volatile(
                    "v_pk_add_f32 %4, %8, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %5, %9, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %6, %10, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %7, %11, %12 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %4, %4, %4\n\t"
                    "v_pk_mul_f32 %5, %5, %5\n\t"
                    "v_pk_mul_f32 %6, %6, %6\n\t"
                    "v_pk_mul_f32 %7, %7, %7\n\t"
                    "v_pk_add_f32 %0, %0, %4\n\t"
                    "v_pk_add_f32 %1, %1, %5\n\t"
                    "v_pk_add_f32 %2, %2, %6\n\t"
                    "v_pk_add_f32 %3, %3, %7"
                    : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                    : "v"(a0[u >> 1]), "v"(a1[u >> 1]), "v"(a2[u >> 1]), "v"(a3[u >> 1]), "v"(cb[j]));
              }
            }
          }
        }
        // slab rows are [code][12 items]: the block's four items are one aligned 16-byte store per code
        // (unused slots receive values nobody reads)
        if (FULLK || b < K) *reinterpret_cast<float4*>(dst + slab_at(b, g)) = float4{s0.x, s1.x, s2.x, s3.x};
        if (FULLK || b + 512 < K) *reinterpret_cast<float4*>(dst + slab_at(b + 512, g)) = float4{s0.y, s1.y, s2.y, s3.y};
      }
#pragma unroll 1
      for (; g < g_hi; g += 2) {       // one or two items left: two chains
        const float4* R0 = reinterpret_cast<const float4*>(res + ((size_t)g * M + p) * SP);
        const float4* R1 = reinterpret_cast<const float4*>(res + ((size_t)(g + 1 < cnt ? g + 1 : g) * M + p) * SP);
        v2f s0 = {0.0f, 0.0f}, s1 = {0.0f, 0.0f};
        float4 n0 = R0[0], n1 = R1[0];
#pragma unroll
        for (int jb = 0; jb < SPq; ++jb) {
          const float4 c0 = n0, c1 = n1;
          if (jb + 1 < SPq) { n0 = R0[jb + 1]; n1 = R1[jb + 1]; }
          const v2f a0[2] = {{c0.x, c0.y}, {c0.z, c0.w}};
          const v2f a1[2] = {{c1.x, c1.y}, {c1.z, c1.w}};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = jb * 4 + u;
            if (j < S) {
              v2f t0, t1;
              if ((u & 1) == 0) {
                asm volatile(
                    "v_pk_add_f32 %2, %4, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %3, %5, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %2, %2, %2\n\t"
                    "v_pk_mul_f32 %3, %3, %3\n\t"
                    "v_pk_add_f32 %0, %0, %2\n\t"
                    "v_pk_add_f32 %1, %1, %3"
                    : "+v"(s0), "+v"(s1), "=&v"(t0), "=&v"(t1)
                    : "v"(a0[u >> 1]), "v"(a1[u >> 1]), "v"(cb[j]));
              } else {
                asm volatile(
                    "v_pk_add_f32 %2, %4, %6 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %3, %5, %6 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_mul_f32 %2, %2, %2\n\t"
                    "v_pk_mul_f32 %3, %3, %3\n\t"
                    "v_pk_add_f32 %0, %0, %2\n\t"
                    "v_pk_add_f32 %1, %1, %3"
                    : "+v"(s0), "+v"(s1), "=&v"(t0), "=&v"(t1)
                    : "v"(a0[u >> 1]), "v"(a1[u >> 1]), "v"(cb[j]));
              }
            }
          }
        }
        if (FULLK || b < K) *reinterpret_cast<v2f*>(dst + slab_at(b, g)) = v2f{s0.x, s1.x};
        if (FULLK || b + 512 < K) *reinterpret_cast<v2f*>(dst + slab_at(b + 512, g)) = v2f{s0.y, s1.y};
      }
    };

    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc = 0;
    auto tick = [&](int slot) { if (a.prof) { const long long t = clock64(); pt[slot] += t - pc; pc = t; } };
    if (a.prof) pc = clock64();
    // slab(0) of the first entry, unoverlapped
    int cnt = __builtin_amdgcn_readfirstlane(dsc[33]);
    load_cb(0);
    build_slab(0, slab, 0, cnt, cnt);
    load_cb(1);
    lds_barrier();
    for (;;) {
      int ngid = 0;       // wave 0: next work index (requested now, used from P(2) on)
      int n_cell = 0, n_first = 0, n_gc = 0, n_item = -1, n_b0 = 0, n_b1 = 0, n_q = 0;
      const int nb = cur ^ 1;
      if (tid == 0) ngid = atomicAdd(a.work_counter, 1);
      for (int p = 0; p + 1 < M; ++p) {
        float* nxt = slab + (size_t)((p + 1) & 1) * G * K;
        tick(5);
        build_slab(p + 1, nxt, 0, cnt, cnt);
        tick(0);   // builds of the main loop
        __builtin_amdgcn_sched_barrier(0);
        // (Issued in one go.  The builder wave blocks ~2.8k cycles per position while the 28 wide loads
        // enter the memory pipe.  Two alternatives were measured and were no faster, the stall just moved
        // into the build: spreading the loads over the last item pair's pass, and building every slab in
        // two passes over the dimensions with the codebook registers refilled half by half.)
        load_cb(p + 2 < M ? p + 2 : 0);   // position 0: the next entry's
        // next entry's descriptor, one dependent global round trip per position (wave 0 only)
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
              if (n_item >= 0) n_q = a.item_query[n_item];   // (published at p == 7)
              if (lane < G) dsc[nb * 16 + lane] = n_item;
              if (lane == 0) {
                dsc[32 + nb * 8 + 1] = cntn; dsc[32 + nb * 8 + 2] = b0;
                dsc[32 + nb * 8 + 3] = nbn; dsc[32 + nb * 8 + 4] = chn; dsc[32 + nb * 8 + 5] = n_cell;
              }
            }
          } else if (p == 7) {
            if (lane < G) dsc[64 + nb * 16 + lane] = n_q;
            if (lane == 0) dsc[32 + nb * 8 + 0] = (ngid < n_work) ? ngid : -1;
          }
        }
        tick(5);
        lds_barrier();
        tick(1);   // waiting at the main loop's barriers
      }
      // P(M-1): the residual table is free (slab(M-1) was built in P(M-2)): stage the next entry's
      const int next_gid = __builtin_amdgcn_readfirstlane(dsc[32 + nb * 8]);
      const int next_cnt = next_gid >= 0 ? __builtin_amdgcn_readfirstlane(dsc[32 + nb * 8 + 1]) : 0;
      stage_residuals(nb, next_cnt, tid, SPEC2_NB * 64, M - 1, M);
      // ... and slab(0) of the next entry, one block of four items per phase: P(M-1), S1, S2.  Buffer 0 is
      // free (the gatherers read buffer 1, then select), position 0 of the next residuals was staged in
      // P(STAGE_P0), position 0 of the codebook was requested after the last build.
      const int c1 = next_cnt < 4 ? next_cnt : 4, c2 = next_cnt < 8 ? next_cnt : 8;
      build_slab(0, slab, 0, c1, next_cnt);
      lds_barrier();
      tick(2);   // P(M-1)
      build_slab(0, slab, c1, c2, next_cnt);
      lds_barrier();
      build_slab(0, slab, c2, next_cnt, next_cnt);
      load_cb(1);
      lds_barrier();
      tick(3);   // S1 + S2
      pt[7] += 1;
      if (next_gid < 0) break;
      cur = nb;
      cnt = next_cnt;
    }
    if (a.prof && tid == 0) {
      for (int i = 0; i < 8; ++i) a.prof[(size_t)blockIdx.x * 8 + i] = pt[i];
      a.prof[(size_t)blockIdx.x * 8 + 6] = wall_clock64();   // 100 MHz, chip-wide: when this workgroup ran dry
    }
  } else {
    // =====================================================================================
    // GATHERERS
    // =====================================================================================
    const int gw = wave - SPEC2_NB;   // 0..7
    v2f acc[G / 2][RMAX];            // ADC sums: acc[h][r] = items (2h, 2h+1) of this lane's row r (96 VGPRs)
    uint32_t cw[RMAX];
    auto bits = [&](int g, int r) { return __float_as_uint((g & 1) ? acc[g >> 1][r].y : acc[g >> 1][r].x); };
    lds_barrier();   // (pairs with the builders' barrier after the first slab)
    for (;;) {
      const int32_t* desc = dsc + cur * 16;
      const int cnt = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 1]);
      const int blk0 = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 2]);
      const int nblk = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 3]);
      const int chunk = __builtin_amdgcn_readfirstlane(dsc[32 + cur * 8 + 4]);
      const int nb = cur ^ 1;
      // rows past the end of the chunk re-read its last block (always in bounds); masked at the end
      auto row_block = [&](int r) {
        const int bl = r * NG + gw;
        return (uint32_t)(blk0 + (bl < nblk - 1 ? bl : nblk - 1));
      };
      auto load_codes = [&](int pair) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) cw[r] = a.packed[(row_block(r) * M2 + (uint32_t)pair) * 64u + (uint32_t)lane];
      };
      // one row at a time: G/4 ds_read_b128 fetch a row's item values
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
      for (int h = 0; h < G / 2; ++h)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[h][r] = v2f{0.0f, 0.0f};
      load_codes(0);
      for (int p = 0; p + 1 < M; ++p) {
        gather(p, slab + (size_t)(p & 1) * G * K);
        __builtin_amdgcn_sched_barrier(0);
        if (p & 1) load_codes((p + 1) >> 1);
        if (p >= STAGE_P0) {
          const int ngid_s = __builtin_amdgcn_readfirstlane(dsc[32 + nb * 8]);
          if (ngid_s >= 0) {
            const int ncnt_s = __builtin_amdgcn_readfirstlane(dsc[32 + nb * 8 + 1]);
            stage_residuals(nb, ncnt_s, tid - SPEC2_NB * 64, NG * 64, p == STAGE_P0 ? 0 : p, p + 1);
          }
        }
        lds_barrier();
      }
      // P(M-1)
      gather(M - 1, slab + (size_t)((M - 1) & 1) * G * K);
      int32_t pid[RMAX];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) pid[r] = a.pos[row_block(r) * 64u + (uint32_t)lane];
      // Selection on the distance bits (>= +0, so they order like the floats); see fused.h.  Column
      // minima (column = lane index over the 8 gatherer waves x 8 row slots) via LDS atomics.
      {
        bool dead[RMAX];
        bool some = false;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) { dead[r] = !(((r * NG + gw) < nblk) && pid[r] >= 0); some |= dead[r]; }
        if (__ballot(some) != 0ull) {   // only the last chunk of a list has padding rows: park them above everything
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
#pragma unroll
            for (int h = 0; h < G / 2; ++h)
              if (dead[r]) acc[h][r] = v2f{__uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu)};
        }
      }
      {
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
      // S1: thresholds, two items per gatherer wave, the two sorts interleaved
      {
        static_assert(G <= 2 * NG, "at most two items per gatherer wave");
        const int g0 = gw, g1 = gw + NG;   // (colmin / tau_s have 16 rows: g1 may be an unused one)
        uint32_t c0 = colmin[g0 * 64 + lane], c1 = colmin[g1 * 64 + lane];
        wave_sort32_x2(c0, c1);
        const uint32_t t0 = __shfl(c0, a.L - 1, 64), t1 = __shfl(c1, a.L - 1, 64);
        // survivors are {bits <= tau and bits < sentinel}: fold both into one bound
        if (lane == 0) {
          tau_s[g0] = min(t0, a.sentinel_bits - 1u);
          tau_s[g1] = min(t1, a.sentinel_bits - 1u);
        }
        colmin[g0 * 64 + lane] = 0xffffffffu;   // ready for the next entry
        colmin[g1 * 64 + lane] = 0xffffffffu;
      }
      lds_barrier();
      // S2: survivors -> this wave's region of each item's buffer
      {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g < cnt) {
            const uint32_t tau = (uint32_t)__builtin_amdgcn_readfirstlane((int)tau_s[g]);
            const int it = __builtin_amdgcn_readfirstlane(desc[g]);
            const size_t region = ((size_t)it * a.upi + chunk) * NG + gw;
            u64* dst = a.surv + region * (size_t)(RMAX * 64);
            if (a.cand_count) {   // freddy.c:971 counts the rows that pass the sentinel guard
              int accepted = 0;
#pragma unroll
              for (int r = 0; r < RMAX; ++r) accepted += __popcll(__ballot(bits(g, r) < a.sentinel_bits));
              if (lane == 0 && accepted) atomicAdd(a.cand_count + a.item_query[it], accepted);
            }
            int run = 0;
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
              const bool pass = bits(g, r) <= tau;
              const u64 mask = __ballot(pass);
              if (mask != 0ull) {
                if (pass) dst[run + lanes_below(mask)] = ((u64)bits(g, r) << 32) | (u64)(uint32_t)pid[r];
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

}  // namespace freddy
