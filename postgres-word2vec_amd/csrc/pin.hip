// pin.hip -- pin_pq / pin_ivf / pin_ivf_multi: the pinned tables' layouts in HBM; append_rows / update_codebook.
#include "internal.h"

#include "kernels.h"
#include "scan_common.h"
#include "coarse.h"   // fragment layouts of the centroids; refine.h: row_term_kernel

static std::vector<float> transpose_codebook(const float* cb, int m, int K, int S) {
  std::vector<float> t((size_t)m * S * K);
  for (int p = 0; p < m; ++p)
    for (int c = 0; c < K; ++c)
      for (int j = 0; j < S; ++j) t[((size_t)p * S + j) * K + c] = cb[((size_t)p * K + c) * S + j];
  return t;
}

// Pack rows of `n_lists` inverted lists into 64-row blocks: [block][M2][64] dwords, two
// int16 codes per dword, plus one scan-position dword per row (-1 on padding rows).
// Which rows share a 16-lane group of a 64-row block decides what the scan kernels' LDS gathers cost: a
// wave-level ds_read_b128 of slab rows takes ~2.4 + 4 x (largest number of lanes of a 16-lane group whose
// rows' codes agree modulo 16 = the same LDS bank group) cycles (tools/lab/ubench6: 14.6 cycles for random
// rows, 6.4 without collisions).  The order of the rows inside a list is free (results are ordered by id
// in the merge), so the rows of every group are picked greedily -- each next row from a window of 64
// candidates, the one that raises the per-position maxima least -- which brings the average maximum
// from 3.06 to ~2.1.  order[] = the list's rows in packing order.
static void arrange_list_rows(const int16_t* codes, int m, int64_t lo, int64_t hi, std::vector<int64_t>& order) {
  const int64_t n = hi - lo;
  order.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) order[(size_t)i] = lo + i;
  if (n <= 16 || m > 16) return;
  static const int WINDOW = (int)env_int("FREDDY_GPU_ARRANGE_WINDOW", 1024);   // candidates looked at for every pick (64: scan 103 us, 256: 101.7, 1024: 99.8; pin time 0.2 / 0.4 / 1.3 s for 3 M rows)
  int cnt[16][16], mx[16];
  for (int64_t k = 0; k < n; ++k) {
    if ((k & 15) == 0) { memset(cnt, 0, sizeof(cnt)); memset(mx, 0, sizeof(mx)); }
    const int64_t wend = std::min<int64_t>(n, k + WINDOW);
    int64_t best = k;
    int best_cost = INT32_MAX;
    for (int64_t j = k; j < wend; ++j) {
      const int16_t* row = codes + (size_t)order[(size_t)j] * m;
      int cost = 0;
      for (int p = 0; p < m; ++p) {
        const int c = cnt[p][row[p] & 15];
        cost += c + (c + 1 > mx[p] ? 100 : 0);
      }
      if (cost < best_cost) { best_cost = cost; best = j; }
    }
    std::swap(order[(size_t)k], order[(size_t)best]);
    const int16_t* row = codes + (size_t)order[(size_t)k] * m;
    for (int p = 0; p < m; ++p) {
      const int c = ++cnt[p][row[p] & 15];
      if (c > mx[p]) mx[p] = c;
    }
  }
  // The 16 rows picked together have to sit in the 16 lanes the LDS serves together -- and for ds_read_b128
  // those are NOT 16 consecutive lanes but {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
  // (MI355X_MICROARCH.md, LDS).  (Round 1 placed each group in consecutive lanes: every hardware group then
  // mixed the halves of two picked groups, and the arrangement bought 2 % instead of what tools/lab/ubench6 promised.)
  static const int GROUP_LANES[64] = {0,  1,  2,  3,  12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27, 4,  5,  6,  7,  8,  9,
                                      10, 11, 16, 17, 18, 19, 28, 29, 30, 31, 32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55,
                                      56, 57, 58, 59, 36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63};
  std::vector<int64_t> blk(64);
  for (int64_t b0 = 0; b0 + 64 <= n; b0 += 64) {   // (a partial last block keeps its rows in the first lanes)
    for (int j = 0; j < 64; ++j) blk[(size_t)GROUP_LANES[j]] = order[(size_t)(b0 + j)];
    for (int j = 0; j < 64; ++j) order[(size_t)(b0 + j)] = blk[(size_t)j];
  }
}

static int pack_lists(freddy_gpu_index* ix, int n_lists, const int32_t* list_off, const int16_t* codes,
                      const int32_t* row_pos /*NULL: row index*/) {
  const int m = ix->m, K = ix->K, M2 = ix->M2;
  std::vector<int32_t> blk_off(n_lists + 1, 0);
  int max_blocks = 0;
  for (int c = 0; c < n_lists; ++c) {
    const int64_t len = (int64_t)list_off[c + 1] - list_off[c];
    if (len < 0) return fail(FREDDY_E_ARG, "list_off is not non-decreasing at list %d", c);
    const int nb = (int)((len + 63) / 64);
    blk_off[c + 1] = blk_off[c] + nb;
    max_blocks = std::max(max_blocks, nb);
  }
  const int64_t n_blocks = blk_off[n_lists];
  std::vector<uint32_t> packed((size_t)std::max<int64_t>(n_blocks, 1) * M2 * 64, 0u);
  std::vector<int32_t> pos((size_t)std::max<int64_t>(n_blocks, 1) * 64, -1);
  // (inverted lists only: the flat PQ table is addressed by row index)
  const bool arrange = row_pos != nullptr;
  std::vector<std::vector<int64_t>> orders(arrange ? (size_t)n_lists : 0);
  if (arrange) {
    std::atomic<int> next_list{0};
    auto worker = [&]() {
      for (int c = next_list.fetch_add(1); c < n_lists; c = next_list.fetch_add(1))
        arrange_list_rows(codes, m, list_off[c], list_off[c + 1], orders[(size_t)c]);
    };
    const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt && (int)t < n_lists; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  for (int c = 0; c < n_lists; ++c) {
    for (int64_t i = 0; i < (int64_t)list_off[c + 1] - list_off[c]; ++i) {
      const int64_t r = arrange ? orders[(size_t)c][(size_t)i] : list_off[c] + i;
      const int64_t b = blk_off[c] + i / 64;
      const int lane = (int)(i % 64);
      const int16_t* row = codes + (size_t)r * m;
      for (int l = 0; l < m; ++l) {
        if (row[l] < 0 || row[l] >= K)
          return fail(FREDDY_E_ARG, "code %d at row %lld position %d is outside [0,%d)", (int)row[l],
                      (long long)r, l, K);
      }
      for (int j = 0; j < M2; ++j) {
        const uint32_t lo = (uint16_t)row[2 * j];
        const uint32_t hi = (2 * j + 1 < m) ? (uint16_t)row[2 * j + 1] : 0u;
        packed[((size_t)b * M2 + j) * 64 + lane] = lo | (hi << 16);
      }
      pos[(size_t)b * 64 + lane] = row_pos ? row_pos[r] : (int32_t)r;
    }
  }
  std::vector<int32_t> blk_cell((size_t)std::max<int64_t>(n_blocks, 1), 0);
  for (int c = 0; c < n_lists; ++c)
    for (int b = blk_off[c]; b < blk_off[c + 1]; ++b) blk_cell[(size_t)b] = c;
  if (upload(&ix->blk_cell, blk_cell.data(), blk_cell.size(), &ix->bytes))
    return fail(FREDDY_E_NOMEM, "device allocation/copy failed while pinning the lists");
  ix->n_blocks = n_blocks;
  ix->max_list_blocks = max_blocks;
  ix->h_list_off.assign(list_off, list_off + n_lists + 1);
  if (upload(&ix->blk_off, blk_off.data(), blk_off.size(), &ix->bytes) ||
      upload(&ix->list_off, list_off, (size_t)n_lists + 1, &ix->bytes) ||
      upload(&ix->packed, packed.data(), packed.size(), &ix->bytes) ||
      upload(&ix->pos, pos.data(), pos.size(), &ix->bytes))
    return fail(FREDDY_E_NOMEM, "device allocation/copy failed while pinning the lists");
  return 0;
}

// packed[block][6][64] (two int16 codes per dword) -> packed8[block][3][64] (four one-byte codes per dword)
__global__ __launch_bounds__(256) void pack8_kernel(const uint32_t* __restrict__ packed, uint32_t* __restrict__ packed8, int64_t n_blocks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // (block, t, lane)
  if (i >= n_blocks * 3 * 64) return;
  const int lane = (int)(i & 63);
  const int64_t bt = i >> 6;
  const int t = (int)(bt % 3);
  const int64_t b = bt / 3;
  const uint32_t p0 = packed[((size_t)b * 6 + 2 * t) * 64 + lane], p1 = packed[((size_t)b * 6 + 2 * t + 1) * 64 + lane];
  packed8[i] = (p0 & 0xffu) | (((p0 >> 16) & 0xffu) << 8) | ((p1 & 0xffu) << 16) | (((p1 >> 16) & 0xffu) << 24);
}
// (Re)build the one-byte code array of a handle whose codes fit a byte (K <= 256, m = 12: the cell-grouped scans' shape).
static int build_packed8(freddy_gpu_index* ix) {
  if (ix->packed8 && ix->packed8_own) { (void)hipFree(ix->packed8); ix->bytes -= ix->packed8_bytes; }   // (rebuilt after append_rows: the old copy no longer counts)
  ix->packed8 = nullptr; ix->packed8_own = false; ix->packed8_bytes = 0;
  if (ix->K > 256 || ix->m != 12 || ix->M2 != 6 || !ix->packed || ix->n_blocks <= 0) return 0;
  const size_t bytes = sizeof(uint32_t) * (size_t)ix->n_blocks * 3 * 64;
  if (hipMalloc((void**)&ix->packed8, bytes) != hipSuccess) { ix->packed8 = nullptr; return 0; }   // (no room: the int16 layout serves)
  ix->packed8_own = true;
  ix->packed8_bytes = (int64_t)bytes;
  ix->bytes += (int64_t)bytes;
  const int64_t n = ix->n_blocks * 3 * 64;
  hipLaunchKernelGGL(pack8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ix->stream, ix->packed, ix->packed8, ix->n_blocks);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return 0;
}

// Everything on the device that is a function of the (residual) codebook: the transposed copy of the generic
// LUT kernel, the paired layout of the exact fused scan, and -- for the filter + refine scan -- the row-major
// copy and the norm bounds.  (Re)built at pin time and by freddy_gpu_update_codebook; the row terms follow
// in refresh_row_terms once the rows are in place.
// Everything that is derived from the codebook.  The new tables are built beside the old ones and swapped in only when
// every upload has succeeded (freddy_gpu_update_codebook on a live handle: a failed call leaves the handle as it was).
static int derive_codebook_tables_into(freddy_gpu_index* ix, const float* codebook);
static int derive_codebook_tables(freddy_gpu_index* ix, const float* codebook) {
  float* const old[] = {ix->cbT, ix->cbP, ix->cbR, ix->pmax, ix->cmaxp, ix->cbF};
  const int64_t bytes_before = ix->bytes;
  ix->cbT = ix->cbP = ix->cbR = ix->pmax = ix->cmaxp = ix->cbF = nullptr;
  const int rc = derive_codebook_tables_into(ix, codebook);
  if (rc) {   // put the old tables back
    float* const fresh[] = {ix->cbT, ix->cbP, ix->cbR, ix->pmax, ix->cmaxp, ix->cbF};
    for (float* p : fresh) if (p) (void)hipFree(p);
    ix->cbT = old[0]; ix->cbP = old[1]; ix->cbR = old[2]; ix->pmax = old[3]; ix->cmaxp = old[4]; ix->cbF = old[5];
    ix->bytes = bytes_before;
    return rc;
  }
  int64_t old_bytes = 0;
  if (old[0]) old_bytes += (int64_t)sizeof(float) * ix->m * ix->S * ix->K;
  if (old[1]) old_bytes += (int64_t)sizeof(float) * ix->m * (((ix->S + 3) & ~3) / 4) * FUSED_T * 8;
  if (old[2]) old_bytes += (int64_t)sizeof(float) * ix->m * ix->K * ix->S;
  if (old[3]) old_bytes += (int64_t)sizeof(float) * ix->m;
  if (old[4]) old_bytes += (int64_t)sizeof(float) * ix->m;
  if (old[5]) old_bytes += (int64_t)sizeof(float) * ix->m * 8 * 7 * 64 * 8;
  ix->bytes -= old_bytes;       // (the footprint changes by the difference, not by a second copy)
  for (float* p : old) if (p) (void)hipFree(p);
  if (ix->kind == KIND_PQ) {    // views of the flat table are rebuilt from the new tables on next use
    if (ix->pq_shadow) { free_index(ix->pq_shadow); ix->pq_shadow = nullptr; }
    if (ix->pq_sub_view) { free_index(ix->pq_sub_view); ix->pq_sub_view = nullptr; }
  }
  return 0;
}
// The codebook in the order the table kernel's v_mfma_f32_16x16x4_f32 B operands are read (m = 12, S = 25, K <= 1024): for
// (position, group g of 16 code slots, step) lane l = (col = l & 15, kq = l >> 4) finds the eight values of dimension
// 4 step + kq for the codes 128 i + 16 g + col + 512 e, (i, e) = (0,0) (0,1) (1,0) ... (3,1), as two 16-byte words: a wave's
// load is 2 KB of consecutive bytes (the transposed copy gave 64-byte pieces of eight different lines).
static int build_fragment_codebook(freddy_gpu_index* ix, const float* codebook) {
  std::vector<float> f((size_t)ix->m * 8 * 7 * 64 * 8, 0.0f);
  for (int p = 0; p < ix->m; ++p)
    for (int g = 0; g < 8; ++g)
      for (int st = 0; st < 7; ++st)
        for (int l = 0; l < 64; ++l)
          for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 2; ++e) {
              const int j = 4 * st + (l >> 4), c = 128 * i + 16 * g + (l & 15) + 512 * e;
              if (j < ix->S && c < ix->K)
                f[((((size_t)p * 8 + g) * 7 + st) * 64 + l) * 8 + i * 2 + e] = codebook[((size_t)p * ix->K + c) * ix->S + j];
            }
  if (upload(&ix->cbF, f.data(), f.size(), &ix->bytes)) return fail(FREDDY_E_NOMEM, "device allocation failed");
  return 0;
}
static int derive_codebook_tables_into(freddy_gpu_index* ix, const float* codebook) {
  std::vector<float> cbT = transpose_codebook(codebook, ix->m, ix->K, ix->S);
  if (upload(&ix->cbT, cbT.data(), cbT.size(), &ix->bytes)) return fail(FREDDY_E_NOMEM, "device allocation failed");
  if (ix->kind == KIND_PQ) {
    // batches over the flat table take the cell-grouped filter + refine scan (pq_shadow_build): its codebook-derived tables,
    // with "centroids" that are zero
    if (ix->m == 12 && ix->S == 25 && ix->K <= FUSED_T * FUSED_E) {
      std::vector<float> cmaxp((size_t)ix->m);
      for (int p = 0; p < ix->m; ++p) {
        double cmax = 0.0;
        for (int c = 0; c < ix->K; ++c) {
          double n2 = 0.0;
          for (int j = 0; j < ix->S; ++j) { const double v = codebook[((size_t)p * ix->K + c) * ix->S + j]; n2 += v * v; }
          cmax = std::max(cmax, std::sqrt(n2));
        }
        cmaxp[p] = (float)(cmax * (1.0 + 1e-6));
      }
      if (upload(&ix->cbR, codebook, (size_t)ix->m * ix->K * ix->S, &ix->bytes) ||
          upload(&ix->pmax, cmaxp.data(), cmaxp.size(), &ix->bytes) ||
          upload(&ix->cmaxp, cmaxp.data(), cmaxp.size(), &ix->bytes))
        return fail(FREDDY_E_NOMEM, "device allocation failed");
      if (int rc = build_fragment_codebook(ix, codebook)) return rc;
    }
    return 0;
  }
  if (ix->kind != KIND_IVF) return 0;
  const int C = ix->C, d = ix->d;
  if (ix->K <= FUSED_T * FUSED_E) {
    // paired layout of the fused kernels: slot t holds codes (t, t+512); 4 dims x 2 codes per 32 bytes.
    // (Splitting the two 16-byte halves of a slot into separate contiguous arrays measured SLOWER: the
    // second load of a slot then no longer hits the lines the first one brought in.)
    const int SP = (ix->S + 3) & ~3, SPq = SP / 4;
    std::vector<float> cbP((size_t)ix->m * SPq * FUSED_T * 8, 0.0f);
    for (int p = 0; p < ix->m; ++p)
      for (int jb = 0; jb < SPq; ++jb)
        for (int tl = 0; tl < FUSED_T; ++tl)
          for (int u = 0; u < 4; ++u)
            for (int e = 0; e < 2; ++e) {
              const int j = jb * 4 + u, c = tl + e * FUSED_T;
              if (j < ix->S && c < ix->K)
                cbP[((((size_t)p * SPq + jb) * FUSED_T + tl) * 4 + u) * 2 + e] = codebook[((size_t)p * ix->K + c) * ix->S + j];
            }
    if (upload(&ix->cbP, cbP.data(), cbP.size(), &ix->bytes)) return fail(FREDDY_E_NOMEM, "device allocation failed");
  }
  // filter + refine tables (fused4.h)
  if (ix->cbP && ix->m == 12 && ix->S == 25) {
    std::vector<float> pmax((size_t)ix->m), cmaxp((size_t)ix->m);
    for (int p = 0; p < ix->m; ++p) {
      double comax = 0.0, cmax = 0.0;
      for (int c = 0; c < C; ++c) {
        double n2 = 0.0;
        for (int j = 0; j < ix->S; ++j) { const double v = ix->h_coarse[(size_t)c * d + p * ix->S + j]; n2 += v * v; }
        comax = std::max(comax, std::sqrt(n2));
      }
      for (int c = 0; c < ix->K; ++c) {
        double n2 = 0.0;
        for (int j = 0; j < ix->S; ++j) { const double v = codebook[((size_t)p * ix->K + c) * ix->S + j]; n2 += v * v; }
        cmax = std::max(cmax, std::sqrt(n2));
      }
      pmax[p] = (float)((comax + cmax) * (1.0 + 1e-6));
      cmaxp[p] = (float)(cmax * (1.0 + 1e-6));
    }
    if (upload(&ix->cbR, codebook, (size_t)ix->m * ix->K * ix->S, &ix->bytes) ||
        upload(&ix->pmax, pmax.data(), pmax.size(), &ix->bytes) ||
        upload(&ix->cmaxp, cmaxp.data(), cmaxp.size(), &ix->bytes))
      return fail(FREDDY_E_NOMEM, "device allocation failed");
    if (int rc = build_fragment_codebook(ix, codebook)) return rc;
  }
  return 0;
}

// rterm[slot] for every row slot of the pinned lists (the (cell, row) part of the filter's cheap distance)
static int refresh_row_terms(freddy_gpu_index* ix) {
  if (ix->rterm) { (void)hipFree(ix->rterm); ix->rterm = nullptr; }
  if (!ix->cbR) return 0;
  const int64_t n_slots = std::max<int64_t>(ix->n_blocks, 1) * 64;
  if (hipMalloc((void**)&ix->rterm, sizeof(float) * (size_t)n_slots) != hipSuccess) return fail(FREDDY_E_NOMEM, "device allocation failed");
  if (ix->n_blocks > 0) {
    hipLaunchKernelGGL(row_term_kernel, dim3((unsigned)((ix->n_blocks * 64 + 255) / 256)), dim3(256), 0, ix->stream, ix->packed,
                       ix->blk_cell, ix->coarse, ix->cbR, ix->rterm, ix->n_blocks * 64, ix->M2, ix->d, ix->m, ix->K, ix->S);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ix->stream) != hipSuccess)
      return fail(FREDDY_E_HIP, "building the row terms failed");
  }
  return 0;
}

int open_device(freddy_gpu_index* ix, int device) {
  // The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the
  // runtime starts: the pipeline's four lanes want a queue each beside the library's own stream (6 queues measured
  // best for ONE process) -- but several backends with six queues each are together slower than one, so a process that
  // finds other live backends takes two (core.hip choose_hw_queues).  Never overrides the environment; without effect if
  // the runtime is already up.
  // (the registry counts backends per PHYSICAL GPU: one process per GPU -- bench.py --gpus N, one PostgreSQL cluster per GPU -- are not neighbours)
  if (!ix->registered) { ix->registered = true; backend_handles(+1, device); }   // (before the count of the others: two backends that start together see each other)
  choose_hw_queues(device);
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(FREDDY_E_ARG, "device %d out of range (%d visible)", device, n);
  HIP_TRY(hipSetDevice(device));
  ix->device = device;
  ix->tune = read_tuning();
  if (int rc = raise_lds_limits_ivfadc(device)) return rc;
  if (int rc = raise_lds_limits_pq(device)) return rc;
  if (int rc = raise_lds_limits_join(device)) return rc;
  if (int rc = raise_lds_limits_exact(device)) return rc;
  HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ix->n_cus = prop.multiProcessorCount;
  return 0;
}

static int check_pq_shape(int d, int m, int K, int64_t N) {
  if (d <= 0 || m <= 0 || K <= 0 || N < 0) return fail(FREDDY_E_ARG, "non-positive dimension");
  if (d % m) return fail(FREDDY_E_ARG, "d=%d is not a multiple of m=%d", d, m);
  if (K > 65536) return fail(FREDDY_E_LIMIT, "K=%d does not fit a 16-bit code", K);
  if ((size_t)m * K * 4 + 4096 > 160 * 1024)
    return fail(FREDDY_E_LIMIT, "LUT of m*K=%d floats does not fit the 160 KiB LDS", m * K);
  if (N > (int64_t)INT32_MAX - 64) return fail(FREDDY_E_LIMIT, "N too large for 32-bit row positions");
  return 0;
}

extern "C" int freddy_gpu_pin_pq(const freddy_pq_desc* t, int device, freddy_gpu_index_t** out) {
  if (!t || !out || !t->codebook || (t->N && (!t->ids || !t->codes))) return fail(FREDDY_E_ARG, "NULL argument");
  if (int rc = check_pq_shape(t->d, t->m, t->K, t->N)) return rc;
  for (int64_t r = 1; r < t->N; ++r)
    if (t->ids[r] <= t->ids[r - 1])
      return fail(FREDDY_E_ARG, "ids must be strictly ascending (canonical scan order); violated at row %lld", (long long)r);
  freddy_gpu_index* ix = new freddy_gpu_index();
  ix->kind = KIND_PQ;
  ix->d = t->d; ix->m = t->m; ix->K = t->K; ix->S = t->d / t->m; ix->M2 = (t->m + 1) / 2; ix->N = t->N;
  int rc = open_device(ix, device);
  if (!rc) rc = derive_codebook_tables(ix, t->codebook);
  if (!rc && upload(&ix->ids, t->ids, (size_t)t->N, &ix->bytes)) rc = fail(FREDDY_E_NOMEM, "device allocation failed");
  if (!rc) {
    const int32_t off[2] = {0, (int32_t)t->N};
    rc = pack_lists(ix, 1, off, t->codes, nullptr);
    if (!rc) rc = build_packed8(ix);
  }
  if (!rc) { ix->h_ids.assign(t->ids, t->ids + t->N); ix->max_id = t->N ? t->ids[t->N - 1] : -1; }
  if (rc) { free_index(ix); return rc; }
  *out = ix;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_pin_ivf(const freddy_ivf_desc* t, int device, freddy_gpu_index_t** out) {
  if (!t || !out || !t->codebook || !t->coarse || !t->list_off || (t->N && (!t->ids || !t->codes)))
    return fail(FREDDY_E_ARG, "NULL argument");
  if (int rc = check_pq_shape(t->d, t->m, t->K, t->N)) return rc;
  if (t->C <= 0) return fail(FREDDY_E_ARG, "C must be positive");
  if (t->list_off[0] != 0 || t->list_off[t->C] != t->N) return fail(FREDDY_E_ARG, "list_off must span [0, N]");
  for (int c = 0; c < t->C; ++c)   // every offset is checked BEFORE any row is touched through it
    if (t->list_off[c] < 0 || t->list_off[c] > t->list_off[c + 1] || (int64_t)t->list_off[c + 1] > t->N)
      return fail(FREDDY_E_ARG, "list_off is not non-decreasing inside [0, N] at list %d", c);
  for (int c = 0; c < t->C; ++c)
    for (int64_t r = t->list_off[c]; r < t->list_off[c + 1]; ++r) {
      if (t->ids[r] < 0) return fail(FREDDY_E_ARG, "negative id at row %lld", (long long)r);
      if (r > t->list_off[c] && t->ids[r] <= t->ids[r - 1])
        return fail(FREDDY_E_ARG, "ids must be strictly ascending inside list %d (row %lld)", c, (long long)r);
    }
  {   // "unique overall": a row id may sit in one list only (the merge orders a query's candidates by id)
    std::vector<int32_t> sorted_ids(t->ids, t->ids + t->N);
    std::sort(sorted_ids.begin(), sorted_ids.end());
    for (int64_t r = 1; r < t->N; ++r)
      if (sorted_ids[(size_t)r] == sorted_ids[(size_t)r - 1])
        return fail(FREDDY_E_ARG, "id %d occurs in more than one list", (int)sorted_ids[(size_t)r]);
  }
  freddy_gpu_index* ix = new freddy_gpu_index();
  ix->kind = KIND_IVF;
  ix->d = t->d; ix->m = t->m; ix->K = t->K; ix->S = t->d / t->m; ix->M2 = (t->m + 1) / 2; ix->N = t->N; ix->C = t->C;
  int rc = open_device(ix, device);
  if (!rc) {
    ix->Cpad = (t->C + WG - 1) / WG * WG;
    std::vector<float> cT((size_t)t->d * ix->Cpad, 0.0f);
    for (int c = 0; c < t->C; ++c)
      for (int i = 0; i < t->d; ++i) cT[(size_t)i * ix->Cpad + c] = t->coarse[(size_t)c * t->d + i];
    if (upload(&ix->coarse, t->coarse, (size_t)t->C * t->d, &ix->bytes) ||
        upload(&ix->coarseT, cT.data(), cT.size(), &ix->bytes))
      rc = fail(FREDDY_E_NOMEM, "device allocation failed");
    if (!rc) {   // MFMA coarse kernel (coarse.h): zero-padded rows, squared norms (fp64, rounded once), largest norm
      ix->dp = (t->d + COARSE_DP_ALIGN - 1) / COARSE_DP_ALIGN * COARSE_DP_ALIGN;
      // fragment order [Cpad / 32][dp / 8][lane = 32 h + r][4]: element t = c[32 g + r][8 i + 4 h + t]
      std::vector<float> cP((size_t)ix->Cpad * ix->dp, 0.0f), cn2((size_t)ix->Cpad, 0.0f);
      const int nit = ix->dp / 8;
      double cmax2 = 0.0;
      for (int c = 0; c < t->C; ++c) {
        double n2 = 0.0;
        for (int i = 0; i < t->d; ++i) {
          const float v = t->coarse[(size_t)c * t->d + i];
          const int it = i >> 3, hh = (i >> 2) & 1, tt = i & 3;
          cP[((((size_t)(c >> 5) * nit + it) * 64) + (size_t)hh * 32 + (c & 31)) * 4 + tt] = v;
          n2 += (double)v * (double)v;
        }
        cn2[(size_t)c] = (float)n2;
        cmax2 = std::max(cmax2, n2);
      }
      ix->cmax = (float)(std::sqrt(cmax2) * (1.0 + 1e-6));
      // the f16-split copy of the centroids for the matrix cores (coarse.h coarse_approx16_body; FREDDY_GPU_COARSE_H16=0: the fp32 tiles).
      // Many cells: the fp32 tiles are bound by the matrix pipe (134 -> 102 us at 13 000 cells); 1000 cells: 32 -> 29 us alone,
      // 54 -> 46 us with four batches in flight (fewer matrix-pipe cycles beside the other batches' kernels)
      if (env_int("FREDDY_GPU_COARSE_H16", 1) != 0 && t->d % 4 == 0) {
        float amax = 0.0f;
        for (size_t i = 0; i < (size_t)t->C * t->d; ++i) amax = std::max(amax, std::fabs(t->coarse[i]));
        int e = 0;
        if (amax > 0.0f && amax < 3e38f) { (void)frexpf(amax, &e); e = 14 - e; }
        ix->coarse_ec = e;
        const int T = (t->d + 15) / 16;
        std::vector<_Float16> cH((size_t)ix->Cpad * T * 2 * 8 * 2, (_Float16)0.0f);   // [Cpad/32][T][2][64][8]
        for (int c = 0; c < t->C; ++c)
          for (int i = 0; i < t->d; ++i) {
            const float v = ldexpf(t->coarse[(size_t)c * t->d + i], e);
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            const int tt = i >> 4, g = (i >> 3) & 1, u = i & 7;
            const size_t base = (((size_t)(c >> 5) * T + tt) * 2) * 64;
            cH[(base + (size_t)g * 32 + (c & 31)) * 8 + u] = hi;
            cH[(base + 64 + (size_t)g * 32 + (c & 31)) * 8 + u] = lo;
          }
        _Float16* dH = nullptr;
        if (upload(&dH, cH.data(), cH.size(), &ix->bytes)) rc = fail(FREDDY_E_NOMEM, "device allocation failed");
        ix->coarseH = dH;
      }
      if (rc) {} else
      if (upload(&ix->coarseP, cP.data(), cP.size(), &ix->bytes) || upload(&ix->cn2, cn2.data(), cn2.size(), &ix->bytes) ||
          hipMalloc((void**)&ix->viol, 4 * sizeof(int32_t)) != hipSuccess || hipMemset(ix->viol, 0, 4 * sizeof(int32_t)) != hipSuccess)
        rc = fail(FREDDY_E_NOMEM, "device allocation failed");
    }
    if (!rc) { ix->h_coarse.assign(t->coarse, t->coarse + (size_t)t->C * t->d); rc = derive_codebook_tables(ix, t->codebook); }
  }
  if (!rc) rc = pack_lists(ix, t->C, t->list_off, t->codes, t->ids);
  if (!rc) rc = build_packed8(ix);
  if (!rc) rc = refresh_row_terms(ix);   // one float per row slot: the (cell, row) part of the filter's cheap distance
  if (!rc) {
    if (ix->rterm) ix->bytes += (int64_t)sizeof(float) * std::max<int64_t>(ix->n_blocks, 1) * 64;
    for (int64_t r = 0; r < t->N; ++r) ix->max_id = std::max(ix->max_id, t->ids[r]);
  }
  if (rc) { free_index(ix); return rc; }
  *out = ix;
  return FREDDY_OK;
}

extern "C" int freddy_gpu_pin_ivf_multi(const freddy_ivf_desc* t, const int* devices, int n_devices, freddy_gpu_index_t** out) {
  if (!devices || n_devices < 1 || !out) return fail(FREDDY_E_ARG, "bad device list");
  freddy_gpu_index* first = nullptr;
  if (int rc = freddy_gpu_pin_ivf(t, devices[0], &first)) return rc;
  for (int g = 1; g < n_devices; ++g) {
    freddy_gpu_index* rep = nullptr;
    if (int rc = freddy_gpu_pin_ivf(t, devices[g], &rep)) { free_index(first); return rc; }
    first->replicas.push_back(rep);
  }
  (void)hipSetDevice(devices[0]);
  *out = first;
  return FREDDY_OK;
}
// ---------------------------------------------------------------------------------------
// insert_batch: HBM index mutation
// ---------------------------------------------------------------------------------------
// new block j of list blk_cell[b] <- old block j of that list (or empty)
__global__ __launch_bounds__(256) void repack_blocks_kernel(const uint32_t* __restrict__ old_packed, const int32_t* __restrict__ old_pos,
                                                           const int32_t* __restrict__ old_blk_off, const int32_t* __restrict__ new_blk_off,
                                                           const int32_t* __restrict__ new_blk_cell, uint32_t* __restrict__ packed,
                                                           int32_t* __restrict__ pos, int64_t n_new_blocks, int M2) {
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= n_new_blocks) return;
  const int c = new_blk_cell[b];
  const int j = (int)(b - new_blk_off[c]);
  const bool have = j < old_blk_off[c + 1] - old_blk_off[c];
  const int64_t ob = (int64_t)old_blk_off[c] + j;
  for (int w = 0; w < M2; ++w) packed[((size_t)b * M2 + w) * 64 + lane] = have ? old_packed[((size_t)ob * M2 + w) * 64 + lane] : 0u;
  pos[(size_t)b * 64 + lane] = have ? old_pos[(size_t)ob * 64 + lane] : -1;
}
// new rows into their slots: slot[i] = row slot (block * 64 + lane) of new row i
__global__ __launch_bounds__(256) void place_rows_kernel(const int64_t* __restrict__ slot, const int32_t* __restrict__ row_pos,
                                                        const int16_t* __restrict__ codes, int64_t n, uint32_t* __restrict__ packed,
                                                        int32_t* __restrict__ pos, int m, int M2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t sl = slot[i], b = sl >> 6;
  const int lane = (int)(sl & 63);
  for (int w = 0; w < M2; ++w) {
    const uint32_t lo = (uint16_t)codes[(size_t)i * m + 2 * w];
    const uint32_t hi = (2 * w + 1 < m) ? (uint16_t)codes[(size_t)i * m + 2 * w + 1] : 0u;
    packed[((size_t)b * M2 + w) * 64 + lane] = lo | (hi << 16);
  }
  pos[(size_t)sl] = row_pos[i];
}
// raw vectors into the 64-row blocked layout: row r -> xb[r / 64][dim][r % 64]
__global__ __launch_bounds__(256) void place_vectors_kernel(const float* __restrict__ src, int64_t first_row, int64_t n, float* __restrict__ xb, int d) {
  const int64_t i = (int64_t)blockIdx.x;
  if (i >= n) return;
  const int64_t r = first_row + i;
  for (int dim = threadIdx.x; dim < d; dim += 256) xb[((r >> 6) * d + dim) * 64 + (r & 63)] = src[(size_t)i * d + dim];
}

template <class T>
static int grow_device_array(T** arr, size_t old_n, size_t new_n, const T* append_host, size_t append_n) {
  T* fresh = nullptr;
  if (hipMalloc((void**)&fresh, sizeof(T) * std::max<size_t>(new_n, 1)) != hipSuccess) return -1;
  if (old_n && hipMemcpy(fresh, *arr, sizeof(T) * old_n, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(fresh); return -2; }
  if (append_n && hipMemcpy(fresh + old_n, append_host, sizeof(T) * append_n, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(fresh); return -2; }
  if (*arr) (void)hipFree(*arr);
  *arr = fresh;
  return 0;
}

// rows of a pq / ivf index: each new row goes to the end of its list; the 64-row block layout is rebuilt on
// the device (old blocks copied to their new places, new rows written into the free slots behind them)
static int append_packed_rows(freddy_gpu_index* ix, int n_lists, int64_t n, const int32_t* cell, const int32_t* row_pos, const int16_t* codes) {
  const int m = ix->m, M2 = ix->M2;
  std::vector<int32_t> new_list_off((size_t)n_lists + 1, 0), add((size_t)n_lists, 0);
  for (int64_t i = 0; i < n; ++i) {
    const int c = cell ? cell[i] : 0;
    if (c < 0 || c >= n_lists) return fail(FREDDY_E_ARG, "coarse_id %d of new row %lld is outside [0, %d)", c, (long long)i, n_lists);
    for (int l = 0; l < m; ++l)
      if (codes[(size_t)i * m + l] < 0 || codes[(size_t)i * m + l] >= ix->K)
        return fail(FREDDY_E_ARG, "code %d of new row %lld is outside [0, %d)", (int)codes[(size_t)i * m + l], (long long)i, ix->K);
    add[(size_t)c]++;
  }
  std::vector<int32_t> old_blk((size_t)n_lists + 1, 0), new_blk((size_t)n_lists + 1, 0);
  int max_blocks = 0;
  for (int c = 0; c < n_lists; ++c) {
    const int64_t old_len = ix->h_list_off[(size_t)c + 1] - ix->h_list_off[(size_t)c];
    old_blk[(size_t)c + 1] = old_blk[(size_t)c] + (int32_t)((old_len + 63) / 64);
    const int64_t len = old_len + add[(size_t)c];
    if ((int64_t)new_list_off[(size_t)c] + len > INT32_MAX - 64) return fail(FREDDY_E_LIMIT, "N too large for 32-bit row positions");
    new_list_off[(size_t)c + 1] = new_list_off[(size_t)c] + (int32_t)len;
    const int nb = (int)((len + 63) / 64);
    new_blk[(size_t)c + 1] = new_blk[(size_t)c] + nb;
    max_blocks = std::max(max_blocks, nb);
  }
  const int64_t n_new_blocks = new_blk[(size_t)n_lists];
  std::vector<int32_t> blk_cell((size_t)std::max<int64_t>(n_new_blocks, 1), 0);
  for (int c = 0; c < n_lists; ++c)
    for (int b = new_blk[(size_t)c]; b < new_blk[(size_t)c + 1]; ++b) blk_cell[(size_t)b] = c;
  std::vector<int64_t> slot((size_t)n);
  std::vector<int32_t> cursor((size_t)n_lists, 0);
  for (int64_t i = 0; i < n; ++i) {
    const int c = cell ? cell[i] : 0;
    const int64_t old_len = ix->h_list_off[(size_t)c + 1] - ix->h_list_off[(size_t)c];
    slot[(size_t)i] = (int64_t)new_blk[(size_t)c] * 64 + old_len + cursor[(size_t)c]++;
  }
  uint32_t* packed = nullptr;
  int32_t *pos = nullptr, *d_blk_cell = nullptr, *d_new_blk = nullptr, *d_list_off = nullptr, *d_row_pos = nullptr;
  int64_t* d_slot = nullptr;
  int16_t* d_codes = nullptr;
  int64_t junk = 0;
  int rc = 0;
  if (hipMalloc((void**)&packed, sizeof(uint32_t) * (size_t)std::max<int64_t>(n_new_blocks, 1) * M2 * 64) != hipSuccess ||
      hipMalloc((void**)&pos, sizeof(int32_t) * (size_t)std::max<int64_t>(n_new_blocks, 1) * 64) != hipSuccess ||
      upload(&d_blk_cell, blk_cell.data(), blk_cell.size(), &junk) || upload(&d_new_blk, new_blk.data(), new_blk.size(), &junk) ||
      upload(&d_list_off, new_list_off.data(), new_list_off.size(), &junk) || upload(&d_slot, slot.data(), slot.size(), &junk) ||
      upload(&d_row_pos, row_pos, (size_t)n, &junk) || upload(&d_codes, codes, (size_t)n * m, &junk))
    rc = fail(FREDDY_E_NOMEM, "device allocation failed while appending rows");
  if (!rc && n_new_blocks > 0) {
    hipLaunchKernelGGL(repack_blocks_kernel, dim3((unsigned)((n_new_blocks + 3) / 4)), dim3(256), 0, ix->stream, ix->packed, ix->pos, ix->blk_off,
                       d_new_blk, d_blk_cell, packed, pos, n_new_blocks, M2);
    if (n > 0)
      hipLaunchKernelGGL(place_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ix->stream, d_slot, d_row_pos, d_codes, n, packed, pos, m, M2);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ix->stream) != hipSuccess) rc = fail(FREDDY_E_HIP, "re-blocking the lists failed");
  }
  void* tmp[] = {d_slot, d_row_pos, d_codes};
  for (void* p : tmp) if (p) (void)hipFree(p);
  if (rc) {
    void* fresh[] = {packed, pos, d_blk_cell, d_new_blk, d_list_off};
    for (void* p : fresh) if (p) (void)hipFree(p);
    return rc;
  }
  void* old[] = {ix->packed, ix->pos, ix->blk_cell, ix->blk_off, ix->list_off};
  for (void* p : old) if (p) (void)hipFree(p);
  ix->packed = packed; ix->pos = pos; ix->blk_cell = d_blk_cell; ix->blk_off = d_new_blk; ix->list_off = d_list_off;
  ix->n_blocks = n_new_blocks;
  ix->max_list_blocks = max_blocks;
  ix->h_list_off = new_list_off;
  ix->N += n;
  if (!ix->shadow_of) { if (int rc = build_packed8(ix)) return rc; }
  return 0;
}

extern "C" int freddy_gpu_append_rows(freddy_gpu_index_t* ix, int64_t n, const int32_t* ids, const int32_t* coarse_id,
                                      const int16_t* codes, const float* vectors) {
  if (!ix) return fail(FREDDY_E_ARG, "NULL index");
  if (n < 0 || (n > 0 && !ids)) return fail(FREDDY_E_ARG, "bad argument");
  if (n == 0) return FREDDY_OK;
  if (!ix->replicas.empty()) {
    // every replica holds the same tables: the primary goes first (argument errors are found there before anything has
    // changed anywhere); a failure after that leaves the devices with different tables -> the handle is poisoned and
    // every search on it fails loudly until it is unpinned
    std::vector<freddy_gpu_index*> reps;
    reps.swap(ix->replicas);
    int rc = freddy_gpu_append_rows(ix, n, ids, coarse_id, codes, vectors);
    reps.swap(ix->replicas);
    if (rc) { if (rc == FREDDY_E_HIP || rc == FREDDY_E_NOMEM) ix->poisoned = true; return rc; }
    for (freddy_gpu_index* r : ix->replicas)
      if ((rc = freddy_gpu_append_rows(r, n, ids, coarse_id, codes, vectors))) { ix->poisoned = true; return rc; }
    return FREDDY_OK;
  }
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  if (ix->kind == KIND_IVPQ) ix->join.tl_valid = false;   // (the cached "id IN (targets)" resolution refers to the rows as they were)
  const int32_t last_id = ix->kind == KIND_IVPQ ? (ix->join.h_ids.empty() ? -1 : ix->join.h_ids.back())
                          : ix->kind == KIND_IVF ? ix->max_id : (ix->h_ids.empty() ? -1 : ix->h_ids.back());
  for (int64_t i = 0; i < n; ++i)
    if (ids[i] <= (i ? ids[i - 1] : last_id))
      return fail(FREDDY_E_ARG, "appended ids must ascend beyond the largest pinned id %d (row %lld has %d)", last_id, (long long)i, ids[i]);
  if (ix->N + n > (int64_t)INT32_MAX - 64) return fail(FREDDY_E_LIMIT, "N too large for 32-bit row positions");
  switch (ix->kind) {
    case KIND_PQ: {
      if (!codes) return fail(FREDDY_E_ARG, "codes are required");
      std::vector<int32_t> row_pos((size_t)n);
      for (int64_t i = 0; i < n; ++i) row_pos[(size_t)i] = (int32_t)(ix->N + i);   // flat table: position = row index
      const int64_t old_n = ix->N;
      if (ix->pq_shadow) { free_index(ix->pq_shadow); ix->pq_shadow = nullptr; }   // (rebuilt by the next batch search)
      if (int rc = append_packed_rows(ix, 1, n, nullptr, row_pos.data(), codes)) return rc;
      if (grow_device_array(&ix->ids, (size_t)old_n, (size_t)(old_n + n), ids, (size_t)n)) return fail(FREDDY_E_NOMEM, "device allocation failed");
      ix->h_ids.insert(ix->h_ids.end(), ids, ids + n);
      ix->max_id = ids[n - 1];
      return FREDDY_OK;
    }
    case KIND_IVF: {
      if (!codes || !coarse_id) return fail(FREDDY_E_ARG, "coarse_id and codes are required");
      if (int rc = append_packed_rows(ix, ix->C, n, coarse_id, ids, codes)) return rc;
      ix->max_id = ids[n - 1];
      return refresh_row_terms(ix);
    }
    case KIND_IVPQ: {
      JoinIndex& j = ix->join;
      if (!codes || !coarse_id || (j.has_vectors && !vectors)) return fail(FREDDY_E_ARG, "coarse_id, codes (and vectors, if pinned) are required");
      for (int64_t i = 0; i < n; ++i) {
        if (coarse_id[i] < 0 || coarse_id[i] >= j.cells) return fail(FREDDY_E_ARG, "coarse_id %d out of range", coarse_id[i]);
        for (int l = 0; l < j.m; ++l)
          if (codes[(size_t)i * j.m + l] < 0 || codes[(size_t)i * j.m + l] >= j.K) return fail(FREDDY_E_ARG, "code out of range at new row %lld", (long long)i);
      }
      const size_t o = (size_t)j.N, nn = (size_t)(j.N + n);
      if (grow_device_array(&j.ids, o, nn, ids, (size_t)n) || grow_device_array(&j.cell, o, nn, coarse_id, (size_t)n) ||
          grow_device_array(&j.codes, o * j.MP, nn * j.MP, join_pad_codes(codes, n, j.m, j.MP).data(), (size_t)n * j.MP) ||
          (j.has_vectors && grow_device_array(&j.vectors, o * j.d, nn * j.d, vectors, (size_t)n * j.d)))
        return fail(FREDDY_E_NOMEM, "device allocation failed");
      if (j.markbits) (void)hipFree(j.markbits);
      j.markbits = nullptr;
      HIP_TRY(hipMalloc((void**)&j.markbits, sizeof(uint32_t) * ((nn + 31) / 32 + 1)));
      j.h_ids.insert(j.h_ids.end(), ids, ids + n);
      j.h_cell.insert(j.h_cell.end(), coarse_id, coarse_id + n);
      j.N += n; ix->N = j.N;
      j.ids_affine = (int64_t)j.h_ids.back() - j.h_ids.front() == j.N - 1;
      return FREDDY_OK;
    }
    case KIND_VEC: {
      if (!vectors) return fail(FREDDY_E_ARG, "vectors are required");
      const int d = ix->d;
      const size_t o = (size_t)ix->N, nn = (size_t)(ix->N + n);
      const int64_t new_blocks = (int64_t)((nn + 63) / 64);
      float* xb = nullptr;
      HIP_TRY(hipMalloc((void**)&xb, sizeof(float) * (size_t)new_blocks * d * 64));
      HIP_TRY(hipMemset(xb, 0, sizeof(float) * (size_t)new_blocks * d * 64));
      if (ix->n_blocks) HIP_TRY(hipMemcpy(xb, ix->xb, sizeof(float) * (size_t)ix->n_blocks * d * 64, hipMemcpyDeviceToDevice));
      if (grow_device_array(&ix->coarse, o * d, nn * d, vectors, (size_t)n * d) || grow_device_array(&ix->ids, o, nn, ids, (size_t)n)) {
        (void)hipFree(xb);
        return fail(FREDDY_E_NOMEM, "device allocation failed");
      }
      hipLaunchKernelGGL(place_vectors_kernel, dim3((unsigned)n), dim3(256), 0, ix->stream, ix->coarse + o * d, (int64_t)o, n, xb, d);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(ix->stream));
      if (ix->xb) (void)hipFree(ix->xb);
      ix->xb = xb; ix->n_blocks = new_blocks; ix->N += n;
      ix->h_ids.insert(ix->h_ids.end(), ids, ids + n);
      return exf_table_stats(ix, (int64_t)o, n);   // (the filter's scale and norm bound cover the new rows)
    }
  }
  return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
}

extern "C" int freddy_gpu_update_codebook(freddy_gpu_index_t* ix, const float* codebook) {
  if (!ix || !codebook) return fail(FREDDY_E_ARG, "NULL argument");
  if (!ix->replicas.empty()) {   // every device or none: a failure after the first device has changed poisons the handle
    size_t done = 0;
    int rc = 0;
    for (freddy_gpu_index* r : ix->replicas) { if ((rc = freddy_gpu_update_codebook(r, codebook))) break; ++done; }
    if (!rc) {
      std::vector<freddy_gpu_index*> none;
      none.swap(ix->replicas);
      rc = freddy_gpu_update_codebook(ix, codebook);
      none.swap(ix->replicas);
      if (!rc) return FREDDY_OK;
      done = ix->replicas.size();
    }
    if (done > 0 || rc == FREDDY_E_HIP || rc == FREDDY_E_NOMEM) ix->poisoned = true;
    return rc;
  }
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipDeviceSynchronize());   // (searches of every stream and lane have drained before the tables change)
  if (ix->kind == KIND_PQ) return derive_codebook_tables(ix, codebook);
  if (ix->kind == KIND_IVF) {
    if (int rc = derive_codebook_tables(ix, codebook)) return rc;
    return refresh_row_terms(ix);
  }
  if (ix->kind == KIND_IVPQ) {
    JoinIndex& j = ix->join;
    std::vector<float> cbT((size_t)j.m * j.S * j.K);
    for (int p = 0; p < j.m; ++p)
      for (int c = 0; c < j.K; ++c)
        for (int i = 0; i < j.S; ++i) cbT[((size_t)p * j.S + i) * j.K + c] = codebook[((size_t)p * j.K + c) * j.S + i];
    HIP_TRY(hipMemcpy(j.cbT, cbT.data(), sizeof(float) * cbT.size(), hipMemcpyHostToDevice));
    return FREDDY_OK;
  }
  return fail(FREDDY_E_KIND, "index handle has the wrong kind for this call");
}

