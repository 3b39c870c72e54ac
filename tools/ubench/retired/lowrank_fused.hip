// ONE launch for the whole low-rank edit with at most 128 concepts  W_new = W_old + (W_old D_e^T) R_e
// (trainscripts/uce_sd_erase.py:56-82 for all modules at once; uce_edit's dual path):
//
//   blocks 0 .. riders-1   the small-system chain of uce_lowrank_riders.h: K = lambda S^-1 + C C^T (Gram riders), its blocked
//                          Cholesky + block inverses (the rider that draws the last ticket), R = rows of K^-1 C (solve riders,
//                          published WRITE-THROUGH; the last one posts stage 4)
//   the other blocks       project their MT*16-row super-tile, T = W_old (G - C_e)^T, exactly as k_lr_project does - but T
//                          stays in LDS -, wait for stage 4 (the chain is as long as the projection GEMM: 30 us each at 50
//                          concepts) and UPDATE THE SAME ROWS: W_new = W_old + T R.
//
// Against projection launch + update launch: no launch boundary, T (2 x 4 rows NEP bytes) never leaves the CU, and the second
// read of W_old comes out of the last-level cache (a workgroup re-reads the 344 KB it streamed ~30 us earlier; the slab
// is 77 MB) instead of HBM - the step's HBM traffic goes from 1.6x to ~1.05x the algorithmic bytes.
// Progress: only the projecting blocks wait, and only for the rider blocks, which are FIRST in the grid - a launch larger than
// the chip's residency (the SDXL slab: 1486 super-tiles) runs its later blocks after the chain has long finished.  Every wait
// is bounded (status -1).  Hand-off words: uce_lowrank_riders.h; the last projecting block re-arms them, so the launch
// replays from a hipGraph.
#undef UCE_CHAIN_DEBUG
#include "uce_lowrank_riders.h"

namespace {

__device__ __forceinline__ __amdgpu_buffer_rsrc_t fu_rsrc(const float* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0 ? (bytes < 0x7fffffffL ? (int)bytes : 0x7fffffff) : 0, 0x00020000);
}

// Update of the workgroup's rows [R0, R0 + MT*16) from the T tile in LDS.  8 waves: wave w takes the 64-column quarter
// w & 3 of every 256-column group and every second 16-row tile (w >> 2); a work item = (column group, tile) = 16 rows x 64 columns =
// 16 accumulator registers, loaded straight from W_old in the accumulator layout.
// PRE: the first PRE work items' rows of W_old are loaded BEFORE the wait for R (`wait()` below: the rider chain ends ~30 us after
// launch start, the projection earlier) and stay in registers across it - the second read of W_old, which bounded the update phase
// of the round-4 form (PRE = 0: every item's rows fetched after the wait, one item ahead), then costs nothing on the critical path:
// after the wait the phase is R fragments + MFMAs + stores.  The item loops are fully unrolled (static register indices).
template <int D, int MT, int NK, int PRE, class WaitFn>
__device__ __forceinline__ void fused_update(const float* __restrict__ W_old, const float* __restrict__ R, float* __restrict__ W_new,
                                             const float* Ts, int tld, long rows, int Ne, long R0, WaitFn wait) {
  constexpr int d = D;
  constexpr int MG = D / 256;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = w & 3, th = w >> 2;
  const int li = lane & 15, lk = lane >> 4;
  constexpr int NT = (MT + 1) / 2;                    // tiles of the first half of the waves; the second half may have one fewer
  const int my_tiles = th == 0 ? NT : MT - NT;
  const long r_bytes = (long)Ne * d * 4;
  const unsigned vo_r = (unsigned)((lk * d + wq * 64 + 4 * li) * 4);
  const unsigned vo_w = (unsigned)((4 * lk * d + wq * 64 + 4 * li) * 4);
  const float* Wb = W_old + R0 * d;
  float* Ob = W_new + R0 * d;
  const long tile_rows = (rows - R0) < MT * 16 ? (rows - R0) : MT * 16;   // rows of this super-tile that exist

  auto ld_r = [&](int t, int gi) -> float4_t {
    const long sh = ((long)4 * t * d + gi * 256) * 4;
    return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(fu_rsrc(R + 4 * t * d + gi * 256, r_bytes - sh), vo_r, 0,
                                                                               16 /* sc1: past the L1 */));
  };
  auto ld_w = [&](int tile, int r, int gi) -> float4_t {
    const long row0 = (long)tile * 16 + r;
    const long sh = (row0 * d + gi * 256) * 4;
    return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(fu_rsrc(Wb + row0 * d + gi * 256, tile_rows * d * 4 - sh),
                                                                               vo_w, 0, 0));
  };
  // item g = gi * NT + it (it < my_tiles): tile th + 2 it of column group gi
  constexpr int NITEM = MG * NT;
  constexpr int NPRE = PRE < NITEM ? PRE : NITEM;
  float4_t pre[NPRE > 0 ? NPRE : 1][4];
#pragma unroll
  for (int g = 0; g < NPRE; ++g) {
    const int gi = g / NT, it = g % NT;
    if (it < my_tiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pre[g][r] = ld_w(th + 2 * it, r, gi);
    }
  }
  wait();                                             // R is complete
  float4_t rr[NK];
#pragma unroll
  for (int t = 0; t < NK; ++t) rr[t] = ld_r(t, 0);
  float4_t res[4];
  if constexpr (NPRE == 0) {
    if (my_tiles > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) res[r] = ld_w(th, r, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int gi = 0; gi < MG; ++gi) {
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      if (it >= my_tiles) continue;                   // (wave-uniform)
      constexpr int dummy = 0;
      (void)dummy;
      const int g = gi * NT + it;
      const int tile = th + 2 * it;
      const bool last_tile = it + 1 == my_tiles;
      float4_t acc[4];                                // acc[q][r]: row tile*16 + 4*lk + r, column 4*li + q of the wave's 64
      if (g < NPRE) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q][r] = pre[g < NPRE ? g : 0][r][q];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q][r] = res[r][q];
      }
      // the next item that is NOT preloaded: its rows in flight under this item's MFMAs
      {
        const int gn_it = last_tile ? 0 : it + 1, gn_gi = last_tile ? gi + 1 : gi;
        const int gn = gn_gi * NT + gn_it;
        if (gn_gi < MG && gn >= NPRE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) res[r] = ld_w(th + 2 * gn_it, r, gn_gi);
        }
      }
      const float* trow = Ts + (tile * 16 + li) * tld + lk;
      float a = trow[0];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NK; ++t) {
        const float an = (t + 1 < NK) ? trow[4 * (t + 1)] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rr[t][q], acc[q], 0, 0, 0);
        if (last_tile && gi + 1 < MG) rr[t] = ld_r(t, gi + 1);     // reload in place: the next group's fragment of step t
        __builtin_amdgcn_sched_barrier(0);
        a = an;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4_t o = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        const long row0 = (long)tile * 16 + r;
        const long sh = (row0 * d + gi * 256) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o), fu_rsrc(Ob + row0 * d + gi * 256, tile_rows * d * 4 - sh), vo_w,
                                               0, 2 /* nt */);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same update on the bf16 matrix cores with fp32-equivalent products (the arithmetic of uce_apply_b3.hip): T and R each
// split exactly into three bf16 planes, a product = the six partial products of weight >= 2^-16 (small terms first), each
// exact inside v_mfma_f32_16x16x32_bf16 - 6 of 16 "bf16 equivalents" per f32 product: the update stops being f32-MFMA-bound
// (from ~40 concepts on it was) and runs at the rate its weight traffic allows.
//   * R arrives pre-split from the solve riders (GramPotrfJob::Rp, [3][NEP/8][4][D/4][8] bf16: 16 bytes = one lane's B operand);
//     a chunk of 256 (NEP = 64) / 128 (NEP = 128) columns (all planes, all concepts: 96 KB) is staged in LDS beside the T tile, the NEXT chunk's
//     pieces wait in registers (12 x 16 bytes per thread) while this one is used;
//   * T stays fp32 in LDS (row stride NEP + 4 floats: conflict-free 16-byte reads) and is split while the A fragments are formed;
//   * wave w: tile slot w >> 1 (tiles slot, slot + 4), column half w & 1; per (tile, 64-column block): 4 rows x 16 bytes of
//     W_old in (prefetched one block ahead), NEP / 32 steps x 4 column residues x 6 MFMAs, 4 rows x 16 bytes out.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 fu_bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float4_t fu_mfma(uint4_t a, uint4_t b, float4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(fu_bf16x8_t, a), __builtin_bit_cast(fu_bf16x8_t, b), c, 0, 0, 0);
}

template <int D, int MT, int CT, int PRE, class WaitFn>
__device__ __forceinline__ void fused_update_b3(const float* __restrict__ W_old, const unsigned short* __restrict__ Rp,
                                                float* __restrict__ W_new, const float* Ts, int tld, unsigned char* Rs, long rows,
                                                long R0, WaitFn wait) {
  constexpr int d = D;
  constexpr int NEP = 64 * CT;
  constexpr int KG = NEP / 8;                         // k-groups of 8 concepts
  constexpr int NS = NEP / 32;                        // MFMA k-steps
  constexpr int CHW = CT == 1 ? 256 : 128;            // columns per staged chunk: 3 planes x NEP concepts x CHW columns x 2 B = 96 KB
  static_assert(D % CHW == 0, "chunk width");
  constexpr int NCH = D / CHW;
  constexpr int NBLK = CHW / 128;                     // 64-column blocks per wave and chunk
  constexpr int PIECES = 3 * KG * CHW;                // 16-byte pieces per chunk (6144)
  constexpr int NP = PIECES / 512;                    // per thread (12)
  static_assert(PIECES % 512 == 0 && NP == 12, "pieces per thread");
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tslot = w >> 1, half = w & 1;
  const int li = lane & 15, lkg = lane >> 4;
  const float* Wb = W_old + R0 * d;
  float* Ob = W_new + R0 * d;
  const long tile_rows = (rows - R0) < MT * 16 ? (rows - R0) : MT * 16;
  const __amdgpu_buffer_rsrc_t rpr = __builtin_amdgcn_make_buffer_rsrc((void*)Rp, 0, 3 * NEP * d * 2, 0x00020000);
  const unsigned vo_w = (unsigned)((4 * lkg * d + 4 * li) * 4);       // rows 4 lkg + r of the tile, columns 4 li .. 4 li + 3 of the block

  // piece e of a chunk (LDS order = [p][kg][q][CHW / 4] x 16 bytes) -> its byte offset in Rp for chunk column base cb
  uint4_t rp[12];                                     // (a literal size: the lambdas below capture it - see uce_gemm.hip on dependent sizes)
  auto issue_chunk = [&](int ch) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e = tid + 512 * i;
      const int c4 = e % (CHW / 4), t = e / (CHW / 4);                 // t = (p * KG + kg) * 4 + q
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rpr, (unsigned)((t * (d / 4) + ch * (CHW / 4) + c4) * 16), 0, 16 /* sc1: past the L1 */);
    }
  };
  auto park_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) *(uint4_t*)(Rs + (tid + 512 * i) * 16) = rp[i];
  };
  auto ld_w = [&](int tile, int r, int col0) -> float4_t {
    const long row0 = (long)tile * 16 + r;
    const long sh = (row0 * d + col0) * 4;
    return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(fu_rsrc(Wb + row0 * d + col0, tile_rows * d * 4 - sh), vo_w, 0, 0));
  };

  // work items of this wave inside one chunk: (block of 64 columns, tile), blocks outer
  const int my_tiles = (MT - tslot + 3) / 4;                           // tiles tslot, tslot + 4, ...
  const int items = my_tiles * NBLK;
  auto item_col = [&](int ch, int it) { return ch * CHW + ((it / my_tiles) * 2 + half) * 64; };
  auto item_tile = [&](int it) { return tslot + 4 * (it % my_tiles); };

  // PRE: the first PRE work items' rows of W_old are loaded BEFORE the wait for R and stay in registers across it (see fused_update);
  // the others follow one item ahead.  Fully unrolled (static register indices): MAXT tiles per wave slot x NBLK blocks x NCH chunks.
  constexpr int MAXT = (MT + 3) / 4;
  constexpr int MAXI = MAXT * NBLK;                                    // items per chunk, at most
  constexpr int NPRE = PRE < MAXI * NCH ? PRE : MAXI * NCH;
  float4_t pre[NPRE > 0 ? NPRE : 1][4];
  // static item number g = (ch * NBLK + b) * MAXT + t  <->  tile tslot + 4 t (t < my_tiles), block b of chunk ch
  auto col_of = [&](int ch, int b) { return ch * CHW + (b * 2 + half) * 64; };
#pragma unroll
  for (int g = 0; g < NPRE; ++g) {
    const int t = g % MAXT, b = (g / MAXT) % NBLK, ch = g / (MAXT * NBLK);
    if (t < my_tiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pre[g][r] = ld_w(tslot + 4 * t, r, col_of(ch, b));
    }
  }
  wait();                                                              // R (and its planes) are complete
  issue_chunk(0);
  float4_t res[4];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    __syncthreads();                                                   // the previous chunk's fragment reads are done
    park_chunk();
    if (ch + 1 < NCH) issue_chunk(ch + 1);
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t >= my_tiles) continue;                                   // (wave-uniform)
        const int g = (ch * NBLK + b) * MAXT + t;
        const int tile = tslot + 4 * t, col0 = col_of(ch, b);
        const int blk = b * 2 + half;
        float4_t acc[4];                              // acc[q][r]: row tile*16 + 4*lkg + r, column col0 + 4*li + q
        if (g < NPRE) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][r] = pre[g < NPRE ? g : 0][r][q];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) res[r] = ld_w(tile, r, col0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][r] = res[r][q];
        }
        const float* trow = Ts + (tile * 16 + li) * tld + 8 * lkg;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float4_t t0 = *(const float4_t*)(trow + 32 * s), t1 = *(const float4_t*)(trow + 32 * s + 4);
          const float tx[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
          uint4_t ah, am, al;
          rp_split8(tx, ah, am, al);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned char* bq = Rs + ((((4 * s + lkg) * 4 + q) * (CHW / 4)) + blk * 16 + li) * 16;
            const uint4_t bh = *(const uint4_t*)bq;
            const uint4_t bm = *(const uint4_t*)(bq + (size_t)KG * 4 * (CHW / 4) * 16);
            const uint4_t bl = *(const uint4_t*)(bq + (size_t)2 * KG * 4 * (CHW / 4) * 16);
            acc[q] = fu_mfma(al, bh, acc[q]);           // small terms first
            acc[q] = fu_mfma(ah, bl, acc[q]);
            acc[q] = fu_mfma(am, bm, acc[q]);
            acc[q] = fu_mfma(am, bh, acc[q]);
            acc[q] = fu_mfma(ah, bm, acc[q]);
            acc[q] = fu_mfma(ah, bh, acc[q]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4_t o = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
          const long row0 = (long)tile * 16 + r;
          const long sh = (row0 * d + col0) * 4;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o), fu_rsrc(Ob + row0 * d + col0, tile_rows * d * 4 - sh), vo_w, 0,
                                                 2 /* nt */);
        }
      }
    }
  }
}

template <int D, int MT, int CT, int NK, int PRE = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lr_fused(
    const float* __restrict__ W_old, const float* __restrict__ Dm, const float* __restrict__ Csub, float* __restrict__ W_new,
    long rows, int Ne, int NEP, GramPotrfJob job) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_gram = gp_riders(job.nb);
  const int has_rider = lr_rider_blocks(job.nb, D);
  if ((int)blockIdx.x < n_gram) {
    gram_potrf_rider<D>(job, smem_raw);
    return;
  }
  if ((int)blockIdx.x < has_rider) {
    solve_rider<D>(job, smem_raw, (int)blockIdx.x - 1);   // column blocks 0 .. n_gram - 2 belong to the Gram riders
    return;
  }
  float* Wc = (float*)smem_raw;                       // [2][MT*16][PJ_LD] staging, then the T tile [MT*16][NEP + 4]
  const int tld = NEP + 4;                            // 16-byte aligned rows, 4 banks apart
  project_dispatch<D, MT, CT>(W_old, Dm, Csub, nullptr, rows, Ne, NEP, Wc, has_rider, Wc, tld);
  const long R0 = (long)((int)blockIdx.x - has_rider) * (MT * 16);
  auto wait = [&]() {
    wait_stage(job, 4);                               // R is complete (the barrier inside also closes the T tile's stores)
    if (threadIdx.x == 0) {
      // seen: count out; the last projecting block re-arms the stage word and this counter for the next launch
      const unsigned t = __hip_atomic_fetch_add(job.ticket + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (unsigned)job.n_proj - 1) {
        __hip_atomic_store(job.ticket + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(job.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if constexpr (NK == 0) {                            // split-bf16 update: the R chunk sits behind the T tile
    fused_update_b3<D, MT, CT, PRE>(W_old, job.Rp, W_new, Wc, tld, smem_raw + (size_t)MT * 16 * (64 * CT + 4) * sizeof(float), rows, R0, wait);
  } else {
    fused_update<D, MT, NK, PRE>(W_old, job.R, W_new, Wc, tld, rows, Ne, R0, wait);
  }
}

template <int D, int MT, int CT, int NK, int PRE = 0>
int launch_fused(const float* W_old, const float* G, const float* Csub, float* W_new, long rows, int N_edit, int NEP64,
                 GramPotrfJob job, hipStream_t st) {
  size_t smem = (size_t)2 * (MT * 16 + 64 * CT) * PJ_LD * sizeof(float);
  if (smem < gp_smem(job.nb)) smem = gp_smem(job.nb);
  size_t ttile = (size_t)MT * 16 * (NEP64 + 4) * sizeof(float);
  if (NK == 0) ttile += 96 * 1024;                    // + the staged chunk of R's bf16 planes
  if (smem < ttile) smem = ttile;
  if (smem > 160 * 1024) return UCE_EINVAL;
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_lr_fused<D, MT, CT, NK, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  const long n_proj = (rows + MT * 16 - 1) / (MT * 16);
  const long nwg = n_proj + lr_rider_blocks(job.nb, D);
  if (n_proj > 0x3fffffffL) return UCE_EINVAL;
  job.fused = 1;
  job.n_proj = (int)n_proj;
  hipLaunchKernelGGL((k_lr_fused<D, MT, CT, NK, PRE>), dim3((unsigned)nwg), dim3(512), smem, st, W_old, G, Csub, W_new, rows, N_edit, NEP64,
                     job);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

// NK = k-steps of 4 concepts the update walks (rows of R beyond N_edit read as zeros, the T tile's padding columns are zero):
// 13 / 16 for one 64-concept tile row, 25 / 32 for two
template <int D, int MT, int PRE = 0>
int launch_fused_d(const float* W_old, const float* G, const float* Csub, float* W_new, long rows, int N_edit, int NEP64,
                   const GramPotrfJob& job, hipStream_t st) {
  if (job.Rp) {                                        // split-bf16 update (NK = 0 marks it)
    if (NEP64 <= 64) return launch_fused<D, MT, 1, 0, PRE>(W_old, G, Csub, W_new, rows, N_edit, NEP64, job, st);
    return launch_fused<D, MT, 2, 0, PRE>(W_old, G, Csub, W_new, rows, N_edit, NEP64, job, st);
  }
  const int nks = (N_edit + 3) / 4;
  if (NEP64 <= 64) {
    if (nks <= 13) return launch_fused<D, MT, 1, 13, PRE>(W_old, G, Csub, W_new, rows, N_edit, NEP64, job, st);
    return launch_fused<D, MT, 1, 16, PRE>(W_old, G, Csub, W_new, rows, N_edit, NEP64, job, st);
  }
  if (nks <= 25) return launch_fused<D, MT, 2, 25, PRE>(W_old, G, Csub, W_new, rows, N_edit, NEP64, job, st);
  return launch_fused<D, MT, 2, 32, PRE>(W_old, G, Csub, W_new, rows, N_edit, NEP64, job, st);
}

}  // namespace

// W_new = W_old + (W_old (G - C_e)^T) R with R = rows of (lamb S^-1 + C C^T)^-1 C, N <= 128 concepts, in ONE launch.
// h->ticket / slabs / Lmat / Linv / status / R as launch_lr_project's rider form.
int launch_lr_fused(const float* W_old, const float* G, const float* C, const float* s, float* W_new, long rows, int d, int N,
                    int N_edit, float lamb, uce_ctx* h, hipStream_t st, unsigned short* Rp, int pre) {
  const int NEP64 = (N_edit + 63) / 64 * 64;
  const int nb = (N + 63) / 64;
  if (!h || nb < 1 || nb > GP_MAXB || N_edit < 1 || N_edit > 128 || !C || !s || !G) return UCE_EINVAL;
  GramPotrfJob job{C, s, N, lamb, h->slabs, h->ticket, h->Lmat, h->Linv, h->status, nb, h->R, N_edit, Rp, NEP64, 1, 0};
  if (d == 768 && pre >= 12) return launch_fused_d<768, 7, 12>(W_old, G, C, W_new, rows, N_edit, NEP64, job, st);
  if (d == 768 && pre >= 8) return launch_fused_d<768, 7, 8>(W_old, G, C, W_new, rows, N_edit, NEP64, job, st);
  if (d == 768) return launch_fused_d<768, 7>(W_old, G, C, W_new, rows, N_edit, NEP64, job, st);
  if (d == 1024) return launch_fused_d<1024, 7>(W_old, G, C, W_new, rows, N_edit, NEP64, job, st);
  if (d == 2048) return launch_fused_d<2048, 5>(W_old, G, C, W_new, rows, N_edit, NEP64, job, st);
  return UCE_EINVAL;
}
