// The register-resident form of the low-rank edit  W_new = W_old + (W_old D_e^T) R_e  for d = 768, N <= 128 (reference: the per-module
// `mat1 @ torch.inverse(mat2)` of trainscripts/uce_sd_erase.py:56-82, collapsed as DESIGN.md section 2 derives) - ONE launch, W_old read
// from HBM ONCE and never again:
//
//   * a main workgroup (7 computing waves + one LDS-DMA producer wave) owns 112 rows; a wave owns 16 of them and keeps its 16 x 768 fp32
//     tile of W_old in 144 registers + 12 KB of LDS from the first load to the final store (SD-1.4's 24 960 x 768 slab = 223
//     workgroups beside the 29 / 33 rider workgroups: one per CU), laid out as the accumulator tiles of the TRANSPOSED update: tile t
//     (16 columns), lane (n = lane & 15, kg = lane >> 4) holds W[row n][16 t + 4 kg + 0..3];
//   * both products run on the f16 matrix cores with fp32-equivalent operands - the two-term split of uce_apply_h2.hip (x s = x_h +
//     x_l under a power-of-two scale per row / column / concept, three MFMAs l*h + h*l + h*h per product, fp32 accumulation): the
//     exact-f32 MFMA form of the two-launch path issues 2 * 2 * rows * d * 64 flop at 157 TF/s (31 us at 50 concepts), this one 3 x that
//     at 2.5 PF/s (6 us), so the step's floor is its HBM traffic (W in once, W out once);
//   * phase A  T^T = D_e W^T : the contraction index (W's column) may be permuted as long as both operands agree, and the
//     accumulator layout above IS a valid B-operand layout of v_mfma_f32_16x16x32_f16 for the k-block of 32 columns made of tiles
//     2b and 2b + 1 (slot j of lane kg <-> column 32 b + 16 (j >> 2) + 4 kg + (j & 3)); the A operand - D_e = G - C_e as f16 (hi,
//     lo) fragments in exactly that order - is prepared ONCE by "D-prep" rider workgroups of the same launch (one per 16 concepts)
//     while the main workgroups wait for their W tiles, and streamed through a three-stage LDS ring by the producer wave;
//   * T stays in registers too: the output tile of phase A (lane: row n, concepts 16 ct + 4 kg + 0..3) is, after its own split, the
//     B operand of phase B  W_new^T += R^T T^T  under the same permutation trick on the concept index;
//   * the small-system chain (Gram -> Cholesky -> solves, uce_lowrank_riders.h) rides in the first workgroups of the launch as in the
//     two-launch form; its solve riders publish R as f16 (hi, lo) fragments + per-column scales (write-through) and count out; the
//     main workgroups - T ready, W in registers - wait for that counter, stream the fragments through the same ring and store
//     (phase B sits on the write floor of HBM: tools/ubench/store_pattern.hip).
//
// Hand-off words (h->ticket, zero between launches): [0] [1] the riders' own, [2] solve riders done (the main workgroups wait for all
// of them - no "last rider posts a stage" hop), [3] main workgroups done (the last one re-arms [1] .. [4]), [4] D-prep riders done.
// Riders come FIRST in the grid: whatever the CU count, they are resident before any main workgroup waits for them.
#define UCE_DBG_SYM g_dbg_res    // -DUCE_CHAIN_DEBUG: this launch's own stamp buffer (tools/dbg_chain.py --resident)
#define UCE_DBG_READ uce_debug_read_res
#include "uce_lowrank_riders.h"

namespace {

typedef _Float16 rs_f16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void rs_lds_void;

constexpr int RS_D = 768;
constexpr int RS_NT = RS_D / 16;          // 48 column tiles of a wave's W tile
constexpr int RS_NKB = RS_D / 32;         // 24 k-blocks of phase A
// NCT = concept tiles of 16 (template parameter of everything below): 4 -> N <= 64 (one 64-block system), 8 -> N <= 128 (two blocks)
constexpr int RS_CW = 7;                  // compute waves of a main workgroup (16 rows each); wave 7 is the LDS-DMA producer
constexpr int RS_ROWS = 16 * RS_CW;       // rows per main workgroup
constexpr int RS_STAGE = 16 * 1024;       // LDS ring stage: 8 / NCT k-blocks of D fragments, 16 / NCT column tiles of R fragments
__host__ __device__ constexpr int rs_nstage(int nct) { return 3 * nct; }       // stages per operand (12 / 24)
#ifndef RS_ABL
#define RS_ABL 0          // measurement builds only: 1 = phase B without its stores, 2 = without its MFMAs
#endif
// Cache policy of the W_new stores.  The nt hint costs 2-5 us of phase B (ablation_stores_mfma_nt.log).  Plain stores leave up to
// 8 x 4 MB of the 76.7 MB dirty in the L2s, written back when the kernel ends - behind everything else; write-through (sc1) stores
// take that out of the tail: whole step at up to 64 concepts 47.4-48.3 -> 45.8-46.4 us same box, three alternating runs
// (profiles/r06/resident/ab_store_policy.log).  At 65-128 concepts the stores are spread over a phase B twice as long and plain
// stores are the faster by ~1 us.  -DRS_ST_AUX=n forces one policy (measurement builds).
#ifdef RS_ST_AUX
template <int NCT> constexpr int rs_st_aux() { return RS_ST_AUX; }
#else
template <int NCT> constexpr int rs_st_aux() { return NCT <= 4 ? 16 : 0; }
#endif
// Cache policy of the fragment loads: sc1 (past the L1 and this XCD's L2), like every other consumer of a payload another workgroup of
// the SAME launch published write-through (uce_lowrank_riders.h: st_sc1).  A plain load measured the same (the producer wave hides
// the latency either way) and is only coherent if no stale line of the previous launch's fragments survives in this XCD's L2.
#ifndef RS_DMA_AUX
#define RS_DMA_AUX 16
#endif
constexpr int RS_NREG = 36;               // column tiles of the W tile held in registers (144 VGPRs) ...
constexpr int RS_NLDS = RS_NT - RS_NREG;  // ... and in LDS (12 KB per wave): 192 + the working set of either phase does not fit 256 registers
// main workgroup LDS: ring [3][16 KB] | R column scales [768] | D concept scales [128] | W tail tiles [7 waves][12][1 KB]
constexpr int RS_RING = 3;                // ring stages: two DMA stages in flight while one is computed on
constexpr int RS_LDS_RSC = RS_RING * RS_STAGE;
constexpr int RS_LDS_DSC = RS_LDS_RSC + RS_D * 4;
constexpr int RS_LDS_WT = RS_LDS_RSC + 4 * 1024;
constexpr size_t RS_MAIN_SMEM = RS_LDS_WT + RS_CW * RS_NLDS * 1024;
static_assert(RS_LDS_DSC + 64 * 4 <= RS_LDS_WT && RS_NREG % 4 == 0, "LDS map");

__device__ __forceinline__ float4_t rs_mfma(uint4_t a, uint4_t b, float4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rs_f16x8_t, a), __builtin_bit_cast(rs_f16x8_t, b), c, 0, 0, 0);
}

struct ResidentJob {
  const float* G;          // [N_edit, d] targets
  const float* Ce;         // [N_edit, d] = the first rows of C
  int Ne;
  uint4_t* Dh;             // D fragments  [k-block 24][concept tile NCT][plane 2][lane 64] x 16 B
  float* Dsc;              // [16 NCT]  2^-e of the concept rows of D_e
  uint4_t* Rh;             // R fragments  [column tile 48][k-block NCT / 2][plane 2][lane 64] x 16 B   (written by the solve riders)
  float* Rsc;              // [768] 2^-e of the columns of R
  int n_main;
};

// ---------------------------------------------------------------------------------------------------------------------------
// D-prep rider ct: concepts 16 ct .. 16 ct + 15 of D_e = G - C_e -> scale per concept, (hi, lo) fragments in phase A's order
// ---------------------------------------------------------------------------------------------------------------------------
template <int NCT>
__device__ __forceinline__ void rs_dprep(const ResidentJob& rj, unsigned* ticket, unsigned char* smem_raw, int ct) {
  unsigned* mx = (unsigned*)smem_raw;                       // [16] row maxima (fp32 bit patterns of |x|)
  const int tid = threadIdx.x;
  const int i = tid & 15;                                   // the concept of this thread's items: item = tid + 512 p, lane = item & 63
  const int concept = 16 * ct + i;
  const bool live = concept < rj.Ne;
  DBG(0);
  if (tid < 16) mx[tid] = 0u;
  float x[3][8];
  unsigned m = 0u;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int item = tid + 512 * p, b = item >> 6, kg = (item >> 4) & 3;
    float4_t g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, c0 = g0, c1 = g0;
    if (live) {
      const size_t o = (size_t)concept * RS_D + 32 * b + 4 * kg;
      g0 = *(const float4_t*)(rj.G + o);
      g1 = *(const float4_t*)(rj.G + o + 16);
      c0 = *(const float4_t*)(rj.Ce + o);
      c1 = *(const float4_t*)(rj.Ce + o + 16);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[p][e] = g0[e] - c0[e];
      x[p][4 + e] = g1[e] - c1[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned a = __float_as_uint(x[p][e]) & 0x7fffffffu;
      m = a > m ? a : m;
    }
  }
  __syncthreads();
  atomicMax(&mx[i], m);
  __syncthreads();
  const int E = rs_clamp_exp(mx[i]);
  const float s = rs_scale(E);
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)rj.Dh, 0, RS_NKB * NCT * 2 * 1024, 0x00020000);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int item = tid + 512 * p, b = item >> 6, lane = item & 63;
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = x[p][e] * s;
    uint4_t hi, lo;
    rs_split8(y, hi, lo);
    const unsigned off = (unsigned)((((b * NCT + ct) * 2) * 64 + lane) * 16);
    __builtin_amdgcn_raw_buffer_store_b128(hi, dr, off, 0, 16 /* sc1: write-through */);
    __builtin_amdgcn_raw_buffer_store_b128(lo, dr, off + 1024, 0, 16);
  }
  if (tid < 16) __hip_atomic_store(rj.Dsc + 16 * ct + tid, rs_inv_scale(rs_clamp_exp(mx[tid])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(ticket + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  DBG(1);
}

// one lane polls `word` until it reaches `want` (bounded; a give-up is reported through the status word), then a barrier
__device__ __forceinline__ void rs_wait_word(const unsigned* word, unsigned want, int* status) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) {
        atomicCAS(status, 0, -1);
        break;
      }
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------
// main workgroup
// ---------------------------------------------------------------------------------------------------------------------------
template <int NCT>
__device__ __forceinline__ void rs_main(const float* __restrict__ W_old, float* __restrict__ W_new, long rows, const ResidentJob& rj,
                                        const GramPotrfJob& job, unsigned char* smem, int wg) {
  constexpr int NSTG = rs_nstage(NCT);      // ring stages per operand
  constexpr int KPS = 8 / NCT;              // k-blocks of D per stage (a k-block = NCT x 2 fragments of 1 KB) = column-tile groups of R
  constexpr int NKC = NCT / 2;              // k-blocks of 32 concepts in phase B
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kg = lane >> 4;
  // ---- LDS ring: stage s of `src` = 16 contiguous 1 KB fragments -> ring buffer s % 3 by LDS-DMA.  Wave 7 is the producer: it
  // issues every piece (an LDS-DMA issue costs a computing wave 100-185 cycles inside a busy phase) and is the only wave that
  // waits on them; the 7 computing waves meet it at one barrier per stage.  Two stages are in flight while one is read.
  if (w == RS_CW) {
    auto dma_stage = [&](const uint4_t* src, int s) {
      const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, NSTG * RS_STAGE, 0x00020000);
#pragma unroll
      for (int c = 0; c < 16; ++c)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(sr, (rs_lds_void*)(smem + (s % RS_RING) * RS_STAGE + c * 1024), 16,
                                                 (unsigned)(s * RS_STAGE + lane * 16), c * 1024, 0, RS_DMA_AUX);
    };
    auto run = [&](const uint4_t* src, int last_wait) {
      __syncthreads();                                        // (the computing waves: the wait for the producer of `src`)
      dma_stage(src, 0);
      dma_stage(src, 1);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // stage 0 landed (the 16 pieces of stage 1 may be outstanding)
      __syncthreads();
#pragma unroll
      for (int s = 0; s + last_wait < NSTG; ++s) {
        if (s + 2 < NSTG) {
          dma_stage(src, s + 2);
          asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // stage s + 1 landed
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
      }
    };
    run(rj.Dh, 0);
    run(rj.Rh, 1);                                            // (no barrier behind the last stage of phase B)
    return;
  }
  const long R0 = (long)wg * RS_ROWS + 16 * w;
  long vr = rows - R0;
  vr = vr < 0 ? 0 : (vr > 16 ? 16 : vr);
  const int w_bytes = (int)vr * RS_D * 4;                   // this wave's rows that exist: beyond them loads give 0, stores nothing
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)(W_old + R0 * RS_D), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)(W_new + R0 * RS_D), 0, w_bytes, 0x00020000);
  const unsigned vo = (unsigned)((n * RS_D + 4 * kg) * 4);

  DBG(0);
  // ---- W tile: 48 loads of 16 B per lane, all in flight - tiles 0..35 into registers, 36..47 straight into this wave's LDS slots
  // (the tile displacement rides in the SCALAR offset, which the hardware range check ignores: `vo` alone says whether the lane's
  //  row exists - and the compiler cannot turn 48 displaced offsets into 48 live registers)
  float4_t wv[RS_NREG];
#pragma unroll
  for (int t = 0; t < RS_NREG; ++t)
    wv[t] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(wr, vo, 64 * t, 0));
  unsigned char* wtail = smem + RS_LDS_WT + w * (RS_NLDS * 1024);
#pragma unroll
  for (int t = RS_NREG; t < RS_NT; ++t)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (rs_lds_void*)(wtail + (t - RS_NREG) * 1024), 16, vo, 64 * t, 0, 0);
  auto wt = [&](int t) -> float4_t { return *(const float4_t*)(wtail + (t - RS_NREG) * 1024 + lane * 16); };
  // row maximum -> scale of the row (lanes n, n + 16, n + 32, n + 48 hold one row)
  float mxf = 0.f;
#pragma unroll
  for (int t = 0; t < RS_NREG; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) mxf = fmaxf(mxf, fabsf(wv[t][r]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the DMA'd tiles (this wave's own slots: no barrier needed)
#pragma unroll
  for (int t = RS_NREG; t < RS_NT; ++t) {
    const float4_t v = wt(t);
#pragma unroll
    for (int r = 0; r < 4; ++r) mxf = fmaxf(mxf, fabsf(v[r]));
  }
  mxf = fmaxf(mxf, __shfl_xor(mxf, 16));
  mxf = fmaxf(mxf, __shfl_xor(mxf, 32));
  DBG(1);
  const int EW = rs_clamp_exp(__float_as_uint(mxf));
  const float sW = rs_scale(EW), iW = rs_inv_scale(EW);
  float* Rsc = (float*)(smem + RS_LDS_RSC);
  float* Dsc = (float*)(smem + RS_LDS_DSC);

  // ---- phase A: T^T = D_e W^T
  rs_wait_word(job.ticket + 4, (unsigned)NCT, job.status);
  DBG(2);
  if (tid < 4 * NCT) {
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)rj.Dsc, 0, 16 * NCT * 4, 0x00020000);
    *(uint4_t*)(Dsc + 4 * tid) = __builtin_amdgcn_raw_buffer_load_b128(sr, (unsigned)(16 * tid), 0, 16);
  }
  __syncthreads();                                            // stage 0 of the D fragments has landed
  float4_t tacc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) tacc[c] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NSTG; ++s) {
#pragma unroll
    for (int q = 0; q < KPS; ++q) {
      const int b = KPS * s + q;
      const unsigned char* fb = smem + (s % RS_RING) * RS_STAGE + q * (NCT * 2048) + lane * 16;
      const float4_t w0 = 2 * b < RS_NREG ? wv[2 * b < RS_NREG ? 2 * b : 0] : wt(2 * b);
      const float4_t w1 = 2 * b + 1 < RS_NREG ? wv[2 * b + 1 < RS_NREG ? 2 * b + 1 : 0] : wt(2 * b + 1);
      float y[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = w0[e] * sW;
        y[4 + e] = w1[e] * sW;
      }
      uint4_t wh, wl;
      rs_split8(y, wh, wl);
#pragma unroll
      for (int c0 = 0; c0 < NCT; c0 += 4) {                 // four concept tiles at a time: four independent accumulator chains
        uint4_t dh[4], dl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          dh[c] = *(const uint4_t*)(fb + (2 * (c0 + c)) * 1024);
          dl[c] = *(const uint4_t*)(fb + (2 * (c0 + c) + 1) * 1024);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) tacc[c0 + c] = rs_mfma(dl[c], wh, tacc[c0 + c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) tacc[c0 + c] = rs_mfma(dh[c], wl, tacc[c0 + c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) tacc[c0 + c] = rs_mfma(dh[c], wh, tacc[c0 + c]);
      }
    }
    __syncthreads();                                          // stage s + 1 landed (producer wave), stage s's buffer is free
  }

  DBG(3);
  // ---- T (lane: row n, concepts 16 ct + 4 kg + r): undo the operand scales, row maximum, scale, split -> B fragments of phase B
  float tmx = 0.f;
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const float4_t ds = *(const float4_t*)(Dsc + 16 * c + 4 * kg);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      tacc[c][r] = tacc[c][r] * iW * ds[r];
      tmx = fmaxf(tmx, fabsf(tacc[c][r]));
    }
  }
  tmx = fmaxf(tmx, __shfl_xor(tmx, 16));
  tmx = fmaxf(tmx, __shfl_xor(tmx, 32));
  const int ET = rs_clamp_exp(__float_as_uint(tmx));
  const float sT = rs_scale(ET), iT = rs_inv_scale(ET);
  uint4_t th[NKC], tl[NKC];
#pragma unroll
  for (int b = 0; b < NKC; ++b) {
    float y[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[e] = tacc[2 * b][e] * sT;
      y[4 + e] = tacc[2 * b + 1][e] * sT;
    }
    rs_split8(y, th[b], tl[b]);
  }

  DBG(4);
  // ---- wait for the chain: every solve rider has counted itself out on ticket[2] (its R fragments drained before that)
  rs_wait_word(job.ticket + 2, (unsigned)(RS_D / SV_COLS), job.status);
  DBG(5);
  if (tid < RS_D / 4) {
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)rj.Rsc, 0, RS_D * 4, 0x00020000);
    *(uint4_t*)(Rsc + 4 * tid) = __builtin_amdgcn_raw_buffer_load_b128(sr, (unsigned)(16 * tid), 0, 16);
  }
  __syncthreads();                                            // stage 0 of the R fragments (and the scales) are in LDS

  // ---- phase B: W_new^T tile t = W^T tile t + R^T T^T.  A group = 8 / NCT column tiles = four independent accumulator chains (tile x
  // k-block of 32 concepts) of three MFMAs each (a dependent MFMA two issues behind its producer stalls on the accumulator)
#pragma unroll
  for (int s = 0; s < NSTG; ++s) {
#ifdef UCE_CHAIN_DEBUG
    if (s < 8) DBG(8 + s);
#endif
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned char* fb = smem + (s % RS_RING) * RS_STAGE + q * 8192 + lane * 16;
      uint4_t rh[KPS][NKC], rl[KPS][NKC];                     // [tile of the group][k-block]
#pragma unroll
      for (int u = 0; u < KPS; ++u)
#pragma unroll
        for (int b = 0; b < NKC; ++b) {
          rh[u][b] = *(const uint4_t*)(fb + ((u * NKC + b) * 2) * 1024);
          rl[u][b] = *(const uint4_t*)(fb + ((u * NKC + b) * 2 + 1) * 1024);
        }
      float4_t ac2[KPS][NKC];
#pragma unroll
      for (int u = 0; u < KPS; ++u)
#pragma unroll
        for (int b = 0; b < NKC; ++b) ac2[u][b] = (float4_t){0.f, 0.f, 0.f, 0.f};
#if RS_ABL != 2
#pragma unroll
      for (int b = 0; b < NKC; ++b)
#pragma unroll
        for (int u = 0; u < KPS; ++u) ac2[u][b] = rs_mfma(rl[u][b], th[b], ac2[u][b]);
#pragma unroll
      for (int b = 0; b < NKC; ++b)
#pragma unroll
        for (int u = 0; u < KPS; ++u) ac2[u][b] = rs_mfma(rh[u][b], tl[b], ac2[u][b]);
#pragma unroll
      for (int b = 0; b < NKC; ++b)
#pragma unroll
        for (int u = 0; u < KPS; ++u) ac2[u][b] = rs_mfma(rh[u][b], th[b], ac2[u][b]);
#endif
      // (the sums of the chains are formed for the whole group BEFORE the first store of the group is built: the same statements in
      //  one loop per tile - `acc = ac2[u][0]; acc += ac2[u][1]; ... store` - compiled to wrong results with hipcc 7.2.0 at -O3, with or
      //  without wait states behind the MFMAs: not a read-after-MFMA hazard; tests/test_edit_gpu.py's changing-input test is the guard)
      float4_t accs[KPS];
#pragma unroll
      for (int u = 0; u < KPS; ++u) {
        accs[u] = ac2[u][0];
#pragma unroll
        for (int b = 1; b < NKC; ++b) accs[u] = accs[u] + ac2[u][b];
      }
#pragma unroll
      for (int u = 0; u < KPS; ++u) {
        const float4_t acc = accs[u];
        const int t = 2 * KPS * s + KPS * q + u;
        const float4_t rs = *(const float4_t*)(Rsc + 16 * t + 4 * kg);
        const float4_t w0 = t < RS_NREG ? wv[t < RS_NREG ? t : 0] : wt(t);
        float4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_fmaf(acc[r], iT * rs[r], w0[r]);
#if RS_ABL == 1
        if (o[0] == 12345.678f)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o), orr, vo, 64 * t, rs_st_aux<NCT>());
      }
    }
    // (the stores stay in flight: nothing in this loop waits on the vector-memory counter)
    if (s + 1 < NSTG) __syncthreads();
  }
  DBG(6);
  // seen the chain's result -> count out; the last main workgroup re-arms the hand-off words for the next launch (every rider has
  // finished: the solve riders counted out before any main workgroup got here, the others long before)
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(job.ticket + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)rj.n_main - 1) {
      __hip_atomic_store(job.ticket + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(job.ticket + 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(job.ticket + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(job.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  DBG(7);
}

template <int NCT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lr_resident(
    const float* __restrict__ W_old, float* __restrict__ W_new, long rows, GramPotrfJob job, ResidentJob rj) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
  constexpr int NB = NCT / 4;                                 // 64-blocks of the dual system
  const int n_gram = gp_riders(NB);
  const int n_chain = lr_rider_blocks(NB, RS_D);
  const int bx = (int)blockIdx.x;
  if (bx < n_gram) {
    gram_potrf_rider<RS_D>(job, smem_raw);
    return;
  }
  if (bx < n_chain) {
    solve_rider<RS_D>(job, smem_raw, bx - 1);                 // column blocks 0 .. n_gram - 2 belong to the Gram riders
    return;
  }
  if (bx < n_chain + NCT) {                                   // D-prep riders: one per concept tile
    rs_dprep<NCT>(rj, job.ticket, smem_raw, bx - n_chain);
    return;
  }
  rs_main<NCT>(W_old, W_new, rows, rj, job, smem_raw, bx - n_chain - NCT);
}

template <int NCT>
int rs_launch(uce_ctx* h, const float* W_old, const float* G, const float* C, const float* s, float* W_new, long rows, int N, int N_edit,
              float lamb, unsigned char* ws, hipStream_t st) {
  constexpr int NB = NCT / 4;
  constexpr int NSTG = rs_nstage(NCT);
  uint4_t* Dh = (uint4_t*)ws;
  uint4_t* Rh = (uint4_t*)(ws + NSTG * RS_STAGE);
  float* Dsc = (float*)(ws + 2 * NSTG * RS_STAGE);
  float* Rsc = Dsc + 16 * NCT;
  const int n_main = (int)((rows + RS_ROWS - 1) / RS_ROWS);
  GramPotrfJob job{C, s, N, lamb, h->slabs, h->ticket, h->Lmat, h->Linv, h->status, NB, h->R, N_edit, nullptr, 16 * NCT, 1, n_main};
  job.Rh = Rh;
  job.Rsc = Rsc;
  ResidentJob rj{G, C, N_edit, Dh, Dsc, Rh, Rsc, n_main};
  size_t smem = gp_smem(NB) > RS_MAIN_SMEM ? gp_smem(NB) : RS_MAIN_SMEM;
  static PerDeviceOnce attr_once;
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_lr_resident<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_once.commit(tok);
  }
  const int nwg = lr_rider_blocks(NB, RS_D) + NCT + n_main;
  hipLaunchKernelGGL(k_lr_resident<NCT>, dim3((unsigned)nwg), dim3(512), smem, st, W_old, W_new, rows, job, rj);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

}  // namespace

bool lr_resident_supported(int d, int N, int N_edit, long rows) {
  return d == RS_D && N >= 1 && N <= 128 && N_edit >= 1 && N_edit <= N && rows >= 1;
}

size_t lr_resident_ws_bytes() {
  return (size_t)2 * rs_nstage(8) * RS_STAGE + (128 + RS_D) * sizeof(float);
}

// ws: lr_resident_ws_bytes() of handle workspace (D fragments | R fragments | D scales | R scales)
int launch_lr_resident(uce_ctx* h, const float* W_old, const float* G, const float* C, const float* s, float* W_new, long rows,
                       int N, int N_edit, float lamb, unsigned char* ws, hipStream_t st) {
  if (!lr_resident_supported(RS_D, N, N_edit, rows) || !h || !ws) return UCE_EINVAL;
  if (N <= 64) return rs_launch<4>(h, W_old, G, C, s, W_new, rows, N, N_edit, lamb, ws, st);
  return rs_launch<8>(h, W_old, G, C, s, W_new, rows, N, N_edit, lamb, ws, st);
}
