// Split-contraction hand-off shared by the direct-to-LDS GEMM / convolution kernels (uce_gemm.hip, uce_conv_dma.hip) in the
// few-tile regime (SURVEY.md section 8(f) row 3 at the reference's own call pattern: one prompt per `pipe()` call,
// evalscripts/generate-images-sd.py:37-42, and the 16 x 16 / 8 x 8 levels of the U-Net at any batch): S workgroups share one
// output tile, each contracting its own range of k-tiles.
//
//   slab (tile, s): BM x BN floats in REGISTER order - float4 number q of thread t sits at float4 index q * NT + t (NT threads), so the
//   stores and the read-back are 16 bytes per lane, lane-contiguous, and nothing about the accumulator layout matters;
//   ticket[tile]  : arrivals so far; all zero between launches (the last arriver re-arms it: the launch can be captured into a
//                   hipGraph and replayed).
// Protocol = the split-K recipe of the CDNA guide (Guideline 16) in the write-through form uce_lowrank_riders.h uses:
// sc1 slab stores -> per-wave vmcnt(0) -> barrier -> one lane draws a relaxed agent-scope ticket; the workgroup that draws the
// LAST ticket re-reads all S slabs with sc1 loads (past the L1; summed in slab order, its own slab included: bit-repeatable
// whatever the arrival order) and runs the kernel's one epilogue.  No release / acquire fence on either side (an agent-scope
// release would write back every dirty line of the XCD's L2).
#pragma once
#include "uce_common.h"

namespace uce_sk {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

// Returns true for the workgroup that holds the reduced accumulators (the last arriver); the others are done.
// `smem`: 4 bytes of LDS nobody else uses until the function returns (the staging ring is drained).
template <int TM, int TN, int TILE_ELEMS, int NT = 512>
__device__ __forceinline__ bool reduce(float16_t (&acc)[TN][TM], float* __restrict__ ws, unsigned* __restrict__ tick, long tile,
                                       int ks, int S, unsigned char* smem, int tid) {
  static_assert(TILE_ELEMS == NT * 16 * TM * TN, "NT threads x their accumulators");
  constexpr int NQ = 4 * TM * TN;                                        // float4 per thread
  float* base = ws + (size_t)tile * S * TILE_ELEMS;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((size_t)S * TILE_ELEMS * 4), 0x00020000);
  {
    const unsigned off0 = (unsigned)(((size_t)ks * TILE_ELEMS) * 4 + (size_t)tid * 16);
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4_t v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, v), r,
                                                 off0 + (unsigned)((((a * TM + b) * 4 + q) * NT) * 16), 0, 16 /* sc1 */);
        }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(tick + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)S - 1) __hip_atomic_store(tick + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
    *(volatile unsigned*)smem = t;
  }
  __syncthreads();
  const unsigned mine = *(volatile unsigned*)smem;
  __syncthreads();                                                       // everybody has read the word: the LDS is free again
  if (mine != (unsigned)S - 1) return false;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  // slabs in groups of SG: the loads of a whole group are in flight at once (an sc1 load is a round trip past the L2 - one slab at
  // a time made the last arriver of a 10-way split the longest part of a small layer), the additions keep the slab order
  constexpr int SG = NQ <= 4 ? 4 : NQ <= 8 ? 2 : 1;
  for (int s0 = 0; s0 < S; s0 += SG) {
    float4_t v[SG][NQ];
#pragma unroll
    for (int g = 0; g < SG; ++g) {
      // (slabs past the last one: an out-of-range offset reads zeros without touching memory)
      const unsigned off0 = s0 + g < S ? (unsigned)(((size_t)(s0 + g) * TILE_ELEMS) * 4 + (size_t)tid * 16) : 0x80000000u;
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        v[g][i] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(r, off0 == 0x80000000u ? off0 : off0 + (unsigned)(i * NT * 16), 0, 16 /* sc1 */));
    }
#pragma unroll
    for (int g = 0; g < SG; ++g)
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4_t x = v[g][(a * TM + b) * 4 + q];
            acc[a][b][4 * q] += x[0];
            acc[a][b][4 * q + 1] += x[1];
            acc[a][b][4 * q + 2] += x[2];
            acc[a][b][4 * q + 3] += x[3];
          }
  }
  return true;
}

// How many ways to split the NK k-tiles of a layer with T output tiles.  Measured on an MI355X (tools/probe_r05_sk.py sweep, 128 x 64
// tiles, one prompt per call): the best S puts about two workgroups on every CU (T S ~ 480: the tiles are bound by LDS bandwidth,
// a second workgroup fills the CU's idle issue slots) as long as each keeps >= 10 k-tiles (below that the slab round trip - write-
// through stores, ticket, read-back: ~5 us - costs more than the shorter loop saves); at most 16 slabs (the last arriver reads all).
//   conv 320 -> 320 @ 64 x 64 x 2 (T = 320): S = 2 33.0 us (S = 1 34.6);  640 @ 32 x 32 (T = 160): S = 3 31.7 (S = 1 56.2);
//   1280 @ 16 x 16 (T = 80): S = 6 32.7 (S = 1 106.8);  1280 @ 8 x 8 (T = 20): S = 12 20.8 (S = 1 104.7);
//   linear 2048 x 640 x 640 (10 k-tiles): S = 1 7.0 (S = 2 12.2);  512 x 1280 x 5120: S = 6 21.4 (S = 1 46.4)
inline int choose_split(long T, int NK, int min_kt, int force = 0) {
  if (force > 0) return force > NK ? NK : force;                        // (UCE_SK_SPLIT: measurements)
  long S = (480 + T / 2) / T;
  if (S > 16) S = 16;
  if (S > NK / min_kt) S = NK / min_kt;
  return S < 1 ? 1 : (int)S;
}

}  // namespace uce_sk
