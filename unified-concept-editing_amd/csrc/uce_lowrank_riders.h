// Device code shared by the launches that carry the low-rank apply (uce_lowrank2.hip: projection / update as two launches;
// uce_lowrank_fused.hip: both in one): the projection body, the rider blocks of the small-system chain (Gram -> blocked
// Cholesky -> triangular solves -> R) and their hand-off words.  See uce_lowrank2.hip for the design notes.
#pragma once
#include "uce_common.h"
#include <type_traits>
#include "uce_potrf64.h"
#include "uce_potrf_la.h"

// -DUCE_CHAIN_DEBUG: wall-clock stamps (100 MHz) of the rider chain's phases, read back with uce_debug_read
// (tools/dbg_chain.py); compiled out of the product library.
#ifdef UCE_CHAIN_DEBUG
#ifndef UCE_DBG_SYM                            // (a second translation unit that hosts the riders names its own buffer + reader)
#define UCE_DBG_SYM g_dbg
#define UCE_DBG_READ uce_debug_read
#endif
#define g_dbg UCE_DBG_SYM
__device__ unsigned long long g_dbg[64][32];   // per block: 16 wall-clock stamps (slots 0-7 Gram / factor / projection role, 8-15 solve role) + 16 shader-clock stamps
#define DBG(slot) do { if (threadIdx.x == 0 && blockIdx.x < 64) { g_dbg[blockIdx.x][slot] = wall_clock64(); g_dbg[blockIdx.x][16 + slot] = clock64(); } } while (0)
extern "C" int UCE_DBG_READ(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(g_dbg));
}
#else
#define DBG(slot) do { } while (0)
#endif

namespace {

constexpr int PJ_KC = 64;    // floats per W k-chunk
constexpr int PJ_LD = 72;    // LDS row stride of the chunk (floats): conflict-free b128 fragment reads

// ---------------------------------------------------------------------------------------------
// projection: 8 waves, wave = (concept tile class c4 = w & 3, M half = w >> 2); the W k-chunk of the MT*16-row
// super-tile and the D_e k-chunk (64 * CT concepts, formed on the fly from G and C_e) are staged once in LDS and shared
// by the waves; each W fragment read from LDS feeds 4 * CT MFMAs, each D_e fragment NMT.
// CT = 2 (64 < N_edit): wave c4 owns the 16-concept tiles c4 and c4 + 4 of a 128-concept batch, so the weights are
// streamed ONCE per 128 concepts (round 2 walked W once per 64 concepts: two passes at the north-star's "100
// concepts", 2.08x the step's algorithmic traffic, 0.144 ms; this form 0.104 ms).
//  * ONE register set per operand stream: chunk c + 1 is loaded during iteration c - 1, parked into the free LDS buffer
//    at the start of iteration c, and the set is reloaded with chunk c + 2 right away (a chunk is 1.5 - 3 us of MFMA
//    work; the two-set form of round 2 needed ~350 VGPRs at CT = 2);
//  * fragments of ONE 16-k group ahead (two named sets) instead of a whole chunk's;
//  * buffer addressing: one 32-bit lane offset per load, the chunk displacement folded into the scalar resource base,
//    num_records doing the bounds work (weight rows >= rows and concept rows >= N_edit read as zeros: no clamps, no mask);
//  * the parks and the next loads sit BETWEEN the MFMA groups of the chunk, so the matrix pipe keeps draining while
//    this wave moves data (round 2: all waves parked right after the barrier, the pipe idle: GEMM alone at 50
//    concepts 33.3 -> 30.7 us, 0.47 -> 0.51 of the f32 MFMA peak on the issued tile; 128-wide: 0.57).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pj_rsrc(const float* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0 ? (bytes < 0x7fffffffL ? (int)bytes : 0x7fffffff) : 0, 0x00020000);
}

template <int D, int MT, int NMT, int CT = 2>
__device__ __forceinline__ void project_body_w(const float* __restrict__ W_old, const float* __restrict__ Dm,
                                               const float* __restrict__ Csub, float* __restrict__ T,
                                               long rows, int Ne, int NEP, float* Wc, int mbase, int blk_off,
                                               int c4_of_wave = -1, bool active = true, int zero_from = 4,
                                               float* Tl = nullptr, int tld = 0) {
  // Tl != nullptr (the fused launch): the workgroup's T tile stays in LDS - Tl [MT*16][tld] floats at the START of the
  // dynamic LDS, over the staging buffers (every wave is past its last fragment read: the final chunk ends on a barrier)
  constexpr int d = D;
  constexpr int SR = MT * 16;
  constexpr int NCB = 64 * CT;                        // concepts per batch
  float* Dc = Wc + 2 * SR * PJ_LD;                    // [2][NCB][PJ_LD]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c4 = c4_of_wave >= 0 ? c4_of_wave : (w & 3);   // column tile of this wave (of the 4 CT a batch holds)
  const int li = lane & 15, lk = lane >> 4;
  const long R0 = (long)(blockIdx.x - blk_off) * SR;

  constexpr int NC = D / PJ_KC;                       // k-chunks (12 / 16 / 32), even
  constexpr int F4 = SR * (PJ_KC / 4);                // float4 per W chunk
  constexpr int NLD = (F4 + 511) / 512;               // per thread
  constexpr int NDL = NCB * (PJ_KC / 4) / 512;        // 2 * CT
  const float cscale = Csub ? 1.f : 0.f;              // D_e = X - cscale * Y (X = G, Y = C_e) or X = Dm
  const float* Ysrc = Csub ? Csub : Dm;
  const long w_valid = ((rows - R0) < SR ? (rows - R0) : SR) * (long)d * 4;   // bytes of this super-tile that exist
  const float* Wb = W_old + R0 * d;
  // lane offsets (bytes): element e = tid + 512 p -> row e >> 4, float4 column e & 15
  unsigned vo_w[NLD], vo_d[NDL];
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    const int e = tid + 512 * p;
    vo_w[p] = (unsigned)(((e >> 4) * d + ((e & 15) << 2)) * 4);      // rows >= SR: beyond w_valid or never parked
  }
#pragma unroll
  for (int p = 0; p < NDL; ++p) {
    const int e = tid + 512 * p;
    vo_d[p] = (unsigned)(((e >> 4) * d + ((e & 15) << 2)) * 4);
  }
  const int nbatch = (NEP + NCB - 1) / NCB;
#pragma unroll 1
  for (int bt = 0; bt < nbatch; ++bt) {
    const int nct = (NEP - bt * NCB) >= NCB ? CT : 1; // NEP is a multiple of 64: the last batch may hold one half
    const long d_valid = (long)(Ne - bt * NCB) * d * 4;               // concept rows of this batch that exist (may be <= 0)
    const float* Xb = Dm + (size_t)bt * NCB * d;
    const float* Yb = Ysrc + (size_t)bt * NCB * d;
    float4_t sw[NLD], sx[NDL], sy[NDL];
    auto issue = [&](int kc) {
      const __amdgpu_buffer_rsrc_t rw = pj_rsrc(Wb + kc * PJ_KC, w_valid - (long)kc * PJ_KC * 4);
      const __amdgpu_buffer_rsrc_t rx = pj_rsrc(Xb + kc * PJ_KC, d_valid - (long)kc * PJ_KC * 4);
      const __amdgpu_buffer_rsrc_t ry = pj_rsrc(Yb + kc * PJ_KC, d_valid - (long)kc * PJ_KC * 4);
#pragma unroll
      for (int p = 0; p < NLD; ++p) sw[p] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rw, vo_w[p], 0, 0));
#pragma unroll
      for (int p = 0; p < NDL; ++p) {
        sx[p] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, vo_d[p], 0, 0));
        sy[p] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(ry, vo_d[p], 0, 0));
      }
    };
    auto park_w = [&](int buf) {
#pragma unroll
      for (int p = 0; p < NLD; ++p) {
        const int e = tid + 512 * p;
        if (e < F4) *(float4_t*)&Wc[(buf * SR + (e >> 4)) * PJ_LD + ((e & 15) << 2)] = sw[p];
      }
    };
    auto park_d = [&](int buf) {
#pragma unroll
      for (int p = 0; p < NDL; ++p) {
        const int e = tid + 512 * p;
        *(float4_t*)&Dc[(buf * NCB + (e >> 4)) * PJ_LD + ((e & 15) << 2)] = sx[p] - cscale * sy[p];
      }
    };
    float4_t acc[CT][NMT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < NMT; ++m) acc[c][m] = (float4_t){0.f, 0.f, 0.f, 0.f};
    struct Frag { float4_t a[NMT]; float4_t b[CT]; };
    // k permutation: MFMA q of 16-k group g uses k = 16g + 4*(lane>>4) + q on both operands
    auto rd = [&](int buf, int g, Frag& f) {
#pragma unroll
      for (int m = 0; m < NMT; ++m)
        f.a[m] = *(const float4_t*)&Wc[(buf * SR + (mbase + m) * 16 + li) * PJ_LD + g * 16 + 4 * lk];
#pragma unroll
      for (int c = 0; c < CT; ++c)
        f.b[c] = *(const float4_t*)&Dc[(buf * NCB + (c * 4 + c4) * 16 + li) * PJ_LD + g * 16 + 4 * lk];
    };
    auto mm = [&](const Frag& f) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int m = 0; m < NMT; ++m)
          acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[m][q], f.b[0][q], acc[0][m], 0, 0, 0);
        if constexpr (CT > 1) {
          if (nct > 1) {
#pragma unroll
            for (int m = 0; m < NMT; ++m)
              acc[1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[m][q], f.b[1][q], acc[1][m], 0, 0, 0);
          }
        }
      }
    };
    // one chunk: buffer `buf` holds chunk kc; the registers hold chunk kc + 1 (parked into buf ^ 1 on the way) and are
    // reloaded with chunk kc + 2
    auto chunk = [&](int kc, int buf) {
      Frag fA, fB;
      rd(buf, 0, fA);
      if (kc + 1 < NC) park_w(buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      rd(buf, 1, fB);
      mm(fA);
      __builtin_amdgcn_sched_barrier(0);
      if (kc + 1 < NC) park_d(buf ^ 1);
      if (kc + 2 < NC) issue(kc + 2);
      __builtin_amdgcn_sched_barrier(0);
      rd(buf, 2, fA);
      mm(fB);
      __builtin_amdgcn_sched_barrier(0);
      rd(buf, 3, fB);
      mm(fA);
      __builtin_amdgcn_sched_barrier(0);
      mm(fB);
      __syncthreads();
    };
    __syncthreads();                                  // previous batch is done with the LDS buffers
    issue(0);
    park_w(0);
    park_d(0);
    issue(1);
    __syncthreads();
#pragma unroll 1
    for (int kc = 0; kc < NC; kc += 2) {
      chunk(kc, 0);
      chunk(kc + 1, 1);
    }
    // D layout: col = lane & 15 (concept), row = 4*(lane>>4) + r
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (c < nct) {
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long gr = R0 + (mbase + m) * 16 + 4 * lk + r;
            if (Tl) {
              if (active) {                                          // rows >= rows were read as zeros: their T is zero
                float* trow = Tl + ((mbase + m) * 16 + 4 * lk + r) * tld + bt * NCB;
                trow[(c * 4 + c4) * 16 + li] = acc[c][m][r];
                if (c4 == 0)
                  for (int dc = zero_from; dc < 4; ++dc) trow[dc * 16 + li] = 0.f;
              }
            } else if (active && gr < rows) {
              T[gr * NEP + bt * NCB + (c * 4 + c4) * 16 + li] = acc[c][m][r];
              if (c4 == 0)                                           // column tiles nobody computes (remapped waves): the padding stays zero
                for (int dc = zero_from; dc < 4; ++dc) T[gr * NEP + bt * NCB + dc * 16 + li] = 0.f;
            }
          }
      }
    }
  }
}

// Optional riders of the projection launch (blocks 0..gp_riders(nb)-1): the whole small-system factorisation
// of the dual form (N <= 128):  K = lambda S^-1 + C C^T  (f64 MFMA over the d features, split by OUTPUT sub-tile: GP_SUB1 / GP_SUB2
// below), then the Cholesky + block inverses by the rider block that finishes its sub-tiles LAST.  Riders need nothing from the
// projection and vice versa, so riding along costs no launch and no event.  The hand-off between rider blocks is the write-through
// form of the CDNA guide's split-K recipe (G16): sc1 stores -> per-wave vmcnt(0) -> barrier -> one lane draws a relaxed agent-scope
// ticket; the block drawing the last ticket reads the tiles with sc1 loads - no release / acquire fence on either side (st_sc1
// below).  Correct for any placement of the rider blocks; bit-repeatable (every sum has a fixed order).
//
// Hand-off words (h->ticket, all zero between launches - nothing of the protocol lives in the kernel arguments, so a
// launch can be captured into a hipGraph and replayed):
//   [0] arrival counter of the Gram riders       (reset by the block that draws the last ticket)
//   [1] stage word of the factorising block:  1 = L_00^-1 is in memory, 2 = L_10 too (two-block systems), 3 = every factor block is
//   [2] completion counter of the solve riders   (the last one to finish resets [1] and [2]; fused launch: posts [1] = 4, resets [2])
//   [3] fused launch: projecting workgroups that have seen stage 4 (the last of them resets [1] and [3])
struct GramPotrfJob {
  const float* C;       // [N, d]; null = no riders
  const float* s;       // [N]
  int N;
  float lamb;
  double* slabs;        // [tiles][64][64]: the lower tiles (0,0) (1,0) (1,1) of the Gram, finished elements
  unsigned* ticket;     // the three hand-off words
  double* Lmat;         // [n, n], n = 64 * nb (only block (1, 0) is written: L_10 of a two-block system)
  double* Linv;         // [nb][64][64]
  int* status;
  int nb;               // 64-blocks of the dual system handled by the riders: 1 or 2
  float* R;             // [N_edit, d] rows of K^-1 C, written by the solve riders
  int N_edit;
  unsigned short* Rp;   // fused launch with the split-bf16 update: R as three bf16 planes [3][NEP/8][4][D/4][8] (below), or null
  int NEP;              // N_edit rounded up to 64
  int fused;            // the launch also UPDATES (uce_lowrank_fused.hip): R is published write-through, the last solve rider
                        // posts stage 4 = "R complete", the projecting workgroups wait for it, count themselves out on
                        // ticket[3], and the last of the `n_proj` re-arms [1] and [3]
  int n_proj;
  // register-resident launch (uce_edit_resident.hip): the solve riders publish R as f16 (hi, lo) MFMA fragments in the order its
  // phase B reads them - [column tile][k-block of 32 concepts][plane][lane] x 16 bytes, lane (i = lane & 15, kg = lane >> 4) holding
  // concepts 32 b + 16 (j >> 2) + 4 kg + (j & 3), j = 0..7, of column 16 t + i - under one power-of-two scale per column (Rsc: its inverse)
  void* Rh;             // null: not wanted
  float* Rsc;           // [d]
};

// The Gram is split by OUTPUT (round 6): a rider owns lower 16 x 16 sub-tiles of the system's 64 x 64 tiles, its eight waves each
// contract an eighth of the features straight from global fragments (no LDS staging) and meet in LDS - the workgroup writes
// FINISHED elements, so every tile has ONE slab and the block that factors sums nothing (the f64 matrix pipe runs at 64 cycles per
// 16 x 16 x 4 step and SIMD: the K-split over four riders per tile of rounds 2-5 spent 5.6 us in MFMAs, 1.8 publishing and 3.5
// summing four slabs; at 50 concepts the chain's first link went from 13.6 to 9.5 us).
constexpr int GP_SUB1 = 10;     // riders of a one-block system (one sub-tile each)
constexpr int GP_SUB2 = 2;      // sub-tiles per rider of a two-block system: 10 + 16 + 10 sub-tiles on 18 riders
constexpr int GP_MAXB = 2;      // largest system the riders take: 128 x 128 (3 lower tiles)
constexpr int GP_LD = 40;       // floats, k-contiguous NT tile stride (conflict-free b128)
constexpr int GP_TLD = 66;      // doubles

__host__ __device__ constexpr int gp_riders(int nb) { return nb <= 1 ? GP_SUB1 : 36 / GP_SUB2; }

// Write-through (sc1) stores / L1-bypassing (sc1) loads of hand-off payloads: a relaxed agent-scope atomic of 8 bytes
// lowers to global_store/load_dwordx2 sc1.  Payload published this way needs NO release fence (buffer_wbl2 writes back
// every dirty line of the XCD's L2 - megabytes of T while the projection streams - and cost several microseconds per
// hand-off here) and the consumer needs no acquire (no L1 invalidate): drained stores -> barrier -> relaxed flag /
// ticket on one side, relaxed poll -> barrier -> sc1 loads on the other (CDNA guide, Guideline 16, the sc1 form).
__device__ __forceinline__ void st_sc1(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same accesses 16 bytes wide (buffer_load/store_dwordx4 ... sc1; cache-policy bit 4 = sc1 on gfx950).  A CU
// sustains only ~10 KB/us of 8-byte L1-bypassing loads (70 KB of slabs: 6.5 us of the chain): the wide form halves the
// requests per byte.  Out-of-range offsets read as zero without touching memory - that is the mask.
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
constexpr unsigned SC1_OOB = 0xFFFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sc1_rsrc(const double* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double2_t ld_sc1_x2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16 /* sc1 */));
}
__device__ __forceinline__ void st_sc1_x2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double2_t v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, v), r, byte_off, 0, 16 /* sc1 */);
}

// The factorising block tells the solve riders how far the factorisation of THIS launch has come: all its
// (write-through) stores drained, a barrier, then the stage word.
__device__ __forceinline__ void publish_stage(const GramPotrfJob& j, unsigned stage) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(j.ticket + 1, stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One lane polls the stage word (relaxed, with s_sleep); the block passes a barrier afterwards and reads the payload
// with sc1 loads.  A factorising block that never ran cannot happen with in-order dispatch: reported, not hung on.
__device__ __forceinline__ void wait_stage(const GramPotrfJob& j, unsigned stage) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(j.ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < stage) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 24)) {                          // ~ seconds
        atomicCAS(j.status, 0, -1);
        break;
      }
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Solve riders (one per SV_COLS = 32 columns of R: the Gram riders that did NOT draw the last ticket take the first
// column blocks, dedicated blocks behind them the rest): wait for the factorisation of THIS launch, then
//   R[:, cols] = rows 0..N_edit-1 of  L^-T L^-1 C[:, cols]
// with the inverted diagonal blocks - the triangular solves that used to be a launch of their own between the projection
// and the update.  f64 tiles in LDS (matrices: stride SV_LD, the 32-column vectors: stride SV_VLD), 8 waves x one
// 16 x 16 MFMA tile, contraction over the non-zero part of the triangular operand (sv_prod).  A two-block system keeps
// all three factor blocks (L_00^-1, L_10, L_11^-1) resident, so its six products run back to back; the first of them
// (Y_0 = L_00^-1 C_0) already starts at stage 1, while block 1 is still being factored.
// ---------------------------------------------------------------------------------------------
constexpr int SV_LD = 66;
constexpr int SV_VLD = 34;
constexpr int SV_COLS = 32;     // columns of R per solve rider: 8 waves x one 16 x 16 tile
constexpr size_t SV_TILE = (size_t)64 * SV_LD * sizeof(double);
constexpr size_t SV_VTILE = (size_t)64 * SV_VLD * sizeof(double);
__host__ __device__ constexpr size_t sv_smem(int nb) { return nb <= 1 ? SV_TILE + 2 * SV_VTILE : 3 * SV_TILE + 3 * SV_VTILE; }

// dst = base - / + op(A) * B :  A [64][64] (stride SV_LD), B / dst / base [64][SV_COLS] (stride SV_VLD) are LDS tiles; TA:
// op(A)[i][k] = A[k][i].  B is read as [k][col].  base == nullptr: dst = op(A) B.  (dst may alias base, never A or B.)
// TRI: A is LOWER triangular (an inverted diagonal block) - the 16-row block rb of op(A) B only contracts over
// k < 16 (rb + 1) (TA: k >= 16 rb).  One 16 x 16 tile per wave; the waves of a SIMD (w, w + 4) take row blocks rb and
// 3 - rb, so every SIMD issues 20 of the 32 MFMAs a full contraction would.  (A 64 x 64 x 64 f64 product is MFMA-bound
// at 2048 cycles on one CU: the riders are sized so that this chain link is ~600 cycles instead.)
template <bool TA, bool TRI>
__device__ __forceinline__ void sv_prod(double* dst, const double* A, const double* B, const double* base, double sign) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int rb = w < 4 ? w : 7 - w;
  const int row0 = 16 * rb, col0 = 16 * (w >> 2);
  const int kb0 = (TRI && TA) ? 4 * rb : 0, kb1 = (TRI && !TA) ? 4 * rb + 4 : 16;
  double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
  // all fragments first (one exposed LDS latency), then the dependent MFMA chain over the live k-steps
  double fa[16], fb[16];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) {
    const int t = 4 * kb + kk;
    fa[kb] = TA ? A[t * SV_LD + row0 + r] : A[(row0 + r) * SV_LD + t];
    fb[kb] = B[t * SV_VLD + col0 + r];
  }
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
    if (kb >= kb0 && kb < kb1) acc = mfma_f64(fa[kb], fb[kb], acc);
  // D layout: row = kk + 4q, col = r
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o = (row0 + kk + 4 * q) * SV_VLD + col0 + r;
    dst[o] = (base ? base[o] : 0.0) + sign * acc[q];
  }
}

// eight fp32 values -> eight bf16 in each of three planes (x = h + m + l exactly; v_cvt_pk_bf16_f32 rounds to nearest even,
// every residual subtraction is exact in fp32)
__device__ __forceinline__ unsigned rp_cvt_pk(float lo, float hi) {
  typedef __bf16 bf16x2_t_ __attribute__((ext_vector_type(2)));
  typedef float float2_t_ __attribute__((ext_vector_type(2)));
  const float2_t_ v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t_));
}
__device__ __forceinline__ void rp_split8(const float (&x)[8], uint4_t& h, uint4_t& m, uint4_t& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = rp_cvt_pk(x[2 * i], x[2 * i + 1]);
    const float r0 = x[2 * i] - __uint_as_float(h[i] << 16), r1 = x[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u);
    m[i] = rp_cvt_pk(r0, r1);
    l[i] = rp_cvt_pk(r0 - __uint_as_float(m[i] << 16), r1 - __uint_as_float(m[i] & 0xffff0000u));
  }
}

// The two-term f16 split of uce_apply_h2.hip on register values: biased exponent E of a row / column maximum -> the scale
// 2^(14 - (E - 127)) that takes the maximum into [2^14, 2^15), its inverse, and eight scaled values -> (hi, lo) f16 x 8
__device__ __forceinline__ int rs_clamp_exp(unsigned abs_bits) {
  const int E = (int)(abs_bits >> 23);
  return E < 30 ? 30 : (E > 240 ? 240 : E);           // (an all-zero / denormal line, or inf / nan in it)
}
__device__ __forceinline__ float rs_scale(int E) { return __uint_as_float((unsigned)(268 - E) << 23); }
__device__ __forceinline__ float rs_inv_scale(int E) { return __uint_as_float((unsigned)(E - 14) << 23); }
__device__ __forceinline__ void rs_split8(const float (&y)[8], uint4_t& hi, uint4_t& lo) {
  typedef _Float16 f16x2_t_ __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 h0 = (_Float16)y[2 * i], h1 = (_Float16)y[2 * i + 1];       // round to nearest even
    const float r0 = y[2 * i] - (float)h0, r1 = y[2 * i + 1] - (float)h1;      // exact
    hi[i] = __builtin_bit_cast(unsigned, f16x2_t_{h0, h1});
    lo[i] = __builtin_bit_cast(unsigned, f16x2_t_{(_Float16)r0, (_Float16)r1});
  }
}

template <int D>
__device__ __forceinline__ void solve_rider(const GramPotrfJob& j, unsigned char* smem_raw, int colblk) {
  double* M0 = (double*)smem_raw;                         // L_00^-1
  double* M1 = M0 + 64 * SV_LD;                           // L_10      (two-block systems)
  double* M2 = M1 + 64 * SV_LD;                           // L_11^-1
  double* V0 = j.nb == 1 ? M1 : M2 + 64 * SV_LD;          // three 64 x 32 vector tiles (one-block systems use two)
  double* V1 = V0 + 64 * SV_VLD;
  double* V2 = V1 + 64 * SV_VLD;
  const int tid = threadIdx.x;
  const int n = 64 * j.nb;
  // 64 concepts x this block's 32 columns of C (rows >= N are zero) -> LDS, widened to f64.  Needs nothing from the
  // factorisation: loaded before the wait.
  auto load_c = [&](double* Vt, int kblk) {
    const int r = tid >> 3, c4 = (tid & 7) << 2;          // 64 rows x 8 float4: exactly one per thread
    const int row = kblk * 64 + r;
    float4_t v = {0.f, 0.f, 0.f, 0.f};
    if (row < j.N) v = *(const float4_t*)(j.C + (size_t)row * D + colblk * SV_COLS + c4);
    Vt[r * SV_VLD + c4] = (double)v[0];
    Vt[r * SV_VLD + c4 + 1] = (double)v[1];
    Vt[r * SV_VLD + c4 + 2] = (double)v[2];
    Vt[r * SV_VLD + c4 + 3] = (double)v[3];
  };
  // a 64 x 64 block of a row-major f64 matrix, published write-through: read it past the L1, 16 bytes per lane, all
  // loads in flight at once
  auto fetch_m = [&](const double* G, int ld, double2_t (&v)[4]) {
    const __amdgpu_buffer_rsrc_t r = sc1_rsrc(G, (unsigned)(64 * ld * sizeof(double)));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int e = 2 * (tid + 512 * p);
      v[p] = ld_sc1_x2(r, (unsigned)(((e >> 6) * ld + (e & 63)) * sizeof(double)));
    }
  };
  auto park_m = [&](double* Ms, const double2_t (&v)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int e = 2 * (tid + 512 * p);
      *(double2_t*)&Ms[(e >> 6) * SV_LD + (e & 63)] = v[p];
    }
  };
  auto store_r = [&](const double* Vt, int kblk) {        // rows of X -> R (fp32), rows < N_edit only
    const int r = tid >> 3, c4 = (tid & 7) << 2;
    const int row = kblk * 64 + r;
    if (row < j.N_edit && !j.Rh) {
      const float4_t v = {(float)Vt[r * SV_VLD + c4], (float)Vt[r * SV_VLD + c4 + 1], (float)Vt[r * SV_VLD + c4 + 2],
                          (float)Vt[r * SV_VLD + c4 + 3]};
      if (j.fused)                                          // read by other CUs of THIS launch: write-through, past the L2
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, v),
                                               __builtin_amdgcn_make_buffer_rsrc((void*)j.R, 0, j.N_edit * D * 4, 0x00020000),
                                               (unsigned)(((size_t)row * D + colblk * SV_COLS + c4) * 4), 0, 16 /* sc1 */);
      else
        *(float4_t*)(j.R + (size_t)row * D + colblk * SV_COLS + c4) = v;
    }
  };
  // The same rows as three bf16 planes r = r_h + r_m + r_l (exact: 8 + 8 + 8 significand bits of the fp32 value) for the
  // split-bf16 update of the fused launch, in the order its B fragments are read: plane p, k-group kg = k / 8, column
  // residue q = col & 3, column quad col >> 2, then the 8 concepts of the group - 16 bytes that are ONE lane's operand of
  // v_mfma_f32_16x16x32_bf16.  Rows >= N_edit of the padded NEP are written as zeros.  256 threads: (column, k-group).
  auto store_rp = [&](const double* Vt, int kblk) {
    if (!j.Rp || kblk * 64 >= j.NEP || tid >= 256) return;
    const int c = tid & 31, g = tid >> 5;
    const int col = colblk * SV_COLS + c;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (kblk * 64 + 8 * g + i < j.N_edit) ? (float)Vt[(8 * g + i) * SV_VLD + c] : 0.f;
    uint4_t pl[3];
    rp_split8(x, pl[0], pl[1], pl[2]);
    const int KGT = j.NEP >> 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)j.Rp, 0, 3 * j.NEP * D * 2, 0x00020000);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      __builtin_amdgcn_raw_buffer_store_b128(pl[p], rs, (unsigned)(((((p * KGT + kblk * 8 + g) * 4 + (col & 3)) * (D / 4)) + (col >> 2)) * 16),
                                             0, 16 /* sc1 */);
  };
  // R as (hi, lo) f16 fragments + column scales for the register-resident launch: rows 0..63 of X in X0, rows 64..127 in X1 (two-
  // block systems; null otherwise).  `cm`: 32 dead LDS words.
  auto store_rf = [&](const double* X0, const double* X1, unsigned* cm) {
    if (!j.Rh) return;
    const int NKC = j.NEP >> 5;                              // k-blocks of 32 concepts (2 or 4)
    if (tid < 32) cm[tid] = 0u;
    __syncthreads();
    {
      const int c = tid & 31, rg = tid >> 5;               // 32 columns x 16 groups of 4 (+ 4) rows
      float m = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * rg + r < j.N_edit) m = fmaxf(m, fabsf((float)X0[(4 * rg + r) * SV_VLD + c]));
        if (X1 && 64 + 4 * rg + r < j.N_edit) m = fmaxf(m, fabsf((float)X1[(4 * rg + r) * SV_VLD + c]));
      }
      atomicMax(&cm[c], __float_as_uint(m));
    }
    __syncthreads();
    if (tid < 128 * NKC) {
      const int lane = tid & 63, i = lane & 15, kg = lane >> 4;
      const int ub = tid >> 6, u = ub / NKC, b = ub - u * NKC;
      const int c = 16 * u + i;
      const float sc = rs_scale(rs_clamp_exp(cm[c]));
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = 32 * b + 16 * (e >> 2) + 4 * kg + (e & 3);
        const double x = row < 64 ? X0[row * SV_VLD + c] : X1[(row - 64) * SV_VLD + c];
        y[e] = row < j.N_edit ? (float)x * sc : 0.f;
      }
      uint4_t hi, lo;
      rs_split8(y, hi, lo);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(j.Rh, 0, (D / 16) * NKC * 2 * 1024, 0x00020000);
      const unsigned off = (unsigned)(((((2 * colblk + u) * NKC + b) * 2) * 64 + lane) * 16);
      __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off, 0, 16 /* sc1 */);
      __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off + 1024, 0, 16);
    }
    if (tid < 32)
      __hip_atomic_store(j.Rsc + colblk * SV_COLS + tid, rs_inv_scale(rs_clamp_exp(cm[tid])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto finish = [&]() {
    // every rider counts itself out; the last one re-arms the stage word and the counter for the next launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(j.ticket + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (register-resident launch: the main workgroups poll this counter themselves and re-arm every word when they are through)
      if (t == (unsigned)(D / SV_COLS) - 1 && !j.Rh) {
        __hip_atomic_store(j.ticket + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // every rider's stores were drained before it drew its ticket: R is complete (fused: the updaters wait for this)
        __hip_atomic_store(j.ticket + 1, j.fused ? 4u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  DBG(8);
  load_c(V0, 0);
  if (j.nb == 2) load_c(V1, 1);
  DBG(9);
  double2_t mv[4];
  if (j.nb == 1) {
    wait_stage(j, 3);
    DBG(10);
    fetch_m(j.Linv, 64, mv);
    park_m(M0, mv);
    __syncthreads();
    DBG(11);
    sv_prod<false, true>(V1, M0, V0, nullptr, 1.0);       // Y = L^-1 C
    __syncthreads();
    sv_prod<true, true>(V0, M0, V1, nullptr, 1.0);        // X = L^-T Y
    __syncthreads();
    DBG(12);
    store_r(V0, 0);
    store_rp(V0, 0);
    store_rf(V0, nullptr, (unsigned*)V1);                 // (Y is dead)
    DBG(13);
    finish();
    return;
  }
  // ---- two-block system:  Y0 = A C0 | Y1 = Dg (C1 - L10 Y0) | X1 = Dg^T Y1 | X0 = A^T (Y0 - L10^T X1),  A = L_00^-1, Dg = L_11^-1
  wait_stage(j, 1);
  fetch_m(j.Linv, 64, mv);
  park_m(M0, mv);
  __syncthreads();
  sv_prod<false, true>(V2, M0, V0, nullptr, 1.0);         // Y0 -> V2
  DBG(10);
  wait_stage(j, 2);                                       // (its barrier also closes the product above)
  fetch_m(j.Lmat + (size_t)64 * n, n, mv);                // block (1, 0) of L: in memory while block 1 is still being eliminated
  park_m(M1, mv);
  __syncthreads();
  sv_prod<false, false>(V1, M1, V2, V1, -1.0);            // C1 - L10 Y0 -> V1
  wait_stage(j, 3);                                       // (barrier)
  fetch_m(j.Linv + 4096, 64, mv);
  park_m(M2, mv);
  __syncthreads();
  DBG(11);
  sv_prod<false, true>(V0, M2, V1, nullptr, 1.0);         // Y1 -> V0 (C0 is dead)
  __syncthreads();
  sv_prod<true, true>(V1, M2, V0, nullptr, 1.0);          // X1 -> V1
  __syncthreads();
  store_r(V1, 1);
  store_rp(V1, 1);
  sv_prod<true, false>(V2, M1, V1, V2, -1.0);             // Y0 - L10^T X1 -> V2
  __syncthreads();
  sv_prod<true, true>(V0, M0, V2, nullptr, 1.0);          // X0 -> V0
  __syncthreads();
  DBG(12);
  store_r(V0, 0);
  store_rp(V0, 0);
  store_rf(V0, V1, (unsigned*)V2);                        // (X1 is still in V1; V2 is dead)
  DBG(13);
  finish();
}

// LDS of the block that factors a two-block system: the elimination scratch (the K_00 staging tile aliases it, and
// later the K_11 tile), the K_10 -> L_10 tile and the L_00^-1 -> Schur complement tile
constexpr size_t GP_F2_SMEM = sizeof(Potrf64Scratch) + 2 * 64 * LD * sizeof(double);

template <int D>
__device__ __forceinline__ void gram_potrf_rider(const GramPotrfJob& j, unsigned char* smem_raw) {
  DBG(0);
  constexpr size_t AB_BYTES = (size_t)4 * 64 * GP_LD * sizeof(float);   // (the first 16 KB: the eight waves' partial sub-tiles)
  // the factorisation scratch ALIASES the Gram staging (As, Bs, P1 are dead once the slab is published), so a
  // single-tile rider needs 74 KB
  Potrf64Scratch* sc = (Potrf64Scratch*)smem_raw;
  static_assert(sizeof(Potrf64Scratch) <= AB_BYTES + 64 * GP_TLD * sizeof(double), "scratch must fit the Gram staging");
  // (all LDS in the dynamic region: a static __shared__ would shift its 16-byte alignment)
  unsigned* s_tick_p = (unsigned*)(smem_raw + AB_BYTES + 64 * GP_TLD * sizeof(double));
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = w >> 2, wq = w & 3;
  const int wr = (wq >> 1) * 32, wc = (wq & 1) * 32;
  const int nriders = gp_riders(j.nb);
  // ---- the Gram, split by OUTPUT: the lower 16 x 16 sub-tiles of the system's tiles in the order  K_00 (ten, lower) | K_10 (sixteen) |
  // K_11 (ten, lower);  a one-block system has one rider per sub-tile, a two-block system two sub-tiles per rider (18 riders).  The
  // eight waves of a rider each contract an eighth of the features straight from global fragments (lane (i, kq): row 16 s + i,
  // features 16 step + 4 kq .. + 3 - the same permutation of the contraction index on both operands) and meet in LDS, summed in wave
  // order: the rider writes FINISHED elements of the tile's one slab.
  {
    constexpr int FW = D / 8, NS = FW / 16;                          // features per wave, 16-feature steps
    const int per = j.nb <= 1 ? 1 : GP_SUB2;
    const int i16 = lane & 15, kq = lane >> 4;
    double* Pw = (double*)smem_raw;                                   // [8][256]
    auto lower = [](int q, int& sa, int& sb) { sa = q >= 6 ? 3 : (q >= 3 ? 2 : (q >= 1 ? 1 : 0)); sb = q - sa * (sa + 1) / 2; };
    float4_t fa[GP_SUB2][NS], fb[GP_SUB2][NS];
    int tl[GP_SUB2], sas[GP_SUB2], sbs[GP_SUB2];
    double mas[GP_SUB2], mbs[GP_SUB2];
#pragma unroll
    for (int u = 0; u < GP_SUB2; ++u) {
      const int idx = (int)blockIdx.x * per + (u < per ? u : 0);
      int t, sa, sb;
      if (idx < 10) { t = 0; lower(idx, sa, sb); }
      else if (idx < 26) { t = 1; sa = (idx - 10) >> 2; sb = (idx - 10) & 3; }
      else { t = 2; lower(idx - 26, sa, sb); }
      tl[u] = t; sas[u] = sa; sbs[u] = sb;
      const int rowa = (t == 0 ? 0 : 64) + 16 * sa + i16, rowb = (t == 2 ? 64 : 0) + 16 * sb + i16;
      mas[u] = rowa < j.N ? 1.0 : 0.0;                                // padding rows contribute zeros
      mbs[u] = rowb < j.N ? 1.0 : 0.0;
      const float* pa = j.C + (size_t)(rowa < j.N ? rowa : 0) * D + w * FW + 4 * kq;
      const float* pb = j.C + (size_t)(rowb < j.N ? rowb : 0) * D + w * FW + 4 * kq;
      if (u < per) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
          fa[u][s2] = *(const float4_t*)(pa + 16 * s2);
          fb[u][s2] = *(const float4_t*)(pb + 16 * s2);
        }
      }
    }
#ifdef UCE_CHAIN_DEBUG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DBG(6);
#endif
#pragma unroll
    for (int u = 0; u < GP_SUB2; ++u) {
      if (u < per) {
        double4_t a4 = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
          for (int t = 0; t < 4; ++t) a4 = mfma_f64(mas[u] * (double)fa[u][s2][t], mbs[u] * (double)fb[u][s2][t], a4);
        // D layout: row = (lane >> 4) + 4 r, col = lane & 15
        if (u) __syncthreads();                                       // (the previous sub-tile's partials have been summed)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pw[w * 256 + (kq + 4 * r) * 16 + i16] = a4[r];
        __syncthreads();
        if (tid < 256) {
          double v = Pw[tid];
#pragma unroll
          for (int ww = 1; ww < 8; ++ww) v += Pw[ww * 256 + tid];
          st_sc1(&j.slabs[(size_t)tl[u] * 4096 + (16 * sas[u] + (tid >> 4)) * 64 + 16 * sbs[u] + (tid & 15)], v);
        }
      }
    }
    DBG(1);
  }
  auto diag_term = [&](int row) -> double {
    const float sv = (row < j.N) ? j.s[row] : 1.f;
    return (row < j.N) ? ((sv > 0.f) ? (double)j.lamb / (double)sv : __builtin_nan("")) : 1.0;
  };
  // one-block systems: the diagonal terms lambda / s_i this thread will add if it turns out to be the factoring block - a load
  // and an f64 division, taken off the path between the ticket and the elimination (they overlap the drain of the stores below)
  double dpre[4] = {0.0, 0.0, 0.0, 0.0};
  if (j.nb <= 1) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int e = 2 * (tid + 512 * p), row = e >> 6, col = e & 63;
      if (row == col || row == col + 1) dpre[p] = diag_term(row);
    }
  }
  // ---- publish the slab, draw a ticket (CDNA guide, Guideline 16 / split-K reduction recipe)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    // (slabs went out write-through and are read back past the L1: no release / acquire fence - see st_sc1)
    const unsigned t = __hip_atomic_fetch_add(j.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_tick_p = t;
    if (t == (unsigned)nriders - 1) {
      __hip_atomic_store(j.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      *j.status = 0;
    }
  }
  __syncthreads();
  DBG(2);
  const unsigned my_ticket = *s_tick_p;
  __syncthreads();                                                // everybody has read the ticket: the LDS is free
  if (my_ticket != (unsigned)nriders - 1) {
    // not the last arriver: this block's Gram duty is over - it becomes the solve rider of column block `ticket`
    solve_rider<D>(j, smem_raw, (int)my_ticket);
    return;
  }
  // Last arriver: all 8 waves factor.  A tile's slab is fetched by ALL 512 threads (8 elements each, every
  // load independent and in flight at once, fixed slab order: bit-repeatable).  Only what the factorisation reads is
  // fetched: rows of real concepts (the padding rows are the identity) and, for the diagonal tiles, the 4 x 4 tiles
  // of the lower triangle.
  const __amdgpu_buffer_rsrc_t slab_r = sc1_rsrc(j.slabs, (unsigned)(nriders * 4096 * sizeof(double)));
  const __amdgpu_buffer_rsrc_t linv_r = sc1_rsrc(j.Linv, (unsigned)(j.nb * 4096 * sizeof(double)));
  // thread -> element pairs e, e + 1 with e = 2 (tid + 512 p): row e >> 6, columns e & 63 (even) and the next
  auto fetch_tile = [&](int t, int nrow4, bool lower, auto& v) {
    constexpr int NBK = sizeof(v) / sizeof(v[0]);                 // slabs per tile: the first extent of v[NBK][4]
#pragma unroll
    for (int b = 0; b < NBK; ++b)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int e = 2 * (tid + 512 * p), row = e >> 6, col = e & 63;
        const bool live = row < nrow4 && (!lower || col <= (row | 3));       // (col even, row | 3 odd: both elements alike)
        v[b][p] = ld_sc1_x2(slab_r, live ? (unsigned)(((t * NBK + b) * 4096 + e) * sizeof(double)) : SC1_OOB);
      }
  };
  auto reduce_tile = [&](const auto& v, int nrow4, int row_base, bool diag, double2_t (&out)[4]) {
    constexpr int NBK = sizeof(v) / sizeof(v[0]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int e = 2 * (tid + 512 * p), row = e >> 6, col = e & 63;
      double2_t a = v[0][p];
#pragma unroll
      for (int b = 1; b < NBK; ++b) a += v[b][p];                            // fixed slab order: bit-repeatable
      if (diag && row == col) a[0] = (row < nrow4 ? a[0] : 0.0) + (j.nb <= 1 ? dpre[p] : diag_term(row_base + row));
      if (diag && row == col + 1) a[1] = (row < nrow4 ? a[1] : 0.0) + (j.nb <= 1 ? dpre[p] : diag_term(row_base + row));
      out[p] = a;
    }
  };
  auto park_tile = [&](double* tile, const double2_t (&o)[4]) {            // -> a [64][GP_TLD] LDS tile
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int e = 2 * (tid + 512 * p);
      *(double2_t*)&tile[(e >> 6) * GP_TLD + (e & 63)] = o[p];
    }
  };
  auto store_linv = [&](int blk, int row, int col, const double (&v)[4]) {   // 4 consecutive columns of a row of L_blk^-1
    const unsigned off = (unsigned)((blk * 4096 + row * 64 + col) * sizeof(double));
    st_sc1_x2(linv_r, off, (double2_t){v[0], v[1]});
    st_sc1_x2(linv_r, off + 16, (double2_t){v[2], v[3]});
  };
  double* Ksum = (double*)smem_raw;                               // [64][GP_TLD] (the Gram staging is dead)
  if (j.nb == 1) {
    {
      const int n4 = (j.N + 3) & ~3;
      double2_t v[1][4], o[4];
      fetch_tile(0, n4, true, v);
      reduce_tile(v, n4, 0, true, o);
      park_tile(Ksum, o);
    }
    __syncthreads();
    DBG(3);
    UCE_POTRF64([&](int row, int col, double (&v)[4]) {
                  const pk_d2 a = *(const pk_d2*)&Ksum[row * GP_TLD + col], b = *(const pk_d2*)&Ksum[row * GP_TLD + col + 2];
                  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
                },
                [&](int row, int col, const double (&v)[4]) { store_linv(0, row, col, v); },
                sc, tid, j.status, 0, j.N);
    DBG(4);
    publish_stage(j, 3);
    DBG(5);
    return;
  }
  // ---- nb == 2: the blocked factorisation of the 128 x 128 system, alone, under the projection GEMM, entirely in
  // LDS / registers (the launch-chain form of uce_solve.hip goes through memory between its steps):
  //   K_00 -> L_00^-1 (stage 1) | L_10 = K_10 L_00^-T | S = K_11 - L_10 L_10^T -> L_11^-1 (stage 2)
  static_assert(GP_TLD == LD, "the staging tile and the step tiles share one stride");
  double (*Mi)[LD] = (double (*)[LD])(smem_raw + sizeof(Potrf64Scratch));                           // K_10 -> L_10
  double (*Li)[LD] = (double (*)[LD])(smem_raw + sizeof(Potrf64Scratch) + 64 * LD * sizeof(double));  // L_00^-1 -> S
  const int n = 128, n2 = j.N - 64;                               // real concepts of block 1 (1..64)
  const int n4b = (n2 + 3) & ~3;
  // K_00 is reduced and factored while the slab loads of K_10 land (64 VGPRs of them ride through the elimination), the
  // loads of K_11 are issued behind the factor and land under the L_10 product - one CU sustains only ~10 KB/us of
  // L1-bypassing loads: summed before the factor, as the first form of this block did, they were 6 us of the chain
  double2_t vb[1][4], vc[1][4];
  {
    double2_t va[1][4], o[4];
    fetch_tile(0, 64, true, va);                                  // K_00
    fetch_tile(1, n4b, false, vb);                                // K_10: lands during the first factor (64 VGPRs pinned)
    reduce_tile(va, 64, 0, true, o);
    park_tile(Ksum, o);
  }
  __syncthreads();
  DBG(3);
  UCE_POTRF64([&](int row, int col, double (&v)[4]) {
                const pk_d2 a = *(const pk_d2*)&Ksum[row * GP_TLD + col], b = *(const pk_d2*)&Ksum[row * GP_TLD + col + 2];
                v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
              },
              [&](int row, int col, const double (&v)[4]) {
                store_linv(0, row, col, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) Li[row][col + e] = v[e];
              },
              sc, tid, j.status, 0, 64);
  fetch_tile(2, n4b, true, vc);                                   // K_11: lands during the L_10 product below
  {
    double2_t o[4];
    reduce_tile(vb, n4b, 64, false, o);
    park_tile(&Mi[0][0], o);
  }
  // Stages 1 and 2 are posted one phase late, behind a barrier every wave reaches with `s_waitcnt vmcnt(0)` long after the
  // stores they cover were issued: nothing is drained on this block's critical path (the solve riders have the whole second
  // elimination to use L_00^-1 and L_10).
  auto post_stage = [&](unsigned stage) {
    if (tid == 0) __hip_atomic_store(j.ticket + 1, stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  __syncthreads();                                                // L_00^-1 is in LDS, the scratch is dead
  DBG(4);
  {
    // L_10 = K_10 L_00^-T on all 8 waves, contraction only over the non-zero part of the triangular operand: the two waves
    // of a SIMD (w, w + 4) take column blocks (0, 3) or (1, 2) - 20 of the 32 k-steps a full contraction would issue
    const int twr = (w & 2) ? 32 : 0, twc = 16 * ((w & 1) ? (half ? 2 : 1) : (half ? 3 : 0));
    // the Schur complement S = K_11 - L_10 L_10^T: wave (wq, half) owns rows wr .. wr+31 x columns wc + 16 half .. +15
    const int wc8 = wc + 16 * half;
    auto prod = [&](double4_t (&a2)[2], const double (*P)[LD], const double (*Q)[LD], double sign, int r0, int c0, int kb_end) {
      const int r = lane & 15, kk = lane >> 4;
#pragma unroll 4
      for (int kb = 0; kb < kb_end; ++kb) {
        const int t = kb * 4 + kk;
        const double b0 = Q[c0 + r][t];
        a2[0] = mfma_f64(sign * P[r0 + r][t], b0, a2[0]);
        a2[1] = mfma_f64(sign * P[r0 + 16 + r][t], b0, a2[1]);
      }
    };
    // D layout of v_mfma_f64_16x16x4: row = (lane>>4) + 4r, col = lane&15
    const int orq = lane >> 4;
    double4_t pp[2] = {(double4_t){0.0, 0.0, 0.0, 0.0}, (double4_t){0.0, 0.0, 0.0, 0.0}};
    prod(pp, Mi, Li, 1.0, twr, twc, (twc + 16) / 4);
    double (*S)[LD] = (double (*)[LD])smem_raw;                   // K_11 -> the (dead) scratch region
    {
      double2_t k11[4];
      reduce_tile(vc, n4b, 64, true, k11);
      park_tile(&S[0][0], k11);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the stores of L_00^-1 (issued before the product above)
    __syncthreads();
    post_stage(1);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) Mi[twr + m * 16 + orq + 4 * r][twc + (lane & 15)] = pp[m][r];
    __syncthreads();
    {
      const __amdgpu_buffer_rsrc_t l10_r = sc1_rsrc(j.Lmat + (size_t)64 * n, (unsigned)(64 * n * sizeof(double)));
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int e = 2 * (tid + 512 * p);
        st_sc1_x2(l10_r, (unsigned)(((e >> 6) * n + (e & 63)) * sizeof(double)), *(const double2_t*)&Mi[e >> 6][e & 63]);
      }
    }
    const int oc = wc8 + (lane & 15);
    double4_t sacc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) sacc[m][r] = S[wr + m * 16 + orq + 4 * r][oc];
    prod(sacc, Mi, Mi, -1.0, wr, wc8, 16);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) Li[wr + m * 16 + orq + 4 * r][oc] = sacc[m][r];
    __syncthreads();
  }
  UCE_POTRF64([&](int row, int col, double (&v)[4]) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drains the publish of L_10 before the elimination's first barrier
                const pk_d2 a = *(const pk_d2*)&Li[row][col], b = *(const pk_d2*)&Li[row][col + 2];
                v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
              },
              [&](int row, int col, const double (&v)[4]) { store_linv(1, row, col, v); },
              sc, tid, j.status, 64, n2, [&]() { post_stage(2); });
  publish_stage(j, 3);
  DBG(5);
}


// Which 16-column tile and which row tiles of the workgroup's MT x 4 (x CT) tile grid a wave takes.  The standing map gives
// wave w column tile w & 3 and one half of the row tiles, so a SIMD (waves w, w + 4) carries MT tiles whatever N_e is - with
// 36 edit concepts (the SDXL debias slab) a quarter of them multiply zero padding, with 2 (BASELINE config 1) three
// quarters.  For CT = 1 and fewer than four LIVE column tiles (N_e <= 48) the waves share the live tiles instead: one live
// tile - a row tile per wave; two - four waves per column tile; three - 3 + 3 + 2 waves, paired on the SIMDs so that none
// carries more than ceil(3 MT / 4) + 1.  Every wave still takes part in the staging and the barriers; a wave without a
// tile of its own repeats tile 0 and does not store; the waves of column tile 0 write the zeros of the tiles nobody computes
// (T's padding columns stay zero as before).
template <int D, int MT, int CT>
__device__ __forceinline__ void project_dispatch(const float* __restrict__ W_old, const float* __restrict__ Dm,
                                                 const float* __restrict__ Csub, float* __restrict__ T, long rows, int Ne,
                                                 int NEP, float* Wc, int blk_off, float* Tl = nullptr, int tld = 0) {
  constexpr int M0 = (MT + 1) / 2;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int live = (Ne + 15) >> 4;
  if (CT > 1 || live >= 4) {
    if (w < 4) project_body_w<D, MT, M0, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, 0, blk_off, -1, true, 4, Tl, tld);
    else project_body_w<D, MT, MT - M0, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, M0, blk_off, -1, true, 4, Tl, tld);
    return;
  }
  if constexpr (CT == 1) {
    int c4, part, parts;                      // this wave: column tile, its index among the `parts` waves of that tile
    if (live == 1) { c4 = 0; part = w; parts = 8; }
    else if (live == 2) { c4 = w & 1; part = w >> 1; parts = 4; }
    else {                                    // column tiles 0, 1: waves {0, 3, 6}, {1, 4, 7}; column tile 2: waves {2, 5}
      c4 = w < 6 ? w % 3 : w - 6;
      part = w < 6 ? w / 3 : 2;
      parts = c4 == 2 ? 2 : 3;
    }
    const int base = MT / parts, extra = MT % parts;             // row tiles [m0, m0 + nm)
    int nm = base + (part < extra ? 1 : 0);
    int m0 = part * base + (part < extra ? part : extra);
    const bool active = nm > 0;
    if (!active) { nm = 1; m0 = 0; }
    switch (nm) {
      case 1: project_body_w<D, MT, 1, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, m0, blk_off, c4, active, live, Tl, tld); break;
      case 2: project_body_w<D, MT, 2, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, m0, blk_off, c4, active, live, Tl, tld); break;
      case 3: project_body_w<D, MT, 3, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, m0, blk_off, c4, active, live, Tl, tld); break;
      default: project_body_w<D, MT, 4, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, m0, blk_off, c4, active, live, Tl, tld); break;
    }
  }
}

constexpr size_t GP_SMEM1 = (size_t)4 * 64 * GP_LD * sizeof(float) + 64 * GP_TLD * sizeof(double) + 16;   // one system tile
__host__ __device__ constexpr size_t gp_smem(int nb) {
  const size_t g = nb <= 1 ? GP_SMEM1 : (GP_SMEM1 > GP_F2_SMEM ? GP_SMEM1 : GP_F2_SMEM);
  return g > sv_smem(nb) ? g : sv_smem(nb);
}
// blocks of the launch that do not project: the Gram riders plus the DEDICATED solve riders (the Gram riders that
// do not factor become solve riders themselves)
__host__ __device__ constexpr int lr_rider_blocks(int nb, int d) { return gp_riders(nb) + d / SV_COLS - (gp_riders(nb) - 1); }


}  // namespace
