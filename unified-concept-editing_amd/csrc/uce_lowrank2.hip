// Two-kernel form of the low-rank apply  W_new = W_old + (W_old D_e^T) R_e :
//
//   k_lr_project<D,MT>  : T [rows, NEP] = W_old D_e^T     needs only D_e = G - C_e, not the solve: in uce_edit ONE
//                         launch carries this GEMM and, in its first 4 / 12 "rider" blocks, the Gram matrix and the
//                         (blocked) Cholesky + block inverses of the dual system (N <= 64 / N <= 128), so the
//                         latency-bound factorisation hides under the f32-MFMA-bound projection without any
//                         cross-stream event (a HIP event hand-off between two streams measured 13-14 us each way).
//   k_lr_update_s<D,..> : W_new = W_old + T R_e           one pass over the weights per 128 edit concepts (129..256: a second
//                         pass in place): algorithmic bytes 8*rows*d (+ the small T and R).
//
// Splitting the update at T costs 2*4*rows*NEP bytes of extra traffic (6.4 MB at N_edit <= 64 for SD-1.4, 4 %) and
// buys (a) overlap of the projection with the latency-bound small-system chain, (b) an update kernel whose LDS
// footprint is tiny, so several workgroups per CU stream W with their phases naturally interleaved.
#include "uce_common.h"
#include "uce_potrf64.h"
#include "uce_potrf_la.h"

// -DUCE_CHAIN_DEBUG: wall-clock stamps (100 MHz) of the rider chain's phases, read back with uce_debug_read
// (tools/dbg_chain.py); compiled out of the product library.
#ifdef UCE_CHAIN_DEBUG
__device__ unsigned long long g_dbg[64][32];   // per block: 16 wall-clock stamps (slots 0-7 Gram / factor / projection role, 8-15 solve role) + 16 shader-clock stamps
#define DBG(slot) do { if (threadIdx.x == 0 && blockIdx.x < 64) { g_dbg[blockIdx.x][slot] = wall_clock64(); g_dbg[blockIdx.x][16 + slot] = clock64(); } } while (0)
extern "C" int uce_debug_read(unsigned long long* out) {
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
                                               int c4_of_wave = -1, bool active = true, int zero_from = 4) {
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
            if (active && gr < rows) {
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
// of the dual form (N <= 128):  K = lambda S^-1 + C C^T  (f64 MFMA over the d features, GP_NB feature slices per
// 64 x 64 tile of the system x 2 wave quads), then the Cholesky + block inverses by the rider block that finishes
// its slab LAST.  Riders need nothing from the projection and vice versa, so riding along costs no launch and no
// event.  The slab hand-off between rider blocks is the split-K reduction of the CDNA guide (G16) in its write-through
// form: sc1 slab stores -> per-wave vmcnt(0) -> barrier -> one lane draws a relaxed agent-scope ticket; the block
// drawing the last ticket reads all slabs with sc1 loads (summed in slab order: bit-repeatable) - no release / acquire
// fence on either side (st_sc1 below).  Correct for any placement of the rider blocks.
//
// Hand-off words (h->ticket, all zero between launches - nothing of the protocol lives in the kernel arguments, so a
// launch can be captured into a hipGraph and replayed):
//   [0] arrival counter of the Gram riders       (reset by the block that draws the last ticket)
//   [1] stage word of the factorising block:  1 = L_00^-1 is in memory, 2 = L_10 too (two-block systems), 3 = every factor block is
//   [2] completion counter of the solve riders   (the last one to finish resets [1] and [2])
struct GramPotrfJob {
  const float* C;       // [N, d]; null = no riders
  const float* s;       // [N]
  int N;
  float lamb;
  double* slabs;        // [tiles * GP_NB][64][64] partial Grams
  unsigned* ticket;     // the three hand-off words
  double* Lmat;         // [n, n], n = 64 * nb (only block (1, 0) is written: L_10 of a two-block system)
  double* Linv;         // [nb][64][64]
  int* status;
  int nb;               // 64-blocks of the dual system handled by the riders: 1 or 2
  float* R;             // [N_edit, d] rows of K^-1 C, written by the solve riders
  int N_edit;
};

constexpr int GP_NB = 4;        // rider blocks per 64x64 tile of the system (split over the feature axis)
constexpr int GP_MAXB = 2;      // largest system the riders take: 128 x 128 (3 lower tiles)
constexpr int GP_LD = 40;       // floats, k-contiguous NT tile stride (conflict-free b128)
constexpr int GP_TLD = 66;      // doubles

__host__ __device__ constexpr int gp_riders(int nb) { return GP_NB * nb * (nb + 1) / 2; }

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
    if (row < j.N_edit)
      *(float4_t*)(j.R + (size_t)row * D + colblk * SV_COLS + c4) =
          (float4_t){(float)Vt[r * SV_VLD + c4], (float)Vt[r * SV_VLD + c4 + 1], (float)Vt[r * SV_VLD + c4 + 2],
                     (float)Vt[r * SV_VLD + c4 + 3]};
  };
  auto finish = [&]() {
    // every rider counts itself out; the last one re-arms the stage word and the counter for the next launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(j.ticket + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (unsigned)(D / SV_COLS) - 1) {
        __hip_atomic_store(j.ticket + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(j.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  sv_prod<true, false>(V2, M1, V1, V2, -1.0);             // Y0 - L10^T X1 -> V2
  __syncthreads();
  sv_prod<true, true>(V0, M0, V2, nullptr, 1.0);          // X0 -> V0
  __syncthreads();
  DBG(12);
  store_r(V0, 0);
  DBG(13);
  finish();
}

// LDS of the block that factors a two-block system: the elimination scratch (the K_00 staging tile aliases it, and
// later the K_11 tile), the K_10 -> L_10 tile and the L_00^-1 -> Schur complement tile
constexpr size_t GP_F2_SMEM = sizeof(Potrf64Scratch) + 2 * 64 * LD * sizeof(double);

template <int D>
__device__ __forceinline__ void gram_potrf_rider(const GramPotrfJob& j, unsigned char* smem_raw) {
  DBG(0);
  float* As = (float*)smem_raw;                                   // [2 halves][64][GP_LD]  rows of block ti
  float* Bs = As + 2 * 64 * GP_LD;                                // [2 halves][64][GP_LD]  rows of block tk
  constexpr size_t AB_BYTES = (size_t)4 * 64 * GP_LD * sizeof(float);
  double* P1 = (double*)(smem_raw + AB_BYTES);                    // [64][GP_TLD]
  // the factorisation scratch ALIASES the Gram staging (As, Bs, P1 are dead once the slab is published), so a
  // single-tile rider needs 74 KB
  Potrf64Scratch* sc = (Potrf64Scratch*)smem_raw;
  static_assert(sizeof(Potrf64Scratch) <= AB_BYTES + 64 * GP_TLD * sizeof(double), "scratch must fit the Gram staging");
  // (all LDS in the dynamic region: a static __shared__ would shift its 16-byte alignment)
  unsigned* s_tick_p = (unsigned*)(smem_raw + AB_BYTES + 64 * GP_TLD * sizeof(double));
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = w >> 2, wq = w & 3;
  const int wr = (wq >> 1) * 32, wc = (wq & 1) * 32;
  const int ht = tid & 255;                                       // thread index within its half
  const int tile = blockIdx.x / GP_NB, blk = blockIdx.x % GP_NB;  // tile of the system, feature slice
  const int ti = tile == 0 ? 0 : 1, tk = tile == 2 ? 1 : 0;       // lower tiles in order (0,0) (1,0) (1,1)
  const int nriders = gp_riders(j.nb);
  double4_t acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};
  float* Ah = As + half * 64 * GP_LD;
  float* Bh = Bs + half * 64 * GP_LD;
  const int lrow = ht >> 3, lc4 = (ht & 7) * 4;
  constexpr int KS = D / (2 * GP_NB);                             // features per (block, half) slice
  constexpr int NCH = KS / 32;                                    // 32-feature chunks (3 / 4 / 8)
  const int kbeg = (blk * 2 + half) * KS;
  // the whole slice is fetched up front (one memory round trip instead of one per chunk)
  float4_t pre[NCH][2], preb[NCH][2];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int ra = ti * 64 + p * 32 + lrow, rb = tk * 64 + p * 32 + lrow;
      pre[ch][p] = *(const float4_t*)(j.C + (size_t)(ra < j.N ? ra : j.N - 1) * D + kbeg + ch * 32 + lc4);
      preb[ch][p] = *(const float4_t*)(j.C + (size_t)(rb < j.N ? rb : j.N - 1) * D + kbeg + ch * 32 + lc4);
    }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = p * 32 + lrow;
      *(float4_t*)&Ah[r * GP_LD + lc4] = (ti * 64 + r < j.N) ? pre[ch][p] : (float4_t){0.f, 0.f, 0.f, 0.f};
      *(float4_t*)&Bh[r * GP_LD + lc4] = (tk * 64 + r < j.N) ? preb[ch][p] : (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kofs = u * 16 + 4 * (lane >> 4);
      const float4_t fa0 = *(const float4_t*)&Ah[(wr + (lane & 15)) * GP_LD + kofs];
      const float4_t fa1 = *(const float4_t*)&Ah[(wr + 16 + (lane & 15)) * GP_LD + kofs];
      const float4_t fb0 = *(const float4_t*)&Bh[(wc + (lane & 15)) * GP_LD + kofs];
      const float4_t fb1 = *(const float4_t*)&Bh[(wc + 16 + (lane & 15)) * GP_LD + kofs];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0][0] = mfma_f64((double)fa0[t], (double)fb0[t], acc[0][0]);
        acc[0][1] = mfma_f64((double)fa0[t], (double)fb1[t], acc[0][1]);
        acc[1][0] = mfma_f64((double)fa1[t], (double)fb0[t], acc[1][0]);
        acc[1][1] = mfma_f64((double)fa1[t], (double)fb1[t], acc[1][1]);
      }
    }
    __syncthreads();
  }
  DBG(1);
  // D layout of the f64 MFMA: row = (lane>>4) + 4r, col = lane & 15
  const int c = lane & 15, rq = lane >> 4;
  if (half == 1) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) P1[(wr + m * 16 + rq + 4 * r) * GP_TLD + wc + n * 16 + c] = acc[m][n][r];
  }
  __syncthreads();
  double* myslab = j.slabs + (size_t)blockIdx.x * 64 * 64;
  if (half == 0) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wr + m * 16 + rq + 4 * r, col = wc + n * 16 + c;
          st_sc1(&myslab[row * 64 + col], acc[m][n][r] + P1[row * GP_TLD + col]);
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
  auto diag_term = [&](int row) -> double {
    const float sv = (row < j.N) ? j.s[row] : 1.f;
    return (row < j.N) ? ((sv > 0.f) ? (double)j.lamb / (double)sv : __builtin_nan("")) : 1.0;
  };
  // Last arriver: all 8 waves factor.  The GP_NB slabs of a tile are summed by ALL 512 threads (8 elements each, every
  // load independent and in flight at once, fixed slab order: bit-repeatable).  Only what the factorisation reads is
  // fetched: rows of real concepts (the padding rows are the identity) and, for the diagonal tiles, the 4 x 4 tiles
  // of the lower triangle.
  const __amdgpu_buffer_rsrc_t slab_r = sc1_rsrc(j.slabs, (unsigned)(nriders * 4096 * sizeof(double)));
  const __amdgpu_buffer_rsrc_t linv_r = sc1_rsrc(j.Linv, (unsigned)(j.nb * 4096 * sizeof(double)));
  // thread -> element pairs e, e + 1 with e = 2 (tid + 512 p): row e >> 6, columns e & 63 (even) and the next
  auto fetch_tile = [&](int t, int nrow4, bool lower, double2_t (&v)[GP_NB][4]) {
#pragma unroll
    for (int b = 0; b < GP_NB; ++b)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int e = 2 * (tid + 512 * p), row = e >> 6, col = e & 63;
        const bool live = row < nrow4 && (!lower || col <= (row | 3));       // (col even, row | 3 odd: both elements alike)
        v[b][p] = ld_sc1_x2(slab_r, live ? (unsigned)(((t * GP_NB + b) * 4096 + e) * sizeof(double)) : SC1_OOB);
      }
  };
  auto reduce_tile = [&](const double2_t (&v)[GP_NB][4], int nrow4, int row_base, bool diag, double2_t (&out)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int e = 2 * (tid + 512 * p), row = e >> 6, col = e & 63;
      double2_t a = v[0][p];
#pragma unroll
      for (int b = 1; b < GP_NB; ++b) a += v[b][p];                          // fixed slab order: bit-repeatable
      if (diag && row == col) a[0] = (row < nrow4 ? a[0] : 0.0) + diag_term(row_base + row);
      if (diag && row == col + 1) a[1] = (row < nrow4 ? a[1] : 0.0) + diag_term(row_base + row);
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
      double2_t v[GP_NB][4], o[4];
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
  double2_t vb[GP_NB][4], vc[GP_NB][4];
  {
    double2_t va[GP_NB][4], o[4];
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
                                                 int NEP, float* Wc, int blk_off) {
  constexpr int M0 = (MT + 1) / 2;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int live = (Ne + 15) >> 4;
  if (CT > 1 || live >= 4) {
    if (w < 4) project_body_w<D, MT, M0, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, 0, blk_off);
    else project_body_w<D, MT, MT - M0, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, M0, blk_off);
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
      case 1: project_body_w<D, MT, 1, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, m0, blk_off, c4, active, live); break;
      case 2: project_body_w<D, MT, 2, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, m0, blk_off, c4, active, live); break;
      case 3: project_body_w<D, MT, 3, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, m0, blk_off, c4, active, live); break;
      default: project_body_w<D, MT, 4, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, m0, blk_off, c4, active, live); break;
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

template <int D, int MT, int CT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lr_project(
    const float* __restrict__ W_old, const float* __restrict__ Dm, const float* __restrict__ Csub,
    float* __restrict__ T, long rows, int Ne, int NEP, GramPotrfJob job) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_gram = job.C ? gp_riders(job.nb) : 0;
  const int has_rider = job.C ? lr_rider_blocks(job.nb, D) : 0;
  if ((int)blockIdx.x < n_gram) {
    gram_potrf_rider<D>(job, smem_raw);
    return;
  }
  if ((int)blockIdx.x < has_rider) {
    solve_rider<D>(job, smem_raw, (int)blockIdx.x - 1);   // column blocks 0 .. n_gram - 2 belong to the Gram riders
    return;
  }
  DBG(0);
  float* Wc = (float*)smem_raw;                       // [2][MT*16][PJ_LD]
  project_dispatch<D, MT, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, has_rider);
  DBG(1);
}

// ---------------------------------------------------------------------------------------------
// update, single-buffered R, buffer addressing.
//  * The register holding k-step t's R fragment is reloaded IN PLACE with the next column group's
//    fragment right after the MFMAs of step t have consumed it.  Program order of the vector-memory
//    stream per group g is   W prefetch(g+1) | R(g+1)[0..NK-1] (between the MFMA steps) | stores(g),
//    so the wait for R(g+1)[t] at step t of group g+1 covers only loads issued a whole MFMA loop
//    earlier: the in-order vmcnt never makes a step wait for a load issued during the current group.
//    Half the registers of a double-buffered fragment set (the predecessor of this kernel: 34.6 us vs 30 us).
//  * Every stream is a buffer load/store: ONE 32-bit lane offset for all of them, the per-step /
//    per-row / per-group displacement folded into the (scalar) resource base, and the resource's
//    num_records doing the bounds work: concept rows >= N_edit read as 0 and weight rows >= rows are
//    neither read nor written - no clamped 64-bit address pairs (2 VGPRs per stream with flat
//    addressing), no per-row branches.  (The scalar offset operand is not part of the hardware range
//    check, hence base shifting instead of soffset.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0 ? bytes : 0, 0x00020000);
}

template <int D, int UP_MT, int WPE, int NK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lr_update_s(
    const float* W_old, const float* __restrict__ T, const float* __restrict__ R,
    float* W_new, long rows, int Ne, int NEP, int t_stride) {
  // (W_old may be W_new: the second pass of a 129..256-concept update adds onto the first pass's output in place - a
  //  workgroup reads only the 16 rows it writes, and every element is read before it is written)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int d = D;
  constexpr int SR = UP_MT * 16;
  const int tld = NEP + 2;
  float* Ts = (float*)smem_raw;                       // [SR][tld]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const long R0 = (long)blockIdx.x * SR;
  constexpr int MG = D / 256;

  const float* Wb = W_old + R0 * d;                   // workgroup-uniform bases
  float* Ob = W_new + R0 * d;
  const int w_bytes = (int)((rows - R0) < SR ? (rows - R0) : SR) * d * 4;   // this tile's valid weight bytes
  const int r_bytes = Ne * d * 4;
  // the one lane offset (bytes): R row lk / W row 4*lk of the step's / tile's base, columns w*64 + 4*li
  const unsigned vo_r = (unsigned)((lk * d + w * 64 + 4 * li) * 4);
  const unsigned vo_w = (unsigned)((4 * lk * d + w * 64 + 4 * li) * 4);

  auto ld_r = [&](int t, int gi) -> float4_t {
    const int sh = (4 * t * d + gi * 256) * 4;
    return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(R + 4 * t * d + gi * 256, r_bytes - sh), vo_r, 0, 0));
  };
  auto ld_w = [&](int m, int r, int gi) -> float4_t {
    const int sh = ((m * 16 + r) * d + gi * 256) * 4;
    return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
        make_rsrc(Wb + (m * 16 + r) * d + gi * 256, w_bytes - sh), vo_w, 0, 0));
  };

  float4_t rr[NK];
  float4_t res[UP_MT][4];
#pragma unroll
  for (int t = 0; t < NK; ++t) rr[t] = ld_r(t, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < UP_MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) res[m][r] = ld_w(m, r, 0);
  __builtin_amdgcn_sched_barrier(0);
  {
    const int f4_row = NEP >> 2;
    for (int e = tid; e < SR * f4_row; e += 256) {
      const int r = e / f4_row, c = (e - r * f4_row) << 2;
      long gr = R0 + r;
      gr = gr < rows ? gr : rows - 1;
      const float4_t v = *(const float4_t*)(T + gr * t_stride + c);
      Ts[r * tld + c] = c < Ne ? v[0] : 0.f;          // pad columns -> 0: the k loop reads unconditionally
      Ts[r * tld + c + 1] = c + 1 < Ne ? v[1] : 0.f;
      Ts[r * tld + c + 2] = c + 2 < Ne ? v[2] : 0.f;
      Ts[r * tld + c + 3] = c + 3 < Ne ? v[3] : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int gi = 0; gi < MG; ++gi) {
    float4_t acc[UP_MT][4];                           // acc[m][q][r]: row m*16 + 4*lk + r, column 4*li + q
#pragma unroll
    for (int m = 0; m < UP_MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][q][r] = res[m][r][q];
    if (gi + 1 < MG) {
#pragma unroll
      for (int m = 0; m < UP_MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[m][r] = ld_w(m, r, gi + 1);
    }
    float a[UP_MT];                                   // T fragments, read from LDS one step ahead
#pragma unroll
    for (int m = 0; m < UP_MT; ++m) a[m] = Ts[(m * 16 + li) * tld + lk];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NK; ++t) {
      float an[UP_MT];
#pragma unroll
      for (int m = 0; m < UP_MT; ++m) an[m] = (t + 1 < NK) ? Ts[(m * 16 + li) * tld + 4 * (t + 1) + lk] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < UP_MT; ++m)
          acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], rr[t][q], acc[m][q], 0, 0, 0);
      if (gi + 1 < MG) rr[t] = ld_r(t, gi + 1);       // reload in place: next group's fragment for step t
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < UP_MT; ++m) a[m] = an[m];
    }
#pragma unroll
    for (int m = 0; m < UP_MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4_t o = {acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]};
        const int sh = ((m * 16 + r) * d + gi * 256) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o),
                                               make_rsrc(Ob + (m * 16 + r) * d + gi * 256, w_bytes - sh), vo_w, 0,
                                               2 /* nt */);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// rows / 16 tiles over 256 CUs with MT tiles per workgroup: time ~ ceil(workgroups / 256) * MT.  `extra`: the rider
// blocks of the launch - they hold a CU each while the chain runs, so a projection workgroup beyond 256 - extra would
// start late and stretch the launch.
int pick_mt2(long rows, int extra) {
  const long t16 = (rows + 15) / 16;
  int best = 8;
  long best_cost = -1;
  for (int mt = 8; mt >= 5; --mt) {
    const long wgs = (t16 + mt - 1) / mt + extra;
    const long cost = ((wgs + 255) / 256) * mt;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mt; }
  }
  return best;
}

// The projection with the persistent Cholesky of a dual system of 3 ... 16 diagonal blocks in its FIRST n_la workgroups
// (uce_potrf_la.h; 129 ... 1024 concepts with at most 128 of them edited: the factorisation needs nothing from the
// projection and the projection nothing from it - they used to be two launches in a row).  The factorisation's workgroups
// come first in the grid, so they are placed before any projection tile (its waits never depend on later workgroups).
template <int D, int MT, int CT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lr_project_la(
    const float* __restrict__ W_old, const float* __restrict__ Dm, const float* __restrict__ Csub,
    float* __restrict__ T, long rows, int Ne, int NEP, PotrfLaJob la, int n_la) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if ((int)blockIdx.x < n_la) {
    potrf_la_body(la, (int)blockIdx.x, n_la, smem_raw);
    return;
  }
  float* Wc = (float*)smem_raw;                       // [2][MT*16][PJ_LD]
  project_dispatch<D, MT, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, n_la);
}

template <int D, int MT, int CT>
int launch_project_la(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit, int NEP64,
                      const PotrfLaJob& la, int own, hipStream_t st) {
  size_t smem = (size_t)2 * (MT * 16 + 64 * CT) * PJ_LD * sizeof(float);
  if (smem < potrf_la_smem()) smem = potrf_la_smem();
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    // (what it needs, not the 160 KB the plain projection asks for: the hosted factorisation has a static LDS word)
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_lr_project_la<D, MT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_once.commit(tok);
  }
  const long nwg = (rows + MT * 16 - 1) / (MT * 16) + own;
  hipLaunchKernelGGL((k_lr_project_la<D, MT, CT>), dim3((unsigned)nwg), dim3(512), smem, st, W_old, Dm, Csub, T, rows, N_edit,
                     NEP64, la, own);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int D, int CT>
int launch_project_la_d(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit, int NEP64,
                        const PotrfLaJob& la, int own, hipStream_t st) {
  // (one tile height per width keeps the instantiations of the hosted factorisation few: 7 x 16 rows tile SD-1.4's slab in 223)
  if constexpr (D == 2048) return launch_project_la<D, 5, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, la, own, st);
  else return launch_project_la<D, 7, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, la, own, st);
}

template <int D, int MT, int CT>
int launch_project(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit,
                   int NEP64, const GramPotrfJob& job, hipStream_t st) {
  size_t smem = (size_t)2 * (MT * 16 + 64 * CT) * PJ_LD * sizeof(float);
  if (job.C && smem < gp_smem(job.nb)) smem = gp_smem(job.nb);
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_lr_project<D, MT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
    attr_once.commit(tok);
  }
  const long nwg = (rows + MT * 16 - 1) / (MT * 16) + (job.C ? lr_rider_blocks(job.nb, D) : 0);
  hipLaunchKernelGGL((k_lr_project<D, MT, CT>), dim3((unsigned)nwg), dim3(512), smem, st, W_old, Dm, Csub, T, rows,
                     N_edit, NEP64, job);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int D, int CT>
int launch_project_d(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit,
                     int NEP64, const GramPotrfJob& job, hipStream_t st) {
  if constexpr (D == 2048) return launch_project<D, 5, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
  else switch (pick_mt2(rows, job.C ? lr_rider_blocks(job.nb, D) : 0)) {
    case 5: return launch_project<D, 5, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
    case 6: return launch_project<D, 6, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
    case 7: return launch_project<D, 7, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
    default: return launch_project<D, 8, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
  }
}

// one pass W_new = W_in + T[:, 0 .. NEPw) R[0 .. N_w): T has row stride t_stride floats, N_w <= 128 concepts
template <int D, int UP_MT, int WPE>
int launch_update_s(const float* W_in, const float* T, const float* R, float* W_new, long rows, int N_w,
                    int NEPw, int t_stride, hipStream_t st) {
  const size_t smem = (size_t)UP_MT * 16 * (NEPw + 2) * sizeof(float);
  const dim3 grid((unsigned)((rows + UP_MT * 16 - 1) / (UP_MT * 16))), block(256);
  const int nks = (N_w + 3) / 4;
  if (nks <= 8)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 8>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else if (nks <= 13)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 13>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else if (nks <= 16)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 16>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else if (nks <= 25)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 25>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else if (nks <= 32)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 32>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else
    return UCE_EINVAL;
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int D>
int launch_update_d(const float* W_old, const float* T, const float* R, float* W_new, long rows, int N_edit,
                    int NEP64, hipStream_t st) {
  // 16-row tiles, 2 waves per SIMD: measured best at N_edit 16..128 on MI355X (32-row tiles: +4-9 %)
  // (3 / 4 waves per SIMD measured in round 3: SDXL slab 564 -> 610 / 537 us, 100 concepts 45 -> 45 / 78 us (spills): kept at 2)
  // (measured in round 3 on the SDXL slab and without effect beyond the 520-570 us run-to-run spread: 32-row tiles; weight
  // rows of two column groups in flight instead of one)
  if (N_edit <= 128) return launch_update_s<D, 1, 2>(W_old, T, R, W_new, rows, N_edit, NEP64, NEP64, st);
  // 129 .. 256 concepts: two passes of the register-resident kernel over 128-concept windows of T and R, the second one in
  // place on the first one's output (round 2's ring-buffered single-pass kernel for this range - R fragments through a
  // 4-deep register ring, in-order vmcnt stalls, spills at d = 2048 - measured 143 us against 117 at 256 concepts on the
  // SD-1.4 slab and 2093 against 1789 on the SDXL slab; it was 3-8 % ahead only just above 128 concepts)
  const int rc = launch_update_s<D, 1, 2>(W_old, T, R, W_new, rows, 128, 128, NEP64, st);
  if (rc) return rc;
  return launch_update_s<D, 1, 2>(W_new, T + 128, R + (size_t)128 * D, W_new, rows, N_edit - 128, NEP64 - 128, NEP64, st);
}

}  // namespace

int lr_rider_cap() { return 64 * GP_MAXB; }

bool lowrank_split_supported(int d, int N_edit) {
  return (d == 768 || d == 1024 || d == 2048) && N_edit >= 1 && N_edit <= 256;
}

// X = Dm with Csub == nullptr, or X = G with Csub = C_e (D_e = G - C_e formed on the fly).  With `h` (and N <= 128) the
// launch also builds and factors the dual system in its rider blocks and solves for R = rows of K^-1 C:
// K = lamb S^-1 + C C^T -> h->Linv (+ block (1, 0) of h->Lmat), h->status, R.  More than 64 edit concepts take the
// 128-concept batches (one pass over W per 128 concepts).
int launch_lr_project(const float* W_old, const float* X, const float* Csub, float* T, long rows, int d,
                      int N_edit, hipStream_t st, uce_ctx* h, const float* C, const float* s, int N, float lamb, float* R) {
  const int NEP64 = (N_edit + 63) / 64 * 64;
  GramPotrfJob job{};
  if (h) {
    const int nb = (N + 63) / 64;
    if (nb < 1 || nb > GP_MAXB || !R || !C || !s) return UCE_EINVAL;
    job = GramPotrfJob{C, s, N, lamb, h->slabs, h->ticket, h->Lmat, h->Linv, h->status, nb, R, N_edit};
  }
  const bool wide = NEP64 > 64;
  if (d == 768)
    return wide ? launch_project_d<768, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st)
                : launch_project_d<768, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st);
  if (d == 1024)
    return wide ? launch_project_d<1024, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st)
                : launch_project_d<1024, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st);
  if (d == 2048)
    return wide ? launch_project_d<2048, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st)
                : launch_project_d<2048, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st);
  return UCE_EINVAL;
}

int launch_lr_project_la(const float* W_old, const float* X, const float* Csub, float* T, long rows, int d, int N_edit,
                         const PotrfLaJob& la, int own, hipStream_t st) {
  const int NEP64 = (N_edit + 63) / 64 * 64;
  if (N_edit < 1 || N_edit > 128) return UCE_EINVAL;
  const bool wide = NEP64 > 64;
  if (d == 768)
    return wide ? launch_project_la_d<768, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st)
                : launch_project_la_d<768, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st);
  if (d == 1024)
    return wide ? launch_project_la_d<1024, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st)
                : launch_project_la_d<1024, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st);
  if (d == 2048)
    return wide ? launch_project_la_d<2048, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st)
                : launch_project_la_d<2048, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st);
  return UCE_EINVAL;
}

int launch_lr_update(const float* W_old, const float* T, const float* R, float* W_new, long rows, int d,
                     int N_edit, hipStream_t st) {
  const int NEP64 = (N_edit + 63) / 64 * 64;
  if (d == 768) return launch_update_d<768>(W_old, T, R, W_new, rows, N_edit, NEP64, st);
  if (d == 1024) return launch_update_d<1024>(W_old, T, R, W_new, rows, N_edit, NEP64, st);
  if (d == 2048) return launch_update_d<2048>(W_old, T, R, W_new, rows, N_edit, NEP64, st);
  return UCE_EINVAL;
}
