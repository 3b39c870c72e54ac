// Two-kernel form of the low-rank apply  W_new = W_old + (W_old D_e^T) R_e :
//
//   k_lr_project<D,MT>  : T [rows, NEP] = W_old D_e^T     needs only D_e = G - C_e, not the solve: in uce_edit ONE
//                         launch carries this GEMM and, in its first 4 / 12 "rider" blocks, the Gram matrix and the
//                         (blocked) Cholesky + block inverses of the dual system (N <= 64 / N <= 128), so the
//                         latency-bound factorisation hides under the f32-MFMA-bound projection without any
//                         cross-stream event (a HIP event hand-off between two streams measured 13-14 us each way).
//   k_lr_update_s<D,..> : W_new = W_old + T R_e           one pass over the weights (N_edit <= 128; k_lr_update is the
//                         ring-buffered form for 129..256): algorithmic bytes 8*rows*d (+ the small T and R).
//
// Splitting the update at T costs 2*4*rows*NEP bytes of extra traffic (6.4 MB at N_edit <= 64 for SD-1.4, 4 %) and
// buys (a) overlap of the projection with the latency-bound small-system chain, (b) an update kernel whose LDS
// footprint is tiny, so several workgroups per CU stream W with their phases naturally interleaved.
#include "uce_common.h"
#include "uce_potrf64.h"
#include <cstdlib>

// -DUCE_CHAIN_DEBUG: wall-clock stamps (100 MHz) of the rider chain's phases, read back with uce_debug_read
// (tools/dbg_chain.py); compiled out of the product library.
#ifdef UCE_CHAIN_DEBUG
__device__ unsigned long long g_dbg[64][16];
#define DBG(slot) do { if (threadIdx.x == 0 && blockIdx.x < 64) { g_dbg[blockIdx.x][slot] = wall_clock64(); g_dbg[blockIdx.x][8 + slot] = clock64(); } } while (0)
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
// projection: 8 waves, wave = (concept tile class c4 = w & 3, M half = w >> 2); the W k-chunk of the
// MT*16-row super-tile and the D_e k-chunk (64 concepts) are staged once in LDS and shared by the waves;
// each D_e fragment read from LDS feeds NMT MFMAs.
// ---------------------------------------------------------------------------------------------
template <int D, int MT, int NMT>
__device__ __forceinline__ void project_body(const float* __restrict__ W_old, const float* __restrict__ Dm,
                                             const float* __restrict__ Csub, float* __restrict__ T,
                                             long rows, int Ne, int NEP, float* Wc, int mbase, int blk_off) {
  constexpr int d = D;
  constexpr int SR = MT * 16;
  float* Dc = Wc + 2 * SR * PJ_LD;                    // [2][64][PJ_LD]  D_e k-chunk of the current batch
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c4 = w & 3;
  const int li = lane & 15, lk = lane >> 4;
  const long R0 = (long)(blockIdx.x - blk_off) * SR;

  constexpr int NC = D / PJ_KC;                       // k-chunks (12 / 16 / 32), even
  constexpr int F4 = SR * (PJ_KC / 4);                // float4 per W chunk
  constexpr int NLD = (F4 + 511) / 512;               // per thread
  struct Stage { float4_t w[NLD]; float4_t x[2]; float4_t y[2]; };
  const float cscale = Csub ? 1.f : 0.f;              // D_e = X - cscale * Y (X = G, Y = C_e) or X = Dm
  const float* Ysrc = Csub ? Csub : Dm;
  const int nbatch = NEP >> 6;
#pragma unroll 1
  for (int bt = 0; bt < nbatch; ++bt) {
    // both operands come in as full 256-byte row segments (16 lanes x 16 B) and go through LDS: the
    // W rows of the super-tile and the 64 concept rows of this batch (masked beyond N_edit)
    auto load_stage = [&](int kc, Stage& st) {
#pragma unroll
      for (int p = 0; p < NLD; ++p) {
        const int e = tid + 512 * p;
        const int r = (e >> 4) < SR ? (e >> 4) : SR - 1, cc = (e & 15) << 2;
        long gr = R0 + r;
        gr = gr < rows ? gr : rows - 1;
        st.w[p] = *(const float4_t*)(W_old + gr * d + kc * PJ_KC + cc);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int e = tid + 512 * p;                  // 64 rows x 16 float4
        const int er = bt * 64 + (e >> 4), cc = (e & 15) << 2;
        const size_t off = (size_t)(er < Ne ? er : Ne - 1) * d + kc * PJ_KC + cc;
        st.x[p] = *(const float4_t*)(Dm + off);
        st.y[p] = *(const float4_t*)(Ysrc + off);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto park_stage = [&](int buf, const Stage& st) {
#pragma unroll
      for (int p = 0; p < NLD; ++p) {
        const int e = tid + 512 * p;
        if (e < F4) *(float4_t*)&Wc[(buf * SR + (e >> 4)) * PJ_LD + ((e & 15) << 2)] = st.w[p];
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int e = tid + 512 * p;
        const float m = (bt * 64 + (e >> 4) < Ne) ? 1.f : 0.f;
        *(float4_t*)&Dc[(buf * 64 + (e >> 4)) * PJ_LD + ((e & 15) << 2)] = (st.x[p] - cscale * st.y[p]) * m;
      }
    };
    float4_t acc[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) acc[m] = (float4_t){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
      // k permutation: MFMA q of 16-k group g uses k = 16g + 4*(lane>>4) + q on both operands
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4_t a[NMT];
#pragma unroll
        for (int m = 0; m < NMT; ++m)
          a[m] = *(const float4_t*)&Wc[(buf * SR + (mbase + m) * 16 + li) * PJ_LD + g * 16 + 4 * lk];
        const float4_t bb = *(const float4_t*)&Dc[(buf * 64 + c4 * 16 + li) * PJ_LD + g * 16 + 4 * lk];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int m = 0; m < NMT; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][q], bb[q], acc[m], 0, 0, 0);
      }
    };
    // chunk c: loaded during iteration c-2 into register set c&1, parked in LDS buffer c&1 during
    // iteration c-1, consumed in iteration c.
    Stage sA, sB;
    __syncthreads();                                  // previous batch is done with the LDS buffers
    load_stage(0, sA);
    load_stage(1, sB);
    park_stage(0, sA);
    load_stage(2 < NC ? 2 : 0, sA);
    __syncthreads();
#pragma unroll 1
    for (int kc = 0; kc < NC; kc += 2) {
      park_stage(1, sB);                              // chunk kc + 1
      load_stage(kc + 3 < NC ? kc + 3 : kc, sB);
      compute(0);                                     // chunk kc
      __syncthreads();
      if (kc + 2 < NC) park_stage(0, sA);             // chunk kc + 2
      load_stage(kc + 4 < NC ? kc + 4 : kc, sA);
      compute(1);                                     // chunk kc + 1
      __syncthreads();
    }
    // D layout: col = lane & 15 (concept), row = 4*(lane>>4) + r
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long gr = R0 + (mbase + m) * 16 + 4 * lk + r;
        if (gr < rows) T[gr * NEP + (bt * 4 + c4) * 16 + li] = acc[m][r];
      }
  }
}

// Optional riders of the projection launch (blocks 0..GP_NB-1): the whole small-system factorisation
// of the dual form when it is a single 64-block (N <= 64):  K = lambda S^-1 + C C^T  (f64 MFMA over
// the d features, split over GP_NB blocks x 2 wave quads), then the 64x64 Cholesky + inverse by the
// rider block that finishes its slab LAST.  Riders need nothing from the projection and vice versa,
// so riding along costs no launch and no event, and - being shorter than the GEMM beside them - no
// time.  The slab hand-off between rider blocks is the split-K reduction of the CDNA guide (G16) in its write-through
// form: sc1 slab stores -> per-wave vmcnt(0) -> barrier -> one lane draws a relaxed agent-scope ticket; the block
// drawing the last ticket reads all slabs with sc1 loads (summed in slab order: bit-repeatable) - no release / acquire
// fence on either side (st_sc1 below).  Correct for any placement of the rider blocks; the ticket word is zero at
// creation and reset by its last taker.
struct GramPotrfJob {
  const float* C;       // [N, d]; null = no riders
  const float* s;       // [N]
  int N;
  float lamb;
  double* slabs;        // [tiles * GP_NB][64][64] partial Grams
  unsigned* ticket;     // one word, zero between launches
  double* Lmat;         // [n, n], n = 64 * nb
  double* Linv;         // [nb][64][64]
  int* status;
  double* M;            // [n, n] assembled system (nb > 1 only)
  int nb;               // 64-blocks of the dual system handled by the riders: 1 or 2
  float* R;             // [N_edit, d] rows of K^-1 C, written by the solve riders (null: no solve riders)
  int N_edit;
  unsigned seq;         // value of ticket[1] that announces THIS launch's factorisation
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
__device__ __forceinline__ double ld_sc1(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The factorising block tells the solve riders that L / L^-1 of this launch are in memory: all its stores drained,
// one agent-scope release, then the sequence word (the hand-off recipe of the CDNA guide, Guideline 16).
// `fence`: the payload was written with plain stores (the 128 x 128 path shares its tile bodies with the launch chain)
// and needs the agent-scope release; the single-tile path stores L / L^-1 write-through and skips it.
__device__ __forceinline__ void announce_factor(const GramPotrfJob& j, bool fence) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (fence) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_store(j.ticket + 1, j.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------
// Solve riders (blocks after the Gram riders, one per SV_COLS = 32 columns of R): wait for the factorisation of THIS launch,
// then  R[:, cols] = rows 0..N_edit-1 of  L^-T L^-1 C[:, cols]  with the inverted diagonal blocks - the triangular
// solves that used to be a launch of their own between the projection and the update.  f64 tiles in LDS (stride SV_LD),
// 8 waves x one 16 x 16 MFMA tile, contraction over the non-zero part of the triangular operand (sv_prod).
// ---------------------------------------------------------------------------------------------
constexpr int SV_LD = 66;
constexpr int SV_COLS = 32;     // columns of R per solve rider: 8 waves x one 16 x 16 tile, twice as many riders as 64 would give
constexpr size_t SV_TILE = (size_t)64 * SV_LD * sizeof(double);
__host__ __device__ constexpr size_t sv_smem(int nb) { return (nb <= 1 ? 3 : 4) * SV_TILE; }

// dst = base - / + op(A) * B :  A [64][64], B / dst / base [64][SV_COLS] are LDS tiles of stride SV_LD; TA: op(A)[i][k] =
// A[k][i].  B is read as [k][col].  base == nullptr: dst = op(A) B.  (dst may alias base, never A or B.)
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
    fb[kb] = B[t * SV_LD + col0 + r];
  }
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
    if (kb >= kb0 && kb < kb1) acc = mfma_f64(fa[kb], fb[kb], acc);
  // D layout: row = kk + 4q, col = r
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o = (row0 + kk + 4 * q) * SV_LD + col0 + r;
    dst[o] = (base ? base[o] : 0.0) + sign * acc[q];
  }
}

template <int D>
__device__ __forceinline__ void solve_rider(const GramPotrfJob& j, unsigned char* smem_raw, int colblk) {
  double* Ms = (double*)smem_raw;                       // the matrix operand of the current product
  double* W0 = Ms + 64 * SV_LD;
  double* W1 = W0 + 64 * SV_LD;
  double* W2 = W1 + 64 * SV_LD;                         // (nb == 2 only)
  const int tid = threadIdx.x;
  const int n = 64 * j.nb;
  // 64 concepts x this block's 64 columns of C (rows >= N are zero) -> LDS, widened to f64.  Needs nothing from the
  // factorisation: loaded before the wait.
  auto load_c = [&](double* Wt, int kblk) {
    for (int e = tid; e < 64 * (SV_COLS / 4); e += 512) {
      const int r = e / (SV_COLS / 4), c4 = (e % (SV_COLS / 4)) << 2;
      const int row = kblk * 64 + r;
      float4_t v = {0.f, 0.f, 0.f, 0.f};
      if (row < j.N) v = *(const float4_t*)(j.C + (size_t)row * D + colblk * SV_COLS + c4);
      Wt[r * SV_LD + c4] = (double)v[0];
      Wt[r * SV_LD + c4 + 1] = (double)v[1];
      Wt[r * SV_LD + c4 + 2] = (double)v[2];
      Wt[r * SV_LD + c4 + 3] = (double)v[3];
    }
  };
  auto load_m = [&](const double* G, int ld) {            // a 64 x 64 block of a row-major f64 matrix -> Ms
    if (j.nb == 1) {                                      // published write-through: read it past the L1
      double v[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int e = tid + 512 * p;
        v[p] = ld_sc1(G + (size_t)(e >> 6) * ld + (e & 63));
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int e = tid + 512 * p;
        Ms[(e >> 6) * SV_LD + (e & 63)] = v[p];
      }
      return;
    }
    for (int e = tid; e < 64 * 32; e += 512) {
      const int r = e >> 5, c2 = (e & 31) << 1;
      typedef double double2_t __attribute__((ext_vector_type(2)));
      const double2_t v = *(const double2_t*)(G + (size_t)r * ld + c2);
      Ms[r * SV_LD + c2] = v[0];
      Ms[r * SV_LD + c2 + 1] = v[1];
    }
  };
  auto store_r = [&](const double* Wt, int kblk) {        // rows of X -> R (fp32), rows < N_edit only
    for (int e = tid; e < 64 * (SV_COLS / 4); e += 512) {
      const int r = e / (SV_COLS / 4), c4 = (e % (SV_COLS / 4)) << 2;
      const int row = kblk * 64 + r;
      if (row < j.N_edit)
        *(float4_t*)(j.R + (size_t)row * D + colblk * SV_COLS + c4) =
            (float4_t){(float)Wt[r * SV_LD + c4], (float)Wt[r * SV_LD + c4 + 1], (float)Wt[r * SV_LD + c4 + 2],
                       (float)Wt[r * SV_LD + c4 + 3]};
    }
  };
  DBG(0);
  load_c(W0, 0);
  DBG(1);
  // ---- wait for this launch's factorisation: one lane polls (relaxed, with s_sleep), one acquire, then plain loads
  if (tid == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(j.ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != j.seq) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 24)) {                          // ~ seconds: the factorising block never ran (cannot happen
        atomicCAS(j.status, 0, -1);                        //   with in-order dispatch); report instead of hanging
        break;
      }
    }
    if (j.nb != 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  DBG(2);
  const double* Linv0 = j.Linv;
  if (j.nb == 1) {
    load_m(Linv0, 64);
    __syncthreads();
    DBG(3);
    sv_prod<false, true>(W1, Ms, W0, nullptr, 1.0);       // Y = L^-1 C
    __syncthreads();
    sv_prod<true, true>(W0, Ms, W1, nullptr, 1.0);        // X = L^-T Y
    __syncthreads();
    DBG(4);
    store_r(W0, 0);
    DBG(5);
    return;
  }
  const double* Linv1 = j.Linv + 4096;
  const double* L10 = j.Lmat + (size_t)64 * n;            // block (1, 0) of L
  load_m(Linv0, 64);
  __syncthreads();
  sv_prod<false, true>(W1, Ms, W0, nullptr, 1.0);         // Y0 = L00^-1 C0
  __syncthreads();
  load_c(W0, 1);
  load_m(L10, n);
  __syncthreads();
  sv_prod<false, false>(W0, Ms, W1, W0, -1.0);            // C1 - L10 Y0
  __syncthreads();
  load_m(Linv1, 64);
  __syncthreads();
  sv_prod<false, true>(W2, Ms, W0, nullptr, 1.0);         // Y1
  __syncthreads();
  sv_prod<true, true>(W0, Ms, W2, nullptr, 1.0);          // X1 = L11^-T Y1
  __syncthreads();
  store_r(W0, 1);
  load_m(L10, n);
  __syncthreads();
  sv_prod<true, false>(W1, Ms, W0, W1, -1.0);             // Y0 - L10^T X1
  __syncthreads();
  load_m(Linv0, 64);
  __syncthreads();
  sv_prod<true, true>(W2, Ms, W1, nullptr, 1.0);          // X0
  __syncthreads();
  store_r(W2, 0);
}

template <int D>
__device__ __forceinline__ void gram_potrf_rider(const GramPotrfJob& j, unsigned char* smem_raw) {
  DBG(0);
  float* As = (float*)smem_raw;                                   // [2 halves][64][GP_LD]  rows of block ti
  float* Bs = As + 2 * 64 * GP_LD;                                // [2 halves][64][GP_LD]  rows of block tk
  constexpr size_t AB_BYTES = (size_t)4 * 64 * GP_LD * sizeof(float);
  double* P1 = (double*)(smem_raw + AB_BYTES);                    // [64][GP_TLD]
  // the factorisation scratch ALIASES the Gram staging (As, Bs, P1 are dead once the slab is published), so a
  // single-tile rider needs 74 KB and two workgroups of the launch fit a CU
  Potrf64Scratch* sc = (Potrf64Scratch*)smem_raw;
  static_assert(sizeof(Potrf64Scratch) <= AB_BYTES + 64 * GP_TLD * sizeof(double), "scratch must fit the Gram staging");
  // (all LDS in the dynamic region: a static __shared__ would shift its 16-byte alignment)
  unsigned* s_last_p = (unsigned*)(smem_raw + AB_BYTES + 64 * GP_TLD * sizeof(double));
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
    *s_last_p = (t == (unsigned)nriders - 1) ? 1u : 0u;
    if (t == (unsigned)nriders - 1) {
      __hip_atomic_store(j.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      *j.status = 0;
    }
  }
  __syncthreads();
  DBG(2);
  if (!*s_last_p) return;
  auto diag_term = [&](int row) -> double {
    const float sv = (row < j.N) ? j.s[row] : 1.f;
    return (row < j.N) ? ((sv > 0.f) ? (double)j.lamb / (double)sv : __builtin_nan("")) : 1.0;
  };
  if (j.nb == 1) {
    // last arriver: all 8 waves factor - waves 0-3 carry the matrix tiles, waves 4-7 the tiles of L^-1.
    // The GP_NB slabs are summed by ALL 512 threads (8 elements each, every load independent and in flight at
    // once, fixed slab order: bit-repeatable) into the LDS tile the matrix waves then pick their 4x4 tiles from.
    double* Ksum = (double*)smem_raw;                             // [64][GP_TLD] (the Gram staging is dead)
    {
      // only what the factorisation reads: the 4 x 4 tiles of the lower triangle, rows of real concepts (the padding
      // rows are identity) - a 50-concept system moves 35 % of the slab bytes
      const int n4 = (j.N + 3) & ~3;
      double v[GP_NB][8];
#pragma unroll
      for (int b = 0; b < GP_NB; ++b)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int e = tid + 512 * p, row = e >> 6, col = e & 63;
          v[b][p] = (row < n4 && col <= (row | 3)) ? ld_sc1(&j.slabs[(size_t)b * 4096 + e]) : 0.0;
        }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int e = tid + 512 * p, row = e >> 6, col = e & 63;
        double acc = v[0][p];
#pragma unroll
        for (int b = 1; b < GP_NB; ++b) acc += v[b][p];
        if (row == col) acc = (row < n4 ? acc : 0.0) + diag_term(row);
        Ksum[row * GP_TLD + col] = acc;
      }
    }
    __syncthreads();
    DBG(3);
    UCE_POTRF64([&](int row, int col, double (&v)[4]) {
                  const pk_d2 a = *(const pk_d2*)&Ksum[row * GP_TLD + col], b = *(const pk_d2*)&Ksum[row * GP_TLD + col + 2];
                  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
                },
                [&](int row, int col, const double (&v)[4]) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) st_sc1(&j.Linv[row * 64 + col + e], v[e]);
                },
                sc, tid, j.status, 0, j.N);
    DBG(4);
    announce_factor(j, false);
    DBG(5);
    return;
  }
  // nb == 2: the last arriver assembles the 128 x 128 system and runs the blocked factorisation that
  // uce_solve.hip spreads over a chain of launches (same tile bodies), alone, under the projection GEMM:
  // factor (0,0) -> tile (1,1) of step 0 (forms L_10, updates and factors block 1).
  const int n = 64 * j.nb;
  for (int t = 0; t < 3; ++t) {
    const int gi = t == 0 ? 0 : 1, gk = t == 2 ? 1 : 0;
    for (int e = tid; e < 64 * 64; e += 512) {
      const int r = e >> 6, cc = e & 63;
      double v = ld_sc1(&j.slabs[(size_t)(t * GP_NB) * 4096 + e]);
#pragma unroll
      for (int b = 1; b < GP_NB; ++b) v += ld_sc1(&j.slabs[(size_t)(t * GP_NB + b) * 4096 + e]);        // fixed order
      const int grow = gi * 64 + r, gcol = gk * 64 + cc;
      if (grow == gcol) v += diag_term(grow);
      j.M[(size_t)grow * n + gcol] = v;
    }
  }
  __syncthreads();                                    // the block's own global writes -> visible to the block
  potrf_first_body8(j.M, n, 1, 0, j.Lmat, j.Linv, j.status, (Potrf64Scratch*)smem_raw, j.N);
  __syncthreads();
  potrf_step_tile(j.M, n, 0, 1, 1, j.Lmat, j.Linv, j.status, smem_raw, j.N);
  announce_factor(j, true);
}

constexpr size_t GP_SMEM1 = (size_t)4 * 64 * GP_LD * sizeof(float) + 64 * GP_TLD * sizeof(double) + 16;   // one system tile
constexpr size_t GP_SMEM = GP_SMEM1 > POTRF_STEP_SMEM + 16 ? GP_SMEM1 : POTRF_STEP_SMEM + 16;              // 128 x 128 systems
__host__ __device__ constexpr size_t gp_smem(int nb) {
  const size_t g = nb <= 1 ? GP_SMEM1 : GP_SMEM;
  return g > sv_smem(nb) ? g : sv_smem(nb);
}

template <int D, int MT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lr_project(
    const float* __restrict__ W_old, const float* __restrict__ Dm, const float* __restrict__ Csub,
    float* __restrict__ T, long rows, int Ne, int NEP, GramPotrfJob job) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_gram = job.C ? gp_riders(job.nb) : 0;
  const int n_solve = (job.C && job.R) ? D / SV_COLS : 0;
  const int has_rider = n_gram + n_solve;
  if ((int)blockIdx.x < n_gram) {
    gram_potrf_rider<D>(job, smem_raw);
    return;
  }
  if ((int)blockIdx.x < has_rider) {
    solve_rider<D>(job, smem_raw, (int)blockIdx.x - n_gram);
    return;
  }
  DBG(0);
  float* Wc = (float*)smem_raw;                       // [2][MT*16][PJ_LD]
  constexpr int M0 = (MT + 1) / 2;
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) < 256)
    project_body<D, MT, M0>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, 0, has_rider);
  else
    project_body<D, MT, MT - M0>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, M0, has_rider);
  DBG(1);
}

// ---------------------------------------------------------------------------------------------
// update: 4 waves, 64 rows per workgroup; wave w owns 64-column groups w, w+4, ...; lane j of a group
// owns 4 consecutive columns, so W, R and the output all move 16 B per lane in full 256 B row
// segments.  T tile [64, NEP] -> LDS once.  Per k-step one R fragment (L2) feeds 16 MFMAs.
// ---------------------------------------------------------------------------------------------
template <int D, int UP_MT, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lr_update(
    const float* __restrict__ W_old, const float* __restrict__ T, const float* __restrict__ R,
    float* __restrict__ W_new, long rows, int Ne, int NEP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int d = D;
  constexpr int SR = UP_MT * 16;
  const int tld = NEP + 2;
  float* Ts = (float*)smem_raw;                       // [64][tld]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const long R0 = (long)blockIdx.x * SR;

  constexpr int MG = D / 256;                         // column groups per wave (3 / 4 / 8)
  constexpr int RD = 4;                               // R fragments in flight
  const int nks = (Ne + 3) >> 2;                      // k-steps that carry concepts
  int rl_g = 0, rl_t = 0;
  auto r_next = [&]() -> float4_t {
    const int e = 4 * rl_t + lk;
    const float4_t v = *(const float4_t*)(R + (size_t)(e < Ne ? e : Ne - 1) * d + (w + 4 * rl_g) * 64 + 4 * li);
    if (++rl_t == nks) { rl_t = 0; rl_g = rl_g + 1 < MG ? rl_g + 1 : rl_g; }
    return v;
  };
  auto res_load = [&](int gi, float4_t (&x)[UP_MT][4]) {
#pragma unroll
    for (int m = 0; m < UP_MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        long gr = R0 + m * 16 + 4 * lk + r;
        gr = gr < rows ? gr : rows - 1;
        x[m][r] = *(const float4_t*)(W_old + gr * d + (w + 4 * gi) * 64 + 4 * li);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  // first loads in flight before T is staged
  float4_t res[UP_MT][4];
  res_load(0, res);
  float4_t ring[RD];
#pragma unroll
  for (int i = 0; i < RD; ++i) ring[i] = r_next();
  {
    const int f4_row = NEP >> 2;
    for (int e = tid; e < SR * f4_row; e += 256) {
      const int r = e / f4_row, c = (e - r * f4_row) << 2;
      long gr = R0 + r;
      gr = gr < rows ? gr : rows - 1;
      const float4_t v = *(const float4_t*)(T + gr * NEP + c);
      Ts[r * tld + c] = v[0];
      Ts[r * tld + c + 1] = v[1];
      Ts[r * tld + c + 2] = v[2];
      Ts[r * tld + c + 3] = v[3];
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
    if (gi + 1 < MG) res_load(gi + 1, res);           // next group's W rows, ahead of this group's stores
#pragma unroll 1
    for (int t = 0; t < nks; ++t) {
      const float4_t b = ring[0];
#pragma unroll
      for (int i = 0; i + 1 < RD; ++i) ring[i] = ring[i + 1];
      ring[RD - 1] = r_next();
      const int e = 4 * t + lk;
      float a[UP_MT];
#pragma unroll
      for (int m = 0; m < UP_MT; ++m) a[m] = (e < Ne) ? Ts[(m * 16 + li) * tld + e] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < UP_MT; ++m)
          acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[q], acc[m][q], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < UP_MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long gr = R0 + m * 16 + 4 * lk + r;
        if (gr < rows) {
          const float4_t o = {acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]};
          __builtin_nontemporal_store(o, (float4_t*)(W_new + gr * d + (w + 4 * gi) * 64 + 4 * li));
        }
      }
  }
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
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0 ? bytes : 0, 0x00020000);
}

template <int D, int UP_MT, int WPE, int NK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lr_update_s(
    const float* __restrict__ W_old, const float* __restrict__ T, const float* __restrict__ R,
    float* __restrict__ W_new, long rows, int Ne, int NEP) {
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
      const float4_t v = *(const float4_t*)(T + gr * NEP + c);
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

// rows / 16 tiles over 256 CUs with MT tiles per workgroup: time ~ ceil(workgroups / 256) * MT
int pick_mt2(long rows) {
  const long t16 = (rows + 15) / 16;
  int best = 8;
  long best_cost = -1;
  for (int mt = 8; mt >= 5; --mt) {
    const long wgs = (t16 + mt - 1) / mt;
    const long cost = ((wgs + 255) / 256) * mt;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mt; }
  }
  return best;
}

template <int D, int MT>
int launch_project(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit,
                   int NEP64, const GramPotrfJob& job, hipStream_t st) {
  size_t smem = (size_t)2 * (MT * 16 + 64) * PJ_LD * sizeof(float);
  if (job.C && smem < gp_smem(job.nb)) smem = gp_smem(job.nb);
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_lr_project<D, MT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
  }
  const long nwg = (rows + MT * 16 - 1) / (MT * 16) + (job.C ? gp_riders(job.nb) + (job.R ? D / SV_COLS : 0) : 0);
  hipLaunchKernelGGL((k_lr_project<D, MT>), dim3((unsigned)nwg), dim3(512), smem, st, W_old, Dm, Csub, T, rows,
                     N_edit, NEP64, job);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int D>
int launch_project_d(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit,
                     int NEP64, const GramPotrfJob& job, hipStream_t st) {
  switch (pick_mt2(rows)) {
    case 5: return launch_project<D, 5>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
    case 6: return launch_project<D, 6>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
    case 7: return launch_project<D, 7>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
    default: return launch_project<D, 8>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
  }
}

template <int D, int UP_MT, int WPE>
int launch_update_v(const float* W_old, const float* T, const float* R, float* W_new, long rows, int N_edit,
                    int NEP64, hipStream_t st) {
  const size_t smem = (size_t)UP_MT * 16 * (NEP64 + 2) * sizeof(float);
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_lr_update<D, UP_MT, WPE>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  const long nwg = (rows + UP_MT * 16 - 1) / (UP_MT * 16);
  hipLaunchKernelGGL((k_lr_update<D, UP_MT, WPE>), dim3((unsigned)nwg), dim3(256), smem, st, W_old, T, R, W_new, rows,
                     N_edit, NEP64);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int D, int UP_MT, int WPE>
int launch_update_s(const float* W_old, const float* T, const float* R, float* W_new, long rows, int N_edit,
                    int NEP64, hipStream_t st) {
  const size_t smem = (size_t)UP_MT * 16 * (NEP64 + 2) * sizeof(float);
  const dim3 grid((unsigned)((rows + UP_MT * 16 - 1) / (UP_MT * 16))), block(256);
  const int nks = (N_edit + 3) / 4;
  if (nks <= 8)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 8>), grid, block, smem, st, W_old, T, R, W_new, rows, N_edit, NEP64);
  else if (nks <= 13)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 13>), grid, block, smem, st, W_old, T, R, W_new, rows, N_edit, NEP64);
  else if (nks <= 16)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 16>), grid, block, smem, st, W_old, T, R, W_new, rows, N_edit, NEP64);
  else if (nks <= 25)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 25>), grid, block, smem, st, W_old, T, R, W_new, rows, N_edit, NEP64);
  else if (nks <= 32)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 32>), grid, block, smem, st, W_old, T, R, W_new, rows, N_edit, NEP64);
  else
    return UCE_EINVAL;
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int D>
int launch_update_d(const float* W_old, const float* T, const float* R, float* W_new, long rows, int N_edit,
                    int NEP64, hipStream_t st) {
  // 16-row tiles, 2 waves per SIMD: measured best at N_edit 16..128 on MI355X (32-row tiles: +4-9 %)
  if (N_edit <= 128) return launch_update_s<D, 1, 2>(W_old, T, R, W_new, rows, N_edit, NEP64, st);
  return launch_update_v<D, 4, 2>(W_old, T, R, W_new, rows, N_edit, NEP64, st);   // 129 <= N_edit <= 256: ring-buffered form
}

}  // namespace

int lr_rider_max_n() {
  static const int cap = getenv("UCE_RIDER_MAX_N") ? atoi(getenv("UCE_RIDER_MAX_N")) : 64 * GP_MAXB;
  return cap < 64 * GP_MAXB ? cap : 64 * GP_MAXB;
}

bool lowrank_split_supported(int d, int N_edit) {
  return (d == 768 || d == 1024 || d == 2048) && N_edit >= 1 && N_edit <= 256;
}

// X = Dm with Csub == nullptr, or X = G with Csub = C_e (D_e = G - C_e formed on the fly).  With `h`
// and a dual system that is a single 64-block (N <= 64) the launch also builds and factors that
// system (block 0): K = lamb S^-1 + C C^T -> h->Lmat (ld 64), h->Linv, h->status.
int launch_lr_project(const float* W_old, const float* X, const float* Csub, float* T, long rows, int d,
                      int N_edit, hipStream_t st, uce_ctx* h, const float* C, const float* s, int N, float lamb, float* R) {
  const int NEP64 = (N_edit + 63) / 64 * 64;
  GramPotrfJob job{};
  if (h) {
    const int nb = (N + 63) / 64;
    if (nb < 1 || nb > GP_MAXB) return UCE_EINVAL;
    job = GramPotrfJob{C, s, N, lamb, h->slabs, h->ticket, h->Lmat, h->Linv, h->status, h->M, nb, R, N_edit, ++h->seq};
  }
  if (d == 768) return launch_project_d<768>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st);
  if (d == 1024) return launch_project_d<1024>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st);
  if (d == 2048) return launch_project<2048, 5>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st);
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
