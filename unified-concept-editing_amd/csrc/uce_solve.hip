// f64 blocked Cholesky + triangular solves for the UCE normal equations
// (reference: `torch.inverse(mat2.float())` at uce_sd_erase.py:82 / uce_sd_debias.py:140, an fp32
// LU inverse recomputed per module; here ONE f64 SPD solve shared by all modules).
//
// Structure (block size 64, everything f64 on v_mfma_f64_16x16x4_f64):
//   k_potrf_first : factor diagonal block 0            -> L_00, L_00^-1
//   k_potrf_step j: one workgroup per trailing tile (i,k), j < k <= i:
//                     P_i = M_ij L_jj^-T (= L_ij),  P_k = M_kj L_jj^-T,  M_ik -= P_i P_k^T
//                   tile (i, j+1) publishes L_ij; tile (j+1, j+1) then factors itself.
//   A chain of launches on one stream replaces grid barriers (a kernel boundary is ~1.5 us on
//   MI355X, cheaper than any software grid barrier, and needs no residency assumptions).
//   k_trisolve    : one workgroup per 16 right-hand-side columns: forward then backward
//                   substitution with the inverted diagonal blocks; Y lives in LDS (n <= 1024).
#include "uce_common.h"
#include "uce_h2split.h"
#include "uce_gram_tile.h"
#include "uce_potrf64.h"
#include <cstdlib>

namespace {

// (tri_decode: uce_gram_tile.h)
// Factors diagonal block 0 (body shared with the fused projection+factor launch, uce_potrf64.h).
__global__ __launch_bounds__(512) void k_potrf_first(const double* __restrict__ M, int n, int nsplit,
                                                     size_t slab_stride, double* __restrict__ Lmat,
                                                     double* __restrict__ Linv, int* status, int n_valid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // every factorisation starts here: reset the status word in this launch (a hipMemsetAsync costs a 4-5 us
  // fill kernel of its own); the first barrier inside the factor orders it before any failure report
  if (threadIdx.x == 0) *status = 0;
  potrf_first_body8(M, n, nsplit, slab_stride, Lmat, Linv, status, (Potrf64Scratch*)smem_raw, n_valid);
}

__global__ __launch_bounds__(512) void k_potrf_step(double* __restrict__ M, int n, int j,
                                                    double* __restrict__ Lmat,
                                                    double* __restrict__ Linv, int* status, int n_valid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int ta, tb;
  tri_decode(blockIdx.x, ta, tb);
  potrf_step_tile(M, n, j, j + 1 + ta, j + 1 + tb, Lmat, Linv, status, smem_raw, n_valid);
}

// ------------------------------------------------------------------------------------------
// The same factorisation as ONE persistent launch with look-ahead (systems of >= 3 diagonal blocks whose tiles are all
// co-resident: the d x d primal system of BASELINE config 3 is 12 blocks = 67 workgroups).  The launch chain above
// serialises, per 64-block, [kernel boundary | loads | panel product | trailing update | 64 x 64 factor]: 21.8 us, of which
// only the factor (11.3 us, a latency chain of 32 pivot pairs) is inherently sequential.  Here
//   * workgroup 0, the WALKER, goes down the diagonal: for block k it takes the tiles M_k,k-1 and M_kk that others have
//     already brought up to date through column k-2, forms L_k,k-1 = M_k,k-1 L_k-1,k-1^-T with the inverse it still holds
//     in LDS, publishes it, forms the Schur complement M_kk - L_k,k-1 L_k,k-1^T and factors it - nothing between two
//     factors but two 64^3 products and one tile load;
//   * one workgroup per off-diagonal tile (i, k), LEFT-looking: it subtracts L_ij L_kj^T for j = 0 .. k-1 as those panels
//     appear (all of it while the walker is busy with later... earlier diagonal blocks), then waits for L_kk^-1, forms
//     L_ik and publishes it.  The tile next to the diagonal, (i, i-1), also accumulates the diagonal tile M_ii (same L_ij
//     operand) and hands both to the walker instead of finishing itself.
// Hand-offs: payload with 16-byte write-through stores -> drained -> barrier -> relaxed flag; the reader polls the flag,
// passes a barrier and reads the payload with 16-byte L1-bypassing (sc1) buffer loads - no acquire: an agent-scope acquire
// invalidates the XCD's L2 under every workgroup on it, and with a few hundred hand-offs per launch those invalidates
// cost the factorisation more than the bypassing loads do (measured: +29 us on the walker's chain).  Workgroups are
// ordered so that nobody but the walker waits for a higher-numbered workgroup (column-major tiles): progress never depends
// on all workgroups being resident at once.  Every wait is bounded (status -1 instead of a hang).  The flags are zero
// between launches: the workgroup that finishes last clears them.
// ------------------------------------------------------------------------------------------
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

// -DUCE_CHAIN_DEBUG: wall-clock stamps (100 MHz) of the walker's phases per diagonal block, read back with
// uce_debug_read_la (tools/dbg_potrf.py); compiled out of the product library.
#ifdef UCE_CHAIN_DEBUG
__device__ unsigned long long g_la_dbg[32][8];
#define LADBG(k, slot) do { if ((threadIdx.x == 0 || threadIdx.x == 256) && (k) < 32) g_la_dbg[k][slot] = wall_clock64(); } while (0)
__device__ __forceinline__ void g_la_stamp(int k) { g_la_dbg[k][6] = wall_clock64(); }
extern "C" int uce_debug_read_la(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_la_dbg), sizeof(g_la_dbg));
}
#else
#define LADBG(k, slot) do { } while (0)
__device__ __forceinline__ void g_la_stamp(int) {}
#endif

struct PotrfLaJob {
  double* M;          // [n, n] system (lower tiles read; the tiles (i, i-1) and (i, i) are overwritten with their updates)
  int n, nb, n_valid;
  double* Lmat;       // [n, n]: off-diagonal blocks of L
  double* Linv;       // [nb][64][64]
  int* status;
  unsigned* flags;    // [nb * nb] panel (i, j) published | [nb] L_kk^-1 published | [nb] tiles (k, k-1), (k, k) handed over | [1] exits
                      // | [nb * nb] block (i, k) of L^-1 published
  double* Wi;         // [n, n]: off-diagonal blocks of L^-1 (null: not wanted)
  H2SplitJob sp;      // sp.blocks rider workgroups behind the factorisation's own: the f16 split of W_old for the dense apply that
                      // follows the solve (uce_apply_h2.hip), streamed on the CUs the factorisation leaves idle
  GramPrimalArgs bt;  // bt.C != null: the same riders then compute the d/64 x d/64 tiles of Bt = C_e^T S_e (G - C_e), the right-hand
                      // side of the solve that follows - nothing in this launch reads it
};

// One lane polls the flag (relaxed, agent scope), the workgroup passes a barrier; the payload is then read with L1-bypassing
// (sc1) loads - no acquire fence (see the header comment).  Bounded: reports instead of hanging.
__device__ __forceinline__ bool la_wait(const unsigned* flag, int* status) {
  bool ok = true;
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 25)) {                        // ~ seconds: report instead of hanging
        atomicCAS(status, 0, -1);
        ok = false;
        break;
      }
    }
  }
  __syncthreads();
  return ok;
}

__device__ __forceinline__ void la_post(unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 64 x 64 tile at G (row stride ld doubles) <-> LDS tile [64][LD]; 512 threads, 16 bytes per lane
__device__ __forceinline__ void la_load_tile(double (*T)[LD], const double* G, int ld) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int e = 2 * (threadIdx.x + 512 * p);
    *(double2_t*)&T[e >> 6][e & 63] = *(const double2_t*)(G + (size_t)(e >> 6) * ld + (e & 63));
  }
}
// L1-bypassing forms (16-byte buffer loads with sc1): a reader that uses them needs no acquire - an agent-scope acquire
// invalidates the XCD's L2 for every workgroup on it, and the W workgroups below would issue hundreds of them beside the
// factorisation they ride along with
__device__ __forceinline__ double2_t la_ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16 /* sc1 */));
}
__device__ __forceinline__ void la_load_tile_sc1(double (*T)[LD], const double* G, int ld, bool transpose) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)G, 0, (int)(64 * ld * sizeof(double)), 0x00020000);
  double2_t v[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int e = 2 * (threadIdx.x + 512 * p);
    v[p] = la_ld_sc1(r, (unsigned)(((e >> 6) * ld + (e & 63)) * sizeof(double)));
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int e = 2 * (threadIdx.x + 512 * p);
    if (transpose) {
      T[e & 63][e >> 6] = v[p][0];
      T[(e & 63) + 1][e >> 6] = v[p][1];
    } else {
      *(double2_t*)&T[e >> 6][e & 63] = v[p];
    }
  }
}
__device__ __forceinline__ void la_publish_tile(const double (*T)[LD], double* G, int ld) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)G, 0, (int)(64 * ld * sizeof(double)), 0x00020000);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int e = 2 * (threadIdx.x + 512 * p);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, *(const double2_t*)&T[e >> 6][e & 63]), r,
                                           (unsigned)(((e >> 6) * ld + (e & 63)) * sizeof(double)), 0, 16 /* sc1 */);
  }
}

// 8 waves: wave (wq = w & 3, half = w >> 2) owns rows wr .. wr + 31 x columns wc8 .. wc8 + 15 of a 64 x 64 product
// acc += sign * P Q^T (both LDS tiles row-major, contraction index contiguous)
struct LaWave {
  int wr, wc8, lane;
  __device__ __forceinline__ LaWave() {
    const int w = threadIdx.x >> 6, wq = w & 3;
    lane = threadIdx.x & 63;
    wr = (wq >> 1) * 32;
    wc8 = (wq & 1) * 32 + 16 * (w >> 2);
  }
  // kb_end < 16: Q is lower triangular (an inverted diagonal block) - column block c of P Q^T only contracts over
  // t < 16 (c + 1).  A 64^3 f64 product is MFMA-bound on one CU (2 us of the walker's critical path per product).
  __device__ __forceinline__ void prod(double4_t (&a2)[2], const double (*P)[LD], const double (*Q)[LD], double sign,
                                       int kb_end = 16) const {
    const int r = lane & 15, kk = lane >> 4;
#pragma unroll 4
    for (int kb = 0; kb < kb_end; ++kb) {
      const int t = kb * 4 + kk;
      const double b0 = Q[wc8 + r][t];
      a2[0] = mfma_f64(sign * P[wr + r][t], b0, a2[0]);
      a2[1] = mfma_f64(sign * P[wr + 16 + r][t], b0, a2[1]);
    }
  }
  // the triangular product's own tile map: the two waves of a SIMD (w, w + 4) take column blocks (0, 3) or (1, 2), so every
  // SIMD issues 20 of the 32 k-steps a full contraction would
  __device__ __forceinline__ void use_tri_map() {
    const int w = threadIdx.x >> 6, hf = w >> 2;
    wr = (w & 2) ? 32 : 0;
    wc8 = 16 * ((w & 1) ? (hf ? 2 : 1) : (hf ? 3 : 0));
  }
  __device__ __forceinline__ int tri_kb_end() const { return (wc8 + 16) / 4; }
  // accumulator <-> tile (D layout of v_mfma_f64_16x16x4: row = (lane >> 4) + 4 r, col = lane & 15)
  template <typename F>
  __device__ __forceinline__ void each(F f) const {
    const int oc = wc8 + (lane & 15), orq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) f(m, r, wr + m * 16 + orq + 4 * r, oc);
  }
};

constexpr size_t POTRF_LA_SMEM = sizeof(Potrf64Scratch) + 2 * 64 * LD * sizeof(double);

__device__ __forceinline__ void la_tile_of_block(int b, int nb, int& i, int& k) {
  // workgroups 1 ..: the off-diagonal tiles in column-major order (column k holds nb - 1 - k tiles)
  int t = b - 1;
  k = 0;
  while (t >= nb - 1 - k) { t -= nb - 1 - k; ++k; }
  i = k + 1 + t;
}

__global__ __launch_bounds__(512) void k_potrf_la(PotrfLaJob j) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int n = j.n, nb = j.nb;
  unsigned* fL = j.flags;                      // [nb * nb]
  unsigned* fInv = j.flags + nb * nb;          // [nb]
  unsigned* fSub = fInv + nb;                  // [nb]
  unsigned* fDone = fSub + nb;                 // [1]
  unsigned* fW = fDone + 1;                    // [nb * nb]
  const int nflags = 2 * nb * nb + 2 * nb + 1;
  const int ntiles = nb * (nb - 1) / 2;
  const LaWave lw;
  LaWave lwt;
  lwt.use_tri_map();
  auto finish = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned s_last;
    if (tid == 0) s_last = __hip_atomic_fetch_add(fDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - (unsigned)j.sp.blocks - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last)
      for (int e = tid; e < nflags; e += 512) __hip_atomic_store(j.flags + e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  if (j.sp.blocks && (int)blockIdx.x >= (int)gridDim.x - j.sp.blocks) {
    // ------------------------------------------------------------------ riders: not part of the factorisation (nobody waits
    // for them, they wait for nobody, they do not count in fDone); the highest block indices, so they are placed after
    // every workgroup of the factorisation
    const int rb = (int)blockIdx.x - ((int)gridDim.x - j.sp.blocks);
    if (j.sp.src)
      for (long row = (long)rb * 8 + (tid >> 6); row < j.sp.rows; row += (long)j.sp.blocks * 8)
        h2_split_row<false>(j.sp.src, j.sp.hi, j.sp.lo, j.sp.inv, row, j.sp.d, tid & 63);
    if (j.bt.C) {
      const int nbt = j.bt.d / 64;
      float (*Ss)[KC] = (float (*)[KC])(smem_raw + 2 * 2 * KC * 64 * sizeof(float));
      for (int t = rb; t < nbt * nbt; t += j.sp.blocks) {
        gram_primal_tile(j.bt, false, t / nbt, t % nbt, 0, smem_raw, Ss);
        __syncthreads();                                         // the next tile's staging overwrites the reduction tile
      }
    }
    return;
  }
  if (blockIdx.x == 0) {
    // ------------------------------------------------------------------ the walker
    Potrf64Scratch* sc = (Potrf64Scratch*)smem_raw;
    double (*S)[LD] = (double (*)[LD])smem_raw;                                              // aliases the scratch
    double (*A)[LD] = (double (*)[LD])(smem_raw + sizeof(Potrf64Scratch));                    // M_k,k-1 -> L_k,k-1
    double (*B)[LD] = (double (*)[LD])(smem_raw + sizeof(Potrf64Scratch) + 64 * LD * sizeof(double));   // L_k-1,k-1^-1
    const __amdgpu_buffer_rsrc_t linv_r =
        __builtin_amdgcn_make_buffer_rsrc((void*)j.Linv, 0, (int)(nb * 4096 * sizeof(double)), 0x00020000);
    if (tid == 0) *j.status = 0;
    la_load_tile(S, j.M, n);                                       // M_00 (written by the launch before this one)
    __syncthreads();
    // Nothing the walker publishes is drained on its own critical path: the flag of a payload is posted one phase later,
    // behind a barrier that every wave reaches with `s_waitcnt vmcnt(0)` long after the stores were issued.  And nothing it
    // consumes is fetched on it: waves 4-7, idle while waves 0-3 eliminate, poll for the next block's two tiles during
    // the last iterations of the factor and pull them in (one into the free A tile, one into 32 VGPRs).
    unsigned* pending = nullptr;                                   // flag of the L_k,k-1 tile whose stores are in flight
    auto post_now = [&](unsigned* f) {
      if (tid == 0 && f) __hip_atomic_store(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    double2_t ps[8];                                               // waves 4-7: pieces of M_k+1,k+1 (M_k+1,k goes straight into A)
    const int st = tid - 256;                                      // index among the 256 side threads
    for (int k = 0; k < nb; ++k) {
      LADBG(k, 0);
      if (k > 0) {
        __syncthreads();                                           // the factor is over: L_k-1,k-1^-1 is in LDS, the scratch is dead
        if (st >= 0) {
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const int e = 2 * (st + 256 * p);
            *(double2_t*)&S[e >> 6][e & 63] = ps[p];
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the stores of L_k-1,k-1^-1
        __syncthreads();
        post_now(fInv + (k - 1));
        LADBG(k, 1);
        double4_t pp[2] = {(double4_t){0.0, 0.0, 0.0, 0.0}, (double4_t){0.0, 0.0, 0.0, 0.0}};
        lwt.prod(pp, A, B, 1.0, lwt.tri_kb_end());                 // L_k,k-1 = M_k,k-1 L_k-1,k-1^-T (L^-1 is lower triangular)
        __syncthreads();
        LADBG(k, 2);
        lwt.each([&](int m, int r, int row, int col) { A[row][col] = pp[m][r]; });
        __syncthreads();
        la_publish_tile(A, j.Lmat + (size_t)k * 64 * n + (size_t)(k - 1) * 64, n);
        pending = fL + k * nb + (k - 1);
        double4_t sacc[2];
        lw.each([&](int m, int r, int row, int col) { sacc[m][r] = S[row][col]; });
        lw.prod(sacc, A, A, -1.0);                                 // Schur complement of the diagonal tile
        __syncthreads();
        lw.each([&](int m, int r, int row, int col) { S[row][col] = sacc[m][r]; });
        __syncthreads();
      }
      LADBG(k, 3);
      const int npiv = (j.n_valid - k * 64) < 64 ? (j.n_valid - k * 64) : 64;
      const double* nextA = j.M + (size_t)(k + 1) * 64 * n + (size_t)k * 64;
      const double* nextS = j.M + (size_t)(k + 1) * 64 * n + (size_t)(k + 1) * 64;
      UCE_POTRF64([&](int row, int col, double (&v)[4]) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drains the publish of L_k,k-1 before the factor's first barrier
                    const pk_d2 a = *(const pk_d2*)&S[row][col], b = *(const pk_d2*)&S[row][col + 2];
                    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
                  },
                  [&](int row, int col, const double (&v)[4]) {
                    const unsigned off = (unsigned)((k * 4096 + row * 64 + col) * sizeof(double));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, (double2_t){v[0], v[1]}), linv_r, off, 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, (double2_t){v[2], v[3]}), linv_r, off + 16, 0, 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) B[row][col + e] = v[e];
                  },
                  sc, tid, j.status, k * 64, npiv, [&]() { post_now(pending); },
                  [&](int kb, int nkb) {
                    // waves 4-7, three iterations before the end of the factor (or at once for a short one): the tiles the
                    // sub-diagonal workgroup (k+1, k) has handed over
                    if (k + 1 >= nb || kb != (nkb > 3 ? nkb - 3 : 0)) return;
                    LADBG(k, 5);
                    if ((tid & 63) == 0) {
                      unsigned spins = 0;
                      while (__hip_atomic_load(fSub + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1u << 25)) {
                          atomicCAS(j.status, 0, -1);
                          break;
                        }
                      }
                    }
                    // M_k+1,k -> the A tile (L_k,k-1 left it when its publish was issued, before this factor began);
                    // M_k+1,k+1 stays in registers until the scratch it belongs in is dead.  L1-bypassing loads: no acquire
                    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)nextA, 0, (int)(64 * n * sizeof(double)), 0x00020000);
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)nextS, 0, (int)(64 * n * sizeof(double)), 0x00020000);
                    double2_t pa[8];
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                      const int e = 2 * (st + 256 * p);
                      pa[p] = la_ld_sc1(ra, (unsigned)(((e >> 6) * n + (e & 63)) * sizeof(double)));
                    }
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                      const int e = 2 * (st + 256 * p);
                      ps[p] = la_ld_sc1(rs, (unsigned)(((e >> 6) * n + (e & 63)) * sizeof(double)));
                    }
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                      const int e = 2 * (st + 256 * p);
                      *(double2_t*)&A[e >> 6][e & 63] = pa[p];
                    }
                    if (tid == 256 && k < 32) g_la_stamp(k);
                  });
      LADBG(k, 4);
    }
    la_post(fInv + (nb - 1));                                      // (nobody waits for it; kept for symmetry of the flag set)
    finish();
    return;
  }

  if ((int)blockIdx.x > ntiles) {
    // ------------------------------------------------------------------ a block (i, k) of L^-1, i > k
    //   W_ik = -L_ii^-1 sum_{j = k .. i-1} L_ij W_jk ,  W_kk = L_kk^-1
    // - what the GEMM-shaped solve (uce_trinv.hip) otherwise builds by recursive doubling in 2 log2(nb) launches AFTER
    // the factorisation.  These workgroups come last in the grid (each waits only for lower-numbered ones: the L tiles,
    // the walker, the W blocks above it in its column) and run beside the factorisation: block (i, k) is complete ~3 us
    // after the walker hands out L_ii^-1, the whole inverse a few microseconds after the last factor.
    int ti, tk;
    la_tile_of_block((int)blockIdx.x - ntiles, nb, ti, tk);
    double (*P)[LD] = (double (*)[LD])smem_raw;
    double (*Q)[LD] = P + 64;
    double4_t acc[2] = {(double4_t){0.0, 0.0, 0.0, 0.0}, (double4_t){0.0, 0.0, 0.0, 0.0}};
    for (int jj = tk; jj < ti; ++jj) {
      la_wait(fL + ti * nb + jj, j.status);
      la_wait(jj == tk ? fInv + tk : fW + jj * nb + tk, j.status);
      la_load_tile_sc1(P, j.Lmat + (size_t)ti * 64 * n + (size_t)jj * 64, n, false);
      if (jj == tk) la_load_tile_sc1(Q, j.Linv + (size_t)tk * 4096, 64, true);
      else la_load_tile_sc1(Q, j.Wi + (size_t)jj * 64 * n + (size_t)tk * 64, n, true);
      __syncthreads();
      lw.prod(acc, P, Q, 1.0);
      __syncthreads();
    }
    la_wait(fInv + ti, j.status);
    la_load_tile_sc1(P, j.Linv + (size_t)ti * 4096, 64, false);
    lw.each([&](int m, int r, int row, int col) { Q[col][row] = acc[m][r]; });       // transposed: the right operand again
    __syncthreads();
    double4_t out[2] = {(double4_t){0.0, 0.0, 0.0, 0.0}, (double4_t){0.0, 0.0, 0.0, 0.0}};
    lw.prod(out, P, Q, -1.0, (lw.wr + 32) / 4);                   // L_ii^-1 is lower triangular: rows wr .. wr+31 contract over t < wr + 32
    __syncthreads();
    lw.each([&](int m, int r, int row, int col) { P[row][col] = out[m][r]; });
    __syncthreads();
    la_publish_tile(P, j.Wi + (size_t)ti * 64 * n + (size_t)tk * 64, n);
    la_post(fW + ti * nb + tk);
    finish();
    return;
  }

  // -------------------------------------------------------------------- an off-diagonal tile (i, k)
  int ti, tk;
  la_tile_of_block((int)blockIdx.x, nb, ti, tk);
  double (*P)[LD] = (double (*)[LD])smem_raw;                      // L_ij, later this tile's accumulator
  double (*Q)[LD] = P + 64;                                        // L_kj, later L_kk^-1
  const bool sub = (ti == tk + 1);                                 // next to the diagonal: also carries M_ii
  double4_t acc[2], dacc[2];
  {
    const double* Mik = j.M + (size_t)ti * 64 * n + (size_t)tk * 64;
    lw.each([&](int m, int r, int row, int col) { acc[m][r] = Mik[(size_t)row * n + col]; });
    if (sub) {
      const double* Mii = j.M + (size_t)ti * 64 * n + (size_t)ti * 64;
      lw.each([&](int m, int r, int row, int col) { dacc[m][r] = Mii[(size_t)row * n + col]; });
    }
  }
  for (int jj = 0; jj < tk; ++jj) {
    la_wait(fL + ti * nb + jj, j.status);
    la_wait(fL + tk * nb + jj, j.status);
    la_load_tile_sc1(P, j.Lmat + (size_t)ti * 64 * n + (size_t)jj * 64, n, false);
    la_load_tile_sc1(Q, j.Lmat + (size_t)tk * 64 * n + (size_t)jj * 64, n, false);
    __syncthreads();
    lw.prod(acc, P, Q, -1.0);
    if (sub) lw.prod(dacc, P, P, -1.0);
    __syncthreads();
  }
  if (sub) {
    // hand both tiles, up to date through column k - 1, to the walker (it owns the last update and the factor)
    double* Mik = j.M + (size_t)ti * 64 * n + (size_t)tk * 64;
    double* Mii = j.M + (size_t)ti * 64 * n + (size_t)ti * 64;
    if (tk > 0) {                                                  // (column 0: the tiles in memory are already final)
      lw.each([&](int m, int r, int row, int col) { P[row][col] = acc[m][r]; Q[row][col] = dacc[m][r]; });
      __syncthreads();
      la_publish_tile(P, Mik, n);
      la_publish_tile(Q, Mii, n);
    }
    la_post(fSub + ti);
    finish();
    return;
  }
  la_wait(fInv + tk, j.status);
  la_load_tile_sc1(Q, j.Linv + (size_t)tk * 4096, 64, false);
  lw.each([&](int m, int r, int row, int col) { P[row][col] = acc[m][r]; });
  __syncthreads();
  double4_t pp[2] = {(double4_t){0.0, 0.0, 0.0, 0.0}, (double4_t){0.0, 0.0, 0.0, 0.0}};
  lwt.prod(pp, P, Q, 1.0, lwt.tri_kb_end());                       // L_ik = M_ik L_kk^-T
  __syncthreads();
  lwt.each([&](int m, int r, int row, int col) { P[row][col] = pp[m][r]; });
  __syncthreads();
  la_publish_tile(P, j.Lmat + (size_t)ti * 64 * n + (size_t)tk * 64, n);
  la_post(fL + ti * nb + tk);
  finish();
}

// ------------------------------------------------------------------------------------------
// triangular solves: X = L^-T L^-1 RHS for 16 columns per workgroup
// ------------------------------------------------------------------------------------------
template <bool RHS32>
__global__ __launch_bounds__(256) void k_trisolve(const double* __restrict__ Lmat,
                                                  const double* __restrict__ Linv, int n, int m,
                                                  const double* __restrict__ rhs64,
                                                  const float* __restrict__ rhs32, int rhs_rows,
                                                  float* __restrict__ out, int out_rows,
                                                  double* __restrict__ Yg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double (*tmp)[16] = (double (*)[16])smem_raw;          // [64][16]
  double* Ylds = (double*)(smem_raw + 64 * 16 * 8);      // [n][16] when it fits
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c0 = blockIdx.x * 16;
  const int c = lane & 15, kk = lane >> 4;
  const int nb = n / 64;
  // Y addressing: LDS [n][16] or global scratch [n][m]
  double* Y = Yg ? (Yg + c0) : Ylds;
  const int ldy = Yg ? m : 16;

  // ---------------- forward: Y_k = Linv_kk (RHS_k - sum_{j<k} L_kj Y_j)
  for (int kb = 0; kb < nb; ++kb) {
    const int r0 = kb * 64 + w * 16;  // this wave's 16 rows
    double4_t acc0, acc1 = (double4_t){0.0, 0.0, 0.0, 0.0}, acc2 = acc1, acc3 = acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + kk + 4 * r;
      double v;
      if (RHS32) v = (row < rhs_rows) ? (double)rhs32[(size_t)row * m + c0 + c] : 0.0;
      else v = rhs64[(size_t)row * m + c0 + c];
      acc0[r] = v;
    }
    const double* Lrow = Lmat + (size_t)(r0 + c) * n;  // A operand: row = r0 + (lane&15)
    // the diagonal-block fragments do not depend on the sum: fetch them first
    const double* Lid = Linv + (size_t)kb * 64 * 64 + (size_t)(w * 16 + c) * 64;
    double li_f[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) li_f[t] = Lid[t * 4 + kk];
    // L_kj fragments: 16-byte loads (lane (i, kk) takes columns 8u + 2kk, +1 of row i; the same k
    // permutation is applied to the Y side, which is free in LDS), two tiles in flight ahead of the
    // MFMAs (strided L2 reads take ~2k cycles here, a 16-MFMA tile only ~1k).
    typedef double double2_t __attribute__((ext_vector_type(2)));
    auto ld_tile = [&](int jb, double2_t (&f)[8]) {
#pragma unroll
      for (int u = 0; u < 8; ++u) f[u] = *(const double2_t*)(Lrow + jb * 64 + 8 * u + 2 * kk);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_tile = [&](int jb, const double2_t (&f)[8]) {
      double bf[16];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bf[2 * u] = Y[(size_t)(jb * 64 + 8 * u + 2 * kk) * ldy + c];
        bf[2 * u + 1] = Y[(size_t)(jb * 64 + 8 * u + 2 * kk + 1) * ldy + c];
      }
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        acc0 = mfma_f64(-f[u][0], bf[2 * u], acc0);
        acc1 = mfma_f64(-f[u][1], bf[2 * u + 1], acc1);
        acc2 = mfma_f64(-f[u + 1][0], bf[2 * u + 2], acc2);
        acc3 = mfma_f64(-f[u + 1][1], bf[2 * u + 3], acc3);
      }
    };
    {
      double2_t fA[8], fB[8], fC[8];
      if (kb > 0) ld_tile(0, fA);
      if (kb > 1) ld_tile(1, fB);
      int jb = 0;
      for (; jb + 2 < kb; jb += 3) {
        ld_tile(jb + 2, fC);
        mma_tile(jb, fA);
        ld_tile(jb + 3 < kb ? jb + 3 : jb + 2, fA);
        mma_tile(jb + 1, fB);
        ld_tile(jb + 4 < kb ? jb + 4 : jb + 2, fB);
        mma_tile(jb + 2, fC);
      }
      if (jb < kb) mma_tile(jb, fA);
      if (jb + 1 < kb) mma_tile(jb + 1, fB);
    }
    acc0 = (acc0 + acc1) + (acc2 + acc3);
#pragma unroll
    for (int r = 0; r < 4; ++r) tmp[w * 16 + kk + 4 * r][c] = acc0[r];
    __syncthreads();
    double4_t y0 = (double4_t){0.0, 0.0, 0.0, 0.0}, y1 = y0;
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      const int k0 = t * 4 + kk;
      y0 = mfma_f64(li_f[t], tmp[k0][c], y0);
      y1 = mfma_f64(li_f[t + 1], tmp[k0 + 4][c], y1);
    }
    y0 += y1;
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[(size_t)(r0 + kk + 4 * r) * ldy + c] = y0[r];
    __syncthreads();
  }

  // ---------------- backward: X_k = Linv_kk^T (Y_k - sum_{j>k} L_jk^T X_j)
  for (int kb = nb - 1; kb >= 0; --kb) {
    const int r0 = kb * 64 + w * 16;
    double4_t acc0, acc1 = (double4_t){0.0, 0.0, 0.0, 0.0}, acc2 = acc1, acc3 = acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] = Y[(size_t)(r0 + kk + 4 * r) * ldy + c];
    // diagonal-block fragments (transposed access) first, then the L_jk^T tiles one ahead
    const double* Lid = Linv + (size_t)kb * 64 * 64;
    double li_f[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) li_f[t] = Lid[(size_t)(t * 4 + kk) * 64 + w * 16 + c];
    // L_jk^T tiles (A[i][k] = L[k][r0 + i]: 16 consecutive rows of one column block per load
    // instruction, 128 B per row group), two tiles in flight ahead of the MFMAs
    auto ld_tile = [&](int jb, double (&f)[16]) {
#pragma unroll
      for (int t = 0; t < 16; ++t) f[t] = Lmat[(size_t)(jb * 64 + t * 4 + kk) * n + r0 + c];
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_tile = [&](int jb, const double (&f)[16]) {
      double bf[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) bf[t] = Y[(size_t)(jb * 64 + t * 4 + kk) * ldy + c];
#pragma unroll
      for (int t = 0; t < 16; t += 4) {
        acc0 = mfma_f64(-f[t], bf[t], acc0);
        acc1 = mfma_f64(-f[t + 1], bf[t + 1], acc1);
        acc2 = mfma_f64(-f[t + 2], bf[t + 2], acc2);
        acc3 = mfma_f64(-f[t + 3], bf[t + 3], acc3);
      }
    };
    {
      double fA[16], fB[16], fC[16];
      const int ntile = nb - 1 - kb;                  // tiles jb = nb-1 .. kb+1, index q -> jb = nb-1-q
      if (ntile > 0) ld_tile(nb - 1, fA);
      if (ntile > 1) ld_tile(nb - 2, fB);
      int q = 0;
      for (; q + 2 < ntile; q += 3) {
        ld_tile(nb - 1 - (q + 2), fC);
        mma_tile(nb - 1 - q, fA);
        ld_tile(nb - 1 - (q + 3 < ntile ? q + 3 : q + 2), fA);
        mma_tile(nb - 2 - q, fB);
        ld_tile(nb - 1 - (q + 4 < ntile ? q + 4 : q + 2), fB);
        mma_tile(nb - 3 - q, fC);
      }
      if (q < ntile) mma_tile(nb - 1 - q, fA);
      if (q + 1 < ntile) mma_tile(nb - 2 - q, fB);
    }
    acc0 = (acc0 + acc1) + (acc2 + acc3);
    __syncthreads();  // every wave has finished reading Y_k rows of this block before tmp reuse
#pragma unroll
    for (int r = 0; r < 4; ++r) tmp[w * 16 + kk + 4 * r][c] = acc0[r];
    __syncthreads();
    double4_t x0 = (double4_t){0.0, 0.0, 0.0, 0.0}, x1 = x0;
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      const int k0 = t * 4 + kk;
      // A[i][k] = Linv[k][w*16 + i]
      x0 = mfma_f64(li_f[t], tmp[k0][c], x0);
      x1 = mfma_f64(li_f[t + 1], tmp[k0 + 4][c], x1);
    }
    x0 += x1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + kk + 4 * r;
      Y[(size_t)row * ldy + c] = x0[r];
      if (row < out_rows) out[(size_t)row * m + c0 + c] = (float)x0[r];
    }
    __syncthreads();
  }
}

}  // namespace

// The walker is the one workgroup that waits for higher-numbered ones (the tiles next to the diagonal), so it needs the
// 1 + nb (nb - 1) / 2 factor workgroups of ITS launch resident; capped so that two launches from two handles / streams fit the
// 256 CUs side by side (2 x 121 at nb = 16: d <= 1024; larger systems take the launch chain).  The W workgroups behind them wait
// only for lower-numbered ones and may start late.
constexpr int POTRF_LA_MAX_NB = 16;

// the persistent launch is taken for this system, and `*own` workgroups are its own
static bool potrf_la_taken(const uce_ctx* h, int n, int nsplit, int* own) {
  const int nb = n / 64;
  if (!(nsplit == 1 && nb >= 3 && nb <= POTRF_LA_MAX_NB && h->sw.potrf_variant != 0 && h->la_flags)) return false;
  const bool with_inverse = h->sw.potrf_variant == 1 && h->Wi != nullptr;
  *own = 1 + nb * (nb - 1) / 2 * (with_inverse ? 2 : 1);
  return true;
}
// (h->sw.potrf_rider_cus, default 250: workgroups the launch may place at once - one per CU: this kernel's LDS -, a few CUs spare)
constexpr int POTRF_LA_MIN_RIDERS = 64;

// uce_edit asks before it hands the launch its rider jobs (the split of W_old, the Bt half of the Gram)
bool potrf_la_has_room(const uce_ctx* h, int n) {
  int own = 0;
  return potrf_la_taken(h, n, 1, &own) && h->sw.potrf_rider_cus - own >= POTRF_LA_MIN_RIDERS;
}

int launch_potrf_slabs(uce_ctx* h, double* M, int n, int nsplit, size_t slab_stride, hipStream_t st, int n_valid) {
  if (n_valid <= 0 || n_valid > n) n_valid = n;
  const int nb = n / 64;
  int own = 0;
  if (potrf_la_taken(h, n, nsplit, &own)) {
    // one persistent launch with look-ahead (k_potrf_la)
    static PerDeviceOnce la_once;
    if (const int tok = la_once.first()) {
      UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_potrf_la, hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTRF_LA_SMEM));
      la_once.commit(tok);
    }
    // variant 1 (default): the launch also forms the off-diagonal blocks of L^-1 (h->Wi) for the GEMM-shaped solve that
    // follows (uce_trinv.hip skips its merge launches); variant 2: factor only
    const bool with_inverse = h->sw.potrf_variant == 1 && h->Wi != nullptr;
    // uce_edit's primal path leaves the W_old of the dense apply here: the factorisation occupies `own` CUs for its whole
    // latency chain (d = 768: 133 for ~220 us), the rest of the chip streams the f16 split meanwhile (one workgroup per
    // CU: this kernel's LDS).  Fewer than 64 free CUs: the apply launches its own split pass.
    H2SplitJob sp{};
    GramPrimalArgs bt{};
    const int room = h->sw.potrf_rider_cus - own;
    if (room >= POTRF_LA_MIN_RIDERS) {
      if (h->h2_pending_src) {
        unsigned short* Ap;
        float *rs, *cb;
        const int rc = apply_h2_workspace(h, h->h2_pending_rows, h->h2_pending_d, &Ap, &rs, &cb);
        if (rc) return rc;
        sp = H2SplitJob{h->h2_pending_src, Ap, Ap + (size_t)h->h2_pending_rows * h->h2_pending_d, rs, h->h2_pending_rows, h->h2_pending_d, 0};
      }
      if (h->bt_pending.C) bt = h->bt_pending;
      if (sp.src || bt.C) sp.blocks = room;
    }
    const PotrfLaJob job{M, n, nb, n_valid, h->Lmat, h->Linv, h->status, h->la_flags, with_inverse ? h->Wi : nullptr, sp, bt};
    hipLaunchKernelGGL(k_potrf_la, dim3(own + sp.blocks), dim3(512), POTRF_LA_SMEM, st, job);
    UCE_LAUNCH_CHECK();
    h->wi_valid = with_inverse;
    if (sp.src) {
      h->h2_done_src = sp.src;
      h->h2_done_rows = sp.rows;
      h->h2_done_d = sp.d;
    }
    if (bt.C) h->bt_pending.C = nullptr;                           // taken (uce_edit launches Bt itself when this is still set)
    h->h2_pending_src = nullptr;
    return UCE_OK;
  }
  h->wi_valid = false;
  const size_t smem = 3 * 64 * LD * sizeof(double);
  const size_t smem_first = sizeof(Potrf64Scratch);
  static_assert(sizeof(Potrf64Scratch) <= POTRF_STEP_SMEM, "scratch must fit the three tile regions");
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_potrf_step, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_potrf_first, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem_first));
    attr_once.commit(tok);
  }
  hipLaunchKernelGGL(k_potrf_first, dim3(1), dim3(512), smem_first, st, (const double*)M, n, nsplit,
                     slab_stride, h->Lmat, h->Linv, h->status, n_valid);
  UCE_LAUNCH_CHECK();
  for (int j = 0; j + 1 < nb; ++j) {
    const int mt = nb - j - 1;
    const int tiles = mt * (mt + 1) / 2;
    hipLaunchKernelGGL(k_potrf_step, dim3(tiles), dim3(512), smem, st, M, n, j, h->Lmat, h->Linv,
                       h->status, n_valid);
    UCE_LAUNCH_CHECK();
  }
  return UCE_OK;
}

int launch_potrf(uce_ctx* h, double* M, int n, hipStream_t st, int n_valid) {
  return launch_potrf_slabs(h, M, n, 1, 0, st, n_valid);
}

int launch_trisolve(uce_ctx* h, int n, int m, const double* rhs64, const float* rhs32, int rhs_rows,
                    float* out, int out_rows, hipStream_t st, double* scratch) {
  // UCE_TRISOLVE_VARIANT=0 (read at uce_create) keeps the substitution kernel at every size (A/B measurements)
  if (h->sw.trisolve_variant && scratch && n >= 192 && m % 64 == 0)
    return launch_trisolve_inv(h, n, m, rhs64, rhs32, rhs_rows, out, out_rows, scratch, st);
  const bool use_lds = n <= 1024;
  const size_t smem = 64 * 16 * 8 + (use_lds ? (size_t)n * 16 * 8 : 0);
  double* Yg = use_lds ? nullptr : h->Yg;
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    const int cap = 64 * 16 * 8 + 1024 * 16 * 8;
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trisolve<true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trisolve<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    attr_once.commit(tok);
  }
  const dim3 grid(m / 16);
  if (rhs32)
    hipLaunchKernelGGL(k_trisolve<true>, grid, dim3(256), smem, st, (const double*)h->Lmat,
                       (const double*)h->Linv, n, m, rhs64, rhs32, rhs_rows, out, out_rows, Yg);
  else
    hipLaunchKernelGGL(k_trisolve<false>, grid, dim3(256), smem, st, (const double*)h->Lmat,
                       (const double*)h->Linv, n, m, rhs64, rhs32, rhs_rows, out, out_rows, Yg);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
