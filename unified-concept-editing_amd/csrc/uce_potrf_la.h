// The blocked f64 Cholesky as ONE persistent launch with look-ahead (uce_solve.hip: k_potrf_la), as a body that another
// launch can host in its first workgroups (uce_lowrank2.hip: k_lr_project_la).
#pragma once
#include "uce_common.h"
#include "uce_h2split.h"
#include "uce_gram_tile.h"
#include "uce_potrf64.h"

namespace {

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
#else
#define LADBG(k, slot) do { } while (0)
__device__ __forceinline__ void g_la_stamp(int) {}
#endif

// (PotrfLaJob: uce_common.h)

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
  // the Schur complement of a diagonal tile is symmetric and only its lower 16 x 16 tiles are read by the elimination
  // (10 of 16): per SIMD 3, 3, 2, 2 tiles instead of 4 each - wave w takes a pair of row tiles of one column tile, waves
  // 4 and 5 a single diagonal tile, waves 6 and 7 nothing.  nm = row tiles of this wave (prod_n / each_n).
  int nm = 2;
  __device__ __forceinline__ void use_schur_map() {
    const int w = threadIdx.x >> 6;
    constexpr int WR[8] = {0, 32, 32, 32, 16, 48, 0, 0}, WC[8] = {0, 0, 16, 32, 16, 48, 0, 0}, NM[8] = {2, 2, 2, 2, 1, 1, 0, 0};
    wr = WR[w];
    wc8 = WC[w];
    nm = NM[w];
  }
  __device__ __forceinline__ void prod_n(double4_t (&a2)[2], const double (*P)[LD], const double (*Q)[LD], double sign) const {
    if (nm == 2) prod(a2, P, Q, sign);
    else if (nm == 1) {
      const int r = lane & 15, kk = lane >> 4;
#pragma unroll 4
      for (int kb = 0; kb < 16; ++kb) {
        const int t = kb * 4 + kk;
        a2[0] = mfma_f64(sign * P[wr + r][t], Q[wc8 + r][t], a2[0]);
      }
    }
  }
  template <typename F>
  __device__ __forceinline__ void each_n(F f) const {
    const int oc = wc8 + (lane & 15), orq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 2; ++m)
      if (m < nm)
#pragma unroll
        for (int r = 0; r < 4; ++r) f(m, r, wr + m * 16 + orq + 4 * r, oc);
  }
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

// The roles of the launch, by block index `bid` of `nblocks` (k_potrf_la: the whole grid; k_lr_project_la, uce_lowrank2.hip: the
// FIRST nblocks workgroups of a projection launch - the dual system's factorisation beside the projection GEMM).
// smem_raw: POTRF_LA_SMEM bytes, 16-byte aligned; 512 threads.
__device__ __forceinline__ void potrf_la_body(const PotrfLaJob& j, const int bid, const int nblocks, unsigned char* smem_raw) {
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
  LaWave lws;
  lws.use_schur_map();
  auto finish = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned s_last;
    if (tid == 0) s_last = __hip_atomic_fetch_add(fDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nblocks - (unsigned)j.sp.blocks - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last)
      for (int e = tid; e < nflags; e += 512) __hip_atomic_store(j.flags + e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  if (j.sp.blocks && bid >= nblocks - j.sp.blocks) {
    // ------------------------------------------------------------------ riders: not part of the factorisation (nobody waits
    // for them, they wait for nobody, they do not count in fDone); the highest block indices, so they are placed after
    // every workgroup of the factorisation
    const int rb = bid - (nblocks - j.sp.blocks);
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
  if (bid == 0) {
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
        lws.each_n([&](int m, int r, int row, int col) { sacc[m][r] = S[row][col]; });
        lws.prod_n(sacc, A, A, -1.0);                              // Schur complement of the diagonal tile (its lower tiles)
        __syncthreads();
        lws.each_n([&](int m, int r, int row, int col) { S[row][col] = sacc[m][r]; });
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

  if (bid > ntiles) {
    // ------------------------------------------------------------------ a block (i, k) of L^-1, i > k
    //   W_ik = -L_ii^-1 sum_{j = k .. i-1} L_ij W_jk ,  W_kk = L_kk^-1
    // - what the GEMM-shaped solve (uce_trinv.hip) otherwise builds by recursive doubling in 2 log2(nb) launches AFTER
    // the factorisation.  These workgroups come last in the grid (each waits only for lower-numbered ones: the L tiles,
    // the walker, the W blocks above it in its column) and run beside the factorisation: block (i, k) is complete ~3 us
    // after the walker hands out L_ii^-1, the whole inverse a few microseconds after the last factor.
    int ti, tk;
    la_tile_of_block(bid - ntiles, nb, ti, tk);
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
  la_tile_of_block(bid, nb, ti, tk);
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


}  // namespace
