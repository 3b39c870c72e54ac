// 64x64 f64 Cholesky + inverse building block, shared by uce_solve.hip (k_potrf_first,
// k_potrf_step) and uce_lowrank2.hip (the launch that runs it beside the projection GEMM).
#pragma once
#include "uce_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Register-tiled Cholesky + inverse of one 64x64 SPD block by 256 threads.
// Thread (ti, tj) = (tid >> 4, tid & 15) owns the 4x4 sub-block rows 4ti.., cols 4tj.. of the
// matrix (a) and of X (x, starts as I, ends as L^-1).  Right-looking block elimination, TWO pivots
// per step (32 steps, one barrier each): at step (k, k+1) the owners publish
// columns k, k+1 of the current Schur complement and rows k, k+1 of X into LDS lines; everybody
// reads the 2x2 pivot block P, inverts it, and applies the rank-2 update
//     a_ij -= [a_ik a_i,k+1] P^-1 [a_jk a_j,k+1]^T ,   x_ij -= [a_ik a_i,k+1] P^-1 [x_kj x_k+1,j]^T .
// The published lines are final as published (later register updates of eliminated rows/columns
// are harmless garbage that is never read), so nothing is masked.  L and L^-1 are assembled from
// the lines in one pass at the end: an even column is scaled by 1/sqrt(p00); an odd one first
// gets the pivot-k elimination it skipped:  (c1 - c0 p01/p00) / sqrt(p11 - p01^2/p00).
// ---------------------------------------------------------------------------------------------
struct Potrf64Scratch {
  double col[64][64];   // col[k][i]: column k of the Schur complement when it was published
  double row[64][64];   // row[k][j]: row k of the partial inverse when it was published
  double rs[64];        // 1/sqrt(effective pivot k)
  double g[64];         // odd k: p01/p00 of its pair
};

// 1 / sqrt(v): hardware estimate + two Newton steps (the sqrt + divide sequence of `1.0 / sqrt(v)` is ~10x the code)
static __device__ __forceinline__ double rsqrt_f64(double v) {
  double y = __builtin_amdgcn_rsq(v);
  y = y * fma(-0.5 * v * y, y, 1.5);
  y = y * fma(-0.5 * v * y, y, 1.5);
  return y;
}

static __device__ __forceinline__ double rcp_f64(double v) {
  double r = __builtin_amdgcn_rcp(v);
  r = fma(fma(-v, r, 1.0), r, r);
  r = fma(fma(-v, r, 1.0), r, r);
  return r;
}

// One pair of pivots K = 4*kb + KR, K + 1 (KR in {0, 2} is static so register indices are static;
// kb is a loop variable: the 32 steps are a 16-trip loop of two bodies, ~3 KB of code.  A fully
// unrolled version of this loop is instruction-fetch bound: 50 KB of run-once straight-line code
// took 31-38 us regardless of how many barriers it contained).
template <int KR>
__device__ __forceinline__ void potrf64_pair(double (&a)[4][4], double (&x)[4][4], Potrf64Scratch* sc,
                                             int ti, int tj, int kb) {
  const int K = 4 * kb + KR;
  if (tj == kb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc->col[K][4 * ti + r] = a[r][KR];
      sc->col[K + 1][4 * ti + r] = a[r][KR + 1];
    }
  }
  if (ti == kb) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sc->row[K][4 * tj + c] = x[KR][c];
      sc->row[K + 1][4 * tj + c] = x[KR + 1][c];
    }
  }
  __syncthreads();
  const double p00 = sc->col[K][K], p01 = sc->col[K][K + 1], p11 = sc->col[K + 1][K + 1];
  const double idet = rcp_f64(fma(p00, p11, -p01 * p01));
  const double q00 = p11 * idet, q01 = -p01 * idet, q11 = p00 * idet;
  double u[4], v[4], c0[4], c1[4], x0[4], x1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double a0 = sc->col[K][4 * ti + r], a1 = sc->col[K + 1][4 * ti + r];
    u[r] = fma(a0, q00, a1 * q01);
    v[r] = fma(a0, q01, a1 * q11);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    c0[c] = sc->col[K][4 * tj + c];
    c1[c] = sc->col[K + 1][4 * tj + c];
    x0[c] = sc->row[K][4 * tj + c];
    x1[c] = sc->row[K + 1][4 * tj + c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a[r][c] = fma(-u[r], c0[c], fma(-v[r], c1[c], a[r][c]));
      x[r][c] = fma(-u[r], x0[c], fma(-v[r], x1[c], x[r][c]));
    }
}

// a[][] holds this thread's 4x4 sub-block of the SPD tile on entry (lower triangle is what
// matters); on exit a = sub-block of L (zero above the diagonal), x = sub-block of L^-1.
__device__ __forceinline__ void potrf64_reg(double (&a)[4][4], double (&x)[4][4], Potrf64Scratch* sc,
                                            int tid, int* status, int col_base) {
  const int ti = tid >> 4, tj = tid & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) x[r][c] = (4 * ti + r == 4 * tj + c) ? 1.0 : 0.0;
#pragma unroll 1
  for (int kb = 0; kb < 16; ++kb) {
    potrf64_pair<0>(a, x, sc, ti, tj, kb);
    potrf64_pair<2>(a, x, sc, ti, tj, kb);
  }
  __syncthreads();
  if (tid < 32) {
    const int k = 2 * tid;
    const double p00 = sc->col[k][k], p01 = sc->col[k][k + 1], p11 = sc->col[k + 1][k + 1];
    const double g = p01 / p00;
    const double p11e = fma(-g, p01, p11);           // pivot k+1 after eliminating pivot k
    // first non-positive (or NaN) pivot wins; everything after it is garbage anyway
    const unsigned long long bad0 = __ballot(!(p00 > 0.0)), bad1 = __ballot(!(p11e > 0.0));
    if ((bad0 | bad1) && tid == 0) {
      const int f0 = bad0 ? 2 * __builtin_ctzll(bad0) : 128, f1 = bad1 ? 2 * __builtin_ctzll(bad1) + 1 : 128;
      atomicCAS(status, 0, col_base + (f0 < f1 ? f0 : f1) + 1);
    }
    sc->rs[k] = 1.0 / sqrt(p00);
    sc->rs[k + 1] = 1.0 / sqrt(p11e);
    sc->g[k] = 0.0;
    sc->g[k + 1] = g;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = 4 * ti + r, col = 4 * tj + c;
      double lv = 0.0, xv = 0.0;
      if (col <= row) {
        lv = sc->col[col][row];
        if (col & 1) lv = fma(-sc->g[col], sc->col[col - 1][row], lv);
        lv *= sc->rs[col];
        xv = sc->row[row][col];
        if (row & 1) xv = fma(-sc->g[row], sc->row[row - 1][col], xv);
        xv *= sc->rs[row];
      }
      a[r][c] = lv;
      x[r][c] = xv;
    }
}

// Factors diagonal block 0.  With nsplit > 1 the block is first summed from the split-K slabs of
// the Gram kernel (single-block systems skip the separate reduction launch).
// ---------------------------------------------------------------------------------------------
// 8-wave variant (512 threads): waves 0-3 own the matrix tiles (a), waves 4-7 own the tiles of X.
// Same block elimination, same LDS lines, same single barrier per pivot pair; each thread issues
// half the f64 FMAs and half the LDS traffic of the 4-wave version.  role = 0 (a) / 1 (x); `t` holds
// this thread's 4x4 tile of its matrix on entry (role 1: ignored, X starts as I) and of L resp.
// L^-1 on exit.
// ---------------------------------------------------------------------------------------------
template <int KR, int ROLE>
__device__ __forceinline__ void potrf64_pair8(double (&t)[4][4], Potrf64Scratch* sc, int ti, int tj, int kb) {
  const int K = 4 * kb + KR;
  if (ROLE == 0) {
    if (tj == kb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc->col[K][4 * ti + r] = t[r][KR];
        sc->col[K + 1][4 * ti + r] = t[r][KR + 1];
      }
    }
  } else {
    if (ti == kb) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sc->row[K][4 * tj + c] = t[KR][c];
        sc->row[K + 1][4 * tj + c] = t[KR + 1][c];
      }
    }
  }
  __syncthreads();
  const double p00 = sc->col[K][K], p01 = sc->col[K][K + 1], p11 = sc->col[K + 1][K + 1];
  const double idet = rcp_f64(fma(p00, p11, -p01 * p01));
  const double q00 = p11 * idet, q01 = -p01 * idet, q11 = p00 * idet;
  double u[4], v[4], y0[4], y1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double a0 = sc->col[K][4 * ti + r], a1 = sc->col[K + 1][4 * ti + r];
    u[r] = fma(a0, q00, a1 * q01);
    v[r] = fma(a0, q01, a1 * q11);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    y0[c] = (ROLE == 0) ? sc->col[K][4 * tj + c] : sc->row[K][4 * tj + c];
    y1[c] = (ROLE == 0) ? sc->col[K + 1][4 * tj + c] : sc->row[K + 1][4 * tj + c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) t[r][c] = fma(-u[r], y0[c], fma(-v[r], y1[c], t[r][c]));
}

// All 512 threads call this (tid = 0..511).  role 0 threads pass their tile of the SPD block.
// `npiv`: the leading npiv rows / columns are the real system, the rest of the block is the identity padding of a
// system whose size is not a multiple of 64 (K_pad = diag(K, I)): its pivots are 1 and touch nothing, so the
// elimination stops after ceil(npiv / 4) of the 16 iterations (a 5-concept edit: 2 iterations instead of 16).
template <int ROLE>
__device__ __forceinline__ void potrf64_reg8(double (&t)[4][4], Potrf64Scratch* sc, int tid256, int* status,
                                             int col_base, int npiv = 64) {
  const int ti = tid256 >> 4, tj = tid256 & 15;
  const int nkb = npiv >= 64 ? 16 : ((npiv + 3) >> 2), kstop = 4 * nkb;
  if (ROLE == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) t[r][c] = (4 * ti + r == 4 * tj + c) ? 1.0 : 0.0;
  }
  if (nkb == 16) {                           // the full block keeps its constant trip count (the compiler pipelines it)
#pragma unroll 1
    for (int kb = 0; kb < 16; ++kb) {
      potrf64_pair8<0, ROLE>(t, sc, ti, tj, kb);
      potrf64_pair8<2, ROLE>(t, sc, ti, tj, kb);
    }
  } else {
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
      potrf64_pair8<0, ROLE>(t, sc, ti, tj, kb);
      potrf64_pair8<2, ROLE>(t, sc, ti, tj, kb);
    }
  }
  __syncthreads();
  if (ROLE == 0 && tid256 < 32) {
    const int k = 2 * tid256;
    const bool live = k < kstop;
    const double p00 = live ? sc->col[k][k] : 1.0, p01 = live ? sc->col[k][k + 1] : 0.0, p11 = live ? sc->col[k + 1][k + 1] : 1.0;
    const double g = p01 * rcp_f64(p00);
    const double p11e = fma(-g, p01, p11);
    const unsigned long long bad0 = __ballot(!(p00 > 0.0)), bad1 = __ballot(!(p11e > 0.0));
    if ((bad0 | bad1) && tid256 == 0) {
      const int f0 = bad0 ? 2 * __builtin_ctzll(bad0) : 128, f1 = bad1 ? 2 * __builtin_ctzll(bad1) + 1 : 128;
      atomicCAS(status, 0, col_base + (f0 < f1 ? f0 : f1) + 1);
    }
    sc->rs[k] = rsqrt_f64(p00);          // (a non-positive pivot gives NaN / inf here; it is reported through `status`)
    sc->rs[k + 1] = rsqrt_f64(p11e);
    sc->g[k] = 0.0;
    sc->g[k + 1] = g;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = 4 * ti + r, col = 4 * tj + c;
      double val = 0.0;
      if (col <= row) {
        if (ROLE == 0) {
          if (col >= kstop) {
            val = (row == col) ? 1.0 : 0.0;                  // identity padding: never published, L = I there
          } else {
            val = sc->col[col][row];
            if (col & 1) val = fma(-sc->g[col], sc->col[col - 1][row], val);
            val *= sc->rs[col];
          }
        } else {
          if (row >= kstop) {
            val = (row == col) ? 1.0 : 0.0;
          } else {
            val = sc->row[row][col];
            if (row & 1) val = fma(-sc->g[row], sc->row[row - 1][col], val);
            val *= sc->rs[row];
          }
        }
      }
      t[r][c] = val;
    }
}

// The 64 x 64 factor the callers use.  (Round 2 also built the rank-4 form with the trailing update on
// v_mfma_f64_16x16x4_f64 - K = 4 is exactly one MFMA per 16 x 16 tile, 16 steps instead of 32, pivot-block inverse by
// 2 x 2 blocks, computed either by every wave after the barrier or by its owner wave one step ahead: parity green,
// 20.5 and 22 us per 64 x 64 block against 17.4 us here (git history: "experiment: MFMA rank-4 form").  A step is
// bound by its dependent chain - barrier, LDS broadcast, reciprocal(s), ~10 dependent f64 operations - not by the
// update arithmetic the MFMA removes, and a 4 x 4 inverse is a longer chain than two 2 x 2 ones.)
#define UCE_POTRF64 potrf64_reg8

// (A four-pivots-per-barrier VALU variant was tried and is SLOWER on MI355X - ~20 us vs ~14 us for the 64x64 factor +
// inverse: the step time is set by the f64 VALU instruction count (f64 FMA issues at half rate on gfx950, ~8
// cycles per wave instruction), not by the barrier / LDS / reciprocal latencies, and the rank-4 form needs ~170
// f64 operations per thread per step (4x4 block inverse, W = C Q, update) against ~50 per pair step.  A faster
// factor has to move the rank-k update onto v_mfma_f64_16x16x4_f64 with the accumulators in D layout.)

// 512-thread version of potrf_first_body (waves 0-3: matrix tiles, waves 4-7: tiles of L^-1)
static __device__ __forceinline__ void potrf_first_body8(const double* __restrict__ M, int n, int nsplit,
                                                         size_t slab_stride, double* __restrict__ Lmat,
                                                         double* __restrict__ Linv, int* status,
                                                         Potrf64Scratch* sc, int n_valid = 1 << 30) {
  const int npiv0 = n_valid < 64 ? n_valid : 64;
  const int tid = threadIdx.x, half = tid >> 8, t256 = tid & 255;
  const int ti = t256 >> 4, tj = t256 & 15;
  double tt[4][4];
  if (half == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t off = (size_t)(4 * ti + r) * n + 4 * tj + c;
        double v = M[off];
        for (int sp = 1; sp < nsplit; ++sp) v += M[(size_t)sp * slab_stride + off];   // index order
        tt[r][c] = v;
      }
    UCE_POTRF64<0>(tt, sc, t256, status, 0, npiv0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Lmat[(size_t)(4 * ti + r) * n + 4 * tj + c] = tt[r][c];
  } else {
    UCE_POTRF64<1>(tt, sc, t256, status, 0, npiv0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Linv[(4 * ti + r) * 64 + 4 * tj + c] = tt[r][c];
  }
}

static __device__ __forceinline__ void potrf_first_body(const double* __restrict__ M, int n, int nsplit,
                                                        size_t slab_stride, double* __restrict__ Lmat,
                                                        double* __restrict__ Linv, int* status,
                                                        Potrf64Scratch* sc) {
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  double a[4][4], x[4][4];
  typedef double double2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[r][c] = 0.0;
  // slabs summed in index order (bit-repeatable); 2 slabs = 16 independent 16-byte loads per batch
  for (int sp = 0; sp < nsplit; sp += 2) {
    double2_t v[2][4][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int sq = (sp + q < nsplit) ? sp + q : sp;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          v[q][r][hh] = *(const double2_t*)(M + (size_t)sq * slab_stride + (size_t)(4 * ti + r) * n + 4 * tj + 2 * hh);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (sp + q < nsplit) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            a[r][2 * hh] += v[q][r][hh][0];
            a[r][2 * hh + 1] += v[q][r][hh][1];
          }
      }
    }
  }
  potrf64_reg(a, x, sc, tid, status, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Lmat[(size_t)(4 * ti + r) * n + 4 * tj + c] = a[r][c];
      Linv[(4 * ti + r) * 64 + 4 * tj + c] = x[r][c];
    }
}



// ---------------------------------------------------------------------------------------------
// One trailing tile (i, k), j < k <= i, of step j of the blocked right-looking Cholesky (block 64), by
// one 512-thread workgroup:  P_i = M_ij L_jj^-T (= L_ij),  P_k = M_kj L_jj^-T,  M_ik -= P_i P_k^T;
// tile (i, j+1) publishes L_ij; tile (j+1, j+1) then factors itself (-> L, L^-1 of block j+1).
// Called once per workgroup by k_potrf_step (uce_solve.hip) and in a loop by the last rider block of
// the projection launch (uce_lowrank2.hip).  smem_raw: POTRF_STEP_SMEM bytes.
// ---------------------------------------------------------------------------------------------
constexpr int LD = 66;  // row stride (doubles) of the 64x64 LDS tiles: conflict-free ds_read_b64
constexpr size_t POTRF_STEP_SMEM = 3 * 64 * LD * sizeof(double);

// one wave's 32x32 quadrant of  acc += sign * P[rows] * Q[cols]^T  (both tiles row-major in LDS,
// contraction index contiguous)
__device__ __forceinline__ void quad_nt(double4_t (&acc)[2][2], const double (*P)[LD],
                                        const double (*Q)[LD], int row0, int col0, int lane,
                                        double sign, int kb0 = 0, int kb1 = 16) {
  const int r = lane & 15, kk = lane >> 4;
#pragma unroll 4
  for (int kb = kb0; kb < kb1; ++kb) {
    const int t = kb * 4 + kk;
    const double a0 = sign * P[row0 + r][t], a1 = sign * P[row0 + 16 + r][t];
    const double b0 = Q[col0 + r][t], b1 = Q[col0 + 16 + r][t];
    acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
    acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
    acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
    acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
  }
}

__device__ __forceinline__ void quad_zero(double4_t (&acc)[2][2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};
}

// accumulator quadrant <-> memory (D layout of v_mfma_f64_16x16x4: row = (lane>>4) + 4r, col = lane&15)
template <typename F>
__device__ __forceinline__ void quad_foreach(int row0, int col0, int lane, F f) {
  const int c = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) f(m, n, r, row0 + m * 16 + rq + 4 * r, col0 + n * 16 + c);
}

static __device__ __forceinline__ void potrf_step_tile(double* __restrict__ M, int n, int j, int i, int k,
                                                       double* __restrict__ Lmat, double* __restrict__ Linv,
                                                       int* status, unsigned char* smem_raw, int n_valid = 1 << 30) {
  double (*Li)[LD] = (double (*)[LD])smem_raw;                       // L_jj^-1
  double (*Mi)[LD] = (double (*)[LD])(smem_raw + 64 * LD * 8);       // M_ij  -> P_i
  double (*Mk)[LD] = (double (*)[LD])(smem_raw + 2 * 64 * LD * 8);   // M_kj  -> P_k

  // 8 waves: quadrant = w & 3, half = w >> 2.  Half 0 forms P_i while half 1 forms P_k; half 0 does the
  // tile update; the diagonal tile is then factored by all 8 waves (potrf64_reg8).
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = w >> 2, wq = w & 3;
  const int wr = (wq >> 1) * 32, wc = (wq & 1) * 32;
  const bool diag = (i == k);

  const double* Linv_j = Linv + (size_t)j * 64 * 64;
  for (int e = tid; e < 64 * 64; e += 512) {
    const int r = e >> 6, c = e & 63;
    Li[r][c] = Linv_j[e];
    Mi[r][c] = M[(size_t)(i * 64 + r) * n + j * 64 + c];
    if (!diag) Mk[r][c] = M[(size_t)(k * 64 + r) * n + j * 64 + c];
  }
  __syncthreads();

  // P_i = M_ij L_jj^-T (half 0) ; P_k likewise (half 1)
  // (splitting the contraction of a diagonal tile over both halves was measured: the two extra barriers cost more
  //  than the 32 MFMAs per wave they save - the chain is latency-, not MFMA-bound)
  double4_t pp[2][2];
  quad_zero(pp);
  if (half == 0) quad_nt(pp, Mi, Li, wr, wc, lane, 1.0);
  else if (!diag) quad_nt(pp, Mk, Li, wr, wc, lane, 1.0);
  __syncthreads();
  if (half == 0)
    quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) { Mi[row][col] = pp[m][nn][r]; });
  else if (!diag)
    quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) { Mk[row][col] = pp[m][nn][r]; });
  __syncthreads();

  // tile (i, j+1) publishes L_ij
  if (k == j + 1) {
    for (int e = tid; e < 64 * 64; e += 512) {
      const int r = e >> 6, c = e & 63;
      Lmat[(size_t)(i * 64 + r) * n + j * 64 + c] = Mi[r][c];
    }
  }

  // M_ik -= P_i P_k^T   (half 0)
  const bool factor_here = diag && i == j + 1;
  double4_t acc[2][2];
  if (half == 0) {
    quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) {
      acc[m][nn][r] = M[(size_t)(i * 64 + row) * n + k * 64 + col];
    });
    quad_nt(acc, Mi, diag ? Mi : Mk, wr, wc, lane, -1.0);
    if (!factor_here) {
      quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) {
        M[(size_t)(i * 64 + row) * n + k * 64 + col] = acc[m][nn][r];
      });
    }
  }
  if (!factor_here) return;
  // the next diagonal block: factor it now (accumulators -> LDS tile -> 4x4 register sub-blocks)
  __syncthreads();
  if (half == 0)
    quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) { Li[row][col] = acc[m][nn][r]; });
  __syncthreads();
  Potrf64Scratch* sc = (Potrf64Scratch*)smem_raw;    // all three tile regions are dead once tt is in registers
  const int t256 = tid & 255;
  const int ti = t256 >> 4, tj = t256 & 15;
  double tt[4][4];
  if (half == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) tt[r][c] = Li[4 * ti + r][4 * tj + c];
  }
  __syncthreads();                                   // Li fully read before the scratch (which overlaps nothing of Li) is used
  const int npiv = (n_valid - i * 64) < 64 ? (n_valid - i * 64) : 64;
  if (half == 0) {
    UCE_POTRF64<0>(tt, sc, t256, status, i * 64, npiv);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Lmat[(size_t)(i * 64 + 4 * ti + r) * n + i * 64 + 4 * tj + c] = tt[r][c];
  } else {
    UCE_POTRF64<1>(tt, sc, t256, status, i * 64, npiv);
    double* Linv_n = Linv + (size_t)i * 64 * 64;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Linv_n[(4 * ti + r) * 64 + 4 * tj + c] = tt[r][c];
  }
}

}  // namespace
