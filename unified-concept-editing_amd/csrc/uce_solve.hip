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
#include "uce_potrf64.h"

namespace {

constexpr int LD = 66;  // row stride (doubles) of the 64x64 LDS tiles: conflict-free ds_read_b64

__device__ __forceinline__ void tri_decode(int t, int& a, int& b) {
  a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((a + 1) * (a + 2) / 2 <= t) ++a;
  while (a * (a + 1) / 2 > t) --a;
  b = t - a * (a + 1) / 2;
}

// Factors diagonal block 0 (body shared with the fused projection+factor launch, uce_potrf64.h).
__global__ __launch_bounds__(512) void k_potrf_first(const double* __restrict__ M, int n, int nsplit,
                                                     size_t slab_stride, double* __restrict__ Lmat,
                                                     double* __restrict__ Linv, int* status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  potrf_first_body8(M, n, nsplit, slab_stride, Lmat, Linv, status, (Potrf64Scratch*)smem_raw);
}

// one wave's 32x32 quadrant of  acc += sign * P[rows] * Q[cols]^T  (both tiles row-major in LDS,
// contraction index contiguous)
__device__ __forceinline__ void quad_nt(double4_t (&acc)[2][2], const double (*P)[LD],
                                        const double (*Q)[LD], int row0, int col0, int lane,
                                        double sign) {
  const int r = lane & 15, kk = lane >> 4;
#pragma unroll 4
  for (int kb = 0; kb < 16; ++kb) {
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

__global__ __launch_bounds__(512) void k_potrf_step(double* __restrict__ M, int n, int j,
                                                    double* __restrict__ Lmat,
                                                    double* __restrict__ Linv, int* status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double (*Li)[LD] = (double (*)[LD])smem_raw;                       // L_jj^-1
  double (*Mi)[LD] = (double (*)[LD])(smem_raw + 64 * LD * 8);       // M_ij  -> P_i
  double (*Mk)[LD] = (double (*)[LD])(smem_raw + 2 * 64 * LD * 8);   // M_kj  -> P_k

  // 8 waves: quadrant = w & 3, half = w >> 2.  Half 0 forms P_i while half 1 forms P_k; half 0 does the
  // tile update; the diagonal tile is then factored by all 8 waves (potrf64_reg8).
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = w >> 2, wq = w & 3;
  const int wr = (wq >> 1) * 32, wc = (wq & 1) * 32;
  int ta, tb;
  tri_decode(blockIdx.x, ta, tb);
  const int i = j + 1 + ta, k = j + 1 + tb;
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
  Potrf64Scratch* sc = (Potrf64Scratch*)&Mi[0][0];   // P_i / P_k regions (66 KB) are dead now
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
  if (half == 0) {
    potrf64_reg8<0>(tt, sc, t256, status, i * 64);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Lmat[(size_t)(i * 64 + 4 * ti + r) * n + i * 64 + 4 * tj + c] = tt[r][c];
  } else {
    potrf64_reg8<1>(tt, sc, t256, status, i * 64);
    double* Linv_n = Linv + (size_t)i * 64 * 64;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Linv_n[(4 * ti + r) * 64 + 4 * tj + c] = tt[r][c];
  }
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

int launch_potrf_slabs(uce_ctx* h, double* M, int n, int nsplit, size_t slab_stride, hipStream_t st) {
  const int nb = n / 64;
  const size_t smem = 3 * 64 * LD * sizeof(double);
  const size_t smem_first = sizeof(Potrf64Scratch);
  static_assert(sizeof(Potrf64Scratch) <= 2 * 64 * LD * sizeof(double), "scratch must fit the P_i/P_k regions");
  static bool attr_set = false;
  if (!attr_set) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_potrf_step, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_potrf_first, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem_first));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_potrf_first, dim3(1), dim3(512), smem_first, st, (const double*)M, n, nsplit,
                     slab_stride, h->Lmat, h->Linv, h->status);
  UCE_LAUNCH_CHECK();
  for (int j = 0; j + 1 < nb; ++j) {
    const int mt = nb - j - 1;
    const int tiles = mt * (mt + 1) / 2;
    hipLaunchKernelGGL(k_potrf_step, dim3(tiles), dim3(512), smem, st, M, n, j, h->Lmat, h->Linv,
                       h->status);
    UCE_LAUNCH_CHECK();
  }
  return UCE_OK;
}

int launch_potrf(uce_ctx* h, double* M, int n, hipStream_t st) {
  return launch_potrf_slabs(h, M, n, 1, 0, st);
}

int launch_trisolve(uce_ctx* h, int n, int m, const double* rhs64, const float* rhs32, int rhs_rows,
                    float* out, int out_rows, hipStream_t st) {
  const bool use_lds = n <= 1024;
  const size_t smem = 64 * 16 * 8 + (use_lds ? (size_t)n * 16 * 8 : 0);
  double* Yg = use_lds ? nullptr : h->Yg;
  static bool attr_set = false;
  if (!attr_set) {
    const int cap = 64 * 16 * 8 + 1024 * 16 * 8;
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trisolve<true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trisolve<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    attr_set = true;
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
