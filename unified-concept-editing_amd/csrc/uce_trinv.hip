// GEMM-shaped triangular solves for systems of >= 3 diagonal blocks (the primal d x d solve of a 1000 + 500 concept
// edit: reference `mat1 @ torch.inverse(mat2)` at uce_sd_erase.py:82):
//
//   X = L^-T L^-1 RHS   with   L^-1 formed EXPLICITLY by recursive doubling, then two triangular GEMMs.
//
// k_trisolve walks all of L inside every 16-column workgroup (48 workgroups, a 2 x 12-step dependent chain each:
// 228 us at n = 768, 5 % of the f64 MFMA rate).  Here the inverted 64 x 64 diagonal blocks the factorisation already
// produced are merged pairwise, level by level,
//        [ A  0 ]^-1    [  A^-1          0   ]
//        [ C  D ]    =  [ -D^-1 C A^-1  D^-1 ]        (two block-GEMM stages per level, log2(n/64) levels)
// and the solve becomes Y = L^-1 RHS, X = L^-T Y: every stage is a grid of independent 64 x 64 output tiles
// (f64 MFMA 16x16x4, 4 waves x 32 x 32 quadrants, operands staged through LDS with the contraction index
// contiguous, next k-chunk prefetched in registers), a kernel boundary between stages.  Tiles only visit the
// k-blocks inside the triangular / segment structure.
#include "uce_common.h"

namespace {

// One workgroup (4 waves) = one 32 x 32 output tile; a k-chunk is 64 wide and each wave contracts its own quarter of
// it (16 k), the four partial tiles are summed through LDS at the end: a 64-deep block product costs a wave 16 MFMAs
// (~0.9k cycles) instead of the 64 of a 64 x 64 tile per workgroup, and the grids are 4x larger (these stages are
// latency chains of few tiles, not throughput problems).
// LDS images, both read conflict-free with ds_read_b64 and written with 16-byte stores, no transposes:
//   "row" operand  R[32][RLD]  (row-major, k contiguous)   fragment = R[row0 + r][k]
//   "k-major" operand  K[64][KLD]  (k rows, 32 columns)    fragment = K[k][col0 + c]   (KLD = 48: the two k rows of a
//                                                           32-lane group sit 32 banks apart)
constexpr int RLD = 66;
constexpr int KLD = 48;
constexpr size_t TRINV_SMEM = (size_t)(32 * RLD + 2 * 64 * KLD) * sizeof(double);   // row image + two k-major images
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// rows [r0, r0+32) x k-chunk kc of a row-major f64 matrix -> 4 double2 per thread
__device__ __forceinline__ void row_load(double2_t (&v)[4], const double* G, int ld, int r0, int kc, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = tid + 256 * p, r = idx >> 5, c2 = idx & 31;
    v[p] = *(const double2_t*)(G + (size_t)(r0 + r) * ld + kc * 64 + 2 * c2);
  }
}
__device__ __forceinline__ void row_park(const double2_t (&v)[4], double* R, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = tid + 256 * p, r = idx >> 5, c2 = idx & 31;
    *(double2_t*)(R + r * RLD + 2 * c2) = v[p];
  }
}
// k rows [64 kc, +64) x columns [c0, c0+32) of a row-major matrix (f64, or f32 with rows >= row_limit zero)
template <bool F32>
__device__ __forceinline__ void kmaj_load(double2_t (&v)[4], const void* G, int ld, int kc, int c0, int tid, int row_limit) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = tid + 256 * p, r = idx >> 4, c2 = idx & 15;
    const int gr = kc * 64 + r;
    if (F32) {
      float2_t f = {0.f, 0.f};
      if (gr < row_limit) f = *(const float2_t*)((const float*)G + (size_t)gr * ld + c0 + 2 * c2);
      v[p] = (double2_t){(double)f[0], (double)f[1]};
    } else {
      v[p] = *(const double2_t*)((const double*)G + (size_t)gr * ld + c0 + 2 * c2);
    }
  }
}
__device__ __forceinline__ void kmaj_park(const double2_t (&v)[4], double* Kk, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = tid + 256 * p, r = idx >> 4, c2 = idx & 15;
    *(double2_t*)(Kk + r * KLD + 2 * c2) = v[p];
  }
}

// C_tile = sign * sum_{k = k0 .. k1} A_sub(k) * B_sub(k)     (32 x 32 output, rows r0.., columns c0..)
//   TA = false: A_sub(k) = A[r0 .. r0+32, 64k .. 64k+64]           (row operand)
//   TA = true : A_sub(k) = A[64k .. 64k+64, r0 .. r0+32]^T         (k-major operand: the transpose is free)
//   B_sub(k)  = B[64k .. 64k+64, c0 .. c0+32]   (f64, or f32 with rows >= b_rows zero)
// Returns this THREAD's 4 outputs of the reduced tile: rows (tid >> 3), columns 4 * (tid & 7) .. +3.
//   Adiag / Bdiag (optional): the inverted 64 x 64 diagonal blocks [nb][64][64]; when given, a 64-block of A (B) that
//   lies ON the diagonal is read from there (the diagonal blocks of L^-1 are never copied into Winv).
template <bool TA, bool B32>
__device__ __forceinline__ void tile_gemm(double (&out)[4], const double* A, int lda, const void* B, int ldb, int b_rows,
                                          int r0, int c0, int k0, int k1, unsigned char* smem,
                                          const double* Adiag = nullptr, const double* Bdiag = nullptr) {
  double* Rs = (double*)smem;                     // [32][RLD]
  double* Ka = Rs + 32 * RLD;                     // [64][KLD]  (A when TA)
  double* Kb = Ka + 64 * KLD;                     // [64][KLD]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  double4_t acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};
  if (k0 <= k1) {
    // two chunks in flight ahead of the MFMAs (an L2 round trip is longer than a chunk's 16 MFMAs per wave); the two
    // register sets are named (a runtime-indexed set would live in scratch memory)
    double2_t va0[4], vb0[4], va1[4], vb1[4];
    auto fetch = [&](int k, double2_t (&xa)[4], double2_t (&xb)[4]) {
      const bool ad = Adiag && k == (r0 >> 6), bd = !B32 && Bdiag && k == (c0 >> 6);
      if (TA) {
        if (ad) kmaj_load<false>(xa, Adiag + (size_t)k * 4096, 64, 0, r0 & 63, tid, 0);
        else kmaj_load<false>(xa, A, lda, k, r0, tid, 0);
      } else {
        if (ad) row_load(xa, Adiag + (size_t)k * 4096, 64, r0 & 63, 0, tid);
        else row_load(xa, A, lda, r0, k, tid);
      }
      if (bd) kmaj_load<false>(xb, Bdiag + (size_t)k * 4096, 64, 0, c0 & 63, tid, 0);
      else kmaj_load<B32>(xb, B, ldb, k, c0, tid, b_rows);
    };
    auto chunk = [&](int k, double2_t (&xa)[4], double2_t (&xb)[4]) {
      __syncthreads();                            // the previous chunk's fragments are read
      if (TA) kmaj_park(xa, Ka, tid);
      else row_park(xa, Rs, tid);
      kmaj_park(xb, Kb, tid);
      if (k + 2 <= k1) fetch(k + 2, xa, xb);      // refill this set: in flight under two chunks of MFMAs
      __syncthreads();
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int t = 16 * w + 4 * kb + kk;       // this wave's quarter of the chunk
        double a0, a1;
        if (TA) { a0 = Ka[t * KLD + r]; a1 = Ka[t * KLD + 16 + r]; }
        else { a0 = Rs[r * RLD + t]; a1 = Rs[(16 + r) * RLD + t]; }
        const double b0 = Kb[t * KLD + r], b1 = Kb[t * KLD + 16 + r];
        acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
        acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
        acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
        acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
      }
    };
    fetch(k0, va0, vb0);
    if (k0 < k1) fetch(k0 + 1, va1, vb1);
    for (int k = k0; k <= k1; k += 2) {
      chunk(k, va0, vb0);
      if (k + 1 <= k1) chunk(k + 1, va1, vb1);
    }
  }
  // ---- sum the four waves' partial tiles (fixed order: bit-repeatable)
  __syncthreads();
  double* Red = (double*)smem;                    // [4][32][33]
  {
    const int c = lane & 15, rq = lane >> 4;      // D layout: row = rq + 4q, col = c
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) Red[(w * 32 + m * 16 + rq + 4 * q) * 33 + n * 16 + c] = acc[m][n][q];
  }
  __syncthreads();
  const int orow = tid >> 3, oc = 4 * (tid & 7);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double s = Red[orow * 33 + oc + q];
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) s += Red[(ww * 32 + orow) * 33 + oc + q];
    out[q] = s;
  }
}
static_assert(4 * 32 * 33 * sizeof(double) <= TRINV_SMEM, "the reduction image must fit the operand images");

// Level with segments of S 64-blocks: pair p = segments [2pS, 2pS + S) (A) and [(2p+1)S, ...) (D, possibly short).
//   STAGE 1: T[i, j]    =  sum_{k in A, k >= j} L[i, k] Winv[k, j]          i in D, j in A
//   STAGE 2: Winv[i, j] = -sum_{k in D, k <= i} Winv[i, k] T[k, j]
// grid (32 x 32 tiles): x = column tile inside the A segment (0 .. 2S-1), y = row tile among the D rows of the level.
template <int STAGE>
__global__ __launch_bounds__(256) void k_trinv_merge(const double* __restrict__ L, double* __restrict__ Winv,
                                                     double* __restrict__ T, const double* __restrict__ Linv, int n, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int yb = blockIdx.y >> 1, p = yb / S, rr = yb % S;
  const int i = (2 * p + 1) * S + rr, a0 = 2 * p * S, j = a0 + (blockIdx.x >> 1);     // 64-blocks of the tile
  const int r0 = i * 64 + (blockIdx.y & 1) * 32, c0 = j * 64 + (blockIdx.x & 1) * 32;
  double o[4];
  if (STAGE == 1) tile_gemm<false, false>(o, L, n, Winv, n, 0, r0, c0, j, a0 + S - 1, smem, nullptr, Linv);
  else tile_gemm<false, false>(o, Winv, n, T, n, 0, r0, c0, (2 * p + 1) * S, i, smem, Linv, nullptr);
  double* dst = ((STAGE == 1) ? T : Winv) + (size_t)(r0 + (threadIdx.x >> 3)) * n + c0 + 4 * (threadIdx.x & 7);
  const double sign = (STAGE == 1) ? 1.0 : -1.0;
  *(double2_t*)dst = (double2_t){sign * o[0], sign * o[1]};
  *(double2_t*)(dst + 2) = (double2_t){sign * o[2], sign * o[3]};
}

// Y[i, :] = sum_{k <= i} Winv[i, k] RHS[k, :]      (RHS f64 [n, m] or f32 with rows >= rhs_rows zero)
template <bool B32>
__global__ __launch_bounds__(256) void k_trinv_fwd(const double* __restrict__ Winv, const double* __restrict__ Linv,
                                                   const void* __restrict__ rhs, int rhs_rows, double* __restrict__ Y,
                                                   int n, int m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // row tiles in DESCENDING order: tile i contracts i/2 + 1 chunks, the long ones are dispatched first
  const int r0 = (gridDim.y - 1 - blockIdx.y) * 32, c0 = blockIdx.x * 32;
  int k1 = r0 >> 6;
  if (B32) {                                   // blocks of zero rows contribute nothing
    const int last = (rhs_rows - 1) / 64;
    k1 = k1 < last ? k1 : last;
  }
  double o[4];
  tile_gemm<false, B32>(o, Winv, n, rhs, m, rhs_rows, r0, c0, 0, k1, smem, Linv, nullptr);
  double* dst = Y + (size_t)(r0 + (threadIdx.x >> 3)) * m + c0 + 4 * (threadIdx.x & 7);
  *(double2_t*)dst = (double2_t){o[0], o[1]};
  *(double2_t*)(dst + 2) = (double2_t){o[2], o[3]};
}

// X[i, :] = sum_{k >= i} Winv[k, i]^T Y[k, :]  ->  out f32, rows < out_rows
__global__ __launch_bounds__(256) void k_trinv_bwd(const double* __restrict__ Winv, const double* __restrict__ Linv,
                                                   const double* __restrict__ Y, float* __restrict__ out, int out_rows,
                                                   int n, int m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  double o[4];
  tile_gemm<true, false>(o, Winv, n, Y, m, 0, r0, c0, r0 >> 6, n / 64 - 1, smem, Linv, nullptr);
  const int gr = r0 + (threadIdx.x >> 3);
  if (gr < out_rows)
    *(float4_t*)(out + (size_t)gr * m + c0 + 4 * (threadIdx.x & 7)) = (float4_t){(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
}

}  // namespace

// Needs h->Lmat / h->Linv of launch_potrf, h->Wi [n, n] and h->Yg [n, m]; `scratch` [n, n] f64 (the factored matrix:
// dead after the factorisation).  m must be a multiple of 64 (it is the embedding width).
int launch_trisolve_inv(uce_ctx* h, int n, int m, const double* rhs64, const float* rhs32, int rhs_rows, float* out,
                        int out_rows, double* scratch, hipStream_t st) {
  const int nb = n / 64;
  const size_t smem = TRINV_SMEM;
  static PerDeviceOnce attr_once;
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trinv_merge<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trinv_merge<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trinv_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trinv_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trinv_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_once.commit(tok);
  }
  // (k_potrf_la has already formed the off-diagonal blocks of L^-1 beside the factorisation: no merge launches)
  for (int S = 1; S < nb && !h->wi_valid; S *= 2) {
    // D rows of the level: every block i with (i / S) odd
    int drows = 0;
    for (int i = 0; i < nb; ++i)
      if ((i / S) & 1) ++drows;
    if (!drows) continue;
    hipLaunchKernelGGL(k_trinv_merge<1>, dim3(2 * S, 2 * drows), dim3(256), smem, st, (const double*)h->Lmat, h->Wi, scratch, (const double*)h->Linv, n, S);
    hipLaunchKernelGGL(k_trinv_merge<2>, dim3(2 * S, 2 * drows), dim3(256), smem, st, (const double*)h->Lmat, h->Wi, scratch, (const double*)h->Linv, n, S);
    UCE_LAUNCH_CHECK();
  }
  const dim3 grid(m / 32, 2 * nb);
  if (rhs32)
    hipLaunchKernelGGL(k_trinv_fwd<true>, grid, dim3(256), smem, st, (const double*)h->Wi, (const double*)h->Linv, (const void*)rhs32, rhs_rows, h->Yg, n, m);
  else
    hipLaunchKernelGGL(k_trinv_fwd<false>, grid, dim3(256), smem, st, (const double*)h->Wi, (const double*)h->Linv, (const void*)rhs64, n, h->Yg, n, m);
  const int out_tiles = (out_rows + 31) / 32;
  hipLaunchKernelGGL(k_trinv_bwd, dim3(m / 32, out_tiles < 2 * nb ? out_tiles : 2 * nb), dim3(256), smem, st, (const double*)h->Wi,
                     (const double*)h->Linv, (const double*)h->Yg, out, out_rows, n, m);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
