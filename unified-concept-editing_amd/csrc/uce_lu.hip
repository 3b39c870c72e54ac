// General (indefinite) d x d solve in f64 for the EDGE path of the edit: negative scales or lambda <= 0 make
// A = lambda I + sum_i s_i c_i c_i^T symmetric INDEFINITE; the reference still inverts it (`mat1 @ torch.inverse(mat2)`,
// trainscripts/uce_sd_erase.py:82 - an LU inverse does not care about definiteness), the hot path's Cholesky cannot.
//
//   X [n, m] f32 = A^-1 B          A [n, n] f64 (destroyed), B [n, m] f64 (destroyed)
//
// Gaussian elimination with partial pivoting on the augmented matrix [A | B], one column at a time: per column k
//   k_lu_pivot   (one workgroup)  arg max |a_ik| over i >= k in a fixed order, swap rows k and p of A (columns >= k) and of B,
//                                 record 1 / a_kk; a pivot below n eps max|A| marks the system singular (status = k + 1)
//   k_lu_update  (grid)           l_i = a_ik / a_kk; a_ij -= l_i a_kj (j > k), b_ij -= l_i b_kj for every row i > k
// then the back substitution per column, from the last: x_k = b_k / u_kk; b_i -= u_ik x_k (i < k).  2 n + n launches of latency-
// bound kernels (n = 768: ~14 ms, n = 2048: ~80 ms): an edge path, outside every measured configuration - the point is that NO
// vendor solver stands behind the product path.  Error ~ n eps cond(A) like any GEPP in f64.
#include "uce_common.h"

namespace {

// one workgroup of 256: pivot search, row swap, reciprocal
__global__ __launch_bounds__(256) void k_lu_pivot(double* __restrict__ A, double* __restrict__ B, int n, int m, int k, double tiny,
                                                  int* __restrict__ status, double* __restrict__ rpiv) {
  __shared__ double bv[256];
  __shared__ int bi[256];
  const int tid = threadIdx.x;
  double best = -1.0;
  int at = k;
  for (int i = k + tid; i < n; i += 256) {                             // (ties: the smallest row index - a fixed order)
    const double v = fabs(A[(size_t)i * n + k]);
    if (v > best) { best = v; at = i; }
  }
  bv[tid] = best;
  bi[tid] = at;
  __syncthreads();
  for (int o = 128; o; o >>= 1) {
    if (tid < o) {
      const double v = bv[tid + o];
      const int j = bi[tid + o];
      if (v > bv[tid] || (v == bv[tid] && j < bi[tid])) { bv[tid] = v; bi[tid] = j; }
    }
    __syncthreads();
  }
  const int p = bi[0];
  const double piv_abs = bv[0];
  if (tid == 0) {
    if (!(piv_abs > tiny) && *status == 0) *status = k + 1;           // singular to working precision (or NaN): first failing column
    rpiv[0] = 1.0 / A[(size_t)p * n + k];
  }
  if (p != k) {
    for (int j = k + tid; j < n; j += 256) {
      const double t = A[(size_t)k * n + j];
      A[(size_t)k * n + j] = A[(size_t)p * n + j];
      A[(size_t)p * n + j] = t;
    }
    for (int j = tid; j < m; j += 256) {
      const double t = B[(size_t)k * m + j];
      B[(size_t)k * m + j] = B[(size_t)p * m + j];
      B[(size_t)p * m + j] = t;
    }
  }
}

// rows i > k: eliminate column k.  Block = 4 rows x 64 column-lanes; columns of [A (j > k) | B] walked in strides of 64
__global__ __launch_bounds__(256) void k_lu_update(double* __restrict__ A, double* __restrict__ B, int n, int m, int k,
                                                   const double* __restrict__ rpiv) {
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int i = k + 1 + blockIdx.x * 4 + r;
  if (i >= n) return;
  const double l = A[(size_t)i * n + k] * rpiv[0];
  const double* ak = A + (size_t)k * n;
  double* ai = A + (size_t)i * n;
  for (int j = k + 1 + lane; j < n; j += 64) ai[j] = fma(-l, ak[j], ai[j]);
  const double* bk = B + (size_t)k * m;
  double* bi = B + (size_t)i * m;
  for (int j = lane; j < m; j += 64) bi[j] = fma(-l, bk[j], bi[j]);
  if (lane == 0) ai[k] = l;                                            // (the multiplier, for whoever wants the factors)
}

// back substitution, column k: block 0 also finishes row k
__global__ __launch_bounds__(256) void k_lu_back(const double* __restrict__ A, double* __restrict__ B, float* __restrict__ X, int n, int m,
                                                 int k) {
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
  const double rk = 1.0 / A[(size_t)k * n + k];
  const double* bk = B + (size_t)k * m;
  const int i = (int)blockIdx.x * 4 + r - 1;                          // row -1 of the grid = row k itself (x_k = b_k / u_kk)
  if (i < 0) {
    for (int j = lane; j < m; j += 64) X[(size_t)k * m + j] = (float)(bk[j] * rk);
    return;
  }
  if (i >= k) return;
  const double u = A[(size_t)i * n + k] * rk;
  double* bi = B + (size_t)i * m;
  for (int j = lane; j < m; j += 64) bi[j] = fma(-u, bk[j], bi[j]);
}

__global__ void k_lu_absmax(const double* __restrict__ A, size_t total, double* __restrict__ out) {
  __shared__ double red[256];
  double v = 0.0;
  for (size_t e = threadIdx.x; e < total; e += 256) v = fmax(v, fabs(A[e]));
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

}  // namespace

extern "C" int uce_solve_general(uce_handle_t h, double* A, double* B, int n, int m, float* X, uce_stream_t stream) {
  if (!h || !A || !B || !X || n <= 0 || m <= 0) return UCE_EINVAL;
  UCE_ENTER(h);
  hipStream_t st = (hipStream_t)stream;
  UceProfScope ps(h, "uce_solve_general", st);
  // scratch: [0] 1 / pivot of the current column, [1] max |A| (both live in the handle's Linv block, unused on this path)
  int rc = uce_ensure(h, 64, 64);
  if (rc) return rc;
  double* scr = h->Linv;
  UCE_HIP_TRY(hipMemsetAsync(h->status, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_lu_absmax, dim3(1), dim3(256), 0, st, (const double*)A, (size_t)n * n, scr + 1);
  UCE_LAUNCH_CHECK();
  double amax = 0.0;
  UCE_HIP_TRY(hipMemcpyAsync(&amax, scr + 1, sizeof(double), hipMemcpyDeviceToHost, st));
  UCE_HIP_TRY(hipStreamSynchronize(st));                              // (an edge path: one host round trip for the singularity threshold)
  const double tiny = (double)n * 2.220446049250313e-16 * amax;
  for (int k = 0; k < n; ++k) {
    hipLaunchKernelGGL(k_lu_pivot, dim3(1), dim3(256), 0, st, A, B, n, m, k, tiny, h->status, scr);
    const int rows_below = n - 1 - k;
    if (rows_below > 0)
      hipLaunchKernelGGL(k_lu_update, dim3((unsigned)((rows_below + 3) / 4)), dim3(256), 0, st, A, B, n, m, k, (const double*)scr);
  }
  UCE_LAUNCH_CHECK();
  for (int k = n - 1; k >= 0; --k)
    hipLaunchKernelGGL(k_lu_back, dim3((unsigned)((k + 1 + 3) / 4)), dim3(256), 0, st, (const double*)A, B, X, n, m, k);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
