// Gram accumulation for the UCE closed form (reference: uce_sd_erase.py:56-79, the rank-1
// `mat2 += s * c c^T` / `mat1 += s * v* c^T` loops; here done ONCE for all modules).
//
//   primal : A  = lambda I + C^T S C          [d,d]   (TN product over the N concepts)
//            Bt = C_e^T S_e (G - C_e)         [d,d]
//   dual   : K  = lambda S^-1 + C C^T         [n_pad,n_pad]  (NT product over the d features)
//
// fp32 inputs are widened to f64 in registers (products of two fp32 values are exact in f64)
// and accumulated with v_mfma_f64_16x16x4_f64.  One workgroup = 4 waves = one 64x64 output
// tile, each wave a 32x32 quadrant (2x2 MFMA tiles).  Split-K partial tiles go to slabs that a
// reduction kernel sums in a fixed order (bit-repeatable; no atomics).
#include "uce_common.h"
#include "uce_gram_tile.h"

namespace {

constexpr int NT_LD = 40;     // row stride (floats) of the k-contiguous NT tiles: conflict-free b128

// ------------------------------------------------------------------------------------------
// primal: tiles [0, nA) are the lower-triangular tiles of A, tiles [nA, nA + nb*nb) those of Bt
// ------------------------------------------------------------------------------------------
// tiles tile0 .. of the enumeration [A lower tiles | Bt tiles]: the whole Gram (tile0 = 0, all tiles), A only, or Bt only
__global__ __launch_bounds__(512) void k_gram_primal(GramPrimalArgs a, int tile0) {
  __shared__ __attribute__((aligned(16))) unsigned char stage_raw[2 * 2 * KC * 64 * sizeof(float)];   // 32 KB
  __shared__ float Ss[2][KC];
  const int nb = a.d / 64;
  const int nA = nb * (nb + 1) / 2;
  const int t = (int)blockIdx.x + tile0;
  int ti, tj;
  const bool isA = t < nA;
  if (isA) tri_decode(t, ti, tj);
  else { ti = (t - nA) / nb; tj = (t - nA) % nb; }
  gram_primal_tile(a, isA, ti, tj, (int)blockIdx.y, stage_raw, Ss);
}

// ------------------------------------------------------------------------------------------
// dual: K = lambda S^-1 + C C^T, n_pad = roundup(N, 64); rows >= N are zero with a unit diagonal
// ------------------------------------------------------------------------------------------
// Blocks with blockIdx.x >= n_tiles do the side jobs of the dual path so that they cost no
// launch of their own: Dm = G - C_e (fp32) and the reset of the solver's status word.
__global__ __launch_bounds__(256) void k_gram_dual(const float* __restrict__ C,
                                                   const float* __restrict__ s, int N, int d,
                                                   float lamb, double* __restrict__ outK, int n_pad,
                                                   int kchunk, size_t slab_stride, int n_tiles,
                                                   const float* __restrict__ G, float* __restrict__ Dm,
                                                   long dm_f4, int* __restrict__ status) {
  if ((int)blockIdx.x >= n_tiles) {
    if (blockIdx.y != 0) return;
    const int nside = gridDim.x - n_tiles;
    if (blockIdx.x == (unsigned)n_tiles && threadIdx.x == 0 && status) *status = 0;
    for (long i = (long)(blockIdx.x - n_tiles) * 256 + threadIdx.x; i < dm_f4; i += (long)nside * 256) {
      const float4_t g = ((const float4_t*)G)[i], c = ((const float4_t*)C)[i];
      ((float4_t*)Dm)[i] = g - c;
    }
    return;
  }
  __shared__ __attribute__((aligned(16))) float As[64][NT_LD];
  __shared__ __attribute__((aligned(16))) float Bs[64][NT_LD];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  int ti, tj;
  tri_decode(blockIdx.x, ti, tj);
  const int split = blockIdx.y;
  const int k_begin = split * kchunk;
  const int k_end = min(d, k_begin + kchunk);

  double4_t acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int lrow = tid >> 3;        // 0..31
  const int lc4 = (tid & 7) * 4;    // 0..28
  for (int k0 = k_begin; k0 < k_end; k0 += KC) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = p * 32 + lrow;
      float4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
      const int ra = ti * 64 + r, rb = tj * 64 + r;
      if (k0 + lc4 < k_end) {  // d is a multiple of 64, chunks are multiples of 32
        if (ra < N) a = *(const float4_t*)(C + (size_t)ra * d + k0 + lc4);
        if (rb < N) b = *(const float4_t*)(C + (size_t)rb * d + k0 + lc4);
      }
      *(float4_t*)&As[r][lc4] = a;
      *(float4_t*)&Bs[r][lc4] = b;
    }
    __syncthreads();
    // k permutation: in 16-k group u, MFMA t uses k = 16u + 4*(lane>>4) + t for BOTH operands
#pragma unroll
    for (int u = 0; u < KC / 16; ++u) {
      const int kofs = u * 16 + 4 * (lane >> 4);
      const float4_t fa0 = *(const float4_t*)&As[wr * 32 + (lane & 15)][kofs];
      const float4_t fa1 = *(const float4_t*)&As[wr * 32 + 16 + (lane & 15)][kofs];
      const float4_t fb0 = *(const float4_t*)&Bs[wc * 32 + (lane & 15)][kofs];
      const float4_t fb1 = *(const float4_t*)&Bs[wc * 32 + 16 + (lane & 15)][kofs];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double a0 = (double)fa0[t], a1 = (double)fa1[t];
        const double b0 = (double)fb0[t], b1 = (double)fb1[t];
        acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
        acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
        acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
        acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
      }
    }
    __syncthreads();
  }
  double* out = outK + (size_t)split * slab_stride;
  store_quadrant(out, n_pad, ti * 64 + wr * 32, tj * 64 + wc * 32, acc, lane, ti != tj, 0.0, s, lamb, N,
                 split == 0);
}

// out[i] = sum over splits of slabs[split][i], fixed order
__global__ void k_reduce_slabs(const double* __restrict__ slabs, size_t slab_stride, int nsplit,
                               double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = slabs[i];
  for (int sidx = 1; sidx < nsplit; ++sidx) v += slabs[(size_t)sidx * slab_stride + i];
  out[i] = v;
}

// G - C rows in fp32 (the low-rank left factor)
__global__ void k_sub_rows(const float* __restrict__ G, const float* __restrict__ C,
                           float* __restrict__ Dm, long n4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4_t g = ((const float4_t*)G)[i], c = ((const float4_t*)C)[i];
  ((float4_t*)Dm)[i] = g - c;
}

}  // namespace

static int pick_split(int tiles, int kchunks_total) {
  if (tiles >= 128) return 1;
  int want = (256 + tiles - 1) / tiles;
  if (want > kchunks_total) want = kchunks_total;
  return want < 1 ? 1 : want;
}

// which = 0: A and Bt (one launch, split over the concepts when the tiles alone do not fill the chip); 1: A only - the
// caller has arranged for Bt elsewhere (uce_edit: rider workgroups of the Cholesky launch); 2: Bt only
int launch_gram_primal(uce_ctx* h, const float* C, const float* G, const float* s, int N, int N_edit,
                       int d, float lamb, double* A, double* Bt, hipStream_t st, int which) {
  const int nb = d / 64;
  const int nA = nb * (nb + 1) / 2;
  const int tile0 = which == 2 ? nA : 0;
  const int tiles = which == 1 ? nA : (which == 2 ? nb * nb : nA + nb * nb);
  const int Kmax = which == 2 ? N_edit : N;
  if (Kmax <= 0) {                                    // (Bt of an edit without edit concepts)
    UCE_HIP_TRY(hipMemsetAsync(Bt, 0, (size_t)d * d * sizeof(double), st));
    return UCE_OK;
  }
  const int chunks = (Kmax + KC - 1) / KC;
  int nsplit = pick_split(tiles, chunks);
  // A alone (78 lower tiles at d = 768): ONE round of workgroups - 3 x 78 = 234 on 256 CUs, not 4 x 78 in two rounds
  if (which == 1 && tiles < 128) {
    nsplit = 256 / tiles < chunks ? 256 / tiles : chunks;
    const size_t cap = h->slabs_bytes / ((size_t)d * d * sizeof(double));   // the slab workspace is sized for the one-launch form
    if ((size_t)nsplit > cap) nsplit = cap ? (int)cap : 1;
  }
  int kchunk = ((chunks + nsplit - 1) / nsplit) * KC;
  nsplit = (Kmax + kchunk - 1) / kchunk;
  const size_t mat = (size_t)d * d;
  if (nsplit == 1) {
    const GramPrimalArgs a{C, G, s, N, N_edit, d, lamb, A, Bt, kchunk, (size_t)0};
    hipLaunchKernelGGL(k_gram_primal, dim3(tiles, 1), dim3(512), 0, st, a, tile0);
    UCE_LAUNCH_CHECK();
    return UCE_OK;
  }
  // slabs: [nsplit][A | Bt] (both), [nsplit][A] or [nsplit][Bt]
  const size_t stride = which == 0 ? 2 * mat : mat;
  const size_t need = (size_t)nsplit * stride * sizeof(double);
  if (need > h->slabs_bytes) return UCE_ENOMEM;
  double* sA = h->slabs;
  double* sB = which == 0 ? h->slabs + mat : h->slabs;
  const GramPrimalArgs a{C, G, s, N, N_edit, d, lamb, sA, sB, kchunk, stride};
  hipLaunchKernelGGL(k_gram_primal, dim3(tiles, nsplit), dim3(512), 0, st, a, tile0);
  UCE_LAUNCH_CHECK();
  // a split whose k-range is beyond N_edit still writes zeros to its Bt slab, so both reduce fully
  const int thr = 256;
  if (which != 2)
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((mat + thr - 1) / thr)), dim3(thr), 0, st,
                       (const double*)sA, stride, nsplit, A, mat);
  if (which != 1)
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((mat + thr - 1) / thr)), dim3(thr), 0, st,
                       (const double*)sB, stride, nsplit, Bt, mat);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

int launch_gram_dual(uce_ctx* h, const float* C, const float* s, int N, int d, float lamb, double* K,
                     int n_pad, const float* G, float* Dm, int N_edit, int* nsplit_out,
                     size_t* slab_stride_out, hipStream_t st) {
  const int nb = n_pad / 64;
  const int tiles = nb * (nb + 1) / 2;
  const int chunks = d / KC;
  int nsplit = pick_split(tiles, chunks);
  if (nb == 1 && nsplit > 8) nsplit = 8;   // k_potrf_first sums the slabs itself: keep that short
  int kchunk = ((chunks + nsplit - 1) / nsplit) * KC;
  nsplit = (d + kchunk - 1) / kchunk;
  const size_t mat = (size_t)n_pad * n_pad;
  const long dm_f4 = (Dm && N_edit > 0) ? (long)N_edit * d / 4 : 0;
  int nside = dm_f4 ? (int)((dm_f4 + 1023) / 1024) : 1;   // 4 float4 per thread
  if (nside > 64) nside = 64;
  *nsplit_out = 1;
  *slab_stride_out = 0;
  if (nsplit == 1) {
    hipLaunchKernelGGL(k_gram_dual, dim3(tiles + nside, 1), dim3(256), 0, st, C, s, N, d, lamb, K, n_pad,
                       kchunk, (size_t)0, tiles, G, Dm, dm_f4, h->status);
    UCE_LAUNCH_CHECK();
    return UCE_OK;
  }
  const size_t need = (size_t)nsplit * mat * sizeof(double);
  if (need > h->slabs_bytes) return UCE_ENOMEM;
  hipLaunchKernelGGL(k_gram_dual, dim3(tiles + nside, nsplit), dim3(256), 0, st, C, s, N, d, lamb, h->slabs,
                     n_pad, kchunk, mat, tiles, G, Dm, dm_f4, h->status);
  UCE_LAUNCH_CHECK();
  if (nb == 1) {  // single block: k_potrf_first sums the slabs itself
    *nsplit_out = nsplit;
    *slab_stride_out = mat;
    return UCE_OK;
  }
  const int thr = 256;
  hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((mat + thr - 1) / thr)), dim3(thr), 0, st,
                     (const double*)h->slabs, mat, nsplit, K, mat);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

int launch_sub_rows(const float* G, const float* C, float* Dm, long n, hipStream_t st) {
  const long n4 = n / 4;
  const int thr = 256;
  hipLaunchKernelGGL(k_sub_rows, dim3((unsigned)((n4 + thr - 1) / thr)), dim3(thr), 0, st, G, C, Dm, n4);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
