// Dense apply  W_new = W_old + W_old Delta  on the bf16 matrix cores with fp32-equivalent products
// (reference: `mat1 @ torch.inverse(mat2)` per module, uce_sd_erase.py:82 - here one launch for all modules).
//
// gfx950's f32 MFMA peaks at 157 TF, its bf16 MFMA at 2.5 PF (16x).  An fp32 value is split exactly into three
// bf16 values  x = x_h + x_m + x_l  (8 + 8 + 8 significand bits; each residual subtraction is exact in fp32),
// and a product x*y is taken as the six partial products whose weight is >= 2^-16 relative:
//     x_l y_h + x_h y_l + x_m y_m + x_m y_h + x_h y_m + x_h y_h       (small terms first)
// Each partial product of two bf16 values is exact inside the MFMA (16-bit significand product, fp32
// accumulation); the dropped terms (x_m y_l, x_l y_m, x_l y_l) are below 2^-23 relative - one fp32 rounding.
// Six bf16 MFMAs replace the 16 "bf16-MFMA-equivalents" an f32 MFMA product costs: 2.7x the f32-MFMA rate.
//
//   k_split3      (I + Delta)^T [d, d] f32 -> three bf16 planes (once per edit, 590 k elements at d = 768)
//   k_apply_b3    NT GEMM, 128 x 128 tile per workgroup, 4 waves x (2 x 2) v_mfma_f32_32x32x16_bf16 tiles,
//                 K step 16, double-buffered LDS with register prefetch; the W_old operand is split on the fly
//                 while it is staged (each element is staged once per column tile), XCD-aware tile order as in k_apply.
// The residual rides in the product:  W_new = W_old (I + Delta), with the identity added to Delta^T before the split
// (fp32 add: the diagonal entry 1 + delta_ii keeps delta_ii to 2^-24 absolute = one fp32 rounding of the result; the three
// planes then hold 1 + delta_ii exactly).  Reading the residual tile separately (accumulator init, the round-2 form) cost
// a second fetch of W_old with 4-byte lane strides: PMC 294 MB per launch for 155.7 MB algorithmic.
#include "uce_common.h"
#include <cstdlib>

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int PLD = 24;           // plane row stride in bf16 elements: 48 B = 3 x 16 B -> conflict-free ds_read_b128
constexpr int PLANE = BM * PLD;   // elements per plane per buffer

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {   // v_cvt_pk_bf16_f32, round to nearest even
  const float2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float lo_f32(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_f32(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

// four consecutive f32 -> 4 bf16 in each of the three planes (8 bytes per plane)
__device__ __forceinline__ void split4(const float4_t x, uint2_t& h, uint2_t& m, uint2_t& l) {
  h[0] = cvt_pk(x[0], x[1]);
  h[1] = cvt_pk(x[2], x[3]);
  const float r0 = x[0] - lo_f32(h[0]), r1 = x[1] - hi_f32(h[0]);      // exact
  const float r2 = x[2] - lo_f32(h[1]), r3 = x[3] - hi_f32(h[1]);
  m[0] = cvt_pk(r0, r1);
  m[1] = cvt_pk(r2, r3);
  l[0] = cvt_pk(r0 - lo_f32(m[0]), r1 - hi_f32(m[0]));
  l[1] = cvt_pk(r2 - lo_f32(m[1]), r3 - hi_f32(m[1]));
}

__global__ __launch_bounds__(256) void k_split3(const float* __restrict__ src, unsigned short* __restrict__ planes,
                                                long n, int d) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;                                   // n is a multiple of 4 (d % 64 == 0)
  uint2_t h, m, l;
  float4_t x = *(const float4_t*)(src + i);
  const int row = (int)(i / d), c0 = (int)(i - (long)row * d);   // four consecutive columns of one row
  if (row >= c0 && row < c0 + 4) x[row - c0] += 1.0f;             // + I
  split4(x, h, m, l);
  *(uint2_t*)(planes + i) = h;
  *(uint2_t*)(planes + n + i) = m;
  *(uint2_t*)(planes + 2 * n + i) = l;
}

__device__ __forceinline__ int xcd_remap_b3(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = b & 7, local = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ float16_t mfma_bf16(uint4_t a, uint4_t b, float16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void k_apply_b3(const float* __restrict__ W_old,
                                                     const unsigned short* __restrict__ Bp,   // [3][d][d] bf16
                                                     float* __restrict__ W_new, long rows, int d) {
  // LDS: [2 buffers][A: 3 planes][B: 3 planes][128 rows][PLD]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = (unsigned short*)smem_raw;
  auto a_plane = [&](int buf, int p) { return lds + (buf * 6 + p) * PLANE; };
  auto b_plane = [&](int buf, int p) { return lds + (buf * 6 + 3 + p) * PLANE; };

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int ncol = (d + BN - 1) / BN;
  const int lid = xcd_remap_b3(blockIdx.x, gridDim.x);
  const long r0 = (long)(lid / ncol) * BM;
  const int j0 = (lid % ncol) * BN;
  const size_t dd = (size_t)d * d;

  // staging coordinates.  A: 128 rows x 16 f32 per step = 2 float4 per thread; B: 128 rows x 16 bf16 per plane
  // = one 16-byte chunk per plane per thread.
  const int arow = tid >> 2, ac4 = (tid & 3) * 4;        // rows arow, arow + 64
  const int brow = tid >> 1, bc8 = (tid & 1) * 8;
  const float* aptr[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    long gr = r0 + arow + 64 * p;
    if (gr > rows - 1) gr = rows - 1;
    aptr[p] = W_old + gr * d + ac4;
  }
  int gj = j0 + brow;
  if (gj > d - 1) gj = d - 1;
  const unsigned short* bptr = Bp + (size_t)gj * d + bc8;

  // (the residual is part of the product: the planes hold (I + Delta)^T)
  float16_t acc[2][2];
  const int ccol = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  float4_t ra[2];
  uint4_t rb[3];
  auto g_load = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 2; ++p) ra[p] = *(const float4_t*)(aptr[p] + kt * BK);
#pragma unroll
    for (int p = 0; p < 3; ++p) rb[p] = *(const uint4_t*)(bptr + p * dd + kt * BK);
  };
  auto s_store = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      uint2_t h, m, l;
      split4(ra[p], h, m, l);
      const int off = (arow + 64 * p) * PLD + ac4;
      *(uint2_t*)(a_plane(buf, 0) + off) = h;
      *(uint2_t*)(a_plane(buf, 1) + off) = m;
      *(uint2_t*)(a_plane(buf, 2) + off) = l;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) *(uint4_t*)(b_plane(buf, p) + brow * PLD + bc8) = rb[p];
  };

  g_load(0);
  s_store(0);
  __syncthreads();

  const int nk = d / BK;
  const int fr = lane & 31, fk = 8 * (lane >> 5);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) g_load(kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    // all twelve fragments are requested up front, in the order the terms consume them
    uint4_t fa[2][3], fb[2][3];
    auto rd_a = [&](int p) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt][p] = *(const uint4_t*)(a_plane(cur, p) + (wm + mt * 32 + fr) * PLD + fk);
    };
    auto rd_b = [&](int p) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) fb[nt][p] = *(const uint4_t*)(b_plane(cur, p) + (wn + nt * 32 + fr) * PLD + fk);
    };
    rd_a(2); rd_b(0); rd_a(0); rd_b(2); rd_a(1); rd_b(1);
    __builtin_amdgcn_sched_barrier(0);
    // six partial products per tile, small terms first; the four tiles are interleaved so that consecutive
    // MFMAs never depend on each other
#pragma unroll
    for (int term = 0; term < 6; ++term) {
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};       // plane of the W_old operand   (0 = high, 1 = mid, 2 = low)
      constexpr int PB[6] = {0, 2, 1, 0, 1, 0};       // plane of the Delta^T operand
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_bf16(fa[mt][PA[term]], fb[nt][PB[term]], acc[mt][nt]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) s_store(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long gr = r0 + wm + mt * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        const int gc = j0 + wn + nt * 32 + ccol;
        if (gr < rows && gc < d) W_new[gr * d + gc] = acc[mt][nt][r];
      }
}

}  // namespace

// planes: h->DeltaP, 3 * d * d bf16
int launch_apply_b3(const float* W_old, const float* DeltaT, unsigned short* planes, float* W_new, long rows, int d,
                    hipStream_t st, uce_ctx* h) {
  const long n = (long)d * d;
  {
    UceProfScope ps(h, "k_split3", st);
    hipLaunchKernelGGL(k_split3, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, DeltaT, planes, n, d);
  }
  UCE_LAUNCH_CHECK();
  const size_t smem = (size_t)2 * 6 * PLANE * sizeof(unsigned short);
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_apply_b3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  const long row_tiles = (rows + BM - 1) / BM;
  const int col_tiles = (d + BN - 1) / BN;
  const long nwg = row_tiles * col_tiles;
  if (nwg > 0x7fffffffL) return UCE_EINVAL;
  {
    UceProfScope ps(h, "k_apply_b3", st);
    hipLaunchKernelGGL(k_apply_b3, dim3((unsigned)nwg), dim3(256), smem, st, W_old, planes, W_new, rows, d);
  }
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
