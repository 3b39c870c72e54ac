// Dense apply  W_new = W_old (I + Delta)  on the f16 matrix cores with fp32-equivalent products, direct-to-LDS form
// (reference: `mat1 @ torch.inverse(mat2)` per module, uce_sd_erase.py:82 - here one launch for all modules).
//
// An fp32 value scaled by a power of two into [2^14, 2^15) splits EXACTLY into two f16 values,
//     x s = x_h + x_l + e,   x_h = rn_f16(x s)  (11 significand bits),  x_l = rn_f16(x s - x_h)  (the next 11; the
//     subtraction is exact in fp32 and leaves <= 13 bits),  |e| <= 2^-22 |x s|: 22 of fp32's 24 significand bits,
// and a product x y is taken as   x_l y_h + x_h y_l + x_h y_h   (small terms first): every partial product of two f16
// values is exact inside the MFMA (22-bit significand, fp32 accumulation), the dropped term x_l y_l is below 2^-22
// relative.  Against fp64 the whole apply measures 3.2e-7 rel. Frobenius on the SD-1.4 slab (bf16 x 3: 4.3e-7, the f32-MFMA
// kernel: 6.8e-7 - fp32 accumulation over 768 terms dominates either way).  THREE f16 MFMAs per fp32-equivalent product - half of the six the three-way bf16 split (uce_apply_b3.hip)
// needs for the same accuracy.  What the bf16 form gets for free and this one has to provide is RANGE: f16 spans
// 2^-14 .. 2^15, so every row of W_old and every row of (I + Delta)^T gets its own power-of-two scale (the row maximum
// goes to [2^14, 2^15); elements more than 2^16 below their row's maximum have a denormal x_l: an absolute error of
// <= 2^-29 of that maximum - nothing a Frobenius norm can see, nor any row of W against I + Delta); the scales factor out of the product as  out[m][n] = acc[m][n] * 2^-e_m * 2^-f_n,  exact.
//
//   k_split_h2 / _h2d  one wave per row: row maximum -> scale -> the two f16 planes + the inverse scale.  W_old
//                      [rows, d] (HBM pass: 4 B in, 4 B out per element) and (I + Delta)^T [d, d] (IDENT: + 1 on the
//                      diagonal - the residual rides in the product, W_old is read by this pass only).
//   k_apply_h2         NT GEMM over the planes: workgroup = 320 rows x 256 columns, 8 waves = 2 (rows) x 4 (columns),
//                      wave tile 160 x 64 = 5 x 2 v_mfma_f32_32x32x16_f16 tiles (30 MFMAs per 14 fragment reads per
//                      16-deep step); a k-tile = 32 columns of all four planes, moved by `buffer_load_dwordx4 ... lds`
//                      straight into LDS (16 rows x 64 B per wave instruction, 9 per wave and k-tile; rows >= rows / >= d
//                      get an out-of-range offset and land as zeros); two 72 KB stages: tile t + 1 lands while the
//                      1 920 MFMA cycles per wave of tile t run; bank swizzle on the SOURCE address as in
//                      uce_conv_dma.hip.  SD-1.4's 24 960 rows x 768 columns = 78 x 3 = 234 workgroups: one round on
//                      256 CUs (the 128 x 128 tiles of the bf16 form: 1 170 workgroups in 2.3 rounds of 2 per CU).
//                      Epilogue: 4-byte stores, a wave instruction covers 2 rows x 128 contiguous bytes.
#include "uce_common.h"
#include "uce_h2split.h"

namespace {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int H2_BM = 320, H2_BN = 256, H2_BK = 32;
constexpr int H2_TM = 5, H2_TN = 2;                                  // 32 x 32 tiles per wave (rows, columns)
constexpr int H2_AH = 0, H2_AL = H2_BM * 64, H2_BH = 2 * H2_BM * 64, H2_BL = H2_BH + H2_BN * 64;
constexpr int H2_STAGE = 2 * (H2_BM + H2_BN) * 64;                   // 73 728 B: four planes x 32 columns
constexpr unsigned H2_OOB = 0x80000000u;

// W_old [rows, d]: one wave per row
__global__ __launch_bounds__(256) void k_split_h2(const float* __restrict__ src, unsigned short* __restrict__ hi,
                                                  unsigned short* __restrict__ lo, float* __restrict__ inv_scale, long rows,
                                                  int d) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < rows) h2_split_row<false>(src, hi, lo, inv_scale, row, d, threadIdx.x & 63);
}
// (I + Delta)^T [d, d]
__global__ __launch_bounds__(256) void k_split_h2d(const float* __restrict__ src, unsigned short* __restrict__ hi,
                                                   unsigned short* __restrict__ lo, float* __restrict__ inv_scale, int d) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < d) h2_split_row<true>(src, hi, lo, inv_scale, row, d, threadIdx.x & 63);
}

__device__ __forceinline__ int h2_xcd_remap(int b, int nwg) {        // consecutive tiles (the column tiles of one row tile) on one XCD
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = b & 7, local = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ float16_t h2_mfma(uint4_t a, uint4_t b, float16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(512) void k_apply_h2(const unsigned short* __restrict__ Ap,   // [2][rows][d] f16 (high, low)
                                                  const unsigned short* __restrict__ Bp,   // [2][d][d] f16 of (I + Delta)^T
                                                  const float* __restrict__ rs,            // [rows] 2^-e of the W_old rows
                                                  const float* __restrict__ cb,            // [d]    2^-f of the (I + Delta)^T rows
                                                  float* __restrict__ W_new, long rows, int d, int ncol) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int li = lane & 31, lh = lane >> 5;
  const int lid = h2_xcd_remap(blockIdx.x, gridDim.x);
  const long m0 = (long)(lid / ncol) * H2_BM;
  const int n0 = (lid % ncol) * H2_BN;

  // ---- staging coordinates (k-tile invariant).  A wave instruction fills 16 rows x 64 B; lane = (row r, piece p);
  // instruction j of wave w is row group 8 j + w: 0..19 W high, 20..39 W low (j = 0..4), then 0..15 Delta high,
  // 16..31 Delta low (j = 0..3).  Piece p of row R holds source piece p ^ ((R >> 2) & 3), and 16 | group base.
  const int r = lane >> 2, c = (lane & 3) ^ ((r >> 2) & 3);
  const unsigned a_plane = (unsigned)(rows * d * 2), b_plane = (unsigned)((long)d * d * 2);
  unsigned a_off[5], b_off[4];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int g = 8 * j + w, pl = g >= 20 ? 1 : 0;
    const long row = m0 + 16 * (g - 20 * pl) + r;
    a_off[j] = row < rows ? pl * a_plane + (unsigned)((row * d + c * 8) * 2) : H2_OOB;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = 8 * j + w, pl = g >= 16 ? 1 : 0;
    const int row = n0 + 16 * (g - 16 * pl) + r;
    b_off[j] = row < d ? pl * b_plane + (unsigned)(((long)row * d + c * 8) * 2) : H2_OOB;
  }
  const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)(2 * a_plane), 0x00020000);
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)(2 * b_plane), 0x00020000);
  auto stage = [&](int st, int kt) {
    unsigned char* sbase = smem + st * H2_STAGE + w * 1024;
    const unsigned ko = (unsigned)(kt * H2_BK * 2);
#pragma unroll
    for (int j = 0; j < 5; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (lds_void*)(sbase + j * 8192), 16, a_off[j] == H2_OOB ? H2_OOB : a_off[j] + ko, 0,
                                               0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (lds_void*)(sbase + H2_BH + j * 8192), 16,
                                               b_off[j] == H2_OOB ? H2_OOB : b_off[j] + ko, 0, 0, 0);
  };

  float16_t acc[H2_TM][H2_TN];
#pragma unroll
  for (int b = 0; b < H2_TM; ++b)
#pragma unroll
    for (int a = 0; a < H2_TN; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[b][a][q] = 0.f;
  int arow[H2_TM], brow[H2_TN];                                      // fragment rows -> byte offset of the row + its swizzle key
#pragma unroll
  for (int b = 0; b < H2_TM; ++b) arow[b] = (wm * H2_TM + b) * 32 + li;
#pragma unroll
  for (int a = 0; a < H2_TN; ++a) brow[a] = (wn * H2_TN + a) * 32 + li;

  float cs[H2_TN];                                                   // column scales / byte offsets of the epilogue
  unsigned co[H2_TN];
#pragma unroll
  for (int a = 0; a < H2_TN; ++a) {
    const int n = n0 + (wn * H2_TN + a) * 32 + li;
    cs[a] = n < d ? cb[n] : 0.f;
    co[a] = n < d ? (unsigned)(n * 4) : H2_OOB;
  }

  const int NK = d / H2_BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < NK; ++kt) {
    if (kt + 1 < NK) stage((kt + 1) & 1, kt + 1);
    const unsigned char* sb = smem + (kt & 1) * H2_STAGE;
#pragma unroll
    for (int s = 0; s < H2_BK / 16; ++s) {
      const int cc = 2 * s + lh;
      uint4_t ah[H2_TM], al[H2_TM], bh[H2_TN], bl[H2_TN];
#pragma unroll
      for (int b = 0; b < H2_TM; ++b) al[b] = *(const uint4_t*)(sb + H2_AL + arow[b] * 64 + ((cc ^ ((arow[b] >> 2) & 3)) << 4));
#pragma unroll
      for (int a = 0; a < H2_TN; ++a) bh[a] = *(const uint4_t*)(sb + H2_BH + brow[a] * 64 + ((cc ^ ((brow[a] >> 2) & 3)) << 4));
#pragma unroll
      for (int b = 0; b < H2_TM; ++b) ah[b] = *(const uint4_t*)(sb + H2_AH + arow[b] * 64 + ((cc ^ ((arow[b] >> 2) & 3)) << 4));
#pragma unroll
      for (int a = 0; a < H2_TN; ++a) bl[a] = *(const uint4_t*)(sb + H2_BL + brow[a] * 64 + ((cc ^ ((brow[a] >> 2) & 3)) << 4));
      // three partial products per tile, small terms first; rows of the result = rows of W_old (first operand)
#pragma unroll
      for (int b = 0; b < H2_TM; ++b)
#pragma unroll
        for (int a = 0; a < H2_TN; ++a) acc[b][a] = h2_mfma(al[b], bh[a], acc[b][a]);
#pragma unroll
      for (int b = 0; b < H2_TM; ++b)
#pragma unroll
        for (int a = 0; a < H2_TN; ++a) acc[b][a] = h2_mfma(ah[b], bl[a], acc[b][a]);
#pragma unroll
      for (int b = 0; b < H2_TM; ++b)
#pragma unroll
        for (int a = 0; a < H2_TN; ++a) acc[b][a] = h2_mfma(ah[b], bh[a], acc[b][a]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // tile kt + 1 has landed (this wave's part)
    __builtin_amdgcn_s_barrier();                                     // ... everybody's; and stage kt & 1 is free again
  }

  // ---- epilogue: D[i][j], i = 8 (q >> 2) + 4 (lane >> 5) + (q & 3) (row of W_old), j = lane & 31 (column):
  // out = acc * 2^-e_row * 2^-f_column (exact), 4-byte buffer stores, 128 contiguous bytes per row and instruction; rows
  // >= rows fall off the end of the descriptor, columns >= d get an out-of-range offset.  The row scales come through
  // LDS (the stages are free after the last barrier): a global load between the stores would wait for the stores
  // (one vmcnt on gfx9).
  float* srs = (float*)smem;
  if (tid < H2_BM) srs[tid] = m0 + tid < rows ? rs[m0 + tid] : 0.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)W_new, 0, (int)(rows * d * 4), 0x00020000);
#pragma unroll
  for (int b = 0; b < H2_TM; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int lm = (wm * H2_TM + b) * 32 + 8 * g + 4 * lh;
      const float4_t r4 = *(const float4_t*)(srs + lm);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned ro = (unsigned)((m0 + lm + k) * d * 4);
#pragma unroll
        for (int a = 0; a < H2_TN; ++a)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[b][a][4 * g + k] * r4[k] * cs[a]), orr,
                                                co[a] == H2_OOB ? H2_OOB : ro + co[a], 0, 0);
      }
    }
}

}  // namespace

bool apply_h2_fits(long rows, int d) {                               // 2 GB buffer descriptors
  return (size_t)rows * (size_t)d * 4 < 0x7fffffffUL && (size_t)d * d * 4 < 0x7fffffffUL;
}

// Workspace: the two f16 planes of W_old + its row scales + the column scales live in h->T (grown here), the planes of
// (I + Delta)^T in h->DeltaP.
int apply_h2_workspace(uce_ctx* h, long rows, int d, unsigned short** Ap, float** rs, float** cb) {
  const size_t wd = (size_t)rows * (size_t)d;
  const size_t rs_off = wd, cb_off = wd + (((size_t)rows + 4 + 63) & ~(size_t)63);
  const int rc = uce_ensure_T_floats(h, cb_off + (size_t)d);
  if (rc) return rc;
  *Ap = (unsigned short*)h->T;
  *rs = h->T + rs_off;
  *cb = h->T + cb_off;
  return UCE_OK;
}

// Any row count: a slab beyond the 2 GB buffer descriptors is walked in row chunks (rows are independent: every chunk sees the same
// planes of (I + Delta)^T and its own row scales - the same bits as one launch would give).  0: launched; < 0: error.
int launch_apply_h2(uce_ctx* h, const float* W_old, const float* DeltaT, float* W_new, long rows, int d, hipStream_t st) {
  if ((size_t)d * d * 4 >= 0x7fffffffUL) return UCE_EINVAL;
  long chunk = rows;
  if (!apply_h2_fits(rows, d)) {
    chunk = (long)((0x7fffffffUL - 1) / ((size_t)d * 4)) / H2_BM * H2_BM;       // whole row tiles under 2 GB
    if (chunk <= 0) return UCE_EINVAL;
  }
  unsigned short* Ap;
  float *rs, *cb;
  const int rc = apply_h2_workspace(h, chunk, d, &Ap, &rs, &cb);
  if (rc) return rc;
  unsigned short* Bp = h->DeltaP;
  // uce_edit: the planes of this W_old may already have been written by rider workgroups of the Cholesky launch (one-chunk slabs)
  const bool split_done = chunk == rows && h->h2_done_src == W_old && h->h2_done_rows == rows && h->h2_done_d == d;
  h->h2_done_src = nullptr;
  {
    UceProfScope ps(h, "k_split_h2d", st);
    hipLaunchKernelGGL(k_split_h2d, dim3((unsigned)((d + 3) / 4)), dim3(256), 0, st, DeltaT, Bp, Bp + (size_t)d * d, cb, d);
  }
  UCE_LAUNCH_CHECK();
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_apply_h2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  const int ncol = (d + H2_BN - 1) / H2_BN;
  for (long r0 = 0; r0 < rows; r0 += chunk) {
    const long rb = rows - r0 < chunk ? rows - r0 : chunk;
    const size_t wd = (size_t)rb * (size_t)d;
    const float* Wc = W_old + (size_t)r0 * d;
    if (!split_done) {
      UceProfScope ps(h, "k_split_h2", st);
      hipLaunchKernelGGL(k_split_h2, dim3((unsigned)((rb + 3) / 4)), dim3(256), 0, st, Wc, Ap, Ap + wd, rs, rb, d);
      UCE_LAUNCH_CHECK();
    }
    const long row_tiles = (rb + H2_BM - 1) / H2_BM;
    const long nwg = row_tiles * ncol;
    if (nwg > 0x7fffffffL) return UCE_EINVAL;
    UceProfScope ps(h, "k_apply_h2", st);
    hipLaunchKernelGGL(k_apply_h2, dim3((unsigned)nwg), dim3(512), 2 * H2_STAGE, st, Ap, Bp, rs, cb, W_new + (size_t)r0 * d, rb, d, ncol);
    UCE_LAUNCH_CHECK();
  }
  return UCE_OK;
}
