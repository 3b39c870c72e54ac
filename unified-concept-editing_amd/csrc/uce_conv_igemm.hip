// Implicit-GEMM 3x3 / stride 1 / pad 1 convolution of channels-last 16-bit activations (SURVEY.md section 8(f) row 3;
// reference: the Conv2d layers diffusers' ResnetBlock2D / Upsample2D / VAE decoder run under `pipe(...)`,
// evalscripts/generate-images-sd.py:37-42):
//
//      Y [N, H, W, Cout] = conv3x3(X [N, H, W, Cin], Wt [Cout, 3, 3, Cin]) (+ bias),   K = 9 * Cin contracted on the
//      bf16 / f16 matrix cores (v_mfma_f32_32x32x16), f32 accumulation.
//
// uce_im2col3x3_nhwc + a library GEMM writes a [pixels, 9*Cin] patch matrix (21 GB per U-Net call at batch 32, 150 GB
// per 16-image VAE decode) and reads it back.  Here the nine taps are gathered on the way INTO LDS: a k-chunk is one tap
// x 64 input channels, i.e. one contiguous 128-byte segment of a (shifted) source pixel per output pixel - 8 lanes x 16 B,
// predicated to zero outside the image - so the activation is read from L2 nine times and nothing else moves.
//   * workgroup = (64*WM) pixels x (64*WN) output channels, 4 waves, each a 64 x 64 quadrant = 2 x 2 MFMA tiles;
//   * operands staged through registers (next chunk's global loads in flight under this chunk's 16 MFMAs per wave), LDS
//     rows of 64 elements + 16 B pad (conflict-free ds_read_b128 fragments), two buffers, ONE barrier per chunk;
//   * swapped product  C^T = Wt X^T : a lane's accumulator column is ONE pixel and its registers are output channels in
//     groups of 4 consecutive -> bias add, conversion and 8-byte LDS writes per lane; the tile then leaves through LDS in
//     whole 16-byte row segments;
//   * `up`: X is the half-resolution tensor and the taps index its 2x nearest-neighbour upsampling (Upsample2D).
//   * block order: tiles that share the pixel rows (different output-channel tiles) sit on one XCD (speed only).
#include "uce_common.h"
#include <cstdlib>

namespace {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

constexpr int CG_BK = 64;                    // input channels per k-chunk
constexpr int CG_LD = CG_BK + 8;             // LDS row stride (elements): 144 B = an odd multiple of 16 B

template <bool F16>
__device__ __forceinline__ float16_t cg_mfma(uint4_t a, uint4_t b, float16_t c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ unsigned cg_pack2(float lo, float hi) {
  const float2_t v = {lo, hi};
  if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <bool F16>
__device__ __forceinline__ float cg_tof(unsigned short v) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __builtin_bit_cast(float, (unsigned)v << 16);
}

template <int WM, int WN>
constexpr size_t cg_smem() { return (size_t)2 * (64 * WM + 64 * WN) * CG_LD * 2; }

template <int WM, int WN, bool F16>
__global__ __launch_bounds__(256) void k_conv3x3_igemm(const unsigned short* __restrict__ X,
                                                       const unsigned short* __restrict__ Wt,
                                                       const unsigned short* __restrict__ bias,
                                                       unsigned short* __restrict__ Y, long M, int H, int W, int Cin,
                                                       int Cout, int up, int mtiles, int ntiles) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int PA = BM * 8 / 256, PB = BN * 8 / 256;      // 16-byte loads per thread per chunk (A: pixels, B: weights)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* As = (unsigned short*)smem;                          // [2][BM][CG_LD]
  unsigned short* Bs = As + 2 * BM * CG_LD;                            // [2][BN][CG_LD]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w % WM, wn = w / WM;
  const int li = lane & 31, lh = lane >> 5;

  // tile of this block: output-channel tiles of one pixel tile are consecutive on one XCD
  long tile = blockIdx.x;
  {
    const long T = (long)mtiles * ntiles;
    if ((T & 7) == 0) tile = (long)(blockIdx.x & 7) * (T >> 3) + (blockIdx.x >> 3);
  }
  const long m0 = (tile / ntiles) * BM;
  const int n0 = (int)(tile % ntiles) * BN;

  const int Hs = H >> up, Ws = W >> up;                                // source (stored) resolution
  const int cchunks = Cin / CG_BK, NK = 9 * cchunks;
  const long K = 9L * Cin;

  // ---- tile-invariant staging slots: A row = pixel (8 lanes x 16 B each), B row = output channel
  int a_y[PA], a_x[PA];
  long a_img[PA];                                                      // element offset of the pixel's image, -1: no pixel
  unsigned a_lds[PA], b_lds[PB];
  long b_src[PB];                                                      // element offset of the weight row (+ lane chunk), -1: none
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int idx = tid + 256 * p, row = idx >> 3, ch = idx & 7;
    const long m = m0 + row;
    a_lds[p] = (unsigned)(row * CG_LD + ch * 8);
    if (m < M) {
      const long img = m / ((long)H * W);
      const int rem = (int)(m - img * (long)H * W);
      a_y[p] = rem / W;
      a_x[p] = rem - a_y[p] * W;
      a_img[p] = img * (long)Hs * Ws * Cin + ch * 8;
    } else {
      a_y[p] = -4;                                                     // every tap lands outside the image
      a_x[p] = -4;
      a_img[p] = 0;
    }
  }
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int idx = tid + 256 * p, row = idx >> 3, ch = idx & 7;
    b_lds[p] = (unsigned)(row * CG_LD + ch * 8);
    b_src[p] = (n0 + row < Cout) ? (long)(n0 + row) * K + ch * 8 : -1;
  }

  // Buffer loads: a tap outside the image (or a weight row >= Cout) gets an offset beyond num_records and reads as
  // zero - no branch around any load, so the whole chunk is one basic block the compiler can interleave with the MFMAs.
  const long x_bytes = (M / ((long)H * W)) * (long)Hs * Ws * Cin * 2;
  const long w_bytes = (long)Cout * K * 2;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, x_bytes > 0x7fffffffL ? 0x7fffffff : (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, w_bytes > 0x7fffffffL ? 0x7fffffff : (int)w_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  uint4_t ra[PA], rb[PB];
  auto fetch = [&](int kc) {
    const int tap = kc / cchunks, c0 = (kc - tap * cchunks) * CG_BK;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int yy = a_y[p] + dy, xx = a_x[p] + dx;
      const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const long off = a_img[p] + ((long)(yy >> up) * Ws + (xx >> up)) * Cin + c0;
      ra[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)(off * 2) : OOB, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < PB; ++p)
      rb[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_src[p] >= 0 ? (unsigned)((b_src[p] + (long)kc * CG_BK) * 2) : OOB, 0, 0);
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PA; ++p) *(uint4_t*)(As + buf * BM * CG_LD + a_lds[p]) = ra[p];
#pragma unroll
    for (int p = 0; p < PB; ++p) *(uint4_t*)(Bs + buf * BN * CG_LD + b_lds[p]) = rb[p];
  };

  float16_t acc[2][2];                                                 // [output-channel tile][pixel tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  fetch(0);
  park(0);
  __syncthreads();
  for (int kc = 0; kc < NK; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < NK) fetch(kc + 1);                                    // in flight under the MFMAs
    const unsigned short* Ab = As + buf * BM * CG_LD + (wm * 64 + li) * CG_LD + 8 * lh;
    const unsigned short* Bb = Bs + buf * BN * CG_LD + (wn * 64 + li) * CG_LD + 8 * lh;
#pragma unroll
    for (int s = 0; s < CG_BK / 16; ++s) {
      const uint4_t p0 = *(const uint4_t*)(Ab + 16 * s);
      const uint4_t p1 = *(const uint4_t*)(Ab + 32 * CG_LD + 16 * s);
      const uint4_t c0 = *(const uint4_t*)(Bb + 16 * s);
      const uint4_t c1 = *(const uint4_t*)(Bb + 32 * CG_LD + 16 * s);
      acc[0][0] = cg_mfma<F16>(c0, p0, acc[0][0]);                     // rows = output channels, columns = pixels
      acc[0][1] = cg_mfma<F16>(c0, p1, acc[0][1]);
      acc[1][0] = cg_mfma<F16>(c1, p0, acc[1][0]);
      acc[1][1] = cg_mfma<F16>(c1, p1, acc[1][1]);
    }
    if (kc + 1 < NK) park(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: + bias, convert, through LDS as [pixel][BN] rows, then whole 16-byte segments to Y
  constexpr int YLD = BN + 8;                                          // elements
  unsigned short* Ys = (unsigned short*)smem;                          // [BM][YLD] (the operand buffers are dead)
  static_assert((size_t)BM * YLD * 2 <= cg_smem<WM, WN>(), "output tile must fit the operand LDS");
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int pix = wm * 64 + pt * 32 + li;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = wn * 64 + ct * 32 + 8 * g + 4 * lh;            // 4 consecutive output channels
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
          for (int q = 0; q < 4; ++q) bv[q] = (n0 + co + q < Cout) ? cg_tof<F16>(bias[n0 + co + q]) : 0.f;
        }
        const uint2_t o2 = {cg_pack2<F16>(acc[ct][pt][4 * g] + bv[0], acc[ct][pt][4 * g + 1] + bv[1]),
                            cg_pack2<F16>(acc[ct][pt][4 * g + 2] + bv[2], acc[ct][pt][4 * g + 3] + bv[3])};
        *(uint2_t*)(Ys + pix * YLD + co) = o2;
      }
    }
  __syncthreads();
  constexpr int OCH = BN / 8;                                          // 16-byte chunks per output row
  for (int e = tid; e < BM * OCH; e += 256) {
    const int row = e / OCH, ch = e - row * OCH;
    const long m = m0 + row;
    const int co = n0 + ch * 8;
    if (m < M && co < Cout) *(uint4_t*)(Y + m * Cout + co) = *(const uint4_t*)(Ys + row * YLD + ch * 8);
  }
}

template <int WM, int WN>
int launch_igemm(const void* x, const void* w, const void* bias, void* y, long M, int H, int W, int Cin, int Cout, int up,
                 int dtype, hipStream_t st) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  const long mtiles = (M + BM - 1) / BM;
  const int ntiles = (Cout + BN - 1) / BN;
  const long nwg = mtiles * ntiles;
  if (nwg > 0x7fffffffL || mtiles > 0x7fffffffL) return UCE_EINVAL;
  const size_t smem = cg_smem<WM, WN>();
  static PerDeviceOnce attr_once;
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_igemm<WM, WN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_igemm<WM, WN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL((k_conv3x3_igemm<WM, WN, true>), dim3((unsigned)nwg), dim3(256), smem, st, (const unsigned short*)x,
                       (const unsigned short*)w, (const unsigned short*)bias, (unsigned short*)y, M, H, W, Cin, Cout, up,
                       (int)mtiles, ntiles);
  else
    hipLaunchKernelGGL((k_conv3x3_igemm<WM, WN, false>), dim3((unsigned)nwg), dim3(256), smem, st, (const unsigned short*)x,
                       (const unsigned short*)w, (const unsigned short*)bias, (unsigned short*)y, M, H, W, Cin, Cout, up,
                       (int)mtiles, ntiles);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

}  // namespace

extern "C" int uce_conv3x3_nhwc_fwd(uce_handle_t h, const void* x, const void* w, const void* bias, void* y, int N, int H,
                                    int W, int Cin, int Cout, int upsample, int stride, const void* residual, int dtype,
                                    uce_stream_t stream) {
  if (!h || !x || !w || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UCE_EINVAL;
  UCE_ENTER(h);
  if (Cin % 32 || Cout % 8) return UCE_EINVAL;
  if (upsample && ((H | W) & 1)) return UCE_EINVAL;
  if ((stride != 1 && stride != 2) || (stride == 2 && upsample)) return UCE_EINVAL;
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  const long M = (long)N * H * W;
  // 32-bit buffer offsets: the stored activation and the weight must each stay below 2 GB (SD-1.4 at batch 32: 84 MB;
  // a 16-image VAE decode at 512 x 512 x 128: 1.07 GB); the host walks larger batches in chunks
  const long src_h = upsample ? H >> 1 : (long)H * stride, src_w = upsample ? W >> 1 : (long)W * stride;
  if ((long)N * src_h * src_w * Cin * 2 >= 0x7fffffffL || (long)Cout * 9 * Cin * 2 >= 0x7fffffffL) return UCE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  UceProfScope ps(h, "uce_conv3x3_nhwc_fwd", st);
  {
    // outputs that are multiples of 128 / 256 / 320 channels: the direct-to-LDS form (uce_conv_dma.hip), which also carries the
    // stride-2 taps and the residual epilogue
    int rc;
    // the one-wave-per-SIMD form first (UCE_CONV_W1; forced tile forms keep the 8-wave kernel)
    if (h->sw.conv_tile == 0 &&
        launch_conv_w1(x, w, bias, y, M, H, W, Cin, Cout, upsample ? 1 : 0, dtype, st, &rc, stride, residual, h->sw.conv_w1))
      return rc;
    if (launch_conv_dma(x, w, bias, y, M, H, W, Cin, Cout, upsample ? 1 : 0, dtype, st, &rc, stride, residual,
                                               h->sw.conv_tile, 1, h))
      return rc;
  }
  if (stride != 1 || residual) return UCE_ENOSYS;            // only the direct-to-LDS form has them
  if (Cin % CG_BK) return UCE_EINVAL;
  // 128 x 128 tiles (a ragged last output-channel tile is masked: Cout = 320 runs 3 tiles, 572 TF/s against 389 for the
  // 256 x 64 form); 256 x 64 only where a 128-wide tile would be at least half empty (Cout <= 64)
  const int rem = Cout % 128;
  if (Cout <= 64 && rem != 0) return launch_igemm<4, 1>(x, w, bias, y, M, H, W, Cin, Cout, upsample ? 1 : 0, dtype, st);
  return launch_igemm<2, 2>(x, w, bias, y, M, H, W, Cin, Cout, upsample ? 1 : 0, dtype, st);
}
