// 3x3 / stride 1 / pad 1 convolutions of the U-Net and VAE at inference as im2col + ONE library GEMM
// ("the rest of the U-Net step").  MIOpen's bf16 implicit-GEMM kernels run these at
// 290-560 TF/s on an MI355X; the same contraction as a plain [pixels, 9*Cin] x [9*Cin, Cout] GEMM runs at 1.0-1.3
// PF/s in hipBLASLt (tools/probe_conv.py), so materialising the patch matrix with a coalesced copy kernel and
// handing the GEMM to the library is 1.7-2.3x faster end to end.
//
//   k_im2col3x3 : x [N, H, W, C] (an NCHW tensor in torch.channels_last memory format), 16-bit elements ->
//                 cols [N*H*W, 9*C], column = (ky*3 + kx)*C + c  (the order of a channels_last Conv2d weight
//                 viewed as [Cout, 9*C]), zero outside the image.  Writes are fully contiguous 16-byte stores, each
//                 input pixel is read nine times out of L2.
#include "uce_common.h"

namespace {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));

// One workgroup per image row (n, y): thread j owns patch-row unit(s) j, j + 256, ... (unit = 16 bytes; tap t = j / C8,
// channel octet c = j % C8, computed once) and walks the row's pixels - no per-element index divisions, stores
// contiguous over j, loads contiguous within a tap.
// up = 1: x is the HALF-resolution tensor [N, H/2, W/2, C] and the patches are those of its 2x nearest-neighbour
// upsampling (diffusers' Upsample2D = interpolate + conv): the upsampled activation is never materialised.
__global__ __launch_bounds__(256) void k_im2col3x3(const uint4_t* __restrict__ x, uint4_t* __restrict__ cols, int H, int W,
                                                   int C8, int units /* = ceil(9*C8 / 256) */, int up) {
  const int row16 = 9 * C8;
  const int ny = blockIdx.x;                       // n * H + y
  const int y = ny % H;
  const long n = ny / H;
  const int Ws = W >> up;
  const uint4_t* xn = x + n * (H >> up) * Ws * C8;
  uint4_t* out = cols + (long)ny * W * row16;
  for (int k = 0; k < units; ++k) {
    const int j = threadIdx.x + 256 * k;
    if (j >= row16) break;
    const int t = j / C8, c = j - t * C8;
    const int yy = y + t / 3 - 1, dx = t % 3 - 1;
    const bool row_ok = yy >= 0 && yy < H;
    const uint4_t* src = xn + ((long)((row_ok ? yy : 0) >> up) * Ws) * C8 + c;
    for (int xw = 0; xw < W; ++xw) {
      const int xx = xw + dx;
      uint4_t v = {0u, 0u, 0u, 0u};
      if (row_ok && xx >= 0 && xx < W) v = src[(long)(xx >> up) * C8];
      __builtin_nontemporal_store(v, out + (long)xw * row16 + j);     // streamed: the GEMM reads it back once
    }
  }
}


// Flat variant: one 16-byte unit per thread-iteration over the whole patch matrix (grid-stride).  Its per-unit index
// arithmetic costs more than k_im2col3x3's, but for wide inputs (C > 1280, where one thread of the row kernel would
// own many units) it keeps more stores in flight: 4.3-4.9 TB/s there against 3.6-4.1 (tools/probe_im2col.py); it
// also takes the few-rows / long-rows cases (VAE decoder, tiny batches) where the row kernel cannot fill the chip.
__global__ __launch_bounds__(256) void k_im2col3x3_flat(const uint4_t* __restrict__ x, uint4_t* __restrict__ cols, int N,
                                                        int H, int W, int C8, int up) {
  const long total = (long)N * H * W * 9 * C8;
  const int row16 = 9 * C8;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / row16;
    const int r = (int)(e - p * row16);
    const int t = r / C8, c = r - t * C8;
    const int xw = (int)(p % W);
    const long q = p / W;
    const int y = (int)(q % H);
    const long n = q / H;
    const int yy = y + t / 3 - 1, xx = xw + t % 3 - 1;
    uint4_t v = {0u, 0u, 0u, 0u};
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = x[((n * (H >> up) + (yy >> up)) * (W >> up) + (xx >> up)) * C8 + c];
    __builtin_nontemporal_store(v, cols + e);
  }
}


// Patch matrix of a 3x3 / pad 1 convolution with FOUR input channels (the U-Net's and the VAE decoder's conv_in on the latents):
// x [N, H, W, 4] -> cols [N*H*W, 64], column (ky*3 + kx)*4 + c for the 36 real entries, zeros behind them (the GEMM kernel's
// k-tile is 32 elements).  One thread per pixel: nine 8-byte loads, eight 16-byte stores (one whole 128-byte row).
__global__ __launch_bounds__(256) void k_im2col3x3_c4(const uint2_t* __restrict__ x, uint4_t* __restrict__ cols, long M, int H,
                                                      int W) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= M) return;
  const int xw = (int)(p % W);
  const long q = p / W;
  const int y = (int)(q % H);
  const long n = q / H;
  unsigned d[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) d[i] = 0u;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = xw + t % 3 - 1;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const uint2_t v = x[(n * H + yy) * W + xx];
      d[2 * t] = v[0];
      d[2 * t + 1] = v[1];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) cols[p * 8 + j] = (uint4_t){d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3]};
}

}  // namespace

extern "C" int uce_im2col3x3_nhwc(uce_handle_t h, const void* x, void* cols, int N, int H, int W, int C, int upsample,
                                  uce_stream_t stream) {
  if (!h || !x || !cols || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return UCE_EINVAL;
  UCE_ENTER(h);
  if (upsample != 0 && upsample != 1) return UCE_EINVAL;
  if (upsample && ((H | W) & 1)) return UCE_EINVAL;
  const int up = upsample;
  const int C8 = C / 8;
  const uint4_t* xs = (const uint4_t*)x;
  uint4_t* cs = (uint4_t*)cols;
  // row kernel: needs enough rows to fill the chip and rows short enough that a thread's serial walk stays short
  // (VAE decoder convolutions at 128^2 .. 512^2 with one image per chunk go to the flat kernel)
  if (C <= 1280 && W <= 64 && (long)N * H >= 256) {
    hipLaunchKernelGGL(k_im2col3x3, dim3((unsigned)(N * H)), dim3(256), 0, (hipStream_t)stream, xs, cs, H, W, C8,
                       (9 * C8 + 255) / 256, up);
  } else {
    const long total = (long)N * H * W * 9 * C8;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_im2col3x3_flat, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, xs, cs, N, H, W, C8, up);
  }
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

extern "C" int uce_im2col3x3_c4(uce_handle_t h, const void* x, void* cols, int N, int H, int W, uce_stream_t stream) {
  if (!h || !x || !cols || N <= 0 || H <= 0 || W <= 0) return UCE_EINVAL;
  UCE_ENTER(h);
  const long M = (long)N * H * W;
  hipLaunchKernelGGL(k_im2col3x3_c4, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint2_t*)x,
                     (uint4_t*)cols, M, H, W);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
