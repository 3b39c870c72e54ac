// 3x3 / stride 1 / pad 1 convolutions of the U-Net and VAE at inference as im2col + ONE library GEMM
// ("the rest of the U-Net step").  MIOpen's bf16 implicit-GEMM kernels run these at
// 290-560 TF/s on an MI355X; the same contraction as a plain [pixels, 9*Cin] x [9*Cin, Cout] GEMM runs at 1.0-1.3
// PF/s in hipBLASLt (tools/probe_conv.py), so materialising the patch matrix with a coalesced copy kernel and
// handing the GEMM to the library is 1.7-2.3x faster end to end.
//
//   k_im2col3x3 : x [N, H, W, C] (an NCHW tensor in torch.channels_last memory format), 16-bit elements ->
//                 cols [N*H*W, 9*C], column = (ky*3 + kx)*C + c  (the order of a channels_last Conv2d weight
//                 viewed as [Cout, 9*C]), zero outside the image.  One 16-byte element per thread-iteration;
//                 writes are fully contiguous, each input pixel is read nine times out of L2.
#include "uce_common.h"

namespace {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_im2col3x3(const uint4_t* __restrict__ x, uint4_t* __restrict__ cols, int N, int H,
                                                   int W, int C8) {
  const long total = (long)N * H * W * 9 * C8;
  const int row16 = 9 * C8;                       // 16-byte units per output row
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
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = x[((n * H + yy) * W + xx) * C8 + c];
    __builtin_nontemporal_store(v, cols + e);     // streamed: the GEMM reads it back once
  }
}

}  // namespace

extern "C" int uce_im2col3x3_nhwc(uce_handle_t h, const void* x, void* cols, int N, int H, int W, int C,
                                  uce_stream_t stream) {
  if (!h || !x || !cols || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return UCE_EINVAL;
  const long total = (long)N * H * W * 9 * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_im2col3x3, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4_t*)x,
                     (uint4_t*)cols, N, H, W, C / 8);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
