// Implicit-GEMM 3x3 / stride 1 / pad 1 convolution, direct-to-LDS form (the second implementation behind
// uce_conv3x3_nhwc_fwd; same contract as uce_conv_igemm.hip, which remains the kernel for narrow outputs):
//
//   * workgroup = 256 pixels x BN output channels, 8 waves, f32 accumulators in registers:
//       BN = 256: waves 2 (pixels) x 4 (channels), wave tile 128 x 64 = 4 x 2 v_mfma_f32_32x32x16 tiles
//       BN = 320: waves 4 x 2, wave tile 64 x 160 = 2 x 5 tiles  (the U-Net's 320 / 640 / 960 / 1280 / 1920-channel layers
//                 tile exactly; with 256-wide tiles a 320-channel layer would waste 37 % of its MFMAs)
//       BN = 128: waves 4 x 2, wave tile 64 x 64 = 2 x 2 tiles   (the VAE decoder's 128-channel layers)
//     6 - 7 fragment reads per 8 - 10 MFMAs (the 128 x 128 kernel: 1 per MFMA);
//   * k-tile = one tap x 32 input channels = one 64-byte segment of a (shifted) source pixel per output pixel, moved by
//     `buffer_load_dwordx4 ... lds` STRAIGHT into LDS (16 rows x 64 B per wave instruction): no staging registers, no
//     ds_write pass; a tap outside the image, a pixel >= M or a weight row >= Cout gets an out-of-range buffer offset
//     and lands as zeros;
//   * the LDS image of such an instruction is lane-linear, so the bank swizzle is applied to the SOURCE: the 16-byte
//     piece p of row R holds source piece p ^ ((R >> 2) & 3) - the 16 rows a b128 fragment read touches for one piece
//     index fall on 16 distinct bank slots; the fragment reads apply the same XOR;
//   * a ring of FOUR LDS stages with three k-tiles in flight: the loads of tile t + 3 are issued when tile t starts
//     computing and are waited for - counted s_waitcnt vmcnt, never 0 - only at the end of tile t + 1, behind RAW
//     s_barriers (a __syncthreads would make the compiler drain the DMA queue).  Every wave waits for its OWN loads of
//     tile t + 1 and then passes the barrier, so after the barrier the whole tile is in LDS (the wait comes one barrier
//     before the first read); a stage is re-filled only after the barrier that follows its last read.
// Measured against the register-staged 128 x 128 kernel in tools/probe_igemm.py; DESIGN.md section 4.
#include "uce_common.h"
#include "uce_epilogue.h"
#include "uce_splitk.h"

namespace {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;



template <bool F16>
__device__ __forceinline__ float16_t cd_mfma(uint4_t a, uint4_t b, float16_t c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ unsigned cd_pack2(float lo, float hi) {
  const float2_t v = {lo, hi};
  if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <bool F16>
__device__ __forceinline__ float cd_tof(unsigned short v) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __builtin_bit_cast(float, (unsigned)v << 16);
}

template <int BM, int BN, int BK, int NSTP = 0>
constexpr size_t cd_smem() { return (size_t)(NSTP ? NSTP : BK == 32 ? 4 : 2) * (BM + BN) * BK * 2; }

// waits until at most `2 * per` of this wave's LDS-DMAs are outstanding (per = its DMAs per k-tile: 2 .. 5)
__device__ __forceinline__ void cd_wait_two_tiles(int per) {
  if (per == 5) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
  else if (per == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  else if (per == 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void cd_wait_one_tile(int per) {
  if (per == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
  else if (per == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else if (per == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
}

// BK: input channels per k-tile (one tap x BK channels).  32 = 64-byte pixel segments and a ring of four stages; 64 = 128-byte
// segments (whole cache lines, half the barriers) and a ring of two - 5-10 % ahead on uce_gemm.hip's shapes wherever Cin % 64 == 0.
// SK: split contraction for the few-tile regime (uce_splitk.h): grid = S x tiles, workgroup (s, tile) walks k-tiles
// [s NK / S, (s + 1) NK / S) of the 9 taps x channel chunks; the last arriver of a tile sums the S slabs and runs the epilogue.
// NSTP: ring stages (0: four for the 64-byte k-tiles, two for the 128-byte ones); the few-tile forms run ONE workgroup per CU
// and keep three 128-byte k-tiles in flight (NSTP = 4)
// NW: waves per workgroup (8 in every form that ships; see uce_gemm.hip for the four-wave form that was measured and left out)
template <int WGM, int WGN, int TM, int TN, bool F16, bool WIDE, int BK = 32, bool SK = false, int NSTP = 0, int NW = 8>
__global__ __launch_bounds__(64 * NW, 2) void k_conv3x3_dma(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                     const unsigned short* __restrict__ bias, unsigned short* __restrict__ Y,
                                                     long M, int H, int W, int Cin, int Cout, int up, int mtiles, int ntiles,
                                                     int sd, const unsigned short* __restrict__ Rs,
                                                     float* __restrict__ skws, unsigned* __restrict__ sktick, int S, int tapin) {
  // sd: stride (1, or 2 = diffusers' Downsample2D: output pixel (y, x) reads source pixels (2y + dy, 2x + dx) of a [N, 2H, 2W, Cin]
  // tensor, pad 1); Rs: optional residual [M, Cout] added in the epilogue (the ResnetBlock2D's `x + conv2(.) + b`)
  static_assert(WGM * WGN == NW && (NW == 8 || (NW == 4 && BK == 64 && NSTP == 2)), "waves (the four-wave form: two-stage ring only - its wait counts)");
  constexpr int CD_BM = 32 * TM * WGM;
  static_assert(CD_BM == 128 || CD_BM == 256, "128 or 256 pixels");
  constexpr int CD_BK = BK;
  constexpr int NST = NSTP ? NSTP : BK == 32 ? 4 : 2;                 // ring stages
  constexpr int PPR = BK / 8;                                         // 16-byte pieces per pixel segment
  constexpr int RPW = 64 / PPR;                                       // rows per DMA wave instruction (16 / 8)
  constexpr int NA = CD_BM / (RPW * NW);                               // A wave instructions per wave and k-tile
  constexpr int BN = 32 * TN * WGN;
  constexpr int STAGE = (CD_BM + BN) * CD_BK * 2;
  constexpr int NB = BN / RPW;                                        // wave instructions per B image
  constexpr int NBJ = (NB + NW - 1) / NW;
  static_assert(NA <= 4 && NBJ <= 5 && NST * STAGE <= 160 * 1024, "staging");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w % WGM, wn = w / WGM;
  const int li = lane & 31, lh = lane >> 5;

  // tile of this block: the output-channel tiles of one pixel tile are consecutive on one XCD (they share the pixels)
  long tile = blockIdx.x;
  int ks = 0;                                                          // this workgroup's slice of the contraction (SK)
  {
    const long T = (long)mtiles * ntiles;
    if constexpr (SK) { ks = (int)(blockIdx.x / T); tile = blockIdx.x - ks * T; }
    if ((T & 7) == 0) tile = (long)(tile & 7) * (T >> 3) + (tile >> 3);
  }
  const long m0 = (tile / ntiles) * CD_BM;
  const int n0 = (int)(tile % ntiles) * BN;
  const int Hi = H * sd, Wi = W * sd;                                 // the image the taps index (before the >> up of the fused upsample)
  const int Hs = Hi >> up, Ws = Wi >> up;
  const int cch = Cin / CD_BK, NKall = 9 * cch;
  const int kb = SK ? (int)((long)ks * NKall / S) : 0;
  const int NK = SK ? (int)((long)(ks + 1) * NKall / S) : NKall;       // (one past the last k-tile of this workgroup)
  const long K = 9L * Cin;

  // ---- staging coordinates (k-tile invariant).  A wave instruction fills RPW rows x 2 BK bytes; lane = (row r, piece p); the
  // bank swizzle is applied to the SOURCE piece: (R >> 2) & 3 for 64-byte rows, (R >> 1) & 7 for 128-byte rows
  auto swz = [](int Rr) { return BK == 32 ? ((Rr >> 2) & 3) : ((Rr >> 1) & 7); };
  const int r = lane / PPR, p = lane % PPR;
  constexpr unsigned OOB = 0x80000000u;
  int a_y[4], a_x[4];                                                  // (fixed sizes: a lambda capturing an array of template-dependent size loses the kernel's host handle - clang, ROCm 7.2)
  unsigned a_base[4];                                                  // byte offset of (image, channel piece); OOB: no pixel
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int R = RPW * (NW * j + w) + r;
    const int c = p ^ swz(R);
    const long m = m0 + R;
    if (m < M) {
      const long img = m / ((long)H * W);
      const int rem = (int)(m - img * (long)H * W);
      a_y[j] = rem / W;
      a_x[j] = (rem - a_y[j] * W) * sd;
      a_y[j] *= sd;
      a_base[j] = (unsigned)((img * (long)Hs * Ws * Cin + c * 8) * 2);
    } else {
      a_y[j] = -4;                                                     // every tap lands outside the image
      a_x[j] = -4;
      a_base[j] = 0;
    }
  }
  unsigned b_base[5];
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int g = NW * j + w;
    const int R = RPW * g + r;
    const int c = p ^ swz(R);
    b_base[j] = (g < NB && n0 + R < Cout) ? (unsigned)(((long)(n0 + R) * K + c * 8) * 2) : OOB;
  }
  int per = NA;                                                        // this wave's DMAs per k-tile
#pragma unroll
  for (int j = 0; j < NBJ; ++j) per += (NW * j + w < NB) ? 1 : 0;
  const long x_bytes = (M / ((long)H * W)) * (long)Hs * Ws * Cin * 2;
  const long w_bytes = (long)Cout * K * 2;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, (int)w_bytes, 0x00020000);

  auto stage = [&](int st, int kt) {
    // k-tile order: tap-major, or chunk-major for large activations (uce_common.h: conv_tap_inner)
    const int chunk = tapin ? kt / 9 : 0;
    const int tap = tapin ? kt - chunk * 9 : kt / cch;
    const int c0 = tapin ? chunk * CD_BK : (kt - tap * cch) * CD_BK;
    const unsigned wk = (unsigned)((tapin ? tap * Cin + c0 : kt * CD_BK) * 2);
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    unsigned char* sbase = smem + st * STAGE;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int yy = a_y[j] + dy, xx = a_x[j] + dx;
      const bool ok = (unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi;
      const unsigned off = a_base[j] + (unsigned)((((yy >> up) * Ws + (xx >> up)) * Cin + c0) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(sbase + (NW * j + w) * 1024), 16, ok ? off : OOB, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      if (NW * j + w < NB)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(sbase + CD_BM * CD_BK * 2 + (NW * j + w) * 1024), 16,
                                                 b_base[j] == OOB ? OOB : b_base[j] + wk, 0, 0, 0);
    }
  };
  // waits until this wave's DMAs of every k-tile but the last (NST - 2) issued have landed
  auto wait_ring = [&]() {
    if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (NST == 3) cd_wait_one_tile(per);
    else cd_wait_two_tiles(per);
  };

  float16_t acc[TN][TM];                                               // [channel tile][pixel tile]
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  int arow[TM], brow[TN];
#pragma unroll
  for (int b = 0; b < TM; ++b) arow[b] = (wm * TM + b) * 32 + li;
#pragma unroll
  for (int a = 0; a < TN; ++a) brow[a] = (wn * TN + a) * 32 + li;

  const int last = NK - 1;                                             // (past the last tile the ring re-loads it: constant counts)
  constexpr int AHEAD = NST - 1;
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) stage(i, kb + i < last ? kb + i : last);
  wait_ring();
  __builtin_amdgcn_s_barrier();
  int slot = 0, fill = AHEAD;
  for (int kt = kb; kt < NK; ++kt) {
    stage(fill, kt + AHEAD < last ? kt + AHEAD : last);
    const unsigned char* Ab = smem + slot * STAGE;
    const unsigned char* Bb = Ab + CD_BM * CD_BK * 2;
#pragma unroll
    for (int s = 0; s < CD_BK / 16; ++s) {
      const int c = 2 * s + lh;
      uint4_t pf[TM], cf[TN];
#pragma unroll
      for (int b = 0; b < TM; ++b) pf[b] = *(const uint4_t*)(Ab + arow[b] * (2 * CD_BK) + ((c ^ swz(arow[b])) << 4));
#pragma unroll
      for (int a = 0; a < TN; ++a) cf[a] = *(const uint4_t*)(Bb + brow[a] * (2 * CD_BK) + ((c ^ swz(brow[a])) << 4));
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = cd_mfma<F16>(cf[a], pf[b], acc[a][b]);   // rows = channels, columns = pixels
    }
    wait_ring();                                                       // this wave's part of tile kt + 1 has landed
    __builtin_amdgcn_s_barrier();
    slot = slot + 1 == NST ? 0 : slot + 1;
    fill = fill + 1 == NST ? 0 : fill + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the ring's tail re-loads

  if constexpr (SK) {
    if (S > 1) {
      if (!uce_sk::reduce<TM, TN, CD_BM * BN, 64 * NW>(acc, skws, sktick, tile, ks, S, smem, tid)) return;
    }
  }

  // ---- epilogue: + bias (+ residual), convert; whole rows through LDS (uce_epilogue.h) or 8-byte stores (a lane holds 4
  //      consecutive output channels of one pixel)
  if constexpr (WIDE) {
    __builtin_amdgcn_s_barrier();                                      // every wave's tail re-loads have landed: the ring is free
    uce_epi::store_rows<TM, TN, F16, false>(acc, smem + w * uce_epi::wave_bytes<TN, false>(), bias, Rs, (long)Cout, Y, (long)Cout,
                                            m0 + wm * TM * 32, n0 + wn * TN * 32, M, Cout, lane);
    return;
  }
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + (wn * TN + a) * 32 + 8 * g + 4 * lh;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias && n < Cout) {
        const uint2_t b2 = *(const uint2_t*)(bias + n);
        bv[0] = cd_tof<F16>((unsigned short)(b2[0] & 0xffffu));
        bv[1] = cd_tof<F16>((unsigned short)(b2[0] >> 16));
        bv[2] = cd_tof<F16>((unsigned short)(b2[1] & 0xffffu));
        bv[3] = cd_tof<F16>((unsigned short)(b2[1] >> 16));
      }
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const long m = m0 + (wm * TM + b) * 32 + li;
        if (m < M && n < Cout) {
          float rv[4] = {0.f, 0.f, 0.f, 0.f};
          if (Rs) {
            const uint2_t r2 = *(const uint2_t*)(Rs + m * Cout + n);
            rv[0] = cd_tof<F16>((unsigned short)(r2[0] & 0xffffu));
            rv[1] = cd_tof<F16>((unsigned short)(r2[0] >> 16));
            rv[2] = cd_tof<F16>((unsigned short)(r2[1] & 0xffffu));
            rv[3] = cd_tof<F16>((unsigned short)(r2[1] >> 16));
          }
          const uint2_t o = {cd_pack2<F16>(acc[a][b][4 * g] + bv[0] + rv[0], acc[a][b][4 * g + 1] + bv[1] + rv[1]),
                             cd_pack2<F16>(acc[a][b][4 * g + 2] + bv[2] + rv[2], acc[a][b][4 * g + 3] + bv[3] + rv[3])};
          *(uint2_t*)(Y + m * Cout + n) = o;
        }
      }
    }
}

template <int WGM, int WGN, int TM, int TN, bool WIDE, int BK = 32, bool SK = false, int NSTP = 0, int NW = 8>
int launch_dma(const void* x, const void* w, const void* bias, void* y, long M, int H, int W, int Cin, int Cout, int up, int dtype,
               hipStream_t st, int sd, const void* res, uce_ctx* h = nullptr, int S = 1) {
  constexpr int BM = 32 * TM * WGM;
  constexpr int BN = 32 * TN * WGN;
  const long mtiles = (M + BM - 1) / BM;
  const int ntiles = (Cout + BN - 1) / BN;
  const long nwg = mtiles * ntiles * (SK ? S : 1);
  if (nwg > 0x7fffffffL || mtiles > 0x7fffffffL) return UCE_EINVAL;
  float* skws = nullptr;
  unsigned* sktick = nullptr;
  if constexpr (SK) {
    if (!h || S < 1 || S > 9 * (Cin / BK)) return UCE_EINVAL;
    if (S > 1) {
      const int rc = uce_ensure_sk(h, (size_t)mtiles * ntiles * S * BM * BN * sizeof(float), (size_t)mtiles * ntiles);
      if (rc != UCE_OK) return rc;
      skws = h->sk_ws;
      sktick = h->sk_tick;
    }
  }
  // (the whole-row epilogue parks 8 wave slabs in the drained ring: the small tiles' ring must hold them)
  constexpr size_t slabs = WIDE ? (size_t)NW * uce_epi::wave_bytes<TN, false>() : 0;
  const size_t smem = cd_smem<BM, BN, BK, NSTP>() > slabs ? cd_smem<BM, BN, BK, NSTP>() : slabs;
  static PerDeviceOnce attr_once;
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_dma<WGM, WGN, TM, TN, false, WIDE, BK, SK, NSTP, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_dma<WGM, WGN, TM, TN, true, WIDE, BK, SK, NSTP, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL((k_conv3x3_dma<WGM, WGN, TM, TN, true, WIDE, BK, SK, NSTP, NW>), dim3((unsigned)nwg), dim3(64 * NW), smem, st, (const unsigned short*)x,
                       (const unsigned short*)w, (const unsigned short*)bias, (unsigned short*)y, M, H, W, Cin, Cout, up,
                       (int)mtiles, ntiles, sd, (const unsigned short*)res, skws, sktick, S, conv_tap_inner(H, W, Cin, up, sd));
  else
    hipLaunchKernelGGL((k_conv3x3_dma<WGM, WGN, TM, TN, false, WIDE, BK, SK, NSTP, NW>), dim3((unsigned)nwg), dim3(64 * NW), smem, st, (const unsigned short*)x,
                       (const unsigned short*)w, (const unsigned short*)bias, (unsigned short*)y, M, H, W, Cin, Cout, up,
                       (int)mtiles, ntiles, sd, (const unsigned short*)res, skws, sktick, S, conv_tap_inner(H, W, Cin, up, sd));
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

}  // namespace

// 0: this form does not take the shape (the caller falls back to the 128 x 128 register-staged kernel); 1: launched; < 0: error.
// Tile: output channels in 320- / 256- / 128-wide tiles (whichever tiles Cout exactly); 256 pixels per workgroup while that
// still gives every CU a workgroup, else 128 pixels, else (the 8 x 8 layers: 2048 pixels at the generation batch) 128 x 128.
// `force` (UCE_CONV_TILE = 1000 * BM + BN) pins one for measurements.
int launch_conv_dma(const void* x, const void* w, const void* bias, void* y, long M, int H, int W, int Cin, int Cout, int up,
                    int dtype, hipStream_t st, int* rc, int sd, const void* res, int force, int wide, uce_ctx* h) {
  *rc = UCE_OK;
  if (Cin % 32 || Cout % 4) return 0;
  // Few-tile regime (h != null): a layer without 200 tiles of 128 pixels x its widest exact tile - the whole U-Net at one prompt per
  // call, the 16 x 16 / 8 x 8 levels at any batch - takes 128 x 128 or 128 x 64 tiles, two or three workgroups per CU, and below 200
  // of those the 9 taps x channel chunks are split S ways (uce_splitk.h).  Any Cout % 8 == 0 (ragged last tile masked).
  // UCE_CONV_TILE = 9128064 / 9128128 pins a form (S by rule; 8... / 7...: its three- / two-stage ring).
  if (h && wide && Cin % 64 == 0 && Cout % 8 == 0 && !((uintptr_t)y & 15) && !((uintptr_t)res & 15) && (!force || force / 1000000 >= 7)) {
    const long mt = (M + 127) / 128;
    const int bw = Cout % 320 == 0 ? 320 : Cout % 256 == 0 ? 256 : 128;
    const long tw = mt * ((Cout + bw - 1) / bw), t128 = mt * ((Cout + 127) / 128), t64 = mt * ((Cout + 63) / 64);
    int bnS = 0, nstS = 2;
    const bool exact = Cout % 128 == 0 || Cout % 320 == 0;              // a wide tile divides Cout (the forms below this block)
    if (force) { bnS = force % 1000; nstS = force / 1000000 - 5; }                      // 7 / 8 / 9: ring of 2 / 3 / 4 stages
    else if (tw < 200) bnS = t128 >= 400 ? 128 : 64;
    else if (!exact && (sd != 1 || res)) bnS = Cout > 64 ? 128 : 64;   // (only this kernel has the stride-2 taps and the residual)
#define UCE_CSK(TN, NSTV) \
  { *rc = launch_dma<4, 2, 1, TN, true, 64, true, NSTV>(x, w, bias, y, M, H, W, Cin, Cout, up, dtype, st, sd, res, h, S); return 1; }
    if (bnS == 128) {
      const int S = uce_sk::choose_split(t128, 9 * (Cin / 64), 10, h->sw.sk_split);
      if (nstS == 2) UCE_CSK(2, 2) else if (nstS == 3) UCE_CSK(2, 3) else UCE_CSK(2, 4)
    }
    if (bnS == 64) {
      const int S = uce_sk::choose_split(t64, 9 * (Cin / 64), 10, h->sw.sk_split);
      if (nstS == 2) UCE_CSK(1, 2) else if (nstS == 3) UCE_CSK(1, 3) else UCE_CSK(1, 4)
    }
#undef UCE_CSK
  }
  int bn = Cout % 320 == 0 ? 320 : Cout % 256 == 0 ? 256 : Cout % 128 == 0 ? 128 : 0;
  if (!bn) return 0;
  const long t256 = ((M + 255) / 256) * (Cout / bn), t128 = ((M + 127) / 128) * (Cout / bn);
  int bm = 256;
  if (t256 < 200) bm = 128;
  if (bm == 128 && t128 < 200 && Cout % 128 == 0) bn = 128;
  int bk = (Cin % 64 == 0 && bn != 128) ? 64 : 32;     // 128-byte k-tiles wherever the channel count allows (the 128-wide tiles keep 64 bytes)
  if (force > 0) { bk = force >= 1000000 ? 64 : 32; bm = (force / 1000) % 1000; bn = force % 1000; }
  if (Cout % bn || Cin % bk) return 0;
  const bool wide_ok = wide && Cout % 8 == 0 && !((uintptr_t)y & 15) && !((uintptr_t)res & 15);
  if (bk == 64 && wide_ok) {                             // UCE_CONV_TILE = 64256320 / 64256256 / 64128320 pins one
#define UCE_CD64(WGM, WGN, TM, TN) { *rc = launch_dma<WGM, WGN, TM, TN, true, 64>(x, w, bias, y, M, H, W, Cin, Cout, up, dtype, st, sd, res); return 1; }
    if (bm == 256 && bn == 320) UCE_CD64(4, 2, 2, 5)
    if (bm == 256 && bn == 256) UCE_CD64(2, 4, 4, 2)
    if (bm == 128 && bn == 320) UCE_CD64(4, 2, 1, 5)
    if (bm == 128 && bn == 256) UCE_CD64(2, 4, 2, 2)
#undef UCE_CD64
  }
#define UCE_CD(WGM, WGN, TM, TN)                                                                                             \
  {                                                                                                                          \
    *rc = wide_ok ? launch_dma<WGM, WGN, TM, TN, true>(x, w, bias, y, M, H, W, Cin, Cout, up, dtype, st, sd, res)            \
                  : launch_dma<WGM, WGN, TM, TN, false>(x, w, bias, y, M, H, W, Cin, Cout, up, dtype, st, sd, res);          \
    return 1;                                                                                                                \
  }
  if (bm == 256 && bn == 320) UCE_CD(4, 2, 2, 5)
  if (bm == 256 && bn == 256) UCE_CD(2, 4, 4, 2)
  if (bm == 256 && bn == 128) UCE_CD(4, 2, 2, 2)
  if (bm == 128 && bn == 320) UCE_CD(4, 2, 1, 5)
  if (bm == 128 && bn == 256) UCE_CD(2, 4, 2, 2)
  if (bm == 128 && bn == 128) UCE_CD(4, 2, 1, 2)
#undef UCE_CD
  return 0;
}
