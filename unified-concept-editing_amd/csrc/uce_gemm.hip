// Linear layers of the U-Net / VAE / text encoder at inference ("the rest of the U-Net step" under
// `pipe(...)` of evalscripts/generate-images-sd.py:37-42 - diffusers' Attention.to_q/to_k/to_v/to_out, FeedForward (GEGLU
// proj + out), Transformer2DModel.proj_in/proj_out, ResnetBlock2D.conv_shortcut, the time embedding):
//
//     Y [M, N] = X [M, K] Wt [N, K]^T  (+ bias[n])  (+ R [M, N])            16-bit in / out, f32 accumulate
//     GEGLU:  Y [M, N/2] = (hidden + b_h) * gelu(gate + b_g)                 with the rows of Wt interleaved per 32 (below)
//
// The direct-to-LDS structure of uce_conv_dma.hip with a single tap: workgroup = BM rows x BN columns, 8 waves, f32
// accumulators in registers (swapped product: a lane's accumulator column is ONE row of Y, its registers are 4 consecutive
// columns -> the bias / residual / GEGLU epilogue is per lane); a k-tile = 32 contraction elements = one 64-byte segment per
// row, moved by `buffer_load_dwordx4 ... lds` straight into LDS (rows >= M and weight rows >= N get an out-of-range offset
// and land as zeros), bank swizzle on the SOURCE address, ring of four stages with three k-tiles in flight, counted
// s_waitcnt vmcnt and raw s_barriers.  What the separate kernel buys over the GEMM library + element-wise passes:
//   * the epilogue operands: `x + to_out(o)`, `x + ff(y)`, `x + proj_out(h) + b` never exist as separate passes, and the
//     feed-forward's [M, 2*inner] projection is never written (hidden * gelu(gate) is formed on the accumulators);
//   * strided operands (ldx / ldy / ldr): q / k / v read from one packed projection, outputs written into slices.
// GEGLU weight layout (host: sd/unet.py geglu_interleave): row 32 t + r of Wt is hidden row 16 t + r for r < 16 and gate row
// 16 t + (r - 16) for r >= 16, so that every 32-row MFMA tile holds 16 outputs' hidden AND gate values and a lane's
// registers 0..7 (hidden) pair with its registers 8..15 (gate).
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
__device__ __forceinline__ float16_t gd_mfma(uint4_t a, uint4_t b, float16_t c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ unsigned gd_pack2(float lo, float hi) {
  const float2_t v = {lo, hi};
  if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <bool F16>
__device__ __forceinline__ float gd_tof(unsigned short v) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __builtin_bit_cast(float, (unsigned)v << 16);
}
template <bool F16>
__device__ __forceinline__ void gd_unpack4(uint2_t v, float* f) {
  f[0] = gd_tof<F16>((unsigned short)(v[0] & 0xffffu));
  f[1] = gd_tof<F16>((unsigned short)(v[0] >> 16));
  f[2] = gd_tof<F16>((unsigned short)(v[1] & 0xffffu));
  f[3] = gd_tof<F16>((unsigned short)(v[1] >> 16));
}

// waits until at most `n` of this wave's LDS-DMAs are outstanding (n = tiles left in flight x its DMAs per k-tile, 2 .. 5)
__device__ __forceinline__ void gd_wait_dma(int n) {
  if (n >= 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
  else if (n == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
  else if (n == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// WIDE: the tile leaves through LDS in whole rows (uce_epilogue.h; needs 16-byte aligned rows of y / residual and N % 8 == 0),
// else 8 bytes per lane straight from the accumulators
// NST: LDS stages of the ring (NST - 1 k-tiles in flight).  4 = one workgroup per CU with a deep ring; 2 / 3 = a ring shallow
// enough (<= 80 KB with the epilogue's slabs) that TWO workgroups share a CU - one's prologue / epilogue under the other's main
// loop, what the short contractions (K = 320: ten k-tiles per output tile) need.
// BK: contraction elements per k-tile.  32 = 64-byte row segments (16 rows per DMA instruction) with the deep ring; 64 = 128-byte
// segments - whole cache lines, 8 rows per instruction, half the barriers per contraction - with a two-stage ring (one k-tile of
// ~2500 MFMA cycles in flight covers the DMA latency).
// SK: split contraction (the few-tile regime: a layer whose output tiles cannot give every CU a workgroup - every layer of the
// U-Net at the CLI's one prompt per call, the 16 x 16 / 8 x 8 levels at any batch).  The grid is S x the tiles; workgroup (s, tile)
// contracts k-tiles [s NK / S, (s + 1) NK / S) and parks its f32 accumulators, in register order, in its slab of `skws`
// (write-through stores); a per-tile ticket (`sktick`, zero between launches: re-armed by the last arriver, so the launch can be
// captured and replayed) picks the workgroup that arrives LAST, which re-reads all S slabs past the L1 in slab order (bit-
// repeatable whatever the arrival order) and runs the one epilogue.  The hand-off is the sc1 form of uce_lowrank_riders.h.
// NW: waves per workgroup (8 in every form that ships).  A four-wave 128 x 128 form with 64 x 64 wave tiles (one fragment read per
// MFMA instead of two) was built for the few-tile regime and measured SLOWER than eight waves on 128 x 64 at every layer of the
// one-prompt U-Net (conv 320 -> 320 @ 64 x 64 x 2: 35.2 us against 33.0; 1280 @ 16 x 16: 38.9 against 32.7 -
// profiles/r05/sk_sweep_b1.jsonl): with one wave per SIMD and a two-stage ring nothing covers the DMA latency.  Not instantiated.
template <int WGM, int WGN, int TM, int TN, bool F16, bool GEGLU, bool WIDE, int NST, int BK = 32, bool SK = false, int NW = 8>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : ((NST == 4 || BK == 64) ? 2 : 4)) void k_gemm_dma(const unsigned short* __restrict__ X, long ldx,
                                                  const unsigned short* __restrict__ Wt,
                                                  const unsigned short* __restrict__ bias,
                                                  const unsigned short* __restrict__ R, long ldr,
                                                  unsigned short* __restrict__ Y, long ldy, long M, int N, int K,
                                                  int mtiles, int ntiles, int outf32,
                                                  const unsigned short* __restrict__ X2, long ldx2, int K1,
                                                  float* __restrict__ skws, unsigned* __restrict__ sktick, int S,
                                                  int scols, float cscale) {
  static_assert(WGM * WGN == NW && (NW == 8 || (NW == 4 && BK == 64 && NST == 2)), "waves (the four-wave form: two-stage ring only - its wait counts)");
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  static_assert(BM == 128 || BM == 256, "A image: whole DMA instructions per wave");
  static_assert(BK == 32 || BK == 64, "k-tile");
  constexpr int PPR = BK / 8;                                          // 16-byte pieces per row segment
  constexpr int RPW = 64 / PPR;                                        // rows per DMA wave instruction (16 / 8)
  constexpr int NA = BM / (RPW * NW);                                   // A wave instructions per wave and k-tile (1, 2 / 2, 4)
  constexpr int NB = BN / RPW;                                         // B wave instructions per k-tile (all waves)
  constexpr int NBJ = (NB + NW - 1) / NW;
  static_assert(NA <= 4 && NBJ <= 5, "staging registers");
  constexpr int STAGE = (BM + BN) * BK * 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w % WGM, wn = w / WGM;
  const int li = lane & 31, lh = lane >> 5;

  // the column tiles of one row tile are consecutive on one XCD (they share the rows of X)
  long tile = blockIdx.x;
  int ks = 0;                                                          // this workgroup's slice of the contraction (SK)
  {
    const long T = (long)mtiles * ntiles;
    if constexpr (SK) { ks = (int)(blockIdx.x / T); tile = blockIdx.x - ks * T; }
    if ((T & 7) == 0) tile = (long)(tile & 7) * (T >> 3) + (tile >> 3);
  }
  const long m0 = (tile / ntiles) * BM;
  const int n0 = (int)(tile % ntiles) * BN;
  const int NKall = K / BK;
  const int kb = SK ? (int)((long)ks * NKall / S) : 0;
  const int NK = SK ? (int)((long)(ks + 1) * NKall / S) : NKall;       // (one past the last k-tile of this workgroup)

  // ---- staging coordinates (k-tile invariant).  A wave instruction fills RPW rows x 2 BK bytes; lane = (row r, piece p).
  // Bank swizzle on the SOURCE piece: row R stores piece p ^ swz(R) at slot p - (R >> 2) & 3 for the 64-byte rows, (R >> 1) & 7 for
  // the 128-byte rows (16 consecutive rows of a b128 fragment read then fall on 16 distinct 16-byte bank slots)
  auto swz = [](int Rr) { return BK == 32 ? ((Rr >> 2) & 3) : ((Rr >> 1) & 7); };
  const int r = lane / PPR, p = lane % PPR;
  constexpr unsigned OOB = 0x80000000u;
  unsigned a_base[4];          // (fixed sizes: a lambda capturing an array of template-dependent size loses the kernel's host handle - clang, ROCm 7.2)
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int Rr = RPW * (NW * j + w) + r;
    const int c = p ^ swz(Rr);
    const long m = m0 + Rr;
    a_base[j] = (m < M) ? (unsigned)((m * ldx + c * 8) * 2) : OOB;
  }
  unsigned b_base[5];
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int g = NW * j + w;
    const int Rr = RPW * g + r;
    const int c = p ^ swz(Rr);
    b_base[j] = (g < NB && n0 + Rr < N) ? (unsigned)(((long)(n0 + Rr) * K + c * 8) * 2) : OOB;
  }
  int per = NA;
#pragma unroll
  for (int j = 0; j < NBJ; ++j) per += (NW * j + w < NB) ? 1 : 0;       // this wave's DMAs per k-tile
  // two-source contraction (X2 != nullptr): columns [0, K1) of a row come from X (row stride ldx), columns [K1, K) from X2 (row
  // stride ldx2) - the 1x1 shortcut convolution of an up block reading x and the skip connection in place (K1 % BK == 0)
  const long x_bytes = X2 ? ((M - 1) * ldx + K1) * 2 : ((M - 1) * ldx + K) * 2;
  const long x2_bytes = X2 ? ((M - 1) * ldx2 + (K - K1)) * 2 : 0;
  const long w_bytes = (long)N * K * 2;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc((void*)(X2 ? X2 : X), 0, (int)x2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, (int)w_bytes, 0x00020000);
  const int kt1 = X2 ? K1 / BK : 0x7fffffff;                           // first k-tile of the second source
  const unsigned dl2 = (unsigned)((ldx2 - ldx) * 2);                   // byte distance of the two sources' row strides (mod 2^32)

  auto stage = [&](int st, int kt) {
    unsigned char* sbase = smem + st * STAGE;
    const unsigned koff = (unsigned)(kt * BK * 2);
    if (kt >= kt1) {                                                   // (wave-uniform)
      const unsigned koff2 = koff - (unsigned)(K1 * 2);
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const unsigned mrow = (unsigned)(m0 + RPW * (NW * j + w) + r);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr2, (lds_void*)(sbase + (NW * j + w) * 1024), 16,
                                                 a_base[j] == OOB ? OOB : a_base[j] + mrow * dl2 + koff2, 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(sbase + (NW * j + w) * 1024), 16,
                                               a_base[j] == OOB ? OOB : a_base[j] + koff, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      if (NW * j + w < NB)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(sbase + BM * BK * 2 + (NW * j + w) * 1024), 16,
                                                 b_base[j] == OOB ? OOB : b_base[j] + koff, 0, 0, 0);
    }
  };

  float16_t acc[TN][TM];                                               // [column tile][row tile]
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
  constexpr int AHEAD = NST - 1;                                       // k-tiles in flight
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) stage(i, kb + i < last ? kb + i : last);
  gd_wait_dma((AHEAD - 1) * per);
  __builtin_amdgcn_s_barrier();
  int slot = 0, fill = AHEAD;                                          // ring positions of tile kt and of tile kt + AHEAD
  for (int kt = kb; kt < NK; ++kt) {
#if !defined(UCE_GEMM_ABLATE) || (UCE_GEMM_ABLATE != 3 && UCE_GEMM_ABLATE != 5)
    stage(fill, kt + AHEAD < last ? kt + AHEAD : last);
#endif
    const unsigned char* Ab = smem + slot * STAGE;
    const unsigned char* Bb = Ab + BM * BK * 2;
#if !defined(UCE_GEMM_ABLATE) || UCE_GEMM_ABLATE != 2
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int c = 2 * s + lh;
      uint4_t pf[TM], cf[TN];
#pragma unroll
      for (int b = 0; b < TM; ++b) pf[b] = *(const uint4_t*)(Ab + arow[b] * (2 * BK) + ((c ^ swz(arow[b])) << 4));
#pragma unroll
      for (int a = 0; a < TN; ++a) cf[a] = *(const uint4_t*)(Bb + brow[a] * (2 * BK) + ((c ^ swz(brow[a])) << 4));
#if defined(UCE_GEMM_ABLATE) && UCE_GEMM_ABLATE == 1
      // (measurement build: the fragment reads without the MFMAs)
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b][0] += __builtin_bit_cast(float, cf[a][0] ^ pf[b][1]);
#else
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = gd_mfma<F16>(cf[a], pf[b], acc[a][b]);   // rows = columns of Y, columns = rows of Y
#endif
    }
#endif
#if !defined(UCE_GEMM_ABLATE) || UCE_GEMM_ABLATE < 4
    gd_wait_dma((AHEAD - 1) * per);                                    // this wave's part of tile kt + 1 has landed
    __builtin_amdgcn_s_barrier();
#endif
    slot = slot + 1 == NST ? 0 : slot + 1;
    fill = fill + 1 == NST ? 0 : fill + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the ring's tail re-loads

  if constexpr (SK) {
    if (S > 1) {
      if (!uce_sk::reduce<TM, TN, BM * BN, 64 * NW>(acc, skws, sktick, tile, ks, S, smem, tid)) return;
    }
  }

  // uce_linear_colscale_fwd: columns [0, scols) of the product leave multiplied by cscale (scols a multiple of 32: whole MFMA
  // tiles; the f32 accumulator is scaled, so the element is rounded once) - the q columns of a packed q | k | v projection
  // carrying scale * log2(e) for the exp2-domain self-attention
  if (scols > 0) {                                                     // (uniform; the factor of a tile is a scalar select: straight-line code)
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const float f = (n0 + (wn * TN + a) * 32 < scols) ? cscale : 1.0f;
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] *= f;
    }
  }

  // ---- epilogue.  Register 4 g + i of tile a = column n0 + (wn TN + a) 32 + 8 g + 4 lh + i, row m0 + (wm TM + b) 32 + li
  if constexpr (WIDE) {
    __builtin_amdgcn_s_barrier();                                      // every wave's tail re-loads have landed: the ring is free
    constexpr int CH = (NST == 4 || TN < 2) ? TN : 2;                  // (shallow ring: the slabs must fit the smaller allocation)
    constexpr bool LEAN = !(NW == 4 || NST == 4 || BK == 64);          // (the launch bounds above: 4 waves per SIMD)
    uce_epi::store_rows<TM, TN, F16, GEGLU, CH, LEAN>(acc, smem + w * uce_epi::wave_bytes<CH, GEGLU>(), bias, R, ldr, Y, ldy,
                                                m0 + wm * TM * 32, n0 + wn * TN * 32, M, N, lane);
  } else if constexpr (GEGLU) {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int nh = n0 + (wn * TN + a) * 32 + 8 * g + 4 * lh;       // hidden rows of the interleaved weight; gates at + 16
        const int no = (n0 + (wn * TN + a) * 32) / 2 + 8 * g + 4 * lh; // output column
        float bh[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias && nh < N) {
          gd_unpack4<F16>(*(const uint2_t*)(bias + nh), bh);
          gd_unpack4<F16>(*(const uint2_t*)(bias + nh + 16), bg);
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          const long m = m0 + (wm * TM + b) * 32 + li;
          if (m < M && nh < N) {
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float hv = acc[a][b][4 * g + i] + bh[i];
              const float gv = acc[a][b][4 * (g + 2) + i] + bg[i];
              o[i] = hv * uce_epi::gelu_erf(gv);
            }
            const uint2_t o2 = {gd_pack2<F16>(o[0], o[1]), gd_pack2<F16>(o[2], o[3])};
            *(uint2_t*)(Y + m * ldy + no) = o2;
          }
        }
      }
  } else {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + (wn * TN + a) * 32 + 8 * g + 4 * lh;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias && n < N) gd_unpack4<F16>(*(const uint2_t*)(bias + n), bv);
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          const long m = m0 + (wm * TM + b) * 32 + li;
          if (m < M && n < N) {
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (R) gd_unpack4<F16>(*(const uint2_t*)(R + m * ldr + n), rv);
            if (outf32) {                                                // f32 result (attention scores ahead of a softmax)
              *(float4_t*)((float*)Y + m * ldy + n) = (float4_t){acc[a][b][4 * g] + bv[0] + rv[0], acc[a][b][4 * g + 1] + bv[1] + rv[1],
                                                                 acc[a][b][4 * g + 2] + bv[2] + rv[2], acc[a][b][4 * g + 3] + bv[3] + rv[3]};
              continue;
            }
            const uint2_t o = {gd_pack2<F16>(acc[a][b][4 * g] + bv[0] + rv[0], acc[a][b][4 * g + 1] + bv[1] + rv[1]),
                               gd_pack2<F16>(acc[a][b][4 * g + 2] + bv[2] + rv[2], acc[a][b][4 * g + 3] + bv[3] + rv[3])};
            *(uint2_t*)(Y + m * ldy + n) = o;
          }
        }
      }
  }
}

// the column scale of the call in flight on this thread (uce_linear_colscale_fwd sets it around linear_entry; every other entry
// point leaves {0, 1}): one kernel signature, no extra parameter through the tile-choice chain
struct ColScale { int cols; float scale; };
static thread_local ColScale g_colscale = {0, 1.f};

template <int WGM, int WGN, int TM, int TN, bool F16, bool GEGLU, bool WIDE, int NST, int BK = 32, bool SK = false, int NW = 8>
int launch_one(const void* x, long ldx, const void* w, const void* bias, const void* res, long ldr, void* y, long ldy, long M, int N,
               int K, hipStream_t st, int outf32 = 0, const void* x2 = nullptr, long ldx2 = 0,
               int K1 = 0, uce_ctx* h = nullptr, int S = 1) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  if (x2 && K1 % BK) return UCE_EINVAL;
  static_assert(NST == 4 || WIDE, "the shallow rings exist with the whole-row epilogue only");
  const long mtiles = (M + BM - 1) / BM;
  const int ntiles = (N + BN - 1) / BN;
  const long nwg = mtiles * ntiles * (SK ? S : 1);
  if (nwg > 0x7fffffffL || mtiles > 0x7fffffffL) return UCE_EINVAL;
  float* skws = nullptr;
  unsigned* sktick = nullptr;
  if constexpr (SK) {
    if (!h || S < 1 || S > K / BK) return UCE_EINVAL;
    if (S > 1) {
      const int rc = uce_ensure_sk(h, (size_t)mtiles * ntiles * S * BM * BN * sizeof(float), (size_t)mtiles * ntiles);
      if (rc != UCE_OK) return rc;
      skws = h->sk_ws;
      sktick = h->sk_tick;
    }
  }
  constexpr size_t ring = (size_t)NST * (BM + BN) * BK * 2;
  constexpr size_t slabs = WIDE ? (size_t)NW * uce_epi::wave_bytes<(NST == 4 || TN < 2) ? TN : 2, GEGLU>() : 0;
  constexpr size_t smem = ring > slabs ? ring : slabs;
  static_assert(NST == 4 || BK == 64 || smem <= 80 * 1024, "two workgroups per CU");
  static_assert(smem <= 160 * 1024, "LDS");
  static PerDeviceOnce attr_once;
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_gemm_dma<WGM, WGN, TM, TN, F16, GEGLU, WIDE, NST, BK, SK, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
    attr_once.commit(tok);
  }
  hipLaunchKernelGGL((k_gemm_dma<WGM, WGN, TM, TN, F16, GEGLU, WIDE, NST, BK, SK, NW>), dim3((unsigned)nwg), dim3(64 * NW), smem, st, (const unsigned short*)x,
                     ldx, (const unsigned short*)w, (const unsigned short*)bias, (const unsigned short*)res, ldr,
                     (unsigned short*)y, ldy, M, N, K, (int)mtiles, ntiles, outf32, (const unsigned short*)x2, ldx2, K1, skws, sktick, S,
                     g_colscale.cols, g_colscale.scale);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int WGM, int WGN, int TM, int TN, bool WIDE, int NST = 4, int BK = 32, bool SK = false, int NW = 8>
int launch_shape(const void* x, long ldx, const void* w, const void* bias, const void* res, long ldr, void* y, long ldy, long M, int N,
                 int K, int geglu, int dtype, hipStream_t st, int outf32 = 0, const void* x2 = nullptr,
                 long ldx2 = 0, int K1 = 0, uce_ctx* h = nullptr, int S = 1) {
  if (dtype == UCE_DTYPE_F16)
    return geglu ? launch_one<WGM, WGN, TM, TN, true, true, WIDE, NST, BK, SK, NW>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, st, 0, nullptr, 0, 0, h, S)
                 : launch_one<WGM, WGN, TM, TN, true, false, WIDE, NST, BK, SK, NW>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, st, outf32, x2, ldx2, K1, h, S);
  return geglu ? launch_one<WGM, WGN, TM, TN, false, true, WIDE, NST, BK, SK, NW>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, st, 0, nullptr, 0, 0, h, S)
               : launch_one<WGM, WGN, TM, TN, false, false, WIDE, NST, BK, SK, NW>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, st, outf32, x2, ldx2, K1, h, S);
}

// padded MFMA work of an N-wide output on BN-wide tiles, relative
inline long waste(int N, int BN) { return (long)((N + BN - 1) / BN) * BN; }

}  // namespace

// tile choice: the column width that wastes the fewest MFMAs on N (320 tiles SD's 320 / 640 / 1280 / 2560 ... exactly,
// 256 the VAE's and the text encoder's widths), and 128-row tiles when 256-row tiles would leave CUs without a workgroup
int launch_linear(const void* x, long ldx, const void* w, const void* bias, const void* res, long ldr, void* y, long ldy, long M, int N,
                  int K, int geglu, int dtype, hipStream_t st, int force_tile, int wide, int outf32,
                  const void* x2 = nullptr, long ldx2 = 0, int K1 = 0, uce_ctx* h = nullptr) {
  // whole-row epilogue where the rows allow 16-byte accesses (every layer of the U-Net); `wide` = 0 keeps the per-lane stores
  const int nout = geglu ? N / 2 : N;
  const bool wide_ok = wide && !outf32 && nout % 8 == 0 && ldy % 8 == 0 && !((uintptr_t)y & 15) && (!res || (ldr % 8 == 0 && !((uintptr_t)res & 15)));
  // Few-tile regime (h != null): a layer that cannot give 200 CUs a 128 x 320 tile takes 128 x 128 or 128 x 64 tiles (two or three
  // workgroups per CU: 48 KB of ring each), and below 200 of those the contraction is split S ways (uce_splitk.h) - the 64 x 64
  // level at one prompt per call (M = 8192: 64 row tiles), the 16 x 16 / 8 x 8 levels, the time MLP, the context projections.
  // UCE_GEMM_TILE = 9128064 / 9128128 pins a form (S by rule; 8... / 7...: its three- / two-stage ring).
  if (h && wide_ok && K % 64 == 0 && !(x2 && K1 % 64) && (!force_tile || force_tile / 1000000 >= 7)) {
    const long mt = (M + 127) / 128;
    const long t320 = mt * ((N + 319) / 320), t128 = mt * ((N + 127) / 128), t64 = mt * ((N + 63) / 64);
    int bnS = 0, nstS = 2;
    if (force_tile) { bnS = force_tile % 1000; nstS = force_tile / 1000000 - 5; }      // 7 / 8 / 9: ring of 2 / 3 / 4 stages
    else if (t320 < 200 && ((M + 255) / 256) * ((N + 255) / 256) < 200) bnS = t128 >= 400 ? 128 : 64;
#define UCE_GSK(TN, NSTV) \
  return launch_shape<4, 2, 1, TN, true, NSTV, 64, true>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, geglu, dtype, st, 0, x2, ldx2, K1, h, S);
    if (bnS == 128) {
      const int S = uce_sk::choose_split(t128, K / 64, 10, h->sw.sk_split);
      if (nstS == 2) { UCE_GSK(2, 2) } else if (nstS == 3) { UCE_GSK(2, 3) } else { UCE_GSK(2, 4) }
    }
    if (bnS == 64) {
      const int S = uce_sk::choose_split(t64, K / 64, 10, h->sw.sk_split);
      if (nstS == 2) { UCE_GSK(1, 2) } else if (nstS == 3) { UCE_GSK(1, 3) } else { UCE_GSK(1, 4) }
    }
#undef UCE_GSK
  }
  const long w320 = waste(N, 320), w256 = waste(N, 256), w128 = waste(N, 128);
  int bm = 256, bn = 320;
  if (w128 < w320 && w128 < w256) {
    bn = 128;
  } else {
    double best = -1.0;
    const int cand[3][2] = {{256, 320}, {256, 256}, {128, 320}};
    const double eff[3] = {1.0, 0.95, 0.8};
    for (int i = 0; i < 3; ++i) {
      if (cand[i][0] == 128 && M <= 128) continue;
      const long T = ((M + cand[i][0] - 1) / cand[i][0]) * ((N + cand[i][1] - 1) / cand[i][1]);
      const double cost = (double)((T + 255) / 256) * cand[i][0] * cand[i][1] / eff[i];
      if (best < 0.0 || cost < best) { best = cost; bm = cand[i][0]; bn = cand[i][1]; }
    }
  }
  int nst = 4;
  // Short contractions (K <= 320: ten k-tiles per output tile, the 64 x 64 level of the U-Net): prologue and epilogue are as
  // long as the main loop, so the shallow-ring forms that put TWO workgroups on a CU win (tools/probe_r04.py, M = 131072:
  // N = 320: 47 us against 55; GEGLU N = 2560: 336 against 381); longer contractions keep the deep ring
  // (at four times those rows - 64 prompts per call - the 128-byte k-tiles are ahead again: M = 524288, N = 320: 210 us against
  // 239; N = 2560: 256 x 256 tiles 1544 against 1594 for the three-stage 128 x 256 form and 1624 for 256 x 320)
  if (K <= 320 && M >= 65536 && M < 196608 && N <= 320) { nst = 2; bm = 128; bn = 320; }
  else if (K <= 320 && K % 64 == 0 && M >= 65536 && N >= 2560 && N % 256 == 0 && !(x2 && K1 % 64)) { nst = 64; bm = 256; bn = 256; }
  else if (K <= 320 && M >= 65536 && N >= 2560 && N % 256 == 0) { nst = 3; bm = 128; bn = 256; }
  // the GEGLU projections of the 32 x 32 / 16 x 16 / 8 x 8 levels (N = 5120 / 10240): 256 x 256 tiles 3-6 % ahead of 256 x 320
  // (CFG batch 256, profiles/r05/gemm_forms_b256.jsonl: N = 5120 1923 us against 1994; N = 10240 1639 against 1720, 404 against 412)
  else if (K % 64 == 0 && N >= 5120 && N % 256 == 0 && M >= 4096 && !(x2 && K1 % 64)) { nst = 64; bm = 256; bn = 256; }
  else if (K % 64 == 0 && bn != 128 && !(x2 && K1 % 64)) nst = 64;         // 128-byte k-tiles, two stages: 5-10 % ahead of the 64-byte ring wherever K allows
  if (force_tile > 0) { nst = force_tile >= 1000000 ? force_tile / 1000000 : 4; bm = (force_tile / 1000) % 1000; bn = force_tile % 1000; }
  if (!wide_ok) nst = 4;
  // 64-wide k-tiles, two stages (UCE_GEMM_TILE = 64256320 / 64256256 / 64128320; K % 64 == 0, whole-row epilogue)
  if (nst == 64 && K % 64 == 0 && wide_ok) {
    if (bm == 256 && bn == 320) return launch_shape<4, 2, 2, 5, true, 2, 64>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, geglu, dtype, st, 0, x2, ldx2, K1);
    if (bm == 256 && bn == 256) return launch_shape<2, 4, 4, 2, true, 2, 64>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, geglu, dtype, st, 0, x2, ldx2, K1);
    if (bm == 128 && bn == 320) return launch_shape<4, 2, 1, 5, true, 2, 64>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, geglu, dtype, st, 0, x2, ldx2, K1);
  }
  if (nst == 64) nst = 4;
  // two workgroups per CU (shallow ring): UCE_GEMM_TILE = 2128320 / 3128256 / 3256128
  if (nst == 2 && bm == 128 && bn == 320) return launch_shape<4, 2, 1, 5, true, 2>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, geglu, dtype, st, 0, x2, ldx2, K1);
  if (nst == 3 && bm == 128 && bn == 256) return launch_shape<2, 4, 2, 2, true, 3>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, geglu, dtype, st, 0, x2, ldx2, K1);
  if (nst == 3 && bm == 256 && bn == 128) return launch_shape<4, 2, 2, 2, true, 3>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, geglu, dtype, st, 0, x2, ldx2, K1);
#define UCE_GD(WGM, WGN, TM, TN)                                                                                                  \
  return wide_ok ? launch_shape<WGM, WGN, TM, TN, true>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, geglu, dtype, st, 0, x2, ldx2, K1) \
                 : launch_shape<WGM, WGN, TM, TN, false>(x, ldx, w, bias, res, ldr, y, ldy, M, N, K, geglu, dtype, st, outf32, x2, ldx2, K1);
  if (bm == 256 && bn == 320) { UCE_GD(4, 2, 2, 5) }
  if (bm == 256 && bn == 256) { UCE_GD(2, 4, 4, 2) }
  if (bm == 128 && bn == 320) { UCE_GD(4, 2, 1, 5) }
  if (bm == 128 && bn == 256) { UCE_GD(2, 4, 2, 2) }
  if (bm == 256 && bn == 128) { UCE_GD(4, 2, 2, 2) }
#undef UCE_GD
  return UCE_EINVAL;
}

static int linear_entry(uce_handle_t h, const void* x, long ldx, const void* w, const void* bias, const void* residual,
                        long ldr, void* y, long ldy, long M, int N, int K, int epilogue, int dtype,
                        uce_stream_t stream, const char* name, const void* x2 = nullptr, long ldx2 = 0, int K1 = 0) {
  if (!h || !x || !w || !y || M <= 0 || N <= 0 || K <= 0) return UCE_EINVAL;
  if (x2 && (K1 <= 0 || K1 >= K || K1 % 32 || ldx < K1 || ldx2 < K - K1 || ldx2 % 8 || ((uintptr_t)x2 & 15))) return UCE_EINVAL;
  UCE_ENTER(h);
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  if (epilogue != UCE_EPILOGUE_NONE && epilogue != UCE_EPILOGUE_GEGLU && epilogue != UCE_EPILOGUE_F32) return UCE_EINVAL;
  const int geglu = epilogue == UCE_EPILOGUE_GEGLU, outf32 = epilogue == UCE_EPILOGUE_F32;
  // 64-byte k-tiles, 8-byte epilogue accesses, 16-byte DMA pieces
  if (K % 32 || N % 4 || (!x2 && ldx < K) || ldx % 8 || ldy % 4 || (residual && (ldr % 4 || geglu))) return UCE_EINVAL;
  if (geglu && N % 32) return UCE_EINVAL;
  if (ldy < (geglu ? N / 2 : N) || (residual && ldr < N)) return UCE_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)w) & 15) || (((uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias) & 7)) return UCE_EINVAL;
  if (outf32 && ((uintptr_t)y & 15)) return UCE_EINVAL;
  // 32-bit buffer offsets: rows are walked in chunks whose X image stays under 2 GB
  const long ldmax = (x2 && ldx2 > ldx) ? ldx2 : ldx;
  const long max_rows = ((1L << 31) - 1 - 2L * K) / (2 * ldmax) + 1;
  if ((long)N * K * 2 >= (1L << 31)) return UCE_EINVAL;
  const int force = h->sw.gemm_tile;
  UceProfScope ps(h, name, (hipStream_t)stream);
  const long ybytes = outf32 ? 4 : 2;
  for (long m0 = 0; m0 < M; m0 += max_rows) {
    const long mb = (M - m0 < max_rows) ? M - m0 : max_rows;
    const int rc = launch_linear((const unsigned short*)x + m0 * ldx, ldx, w, bias,
                                 residual ? (const void*)((const unsigned short*)residual + m0 * ldr) : nullptr, ldr,
                                 (unsigned char*)y + m0 * ldy * ybytes, ldy, mb, N, K, geglu, dtype, (hipStream_t)stream, force,
                                 1, outf32,
                                 x2 ? (const void*)((const unsigned short*)x2 + m0 * ldx2) : nullptr, ldx2, K1, h);
    if (rc != UCE_OK) return rc;
  }
  return UCE_OK;
}

extern "C" int uce_linear_fwd(uce_handle_t h, const void* x, long ldx, const void* w, const void* bias, const void* residual,
                              long ldr, void* y, long ldy, long M, int N, int K, int epilogue, int dtype, uce_stream_t stream) {
  return linear_entry(h, x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, epilogue, dtype, stream, "uce_linear_fwd");
}

extern "C" int uce_linear_colscale_fwd(uce_handle_t h, const void* x, long ldx, const void* w, void* y, long ldy, long M, int N,
                                      int K, int scale_cols, float scale, int dtype, uce_stream_t stream) {
  if (scale_cols < 0 || scale_cols > N || scale_cols % 32) return UCE_EINVAL;
  g_colscale = {scale_cols, scale};
  const int rc = linear_entry(h, x, ldx, w, nullptr, nullptr, 0, y, ldy, M, N, K, UCE_EPILOGUE_NONE, dtype, stream, "uce_linear_colscale_fwd");
  g_colscale = {0, 1.f};
  return rc;
}

extern "C" int uce_linear_cat_fwd(uce_handle_t h, const void* x, long ldx, const void* x2, long ldx2, int K1, const void* w,
                                  const void* bias, const void* residual, long ldr, void* y, long ldy, long M, int N, int K,
                                  int epilogue, int dtype, uce_stream_t stream) {
  if (!x2) return UCE_EINVAL;
  return linear_entry(h, x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, epilogue, dtype, stream, "uce_linear_cat_fwd", x2,
                      ldx2, K1);
}
