// Implicit-GEMM 3x3 / pad 1 convolution, ONE wave per SIMD (the third implementation behind uce_conv3x3_nhwc_fwd; same contract
// as uce_conv_dma.hip, whose 8-wave form it replaces where it applies).  Why: the 8-wave 256 x 320 form reads 7 LDS fragments
// per 10 MFMAs and, with the DMA writes on top, keeps the CU's LDS pipe ~90 % busy at full MFMA rate - its matrix pipe measured
// 45-50 % busy.  The GEMM library's best kernel on this chip (rocprofv3: ..._MT256x256x64_MI16x16x1) is a 4-wave workgroup with
// 128 x 128 wave tiles; tools/ubench/gemm_w1.hip rebuilt that shape: 1.22-1.30 PF/s in the main loop against 1.05 for the 8-wave
// structure.  Here:
//   * workgroup = 256 pixels x BN output channels (BN = 320 / 256), 4 waves = 2 (pixels) x 2 (channels), wave tile 128 x BN/2 =
//     8 x TNW v_mfma_f32_16x16x32 tiles (TNW = 10 / 8): 18 / 16 fragment reads per 80 / 64 MFMAs (0.23 per MFMA against 0.7);
//   * accumulators: 64 tiles in the wave's 256 AGPRs, the last 16 of BN = 320 in VGPRs; the MFMAs are inline asm with the
//     accumulator as a read-write "a" / "v" operand - left to the compiler the accumulators travel between the register files
//     around every MFMA (v_accvgpr moves were half the instructions of the loop);
//   * k-tile = one tap x 64 input channels (128-byte pixel segments), `buffer_load ... lds` into a ring of two stages; the wave's
//     DMAs of the outgoing k-tile and the fragment reads of the next k-step are issued from INSIDE the MFMA stream (one per few
//     MFMAs, fenced), so the single wave of a SIMD never stops issuing matrix work to move data;
//   * whole-row epilogue: 16-pixel slabs through the wave's LDS region, 16-byte coalesced stores, residual read the same way.
// Cin % 64 == 0, Cout % BN == 0, stride 1 / 2, fused 2x nearest upsample, bias, residual.  DESIGN.md section 4.34.
#include "uce_common.h"
#include <type_traits>

namespace {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

template <bool F16>
__device__ __forceinline__ unsigned w1_pack2(float lo, float hi) {
  const float2_t v = {lo, hi};
  if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <bool F16>
__device__ __forceinline__ float w1_tof(unsigned short v) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __builtin_bit_cast(float, (unsigned)v << 16);
}
template <bool F16>
__device__ __forceinline__ unsigned w1_add2(unsigned a, unsigned b) {     // (a.lo + b.lo, a.hi + b.hi) in f32, one rounding
  return w1_pack2<F16>(w1_tof<F16>((unsigned short)(a & 0xffffu)) + w1_tof<F16>((unsigned short)(b & 0xffffu)),
                       w1_tof<F16>((unsigned short)(a >> 16)) + w1_tof<F16>((unsigned short)(b >> 16)));
}

// one MFMA with its accumulator pinned: tiles a < 8 (64 x 4 registers = the whole AGPR file) in AGPRs, the rest in VGPRs
template <bool F16, bool AGPR>
__device__ __forceinline__ void w1_mfma(float4_t& acc, const uint4_t& wfrag, const uint4_t& xfrag) {
  if constexpr (F16) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(wfrag), "v"(xfrag));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(wfrag), "v"(xfrag));
  } else {
    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(wfrag), "v"(xfrag));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wfrag), "v"(xfrag));
  }
}

constexpr int W1_BM = 256, W1_BK = 64;
constexpr int W1_ROWB = W1_BK * 2;                   // bytes per row segment (one cache line)
constexpr int W1_NA = W1_BM / 8 / 4;                 // A DMA instructions per wave and k-tile (8 rows x 128 B each)

template <int TNW>
constexpr size_t w1_smem() { return (size_t)2 * (W1_BM + 32 * TNW) * W1_ROWB; }

// Whole-row epilogue of k_conv3x3_w1.  Memory operations of a wave complete in order (one vmcnt for loads and stores), so a
// residual load issued behind a store waits for that store's acknowledgement: the row segments leave in sub-batches of SB pieces,
// and the residual pieces of sub-batch q + 1 are requested BEFORE the stores of sub-batch q go out (across the two passes as
// well) - the wait for them is a `vmcnt(stores of q)`.  Loads and stores are buffer operations on a descriptor of the wave's 128
// pixels clipped to M: no branch, nothing for the compiler to sink into a conditional block and serialise there (before: one
// residual load + vmcnt(0) + store per piece, 2 x BN / 16 round trips per wave with one wave per SIMD to hide them).
constexpr int W1_PB = 4;                             // pixel tiles per pass (two passes)
template <int TNW, bool F16, bool RES>
__device__ __forceinline__ void w1_epilogue(float4_t (&acc)[10][8], unsigned char* slab, const unsigned short* __restrict__ bias,
                                            const unsigned short* __restrict__ Rs, unsigned short* __restrict__ Y, long mrow,
                                            int ncol0, long M, int Cout, int lane) {
  constexpr int BN = 32 * TNW;
  constexpr int SROW = BN + 16;                      // bytes per slab row (BN / 2 channels of 2 bytes)
  constexpr int CPR = BN / 16;                       // 16-byte pieces per slab row
  constexpr int ITP = (16 * W1_PB * CPR) / 64;       // pieces per lane and pass (= CPR)
  constexpr int SB = ITP % 5 == 0 ? 5 : (ITP % 4 == 0 ? 4 : ITP);
  constexpr int NSB = ITP / SB;                      // sub-batches per pass
  const int l16 = lane & 15, lq = lane >> 4;
  const long left = M - mrow;
  const int rows = left <= 0 ? 0 : (left < 128 ? (int)left : 128);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(Y + mrow * Cout), 0, rows * Cout * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? Rs + mrow * Cout : Y), 0, RES ? rows * Cout * 2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, bias ? Cout * 2 : 0, 0x00020000);
  auto piece_off = [&](int q, int j) {               // byte offset (within the wave's rows) of piece j of sub-batch q
    const int pb = q / NSB, it = (q % NSB) * SB + j;
    const int idx = it * 64 + lane;
    const int row = idx / CPR, ch = idx - row * CPR;
    return (unsigned)(((16 * W1_PB * pb + row) * Cout + ncol0 + ch * 8) * 2);
  };
  uint4_t r[2][SB];
  if constexpr (RES) {
#pragma unroll
    for (int j = 0; j < SB; ++j) r[0][j] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(rr, piece_off(0, j), 0, 0));
    __builtin_amdgcn_sched_barrier(0);
  }
  float bv[10][4];
  {
    uint2_t b2[10];
#pragma unroll
    for (int a = 0; a < TNW; ++a)
      b2[a] = __builtin_bit_cast(uint2_t, __builtin_amdgcn_raw_buffer_load_b64(br, (unsigned)(ncol0 + 16 * a + 4 * lq) * 2u, 0, 0));
#pragma unroll
    for (int a = 0; a < TNW; ++a) {                  // (no bias: the empty descriptor reads 0)
      bv[a][0] = w1_tof<F16>((unsigned short)(b2[a][0] & 0xffffu));
      bv[a][1] = w1_tof<F16>((unsigned short)(b2[a][0] >> 16));
      bv[a][2] = w1_tof<F16>((unsigned short)(b2[a][1] & 0xffffu));
      bv[a][3] = w1_tof<F16>((unsigned short)(b2[a][1] >> 16));
    }
  }
#pragma unroll
  for (int pb = 0; pb < 8 / W1_PB; ++pb) {
#pragma unroll
    for (int bb = 0; bb < W1_PB; ++bb) {
      const int b = pb * W1_PB + bb;
#pragma unroll
      for (int a = 0; a < TNW; ++a)
        *(uint2_t*)(slab + (16 * bb + l16) * SROW + (16 * a + 4 * lq) * 2) =
            (uint2_t){w1_pack2<F16>(acc[a][b][0] + bv[a][0], acc[a][b][1] + bv[a][1]),
                      w1_pack2<F16>(acc[a][b][2] + bv[a][2], acc[a][b][3] + bv[a][3])};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
      const int q = pb * NSB + s;
      if constexpr (RES) {
        if (q + 1 < (8 / W1_PB) * NSB) {             // (resolved when the loops are unrolled)
#pragma unroll
          for (int j = 0; j < SB; ++j)
            r[(q + 1) & 1][j] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(rr, piece_off(q + 1, j), 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);           // (the scheduler, short of registers here, sinks every load to its use)
      }
      uint4_t v[SB];
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const int idx = (s * SB + j) * 64 + lane;
        const int row = idx / CPR, ch = idx - row * CPR;
        v[j] = *(const uint4_t*)(slab + row * SROW + ch * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        uint4_t o = v[j];
        if constexpr (RES) {
          const uint4_t r4 = r[q & 1][j];
          o = (uint4_t){w1_add2<F16>(o[0], r4[0]), w1_add2<F16>(o[1], r4[1]), w1_add2<F16>(o[2], r4[2]), w1_add2<F16>(o[3], r4[3])};
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, yr, piece_off(q, j), 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slab is read before the next pass overwrites it
  }
}

// TAPS: 9 = the 3 x 3 convolution.  (1 = its centre tap alone on a 1 x 1 "image" per row, i.e. a linear layer over contiguous rows:
// measured behind UCE_GEMM_W1 in round 4, level with or behind k_gemm_dma on every U-Net shape - no longer instantiated, HISTORY.md)
template <int TNW, bool F16, int TAPS>
__global__ __launch_bounds__(256, 1) void k_conv3x3_w1(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                       const unsigned short* __restrict__ bias, unsigned short* __restrict__ Y,
                                                       long M, int H, int W, int Cin, int Cout, int up, int mtiles, int ntiles,
                                                       int sd, const unsigned short* __restrict__ Rs, int tapin) {
  constexpr int BN = 32 * TNW;
  constexpr int NB = BN / 8 / 4;                     // B DMA instructions per wave and k-tile
  constexpr int PER = W1_NA + NB;
  constexpr int STAGE = (W1_BM + BN) * W1_ROWB;
  constexpr int NMF = TNW * 8, NFR = 8 + TNW;        // MFMAs and fragment reads per k-step of 32
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int l16 = lane & 15, lq = lane >> 4;

  long tile = blockIdx.x;                            // the output-channel tiles of one pixel tile are consecutive on one XCD
  {
    const long T = (long)mtiles * ntiles;
    if ((T & 7) == 0) tile = (long)(blockIdx.x & 7) * (T >> 3) + (blockIdx.x >> 3);
  }
  const long m0 = (tile / ntiles) * W1_BM;
  const int n0 = (int)(tile % ntiles) * BN;
  const int Hi = H * sd, Wi = W * sd;                // the image the taps index (before the >> up of the fused upsample)
  const int Hs = Hi >> up, Ws = Wi >> up;
  const int cch = Cin / W1_BK, NK = TAPS * cch;
  const long K = (long)TAPS * Cin;

  // ---- staging coordinates (k-tile invariant): a wave instruction fills 8 rows x 128 B, lane = (row r, 16-byte piece p); the bank
  // swizzle - row R keeps source piece p ^ ((R >> 1) & 7) in slot p - is applied to the SOURCE address
  const int r = lane >> 3, p = lane & 7;
  constexpr unsigned OOB = 0x80000000u;
  int a_yx[W1_NA];                                   // (y << 16) | x of the source pixel's centre tap; y = 0x7ff0: no pixel
  unsigned a_base[W1_NA];                            // byte offset of (image, channel piece)
#pragma unroll
  for (int j = 0; j < W1_NA; ++j) {
    const int R = 8 * (4 * j + w) + r;
    const int c = p ^ ((R >> 1) & 7);
    const long m = m0 + R;
    if (m < M) {
      const long img = m / ((long)H * W);
      const int rem = (int)(m - img * (long)H * W);
      const int y = rem / W, x = rem - y * W;
      a_yx[j] = ((y * sd) << 16) | (x * sd);
      a_base[j] = (unsigned)((img * (long)Hs * Ws * Cin + c * 8) * 2);
    } else {
      a_yx[j] = 0x7ff07ff0;
      a_base[j] = 0;
    }
  }
  unsigned b_base[10];                               // (fixed sizes: a lambda capturing an array of template-dependent size loses the kernel's host handle - clang, ROCm 7.2)
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int R = 8 * (4 * j + w) + r;
    const int c = p ^ ((R >> 1) & 7);
    b_base[j] = (n0 + R < Cout) ? (unsigned)(((long)(n0 + R) * K + c * 8) * 2) : OOB;
  }
  const long x_bytes = (M / ((long)H * W)) * (long)Hs * Ws * Cin * 2;
  const long w_bytes = (long)Cout * K * 2;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, (int)w_bytes, 0x00020000);

  // DMA number i (0 .. PER - 1: the A image's, then the B image's) of k-tile kt into stage st
  auto dma = [&](int st, int kt, int i) __attribute__((always_inline)) {
    unsigned char* sbase = smem + st * STAGE;
    if (i < W1_NA) {
      // k-tile order: tap-major (the weight's own [tap][Cin] order) or, for large activations, chunk-major - the nine taps of a
      // 64-channel chunk back to back, so that the pixel segments a tap shares with its neighbours are re-read while they are still
      // in the XCD's L2 (tap-major: a segment comes back cch k-tiles = 164 KB of A per CU later, 5 MB per XCD; chunk-major: 33 KB)
      const int chunk = tapin ? kt / 9 : 0;
      const int tap = TAPS == 1 ? 0 : (tapin ? kt - chunk * 9 : kt / cch);
      const int c0 = TAPS == 1 ? kt * W1_BK : (tapin ? chunk * W1_BK : (kt - tap * cch) * W1_BK);
      const int dy = TAPS == 1 ? 0 : tap / 3 - 1, dx = TAPS == 1 ? 0 : tap - (tap / 3) * 3 - 1;
      const int yy = (a_yx[i] >> 16) + dy, xx = (a_yx[i] & 0xffff) + dx;
      const bool ok = (unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi;
      const unsigned off = a_base[i] + (unsigned)((((yy >> up) * Ws + (xx >> up)) * Cin + c0) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(sbase + (4 * i + w) * 1024), 16, ok ? off : OOB, 0, 0, 0);
    } else {
      const int j = i - W1_NA;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(sbase + W1_BM * W1_ROWB + (4 * j + w) * 1024), 16,
                                               b_base[j] == OOB ? OOB : b_base[j] + (unsigned)((tapin ? (kt % 9) * Cin + (kt / 9) * W1_BK : kt * W1_BK) * 2), 0, 0, 0);
    }
  };

  float4_t acc[10][8];                               // [channel tile][pixel tile]: 4 consecutive channels of one pixel per lane (TNW rows used)
#pragma unroll
  for (int a = 0; a < TNW; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row R of either image keeps the swizzle key (R >> 1) & 7 = (l16 >> 1) & 7 for every tile of the wave (the
  // tiles start at multiples of 16 rows), so a k-step has ONE piece offset and the tiles are immediate offsets of 2048 bytes
  const int key = (l16 >> 1) & 7;
  const int arow = (wm * 128 + l16) * W1_ROWB;
  const int brow = W1_BM * W1_ROWB + (wn * (BN / 2) + l16) * W1_ROWB;
  uint4_t xf[2][8], wf[2][10];
  auto frag = [&](int buf, const unsigned char* sb, int s, int f) __attribute__((always_inline)) {
    const int po = ((4 * s + lq) ^ key) << 4;
    if (f < 8) xf[buf][f] = *(const uint4_t*)(sb + arow + po + f * 16 * W1_ROWB);
    else wf[buf][f - 8] = *(const uint4_t*)(sb + brow + po + (f - 8) * 16 * W1_ROWB);
  };

  const int last = NK - 1;
#pragma unroll
  for (int i = 0; i < PER; ++i) dma(0, 0, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int f = 0; f < NFR; ++f) frag(0, smem, 0, f);
  int slot = 0;
  for (int kt = 0; kt < NK; ++kt) {
    const int ktn = kt < last ? kt + 1 : last;       // the k-tile this iteration sends off, into the other stage (past the end: a re-load)
    const int fill = slot ^ 1;
    const unsigned char* sb = smem + slot * STAGE;
    // the two k-steps of the tile, S a compile-time constant (the fragment buffers are register arrays)
    auto kstep = [&](auto s_c) __attribute__((always_inline)) {
      constexpr int S = decltype(s_c)::value;
      constexpr int cur = S, nxt = S ^ 1;
#pragma unroll
      for (int a = 0; a < TNW; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          if (a < 8) w1_mfma<F16, true>(acc[a][b], wf[cur][a], xf[cur][b]);     // (resolved when the loops are unrolled)
          else w1_mfma<F16, false>(acc[a][b], wf[cur][a], xf[cur][b]);
          const int i = a * 8 + b;
          // fragment f of the second k-step goes out behind MFMA (f + 1) NMF / NFR - 1 of the first (evenly spread, in the order the
          // next step needs them: X fragments, then W fragments); the wave's DMAs of the outgoing k-tile likewise
          if constexpr (S == 0) {
#pragma unroll
            for (int f = 0; f < NFR; ++f)
              if (i == ((f + 1) * NMF) / NFR - 1) {
                frag(nxt, sb, 1, f);
                __builtin_amdgcn_sched_barrier(0);
              }
#pragma unroll
            for (int g = 0; g < PER; ++g)
              if (i == ((g + 1) * NMF) / PER - 2) {
                dma(fill, ktn, g);
                __builtin_amdgcn_sched_barrier(0);
              }
          } else {
            // The tile's barrier stands HOLD MFMAs before the end of its second k-step: every fragment of this stage was read during
            // the first step, so from here on the wave works out of registers - it waits for its DMAs of k-tile kt + 1, passes the
            // barrier and reads the next tile's first fragments UNDER the last MFMAs of this one (read behind the last MFMA they
            // cost a bare LDS round trip plus 18 issue slots per tile with no other wave on the SIMD to fill them).
            constexpr int HOLD = NFR;                // one fragment read behind each of the last NFR MFMAs
            if (i == NMF - HOLD - 1) {
              asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              __builtin_amdgcn_sched_barrier(0);
            }
            if (i >= NMF - HOLD) {
              if (kt < last) frag(nxt, smem + fill * STAGE, 0, i - (NMF - HOLD));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
    };
    kstep(std::integral_constant<int, 0>{});
    kstep(std::integral_constant<int, 1>{});
    slot = fill;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // every wave is past its last MFMA on LDS-fed registers: the ring is free

  // ---- epilogue: 64-pixel slabs (four pixel tiles per pass, two passes) through this wave's LDS region.  acc[a][b] = channels
  // n0 + wn BN/2 + 16 a + 4 lq + {0..3} of pixel m0 + wm 128 + 16 b + l16.  Slab row = one pixel, BN/2 channels (+ 16 bytes: rows
  // two bank groups apart); the rows leave as 16-byte pieces, consecutive lanes on consecutive pieces of a row.
  unsigned char* slab = smem + w * (16 * W1_PB) * (BN + 16);
  const int ncol0 = n0 + wn * (BN / 2);
  const long mrow = m0 + wm * 128;                   // the wave's 128 pixels (uniform)
  if (Rs) w1_epilogue<TNW, F16, true>(acc, slab, bias, Rs, Y, mrow, ncol0, M, Cout, lane);
  else w1_epilogue<TNW, F16, false>(acc, slab, bias, Rs, Y, mrow, ncol0, M, Cout, lane);
}

template <int TNW, int TAPS>
int launch_w1(const void* x, const void* w, const void* bias, void* y, long M, int H, int W, int Cin, int Cout, int up, int dtype,
              hipStream_t st, int sd, const void* res) {
  constexpr int BN = 32 * TNW;
  const long mtiles = (M + W1_BM - 1) / W1_BM;
  const int ntiles = Cout / BN;
  const long nwg = mtiles * ntiles;
  if (nwg > 0x7fffffffL || mtiles > 0x7fffffffL) return UCE_EINVAL;
  const size_t smem = w1_smem<TNW>();
  static PerDeviceOnce attr_once;
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_w1<TNW, false, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_w1<TNW, true, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL((k_conv3x3_w1<TNW, true, TAPS>), dim3((unsigned)nwg), dim3(256), smem, st, (const unsigned short*)x, (const unsigned short*)w,
                       (const unsigned short*)bias, (unsigned short*)y, M, H, W, Cin, Cout, up, (int)mtiles, ntiles, sd,
                       (const unsigned short*)res, TAPS == 9 ? conv_tap_inner(H, W, Cin, up, sd) : 0);
  else
    hipLaunchKernelGGL((k_conv3x3_w1<TNW, false, TAPS>), dim3((unsigned)nwg), dim3(256), smem, st, (const unsigned short*)x, (const unsigned short*)w,
                       (const unsigned short*)bias, (unsigned short*)y, M, H, W, Cin, Cout, up, (int)mtiles, ntiles, sd,
                       (const unsigned short*)res, TAPS == 9 ? conv_tap_inner(H, W, Cin, up, sd) : 0);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

}  // namespace

// 0: this form does not take the shape; 1: launched (*rc = status).  mode (UCE_CONV_W1): 1 = by rule (enough 256-pixel tiles to give
// every CU one), 2 = wherever the shape allows.
int launch_conv_w1(const void* x, const void* w, const void* bias, void* y, long M, int H, int W, int Cin, int Cout, int up, int dtype,
                   hipStream_t st, int* rc, int sd, const void* res, int mode) {
  *rc = UCE_OK;
  if (mode <= 0 || Cin % W1_BK || Cout % 8) return 0;
  if (((uintptr_t)y & 15) || ((uintptr_t)res & 15) || ((uintptr_t)bias & 7)) return 0;
  const int bn = Cout % 320 == 0 ? 320 : Cout % 256 == 0 ? 256 : 0;
  if (!bn) return 0;
  const long tiles = ((M + W1_BM - 1) / W1_BM) * (Cout / bn);
  // measured at 64 prompts per call (tools/probe_r04.py conv, UCE_PROBE_B=128, one box; us, 8-wave form | this one): 320 -> 320 @ 64^2
  // 903 | 853, 640 -> 640 @ 32^2 819 | 757, 1280 -> 1280 @ 16^2 780 | 704, 2560 -> 1280 1532 | 1410, 1920 -> 640 @ 32^2 2364 | 2175,
  // 960 -> 320 @ 64^2 2484 | 2337, stride 2: 320 -> 320 234 | 231, 640 -> 640 205 | 186; with fewer than 256 tiles (the 8 x 8 level: 246 | 309,
  // 1280 -> 1280 stride 2: 247 | 311) the 8-wave form stays
  if (mode == 1 && tiles < 256) return 0;
  *rc = bn == 320 ? launch_w1<10, 9>(x, w, bias, y, M, H, W, Cin, Cout, up, dtype, st, sd, res)
                  : launch_w1<8, 9>(x, w, bias, y, M, H, W, Cin, Cout, up, dtype, st, sd, res);
  return 1;
}
