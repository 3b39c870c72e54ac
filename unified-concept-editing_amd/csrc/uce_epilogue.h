// Epilogue shared by the direct-to-LDS GEMM / convolution kernels (uce_gemm.hip, uce_conv_dma.hip): the swapped product leaves
// a lane with ONE output row (pixel) and, per 32 x 32 MFMA tile, four runs of 4 consecutive columns - stored straight from
// the registers that is 8 bytes per lane at a row stride: every store instruction touches 32 different lines and the
// output leaves as partial lines (measured: a 131 072 x 2560 projection wrote its 671 MB at 1.3 TB/s).  Here each wave parks
// its 32-row slab in its own LDS region (the staging ring is free after the main loop) and reads it back as whole rows:
// 16-byte pieces, consecutive lanes on consecutive pieces of a row, so the tile leaves in full lines - and the residual
// operand is read the same way.
//   MODE 0: y = acc + bias (+ residual)          MODE 1 (GEGLU): y = (acc_h + b_h) * gelu(acc_g + b_g), half the columns
// Wave-private: no workgroup barrier inside (DS operations of one wave execute in order); the caller has passed a barrier
// after every wave's last read of the ring AND after every wave's own `s_waitcnt vmcnt(0)` (the ring's tail re-loads).
#pragma once
#include "uce_common.h"

namespace uce_epi {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <bool F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const float2_t v = {lo, hi};
  if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <bool F16>
__device__ __forceinline__ float tof(unsigned short v) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __builtin_bit_cast(float, (unsigned)v << 16);
}
template <bool F16>
__device__ __forceinline__ void unpack4(uint2_t v, float* f) {
  f[0] = tof<F16>((unsigned short)(v[0] & 0xffffu));
  f[1] = tof<F16>((unsigned short)(v[0] >> 16));
  f[2] = tof<F16>((unsigned short)(v[1] & 0xffffu));
  f[3] = tof<F16>((unsigned short)(v[1] >> 16));
}
template <bool F16>
__device__ __forceinline__ unsigned add2(unsigned a, unsigned b) {       // (a.lo + b.lo, a.hi + b.hi) in f32, one rounding
  return pack2<F16>(tof<F16>((unsigned short)(a & 0xffffu)) + tof<F16>((unsigned short)(b & 0xffffu)),
                    tof<F16>((unsigned short)(a >> 16)) + tof<F16>((unsigned short)(b >> 16)));
}

// erf to 1.5e-7 absolute (Abramowitz & Stegun 7.1.26) on one v_rcp, one v_exp and nine FMAs - the library erff is a branchy
// ~30-instruction polynomial, and the GEGLU epilogue evaluates it once per output element with nothing to overlap it.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return __builtin_copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float g) { return 0.5f * g * (1.0f + erf_as(g * 0.70710678118654752f)); }

// CH: 32-column MFMA tiles parked at once (the row of a slab is CH * 64 bytes, CH * 32 for GEGLU).  CH = TN parks the wave's
// whole width (fewest passes); CH = 2 keeps the region at 4.5 KB per wave for the two-workgroups-per-CU tile forms.
template <int CH, bool GEGLU>
constexpr int row_bytes() { return (GEGLU ? CH * 32 : CH * 64) + 16; }        // + 16: rows two bank groups apart
template <int CH, bool GEGLU>
constexpr int wave_bytes() { return 32 * row_bytes<CH, GEGLU>(); }

// one chunk: tiles CA .. CA + CW - 1 of row tile b
template <int TM, int TN, bool F16, bool GEGLU, int CH, int CA>
__device__ __forceinline__ void store_chunk(float16_t (&acc)[TN][TM], int b, unsigned char* lds, const unsigned short* __restrict__ bias,
                                            const unsigned short* __restrict__ res, long ldr, unsigned short* __restrict__ Y,
                                            long ldy, long mrow0, int ncol0, long M, int N, int lane) {
  constexpr int CW = (TN - CA < CH) ? TN - CA : CH;
  constexpr int ROWB = row_bytes<CH, GEGLU>();
  constexpr int CPR = (GEGLU ? CW * 32 : CW * 64) / 16;                   // 16-byte pieces per row of the chunk
  constexpr int ITER = (32 * CPR) / 64;                                  // 32 rows x CPR pieces over 64 lanes
  const int li = lane & 31, lh = lane >> 5;
  const int ocol0 = (GEGLU ? ncol0 / 2 : ncol0) + (GEGLU ? CA * 16 : CA * 32);
  const int NO = GEGLU ? N / 2 : N;
  // ---- park the chunk: 8 bytes per lane per (tile, register group)
#pragma unroll
  for (int aa = 0; aa < CW; ++aa) {
    const int a = CA + aa;
    if constexpr (GEGLU) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int nh = ncol0 + a * 32 + 8 * g + 4 * lh;                  // hidden columns; their gates sit 16 further
        float bh[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias && nh < N) {
          unpack4<F16>(*(const uint2_t*)(bias + nh), bh);
          unpack4<F16>(*(const uint2_t*)(bias + nh + 16), bg);
        }
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (acc[a][b][4 * g + i] + bh[i]) * gelu_erf(acc[a][b][4 * (g + 2) + i] + bg[i]);
        *(uint2_t*)(lds + li * ROWB + (aa * 16 + 8 * g + 4 * lh) * 2) = (uint2_t){pack2<F16>(o[0], o[1]), pack2<F16>(o[2], o[3])};
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = ncol0 + a * 32 + 8 * g + 4 * lh;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias && n < N) unpack4<F16>(*(const uint2_t*)(bias + n), bv);
        *(uint2_t*)(lds + li * ROWB + (aa * 32 + 8 * g + 4 * lh) * 2) =
            (uint2_t){pack2<F16>(acc[a][b][4 * g] + bv[0], acc[a][b][4 * g + 1] + bv[1]),
                      pack2<F16>(acc[a][b][4 * g + 2] + bv[2], acc[a][b][4 * g + 3] + bv[3])};
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // ---- whole row segments out: lane = (row, 16-byte piece), consecutive lanes on consecutive pieces
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / CPR, ch = idx - row * CPR;
    uint4_t v = *(const uint4_t*)(lds + row * ROWB + ch * 16);
    const long m = mrow0 + b * 32 + row;
    const int n = ocol0 + ch * 8;
    if (m < M && n < NO) {
      if constexpr (!GEGLU) {
        if (res) {
          const uint4_t r4 = *(const uint4_t*)(res + m * ldr + n);
          v = (uint4_t){add2<F16>(v[0], r4[0]), add2<F16>(v[1], r4[1]), add2<F16>(v[2], r4[2]), add2<F16>(v[3], r4[3])};
        }
      }
      *(uint4_t*)(Y + m * ldy + n) = v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // the chunk is read before the next one overwrites it
  if constexpr (CA + CH < TN)
    store_chunk<TM, TN, F16, GEGLU, CH, CA + CH>(acc, b, lds, bias, res, ldr, Y, ldy, mrow0, ncol0, M, N, lane);
}

// acc[a][b][4 g + i]: column 32 a + 8 g + 4 lh + i of the wave's TN x 32 columns, row 32 b + li of its TM x 32 rows.
//   lds     this wave's region (wave_bytes<CH, GEGLU>() bytes)
//   bias    indexed by the (interleaved, for GEGLU) column; may be null.   res / ldr: residual rows (MODE 0), may be null
//   mrow0   first row of the wave's slab (global), ncol0 its first column in the index space of `bias` (a multiple of 32)
//   M, N    bounds in that space (GEGLU: N counts the interleaved columns; the output has N / 2)
template <int TM, int TN, bool F16, bool GEGLU, int CH = TN>
__device__ __forceinline__ void store_rows(float16_t (&acc)[TN][TM], unsigned char* lds, const unsigned short* __restrict__ bias,
                                           const unsigned short* __restrict__ res, long ldr, unsigned short* __restrict__ Y,
                                           long ldy, long mrow0, int ncol0, long M, int N, int lane) {
#pragma unroll
  for (int b = 0; b < TM; ++b)
    store_chunk<TM, TN, F16, GEGLU, CH, 0>(acc, b, lds, bias, res, ldr, Y, ldy, mrow0, ncol0, M, N, lane);
}

}  // namespace uce_epi
