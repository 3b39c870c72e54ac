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

// GELU, erf form (diffusers GEGLU: F.gelu), on ONE transcendental.  With u = |g|:  gelu(g) = max(g, 0) - u h(u),
//   h(u) = 0.5 erfc(u / sqrt 2) = 2^( P(u) - u^2 log2(e) / 2 ),   P(u) ~ log2(0.5 erfcx(u / sqrt 2))  on [0, 6.25]
// P: degree 7, Chebyshev fit (tools/fit_gelu.py), the argument clamped at 6.25 (beyond it u h(u) < 3e-9).  Against fp64 over
// |g| <= 12 in f32 arithmetic: 5.9e-7 absolute, 5.9e-6 relative (bf16 resolves 4e-3, f16 5e-4).  17 VALU issue slots per element
// (min, 7 FMA, mul, FMA, v_exp_f32 at four, max, FMA, the product with the hidden value) where the Abramowitz-Stegun 7.1.26 form it
// replaces (a v_rcp_f32 AND a v_exp_f32, nine FMAs, copysign: 1.5e-7 absolute) took 23 - the GEGLU epilogue evaluates it 64 times per
// lane and tile with no matrix work beside it (one workgroup per CU): ~40 % of the 256 x 256 tile's time at K = 320.
__device__ __forceinline__ float gelu_erf(float g) {
  const float u = __builtin_fminf(__builtin_fabsf(g), 6.25f);
  float p = fmaf(-1.837149047e-06f, u, 6.163556782e-05f);
  p = fmaf(p, u, -9.307270628e-04f);
  p = fmaf(p, u, 8.508252472e-03f);
  p = fmaf(p, u, -5.395976999e-02f);
  p = fmaf(p, u, 2.628816807e-01f);
  p = fmaf(p, u, -1.151250035e+00f);
  p = fmaf(p, u, -9.999953876e-01f);
  const float h = __builtin_amdgcn_exp2f(fmaf(u * u, -0.72134752044448170f, p));
  return fmaf(-u, h, __builtin_fmaxf(g, 0.0f));      // (the clamped u: beyond 6.25 the term is < 1.4e-9 either way, and +inf stays +inf)
}

// CH: 32-column MFMA tiles parked at once (the row of a slab is CH * 64 bytes, CH * 32 for GEGLU).  CH = TN parks the wave's
// whole width (fewest passes); CH = 2 keeps the region at 4.5 KB per wave for the two-workgroups-per-CU tile forms.
template <int CH, bool GEGLU>
constexpr int row_bytes() { return (GEGLU ? CH * 32 : CH * 64) + 16; }        // + 16: rows two bank groups apart
template <int CH, bool GEGLU>
constexpr int wave_bytes() { return 32 * row_bytes<CH, GEGLU>(); }

// Memory operations of a wave complete IN ORDER on gfx9 (one vmcnt for loads and stores): a load issued after a store cannot
// be waited for without waiting for that store's acknowledgement too.  The epilogue therefore never issues a load behind a
// store it does not want to wait for: the bias of the wave's columns is folded into the accumulators up front (loads in
// batches), and the residual rows of chunk k + 1 are requested BEFORE chunk k's stores go out, so the wait for them is a
// `vmcnt(stores of chunk k)` that the stores slip past.  Residual loads and output stores are BUFFER operations on a
// descriptor of the wave's own rows (base = its first row, extent = its rows inside M): rows beyond M fall outside the
// descriptor, columns beyond N get the out-of-range offset - no branch anywhere, so the compiler cannot sink a load (or the
// LDS read) into a conditional store block and serialise it there.  (Before: one bias load + `vmcnt(0)` per (tile, register
// group) and one LDS read + residual load + `vmcnt(0)` + store per 16-byte piece - every one a full round trip, with 8 waves
// per CU to hide it.)
template <int TN, int CH>
constexpr int col_chunks() { return (TN + CH - 1) / CH; }
template <int TN, bool GEGLU, int CH, int CA>
struct chunk_geom {
  static constexpr int CW = (TN - CA < CH) ? TN - CA : CH;               // 32-column tiles of this chunk
  static constexpr int CPR = (GEGLU ? CW * 32 : CW * 64) / 16;           // 16-byte pieces per row of the chunk
  static constexpr int ITER = (32 * CPR) / 64;                           // 32 rows x CPR pieces over 64 lanes
};
template <int CH, bool GEGLU>
constexpr int max_iter() { return (32 * ((GEGLU ? CH * 32 : CH * 64) / 16)) / 64; }
// bit 31 for a column beyond the bound: the offset leaves every descriptor (arithmetic, not a select - the compiler turns a
// select between a computed offset and a constant into a divergent branch around the load)
__device__ __forceinline__ unsigned oob_bit(int n, int NO) { return (unsigned)(NO - 1 - n) & 0x80000000u; }

// descriptor of `rows` rows of `ld` elements from `base` (uniform arguments)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const unsigned short* base, long ld, int rows) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((long)rows * ld * 2), 0x00020000);
}

// residual rows of chunk K (row tile K / NCA, column chunk K % NCA): lane = (row, 16-byte piece)
template <int TN, int CH, int K, int IT>
__device__ __forceinline__ void fetch_residual(uint4_t (&r)[IT], __amdgpu_buffer_rsrc_t rr, long ldr, int ocol0w, int NO, int lane) {
  constexpr int NCA = col_chunks<TN, CH>();
  constexpr int b = K / NCA, CA = (K % NCA) * CH;
  using G = chunk_geom<TN, false, CH, CA>;
#pragma unroll
  for (int it = 0; it < G::ITER; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / G::CPR, ch = idx - row * G::CPR;
    const int n = ocol0w + CA * 32 + ch * 8;
    const unsigned off = (((unsigned)(b * 32 + row) * (unsigned)ldr + (unsigned)n) * 2u) | oob_bit(n, NO);
    r[it] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0));
  }
}

// chunk K: park (convert / GEGLU) -> request chunk K + 1's residual rows -> whole row segments out -> next chunk
template <int TM, int TN, bool F16, bool GEGLU, int CH, bool RES, int K, int IT, bool PIPE, bool LEAN>
__device__ __forceinline__ void store_chunk(float16_t (&acc)[TN][TM], unsigned char* lds, uint4_t (&r)[IT], __amdgpu_buffer_rsrc_t rr,
                                            long ldr, __amdgpu_buffer_rsrc_t yr, long ldy, int ocol0w, int NO, int lane) {
  constexpr int NCA = col_chunks<TN, CH>();
  constexpr int b = K / NCA, CA = (K % NCA) * CH;
  using G = chunk_geom<TN, GEGLU, CH, CA>;
  constexpr int ROWB = row_bytes<CH, GEGLU>();
  const int li = lane & 31, lh = lane >> 5;
  if constexpr (RES && !PIPE && !LEAN) fetch_residual<TN, CH, K, IT>(r, rr, ldr, ocol0w, NO, lane);   // (under the parking VALU)
  // ---- park the chunk: 8 bytes per lane per (tile, register group); the bias is in the accumulators already
#pragma unroll
  for (int aa = 0; aa < G::CW; ++aa) {
    const int a = CA + aa;
    if constexpr (GEGLU) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {                                      // hidden columns 8 g + 4 lh + i; their gates sit 16 further
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = acc[a][b][4 * g + i] * gelu_erf(acc[a][b][4 * (g + 2) + i]);
        *(uint2_t*)(lds + li * ROWB + (aa * 16 + 8 * g + 4 * lh) * 2) = (uint2_t){pack2<F16>(o[0], o[1]), pack2<F16>(o[2], o[3])};
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(uint2_t*)(lds + li * ROWB + (aa * 32 + 8 * g + 4 * lh) * 2) =
            (uint2_t){pack2<F16>(acc[a][b][4 * g], acc[a][b][4 * g + 1]), pack2<F16>(acc[a][b][4 * g + 2], acc[a][b][4 * g + 3])};
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  uint4_t rn[PIPE ? IT : 1];
  if constexpr (RES && PIPE && K + 1 < TM * NCA) fetch_residual<TN, CH, K + 1, IT>(rn, rr, ldr, ocol0w, NO, lane);
  if constexpr (RES && !PIPE && LEAN) fetch_residual<TN, CH, K, IT>(r, rr, ldr, ocol0w, NO, lane);    // (no registers to spare earlier)
  // ---- whole row segments out: lane = (row, 16-byte piece), consecutive lanes on consecutive pieces
  const int ocol0 = ocol0w + (GEGLU ? CA * 16 : CA * 32);
  uint4_t v[G::ITER];
#pragma unroll
  for (int it = 0; it < G::ITER; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / G::CPR, ch = idx - row * G::CPR;
    v[it] = *(const uint4_t*)(lds + row * ROWB + ch * 16);
  }
#pragma unroll
  for (int it = 0; it < G::ITER; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / G::CPR, ch = idx - row * G::CPR;
    const int n = ocol0 + ch * 8;
    uint4_t o = v[it];
    if constexpr (RES)
      o = (uint4_t){add2<F16>(o[0], r[it][0]), add2<F16>(o[1], r[it][1]), add2<F16>(o[2], r[it][2]), add2<F16>(o[3], r[it][3])};
    const unsigned off = (((unsigned)(b * 32 + row) * (unsigned)ldy + (unsigned)n) * 2u) | oob_bit(n, NO);
    __builtin_amdgcn_raw_buffer_store_b128(o, yr, off, 0, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // the chunk is read before the next one overwrites it
  if constexpr (K + 1 < TM * NCA) {
    if constexpr (PIPE) store_chunk<TM, TN, F16, GEGLU, CH, RES, K + 1, IT, PIPE, LEAN>(acc, lds, rn, rr, ldr, yr, ldy, ocol0w, NO, lane);
    else store_chunk<TM, TN, F16, GEGLU, CH, RES, K + 1, IT, PIPE, LEAN>(acc, lds, r, rr, ldr, yr, ldy, ocol0w, NO, lane);
  }
}

// bias into the accumulators, BG column tiles at a time: their loads first, then the adds.  `br` covers the N bias values (no
// bias: an empty descriptor) - columns beyond it read 0, and a null bias adds 0.f as the per-group epilogue does (- 0 + 0 = + 0).
template <int TM, int TN, bool F16, int BGMAX, int A0>
__device__ __forceinline__ void add_bias(float16_t (&acc)[TN][TM], __amdgpu_buffer_rsrc_t br, int ncol0, int lh) {
  constexpr int BG = (TN - A0 < BGMAX) ? TN - A0 : BGMAX;                // 8 BG transient registers
  uint2_t bq[BG][4];
#pragma unroll
  for (int a = 0; a < BG; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bq[a][g] = __builtin_bit_cast(uint2_t, __builtin_amdgcn_raw_buffer_load_b64(br, (unsigned)(ncol0 + (A0 + a) * 32 + 8 * g + 4 * lh) * 2u, 0, 0));
#pragma unroll
  for (int a = 0; a < BG; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float bv[4];
      unpack4<F16>(bq[a][g], bv);
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[A0 + a][b][4 * g + i] += bv[i];
    }
  if constexpr (A0 + BG < TN) add_bias<TM, TN, F16, BGMAX, A0 + BG>(acc, br, ncol0, lh);
}

// acc[a][b][4 g + i]: column 32 a + 8 g + 4 lh + i of the wave's TN x 32 columns, row 32 b + li of its TM x 32 rows.
//   lds     this wave's region (wave_bytes<CH, GEGLU>() bytes)
//   bias    indexed by the (interleaved, for GEGLU) column; may be null.   res / ldr: residual rows (MODE 0), may be null
//   mrow0   first row of the wave's slab (global), ncol0 its first column in the index space of `bias` (a multiple of 32)
//   M, N    bounds in that space (GEGLU: N counts the interleaved columns; the output has N / 2)
//   LEAN    the kernel runs under a 128-register cap
template <int TM, int TN, bool F16, bool GEGLU, int CH = TN, bool LEAN = false>
__device__ __forceinline__ void store_rows(float16_t (&acc)[TN][TM], unsigned char* lds, const unsigned short* __restrict__ bias,
                                           const unsigned short* __restrict__ res, long ldr, unsigned short* __restrict__ Y,
                                           long ldy, long mrow0, int ncol0, long M, int N, int lane) {
  const int lh = lane >> 5;
  const int NO = GEGLU ? N / 2 : N;
  const int ocol0w = GEGLU ? ncol0 / 2 : ncol0;
  add_bias<TM, TN, F16, LEAN ? 1 : 3, 0>(acc, __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, bias ? N * 2 : 0, 0x00020000), ncol0, lh);
  // the wave's rows (uniform: the caller's wave index is a scalar; the two halves go through readfirstlane so that the
  // descriptors are scalar whatever the compiler can prove)
  const long mrow = ((long)__builtin_amdgcn_readfirstlane((int)(mrow0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)mrow0);
  const long left = M - mrow;
  const int rows = left <= 0 ? 0 : (left < TM * 32 ? (int)left : TM * 32);
  const __amdgpu_buffer_rsrc_t yr = rows_rsrc(Y + mrow * ldy, ldy, rows);
  constexpr int IT = max_iter<CH, GEGLU>();
  // two chunks' residual rows in registers where the budget has them: LEAN = a 128-register kernel (two workgroups of 512 per CU)
  constexpr bool PIPE = IT <= 4 && (!LEAN || TM * TN <= 4);
  uint4_t r[IT];
  if constexpr (!GEGLU) {
    if (res) {                                                           // (uniform)
      const __amdgpu_buffer_rsrc_t rr = rows_rsrc(res + mrow * ldr, ldr, rows);
      if constexpr (PIPE) fetch_residual<TN, CH, 0, IT>(r, rr, ldr, ocol0w, NO, lane);
      store_chunk<TM, TN, F16, false, CH, true, 0, IT, PIPE, LEAN>(acc, lds, r, rr, ldr, yr, ldy, ocol0w, NO, lane);
      return;
    }
  }
  store_chunk<TM, TN, F16, GEGLU, CH, false, 0, IT, false, LEAN>(acc, lds, r, yr, ldr, yr, ldy, ocol0w, NO, lane);
}

}  // namespace uce_epi
