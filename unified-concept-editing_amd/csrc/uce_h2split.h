// The f16 x 2 split of one fp32 row (see uce_apply_h2.hip), shared by k_split_h2 and the rider workgroups of the
// persistent Cholesky launch (uce_solve.hip: k_potrf_la) that split W_old in that launch's shadow.
#pragma once
#include "uce_common.h"


typedef unsigned int h2s_uint2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h2s_f16x2_t __attribute__((ext_vector_type(2)));

// one wave: row maximum -> power-of-two scale s (maximum -> [2^14, 2^15)) -> hi = rn_f16(x s), lo = rn_f16(x s - hi);
// IDENT adds 1 on the diagonal first ((I + Delta)^T)
template <bool IDENT>
__device__ __forceinline__ void h2_split_row(const float* __restrict__ src, unsigned short* __restrict__ hi,
                                             unsigned short* __restrict__ lo, float* __restrict__ inv_scale, long row, int d,
                                             int lane) {
  const float4_t* p = (const float4_t*)(src + row * d);
  const int nc = d >> 2;
  auto fetch = [&](int c) {
    float4_t x = p[c];
    if constexpr (IDENT) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((long)(4 * c + k) == row) x[k] += 1.0f;
    }
    return x;
  };
  unsigned mx = 0;
  for (int c = lane; c < nc; c += 64) {
    const float4_t x = fetch(c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned b = __float_as_uint(x[k]) & 0x7fffffffu;
      mx = b > mx ? b : mx;
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    const unsigned other = (unsigned)__shfl_xor((int)mx, o);
    mx = other > mx ? other : mx;
  }
  int E = (int)(mx >> 23);                                           // biased exponent of the row maximum
  E = E < 30 ? 30 : (E > 240 ? 240 : E);                             // (an all-zero / denormal row, or inf / nan in it)
  const float s = __uint_as_float((unsigned)(268 - E) << 23);        // 2^(14 - (E - 127))
  for (int c = lane; c < nc; c += 64) {
    const float4_t x = fetch(c);                                     // (second touch of a 3 KB row: cache)
    float r[4];
    _Float16 h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float y = x[k] * s;                                      // exact
      h[k] = (_Float16)y;                                            // round to nearest even
      r[k] = y - (float)h[k];                                        // exact
    }
    const h2s_uint2_t vh = {__builtin_bit_cast(unsigned, h2s_f16x2_t{h[0], h[1]}),
                            __builtin_bit_cast(unsigned, h2s_f16x2_t{h[2], h[3]})};
    const h2s_uint2_t vl = {__builtin_bit_cast(unsigned, h2s_f16x2_t{(_Float16)r[0], (_Float16)r[1]}),
                            __builtin_bit_cast(unsigned, h2s_f16x2_t{(_Float16)r[2], (_Float16)r[3]})};
    *(h2s_uint2_t*)(hi + row * d + 4 * c) = vh;
    *(h2s_uint2_t*)(lo + row * d + 4 * c) = vl;
  }
  if (lane == 0) inv_scale[row] = __uint_as_float((unsigned)(E - 14) << 23);   // 2^-(14 - (E - 127))
}
