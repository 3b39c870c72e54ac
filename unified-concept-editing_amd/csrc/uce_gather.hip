// Last-token gather of the batched concept-embedding extraction (reference: one text-encoder call per
// string and `t_emb[0][0, attention_mask.sum() - 2, :]`, uce_sd_erase.py:25-42 / uce_sd_debias.py:49-66):
//   out[i, :] (f32) = hidden[i, idx[i], :]   for a text-encoder output hidden [B, L, d] in bf16 / f16 / f32.
// One workgroup per string, 16-byte loads of the one row that is needed (the other L - 1 rows are never touched),
// widened to fp32 - the C / G rows uce_edit consumes - in the same pass.
#include "uce_common.h"

namespace {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

template <int DT>   // UCE_DTYPE_*
__global__ __launch_bounds__(256) void k_gather_last(const void* __restrict__ hidden, const int* __restrict__ idx,
                                                     float* __restrict__ out, int L, int d) {
  const int i = blockIdx.x;
  int t = idx[i];
  t = t < 0 ? 0 : (t >= L ? L - 1 : t);
  float* o = out + (size_t)i * d;
  if constexpr (DT == UCE_DTYPE_F32) {
    const float4_t* src = (const float4_t*)((const float*)hidden + ((size_t)i * L + t) * d);
    for (int c = threadIdx.x; c < d / 4; c += 256) ((float4_t*)o)[c] = src[c];
  } else {
    const uint4_t* src = (const uint4_t*)((const unsigned short*)hidden + ((size_t)i * L + t) * d);
    for (int c = threadIdx.x; c < d / 8; c += 256) {
      const uint4_t v = src[c];
      float f[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if constexpr (DT == UCE_DTYPE_BF16) {
          f[2 * u] = __builtin_bit_cast(float, v[u] << 16);
          f[2 * u + 1] = __builtin_bit_cast(float, v[u] & 0xffff0000u);
        } else {
          f[2 * u] = (float)__builtin_bit_cast(_Float16, (unsigned short)(v[u] & 0xffffu));
          f[2 * u + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(v[u] >> 16));
        }
      }
      ((float4_t*)o)[2 * c] = (float4_t){f[0], f[1], f[2], f[3]};
      ((float4_t*)o)[2 * c + 1] = (float4_t){f[4], f[5], f[6], f[7]};
    }
  }
}

}  // namespace

extern "C" int uce_gather_last_token(uce_handle_t h, const void* hidden, const int* idx, float* out, int B, int L, int d,
                                     int dtype, uce_stream_t stream) {
  if (!h || !hidden || !idx || !out || B < 0 || L <= 0 || d <= 0 || (d & 7)) return UCE_EINVAL;
  UCE_ENTER(h);
  if (B == 0) return UCE_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UCE_DTYPE_F32) hipLaunchKernelGGL(k_gather_last<UCE_DTYPE_F32>, dim3(B), dim3(256), 0, st, hidden, idx, out, L, d);
  else if (dtype == UCE_DTYPE_BF16) hipLaunchKernelGGL(k_gather_last<UCE_DTYPE_BF16>, dim3(B), dim3(256), 0, st, hidden, idx, out, L, d);
  else if (dtype == UCE_DTYPE_F16) hipLaunchKernelGGL(k_gather_last<UCE_DTYPE_F16>, dim3(B), dim3(256), 0, st, hidden, idx, out, L, d);
  else return UCE_EINVAL;
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
