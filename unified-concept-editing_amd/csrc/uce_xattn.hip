// Cross-attention forward for the edited U-Net's attn2 layers at inference
// (reference: diffusers AttnProcessor2_0 -> F.scaled_dot_product_attention(q, k, v), reached from
// evalscripts/generate-images-sd.py:37-42; SD-1.4 shapes: Lk = 77, dh in {40, 80, 160}).
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); K_h and V_h^T (<= 128 keys) are
// staged once in LDS.  Each wave owns 32 query rows:
//   S^T = K Q^T   "swapped" so a lane's accumulator column is ONE query row: softmax over the
//                 keys is in-register plus a single exchange with lane^32;
//   O   = P V     P fragments come straight out of the S^T accumulators (the contraction order over
//                 keys is permuted identically on the V side), V^T fragments are 8-byte LDS reads.
// bf16/f16 MFMA 32x32x16, f32 softmax and accumulation, no online rescaling (all keys resident).
// HBM-bound: Q is read once, O written once (through LDS so global stores are 16 B per lane).
#include "uce_common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));

template <bool F16>
__device__ __forceinline__ float16_t mfma32(uint4_t a, uint4_t b, float16_t c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b),
                                                  c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  if constexpr (F16) {
    const _Float16 a = (_Float16)lo, b = (_Float16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
  } else {
    unsigned ul = __float_as_uint(lo), uh = __float_as_uint(hi);
    ul = (ul + 0x7fffu + ((ul >> 16) & 1u)) >> 16;   // RNE (inputs are finite)
    uh = (uh + 0x7fffu + ((uh >> 16) & 1u)) >> 16;
    return ul | (uh << 16);
  }
}

// DHP: head dim padded to a multiple of 16; KT: key tiles of 32 (3 -> up to 96 keys, 4 -> 128)
template <int DHP, int KT, bool F16>
__global__ __launch_bounds__(256) void k_xattn(const unsigned short* __restrict__ Q,
                                               const unsigned short* __restrict__ K,
                                               const unsigned short* __restrict__ V,
                                               unsigned short* __restrict__ O, int H, int Lq, int Lk,
                                               int dh, float scale_log2e) {
  constexpr int NDV = (DHP + 31) / 32;      // output column tiles
  constexpr int DVP = NDV * 32;
  constexpr int LKP = KT * 32;
  constexpr int KLD = DHP + 8;              // K_lds row stride (elements): odd multiple of 16 B
  constexpr int VLD = LKP + 4;              // Vt row stride (elements)
  constexpr int OLD = DHP + 8;              // O staging row stride
  constexpr int NS = DHP / 16;              // contraction steps of S^T = K Q^T
  constexpr int KV_BYTES = (LKP * KLD + DVP * VLD) * 2;
  constexpr int O_BYTES = 4 * 32 * OLD * 2;
  constexpr int SMEM = KV_BYTES > O_BYTES ? KV_BYTES : O_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned short* Ks = (unsigned short*)smem;                 // [LKP][KLD]
  unsigned short* Vt = Ks + LKP * KLD;                        // [DVP][VLD]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = H * dh;
  const long q0 = (long)blockIdx.x * 128 + w * 32;
  const int lq = lane & 31, lh = lane >> 5;

  // ---- this lane's Q fragments (row q0 + lq, dims 16s + 8*lh .. +7), straight from HBM
  uint4_t qf[NS];
  {
    long row = q0 + lq;
    if (row > Lq - 1) row = Lq - 1;
    const unsigned short* qrow = Q + ((size_t)b * Lq + row) * C + (size_t)h * dh;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int dim = 16 * s + 8 * lh;
      qf[s] = (dim < dh) ? *(const uint4_t*)(qrow + dim) : (uint4_t){0u, 0u, 0u, 0u};
    }
  }

  // ---- stage K_h (zero padded) and V_h^T
  {
    const unsigned short* kbase = K + (size_t)b * Lk * C + (size_t)h * dh;
    const unsigned short* vbase = V + (size_t)b * Lk * C + (size_t)h * dh;
    constexpr int KCH = DHP / 8;  // 16-byte chunks per key row
    for (int e = tid; e < LKP * KCH; e += 256) {
      const int key = e / KCH, dim = (e - key * KCH) * 8;
      uint4_t val = {0u, 0u, 0u, 0u};
      if (key < Lk && dim < dh) val = *(const uint4_t*)(kbase + (size_t)key * C + dim);
      *(uint4_t*)(Ks + key * KLD + dim) = val;
    }
    constexpr int VCH = DVP / 8;
    for (int e = tid; e < LKP * VCH; e += 256) {
      const int key = e / VCH, dv = (e - key * VCH) * 8;
      uint4_t val = {0u, 0u, 0u, 0u};
      if (key < Lk && dv < dh) val = *(const uint4_t*)(vbase + (size_t)key * C + dv);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        Vt[(dv + 2 * t) * VLD + key] = (unsigned short)(val[t] & 0xffffu);
        Vt[(dv + 2 * t + 1) * VLD + key] = (unsigned short)(val[t] >> 16);
      }
    }
  }
  __syncthreads();

  // ---- S^T = K Q^T : accumulator column = query lq, register r of tile kt = key kt*32 + (r&3) + 8*(r>>2) + 4*lh
  float16_t sacc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const uint4_t kf = *(const uint4_t*)(Ks + (kt * 32 + lq) * KLD + 16 * s + 8 * lh);
      sacc[kt] = mfma32<F16>(kf, qf[s], sacc[kt]);
    }
  }

  // ---- softmax over keys (f32), masked beyond Lk
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float sv = (key < Lk) ? sacc[kt][r] : -INFINITY;
      sacc[kt][r] = sv;
      m = fmaxf(m, sv);
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = exp2f((sacc[kt][r] - m) * scale_log2e);
      sacc[kt][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;

  // ---- P fragments: slot e of step s of tile kt <- register 8s + e
  uint4_t pf[KT][2];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        pf[kt][s][t] = pack2<F16>(sacc[kt][8 * s + 2 * t] * inv, sacc[kt][8 * s + 2 * t + 1] * inv);

  // ---- O = P V : B operand slot e <-> key kt*32 + 16s + 4*lh + (e&3) + 8*(e>>2)
  float16_t oacc[NDV];
#pragma unroll
  for (int nt = 0; nt < NDV; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[nt][r] = 0.f;
    const unsigned short* vrow = Vt + (nt * 32 + lq) * VLD + 4 * lh;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint2_t lo = *(const uint2_t*)(vrow + kt * 32 + 16 * s);
        const uint2_t hi = *(const uint2_t*)(vrow + kt * 32 + 16 * s + 8);
        const uint4_t vf = {lo[0], lo[1], hi[0], hi[1]};
        oacc[nt] = mfma32<F16>(pf[kt][s], vf, oacc[nt]);
      }
  }

  // ---- O tile -> LDS (reusing the K/V region) -> 16-byte global stores
  __syncthreads();
  unsigned short* Os = (unsigned short*)smem + w * 32 * OLD;   // [32][OLD] per wave
#pragma unroll
  for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int dv = nt * 32 + lq;
      if (dv < DHP) {
        float x = oacc[nt][r];
        unsigned short hv;
        if constexpr (F16) { const _Float16 t = (_Float16)x; hv = __builtin_bit_cast(unsigned short, t); }
        else { unsigned u = __float_as_uint(x); hv = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
        Os[qr * OLD + dv] = hv;
      }
    }
  __syncthreads();
  {
    const int chunks = dh >> 3;  // 16-byte chunks per output row
    for (int e = lane; e < 32 * chunks; e += 64) {
      const int qr = e / chunks, ch = e - qr * chunks;
      const long row = q0 + qr;
      if (row < Lq) {
        const uint4_t val = *(const uint4_t*)(Os + qr * OLD + ch * 8);
        *(uint4_t*)(O + ((size_t)b * Lq + row) * C + (size_t)h * dh + ch * 8) = val;
      }
    }
  }
}

template <int DHP, int KT>
int launch_cfg(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh,
               float scale, int dtype, hipStream_t st) {
  const dim3 grid((Lq + 127) / 128, H, B);
  const float sl2 = scale * 1.4426950408889634f;
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL((k_xattn<DHP, KT, true>), grid, dim3(256), 0, st, (const unsigned short*)q,
                       (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)o, H, Lq, Lk, dh, sl2);
  else
    hipLaunchKernelGGL((k_xattn<DHP, KT, false>), grid, dim3(256), 0, st, (const unsigned short*)q,
                       (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)o, H, Lq, Lk, dh, sl2);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int KT>
int launch_dh(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh,
              float scale, int dtype, hipStream_t st) {
  if (dh <= 48) return launch_cfg<48, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st);
  if (dh <= 64) return launch_cfg<64, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st);
  if (dh <= 80) return launch_cfg<80, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st);
  if (dh <= 96) return launch_cfg<96, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st);
  if (dh <= 128) return launch_cfg<128, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st);
  return launch_cfg<160, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st);
}

}  // namespace

int launch_xattn(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh,
                 float scale, int dtype, hipStream_t st) {
  if (Lk <= 96) return launch_dh<3>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st);
  return launch_dh<4>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st);
}

extern "C" int uce_xattn_fwd(uce_handle_t h, const void* q, const void* k, const void* v, void* o, int B,
                             int H, int Lq, int Lk, int dh, float scale, int dtype, uce_stream_t stream) {
  if (!h || !q || !k || !v || !o || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || Lk > 128) return UCE_EINVAL;
  if (dh <= 0 || dh > 160 || (dh & 7)) return UCE_EINVAL;
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  if (B > 65535 || H > 65535) return UCE_EINVAL;
  return launch_xattn(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, (hipStream_t)stream);
}
