// Cross-attention forward for the edited U-Net's attn2 layers at inference
// (reference: diffusers AttnProcessor2_0 -> F.scaled_dot_product_attention(q, k, v), reached from
// evalscripts/generate-images-sd.py:37-42; SD-1.4 shapes: Lk = 77, dh in {40, 80, 160}).
//
// One workgroup = 4 waves of one (batch, head); K_h and V_h^T (<= 128 keys) are staged once in LDS and
// the workgroup then walks `iters` consecutive 128-row query tiles (so the staging and its latency are
// paid once per 128*iters rows, and the next tile's Q fragments are in flight while the current tile
// is computed: at large batch this is a streaming kernel).  Each wave owns 32 query rows of a tile:
//   S^T = K Q^T   "swapped" so a lane's accumulator column is ONE query row: softmax over the
//                 keys is in-register plus a single exchange with lane^32;
//   O^T = V^T P^T P fragments come straight out of the S^T accumulators (the contraction order over
//                 keys is permuted identically on the V side), V^T fragments are 8-byte LDS reads; the
//                 output column is again the lane's query row: 1/sum, the bf16 conversion
//                 (v_cvt_pk_bf16_f32) and the 8-byte stores of 4 consecutive dims are all per lane.
// bf16/f16 MFMA 32x32x16, f32 softmax and accumulation, no online rescaling (all keys resident).
// Q is read once, O written once; at SD-1.4's dh = 40 the per-row VALU work of the softmax (96 padded
// keys) is of the same order as the memory time, so the instruction count per row matters as much.
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

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// two f32 -> one packed pair, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 / v_cvt_pkrtz is NOT used)
template <bool F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const float2_t v = {lo, hi};
  if constexpr (F16)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  else
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// DHP: head dim padded to a multiple of 16; KT: key tiles of 32 (3 -> up to 96 keys, 4 -> 128)
template <int DHP, int KT, bool F16>
__global__ __launch_bounds__(256) void k_xattn(const unsigned short* __restrict__ Q,
                                               const unsigned short* __restrict__ K,
                                               const unsigned short* __restrict__ V,
                                               unsigned short* __restrict__ O, int H, int Lq, int Lk,
                                               int dh, float scale_log2e, int iters) {
  constexpr int NDV = (DHP + 31) / 32;      // output column tiles
  constexpr int DVP = NDV * 32;
  constexpr int LKP = KT * 32;
  constexpr int KLD = DHP + 8;              // K_lds row stride (elements): odd multiple of 16 B
  constexpr int VLD = LKP + 4;              // Vt row stride (elements)
  constexpr int NS = DHP / 16;              // contraction steps of S^T = K Q^T
  constexpr int KV_BYTES = (LKP * KLD + DVP * VLD) * 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[KV_BYTES];
  unsigned short* Ks = (unsigned short*)smem;                 // [LKP][KLD]
  unsigned short* Vt = Ks + LKP * KLD;                        // [DVP][VLD]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = H * dh;
  const int lq = lane & 31, lh = lane >> 5;
  const long tile0 = (long)blockIdx.x * iters;

  // this lane's Q fragments of tile t (row q0 + lq, dims 16s + 8*lh .. +7), straight from HBM
  auto load_q = [&](long t, uint4_t (&f)[NS]) {
    long row = t * 128 + w * 32 + lq;
    if (row > Lq - 1) row = Lq - 1;
    const unsigned short* qrow = Q + ((size_t)b * Lq + row) * C + (size_t)h * dh;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int dim = 16 * s + 8 * lh;
      f[s] = (dim < dh) ? *(const uint4_t*)(qrow + dim) : (uint4_t){0u, 0u, 0u, 0u};
    }
  };
  uint4_t qf[NS];
  load_q(tile0, qf);

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

  for (int it = 0; it < iters; ++it) {
  const long q0 = (tile0 + it) * 128 + w * 32;
  if ((tile0 + it) * 128 >= Lq) break;                         // workgroup-uniform
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

  if (it + 1 < iters) load_q(tile0 + it + 1, qf);              // next tile's Q: in flight under softmax / PV / store

  // ---- softmax over keys (f32): a key tile is masked only if it holds padding (scalar test per tile)
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if ((kt + 1) * 32 > Lk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        sacc[kt][r] = (key < Lk) ? sacc[kt][r] : -INFINITY;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, sacc[kt][r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32));
  const float mc = m * scale_log2e;
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], scale_log2e, -mc));   // in [0, 1]; exp2(-inf) = 0
      sacc[kt][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;

  // ---- P fragments (unnormalised, in [0, 1]): slot e of step s of tile kt <- register 8s + e
  uint4_t pf[KT][2];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) pf[kt][s][t] = pack2<F16>(sacc[kt][8 * s + 2 * t], sacc[kt][8 * s + 2 * t + 1]);

  // ---- O^T = V^T P^T : swapped again, so the accumulator column stays this lane's query row and the 1/sum
  // and the output conversion are per lane; slot e <-> key kt*32 + 16s + 4*lh + (e&3) + 8*(e>>2) on both sides.
  // Register r of tile nt = output dim nt*32 + (r&3) + 8*(r>>2) + 4*lh: four consecutive dims = one 8-byte store.
  const long row = q0 + lq;
  unsigned short* orow = O + ((size_t)b * Lq + (row < Lq ? row : Lq - 1)) * C + (size_t)h * dh;
#pragma unroll
  for (int nt = 0; nt < NDV; ++nt) {
    float16_t oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    const unsigned short* vrow = Vt + (nt * 32 + lq) * VLD + 4 * lh;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint2_t lo = *(const uint2_t*)(vrow + kt * 32 + 16 * s);
        const uint2_t hi = *(const uint2_t*)(vrow + kt * 32 + 16 * s + 8);
        const uint4_t vf = {lo[0], lo[1], hi[0], hi[1]};
        oacc = mfma32<F16>(vf, pf[kt][s], oacc);
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dv = nt * 32 + 8 * g + 4 * lh;
      if (dv < dh && row < Lq) {
        const uint2_t o2 = {pack2<F16>(oacc[4 * g] * inv, oacc[4 * g + 1] * inv),
                            pack2<F16>(oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv)};
        *(uint2_t*)(orow + dv) = o2;
      }
    }
  }
  }
}

template <int DHP, int KT>
int launch_cfg(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh,
               float scale, int dtype, hipStream_t st) {
  // query tiles per workgroup: amortise the K/V staging once there are more than ~4 workgroups per CU
  const long tiles_q = (Lq + 127) / 128;
  const long total = tiles_q * H * B;
  int iters = (int)(total / 1024);
  iters = iters < 1 ? 1 : (iters > 8 ? 8 : iters);
  if (iters > tiles_q) iters = (int)tiles_q;
  const dim3 grid((unsigned)((tiles_q + iters - 1) / iters), H, B);
  const float sl2 = scale * 1.4426950408889634f;
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL((k_xattn<DHP, KT, true>), grid, dim3(256), 0, st, (const unsigned short*)q,
                       (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)o, H, Lq, Lk, dh, sl2,
                       iters);
  else
    hipLaunchKernelGGL((k_xattn<DHP, KT, false>), grid, dim3(256), 0, st, (const unsigned short*)q,
                       (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)o, H, Lq, Lk, dh, sl2,
                       iters);
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
