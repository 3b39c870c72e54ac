// Self-attention forward (no mask) for the U-Net's attn1 layers at inference: O = softmax(Q K^T * scale) V per
// (batch, head), any Lk (SD-1.4: Lq = Lk in {4096, 1024, 256, 64}, dh in {40, 80, 160}).  This is "the rest of
// the U-Net step" of SURVEY.md section 8(f) row 3 - diffusers' AttnProcessor2_0 -> F.scaled_dot_product_attention
// for attn1, reached from evalscripts/generate-images-sd.py:37-42 - built as the streaming sibling of k_xattn.
//
//   k_vt        V [B, Lk, H*dh] -> V^T [B, H, DVP, LkP] (16-bit elements, zero padded): the P V product needs the
//               key index contiguous per lane on the V side; one extra pass over V (a few % of the attention).
//   k_sattn     one workgroup = 4 waves = 128 query rows of one (b, h); keys in tiles of 64, K tile and V^T tile
//               double-buffered in LDS (straight 16-byte copies, register prefetch of the next tile).
//               Per wave 32 query rows:  S^T = K Q^T "swapped", so a lane's accumulator column is ONE query:
//               the running max / sum of the online softmax, the rescale of the output accumulators and the
//               final 1/sum are all per lane (one lane^32 exchange per tile for the max, none for the sum);
//               O^T = V^T P^T with the P fragments straight out of the S^T accumulators (k_xattn's key
//               permutation).  bf16/f16 MFMA 32x32x16, f32 softmax and accumulation.
// The loop is VALU-bound on the softmax, not MFMA-bound (tools/ubench/valu_mfma.hip: v_exp_f32 8.2 cycles of the SIMD's VALU
// per wave, v_fma_f32 4.75 for one wave and 2.4 when two waves alternate, cvt_pk / max3 4.3; per 64 x 64 tile and wave
// ~310 VALU instructions = ~1900 cycles against 896 of matrix pipe).  k_sattn_h below interleaves the two pipes explicitly.
#include "uce_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));

template <bool F16>
__device__ __forceinline__ float16_t mfma32(uint4_t a, uint4_t b, float16_t c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0,
                                                  0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                   0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const float2_t v = {lo, hi};
  if constexpr (F16)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  else
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

constexpr int KT = 64;            // keys per tile

// max over the lane pair (l, l ^ 32) with one v_permlane32_swap - as inline assembly: hipcc 7.2.0 lowers
// __builtin_amdgcn_permlane32_swap with BOTH results in the first operand's register (tools/ubench/permlane32_swap_builtin.hip), so
// the half of the exchange that carries the partner's value to lanes 0-31 is lost and `max(sw[0], sw[1])` is the maximum over the
// LOWER lane's 16 keys for both lanes of a pair.  That is still one reference point per query, so the softmax stays right until one
// of the other 16 keys exceeds it by 2^128 - rounds 3-5 shipped it; bench.py's peaked-logit case (q, k x 5) found it: inf / NaN
// rows.  s_nop 1: the wait states the compiler itself puts between a VALU write of an operand and the swap.
// v_max3_f32 / v_max_f32 as issued: fmaxf() canonicalises operands the compiler cannot prove quiet (every MFMA result) - a
// `v_max_f32 x, x, x` per input, a third of the running-maximum instructions of a key tile.  The scores are never signalling NaNs.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float lane_pair_max(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return vmax2(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}

// maximum of a 64-key tile's 32 scores per lane (two 32 x 32 accumulators), then over the lane pair: 16 v_max3 / v_max as issued
// + one swap, where fmaxf() over the same values is 32 maxima and 32 canonicalising ones
__device__ __forceinline__ float tile_max32(const float16_t (&sc)[2]) {
  float a = vmax3(sc[0][0], sc[0][1], sc[0][2]), b = vmax3(sc[1][0], sc[1][1], sc[1][2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) {
    a = vmax3(a, sc[0][r], sc[0][r + 1]);
    b = vmax3(b, sc[1][r], sc[1][r + 1]);
  }
  return lane_pair_max(vmax3(vmax2(a, sc[0][15]), b, sc[1][15]));
}

// (query tile, head, batch) of a workgroup.  The hardware deals consecutive workgroups round-robin to the 8 XCDs, each with
// its own L2: with the natural order the query tiles of one (batch, head) land on all 8 XCDs and every XCD streams every
// K / V^T from HBM (PMC: 2.1 GB per launch at L = 4096, B = 32 for 0.34 GB of Q + K + V + O).  Remapped, the workgroups
// an XCD receives walk the query tiles of ONE (batch, head) after another, whose K / V^T (0.65 MB) then stay in that L2.
__device__ __forceinline__ void sattn_block(int& qx, int& h, int& b) {
  const unsigned gx = gridDim.x, gy = gridDim.y, nbh = gridDim.y * gridDim.z;
  if ((nbh & 7u) == 0u) {
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = lin & 7u, slot = lin >> 3;
    const unsigned bh = (slot / gx) * 8u + xcd;
    qx = (int)(slot % gx);
    h = (int)(bh % gy);
    b = (int)(bh / gy);
  } else {
    qx = (int)blockIdx.x;
    h = (int)blockIdx.y;
    b = (int)blockIdx.z;
  }
}

// V [B, Lk, C] -> Vt [B, H, DVP, LkP]; 64 keys x 64 dims per workgroup through LDS
// ones_row >= 0: that (padding) row of V^T is set to 1.0 (`one`, in the element type) for the real keys, so the P V
// product accumulates the softmax denominator in that output row for free (k_sattn's sum_mfma path).
__global__ __launch_bounds__(256) void k_vt(const unsigned short* __restrict__ V, unsigned short* __restrict__ Vt, int H,
                                            int Lk, int dh, int DVP, int LkP, int ones_row, unsigned short one, long ld) {
  __shared__ unsigned short tile[64][66];
  const int b = blockIdx.z / H, h = blockIdx.z % H;
  const int k0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  {
    const int dv = tid & 63, kk = tid >> 6;          // coalesced along the head dims
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int key = k0 + kk + 4 * p;
      unsigned short v = 0;
      if (key < Lk && d0 + dv < dh) v = V[((size_t)b * Lk + key) * ld + (size_t)h * dh + d0 + dv];
      if (key < Lk && d0 + dv == ones_row) v = one;
      tile[kk + 4 * p][dv] = v;
    }
  }
  __syncthreads();
  {
    const int key = tid & 63, dd = tid >> 6;         // coalesced along the keys
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int dv = d0 + dd + 4 * p;
      if (dv < DVP) Vt[(((size_t)b * H + h) * DVP + dv) * LkP + k0 + key] = tile[key][dd + 4 * p];
    }
  }
}

// DHP: head dim padded to a multiple of 16.  QT: 32-row query tiles per wave (1 or 2).  With QT = 2 (dh <= 48: the
// accumulators of two tiles fit the register file at two waves per SIMD) every K / V^T fragment read from LDS feeds two
// MFMAs and a key tile's barrier covers twice the flops - the 4096-token layers were bound by LDS fragment traffic and
// barrier stalls (22 KB of LDS reads per wave and key tile at QT = 1), not by the matrix cores.
// VTI: no V^T pre-pass - `Vt` is V itself ([B, Lk, ld] rows like K) and the tile is transposed on its way into LDS (16-byte
// global loads, eight 2-byte LDS stores per load); the padding rows of the V^T image (and its row of ones) are written once.
// ld: row stride (elements) of Q, K and V - H * dh for separate tensors, 3 * H * dh for one packed projection.
// NBUF = 1: ONE K / V^T image in LDS (a second barrier per key tile, the next tile still prefetched into registers) under a
// two-workgroups-per-CU register cap - the dh = 160 layers (16 x 16 level: 4 key tiles per workgroup) ran one wave per SIMD with
// 86 KB of ring, every global load, barrier and softmax of a tile exposed.
// VTR (with VTI): V row-major in LDS + ds_read_b64_tr_b16 for the P V fragments (k_sattn_h's form); key rows VRS elements apart
// (a stride of 16 / 48 banks: the four key rows a 32-lane half reads fall on disjoint bank sets).
template <int DHP, bool F16, int QT, bool VTI, int NBUF = 2, bool VTR = false>
__global__ __launch_bounds__(256, NBUF == 1 ? 2 : 1) void k_sattn(const unsigned short* __restrict__ Q, const unsigned short* __restrict__ K,
                                               const unsigned short* __restrict__ Vt, unsigned short* __restrict__ O,
                                               int H, int Lq, int Lk, int LkP, int dh, float scale_log2e, long ld,
                                               unsigned short one, float lazy) {
  constexpr int NDV = (DHP + 31) / 32;      // output row tiles (of O^T)
  constexpr int DVP = NDV * 32;
  constexpr int KLD = DHP + 8;              // K tile row stride (elements): odd multiple of 16 B
  constexpr int VLD = KT + 4;               // V^T tile row stride (elements)
  constexpr int NS = DHP / 16;              // contraction steps of S^T = K Q^T
  constexpr int KCH = DHP / 8;              // 16-byte chunks per key row
  constexpr int NKL = (KT * KCH + 255) / 256;     // K chunks per thread per tile
  constexpr int VCH = KT / 8;               // 16-byte chunks per V^T row (8)
  constexpr int NVL = VTI ? NKL : (DVP * VCH + 255) / 256;    // V^T chunks per thread per tile (VTI: chunks of V rows, like K)
  static_assert(!VTR || VTI, "the transposing reads replace the inline transpose");
  constexpr int VRS = (DVP % 128 == 32 || DVP % 128 == 96) ? DVP : DVP + 32;
  constexpr int BUF = KT * KLD + (VTR ? KT * VRS : DVP * VLD);       // elements per buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // 2 buffers of BUF elements
  unsigned short* smem = (unsigned short*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int qx, h, b;
  sattn_block(qx, h, b);
  const float lazy_raw = lazy / scale_log2e;    // the lazy-maximum threshold in units of the raw scores (scale > 0)
  const int C = H * dh;
  const int lq = lane & 31, lh = lane >> 5;
  const long q0 = (long)qx * (128 * QT) + w * (32 * QT);

  // ---- this lane's Q fragments (rows q0 + 32 t + lq, dims 16s + 8*lh .. +7)
  uint4_t qf[QT][NS];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const long row = q0 + 32 * t + lq;
    const unsigned short* qrow = Q + ((size_t)b * Lq + (row < Lq ? row : Lq - 1)) * ld + (size_t)h * dh;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int dim = 16 * s + 8 * lh;
      qf[t][s] = (dim < dh) ? *(const uint4_t*)(qrow + dim) : (uint4_t){0u, 0u, 0u, 0u};
    }
  }

  const unsigned short* kbase = K + (size_t)b * Lk * ld + (size_t)h * dh;
  const unsigned short* vbase = VTI ? Vt + (size_t)b * Lk * ld + (size_t)h * dh : Vt + ((size_t)b * H + h) * DVP * LkP;
  uint4_t rk[NKL], rv[NVL];
  // K (and the inline V) tile by buffer loads (see k_sattn_h): keys >= Lk fall off the descriptor's end, chunks of dims >= dh carry
  // an out-of-range offset - zeros either way (the padding keys carry P = 0)
  unsigned kv_off[NKL];
#pragma unroll
  for (int i = 0; i < NKL; ++i) {
    const int e = tid + 256 * i;
    const int key = e / KCH, dim = (e - key * KCH) * 8;
    kv_off[i] = (e < KT * KCH && dim < dh) ? (unsigned)(((long)key * ld + dim) * 2) : 0x80000000u;
  }
  auto tile_rsrc = [&](const unsigned short* base, int t) {
    const long left = ((long)(Lk - t * KT - 1) * ld + dh) * 2;         // bytes up to the end of the last key's head slice
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)t * KT * ld), 0, (int)(left < 0x7fffffffL ? left : 0x7fffffffL),
                                             0x00020000);
  };
  auto g_load = [&](int t) {
    const int key0 = t * KT;
    {
      const __amdgpu_buffer_rsrc_t r = tile_rsrc(kbase, t);
#pragma unroll
      for (int i = 0; i < NKL; ++i) rk[i] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(r, kv_off[i], 0, 0));
    }
    if constexpr (VTI) {
      const __amdgpu_buffer_rsrc_t r = tile_rsrc(vbase, t);
#pragma unroll
      for (int i = 0; i < NVL; ++i) rv[i] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(r, kv_off[i], 0, 0));
    } else {
#pragma unroll
      for (int i = 0; i < NVL; ++i) {
        const int e = tid + 256 * i;
        const int dv = e / VCH, kc = (e - dv * VCH) * 8;
        if (e < DVP * VCH) rv[i] = *(const uint4_t*)(vbase + (size_t)dv * LkP + key0 + kc);   // zero padded by k_vt
      }
    }
  };
  auto s_store = [&](int buf) {
    unsigned short* Ks = smem + buf * BUF;
    unsigned short* Vs = Ks + KT * KLD;
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
      const int e = tid + 256 * i;
      const int key = e / KCH, dim = (e - key * KCH) * 8;
      if (e < KT * KCH) *(uint4_t*)(Ks + key * KLD + dim) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int e = tid + 256 * i;
      if constexpr (VTR) {                     // as it arrived: 8 dims of one key
        const int key = e / KCH, dim = (e - key * KCH) * 8;
        if (e < KT * KCH && dim < dh) *(uint4_t*)(Vs + key * VRS + dim) = rv[i];
      } else if constexpr (VTI) {              // transposed: element q of the chunk -> row dim + q, column key
        const int key = e / KCH, dim = (e - key * KCH) * 8;
        if (e < KT * KCH && dim < dh) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            Vs[(dim + 2 * q) * VLD + key] = (unsigned short)(rv[i][q] & 0xffffu);
            Vs[(dim + 2 * q + 1) * VLD + key] = (unsigned short)(rv[i][q] >> 16);
          }
        }
      } else {
        const int dv = e / VCH, kc = (e - dv * VCH) * 8;
        if (e < DVP * VCH) {
          // VLD * 2 bytes is a multiple of 8 but not of 16: two 8-byte stores
          *(uint2_t*)(Vs + dv * VLD + kc) = (uint2_t){rv[i][0], rv[i][1]};
          *(uint2_t*)(Vs + dv * VLD + kc + 4) = (uint2_t){rv[i][2], rv[i][3]};
        }
      }
    }
  };
  if constexpr (VTR) {
    for (int e = tid; e < NBUF * DVP * KT; e += 256) {          // the padding dims of every key row: zeros, dim DVP - 1 = 1
      const int buf = e / (DVP * KT), rem = e - buf * (DVP * KT);
      const int key = rem / DVP, dv = rem - key * DVP;
      if (dv >= dh) smem[buf * BUF + KT * KLD + key * VRS + dv] = (dv == DVP - 1) ? one : (unsigned short)0;
    }
  } else if constexpr (VTI) {
    // rows dh .. DVP - 1 of both V^T images never change: zeros, and ones in the last row when it is a padding row
    for (int e = tid; e < NBUF * (DVP - 0) * KT; e += 256) {
      const int buf = e / (DVP * KT), rem = e - buf * (DVP * KT);
      const int dv = rem / KT, key = rem - dv * KT;
      if (dv >= dh) smem[buf * BUF + KT * KLD + dv * VLD + key] = (dv == DVP - 1) ? one : (unsigned short)0;
    }
  }
  const int tr_off = (4 * lh + ((lane & 15) >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  // dh < DVP: row DVP - 1 of V^T is all ones (k_vt), so O^T's last row IS the running softmax denominator - summed
  // by the matrix core, rescaled with the other rows - and the 32 VALU adds per tile go away.
  const bool sum_mfma = dh < DVP;
  float m[QT], lsum[QT];                    // running max (shared by lane and lane^32), this lane's partial sum
  float16_t oacc[QT][NDV];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -INFINITY;
    lsum[t] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][nt][r] = 0.f;
  }

  const int ntiles = (Lk + KT - 1) / KT;
  g_load(0);
  s_store(0);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = NBUF == 2 ? (kt & 1) : 0;
    const unsigned short* Ks = smem + cur * BUF;
    const unsigned short* Vs = Ks + KT * KLD;
    if (kt + 1 < ntiles) g_load(kt + 1);

    // ---- S^T = K Q^T for the 64 keys: register r of sub-tile j = key 32j + (r&3) + 8*(r>>2) + 4*lh
    float16_t sacc[QT][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[t][j][r] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const uint4_t kf = *(const uint4_t*)(Ks + (j * 32 + lq) * KLD + 16 * s + 8 * lh);
#pragma unroll
        for (int t = 0; t < QT; ++t) sacc[t][j] = mfma32<F16>(kf, qf[t][s], sacc[t][j]);
      }
    }
    if ((kt + 1) * KT > Lk) {                 // the last tile may hold padding keys
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt * KT + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sacc[t][j][r] = (key < Lk) ? sacc[t][j][r] : -INFINITY;
          }
    }
    // ---- online softmax, per query tile
    uint4_t pf[QT][2][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const float mt = tile_max32(sacc[t]);
      // LAZY running maximum: raised only when a tile's maximum is more than `lazy` powers of two above it, so exp2's argument
      // stays <= lazy (P <= 2^lazy: the same relative precision in bf16 / f16, f32 sums) and the rescale of O below is rare.  With
      // the exact running maximum SOME lane's maximum moves in almost every tile of a 64-query wave (probability
      // 1 - (1 - 1/t)^64 at tile t for exchangeable scores), so the "rare" branch ran nearly always.
      const float m_new = (mt > m[t] + lazy_raw) ? mt : m[t];         // finite: every tile holds at least one real key
      const float alpha = __builtin_amdgcn_exp2f((m[t] - m_new) * scale_log2e);   // exp2(-inf) = 0 on the first tile
      const float mc = m_new * scale_log2e;
      m[t] = m_new;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sacc[t][j][r] = __builtin_amdgcn_exp2f(fmaf(sacc[t][j][r], scale_log2e, -mc));
      if (!sum_mfma) {
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) ps += sacc[t][j][r];
        lsum[t] = fmaf(lsum[t], alpha, ps);
      }
      if (__any(alpha != 1.0f)) {               // the max rarely moves after the first tiles: skip the rescale
#pragma unroll
        for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[t][nt][r] *= alpha;
      }
      // ---- P fragments (unnormalised): slot e of step s2 of sub-tile j <- register 8*s2 + e
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int q = 0; q < 4; ++q) pf[t][j][s2][q] = pack2<F16>(sacc[t][j][8 * s2 + 2 * q], sacc[t][j][8 * s2 + 2 * q + 1]);
    }
    // ---- O^T += V^T P^T : slot e <-> key 32j + 16*s2 + 4*lh + (e&3) + 8*(e>>2) on both sides
#pragma unroll
    for (int nt = 0; nt < NDV; ++nt) {
      const unsigned short* vrow = Vs + (nt * 32 + lq) * VLD + 4 * lh;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          uint4_t vf;
          if constexpr (VTR) {                  // keys 32 j + 16 s2 + 4 lh + {0..3, 8..11} of dim 32 nt + lq
            typedef short v4s_t __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;
            const unsigned short* a = Vs + tr_off + (32 * j + 16 * s2) * VRS + 32 * nt;
            const uint2_t l2 = __builtin_bit_cast(uint2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)a));
            const uint2_t h2 = __builtin_bit_cast(uint2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(a + 8 * VRS)));
            vf = (uint4_t){l2[0], l2[1], h2[0], h2[1]};
          } else {
            const uint2_t lo = *(const uint2_t*)(vrow + j * 32 + 16 * s2);
            const uint2_t hi = *(const uint2_t*)(vrow + j * 32 + 16 * s2 + 8);
            vf = (uint4_t){lo[0], lo[1], hi[0], hi[1]};
          }
#pragma unroll
          for (int t = 0; t < QT; ++t) oacc[t][nt] = mfma32<F16>(vf, pf[t][j][s2], oacc[t][nt]);
        }
    }
    if (kt + 1 < ntiles) {
      if constexpr (NBUF == 1) __syncthreads();    // every wave is done reading the one image
      s_store(NBUF == 2 ? (cur ^ 1) : 0);
    }
    __syncthreads();
  }

  // ---- 1/sum, convert, store: register r of tile nt = output dim nt*32 + (r&3) + 8*(r>>2) + 4*lh
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const long row = q0 + 32 * t + lq;
    float denom = lsum[t] + __shfl_xor(lsum[t], 32);
    if (sum_mfma) {                           // row DVP - 1 = tile NDV - 1, register 15 of the lh = 1 lanes
      const float l1 = oacc[t][NDV - 1][15];
      const float l0 = __shfl_xor(l1, 32);
      denom = lh ? l1 : l0;
    }
    const float inv = 1.0f / denom;
    unsigned short* orow = O + ((size_t)b * Lq + (row < Lq ? row : Lq - 1)) * C + (size_t)h * dh;
#pragma unroll
    for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = nt * 32 + 8 * g + 4 * lh;
        if (dv < dh && row < Lq) {
          const uint2_t o2 = {pack2<F16>(oacc[t][nt][4 * g] * inv, oacc[t][nt][4 * g + 1] * inv),
                              pack2<F16>(oacc[t][nt][4 * g + 2] * inv, oacc[t][nt][4 * g + 3] * inv)};
          *(uint2_t*)(orow + dv) = o2;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// k_sattn_p: the same kernel software-pipelined over the key tiles (the att[2] double pipeline of the CDNA guide, T15):
// iteration t issues the S^T MFMAs of tile t + 1 FIRST, then runs the softmax of tile t on the VALU while they execute,
// then the P V MFMAs of tile t - inside one wave the MFMA -> VALU -> MFMA dependency chain of a tile (what kept the
// matrix pipe at a third of its time in k_sattn: neither pipe saturated, the wave waiting on its own previous phase)
// is broken across two tiles.  K is therefore staged one tile further ahead than V^T: separate double buffers.
// ---------------------------------------------------------------------------------------------
// (forced to three waves per SIMD at dh = 40 - 168 VGPRs, 22 spilled - it ran 2071 us against 1747 at two waves and 1580 for
//  k_sattn with two query tiles per wave: measured in round 3, not adopted)
// VTR (with VTI; DVP = 64 or 96): V stays row-major in LDS and the P V fragments come out of ds_read_b64_tr_b16, as in k_sattn_h -
// three 16-byte stores per thread and tile in place of 24 scattered 2-byte ones.
template <int DHP, bool F16, bool VTI, bool VTR = false>
__global__ __launch_bounds__(256) void k_sattn_p(const unsigned short* __restrict__ Q, const unsigned short* __restrict__ K,
                                                 const unsigned short* __restrict__ Vt, unsigned short* __restrict__ O,
                                                 int H, int Lq, int Lk, int LkP, int dh, float scale_log2e, long ld,
                                                 unsigned short one, float lazy) {
  constexpr int NDV = (DHP + 31) / 32;
  constexpr int DVP = NDV * 32;
  constexpr int KLD = DHP + 8;
  constexpr int VLD = KT + 4;
  constexpr int NS = DHP / 16;
  constexpr int KCH = DHP / 8;
  constexpr int NKL = (KT * KCH + 255) / 256;
  constexpr int VCH = KT / 8;
  constexpr int NVL = VTI ? NKL : (DVP * VCH + 255) / 256;
  static_assert(!VTR || (VTI && (DVP == 64 || DVP == 96)), "row-major V: the inline form, key rows 192 bytes apart");
  constexpr int VRS = 96;                                              // VTR: elements per key row of V (48 banks: see k_sattn_h)
  constexpr int VB = VTR ? KT * VRS : DVP * VLD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Kbuf = (unsigned short*)smem_raw;                  // [2][KT * KLD]
  unsigned short* Vbuf = Kbuf + 2 * KT * KLD;                        // [2][VB]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int qx, h, b;
  sattn_block(qx, h, b);
  const float lazy_raw = lazy / scale_log2e;    // the lazy-maximum threshold in units of the raw scores (scale > 0)
  const int C = H * dh;
  const int lq = lane & 31, lh = lane >> 5;
  const long row = (long)qx * 128 + w * 32 + lq;

  uint4_t qf[NS];
  {
    const unsigned short* qrow = Q + ((size_t)b * Lq + (row < Lq ? row : Lq - 1)) * ld + (size_t)h * dh;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int dim = 16 * s + 8 * lh;
      qf[s] = (dim < dh) ? *(const uint4_t*)(qrow + dim) : (uint4_t){0u, 0u, 0u, 0u};
    }
  }
  const unsigned short* kbase = K + (size_t)b * Lk * ld + (size_t)h * dh;
  const unsigned short* vbase = VTI ? Vt + (size_t)b * Lk * ld + (size_t)h * dh : Vt + ((size_t)b * H + h) * DVP * LkP;
  uint4_t rk[NKL], rv[NVL];
  // K (and the inline V) tile by buffer loads, as in k_sattn_h: a fixed per-thread byte offset against a descriptor that moves with
  // the tile; keys >= Lk fall off its end and chunks of dims >= dh carry an out-of-range offset - both come back as zeros
  unsigned kv_off[NKL];
#pragma unroll
  for (int i = 0; i < NKL; ++i) {
    const int e = tid + 256 * i;
    const int key = e / KCH, dim = (e - key * KCH) * 8;
    kv_off[i] = (e < KT * KCH && dim < dh) ? (unsigned)(((long)key * ld + dim) * 2) : 0x80000000u;
  }
  auto tile_rsrc = [&](const unsigned short* base, int t) {
    const long left = ((long)(Lk - t * KT - 1) * ld + dh) * 2;         // bytes up to the end of the last key's head slice
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)t * KT * ld), 0, (int)(left < 0x7fffffffL ? left : 0x7fffffffL),
                                             0x00020000);
  };
  auto g_load_k = [&](int t) {
    const __amdgpu_buffer_rsrc_t r = tile_rsrc(kbase, t);
#pragma unroll
    for (int i = 0; i < NKL; ++i) rk[i] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(r, kv_off[i], 0, 0));
  };
  auto g_load_v = [&](int t) {
    if constexpr (VTI) {
      const __amdgpu_buffer_rsrc_t r = tile_rsrc(vbase, t);
#pragma unroll
      for (int i = 0; i < NVL; ++i) rv[i] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(r, kv_off[i], 0, 0));
    } else {
      const int key0 = t * KT;
#pragma unroll
      for (int i = 0; i < NVL; ++i) {
        const int e = tid + 256 * i;
        const int dv = e / VCH, kc = (e - dv * VCH) * 8;
        if (e < DVP * VCH) rv[i] = *(const uint4_t*)(vbase + (size_t)dv * LkP + key0 + kc);   // zero padded by k_vt
      }
    }
  };
  auto s_store_k = [&](int buf) {
    unsigned short* Ks = Kbuf + buf * KT * KLD;
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
      const int e = tid + 256 * i;
      const int key = e / KCH, dim = (e - key * KCH) * 8;
      if (e < KT * KCH) *(uint4_t*)(Ks + key * KLD + dim) = rk[i];
    }
  };
  auto s_store_v = [&](int buf) {
    unsigned short* Vs = Vbuf + buf * VB;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int e = tid + 256 * i;
      if constexpr (VTR) {
        const int key = e / KCH, dim = (e - key * KCH) * 8;
        if (e < KT * KCH && dim < dh) *(uint4_t*)(Vs + key * VRS + dim) = rv[i];
      } else if constexpr (VTI) {
        const int key = e / KCH, dim = (e - key * KCH) * 8;
        if (e < KT * KCH && dim < dh) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            Vs[(dim + 2 * q) * VLD + key] = (unsigned short)(rv[i][q] & 0xffffu);
            Vs[(dim + 2 * q + 1) * VLD + key] = (unsigned short)(rv[i][q] >> 16);
          }
        }
      } else {
        const int dv = e / VCH, kc = (e - dv * VCH) * 8;
        if (e < DVP * VCH) {
          *(uint2_t*)(Vs + dv * VLD + kc) = (uint2_t){rv[i][0], rv[i][1]};
          *(uint2_t*)(Vs + dv * VLD + kc + 4) = (uint2_t){rv[i][2], rv[i][3]};
        }
      }
    }
  };
  if constexpr (VTR) {
    for (int e = tid; e < 2 * DVP * KT; e += 256) {                    // the padding dims of every key row: zeros, dim DVP - 1 = 1
      const int buf = e / (DVP * KT), rem = e - buf * (DVP * KT);
      const int key = rem / DVP, dv = rem - key * DVP;
      if (dv >= dh) Vbuf[buf * VB + key * VRS + dv] = (dv == DVP - 1) ? one : (unsigned short)0;
    }
  } else if constexpr (VTI) {
    for (int e = tid; e < 2 * DVP * KT; e += 256) {
      const int buf = e / (DVP * KT), rem = e - buf * (DVP * KT);
      const int dv = rem / KT, key = rem - dv * KT;
      if (dv >= dh) Vbuf[buf * VB + dv * VLD + key] = (dv == DVP - 1) ? one : (unsigned short)0;
    }
  }
  const int tr_off = (4 * lh + ((lane & 15) >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  // S^T = K Q^T of the tile in K buffer `buf`
  auto qk = [&](int buf, float16_t (&sacc)[2]) {
    const unsigned short* Ks = Kbuf + buf * KT * KLD;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[j][r] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const uint4_t kf = *(const uint4_t*)(Ks + (j * 32 + lq) * KLD + 16 * s + 8 * lh);
        sacc[j] = mfma32<F16>(kf, qf[s], sacc[j]);
      }
    }
  };

  const bool sum_mfma = dh < DVP;
  float m = -INFINITY, lsum = 0.f;
  float16_t oacc[NDV];
#pragma unroll
  for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[nt][r] = 0.f;

  const int ntiles = (Lk + KT - 1) / KT;
  // prologue: K(0), V^T(0), K(1) staged; S^T(0) computed
  g_load_k(0);
  g_load_v(0);
  s_store_k(0);
  s_store_v(0);
  if (ntiles > 1) {
    g_load_k(1);
    s_store_k(1);
  }
  __syncthreads();
  float16_t sA[2], sB[2];
  qk(0, sA);
  __syncthreads();                              // K(0) is overwritten by iteration 0's store of K(2)

  // one key tile: sc = S^T(t) (computed one iteration earlier), sn <- S^T(t + 1)
  auto tile = [&](int t, float16_t (&sc)[2], float16_t (&sn)[2]) {
    const int cur = t & 1;
    if (t + 2 < ntiles) g_load_k(t + 2);
    if (t + 1 < ntiles) {
      g_load_v(t + 1);
      qk(cur ^ 1, sn);                          // the next tile's scores: in flight under this tile's softmax
    }
    if ((t + 1) * KT > Lk) {                    // the last tile may hold padding keys
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KT + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          sc[j][r] = (key < Lk) ? sc[j][r] : -INFINITY;
        }
    }
    const float mt = tile_max32(sc);
    const float m_new = (mt > m + lazy_raw) ? mt : m;          // lazy running maximum (see k_sattn)
    const float alpha = __builtin_amdgcn_exp2f((m - m_new) * scale_log2e);
    const float mc = m_new * scale_log2e;
    m = m_new;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[j][r] = __builtin_amdgcn_exp2f(fmaf(sc[j][r], scale_log2e, -mc));
    if (!sum_mfma) {
      float ps = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ps += sc[j][r];
      lsum = fmaf(lsum, alpha, ps);
    }
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[nt][r] *= alpha;
    }
    uint4_t pf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int q = 0; q < 4; ++q) pf[j][s2][q] = pack2<F16>(sc[j][8 * s2 + 2 * q], sc[j][8 * s2 + 2 * q + 1]);
    const unsigned short* Vs = Vbuf + cur * VB;
#pragma unroll
    for (int nt = 0; nt < NDV; ++nt) {
      const unsigned short* vrow = Vs + (nt * 32 + lq) * VLD + 4 * lh;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          uint4_t vf;
          if constexpr (VTR) {                  // keys 32 j + 16 s2 + 4 lh + {0..3, 8..11} of dim 32 nt + lq
            typedef short v4s_t __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;
            const unsigned short* a = Vs + tr_off + (32 * j + 16 * s2) * VRS + 32 * nt;
            const uint2_t l2 = __builtin_bit_cast(uint2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)a));
            const uint2_t h2 = __builtin_bit_cast(uint2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(a + 8 * VRS)));
            vf = (uint4_t){l2[0], l2[1], h2[0], h2[1]};
          } else {
            const uint2_t lo = *(const uint2_t*)(vrow + j * 32 + 16 * s2);
            const uint2_t hi = *(const uint2_t*)(vrow + j * 32 + 16 * s2 + 8);
            vf = (uint4_t){lo[0], lo[1], hi[0], hi[1]};
          }
          oacc[nt] = mfma32<F16>(vf, pf[j][s2], oacc[nt]);
        }
    }
    // K(t + 2) -> the buffer K(t) left one iteration ago; V^T(t + 1) -> the buffer V^T(t - 1) left
    if (t + 2 < ntiles) s_store_k(cur);
    if (t + 1 < ntiles) s_store_v(cur ^ 1);
    __syncthreads();
  };
  for (int t = 0; t < ntiles; t += 2) {
    tile(t, sA, sB);
    if (t + 1 < ntiles) tile(t + 1, sB, sA);
  }

  float denom = lsum + __shfl_xor(lsum, 32);
  if (sum_mfma) {
    const float l1 = oacc[NDV - 1][15];
    const float l0 = __shfl_xor(l1, 32);
    denom = lh ? l1 : l0;
  }
  const float inv = 1.0f / denom;
  unsigned short* orow = O + ((size_t)b * Lq + (row < Lq ? row : Lq - 1)) * C + (size_t)h * dh;
#pragma unroll
  for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dv = nt * 32 + 8 * g + 4 * lh;
      if (dv < dh && row < Lq) {
        const uint2_t o2 = {pack2<F16>(oacc[nt][4 * g] * inv, oacc[nt][4 * g + 1] * inv),
                            pack2<F16>(oacc[nt][4 * g + 2] * inv, oacc[nt][4 * g + 3] * inv)};
        *(uint2_t*)(orow + dv) = o2;
      }
    }
}

// ---------------------------------------------------------------------------------------------
// k_sattn_h: two query tiles per wave (k_sattn's QT = 2: every K / V^T fragment read from LDS feeds two MFMAs), pipelined
// over HALF key tiles of 32 keys so that both pipes have work from different halves in every stretch of the stream:
//   sub-step h:   [ P V of half h - 1  |  running max of half h ]   ->  (rarely: rescale O)  ->
//                 [ K Q^T of half h + 1  |  exp2 + convert of half h ]
// The MFMAs of each bracket do not depend on its VALU work and the instruction stream alternates them explicitly
// (sched_group_barrier): k_sattn ran max -> exp -> P V -> K Q^T in sequence inside a wave and the two waves of a SIMD fell into
// step with each other (both in their softmax, then both in their MFMAs: 2800 cycles per 64 x 64 tile for 1170 of VALU and
// 896 of matrix pipe).  P V trails by one half so the rescale of O still covers it; K is staged two tiles ahead and V^T
// one behind: three LDS buffers each.  dh < DVP only (the denominator comes out of the ones row of V^T).
// ---------------------------------------------------------------------------------------------
// VTR (with VTI): V stays ROW-MAJOR in LDS ([key][64 dims], 192-byte rows) - stored as it arrives, 16 bytes per lane - and the P V
// fragments (4 consecutive keys of one dim per lane) come out of `ds_read_b64_tr_b16`, gfx950's transposing LDS read: per 16-lane
// group the hardware reads a [4 keys][16 dims] block (lane 4 j + c supplies the address of key j, dims 4 c .. 4 c + 3) and hands
// lane l the column l & 15.  Replaces the inline transpose's eight 2-byte scattered stores per 16-byte chunk (LDS bank conflicts on
// 0.32 of the LDS cycles, profiles/r04/session2_sattn_h_pmc_sq.txt).  The 192-byte row stride puts the four key rows of a
// 32-lane half on four disjoint 16-bank sets ((a / 4) % 64: 0, 48, 32, 16); the ones column (dim DVP - 1) replaces the ones row.
// REL (exp2-domain scores; launched for dh = DHP - 8 only): Q arrives carrying scale * log2(e) already - from the epilogue of the
// projection that produced it (uce_linear_colscale_fwd: ONE rounding of the f32 product, so nothing is lost against rounding q and
// scaling the f32 score) - and the head's first padding dim holds 1.0 on the K side and minus the running maximum on the Q side:
// a score leaves the matrix pipe as the ARGUMENT of exp2, no v_fma_f32 per element in the softmax (the loop is bound by VALU
// issue).  The maximum is only a reference point (O and the denominator are formed with the same one): it is kept rounded to the
// element type, so its product with 1.0 is exact.  (Round 6 built this with the scale applied to the bf16 Q fragments inside the
// kernel - a second rounding of q: 2.3e-3 -> 3.0e-3 against fp64, retired; the scale belongs in the producer.)
template <int DHP, bool F16, bool VTI, bool VTR = false, bool REL = false>
__global__ __launch_bounds__(256, 2) void k_sattn_h(const unsigned short* __restrict__ Q, const unsigned short* __restrict__ K,
                                                 const unsigned short* __restrict__ Vt, unsigned short* __restrict__ O,
                                                 int H, int Lq, int Lk, int LkP, int dh, float scale_log2e, long ld,
                                                 unsigned short one, float lazy) {
  constexpr int NDV = (DHP + 31) / 32;
  constexpr int DVP = NDV * 32;
  static_assert(DHP < DVP, "the softmax denominator rides in the padding row of V^T");
  constexpr int KLD = DHP + 8;
  constexpr int VLD = KT + 4;
  constexpr int NS = DHP / 16;
  constexpr int KCH = DHP / 8;
  constexpr int NKL = (KT * KCH + 255) / 256;
  constexpr int VCH = KT / 8;
  constexpr int NVL = VTI ? NKL : (DVP * VCH + 255) / 256;
  static_assert(!VTR || VTI, "the transposing reads replace the inline transpose");
  constexpr int VRS = 96;                                              // VTR: elements per key row of V (64 dims + 32: see above)
  // REL: the Q-side slot of dim dh = DHP - 8 - fragment QSS, first element, in the lanes of half QSH
  constexpr int QSS = (DHP - 8) / 16, QSH = ((DHP - 8) / 8) & 1;
  constexpr int KB = KT * KLD, VB = VTR ? KT * VRS : DVP * VLD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Kbuf = (unsigned short*)smem_raw;                  // [3][KB]
  unsigned short* Vbuf = Kbuf + 3 * KB;                              // [3][VB]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int qx, h, b;
  sattn_block(qx, h, b);
  const float lazy_raw = lazy / scale_log2e;    // the lazy-maximum threshold in units of the raw scores (scale > 0)
  const int C = H * dh;
  const int lq = lane & 31, lh = lane >> 5;
  const long q0 = (long)qx * 256 + w * 64;

  uint4_t qf[2][NS];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const long row = q0 + 32 * x + lq;
    const unsigned short* qrow = Q + ((size_t)b * Lq + (row < Lq ? row : Lq - 1)) * ld + (size_t)h * dh;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int dim = 16 * s + 8 * lh;
      qf[x][s] = (dim < dh) ? *(const uint4_t*)(qrow + dim) : (uint4_t){0u, 0u, 0u, 0u};
    }
  }
  const unsigned short* kbase = K + (size_t)b * Lk * ld + (size_t)h * dh;
  const unsigned short* vbase = VTI ? Vt + (size_t)b * Lk * ld + (size_t)h * dh : Vt + ((size_t)b * H + h) * DVP * LkP;
  uint4_t rk[NKL], rv[NVL];
  // K (and the inline V) tile by BUFFER loads: the thread's chunk (key, dim) of a tile is a fixed byte offset, the tile's first key
  // row moves the descriptor's base (scalar arithmetic), keys >= Lk fall off the end of its range and chunks of dims >= dh carry an
  // out-of-range offset - both come back as zeros.  (As plain global loads the same tile took ~50 vector instructions of address
  // work, zero fills and 12 exec-mask branches per wave and key tile - a tenth of the loop's VALU issue.)
  unsigned kv_off[NKL];
  unsigned k_one[NKL];                        // REL: the 1.0 that meets -max on the Q side, OR-ed into the (zero) chunk of dim dh on its way to LDS
#pragma unroll
  for (int i = 0; i < NKL; ++i) {
    const int e = tid + 256 * i;
    const int key = e / KCH, dim = (e - key * KCH) * 8;
    kv_off[i] = (e < KT * KCH && dim < dh) ? (unsigned)(((long)key * ld + dim) * 2) : 0x80000000u;
    k_one[i] = (REL && e < KT * KCH && dim == dh) ? (unsigned)one : 0u;
  }
  auto tile_rsrc = [&](const unsigned short* base, int t) {
    const long left = ((long)(Lk - t * KT - 1) * ld + dh) * 2;         // bytes up to the end of the last key's head slice
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)t * KT * ld), 0, (int)(left < 0x7fffffffL ? left : 0x7fffffffL),
                                             0x00020000);
  };
  auto g_load_k = [&](int t) {
    const __amdgpu_buffer_rsrc_t r = tile_rsrc(kbase, t);
#pragma unroll
    for (int i = 0; i < NKL; ++i) rk[i] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(r, kv_off[i], 0, 0));
  };
  auto g_load_v = [&](int t) {
    if constexpr (VTI) {
      const __amdgpu_buffer_rsrc_t r = tile_rsrc(vbase, t);
#pragma unroll
      for (int i = 0; i < NVL; ++i) rv[i] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(r, kv_off[i], 0, 0));
    } else {
      const int key0 = t * KT;
#pragma unroll
      for (int i = 0; i < NVL; ++i) {
        const int e = tid + 256 * i;
        const int dv = e / VCH, kc = (e - dv * VCH) * 8;
        if (e < DVP * VCH) rv[i] = *(const uint4_t*)(vbase + (size_t)dv * LkP + key0 + kc);   // zero padded by k_vt
      }
    }
  };
  auto s_store_k = [&](int buf) {
    unsigned short* Ks = Kbuf + buf * KB;
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
      const int e = tid + 256 * i;
      const int key = e / KCH, dim = (e - key * KCH) * 8;
      if constexpr (REL) rk[i][0] |= k_one[i];
      if (e < KT * KCH) *(uint4_t*)(Ks + key * KLD + dim) = rk[i];
    }
  };
  auto s_store_v = [&](int buf) {
    unsigned short* Vs = Vbuf + buf * VB;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int e = tid + 256 * i;
      if constexpr (VTR) {
        const int key = e / KCH, dim = (e - key * KCH) * 8;
        if (e < KT * KCH && dim < dh) *(uint4_t*)(Vs + key * VRS + dim) = rv[i];
      } else if constexpr (VTI) {
        const int key = e / KCH, dim = (e - key * KCH) * 8;
        if (e < KT * KCH && dim < dh) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            Vs[(dim + 2 * q) * VLD + key] = (unsigned short)(rv[i][q] & 0xffffu);
            Vs[(dim + 2 * q + 1) * VLD + key] = (unsigned short)(rv[i][q] >> 16);
          }
        }
      } else {
        const int dv = e / VCH, kc = (e - dv * VCH) * 8;
        if (e < DVP * VCH) {
          *(uint2_t*)(Vs + dv * VLD + kc) = (uint2_t){rv[i][0], rv[i][1]};
          *(uint2_t*)(Vs + dv * VLD + kc + 4) = (uint2_t){rv[i][2], rv[i][3]};
        }
      }
    }
  };
  if constexpr (VTR) {
    for (int e = tid; e < 3 * DVP * KT; e += 256) {                    // the padding dims of every key row: zeros, dim DVP - 1 = 1
      const int buf = e / (DVP * KT), rem = e - buf * (DVP * KT);
      const int key = rem / DVP, dv = rem - key * DVP;
      if (dv >= dh) Vbuf[buf * VB + key * VRS + dv] = (dv == DVP - 1) ? one : (unsigned short)0;
    }
  } else if constexpr (VTI) {
    for (int e = tid; e < 3 * DVP * KT; e += 256) {
      const int buf = e / (DVP * KT), rem = e - buf * (DVP * KT);
      const int dv = rem / KT, key = rem - dv * KT;
      if (dv >= dh) Vbuf[buf * VB + dv * VLD + key] = (dv == DVP - 1) ? one : (unsigned short)0;
    }
  }
  // the P V fragment of dim tile nt, key group s2 of a 32-key half whose first key row / column is at Vp:
  // keys 16 s2 + 4 lh + {0..3, 8..11} of dim 32 nt + lq (the key order of the P fragments)
  const int tr_off = (4 * lh + ((lane & 15) >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  auto v_frag = [&](const unsigned short* Vp, int nt, int s2) -> uint4_t {
    if constexpr (VTR) {
      typedef short v4s_t __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;
      const unsigned short* a = Vp + tr_off + 16 * s2 * VRS + 32 * nt;
      const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)a);
      const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(a + 8 * VRS));
      const uint2_t l2 = __builtin_bit_cast(uint2_t, lo), h2 = __builtin_bit_cast(uint2_t, hi);
      return (uint4_t){l2[0], l2[1], h2[0], h2[1]};
    } else {
      const unsigned short* vrow = Vp + (nt * 32 + lq) * VLD + 4 * lh + 16 * s2;
      const uint2_t lo = *(const uint2_t*)(vrow);
      const uint2_t hi = *(const uint2_t*)(vrow + 8);
      return (uint4_t){lo[0], lo[1], hi[0], hi[1]};
    }
  };
  constexpr int VHALF = VTR ? 32 * VRS : 32;                           // the upper 32 keys of a V buffer

  float m[2] = {-INFINITY, -INFINITY};
  float16_t oacc[2][NDV];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[x][nt][r] = 0.f;
  uint4_t pf[2][2];                           // P fragments of the half before the current one (zero before the first)
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) pf[x][s2] = (uint4_t){0u, 0u, 0u, 0u};

  const int ntiles = (Lk + KT - 1) / KT;
  g_load_k(0);
  g_load_v(0);
  s_store_k(0);
  s_store_v(0);
  if (ntiles > 1) {
    g_load_k(1);
    s_store_k(1);
  }
  __syncthreads();
  float16_t sA[2], sB[2];
  {
    const unsigned short* Ks = Kbuf + lq * KLD + 8 * lh;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) sA[x][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const uint4_t kf = *(const uint4_t*)(Ks + 16 * s);
#pragma unroll
      for (int x = 0; x < 2; ++x) sA[x] = mfma32<F16>(kf, qf[x][s], sA[x]);
    }
  }

  // one sub-step, in two parts.  head: sc = S^T of half hh (complete; MASK: the tile may hold padding keys - only the last
  // does); P V of half hh - 1 from the 32 key columns at Vp beside the running max of half hh.  tail: sn <- S^T of half
  // hh + 1 (rows of the K buffer at Kn; NEXT = false: there is none) beside exp2 and the P fragments of half hh.
  float mc[2] = {0.f, 0.f};
  auto sub_head = [&](auto mask_c, int hh, float16_t (&sc)[2], const unsigned short* Vp) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_c)::value;
    if constexpr (MASK) {
      if ((hh + 1) * 32 > Lk) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = hh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sc[x][r] = (key < Lk) ? sc[x][r] : -INFINITY;
          }
      }
    }
    // ---- bracket 1: O^T += V^T P^T of the previous half | running max of this one.  The source order IS the issue order:
    // one MFMA, a slice of the VALU work, a scheduling fence (left to itself the scheduler emits the VALU work first and the
    // MFMAs in one run behind it, and sched_group_barrier pipelines did not change that).
    float mt[2];
    {
      uint4_t vf[NDV][2];
#pragma unroll
      for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) vf[nt][s2] = v_frag(Vp, nt, s2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4 * NDV; ++i) {
        const int nt = i >> 2, s2 = (i >> 1) & 1, x = i & 1;
        oacc[x][nt] = mfma32<F16>(vf[nt][s2], pf[x][s2], oacc[x][nt]);
        if (i < 2) {                            // the tile maximum of query tile i (both key halves of the lane pair)
          float ma = vmax3(sc[i][0], sc[i][1], sc[i][2]), mb = vmax3(sc[i][8], sc[i][9], sc[i][10]);
#pragma unroll
          for (int r = 3; r < 7; r += 2) {
            ma = vmax3(ma, sc[i][r], sc[i][r + 1]);
            mb = vmax3(mb, sc[i][8 + r], sc[i][9 + r]);
          }
          ma = vmax3(ma, sc[i][7], sc[i][15]);
          mt[i] = lane_pair_max(vmax2(ma, mb));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (REL) {
      // sc is relative to mc already (exp2 domain): it was formed with the reference of the previous sub-step.  Lazy running
      // maximum as in k_sattn; the first half always moves it (mc starts at 0, whatever the scores are), in either direction.
      // A move corrects this half's scores (sc -= d) and rewrites the Q-side slot, so the S^T of the next half - issued in the
      // tail below - is formed against the new reference.
      if (__any(mt[0] > lazy || mt[1] > lazy) || hh == 0) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const float want = mc[x] + (hh == 0 ? mt[x] : fmaxf(mt[x], 0.f));
          const unsigned nb = pack2<F16>(-want, 0.f) & 0xffffu;               // -max, rounded to the element type
          const float m_new = F16 ? -(float)__builtin_bit_cast(f16x2_t, nb)[0] : -__builtin_bit_cast(float, nb << 16);
          const float d = m_new - mc[x];
          const float alpha = __builtin_amdgcn_exp2f(-d);
          mc[x] = m_new;
          if (lh == QSH) qf[x][QSS][0] = (qf[x][QSS][0] & 0xffff0000u) | nb;
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[x][r] -= d;
#pragma unroll
          for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[x][nt][r] *= alpha;
        }
      }
    } else {
      if (__any(mt[0] > m[0] + lazy_raw || mt[1] > m[1] + lazy_raw)) {  // lazy running maximum: see k_sattn
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const float m_new = fmaxf(m[x], mt[x]);
          const float alpha = __builtin_amdgcn_exp2f((m[x] - m_new) * scale_log2e);   // exp2(-inf) = 0 on the first half
          m[x] = m_new;
          mc[x] = m_new * scale_log2e;
#pragma unroll
          for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[x][nt][r] *= alpha;
        }
      }
    }
  };
  // ---- bracket 2: S^T of the next half | exp2 and the P fragments of this one (16 register pairs over 2 NS MFMAs)
  auto sub_tail = [&](auto next_c, float16_t (&sc)[2], float16_t (&sn)[2], const unsigned short* Kn) __attribute__((always_inline)) {
    constexpr bool NEXT = decltype(next_c)::value;
    {
      constexpr int NM = 2 * NS, PPC = (16 + NM - 1) / NM;       // MFMAs, pairs per MFMA
      uint4_t kf[NS];
      if constexpr (NEXT) {
#pragma unroll
        for (int s = 0; s < NS; ++s) kf[s] = *(const uint4_t*)(Kn + 16 * s);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int r = 0; r < 16; ++r) sn[x][r] = 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        if constexpr (NEXT) sn[i & 1] = mfma32<F16>(kf[i >> 1], qf[i & 1][i >> 1], sn[i & 1]);
#pragma unroll
        for (int p = i * PPC; p < (i + 1) * PPC && p < 16; ++p) {
          const int x = p >> 3, r0 = 2 * (p & 7);
          const float e0 = __builtin_amdgcn_exp2f(REL ? sc[x][r0] : fmaf(sc[x][r0], scale_log2e, -mc[x]));
          const float e1 = __builtin_amdgcn_exp2f(REL ? sc[x][r0 + 1] : fmaf(sc[x][r0 + 1], scale_log2e, -mc[x]));
          pf[x][r0 >> 3][(r0 & 7) >> 1] = pack2<F16>(e0, e1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // key tile t: K(t) in buffer kb, V^T(t) in vb; K(t + 1) at kb + 1, V^T(t - 1) at vb - 1 (mod 3).  The tile's ONE barrier
  // stands between the two parts of its second sub-step: before it the stores of K(t + 2) and V^T(t + 1) (fetched at the
  // top of the tile); the exp2 bracket closes the loop body, so it stays beside its MFMAs
  // (with the stores behind it the compiler sinks it past them, towards the only use of the P fragments in the next tile).
  int kb = 0, vb = 0;
  auto tile = [&](auto last_c, auto mask_c, int t) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    const int kb1 = kb == 2 ? 0 : kb + 1, kb2 = kb == 0 ? 2 : kb - 1;
    const int vb1 = vb == 2 ? 0 : vb + 1, vbp = vb == 0 ? 2 : vb - 1;
    if (!LAST) {
      if (t + 2 < ntiles) g_load_k(t + 2);
      g_load_v(t + 1);
    }
    // half 2t: P V of half 2t - 1 (tile t - 1, upper keys; before the first tile: zeros times tile 0), S^T of half 2t + 1
    sub_head(mask_c, 2 * t, sA, t ? Vbuf + vbp * VB + VHALF : Vbuf + vb * VB);
    sub_tail(std::true_type{}, sA, sB, Kbuf + kb * KB + (32 + lq) * KLD + 8 * lh);
    // half 2t + 1: P V of half 2t, S^T of half 2t + 2
    sub_head(mask_c, 2 * t + 1, sB, Vbuf + vb * VB);
    if (!LAST) {
      if (t + 2 < ntiles) s_store_k(kb2);
      s_store_v(vb1);
      __syncthreads();
    }
    sub_tail(std::integral_constant<bool, !LAST>{}, sB, sA, Kbuf + kb1 * KB + lq * KLD + 8 * lh);
    if (!LAST) {
      kb = kb1;
      vb = vb1;
    }
  };
  for (int t = 0; t + 1 < ntiles; ++t) tile(std::false_type{}, std::false_type{}, t);
  if (Lk & (KT - 1))
    tile(std::true_type{}, std::true_type{}, ntiles - 1);
  else
    tile(std::true_type{}, std::false_type{}, ntiles - 1);
  // P V of the last half
  {
    const unsigned short* Vp = Vbuf + vb * VB + VHALF;
#pragma unroll
    for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const uint4_t vf = v_frag(Vp, nt, s2);
#pragma unroll
        for (int x = 0; x < 2; ++x) oacc[x][nt] = mfma32<F16>(vf, pf[x][s2], oacc[x][nt]);
      }
  }

  // ---- 1 / sum (row DVP - 1 of O^T = tile NDV - 1, register 15 of the lh = 1 lanes), convert, store
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const long row = q0 + 32 * x + lq;
    const float l1 = oacc[x][NDV - 1][15];
    const float l0 = __shfl_xor(l1, 32);
    const float inv = 1.0f / (lh ? l1 : l0);
    unsigned short* orow = O + ((size_t)b * Lq + (row < Lq ? row : Lq - 1)) * C + (size_t)h * dh;
#pragma unroll
    for (int nt = 0; nt < NDV; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = nt * 32 + 8 * g + 4 * lh;
        if (dv < dh && row < Lq) {
          const uint2_t o2 = {pack2<F16>(oacc[x][nt][4 * g] * inv, oacc[x][nt][4 * g + 1] * inv),
                              pack2<F16>(oacc[x][nt][4 * g + 2] * inv, oacc[x][nt][4 * g + 3] * inv)};
          *(uint2_t*)(orow + dv) = o2;
        }
      }
  }
}

template <int DHP, bool VTI, bool VTR = false, bool REL = false>
int launch_cfg_h(const void* q, const void* k, const void* vt, void* o, int B, int H, int Lq, int Lk, int LkP, int dh,
                 float scale, int dtype, hipStream_t st, long ld, float lazy) {
  const dim3 grid((Lq + 255) / 256, H, B);
  const float sl2 = scale * 1.4426950408889634f;
  constexpr int NDV = (DHP + 31) / 32;
  const size_t smem = (size_t)3 * (KT * (DHP + 8) + (VTR ? KT * 96 : NDV * 32 * (KT + 4))) * sizeof(unsigned short);
  const unsigned short one = dtype == UCE_DTYPE_F16 ? 0x3C00 : 0x3F80;
  static PerDeviceOnce attr_once;
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_sattn_h<DHP, true, VTI, VTR, REL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_sattn_h<DHP, false, VTI, VTR, REL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL((k_sattn_h<DHP, true, VTI, VTR, REL>), grid, dim3(256), smem, st, (const unsigned short*)q, (const unsigned short*)k,
                       (const unsigned short*)vt, (unsigned short*)o, H, Lq, Lk, LkP, dh, sl2, ld, one, lazy);
  else
    hipLaunchKernelGGL((k_sattn_h<DHP, false, VTI, VTR, REL>), grid, dim3(256), smem, st, (const unsigned short*)q, (const unsigned short*)k,
                       (const unsigned short*)vt, (unsigned short*)o, H, Lq, Lk, LkP, dh, sl2, ld, one, lazy);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}


template <int DHP, bool VTI, bool VTR = false>
int launch_cfg_p(const void* q, const void* k, const void* vt, void* o, int B, int H, int Lq, int Lk, int LkP, int dh,
                 float scale, int dtype, hipStream_t st, long ld, float lazy) {
  const dim3 grid((Lq + 127) / 128, H, B);
  const float sl2 = scale * 1.4426950408889634f;
  constexpr int NDV = (DHP + 31) / 32;
  const size_t smem = (size_t)2 * (KT * (DHP + 8) + (VTR ? KT * 96 : NDV * 32 * (KT + 4))) * sizeof(unsigned short);
  const unsigned short one = dtype == UCE_DTYPE_F16 ? 0x3C00 : 0x3F80;
  static PerDeviceOnce attr_once;
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_sattn_p<DHP, true, VTI, VTR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_sattn_p<DHP, false, VTI, VTR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL((k_sattn_p<DHP, true, VTI, VTR>), grid, dim3(256), smem, st, (const unsigned short*)q, (const unsigned short*)k,
                       (const unsigned short*)vt, (unsigned short*)o, H, Lq, Lk, LkP, dh, sl2, ld, one, lazy);
  else
    hipLaunchKernelGGL((k_sattn_p<DHP, false, VTI, VTR>), grid, dim3(256), smem, st, (const unsigned short*)q, (const unsigned short*)k,
                       (const unsigned short*)vt, (unsigned short*)o, H, Lq, Lk, LkP, dh, sl2, ld, one, lazy);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int DHP, int QT, bool VTI, int NBUF = 2, bool VTR = false>
int launch_cfg(const void* q, const void* k, const void* vt, void* o, int B, int H, int Lq, int Lk, int LkP, int dh,
               float scale, int dtype, hipStream_t st, long ld, float lazy) {
  const dim3 grid((Lq + 128 * QT - 1) / (128 * QT), H, B);
  const float sl2 = scale * 1.4426950408889634f;
  constexpr int NDV = (DHP + 31) / 32;
  constexpr int DVP = NDV * 32, VRS = (DVP % 128 == 32 || DVP % 128 == 96) ? DVP : DVP + 32;
  const size_t smem = (size_t)NBUF * (KT * (DHP + 8) + (VTR ? KT * VRS : DVP * (KT + 4))) * sizeof(unsigned short);
  const unsigned short one = dtype == UCE_DTYPE_F16 ? 0x3C00 : 0x3F80;
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_sattn<DHP, true, QT, VTI, NBUF, VTR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_sattn<DHP, false, QT, VTI, NBUF, VTR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL((k_sattn<DHP, true, QT, VTI, NBUF, VTR>), grid, dim3(256), smem, st, (const unsigned short*)q, (const unsigned short*)k,
                       (const unsigned short*)vt, (unsigned short*)o, H, Lq, Lk, LkP, dh, sl2, ld, one, lazy);
  else
    hipLaunchKernelGGL((k_sattn<DHP, false, QT, VTI, NBUF, VTR>), grid, dim3(256), smem, st, (const unsigned short*)q, (const unsigned short*)k,
                       (const unsigned short*)vt, (unsigned short*)o, H, Lq, Lk, LkP, dh, sl2, ld, one, lazy);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

// k_sattn_h (dh <= 48): by rule where the 256-row workgroups fill the chip twice over and there are at least 8 key tiles.
// Measured at H = 8, dh = 40, packed q|k|v, us per launch, k_sattn QT = 2 with the V^T pre-pass | k_sattn_h inline V^T:
//   B = 32:  L = 4096  1349 | 1168     L = 1024  107 | 106
//   B = 128: L = 4096  5171 | 4898     L = 1024  416 | 362     L = 256  42.6 | 44.5 (stays with k_sattn)
static bool sattn_use_h(int qt_variant, int dh, int Lq, int Lk, int H, int B) {
  if (dh > 48) return false;
  if (qt_variant == 4) return true;
  // (one prompt per call - 256 workgroups at L = 4096 - takes it too: 97 us against 113 for k_sattn behind the V^T pre-pass,
  // profiles/r05/sattn_b1.jsonl; the 1024 / 256-token layers of that batch are launch-bound on every form)
  const long wg = (long)((Lq + 255) / 256) * H * B;
  return qt_variant == 0 && ((wg >= 1024 && Lk >= 512) || (wg >= 256 && Lk >= 2048));
}

// the kernel for one shape, with (VTI) or without the V^T pre-pass already run
template <bool VTI>
int launch_body(const void* q, const void* k, const void* vt, void* o, int B, int H, int Lq, int Lk, int LkP, int dh, float scale,
                int dtype, hipStream_t st, int qt_variant, long ld, float lazy, int vti = 0, bool exp2q = false) {
  // measured on MI355X at the generation batch (B = 32, H = 8; us per launch, k_sattn QT = 1 | QT = 2 | k_sattn_p):
  //   L = 4096, dh = 40:  1691 | 1576 | 1788   (the pipelined form drops from 3 to 2 waves per SIMD at dh = 40 and loses)
  //   L = 1024, dh = 80:   212 |  -   |  197   (two waves per SIMD either way: the pipeline wins)
  if (sattn_use_h(qt_variant, dh, Lq, Lk, H, B)) {
    // inline V: row-major in LDS + transposing reads (UCE_SATTN_VTI = 0 / 3), or transposed on the way in (= 1)
    if constexpr (VTI) {
      // q already carries scale * log2(e) (uce_sattn_packed_exp2_fwd): the exp2-domain form where the head has a padding dim to spare
      if (exp2q && dh == 40 && vti != 1) return launch_cfg_h<48, true, true, true>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy);
      if (vti != 1) return launch_cfg_h<48, true, true>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy);
    }
    return launch_cfg_h<48, VTI>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy);
  }
  if (qt_variant == 3 && dh <= 48) return launch_cfg_p<48, VTI>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy);
  // 64 < dh <= 80 (the 32 x 32 level, L = 1024): with the K / V tiles by buffer loads and V row-major in LDS the one-tile kernel needs
  // 164 VGPRs - THREE workgroups per CU (47 KB of LDS each) - and passes the pipelined one, which holds two tiles' scores and stays at
  // two: L = 1024 at B = 128 471 against 552 us same box.  The pipelined kernel keeps the V^T pre-pass case and UCE_SATTN_QT = 3.
  if ((qt_variant == 3 || (qt_variant == 0 && !(VTI && vti != 1))) && dh > 64 && dh <= 80) {
    if constexpr (VTI) {                        // inline V: row-major + transposing reads unless UCE_SATTN_VTI = 1 asks for the 2-byte stores
      if (vti != 1) return launch_cfg_p<80, true, true>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy);
    }
    return launch_cfg_p<80, VTI>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy);
  }
  if (dh <= 48) {
    // two query tiles per wave once there are enough 256-row workgroups to fill the chip twice over
    const long wg2 = (long)((Lq + 255) / 256) * H * B;
    if (qt_variant != 1 && (qt_variant == 2 || wg2 >= 1024))
      return launch_cfg<48, 2, VTI>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy);
    return launch_cfg<48, 1, VTI>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy);
  }
  // one query tile per wave; inline V row-major + transposing reads unless UCE_SATTN_VTI = 1 asks for the 2-byte stores
  constexpr bool R = VTI;
  const bool tr = R && vti != 1;
#define UCE_SA1(DHP_, NB_)                                                                                                          \
  return tr ? launch_cfg<DHP_, 1, VTI, NB_, R>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy)                       \
            : launch_cfg<DHP_, 1, VTI, NB_>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, ld, lazy);
  if (dh <= 64) { UCE_SA1(64, 2) }
  if (dh <= 80) { UCE_SA1(80, 2) }
  if (dh <= 96) { UCE_SA1(96, 2) }
  if (dh <= 128) { UCE_SA1(128, 2) }
  // dh = 160 (the 16 x 16 level): two workgroups per CU on one LDS image each where there are workgroups for it (UCE_SATTN_QT = 1
  // keeps the two-image form)
  if (qt_variant == 0 && (long)((Lq + 127) / 128) * H * B >= 512) { UCE_SA1(160, 1) }
  UCE_SA1(160, 2)
#undef UCE_SA1
}

}  // namespace

int sattn_dvp(int dh) {
  const int dhp = dh <= 48 ? 48 : dh <= 64 ? 64 : dh <= 80 ? 80 : dh <= 96 ? 96 : dh <= 128 ? 128 : 160;
  return (dhp + 31) / 32 * 32;
}

size_t sattn_vt_elems(int B, int H, int Lk, int dh) {
  return (size_t)B * H * sattn_dvp(dh) * ((Lk + KT - 1) / KT * KT);
}

// V^T by the pre-pass (k_vt: straight 16-byte tile copies in the key loop) or transposed on the way into LDS (no pre-pass, no
// scratch): `vti` = UCE_SATTN_VTI (0: by rule - inline up to 1024 keys, where the pre-pass and its launch are a visible share
// of a short kernel; 1: always inline, transposed by 2-byte stores; 2: always the pre-pass; 3: always inline - k_sattn_h keeps V
// row-major and reads it with ds_read_b64_tr_b16 under 0 and 3)
bool sattn_inline_vt(int Lk, int vti, bool use_h) { return vti == 1 || vti == 3 || (vti == 0 && (Lk <= 1024 || use_h)); }

// qt_variant (UCE_SATTN_QT, read at uce_create): 0 = measured best by shape, 1 = always k_sattn with one query tile per wave,
// 2 = two query tiles wherever dh <= 48, 3 = the pipelined kernel wherever it exists (dh <= 48, 64 < dh <= 80)
// ld: row stride (elements) of q, k and v; o rows are H * dh apart.
// the running maximum of the online softmax is raised only when a key tile's maximum exceeds it by more than 2^8 (P <= 256 keeps the
// relative precision of bf16 / f16; sums are f32): measured 4, 8, 16 -> 4 882 / 4 843 / 4 860 us at L = 4096, B = 128 (round 4)
constexpr float SATTN_LAZY = 8.f;
constexpr int SATTN_SHORT_KEYS = 128;

int launch_sattn(const void* q, const void* k, const void* v, void* vt, void* o, int B, int H, int Lq, int Lk, int dh,
                 float scale, int dtype, hipStream_t st, int qt_variant, long ld, int vti, float lazy, bool exp2q) {
  const int LkP = (Lk + KT - 1) / KT * KT;
  if (ld <= 0) ld = (long)H * dh;
  // at most 128 keys: every key resident, plain softmax - k_xattn's form (uce_xattn.hip); UCE_SATTN_QT != 0 keeps the streaming kernels
  if (qt_variant == 0 && Lk <= SATTN_SHORT_KEYS) return launch_xattn_short_self(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
  if (sattn_inline_vt(Lk, vti, sattn_use_h(qt_variant, dh, Lq, Lk, H, B))) return launch_body<true>(q, k, v, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, qt_variant, ld, lazy, vti, exp2q);
  const int DVP = sattn_dvp(dh);
  const int ones_row = dh < DVP ? DVP - 1 : -1;
  const unsigned short one = dtype == UCE_DTYPE_F16 ? 0x3C00 : 0x3F80;
  hipLaunchKernelGGL(k_vt, dim3(LkP / 64, (DVP + 63) / 64, B * H), dim3(256), 0, st, (const unsigned short*)v,
                     (unsigned short*)vt, H, Lk, dh, DVP, LkP, ones_row, one, ld);
  UCE_LAUNCH_CHECK();
  return launch_body<false>(q, k, vt, o, B, H, Lq, Lk, LkP, dh, scale, dtype, st, qt_variant, ld, lazy);
}

static int sattn_check(uce_handle_t h, const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh,
                       int dtype) {
  if (!h || !q || !k || !v || !o || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return UCE_EINVAL;
  if (dh <= 0 || dh > 160 || (dh & 7)) return UCE_EINVAL;
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  if (B > 65535 || H > 65535 || (long)B * H > 65535) return UCE_EINVAL;
  return UCE_OK;
}

extern "C" int uce_sattn_fwd(uce_handle_t h, const void* q, const void* k, const void* v, void* o, int B, int H, int Lq,
                             int Lk, int dh, float scale, int dtype, uce_stream_t stream) {
  if (const int rc = sattn_check(h, q, k, v, o, B, H, Lq, Lk, dh, dtype)) return rc;
  UCE_ENTER(h);
  if (!sattn_inline_vt(Lk, h->sw.sattn_vti, sattn_use_h(h->sw.sattn_qt, dh, Lq, Lk, H, B))) {
    const int rc = uce_ensure_Vt(h, sattn_vt_elems(B, H, Lk, dh));
    if (rc) return rc;
  }
  UceProfScope ps(h, "uce_sattn_fwd", (hipStream_t)stream);
  return launch_sattn(q, k, v, h->Vt, o, B, H, Lq, Lk, dh, scale, dtype, (hipStream_t)stream, h->sw.sattn_qt, (long)H * dh,
                      h->sw.sattn_vti, SATTN_LAZY);
}

extern "C" int uce_sattn_packed_fwd(uce_handle_t h, const void* qkv, void* o, int B, int H, int L, int dh, float scale, int dtype,
                                    uce_stream_t stream) {
  const unsigned short* p = (const unsigned short*)qkv;
  const long C = (long)H * dh;
  if (const int rc = sattn_check(h, p, p, p, o, B, H, L, L, dh, dtype)) return rc;
  UCE_ENTER(h);
  if (!sattn_inline_vt(L, h->sw.sattn_vti, sattn_use_h(h->sw.sattn_qt, dh, L, L, H, B))) {
    const int rc = uce_ensure_Vt(h, sattn_vt_elems(B, H, L, dh));
    if (rc) return rc;
  }
  UceProfScope ps(h, "uce_sattn_packed_fwd", (hipStream_t)stream);
  return launch_sattn(p, p + C, p + 2 * C, h->Vt, o, B, H, L, L, dh, scale, dtype, (hipStream_t)stream, h->sw.sattn_qt, 3 * C,
                      h->sw.sattn_vti, SATTN_LAZY);
}

// exp2-domain self-attention on a packed projection whose q columns already carry scale * log2(e) (uce_linear_colscale_fwd wrote them):
// softmax numerator = exp2(q' . k - max).  Where the head has a padding dim to spare and the two-tile streaming kernel runs (dh = 40
// at SD-1.4's 64 x 64 level) the scores leave the matrix pipe as the argument of exp2 (k_sattn_h<.., REL>); every other shape takes
// its usual kernel with a unit factor in place of scale * log2(e) - the same arithmetic, one multiplication by 1.0 per score.
extern "C" int uce_sattn_exp2_form(uce_handle_t h, int B, int H, int L, int dh) {
  if (!h || B <= 0 || H <= 0 || L <= 0) return 0;
  const int qt = h->sw.sattn_qt;                                       // (4 = the two-tile kernel at every shape: tests)
  return dh == 40 && (qt == 4 || (qt == 0 && L > SATTN_SHORT_KEYS)) && h->sw.sattn_vti != 1 && h->sw.sattn_vti != 2 &&
         sattn_use_h(qt, dh, L, L, H, B);
}

extern "C" int uce_sattn_packed_exp2_fwd(uce_handle_t h, const void* qkv, void* o, int B, int H, int L, int dh, int dtype,
                                         uce_stream_t stream) {
  const unsigned short* p = (const unsigned short*)qkv;
  const long C = (long)H * dh;
  if (const int rc = sattn_check(h, p, p, p, o, B, H, L, L, dh, dtype)) return rc;
  UCE_ENTER(h);
  if (!sattn_inline_vt(L, h->sw.sattn_vti, sattn_use_h(h->sw.sattn_qt, dh, L, L, H, B))) {
    const int rc = uce_ensure_Vt(h, sattn_vt_elems(B, H, L, dh));
    if (rc) return rc;
  }
  UceProfScope ps(h, "uce_sattn_packed_exp2_fwd", (hipStream_t)stream);
  // scale = 1 / log2(e): the kernels multiply it by log2(e) again - a unit factor (0.6931471805599453 * 1.4426950408889634 rounds to 1.0f)
  return launch_sattn(p, p + C, p + 2 * C, h->Vt, o, B, H, L, L, dh, 0.6931471805599453f, dtype, (hipStream_t)stream, h->sw.sattn_qt,
                      3 * C, h->sw.sattn_vti, SATTN_LAZY, true);
}
