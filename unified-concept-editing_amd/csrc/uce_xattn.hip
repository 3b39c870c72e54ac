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
#include <cstdlib>

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
                                               int dh, float scale_log2e, int iters, long ld) {
  // ld: row stride (elements) of Q, K and V - H * dh for separate tensors, 3 * H * dh for the packed q | k | v projection of a
  // SHORT self-attention (uce_sattn_*_fwd with at most 128 keys: all keys resident, plain softmax - this kernel); O rows: H * dh
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
    const unsigned short* qrow = Q + ((size_t)b * Lq + row) * ld + (size_t)h * dh;
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
    const unsigned short* kbase = K + (size_t)b * Lk * ld + (size_t)h * dh;
    const unsigned short* vbase = V + (size_t)b * Lk * ld + (size_t)h * dh;
    constexpr int KCH = DHP / 8;  // 16-byte chunks per key row
    for (int e = tid; e < LKP * KCH; e += 256) {
      const int key = e / KCH, dim = (e - key * KCH) * 8;
      uint4_t val = {0u, 0u, 0u, 0u};
      if (key < Lk && dim < dh) val = *(const uint4_t*)(kbase + (size_t)key * ld + dim);
      *(uint4_t*)(Ks + key * KLD + dim) = val;
    }
    constexpr int VCH = DVP / 8;
    for (int e = tid; e < LKP * VCH; e += 256) {
      const int key = e / VCH, dv = (e - key * VCH) * 8;
      uint4_t val = {0u, 0u, 0u, 0u};
      if (key < Lk && dv < dh) val = *(const uint4_t*)(vbase + (size_t)key * ld + dv);
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


// ---------------------------------------------------------------------------------------------------------
// Group kernel (the default for dh in {40, 80, 160}, i.e. SD-1.x, and dh = 64, i.e. SD-2.x / SDXL): one workgroup = one batch sample x
// one 320-channel column group (8 / 5 / 4 / 2 heads = 640 bytes of every [B, L, C] row = five whole 128-byte lines) x a run
// of consecutive query tiles.  k_xattn gives each (batch, head) its own workgroup, so the 80 / 160 / 320 bytes a
// head owns of a row are fetched - and the partial lines of O written - by different workgroups on different XCDs
// (2.6 L2 fills per line at dh = 40: 0.33 of the HBM roofline).  Here every byte of Q is loaded exactly once with
// 16-byte lane accesses over whole 640-byte row segments, staged in LDS, and every byte of O leaves the same way:
//   * K (row-major, 80 keys x 640 B) and V^T (320 dims x 96 keys) of the group live in LDS for the whole run;
//   * wave (head, 32-row slice) reads its Q fragments out of the staged tile, runs k_xattn's swapped
//     S^T = K Q^T -> in-register softmax -> O^T = V^T P^T, and writes its O block back INTO THE SAME LDS region
//     (the region a wave reads Q from is exactly the region it writes O to: no barrier in between);
//   * after one barrier the tile leaves with coalesced 16-byte stores and the same thread parks the next tile's Q
//     (prefetched into registers while the MFMAs ran) in the slot it just drained: two barriers per tile.
// ---------------------------------------------------------------------------------------------------------
constexpr int XG_C = 320;                  // channels per column group
constexpr int XG_ROW = XG_C * 2 + 16;      // LDS row stride (bytes): an odd multiple of 16 B -> conflict-free b128 fragment reads
constexpr int XG_KEYS = 80;                // key rows held in LDS (Lk <= 80; CLIP: 77)
constexpr int XG_VLD = 100;                // V^T row stride (elements): 96 key slots + pad, conflict-free b64 reads
constexpr int XG_KT = 3;                   // 32-key tiles

constexpr int XG_VROWS = XG_C + 2;         // + two rows of ones: the P V MFMA accumulates the softmax denominator itself
constexpr int XG_KROWS = XG_KEYS + 1;      // + one all-masked key row: what the key slots >= XG_KEYS of the last tile read

template <int DH, int NW>
constexpr size_t xg_smem() { return (size_t)XG_KROWS * XG_ROW + (size_t)XG_VROWS * XG_VLD * 2 + (size_t)(32 * (NW / (XG_C / DH))) * XG_ROW; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xg_rsrc(const void* base, long bytes) {
  const int n = bytes < 0 ? 0 : (bytes > 0x7fffffffL ? 0x7fffffff : (int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n, 0x00020000);
}

template <int DH, int NW, bool F16>
__global__ __launch_bounds__(NW * 64) void k_xattn_g(const unsigned short* __restrict__ Q,
                                                     const unsigned short* __restrict__ K,
                                                     const unsigned short* __restrict__ V,
                                                     unsigned short* __restrict__ O, int C, int Lq, int Lk,
                                                     float scale_log2e, int iters, int wpb, int B, int remap) {
  constexpr int HG = XG_C / DH;             // heads per group
  constexpr int RS = NW / HG;               // 32-row slices per tile
  constexpr int TR = 32 * RS;               // query rows per tile
  constexpr int NT = NW * 64;
  constexpr int NS = (DH + 15) / 16;        // contraction steps of S^T = K Q^T
  constexpr int NDV = (DH + 31) / 32;       // output dim tiles
  constexpr int KT = XG_KT;
  constexpr int CH = XG_C / 8;              // 16-byte chunks per row segment
  constexpr int NQ = (TR * CH + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;                                                    // [XG_KROWS][XG_ROW]
  unsigned short* Vt = (unsigned short*)(smem + XG_KROWS * XG_ROW);            // [XG_VROWS][XG_VLD]
  unsigned char* Xs = smem + XG_KROWS * XG_ROW + XG_VROWS * XG_VLD * 2;        // [TR][XG_ROW]  Q in, O out
  // head dims that leave padding rows in the last 32-dim tile (40, 80): rows dv = DH and DH + 4 of that tile read a
  // row of ones, so register 4 / 8 of the tile's accumulator IS sum_k P[k] for this lane's query (lh = 0 / 1)
  constexpr bool ONES = (DH % 32) != 0 && (DH % 32) <= 24;
  constexpr int ONE_REG = ((DH % 32) / 8) * 4;
  // dh = 40: the last 16-dim contraction step has 8 spare dims (the lh = 1 lanes).  Those lanes read the 16-byte PAD of
  // their LDS row instead of the neighbouring head: the Q pads hold (1, 0, ..., 0), the K pads (bias_key, 0, ..., 0)
  // with bias = 0 for real keys and a huge negative number for key slots >= Lk - the MFMA itself masks the padding
  // keys and nothing is selected per tile.
  constexpr bool PADTRICK = (DH % 16) == 8;
  constexpr unsigned ONE16 = F16 ? 0x3c00u : 0x3f80u;
  constexpr unsigned NEG16 = F16 ? 0xfbffu : 0xf149u;          // -65504 (f16) / -1e30 (bf16)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int G = C / XG_C;
  // workgroups that share (b, group) - and with it K / V - sit on one XCD when the batch allows it (speed only)
  int b, g, x;
  {
    const int L = blockIdx.x, per_b = wpb * G;
    if (remap) {
      const int xcd = L & 7, slot = L >> 3;
      b = xcd + 8 * (slot / per_b);
      const int rem = slot % per_b;
      g = rem / wpb;
      x = rem % wpb;
    } else {
      b = L / per_b;
      const int rem = L % per_b;
      g = rem / wpb;
      x = rem % wpb;
    }
  }
  const int hl = w % HG, rs = w / HG;
  const int hc = hl * DH;                   // first channel of this wave's head inside the group
  const int lq = lane & 31, lh = lane >> 5;
  const int tiles_q = (Lq + TR - 1) / TR;
  const int t0 = x * iters;
  const int t1 = (t0 + iters < tiles_q) ? t0 + iters : tiles_q;
  if (t0 >= t1) return;                     // workgroup-uniform

  const unsigned short* Qg = Q + (size_t)b * Lq * C + (size_t)g * XG_C;
  unsigned short* Og = O + (size_t)b * Lq * C + (size_t)g * XG_C;
  const long row_bytes = (long)C * 2;

  // tile-invariant per-thread slots of the cooperative copies: chunk e = tid + NT*p -> (row, 16-byte chunk)
  unsigned slot_lds[NQ], slot_glb[NQ];
#pragma unroll
  for (int p = 0; p < NQ; ++p) {
    const int e = tid + NT * p;
    const int row = e / CH, ch = e - row * CH;
    slot_lds[p] = (unsigned)(row * XG_ROW + ch * 16);
    slot_glb[p] = (e < TR * CH) ? (unsigned)(row * row_bytes + ch * 16) : 0x7fffffffu;    // beyond the tile: out of range
  }
  // a tile's rows through one buffer resource: rows >= Lq are out of range (loads give 0, stores are dropped)
  auto tile_bytes = [&](int t) -> long {
    const long rows = (long)Lq - (long)t * TR;
    return rows <= 0 ? 0 : ((rows < TR ? rows : TR) - 1) * row_bytes + XG_C * 2;
  };
  uint4_t qn[NQ];
  auto load_q_tile = [&](int t) {
    const __amdgpu_buffer_rsrc_t r = xg_rsrc(Qg + (size_t)t * TR * C, tile_bytes(t));
#pragma unroll
    for (int p = 0; p < NQ; ++p) qn[p] = __builtin_amdgcn_raw_buffer_load_b128(r, slot_glb[p], 0, 0);
  };
  load_q_tile(t0);

  // ---- K rows and V^T of the group, once per workgroup
  {
    const unsigned short* kb = K + (size_t)b * Lk * C + (size_t)g * XG_C;
    const unsigned short* vb = V + (size_t)b * Lk * C + (size_t)g * XG_C;
    // four independent 16-byte loads in flight per thread before the first LDS store (one memory round trip per
    // four chunks instead of one per chunk: the staging is pure latency)
    constexpr int UNR = 4;
    for (int e0 = tid; e0 < XG_KROWS * (CH + 1); e0 += NT * UNR) {
      uint4_t val[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int e = e0 + NT * u;
        const int key = e / (CH + 1), ch = e - key * (CH + 1);
        val[u] = (uint4_t){0u, 0u, 0u, 0u};
        if (e < XG_KROWS * (CH + 1)) {
          if (ch < CH) {
            if (key < Lk) val[u] = *(const uint4_t*)(kb + (size_t)key * C + ch * 8);
          } else if (PADTRICK && key >= Lk) {
            val[u][0] = NEG16;                            // the row's pad: (bias, 0, ..., 0)
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int e = e0 + NT * u;
        const int key = e / (CH + 1), ch = e - key * (CH + 1);
        if (e < XG_KROWS * (CH + 1)) *(uint4_t*)(Ks + key * XG_ROW + ch * 16) = val[u];
      }
    }
    for (int e = tid; e < 2 * XG_VLD; e += NT) Vt[XG_C * XG_VLD + e] = (unsigned short)ONE16;
    for (int e0 = tid; e0 < 96 * CH; e0 += NT * UNR) {   // consecutive lanes -> consecutive keys: 2-byte LDS writes side by side
      uint4_t val[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int e = e0 + NT * u;
        const int c8 = e / 96, key = e - c8 * 96;
        val[u] = (uint4_t){0u, 0u, 0u, 0u};
        if (e < 96 * CH && key < Lk) val[u] = *(const uint4_t*)(vb + (size_t)key * C + c8 * 8);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int e = e0 + NT * u;
        const int c8 = e / 96, key = e - c8 * 96;
        if (e < 96 * CH) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            Vt[(c8 * 8 + 2 * t) * XG_VLD + key] = (unsigned short)(val[u][t] & 0xffffu);
            Vt[(c8 * 8 + 2 * t + 1) * XG_VLD + key] = (unsigned short)(val[u][t] >> 16);
          }
        }
      }
    }
    for (int r = tid; r < TR; r += NT) *(uint4_t*)(Xs + r * XG_ROW + XG_C * 2) = (uint4_t){PADTRICK ? ONE16 : 0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int p = 0; p < NQ; ++p)
    if (tid + NT * p < TR * CH) *(uint4_t*)(Xs + slot_lds[p]) = qn[p];
  __syncthreads();

  // tile-invariant fragment addresses (bytes): step s of this lane's Q row / of key row kt*32 + lq
  const unsigned xrow = (unsigned)((rs * 32 + lq) * XG_ROW);
  unsigned frag_off[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int dim = 16 * s + 8 * lh;
    frag_off[s] = (PADTRICK && s == NS - 1 && lh == 1) ? (unsigned)(XG_C * 2) : (unsigned)((hc + dim) * 2);
  }
  unsigned krow[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kt * 32 + lq;
    krow[kt] = (unsigned)((key < XG_KEYS ? key : XG_KEYS) * XG_ROW);   // slots >= XG_KEYS: the all-masked row
  }

  for (int t = t0; t < t1; ++t) {
    // ---- this wave's Q fragments out of the staged tile (row rs*32 + lq, dims 16s + 8*lh .. +7 of head hl)
    uint4_t qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const uint4_t v = *(const uint4_t*)(Xs + xrow + frag_off[s]);
      qf[s] = (PADTRICK || 16 * s + 8 * lh < DH) ? v : (uint4_t){0u, 0u, 0u, 0u};
    }
    const bool has_next = t + 1 < t1;
    if (has_next) load_q_tile(t + 1);                  // in flight under the MFMAs and the softmax

    // ---- S^T = K Q^T
    float16_t sacc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const uint4_t v = *(const uint4_t*)(Ks + krow[kt] + frag_off[s]);
        const uint4_t kf = (PADTRICK || 16 * s + 8 * lh < DH) ? v : (uint4_t){0u, 0u, 0u, 0u};
        sacc[kt] = mfma32<F16>(kf, qf[s], sacc[kt]);
      }
    }
    // ---- softmax over the keys (f32, in registers; key slots >= Lk are masked by the bias dim or here)
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (!PADTRICK && (kt + 1) * 32 > Lk) {
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
    // keys 80..95 (registers 8..15 of the last tile) are padding for every lane when Lk <= 80 (CLIP: 77): no exp,
    // no P fragment, no P V step for them
    const bool short_keys = Lk <= 80;                  // workgroup-uniform
    float sum = 0.f;
    uint4_t pf[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (kt == KT - 1 && s == 1 && short_keys) {
          pf[kt][s] = (uint4_t){0u, 0u, 0u, 0u};
          continue;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][8 * s + 2 * u], scale_log2e, -mc));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][8 * s + 2 * u + 1], scale_log2e, -mc));
          if constexpr (!ONES) sum += p0 + p1;
          pf[kt][s][u] = pack2<F16>(p0, p1);
        }
      }
    float inv = 0.f;
    if constexpr (!ONES) {
      sum += __shfl_xor(sum, 32);
      inv = 1.0f / sum;
    }

    // ---- O^T = V^T P^T, written back over this wave's own Q block of the staged tile
    unsigned char* orow = Xs + xrow + hc * 2;
#pragma unroll
    for (int it = 0; it < NDV; ++it) {
      // with the ones rows the last dim tile carries the denominator: it goes first
      const int nt = ONES ? (it == 0 ? NDV - 1 : it - 1) : it;
      float16_t oacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
      const int dvr = nt * 32 + lq;
      int vr = hc + (dvr < DH ? dvr : DH - 1);
      if (ONES && nt == NDV - 1) vr = (dvr == DH) ? XG_C : (dvr == DH + 4 ? XG_C + 1 : vr);
      const unsigned short* vrow = Vt + vr * XG_VLD + 4 * lh;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (kt == KT - 1 && s == 1 && short_keys) continue;
          const uint2_t lo = *(const uint2_t*)(vrow + kt * 32 + 16 * s);
          const uint2_t hi = *(const uint2_t*)(vrow + kt * 32 + 16 * s + 8);
          const uint4_t vf = {lo[0], lo[1], hi[0], hi[1]};
          oacc = mfma32<F16>(vf, pf[kt][s], oacc);
        }
      if (ONES && it == 0) inv = __builtin_amdgcn_rcpf(oacc[ONE_REG]);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int dv = nt * 32 + 8 * g4 + 4 * lh;
        if (dv < DH) {
          const uint2_t o2 = {pack2<F16>(oacc[4 * g4] * inv, oacc[4 * g4 + 1] * inv),
                              pack2<F16>(oacc[4 * g4 + 2] * inv, oacc[4 * g4 + 3] * inv)};
          *(uint2_t*)(orow + dv * 2) = o2;
        }
      }
    }
    __syncthreads();                                   // the O tile is complete
    // ---- the tile leaves in whole row segments; the same thread refills the slot with the next tile's Q
    {
      const __amdgpu_buffer_rsrc_t r = xg_rsrc(Og + (size_t)t * TR * C, tile_bytes(t));
#pragma unroll
      for (int p = 0; p < NQ; ++p) {
        if (tid + NT * p < TR * CH) {
          const uint4_t o4 = *(const uint4_t*)(Xs + slot_lds[p]);
          __builtin_amdgcn_raw_buffer_store_b128(o4, r, slot_glb[p], 0, 0);
          if (has_next) *(uint4_t*)(Xs + slot_lds[p]) = qn[p];
        }
      }
    }
    __syncthreads();                                   // the next Q tile is staged
  }
}

template <int DH, int NW>
int launch_group(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, float scale,
                 int dtype, hipStream_t st) {
  constexpr int TR = 32 * (NW / (XG_C / DH));
  const int C = H * DH, G = C / XG_C;
  const int tiles_q = (Lq + TR - 1) / TR;
  // one workgroup per CU (the K / V^T / tile footprint fills the LDS): about 256 workgroups, each walking `iters` tiles
  int wpb = (256 + G * B - 1) / (G * B);
  wpb = wpb < 1 ? 1 : (wpb > tiles_q ? tiles_q : wpb);
  const int iters = (tiles_q + wpb - 1) / wpb;
  wpb = (tiles_q + iters - 1) / iters;
  const long nwg = (long)wpb * G * B;
  if (nwg > 0x7fffffffL) return UCE_EINVAL;
  const int remap = (B % 8 == 0) ? 1 : 0;
  const size_t smem = xg_smem<DH, NW>();
  const float sl2 = scale * 1.4426950408889634f;
  static PerDeviceOnce attr_once;
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_xattn_g<DH, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_xattn_g<DH, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.commit(tok);
  }
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL((k_xattn_g<DH, NW, true>), dim3((unsigned)nwg), dim3(NW * 64), smem, st, (const unsigned short*)q,
                       (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)o, C, Lq, Lk, sl2, iters, wpb, B, remap);
  else
    hipLaunchKernelGGL((k_xattn_g<DH, NW, false>), dim3((unsigned)nwg), dim3(NW * 64), smem, st, (const unsigned short*)q,
                       (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)o, C, Lq, Lk, sl2, iters, wpb, B, remap);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int DHP, int KT>
int launch_cfg(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh,
               float scale, int dtype, hipStream_t st, long ld) {
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
                       iters, ld);
  else
    hipLaunchKernelGGL((k_xattn<DHP, KT, false>), grid, dim3(256), 0, st, (const unsigned short*)q,
                       (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)o, H, Lq, Lk, dh, sl2,
                       iters, ld);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int KT>
int launch_dh(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh,
              float scale, int dtype, hipStream_t st, long ld) {
  if (dh <= 48) return launch_cfg<48, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
  if (dh <= 64) return launch_cfg<64, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
  if (dh <= 80) return launch_cfg<80, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
  if (dh <= 96) return launch_cfg<96, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
  if (dh <= 128) return launch_cfg<128, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
  return launch_cfg<160, KT>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
}

}  // namespace

int launch_xattn(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh,
                 float scale, int dtype, hipStream_t st, int variant) {
  // SD-1.x shapes: the 640-byte column-group kernel (whole lines of Q / O per workgroup); `variant` = the handle's
  // UCE_XATTN_VARIANT (read at uce_create): 0 keeps the per-(batch, head) kernel, 2 = the group kernel at every size,
  // 3 = its 8-wave dh = 40 form (tests force them through a handle of their own)
  if (variant && Lk <= XG_KEYS && (H * dh) % XG_C == 0 && scale > 0.f) {
    // worth it once a workgroup walks several tiles behind one K / V^T staging (the generation batch); the B = 2
    // launches of row-by-row generation stay on k_xattn (launch-bound either way)
    const long rows_groups = (long)B * Lq * ((H * dh) / XG_C);
    const bool big = variant == 2 || rows_groups >= 256L * 4 * 64;
    if (big && dh == 40) {
      if (variant == 3) return launch_group<40, 8>(q, k, v, o, B, H, Lq, Lk, scale, dtype, st);
      return launch_group<40, 16>(q, k, v, o, B, H, Lq, Lk, scale, dtype, st);
    }
    if (big && dh == 64) return launch_group<64, 10>(q, k, v, o, B, H, Lq, Lk, scale, dtype, st);   // SD-2.x / SDXL: 5 heads per group
    if (big && dh == 80) return launch_group<80, 8>(q, k, v, o, B, H, Lq, Lk, scale, dtype, st);
    if (big && dh == 160) return launch_group<160, 4>(q, k, v, o, B, H, Lq, Lk, scale, dtype, st);
  }
  const long ld = (long)H * dh;
  if (Lk <= 96) return launch_dh<3>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
  return launch_dh<4>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
}

// A SHORT self-attention (at most 128 keys: SD-1.4's 8 x 8 level, L = 64) is k_xattn's case - every key resident in LDS, a plain
// softmax, no running maximum - not the streaming kernel's (uce_sattn.hip: one 64-key tile, half of its 128-row workgroup idle:
// 45.7 us at B = 128 against 32.5 for torch's SDPA; through here the driver line of round 6 says).  ld: row stride of q, k, v.
int launch_xattn_short_self(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh, float scale,
                            int dtype, hipStream_t st, long ld) {
  if (Lk <= 64) return launch_dh<2>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
  if (Lk <= 96) return launch_dh<3>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
  return launch_dh<4>(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, st, ld);
}

extern "C" int uce_xattn_fwd(uce_handle_t h, const void* q, const void* k, const void* v, void* o, int B,
                             int H, int Lq, int Lk, int dh, float scale, int dtype, uce_stream_t stream) {
  if (!h || !q || !k || !v || !o || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || Lk > 128) return UCE_EINVAL;
  UCE_ENTER(h);
  if (dh <= 0 || dh > 160 || (dh & 7)) return UCE_EINVAL;
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  if (B > 65535 || H > 65535) return UCE_EINVAL;
  UceProfScope ps(h, "uce_xattn_fwd", (hipStream_t)stream);
  return launch_xattn(q, k, v, o, B, H, Lq, Lk, dh, scale, dtype, (hipStream_t)stream, h->sw.xattn_variant);
}
