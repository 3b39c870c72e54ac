// Apply kernels: W_new = W_old + W_old Delta for EVERY cross-attention K/V projection at once
// (reference: `mat1 @ torch.inverse(mat2)` per module, uce_sd_erase.py:82; the host keeps all
// modules' weights in one [rows, d] slab so one launch covers the whole U-Net).
//
// (the dense form W_old (I + Delta) lives in uce_apply_h2.hip; the exact-f32 MFMA kernel it replaced is retired)
// k_delta_factors : DeltaT = R^T Dm  (small TN GEMM, f32 MFMA) for the dual path with large N_edit.
// k_cast_bf16     : f32 -> bf16 RNE cast of the edited slab into the U-Net's parameters.
#include "uce_common.h"
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TLD = 36;  // LDS row stride (floats): 144 B, conflict-free ds_read_b128 for 32-row fragments

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  // contiguous chunk of the logical grid per XCD (block b runs on XCD b % 8), bijective for any nwg
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = b & 7, local = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// DeltaT[j][k] = sum_e R[e][j] * Dm[e][k]   (Delta = Dm^T R), 64x64 tile per workgroup, f32 MFMA 16x16x4
__global__ __launch_bounds__(256) void k_delta_factors(const float* __restrict__ Dm,
                                                       const float* __restrict__ R, int N_edit, int d,
                                                       float* __restrict__ DeltaT) {
  __shared__ __attribute__((aligned(16))) float Rs[32][64];
  __shared__ __attribute__((aligned(16))) float Ds[32][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
  const int nb = d / 64;
  const int tj = blockIdx.x / nb, tk = blockIdx.x % nb;
  float4_t acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (float4_t){0.f, 0.f, 0.f, 0.f};
  const int lrow = tid >> 4, lc4 = (tid & 15) * 4;
  for (int e0 = 0; e0 < N_edit; e0 += 32) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int e = e0 + p * 16 + lrow;
      float4_t x = {0.f, 0.f, 0.f, 0.f}, y = x;
      if (e < N_edit) {
        x = *(const float4_t*)(R + (size_t)e * d + tj * 64 + lc4);
        y = *(const float4_t*)(Dm + (size_t)e * d + tk * 64 + lc4);
      }
      *(float4_t*)&Rs[p * 16 + lrow][lc4] = x;
      *(float4_t*)&Ds[p * 16 + lrow][lc4] = y;
    }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      const int kk = kb * 4 + (lane >> 4);
      const float a0 = Rs[kk][wr + (lane & 15)], a1 = Rs[kk][wr + 16 + (lane & 15)];
      const float b0 = Ds[kk][wc + (lane & 15)], b1 = Ds[kk][wc + 16 + (lane & 15)];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // D layout of 16x16 f32 MFMA: col = lane & 15, row = 4*(lane>>4) + r
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = tj * 64 + wr + m * 16 + 4 * (lane >> 4) + r;
        const int col = tk * 64 + wc + n * 16 + (lane & 15);
        DeltaT[(size_t)row * d + col] = acc[m][n][r];
      }
}

__global__ void k_cast_bf16(const float* __restrict__ src, unsigned short* __restrict__ dst, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4_t v = *(const float4_t*)(src + i);
    unsigned short o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      unsigned int u = __float_as_uint(v[t]);
      if ((u & 0x7fffffffu) > 0x7f800000u) o[t] = (unsigned short)((u >> 16) | 0x40);  // quiet NaN
      else o[t] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);           // RNE
    }
    *(uint2*)(dst + i) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16),
                                    (unsigned)o[2] | ((unsigned)o[3] << 16));
  } else {
    for (; i < n; ++i) {
      unsigned int u = __float_as_uint(src[i]);
      dst[i] = ((u & 0x7fffffffu) > 0x7f800000u) ? (unsigned short)((u >> 16) | 0x40)
                                                 : (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
  }
}

}  // namespace

int launch_delta_from_factors(const float* Dm, const float* R, int N_edit, int d, float* DeltaT,
                              hipStream_t st) {
  const int nb = d / 64;
  hipLaunchKernelGGL(k_delta_factors, dim3(nb * nb), dim3(256), 0, st, Dm, R, N_edit, d, DeltaT);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

extern "C" int uce_cast_bf16(uce_handle_t h, const float* src, void* dst_bf16, long n, uce_stream_t stream) {
  if (!h || !src || !dst_bf16 || n < 0) return UCE_EINVAL;
  UCE_ENTER(h);
  if (n == 0) return UCE_OK;
  const long n4 = (n + 3) / 4;
  hipLaunchKernelGGL(k_cast_bf16, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (unsigned short*)dst_bf16, n);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

// =============================================================================================
// Low-rank apply (dual path, N_edit <= 256):  W_new = W_old + (W_old Dm^T) R  in ONE pass over
// the weights: 16-row tiles, the W tile lives in LDS between the two GEMMs so HBM sees each
// weight exactly once in and once out (algorithmic bytes = 8 * rows * d).
//   phase 1  T[16, NEP]  = Ws[16, d] * Dm^T      v_mfma_f32_16x16x4_f32, K split over the 4 waves
//   phase 2  out[16, d]  = Ws + T * R            16x16x4, column groups of 64 split over the waves;
//            lane j owns columns 4j..4j+3 of a group so R loads / W stores are 16 B per lane.
// =============================================================================================
namespace {

constexpr int LR_BM = 16;

__global__ __launch_bounds__(256, 2) void k_apply_lowrank_generic(const float* __restrict__ W_old,
                                                          const float* __restrict__ Dm,
                                                          const float* __restrict__ R,
                                                          float* __restrict__ W_new, long rows, int d,
                                                          int Ne, int NEP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wld = d + 8;      // stride = 8 mod 16 floats: conflict-free b128 fragment reads
  const int tld = NEP + 2;    // 2 mod 32 (x odd): conflict-free ds_read_b32 of the phase-2 A fragments
  float* Ws = (float*)smem_raw;                 // [16][wld]
  float* Ts = Ws + LR_BM * wld;                 // [16][tld]
  float* Tp = Ts + LR_BM * tld;                 // [4][16][tld] per-wave partials of phase 1

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long r0 = (long)blockIdx.x * LR_BM;
  const int li = lane & 15, lk = lane >> 4;

  // ---- phase 0: W tile -> LDS (coalesced 16 B loads)
  const int f4_per_row = d >> 2;
  for (int e = tid; e < LR_BM * f4_per_row; e += 256) {
    const int r = e / f4_per_row, c4 = (e - r * f4_per_row) << 2;
    long gr = r0 + r;
    if (gr > rows - 1) gr = rows - 1;
    *(float4_t*)&Ws[r * wld + c4] = *(const float4_t*)(W_old + gr * d + c4);
  }
  __syncthreads();

  // ---- phase 1: this wave's K quarter for every 16-column tile of T.
  // Work units u = (column tile ct, chunk of 4 16-k groups); the Dm fragments of unit u+1 are
  // fetched (L2 latency ~1-2k cycles) while the MFMAs of unit u run.
  const int kq = d >> 2;               // floats per K quarter (multiple of 16)
  const int kbeg = w * kq;
  const int ngrp1 = kq >> 4;           // 16-k groups in the quarter
  const int nchunk = (ngrp1 + 3) >> 2;
  const int nct = NEP >> 4;
  const int nunit = nct * nchunk;
  const float* wrow = Ws + li * wld + kbeg + 4 * lk;

  auto load_unit = [&](int u, float4_t (&b)[4]) {
    const int ct = u / nchunk, g0 = (u - ct * nchunk) << 2;
    const int e_row = ct * 16 + li;
    const float* drow = Dm + (size_t)(e_row < Ne ? e_row : 0) * d + kbeg + 4 * lk;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      b[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
      if (e_row < Ne && g0 + t < ngrp1) b[t] = *(const float4_t*)(drow + (g0 + t) * 16);
    }
  };
  float4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  auto compute_unit = [&](int u, const float4_t (&b)[4]) {
    const int ct = u / nchunk, ch = u - ct * nchunk, g0 = ch << 2;
    if (ch == 0) { acc0 = (float4_t){0.f, 0.f, 0.f, 0.f}; acc1 = acc0; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (g0 + t < ngrp1) {   // wave-uniform
        // k permutation: MFMA q of 16-k group g uses k = 16g + 4*(lane>>4) + q on both operands
        const float4_t a = *(const float4_t*)(wrow + (g0 + t) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (t & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[t][q], acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[t][q], acc0, 0, 0, 0);
        }
      }
    }
    if (ch == nchunk - 1) {
      const float4_t sum = acc0 + acc1;
      // D layout: col = lane & 15 (column of T), row = 4*(lane>>4) + r
#pragma unroll
      for (int r = 0; r < 4; ++r) Tp[(w * 16 + 4 * lk + r) * tld + ct * 16 + li] = sum[r];
    }
  };
  {
    float4_t bA[4], bB[4];
    load_unit(0, bA);
    int u = 0;
    for (; u + 1 < nunit; u += 2) {
      load_unit(u + 1, bB);
      compute_unit(u, bA);
      if (u + 2 < nunit) load_unit(u + 2, bA);
      compute_unit(u + 1, bB);
    }
    if (u < nunit) compute_unit(u, bA);
  }
  __syncthreads();
  for (int e = tid; e < LR_BM * NEP; e += 256) {
    const int r = e / NEP, c = e - r * NEP;
    Ts[r * tld + c] = (Tp[(0 * 16 + r) * tld + c] + Tp[(1 * 16 + r) * tld + c]) +
                      (Tp[(2 * 16 + r) * tld + c] + Tp[(3 * 16 + r) * tld + c]);
  }
  __syncthreads();

  // ---- phase 2: out = Ws + Ts * R.  Units = (64-column group g, chunk of 16 e-rows x 4 lanes-k);
  // the R fragments of the next unit are in flight during the MFMAs of the current one.
  const int ngrp = d >> 6;
  const int ne4 = (Ne + 3) & ~3;                 // contraction length actually needed
  const int nech = (ne4 + 31) >> 5;              // chunks of 32 e-rows (8 MFMA k-steps)
  const int my_groups = (ngrp - w + 3) >> 2;     // groups w, w+4, ...
  const int nunit2 = my_groups * nech;
  auto load_unit2 = [&](int u, float4_t (&b)[8]) {
    const int gi = u / nech, e0 = (u - gi * nech) << 5;
    const int c0 = (w + 4 * gi) * 64 + 4 * li;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e_row = e0 + 4 * t + lk;
      b[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
      if (e_row < Ne) b[t] = *(const float4_t*)(R + (size_t)e_row * d + c0);
    }
  };
  float4_t acc[4];                               // acc[q][r]: row 4*lk + r, column c0 + q
  auto compute_unit2 = [&](int u, const float4_t (&b)[8]) {
    const int gi = u / nech, ch = u - gi * nech, e0 = ch << 5;
    const int c0 = (w + 4 * gi) * 64 + 4 * li;
    if (ch == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4_t wv = *(const float4_t*)&Ws[(4 * lk + r) * wld + c0];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][r] = wv[q];
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (e0 + 4 * t < ne4) {   // wave-uniform
        const float a = Ts[li * tld + e0 + 4 * t + lk];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t][q], acc[q], 0, 0, 0);
      }
    }
    if (ch == nech - 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long gr = r0 + 4 * lk + r;
        if (gr < rows) {
          const float4_t o = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
          *(float4_t*)(W_new + gr * d + c0) = o;
        }
      }
    }
  };
  {
    float4_t bA[8], bB[8];
    if (nunit2 > 0) load_unit2(0, bA);
    int u = 0;
    for (; u + 1 < nunit2; u += 2) {
      load_unit2(u + 1, bB);
      compute_unit2(u, bA);
      if (u + 2 < nunit2) load_unit2(u + 2, bA);
      compute_unit2(u + 1, bB);
    }
    if (u < nunit2) compute_unit2(u, bA);
  }
}

// G = C_edit + Dsum @ C_debias  (f64 accumulate, one thread per 4 output floats)
__global__ void k_debias_targets(const float* __restrict__ Ce, const float* __restrict__ Cd,
                                 const double* __restrict__ Dsum, int Ne, int Nd, int d,
                                 float* __restrict__ G) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int f4 = d >> 2;
  if (i >= (long)Ne * f4) return;
  const int e = (int)(i / f4), c4 = (int)(i - (long)e * f4) << 2;
  const float4_t c = *(const float4_t*)(Ce + (size_t)e * d + c4);
  double g[4] = {(double)c[0], (double)c[1], (double)c[2], (double)c[3]};
  for (int t = 0; t < Nd; ++t) {
    const double sc = Dsum[(size_t)e * Nd + t];
    const float4_t v = *(const float4_t*)(Cd + (size_t)t * d + c4);
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] += sc * (double)v[q];
  }
  *(float4_t*)(G + (size_t)e * d + c4) = (float4_t){(float)g[0], (float)g[1], (float)g[2], (float)g[3]};
}

}  // namespace

bool apply_lowrank_fits(int d, int N_edit) {
  const int NEP = N_edit <= 0 ? 16 : (N_edit + 15) / 16 * 16;
  return N_edit <= 256 && ((size_t)LR_BM * (d + 8) + (size_t)5 * LR_BM * (NEP + 2)) * sizeof(float) <= 160 * 1024;
}

int launch_apply_lowrank(const float* W_old, const float* Dm, const float* R, float* W_new, long rows,
                         int d, int N_edit, hipStream_t st) {
  // single-launch fused form for widths / slab sizes the two-kernel path (uce_lowrank2.hip) does not cover
  int NEP = (N_edit + 15) / 16 * 16;
  if (NEP == 0) NEP = 16;
  const size_t smem = ((size_t)LR_BM * (d + 8) + (size_t)5 * LR_BM * (NEP + 2)) * sizeof(float);
  if (smem > 160 * 1024) return UCE_EINVAL;
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_apply_lowrank_generic, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
    attr_once.commit(tok);
  }
  const long nwg = (rows + LR_BM - 1) / LR_BM;
  if (nwg > 0x7fffffffL) return UCE_EINVAL;
  hipLaunchKernelGGL(k_apply_lowrank_generic, dim3((unsigned)nwg), dim3(256), smem, st, W_old, Dm, R, W_new, rows, d,
                     N_edit, NEP);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

extern "C" int uce_debias_targets(uce_handle_t h, const float* C_edit, const float* C_debias,
                                  const double* Dsum, int N_edit, int N_debias, int d, float* G,
                                  uce_stream_t stream) {
  if (!h || !C_edit || !G || N_edit <= 0 || N_debias < 0 || d <= 0 || d % 64) return UCE_EINVAL;
  UCE_ENTER(h);
  if (N_debias > 0 && (!C_debias || !Dsum)) return UCE_EINVAL;
  const long n = (long)N_edit * (d / 4);
  hipLaunchKernelGGL(k_debias_targets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     C_edit, C_debias, Dsum, N_edit, N_debias, d, G);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
