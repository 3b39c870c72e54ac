// Apply kernels: W_new = W_old + W_old Delta for EVERY cross-attention K/V projection at once
// (reference: `mat1 @ torch.inverse(mat2)` per module, uce_sd_erase.py:82; the host keeps all
// modules' weights in one [rows, d] slab so one launch covers the whole U-Net).
//
// k_apply         : dense form, NT GEMM [rows,d] x DeltaT[d,d]^T on v_mfma_f32_32x32x2_f32 (exact
//                   f32 products), residual folded into the accumulator init.  MFMA-bound.
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

__global__ __launch_bounds__(256, 2) void k_apply(const float* __restrict__ W_old,
                                                  const float* __restrict__ DeltaT,
                                                  float* __restrict__ W_new, long rows, int d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float (*As)[BM][TLD] = (float (*)[BM][TLD])smem_raw;                              // [2][128][36]
  float (*Bs)[BN][TLD] = (float (*)[BN][TLD])(smem_raw + 2 * BM * TLD * sizeof(float));

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int ncol = (d + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const long r0 = (long)(lid / ncol) * BM;
  const int j0 = (lid % ncol) * BN;

  // staging coordinates: 4 float4 per thread per operand per K step
  const int srow = tid >> 3;        // 0..31 (+32p)
  const int sc4 = (tid & 7) * 4;    // 0..28
  const float* aptr[4];
  const float* bptr[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    long gr = r0 + srow + 32 * p;
    if (gr > rows - 1) gr = rows - 1;
    int gj = j0 + srow + 32 * p;
    if (gj > d - 1) gj = d - 1;
    aptr[p] = W_old + gr * d + sc4;
    bptr[p] = DeltaT + (long)gj * d + sc4;
  }

  // accumulators start at the residual W_old tile (D layout of 32x32 MFMA)
  float16_t acc[2][2];
  const int ccol = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long gr = r0 + wm + mt * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        const int gc = j0 + wn + nt * 32 + ccol;
        acc[mt][nt][r] = (gr < rows && gc < d) ? W_old[gr * d + gc] : 0.f;
      }

  float4_t ra[4], rb[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    ra[p] = *(const float4_t*)(aptr[p]);
    rb[p] = *(const float4_t*)(bptr[p]);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    *(float4_t*)&As[0][srow + 32 * p][sc4] = ra[p];
    *(float4_t*)&Bs[0][srow + 32 * p][sc4] = rb[p];
  }
  __syncthreads();

  const int nk = d / BK;
  const int fr = lane & 31, fk = 4 * (lane >> 5);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        ra[p] = *(const float4_t*)(aptr[p] + (kt + 1) * BK);
        rb[p] = *(const float4_t*)(bptr[p] + (kt + 1) * BK);
      }
    }
#pragma unroll
    for (int u = 0; u < BK / 8; ++u) {
      // k permutation inside each 8-k group: MFMA t takes k = 8u + 4*(lane>>5) + t from BOTH operands
      float4_t fa[2], fb[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt] = *(const float4_t*)&As[cur][wm + mt * 32 + fr][u * 8 + fk];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) fb[nt] = *(const float4_t*)&Bs[cur][wn + nt * 32 + fr][u * 8 + fk];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][t], fb[nt][t], acc[mt][nt], 0, 0, 0);
    }
    if (kt + 1 < nk) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        *(float4_t*)&As[cur ^ 1][srow + 32 * p][sc4] = ra[p];
        *(float4_t*)&Bs[cur ^ 1][srow + 32 * p][sc4] = rb[p];
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long gr = r0 + wm + mt * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        const int gc = j0 + wn + nt * 32 + ccol;
        if (gr < rows && gc < d) W_new[gr * d + gc] = acc[mt][nt][r];
      }
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

int launch_apply(const float* W_old, const float* DeltaT, float* W_new, long rows, int d, hipStream_t st) {
  static const int one_per_cu = getenv("UCE_APPLY_1WG") ? atoi(getenv("UCE_APPLY_1WG")) : 0;
  const size_t smem = one_per_cu ? (size_t)100 * 1024 : (size_t)2 * (BM + BN) * TLD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_apply, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const long row_tiles = (rows + BM - 1) / BM;
  const int col_tiles = (d + BN - 1) / BN;
  const long nwg = row_tiles * col_tiles;
  if (nwg > 0x7fffffffL) return UCE_EINVAL;
  hipLaunchKernelGGL(k_apply, dim3((unsigned)nwg), dim3(256), smem, st, W_old, DeltaT, W_new, rows, d);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

int launch_delta_from_factors(const float* Dm, const float* R, int N_edit, int d, float* DeltaT,
                              hipStream_t st) {
  const int nb = d / 64;
  hipLaunchKernelGGL(k_delta_factors, dim3(nb * nb), dim3(256), 0, st, Dm, R, N_edit, d, DeltaT);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

extern "C" int uce_cast_bf16(uce_handle_t h, const float* src, void* dst_bf16, long n, uce_stream_t stream) {
  if (!h || !src || !dst_bf16 || n < 0) return UCE_EINVAL;
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

// Streaming variant for the embedding widths that matter (768 / 1024 / 2048): NO weight tile in
// LDS.  A 16-row tile per workgroup, 4 waves:
//   phase 1  T = W_tile Dm^T.  Wave w owns K quarter w; per pipeline unit it loads 2 16-k groups
//            of its W rows (A fragments, straight from HBM) and of 4 x 16 Dm rows (B fragments,
//            L2) and issues 32 MFMAs (16x16x4 f32) into 4 accumulators (one per 16-concept tile).
//            Partial T's meet in LDS (the only LDS use: ~20 KB, so occupancy is VGPR-bound).
//   phase 2  out = W_tile + T R.  Wave w owns column groups w, w+4, ...; lane j of a group owns 4
//            consecutive columns, so the residual (re-read of the W tile, an L2 hit), the R
//            fragments and the output are all 16 B per lane.  k-step-major MFMA order: one LDS
//            read of T feeds every group.  Results stay in registers and are stored at the very
//            end, so no load ever queues behind a store (vmcnt retires in order on gfx950).
// Every global load is unconditional (clamped address, zero applied afterwards) and issued two
// units ahead of its MFMAs, pinned with sched_barrier: hipcc then emits counted vmcnt waits.
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_apply_lowrank(
    const float* __restrict__ W_old, const float* __restrict__ Dm, const float* __restrict__ R,
    float* __restrict__ W_new, long rows, int Ne, int NEP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int d = D;
  const int tld = NEP + 2;
  float* Ts = (float*)smem_raw;                 // [16][tld]
  float* Tp = Ts + LR_BM * tld;                 // [4][16][tld] per-wave partials of phase 1

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const long r0 = (long)blockIdx.x * LR_BM;

  // ---------------- phase 1
  constexpr int KQ = D / 4;          // floats per K quarter
  constexpr int NG = KQ / 16;        // 16-k groups per quarter (12 / 16 / 32)
  constexpr int GRP = 2;             // groups per pipeline unit
  constexpr int NU = NG / GRP;       // units per batch of 4 concept tiles
  const int kbeg = w * KQ;
  long arow = r0 + li;
  arow = arow < rows ? arow : rows - 1;
  const float* aptr = W_old + arow * d + kbeg + 4 * lk;
  const int nbatch = (NEP + 63) >> 6;
  const int nunit = nbatch * NU;

  struct Frag1 { float4_t a[GRP]; float4_t b[4][GRP]; };
  auto load1 = [&](int u, Frag1& f) {
    const int bt = u / NU, uu = u - bt * NU;
#pragma unroll
    for (int g = 0; g < GRP; ++g) f.a[g] = *(const float4_t*)(aptr + (uu * GRP + g) * 16);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int e_row = (bt * 4 + ct) * 16 + li;
      const int e_cl = e_row < Ne ? e_row : Ne - 1;
      const float* drow = Dm + (size_t)e_cl * d + kbeg + 4 * lk + uu * GRP * 16;
#pragma unroll
      for (int g = 0; g < GRP; ++g) f.b[ct][g] = *(const float4_t*)(drow + g * 16);
    }
    __builtin_amdgcn_sched_barrier(0);   // the whole batch is issued HERE, two units ahead of its use
  };
  float4_t acc1[4];
  auto comp1 = [&](int u, const Frag1& f) {
    const int bt = u / NU, uu = u - bt * NU;
    if (uu == 0) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc1[ct] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    const int nct_b = min(4, (NEP >> 4) - bt * 4);   // concept tiles that exist in this batch (uniform)
    // k permutation: MFMA q of group g uses k = 16g + 4*(lane>>4) + q on both operands
    if (nct_b == 4) {
#pragma unroll
      for (int g = 0; g < GRP; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const float bv = ((bt * 4 + ct) * 16 + li < Ne) ? f.b[ct][g][q] : 0.f;
            acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[g][q], bv, acc1[ct], 0, 0, 0);
          }
    } else {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        if (ct < nct_b) {
#pragma unroll
          for (int g = 0; g < GRP; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float bv = ((bt * 4 + ct) * 16 + li < Ne) ? f.b[ct][g][q] : 0.f;
              acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[g][q], bv, acc1[ct], 0, 0, 0);
            }
        }
    }
    if (uu == NU - 1) {
      // D layout: col = lane & 15 (concept within the tile), row = 4*(lane>>4) + r
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        if (ct < nct_b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Tp[(w * 16 + 4 * lk + r) * tld + (bt * 4 + ct) * 16 + li] = acc1[ct][r];
        }
    }
  };
  {
    Frag1 fA, fB, fC;
    load1(0, fA);
    load1(nunit > 1 ? 1 : 0, fB);
    int u = 0;
    for (; u + 2 < nunit; u += 3) {
      load1(u + 2, fC);
      comp1(u, fA);
      load1(u + 3 < nunit ? u + 3 : u + 2, fA);
      comp1(u + 1, fB);
      load1(u + 4 < nunit ? u + 4 : u + 2, fB);
      comp1(u + 2, fC);
    }
    if (u < nunit) comp1(u, fA);
    if (u + 1 < nunit) comp1(u + 1, fB);
  }
  __syncthreads();
  for (int e = tid; e < LR_BM * NEP; e += 256) {
    const int r = e / NEP, c = e - r * NEP;
    Ts[r * tld + c] = (Tp[(0 * 16 + r) * tld + c] + Tp[(1 * 16 + r) * tld + c]) +
                      (Tp[(2 * 16 + r) * tld + c] + Tp[(3 * 16 + r) * tld + c]);
  }
  __syncthreads();

  // ---------------- phase 2
  constexpr int MG = D / 256;                  // column groups per wave (3 / 4 / 8)
  constexpr int GP = MG > 4 ? 4 : MG;          // groups per pass (accumulators live in registers)
  const int nks = (Ne + 3) >> 2;               // k-steps that carry concepts
  const int nu2 = (nks + 1) >> 1;              // pipeline units of 2 k-steps
  struct Frag2 { float4_t b[2][GP]; };
#pragma unroll 1
  for (int pass = 0; pass < MG / GP; ++pass) {
    const int gbase = w + 4 * pass * GP;       // groups gbase, gbase + 4, ...
    auto load2 = [&](int u, Frag2& f) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int e_row = 8 * u + 4 * t + lk;
        const int e_cl = e_row < Ne ? e_row : Ne - 1;
        const float* rrow = R + (size_t)e_cl * d + 4 * li;
#pragma unroll
        for (int g = 0; g < GP; ++g) f.b[t][g] = *(const float4_t*)(rrow + (gbase + 4 * g) * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    float4_t acc[GP][4];                       // acc[g][q][r]: row 4*lk + r, column (gbase+4g)*64 + 4*li + q
    Frag2 fA, fB, fC;
    load2(0, fA);
    load2(nu2 > 1 ? 1 : 0, fB);
    {
      // residual: the W tile again (L2), already in the accumulator layout
      float4_t wv[GP][4];
#pragma unroll
      for (int g = 0; g < GP; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          long gr = r0 + 4 * lk + r;
          gr = gr < rows ? gr : rows - 1;
          wv[g][r] = *(const float4_t*)(W_old + gr * d + (gbase + 4 * g) * 64 + 4 * li);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < GP; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[g][q][r] = wv[g][r][q];
    }
    auto comp2 = [&](int u, const Frag2& f) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int e_row = 8 * u + 4 * t + lk;
        const float a = (e_row < Ne) ? Ts[li * tld + e_row] : 0.f;   // zero A also kills clamped R rows
#pragma unroll
        for (int g = 0; g < GP; ++g)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[g][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, f.b[t][g][q], acc[g][q], 0, 0, 0);
      }
    };
    int u = 0;
    for (; u + 2 < nu2; u += 3) {
      load2(u + 2, fC);
      comp2(u, fA);
      load2(u + 3 < nu2 ? u + 3 : u + 2, fA);
      comp2(u + 1, fB);
      load2(u + 4 < nu2 ? u + 4 : u + 2, fB);
      comp2(u + 2, fC);
    }
    if (u < nu2) comp2(u, fA);
    if (u + 1 < nu2) comp2(u + 1, fB);
#pragma unroll
    for (int g = 0; g < GP; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long gr = r0 + 4 * lk + r;
        if (gr < rows) {
          const float4_t o = {acc[g][0][r], acc[g][1][r], acc[g][2][r], acc[g][3][r]};
          *(float4_t*)(W_new + gr * d + (gbase + 4 * g) * 64 + 4 * li) = o;
        }
      }
  }
}

// =============================================================================================
// Super-tile low-rank apply (49 <= N_edit <= 256): one workgroup = MT 16-row tiles (MT*16 rows),
// 4 waves, one per SIMD.  Both GEMMs re-use every fetched operand MT times, which is what the
// 16-row kernels above cannot do (their Dm / R fragments feed a single MFMA and the L1/TA path,
// not the MFMA pipe, sets their speed):
//   phase 1  T = W_st Dm^T: wave w owns concept tile (16 columns of T) 4*batch + w.  The W
//            k-chunk (MT*16 rows x 64 floats, full 256 B row segments) is staged once in LDS
//            (double buffered, register prefetch one chunk ahead) and shared by the 4 waves; each
//            wave's Dm fragments come straight from L2 and feed MT MFMAs each.
//   phase 2  out = W_st + T R: wave w owns 64-column groups w, w+4, ...; for a group it keeps
//            MT x 4 accumulators (initialised with the residual, an L2 re-read of W) and walks the
//            concept k-steps: one 16 B R fragment per lane feeds 4*MT MFMAs; T comes from LDS.
// MFMA-bound by construction (f32 16x16x4): MT*16 rows cost 32 cycles * MT*(NEP*D/64 + nks*D/64)/...
// =============================================================================================

constexpr int ST_KC = 64;    // floats per W k-chunk
constexpr int ST_LD = 72;    // LDS row stride of the chunk (floats): conflict-free b128 fragment reads

// 8 waves: wave = (concept tile / column-group class c4 = w & 3, M half = w >> 2).  The two waves
// that share a SIMD split the MT row tiles between them, so one wave's LDS reads, address math and
// waits overlap the other's MFMAs (the matrix pipe is per SIMD and shared by its waves).
template <int D, int MT, int NMT>
struct StBody {
  // phase 1 for NMT row tiles starting at mbase; returns nothing, writes T into Ts
  static __device__ __forceinline__ void run(const float* __restrict__ W_old, const float* __restrict__ Dm,
                                             const float* __restrict__ R, float* __restrict__ W_new,
                                             long rows, int Ne, int NEP, float* Wc, float* Ts, int mbase, int mode) {
    constexpr int d = D;
    constexpr int SR = MT * 16;
    const int tld = NEP + 2;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c4 = w & 3;
    const int li = lane & 15, lk = lane >> 4;
    const long R0 = (long)blockIdx.x * SR;

    constexpr int NC = D / ST_KC;                       // k-chunks (12 / 16 / 32), even
    constexpr int F4 = SR * (ST_KC / 4);                // float4 per chunk
    constexpr int NLD = (F4 + 511) / 512;               // per thread
    auto load_chunk = [&](int kc, float4_t (&v)[NLD]) {
#pragma unroll
      for (int p = 0; p < NLD; ++p) {
        const int e = tid + 512 * p;
        const int r = (e >> 4) < SR ? (e >> 4) : SR - 1, cc = (e & 15) << 2;
        long gr = R0 + r;
        gr = gr < rows ? gr : rows - 1;
        v[p] = *(const float4_t*)(W_old + gr * d + kc * ST_KC + cc);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto park_chunk = [&](int buf, const float4_t (&v)[NLD]) {
#pragma unroll
      for (int p = 0; p < NLD; ++p) {
        const int e = tid + 512 * p;
        if (e < F4) *(float4_t*)&Wc[(buf * SR + (e >> 4)) * ST_LD + ((e & 15) << 2)] = v[p];
      }
    };
    const int nbatch = NEP >> 6;
#pragma unroll 1
    for (int bt = 0; bt < nbatch; ++bt) {
      const int e_row = (bt * 4 + c4) * 16 + li;        // this lane's concept (B operand column)
      const float bmask = e_row < Ne ? 1.f : 0.f;
      const float* dptr = Dm + (size_t)(e_row < Ne ? e_row : Ne - 1) * d + 4 * lk;
      auto load_dm = [&](int kc, float4_t (&b)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = *(const float4_t*)(dptr + kc * ST_KC + g * 16);
        __builtin_amdgcn_sched_barrier(0);
      };
      float4_t acc[NMT];
#pragma unroll
      for (int m = 0; m < NMT; ++m) acc[m] = (float4_t){0.f, 0.f, 0.f, 0.f};
      auto compute = [&](int buf, const float4_t (&b)[4]) {
        // k permutation: MFMA q of 16-k group g uses k = 16g + 4*(lane>>4) + q on both operands
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4_t a[NMT];
#pragma unroll
          for (int m = 0; m < NMT; ++m)
            a[m] = *(const float4_t*)&Wc[(buf * SR + (mbase + m) * 16 + li) * ST_LD + g * 16 + 4 * lk];
          const float4_t bb = b[g] * bmask;
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < NMT; ++m)
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][q], bb[q], acc[m], 0, 0, 0);
        }
      };
      // W chunk c: loaded (HBM) during iteration c-2 into register set c&1, parked in LDS buffer
      // c&1 during iteration c-1, consumed in iteration c.
      float4_t wA[NLD], wB[NLD], dmA[4], dmB[4];
      __syncthreads();                                  // previous batch is done with Wc
      load_chunk(0, wA);
      load_chunk(1, wB);
      load_dm(0, dmA);
      load_dm(1, dmB);
      park_chunk(0, wA);
      load_chunk(2 < NC ? 2 : 0, wA);
      __syncthreads();
#pragma unroll 1
      for (int kc = 0; kc < ((mode & 1) ? 2 : NC); kc += 2) {
        park_chunk(1, wB);                              // chunk kc + 1
        load_chunk(kc + 3 < NC ? kc + 3 : kc, wB);
        compute(0, dmA);                                // chunk kc
        load_dm(kc + 2 < NC ? kc + 2 : kc, dmA);
        __syncthreads();
        if (kc + 2 < NC) park_chunk(0, wA);             // chunk kc + 2
        load_chunk(kc + 4 < NC ? kc + 4 : kc, wA);
        compute(1, dmB);                                // chunk kc + 1
        load_dm(kc + 3 < NC ? kc + 3 : kc, dmB);
        __syncthreads();
      }
      // D layout: col = lane & 15 (concept), row = 4*(lane>>4) + r
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Ts[((mbase + m) * 16 + 4 * lk + r) * tld + (bt * 4 + c4) * 16 + li] = acc[m][r];
    }

    // ---------------- phase 2 (its first loads are issued before the barrier that publishes T)
    constexpr int MG = D / 256;                         // column groups per class (3 / 4 / 8)
    constexpr int RD = 4;                               // R fragments in flight
    const int nks = (Ne + 3) >> 2;                      // k-steps that carry concepts
    int rl_g = 0, rl_t = 0;                             // (group, k-step) of the next R fragment to fetch
    auto r_next = [&]() -> float4_t {
      const int e = 4 * rl_t + lk;
      const float4_t v = *(const float4_t*)(R + (size_t)(e < Ne ? e : Ne - 1) * d + (c4 + 4 * rl_g) * 64 + 4 * li);
      if (++rl_t == nks) { rl_t = 0; rl_g = rl_g + 1 < MG ? rl_g + 1 : rl_g; }
      return v;
    };
    auto res_load = [&](int gi, float4_t (&x)[NMT][4]) {
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          long gr = R0 + (mbase + m) * 16 + 4 * lk + r;
          gr = gr < rows ? gr : rows - 1;
          x[m][r] = *(const float4_t*)(W_old + gr * d + (c4 + 4 * gi) * 64 + 4 * li);
        }
      __builtin_amdgcn_sched_barrier(0);
    };
    float4_t ring[RD];
#pragma unroll
    for (int i = 0; i < RD; ++i) ring[i] = r_next();
    float4_t res[NMT][4];
    res_load(0, res);
    __syncthreads();
#pragma unroll
    for (int gi = 0; gi < MG; ++gi) {
      float4_t acc[NMT][4];                             // acc[m][q][r]: row (mbase+m)*16 + 4*lk + r, column 4*li + q
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[m][q][r] = res[m][r][q];
      if (gi + 1 < MG) res_load(gi + 1, res);           // next group's residual, ahead of this group's stores
#pragma unroll 1
      for (int t = 0; t < ((mode & 2) ? 1 : nks); ++t) {
        const float4_t b = ring[0];
#pragma unroll
        for (int i = 0; i + 1 < RD; ++i) ring[i] = ring[i + 1];
        ring[RD - 1] = r_next();
        const int e = 4 * t + lk;
        float a[NMT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) a[m] = (e < Ne) ? Ts[((mbase + m) * 16 + li) * tld + e] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int m = 0; m < NMT; ++m)
            acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[q], acc[m][q], 0, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long gr = R0 + (mbase + m) * 16 + 4 * lk + r;
          if (gr < rows) {
            const float4_t o = {acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]};
            // streaming store: keep W_old (re-read as the residual) rather than W_new in L2 / MALL
            __builtin_nontemporal_store(o, (float4_t*)(W_new + gr * d + (c4 + 4 * gi) * 64 + 4 * li));
          }
        }
    }
  }
};

template <int D, int MT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_apply_lowrank_st(
    const float* __restrict__ W_old, const float* __restrict__ Dm, const float* __restrict__ R,
    float* __restrict__ W_new, long rows, int Ne, int NEP, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Wc = (float*)smem_raw;                       // [2][MT*16][ST_LD]
  float* Ts = Wc + 2 * MT * 16 * ST_LD;               // [MT*16][NEP + 2]
  constexpr int M0 = (MT + 1) / 2;
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) < 256)
    StBody<D, MT, M0>::run(W_old, Dm, R, W_new, rows, Ne, NEP, Wc, Ts, 0, mode);
  else
    StBody<D, MT, MT - M0>::run(W_old, Dm, R, W_new, rows, Ne, NEP, Wc, Ts, M0, mode);
}

template <int D, int MT>
int launch_st(const float* W_old, const float* Dm, const float* R, float* W_new, long rows, int N_edit,
              int NEP64, hipStream_t st) {
  const size_t smem = ((size_t)2 * MT * 16 * ST_LD + (size_t)MT * 16 * (NEP64 + 2)) * sizeof(float);
  if (smem > 160 * 1024) return UCE_ENOMEM;   // caller falls back to the 16-row kernels
  static bool attr_set = false;
  if (!attr_set) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_apply_lowrank_st<D, MT>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const long nwg = (rows + MT * 16 - 1) / (MT * 16);
  hipLaunchKernelGGL((k_apply_lowrank_st<D, MT>), dim3((unsigned)nwg), dim3(512), smem, st, W_old, Dm, R, W_new,
                     rows, N_edit, NEP64, getenv("UCE_LR_MODE") ? atoi(getenv("UCE_LR_MODE")) : 0);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

// rows / 16 tiles over 256 CUs with MT tiles per workgroup: time ~ ceil(workgroups / 256) * MT
int pick_mt(long rows) {
  const long t16 = (rows + 15) / 16;
  int best = 8;
  long best_cost = -1;
  for (int mt = 8; mt >= 5; --mt) {
    const long wgs = (t16 + mt - 1) / mt;
    const long cost = ((wgs + 255) / 256) * mt;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mt; }
  }
  return best;
}

template <int D>
int launch_st_d(const float* W_old, const float* Dm, const float* R, float* W_new, long rows, int N_edit,
                int NEP64, hipStream_t st) {
  if (D >= 2048) return launch_st<D, 5>(W_old, Dm, R, W_new, rows, N_edit, NEP64, st);   // register budget
  switch (pick_mt(rows)) {
    case 5: return launch_st<D, 5>(W_old, Dm, R, W_new, rows, N_edit, NEP64, st);
    case 6: return launch_st<D, 6>(W_old, Dm, R, W_new, rows, N_edit, NEP64, st);
    case 7: return launch_st<D, 7>(W_old, Dm, R, W_new, rows, N_edit, NEP64, st);
    default: return launch_st<D, 8>(W_old, Dm, R, W_new, rows, N_edit, NEP64, st);
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
  int NEP = (N_edit + 15) / 16 * 16;
  if (NEP == 0) NEP = 16;
  const size_t smem = ((size_t)LR_BM * (d + 8) + (size_t)5 * LR_BM * (NEP + 2)) * sizeof(float);
  if (smem > 160 * 1024) return UCE_EINVAL;
  static bool attr_set = false;
  if (!attr_set) {
    const int cap = 160 * 1024;
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_apply_lowrank<768>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_apply_lowrank<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_apply_lowrank<2048>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_apply_lowrank_generic, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    attr_set = true;
  }
  const long nwg = (rows + LR_BM - 1) / LR_BM;
  if (nwg > 0x7fffffffL) return UCE_EINVAL;
  static const int variant = getenv("UCE_LOWRANK_VARIANT") ? atoi(getenv("UCE_LOWRANK_VARIANT")) : 0;
  if (variant != 1 && N_edit > 48 && rows >= 16 * 5 * 64 && (d == 768 || d == 1024 || d == 2048)) {
    const int NEP64 = (N_edit + 63) / 64 * 64;
    int rc = UCE_ENOMEM;
    if (d == 768) rc = launch_st_d<768>(W_old, Dm, R, W_new, rows, N_edit, NEP64, st);
    else if (d == 1024) rc = launch_st_d<1024>(W_old, Dm, R, W_new, rows, N_edit, NEP64, st);
    else rc = launch_st_d<2048>(W_old, Dm, R, W_new, rows, N_edit, NEP64, st);
    if (rc != UCE_ENOMEM) return rc;
  }
  const dim3 grid((unsigned)nwg), block(256);
  const size_t smem_s = (size_t)5 * LR_BM * (NEP + 2) * sizeof(float);   // streaming kernels: T only
  if (N_edit > 0 && d == 768)
    hipLaunchKernelGGL(k_apply_lowrank<768>, grid, block, smem_s, st, W_old, Dm, R, W_new, rows, N_edit, NEP);
  else if (N_edit > 0 && d == 1024)
    hipLaunchKernelGGL(k_apply_lowrank<1024>, grid, block, smem_s, st, W_old, Dm, R, W_new, rows, N_edit, NEP);
  else if (N_edit > 0 && d == 2048)
    hipLaunchKernelGGL(k_apply_lowrank<2048>, grid, block, smem_s, st, W_old, Dm, R, W_new, rows, N_edit, NEP);
  else
    hipLaunchKernelGGL(k_apply_lowrank_generic, grid, block, smem, st, W_old, Dm, R, W_new, rows, d, N_edit, NEP);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

extern "C" int uce_debias_targets(uce_handle_t h, const float* C_edit, const float* C_debias,
                                  const double* Dsum, int N_edit, int N_debias, int d, float* G,
                                  uce_stream_t stream) {
  if (!h || !C_edit || !G || N_edit <= 0 || N_debias < 0 || d <= 0 || d % 64) return UCE_EINVAL;
  if (N_debias > 0 && (!C_debias || !Dsum)) return UCE_EINVAL;
  const long n = (long)N_edit * (d / 4);
  hipLaunchKernelGGL(k_debias_targets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     C_edit, C_debias, Dsum, N_edit, N_debias, d, G);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
