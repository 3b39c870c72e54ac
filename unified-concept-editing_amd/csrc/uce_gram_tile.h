// Pieces of the primal Gram shared by uce_gram.hip (k_gram_primal) and uce_solve.hip (the rider workgroups of the
// persistent Cholesky launch that compute Bt = C_e^T S_e (G - C_e) while the factorisation of A runs).
#pragma once
#include "uce_common.h"

namespace {

constexpr int KC = 32;        // K-chunk staged in LDS per iteration

// lower-triangular tile enumeration: t -> (ti, tj) with ti >= tj
__device__ __forceinline__ void tri_decode(int t, int& ti, int& tj) {
  int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((a + 1) * (a + 2) / 2 <= t) ++a;
  while (a * (a + 1) / 2 > t) --a;
  ti = a;
  tj = t - a * (a + 1) / 2;
}

// store one wave's 32x32 quadrant held in 2x2 f64 MFMA accumulators
__device__ __forceinline__ void store_quadrant(double* out, int ld, int row0, int col0,
                                               const double4_t (&acc)[2][2], int lane, bool mirror,
                                               double diag_val, const float* inv_s, float lamb,
                                               int n_valid, bool add_diag) {
  const int c = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + m * 16 + rq + 4 * r;
        const int col = col0 + n * 16 + c;
        double v = acc[m][n][r];
        if (add_diag && row == col) {
          if (inv_s) {
            // a non-positive scale makes K indefinite: poison the pivot so potrf reports it
            const float sv = (row < n_valid) ? inv_s[row] : 1.f;
            v += (row < n_valid) ? ((sv > 0.f) ? (double)lamb / (double)sv : __builtin_nan("")) : 1.0;
          }
          else v += diag_val;
        }
        out[(size_t)row * ld + col] = v;
        if (mirror) out[(size_t)col * ld + row] = v;
      }
}

// One 64 x 64 tile of the primal Gram by one 512-thread workgroup (k_gram_primal, and the rider workgroups of the
// persistent Cholesky launch that compute Bt in its shadow: uce_solve.hip).
// 8 waves: two quads of 4 waves, each quad a full 64 x 64 tile over ALTERNATE 32-concept chunks (its own LDS staging);
// quad 1's accumulators are added to quad 0's through LDS at the end (fixed order: bit-repeatable).  With one wave per
// SIMD (round 2: 4 waves) the f64 MFMA pipe sat idle through every staging write, barrier and fragment read of its only
// wave: 0.33 of the f64 peak; two waves per SIMD cover each other's stalls.
// LDS: stage_raw 2 * 2 * KC * 64 floats (32 KB, 16-byte aligned), Ss [2][KC] floats.  Every thread returns; a caller that
// runs several tiles puts a barrier between them (quad 0 reads the reduction tile that aliases the staging).

__device__ __forceinline__ void gram_primal_tile(const GramPrimalArgs& a, bool isA, int ti, int tj, int split,
                                                 unsigned char* stage_raw, float (*Ss)[KC]) {
  float (*Xs)[KC][64] = (float (*)[KC][64])stage_raw;                         // [2][KC][64]
  float (*Ys)[KC][64] = (float (*)[KC][64])(stage_raw + 2 * KC * 64 * sizeof(float));
  double (*Red)[64] = (double (*)[64])stage_raw;   // [64][64]: quad 1's tile on its way to quad 0 (the staging is dead by then)
  const float* __restrict__ C = a.C;
  const float* __restrict__ G = a.G;
  const int d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = w >> 2, wq = w & 3, ht = tid & 255;
  const int wr = wq >> 1, wc = wq & 1;
  const int Ktot = isA ? a.N : a.N_edit;
  const int k_begin = split * a.kchunk;
  const int k_end = min(Ktot, k_begin + a.kchunk);

  double4_t acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int lrow = ht >> 4;           // 0..15
  const int lc4 = (ht & 15) * 4;      // 0..60
  // register prefetch of this quad's next 32-concept chunk: its global loads are in flight while the current one is multiplied
  float4_t px[KC / 16], py[KC / 16];
  float ps = 0.f;
  auto g_load = [&](int k0) {
#pragma unroll
    for (int p = 0; p < KC / 16; ++p) {
      const int n = k0 + p * 16 + lrow;
      px[p] = (float4_t){0.f, 0.f, 0.f, 0.f};
      py[p] = px[p];
      if (n < k_end) {
        px[p] = *(const float4_t*)(C + (size_t)n * d + ti * 64 + lc4);
        const float4_t cy = *(const float4_t*)(C + (size_t)n * d + tj * 64 + lc4);
        // stage X = C[:, ti tile], Y = C[:, tj tile] (A) or (G - C)[:, tj tile] (Bt)
        if (isA) py[p] = cy;
        else py[p] = *(const float4_t*)(G + (size_t)n * d + tj * 64 + lc4) - cy;
      }
    }
    if (ht < KC) ps = (k0 + ht < k_end) ? a.s[k0 + ht] : 0.f;
  };
  // quad `half` owns chunks half, half + 2, ...; the loop count is the same for both quads (barriers are workgroup-wide):
  // a quad whose chunk lies beyond k_end stages zeros
  const int kq = k_begin + half * KC;
  if (k_begin < k_end) g_load(kq);
  for (int k0 = k_begin; k0 < k_end; k0 += 2 * KC) {
#pragma unroll
    for (int p = 0; p < KC / 16; ++p) {
      *(float4_t*)&Xs[half][p * 16 + lrow][lc4] = px[p];
      *(float4_t*)&Ys[half][p * 16 + lrow][lc4] = py[p];
    }
    if (ht < KC) Ss[half][ht] = ps;
    __syncthreads();
    if (k0 + 2 * KC < k_end) g_load(k0 + 2 * KC + half * KC);
#pragma unroll
    for (int kb = 0; kb < KC / 4; ++kb) {
      const int kk = kb * 4 + (lane >> 4);
      const double sc = (double)Ss[half][kk];
      const double a0 = (double)Xs[half][kk][wr * 32 + (lane & 15)] * sc;
      const double a1 = (double)Xs[half][kk][wr * 32 + 16 + (lane & 15)] * sc;
      const double b0 = (double)Ys[half][kk][wc * 32 + (lane & 15)];
      const double b1 = (double)Ys[half][kk][wc * 32 + 16 + (lane & 15)];
      acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
      acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
      acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
      acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
  // quad 1 -> LDS -> quad 0 (D layout of the f64 MFMA: row = (lane>>4) + 4r, col = lane & 15)
  {
    const int c = lane & 15, rq = lane >> 4;
    if (half == 1) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) Red[wr * 32 + m * 16 + rq + 4 * r][wc * 32 + n * 16 + c] = acc[m][n][r];
    }
    __syncthreads();
    if (half == 1) return;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] += Red[wr * 32 + m * 16 + rq + 4 * r][wc * 32 + n * 16 + c];
  }
  double* out = (isA ? a.outA : a.outBt) + (size_t)split * a.slab_stride;
  store_quadrant(out, d, ti * 64 + wr * 32, tj * 64 + wc * 32, acc, lane, isA && ti != tj,
                 (double)a.lamb, nullptr, a.lamb, 0, isA && split == 0);
}

}  // namespace
