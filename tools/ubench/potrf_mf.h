// R&D probe (not part of the product; measured against potrf64_pk by tools/ubench/potrf.hip -DMF): the 64 x 64 f64
// Cholesky-inverse block as a blocked factorisation whose 16 x 16 base case runs inside one wave.  Correct for every
// npiv (|X A X^T - I| ~ 1e-15), 46 900 cycles against 27 700 - see the note at the end of uce_potrf64.h.
// tools/ubench/potrf_mf_emu.py is the lane-accurate numpy model the kernel was written against.
#pragma once
#include "uce_potrf64.h"

namespace {

// ---------------------------------------------------------------------------------------------
// The same block (A = L L^T, hands out L^-1) as a BLOCKED factorisation whose 16 x 16 base case runs inside ONE wave:
// no barrier, no LDS round trip per pivot.
//
//  * Tiles of 16 x 16 live in the D layout of v_mfma_f64_16x16x4 (lane (g = lane >> 4, c = lane & 15), register r holds
//    element (g + 4 r, c)): four f64 registers per lane.  With X, Y in that layout,  prod(acc, X, Y): acc += X^T Y  is
//    four MFMAs (register r of X as the A operand is X^T[i = c][k = g + 4 r], register r of Y as the B operand is
//    Y[k = g + 4 r][j = c]) - every product of the blocked algorithm is written in that form, so no tile is ever
//    transposed or moved between lanes.
//  * Base case (mf_base), LDL^T elimination of [A | I] on the symmetric tile A, one pivot per step: the pivot row k sits
//    in the 16 lanes g = k & 3 of register k >> 2, INDEXED BY COLUMN in lane & 15 - which is exactly an MFMA operand
//    vector in contraction slot k & 3 (the other three slots zero).  By symmetry it is also column k, so
//        A  -= (u / p) u^T      = mfma(aop, bopA, A),   aop = -u / p on i > k, bopA = u on j > k
//        X  -= (u / p) X[k,:]   = mfma(aop, bopX, X),   bopX = row k of X
//        X^T -= X[k,:]^T (u/p)^T = mfma(bopX, aop, X^T)
//    - the matrix cores do the broadcast a VALU elimination needs an LDS round trip and a barrier for.  The chain per
//    pivot is MFMA -> v_readlane of two scalars -> the NEXT pivot predicted on the scalar side,
//    p_k+1 = S[k+1][k+1] - S[k][k+1]^2 / p_k, and its reciprocal (v_rcp_f64 + two Newton steps), which overlaps the
//    MFMA of step k.  Square roots are off the chain: M = unit lower L~^-1, pivots p, L^-1 = diag(p)^-1/2 M at the end.
//  * Blocked level (4 x 4 tiles, upper triangle kept):  H_pi = M_pp A_pi;  A_ij -= (D_p^-1 H_pi)^T H_pj;  the inverse
//    Y_ip = -M_ii sum_{p <= j < i} (D_j^-1 H_ji)^T Y_jp, Y_pp = M_pp;  X = diag(p)^-1/2 Y.
//    Wave 0 runs the critical path alone: base p -> H_p,p+1 -> its last update of A_p+1,p+1 -> base p + 1, all in
//    registers.  Wave c (1..3) owns COLUMN c, left-looking: it prepares A_qc at version q for q < c, the superdiagonal
//    tile and the diagonal tile's partial sum for wave 0, and row c of the inverse.  The waves meet through LDS tiles +
//    workgroup-scope flags in LDS (no barrier: a barrier would stall wave 0), everything is ready long before it is
//    asked for (a base case is ~1 300 cycles, a tile product ~250).
//  * Code size matters as much as cycles: called once per launch (the riders) the block runs at instruction-fetch
//    speed, so the base loop is rolled over the register index (uniform selects), the column programs are loops.
// LDS (the 64 KB of Potrf64Scratch), tiles row-major [16][16] f64 (conflict-free D-layout reads):
//   [0, 32 K)   slot (I, J): I <= J  A_IJ;  I > J  H_JI (the lower half of the input is never read)
//   [32 K, 64 K) slot (I, I) M_II;  I > J  Y_IJ (T_IJ while it accumulates);  (0,1) (0,2) (0,3) (1,2): M_pp^T, p = 0..3;
//                (1,3): pivots [64] + reciprocals [64];  (2,3): flags
// ---------------------------------------------------------------------------------------------
constexpr int MF_X = 32768;
constexpr int MF_SCAL = MF_X + 7 * 2048, MF_FLAGS = MF_X + 11 * 2048;
__device__ __forceinline__ constexpr int mf_slot(int I, int J) { return (4 * I + J) * 2048; }
__device__ __forceinline__ constexpr int mf_mt_slot(int p) { return MF_X + (p < 3 ? p + 1 : 6) * 2048; }
// flag words: [0..3] BASE[p]: M_pp, M_pp^T, pivots of block p are in LDS | [4..7] SUP[c]: A_c-1,c (version c-1) and the
// partial A_cc are in LDS | [8..23] HF[s][q]: H_sq is in LDS | [24..27] YROW[c]: row c of Y is in LDS
enum { MF_BASE = 0, MF_SUP = 4, MF_HF = 8, MF_YROW = 24, MF_NFLAGS = 28 };

// LDS-only hand-off: the payload's ds_writes are waited for (lgkmcnt - NOT a release fence, which would also drain the
// caller's global loads that ride through the factor), then the flag; LDS serves a wave's requests in order, so a
// reader that sees the flag sees the payload.
__device__ __forceinline__ void mf_post(char* lds, int f) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __hip_atomic_store((unsigned*)(lds + MF_FLAGS) + f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void mf_wait(char* lds, int f) {
  while (__hip_atomic_load((unsigned*)(lds + MF_FLAGS) + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u)
    __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ double4_t mf_ld(const char* lds, int off, int g, int c) {
  const char* b = lds + off + g * 128 + c * 8;
  return (double4_t){*(const double*)b, *(const double*)(b + 512), *(const double*)(b + 1024), *(const double*)(b + 1536)};
}
__device__ __forceinline__ void mf_st(char* lds, int off, int g, int c, double4_t v) {
  char* b = lds + off + g * 128 + c * 8;
  *(double*)b = v[0]; *(double*)(b + 512) = v[1]; *(double*)(b + 1024) = v[2]; *(double*)(b + 1536) = v[3];
}
// acc += X^T Y (all three in the D layout)
__device__ __forceinline__ double4_t mf_prod(double4_t acc, double4_t X, double4_t Y) {
#pragma unroll
  for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[r], Y[r], acc, 0, 0, 0);
  return acc;
}
// rows of H scaled by -1 / p (pivots of block s): the operand -(D_s^-1 H)
__device__ __forceinline__ double4_t mf_ld_scaled(const char* lds, int off, int s, int g, int c) {
  const double4_t h = mf_ld(lds, off, g, c);
  const double* ip = (const double*)(lds + MF_SCAL + 512) + 16 * s + g;
  return (double4_t){-h[0] * ip[0], -h[1] * ip[4], -h[2] * ip[8], -h[3] * ip[12]};
}
__device__ __forceinline__ double mf_readlane(double v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)__double2loint(v), l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)__double2hiint(v), l);
  return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ double mf_sel(double4_t v, int i) {          // uniform i
  const double a = (i & 1) ? v[1] : v[0], b = (i & 1) ? v[3] : v[2];
  return (i & 2) ? b : a;
}

// The base case on one wave.  A: symmetric 16 x 16 tile (D layout), consumed.  Out: X = M (unit lower, L~^-1), XT = M^T,
// pv / ipv: lane (g, c) holds pivot c and its reciprocal (1 where k >= klim: the identity padding is not eliminated).
__device__ __forceinline__ void mf_base(double4_t A, double4_t& X, double4_t& XT, double& pv, double& ipv, int klim, int g, int c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) X[r] = (g + 4 * r == c) ? 1.0 : 0.0;
  XT = X;
  pv = 1.0;
  ipv = 1.0;
  double p = mf_readlane(A[0], 0), ip = rcp_f64(p);
#pragma unroll 1
  for (int kr = 0; kr < 4; ++kr) {
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int k = 4 * kr + kg;
      if (k < klim) {
        const double uk = mf_sel(A, kr), xk = mf_sel(X, kr);             // rows k of S_k and of X: lanes g == kg
        pv = (c == k) ? p : pv;
        ipv = (c == k) ? ip : ipv;
        double pn = 1.0, ipn = 1.0;
        if (k + 1 < klim) {                                              // the next pivot, on the scalar side
          const double b = mf_readlane(uk, 16 * kg + k + 1);             // S_k[k][k+1]
          const double un = kg == 3 ? mf_sel(A, kr + 1) : uk;
          const double a = mf_readlane(un, 16 * ((kg + 1) & 3) + k + 1); // S_k[k+1][k+1]
          pn = fma(-(b * ip), b, a);
          ipn = rcp_f64(pn);
        }
        const bool rowk = g == kg, act = rowk && c > k;
        const double aop = act ? -ip * uk : 0.0, bopA = act ? uk : 0.0, bopX = rowk ? xk : 0.0;
        A = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bopA, A, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bopX, X, 0, 0, 0);
        XT = __builtin_amdgcn_mfma_f64_16x16x4f64(bopX, aop, XT, 0, 0, 0);
        p = pn;
        ip = ipn;
      }
    }
  }
}

#ifdef MF_STAMPS
__device__ unsigned long long g_mfst[32];
#define MFS(i) do { if (lane == 0) g_mfst[i] = clock64(); } while (0)
#else
#define MFS(i) do { } while (0)
#endif

template <class LoadA, class StoreX, class Hook = PotrfNoHook, class Side = PotrfNoSide>
__device__ __forceinline__ void potrf64_mf(LoadA loadA, StoreX storeX, Potrf64Scratch* sc, int tid, int* status, int col_base,
                                           int npiv, Hook after_load = Hook(), Side side = Side()) {
  char* lds = (char*)sc;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, c = lane & 15;
  if (tid == 0) { MFS(0); }
  if (npiv > 64) npiv = 64;
  const int nbase = (npiv + 15) >> 4;                                    // 16-blocks that hold real pivots
  {
    // every thread fetches 8 consecutive columns of one row; outside the real block the input is replaced by the
    // identity, so nothing depends on what the caller left in the padding
    const int R = tid >> 3, C0 = (tid & 7) * 8;
    double v[8];
    {
      double q[4];
      loadA(R, C0, q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = q[e];
      loadA(R, C0 + 4, q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 + e] = q[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (R >= npiv || C0 + e >= npiv) v[e] = (R == C0 + e) ? 1.0 : 0.0;
    __syncthreads();                                                     // loadA may have read what the scratch aliases
    after_load();
    char* dst = lds + mf_slot(R >> 4, C0 >> 4) + (R & 15) * 128 + (C0 & 15) * 8;
#pragma unroll
    for (int e = 0; e < 8; e += 2) *(pk_d2*)(dst + 8 * e) = (pk_d2){v[e], v[e + 1]};
    if (tid < MF_NFLAGS) ((unsigned*)(lds + MF_FLAGS))[tid] = 0u;
    if (tid < 64) {
      ((double*)(lds + MF_SCAL))[tid] = 1.0;
      ((double*)(lds + MF_SCAL + 512))[tid] = 1.0;
    }
  }
  __syncthreads();
  const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  if (wave == 0) {
    // ---- the critical path: base p -> H_p,p+1 -> A_p+1,p+1 -> base p + 1
    double4_t X = zero4, XT = zero4, MTs = zero4;
    double pv = 1.0, ipv = 1.0;
#pragma unroll 1
    for (int p = 0; p < nbase; ++p) {
      double4_t A;
      MFS(1 + 3 * p);
      if (p == 0) A = mf_ld(lds, mf_slot(0, 0), g, c);
      else {
        mf_wait(lds, MF_SUP + p);
        const double4_t S = mf_ld(lds, mf_slot(p - 1, p), g, c), Dg = mf_ld(lds, mf_slot(p, p), g, c);
        const double4_t H = mf_prod(zero4, XT, S), Hs = mf_prod(zero4, MTs, S);   // H_p-1,p and -(D^-1 H)
        mf_st(lds, mf_slot(p, p - 1), g, c, H);
        mf_post(lds, MF_HF + 4 * (p - 1) + p);
        A = mf_prod(Dg, Hs, H);
      }
      const int klim = npiv - 16 * p < 16 ? npiv - 16 * p : 16;
      MFS(2 + 3 * p);
      mf_base(A, X, XT, pv, ipv, klim, g, c);
      MFS(3 + 3 * p);
#pragma unroll
      for (int r = 0; r < 4; ++r) MTs[r] = -XT[r] * ipv;                 // (M^T D^-1): columns scaled
      mf_st(lds, MF_X + mf_slot(p, p), g, c, X);
      mf_st(lds, mf_mt_slot(p), g, c, XT);
      if (g == 0) {
        ((double*)(lds + MF_SCAL))[16 * p + c] = pv;
        ((double*)(lds + MF_SCAL + 512))[16 * p + c] = ipv;
      }
      mf_post(lds, MF_BASE + p);
    }
    MFS(13);
  } else if (wave < 4 && wave < nbase) {
    // ---- column / row `wave` of the block structure
    const int cc = wave;
#pragma unroll 1
    for (int q = 0; q < cc; ++q) {
      double4_t acc = mf_ld(lds, mf_slot(q, cc), g, c);                  // A_qc -> version q
#pragma unroll 1
      for (int s = 0; s < q; ++s) {
        mf_wait(lds, MF_HF + 4 * s + q);
        acc = mf_prod(acc, mf_ld_scaled(lds, mf_slot(q, s), s, g, c), mf_ld(lds, mf_slot(cc, s), g, c));
      }
      if (q == cc - 1) {
        mf_st(lds, mf_slot(q, cc), g, c, acc);                           // the superdiagonal tile: wave 0 forms H_c-1,c
        double4_t dg = mf_ld(lds, mf_slot(cc, cc), g, c);
#pragma unroll 1
        for (int s = 0; s < q; ++s) dg = mf_prod(dg, mf_ld_scaled(lds, mf_slot(cc, s), s, g, c), mf_ld(lds, mf_slot(cc, s), g, c));
        mf_st(lds, mf_slot(cc, cc), g, c, dg);
        mf_post(lds, MF_SUP + cc);
      } else {
        mf_wait(lds, MF_BASE + q);
        mf_st(lds, mf_slot(cc, q), g, c, mf_prod(zero4, mf_ld(lds, mf_mt_slot(q), g, c), acc));   // H_qc = M_qq A_qc
        mf_post(lds, MF_HF + 4 * q + cc);
      }
    }
    // row cc of the inverse: T_p = -sum_j (D_j^-1 H_jc)^T Y_jp accumulates in the slot of Y_cp; Y_cp = M_cc T_p
#pragma unroll 1
    for (int jj = 0; jj < cc; ++jj) {                                    // contributions of block row jj, in the order they become available
      if (jj == cc - 1) mf_wait(lds, MF_HF + 4 * jj + cc);               // H_c-1,c comes from wave 0
      if (jj > 0) mf_wait(lds, MF_YROW + jj);
      const double4_t hs = mf_ld_scaled(lds, mf_slot(cc, jj), jj, g, c);
#pragma unroll 1
      for (int p = 0; p <= jj; ++p) {
        const double4_t t0 = jj == p ? zero4 : mf_ld(lds, MF_X + mf_slot(cc, p), g, c);   // first contribution of column p: j = p
        mf_st(lds, MF_X + mf_slot(cc, p), g, c, mf_prod(t0, hs, mf_ld(lds, MF_X + mf_slot(jj, p), g, c)));
      }
    }
    mf_wait(lds, MF_BASE + cc);
    {
      const double4_t mt = mf_ld(lds, mf_mt_slot(cc), g, c);
#pragma unroll 1
      for (int p = 0; p < cc; ++p)
        mf_st(lds, MF_X + mf_slot(cc, p), g, c, mf_prod(zero4, mt, mf_ld(lds, MF_X + mf_slot(cc, p), g, c)));
    }
    mf_post(lds, MF_YROW + cc);
    MFS(13 + cc);
  } else if (wave >= 4) {
    const int nkb = (npiv + 3) >> 2;
    for (int kb = 0; kb < nkb; ++kb) side(kb, nkb);
  }
  if (tid == 0) { MFS(17); }
  __syncthreads();
  if (tid == 0) { MFS(18); }
  // ---- L^-1 = diag(p)^-1/2 Y, rows >= npiv the identity; a non-positive pivot is reported (smallest index wins)
  {
    const int R = tid >> 3, I = R >> 4;
    const double pR = ((const double*)(lds + MF_SCAL))[R];
    if ((tid & 7) == 0 && R < npiv && !(pR > 0.0)) {
      const int want = col_base + R + 1;
      int cur = *(volatile int*)status;
      while (cur == 0 || cur > want) {
        const int prev = atomicCAS(status, cur, want);
        if (prev == cur) break;
        cur = prev;
      }
    }
    const double rs = rsqrt_f64(pR);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int C0 = (tid & 7) * 8 + 4 * h, J = C0 >> 4;
      double v[4];
      const char* src = lds + MF_X + mf_slot(I, J <= I ? J : I) + (R & 15) * 128 + (C0 & 15) * 8;
      const pk_d2 a = *(const pk_d2*)src, b = *(const pk_d2*)(src + 16);
      const double y[4] = {a[0], a[1], b[0], b[1]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int C = C0 + e;
        v[e] = R >= npiv ? (R == C ? 1.0 : 0.0) : (C <= R ? y[e] * rs : 0.0);
      }
      storeX(R, C0, v);
    }
  }
}

}  // namespace
