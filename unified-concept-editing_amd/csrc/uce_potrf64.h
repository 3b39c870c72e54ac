// 64x64 f64 Cholesky + inverse building block, shared by uce_solve.hip (k_potrf_first,
// k_potrf_step) and uce_lowrank2.hip (the launch that runs it beside the projection GEMM).
#pragma once
#include "uce_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Cholesky inverse of one 64x64 SPD block: A = L L^T, the block hands out L^-1.
// Right-looking block elimination on [A | I], TWO pivots per step (32 steps, one barrier each).  Threads own 4x4
// register tiles of the current Schur complement (a) and of X (starts as I, ends as L^-1 up to the row scaling).  At
// step (k, k+1) the owners publish columns k, k+1 of the Schur complement and rows k, k+1 of X into LDS LINES;
// everybody reads the 2x2 pivot block P, inverts it, and applies the rank-2 update
//     a_ij -= [a_ik a_i,k+1] P^-1 [a_jk a_j,k+1]^T ,   x_ij -= [a_ik a_i,k+1] P^-1 [x_kj x_k+1,j]^T .
// The published lines are final as published (later register updates of eliminated rows/columns are harmless
// garbage that is never read), so nothing is masked.  L^-1 is assembled from the row lines in one pass at the end:
// an even row is scaled by 1/sqrt(p00); an odd one first gets the pivot-k elimination it skipped:
// (r1 - r0 p01/p00) / sqrt(p11 - p01^2/p00).
// ---------------------------------------------------------------------------------------------
struct Potrf64Scratch {
  double col[64][64];   // column line k: column k of the Schur complement when it was published (element order: pk_pos)
  double row[64][64];   // row line k: row k of the partial inverse when it was published
};

// 1 / sqrt(v): hardware estimate + two Newton steps (the sqrt + divide sequence of `1.0 / sqrt(v)` is ~10x the code)
static __device__ __forceinline__ double rsqrt_f64(double v) {
  double y = __builtin_amdgcn_rsq(v);
  y = y * fma(-0.5 * v * y, y, 1.5);
  y = y * fma(-0.5 * v * y, y, 1.5);
  return y;
}

static __device__ __forceinline__ double rcp_f64(double v) {
  double r = __builtin_amdgcn_rcp(v);
  r = fma(fma(-v, r, 1.0), r, r);
  r = fma(fma(-v, r, 1.0), r, r);
  return r;
}

// ---------------------------------------------------------------------------------------------
// How the elimination is laid out on the CU (tools/ubench/potrf.hip times the block alone).  A pair step costs what
// its LDS round trip costs - write lines, barrier, ~10 b128 reads per wave at ~20-30 cycles each, 60 f64 VALU
// instructions at 4 cycles - so the layout minimises LDS instructions, exposed latencies and branches:
//  * only LIVE tiles carry work, packed into four waves (one per SIMD): tile (ti, tj) of the matrix is live while
//    kb <= tj and only the lower block triangle is ever read; tile (ti, tj) of X only once kb >= tj (row K of X is
//    zero right of column K + 1).  In 16-column groups:  matrix 0..15: kb < 4 | 16..31: kb < 8 | 32..63: always;
//    X 0..15: always | 16..31: kb >= 4 | 32..47: kb >= 8 | 48..63: kb >= 12.  So
//        wave 0: matrix cols 32..63 (48 lanes, lower block triangle) + X cols 48..63 (16 lanes)
//        wave 1: X cols 0..15
//        wave 2: matrix cols 16..31 (rows >= 16); from kb = 8 on: X cols 32..47 (rows >= 32)
//        wave 3: matrix cols 0..15;               from kb = 4 on: X cols 16..31 (rows >= 16)
//    Waves 4-7 of the 512-thread callers only meet the barriers and come back for the assembly.
//  * X tiles are held TRANSPOSED, so both roles run one instruction stream:  w[b][a] -= [F0[a] F1[a]] Q [S0[b] S1[b]]^T
//    with S = the column lines at the tile's second index and F = the column (matrix) or row (X) lines at its first
//    index; the line a thread reads F from is the line it publishes to.  No role branches in the step; tiles that
//    are not live yet subtract exact zeros (the row lines start as the identity), dead tiles compute garbage nobody
//    reads.
//  * every LDS read of a step is issued right after the barrier (one exposed latency, not three dependent ones);
//  * the two columns the NEXT pair publishes are updated first and written to LDS before the rest of the update;
//  * run-once code (tile load, line initialisation, assembly) executes at instruction-fetch speed, ~1.7 cycles per
//    byte: no divergent blocks (the compiler parks them far away), vector loads, one assembly pass for L^-1 only.
// `loadA(row, col, v[4])` supplies 4 consecutive columns of a row of the SPD block (it may read LDS that the scratch
// aliases: everything is fetched, then a barrier, then the scratch is touched).
// ---------------------------------------------------------------------------------------------
typedef double pk_d2 __attribute__((ext_vector_type(2)));

// A line (512 B) keeps element j at pk_pos(j): the first two elements of every 4-group in its lower half, the last two
// in its upper half - the 16 tiles' 16-byte pieces are contiguous, so the b128 reads of a step are conflict-free
// (with the natural order they sit at a 32-byte stride: two-way conflicts, measured +27 % per read).
__device__ __forceinline__ constexpr int pk_pos(int j) { return ((j & 2) << 4) + ((j >> 2) << 1) + (j & 1); }

template <int KR>
__device__ __forceinline__ void pk_step(double (&w)[4][4], char* lds, unsigned foff, unsigned soff, int i2, int kb, int kstop) {
  const int K = 4 * kb + KR;
  char* lk = lds + K * 512;                                  // line K (column array; the row array sits 32 KB above)
  const unsigned poff = (KR ? 256u : 0u) + 16u * (unsigned)kb;
  const pk_d2 pp = *(const pk_d2*)(lk + poff);               // col[K][K], col[K][K + 1]
  const double p11 = *(const double*)(lk + 512 + poff + 8);  // col[K + 1][K + 1]
  pk_d2 s0[2], s1[2], f0[2], f1[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    s0[h] = *(const pk_d2*)(lk + soff + 256 * h);
    s1[h] = *(const pk_d2*)(lk + 512 + soff + 256 * h);
    f0[h] = *(const pk_d2*)(lk + foff + 256 * h);
    f1[h] = *(const pk_d2*)(lk + 512 + foff + 256 * h);
  }
  const double p00 = pp[0], p01 = pp[1];
  const double idet = rcp_f64(fma(p00, p11, -p01 * p01));
  const double q00 = p11 * idet, q01 = -p01 * idet, q11 = p00 * idet;
  double u[4], v[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const double c0 = s0[b >> 1][b & 1], c1 = s1[b >> 1][b & 1];
    u[b] = fma(c0, q00, c1 * q01);
    v[b] = fma(c0, q01, c1 * q11);
  }
  constexpr int BN = (KR + 2) & 3;                            // the columns the next pair publishes: updated first
#pragma unroll
  for (int b = BN; b < BN + 2; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) w[b][a] = fma(-f0[a >> 1][a & 1], u[b], fma(-f1[a >> 1][a & 1], v[b], w[b][a]));
  const int kbn = KR == 0 ? kb : kb + 1;
  if (i2 == kbn && K + 2 < kstop) {
    char* ln = lk + 1024 + foff;
    *(pk_d2*)(ln) = (pk_d2){w[BN][0], w[BN][1]};
    *(pk_d2*)(ln + 256) = (pk_d2){w[BN][2], w[BN][3]};
    *(pk_d2*)(ln + 512) = (pk_d2){w[BN + 1][0], w[BN + 1][1]};
    *(pk_d2*)(ln + 768) = (pk_d2){w[BN + 1][2], w[BN + 1][3]};
  }
#pragma unroll
  for (int b = KR; b < KR + 2; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) w[b][a] = fma(-f0[a >> 1][a & 1], u[b], fma(-f1[a >> 1][a & 1], v[b], w[b][a]));
}

#ifdef PK_STAMPS
__device__ unsigned long long g_pkst[8];
#define PKS(i) do { if (tid == 0) g_pkst[i] = clock64(); } while (0)
#else
#define PKS(i) do { } while (0)
#endif

struct PotrfNoHook {
  __device__ __forceinline__ void operator()() const {}
};

// `after_load`: called by every thread right after the barrier that follows the tile load (the persistent Cholesky posts a
// hand-off flag there: the stores it covers were drained by the waves before that barrier, off the critical path).
struct PotrfNoSide {
  __device__ __forceinline__ void operator()(int, int) const {}
};

// `side(kb, nkb)`: called once per elimination iteration by waves 4-7 - they carry no tile and otherwise only meet the
// barriers - so a caller can have them fetch what comes after the factor while waves 0-3 eliminate.
template <class LoadA, class StoreX, class Hook = PotrfNoHook, class Side = PotrfNoSide>
__device__ __forceinline__ void potrf64_pk(LoadA loadA, StoreX storeX, Potrf64Scratch* sc, int tid, int* status, int col_base,
                                           int npiv, Hook after_load = Hook(), Side side = Side()) {
  const int wave = tid >> 6, lane = tid & 63;
  PKS(0);
  // role 0: matrix tile, (i1, i2) = (row block, column block); role 1: X tile, transposed, (i1, i2) = (column block, row block)
  int role = -1, i1 = 0, i2 = -1;          // what this thread carries now (i2 < 0: nothing - it never publishes)
  int sw = 64, j1 = 0, j2 = -1;            // at kb == sw it becomes the X tile (j1, j2)
  if (wave == 0) {
    if (lane < 32) { role = 0; i2 = 8 + (lane >> 3); i1 = 8 + (lane & 7); }
    else if (lane < 48) { role = 0; i2 = 12 + ((lane - 32) >> 2); i1 = 12 + (lane & 3); }
    else { role = 1; i1 = 12 + ((lane - 48) >> 2); i2 = 12 + (lane & 3); }
  } else if (wave == 1) {
    role = 1; i2 = lane & 15; i1 = lane >> 4;
  } else if (wave == 2) {
    if (lane < 48) { role = 0; i2 = 4 + lane / 12; i1 = 4 + lane % 12; }
    sw = 8;
    if (lane < 32) { j2 = 8 + (lane & 7); j1 = 8 + (lane >> 3); }
  } else if (wave == 3) {
    role = 0; i1 = lane & 15; i2 = lane >> 4;
    sw = 4;
    if (lane < 48) { j1 = 4 + lane / 12; j2 = 4 + lane % 12; }
  }
  double w[4][4];                          // w[b][a]: element (4 i1 + a, 4 i2 + b) of the tile's matrix (X: of X^T)
  {
    // every lane loads (no divergent blocks: run-once code is fetch-bound, and the compiler parks masked blocks far away);
    // lanes that carry no matrix tile fetch tile (0, 0) and drop it
    const int li = role == 0 ? i1 : 0, lj = role == 0 ? i2 : 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double r4[4];
      loadA(4 * li + a, 4 * lj, r4);                                    // 4 consecutive columns of one row
#pragma unroll
      for (int b = 0; b < 4; ++b) w[b][a] = role == 0 ? r4[b] : ((a == b && i1 == i2) ? 1.0 : 0.0);   // X tiles start as the identity
    }
  }
  __syncthreads();                                                      // loadA may have read what the scratch aliases
  after_load();
  PKS(1);
  const int nkb = npiv >= 64 ? 16 : ((npiv + 3) >> 2), kstop = 4 * nkb;
  char* lds = (char*)sc;
  unsigned foff = (role == 1 ? 32768u : 0u) + 16u * (unsigned)i1, soff = 16u * (unsigned)(i2 < 0 ? 0 : i2);
  {
    // row lines start as the identity: a tile of X that is not live yet subtracts exact zeros, and a diagonal tile
    // that joins its wave only at a segment boundary is still pristine when its first pair of rows is due.  (The
    // pieces the first pair publishes - rows 0, 1, columns 0..15 - are left to it: no barrier in between.)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = tid + 512 * q, r = ch >> 5, i = ch & 15, j0 = 4 * i + 2 * ((ch >> 4) & 1);   // 16-byte piece: elements j0, j0 + 1 of line r
      if (!(r < 2 && i < 4)) *(pk_d2*)(lds + 32768 + ch * 16) = (pk_d2){r == j0 ? 1.0 : 0.0, r == j0 + 1 ? 1.0 : 0.0};
    }
  }
  if (i2 == 0) {                                                        // the first pair
    *(pk_d2*)(lds + foff) = (pk_d2){w[0][0], w[0][1]};
    *(pk_d2*)(lds + foff + 256) = (pk_d2){w[0][2], w[0][3]};
    *(pk_d2*)(lds + foff + 512) = (pk_d2){w[1][0], w[1][1]};
    *(pk_d2*)(lds + foff + 768) = (pk_d2){w[1][2], w[1][3]};
  }
  __syncthreads();
  PKS(2);
  // three segments: the tile map changes at kb = 4 (wave 3) and kb = 8 (wave 2)
#pragma unroll 1
  for (int seg = 0; seg < 3; ++seg) {
    const int kb0 = seg == 0 ? 0 : 4 * seg, kb1 = seg == 2 ? 16 : 4 * seg + 4;
    if (kb0 >= nkb) break;
    if (kb0 == sw) {
      i1 = j1; i2 = j2;
      foff = 32768u + 16u * (unsigned)i1;
      soff = 16u * (unsigned)(i2 < 0 ? 0 : i2);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) w[b][a] = (a == b && i1 == i2) ? 1.0 : 0.0;
    }
    const int kbe = kb1 < nkb ? kb1 : nkb;
#pragma unroll 1
    for (int kb = kb0; kb < kbe; ++kb) {
      if (wave < 4) pk_step<0>(w, lds, foff, soff, i2, kb, kstop);
      else side(kb, nkb);
      __syncthreads();
      if (wave < 4) pk_step<2>(w, lds, foff, soff, i2, kb, kstop);
      __syncthreads();
    }
  }
  PKS(3);
  // assembly of L^-1 from the row lines, all 512 threads: thread = rows k = 4 A + 2 hq and k + 1 (one pivot pair) x
  // columns 4 B .. 4 B + 3.  Row k is scaled by 1 / sqrt(p00); row k + 1 first gets the pivot-k elimination it skipped:
  // (r1 - r0 p01 / p00) / sqrt(p11 - p01^2 / p00).  Every thread derives its pair's scalars itself (16 threads per
  // pair repeat ~20 instructions: cheaper than a phase + barrier of run-once code, which executes at instruction-
  // fetch speed, ~1.7 cycles per byte).  storeX(row, col, v[4]) receives 4 consecutive columns of a row.
  // L itself is not assembled: nothing reads the DIAGONAL blocks of the factor - the blocked factorisation and every
  // solve work with L_jj^-1 (the off-diagonal blocks L_ij = M_ij L_jj^-T are formed by the tile update).
  {
    const int hq = tid >> 8, t256 = tid & 255;
    const int A = t256 >> 4, B = t256 & 15;
    const int k = 4 * A + 2 * hq;
    const bool pad = k >= kstop;                                        // identity padding: never published
    const char* lb = lds + 32768 + k * 512 + 16 * B;
    const pk_d2 e0[2] = {*(const pk_d2*)(lb), *(const pk_d2*)(lb + 256)};
    const pk_d2 o0[2] = {*(const pk_d2*)(lb + 512), *(const pk_d2*)(lb + 768)};
    const char* pl = lds + k * 512 + (hq ? 256 : 0) + 16 * A;           // pk_pos(k) of column line k
    const pk_d2 pp = *(const pk_d2*)pl;
    const double p00 = pad ? 1.0 : pp[0], p01 = pad ? 0.0 : pp[1], p11 = pad ? 1.0 : *(const double*)(pl + 512 + 8);
    const double g = p01 * rcp_f64(p00);
    const double p11e = fma(-g, p01, p11);                              // pivot k + 1 after eliminating pivot k
    if (B == 0 && (!(p00 > 0.0) || !(p11e > 0.0))) {                    // smallest failing pivot index wins
      const int want = col_base + k + (!(p00 > 0.0) ? 1 : 2);
      int cur = *(volatile int*)status;
      while (cur == 0 || cur > want) {
        const int prev = atomicCAS(status, cur, want);
        if (prev == cur) break;
        cur = prev;
      }
    }
    const double rs0 = rsqrt_f64(p00), rs1 = rsqrt_f64(p11e);           // (a non-positive pivot gives NaN / inf; reported above)
    double v0[4], v1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = 4 * B + e;
      const double x0 = e0[e >> 1][e & 1] * rs0, x1 = fma(-g, e0[e >> 1][e & 1], o0[e >> 1][e & 1]) * rs1;
      v0[e] = pad ? (k == col ? 1.0 : 0.0) : (col <= k ? x0 : 0.0);
      v1[e] = pad ? (k + 1 == col ? 1.0 : 0.0) : (col <= k + 1 ? x1 : 0.0);
    }
    storeX(k, 4 * B, v0);
    storeX(k + 1, 4 * B, v1);
  }
  PKS(5);
}

// The 64 x 64 factor the callers use.  History, all measured on MI355X with tools/ubench/potrf.hip (cycles for the full
// block, 2.4 GHz):  8 waves, every tile at every step, branches per role, L and L^-1 assembled with strided line reads:
// 43 400;  this form: 27 100 (pair step 1400 -> 800 cycles, run-once code 7 000 -> 3 200).  Also tried and dropped:
// rank-4 steps with the trailing update on v_mfma_f64_16x16x4_f64 (K = 4 is exactly one MFMA per 16 x 16 tile, 16 steps
// instead of 32; 49 000 - 53 000: a 4 x 4 pivot-block inverse is a longer dependent chain than two 2 x 2 ones and the
// update arithmetic the MFMA removes is a fifth of the step), four pivots per barrier on the VALU (48 000), a fully
// unrolled loop (instruction-fetch bound: 50 KB of run-once code took 75 000 - 90 000).  Round 3: a blocked form whose
// 16 x 16 base case runs inside ONE wave with no barrier and no LDS round trip per pivot (the pivot row is an MFMA operand
// vector as it lies in the D layout; tools/ubench/potrf_mf.h, correct on every npiv): 46 900 cycles - a pivot costs
// ~550 cycles there, not the ~130 of its dependency chain (68-cycle f64 MFMA, 19-cycle v_rcp_f64, 5-cycle f64 FMA:
// tools/ubench/lat.hip), because a wave64 VALU instruction issues in 4 cycles and the step needs ~50 of them (uniform
// register selects, operand masks, the scalar-side pivot prediction) plus three 64-cycle f64 MFMAs on one SIMD.
#define UCE_POTRF64 potrf64_pk

// Factors diagonal block 0 (512 threads).  With nsplit > 1 the block is first summed from the split-K slabs of the
// Gram kernel (single-block systems skip the separate reduction launch).
static __device__ __forceinline__ void potrf_first_body8(const double* __restrict__ M, int n, int nsplit,
                                                         size_t slab_stride, double* __restrict__ Lmat,
                                                         double* __restrict__ Linv, int* status,
                                                         Potrf64Scratch* sc, int n_valid = 1 << 30) {
  const int npiv0 = n_valid < 64 ? n_valid : 64;
  const int tid = threadIdx.x;
  UCE_POTRF64([&](int row, int col, double (&v)[4]) {
    const size_t off = (size_t)row * n + col;
    pk_d2 a = *(const pk_d2*)(M + off), b = *(const pk_d2*)(M + off + 2);
    for (int sp = 1; sp < nsplit; ++sp) {                                          // index order
      a += *(const pk_d2*)(M + (size_t)sp * slab_stride + off);
      b += *(const pk_d2*)(M + (size_t)sp * slab_stride + off + 2);
    }
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
  }, [&](int row, int col, const double (&v)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) Linv[row * 64 + col + e] = v[e];
  }, sc, tid, status, 0, npiv0);
}


// ---------------------------------------------------------------------------------------------
// One trailing tile (i, k), j < k <= i, of step j of the blocked right-looking Cholesky (block 64), by
// one 512-thread workgroup:  P_i = M_ij L_jj^-T (= L_ij),  P_k = M_kj L_jj^-T,  M_ik -= P_i P_k^T;
// tile (i, j+1) publishes L_ij; tile (j+1, j+1) then factors itself (-> L, L^-1 of block j+1).
// Called once per workgroup by k_potrf_step (uce_solve.hip) and in a loop by the last rider block of
// the projection launch (uce_lowrank2.hip).  smem_raw: POTRF_STEP_SMEM bytes.
// ---------------------------------------------------------------------------------------------
constexpr int LD = 66;  // row stride (doubles) of the 64x64 LDS tiles: conflict-free ds_read_b64
constexpr size_t POTRF_STEP_SMEM = 3 * 64 * LD * sizeof(double);

// one wave's 32x32 quadrant of  acc += sign * P[rows] * Q[cols]^T  (both tiles row-major in LDS,
// contraction index contiguous)
__device__ __forceinline__ void quad_nt(double4_t (&acc)[2][2], const double (*P)[LD],
                                        const double (*Q)[LD], int row0, int col0, int lane,
                                        double sign, int kb0 = 0, int kb1 = 16) {
  const int r = lane & 15, kk = lane >> 4;
#pragma unroll 4
  for (int kb = kb0; kb < kb1; ++kb) {
    const int t = kb * 4 + kk;
    const double a0 = sign * P[row0 + r][t], a1 = sign * P[row0 + 16 + r][t];
    const double b0 = Q[col0 + r][t], b1 = Q[col0 + 16 + r][t];
    acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
    acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
    acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
    acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
  }
}

__device__ __forceinline__ void quad_zero(double4_t (&acc)[2][2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};
}

// accumulator quadrant <-> memory (D layout of v_mfma_f64_16x16x4: row = (lane>>4) + 4r, col = lane&15)
template <typename F>
__device__ __forceinline__ void quad_foreach(int row0, int col0, int lane, F f) {
  const int c = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) f(m, n, r, row0 + m * 16 + rq + 4 * r, col0 + n * 16 + c);
}

static __device__ __forceinline__ void potrf_step_tile(double* __restrict__ M, int n, int j, int i, int k,
                                                       double* __restrict__ Lmat, double* __restrict__ Linv,
                                                       int* status, unsigned char* smem_raw, int n_valid = 1 << 30) {
  double (*Li)[LD] = (double (*)[LD])smem_raw;                       // L_jj^-1
  double (*Mi)[LD] = (double (*)[LD])(smem_raw + 64 * LD * 8);       // M_ij  -> P_i
  double (*Mk)[LD] = (double (*)[LD])(smem_raw + 2 * 64 * LD * 8);   // M_kj  -> P_k

  // 8 waves: quadrant = w & 3, half = w >> 2.  Half 0 forms P_i while half 1 forms P_k; half 0 does the
  // tile update; the diagonal tile is then factored by all 8 waves (potrf64_reg8).
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = w >> 2, wq = w & 3;
  const int wr = (wq >> 1) * 32, wc = (wq & 1) * 32;
  const bool diag = (i == k);

  const double* Linv_j = Linv + (size_t)j * 64 * 64;
  for (int e = tid; e < 64 * 64; e += 512) {
    const int r = e >> 6, c = e & 63;
    Li[r][c] = Linv_j[e];
    Mi[r][c] = M[(size_t)(i * 64 + r) * n + j * 64 + c];
    if (!diag) Mk[r][c] = M[(size_t)(k * 64 + r) * n + j * 64 + c];
  }
  __syncthreads();

  // P_i = M_ij L_jj^-T (half 0) ; P_k likewise (half 1)
  // (splitting the contraction of a diagonal tile over both halves was measured: the two extra barriers cost more
  //  than the 32 MFMAs per wave they save - the chain is latency-, not MFMA-bound)
  double4_t pp[2][2];
  quad_zero(pp);
  if (half == 0) quad_nt(pp, Mi, Li, wr, wc, lane, 1.0);
  else if (!diag) quad_nt(pp, Mk, Li, wr, wc, lane, 1.0);
  __syncthreads();
  if (half == 0)
    quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) { Mi[row][col] = pp[m][nn][r]; });
  else if (!diag)
    quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) { Mk[row][col] = pp[m][nn][r]; });
  __syncthreads();

  // tile (i, j+1) publishes L_ij
  if (k == j + 1) {
    for (int e = tid; e < 64 * 64; e += 512) {
      const int r = e >> 6, c = e & 63;
      Lmat[(size_t)(i * 64 + r) * n + j * 64 + c] = Mi[r][c];
    }
  }

  // M_ik -= P_i P_k^T   (half 0)
  const bool factor_here = diag && i == j + 1;
  double4_t acc[2][2];
  if (half == 0) {
    quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) {
      acc[m][nn][r] = M[(size_t)(i * 64 + row) * n + k * 64 + col];
    });
    quad_nt(acc, Mi, diag ? Mi : Mk, wr, wc, lane, -1.0);
    if (!factor_here) {
      quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) {
        M[(size_t)(i * 64 + row) * n + k * 64 + col] = acc[m][nn][r];
      });
    }
  }
  if (!factor_here) return;
  // the next diagonal block: factor it now (accumulators -> LDS tile -> 4x4 register sub-blocks)
  __syncthreads();
  if (half == 0)
    quad_foreach(wr, wc, lane, [&](int m, int nn, int r, int row, int col) { Li[row][col] = acc[m][nn][r]; });
  __syncthreads();
  Potrf64Scratch* sc = (Potrf64Scratch*)smem_raw;    // aliases the tile regions: the factor fetches Li, then a barrier, then the scratch
  const int npiv = (n_valid - i * 64) < 64 ? (n_valid - i * 64) : 64;
  double* Linv_n = Linv + (size_t)i * 64 * 64;
  UCE_POTRF64([&](int row, int col, double (&v)[4]) {
                const pk_d2 a = *(const pk_d2*)&Li[row][col], b = *(const pk_d2*)&Li[row][col + 2];
                v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
              },
              [&](int row, int col, const double (&v)[4]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Linv_n[row * 64 + col + e] = v[e];
              },
              sc, tid, status, i * 64, npiv);
}

}  // namespace
