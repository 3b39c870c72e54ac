// f64 blocked Cholesky + triangular solves for the UCE normal equations
// (reference: `torch.inverse(mat2.float())` at uce_sd_erase.py:82 / uce_sd_debias.py:140, an fp32
// LU inverse recomputed per module; here ONE f64 SPD solve shared by all modules).
//
// Structure (block size 64, everything f64 on v_mfma_f64_16x16x4_f64):
//   k_potrf_first : factor diagonal block 0            -> L_00, L_00^-1
//   k_potrf_step j: one workgroup per trailing tile (i,k), j < k <= i:
//                     P_i = M_ij L_jj^-T (= L_ij),  P_k = M_kj L_jj^-T,  M_ik -= P_i P_k^T
//                   tile (i, j+1) publishes L_ij; tile (j+1, j+1) then factors itself.
//   A chain of launches on one stream replaces grid barriers (a kernel boundary is ~1.5 us on
//   MI355X, cheaper than any software grid barrier, and needs no residency assumptions).
//   k_trisolve    : one workgroup per 16 right-hand-side columns: forward then backward
//                   substitution with the inverted diagonal blocks; Y lives in LDS (n <= 1024).
#include "uce_common.h"
#include "uce_h2split.h"
#include "uce_gram_tile.h"
#include "uce_potrf64.h"
#include "uce_potrf_la.h"
#include <cstdlib>

namespace {

// (tri_decode: uce_gram_tile.h)
// Factors diagonal block 0 (body shared with the fused projection+factor launch, uce_potrf64.h).
__global__ __launch_bounds__(512) void k_potrf_first(const double* __restrict__ M, int n, int nsplit,
                                                     size_t slab_stride, double* __restrict__ Lmat,
                                                     double* __restrict__ Linv, int* status, int n_valid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // every factorisation starts here: reset the status word in this launch (a hipMemsetAsync costs a 4-5 us
  // fill kernel of its own); the first barrier inside the factor orders it before any failure report
  if (threadIdx.x == 0) *status = 0;
  potrf_first_body8(M, n, nsplit, slab_stride, Lmat, Linv, status, (Potrf64Scratch*)smem_raw, n_valid);
}

__global__ __launch_bounds__(512) void k_potrf_step(double* __restrict__ M, int n, int j,
                                                    double* __restrict__ Lmat,
                                                    double* __restrict__ Linv, int* status, int n_valid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int ta, tb;
  tri_decode(blockIdx.x, ta, tb);
  potrf_step_tile(M, n, j, j + 1 + ta, j + 1 + tb, Lmat, Linv, status, smem_raw, n_valid);
}

// (the persistent factorisation itself: uce_potrf_la.h)
#ifdef UCE_CHAIN_DEBUG
extern "C" int uce_debug_read_la(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_la_dbg), sizeof(g_la_dbg));
}
#endif

__global__ __launch_bounds__(512) void k_potrf_la(PotrfLaJob j) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  potrf_la_body(j, (int)blockIdx.x, (int)gridDim.x, smem_raw);
}

// ------------------------------------------------------------------------------------------
// triangular solves: X = L^-T L^-1 RHS for 16 columns per workgroup
// ------------------------------------------------------------------------------------------
template <bool RHS32>
__global__ __launch_bounds__(256) void k_trisolve(const double* __restrict__ Lmat,
                                                  const double* __restrict__ Linv, int n, int m,
                                                  const double* __restrict__ rhs64,
                                                  const float* __restrict__ rhs32, int rhs_rows,
                                                  float* __restrict__ out, int out_rows,
                                                  double* __restrict__ Yg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double (*tmp)[16] = (double (*)[16])smem_raw;          // [64][16]
  double* Ylds = (double*)(smem_raw + 64 * 16 * 8);      // [n][16] when it fits
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c0 = blockIdx.x * 16;
  const int c = lane & 15, kk = lane >> 4;
  const int nb = n / 64;
  // Y addressing: LDS [n][16] or global scratch [n][m]
  double* Y = Yg ? (Yg + c0) : Ylds;
  const int ldy = Yg ? m : 16;

  // ---------------- forward: Y_k = Linv_kk (RHS_k - sum_{j<k} L_kj Y_j)
  for (int kb = 0; kb < nb; ++kb) {
    const int r0 = kb * 64 + w * 16;  // this wave's 16 rows
    double4_t acc0, acc1 = (double4_t){0.0, 0.0, 0.0, 0.0}, acc2 = acc1, acc3 = acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + kk + 4 * r;
      double v;
      if (RHS32) v = (row < rhs_rows) ? (double)rhs32[(size_t)row * m + c0 + c] : 0.0;
      else v = rhs64[(size_t)row * m + c0 + c];
      acc0[r] = v;
    }
    const double* Lrow = Lmat + (size_t)(r0 + c) * n;  // A operand: row = r0 + (lane&15)
    // the diagonal-block fragments do not depend on the sum: fetch them first
    const double* Lid = Linv + (size_t)kb * 64 * 64 + (size_t)(w * 16 + c) * 64;
    double li_f[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) li_f[t] = Lid[t * 4 + kk];
    // L_kj fragments: 16-byte loads (lane (i, kk) takes columns 8u + 2kk, +1 of row i; the same k
    // permutation is applied to the Y side, which is free in LDS), two tiles in flight ahead of the
    // MFMAs (strided L2 reads take ~2k cycles here, a 16-MFMA tile only ~1k).
    typedef double double2_t __attribute__((ext_vector_type(2)));
    auto ld_tile = [&](int jb, double2_t (&f)[8]) {
#pragma unroll
      for (int u = 0; u < 8; ++u) f[u] = *(const double2_t*)(Lrow + jb * 64 + 8 * u + 2 * kk);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_tile = [&](int jb, const double2_t (&f)[8]) {
      double bf[16];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bf[2 * u] = Y[(size_t)(jb * 64 + 8 * u + 2 * kk) * ldy + c];
        bf[2 * u + 1] = Y[(size_t)(jb * 64 + 8 * u + 2 * kk + 1) * ldy + c];
      }
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        acc0 = mfma_f64(-f[u][0], bf[2 * u], acc0);
        acc1 = mfma_f64(-f[u][1], bf[2 * u + 1], acc1);
        acc2 = mfma_f64(-f[u + 1][0], bf[2 * u + 2], acc2);
        acc3 = mfma_f64(-f[u + 1][1], bf[2 * u + 3], acc3);
      }
    };
    {
      double2_t fA[8], fB[8], fC[8];
      if (kb > 0) ld_tile(0, fA);
      if (kb > 1) ld_tile(1, fB);
      int jb = 0;
      for (; jb + 2 < kb; jb += 3) {
        ld_tile(jb + 2, fC);
        mma_tile(jb, fA);
        ld_tile(jb + 3 < kb ? jb + 3 : jb + 2, fA);
        mma_tile(jb + 1, fB);
        ld_tile(jb + 4 < kb ? jb + 4 : jb + 2, fB);
        mma_tile(jb + 2, fC);
      }
      if (jb < kb) mma_tile(jb, fA);
      if (jb + 1 < kb) mma_tile(jb + 1, fB);
    }
    acc0 = (acc0 + acc1) + (acc2 + acc3);
#pragma unroll
    for (int r = 0; r < 4; ++r) tmp[w * 16 + kk + 4 * r][c] = acc0[r];
    __syncthreads();
    double4_t y0 = (double4_t){0.0, 0.0, 0.0, 0.0}, y1 = y0;
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      const int k0 = t * 4 + kk;
      y0 = mfma_f64(li_f[t], tmp[k0][c], y0);
      y1 = mfma_f64(li_f[t + 1], tmp[k0 + 4][c], y1);
    }
    y0 += y1;
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[(size_t)(r0 + kk + 4 * r) * ldy + c] = y0[r];
    __syncthreads();
  }

  // ---------------- backward: X_k = Linv_kk^T (Y_k - sum_{j>k} L_jk^T X_j)
  for (int kb = nb - 1; kb >= 0; --kb) {
    const int r0 = kb * 64 + w * 16;
    double4_t acc0, acc1 = (double4_t){0.0, 0.0, 0.0, 0.0}, acc2 = acc1, acc3 = acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] = Y[(size_t)(r0 + kk + 4 * r) * ldy + c];
    // diagonal-block fragments (transposed access) first, then the L_jk^T tiles one ahead
    const double* Lid = Linv + (size_t)kb * 64 * 64;
    double li_f[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) li_f[t] = Lid[(size_t)(t * 4 + kk) * 64 + w * 16 + c];
    // L_jk^T tiles (A[i][k] = L[k][r0 + i]: 16 consecutive rows of one column block per load
    // instruction, 128 B per row group), two tiles in flight ahead of the MFMAs
    auto ld_tile = [&](int jb, double (&f)[16]) {
#pragma unroll
      for (int t = 0; t < 16; ++t) f[t] = Lmat[(size_t)(jb * 64 + t * 4 + kk) * n + r0 + c];
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_tile = [&](int jb, const double (&f)[16]) {
      double bf[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) bf[t] = Y[(size_t)(jb * 64 + t * 4 + kk) * ldy + c];
#pragma unroll
      for (int t = 0; t < 16; t += 4) {
        acc0 = mfma_f64(-f[t], bf[t], acc0);
        acc1 = mfma_f64(-f[t + 1], bf[t + 1], acc1);
        acc2 = mfma_f64(-f[t + 2], bf[t + 2], acc2);
        acc3 = mfma_f64(-f[t + 3], bf[t + 3], acc3);
      }
    };
    {
      double fA[16], fB[16], fC[16];
      const int ntile = nb - 1 - kb;                  // tiles jb = nb-1 .. kb+1, index q -> jb = nb-1-q
      if (ntile > 0) ld_tile(nb - 1, fA);
      if (ntile > 1) ld_tile(nb - 2, fB);
      int q = 0;
      for (; q + 2 < ntile; q += 3) {
        ld_tile(nb - 1 - (q + 2), fC);
        mma_tile(nb - 1 - q, fA);
        ld_tile(nb - 1 - (q + 3 < ntile ? q + 3 : q + 2), fA);
        mma_tile(nb - 2 - q, fB);
        ld_tile(nb - 1 - (q + 4 < ntile ? q + 4 : q + 2), fB);
        mma_tile(nb - 3 - q, fC);
      }
      if (q < ntile) mma_tile(nb - 1 - q, fA);
      if (q + 1 < ntile) mma_tile(nb - 2 - q, fB);
    }
    acc0 = (acc0 + acc1) + (acc2 + acc3);
    __syncthreads();  // every wave has finished reading Y_k rows of this block before tmp reuse
#pragma unroll
    for (int r = 0; r < 4; ++r) tmp[w * 16 + kk + 4 * r][c] = acc0[r];
    __syncthreads();
    double4_t x0 = (double4_t){0.0, 0.0, 0.0, 0.0}, x1 = x0;
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      const int k0 = t * 4 + kk;
      // A[i][k] = Linv[k][w*16 + i]
      x0 = mfma_f64(li_f[t], tmp[k0][c], x0);
      x1 = mfma_f64(li_f[t + 1], tmp[k0 + 4][c], x1);
    }
    x0 += x1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + kk + 4 * r;
      Y[(size_t)row * ldy + c] = x0[r];
      if (row < out_rows) out[(size_t)row * m + c0 + c] = (float)x0[r];
    }
    __syncthreads();
  }
}

}  // namespace

// The walker is the one workgroup that waits for higher-numbered ones (the tiles next to the diagonal), so it needs the
// 1 + nb (nb - 1) / 2 factor workgroups of ITS launch resident; capped so that two launches from two handles / streams fit the
// 256 CUs side by side (2 x 121 at nb = 16: d <= 1024; larger systems take the launch chain).  The W workgroups behind them wait
// only for lower-numbered ones and may start late.
constexpr int POTRF_LA_MAX_NB = 16;

// the persistent launch is taken for this system, and `*own` workgroups are its own
static bool potrf_la_taken(const uce_ctx* h, int n, int nsplit, int* own) {
  const int nb = n / 64;
  if (!(nsplit == 1 && nb >= 3 && nb <= POTRF_LA_MAX_NB && h->sw.potrf_variant != 0 && h->la_flags)) return false;
  const bool with_inverse = h->sw.potrf_variant == 1 && h->Wi != nullptr;
  *own = 1 + nb * (nb - 1) / 2 * (with_inverse ? 2 : 1);
  return true;
}
// (h->sw.potrf_rider_cus, default 250: workgroups the launch may place at once - one per CU: this kernel's LDS -, a few CUs spare)
constexpr int POTRF_LA_MIN_RIDERS = 64;

// uce_edit asks before it hands the launch its rider jobs (the split of W_old, the Bt half of the Gram)
bool potrf_la_has_room(const uce_ctx* h, int n) {
  int own = 0;
  return potrf_la_taken(h, n, 1, &own) && h->sw.potrf_rider_cus - own >= POTRF_LA_MIN_RIDERS;
}

size_t potrf_la_smem() { return POTRF_LA_SMEM; }

bool potrf_la_job(uce_ctx* h, double* M, int n, int n_valid, PotrfLaJob* job, int* own) {
  if (!potrf_la_taken(h, n, 1, own)) return false;
  if (n_valid <= 0 || n_valid > n) n_valid = n;
  const bool with_inverse = h->sw.potrf_variant == 1 && h->Wi != nullptr;
  *job = PotrfLaJob{M, n, n / 64, n_valid, h->Lmat, h->Linv, h->status, h->la_flags, with_inverse ? h->Wi : nullptr, H2SplitJob{},
                    GramPrimalArgs{}};
  h->wi_valid = with_inverse;
  return true;
}

int launch_potrf_slabs(uce_ctx* h, double* M, int n, int nsplit, size_t slab_stride, hipStream_t st, int n_valid) {
  if (n_valid <= 0 || n_valid > n) n_valid = n;
  const int nb = n / 64;
  int own = 0;
  if (potrf_la_taken(h, n, nsplit, &own)) {
    // one persistent launch with look-ahead (k_potrf_la)
    static PerDeviceOnce la_once;
    if (const int tok = la_once.first()) {
      UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_potrf_la, hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTRF_LA_SMEM));
      la_once.commit(tok);
    }
    // variant 1 (default): the launch also forms the off-diagonal blocks of L^-1 (h->Wi) for the GEMM-shaped solve that
    // follows (uce_trinv.hip skips its merge launches); variant 2: factor only
    const bool with_inverse = h->sw.potrf_variant == 1 && h->Wi != nullptr;
    // uce_edit's primal path leaves the W_old of the dense apply here: the factorisation occupies `own` CUs for its whole
    // latency chain (d = 768: 133 for ~220 us), the rest of the chip streams the f16 split meanwhile (one workgroup per
    // CU: this kernel's LDS).  Fewer than 64 free CUs: the apply launches its own split pass.
    H2SplitJob sp{};
    GramPrimalArgs bt{};
    const int room = h->sw.potrf_rider_cus - own;
    if (room >= POTRF_LA_MIN_RIDERS) {
      if (h->h2_pending_src) {
        unsigned short* Ap;
        float *rs, *cb;
        const int rc = apply_h2_workspace(h, h->h2_pending_rows, h->h2_pending_d, &Ap, &rs, &cb);
        if (rc) return rc;
        sp = H2SplitJob{h->h2_pending_src, Ap, Ap + (size_t)h->h2_pending_rows * h->h2_pending_d, rs, h->h2_pending_rows, h->h2_pending_d, 0};
      }
      if (h->bt_pending.C) bt = h->bt_pending;
      if (sp.src || bt.C) sp.blocks = room;
    }
    const PotrfLaJob job{M, n, nb, n_valid, h->Lmat, h->Linv, h->status, h->la_flags, with_inverse ? h->Wi : nullptr, sp, bt};
    hipLaunchKernelGGL(k_potrf_la, dim3(own + sp.blocks), dim3(512), POTRF_LA_SMEM, st, job);
    UCE_LAUNCH_CHECK();
    h->wi_valid = with_inverse;
    if (sp.src) {
      h->h2_done_src = sp.src;
      h->h2_done_rows = sp.rows;
      h->h2_done_d = sp.d;
    }
    if (bt.C) h->bt_pending.C = nullptr;                           // taken (uce_edit launches Bt itself when this is still set)
    h->h2_pending_src = nullptr;
    return UCE_OK;
  }
  h->wi_valid = false;
  const size_t smem = 3 * 64 * LD * sizeof(double);
  const size_t smem_first = sizeof(Potrf64Scratch);
  static_assert(sizeof(Potrf64Scratch) <= POTRF_STEP_SMEM, "scratch must fit the three tile regions");
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_potrf_step, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_potrf_first, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem_first));
    attr_once.commit(tok);
  }
  hipLaunchKernelGGL(k_potrf_first, dim3(1), dim3(512), smem_first, st, (const double*)M, n, nsplit,
                     slab_stride, h->Lmat, h->Linv, h->status, n_valid);
  UCE_LAUNCH_CHECK();
  for (int j = 0; j + 1 < nb; ++j) {
    const int mt = nb - j - 1;
    const int tiles = mt * (mt + 1) / 2;
    hipLaunchKernelGGL(k_potrf_step, dim3(tiles), dim3(512), smem, st, M, n, j, h->Lmat, h->Linv,
                       h->status, n_valid);
    UCE_LAUNCH_CHECK();
  }
  return UCE_OK;
}

int launch_potrf(uce_ctx* h, double* M, int n, hipStream_t st, int n_valid) {
  return launch_potrf_slabs(h, M, n, 1, 0, st, n_valid);
}

int launch_trisolve(uce_ctx* h, int n, int m, const double* rhs64, const float* rhs32, int rhs_rows,
                    float* out, int out_rows, hipStream_t st, double* scratch) {
  // UCE_TRISOLVE_VARIANT=0 (read at uce_create) keeps the substitution kernel at every size (A/B measurements)
  if (scratch && n >= 192 && m % 64 == 0)
    return launch_trisolve_inv(h, n, m, rhs64, rhs32, rhs_rows, out, out_rows, scratch, st);
  const bool use_lds = n <= 1024;
  const size_t smem = 64 * 16 * 8 + (use_lds ? (size_t)n * 16 * 8 : 0);
  double* Yg = use_lds ? nullptr : h->Yg;
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    const int cap = 64 * 16 * 8 + 1024 * 16 * 8;
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trisolve<true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_trisolve<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    attr_once.commit(tok);
  }
  const dim3 grid(m / 16);
  if (rhs32)
    hipLaunchKernelGGL(k_trisolve<true>, grid, dim3(256), smem, st, (const double*)h->Lmat,
                       (const double*)h->Linv, n, m, rhs64, rhs32, rhs_rows, out, out_rows, Yg);
  else
    hipLaunchKernelGGL(k_trisolve<false>, grid, dim3(256), smem, st, (const double*)h->Lmat,
                       (const double*)h->Linv, n, m, rhs64, rhs32, rhs_rows, out, out_rows, Yg);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
