// Two-kernel form of the low-rank apply  W_new = W_old + (W_old D_e^T) R_e :
//
//   k_lr_project<D,MT>  : T [rows, NEP] = W_old D_e^T     needs only D_e = G - C_e, not the solve: in uce_edit ONE
//                         launch carries this GEMM and, in its first 4 / 12 "rider" blocks, the Gram matrix and the
//                         (blocked) Cholesky + block inverses of the dual system (N <= 64 / N <= 128), so the
//                         latency-bound factorisation hides under the f32-MFMA-bound projection without any
//                         cross-stream event (a HIP event hand-off between two streams measured 13-14 us each way).
//   k_lr_update_s<D,..> : W_new = W_old + T R_e           one pass over the weights per 128 edit concepts (129..256: a second
//                         pass in place): algorithmic bytes 8*rows*d (+ the small T and R).
//
// Splitting the update at T costs 2*4*rows*NEP bytes of extra traffic (6.4 MB at N_edit <= 64 for SD-1.4, 4 %) and
// buys (a) overlap of the projection with the latency-bound small-system chain, (b) an update kernel whose LDS
// footprint is tiny, so several workgroups per CU stream W with their phases naturally interleaved.
#include "uce_lowrank_riders.h"

namespace {

template <int D, int MT, int CT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lr_project(
    const float* __restrict__ W_old, const float* __restrict__ Dm, const float* __restrict__ Csub,
    float* __restrict__ T, long rows, int Ne, int NEP, GramPotrfJob job) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_gram = job.C ? gp_riders(job.nb) : 0;
  const int has_rider = job.C ? lr_rider_blocks(job.nb, D) : 0;
  if ((int)blockIdx.x < n_gram) {
    gram_potrf_rider<D>(job, smem_raw);
    return;
  }
  if ((int)blockIdx.x < has_rider) {
    solve_rider<D>(job, smem_raw, (int)blockIdx.x - 1);   // column blocks 0 .. n_gram - 2 belong to the Gram riders
    return;
  }
  DBG(0);
  float* Wc = (float*)smem_raw;                       // [2][MT*16][PJ_LD]
  project_dispatch<D, MT, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, has_rider);
  DBG(1);
}

// ---------------------------------------------------------------------------------------------
// update, single-buffered R, buffer addressing.
//  * The register holding k-step t's R fragment is reloaded IN PLACE with the next column group's
//    fragment right after the MFMAs of step t have consumed it.  Program order of the vector-memory
//    stream per group g is   W prefetch(g+1) | R(g+1)[0..NK-1] (between the MFMA steps) | stores(g),
//    so the wait for R(g+1)[t] at step t of group g+1 covers only loads issued a whole MFMA loop
//    earlier: the in-order vmcnt never makes a step wait for a load issued during the current group.
//    Half the registers of a double-buffered fragment set (the predecessor of this kernel: 34.6 us vs 30 us).
//  * Every stream is a buffer load/store: ONE 32-bit lane offset for all of them, the per-step /
//    per-row / per-group displacement folded into the (scalar) resource base, and the resource's
//    num_records doing the bounds work: concept rows >= N_edit read as 0 and weight rows >= rows are
//    neither read nor written - no clamped 64-bit address pairs (2 VGPRs per stream with flat
//    addressing), no per-row branches.  (The scalar offset operand is not part of the hardware range
//    check, hence base shifting instead of soffset.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0 ? bytes : 0, 0x00020000);
}

template <int D, int UP_MT, int WPE, int NK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lr_update_s(
    const float* W_old, const float* __restrict__ T, const float* __restrict__ R,
    float* W_new, long rows, int Ne, int NEP, int t_stride) {
  // (W_old may be W_new: the second pass of a 129..256-concept update adds onto the first pass's output in place - a
  //  workgroup reads only the 16 rows it writes, and every element is read before it is written)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int d = D;
  constexpr int SR = UP_MT * 16;
  const int tld = NEP + 2;
  float* Ts = (float*)smem_raw;                       // [SR][tld]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const long R0 = (long)blockIdx.x * SR;
  constexpr int MG = D / 256;

  const float* Wb = W_old + R0 * d;                   // workgroup-uniform bases
  float* Ob = W_new + R0 * d;
  const int w_bytes = (int)((rows - R0) < SR ? (rows - R0) : SR) * d * 4;   // this tile's valid weight bytes
  const int r_bytes = Ne * d * 4;
  // the one lane offset (bytes): R row lk / W row 4*lk of the step's / tile's base, columns w*64 + 4*li
  const unsigned vo_r = (unsigned)((lk * d + w * 64 + 4 * li) * 4);
  const unsigned vo_w = (unsigned)((4 * lk * d + w * 64 + 4 * li) * 4);

  auto ld_r = [&](int t, int gi) -> float4_t {
    const int sh = (4 * t * d + gi * 256) * 4;
    return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(R + 4 * t * d + gi * 256, r_bytes - sh), vo_r, 0, 0));
  };
  auto ld_w = [&](int m, int r, int gi) -> float4_t {
    const int sh = ((m * 16 + r) * d + gi * 256) * 4;
    return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
        make_rsrc(Wb + (m * 16 + r) * d + gi * 256, w_bytes - sh), vo_w, 0, 0));
  };

  float4_t rr[NK];
  float4_t res[UP_MT][4];
#pragma unroll
  for (int t = 0; t < NK; ++t) rr[t] = ld_r(t, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < UP_MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) res[m][r] = ld_w(m, r, 0);
  __builtin_amdgcn_sched_barrier(0);
  {
    const int f4_row = NEP >> 2;
    for (int e = tid; e < SR * f4_row; e += 256) {
      const int r = e / f4_row, c = (e - r * f4_row) << 2;
      long gr = R0 + r;
      gr = gr < rows ? gr : rows - 1;
      const float4_t v = *(const float4_t*)(T + gr * t_stride + c);
      Ts[r * tld + c] = c < Ne ? v[0] : 0.f;          // pad columns -> 0: the k loop reads unconditionally
      Ts[r * tld + c + 1] = c + 1 < Ne ? v[1] : 0.f;
      Ts[r * tld + c + 2] = c + 2 < Ne ? v[2] : 0.f;
      Ts[r * tld + c + 3] = c + 3 < Ne ? v[3] : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int gi = 0; gi < MG; ++gi) {
    float4_t acc[UP_MT][4];                           // acc[m][q][r]: row m*16 + 4*lk + r, column 4*li + q
#pragma unroll
    for (int m = 0; m < UP_MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][q][r] = res[m][r][q];
    if (gi + 1 < MG) {
#pragma unroll
      for (int m = 0; m < UP_MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[m][r] = ld_w(m, r, gi + 1);
    }
    float a[UP_MT];                                   // T fragments, read from LDS one step ahead
#pragma unroll
    for (int m = 0; m < UP_MT; ++m) a[m] = Ts[(m * 16 + li) * tld + lk];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NK; ++t) {
      float an[UP_MT];
#pragma unroll
      for (int m = 0; m < UP_MT; ++m) an[m] = (t + 1 < NK) ? Ts[(m * 16 + li) * tld + 4 * (t + 1) + lk] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < UP_MT; ++m)
          acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], rr[t][q], acc[m][q], 0, 0, 0);
      if (gi + 1 < MG) rr[t] = ld_r(t, gi + 1);       // reload in place: next group's fragment for step t
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < UP_MT; ++m) a[m] = an[m];
    }
#pragma unroll
    for (int m = 0; m < UP_MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4_t o = {acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]};
        const int sh = ((m * 16 + r) * d + gi * 256) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o),
                                               make_rsrc(Ob + (m * 16 + r) * d + gi * 256, w_bytes - sh), vo_w, 0,
                                               2 /* nt */);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// rows / 16 tiles over 256 CUs with MT tiles per workgroup: time ~ ceil(workgroups / 256) * MT.  `extra`: the rider
// blocks of the launch - they hold a CU each while the chain runs, so a projection workgroup beyond 256 - extra would
// start late and stretch the launch.
int pick_mt2(long rows, int extra) {
  const long t16 = (rows + 15) / 16;
  int best = 8;
  long best_cost = -1;
  for (int mt = 8; mt >= 5; --mt) {
    const long wgs = (t16 + mt - 1) / mt + extra;
    const long cost = ((wgs + 255) / 256) * mt;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mt; }
  }
  return best;
}

// The projection with the persistent Cholesky of a dual system of 3 ... 16 diagonal blocks in its FIRST n_la workgroups
// (uce_potrf_la.h; 129 ... 1024 concepts with at most 128 of them edited: the factorisation needs nothing from the
// projection and the projection nothing from it - they used to be two launches in a row).  The factorisation's workgroups
// come first in the grid, so they are placed before any projection tile (its waits never depend on later workgroups).
template <int D, int MT, int CT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lr_project_la(
    const float* __restrict__ W_old, const float* __restrict__ Dm, const float* __restrict__ Csub,
    float* __restrict__ T, long rows, int Ne, int NEP, PotrfLaJob la, int n_la) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if ((int)blockIdx.x < n_la) {
    potrf_la_body(la, (int)blockIdx.x, n_la, smem_raw);
    return;
  }
  float* Wc = (float*)smem_raw;                       // [2][MT*16][PJ_LD]
  project_dispatch<D, MT, CT>(W_old, Dm, Csub, T, rows, Ne, NEP, Wc, n_la);
}

template <int D, int MT, int CT>
int launch_project_la(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit, int NEP64,
                      const PotrfLaJob& la, int own, hipStream_t st) {
  size_t smem = (size_t)2 * (MT * 16 + 64 * CT) * PJ_LD * sizeof(float);
  if (smem < potrf_la_smem()) smem = potrf_la_smem();
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    // (what it needs, not the 160 KB the plain projection asks for: the hosted factorisation has a static LDS word)
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_lr_project_la<D, MT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_once.commit(tok);
  }
  const long nwg = (rows + MT * 16 - 1) / (MT * 16) + own;
  hipLaunchKernelGGL((k_lr_project_la<D, MT, CT>), dim3((unsigned)nwg), dim3(512), smem, st, W_old, Dm, Csub, T, rows, N_edit,
                     NEP64, la, own);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int D, int CT>
int launch_project_la_d(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit, int NEP64,
                        const PotrfLaJob& la, int own, hipStream_t st) {
  // (one tile height per width keeps the instantiations of the hosted factorisation few: 7 x 16 rows tile SD-1.4's slab in 223)
  if constexpr (D == 2048) return launch_project_la<D, 5, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, la, own, st);
  else return launch_project_la<D, 7, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, la, own, st);
}

template <int D, int MT, int CT>
int launch_project(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit,
                   int NEP64, const GramPotrfJob& job, hipStream_t st) {
  size_t smem = (size_t)2 * (MT * 16 + 64 * CT) * PJ_LD * sizeof(float);
  if (job.C && smem < gp_smem(job.nb)) smem = gp_smem(job.nb);
  static PerDeviceOnce attr_once;   // hipFuncSetAttribute is per device
  if (const int tok = attr_once.first()) {
    UCE_HIP_TRY(hipFuncSetAttribute((const void*)k_lr_project<D, MT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
    attr_once.commit(tok);
  }
  const long nwg = (rows + MT * 16 - 1) / (MT * 16) + (job.C ? lr_rider_blocks(job.nb, D) : 0);
  hipLaunchKernelGGL((k_lr_project<D, MT, CT>), dim3((unsigned)nwg), dim3(512), smem, st, W_old, Dm, Csub, T, rows,
                     N_edit, NEP64, job);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int D, int CT>
int launch_project_d(const float* W_old, const float* Dm, const float* Csub, float* T, long rows, int N_edit,
                     int NEP64, const GramPotrfJob& job, hipStream_t st) {
  if constexpr (D == 2048) return launch_project<D, 5, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
  else switch (pick_mt2(rows, job.C ? lr_rider_blocks(job.nb, D) : 0)) {
    case 5: return launch_project<D, 5, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
    case 6: return launch_project<D, 6, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
    case 7: return launch_project<D, 7, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
    default: return launch_project<D, 8, CT>(W_old, Dm, Csub, T, rows, N_edit, NEP64, job, st);
  }
}

// one pass W_new = W_in + T[:, 0 .. NEPw) R[0 .. N_w): T has row stride t_stride floats, N_w <= 128 concepts
template <int D, int UP_MT, int WPE>
int launch_update_s(const float* W_in, const float* T, const float* R, float* W_new, long rows, int N_w,
                    int NEPw, int t_stride, hipStream_t st) {
  const size_t smem = (size_t)UP_MT * 16 * (NEPw + 2) * sizeof(float);
  const dim3 grid((unsigned)((rows + UP_MT * 16 - 1) / (UP_MT * 16))), block(256);
  const int nks = (N_w + 3) / 4;
  if (nks <= 8)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 8>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else if (nks <= 13)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 13>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else if (nks <= 16)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 16>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else if (nks <= 25)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 25>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else if (nks <= 32)
    hipLaunchKernelGGL((k_lr_update_s<D, UP_MT, WPE, 32>), grid, block, smem, st, W_in, T, R, W_new, rows, N_w, NEPw, t_stride);
  else
    return UCE_EINVAL;
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

template <int D>
int launch_update_d(const float* W_old, const float* T, const float* R, float* W_new, long rows, int N_edit,
                    int NEP64, hipStream_t st) {
  // 16-row tiles, 2 waves per SIMD: measured best at N_edit 16..128 on MI355X (32-row tiles: +4-9 %)
  // (3 / 4 waves per SIMD measured in round 3: SDXL slab 564 -> 610 / 537 us, 100 concepts 45 -> 45 / 78 us (spills): kept at 2)
  // (measured in round 3 on the SDXL slab and without effect beyond the 520-570 us run-to-run spread: 32-row tiles; weight
  // rows of two column groups in flight instead of one)
  if (N_edit <= 128) return launch_update_s<D, 1, 2>(W_old, T, R, W_new, rows, N_edit, NEP64, NEP64, st);
  // 129 .. 256 concepts: two passes of the register-resident kernel over 128-concept windows of T and R, the second one in
  // place on the first one's output (round 2's ring-buffered single-pass kernel for this range - R fragments through a
  // 4-deep register ring, in-order vmcnt stalls, spills at d = 2048 - measured 143 us against 117 at 256 concepts on the
  // SD-1.4 slab and 2093 against 1789 on the SDXL slab; it was 3-8 % ahead only just above 128 concepts)
  const int rc = launch_update_s<D, 1, 2>(W_old, T, R, W_new, rows, 128, 128, NEP64, st);
  if (rc) return rc;
  return launch_update_s<D, 1, 2>(W_new, T + 128, R + (size_t)128 * D, W_new, rows, N_edit - 128, NEP64 - 128, NEP64, st);
}

}  // namespace

int lr_rider_cap() { return 64 * GP_MAXB; }

bool lowrank_split_supported(int d, int N_edit) {
  return (d == 768 || d == 1024 || d == 2048) && N_edit >= 1 && N_edit <= 256;
}

// X = Dm with Csub == nullptr, or X = G with Csub = C_e (D_e = G - C_e formed on the fly).  With `h` (and N <= 128) the
// launch also builds and factors the dual system in its rider blocks and solves for R = rows of K^-1 C:
// K = lamb S^-1 + C C^T -> h->Linv (+ block (1, 0) of h->Lmat), h->status, R.  More than 64 edit concepts take the
// 128-concept batches (one pass over W per 128 concepts).
int launch_lr_project(const float* W_old, const float* X, const float* Csub, float* T, long rows, int d,
                      int N_edit, hipStream_t st, uce_ctx* h, const float* C, const float* s, int N, float lamb, float* R) {
  const int NEP64 = (N_edit + 63) / 64 * 64;
  GramPotrfJob job{};
  if (h) {
    const int nb = (N + 63) / 64;
    if (nb < 1 || nb > GP_MAXB || !R || !C || !s) return UCE_EINVAL;
    job = GramPotrfJob{C, s, N, lamb, h->slabs, h->ticket, h->Lmat, h->Linv, h->status, nb, R, N_edit, nullptr, NEP64, 0, 0};
  }
  const bool wide = NEP64 > 64;
  if (d == 768)
    return wide ? launch_project_d<768, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st)
                : launch_project_d<768, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st);
  if (d == 1024)
    return wide ? launch_project_d<1024, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st)
                : launch_project_d<1024, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st);
  if (d == 2048)
    return wide ? launch_project_d<2048, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st)
                : launch_project_d<2048, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, job, st);
  return UCE_EINVAL;
}

int launch_lr_project_la(const float* W_old, const float* X, const float* Csub, float* T, long rows, int d, int N_edit,
                         const PotrfLaJob& la, int own, hipStream_t st) {
  const int NEP64 = (N_edit + 63) / 64 * 64;
  if (N_edit < 1 || N_edit > 128) return UCE_EINVAL;
  const bool wide = NEP64 > 64;
  if (d == 768)
    return wide ? launch_project_la_d<768, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st)
                : launch_project_la_d<768, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st);
  if (d == 1024)
    return wide ? launch_project_la_d<1024, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st)
                : launch_project_la_d<1024, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st);
  if (d == 2048)
    return wide ? launch_project_la_d<2048, 2>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st)
                : launch_project_la_d<2048, 1>(W_old, X, Csub, T, rows, N_edit, NEP64, la, own, st);
  return UCE_EINVAL;
}

int launch_lr_update(const float* W_old, const float* T, const float* R, float* W_new, long rows, int d,
                     int N_edit, hipStream_t st) {
  const int NEP64 = (N_edit + 63) / 64 * 64;
  if (d == 768) return launch_update_d<768>(W_old, T, R, W_new, rows, N_edit, NEP64, st);
  if (d == 1024) return launch_update_d<1024>(W_old, T, R, W_new, rows, N_edit, NEP64, st);
  if (d == 2048) return launch_update_d<2048>(W_old, T, R, W_new, rows, N_edit, NEP64, st);
  return UCE_EINVAL;
}
