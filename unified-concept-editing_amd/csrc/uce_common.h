// Shared declarations for libuce_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/uce_hip.h"

#define UCE_HIP_TRY(expr)                                   \
  do {                                                      \
    hipError_t e_ = (expr);                                 \
    if (e_ != hipSuccess) return UCE_EHIP - (int)e_;        \
  } while (0)

#define UCE_LAUNCH_CHECK()                                  \
  do {                                                      \
    hipError_t e_ = hipGetLastError();                      \
    if (e_ != hipSuccess) return UCE_EHIP - (int)e_;        \
  } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef short short8_t __attribute__((ext_vector_type(8)));

// Function attributes (the > 64 KB dynamic-LDS opt-in) are per DEVICE: a launcher sets them the first time it runs
// on each device of the process, not once per process.  first() returns 0 when the current device is done, else a
// token (device ordinal + 1) that the launcher hands to commit() AFTER every hipFuncSetAttribute call succeeded - a
// failed call is retried by the next launch instead of being remembered as done.  (The entry points of uce_api.hip
// make the handle's device current before any launcher runs, so "current device" = the device of the launch.)
struct PerDeviceOnce {
  unsigned long long seen[4] = {0, 0, 0, 0};      // up to 256 device ordinals
  int first() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) return -1;     // unknown: set the attributes, remember nothing
    return (__atomic_load_n(&seen[dev >> 6], __ATOMIC_ACQUIRE) & (1ull << (dev & 63))) ? 0 : dev + 1;
  }
  void commit(int token) {
    if (token > 0) __atomic_fetch_or(&seen[(token - 1) >> 6], 1ull << ((token - 1) & 63), __ATOMIC_RELEASE);
  }
};

// Every launching entry point runs with the HANDLE's device current (function attributes, allocations and launches
// then all refer to the device the handle was created on, whatever the caller's current device is); restored on exit.
struct UceDeviceGuard {
  int prev = -1;
  explicit UceDeviceGuard(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur;
  }
  ~UceDeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  UceDeviceGuard(const UceDeviceGuard&) = delete;
  UceDeviceGuard& operator=(const UceDeviceGuard&) = delete;
};
#define UCE_ENTER(h) UceDeviceGuard uce_guard_((h)->device)

constexpr int UCE_NB = 64;  // block size of the f64 Cholesky / triangular solves

// A/B switches for measurements (defaults = the measured best), read from the environment ONCE, when the handle is created:
//   UCE_XATTN_VARIANT   1 (default): column-group kernel at generation-batch sizes | 0: always k_xattn | 2: the group kernel
//                       at every size | 3: its 8-wave dh = 40 form
//   UCE_SPLIT_MAX_NE / UCE_SPLIT_MAX_N  uce_edit takes the project + update form up to this many edit concepts / concepts in all (beyond: Delta + dense apply)
//   UCE_PROJECT_LA      1: N > 128 with <= 128 edit concepts: the persistent Cholesky inside the projection launch | 0: in front of it
//   UCE_POTRF_RIDER_CUS workgroups (CUs) the persistent Cholesky launch may occupy with its riders included (default 250; 0: no riders)
//   UCE_RIDER_MAX_N     largest dual system (64 or 128) factored by rider blocks of the projection launch; 0: never
//   UCE_SATTN_QT        0: self-attention kernel by measured rule | 1: k_sattn, one query tile per wave | 2: two query tiles
//                       wherever dh <= 48 | 3: the software-pipelined k_sattn_p wherever it exists
//   UCE_SATTN_VTI       0: V^T of the self-attention transposed on the way into LDS up to 1024 keys, by the k_vt pre-pass beyond |
//                       1: always inline (2-byte transposing stores) | 2: always the pre-pass | 3: always inline (k_sattn_h: row-major V +
//                       ds_read_b64_tr_b16, also what 0 takes there)
//   UCE_CONV_W1         one-wave-per-SIMD convolution (uce_conv_w1.hip: 4 waves, 128 x 160 / 128 x 128 wave tiles, accumulators pinned in
//                       AGPRs): 1 (default) = where a layer gives every CU a 256-pixel tile | 2 = wherever the shape allows | 0 = off
//   UCE_CONV_TILE       0: tile of the direct-to-LDS convolution by rule | 1000 * BM + BN: forced
//   UCE_GEMM_TILE       0: tile of uce_linear_fwd by rule | 1000 * BM + BN (256320, 256256, 128320, 128256, 256128): forced
//   UCE_GN_FUSED        1: GroupNorm of small activations in ONE launch (grid-wide wait inside a sample) | 0: always stats + apply
//   UCE_SK_SPLIT        0: slabs per tile of the few-tile GEMM / convolution forms by rule (uce_splitk.h) | S: forced
//   UCE_EDIT_RESIDENT   1 (default): d = 768, N <= 64: the ONE-launch register-resident edit (uce_edit_resident.hip) | 0: projection + update launches
//   UCE_POTRF_VARIANT   1: one persistent look-ahead launch for systems of 3..16 diagonal blocks that also forms L^-1 |
//                       2: the same launch, factor only (L^-1 by the merge launches of uce_trinv.hip) | 0: the launch chain
struct UceSwitches {
  int xattn_variant, rider_max_n, potrf_variant, sattn_qt, potrf_rider_cus, split_max_ne, split_max_n, project_la,
      gemm_tile, sattn_vti, conv_tile, conv_w1, sk_split, gn_fused, edit_resident;
};

// Workspace owned by a handle.  Everything is sized by (d_cap, n_cap): the largest embedding
// width and the largest SPD system (primal: n = d, dual: n = roundup(N, 64)) seen so far.
// arguments of one primal Gram tile job (uce_gram_tile.h)
struct GramPrimalArgs {
  const float* C;       // [N, d]
  const float* G;       // [N_edit, d]
  const float* s;       // [N]
  int N, N_edit, d;
  float lamb;
  double* outA;         // [d, d] (+ split * slab_stride)
  double* outBt;
  int kchunk;           // concepts per split
  size_t slab_stride;
};

// W_old [rows, d] -> planes hi / lo [rows, d] f16 + inv [rows] (2^-e per row); `blocks` rider workgroups (0: none)
struct H2SplitJob {
  const float* src;
  unsigned short* hi;
  unsigned short* lo;
  float* inv;
  long rows;
  int d;
  int blocks;
};

// job of the persistent Cholesky launch (uce_potrf_la.h)
struct PotrfLaJob {
  double* M;          // [n, n] system (lower tiles read; the tiles (i, i-1) and (i, i) are overwritten with their updates)
  int n, nb, n_valid;
  double* Lmat;       // [n, n]: off-diagonal blocks of L
  double* Linv;       // [nb][64][64]
  int* status;
  unsigned* flags;    // [nb * nb] panel (i, j) published | [nb] L_kk^-1 published | [nb] tiles (k, k-1), (k, k) handed over | [1] exits
                      // | [nb * nb] block (i, k) of L^-1 published
  double* Wi;         // [n, n]: off-diagonal blocks of L^-1 (null: not wanted)
  H2SplitJob sp;      // sp.blocks rider workgroups behind the factorisation's own: the f16 split of W_old for the dense apply that
                      // follows the solve (uce_apply_h2.hip), streamed on the CUs the factorisation leaves idle
  GramPrimalArgs bt;  // bt.C != null: the same riders then compute the d/64 x d/64 tiles of Bt = C_e^T S_e (G - C_e), the right-hand
                      // side of the solve that follows - nothing in this launch reads it
};

struct uce_ctx {
  int device;
  UceSwitches sw;
  int d_cap;      // embedding width capacity
  int n_cap;      // SPD system capacity (multiple of 64)
  double* M;      // [n_cap, n_cap]  system matrix, trailing tiles updated in place
  double* Lmat;   // [n_cap, n_cap]  Cholesky factor (lower blocks)
  double* Linv;   // [n_cap/64, 64, 64] inverses of the diagonal blocks of L
  double* slabs;  // split-K partial sums of the Gram kernels
  size_t slabs_bytes;
  double* Bt;     // [d_cap, d_cap] right-hand side of the primal solve
  double* Yg;     // [n_cap, d_cap] intermediate of the triangular solves (Y = L^-1 RHS)
  double* Wi;     // [n_cap, n_cap] explicit L^-1 of the GEMM-shaped triangular solves (uce_trinv.hip)
  bool wi_valid;  // the last factorisation on this handle already formed the off-diagonal blocks of L^-1 in Wi (k_potrf_la)
  float* DeltaT;  // [d_cap, d_cap]
  unsigned short* DeltaP;  // [2][d_cap, d_cap] f16 planes (high, low) of (I + Delta)^T (uce_apply_h2.hip)
  float* Dm;      // [n_cap, d_cap]
  float* R;       // [n_cap, d_cap]
  int* status;    // device word: 0 or (1-based) index of the first non-positive pivot
  unsigned* la_flags;   // hand-off flags of the persistent Cholesky (uce_solve.hip: k_potrf_la), zero between launches
  unsigned* ticket;  // hand-off words of the rider blocks (uce_lowrank2.hip), all zero between launches: [0] arrival counter
                     // of the Gram riders, [1] stage word of the factorising block, [2] completion counter of the solve riders,
                     // [3] main workgroups of a one-launch edit past the wait for R, [4] its D-prep riders done (8 words allocated)
  unsigned char* res_ws;   // fragment planes + scales of the register-resident edit launch (uce_edit_resident.hip), allocated at uce_create
  float* T;       // [rows_cap, nep_cap] projection W_old D_e^T of the two-kernel low-rank apply; the dense f16 apply keeps the
                  // planes of W_old and the scales here (uce_apply_h2.hip: apply_h2_workspace)
  size_t T_elems;
  // uce_edit, primal path: W_old waiting to be split into f16 planes by rider workgroups of the Cholesky launch
  // (h2_pending_*: set before uce_solve_delta, consumed by launch_potrf_slabs), and the W_old whose planes are in h->T
  // (h2_done_*: set by that launch, consumed by launch_apply_h2)
  const float* h2_pending_src;
  long h2_pending_rows;
  int h2_pending_d;
  const float* h2_done_src;
  long h2_done_rows;
  int h2_done_d;
  // ... and the Bt half of the primal Gram (the right-hand side: not needed before the solve), for the same riders.
  // bt_pending.C != null after uce_solve_delta's factorisation: nobody took it, uce_edit launches it itself.
  GramPrimalArgs bt_pending;
  void* Vt;       // V^T scratch of uce_sattn_fwd ([B, H, DVP, LkP] 16-bit elements)
  size_t Vt_elems;
  // split-contraction scratch of the few-tile GEMM / convolution forms (uce_splitk.h): slabs + per-tile tickets (zero between launches).
  // One launch at a time may use it: launches of one handle are ordered by the caller's stream (one handle per thread / stream).
  // one-launch GroupNorm of small activations (uce_norm.hip: k_gn_fused): partial sums [<= 1024 workgroups][64 groups][2] and the
  // arrive / depart counters of up to 1024 samples (zero between launches); allocated at uce_create
  float* gn_partial;
  unsigned* gn_counters;
  int gn_fused_cap;      // workgroups of k_gn_fused resident at once on this device (0: not asked yet; uce_norm.hip: gn_fused_capacity)
  float* sk_ws;
  size_t sk_bytes;
  unsigned* sk_tick;
  size_t sk_tiles;
  void* retired[32];   // outgrown Vt / split-contraction buffers: kept alive until uce_destroy (captured hipGraphs may still name them)
  int n_retired;
  struct uce_prof* prof;   // per-launch HIP-event brackets of uce_edit (uce_profile_begin / _end); null = off
};

// Measurement aid (uce_profile_begin / uce_profile_end): when h->prof is set, every kernel launch (or launch
// chain) uce_edit issues is bracketed by two HIP events on the caller's stream.
void uce_prof_mark(uce_ctx* h, const char* name, hipStream_t st, bool begin);
// UCE_ROCTX=1: every such scope is also a roctx range (roctxRangePushA / roctxRangePop from libroctx64.so, resolved at the first
// scope; rocprofv3 --marker-trace shows the launches of the edit / attention entry points under their names); off = one load.
void uce_roctx(const char* name, bool begin);
extern int g_uce_roctx;      // -1: not looked at yet, 0: off, 1: on
struct UceProfScope {
  uce_ctx* h; const char* name; hipStream_t st;
  UceProfScope(uce_ctx* h_, const char* n, hipStream_t s) : h(h_), name(n), st(s) {
    if (g_uce_roctx) uce_roctx(name, true);
    if (h && h->prof) uce_prof_mark(h, name, st, true);
  }
  ~UceProfScope() {
    if (h && h->prof) uce_prof_mark(h, name, st, false);
    if (g_uce_roctx) uce_roctx(name, false);
  }
};

// k-tile order of the implicit-GEMM convolutions: CHUNK-major - the nine taps of a 64-channel chunk back to back - instead of the
// weight's own tap-major order.  A tap's pixel segments are shared with its neighbours; tap-major brings a segment back cch k-tiles
// later (Cin = 320: 164 KB of A per CU, 5 MB per XCD in between - more than its 4 MB L2), chunk-major one k-tile later (33 KB).
// Measured on an MI355X at 128 prompts per call (tools/probe_r04.py conv, profiles/r05/conv_tap_order_b128_*.jsonl; us, tap-major |
// chunk-major, k_conv3x3_w1): 320 -> 320 @ 64^2 838 | 802, 960 -> 320 @ 64^2 2316 | 2092, 640 -> 320 @ 64^2 1507 | 1454, 1920 -> 640 @ 32^2
// 2123 | 2011, 1280 -> 640 @ 32^2 1414 | 1374; alone a few small-activation layers are level or behind (320 -> 640 @ 32^2 370 | 388, 2560 ->
// 1280 @ 16^2 1323 | 1343), inside the generation loop chunk-major EVERYWHERE is the fastest: same box, 128 images at 128 prompts per
// call, 8.73 / 8.73 images/s tap-major, 8.86 / 8.85 chunk-major from 1 M source elements per image, 9.05 always (tools/ab_gen.sh).
extern int g_uce_conv_tapin;      // UCE_CONV_TAPIN (measurements, read at uce_create): -1 / 1 = chunk-major (default), 0 = tap-major
inline int conv_tap_inner(int H, int W, int Cin, int up, int sd) {
  (void)H; (void)W; (void)Cin; (void)up; (void)sd;
  return g_uce_conv_tapin == 0 ? 0 : 1;
}

// ---- internal launchers (defined across the .hip files) -------------------------------------
int uce_ensure(uce_ctx* h, int d, int n);
// uce_conv_dma.hip: 1 = launched (*rc = status), 0 = shape not taken by the direct-to-LDS form
// (the caller decides by UceSwitches::conv_dma whether to ask)
int launch_conv_w1(const void* x, const void* w, const void* bias, void* y, long M, int H, int W, int Cin, int Cout, int up, int dtype,
                   hipStream_t st, int* rc, int sd, const void* res, int mode);
int launch_conv_dma(const void* x, const void* w, const void* bias, void* y, long M, int H, int W, int Cin, int Cout, int up,
                    int dtype, hipStream_t st, int* rc, int sd = 1, const void* res = nullptr, int force = 0, int wide = 1,
                    uce_ctx* h = nullptr);

int launch_gram_primal(uce_ctx* h, const float* C, const float* G, const float* s, int N, int N_edit,
                       int d, float lamb, double* A, double* Bt, hipStream_t st, int which = 0);
// Also writes Dm = G - C_e and resets h->status.  When the system is a single 64-block and the
// Gram was split over K, the slabs are left unreduced: *nsplit_out > 1 and the matrix to factor
// is h->slabs with *slab_stride_out (k_potrf_first sums them).
int launch_gram_dual(uce_ctx* h, const float* C, const float* s, int N, int d, float lamb, double* K,
                     int n_pad, const float* G, float* Dm, int N_edit, int* nsplit_out,
                     size_t* slab_stride_out, hipStream_t st);
// n_valid (0 = n): rows / columns >= n_valid are the identity padding of the system (their pivots are skipped).
int launch_potrf_slabs(uce_ctx* h, double* M, int n, int nsplit, size_t slab_stride, hipStream_t st, int n_valid = 0);
int launch_potrf(uce_ctx* h, double* M, int n, hipStream_t st, int n_valid = 0);
bool potrf_la_has_room(const uce_ctx* h, int n);
// the job of the persistent launch for the n x n system M, if that launch is the form the system takes (else false): for a
// caller that hosts the factorisation in a launch of its own (launch_lr_project_la); *own = its workgroups
bool potrf_la_job(uce_ctx* h, double* M, int n, int n_valid, PotrfLaJob* job, int* own);
size_t potrf_la_smem();   // the persistent Cholesky launch will run for an n x n system and has CUs for riders
// X = M^-1 RHS after launch_potrf.  RHS is f64 [n, m] (rhs64) or f32 [rhs_rows, m] (rhs32, rows
// beyond rhs_rows are zero).  out f32 [out_rows, m] gets rows 0..out_rows-1 of X.
// `scratch` [n, n] f64 (optional): the factored matrix, dead after launch_potrf - with it, systems of >= 3 diagonal
// blocks take the GEMM-shaped path (explicit L^-1 by recursive doubling, uce_trinv.hip).
int launch_trisolve(uce_ctx* h, int n, int m, const double* rhs64, const float* rhs32, int rhs_rows,
                    float* out, int out_rows, hipStream_t st, double* scratch = nullptr);
int launch_trisolve_inv(uce_ctx* h, int n, int m, const double* rhs64, const float* rhs32, int rhs_rows, float* out,
                        int out_rows, double* scratch, hipStream_t st);
// f16 x 2 dense apply (uce_apply_h2.hip): any row count (slabs beyond a 2 GB buffer descriptor are walked in row chunks).
// apply_h2_fits: the whole slab is ONE chunk (only then can rider workgroups pre-split it)
bool apply_h2_fits(long rows, int d);
int apply_h2_workspace(uce_ctx* h, long rows, int d, unsigned short** Ap, float** rs, float** cb);
int launch_apply_h2(uce_ctx* h, const float* W_old, const float* DeltaT, float* W_new, long rows, int d, hipStream_t st);
int launch_apply_lowrank(const float* W_old, const float* Dm, const float* R, float* W_new, long rows,
                         int d, int N_edit, hipStream_t st);
bool apply_lowrank_fits(int d, int N_edit);
int launch_delta_from_factors(const float* Dm, const float* R, int N_edit, int d, float* DeltaT,
                              hipStream_t st);
int launch_sub_rows(const float* G, const float* C, float* Dm, long n, hipStream_t st);
bool lowrank_split_supported(int d, int N_edit);
int lr_rider_cap();     // largest dual system (rows, multiple of 64) the projection launch's riders can factor
// With `h`: the launch also carries the whole small-system chain (Gram + Cholesky riders, then the solve riders
// that write R [N_edit, d] = rows of K^-1 C) - the caller needs no separate triangular-solve launch.
int launch_lr_project(const float* W_old, const float* X, const float* Csub, float* T, long rows, int d,
                      int N_edit, hipStream_t st, uce_ctx* h = nullptr, const float* C = nullptr,
                      const float* s = nullptr, int N = 0, float lamb = 0.f, float* R = nullptr);
int launch_lr_update(const float* W_old, const float* T, const float* R, float* W_new, long rows, int d,
                     int N_edit, hipStream_t st);
// uce_edit_resident.hip: the whole step in one launch, W_old held in registers between its two products
bool lr_resident_supported(int d, int N, int N_edit, long rows);
size_t lr_resident_ws_bytes();
int launch_lr_resident(uce_ctx* h, const float* W_old, const float* G, const float* C, const float* s, float* W_new, long rows,
                       int N, int N_edit, float lamb, unsigned char* ws, hipStream_t st);
// the projection with the persistent Cholesky of the dual system (la, own workgroups) in the first workgroups of the launch
int launch_lr_project_la(const float* W_old, const float* X, const float* Csub, float* T, long rows, int d, int N_edit,
                         const PotrfLaJob& la, int own, hipStream_t st);
int uce_ensure_T(uce_ctx* h, long rows, int N_edit);
int uce_ensure_T_floats(uce_ctx* h, size_t need);   // h->T holds >= need floats
int uce_ensure_Vt(uce_ctx* h, size_t elems);
// handle-owned scratch of the split-contraction forms: `bytes` of slabs and `tiles` tickets (zeroed when allocated; the kernels
// leave them zero).  Grows geometrically; outgrown buffers are retired, not freed (captured hipGraphs may still name them).
int uce_ensure_sk(uce_ctx* h, size_t bytes, size_t tiles);
size_t sattn_vt_elems(int B, int H, int Lk, int dh);
int launch_sattn(const void* q, const void* k, const void* v, void* vt, void* o, int B, int H, int Lq, int Lk, int dh,
                 float scale, int dtype, hipStream_t st, int qt_variant = 0, long ld = 0, int vti = 0, float lazy = 8.f,
                 bool exp2q = false);
int launch_xattn(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk,
                 int dh, float scale, int dtype, hipStream_t st, int variant = 1);
// all keys resident (Lk <= 128), row stride ld of q / k / v: the short self-attention layers
int launch_xattn_short_self(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int dh, float scale,
                            int dtype, hipStream_t st, long ld);

static __device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
