// C ABI of libuce_hip.so (include/uce_hip.h): handle lifetime, argument checks, and the
// orchestration of the edit pipeline (gram -> potrf chain -> trisolve -> apply) on one stream.
#include "uce_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <new>
#include <vector>

// ---- per-launch HIP-event brackets (uce_profile_begin / uce_profile_end) ----------------------
struct uce_prof {
  struct Rec { const char* name; hipEvent_t e0, e1; };
  std::vector<Rec> recs;
};

void uce_prof_mark(uce_ctx* h, const char* name, hipStream_t st, bool begin) {
  if (!h || !h->prof) return;
  if (begin) {
    uce_prof::Rec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, st);
    h->prof->recs.push_back(r);
  } else if (!h->prof->recs.empty() && h->prof->recs.back().name == name) {
    (void)hipEventRecord(h->prof->recs.back().e1, st);
  }
}

// roctx ranges (tracing): resolved lazily so that the library never links the profiler
int g_uce_roctx = -1;
void uce_roctx(const char* name, bool begin) {
  static int (*push)(const char*) = nullptr;
  static int (*pop)() = nullptr;
  if (g_uce_roctx < 0) {
    const char* e = getenv("UCE_ROCTX");
    int on = 0;
    if (e && *e && *e != '0') {
      void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
      if (lib) {
        push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
        pop = (int (*)())dlsym(lib, "roctxRangePop");
        on = push && pop;
      }
    }
    g_uce_roctx = on;
    if (!on) return;
  }
  if (!g_uce_roctx) return;
  if (begin) (void)push(name);
  else (void)pop();
}

static void prof_clear(uce_ctx* h) {
  if (!h->prof) return;
  for (auto& r : h->prof->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  delete h->prof;
  h->prof = nullptr;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

static void free_ws(uce_ctx* h) {
  void* ptrs[] = {h->M, h->Lmat, h->Linv, h->slabs, h->Bt, h->Yg, h->Wi, h->DeltaT, h->DeltaP, h->Dm, h->R};  // (T is separate)
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  h->M = h->Lmat = h->Linv = h->slabs = h->Bt = h->Yg = h->Wi = nullptr;
  h->DeltaT = h->Dm = h->R = nullptr;
  h->DeltaP = nullptr;
  h->slabs_bytes = 0;
  h->d_cap = h->n_cap = 0;
}

int uce_ensure(uce_ctx* h, int d, int n) {
  if (d <= h->d_cap && n <= h->n_cap) return UCE_OK;
  const int dc = d > h->d_cap ? d : h->d_cap;
  const int nc = n > h->n_cap ? n : h->n_cap;
  UCE_HIP_TRY(hipSetDevice(h->device));
  UCE_HIP_TRY(hipDeviceSynchronize());
  free_ws(h);
  const size_t nn = (size_t)nc * nc, dd = (size_t)dc * dc;
  // split-K slabs: primal needs nsplit*2*d*d (only for small d), dual nsplit*n*n with nsplit <= 256 tiles
  size_t slabs = 0;
  {
    // primal: tiles >= 128 (d >= 576) -> no slabs; otherwise nsplit <= ceil(256 / tiles)
    const int nb = dc / 64;
    const int tiles = nb * (nb + 1) / 2 + nb * nb;
    if (tiles < 128) slabs = (size_t)((256 + tiles - 1) / tiles) * 2 * dd * sizeof(double);
    // dual: for every n' <= nc, nsplit(n') * n'^2 <= max(256 * 64^2 * ..): bound by tiles(n')*split <= 256+tiles
    const int nbn = nc / 64;
    size_t worst = 0;
    for (int b = 1; b <= nbn; ++b) {
      const int t = b * (b + 1) / 2;
      const int sp = t >= 128 ? 1 : (256 + t - 1) / t;
      const size_t need = (size_t)sp * (size_t)(b * 64) * (b * 64) * sizeof(double);
      if (need > worst) worst = need;
    }
    if (worst > slabs) slabs = worst;
  }
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16);
  };
  alloc((void**)&h->M, nn * sizeof(double));
  alloc((void**)&h->Lmat, nn * sizeof(double));
  alloc((void**)&h->Linv, (size_t)(nc / 64) * 64 * 64 * sizeof(double));
  alloc((void**)&h->slabs, slabs);
  alloc((void**)&h->Bt, dd * sizeof(double));
  alloc((void**)&h->Yg, (size_t)nc * dc * sizeof(double));
  alloc((void**)&h->Wi, nn * sizeof(double));
  alloc((void**)&h->DeltaT, dd * sizeof(float));
  alloc((void**)&h->DeltaP, 2 * dd * sizeof(unsigned short));
  alloc((void**)&h->Dm, (size_t)nc * dc * sizeof(float));
  alloc((void**)&h->R, (size_t)nc * dc * sizeof(float));
  if (e != hipSuccess) {
    free_ws(h);
    return UCE_ENOMEM;
  }
  h->slabs_bytes = slabs;
  h->d_cap = dc;
  h->n_cap = nc;
  return UCE_OK;
}

int uce_ensure_T(uce_ctx* h, long rows, int N_edit) {
  return uce_ensure_T_floats(h, (size_t)rows * (size_t)((N_edit + 63) / 64 * 64));
}

int uce_ensure_T_floats(uce_ctx* h, size_t need) {
  if (need <= h->T_elems) return UCE_OK;
  UCE_HIP_TRY(hipSetDevice(h->device));
  UCE_HIP_TRY(hipDeviceSynchronize());
  if (h->T) (void)hipFree(h->T);
  h->T = nullptr;
  h->T_elems = 0;
  if (hipMalloc((void**)&h->T, need * sizeof(float)) != hipSuccess) return UCE_ENOMEM;
  h->T_elems = need;
  return UCE_OK;
}

int uce_ensure_Vt(uce_ctx* h, size_t elems) {
  if (elems <= h->Vt_elems) return UCE_OK;
  UCE_HIP_TRY(hipSetDevice(h->device));
  // grow geometrically; the old buffer is retired, not freed: a captured hipGraph may still launch with it
  size_t want = h->Vt_elems ? h->Vt_elems : (size_t)1 << 20;
  while (want < elems) want *= 2;
  void* p = nullptr;
  if (hipMalloc(&p, want * sizeof(unsigned short)) != hipSuccess) return UCE_ENOMEM;
  if (h->Vt) {
    if (h->n_retired < 32) h->retired[h->n_retired++] = h->Vt;
    else { UCE_HIP_TRY(hipDeviceSynchronize()); (void)hipFree(h->Vt); }
  }
  h->Vt = p;
  h->Vt_elems = want;
  return UCE_OK;
}

int uce_ensure_sk(uce_ctx* h, size_t bytes, size_t tiles) {
  if (bytes <= h->sk_bytes && tiles <= h->sk_tiles) return UCE_OK;
  UCE_HIP_TRY(hipSetDevice(h->device));
  auto retire = [&](void* p) -> int {
    if (!p) return UCE_OK;
    if (h->n_retired < 32) { h->retired[h->n_retired++] = p; return UCE_OK; }
    UCE_HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(p);
    return UCE_OK;
  };
  if (bytes > h->sk_bytes) {
    size_t want = h->sk_bytes ? h->sk_bytes : (size_t)32 << 20;
    while (want < bytes) want *= 2;
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) return UCE_ENOMEM;
    const int rc = retire(h->sk_ws);
    if (rc != UCE_OK) return rc;
    h->sk_ws = (float*)p;
    h->sk_bytes = want;
  }
  if (tiles > h->sk_tiles) {
    size_t want = h->sk_tiles ? h->sk_tiles : 4096;
    while (want < tiles) want *= 2;
    void* p = nullptr;
    if (hipMalloc(&p, want * sizeof(unsigned)) != hipSuccess) return UCE_ENOMEM;
    // (a blocking memset on the null stream: the tickets must be zero before the first launch on any stream reads them)
    if (hipMemset(p, 0, want * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return UCE_EHIP; }
    const int rc = retire(h->sk_tick);
    if (rc != UCE_OK) return rc;
    h->sk_tick = (unsigned*)p;
    h->sk_tiles = want;
  }
  return UCE_OK;
}

constexpr size_t LA_FLAGS = 2 * 22 * 22 + 2 * 22 + 8;          // k_potrf_la: systems of up to 22 diagonal blocks

int g_uce_conv_tapin = -1;

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

extern "C" {

int uce_version(void) { return 114; }

const char* uce_strerror(int code) {
  switch (code) {
    case UCE_OK: return "ok";
    case UCE_EINVAL: return "invalid argument";
    case UCE_ENOMEM: return "workspace allocation failed";
    case UCE_EDOM: return "system is not positive definite (check lambda > 0 and scales > 0)";
    case UCE_ENOSYS: return "not available in this build";
    case UCE_ECOMM: return "collective (RCCL) call failed";
    case UCE_ETIMEDOUT: return "an in-launch hand-off between workgroups timed out (hand-off words re-armed; retry)";
    default:
      if (code <= UCE_EHIP) return hipGetErrorString((hipError_t)(UCE_EHIP - code));
      return "unknown error";
  }
}

int uce_create(uce_handle_t* out, int device) {
  if (!out) return UCE_EINVAL;
  int ndev = 0;
  UCE_HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return UCE_EINVAL;
  UCE_HIP_TRY(hipSetDevice(device));
  uce_ctx* h = new (std::nothrow) uce_ctx();
  if (!h) return UCE_ENOMEM;
  *h = uce_ctx{};
  h->device = device;
  {
    const int cap = lr_rider_cap();
    const int want = env_int("UCE_RIDER_MAX_N", cap);
    h->sw = UceSwitches{env_int("UCE_XATTN_VARIANT", 1), want < cap ? want : cap, env_int("UCE_POTRF_VARIANT", 1), env_int("UCE_SATTN_QT", 0),
                        env_int("UCE_POTRF_RIDER_CUS", 250), env_int("UCE_SPLIT_MAX_NE", 128), env_int("UCE_SPLIT_MAX_N", 1 << 30),
                        env_int("UCE_PROJECT_LA", 1), env_int("UCE_GEMM_TILE", 0), env_int("UCE_SATTN_VTI", 0), env_int("UCE_CONV_TILE", 0),
                        env_int("UCE_CONV_W1", 1), env_int("UCE_SK_SPLIT", 0), env_int("UCE_GN_FUSED", 1), env_int("UCE_EDIT_RESIDENT", 1)};
  }
  g_uce_conv_tapin = env_int("UCE_CONV_TAPIN", -1);
  hipError_t e = hipMalloc((void**)&h->status, sizeof(int));
  if (e != hipSuccess) { delete h; return UCE_ENOMEM; }
  (void)hipMemset(h->status, 0, sizeof(int));
  if (hipMalloc((void**)&h->ticket, 8 * sizeof(unsigned)) != hipSuccess) { (void)hipFree(h->status); delete h; return UCE_ENOMEM; }
  (void)hipMemset(h->ticket, 0, 8 * sizeof(unsigned));
  if (hipMalloc((void**)&h->res_ws, lr_resident_ws_bytes()) != hipSuccess) h->res_ws = nullptr;      // (without it: the two-launch form)
  if (hipMalloc((void**)&h->la_flags, LA_FLAGS * sizeof(unsigned)) == hipSuccess) (void)hipMemset(h->la_flags, 0, LA_FLAGS * sizeof(unsigned));
  else h->la_flags = nullptr;                                  // (the launch chain is used instead)
  // (k_gn_fused: 1024 workgroups x 64 groups x 2 floats, 2 counters x 1024 samples; without them the two-kernel form runs)
  if (hipMalloc((void**)&h->gn_partial, (size_t)1024 * 64 * 2 * sizeof(float)) != hipSuccess) h->gn_partial = nullptr;
  if (h->gn_partial) {
    if (hipMalloc((void**)&h->gn_counters, 2 * 1024 * sizeof(unsigned)) == hipSuccess) (void)hipMemset(h->gn_counters, 0, 2 * 1024 * sizeof(unsigned));
    else { (void)hipFree(h->gn_partial); h->gn_partial = nullptr; h->gn_counters = nullptr; }
  }
  (void)hipDeviceSynchronize();                                // the zeroed hand-off words are in memory before any stream launches
  *out = h;
  return UCE_OK;
}

int uce_destroy(uce_handle_t h) {
  if (!h) return UCE_EINVAL;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  free_ws(h);
  if (h->status) (void)hipFree(h->status);
  if (h->T) (void)hipFree(h->T);
  if (h->Vt) (void)hipFree(h->Vt);
  if (h->gn_partial) (void)hipFree(h->gn_partial);
  if (h->gn_counters) (void)hipFree(h->gn_counters);
  if (h->sk_ws) (void)hipFree(h->sk_ws);
  if (h->sk_tick) (void)hipFree(h->sk_tick);
  for (int i = 0; i < h->n_retired; ++i) (void)hipFree(h->retired[i]);
  if (h->ticket) (void)hipFree(h->ticket);
  if (h->res_ws) (void)hipFree(h->res_ws);
  if (h->la_flags) (void)hipFree(h->la_flags);
  prof_clear(h);
  delete h;
  return UCE_OK;
}

int uce_reserve(uce_handle_t h, int d_max, int n_max) {
  if (!h || d_max <= 0 || n_max <= 0 || d_max % 64) return UCE_EINVAL;
  UCE_ENTER(h);
  return uce_ensure(h, d_max, round_up(n_max, 64));
}

int uce_reserve_rows(uce_handle_t h, long rows_max, int n_edit_max) {
  if (!h || rows_max <= 0 || n_edit_max <= 0) return UCE_EINVAL;
  UCE_ENTER(h);
  return uce_ensure_T(h, rows_max, n_edit_max);
}

int uce_lowrank_project(uce_handle_t h, const float* W_old, const float* Dm, float* T, long rows, int d,
                        int N_edit, uce_stream_t stream) {
  if (!h || !W_old || !Dm || !T || rows < 0 || N_edit <= 0 || !lowrank_split_supported(d, N_edit)) return UCE_EINVAL;
  UCE_ENTER(h);
  if (rows == 0) return UCE_OK;
  return launch_lr_project(W_old, Dm, nullptr, T, rows, d, N_edit, (hipStream_t)stream);
}

int uce_lowrank_update(uce_handle_t h, const float* W_old, const float* T, const float* R, float* W_new,
                       long rows, int d, int N_edit, uce_stream_t stream) {
  if (!h || !W_old || !T || !R || !W_new || rows < 0 || N_edit <= 0 || W_old == W_new ||
      !lowrank_split_supported(d, N_edit))
    return UCE_EINVAL;
  UCE_ENTER(h);
  if (rows == 0) return UCE_OK;
  return launch_lr_update(W_old, T, R, W_new, rows, d, N_edit, (hipStream_t)stream);
}

int uce_gram(uce_handle_t h, const float* C, const float* G, const float* s, int N, int N_edit, int d,
             float lamb, double* A, double* Bt, uce_stream_t stream) {
  if (!h || !C || !s || !A || !Bt || N <= 0 || N_edit < 0 || N_edit > N || d <= 0 || d % 64) return UCE_EINVAL;
  UCE_ENTER(h);
  if (N_edit > 0 && !G) return UCE_EINVAL;
  int rc = uce_ensure(h, d, d);
  if (rc) return rc;
  UceProfScope ps(h, "k_gram_primal", (hipStream_t)stream);
  return launch_gram_primal(h, C, G ? G : C, s, N, N_edit, d, lamb, A, Bt, (hipStream_t)stream);
}

int uce_solve_delta(uce_handle_t h, double* A, const double* Bt, int d, float* DeltaT, uce_stream_t stream) {
  if (!h || !A || !Bt || !DeltaT || d <= 0 || d % 64) return UCE_EINVAL;
  UCE_ENTER(h);
  int rc = uce_ensure(h, d, d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  {
    UceProfScope ps(h, "potrf", st);
    rc = launch_potrf(h, A, d, st);                  // k_potrf_first resets the status word
  }
  if (rc) return rc;
  UceProfScope ps(h, "k_trisolve", st);
  return launch_trisolve(h, d, d, Bt, nullptr, d, DeltaT, d, st, A);
}

int uce_solve_rhs(uce_handle_t h, double* A, const double* B, int d, int m, float* X, uce_stream_t stream) {
  if (!h || !A || !B || !X || d <= 0 || d % 64 || m <= 0 || m % 64) return UCE_EINVAL;
  UCE_ENTER(h);
  int rc = uce_ensure(h, d > m ? d : m, d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  {
    UceProfScope ps(h, "potrf", st);
    rc = launch_potrf(h, A, d, st);
  }
  if (rc) return rc;
  UceProfScope ps(h, "k_trisolve", st);
  return launch_trisolve(h, d, m, B, nullptr, d, X, d, st, A);
}

int uce_apply(uce_handle_t h, const float* W_old, const float* DeltaT, float* W_new, long rows, int d,
              uce_stream_t stream) {
  if (!h || !W_old || !DeltaT || !W_new || rows < 0 || d <= 0 || d % 64 || W_old == W_new) return UCE_EINVAL;
  UCE_ENTER(h);
  if (rows == 0) return UCE_OK;
  // f16 matrix cores with a two-way split of both operands (fp32-equivalent products from three f16 MFMAs, uce_apply_h2.hip); slabs
  // beyond its 2 GB buffer descriptors are walked in row chunks inside launch_apply_h2.  (The exact-f32 MFMA kernel and the three-way
  // bf16 split it replaced are under tools/ubench/retired with their numbers in HISTORY.md.)
  const int rc = uce_ensure(h, d, 64);
  if (rc) return rc;
  return launch_apply_h2(h, W_old, DeltaT, W_new, rows, d, (hipStream_t)stream);
}

int uce_dual_factors(uce_handle_t h, const float* C, const float* G, const float* s, int N, int N_edit,
                     int d, float lamb, float* Dm, float* R, uce_stream_t stream) {
  if (!h || !C || !s || !Dm || !R || N <= 0 || N_edit < 0 || N_edit > N || d <= 0 || d % 64) return UCE_EINVAL;
  UCE_ENTER(h);
  if (N_edit > 0 && !G) return UCE_EINVAL;
  const int n_pad = round_up(N, 64);
  int rc = uce_ensure(h, d, n_pad);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  int nsplit = 1;
  size_t slab_stride = 0;
  {
    UceProfScope ps(h, "k_gram_dual", st);
    rc = launch_gram_dual(h, C, s, N, d, lamb, h->M, n_pad, G, Dm, N_edit, &nsplit, &slab_stride, st);
  }
  if (rc) return rc;
  {
    UceProfScope ps(h, "potrf", st);
    rc = (nsplit > 1) ? launch_potrf_slabs(h, h->slabs, n_pad, nsplit, slab_stride, st, N)
                      : launch_potrf(h, h->M, n_pad, st, N);
  }
  if (rc) return rc;
  if (N_edit == 0) return UCE_OK;
  UceProfScope ps(h, "k_trisolve", st);
  return launch_trisolve(h, n_pad, d, nullptr, C, N, R, N_edit, st, h->M);
}

int uce_delta_from_factors(uce_handle_t h, const float* Dm, const float* R, int N_edit, int d,
                           float* DeltaT, uce_stream_t stream) {
  if (!h || !DeltaT || N_edit < 0 || d <= 0 || d % 64) return UCE_EINVAL;
  UCE_ENTER(h);
  if (N_edit > 0 && (!Dm || !R)) return UCE_EINVAL;
  UceProfScope ps(h, "k_delta_factors", (hipStream_t)stream);
  return launch_delta_from_factors(Dm, R, N_edit, d, DeltaT, (hipStream_t)stream);
}

int uce_apply_lowrank(uce_handle_t h, const float* W_old, const float* Dm, const float* R, float* W_new,
                      long rows, int d, int N_edit, uce_stream_t stream) {
  if (!h || !W_old || !W_new || rows < 0 || d <= 0 || d % 64 || N_edit < 0 || N_edit > 256 || W_old == W_new)
    return UCE_EINVAL;
  UCE_ENTER(h);
  if (N_edit > 0 && (!Dm || !R)) return UCE_EINVAL;
  if (rows == 0) return UCE_OK;
  if (N_edit >= 1 && rows >= 1024 && lowrank_split_supported(d, N_edit)) {
    // the two-kernel form (projection into the handle's T, then the update pass)
    int rc = uce_ensure_T(h, rows, N_edit);
    if (rc) return rc;
    {
      UceProfScope ps(h, "k_lr_project", (hipStream_t)stream);
      rc = launch_lr_project(W_old, Dm, nullptr, h->T, rows, d, N_edit, (hipStream_t)stream);
    }
    if (rc) return rc;
    UceProfScope ps(h, "k_lr_update", (hipStream_t)stream);
    return launch_lr_update(W_old, h->T, R, W_new, rows, d, N_edit, (hipStream_t)stream);
  }
  UceProfScope ps(h, "k_apply_lowrank_generic", (hipStream_t)stream);
  return launch_apply_lowrank(W_old, Dm, R, W_new, rows, d, N_edit, (hipStream_t)stream);
}

// Jobs uce_edit leaves for the rider workgroups of the persistent Cholesky launch (h->h2_pending_*, h->bt_pending) and
// what that launch reports back (h->h2_done_*): whatever way uce_edit is left, nothing stays behind for a later
// factorisation or apply on the same handle.
namespace {
struct RiderJobs {
  uce_ctx* h;
  explicit RiderJobs(uce_ctx* h_) : h(h_) { clear(); }
  ~RiderJobs() { clear(); }
  void clear() {
    h->h2_pending_src = nullptr;
    h->h2_done_src = nullptr;
    h->bt_pending = GramPrimalArgs{};
  }
};
}  // namespace

int uce_edit(uce_handle_t h, const float* C, const float* G, const float* s, int N, int N_edit, int d,
             float lamb, const float* W_old, float* W_new, long rows, int algo, uce_stream_t stream) {
  if (!h || !C || !s || !W_old || !W_new || N <= 0 || N_edit < 0 || N_edit > N || d <= 0 || d % 64 ||
      rows < 0 || W_old == W_new)
    return UCE_EINVAL;
  UCE_ENTER(h);
  if (N_edit > 0 && !G) return UCE_EINVAL;
  if (algo == UCE_ALGO_AUTO) algo = (round_up(N, 64) < d) ? UCE_ALGO_DUAL : UCE_ALGO_PRIMAL;
  int rc;
  if (algo == UCE_ALGO_PRIMAL) {
    rc = uce_ensure(h, d, d);
    if (rc) return rc;
    // Two pieces of this step depend on nothing but its inputs and are not needed before the Cholesky is over: the f16
    // split of W_old (for the dense apply) and Bt, the right-hand side of the solve.  When the factorisation is the
    // persistent launch (133 CUs busy for ~215 us at d = 768, latency-bound), rider workgroups of THAT launch do both on
    // the CUs it leaves idle, and the Gram launch in front of it computes A alone, split over the concepts.
    const bool ride = potrf_la_has_room(h, d);
    const bool ride_bt = ride && N_edit > 0 && N_edit <= 2048;
    RiderJobs jobs(h);
    if (ride) {
      // (each rider contracts its d/64 x d/64 tile of Bt over ALL edit concepts: beyond ~2000 of them that outlasts the ~215 us
      //  factorisation and the "free" Bt would become the critical path of the launch - the Gram launch computes it then)
      if (ride_bt)
        h->bt_pending = GramPrimalArgs{C, G, s, N, N_edit, d, lamb, h->M, h->Bt, (N_edit + 31) / 32 * 32, (size_t)0};
      if (rows > 0 && apply_h2_fits(rows, d)) {
        h->h2_pending_src = W_old;
        h->h2_pending_rows = rows;
        h->h2_pending_d = d;
      }
    }
    {
      UceProfScope ps(h, "k_gram_primal", (hipStream_t)stream);
      rc = launch_gram_primal(h, C, G ? G : C, s, N, N_edit, d, lamb, h->M, h->Bt, (hipStream_t)stream,
                              ride_bt ? 1 : 0);
    }
    if (rc) return rc;
    {
      UceProfScope ps(h, "potrf", (hipStream_t)stream);
      rc = launch_potrf(h, h->M, d, (hipStream_t)stream);
    }
    if (!rc && h->bt_pending.C) {                                   // the factorisation took another form: Bt in a launch of its own
      UceProfScope ps(h, "k_gram_primal", (hipStream_t)stream);
      rc = launch_gram_primal(h, C, G, s, N, N_edit, d, lamb, h->M, h->Bt, (hipStream_t)stream, 2);
    }
    if (rc) return rc;
    {
      UceProfScope ps(h, "k_trisolve", (hipStream_t)stream);
      rc = launch_trisolve(h, d, d, h->Bt, nullptr, d, h->DeltaT, d, (hipStream_t)stream, h->M);
    }
    if (rc) return rc;
    return uce_apply(h, W_old, h->DeltaT, W_new, rows, d, stream);
  }
  if (algo != UCE_ALGO_DUAL) return UCE_EINVAL;
  const int n_pad = round_up(N, 64);
  rc = uce_ensure(h, d, n_pad);
  if (rc) return rc;
  if (N_edit >= 1 && rows >= 1024 && lowrank_split_supported(d, N_edit) && N_edit <= h->sw.split_max_ne && N <= h->sw.split_max_n) {
    // (more than 128 edit concepts: the dense form below - Delta + the f16 apply - is 25-33 % ahead of two projection and two
    //  update passes since round 3, tools/ab_split.py; UCE_SPLIT_MAX_NE=256 brings the two-pass form back)
    // N <= 128: THREE launches on the caller's stream, no events (forking the Gram -> Cholesky -> solve chain onto a
    // side stream beside a rider-less projection was measured at 105 us against 70: the two cross-stream event
    // waits cost more than the overlap buys):
    //   1 projection T = W_old (G - C_e)^T   ||   rider blocks of the same launch: K = lambda S^-1 + C C^T and
    //     its (blocked) Cholesky + block inverses (hidden under the GEMM)
    //   2 triangular solves -> R     3 update W_new = W_old + T R
    // N > 128: Gram launch + potrf launch chain first, then the same three without the riders.
    rc = uce_ensure_T(h, rows, N_edit);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool riders = n_pad <= h->sw.rider_max_n;
    if (riders && h->sw.edit_resident && h->res_ws && lr_resident_supported(d, N, N_edit, rows)) {
      // ONE launch: W_old read once and held in registers between the projection and the update, both on the f16 matrix cores
      // with the two-term split; the chain rides in the first workgroups as below (uce_edit_resident.hip)
      UceProfScope ps(h, "k_lr_resident", st);
      return launch_lr_resident(h, W_old, G, C, s, W_new, rows, N, N_edit, lamb, h->res_ws, st);
    }
    if (riders) {
      // TWO launches: projection || (Gram -> Cholesky -> triangular solves, all in rider blocks of the same launch),
      // then the update
      UceProfScope ps(h, "k_lr_project", st);
      rc = launch_lr_project(W_old, G, C, h->T, rows, d, N_edit, st, h, C, s, N, lamb, h->R);
      if (rc) return rc;
    } else {
      int nsplit = 1;
      size_t slab_stride = 0;
      {
        UceProfScope ps(h, "k_gram_dual", st);
        rc = launch_gram_dual(h, C, s, N, d, lamb, h->M, n_pad, nullptr, nullptr, 0, &nsplit, &slab_stride, st);
      }
      if (rc) return rc;
      // 3 ... 16 diagonal blocks (129 ... 1024 concepts): the persistent Cholesky runs in the FIRST workgroups of the
      // projection launch - neither needs anything from the other (tools/ab_split.py; UCE_PROJECT_LA=0: two launches)
      PotrfLaJob la{};
      int own = 0;
      if (nsplit == 1 && N_edit <= 128 && h->sw.project_la && potrf_la_job(h, h->M, n_pad, N, &la, &own)) {
        UceProfScope ps(h, "k_lr_project", st);
        rc = launch_lr_project_la(W_old, G, C, h->T, rows, d, N_edit, la, own, st);
        if (rc) return rc;
      } else {
        {
          UceProfScope ps(h, "potrf", st);
          rc = (nsplit > 1) ? launch_potrf_slabs(h, h->slabs, n_pad, nsplit, slab_stride, st, N)
                            : launch_potrf(h, h->M, n_pad, st, N);
        }
        if (rc) return rc;
        UceProfScope ps(h, "k_lr_project", st);
        rc = launch_lr_project(W_old, G, C, h->T, rows, d, N_edit, st);
        if (rc) return rc;
      }
    }
    if (!riders) {
      UceProfScope ps(h, "k_trisolve", st);
      rc = launch_trisolve(h, n_pad, d, nullptr, C, N, h->R, N_edit, st, h->M);
      if (rc) return rc;
    }
    UceProfScope ps(h, "k_lr_update", st);
    return launch_lr_update(W_old, h->T, h->R, W_new, rows, d, N_edit, st);
  }
  // More than 256 edit concepts against N < d: the dual system's Cholesky (persistent from 3 diagonal blocks), then Delta and
  // the dense apply - whose f16 split of W_old again rides in the Cholesky launch
  RiderJobs jobs(h);
  const bool dense = !apply_lowrank_fits(d, N_edit) || (N_edit >= 1 && (N_edit > h->sw.split_max_ne || N > h->sw.split_max_n));
  if (dense && rows > 0 && apply_h2_fits(rows, d) && potrf_la_has_room(h, n_pad)) {
    h->h2_pending_src = W_old;
    h->h2_pending_rows = rows;
    h->h2_pending_d = d;
  }
  rc = uce_dual_factors(h, C, G, s, N, N_edit, d, lamb, h->Dm, h->R, stream);
  h->h2_pending_src = nullptr;
  if (rc) return rc;
  if (!dense)
    return uce_apply_lowrank(h, W_old, h->Dm, h->R, W_new, rows, d, N_edit, stream);
  rc = uce_delta_from_factors(h, h->Dm, h->R, N_edit, d, h->DeltaT, stream);
  if (rc) return rc;
  return uce_apply(h, W_old, h->DeltaT, W_new, rows, d, stream);
}

int uce_profile_begin(uce_handle_t h) {
  if (!h) return UCE_EINVAL;
  prof_clear(h);
  h->prof = new (std::nothrow) uce_prof();
  return h->prof ? UCE_OK : UCE_ENOMEM;
}

int uce_profile_end(uce_handle_t h, uce_stream_t stream, char* report, size_t cap) {
  if (!h || !h->prof || !report || cap == 0) return UCE_EINVAL;
  UCE_ENTER(h);
  UCE_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  struct Agg { const char* name; double ms; int n; };
  std::vector<Agg> agg;
  for (auto& r : h->prof->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    size_t i = 0;
    for (; i < agg.size(); ++i)
      if (!strcmp(agg[i].name, r.name)) break;
    if (i == agg.size()) agg.push_back(Agg{r.name, 0.0, 0});
    agg[i].ms += ms;
    agg[i].n += 1;
  }
  prof_clear(h);
  size_t off = 0;
  report[0] = 0;
  for (auto& a : agg) {
    const int w = snprintf(report + off, cap - off, "%s %.6f %d\n", a.name, a.ms, a.n);
    if (w < 0 || (size_t)w >= cap - off) return UCE_EINVAL;
    off += (size_t)w;
  }
  return UCE_OK;
}

// ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t (ncclUint8 = 1), int root, ncclComm_t, hipStream_t)
typedef int (*nccl_broadcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);

static nccl_broadcast_fn resolve_nccl_broadcast() {
  static nccl_broadcast_fn fn = []() -> nccl_broadcast_fn {
    void* sym = nullptr;
    if (const char* path = getenv("UCE_RCCL_LIB")) {               // an explicit library (the one `comm` was created with)
      if (void* lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL)) sym = dlsym(lib, "ncclBroadcast");
    }
    if (!sym) sym = dlsym(RTLD_DEFAULT, "ncclBroadcast");          // the RCCL the host process already carries
    if (!sym) {
      void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (lib) sym = dlsym(lib, "ncclBroadcast");
    }
    return (nccl_broadcast_fn)sym;
  }();
  return fn;
}

int uce_bcast(uce_handle_t h, void* buf, size_t bytes, int root, void* comm, uce_stream_t stream) {
  if (!h || !buf || !comm || root < 0) return UCE_EINVAL;
  UCE_ENTER(h);
  if (bytes == 0) return UCE_OK;
  const nccl_broadcast_fn bc = resolve_nccl_broadcast();
  if (!bc) return UCE_ENOSYS;
  const int rc = bc(buf, buf, bytes, /* ncclUint8 */ 1, root, comm, (hipStream_t)stream);
  return rc == 0 ? UCE_OK : UCE_ECOMM;
}

int uce_status(uce_handle_t h, int* info, uce_stream_t stream) {
  if (!h || !info) return UCE_EINVAL;
  UCE_ENTER(h);
  int v = 0;
  UCE_HIP_TRY(hipMemcpyAsync(&v, h->status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  UCE_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  *info = v;
  if (v < 0) {
    // a bounded in-launch wait gave up (a hand-off of the cooperating workgroups never arrived): NOT a property of the system.
    // A late poster may have set flags after the launch's last block cleared them - re-arm every hand-off word and the status
    // word before the next launch on this handle can pass its waits early.
    (void)hipMemsetAsync(h->ticket, 0, 8 * sizeof(unsigned), (hipStream_t)stream);
    if (h->la_flags) (void)hipMemsetAsync(h->la_flags, 0, LA_FLAGS * sizeof(unsigned), (hipStream_t)stream);
    if (h->gn_counters) (void)hipMemsetAsync(h->gn_counters, 0, 2 * 1024 * sizeof(unsigned), (hipStream_t)stream);   // (k_gn_fused's arrive / depart words)
    (void)hipMemsetAsync(h->status, 0, sizeof(int), (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    return UCE_ETIMEDOUT;
  }
  return v ? UCE_EDOM : UCE_OK;
}

}  // extern "C"
