// GroupNorm (+ SiLU) forward for channels-last activations of the U-Net / VAE at inference - SURVEY.md section
// 8(f) row 3, "the rest of the U-Net step".  diffusers' ResnetBlock2D computes conv(silu(group_norm(x))): torch
// runs that as three GroupNorm kernels (moments, fused parameters, a STRIDED element-wise apply for NHWC tensors)
// plus a SiLU pass - five trips over the activation.  Here: two kernels, three trips, all 16-byte coalesced.
//
//   x, y : [N, H*W, C] (= NCHW tensors in torch.channels_last memory format), bf16 or f16; gamma, beta : [C], same
//          dtype; G groups of C/G consecutive channels; statistics and arithmetic in f32 (final reduction in f64).
//   k_gn_stats : grid (chunks, N).  A workgroup walks its chunk of pixels; thread t owns channel octet(s) t % OC
//                of pixel rows t / OC + k*RPI, accumulates per-channel sum / sum of squares in registers, the
//                workgroup folds them per group through LDS -> partial[n][chunk][g] = (sum, sumsq).
//   k_gn_apply : grid (chunks, N).  Folds the partials of its n (fixed order: bit-repeatable) into mean / rstd,
//                builds per-channel scale and shift, streams its chunk:  y = silu?(x * a_c + b_c).
#include "uce_common.h"

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

template <bool F16>
__device__ __forceinline__ void unpack8(const uint4_t v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (F16) {
      const unsigned u = v[i];
      f[2 * i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
      f[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
    } else {
      f[2 * i] = __uint_as_float(v[i] << 16);
      f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
  }
}

template <bool F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const float2_t v = {lo, hi};
  if constexpr (F16)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  else
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

constexpr int MAXO = 2;     // channel octets per thread: C <= 8 * 256 * MAXO = 4096

struct GnMap {
  int OC, RPI, NO, active;  // octets per pixel, pixel rows per iteration, octets per thread, active threads
};
__host__ __device__ inline GnMap gn_map(int C) {
  GnMap m;
  m.OC = C / 8;
  m.NO = (m.OC + 255) / 256;
  const int tpr = (m.OC + m.NO - 1) / m.NO;          // threads per pixel row
  m.RPI = 256 / tpr;
  m.active = m.RPI * tpr;
  return m;
}

// `addend` (optional, [N, C], same dtype): x + addend[n][c] is what gets normalised (the ResnetBlock2D's
// "h + time_emb_proj(...)[:, :, None, None]" and the bias of the convolution that produced x, folded in).
// `x2` (optional): the channels [C1, C) of every pixel come from a second tensor x2 [N, HW, C - C1] - the skip connection of an
// up block, whose torch.cat([x, skip], dim=1) is then never written (C1 % 8 == 0; x2 == nullptr: C1 == C).
// (non-temporal loads HERE were measured and lose 5-15 %: the last-level cache keeps the tail of the tensor for the apply pass)
template <bool F16>
__global__ __launch_bounds__(256) void k_gn_stats(const unsigned short* __restrict__ x, const unsigned short* __restrict__ x2, int C1,
                                                  const unsigned short* __restrict__ addend,
                                                  float* __restrict__ partial, int HW, int C, int G, int chunks, long ald) {
  extern __shared__ __attribute__((aligned(16))) float red[];     // [RPI][C][2]
  const GnMap mp = gn_map(C);
  const int tpr = mp.active / mp.RPI;
  const int tid = threadIdx.x;
  const int n = blockIdx.y, ch = blockIdx.x;
  const int p0 = (int)((long)HW * ch / chunks), p1 = (int)((long)HW * (ch + 1) / chunks);
  const int oc0 = tid % tpr, prow = tid / tpr;
  float s[MAXO][8], q[MAXO][8];
#pragma unroll
  for (int j = 0; j < MAXO; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) s[j][i] = q[j][i] = 0.f;
  if (tid < mp.active) {
    const unsigned short* src[MAXO];                   // this thread's octet j of pixel 0 of sample n, and its pixel stride
    int pst[MAXO];
    float ad[MAXO][8];
#pragma unroll
    for (int j = 0; j < MAXO; ++j) {
      const int oc = oc0 + j * tpr;
      const bool second = oc * 8 >= C1;
      pst[j] = second ? C - C1 : C1;
      src[j] = second ? x2 + (size_t)n * HW * (C - C1) + (oc * 8 - C1) : x + (size_t)n * HW * C1 + oc * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) ad[j][i] = 0.f;
      if (addend && j < mp.NO && oc < mp.OC) unpack8<F16>(*(const uint4_t*)(addend + (size_t)n * ald + oc * 8), ad[j]);
    }
    auto take = [&](const uint4_t (&v)[MAXO]) {
#pragma unroll
      for (int j = 0; j < MAXO; ++j) {
        const int oc = oc0 + j * tpr;
        if (j < mp.NO && oc < mp.OC) {
          float f[8];
          unpack8<F16>(v[j], f);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float w = f[i] + ad[j][i];
            s[j][i] += w;
            q[j][i] = fmaf(w, w, q[j][i]);
          }
        }
      }
    };
    auto fetch = [&](int p, uint4_t (&v)[MAXO]) {
#pragma unroll
      for (int j = 0; j < MAXO; ++j) {
        const int oc = oc0 + j * tpr;
        v[j] = (uint4_t){0u, 0u, 0u, 0u};
        if (j < mp.NO && oc < mp.OC) v[j] = *(const uint4_t*)(src[j] + (size_t)p * pst[j]);
      }
    };
    // four pixel rows in flight per thread (with few samples a workgroup's chunk is a handful of dependent round trips:
    // one prompt per call spent 12 us per launch on 5 MB); the accumulation order is the pixel order either way
    int p = p0 + prow;
    for (; p + 3 * mp.RPI < p1; p += 4 * mp.RPI) {
      uint4_t v0[MAXO], v1[MAXO], v2[MAXO], v3[MAXO];
      fetch(p, v0);
      fetch(p + mp.RPI, v1);
      fetch(p + 2 * mp.RPI, v2);
      fetch(p + 3 * mp.RPI, v3);
      take(v0);
      take(v1);
      take(v2);
      take(v3);
    }
    for (; p < p1; p += mp.RPI) {
      uint4_t v0[MAXO];
      fetch(p, v0);
      take(v0);
    }
#pragma unroll
    for (int j = 0; j < MAXO; ++j) {
      const int oc = oc0 + j * tpr;
      if (j < mp.NO && oc < mp.OC) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          red[((size_t)prow * C + oc * 8 + i) * 2] = s[j][i];
          red[((size_t)prow * C + oc * 8 + i) * 2 + 1] = q[j][i];
        }
      }
    }
  }
  __syncthreads();
  // one thread per group: fixed summation order (pixel rows outer, channels inner)
  if (tid < G) {
    const int cpg = C / G;
    double a = 0.0, b = 0.0;
    for (int r = 0; r < mp.RPI; ++r)
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
        a += (double)red[((size_t)r * C + c) * 2];
        b += (double)red[((size_t)r * C + c) * 2 + 1];
      }
    float* out = partial + (((size_t)n * chunks + ch) * G + tid) * 2;
    out[0] = (float)a;
    out[1] = (float)b;
  }
}

// NTM: cache policy of the pass over the activation - 0 plain; 1 non-temporal stores; 2 non-temporal loads as well.  An activation far
// larger than the caches is read once and written once here: the hints take the normalised tensor's lines out of the way of the
// loads (N = 256: 64 x 64 x 320 432 -> 378 us for statistics + apply, 16 x 16 x 2560 226 -> 172; profiles/r06/gn_nontemporal_ab.log);
// a tensor the 256 MB last-level cache can hold keeps plain loads (the statistics pass has just brought it in) and, below 100 MB,
// plain stores (the next kernel reads it back).
template <bool F16, int NTM = 0>
__global__ __launch_bounds__(256) void k_gn_apply(const unsigned short* __restrict__ x, const unsigned short* __restrict__ x2, int C1,
                                                  const unsigned short* __restrict__ addend,
                                                  const unsigned short* __restrict__ gamma,
                                                  const unsigned short* __restrict__ beta, const float* __restrict__ partial,
                                                  unsigned short* __restrict__ y, int HW, int C, int G, int chunks,
                                                  float eps, int silu, long ald) {
  __shared__ float mean_s[64], rstd_s[64];
  __shared__ double fold_s[2][8][64];
  const GnMap mp = gn_map(C);
  const int tpr = mp.active / mp.RPI;
  const int tid = threadIdx.x;
  const int n = blockIdx.y, ch = blockIdx.x;      // (walking the tensor backwards - the tail the statistics pass read last - measured the same)
  const int cpg = C / G;
  // the partials of this sample, folded by all 256 threads: thread (sub, g) sums chunks sub, sub + nsub, ... of group g,
  // then one thread per group adds the nsub sums - a fixed order (bit-repeatable), 8 dependent steps instead of 64
  const int nsub = G <= 32 ? 8 : 4;
  {
    const int g = tid % (256 / nsub), sub = tid / (256 / nsub);
    double a = 0.0, b = 0.0;
    if (g < G)
      for (int c = sub; c < chunks; c += nsub) {
        const float* p = partial + (((size_t)n * chunks + c) * G + g) * 2;
        a += (double)p[0];
        b += (double)p[1];
      }
    if (g < 64) {
      fold_s[0][sub][g] = a;
      fold_s[1][sub][g] = b;
    }
  }
  __syncthreads();
  if (tid < G) {
    double a = 0.0, b = 0.0;
    for (int sub = 0; sub < nsub; ++sub) {
      a += fold_s[0][sub][tid];
      b += fold_s[1][sub][tid];
    }
    const double cnt = (double)HW * cpg;
    const double mu = a / cnt;
    double var = b / cnt - mu * mu;
    var = var > 0.0 ? var : 0.0;
    mean_s[tid] = (float)mu;
    rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if (tid >= mp.active) return;
  const int p0 = (int)((long)HW * ch / chunks), p1 = (int)((long)HW * (ch + 1) / chunks);
  const int oc0 = tid % tpr, prow = tid / tpr;
  float sa[MAXO][8], sb[MAXO][8];
  const unsigned short* src[MAXO];
  int pst[MAXO];
#pragma unroll
  for (int j = 0; j < MAXO; ++j) {
    const int oc = oc0 + j * tpr;
    const bool second = oc * 8 >= C1;
    pst[j] = second ? C - C1 : C1;
    src[j] = second ? x2 + (size_t)n * HW * (C - C1) + (oc * 8 - C1) : x + (size_t)n * HW * C1 + oc * 8;
    if (j < mp.NO && oc < mp.OC) {
      float g8[8], b8[8];
      unpack8<F16>(*(const uint4_t*)(gamma + oc * 8), g8);
      unpack8<F16>(*(const uint4_t*)(beta + oc * 8), b8);
      float a8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a8[i] = 0.f;
      if (addend) unpack8<F16>(*(const uint4_t*)(addend + (size_t)n * ald + oc * 8), a8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int grp = (oc * 8 + i) / cpg;
        sa[j][i] = g8[i] * rstd_s[grp];
        sb[j][i] = fmaf(a8[i] - mean_s[grp], sa[j][i], b8[i]);      // ((x + a) - mean) * rstd * gamma + beta
      }
    }
  }
  unsigned short* yb = y + (size_t)n * HW * C;
  auto fetch = [&](int p, uint4_t (&v)[MAXO]) {
#pragma unroll
    for (int j = 0; j < MAXO; ++j) {
      const int oc = oc0 + j * tpr;
      v[j] = (uint4_t){0u, 0u, 0u, 0u};
      if (j < mp.NO && oc < mp.OC) {
        if constexpr (NTM >= 2) v[j] = __builtin_nontemporal_load((const uint4_t*)(src[j] + (size_t)p * pst[j]));
        else v[j] = *(const uint4_t*)(src[j] + (size_t)p * pst[j]);
      }
    }
  };
  auto put = [&](int p, const uint4_t (&v)[MAXO]) {
#pragma unroll
    for (int j = 0; j < MAXO; ++j) {
      const int oc = oc0 + j * tpr;
      if (j < mp.NO && oc < mp.OC) {
        float f[8];
        unpack8<F16>(v[j], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float w = fmaf(f[i], sa[j][i], sb[j][i]);
          if (silu) w = w * __builtin_amdgcn_rcpf(1.0f + __expf(-w));
          f[i] = w;
        }
        const uint4_t o = {pack2<F16>(f[0], f[1]), pack2<F16>(f[2], f[3]), pack2<F16>(f[4], f[5]), pack2<F16>(f[6], f[7])};
        if constexpr (NTM >= 1) __builtin_nontemporal_store(o, (uint4_t*)(yb + (size_t)p * C + oc * 8));
        else *(uint4_t*)(yb + (size_t)p * C + oc * 8) = o;
      }
    }
  };
  int p = p0 + prow;
  for (; p + 3 * mp.RPI < p1; p += 4 * mp.RPI) {         // four pixel rows in flight per thread (see k_gn_stats)
    uint4_t v0[MAXO], v1[MAXO], v2[MAXO], v3[MAXO];
    fetch(p, v0);
    fetch(p + mp.RPI, v1);
    fetch(p + 2 * mp.RPI, v2);
    fetch(p + 3 * mp.RPI, v3);
    put(p, v0);
    put(p + mp.RPI, v1);
    put(p + 2 * mp.RPI, v2);
    put(p + 3 * mp.RPI, v3);
  }
  for (; p < p1; p += mp.RPI) {
    uint4_t v0[MAXO];
    fetch(p, v0);
    put(p, v0);
  }
}


// ---- ONE launch for small activations (one prompt per call: every GroupNorm of the U-Net; the small levels at any batch).
// The two-kernel form costs two launches and reads x twice; at a CFG batch of 2 a launch (~4-5 us of ramp, latency-bound
// round trips) is longer than the data takes (5 MB).  Here a workgroup keeps its chunk of pixels in REGISTERS (KR rows per
// thread), publishes its partial sums, waits until the chunks of its sample have all arrived, folds them and normalises its rows
// from the registers.  The whole grid must be resident at once (grid <= GNF_MAX_WG workgroups of 256 threads, <= 128 VGPRs - the
// launch bounds - and <= 32 KB of LDS: 4 per CU fit) - a grid-wide wait inside a sample; the spin is bounded (GNF_SPIN_MAX polls, ~1 s: a device that cannot
// hold the grid poisons the output with NaN rather than hanging).
//   partial [N][chunks][G][2] floats and arrive / depart [N] counters live in the HANDLE (zero between launches: the workgroup that
//   departs last re-arms both - nothing of the protocol is in the kernel arguments, so the launch can be captured and replayed).
// Hand-off: write-through (sc1) stores -> vmcnt(0) -> barrier -> relaxed agent-scope ticket; poll -> barrier -> sc1 loads
// (uce_lowrank_riders.h has the reasoning).
constexpr double GN_NT_STORE_BYTES = 100e6, GN_NT_LOAD_BYTES = 256e6;   // k_gn_apply's cache policy by activation size
constexpr int GNF_MAX_WG = 1024;
constexpr int GNF_MAX_N = 1024;
constexpr unsigned GNF_SPIN_MAX = 1u << 24;

template <bool F16, int NOT>
__global__ __launch_bounds__(256, 4) void k_gn_fused(const unsigned short* __restrict__ x, const unsigned short* __restrict__ x2, int C1,
                                                  const unsigned short* __restrict__ addend, const unsigned short* __restrict__ gamma,
                                                  const unsigned short* __restrict__ beta, float* __restrict__ partial,
                                                  unsigned* __restrict__ counters, unsigned short* __restrict__ y, int HW, int C,
                                                  int G, int chunks, float eps, int silu, long ald, int* __restrict__ status) {
  constexpr int KR = 8 / NOT;                                           // pixel rows a thread keeps
  extern __shared__ __attribute__((aligned(16))) float red[];          // [RPI][C][2]
  __shared__ float mean_s[64], rstd_s[64];
  __shared__ double fold_s[2][8][64];
  __shared__ unsigned flag_s;
  const GnMap mp = gn_map(C);
  const int tpr = mp.active / mp.RPI;
  const int tid = threadIdx.x;
  const int n = blockIdx.y, ch = blockIdx.x;
  const int P = mp.RPI * KR;
  const int p0 = ch * P, p1 = (p0 + P < HW) ? p0 + P : HW;
  const int oc0 = tid % tpr, prow = tid / tpr;
  const int cpg = C / G;
  const bool live = tid < mp.active;
  uint4_t v[KR][NOT];
  float ad[NOT][8];
  const unsigned short* src[NOT];
  int pst[NOT];
  bool has[NOT];
#pragma unroll
  for (int j = 0; j < NOT; ++j) {
    const int oc = oc0 + j * tpr;
    has[j] = live && j < mp.NO && oc < mp.OC;
    const bool second = oc * 8 >= C1;
    pst[j] = second ? C - C1 : C1;
    src[j] = second ? x2 + (size_t)n * HW * (C - C1) + (oc * 8 - C1) : x + (size_t)n * HW * C1 + oc * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) ad[j][i] = 0.f;
    if (addend && has[j]) unpack8<F16>(*(const uint4_t*)(addend + (size_t)n * ald + oc * 8), ad[j]);
  }
  // ---- the chunk into registers (every load in flight at once), then the per-channel sums in pixel order
#pragma unroll
  for (int k = 0; k < KR; ++k)
#pragma unroll
    for (int j = 0; j < NOT; ++j) {
      const int p = p0 + prow + k * mp.RPI;
      v[k][j] = (uint4_t){0u, 0u, 0u, 0u};
      if (has[j] && p < p1) v[k][j] = *(const uint4_t*)(src[j] + (size_t)p * pst[j]);
    }
  {
    float s[NOT][8], q[NOT][8];
#pragma unroll
    for (int j = 0; j < NOT; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) s[j][i] = q[j][i] = 0.f;
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
      for (int j = 0; j < NOT; ++j) {
        const int p = p0 + prow + k * mp.RPI;
        if (has[j] && p < p1) {
          float f[8];
          unpack8<F16>(v[k][j], f);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float w = f[i] + ad[j][i];
            s[j][i] += w;
            q[j][i] = fmaf(w, w, q[j][i]);
          }
        }
      }
#pragma unroll
    for (int j = 0; j < NOT; ++j) {
      const int oc = oc0 + j * tpr;
      if (has[j]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          red[((size_t)prow * C + oc * 8 + i) * 2] = s[j][i];
          red[((size_t)prow * C + oc * 8 + i) * 2 + 1] = q[j][i];
        }
      }
    }
  }
  __syncthreads();
  // (two threads per group - sums and sums of squares - in a fixed order: pixel rows outer, channels inner)
  if (tid < 2 * G) {
    const int g = tid >> 1, which = tid & 1;
    double a = 0.0;
    for (int r = 0; r < mp.RPI; ++r)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) a += (double)red[((size_t)r * C + c) * 2 + which];
    __hip_atomic_store(partial + (((size_t)n * chunks + ch) * G + g) * 2 + which, (float)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- wait for the sample's other chunks
  unsigned* arrive = counters + 2 * n;
  unsigned* depart = arrive + 1;
  if (tid == 0) {
    unsigned polls = 0;
    if (__hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)chunks - 1)   // (the last arriver need not poll)
      while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)chunks && ++polls < GNF_SPIN_MAX)
        __builtin_amdgcn_s_sleep(1);
    flag_s = polls < GNF_SPIN_MAX ? 1u : 0u;
    // an expired wait (the sample's other workgroups were not all resident: a partitioned / shared / CU-masked device beyond what
    // gn_fused_capacity() saw) is REPORTED: uce_status turns the negative word into UCE_ETIMEDOUT and re-arms the counters; the
    // NaN below marks the output so that nothing downstream looks valid either
    if (polls >= GNF_SPIN_MAX) atomicCAS(status, 0, -2);
    if (__hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)chunks - 1) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // everybody is past the poll: re-arm
      __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  const bool ok = flag_s != 0u;
  // ---- fold the partials of this sample (k_gn_apply's order: thread (sub, g) takes chunks sub, sub + nsub, ..; then the nsub sums)
  const int nsub = G <= 32 ? 8 : 4;
  {
    const int g = tid % (256 / nsub), sub = tid / (256 / nsub);
    double a = 0.0, b = 0.0;
    if (g < G) {
      const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)(partial + (size_t)n * chunks * G * 2), 0,
                                                                          (int)((size_t)chunks * G * 2 * sizeof(float)), 0x00020000);
      typedef unsigned uint2v __attribute__((ext_vector_type(2)));
      constexpr int FU = 16;                                             // independent round trips in flight (one round up to 16 nsub chunks)
      for (int c0 = sub; c0 < chunks; c0 += FU * nsub) {
        uint2v t[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const int c = c0 + u * nsub;
          t[u] = __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(pr, c < chunks ? (unsigned)(((size_t)c * G + g) * 8) : 0x80000000u, 0, 16 /* sc1 */));
        }
#pragma unroll
        for (int u = 0; u < FU; ++u)
          if (c0 + u * nsub < chunks) {
            a += (double)__uint_as_float(t[u][0]);
            b += (double)__uint_as_float(t[u][1]);
          }
      }
    }
    if (g < 64) {
      fold_s[0][sub][g] = a;
      fold_s[1][sub][g] = b;
    }
  }
  __syncthreads();
  if (tid < G) {
    double a = 0.0, b = 0.0;
    for (int sub = 0; sub < nsub; ++sub) {
      a += fold_s[0][sub][tid];
      b += fold_s[1][sub][tid];
    }
    const double cnt = (double)HW * cpg;
    const double mu = a / cnt;
    double var = b / cnt - mu * mu;
    var = var > 0.0 ? var : 0.0;
    mean_s[tid] = ok ? (float)mu : __builtin_nanf("");
    rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if (!live) return;
  // ---- normalise the rows held in registers
  unsigned short* yb = y + (size_t)n * HW * C;
#pragma unroll
  for (int j = 0; j < NOT; ++j) {
    if (!has[j]) continue;
    const int oc = oc0 + j * tpr;
    float g8[8], b8[8], sa[8], sb[8];
    unpack8<F16>(*(const uint4_t*)(gamma + oc * 8), g8);
    unpack8<F16>(*(const uint4_t*)(beta + oc * 8), b8);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int grp = (oc * 8 + i) / cpg;
      sa[i] = g8[i] * rstd_s[grp];
      sb[i] = fmaf(ad[j][i] - mean_s[grp], sa[i], b8[i]);              // ((x + a) - mean) * rstd * gamma + beta
    }
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int p = p0 + prow + k * mp.RPI;
      if (p < p1) {
        float f[8];
        unpack8<F16>(v[k][j], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float w = fmaf(f[i], sa[i], sb[i]);
          if (silu) w = w * __builtin_amdgcn_rcpf(1.0f + __expf(-w));
          f[i] = w;
        }
        const uint4_t o = {pack2<F16>(f[0], f[1]), pack2<F16>(f[2], f[3]), pack2<F16>(f[4], f[5]), pack2<F16>(f[6], f[7])};
        *(uint4_t*)(yb + (size_t)p * C + oc * 8) = o;
      }
    }
  }
}

}  // namespace

extern "C" int uce_groupnorm_chunks(int HW) {
  int c = HW / 16;                       // >= 16 pixels per workgroup; small feature maps still fill the chip
  return c < 1 ? 1 : (c > 64 ? 64 : c);
}

// ws: N * uce_groupnorm_chunks(HW) * G * 2 floats, owned by the caller
// Workgroups of k_gn_fused the device keeps resident AT ONCE (its grid-wide wait inside a sample needs the whole grid on the chip):
// occupancy of the largest-footprint instantiation (32 KB of dynamic LDS) x compute units of THIS device as the runtime reports
// them - a CPX-partitioned or smaller part gets its own figure instead of 256 x 4 - capped by the handle's partial-sum buffer.
// Asked once per handle (the handle is bound to one device).
static int gn_fused_capacity(uce_handle_t h) {
  if (h->gn_fused_cap > 0) return h->gn_fused_cap;
  int cap = GNF_MAX_WG;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) {
    int per_cu = 4;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_gn_fused<false, 2>, 256, 32 * 1024) == hipSuccess && nb > 0) per_cu = nb;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_gn_fused<false, 1>, 256, 32 * 1024) == hipSuccess && nb > 0 && nb < per_cu) per_cu = nb;
    const long c = (long)per_cu * prop.multiProcessorCount;
    cap = c < GNF_MAX_WG ? (int)c : GNF_MAX_WG;
  }
  h->gn_fused_cap = cap;
  return cap;
}

static int groupnorm_launch(uce_handle_t h, const void* x, const void* x2, int C1, const void* addend, const void* gamma,
                            const void* beta, void* y, float* ws, int N, int HW, int C, int G, float eps, int silu, int dtype,
                            long addend_ld, uce_stream_t stream) {
  if (!h || !x || !gamma || !beta || !y || !ws || N <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 64) return UCE_EINVAL;
  const long ald = addend_ld > 0 ? addend_ld : C;
  if (addend && (ald < C || ald % 8)) return UCE_EINVAL;
  UCE_ENTER(h);
  if (C % 8 || C % G || C > 8 * 256 * MAXO || N > 65535) return UCE_EINVAL;
  if (x2 ? (C1 <= 0 || C1 >= C || C1 % 8) : C1 != C) return UCE_EINVAL;
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  const int chunks = uce_groupnorm_chunks(HW);
  const GnMap mp = gn_map(C);
  const size_t smem = (size_t)mp.RPI * C * 2 * sizeof(float);
  if (smem > 64 * 1024) return UCE_EINVAL;
  const dim3 grid(chunks, N), block(256);
  hipStream_t st = (hipStream_t)stream;
  // small activations: ONE launch, the chunk held in registers across a grid-wide wait inside each sample (k_gn_fused)
  if (h->gn_partial && h->sw.gn_fused && smem <= 32 * 1024 && N <= GNF_MAX_N) {
    const int P = mp.RPI * (8 / mp.NO);
    const long fch = ((long)HW + P - 1) / P;
    if (fch * N <= gn_fused_capacity(h)) {
      const dim3 fgrid((unsigned)fch, N);
#define UCE_GNF(F16V, NOV)                                                                                                          \
  hipLaunchKernelGGL((k_gn_fused<F16V, NOV>), fgrid, block, smem, st, (const unsigned short*)x, (const unsigned short*)x2, C1,       \
                     (const unsigned short*)addend, (const unsigned short*)gamma, (const unsigned short*)beta, h->gn_partial,       \
                     h->gn_counters, (unsigned short*)y, HW, C, G, (int)fch, eps, silu, ald, h->status)
      if (dtype == UCE_DTYPE_F16) { if (mp.NO == 1) UCE_GNF(true, 1); else UCE_GNF(true, 2); }
      else { if (mp.NO == 1) UCE_GNF(false, 1); else UCE_GNF(false, 2); }
#undef UCE_GNF
      UCE_LAUNCH_CHECK();
      return UCE_OK;
    }
  }
  // cache policy of the apply pass by the size of the activation (see k_gn_apply)
  const double act_bytes = 2.0 * (double)N * HW * C;
  const int ntm = act_bytes >= GN_NT_LOAD_BYTES ? 2 : (act_bytes >= GN_NT_STORE_BYTES ? 1 : 0);
#define UCE_GNA(F16V, NTV)                                                                                                          \
  hipLaunchKernelGGL((k_gn_apply<F16V, NTV>), grid, block, 0, st, (const unsigned short*)x, (const unsigned short*)x2, C1,           \
                     (const unsigned short*)addend, (const unsigned short*)gamma, (const unsigned short*)beta, (const float*)ws,    \
                     (unsigned short*)y, HW, C, G, chunks, eps, silu, ald)
  if (dtype == UCE_DTYPE_F16) {
    hipLaunchKernelGGL(k_gn_stats<true>, grid, block, smem, st, (const unsigned short*)x, (const unsigned short*)x2, C1,
                       (const unsigned short*)addend, ws, HW, C, G, chunks, ald);
    if (ntm == 2) UCE_GNA(true, 2); else if (ntm == 1) UCE_GNA(true, 1); else UCE_GNA(true, 0);
  } else {
    hipLaunchKernelGGL(k_gn_stats<false>, grid, block, smem, st, (const unsigned short*)x, (const unsigned short*)x2, C1,
                       (const unsigned short*)addend, ws, HW, C, G, chunks, ald);
    if (ntm == 2) UCE_GNA(false, 2); else if (ntm == 1) UCE_GNA(false, 1); else UCE_GNA(false, 0);
  }
#undef UCE_GNA
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

extern "C" int uce_groupnorm_nhwc_fwd(uce_handle_t h, const void* x, const void* addend, const void* gamma, const void* beta,
                                      void* y, float* ws, int N, int HW, int C, int G, float eps, int silu, int dtype,
                                      long addend_ld, uce_stream_t stream) {
  return groupnorm_launch(h, x, nullptr, C, addend, gamma, beta, y, ws, N, HW, C, G, eps, silu, dtype, addend_ld, stream);
}

extern "C" int uce_groupnorm_cat_nhwc_fwd(uce_handle_t h, const void* x, const void* x2, int C1, const void* addend,
                                          const void* gamma, const void* beta, void* y, float* ws, int N, int HW, int C, int G,
                                          float eps, int silu, int dtype, long addend_ld, uce_stream_t stream) {
  if (!x2) return UCE_EINVAL;
  return groupnorm_launch(h, x, x2, C1, addend, gamma, beta, y, ws, N, HW, C, G, eps, silu, dtype, addend_ld, stream);
}

// -------------------------------------------------------------------------------------------------------------
// GEGLU of the transformer feed-forward (diffusers GEGLU: hidden, gate = proj(x).chunk(2, -1); hidden * gelu(gate),
// exact erf form): torch runs a strided gelu and a strided multiply on the two halves; here one pass.
//   x [rows, 2*inner] -> y [rows, inner], 16-bit, inner % 8 == 0
// -------------------------------------------------------------------------------------------------------------
namespace {
template <bool F16>
__global__ __launch_bounds__(256) void k_geglu(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                               long rows, int inner) {
  const int oct = inner / 8;
  const long total = rows * oct;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / oct;
    const int c = (int)(e - r * oct) * 8;
    float hv[8], gv[8];
    unpack8<F16>(*(const uint4_t*)(x + r * 2 * inner + c), hv);
    unpack8<F16>(*(const uint4_t*)(x + r * 2 * inner + inner + c), gv);
#pragma unroll
    for (int i = 0; i < 8; ++i) hv[i] *= 0.5f * gv[i] * (1.0f + erff(gv[i] * 0.70710678118654752f));
    const uint4_t o = {pack2<F16>(hv[0], hv[1]), pack2<F16>(hv[2], hv[3]), pack2<F16>(hv[4], hv[5]), pack2<F16>(hv[6], hv[7])};
    *(uint4_t*)(y + r * inner + c) = o;
  }
}
}  // namespace

extern "C" int uce_geglu_fwd(uce_handle_t h, const void* x, void* y, long rows, int inner, int dtype, uce_stream_t stream) {
  if (!h || !x || !y || rows <= 0 || inner <= 0 || inner % 8) return UCE_EINVAL;
  UCE_ENTER(h);
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  const long total = rows * (inner / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL(k_geglu<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                       (unsigned short*)y, rows, inner);
  else
    hipLaunchKernelGGL(k_geglu<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                       (unsigned short*)y, rows, inner);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

// -------------------------------------------------------------------------------------------------------------
// y = a + b + bias[c]  for channels-last 16-bit activations [N*HW, C]: the residual join of a ResnetBlock2D /
// Transformer2DModel with the bias of the (bias-free launched) convolution that produced `b` folded in - one pass
// instead of MIOpen's separate bias kernel plus a torch add.  b may be null (y = a + bias).
// -------------------------------------------------------------------------------------------------------------
namespace {
template <bool F16>
__global__ __launch_bounds__(256) void k_add_bias(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                  const unsigned short* __restrict__ bias, unsigned short* __restrict__ y,
                                                  long pixels, int C) {
  const int oct = C / 8;
  const long total = pixels * oct;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int oc = (int)(e % oct);
    float av[8], bv[8], cv[8];
    unpack8<F16>(*(const uint4_t*)(a + e * 8), av);
    if (b) unpack8<F16>(*(const uint4_t*)(b + e * 8), bv);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) bv[i] = 0.f;
    }
    if (bias) unpack8<F16>(*(const uint4_t*)(bias + oc * 8), cv);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) cv[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) av[i] = av[i] + (bv[i] + cv[i]);
    const uint4_t o = {pack2<F16>(av[0], av[1]), pack2<F16>(av[2], av[3]), pack2<F16>(av[4], av[5]), pack2<F16>(av[6], av[7])};
    *(uint4_t*)(y + e * 8) = o;
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// Classifier-free-guidance combine + PNDM (PLMS) scheduler step in ONE pass over the latents (SURVEY.md 8f row 3;
// reference: `noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)` and `scheduler.step(...)` inside
// `pipe(...)`, evalscripts/generate-images-sd.py:37-42 - a dozen tiny elementwise launches per denoising step in torch):
//   e      = eps[0:n] + g * (eps[n:2n] - eps[0:n])       (cfg; otherwise e = eps[0:n])   -> eps_out (history entry)
//   m      = w0 * e + w1 * h1 + w2 * h2 + w3 * h3          (the linear multistep combination of the stored outputs)
//   prev   = cs * sample - ce * m                          -> prev_out
// f32 arithmetic on 16-bit tensors, one rounding per output (e is re-read as rounded: it is what the history holds).
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void k_cfg_pndm(const unsigned short* __restrict__ eps, int cfg, float g,
                                                  const unsigned short* __restrict__ h1, const unsigned short* __restrict__ h2,
                                                  const unsigned short* __restrict__ h3, float w0, float w1, float w2, float w3,
                                                  const unsigned short* __restrict__ sample, float cs, float ce,
                                                  unsigned short* __restrict__ eps_out, unsigned short* __restrict__ prev_out,
                                                  long n8) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long)gridDim.x * 256) {
    float eu[8], ec[8], a[8], b[8], c[8], sv[8];
    unpack8<F16>(*(const uint4_t*)(eps + e * 8), eu);
    if (cfg) {
      unpack8<F16>(*(const uint4_t*)(eps + (n8 + e) * 8), ec);
#pragma unroll
      for (int i = 0; i < 8; ++i) eu[i] = fmaf(g, ec[i] - eu[i], eu[i]);
    }
    const uint4_t er = {pack2<F16>(eu[0], eu[1]), pack2<F16>(eu[2], eu[3]), pack2<F16>(eu[4], eu[5]), pack2<F16>(eu[6], eu[7])};
    *(uint4_t*)(eps_out + e * 8) = er;
    unpack8<F16>(er, eu);                                           // the rounded value, as the history stores it
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = b[i] = c[i] = 0.f;
    if (h1) unpack8<F16>(*(const uint4_t*)(h1 + e * 8), a);
    if (h2) unpack8<F16>(*(const uint4_t*)(h2 + e * 8), b);
    if (h3) unpack8<F16>(*(const uint4_t*)(h3 + e * 8), c);
    unpack8<F16>(*(const uint4_t*)(sample + e * 8), sv);
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float m = fmaf(w3, c[i], fmaf(w2, b[i], fmaf(w1, a[i], w0 * eu[i])));
      o[i] = fmaf(cs, sv[i], -ce * m);
    }
    *(uint4_t*)(prev_out + e * 8) = (uint4_t){pack2<F16>(o[0], o[1]), pack2<F16>(o[2], o[3]), pack2<F16>(o[4], o[5]), pack2<F16>(o[6], o[7])};
  }
}

extern "C" int uce_cfg_pndm_step(uce_handle_t h, const void* eps, int cfg, float guidance, const void* h1, const void* h2,
                                 const void* h3, const float* w, const void* sample, float cs, float ce, void* eps_out,
                                 void* prev_out, long n, int dtype, uce_stream_t stream) {
  if (!h || !eps || !w || !sample || !eps_out || !prev_out || n <= 0 || (n & 7)) return UCE_EINVAL;
  UCE_ENTER(h);
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  const long n8 = n / 8;
  long blocks = (n8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL(k_cfg_pndm<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)eps, cfg,
                       guidance, (const unsigned short*)h1, (const unsigned short*)h2, (const unsigned short*)h3, w[0], w[1], w[2],
                       w[3], (const unsigned short*)sample, cs, ce, (unsigned short*)eps_out, (unsigned short*)prev_out, n8);
  else
    hipLaunchKernelGGL(k_cfg_pndm<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)eps, cfg,
                       guidance, (const unsigned short*)h1, (const unsigned short*)h2, (const unsigned short*)h3, w[0], w[1], w[2],
                       w[3], (const unsigned short*)sample, cs, ce, (unsigned short*)eps_out, (unsigned short*)prev_out, n8);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

extern "C" int uce_add_bias_nhwc_fwd(uce_handle_t h, const void* a, const void* b, const void* bias, void* y, long pixels,
                                     int C, int dtype, uce_stream_t stream) {
  if (!h || !a || !y || pixels <= 0 || C <= 0 || C % 8) return UCE_EINVAL;
  UCE_ENTER(h);
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  const long total = pixels * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL(k_add_bias<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)a,
                       (const unsigned short*)b, (const unsigned short*)bias, (unsigned short*)y, pixels, C);
  else
    hipLaunchKernelGGL(k_add_bias<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)a,
                       (const unsigned short*)b, (const unsigned short*)bias, (unsigned short*)y, pixels, C);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

// -------------------------------------------------------------------------------------------------------------
// LayerNorm forward over the last dim of [rows, C] 16-bit activations (the three norms of a BasicTransformerBlock;
// diffusers' attention.py, reached from evalscripts/generate-images-sd.py:37-42), optionally fused with the residual
// join in front of it:  s = x + residual (rounded to the element type, written to sum_out),  y = LN(s) * gamma + beta.
// torch's vectorized_layer_norm kernel runs these at ~1.2 TB/s on an MI355X and the join is a separate pass.
//   TPR (8..64, power of two) consecutive lanes share a row, each holding NU 16-byte units (unit index sub + k*TPR),
//   so a wave streams 64/TPR rows per iteration; mean, then the centred sum of squares, reduced with lane shuffles;
//   f32 statistics; gamma / beta stay packed in registers for the thread's fixed units.
// -------------------------------------------------------------------------------------------------------------
namespace {
template <int NU, bool F16, bool RES>
__global__ __launch_bounds__(256) void k_layernorm(const unsigned short* __restrict__ x, const unsigned short* __restrict__ res,
                                                   const unsigned short* __restrict__ gamma,
                                                   const unsigned short* __restrict__ beta, unsigned short* __restrict__ y,
                                                   unsigned short* __restrict__ sum_out, long rows, int C, int tpr, float eps) {
  const int units = C / 8;
  const int sub = threadIdx.x & (tpr - 1);
  const long rpb = 256 / tpr;                                  // rows per workgroup per iteration
  uint4_t gq[NU], bq[NU];
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int u = sub + k * tpr;
    gq[k] = bq[k] = uint4_t{0u, 0u, 0u, 0u};
    if (u < units) {
      gq[k] = *(const uint4_t*)(gamma + u * 8);
      bq[k] = *(const uint4_t*)(beta + u * 8);
    }
  }
  const float inv_c = 1.0f / (float)C;
  for (long row = (long)blockIdx.x * rpb + threadIdx.x / tpr; row < rows; row += (long)gridDim.x * rpb) {
    uint4_t raw[NU], rr[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const int u = sub + k * tpr;
      raw[k] = rr[k] = uint4_t{0u, 0u, 0u, 0u};
      if (u < units) {
        raw[k] = *(const uint4_t*)(x + row * C + u * 8);
        if (RES) rr[k] = *(const uint4_t*)(res + row * C + u * 8);
      }
    }
    float v[NU][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      unpack8<F16>(raw[k], v[k]);
      if (RES) {
        float r8[8];
        unpack8<F16>(rr[k], r8);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[k][i] += r8[i];
        const uint4_t sq = {pack2<F16>(v[k][0], v[k][1]), pack2<F16>(v[k][2], v[k][3]), pack2<F16>(v[k][4], v[k][5]),
                            pack2<F16>(v[k][6], v[k][7])};
        if (sub + k * tpr < units) *(uint4_t*)(sum_out + row * C + (sub + k * tpr) * 8) = sq;
        unpack8<F16>(sq, v[k]);                                // normalise what the next layer will see
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[k][i];                // padding units hold zeros
    }
    for (int o = tpr >> 1; o; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NU; ++k)
      if (sub + k * tpr < units) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float dlt = v[k][i] - mean;
          q = fmaf(dlt, dlt, q);
        }
      }
    for (int o = tpr >> 1; o; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * inv_c + eps);
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const int u = sub + k * tpr;
      if (u < units) {
        float g8[8], b8[8], o8[8];
        unpack8<F16>(gq[k], g8);
        unpack8<F16>(bq[k], b8);
#pragma unroll
        for (int i = 0; i < 8; ++i) o8[i] = fmaf((v[k][i] - mean) * rstd, g8[i], b8[i]);
        const uint4_t oq = {pack2<F16>(o8[0], o8[1]), pack2<F16>(o8[2], o8[3]), pack2<F16>(o8[4], o8[5]), pack2<F16>(o8[6], o8[7])};
        *(uint4_t*)(y + row * C + u * 8) = oq;
      }
    }
  }
}

template <int NU>
void launch_layernorm(const void* x, const void* res, const void* gamma, const void* beta, void* y, void* sum_out, long rows,
                      int C, int tpr, float eps, int dtype, hipStream_t st) {
  const long rpb = 256 / tpr;
  long blocks = (rows + rpb - 1) / rpb;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const dim3 grid((unsigned)blocks), block(256);
  const unsigned short *xs = (const unsigned short*)x, *rs = (const unsigned short*)res, *gs = (const unsigned short*)gamma,
                       *bs = (const unsigned short*)beta;
  unsigned short *ys = (unsigned short*)y, *ss = (unsigned short*)sum_out;
  const bool f16 = dtype == UCE_DTYPE_F16;
  if (res) {
    if (f16) hipLaunchKernelGGL((k_layernorm<NU, true, true>), grid, block, 0, st, xs, rs, gs, bs, ys, ss, rows, C, tpr, eps);
    else hipLaunchKernelGGL((k_layernorm<NU, false, true>), grid, block, 0, st, xs, rs, gs, bs, ys, ss, rows, C, tpr, eps);
  } else {
    if (f16) hipLaunchKernelGGL((k_layernorm<NU, true, false>), grid, block, 0, st, xs, rs, gs, bs, ys, ss, rows, C, tpr, eps);
    else hipLaunchKernelGGL((k_layernorm<NU, false, false>), grid, block, 0, st, xs, rs, gs, bs, ys, ss, rows, C, tpr, eps);
  }
}
}  // namespace

extern "C" int uce_layernorm_fwd(uce_handle_t h, const void* x, const void* residual, const void* gamma, const void* beta,
                                 void* y, void* sum_out, long rows, int C, float eps, int dtype, uce_stream_t stream) {
  if (!h || !x || !gamma || !beta || !y || rows <= 0 || C <= 0 || C % 8) return UCE_EINVAL;
  UCE_ENTER(h);
  if ((residual != nullptr) != (sum_out != nullptr)) return UCE_EINVAL;
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  const int units = C / 8;
  if (units > 64 * 5) return UCE_ENOSYS;                        // C <= 2560
  int tpr = 8;
  while ((units + tpr - 1) / tpr > 5) tpr *= 2;
  const int nu = (units + tpr - 1) / tpr;
  hipStream_t st = (hipStream_t)stream;
  switch (nu) {
    case 1: launch_layernorm<1>(x, residual, gamma, beta, y, sum_out, rows, C, tpr, eps, dtype, st); break;
    case 2: launch_layernorm<2>(x, residual, gamma, beta, y, sum_out, rows, C, tpr, eps, dtype, st); break;
    case 3: launch_layernorm<3>(x, residual, gamma, beta, y, sum_out, rows, C, tpr, eps, dtype, st); break;
    case 4: launch_layernorm<4>(x, residual, gamma, beta, y, sum_out, rows, C, tpr, eps, dtype, st); break;
    default: launch_layernorm<5>(x, residual, gamma, beta, y, sum_out, rows, C, tpr, eps, dtype, st); break;
  }
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}

// -------------------------------------------------------------------------------------------------------------
// Row softmax between the two products of the VAE mid-block attention (one head of 512 dims: diffusers AutoencoderKL's
// `Attention` under `pipe(...).images`, evalscripts/generate-images-sd.py:37-42 -> vae.decode): s [rows, L] f32 scores in,
// p [rows, L] 16-bit probabilities out.  One workgroup per row: the row (<= 64 KB) lives in registers (up to 16 floats per
// thread), max and sum by wave shuffles + one LDS exchange, f32 arithmetic, one rounding.
// -------------------------------------------------------------------------------------------------------------
namespace {
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
template <bool F16>
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ s, unsigned short* __restrict__ p, int L, float scale_log2e) {
  __shared__ float red[8];
  const long row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* src = s + row * L;
  unsigned short* dst = p + row * L;
  constexpr int MAXQ = 16;                           // quads (4 floats) per thread: L <= 256 * 4 * 16 = 16384
  float4_t v[MAXQ];
  const int nq = L / 4;
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int q = tid + 256 * i;
    if (q < nq) {
      v[i] = *(const float4_t*)(src + 4 * q);
      m = fmaxf(m, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float mc = m * scale_log2e;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int q = tid + 256 * i;
    if (q < nq) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[i][j] = __builtin_amdgcn_exp2f(fmaf(v[i][j], scale_log2e, -mc));
        sum += v[i][j];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int q = tid + 256 * i;
    if (q < nq) *(uint2_t*)(dst + 4 * q) = (uint2_t){pack2<F16>(v[i][0] * inv, v[i][1] * inv), pack2<F16>(v[i][2] * inv, v[i][3] * inv)};
  }
}
}  // namespace

extern "C" int uce_softmax_rows(uce_handle_t h, const float* s, void* p, long rows, int L, float scale, int dtype, uce_stream_t stream) {
  if (!h || !s || !p || rows <= 0 || L <= 0 || L % 8 || L > 16384 || rows > 0x7fffffffL || !(scale > 0.f)) return UCE_EINVAL;
  UCE_ENTER(h);
  if (dtype != UCE_DTYPE_BF16 && dtype != UCE_DTYPE_F16) return UCE_ENOSYS;
  const float sl2 = scale * 1.4426950408889634f;
  if (dtype == UCE_DTYPE_F16)
    hipLaunchKernelGGL(k_softmax_rows<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (unsigned short*)p, L, sl2);
  else
    hipLaunchKernelGGL(k_softmax_rows<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (unsigned short*)p, L, sl2);
  UCE_LAUNCH_CHECK();
  return UCE_OK;
}
