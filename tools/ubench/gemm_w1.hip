// R&D probe (not part of the product): bf16 NT GEMM C[M,N] = A[M,K] B[N,K]^T with ONE wave per SIMD - the shape the GEMM
// library's best kernel on this chip has (rocprofv3: Custom_Cijk_..._MT256x256x64_MI16x16x1, 1.27-1.36 PF/s on the compute-bound
// linear shapes where k_gemm_dma's 8-wave 256 x 320 form reaches 1.0):
//   * workgroup 256 x 256, 4 waves = 2 (M) x 2 (N), wave tile 128 x 128 = 8 x 8 v_mfma_f32_16x16x32_bf16, f32 accumulators in
//     256 AGPRs (compile WITHOUT -amdgpu-mfma-vgpr-form): 16 fragment reads per 64 MFMAs - 0.25 per MFMA against 0.7 for the
//     64 x 160 wave tile of k_gemm_dma, whose LDS pipe is ~90 % busy at full MFMA rate;
//   * k-tiles of BK (64: two stages, 32: four stages) moved by global_load_lds_dwordx4 straight into LDS, bank swizzle on the source
//     piece, counted vmcnt, raw barriers; fragments of the next k-step are read while the MFMAs of the current one run.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DBKV=32] tools/ubench/gemm_w1.hip -o /tmp/gemm_w1 && /tmp/gemm_w1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
#include <type_traits>

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

#ifndef BKV
#define BKV 64
#endif
#ifndef BNV
#define BNV 256
#endif
constexpr int BM = 256, BN = BNV, BK = BKV;
constexpr int TNW = BN / 32;                        // 16-wide n tiles per wave (2 waves across N): 8 (BN = 256) or 10 (BN = 320)
constexpr int NACC = TNW < 8 ? TNW : 8;             // n-tile rows whose accumulators live in AGPRs (64 tiles = all 256 of them)
#ifndef NSTV
#define NSTV (BKV == 64 ? 2 : 4)
#endif
constexpr int NST = NSTV;
constexpr int ROWB = BK * 2;                        // bytes per row segment
constexpr int PPR = BK / 8;                         // 16-byte pieces per row
constexpr int RPW = 64 / PPR;                       // rows per DMA wave instruction
constexpr int STAGE = (BM + BN) * ROWB;
constexpr int NI = BM / RPW / 4;                    // DMA instructions per wave and k-tile for the A image (8 / 4)
constexpr int NIB = BN / RPW / 4;                   // ... for the B image

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const float2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ int swz(int R) { return BK == 64 ? ((R >> 1) & 7) : ((R >> 2) & 3); }

__global__ __launch_bounds__(256, 1) void k_gemm_w1(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                   unsigned short* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int l16 = lane & 15, lq = lane >> 4;
  const int ntile = N / BN;
  const int m0 = (blockIdx.x / ntile) * BM, n0 = (blockIdx.x % ntile) * BN;

  // staging: instruction j of wave w fills rows RPW (4 j + w) .. + RPW - 1 of the A image and of the B image.  Buffer loads:
  // the lane's offset is k-tile invariant (one VGPR per instruction), the k-tile displacement rides in the scalar offset, the
  // LDS destination in M0 - no vector ALU work per issue, so a DMA can sit anywhere in the MFMA stream.
  const int r = lane / PPR, p = lane % PPR;
  unsigned aofs[NI], bofs[NIB];
#pragma unroll
  for (int j = 0; j < NIB; ++j) {
    const int R = RPW * (4 * j + w) + r;
    const int c = p ^ swz(R);
    if (j < NI) aofs[j] = (unsigned)(((size_t)R * K + c * 8) * 2);
    bofs[j] = (unsigned)(((size_t)R * K + c * 8) * 2);
  }
  const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * K), 0, (int)((size_t)BM * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * K), 0, (int)((size_t)BN * K * 2), 0x00020000);
  // DMA number i (0 .. 2 NI - 1) of k-tile kt into stage st
  constexpr int PER = NI + NIB;                     // this wave's DMAs per k-tile: the A image's first, then the B image's
  auto dma = [&](int st, int kt, int i) __attribute__((always_inline)) {
    if (i < NI) {
      unsigned char* d = smem + st * STAGE + (4 * i + w) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (lds_void*)d, 16, aofs[i], kt * BK * 2, 0, 0);
    } else {
      const int j = i - NI;
      unsigned char* d = smem + st * STAGE + BM * ROWB + (4 * j + w) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (lds_void*)d, 16, bofs[j], kt * BK * 2, 0, 0);
    }
  };
  auto stage = [&](int st, int kt) {
#pragma unroll
    for (int i = 0; i < PER; ++i) dma(st, kt, i);
  };

  float4_t acc[TNW][8];                             // [n tile][m tile]
#pragma unroll
  for (int a = 0; a < TNW; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // fragment byte offsets inside a stage (k-step invariant part): row * ROWB, and the row's swizzle key
  int aoff[8], boff[TNW], akey[8], bkey[TNW];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int R = wm * 128 + b * 16 + l16;
    aoff[b] = R * ROWB;
    akey[b] = swz(R);
  }
#pragma unroll
  for (int a = 0; a < TNW; ++a) {
    const int R = wn * (BN / 2) + a * 16 + l16;
    boff[a] = BM * ROWB + R * ROWB;
    bkey[a] = swz(R);
  }
  constexpr int KS = BK / 32;                       // k-steps of 32 per k-tile
  const int NK = K / BK, last = NK - 1;
  constexpr int AHEAD = NST - 1;
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) stage(i, i < last ? i : last);
  // READY: tiles beyond the current one that are visible at the top of an iteration.  The deep ring (BK = 32, four stages) keeps
  // two tiles in flight and one landed ahead, so the first fragments of tile kt + 1 are read under the MFMAs of tile kt; the
  // two-stage ring (BK = 64) has the next tile still in flight and reads them right behind the barrier.
  constexpr int READY = NST >= 4 ? 1 : 0;
  constexpr int OUT = (AHEAD - 1 - READY) * PER;    // DMAs of this wave that may still be outstanding behind a wait
  if constexpr (OUT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OUT) : "memory");
  __builtin_amdgcn_s_barrier();

  uint4_t xf[2][8], wf[2][TNW];                     // fragments of the current and the next k-step
  auto frags = [&](int buf, const unsigned char* sb, int s) {
    const int c = 4 * s + lq;
#pragma unroll
    for (int b = 0; b < 8; ++b) xf[buf][b] = *(const uint4_t*)(sb + aoff[b] + ((c ^ akey[b]) << 4));
#pragma unroll
    for (int a = 0; a < TNW; ++a) wf[buf][a] = *(const uint4_t*)(sb + boff[a] + ((c ^ bkey[a]) << 4));
  };
  frags(0, smem, 0);
  int slot = 0, fill = AHEAD % NST;
  // one k-tile; PAR = fragment buffer of its first k-step (a compile-time constant: the buffers are register arrays)
  auto tile = [&](auto par_c, int kt) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    const int ktn = kt + AHEAD < last ? kt + AHEAD : last;   // the k-tile this iteration sends off (into the slot freed last)
    const unsigned char* sb = smem + slot * STAGE;
    const int nslot = slot + 1 == NST ? 0 : slot + 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int cur = (PAR + s) & 1, nxt = cur ^ 1;
      // where the fragments of the next k-step come from (this tile's next step, or - deep ring - the next tile's first one)
      const bool more = (s + 1 < KS) || (READY && kt + 1 < NK);
      const unsigned char* nb = (s + 1 < KS) ? sb : smem + nslot * STAGE;
      const int nc = 4 * ((s + 1 < KS) ? s + 1 : 0) + lq;
      // 64 MFMAs with their accumulators pinned in AGPRs (inline asm: left to the compiler the accumulators travel between the
      // register files around every MFMA); ONE fragment read of the next k-step behind every fourth MFMA, in the order the next
      // step needs them (its X fragments, then W fragments 0..7), fenced so the stream keeps this order
#pragma unroll
      for (int a = 0; a < TNW; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          if (a < NACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[a][b]) : "v"(wf[cur][a]), "v"(xf[cur][b]));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(wf[cur][a]), "v"(xf[cur][b]));
          constexpr int NMF = TNW * 8, NFR = 8 + TNW;      // MFMAs and fragment reads per k-step
          const int i = a * 8 + b;
          // fragment f of the next k-step goes out behind MFMA (f + 1) NMF / NFR - 1: evenly spread, in the order the next step
          // needs them (its X fragments, then W fragments 0 .. TNW - 1); the wave's DMAs of the outgoing k-tile likewise over the
          // first k-step
#pragma unroll
          for (int f = 0; f < NFR; ++f)
            if (i == ((f + 1) * NMF) / NFR - 1) {
              if (more) {
                if (f < 8) xf[nxt][f] = *(const uint4_t*)(nb + aoff[f] + ((nc ^ akey[f]) << 4));
                else wf[nxt][f - 8] = *(const uint4_t*)(nb + boff[f - 8] + ((nc ^ bkey[f - 8]) << 4));
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          if (s == 0) {
#pragma unroll
            for (int g = 0; g < PER; ++g)
              if (i == ((g + 1) * NMF) / PER - 2) {
                dma(fill, ktn, g);
                __builtin_amdgcn_sched_barrier(0);
              }
          }
        }
    }
    // the wave's own DMAs of the tile that must be visible next have landed; everybody is past its reads of the slot filled next
    if constexpr (OUT == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(OUT) : "memory");
    __builtin_amdgcn_s_barrier();
    if (!READY && kt + 1 < NK) frags((PAR + KS) & 1, smem + nslot * STAGE, 0);
    slot = nslot;
    fill = fill + 1 == NST ? 0 : fill + 1;
  };
  if constexpr (KS & 1) {                           // the first buffer alternates from tile to tile: two tiles per trip
    int kt = 0;
    for (; kt + 1 < NK; kt += 2) {
      tile(std::integral_constant<int, 0>{}, kt);
      tile(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < NK) tile(std::integral_constant<int, 0>{}, kt);
  } else {
    for (int kt = 0; kt < NK; ++kt) tile(std::integral_constant<int, 0>{}, kt);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // D layout (swapped product): lane column = l16 -> row m of C, registers q -> n = 4 lq + q of the 16-wide n tile
#pragma unroll
  for (int a = 0; a < TNW; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int m = m0 + wm * 128 + b * 16 + l16;
      const int n = n0 + wn * (BN / 2) + a * 16 + 4 * lq;
      const uint2_t o = {pack2(acc[a][b][0], acc[a][b][1]), pack2(acc[a][b][2], acc[a][b][3])};
      *(uint2_t*)(C + (size_t)m * N + n) = o;
    }
}

static unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf2f(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int shapes[][3] = {{512, 1280, 256}, {4096, 3840, 4096}, {8192, 7680, 8192}, {32768, 3840, 1280}, {32768, 10240, 1280},
                           {131072, 5120, 640}, {131072, 1280, 2880}, {524288, 1280, 320}, {8192, 1280, 11520}};
  for (auto& sh : shapes) {
    int M = sh[0], N = sh[1], K = sh[2];
    if (argc == 4) {
      if (&sh != &shapes[0]) break;
      M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]);
    }
    std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 1234567u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd());
    unsigned short *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_gemm_w1, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE);
    const int grid = (M / BM) * (N / BN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k_gemm_w1<<<grid, 256, NST * STAGE>>>(dA, dB, dC, M, N, K);
    hipEventRecord(e0);
    const int iters = 10;
    for (int i = 0; i < iters; ++i) k_gemm_w1<<<grid, 256, NST * STAGE>>>(dA, dB, dC, M, N, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    std::vector<unsigned short> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int t = 0; t < 64; ++t) {
      const int m = (int)(((long)t * 7919 + 13) % M), n = (int)(((long)t * 104729 + 7) % N);
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
      const double got = bf2f(hC[(size_t)m * N + n]);
      worst = fmax(worst, fabs(got - ref) / (fabs(ref) + sqrt((double)K) * 0.02));
    }
    printf("BK=%d M=%d N=%d K=%d: %.1f us, %.0f TF/s, worst scaled error %.2e (%s)\n", BK, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, worst,
           hipGetErrorString(hipGetLastError()));
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
