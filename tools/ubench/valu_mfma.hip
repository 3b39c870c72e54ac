// Issue cost of the softmax VALU instructions and how they overlap with MFMAs on gfx950 - inside one wave and between the
// two waves of a SIMD.  One workgroup; per mode the cycles (s_memtime) of wave 0 and of wave 4 for ITER rounds.
//   hipcc --offload-arch=gfx950 -O3 valu_mfma.hip -o valu_mfma && ./valu_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#define STAMP(t) asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
constexpr int ITER = 256;

#define EXP16 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n" \
  "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n" \
  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#define FMA16 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %0\n" \
  "v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %0\n" \
  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#define CVT16 asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n" \
  "v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n" \
  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#define MAX16 asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n" \
  "v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n" \
  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#define PKM16 asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n" \
  "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n" \
  : "+v"(p2[0]), "+v"(p2[1]), "+v"(p2[2]), "+v"(p2[3]) : "v"(p2[4]))
#define PKM2 asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n" : "+v"(p2[0]), "+v"(p2[1]) : "v"(p2[4]))
#define MUL16 asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %4\n v_mul_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_mul_f32 %7, %7, %0\n" \
  "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %4\n v_mul_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_mul_f32 %7, %7, %0\n" \
  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#define EXP4 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
#define FMA4 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0\n" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
#define FMA8 asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %0\n" \
  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#define MF(c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c, 0, 0, 0)
#define FENCE __builtin_amdgcn_sched_barrier(0)

template <int mode>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, float seed) {
  const int tid = threadIdx.x, wave = tid >> 6, grp = wave >> 2;
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + tid * 1e-6f;
  typedef float float2_t __attribute__((ext_vector_type(2)));
  float2_t p2[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) p2[i] = (float2_t){1.0f + seed * i, 1.0f - seed * i};
  float16_t c0, c1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  bf16x8_t fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
  unsigned long long t0, t1;
  __syncthreads();
  STAMP(t0);
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) switch (mode) {
      case 0: EXP16; EXP16; break;                                   // 32 exp
      case 1: FMA16; FMA16; break;                                   // 32 fma
      case 2: CVT16; CVT16; break;                                   // 32 cvt_pk
      case 3: MAX16; MAX16; break;                                   // 32 max3
      case 4: MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); break;   // 8 mfma, two accumulators
      case 5:                                                        // one wave: 8 x (mfma, 4 exp)
        MF(c0); FENCE; EXP4; FENCE; MF(c1); FENCE; EXP4; FENCE; MF(c0); FENCE; EXP4; FENCE; MF(c1); FENCE; EXP4; FENCE;
        MF(c0); FENCE; EXP4; FENCE; MF(c1); FENCE; EXP4; FENCE; MF(c0); FENCE; EXP4; FENCE; MF(c1); FENCE; EXP4; FENCE; break;
      case 6:                                                        // one wave: 8 x (mfma, 8 fma)
        MF(c0); FENCE; FMA8; FENCE; MF(c1); FENCE; FMA8; FENCE; MF(c0); FENCE; FMA8; FENCE; MF(c1); FENCE; FMA8; FENCE;
        MF(c0); FENCE; FMA8; FENCE; MF(c1); FENCE; FMA8; FENCE; MF(c0); FENCE; FMA8; FENCE; MF(c1); FENCE; FMA8; FENCE; break;
      case 7:                                                        // group 0: 32 exp, group 1: 8 mfma (two waves per SIMD)
        if (grp == 0) { EXP16; EXP16; } else { MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); } break;
      case 8:                                                        // group 0: 64 fma, group 1: 8 mfma
        if (grp == 0) { FMA16; FMA16; FMA16; FMA16; } else { MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); } break;
      case 9:                                                        // both groups: 8 x (mfma, 4 exp, 4 fma)
        MF(c0); FENCE; EXP4; FMA4; FENCE; MF(c1); FENCE; EXP4; FMA4; FENCE; MF(c0); FENCE; EXP4; FMA4; FENCE; MF(c1); FENCE; EXP4; FMA4; FENCE;
        MF(c0); FENCE; EXP4; FMA4; FENCE; MF(c1); FENCE; EXP4; FMA4; FENCE; MF(c0); FENCE; EXP4; FMA4; FENCE; MF(c1); FENCE; EXP4; FMA4; FENCE; break;
      case 11: asm volatile("" : "+v"(a[0])); break;
      case 12: PKM16; PKM16; break;                                  // 32 pk_mul (64 products)
      case 13: MUL16; MUL16; break;                                  // 32 mul
      case 14:                                                       // both groups: 8 x (mfma, 4 exp, 2 pk_mul)
        MF(c0); FENCE; EXP4; PKM2; FENCE; MF(c1); FENCE; EXP4; PKM2; FENCE; MF(c0); FENCE; EXP4; PKM2; FENCE; MF(c1); FENCE; EXP4; PKM2; FENCE;
        MF(c0); FENCE; EXP4; PKM2; FENCE; MF(c1); FENCE; EXP4; PKM2; FENCE; MF(c0); FENCE; EXP4; PKM2; FENCE; MF(c1); FENCE; EXP4; PKM2; FENCE; break;
      case 15:                                                       // both groups: 8 x (mfma, 4 exp)
        MF(c0); FENCE; EXP4; FENCE; MF(c1); FENCE; EXP4; FENCE; MF(c0); FENCE; EXP4; FENCE; MF(c1); FENCE; EXP4; FENCE;
        MF(c0); FENCE; EXP4; FENCE; MF(c1); FENCE; EXP4; FENCE; MF(c0); FENCE; EXP4; FENCE; MF(c1); FENCE; EXP4; FENCE; break;
      case 10:                                                       // 8 mfma, then 32 exp + 32 fma (phases in sequence)
        MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); MF(c0); MF(c1); FENCE; EXP16; EXP16; FMA16; FMA16; break;
    }
  }
  STAMP(t1);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += p2[i][0] + p2[i][1];
  out[tid] = s;
  if ((tid & 63) == 0) cyc[wave] = t1 - t0;
}

template <int M>
void launch1(int nw, float* out, unsigned long long* cyc) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(64 * nw), 0, 0, out, cyc, 0.001f); }
void launch(int mode, int nw, float* out, unsigned long long* cyc) {
  switch (mode) {
    case 0: launch1<0>(nw, out, cyc); break; case 1: launch1<1>(nw, out, cyc); break; case 2: launch1<2>(nw, out, cyc); break;
    case 3: launch1<3>(nw, out, cyc); break; case 4: launch1<4>(nw, out, cyc); break; case 5: launch1<5>(nw, out, cyc); break;
    case 6: launch1<6>(nw, out, cyc); break; case 7: launch1<7>(nw, out, cyc); break; case 8: launch1<8>(nw, out, cyc); break;
    case 9: launch1<9>(nw, out, cyc); break; case 10: launch1<10>(nw, out, cyc); break; case 11: launch1<11>(nw, out, cyc); break;
    case 12: launch1<12>(nw, out, cyc); break; case 13: launch1<13>(nw, out, cyc); break; case 14: launch1<14>(nw, out, cyc); break;
    case 15: launch1<15>(nw, out, cyc); break;
  }
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 8 * 8);
  const char* names[] = {"32 exp", "32 fma", "32 cvt_pk_bf16", "32 max3", "8 mfma32x32x16", "8 x (mfma | 4 exp), one stream", "8 x (mfma | 8 fma), one stream",
                         "waves 0-3: 32 exp, waves 4-7: 8 mfma", "waves 0-3: 64 fma, waves 4-7: 8 mfma", "8 x (mfma | 4 exp + 4 fma), one stream", "8 mfma then 32 exp + 32 fma", "empty",
                         "32 pk_mul_f32 (64 products)", "32 mul_f32", "8 x (mfma | 4 exp + 2 pk_mul), one stream", "8 x (mfma | 4 exp), one stream (=5)"};
  for (int nw = 4; nw <= 8; nw += 4)
    for (int mode = 0; mode <= 15; ++mode) {
      if (nw == 4 && (mode == 7 || mode == 8)) continue;
      unsigned long long h[8] = {0};
      for (int rep = 0; rep < 2; ++rep) {
        launch(mode, nw, out, cyc);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      printf("%d waves/SIMD  %-46s wave0 %7.1f  wave%d %7.1f  cycles per round\n", nw / 4, names[mode], (double)h[0] / ITER / 4, nw - 1, (double)h[nw - 1] / ITER / 4);
    }
  return 0;
}
