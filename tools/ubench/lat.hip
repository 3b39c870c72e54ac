// Latency probes on gfx950 (one workgroup): dependent f64 FMA chain, v_rcp_f64, LDS write->barrier->read round trip,
// s_memtime overhead.  hipcc --offload-arch=gfx950 -O3 lat.hip -o lat && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>

// s_memtime ordered against the value `v` (the compiler may not move the arithmetic on v across the stamp)
#define STAMP(t, v) asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(v) :: "memory")

__global__ void k(double* out, unsigned long long* cyc, double seed, int nwaves_active) {
  __shared__ double lds[64 * 64];
  const int tid = threadIdx.x, wave = tid >> 6;
  unsigned long long t0, t1;
  double x = seed + tid * 1e-9, y = 1.000000001;
  // (0) s_memtime back to back
  STAMP(t0, x);
  STAMP(t1, x);
  if (tid == 0) cyc[0] = t1 - t0;
  // (1) 64 dependent f64 FMAs
  STAMP(t0, x);
#pragma unroll
  for (int i = 0; i < 64; ++i) x = fma(x, y, 1e-12);
  STAMP(t1, x);
  if (tid == 0) cyc[1] = t1 - t0;
  // (2) 64 independent f64 FMAs (8 chains x 8)
  double z[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = x + j;
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(z[j]));
  STAMP(t0, z[0]);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = fma(z[j], y, 1e-12);
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(z[j]));
  STAMP(t1, z[7]);
  if (tid == 0) cyc[2] = t1 - t0;
  // (3) 16 dependent v_rcp_f64
  double r = x;
  STAMP(t0, r);
#pragma unroll
  for (int i = 0; i < 16; ++i) r = __builtin_amdgcn_rcp(r);
  STAMP(t1, r);
  if (tid == 0) cyc[3] = t1 - t0;
  // (4) LDS: write b128 -> waitcnt -> barrier -> broadcast read b128 -> use, 16 rounds
  double acc = 0.0;
  __syncthreads();
  STAMP(t0, acc);
  for (int i = 0; i < 16; ++i) {
    if (wave < nwaves_active) {
      if ((tid & 63) < 4) { lds[(i & 3) * 64 + 2 * tid] = x + i; lds[(i & 3) * 64 + 2 * tid + 1] = r; }
    }
    __syncthreads();
    if (wave < nwaves_active) acc += lds[(i & 3) * 64 + 2] + lds[(i & 3) * 64 + 3];
  }
  STAMP(t1, acc);
  if (tid == 0) cyc[4] = t1 - t0;
  // (5) LDS read latency alone: 16 dependent reads (pointer chase)
  int idx = tid & 63;
  lds[tid & 4095] = 0.0;
  __syncthreads();
  int* li = (int*)lds;
  li[tid] = (tid + 1) & 63;
  __syncthreads();
  STAMP(t0, idx);
#pragma unroll
  for (int i = 0; i < 16; ++i) idx = li[idx];
  STAMP(t1, idx);
  if (tid == 0) cyc[5] = t1 - t0;
  // (6) barrier alone, 16 rounds
  STAMP(t0, idx);
#pragma unroll
  for (int i = 0; i < 16; ++i) __syncthreads();
  STAMP(t1, idx);
  if (tid == 0) cyc[6] = t1 - t0;
  // (7) 16 dependent f32 FMAs x4
  float f = (float)x;
  STAMP(t0, f);
#pragma unroll
  for (int i = 0; i < 64; ++i) f = fmaf(f, 1.0000001f, 1e-12f);
  STAMP(t1, f);
  if (tid == 0) cyc[7] = t1 - t0;
  // (8) 16 dependent v_mfma_f64_16x16x4 (accumulator chain), (9) 16 x (mfma -> readlane -> fma on the scalar -> operand)
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 ca = {x, x, x, x};
  double ma = x * 1e-3, mb = y;
  STAMP(t0, ma);
#pragma unroll
  for (int i = 0; i < 16; ++i) ca = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, ca, 0, 0, 0);
  asm volatile("" : "+v"(ca));
  double cs = ca[0];
  STAMP(t1, cs);
  if (tid == 0) cyc[8] = t1 - t0;
  STAMP(t0, cs);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    ca = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, ca, 0, 0, 0);
    const int lo = __builtin_amdgcn_readlane(__double2loint(ca[0]), 5), hi = __builtin_amdgcn_readlane(__double2hiint(ca[0]), 5);
    ma = fma(__hiloint2double(hi, lo), 1e-9, ma);
  }
  cs = ca[1];
  STAMP(t1, cs);
  if (tid == 0) cyc[9] = t1 - t0;
  // (10) 16 independent mfma f64 (4 accumulators x 4)
  d4 cb[4] = {ca, ca, ca, ca};
  STAMP(t0, ma);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) cb[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, cb[j], 0, 0, 0);
  cs = cb[0][0] + cb[1][0] + cb[2][0] + cb[3][0];
  STAMP(t1, cs);
  if (tid == 0) cyc[10] = t1 - t0;
  out[tid] = cs + x + z[0] + z[1] + z[2] + z[3] + z[4] + z[5] + z[6] + z[7] + r + acc + idx + f;
}

int main() {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 1024 * 8);
  hipMalloc(&cyc, 64 * 8);
  for (int threads : {64, 256, 512}) {
    for (int rep = 0; rep < 3; ++rep) k<<<1, threads>>>(out, cyc, 1.5, 8);
    hipDeviceSynchronize();
    unsigned long long h[11];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads %d: memtime pair %llu | 64 dep f64 fma %llu (%.1f each) | 64 indep f64 fma %llu (%.1f each) | 16 dep rcp_f64 %llu (%.1f each) | "
           "16 LDS publish rounds %llu (%.1f each) | 16 dep LDS reads %llu (%.1f each) | 16 barriers %llu (%.1f each) | 64 dep f32 fma %llu (%.1f)\n",
           threads, h[0], h[1], h[1] / 64.0, h[2], h[2] / 64.0, h[3], h[3] / 16.0, h[4], h[4] / 16.0, h[5], h[5] / 16.0, h[6], h[6] / 16.0,
           h[7], h[7] / 64.0);
    printf("   16 dep mfma_f64_16x16x4 %llu (%.1f each) | 16 x (mfma -> readlane -> fma -> operand) %llu (%.1f each) | 16 indep mfma_f64 %llu (%.1f each)\n",
           h[8], h[8] / 16.0, h[9], h[9] / 16.0, h[10], h[10] / 16.0);
  }
  return 0;
}
