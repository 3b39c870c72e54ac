// LDS read throughput on gfx950 by access pattern: per wave 64 back-to-back ds_read_b128 (or b64), 1 / 4 / 8 waves.
//   hipcc --offload-arch=gfx950 -O3 ldsbw.hip -o ldsbw && ./ldsbw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int PAT, int B64>
__global__ void k(double* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) double lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 8192; e += blockDim.x) lds[e] = e * 0.5;
  __syncthreads();
  // byte offset of this lane's 16-byte chunk
  unsigned off;
  if (PAT == 0) off = lane * 16;                 // 64 distinct, contiguous
  else if (PAT == 1) off = (lane & 15) * 32;     // 16 distinct, 32-byte stride (the tile-row pattern)
  else if (PAT == 2) off = (lane >> 4) * 32;     // 4 distinct
  else if (PAT == 3) off = 0;                    // uniform
  else off = (lane & 15) * 16;                   // 16 distinct, contiguous
  const char* base = (const char*)lds + off;
  d2 acc = {0.0, 0.0};
  double acc1 = 0.0;
  unsigned long long t0, t1;
  asm volatile("s_barrier\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  // 4 batches of 16 independent reads (all 16 in flight, then consumed)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    d2 v[16];
    double s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (B64) s[i] = *(const double*)(base + (g * 16 + i) * 512);
      else v[i] = *(const d2*)(base + (g * 16 + i) * 512);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (B64) asm volatile("" : "+v"(s[i])); else asm volatile("" : "+v"(v[i]));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (B64) acc1 += s[i]; else acc += v[i];
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(acc), "+v"(acc1) :: "memory");
  if (lane == 0) cyc[tid >> 6] = t1 - t0;
  out[tid] = acc[0] + acc[1] + acc1;
}

template <int PAT, int B64>
void run(const char* name, double* out, unsigned long long* cyc) {
  for (int threads : {64, 256, 512}) {
    for (int r = 0; r < 3; ++r) k<PAT, B64><<<1, threads>>>(out, cyc);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("%-34s %s waves %d: %5llu cycles for 64 reads per wave -> %.1f cycles per wave-read, %.0f B/clk delivered\n", name,
           B64 ? "b64 " : "b128", threads / 64, mx, (double)mx / 64.0, (double)(threads / 64) * 64 * 64 * (B64 ? 8 : 16) / (double)mx);
  }
}

int main() {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 512 * 8);
  hipMalloc(&cyc, 64);
  run<0, 0>("64 distinct contiguous", out, cyc);
  run<1, 0>("16 distinct, 32 B stride", out, cyc);
  run<4, 0>("16 distinct contiguous", out, cyc);
  run<2, 0>("4 distinct", out, cyc);
  run<3, 0>("uniform", out, cyc);
  run<0, 1>("64 distinct contiguous", out, cyc);
  run<1, 1>("16 distinct, 32 B stride", out, cyc);
  run<3, 1>("uniform", out, cyc);
  return 0;
}
