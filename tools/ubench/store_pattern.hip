// R&D probe (not part of the product): how fast does a CU retire 16-byte-per-lane stores, by the shape of the 1 KB a wave instruction writes?
//   A: 16 rows x 64 B   (the accumulator layout of uce_edit_resident.hip's phase B: lane (n, kg) -> row n, bytes 16 kg .. 16 kg + 15)
//   B:  4 rows x 256 B  (after a transpose through LDS)
//   C:  1 row  x 1 KB   (fully contiguous)
//   D:  8 rows x 128 B  (whole cache lines, two tiles transposed through LDS)
// 223 workgroups x 7 waves (one per CU as in the product launch), each wave writes 16 rows x 3072 B = 48 KB, row stride 3072 B.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k_store(float* out, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w == 7) return;
  float* base = out + ((size_t)blockIdx.x * 112 + 16 * w) * 768;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 16 * 3072, 0x00020000);
  uint4_t v = {(unsigned)lane, 1u, 2u, 3u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 48; ++t) {
      unsigned off;
      if (PAT == 0) off = (unsigned)((lane & 15) * 3072 + (lane >> 4) * 16 + 64 * t);                 // 16 rows x 64 B
      else if (PAT == 1) off = (unsigned)(((lane >> 4) + 4 * (t & 3)) * 3072 + (lane & 15) * 16 + 256 * (t >> 2));   // 4 rows x 256 B
      else if (PAT == 2) off = (unsigned)((t / 3) * 3072 + (t % 3) * 1024 + lane * 16);                // 1 KB contiguous
      else off = (unsigned)(((lane >> 3) + 8 * (t & 1)) * 3072 + (lane & 7) * 16 + 128 * (t >> 1));   // 8 rows x 128 B (whole lines)
      __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
    }
  }
}

int main() {
  const int nwg = 223;
  float* out;
  hipMalloc(&out, (size_t)nwg * 112 * 768 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int pat = 0; pat < 4; ++pat) {
    for (int rep = 0; rep < 3; ++rep) {
      auto launch = [&](int iters) {
        if (pat == 0) hipLaunchKernelGGL(k_store<0>, dim3(nwg), dim3(512), 0, 0, out, iters);
        else if (pat == 1) hipLaunchKernelGGL(k_store<1>, dim3(nwg), dim3(512), 0, 0, out, iters);
        else if (pat == 2) hipLaunchKernelGGL(k_store<2>, dim3(nwg), dim3(512), 0, 0, out, iters);
        else hipLaunchKernelGGL(k_store<3>, dim3(nwg), dim3(512), 0, 0, out, iters);
      };
      for (int i = 0; i < 20; ++i) launch(1);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 200; ++i) launch(1);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / 200;
      printf("pattern %c: %.2f us per 76.7 MB launch  (%.0f GB/s, %.1f B/cycle/CU at 2.1 GHz on %d CUs)\n", "ABCD"[pat], us, 76.7e6 / us / 1e3,
             76.7e6 / nwg / (us * 2100.0), nwg);
    }
  }
  return 0;
}
