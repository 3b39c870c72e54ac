// hipcc 7.2.0 (ROCm 7.2.0), gfx950: __builtin_amdgcn_permlane32_swap hands back its FIRST result twice.
//   hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S -o - permlane32_swap_builtin.hip | grep -A12 "^_Z2k2"
// k2 stores sw[0] and sw[1] from the same register (v1); the swapped partner register (v2) is never read.  The product code uses
// the instruction through inline assembly instead (uce_sattn.hip: lane_pair_max).
#include <hip/hip_runtime.h>
__global__ void k(float* o, const float* in) {
  float x = in[threadIdx.x];
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("" : "+v"(b));
  const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
}
__global__ void k2(float* o, const float* in) {
  float x = in[threadIdx.x];
  unsigned a = __builtin_bit_cast(unsigned, x);
  const auto sw = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  o[threadIdx.x] = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
  o[threadIdx.x + 64] = __builtin_bit_cast(float, sw[1]);
}
