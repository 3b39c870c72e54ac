// R&D probe (not part of the product): k_conv3x3_w1 as a PERSISTENT kernel.  The product kernel
// (csrc/uce_conv_w1.hip, included below so both are the same translation unit and the same helpers) runs one 256 x BN output tile per
// workgroup, alone on its CU: the first DMA round trip and the epilogue of every tile are exposed.  Here a workgroup walks the tiles
// L = blockIdx.x, blockIdx.x + gridDim.x, ... : at the LAST k-tile of a tile the other ring stage is free, so the NEXT tile's first
// k-tile is sent there (with the next tile's staging coordinates, recomputed in place - the current tile issues no more DMAs) and
// lands under the last MFMAs and the epilogue, whose slabs use the stage that was just consumed (32-pixel slabs: 43 KB <= one stage).
// The harness compares the persistent kernel's output bit for bit with the product kernel's (same accumulation order) and times both.
// STATUS (profiles/r04/session2_ubench_conv_w1_persistent.txt): the first form - run-time `tail` conditions inside the MFMA stream -
// was bit-exact on all eight shapes and 2x slower (the conditions split the stream into ~300 basic blocks and the staging
// coordinates went to scratch); THIS form (last k-tile peeled, hot loop = the product kernel's) runs at the product kernel's speed
// within the run-to-run spread of the same launch (1 168 us against 1 168 / 992; 934 against 1 004 / 835) - the persistent walk buys
// nothing measurable - and 2.1 % of its outputs differ from the product kernel's (deterministic, unresolved).  Not a candidate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I unified-concept-editing_amd/csrc -I include tools/ubench/conv_w1p.hip -o /tmp/conv_w1p
#include "../../unified-concept-editing_amd/csrc/uce_conv_w1.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

template <int TNW, bool F16>
__global__ __launch_bounds__(256, 1) void k_conv3x3_w1p(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                        const unsigned short* __restrict__ bias, unsigned short* __restrict__ Y,
                                                        long M, int H, int W, int Cin, int Cout, int up, int mtiles, int ntiles,
                                                        int sd, const unsigned short* __restrict__ Rs) {
  constexpr int BN = 32 * TNW;
  constexpr int NB = BN / 8 / 4;
  constexpr int PER = W1_NA + NB;
  constexpr int STAGE = (W1_BM + BN) * W1_ROWB;
  constexpr int NMF = TNW * 8, NFR = 8 + TNW;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int l16 = lane & 15, lq = lane >> 4;
  const long T = (long)mtiles * ntiles;
  const int Hi = H * sd, Wi = W * sd;
  const int Hs = Hi >> up, Ws = Wi >> up;
  const int cch = Cin / W1_BK, NK = 9 * cch;
  const long K = 9L * Cin;
  const int r = lane >> 3, p = lane & 7;
  constexpr unsigned OOB = 0x80000000u;
  const long x_bytes = (M / ((long)H * W)) * (long)Hs * Ws * Cin * 2;
  const long w_bytes = (long)Cout * K * 2;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, (int)w_bytes, 0x00020000);

  // logical position L of the walk -> tile (the channel tiles of one pixel tile on one XCD, as in the product kernel)
  auto tile_of = [&](long L) -> long { return ((T & 7) == 0) ? (L & 7) * (T >> 3) + (L >> 3) : L; };

  int a_yx[W1_NA];
  unsigned a_base[W1_NA];
  unsigned b_base[10];
  auto coords = [&](long m0c, int n0c) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < W1_NA; ++j) {
      const int R = 8 * (4 * j + w) + r;
      const int c = p ^ ((R >> 1) & 7);
      const long m = m0c + R;
      if (m < M) {
        const long img = m / ((long)H * W);
        const int rem = (int)(m - img * (long)H * W);
        const int y = rem / W, x = rem - y * W;
        a_yx[j] = ((y * sd) << 16) | (x * sd);
        a_base[j] = (unsigned)((img * (long)Hs * Ws * Cin + c * 8) * 2);
      } else {
        a_yx[j] = 0x7ff07ff0;
        a_base[j] = 0;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int R = 8 * (4 * j + w) + r;
      const int c = p ^ ((R >> 1) & 7);
      b_base[j] = (n0c + R < Cout) ? (unsigned)(((long)(n0c + R) * K + c * 8) * 2) : OOB;
    }
  };
  auto dma = [&](int st, int kt, int i) __attribute__((always_inline)) {
    unsigned char* sbase = smem + st * STAGE;
    if (i < W1_NA) {
      const int tap = kt / cch, c0 = (kt - tap * cch) * W1_BK;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int yy = (a_yx[i] >> 16) + dy, xx = (a_yx[i] & 0xffff) + dx;
      const bool ok = (unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi;
      const unsigned off = a_base[i] + (unsigned)((((yy >> up) * Ws + (xx >> up)) * Cin + c0) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(sbase + (4 * i + w) * 1024), 16, ok ? off : OOB, 0, 0, 0);
    } else {
      const int j = i - W1_NA;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(sbase + W1_BM * W1_ROWB + (4 * j + w) * 1024), 16,
                                               b_base[j] == OOB ? OOB : b_base[j] + (unsigned)(kt * W1_BK * 2), 0, 0, 0);
    }
  };

  float4_t acc[10][8];
  const int key = (l16 >> 1) & 7;
  const int arow = (wm * 128 + l16) * W1_ROWB;
  const int brow = W1_BM * W1_ROWB + (wn * (BN / 2) + l16) * W1_ROWB;
  uint4_t xf[2][8], wf[2][10];
  auto frag = [&](int buf, const unsigned char* sb, int s, int f) __attribute__((always_inline)) {
    const int po = ((4 * s + lq) ^ key) << 4;
    if (f < 8) xf[buf][f] = *(const uint4_t*)(sb + arow + po + f * 16 * W1_ROWB);
    else wf[buf][f - 8] = *(const uint4_t*)(sb + brow + po + (f - 8) * 16 * W1_ROWB);
  };
  const int last = NK - 1;

  // ---- the first tile's first k-tile
  long L = blockIdx.x;
  long tile = tile_of(L);
  long m0 = (tile / ntiles) * W1_BM;
  int n0 = (int)(tile % ntiles) * BN;
  coords(m0, n0);
#pragma unroll
  for (int i = 0; i < PER; ++i) dma(0, 0, i);
  int slot = 0;

  for (; L < T; L += gridDim.x) {
    const bool has_next = L + gridDim.x < T;
    long m0n = 0;
    int n0n = 0;
    if (has_next) {
      const long tn = tile_of(L + gridDim.x);
      m0n = (tn / ntiles) * W1_BM;
      n0n = (int)(tn % ntiles) * BN;
    }
#pragma unroll
    for (int a = 0; a < TNW; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};
    // this tile's first k-tile (sent by the prologue or by the previous tile) has landed in `slot`
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < NFR; ++f) frag(0, smem + slot * STAGE, 0, f);

    // one k-tile; TAIL (compile time: the hot loop carries no run-time conditions) = the tile's last one
    auto ktile = [&](auto tail_c, int kt) __attribute__((always_inline)) {
      constexpr bool tail = decltype(tail_c)::value;
      const int fill = slot ^ 1;
      const unsigned char* sb = smem + slot * STAGE;
      if constexpr (tail) {
        if (has_next) coords(m0n, n0n);                // (the current tile sends nothing any more: its coordinates are free)
      }
      auto kstep = [&](auto s_c) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value;
        constexpr int cur = S, nxt = S ^ 1;
#pragma unroll
        for (int a = 0; a < TNW; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            if (a < 8) w1_mfma<F16, true>(acc[a][b], wf[cur][a], xf[cur][b]);
            else w1_mfma<F16, false>(acc[a][b], wf[cur][a], xf[cur][b]);
            const int i = a * 8 + b;
            if constexpr (S == 0) {
#pragma unroll
              for (int f = 0; f < NFR; ++f)
                if (i == ((f + 1) * NMF) / NFR - 1) {
                  frag(nxt, sb, 1, f);
                  __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
              for (int g = 0; g < PER; ++g)
                if (i == ((g + 1) * NMF) / PER - 2) {
                  if constexpr (!tail) dma(fill, kt + 1, g);
                  else if (has_next) dma(fill, 0, g);
                  __builtin_amdgcn_sched_barrier(0);
                }
            } else {
              constexpr int HOLD = NFR;
              if (i == NMF - HOLD - 1) {
                // the next tile's first k-tile stays in flight behind the tile's last barrier: it is waited for after the epilogue
                if constexpr (tail) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                // (compiler fence: the raw barrier does not order memory operations for the compiler - without it the epilogue's
                // slab writes of accumulators that are final early in this k-step are hoisted above the barrier, into a stage
                // another wave is still reading its second-step fragments from: 2 % of the outputs differed)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
              }
              if (i >= NMF - HOLD) {
                if constexpr (!tail) frag(nxt, smem + fill * STAGE, 0, i - (NMF - HOLD));
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
      };
      kstep(std::integral_constant<int, 0>{});
      kstep(std::integral_constant<int, 1>{});
      slot = fill;
    };
    for (int kt = 0; kt < last; ++kt) ktile(std::false_type{}, kt);
    ktile(std::true_type{}, last);
    // `slot` now names the stage the NEXT tile's first k-tile is landing in; the stage this tile's last k-tile used is the other one
    // and every wave is past its reads of it (the barrier inside the last k-step): its memory holds the epilogue's slabs
    constexpr int SROW = BN + 16;
    constexpr int CPR = BN / 16;
    constexpr int PB = 2;                              // 32-pixel slabs: 4 x 32 x SROW = 43 KB <= one stage
    unsigned char* slab = smem + (slot ^ 1) * STAGE + w * (16 * PB) * SROW;
    const int ncol0 = n0 + wn * (BN / 2);
    float bv[10][4];
#pragma unroll
    for (int a = 0; a < TNW; ++a) {
      bv[a][0] = bv[a][1] = bv[a][2] = bv[a][3] = 0.f;
      if (bias) {
        const uint2_t b2 = *(const uint2_t*)(bias + ncol0 + 16 * a + 4 * lq);
        bv[a][0] = w1_tof<F16>((unsigned short)(b2[0] & 0xffffu));
        bv[a][1] = w1_tof<F16>((unsigned short)(b2[0] >> 16));
        bv[a][2] = w1_tof<F16>((unsigned short)(b2[1] & 0xffffu));
        bv[a][3] = w1_tof<F16>((unsigned short)(b2[1] >> 16));
      }
    }
#pragma unroll
    for (int pb = 0; pb < 8 / PB; ++pb) {
#pragma unroll
      for (int bb = 0; bb < PB; ++bb) {
        const int b = pb * PB + bb;
#pragma unroll
        for (int a = 0; a < TNW; ++a)
          *(uint2_t*)(slab + (16 * bb + l16) * SROW + (16 * a + 4 * lq) * 2) =
              (uint2_t){w1_pack2<F16>(acc[a][b][0] + bv[a][0], acc[a][b][1] + bv[a][1]),
                        w1_pack2<F16>(acc[a][b][2] + bv[a][2], acc[a][b][3] + bv[a][3])};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long mrow0 = m0 + wm * 128 + 16 * PB * pb;
#pragma unroll
      for (int it = 0; it < (16 * PB * CPR) / 64; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / CPR, ch = idx - row * CPR;
        const long m = mrow0 + row;
        if (m < M) {
          uint4_t v = *(const uint4_t*)(slab + row * SROW + ch * 16);
          const int n = ncol0 + ch * 8;
          if (Rs) {
            const uint4_t r4 = *(const uint4_t*)(Rs + m * Cout + n);
            v = (uint4_t){w1_add2<F16>(v[0], r4[0]), w1_add2<F16>(v[1], r4[1]), w1_add2<F16>(v[2], r4[2]), w1_add2<F16>(v[3], r4[3])};
          }
          *(uint4_t*)(Y + m * Cout + n) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    m0 = m0n;
    n0 = n0n;
  }
}

unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace

int main(int argc, char** argv) {
  // N, H, W, Cin, Cout, stride, residual
  const int shapes[][7] = {{8, 16, 16, 64, 320, 1, 1},   {3, 20, 14, 192, 640, 1, 1},  {4, 32, 32, 320, 320, 2, 0},
                           {32, 64, 64, 320, 320, 1, 1}, {128, 64, 64, 320, 320, 1, 1}, {128, 32, 32, 640, 640, 1, 1},
                           {128, 16, 16, 1280, 1280, 1, 1}, {128, 64, 64, 960, 320, 1, 1}};
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  for (auto& sh : shapes) {
    const int N = sh[0], H = sh[1], W = sh[2], Cin = sh[3], Cout = sh[4], sd = sh[5], res = sh[6];
    const long M = (long)N * H * W;                                   // output pixels; the source image is (H sd) x (W sd)
    const size_t nx = (size_t)N * H * sd * W * sd * Cin, nw = (size_t)Cout * 9 * Cin, ny = (size_t)M * Cout;
    std::vector<unsigned short> hx(nx), hw(nw), hb(Cout), hr(res ? ny : 1);
    unsigned s = 2463534242u;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = f2bf(rnd());
    for (auto& v : hw) v = f2bf(rnd() * 0.05f);
    for (auto& v : hb) v = f2bf(rnd());
    for (auto& v : hr) v = f2bf(rnd());
    unsigned short *dx, *dw, *db, *dr, *dy0, *dy1;
    hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&db, Cout * 2); hipMalloc(&dr, hr.size() * 2);
    hipMalloc(&dy0, ny * 2); hipMalloc(&dy1, ny * 2);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), Cout * 2, hipMemcpyHostToDevice); hipMemcpy(dr, hr.data(), hr.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dy0, 0xff, ny * 2); hipMemset(dy1, 0xee, ny * 2);
    const long mtiles = (M + W1_BM - 1) / W1_BM;
    const int ntiles = Cout / 320;
    const long T = mtiles * ntiles;
    const unsigned grid = (unsigned)(T < ncu ? T : ncu);
    const size_t smem = w1_smem<10>();
    hipFuncSetAttribute((const void*)k_conv3x3_w1p<10, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto product = [&]() { int rc; launch_conv_w1(dx, dw, db, dy0, M, H, W, Cin, Cout, 0, UCE_DTYPE_BF16, 0, &rc, sd, res ? dr : nullptr, 2); };
    auto persistent = [&]() {
      k_conv3x3_w1p<10, false><<<grid, 256, smem>>>(dx, dw, db, dy1, M, H, W, Cin, Cout, 0, (int)mtiles, ntiles, sd, res ? dr : nullptr);
    };
    auto timed = [&](auto fn) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      fn(); fn();
      hipEventRecord(e0);
      const int iters = 5;
      for (int i = 0; i < iters; ++i) fn();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      return ms / iters * 1e3f;
    };
    const float t0 = timed(product), t1 = timed(persistent), t0b = timed(product);
    std::vector<unsigned short> y0(ny), y1(ny);
    hipMemcpy(y0.data(), dy0, ny * 2, hipMemcpyDeviceToHost); hipMemcpy(y1.data(), dy1, ny * 2, hipMemcpyDeviceToHost);
    size_t diff = 0;
    for (size_t i = 0; i < ny; ++i) diff += y0[i] != y1[i];
    printf("N=%d %dx%d %d->%d stride %d res %d: tiles %ld grid %u | product %.1f us, persistent %.1f us, product again %.1f us | differing outputs %zu of %zu (%s)\n",
           N, H, W, Cin, Cout, sd, res, T, grid, t0, t1, t0b, diff, ny, hipGetErrorString(hipGetLastError()));
    hipFree(dx); hipFree(dw); hipFree(db); hipFree(dr); hipFree(dy0); hipFree(dy1);
  }
  return 0;
}
