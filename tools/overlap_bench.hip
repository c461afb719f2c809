// Dev tool: do MFMA and VALU / transcendental work of DIFFERENT waves on one SIMD overlap?  Every wave runs rounds of
// NM independent 32x32x16 MFMAs followed by a softmax-like VALU block (sub, exp2, add per value; 32 values); waves start
// their rounds out of phase.  Times for MFMA only, VALU only and both, at 1, 2 and 4 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vtc_amd/csrc tools/overlap_bench.hip -o tools/overlap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "common.hip.h"
using namespace vtc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 1 = MFMA, 2 = VALU, 3 = both
__global__ void k(float* out, int rounds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  u16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (unsigned short)(0x3c00 + lane); b[i] = (unsigned short)(0x3800 + i); }
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = (lane + i) * 1e-3f;
  float m = 0.25f, l = 0.f;
  if (MODE == 3 && (wave & 1)) {   // odd waves start with the VALU block: out of phase with their SIMD partners
    for (int i = 0; i < 32; ++i) { v[i] = __builtin_amdgcn_exp2f(v[i] - m); l += v[i]; }
  }
  for (int r = 0; r < rounds; ++r) {
    if (MODE & 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc[0] = mfma32<true>(a, b, acc[0]); acc[1] = mfma32<true>(b, a, acc[1]); }
    }
    if (MODE & 2) {
#pragma unroll
      for (int i = 0; i < 32; ++i) { v[i] = __builtin_amdgcn_exp2f(v[i] - m); l += v[i]; }
      m += 1e-6f;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = l;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> float run(float* out, int threads, int rounds) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, rounds);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, rounds);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 1024 * 4));
  const int rounds = 4000;
  for (int threads : {256, 512, 1024}) {
    const float a = run<1>(out, threads, rounds), b = run<2>(out, threads, rounds), c = run<3>(out, threads, rounds);
    const double per = 1e6 / rounds;   // ns per round
    printf("%d waves per SIMD: per round (16 MFMA 32x32x16 | 32 x {sub, exp2, add}) MFMA only %.0f ns, VALU only %.0f ns, both %.0f ns  "
           "(sum %.0f, max %.0f)\n", threads / 256, a * per, b * per, c * per, (a + b) * per, (a > b ? a : b) * per);
  }
  return 0;
}
