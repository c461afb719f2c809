// Dev tool: issue rate of v_mfma_f32_16x16x32_f16 by number of independent accumulators (distance between dependent
// MFMAs) and waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vtc_amd/csrc tools/mfma_bench.hip -o tools/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "common.hip.h"
using namespace vtc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ void k(float* out, int rounds, long long* cyc) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u16x8 a[4], b[2];
  for (int t = 0; t < 4; ++t) for (int i = 0; i < 8; ++i) a[t][i] = (unsigned short)(0x3c00 + lane + t);
  for (int t = 0; t < 2; ++t) for (int i = 0; i < 8; ++i) b[t][i] = (unsigned short)(0x3800 + i + t);
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = mfma16<true>(b[i & 1], a[i & 3], acc[i]);
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC> void run(float* out, long long* cyc, int threads) {
  const int rounds = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(threads), 0, 0, out, rounds, cyc);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(threads), 0, 0, out, rounds, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[256]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
  double m = 0; for (auto x : h) m += x; m /= 256;
  const double n = rounds * 16.0 * (threads / 256);   // MFMAs per SIMD
  printf("%d accumulators, %d waves per SIMD: %.2f ns and %.1f s_memtime ticks per MFMA and SIMD\n", NACC, threads / 256, ms * 1e6 / n, m / n);
}

int main() {
  float* out; long long* cyc; CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  for (int threads : {256, 512}) { run<16>(out, cyc, threads); run<8>(out, cyc, threads); run<4>(out, cyc, threads); run<2>(out, cyc, threads); }
  return 0;
}
