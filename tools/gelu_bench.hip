// Dev tool: VALU cost of GELU variants (cycles per wave and element with 2 waves per SIMD, as in the GEMM epilogues).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I vtc_amd/csrc tools/gelu_bench.hip -o tools/gelu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "common.hip.h"
using namespace vtc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 gelu_pk(f32x2 x) {   // gelu_f on two elements with packed fp32 arithmetic
  const f32x2 s = x * x;
  f32x2 p = __builtin_elementwise_fma(f32x2{-3.2290622584696393e-06f, -3.2290622584696393e-06f}, s, f32x2{8.823935058899224e-05f, 8.823935058899224e-05f});
  p = __builtin_elementwise_fma(p, s, f32x2{0.0003602687211241573f, 0.0003602687211241573f});
  p = __builtin_elementwise_fma(p, s, f32x2{-0.10522668063640594f, -0.10522668063640594f});
  p = __builtin_elementwise_fma(p, s, f32x2{-2.3020453453063965f, -2.3020453453063965f});
  const f32x2 z = x * p;
  const f32x2 e = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
  const f32x2 d = e + f32x2{1.f, 1.f};
  const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return x * r;
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, long long* cyc) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (threadIdx.x * 16 + i) * 1e-4f - 0.3f;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = gelu_f(v[i]) + 0.25f;
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) { f32x2 r = gelu_pk(f32x2{v[i], v[i + 1]}); v[i] = r[0] + 0.25f; v[i + 1] = r[1] + 0.25f; }
    } else if (MODE == 2) {       // transcendental only
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[i]));
    } else {                      // 8 dependent fma only
#pragma unroll
      for (int i = 0; i < 16; ++i) { float p = v[i]; p = fmaf(p, p, 0.1f); p = fmaf(p, v[i], 0.2f); p = fmaf(p, v[i], 0.3f); p = fmaf(p, v[i], 0.1f);
        p = fmaf(p, v[i], 0.2f); p = fmaf(p, v[i], 0.3f); p = fmaf(p, v[i], 0.1f); v[i] = fmaf(p, 0.001f, 0.2f); }
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
  CK(hipDeviceSynchronize());
  long long h[256]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
  double m = 0; for (auto x : h) m += x; m /= 256;
  // s_memtime ticks at 100 MHz; report per wave and element using the wall time of the kernel instead
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, iters, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s %8.3f ms  -> %.2f ns per wave-element (2 waves per SIMD share it), memtime ticks %.0f\n", name, ms, ms * 1e6 / (iters * 16.0 * 2), m);
}

int main() {
  float* out; long long* cyc; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  run<0>("gelu_f scalar", out, cyc); run<1>("gelu packed fp32", out, cyc); run<2>("exp2 + add + rcp", out, cyc); run<3>("8 fma", out, cyc);
  // accuracy of the packed form against erf
  return 0;
}
