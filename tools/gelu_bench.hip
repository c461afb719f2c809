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

// GELU by table: Phi(x) from a 1536-entry (value, slope) table in LDS, first-order around the NEAREST node
// (|err| of x * Phi <= h^2 / 8 * max |x^2 phi(x)| = 2.2e-6 at h = 12 / 1535); 4 - 4.5 VALU slots per element instead of 12
// (no transcendental), plus one 8-byte LDS gather per element.
constexpr int kGeluNodes = 1536;
template <bool SAT>
__device__ __forceinline__ f32x2 gelu_tab_pk(f32x2 x, const char* tab) {
  f32x2 u;
  if constexpr (SAT) {   // u = sat(x / 12 + 0.5): the clamp output modifier of the packed fma
    const f32x2 k1 = {1.f / 12.f, 1.f / 12.f}, k2 = {0.5f, 0.5f};
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(u) : "v"(x), "v"(k1), "v"(k2));
  } else {
    u = __builtin_elementwise_fma(x, f32x2{1.f / 12.f, 1.f / 12.f}, f32x2{0.5f, 0.5f});
    u[0] = __builtin_amdgcn_fmed3f(u[0], 0.f, 1.f); u[1] = __builtin_amdgcn_fmed3f(u[1], 0.f, 1.f);
  }
  const f32x2 big = {8388608.f, 8388608.f}, sc = {(float)(kGeluNodes - 1), (float)(kGeluNodes - 1)};
  const f32x2 t = __builtin_elementwise_fma(u, sc, big);          // 2^23 + round(u * 1535)
  const f32x2 r = t - big;
  const f32x2 f = __builtin_elementwise_fma(u, sc, -r);           // in [-0.5, 0.5]
  const unsigned a0 = (__float_as_uint(t[0]) << 3) + 0xA8000000u;  // 8 * node  (0x4B000000 << 3 wraps to 0x58000000)
  const unsigned a1 = (__float_as_uint(t[1]) << 3) + 0xA8000000u;
  const float2 e0 = *(const float2*)(tab + a0), e1 = *(const float2*)(tab + a1);
  const f32x2 phi = __builtin_elementwise_fma(f, f32x2{e0.y, e1.y}, f32x2{e0.x, e1.x});
  return x * phi;
}
__global__ void gelu_tab_check(const float* x, float* y, int n, const float2* tab_g, int sat) {
  __shared__ float2 tab[kGeluNodes];
  for (int i = threadIdx.x; i < kGeluNodes; i += blockDim.x) tab[i] = tab_g[i];
  __syncthreads();
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) {
    const f32x2 r = sat ? gelu_tab_pk<true>(f32x2{x[i], x[i + 1]}, (const char*)tab) : gelu_tab_pk<false>(f32x2{x[i], x[i + 1]}, (const char*)tab);
    y[i] = r[0]; y[i + 1] = r[1];
  }
}
__device__ const float2* g_tab;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, long long* cyc) {
  __shared__ float2 tab[(MODE >= 4 && MODE != 6) ? kGeluNodes : 1];
  if (MODE >= 4 && MODE != 6) {
    for (int i = threadIdx.x; i < kGeluNodes; i += 512) tab[i] = g_tab[i];
    __syncthreads();
  }
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
    } else if (MODE == 4 || MODE == 5) {   // table in LDS
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f32x2 r = MODE == 4 ? gelu_tab_pk<true>(f32x2{v[i], v[i + 1]}, (const char*)tab) : gelu_tab_pk<false>(f32x2{v[i], v[i + 1]}, (const char*)tab);
        v[i] = r[0] * 3.1f - 1.1f; v[i + 1] = r[1] * 2.3f - 0.7f;      // keeps the lanes on different, moving table entries
      }
    } else if (MODE >= 7 && MODE <= 9) {   // mixed: MODE 7: 8 of 16 by table, MODE 8: 6 of 16, MODE 9: 4 of 16; the rest by exp2 + rcp
      constexpr int NT = MODE == 7 ? 8 : MODE == 8 ? 6 : 4;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f32x2 r = i < NT ? gelu_tab_pk<true>(f32x2{v[i], v[i + 1]}, (const char*)tab) : gelu_pk(f32x2{v[i], v[i + 1]});
        v[i] = r[0] * 3.1f - 1.1f; v[i + 1] = r[1] * 2.3f - 0.7f;
      }
    } else if (MODE == 6) {       // packed form with the same feedback as the table modes (like for like)
#pragma unroll
      for (int i = 0; i < 16; i += 2) { f32x2 r = gelu_pk(f32x2{v[i], v[i + 1]}); v[i] = r[0] * 3.1f - 1.1f; v[i + 1] = r[1] * 2.3f - 0.7f; }
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
  // the table: node i <-> x_i = (i / 1535 - 0.5) * 12; value Phi(x_i), slope Phi'(x_i) * 12 / 1535 per unit of f; the end nodes
  // are exactly 0 and 1 (x <= -6 -> -0, x >= 6 -> x)
  std::vector<float> tab(2 * kGeluNodes);
  for (int i = 0; i < kGeluNodes; ++i) {
    const double x = ((double)i / (kGeluNodes - 1) - 0.5) * 12.0;
    tab[2 * i] = (float)(0.5 * erfc(-x / sqrt(2.0)));
    tab[2 * i + 1] = (float)(exp(-0.5 * x * x) / sqrt(2.0 * M_PI) * 12.0 / (kGeluNodes - 1));
  }
  tab[0] = 0.f; tab[1] = 0.f; tab[2 * kGeluNodes - 2] = 1.f; tab[2 * kGeluNodes - 1] = 0.f;
  float2* tab_d; CK(hipMalloc(&tab_d, tab.size() * 4)); CK(hipMemcpy(tab_d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), &tab_d, sizeof tab_d));
  run<0>("gelu_f scalar", out, cyc); run<1>("gelu packed fp32", out, cyc); run<2>("exp2 + add + rcp", out, cyc); run<3>("8 fma", out, cyc);
  run<6>("gelu packed, moving inputs", out, cyc); run<4>("gelu table (sat modifier)", out, cyc); run<5>("gelu table (med3)", out, cyc);
  run<7>("mixed: 8 of 16 by table", out, cyc); run<8>("mixed: 6 of 16 by table", out, cyc); run<9>("mixed: 4 of 16 by table", out, cyc);
  // accuracy of the table forms against erf, |x| <= 8 and a few far values
  const int n = 1 << 22;
  std::vector<float> hx(n), hy(n);
  for (int i = 0; i < n; ++i) hx[i] = -8.f + 16.f * (float)i / (float)(n - 1);
  hx[0] = -1e30f; hx[1] = 1e30f; hx[2] = -65504.f; hx[3] = 65504.f; hx[4] = 0.f; hx[5] = -0.f;
  float *dx, *dy; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice));
  for (int sat = 1; sat >= 0; --sat) {
    hipLaunchKernelGGL(gelu_tab_check, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dy, n, tab_d, sat);
    CK(hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost));
    double worst = 0, at = 0;
    for (int i = 0; i < n; ++i) {
      const double x = hx[i], want = fabs(x) > 40 ? (x > 0 ? x : 0.0) : 0.5 * x * erfc(-x / sqrt(2.0));
      const double err = fabs((double)hy[i] - want) / (fabs(x) > 40 ? fabs(x) : 1.0);
      if (!(err <= worst)) { worst = err; at = x; }
    }
    printf("table GELU (%s): max |err| %.3e at x = %g;  f(-1e30) = %g, f(1e30) = %g, f(0) = %g\n", sat ? "sat modifier" : "med3", worst, at,
           hy[0], hy[1], hy[4]);
  }
  return 0;
}
