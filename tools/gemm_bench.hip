// Dev tool: time the GEMM kernels on the encoder's real shapes (random data).  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I vtc_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "gemm.hip.h"
#include "gemm256.hip.h"
using namespace vtc;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int M, N, K; int rows_per_batch; long long row_stride, batch_stride; int epi; };

template <int EPI> void run(const Shape& s, GemmParams p, int which, hipStream_t st) {
  if (which == 128) launch_gemm<true, 128, EPI>(p, 1, st);
  else if (which == 256) { if (p.N % 256 == 0) launch_gemm256<true, EPI, 256>(p, 1, st); }
  else if (which == 192) { if (p.N % 192 == 0) launch_gemm256<true, EPI, 192>(p, 1, st); }
  else if (which == 1281) launch_gemm<true, 128, EPI, 1>(p, 1, st);
  else if (which == 1282) launch_gemm<true, 128, EPI, 2>(p, 1, st);
  else if (which == 1283) launch_gemm<true, 128, EPI, 3>(p, 1, st);
  else if (which == 1284) launch_gemm<true, 128, EPI, 4>(p, 1, st);
  else if (which == 1286) launch_gemm<true, 128, EPI, 6>(p, 1, st);
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 10;
  const char* only = argc > 2 ? argv[2] : "";      // substring filter on the shape name
  const int only_tile = argc > 3 ? atoi(argv[3]) : 0;
  std::vector<Shape> shapes = {
      {"conv1  M=1.02M N=512  K=1536 (implicit)", 64 * 15999, 512, 1536, 15999, 1024, 31999LL * 512, EPI_GELU},
      {"conv1c M=1.02M N=512  K=1536 (implicit, A cache-resident: windows 1 row apart)", 64 * 15999, 512, 1536, 15999, 1024, 1024LL, EPI_GELU},
      {"conv2  M=512K  N=512  K=1536 (implicit)", 64 * 7999, 512, 1536, 7999, 1024, 15999LL * 512, EPI_GELU},
      {"conv5  M=64K   N=512  K=1024 (implicit)", 64 * 999, 512, 1024, 999, 1024, 1999LL * 512, EPI_GELU},
      {"qkv    M=31936 N=2304 K=768", 31936, 2304, 768, 31936, 768, 0, EPI_BIAS},
      {"oproj  M=31936 N=768  K=768  (+resid16)  ", 31936, 768, 768, 31936, 768, 0, EPI_BIAS | EPI_RESID16 | EPI_F32},
      {"ffn1   M=31936 N=3072 K=768  (gelu)", 31936, 3072, 768, 31936, 768, 0, EPI_BIAS | EPI_GELU},
      {"ffn1'  M=31936 N=3072 K=768  (bias only)", 31936, 3072, 768, 31936, 768, 0, EPI_BIAS},
      {"qkv'   M=31936 N=2304 K=768  (gelu)", 31936, 2304, 768, 31936, 768, 0, EPI_BIAS | EPI_GELU},
      {"ffn2   M=31936 N=768  K=3072 (+resid16)  ", 31936, 768, 3072, 31936, 3072, 0, EPI_BIAS | EPI_RESID16 | EPI_F32},
      {"square M=8192  N=8192 K=8192", 8192, 8192, 8192, 8192, 8192, 0, EPI_BIAS},
      {"sq4096 M=4096  N=4096 K=4096", 4096, 4096, 4096, 4096, 4096, 0, EPI_BIAS},
  };
  size_t maxA = 0, maxW = 0, maxC = 0;
  for (auto& s : shapes) {
    size_t a = s.batch_stride ? (size_t)(s.M / s.rows_per_batch) * s.batch_stride + s.K : (size_t)s.M * s.K;
    maxA = std::max(maxA, a); maxW = std::max(maxW, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N);
  }
  h16 *A, *W; float *C, *R, *bias;
  CK(hipMalloc(&A, maxA * 2 + 4096)); CK(hipMalloc(&W, maxW * 2)); CK(hipMalloc(&C, maxC * 4)); CK(hipMalloc(&R, maxC * 4));
  CK(hipMalloc(&bias, 8192 * 4));
  {
    std::vector<unsigned short> h(std::max(maxA, maxW));
    srand(1);
    for (auto& v : h) { _Float16 x = (_Float16)((rand() % 2001 - 1000) / 1000.0f); memcpy(&v, &x, 2); }
    CK(hipMemcpy(A, h.data(), maxA * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), maxW * 2, hipMemcpyHostToDevice));
    CK(hipMemset(R, 0, maxC * 4)); CK(hipMemset(bias, 0, 8192 * 4));
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& s : shapes) {
    if (!strstr(s.name, only)) continue;
    GemmParams p{};
    p.A = A; p.W = W; p.C = C; p.bias = bias; p.resid = R; p.M = s.M; p.N = s.N; p.K = s.K; p.ldc = s.N;
    p.rows_per_batch = s.rows_per_batch; p.a_row_stride = s.row_stride; p.a_batch_stride = s.batch_stride;
    for (int which : {128, 256, 192}) {
      if (only_tile && which != only_tile) continue;
      auto go = [&]() {
        switch (s.epi) {
          case EPI_GELU: run<EPI_GELU>(s, p, which, st); break;
          case EPI_BIAS: run<EPI_BIAS>(s, p, which, st); break;
          case EPI_BIAS | EPI_GELU: run<EPI_BIAS | EPI_GELU>(s, p, which, st); break;
          default: run<EPI_BIAS | EPI_RESID16 | EPI_F32>(s, p, which, st); break;
        }
      };
      go(); go();
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) go();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      printf("%-46s tile %4d  %8.3f ms  %7.1f TFLOP/s\n", s.name, which, ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
#ifdef VTC_STAMPS
      if (which != 128) {
        std::vector<long long> h(256 * 8);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8));
        double d[7] = {0, 0, 0, 0, 0, 0, 0}; int n = 0;
        for (int b = 0; b < 256; ++b) {
          const long long* t = &h[b * 8];
          if (!t[7]) continue;   // block ran fewer than two tiles
          for (int k = 0; k < 7; ++k) d[k] += (double)(t[k + 1] - t[k]);
          ++n;
        }
        if (n) printf("    stamps (cycles, mean of %d blocks): loop0 %.0f  prologue-issue %.0f  epilogue %.0f  wait+barrier %.0f  loop1 %.0f  "
                      "prologue-issue %.0f  epilogue %.0f\n", n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n, d[6] / n);
      }
#endif
    }
  }
  return 0;
}
