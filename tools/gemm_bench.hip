// Dev tool: time the GEMM kernels on the encoder's real shapes (random data).  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I vtc_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <vector>
#include "gemm.hip.h"
#include "gemm256.hip.h"
using namespace vtc;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int M, N, K; int rows_per_batch; long long row_stride, batch_stride; int epi; };

template <int EPI> void run(const Shape& s, GemmParams p, int which, hipStream_t st) {
  if constexpr (EPI & (EPI_LNRES | EPI_LNIN)) {
    if (which == 192) launch_gemm256<true, EPI, 192>(p, 1, st);
    if constexpr (!(EPI & EPI_LNRES)) { if (which == 256 && p.N % 256 == 0) launch_gemm256<true, EPI, 256>(p, 1, st); }
  } else {
    if (which == 128) launch_gemm<true, 128, EPI>(p, 1, st);
    else if (which == 256) { if (p.N % 256 == 0) launch_gemm256<true, EPI, 256>(p, 1, st); }
    else if (which == 192) { if (p.N % 192 == 0) launch_gemm256<true, EPI, 192>(p, 1, st); }
    else if (which == 1281) launch_gemm<true, 128, EPI, 1>(p, 1, st);
    else if (which == 1282) launch_gemm<true, 128, EPI, 2>(p, 1, st);
    else if (which == 1283) launch_gemm<true, 128, EPI, 3>(p, 1, st);
    else if (which == 1284) launch_gemm<true, 128, EPI, 4>(p, 1, st);
    else if (which == 1286) launch_gemm<true, 128, EPI, 6>(p, 1, st);
  }
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
      {"qkv0   M=31936 N=2304 K=768  (no bias)", 31936, 2304, 768, 31936, 768, 0, 0},
      {"oproj  M=31936 N=768  K=768  (+resid16)  ", 31936, 768, 768, 31936, 768, 0, EPI_BIAS | EPI_RESID16 | EPI_F32},
      {"qkvc   M=31936 N=2304 K=768  (A cache-resident: rows 16 B apart)", 31936, 2304, 768, 31936, 8, 0, EPI_BIAS},
      {"ffn2c  M=31936 N=768  K=3072 (+resid16, A cache-resident: rows 16 B apart)", 31936, 768, 3072, 31936, 8, 0, EPI_BIAS | EPI_RESID16 | EPI_F32},
      {"oprojL M=31936 N=768  K=768  (LayerNorm residual)", 31936, 768, 768, 31936, 768, 0, EPI_BIAS | EPI_LNRES},
      {"ffn2L  M=31936 N=768  K=3072 (LayerNorm residual)", 31936, 768, 3072, 31936, 3072, 0, EPI_BIAS | EPI_LNRES},
      {"qkvL   M=31936 N=2304 K=768  (LayerNorm in)", 31936, 2304, 768, 31936, 768, 0, EPI_BIAS | EPI_LNIN},
      {"ffn1L  M=31936 N=3072 K=768  (LayerNorm in, gelu)", 31936, 3072, 768, 31936, 768, 0, EPI_BIAS | EPI_GELU | EPI_LNIN},
      {"ffn1   M=31936 N=3072 K=768  (gelu)", 31936, 3072, 768, 31936, 768, 0, EPI_BIAS | EPI_GELU},
      {"ffn1'  M=31936 N=3072 K=768  (bias only)", 31936, 3072, 768, 31936, 768, 0, EPI_BIAS},
      {"qkv'   M=31936 N=2304 K=768  (gelu)", 31936, 2304, 768, 31936, 768, 0, EPI_BIAS | EPI_GELU},
      {"ffn2   M=31936 N=768  K=3072 (+resid16)  ", 31936, 768, 3072, 31936, 3072, 0, EPI_BIAS | EPI_RESID16 | EPI_F32},
      {"square M=8192  N=8192 K=8192", 8192, 8192, 8192, 8192, 8192, 0, EPI_BIAS},
      {"sq4096 M=4096  N=4096 K=4096", 4096, 4096, 4096, 4096, 4096, 0, EPI_BIAS},
  };
  size_t maxA = 0, maxW = 0, maxC = 0;
  for (auto& s : shapes) {
    if (!strstr(s.name, only)) continue;     // buffers sized for the shapes that will run (start-up is host-side random data)
    size_t a = s.batch_stride ? (size_t)(s.M / s.rows_per_batch) * s.batch_stride + s.K : (size_t)s.M * s.K;
    maxA = std::max(maxA, a); maxW = std::max(maxW, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N);
  }
  h16 *A, *W; float *C, *R, *bias;
  CK(hipMalloc(&A, maxA * 2 + 4096)); CK(hipMalloc(&W, maxW * 2)); CK(hipMalloc(&C, maxC * 4)); CK(hipMalloc(&R, maxC * 4));
  CK(hipMalloc(&bias, 8192 * 4));
  float *lnstats, *lnpart; CK(hipMalloc(&lnstats, 32768 * 8)); CK(hipMalloc(&lnpart, 32768 * 128));
  CK(hipMemset(lnstats, 0, 32768 * 8)); CK(hipMemset(lnpart, 0, 32768 * 128));
  std::vector<unsigned short> hA(std::max(maxA, maxW));   // A and W are the same random values (W = its first N*K)
  {
    srand(1);
    for (auto& v : hA) { _Float16 x = (_Float16)((rand() % 2001 - 1000) / 1000.0f / 16.0f); memcpy(&v, &x, 2); }
    CK(hipMemcpy(A, hA.data(), maxA * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hA.data(), maxW * 2, hipMemcpyHostToDevice));
    CK(hipMemset(R, 0, maxC * 4)); CK(hipMemset(bias, 0, 8192 * 4));
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& s : shapes) {
    if (!strstr(s.name, only)) continue;
    GemmParams p{};
    p.A = A; p.W = W; p.C = C; p.bias = bias; p.resid = R; p.M = s.M; p.N = s.N; p.K = s.K; p.ldc = s.N;
    p.rows_per_batch = s.rows_per_batch; p.a_row_stride = s.row_stride; p.a_batch_stride = s.batch_stride;
    p.ln_stats = lnstats; p.colsum = bias; p.res_h = (const h16*)R; p.res_g = bias; p.part = lnpart; p.part_rows = s.M;
    if ((s.epi & (EPI_LNRES | EPI_LNIN)) && s.M > 32768) continue;
    for (int which : {128, 256, 192}) {
      if (only_tile && which != only_tile) continue;
      auto go = [&]() {
        switch (s.epi) {
          case 0: run<0>(s, p, which, st); break;
          case EPI_GELU: run<EPI_GELU>(s, p, which, st); break;
          case EPI_BIAS: run<EPI_BIAS>(s, p, which, st); break;
          case EPI_BIAS | EPI_GELU: run<EPI_BIAS | EPI_GELU>(s, p, which, st); break;
          case EPI_BIAS | EPI_LNRES: run<EPI_BIAS | EPI_LNRES>(s, p, which, st); break;
          case EPI_BIAS | EPI_LNIN: run<EPI_BIAS | EPI_LNIN>(s, p, which, st); break;
          case EPI_BIAS | EPI_GELU | EPI_LNIN: run<EPI_BIAS | EPI_GELU | EPI_LNIN>(s, p, which, st); break;
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
      if (getenv("VTC_CHECK") && !(s.epi & EPI_F32) && which != 128 && (which != 256 || s.N % 256 == 0)) {
        // sampled check of the 16-bit output against host dot products (bias and residual are zero here)
        std::vector<unsigned short> hc((size_t)s.M * s.N);
        CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
        auto f16 = [](unsigned short u) { _Float16 x; memcpy(&x, &u, 2); return (double)x; };
        double worst = 0; int bad = 0;
        srand(7);
        for (int t = 0; t < 20000; ++t) {
          int m = t < 64 ? s.M - 1 - t : rand() % s.M, n = t < 64 ? s.N - 1 - (t % s.N) : rand() % s.N;
          if (t >= 64 && t < 4096) { m = (rand() % ((s.M + 255) / 256)) * 256 + (t % 256); if (m >= s.M) m = s.M - 1; n = (t * 7) % s.N; }
          const int b = m / s.rows_per_batch, rr = m - b * s.rows_per_batch;
          const size_t a0 = (size_t)b * s.batch_stride + (size_t)rr * s.row_stride;
          double acc = 0;
          for (int k = 0; k < s.K; ++k) acc += f16(hA[a0 + k]) * f16(hA[(size_t)n * s.K + k]);
          if (s.epi & EPI_GELU) acc = 0.5 * acc * (1.0 + erf(acc / sqrt(2.0)));
          const double got = f16(hc[(size_t)m * s.N + n]), err = fabs(got - acc);
          worst = std::max(worst, err);
          if (err > 0.02 + 0.004 * fabs(acc)) { if (bad < 5) printf("    MISMATCH m=%d n=%d want %.4f got %.4f\n", m, n, acc, got); ++bad; }
        }
        printf("    check: worst |err| %.4f, %d of 20000 samples out of tolerance\n", worst, bad);
      }
#ifdef VTC_STAMPS
      if (which != 128) {
        std::vector<long long> h(256 * 16);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8));
        double loop0 = 0, epi0 = 0, loop1 = 0, epi1 = 0, ld = 0, ar = 0, fl = 0; int n = 0;
        for (int b = 0; b < 256; ++b) {
          const long long* t = &h[b * 16];
          if (!t[11]) continue;   // block ran fewer than two tiles
          loop0 += (double)(t[1] - t[0]); epi0 += (double)(t[3] - t[2]); loop1 += (double)(t[9] - t[8]); epi1 += (double)(t[11] - t[10]);
          if (t[12]) ld += (double)(t[12] - t[10]);
          if (t[13]) { ar += (double)(t[13] - (t[12] ? t[12] : t[10])); fl += (double)(t[11] - t[13]); }
          ++n;
        }
        if (n) printf("    stamps (ticks, mean of %d blocks): loop0 %.0f  epilogue %.0f  loop1 %.0f  epilogue %.0f  [second tile: operand loads %.0f, "
                      "arithmetic %.0f, output %.0f]\n", n, loop0 / n, epi0 / n, loop1 / n, epi1 / n, ld / n, ar / n, fl / n);
      }
#endif
    }
  }
  return 0;
}
