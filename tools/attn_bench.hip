// Dev tool: time attention_kernel on the encoder's shape (64 windows x 12 heads, T = 499) and, when built with
// -DVTC_ATTN_STAMPS, print where a wave's time goes per 64-key tile.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DVTC_ATTN_STAMPS] -I vtc_amd/csrc -I tools tools/attn_bench.hip -o tools/attn_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <vector>
#include "attention.hip.h"
#include "attention64.hip.h"
#include "attention_persistent.hip.h"
#include "attention_pp.hip.h"
#include "attention_b2.hip.h"
using namespace vtc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int nw = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 499, iters = 20;
  const int extra_lds = argc > 3 ? atoi(argv[3]) : 0;   // dynamic LDS nobody uses: > 16 KiB leaves one block per CU
  const long long M = (long long)nw * T;
  h16 *qkv, *ctx;
  CK(hipMalloc(&qkv, M * 2304 * 2)); CK(hipMalloc(&ctx, M * 768 * 2));
  {
    std::vector<unsigned short> h(M * 2304);
    srand(3);
    for (auto& v : h) { _Float16 x = (_Float16)((rand() % 2001 - 1000) / 1000.0f); memcpy(&v, &x, 2); }
    CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  }
  const int nkt = (T + 63) / 64, nqb = ((T + 31) / 32 + kAttnWaves - 1) / kAttnWaves;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto go = [&]() { hipLaunchKernelGGL((attention_kernel<true>), dim3(nqb, nw * 12), dim3(512), extra_lds, 0, qkv, ctx, T, nkt); };
  go(); go(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) go();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
  printf("attention nw=%d T=%d: %.1f us  %.1f TFLOP/s\n", nw, T, ms * 1e3, 4.0 * nw * 12 * (double)T * T * 64 / ms / 1e9);
  {   // the 64-queries-per-wave kernel on the same input: time, and its context rows against the kernel above
    std::vector<unsigned short> ref((size_t)M * 768), got((size_t)M * 768);
    CK(hipMemcpy(ref.data(), ctx, ref.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemset(ctx, 0, M * 768 * 2));
    const int nqb64 = ((T + 63) / 64 + kAttn64Waves - 1) / kAttn64Waves;
    auto go64 = [&]() { hipLaunchKernelGGL((attention64_kernel<true>), dim3(nqb64, nw * 12), dim3(256), extra_lds, 0, qkv, ctx, T, nkt); };
    go64(); go64(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) go64();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms64; CK(hipEventElapsedTime(&ms64, e0, e1)); ms64 /= iters;
    CK(hipMemcpy(got.data(), ctx, got.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0; size_t ndiff = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
      _Float16 a, c; memcpy(&a, &ref[i], 2); memcpy(&c, &got[i], 2);
      const double d = fabs((double)a - (double)c);
      if (d > worst) worst = d;
      ndiff += ref[i] != got[i];
    }
    {
      CK(hipMemset(ctx, 0, M * 768 * 2));
      const int n_items = nqb * nw * 12;
      const int grid = n_items < 512 ? n_items : 512;
      auto gop = [&]() { hipLaunchKernelGGL((attention_persistent_kernel<true>), dim3(grid), dim3(512), extra_lds, 0, qkv, ctx, T, nkt, nqb, n_items); };
      gop(); gop(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) gop();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float msp; CK(hipEventElapsedTime(&msp, e0, e1)); msp /= iters;
      std::vector<unsigned short> gp((size_t)M * 768);
      CK(hipMemcpy(gp.data(), ctx, gp.size() * 2, hipMemcpyDeviceToHost));
      size_t nd = 0; double w = 0;
      for (size_t i = 0; i < ref.size(); ++i) {
        if (ref[i] != gp[i]) { ++nd; _Float16 a, c; memcpy(&a, &ref[i], 2); memcpy(&c, &gp[i], 2); const double d = fabs((double)a - (double)c); if (d > w) w = d; }
      }
      printf("attention (persistent) nw=%d T=%d: %.1f us  %.1f TFLOP/s; vs attention_kernel: %zu of %zu elements differ, worst |d| %.3g\n", nw, T,
             msp * 1e3, 4.0 * nw * 12 * (double)T * T * 64 / msp / 1e9, nd, ref.size(), w);
    }
    {
      CK(hipMemset(ctx, 0, M * 768 * 2));
      auto gop = [&]() { hipLaunchKernelGGL((attention_b2_kernel<true>), dim3(nqb, nw * 12), dim3(512), extra_lds, 0, qkv, ctx, T, nkt); };
      gop(); gop(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) gop();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float msp; CK(hipEventElapsedTime(&msp, e0, e1)); msp /= iters;
      std::vector<unsigned short> gp((size_t)M * 768);
      CK(hipMemcpy(gp.data(), ctx, gp.size() * 2, hipMemcpyDeviceToHost));
      size_t nd = 0;
      for (size_t i = 0; i < ref.size(); ++i) nd += ref[i] != gp[i];
      printf("attention (barrier every 2 tiles) nw=%d T=%d: %.1f us  %.1f TFLOP/s; vs attention_kernel: %zu of %zu elements differ\n",
             nw, T, msp * 1e3, 4.0 * nw * 12 * (double)T * T * 64 / msp / 1e9, nd, ref.size());
    }
    for (int prio = 0; prio < 2; ++prio) {
      CK(hipMemset(ctx, 0, M * 768 * 2));
      auto gop = [&]() {
        if (prio) hipLaunchKernelGGL((attention_pp_kernel<true, true>), dim3(nqb, nw * 12), dim3(512), extra_lds, 0, qkv, ctx, T, nkt);
        else      hipLaunchKernelGGL((attention_pp_kernel<true, false>), dim3(nqb, nw * 12), dim3(512), extra_lds, 0, qkv, ctx, T, nkt);
      };
      gop(); gop(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) gop();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float msp; CK(hipEventElapsedTime(&msp, e0, e1)); msp /= iters;
      std::vector<unsigned short> gp((size_t)M * 768);
      CK(hipMemcpy(gp.data(), ctx, gp.size() * 2, hipMemcpyDeviceToHost));
      size_t nd = 0; double w = 0;
      for (size_t i = 0; i < ref.size(); ++i) {
        if (ref[i] != gp[i]) { ++nd; _Float16 a, c; memcpy(&a, &ref[i], 2); memcpy(&c, &gp[i], 2); const double d = fabs((double)a - (double)c); if (d > w) w = d; }
      }
      printf("attention (ping-pong%s) nw=%d T=%d: %.1f us  %.1f TFLOP/s; vs attention_kernel: %zu of %zu elements differ, worst |d| %.3g\n",
             prio ? ", setprio" : "", nw, T, msp * 1e3, 4.0 * nw * 12 * (double)T * T * 64 / msp / 1e9, nd, ref.size(), w);
    }
    printf("attention64 nw=%d T=%d: %.1f us  %.1f TFLOP/s; vs attention_kernel: %zu of %zu elements differ, worst |d| %.3g\n", nw, T,
           ms64 * 1e3, 4.0 * nw * 12 * (double)T * T * 64 / ms64 / 1e9, ndiff, ref.size(), worst);
  }
#ifdef VTC_ATTN_STAMPS
  std::vector<long long> h(64 * 8 * 10 * 5);
  CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_attn_stamps), h.size() * 8));
  double d[5] = {0, 0, 0, 0, 0}; int n = 0;
  for (int b = 0; b < 64; ++b) for (int w = 0; w < 8; ++w) for (int kt = 1; kt < nkt && kt < 10; ++kt) {
    const long long* t = &h[((b * 8 + w) * 10 + kt) * 5];
    const long long* tp = &h[((b * 8 + w) * 10 + kt - 1) * 5];
    if (!t[4] || !tp[4]) continue;
    d[0] += (double)(t[0] - tp[4]); d[1] += (double)(t[1] - t[0]); d[2] += (double)(t[2] - t[1]);
    d[3] += (double)(t[3] - t[2]); d[4] += (double)(t[4] - t[3]); ++n;
  }
  if (n) printf("  per tile and wave (cycles, mean of %d): loop-back %.0f  wait+barrier+stage %.0f  QK %.0f  softmax %.0f  PV %.0f\n", n,
                d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n);
  {
    const int nb = nqb * nw * 12 < 4096 ? nqb * nw * 12 : 4096;
    std::vector<long long> hb(4096 * 4);
    CK(hipMemcpyFromSymbol(hb.data(), HIP_SYMBOL(g_attn_blk), hb.size() * 8));
    long long t0 = hb[0], t1 = 0;
    double pro = 0, loop = 0, epi = 0;
    for (int b = 0; b < nb; ++b) { t0 = std::min(t0, hb[b * 4]); t1 = std::max(t1, hb[b * 4 + 3]);
      pro += hb[b * 4 + 1] - hb[b * 4]; loop += hb[b * 4 + 2] - hb[b * 4 + 1]; epi += hb[b * 4 + 3] - hb[b * 4 + 2]; }
    // 100 MHz realtime counter: 10 ns ticks
    printf("  blocks (first %d): prologue %.2f us  loop %.2f us  epilogue %.2f us; span %.1f us; mean concurrency %.1f blocks\n", nb,
           pro / nb * 0.01, loop / nb * 0.01, epi / nb * 0.01, (t1 - t0) * 0.01, (pro + loop + epi) / (double)(t1 - t0));
  }
  const long long* a = &h[0];
  printf("  block 0 wave 0 total first->last tile: %lld cycles\n", h[((0 * 8 + 0) * 10 + (nkt < 10 ? nkt - 1 : 9)) * 5 + 4] - a[0]);
#endif
  return 0;
}
