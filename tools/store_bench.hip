// Dev tool: per-CU store throughput of epilogue-like store shapes (one 512-thread block per CU, every wave
// issues NST dwordx4 stores back to back).  Shape = how many output rows one wave-instruction touches.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/store_bench.hip -o tools/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int ROWS, int NST>
__global__ __launch_bounds__(512) void store_kernel(char* out, long long row_stride, long long* cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = 64 / ROWS;                 // lanes per row
  const int r = lane / LPR, c = lane % LPR;
  // block tile: rows [blockIdx.x*256, +256); a wave owns 32 consecutive tile rows per "pass" region
  char* base = out + ((long long)blockIdx.x * 256 + wave * 32) * row_stride;
  const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  __syncthreads();
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    // instruction i covers ROWS rows x (LPR*16) bytes; successive instructions walk columns then rows
    const int per_row_instr = 512 / (LPR * 16) > 0 ? 512 / (LPR * 16) : 1;   // instructions to fill 512 B of a row
    const int rr = (i / per_row_instr) * ROWS + r;
    const int cc = (i % per_row_instr) * (LPR * 16) + c * 16;
    *(u32x4*)(base + (long long)(rr % 32) * row_stride + cc) = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int ROWS, int NST> void run(char* out, long long* cyc, int grid) {
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((store_kernel<ROWS, NST>), dim3(grid), dim3(512), 0, 0, out, 4608LL, cyc);
  CK(hipDeviceSynchronize());
  long long h[256]; CK(hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost));
  double m = 0; for (int i = 0; i < grid; ++i) m += h[i];
  m /= grid;
  printf("rows/instr %2d  stores/wave %2d  grid %3d: %8.0f cycles  -> %.1f B/cycle/CU\n", ROWS, NST, grid, m, 8.0 * NST * 1024 / m);
}

int main() {
  char* out; long long* cyc;
  CK(hipMalloc(&out, 256LL * 256 * 4608 + (1 << 20))); CK(hipMalloc(&cyc, 256 * 8));
  for (int grid : {256, 16}) {
    run<16, 16>(out, cyc, grid); run<8, 16>(out, cyc, grid); run<4, 16>(out, cyc, grid); run<2, 16>(out, cyc, grid); run<1, 16>(out, cyc, grid);
    run<16, 4>(out, cyc, grid); run<2, 4>(out, cyc, grid);
  }
  return 0;
}
