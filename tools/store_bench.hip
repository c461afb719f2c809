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

// Epilogue-shaped stores of one 256 x 192 16-bit tile (96 KB per block, row stride as in the QKV output):
//   MODE 0: as gemm256's C^T epilogue: per wave and row group a b128 (16 rows x 64 B) and a b64 (16 rows x 32 B)
//   MODE 1: the same bytes as whole 128-B lines: the tile as [256 rows][384 B], 24 lanes x 16 B per row
//   MODE 2: 64-B pieces only (16 rows x 64 B per instruction, 1.5 instructions' worth: the BN = 256 shape)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void tile_store_kernel(char* out, long long row_stride, long long* cycles, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
  const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  const u32x2 v2 = {(unsigned)lane, 1u};
  __syncthreads();
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    char* base = out + ((long long)(blockIdx.x * tiles + t) * 256) * row_stride;
    if (MODE == 0) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int row = (g >> 2) * 128 + wm * 64 + (g & 3) * 16 + fr;
        *(u32x4*)(base + row * row_stride + (wn * 32 + 8 * fq) * 2) = v;
        *(u32x2*)(base + row * row_stride + (128 + wn * 16 + 4 * fq) * 2) = v2;
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int idx = (wave * 12 + k) * 64 + lane;
        const int row = idx / 24, c = idx % 24;
        *(u32x4*)(base + row * row_stride + c * 16) = v;
      }
    } else {
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        const int row = ((g * 8 + wave) % 16) * 16 + fr;
        *(u32x4*)(base + row * row_stride + (g / 2) * 64 + fq * 16) = v;
      }
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();      // all stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long long t2 = (long long)__builtin_amdgcn_s_memtime();      // all stores acknowledged
  if (threadIdx.x == 0) { cycles[blockIdx.x * 2] = t1 - t0; cycles[blockIdx.x * 2 + 1] = t2 - t0; }
}

template <int MODE> void run_tile(char* out, long long* cyc, int grid, int tiles) {
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((tile_store_kernel<MODE>), dim3(grid), dim3(512), 0, 0, out, 4608LL, cyc, tiles);
  CK(hipDeviceSynchronize());
  long long h[512]; CK(hipMemcpy(h, cyc, grid * 16, hipMemcpyDeviceToHost));
  double a = 0, b = 0; for (int i = 0; i < grid; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
  a /= grid; b /= grid;
  printf("tile mode %d  grid %3d  tiles %d: issued after %7.0f cycles, acknowledged after %7.0f -> %.1f B/cycle/CU\n", MODE, grid, tiles, a, b,
         tiles * 98304.0 / b);
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
  CK(hipMalloc(&out, 4LL * 256 * 256 * 4608 + (1 << 20))); CK(hipMalloc(&cyc, 512 * 8));
  for (int grid : {256, 8})
    for (int tiles : {1, 4}) { run_tile<0>(out, cyc, grid, tiles); run_tile<1>(out, cyc, grid, tiles); run_tile<2>(out, cyc, grid, tiles); }
  for (int grid : {256, 16}) {
    run<16, 16>(out, cyc, grid); run<8, 16>(out, cyc, grid); run<4, 16>(out, cyc, grid); run<2, 16>(out, cyc, grid); run<1, 16>(out, cyc, grid);
    run<16, 4>(out, cyc, grid); run<2, 4>(out, cyc, grid);
  }
  return 0;
}
