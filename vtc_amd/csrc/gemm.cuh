// C[M,N] = A[M,K] * W[N,K]^T on 16-bit MFMA (v_mfma_f32_16x16x32_{bf16,f16}), fp32 accumulate.
//
// One kernel serves every dense contraction of the encoder:
//   * conv layers 1..6 as implicit GEMM: activations are channel-last (B, L, 512), so the im2col row of
//     output frame t is the CONTIGUOUS slice x[s*t : s*t+k, :]; A is addressed with overlapping rows
//     (row stride s*512, K = k*512) and nothing is materialised.
//   * grouped positional conv: per group g the padded, group-major input (B,16,T+128,48) gives rows of
//     K = 128*48 at row stride 48 (same overlapping-row trick); z = group.
//   * projection, QKV, out-proj, FFN1, FFN2: plain row-major GEMM.
// Tile 128 x BN x 64, 4 waves (2x2), each wave 64 x BN/2 = 4 x (BN/32) MFMA 16x16 tiles.
// Global -> LDS by global_load_lds_dwordx4 (1 KiB per wave instruction, lane-linear LDS image);
// the XOR swizzle lives on the per-lane SOURCE address and on the ds_read_b128 address.
#pragma once
#include "common.cuh"

namespace vtc {

enum : int { EPI_BIAS = 1, EPI_GELU = 2, EPI_RESID = 4, EPI_F32 = 8 };

struct GemmParams {
  const h16* A;
  const h16* W;
  void* C;
  const float* bias;
  const float* resid;
  int M, N, K;
  int ldc;
  int rows_per_batch;
  long long a_row_stride;    // elements
  long long a_batch_stride;  // elements
  long long a_z_stride;      // elements
  long long w_z_stride;      // elements
  int c_col_z;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ int lds_swz(int row) { return (row >> 1) & 7; }

template <bool F16, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int BM = 128;
  constexpr int WN = BN / 2;
  constexpr int NJ = WN / 16;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int W_INSTR = BN / 32;  // glds instructions per wave for the W tile
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const unsigned ntiles = (unsigned)(p.tiles_m * p.tiles_n);
  const unsigned t = xcd_remap(blockIdx.x, ntiles);
  const int tile_n = (int)(t % (unsigned)p.tiles_n);
  const int tile_m = (int)(t / (unsigned)p.tiles_n);
  const int z = blockIdx.y;

  // ---- per-lane global source pointers (bytes), one per glds instruction
  const char* a_src[4];
  const char* w_src[W_INSTR];
  {
    const int lr = lane >> 3, lc = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wave * 32 + i * 8 + lr;
      int m = tile_m * BM + r;
      m = m < p.M ? m : p.M - 1;
      const int b = m / p.rows_per_batch;
      const int rr = m - b * p.rows_per_batch;
      const long long off = (long long)z * p.a_z_stride + (long long)b * p.a_batch_stride +
                            (long long)rr * p.a_row_stride + (long long)((lc ^ lds_swz(r)) * 8);
      a_src[i] = (const char*)(p.A + off);
    }
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
      const int r = wave * (BN / 4) + i * 8 + lr;
      const long long n = (long long)tile_n * BN + r;
      const long long off = (long long)z * p.w_z_stride + n * p.K + (long long)((lc ^ lds_swz(r)) * 8);
      w_src[i] = (const char*)(p.W + off);
    }
  }

  auto stage = [&](int buf, int kb) {
    char* base = smem + buf * STAGE_BYTES;
    const long long koff = (long long)kb * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(GPTR(a_src[i] + koff), LPTR(base + (wave * 32 + i * 8) * 128), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i)
      __builtin_amdgcn_global_load_lds(GPTR(w_src[i] + koff),
                                       LPTR(base + BM * 128 + (wave * (BN / 4) + i * 8) * 128), 16, 0, 0);
  };

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes, within a stage); row = tile row, chunk = 16-B k chunk
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[4], b_off[NJ], a_sw[4], b_sw[NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + i * 16 + fr;
    a_off[i] = r * 128;
    a_sw[i] = lds_swz(r);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int r = wn * WN + j * 16 + fr;
    b_off[j] = BM * 128 + r * 128;
    b_sw[j] = lds_swz(r);
  }

  const int nk = p.K >> 6;
  stage(0, 0);
  for (int kb = 0; kb < nk; ++kb) {
    const int cur = kb & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kb + 1 < nk) stage(cur ^ 1, kb + 1);
    const char* base = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u16x8 af[4], bf[NJ];
      const int c = ks * 4 + fq;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const u16x8*)(base + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = *(const u16x8*)(base + b_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16<F16>(af[i], bf[j], acc[i][j]);
    }
  }

  // ---- epilogue: C/D map col = lane&15, row = (lane>>4)*4 + reg
  const int col0 = tile_n * BN + wn * WN + fr;
  const int row0 = tile_m * BM + wm * 64 + fq * 4;
  const int ccol_z = z * p.c_col_z;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = col0 + j * 16;
    if (col >= p.N) continue;
    float bv = 0.f;
    if constexpr (EPI & EPI_BIAS) bv = p.bias[ccol_z + col];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + i * 16 + r;
        if (row >= p.M) continue;
        float v = acc[i][j][r] + bv;
        if constexpr (EPI & EPI_GELU) v = gelu_f(v);
        const long long idx = (long long)row * p.ldc + ccol_z + col;
        if constexpr (EPI & EPI_RESID) v += p.resid[idx];
        if constexpr (EPI & EPI_F32) ((float*)p.C)[idx] = v;
        else ((h16*)p.C)[idx] = to_h16<F16>(v);
      }
    }
  }
}

template <bool F16, int BN, int EPI>
inline void launch_gemm(GemmParams p, int nz, hipStream_t s) {
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)nz);
  hipLaunchKernelGGL((gemm_kernel<F16, BN, EPI>), grid, dim3(256), 0, s, p);
}

}  // namespace vtc
