// Grouped positional convolution (768 ch, 16 groups of 48, k = 128, pad 64, last frame dropped) + bias + GELU +
// residual, HF modeling_hubert.py:45-103, 439-440.
//
// Per (window, group) the padded, group-major input X[(T+128)][48] makes the im2col matrix Toeplitz:
// A[m][k = j*48 + ci] = X[(m + j)*48 + ci] = Xflat[m*48 + k].  So the WHOLE A operand of a 256-frame tile is a
// 36 KiB slab (256+128 rows x 96 B) that is loaded into LDS once; the MFMA A fragment of row m, 16-B k-chunk c
// is simply slab[m*96 + c*16] (conflict-free for ds_read_b128: 96 B row stride walks all 16 slots).  Only the
// weights (48 x 6144 per group, L2 resident) stream, 8 KiB per 64-deep K step, double buffered.
// Block = 4 waves, wave tile 64 frames x 48 channels (4 x 3 MFMA 16x16x32 tiles).
#pragma once
#include "common.hip.h"
#include "gemm.hip.h"

namespace vtc {

constexpr int kPcRows = 256;                        // frames per block
constexpr int kPcSlab = (kPcRows + 128) * 96;       // 36864 B
constexpr int kPcWStage = 64 * 128;                 // 64 weight rows (48 real + 16 zero) x 64 k x 2 B

template <bool F16>
__global__ __launch_bounds__(256) void posconv_kernel(const h16* __restrict__ posin, const h16* __restrict__ wpk,
                                                      const float* __restrict__ bias, const float* __restrict__ h0,
                                                      float* __restrict__ out, int T) {
  __shared__ __attribute__((aligned(16))) char smem[kPcSlab + 2 * kPcWStage];
  const int t0 = blockIdx.x * kPcRows, g = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int TP = T + 128;

  // ---- slab: rows [t0, t0+384) of this (window, group), a linear copy (clamped at the end of the slab)
  {
    const char* base = (const char*)(posin + ((long long)b * 16 + g) * TP * 48);
    const long long lim = (long long)TP * 96 - 16;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int piece = wave * 9 + i;  // 36 pieces of 1 KiB
      long long off = (long long)t0 * 96 + piece * 1024 + lane * 16;
      off = off < lim ? off : lim;
      __builtin_amdgcn_global_load_lds(GPTR(base + off), LPTR(smem + piece * 1024), 16, 0, 0);
    }
  }
  // ---- weight staging: 64 rows x 128 B per K step, 2 pieces per wave, XOR swizzle on the source chunk
  const char* wsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + (lane >> 3);
    wsrc[i] = (const char*)(wpk + ((long long)g * 64 + row) * 6144 + (((lane & 7) ^ lds_swz(row)) * 8));
  }
  auto stage_w = [&](int buf, int kb) {
    char* dst = smem + kPcSlab + buf * kPcWStage + wave * 2048;
    __builtin_amdgcn_global_load_lds(GPTR(wsrc[0] + (long long)kb * 128), LPTR(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GPTR(wsrc[1] + (long long)kb * 128), LPTR(dst + 1024), 16, 0, 0);
  };

  f32x4 acc[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int a_off[4], b_off[3], b_sw[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = (wave * 64 + i * 16 + fr) * 96 + fq * 16;
#pragma unroll
  for (int j = 0; j < 3; ++j) { const int r = j * 16 + fr; b_off[j] = r * 128; b_sw[j] = lds_swz(r); }

  stage_w(0, 0);
  for (int kb = 0; kb < 96; ++kb) {
    const int cur = kb & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kb + 1 < 96) stage_w(cur ^ 1, kb + 1);
    const char* wb = smem + kPcSlab + cur * kPcWStage;
    const char* ab = smem + kb * 128;   // k advances 64 elements = 128 B along the flattened slab
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u16x8 af[4], bf[3];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const u16x8*)(ab + a_off[i] + ks * 64);
#pragma unroll
      for (int j = 0; j < 3; ++j) bf[j] = *(const u16x8*)(wb + b_off[j] + (((ks * 4 + fq) ^ b_sw[j]) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma16<F16>(af[i], bf[j], acc[i][j]);
    }
  }

  // ---- epilogue: x + gelu(conv + bias) -> fp32 (B*T, 768).  C/D map: col = lane&15, row = (lane>>4)*4 + reg.
  // All residual loads first (clamped rows, no branches), then the predicated stores.
  float bv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) bv[j] = bias[g * 48 + j * 16 + fr];
  float rv[4][3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int t = t0 + wave * 64 + i * 16 + fq * 4 + r;
      t = t < T ? t : T - 1;
      const float* src = h0 + ((long long)b * T + t) * 768 + g * 48 + fr;
#pragma unroll
      for (int j = 0; j < 3; ++j) rv[i][j][r] = src[j * 16];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = t0 + wave * 64 + i * 16 + fq * 4 + r;
      if (t < T) {
        float* dst = out + ((long long)b * T + t) * 768 + g * 48 + fr;
#pragma unroll
        for (int j = 0; j < 3; ++j) dst[j * 16] = rv[i][j][r] + gelu_f(acc[i][j][r] + bv[j]);
      }
    }
}

}  // namespace vtc
