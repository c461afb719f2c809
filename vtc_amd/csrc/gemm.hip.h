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
#include "common.hip.h"

namespace vtc {

enum : int { EPI_BIAS = 1, EPI_GELU = 2, EPI_RESID = 4, EPI_F32 = 8, EPI_RESID16 = 16,   // RESID16: residual stored 16-bit
             // LayerNorm-free transformer layers (gemm256 only; see gemm256.hip.h "LayerNorm without a LayerNorm kernel"):
             EPI_LNIN = 32,    // A holds PRE-LayerNorm rows; the normalisation is applied to the accumulators
             EPI_LNRES = 64 }; // residual = LayerNorm of 16-bit pre-LN rows; output = pre-LN rows + partial row statistics

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
  int group_n;               // gemm256: tile columns per L2 group (divides tiles_n), see tile_coords()
  // EPI_LNIN / EPI_LNRES (gemm256)
  const float* ln_stats;     // [M][2] (mean, rstd) of the LayerNorm that feeds A (LNIN) or the residual (LNRES)
  const float* colsum;       // LNIN: [N] sum over k of the 16-bit weights W[n][k] (gamma is folded into W, beta into bias)
  const h16* res_h;          // LNRES: 16-bit pre-LN rows the residual is the LayerNorm of (may alias C: a lane reads exactly
                             //        the elements it later writes)
  const float* res_g;        // LNRES: [N] gamma of that LayerNorm (its beta is folded into bias)
  float* part;               // LNRES: [16][part_rows][2] (mean, M2) of the output row over 16 groups of 48 columns; group-major, so
  int part_rows;             //        the 16 rows x 8 B a wave writes at a time are one whole 128-B line
};

__device__ __forceinline__ int lds_swz(int row) { return (row >> 1) & 7; }

// ---------------------------------------------------------------------------------------------
// Epilogue through LDS: a wave parks a 64-row x 64-col fp32 piece of its accumulators in a private
// LDS region (pitch 68 floats), then re-reads it ROW-wise so that bias / GELU / residual / conversion
// run on 4 or 8 consecutive columns per lane and global accesses are 16 B per lane, branch-free
// (only the store is predicated).  The 64 local columns map to global columns
// col_base + (c >> 5) * col_group_stride + (c & 31).
// ---------------------------------------------------------------------------------------------
constexpr int kEpiPitch = 68;                       // floats; 272 B rows keep float4 reads 16-B aligned
constexpr int kEpiWaveBytes = 64 * kEpiPitch * 4;   // 17408

template <bool F16, int EPI>
__device__ __forceinline__ void epilogue_rows(const float* lds, const GemmParams& p, int row_base, int col_base,
                                              int col_group_stride, int ccol_z, int lane) {
  if constexpr (EPI & EPI_F32) {
    const int c4 = (lane & 15) * 4;
    const int col = col_base + (c4 >> 5) * col_group_stride + (c4 & 31);
    const bool col_ok = col < p.N;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI & EPI_BIAS) bv = *(const float4*)(p.bias + ccol_z + (col_ok ? col : 0));
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int r = it * 4 + (lane >> 4);
      const int row = row_base + r;
      const int row_c = row < p.M ? row : p.M - 1;
      float4 v = *(const float4*)(lds + r * kEpiPitch + c4);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      if constexpr (EPI & EPI_GELU) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
      const long long idx = (long long)row_c * p.ldc + ccol_z + (col_ok ? col : 0);
      if constexpr (EPI & EPI_RESID) {
        const float4 rv = *(const float4*)(p.resid + idx);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      if constexpr (EPI & EPI_RESID16) {
        const u16x4 rv = *(const u16x4*)((const h16*)p.resid + idx);
        v.x += from_h16<F16>(rv[0]); v.y += from_h16<F16>(rv[1]); v.z += from_h16<F16>(rv[2]); v.w += from_h16<F16>(rv[3]);
      }
      if (row < p.M && col_ok) *(float4*)((float*)p.C + idx) = v;
    }
  } else {
    const int c8 = (lane & 7) * 8;
    const int col = col_base + (c8 >> 5) * col_group_stride + (c8 & 31);
    const bool col_ok = col < p.N;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI & EPI_BIAS) {
      const float4 b0 = *(const float4*)(p.bias + ccol_z + (col_ok ? col : 0));
      const float4 b1 = *(const float4*)(p.bias + ccol_z + (col_ok ? col : 0) + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      const int r = it * 8 + (lane >> 3);
      const int row = row_base + r;
      const float4 v0 = *(const float4*)(lds + r * kEpiPitch + c8);
      const float4 v1 = *(const float4*)(lds + r * kEpiPitch + c8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] += bv[k];
        if constexpr (EPI & EPI_GELU) v[k] = gelu_f(v[k]);
      }
      if constexpr (EPI & EPI_RESID16) {
        const int row_c = row < p.M ? row : p.M - 1;
        const u16x8 rv = *(const u16x8*)((const h16*)p.resid + (long long)row_c * p.ldc + ccol_z + (col_ok ? col : 0));
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += from_h16<F16>(rv[k]);
      }
      if (row < p.M && col_ok) {
        const long long idx = (long long)row * p.ldc + ccol_z + col;
        *(uint4*)((h16*)p.C + idx) = make_uint4(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]), pack2<F16>(v[4], v[5]),
                                                pack2<F16>(v[6], v[7]));
      }
    }
  }
}

// ABL (dev-only ablations, tools/gemm_bench.hip): 1 = no global loads after the first tile, 2 = ds_read only
// on the first K-tile, 4 = no MFMA.  ABL = 0 in the library.
template <bool F16, int BN, int EPI, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int BM = 128;
  constexpr int WN = BN / 2;
  constexpr int NJ = WN / 16;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int W_INSTR = BN / 32;  // glds instructions per wave for the W tile
  constexpr int SMEM_BYTES = 2 * STAGE_BYTES > 4 * kEpiWaveBytes ? 2 * STAGE_BYTES : 4 * kEpiWaveBytes;
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

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
  u16x8 af[2][4], bf[2][NJ];
  stage(0, 0);
  for (int kb = 0; kb < nk; ++kb) {
    const int cur = kb & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kb + 1 < nk && !((ABL & 1) && kb > 0)) stage(cur ^ 1, kb + 1);
    const char* base = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + fq;
      if (!((ABL & 2) && kb > 0)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[ks][i] = *(const u16x8*)(base + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[ks][j] = *(const u16x8*)(base + b_off[j] + ((c ^ b_sw[j]) << 4));
      }
      if constexpr (ABL & 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(af[ks][i]));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(bf[ks][j]));
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16<F16>(af[ks][i], bf[ks][j], acc[i][j]);
      }
    }
  }

  // ---- epilogue through LDS (C/D map: col = lane&15, row = (lane>>4)*4 + reg)
  __syncthreads();  // every wave is done reading the last K-tile
  float* my = (float*)(smem + wave * kEpiWaveBytes);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(i * 16 + fq * 4 + r) * kEpiPitch + j * 16 + fr] = acc[i][j][r];
  __syncthreads();
  if constexpr (NJ == 4) {
    epilogue_rows<F16, EPI>(my, p, tile_m * BM + wm * 64, tile_n * BN + wn * WN, 32, z * p.c_col_z, lane);
  } else {
    // 32 live columns per wave: lanes whose column falls in the dead half are masked by col < N or by pairing
    GemmParams q = p;
    q.N = min(p.N, tile_n * BN + wn * WN + WN);
    epilogue_rows<F16, EPI>(my, q, tile_m * BM + wm * 64, tile_n * BN + wn * WN, 32, z * p.c_col_z, lane);
  }
}

template <bool F16, int BN, int EPI, int ABL = 0>
inline void launch_gemm(GemmParams p, int nz, hipStream_t s) {
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)nz);
  hipLaunchKernelGGL((gemm_kernel<F16, BN, EPI, ABL>), grid, dim3(256), 0, s, p);
}

}  // namespace vtc
