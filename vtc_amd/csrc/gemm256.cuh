// 256 x 256 x 64 MFMA GEMM, 8 waves, 8-phase schedule (two K-tiles per loop iteration).
// Same contract as gemm.cuh (C = A * W^T, overlapping-row A addressing, fused epilogues); requires
// K % 128 == 0 and N % 256 == 0 (W rows are read unguarded).
//
// Why a second kernel: a 128 x 128 tile needs 1 B of L2->LDS fill per 64 FLOP, which at the MFMA rate is
// more than the L2 can deliver; 256 x 256 halves that.  Structure (cdna_hip_programming.md section 5,
// "The 256^2 8-phase template"):
//   * LDS 128 KiB = 2 K-tile buffers x 4 half-tiles (A0,B0,B1,A1) of 128 rows x 128 B, XOR-swizzled.
//   * a wave (wm 0..1, wn 0..3) owns rows {h*128 + wm*64 ..+64} and cols {h*128 + wn*32 ..+32}, h = 0,1, so its
//     four 64 x 32 quadrants consume the half-tiles in the order they are staged: A0+B0, B1, A1.
//   * every phase: ds_read the fragments it needs, issue ONE half-tile of global_load_lds prefetch,
//     barrier, lgkmcnt(0), 16 MFMA, barrier.  vmcnt is only waited at the K-tile boundary phases, with a
//     COUNT (3 half-tiles stay in flight across the barriers), never 0 in steady state.
//   * a slot is restaged at least one full phase after its last ds_read; a staged tile is first read one
//     phase after the wait + barrier that retired it.
#pragma once
#include "common.cuh"
#include "gemm.cuh"

namespace vtc {

constexpr int kHalfBytes = 128 * 128;      // one half-tile: 128 rows x 64 elements x 2 B
constexpr int kBufBytes = 4 * kHalfBytes;  // A0, B0, B1, A1
enum : int { HT_A0 = 0, HT_B0 = 1, HT_B1 = 2, HT_A1 = 3 };

template <bool F16, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmParams p) {
  constexpr int SMEM_BYTES = 2 * kBufBytes > 8 * kEpiWaveBytes ? 2 * kBufBytes : 8 * kEpiWaveBytes;
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const unsigned ntiles = (unsigned)(p.tiles_m * p.tiles_n);
  const unsigned tl = xcd_remap(blockIdx.x, ntiles);
  const int tile_n = (int)(tl % (unsigned)p.tiles_n);
  const int tile_m = (int)(tl / (unsigned)p.tiles_n);
  const int z = blockIdx.y;

  // ---- per-lane global sources: [half-tile kind][instruction 0/1]
  const char* src[4][2];
  {
    const int lr = lane >> 3, lc = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave * 2 + i) * 8 + lr;  // row inside a half-tile
      const long long sw = (long long)((lc ^ lds_swz(r)) * 8);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int m = tile_m * 256 + h * 128 + r;
        m = m < p.M ? m : p.M - 1;
        const int b = m / p.rows_per_batch;
        const int rr = m - b * p.rows_per_batch;
        src[h ? HT_A1 : HT_A0][i] = (const char*)(p.A + ((long long)z * p.a_z_stride + (long long)b * p.a_batch_stride +
                                                         (long long)rr * p.a_row_stride + sw));
        const long long n = (long long)tile_n * 256 + h * 128 + r;
        src[h ? HT_B1 : HT_B0][i] = (const char*)(p.W + ((long long)z * p.w_z_stride + n * p.K + sw));
      }
    }
  }
  const int stage_off = wave * 2 * 1024;  // this wave's two 1 KiB pieces inside a half-tile

  auto issue = [&](int buf, int kind, int kt) {
    char* dst = smem + buf * kBufBytes + kind * kHalfBytes + stage_off;
    const long long koff = (long long)kt * 128;
    __builtin_amdgcn_global_load_lds(GPTR(src[kind][0] + koff), LPTR(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GPTR(src[kind][1] + koff), LPTR(dst + 1024), 16, 0, 0);
  };

  f32x4 acc[2][2][4][2];  // [mq][nq][i][j]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a half-tile
  const int fr = lane & 15, fq = lane >> 4;
  int a_row[4], a_sw[4], b_row[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int r = wm * 64 + i * 16 + fr; a_row[i] = r * 128; a_sw[i] = lds_swz(r); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int r = wn * 32 + j * 16 + fr; b_row[j] = r * 128; b_sw[j] = lds_swz(r); }

  u16x8 fa[2][4], fb0[2][2], fb1[2][2];  // [ks][tile]

  auto read_a = [&](int buf, int kind) {
    const char* base = smem + buf * kBufBytes + kind * kHalfBytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i] = *(const u16x8*)(base + a_row[i] + (((ks * 4 + fq) ^ a_sw[i]) << 4));
  };
  auto read_b = [&](int buf, int kind, u16x8 (&fb)[2][2]) {
    const char* base = smem + buf * kBufBytes + kind * kHalfBytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[ks][j] = *(const u16x8*)(base + b_row[j] + (((ks * 4 + fq) ^ b_sw[j]) << 4));
  };
  auto mma = [&](int mq, int nq, u16x8 (&fb)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[mq][nq][i][j] = mfma16<F16>(fa[ks][i], fb[ks][j], acc[mq][nq][i][j]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto sync_reads = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto end_phase = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  const int nk = p.K >> 6;  // even, >= 2

  // ---- prologue: all of tile 0, three half-tiles of tile 1
  issue(0, HT_A0, 0); issue(0, HT_B0, 0); issue(0, HT_B1, 0); issue(0, HT_A1, 0);
  issue(1, HT_A0, 1); issue(1, HT_B0, 1); issue(1, HT_B1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();

#define VTC_TILE(BUF, T)                                                                                   \
  {                                                                                                        \
    /* phase 1: quadrant (0,0); needs A0,B0 of this tile; stage the last half-tile of tile T+1 */          \
    read_b(BUF, HT_B0, fb0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    read_a(BUF, HT_A0);                                                                                    \
    if ((T) + 1 < nk) issue((BUF) ^ 1, HT_A1, (T) + 1);                                                    \
    sync_reads();                                                                                          \
    mma(0, 0, fb0);                                                                                        \
    end_phase();                                                                                           \
    /* phase 2: quadrant (0,1); needs B1 */                                                                \
    read_b(BUF, HT_B1, fb1);                                                                               \
    if ((T) + 2 < nk) issue(BUF, HT_A0, (T) + 2);                                                          \
    sync_reads();                                                                                          \
    mma(0, 1, fb1);                                                                                        \
    end_phase();                                                                                           \
    /* phase 3: quadrant (1,1); needs A1 */                                                                \
    read_a(BUF, HT_A1);                                                                                    \
    if ((T) + 2 < nk) issue(BUF, HT_B0, (T) + 2);                                                          \
    sync_reads();                                                                                          \
    mma(1, 1, fb1);                                                                                        \
    end_phase();                                                                                           \
    /* phase 4: quadrant (1,0); registers only; K-tile boundary wait */                                    \
    if ((T) + 2 < nk) {                                                                                    \
      issue(BUF, HT_B1, (T) + 2);                                                                          \
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                     \
    } else {                                                                                               \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    }                                                                                                      \
    __builtin_amdgcn_s_barrier();                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    mma(1, 0, fb0);                                                                                        \
    end_phase();                                                                                           \
  }

  for (int t = 0; t < nk; t += 2) {
    VTC_TILE(0, t)
    VTC_TILE(1, t + 1)
  }
#undef VTC_TILE

  // ---- epilogue through LDS, two passes of 64 rows x (2 x 32) cols per wave (the loop's last barrier
  //      guarantees every wave is done reading the K-tiles)
  float* my = (float*)(smem + wave * kEpiWaveBytes);
#pragma unroll
  for (int mq = 0; mq < 2; ++mq) {
    if (mq) __syncthreads();
#pragma unroll
    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            my[(i * 16 + fq * 4 + r) * kEpiPitch + nq * 32 + j * 16 + fr] = acc[mq][nq][i][j][r];
    __syncthreads();
    epilogue_rows<F16, EPI>(my, p, tile_m * 256 + mq * 128 + wm * 64, tile_n * 256 + wn * 32, 128, z * p.c_col_z, lane);
  }
}

template <bool F16, int EPI>
inline void launch_gemm256(GemmParams p, int nz, hipStream_t s) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)nz);
  hipLaunchKernelGGL((gemm256_kernel<F16, EPI>), grid, dim3(512), 0, s, p);
}

}  // namespace vtc
