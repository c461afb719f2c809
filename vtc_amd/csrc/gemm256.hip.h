// Persistent 256 x BN x 64 MFMA GEMM (BN = 256 or 192), 8 waves, 8-phase schedule (two K-tiles per loop
// iteration).  Same contract as gemm.hip.h (C = A * W^T, overlapping-row A addressing, fused epilogues);
// requires K % 128 == 0 and N % BN == 0 (W rows are read unguarded).
//
// Why: tools/gemm_bench ablations show the 128 x 128 kernel is bound by the L2 -> LDS fill (it runs as fast
// with its MFMAs removed: ~13.4 TB/s of fill = 64 FLOP/B x 0.86 PFLOP/s).  A 256 x 256 tile needs half the
// fill bytes per FLOP, 256 x 192 0.58x, and BN = 192 tiles N = 768 / 2304 / 3072 without a ragged last
// round over the 256 CUs (M = 31936: 500 / 1500 / 2000 tiles).
//
// Structure (cdna_hip_programming.md section 5, "The 256^2 8-phase template"):
//   * LDS: 2 K-tile buffers x 4 half-tiles (A0,B0,B1,A1; 128 rows x 128 B each, B1 64 rows when BN = 192),
//     XOR-swizzled.
//   * a wave (wm 0..1, wn 0..3) owns rows {h*128 + wm*64 ..+64}, h = 0,1, and cols {wn*32 ..+32} of B0 plus
//     {wn*NB1 ..} of B1, so its four quadrants consume the half-tiles in the order they are staged.
//   * every phase: ds_read the fragments it needs, issue ONE half-tile of global_load_lds prefetch,
//     barrier, lgkmcnt(0), MFMAs, barrier.  vmcnt is only waited at the K-tile boundary phases, with a
//     COUNT (3 half-tiles stay in flight across the barriers), never 0 in steady state.
//   * a slot is restaged at least one full phase after its last ds_read; a staged tile is first read one
//     phase after the wait + barrier that retired it.
//   * the MFMAs compute C^T (W fragment as the A operand): a lane then holds ONE output row and, per 16 x 16
//     tile, 4 consecutive output columns.  W rows are staged into LDS in a permuted order (perm32 below) so
//     that the two tiles of a 32-column group give a lane 8 consecutive columns (16 B).  The GELU kernels and the fp32
//     output store those straight from the accumulators; the other 16-bit outputs go through two LDS regions behind the
//     staging buffers and leave as whole rows (see the epilogue: partial 128-B lines are what a CU stores slowly).
//   * persistent: a block walks tiles b, b+G, b+2G, ...; the NEXT tile's prologue loads are issued before
//     the current tile's epilogue (registers only), so loads and stores overlap.
//
// LayerNorm without a LayerNorm kernel (post-LN transformer layers, modeling_hubert.py:387-397).  The residual stream is
// kept as 16-bit PRE-LayerNorm rows u (what the next GEMM reads as its A operand) plus per-row statistics (mean, rstd):
//   EPI_LNRES (out-proj, FFN2):  v = acc + bias' + gamma_c * ((u - mean_r) * rstd_r)   [bias' = bias + beta]
//       writes v (16 bit) and, per row, the (mean, M2) of the fp32 v over each wave's 48 columns: 16 partials per row in a
//       fixed (group-major) layout, merged in a fixed order by ln_stats_finalize_kernel: bit-reproducible, no atomics.
//   EPI_LNIN (QKV, FFN1):  LN(u) W^T + b = rstd_r * (u W'^T - mean_r * s_c) + b'  with W' = gamma o W rounded to 16 bit,
//       s_c = sum_k W'[c][k] (of the rounded weights, so the mean cancels exactly), b' = b + W beta.
// Rounding: the residual stream is rounded to 16 bit once per LayerNorm (u), where the LayerNorm-kernel form rounded
// twice (the pre-LN rows and the LayerNorm output); statistics come from the fp32 values.  24 LayerNorm launches per
// forward (4 B per element each) become 24 launches of a 5 us statistics merge.
// Tried and dropped: carrying the low halves u_l = u - fp16(u) as a second 16-bit array (residual exact to ~22 bits: 37 %
// less logit error instead of 17 %).  Twice the epilogue loads and stores need either a second pass of loads behind the
// first pass's stores (a full vmcnt(0) drain that waits for the store acknowledgements of the chip-wide output burst:
// +33..45 us per launch) or ~100 more live registers than the kernel has (96 spilled).
#pragma once
#include "common.hip.h"
#include "gemm.hip.h"

namespace vtc {

constexpr int kHalfBytes = 128 * 128;  // one full half-tile: 128 rows x 64 elements x 2 B

// LDS row rho (0..31) of a 32-row group of W holds W row perm32(rho): MFMA tile j = rho>>4, operand row
// i' = rho&15 lands in accumulator (quad i'>>2, reg i'&3) and is output column 8*(i'>>2) + 4*j + (i'&3).
__device__ __forceinline__ int perm32(int rho) { return 8 * ((rho & 15) >> 2) + 4 * (rho >> 4) + (rho & 3); }

#ifndef VTC_STORE_AUX
#define VTC_STORE_AUX 0   // cache policy bits of the epilogue stores (tools/gemm_bench builds try others)
#endif
#ifdef VTC_STAMPS   // tools/gemm_bench diagnostic build: s_memtime stamps of the first two tiles of every block
__device__ long long g_stamps[256 * 16];
#define VTC_STAMP(k) do { if (tile_it < 2) stamp_t[tile_it * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define VTC_STAMP(k) do { } while (0)
#endif

template <int BN>
struct Tile256 {
  static_assert(BN == 256 || BN == 192, "BN must be 256 or 192");
  static constexpr int NJ1 = BN == 256 ? 2 : 1;       // 16-col MFMA tiles per wave in the B1 half
  static constexpr int B1_ROWS = 64 * NJ1;            // rows of W in the B1 half-tile
  static constexpr int B1_INSTR = NJ1;                // glds instructions per wave for B1
  static constexpr int OFF_A0 = 0, OFF_B0 = kHalfBytes, OFF_B1 = 2 * kHalfBytes;
  static constexpr int OFF_A1 = OFF_B1 + B1_ROWS * 128;
  static constexpr int BUF = OFF_A1 + kHalfBytes;
  static constexpr int SMEM = 2 * BUF;
  // 16-bit outputs leave through LDS (see the epilogue): two regions of GPR row groups x 16 rows x both wave rows
  static constexpr int GPR = BN == 192 ? 2 : 1;
  static constexpr int PITCH = BN * 2;
  static constexpr int REGION = 2 * GPR * 16 * PITCH;
  static_assert(SMEM + 2 * REGION <= 160 * 1024, "staging + output regions must fit the CU's LDS");
};

// Tile order: columns are walked in groups of group_n; inside a group tile_n runs fastest, then tile_m.  With the
// XCD remap (consecutive tiles share an XCD) the W panel of a group (group_n x BN x K, sized to stay in the 4 MiB
// L2 by launch_gemm256) is fetched once per XCD instead of once per round, and the A rows once per group.
__device__ __forceinline__ void tile_coords(const GemmParams& p, unsigned t, int& tile_m, int& tile_n) {
  const unsigned per_group = (unsigned)(p.tiles_m * p.group_n);
  const unsigned g = t / per_group, r = t - g * per_group;
  tile_m = (int)(r / (unsigned)p.group_n);
  tile_n = (int)(g * (unsigned)p.group_n + r % (unsigned)p.group_n);
}

template <bool F16, int EPI, int BN>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmParams p) {
  using TC = Tile256<BN>;
  constexpr int NJ1 = TC::NJ1;
  static_assert(!(EPI & EPI_RESID), "fp32 residual is not supported here (the encoder keeps a 16-bit residual stream)");
  static_assert(!(EPI & EPI_LNRES) || (BN == 192 && !(EPI & (EPI_F32 | EPI_GELU | EPI_RESID16 | EPI_LNIN))),
                "EPI_LNRES: 192-wide tiles (16 groups of 48 columns per 768-wide row), 16-bit output");
#ifndef VTC_EPI_MODE
#define VTC_EPI_MODE 1   // tools/gemm_bench A/B: 0 = arithmetic first, GELU kernels store directly; 1 = pipelined rounds, GELU
#endif                   // direct; 2 = pipelined rounds for every 16-bit output
  constexpr bool OUT_LDS = !(EPI & EPI_F32) && (VTC_EPI_MODE == 2 || !(EPI & EPI_GELU));
  __shared__ __attribute__((aligned(16))) char smem[TC::SMEM + (OUT_LDS ? 2 * TC::REGION : 0)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int z = blockIdx.y;
  const unsigned ntiles = (unsigned)(p.tiles_m * p.tiles_n);

  // per-lane global sources of the tile being staged: A0, A1, B0 (2 instructions each), B1 (NJ1)
  const char* src_a0[2];
  const char* src_a1[2];
  const char* src_b0[2];
  const char* src_b1[NJ1];

  auto set_sources = [&](int tile_m, int tile_n) {
    // lane-derived values are recomputed here from an opaque copy of the thread id: kept live across the K loop they cost
    // ~12 registers the 256-wide variant does not have (they were spilled and reloaded behind a vmcnt(0) every tile)
    int l_;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));   // lane id, never hoisted
    const int lr = l_ >> 3, lc = l_ & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave * 2 + i) * 8 + lr;  // row inside a 128-row half-tile
      const long long sw = (long long)((lc ^ lds_swz(r)) * 8);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int m = tile_m * 256 + h * 128 + r;
        m = m < p.M ? m : p.M - 1;
        const int b = m / p.rows_per_batch;
        const int rr = m - b * p.rows_per_batch;
        const char* a = (const char*)(p.A + ((long long)z * p.a_z_stride + (long long)b * p.a_batch_stride +
                                             (long long)rr * p.a_row_stride + sw));
        if (h) src_a1[i] = a; else src_a0[i] = a;
      }
      const int wr = (r & ~31) + perm32(r & 31);
      src_b0[i] = (const char*)(p.W + ((long long)z * p.w_z_stride + ((long long)tile_n * BN + wr) * p.K + sw));
    }
#pragma unroll
    for (int i = 0; i < NJ1; ++i) {
      const int r = (wave * NJ1 + i) * 8 + lr;  // row inside the B1 half-tile
      const long long sw = (long long)((lc ^ lds_swz(r)) * 8);
      const int wr = NJ1 == 2 ? (r & ~31) + perm32(r & 31) : r;   // 16-column groups need no permutation
      src_b1[i] = (const char*)(p.W + ((long long)z * p.w_z_stride + ((long long)tile_n * BN + 128 + wr) * p.K + sw));
    }
  };

  auto issue_a0 = [&](int buf, int kt) {
    char* d = smem + buf * TC::BUF + TC::OFF_A0 + wave * 2048;
    __builtin_amdgcn_global_load_lds(GPTR(src_a0[0] + (long long)kt * 128), LPTR(d), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GPTR(src_a0[1] + (long long)kt * 128), LPTR(d + 1024), 16, 0, 0);
  };
  auto issue_a1 = [&](int buf, int kt) {
    char* d = smem + buf * TC::BUF + TC::OFF_A1 + wave * 2048;
    __builtin_amdgcn_global_load_lds(GPTR(src_a1[0] + (long long)kt * 128), LPTR(d), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GPTR(src_a1[1] + (long long)kt * 128), LPTR(d + 1024), 16, 0, 0);
  };
  auto issue_b0 = [&](int buf, int kt) {
    char* d = smem + buf * TC::BUF + TC::OFF_B0 + wave * 2048;
    __builtin_amdgcn_global_load_lds(GPTR(src_b0[0] + (long long)kt * 128), LPTR(d), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GPTR(src_b0[1] + (long long)kt * 128), LPTR(d + 1024), 16, 0, 0);
  };
  auto issue_b1 = [&](int buf, int kt) {
    char* d = smem + buf * TC::BUF + TC::OFF_B1 + wave * (NJ1 * 1024);
#pragma unroll
    for (int i = 0; i < NJ1; ++i)
      __builtin_amdgcn_global_load_lds(GPTR(src_b1[i] + (long long)kt * 128), LPTR(d + i * 1024), 16, 0, 0);
  };

  // fragment addresses inside a half-tile
  int a_row[4], a_sw[4], b0_row[2], b0_sw[2], b1_row[NJ1], b1_sw[NJ1];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int r = wm * 64 + i * 16 + fr; a_row[i] = r * 128; a_sw[i] = lds_swz(r); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int r = wn * 32 + j * 16 + fr; b0_row[j] = r * 128; b0_sw[j] = lds_swz(r); }
#pragma unroll
  for (int j = 0; j < NJ1; ++j) { const int r = wn * (16 * NJ1) + j * 16 + fr; b1_row[j] = r * 128; b1_sw[j] = lds_swz(r); }

  f32x4 acc0[2][4][2];    // [mq][i][j]   columns in B0
  f32x4 acc1[2][4][NJ1];  // [mq][i][j]   columns in B1
  u16x8 fa[2][4], fb0[2][2], fb1[2][NJ1];  // [ks][tile]
#ifdef VTC_ABL_NOREAD   // tools/gemm_bench timing ablation (results are garbage): fragments never read from LDS
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) fa[ks][i][e] = (unsigned short)(0x3c00 + lane + i);
#pragma unroll
    for (int j = 0; j < 2; ++j) for (int e = 0; e < 8; ++e) fb0[ks][j][e] = (unsigned short)(0x3800 + lane + j);
#pragma unroll
    for (int j = 0; j < NJ1; ++j) for (int e = 0; e < 8; ++e) fb1[ks][j][e] = (unsigned short)(0x3400 + lane + j);
  }
#define VTC_ABL_READ_RETURN return
#else
#define VTC_ABL_READ_RETURN
#endif
#if defined(VTC_ABL_NOREAD) && defined(VTC_ABL_KEEP_W)   // ... but the W fragments are read
#define VTC_ABL_READ_RETURN_W
#else
#define VTC_ABL_READ_RETURN_W VTC_ABL_READ_RETURN
#endif
#if defined(VTC_ABL_NOREAD) && defined(VTC_ABL_KEEP_A)   // ... but the A fragments are read
#define VTC_ABL_READ_RETURN_A
#else
#define VTC_ABL_READ_RETURN_A VTC_ABL_READ_RETURN
#endif
#ifdef VTC_ABL_NOMFMA   // timing ablation: no MFMAs
#define VTC_ABL_MMA_RETURN return
#else
#define VTC_ABL_MMA_RETURN
#endif

  auto read_a = [&](int buf, int off) {
    VTC_ABL_READ_RETURN_A;
    const char* base = smem + buf * TC::BUF + off;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i] = *(const u16x8*)(base + a_row[i] + (((ks * 4 + fq) ^ a_sw[i]) << 4));
  };
  auto read_b0 = [&](int buf) {
    VTC_ABL_READ_RETURN_W;
    const char* base = smem + buf * TC::BUF + TC::OFF_B0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) fb0[ks][j] = *(const u16x8*)(base + b0_row[j] + (((ks * 4 + fq) ^ b0_sw[j]) << 4));
  };
  auto read_b1 = [&](int buf) {
    VTC_ABL_READ_RETURN_W;
    const char* base = smem + buf * TC::BUF + TC::OFF_B1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < NJ1; ++j) fb1[ks][j] = *(const u16x8*)(base + b1_row[j] + (((ks * 4 + fq) ^ b1_sw[j]) << 4));
  };
  auto mma0 = [&](int mq) {
    VTC_ABL_MMA_RETURN;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc0[mq][i][j] = mfma16<F16>(fb0[ks][j], fa[ks][i], acc0[mq][i][j]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma1 = [&](int mq) {
    VTC_ABL_MMA_RETURN;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ1; ++j) acc1[mq][i][j] = mfma16<F16>(fb1[ks][j], fa[ks][i], acc1[mq][i][j]);
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
  auto prologue = [&]() {  // first tile of the block only: all of K-tile 0, A0 + B0 of K-tile 1 (the steady state)
    issue_a0(0, 0); issue_b0(0, 0); issue_b1(0, 0); issue_a1(0, 0);
    issue_a0(1, 1); issue_b0(1, 1);
  };

  const int nk = p.K >> 6;  // even, >= 4
  const unsigned G = gridDim.x;
  unsigned tl = blockIdx.x;
  {
    int tm0, tn0;
    tile_coords(p, xcd_remap(tl, ntiles), tm0, tn0);
    set_sources(tm0, tn0);
  }
  prologue();

  // Staging order per K-tile T (one half-tile per phase): B1(T+1), A1(T+1), A0(T+2), B0(T+2).  Every slot is
  // restaged >= 2 phases after its last ds_read, which is what the half-phase stagger between the two wave
  // rows (below) needs; at the K-tile boundary wait A0(T+2) and B0(T+2) (4 glds) stay in flight.
  // The pipeline runs on ACROSS tiles: in the last two K-tiles of a tile, "T+1" / "T+2" are K-tiles 0 and 1 of
  // the block's next tile (HOOK switches the source pointers between phases 2 and 3 of K-tile nk-2), so a
  // tile ends in exactly the state the next one starts from and there is no per-tile load burst or ramp.
#define VTC_KTILE(BUF, C1, K1, C2, K2, HOOK)                                                               \
  {                                                                                                        \
    /* phase 1: quadrant (0, B0); reads A0,B0 */                                                           \
    read_b0(BUF);                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    read_a(BUF, TC::OFF_A0);                                                                               \
    if (C1) issue_b1((BUF) ^ 1, (K1));                                                                     \
    sync_reads();                                                                                          \
    mma0(0);                                                                                               \
    end_phase();                                                                                           \
    /* phase 2: quadrant (0, B1); reads B1 */                                                              \
    read_b1(BUF);                                                                                          \
    if (C1) issue_a1((BUF) ^ 1, (K1));                                                                     \
    sync_reads();                                                                                          \
    mma1(0);                                                                                               \
    end_phase();                                                                                           \
    HOOK;                                                                                                  \
    /* phase 3: quadrant (1, B1); reads A1 */                                                              \
    read_a(BUF, TC::OFF_A1);                                                                               \
    if (C2) issue_a0(BUF, (K2));                                                                           \
    sync_reads();                                                                                          \
    mma1(1);                                                                                               \
    end_phase();                                                                                           \
    /* phase 4: quadrant (1, B0); registers only; K-tile boundary wait */                                  \
    if (C2) {                                                                                              \
      issue_b0(BUF, (K2));                                                                                 \
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                     \
    } else {                                                                                               \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    }                                                                                                      \
    __builtin_amdgcn_s_barrier();                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    mma0(1);                                                                                               \
    end_phase();                                                                                           \
  }

  bool first_tile = true;
#ifdef VTC_STAMPS
  long long stamp_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int tile_it = 0;
#endif

  while (tl < ntiles) {
    const unsigned tcur = xcd_remap(tl, ntiles);
    int tile_m, tile_n;
    tile_coords(p, tcur, tile_m, tile_n);

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc0[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ1; ++j) acc1[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }

    if (first_tile) {   // later tiles arrive here in the steady state (see VTC_KTILE)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      first_tile = false;
    }
    VTC_STAMP(0);
    const unsigned tl_next = tl + G;
    const bool has_next = tl_next < ntiles;

    // Stagger: wave row 1 (waves 4-7, the SIMD partners of waves 0-3) runs one barrier behind, so that on
    // every SIMD one wave is in its MFMA cluster while its partner reads LDS / issues loads.
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t + 2 < nk; t += 2) {
      VTC_KTILE(0, true, t + 1, true, t + 2, )
      VTC_KTILE(1, true, t + 2, true, t + 3, )
    }
    VTC_KTILE(0, true, nk - 1, has_next, 0, if (has_next) {
      int tmn; int tnn;
      tile_coords(p, xcd_remap(tl_next, ntiles), tmn, tnn);
      set_sources(tmn, tnn);
    })
    VTC_KTILE(1, has_next, 0, has_next, 1, )
    if (wm == 0) __builtin_amdgcn_s_barrier();
    VTC_STAMP(1);
    tl = tl_next;
    VTC_STAMP(2);

    // ---- epilogue, registers only.  C^T accumulators: lane (fr, fq) owns output row i*16 + fr of each 16-row
    // group and columns 8*fq .. +8 of its B0 group (acc0[..][j=0,1][r]) plus 8*fq .. +8 (BN = 256) or
    // 4*fq .. +4 (BN = 192) of its B1 group.  Global accesses go through buffer descriptors: rows outside the
    // matrix get the offset 0x80000000 and the hardware drops the access (loads return 0), so there are no
    // branches and vmcnt stays countable.  The residual of the whole tile is fetched before the first store.
    const int ccol_z = z * p.c_col_z;
    constexpr int ESZ = (EPI & EPI_F32) ? 4 : 2;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int W1 = 4 * NJ1;   // columns per lane in the B1 group
    const __amdgpu_buffer_rsrc_t rc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)((long long)p.M * p.ldc * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.resid, 0, (EPI & EPI_RESID16) ? (int)((long long)p.M * p.ldc * 2) : 0, 0x00020000);
    const int col0 = ccol_z + tile_n * BN + wn * 32 + 8 * fq;
    const int col1 = ccol_z + tile_n * BN + 128 + wn * (16 * NJ1) + W1 * fq;
    float bv0[8], bv1[W1];
#pragma unroll
    for (int k = 0; k < 8; ++k) bv0[k] = 0.f;
#pragma unroll
    for (int k = 0; k < W1; ++k) bv1[k] = 0.f;
    if constexpr (EPI & EPI_BIAS) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 b = *(const float4*)(p.bias + col0 + 4 * q);
        bv0[4 * q] = b.x; bv0[4 * q + 1] = b.y; bv0[4 * q + 2] = b.z; bv0[4 * q + 3] = b.w;
      }
#pragma unroll
      for (int q = 0; q < NJ1; ++q) {
        const float4 b = *(const float4*)(p.bias + col1 + 4 * q);
        bv1[4 * q] = b.x; bv1[4 * q + 1] = b.y; bv1[4 * q + 2] = b.z; bv1[4 * q + 3] = b.w;
      }
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned rowoff[8];   // element offset of (row, column 0 of the z slice), or OOB
    unsigned rowidx[8];   // the row itself, or OOB (offsets into the per-row statistics)
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int row = tile_m * 256 + (g >> 2) * 128 + wm * 64 + (g & 3) * 16 + fr;
      rowoff[g] = row < p.M ? (unsigned)(row * p.ldc) : OOB;
      rowidx[g] = row < p.M ? (unsigned)row : OOB;
    }
    // 16-bit outputs leave through LDS.  Straight from the C^T accumulators a store instruction writes 16 rows x 64 B (or
    // 32 B): partial 128-B lines, which a CU retires at ~10 B/clk however many CUs store (tools/store_bench: 9.8 k cycles
    // per 256 x 192 tile, against 2.7 k for the same bytes as whole lines).  So the four waves of a wave row put GPR row
    // groups (16 rows x BN columns each) into an LDS region, swizzled so that neither side has bank conflicts, and after one
    // barrier read them back as whole rows: every store instruction then writes eight full lines.  Two regions alternate,
    // so there is one barrier per round (the lgkmcnt(0) in front of it also retires the reads of the round before last).
    constexpr int GPR = TC::GPR, PITCH = TC::PITCH, CPR = BN / 8, NRD = GPR * 16 * CPR / 256;
    [[maybe_unused]] auto eswz = [](int chunk, int l) {   // 16-B chunk of row l -> its place in the LDS row
      return BN == 192 ? ((chunk & ~7) | ((chunk ^ (l >> 1)) & 7)) : ((chunk & ~15) | ((chunk ^ l) & 15));
    };
    [[maybe_unused]] u32x4 po0[8];                                  // packed outputs of the 8 row groups: B0 columns
    [[maybe_unused]] u32x4 po1[NJ1 == 2 ? 8 : 4];                   // B1 columns (BN = 192: two row groups per entry)
    [[maybe_unused]] auto put = [&](int g, u32x4 o0, u32x4 o1) {    // arithmetic first, for all groups; LDS rounds after
      po0[g] = o0;
      if constexpr (NJ1 == 2) po1[g] = o1;
      else { po1[g >> 1][2 * (g & 1)] = o1[0]; po1[g >> 1][2 * (g & 1) + 1] = o1[1]; }
    };
    [[maybe_unused]] int el_ = 0;   // lane id from an opaque source: values derived from it are not kept live across the K loop
    if constexpr (OUT_LDS) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el_));
    [[maybe_unused]] char* const ebase = smem + TC::SMEM + wm * (GPR * 16 * PITCH);
    [[maybe_unused]] auto write_round = [&](int rd) {
      const int fr_ = el_ & 15, fq_ = el_ >> 4;
      char* const wbase = ebase + fr_ * PITCH + (rd & 1) * TC::REGION;
      const int w0 = eswz(wn * 4 + fq_, fr_) * 16;
      const int w1 = NJ1 == 2 ? eswz(16 + wn * 4 + fq_, fr_) * 16 : eswz(16 + wn * 2 + (fq_ >> 1), fr_) * 16 + (fq_ & 1) * 8;
#pragma unroll
      for (int gg = 0; gg < GPR; ++gg) {
        const int g = rd * GPR + gg;
        char* r = wbase + gg * 16 * PITCH;
        *(u32x4*)(r + w0) = po0[g];
        if constexpr (NJ1 == 2) *(u32x4*)(r + w1) = po1[g];
        else *(u32x2*)(r + w1) = u32x2{po1[g >> 1][2 * (g & 1)], po1[g >> 1][2 * (g & 1) + 1]};
      }
    };
    // one round: wait for everybody's writes of round rd, read it back as whole rows, put the next round's rows into the
    // other region (its readers have passed this barrier), then store
    [[maybe_unused]] auto move_round = [&](int rd, bool write_next) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const char* rb = ebase + (rd & 1) * TC::REGION;
      const int g0 = rd * GPR;
      u32x4 v[NRD];
      int l[NRD], c[NRD];
#pragma unroll
      for (int k = 0; k < NRD; ++k) {
        const int q = (wn * NRD + k) * 64 + el_;
        l[k] = q / CPR; c[k] = q - l[k] * CPR;
        v[k] = *(const u32x4*)(rb + l[k] * PITCH + eswz(c[k], l[k]) * 16);
      }
      if (write_next) write_round(rd + 1);
#pragma unroll
      for (int k = 0; k < NRD; ++k) {
        const int row = tile_m * 256 + (g0 >> 2) * 128 + wm * 64 + (g0 & 3) * 16 + l[k];
#ifdef VTC_ABL_NOSTORE
        const unsigned off = OOB;
#else
        const unsigned off = row < p.M ? (unsigned)(row * p.ldc + ccol_z + tile_n * BN) * 2u + (unsigned)c[k] * 16u : OOB;
#endif
        __builtin_amdgcn_raw_buffer_store_b128(v[k], rc, off, 0, VTC_STORE_AUX);
      }
    };
    // the tile's row groups through arithmetic (group(g) ends in put(g, ..)) and out: either all arithmetic first, or
    // round by round with the arithmetic of round r + 1 in front of the barrier of round r, so that the LDS latencies,
    // the barrier skew and the store issue of a round hide under the next round's VALU work
    // (before_stores() runs once, in front of the first global store: the place for the wait on loads that the first row
    // groups did not need, see EPI_LNRES)
    [[maybe_unused]] auto run_rounds = [&](auto&& group, auto&& before_stores) {
      constexpr int R = 8 / GPR;
      if constexpr (VTC_EPI_MODE == 0) {
#pragma unroll
        for (int g = 0; g < 8; ++g) group(g);
        VTC_STAMP(5);
        before_stores();
        write_round(0);
#pragma unroll
        for (int rd = 0; rd < R; ++rd) move_round(rd, rd + 1 < R);
      } else {
#pragma unroll
        for (int gg = 0; gg < GPR; ++gg) group(gg);
        write_round(0);
#pragma unroll
        for (int rd = 0; rd < R; ++rd) {
          if (rd + 1 < R) {
#pragma unroll
            for (int gg = 0; gg < GPR; ++gg) group((rd + 1) * GPR + gg);
          }
          if (rd == 0) before_stores();
          move_round(rd, rd + 1 < R);
        }
      }
    };
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rstat = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.ln_stats, 0, (EPI & (EPI_LNIN | EPI_LNRES)) ? (int)((long long)p.M * 8) : 0, 0x00020000);
    if constexpr (EPI & EPI_LNRES) {
      // ---- residual = LayerNorm(u) from the 16-bit pre-LN rows and their statistics; partial statistics of the new rows
      const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)p.res_h, 0, (int)((long long)p.M * p.ldc * 2), 0x00020000);
      const __amdgpu_buffer_rsrc_t rpart =
          __builtin_amdgcn_make_buffer_rsrc((void*)p.part, 0, (int)(((long long)15 * p.part_rows + p.M) * 8), 0x00020000);
      float gv0[8], gv1[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 t = *(const float4*)(p.res_g + col0 + 4 * q);
        gv0[4 * q] = t.x; gv0[4 * q + 1] = t.y; gv0[4 * q + 2] = t.z; gv0[4 * q + 3] = t.w;
      }
      {
        const float4 t = *(const float4*)(p.res_g + col1);
        gv1[0] = t.x; gv1[1] = t.y; gv1[2] = t.z; gv1[3] = t.w;
      }
      const unsigned part_grp = (unsigned)((tile_n * 4 + wn) * p.part_rows);   // first row of this wave's column group
      // vmcnt counts loads and stores alike and the compiler has to assume they complete out of order, so any wait for a
      // load while a store is pending is a full drain (vmcnt(0)): it then waits for the store acknowledgements of the whole
      // chip's output burst, microseconds per tile (measured: a second pass of loads behind the first pass's stores costs
      // 15 us per launch).  Hence every load of the tile is issued and waited for (the empty asm "uses" each register while
      // only loads are pending, so the wait is a counted one) before the first store.
      auto row_of = [&](int g) { return tile_m * 256 + (g >> 2) * 128 + wm * 64 + (g & 3) * 16 + fr; };
      u32x4 h0[8];
      u32x2 h1[8], st[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int row = row_of(g);
        const unsigned ro = row < p.M ? (unsigned)(row * p.ldc) : OOB;
#if defined(VTC_ABL_LNRES) && (VTC_ABL_LNRES & 2)   // tools/gemm_bench timing ablation (results are garbage): no residual / statistics loads
        h0[g] = u32x4{(unsigned)ro, 1u, 2u, 3u}; h1[g] = u32x2{4u, 5u}; st[g] = u32x2{0u, 0x3f800000u};
#else
        h0[g] = __builtin_amdgcn_raw_buffer_load_b128(rh, ro == OOB ? OOB : (ro + (unsigned)col0) * 2u, 0, 0);
        h1[g] = __builtin_amdgcn_raw_buffer_load_b64(rh, ro == OOB ? OOB : (ro + (unsigned)col1) * 2u, 0, 0);
        st[g] = __builtin_amdgcn_raw_buffer_load_b64(rstat, row < p.M ? (unsigned)row * 8u : OOB, 0, 0);
#endif
      }
      // The first row groups start as soon as THEIR loads are back (loads return in order and nothing else is pending, so
      // the compiler's waits are counted ones); the rest of the burst arrives under that arithmetic.  The partial statistics
      // of those groups are held back, because a store in flight would turn every later load wait into a drain.
      constexpr int kEarly = VTC_EPI_MODE == 0 ? 8 : 2 * GPR;   // groups computed before the first store
      u32x2 pm_keep[kEarly];
      VTC_STAMP(4);
      run_rounds([&](int g) {
        const int mq = g >> 2, i = g & 3;
        const int row = row_of(g);
        const float a_r = __uint_as_float(st[g][1]);                 // rstd
        const float b_r = -__uint_as_float(st[g][0]) * a_r;          // -mean * rstd
        float v[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float ua = from_h16<F16>((h16)(h0[g][k] & 0xffffu)), ub = from_h16<F16>((h16)(h0[g][k] >> 16));
          const int c = 2 * k;
          v[c] = fmaf(gv0[c], fmaf(a_r, ua, b_r), acc0[mq][i][c >> 2][c & 3] + bv0[c]);
          v[c + 1] = fmaf(gv0[c + 1], fmaf(a_r, ub, b_r), acc0[mq][i][(c + 1) >> 2][(c + 1) & 3] + bv0[c + 1]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float ua = from_h16<F16>((h16)(h1[g][k] & 0xffffu)), ub = from_h16<F16>((h16)(h1[g][k] >> 16));
          const int c = 2 * k;
          v[8 + c] = fmaf(gv1[c], fmaf(a_r, ua, b_r), acc1[mq][i][0][c] + bv1[c]);
          v[8 + c + 1] = fmaf(gv1[c + 1], fmaf(a_r, ub, b_r), acc1[mq][i][0][c + 1] + bv1[c + 1]);
        }
        // statistics of the row over this wave's 48 columns (4 lanes x 12), fixed order
#if defined(VTC_ABL_LNRES) && (VTC_ABL_LNRES & 1)   // timing ablation: no row statistics
        const float mean = v[0], qd = v[1];
#else
        float sm = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) + ((v[8] + v[9]) + (v[10] + v[11]));
        sm += __shfl_xor(sm, 16);
        sm += __shfl_xor(sm, 32);
        const float mean = sm * (1.f / 48.f);
        float qd = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) { const float d = v[k] - mean; qd = fmaf(d, d, qd); }
        qd += __shfl_xor(qd, 16);
        qd += __shfl_xor(qd, 32);
#endif
        const u32x2 pm = {__float_as_uint(mean), __float_as_uint(qd)};
        const u32x4 o0 = {pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]), pack2<F16>(v[4], v[5]), pack2<F16>(v[6], v[7])};
        const u32x2 o1 = {pack2<F16>(v[8], v[9]), pack2<F16>(v[10], v[11])};
        put(g, o0, u32x4{o1[0], o1[1], 0u, 0u});
        if (g < kEarly) pm_keep[g] = pm;
        else __builtin_amdgcn_raw_buffer_store_b64(pm, rpart, (row >= p.M || fq != 0) ? OOB : (part_grp + (unsigned)row) * 8u, 0, 0);
      }, [&]() {
#pragma unroll
        for (int g = 0; g < 8; ++g) asm volatile("" ::"v"(h0[g]), "v"(h1[g]), "v"(st[g]));   // every load is back: stores may start
#pragma unroll
        for (int g = 0; g < kEarly; ++g) {
          const int row = row_of(g);
          __builtin_amdgcn_raw_buffer_store_b64(pm_keep[g], rpart, (row >= p.M || fq != 0) ? OOB : (part_grp + (unsigned)row) * 8u, 0, 0);
        }
      });
    } else {
    [[maybe_unused]] float ln_mu[8], ln_rs[8], cs0[8], cs1[W1];
    if constexpr (EPI & EPI_LNIN) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rstat, rowidx[g] == OOB ? OOB : rowidx[g] * 8u, 0, 0);
        ln_mu[g] = __uint_as_float(t[0]); ln_rs[g] = __uint_as_float(t[1]);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 t = *(const float4*)(p.colsum + col0 + 4 * q);
        cs0[4 * q] = t.x; cs0[4 * q + 1] = t.y; cs0[4 * q + 2] = t.z; cs0[4 * q + 3] = t.w;
      }
#pragma unroll
      for (int q = 0; q < NJ1; ++q) {
        const float4 t = *(const float4*)(p.colsum + col1 + 4 * q);
        cs1[4 * q] = t.x; cs1[4 * q + 1] = t.y; cs1[4 * q + 2] = t.z; cs1[4 * q + 3] = t.w;
      }
      // wait for all of them now, while only loads are pending (see EPI_LNRES above): a first use behind the first
      // store would be a full drain
#pragma unroll
      for (int g = 0; g < 8; ++g) asm volatile("" ::"v"(ln_mu[g]), "v"(ln_rs[g]));
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(cs0[k]));
#pragma unroll
      for (int k = 0; k < W1; ++k) asm volatile("" ::"v"(cs1[k]));
    }
#ifdef VTC_ABL_NOSTORE   // tools/gemm_bench ablation: every store is dropped by the bounds check
#define VTC_OFF_ST(g, col, sz) OOB
#else
#define VTC_OFF_ST(g, col, sz) (rowoff[g] == OOB ? OOB : (rowoff[g] + (unsigned)(col)) * (sz))
#endif
#define VTC_OFF_LD(g, col, sz) (rowoff[g] == OOB ? OOB : (rowoff[g] + (unsigned)(col)) * (sz))
    u32x4 res0[8];
    u32x4 res1[8];
    if constexpr (EPI & EPI_RESID16) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        res0[g] = __builtin_amdgcn_raw_buffer_load_b128(rr, VTC_OFF_LD(g, col0, 2u), 0, 0);
        if constexpr (NJ1 == 2) {
          res1[g] = __builtin_amdgcn_raw_buffer_load_b128(rr, VTC_OFF_LD(g, col1, 2u), 0, 0);
        } else {
          const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rr, VTC_OFF_LD(g, col1, 2u), 0, 0);
          res1[g] = u32x4{t[0], t[1], 0u, 0u};
        }
      }
    }
    auto group = [&](int g) {
      const int mq = g >> 2, i = g & 3;
      float v0[8], v1[W1];
      if constexpr (EPI & EPI_LNIN) {   // LN(u) W^T + b = rstd * (u W'^T - mean * colsum) + b'
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v0[4 * j + r] = fmaf(ln_rs[g], fmaf(-ln_mu[g], cs0[4 * j + r], acc0[mq][i][j][r]), bv0[4 * j + r]);
#pragma unroll
        for (int j = 0; j < NJ1; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v1[4 * j + r] = fmaf(ln_rs[g], fmaf(-ln_mu[g], cs1[4 * j + r], acc1[mq][i][j][r]), bv1[4 * j + r]);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v0[4 * j + r] = acc0[mq][i][j][r] + bv0[4 * j + r];
#pragma unroll
        for (int j = 0; j < NJ1; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v1[4 * j + r] = acc1[mq][i][j][r] + bv1[4 * j + r];
      }
      if constexpr (EPI & EPI_GELU) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) gelu2(v0[k], v0[k + 1]);
#pragma unroll
        for (int k = 0; k < W1; k += 2) gelu2(v1[k], v1[k + 1]);
      }
      if constexpr (EPI & EPI_RESID16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v0[2 * k] += from_h16<F16>((h16)(res0[g][k] & 0xffffu));
          v0[2 * k + 1] += from_h16<F16>((h16)(res0[g][k] >> 16));
        }
#pragma unroll
        for (int k = 0; k < W1 / 2; ++k) {
          v1[2 * k] += from_h16<F16>((h16)(res1[g][k] & 0xffffu));
          v1[2 * k + 1] += from_h16<F16>((h16)(res1[g][k] >> 16));
        }
      }
      if constexpr (EPI & EPI_F32) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const u32x4 o = {__float_as_uint(v0[4 * q]), __float_as_uint(v0[4 * q + 1]), __float_as_uint(v0[4 * q + 2]),
                           __float_as_uint(v0[4 * q + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(o, rc, VTC_OFF_ST(g, col0 + 4 * q, 4u), 0, VTC_STORE_AUX);
        }
#pragma unroll
        for (int q = 0; q < NJ1; ++q) {
          const u32x4 o = {__float_as_uint(v1[4 * q]), __float_as_uint(v1[4 * q + 1]), __float_as_uint(v1[4 * q + 2]),
                           __float_as_uint(v1[4 * q + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(o, rc, VTC_OFF_ST(g, col1 + 4 * q, 4u), 0, VTC_STORE_AUX);
        }
      } else {
        const u32x4 o0 = {pack2<F16>(v0[0], v0[1]), pack2<F16>(v0[2], v0[3]), pack2<F16>(v0[4], v0[5]), pack2<F16>(v0[6], v0[7])};
        u32x4 o1 = {pack2<F16>(v1[0], v1[1]), pack2<F16>(v1[2], v1[3]), 0u, 0u};
        if constexpr (NJ1 == 2) { o1[2] = pack2<F16>(v1[4], v1[5]); o1[3] = pack2<F16>(v1[6], v1[7]); }
        if constexpr (OUT_LDS) {
          put(g, o0, o1);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(o0, rc, VTC_OFF_ST(g, col0, 2u), 0, VTC_STORE_AUX);
          if constexpr (NJ1 == 2) __builtin_amdgcn_raw_buffer_store_b128(o1, rc, VTC_OFF_ST(g, col1, 2u), 0, VTC_STORE_AUX);
          else __builtin_amdgcn_raw_buffer_store_b64(u32x2{o1[0], o1[1]}, rc, VTC_OFF_ST(g, col1, 2u), 0, VTC_STORE_AUX);
        }
      }
    };
    if constexpr (OUT_LDS) {
      run_rounds(group, []() {});
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) group(g);
    }
    }   // !EPI_LNRES
#undef VTC_OFF_ST
#undef VTC_OFF_LD
    VTC_STAMP(3);
#ifdef VTC_STAMPS
    ++tile_it;
#endif
  }
#undef VTC_KTILE
#ifdef VTC_STAMPS
  if (tid == 0 && blockIdx.y == 0)
    for (int k = 0; k < 16; ++k) g_stamps[blockIdx.x * 16 + k] = stamp_t[k];
#endif
}

template <bool F16, int EPI, int BN>
inline void launch_gemm256(GemmParams p, int nz, hipStream_t s) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = p.N / BN;
  // Column-group width: the divisor g of tiles_n with the least estimated fabric traffic.  A is re-read once per group
  // (tiles_n / g times); a group's W panel is fetched once per XCD when it fits beside the streaming A rows in the
  // 4 MiB L2 (<= 2.5 MB), otherwise once per XCD and round of tiles.
  {
    const double a_bytes = 2.0 * p.M * (p.a_row_stride < p.K ? p.a_row_stride : p.K);
    const double w_bytes = 2.0 * p.N * p.K;
    const double rounds = (double)p.tiles_m * p.tiles_n / 256.0;
    double best = 1e300;
    p.group_n = p.tiles_n;
    for (int g = 1; g <= p.tiles_n; ++g) {
      if (p.tiles_n % g) continue;
      const bool fits = (double)g * BN * p.K * 2 <= 2.5e6;
      const double est = a_bytes * (p.tiles_n / g) + w_bytes * 8.0 * (fits ? 1.0 : (rounds > 1.0 ? rounds : 1.0));
      if (est < best) { best = est; p.group_n = g; }
    }
  }
#ifdef VTC_STAMPS
  if (const char* e = getenv("VTC_GROUP_N")) {   // diagnostic override, only a positive divisor of tiles_n is taken
    const int g = atoi(e);
    if (g > 0 && p.tiles_n % g == 0) p.group_n = g;
  }
#endif
  const int ntiles = p.tiles_m * p.tiles_n;
  int cap = 256;
#ifdef VTC_STAMPS
  if (const char* e = getenv("VTC_GRID")) cap = atoi(e);   // diagnostic: run on a subset of the CUs
#endif
  dim3 grid((unsigned)(ntiles < cap ? ntiles : cap), (unsigned)nz);
  hipLaunchKernelGGL((gemm256_kernel<F16, EPI, BN>), grid, dim3(512), 0, s, p);
}

}  // namespace vtc
