// Persistent 256 x BN x 64 MFMA GEMM (BN = 256 or 192), 8 waves, 8-phase schedule (two K-tiles per loop
// iteration).  Same contract as gemm.cuh (C = A * W^T, overlapping-row A addressing, fused epilogues);
// requires K % 128 == 0 and N % BN == 0 (W rows are read unguarded).
//
// Why: tools/gemm_bench ablations show the 128 x 128 kernel is bound by the L2 -> LDS fill (it runs as fast
// with its MFMAs removed: ~13.4 TB/s of fill = 64 FLOP/B x 0.86 PFLOP/s).  A 256 x 256 tile needs half the
// fill bytes per FLOP, 256 x 192 0.58x, and BN = 192 tiles N = 768 / 2304 / 3072 without a ragged last
// round over the 256 CUs (M = 31936: 500 / 1500 / 2000 tiles).
//
// Structure (cdna_hip_programming.md section 5, "The 256^2 8-phase template"):
//   * LDS: 2 K-tile buffers x 4 half-tiles (A0,B0,B1,A1; 128 rows x 128 B each, B1 64 rows when BN = 192),
//     XOR-swizzled; plus a separate 32 KiB epilogue region (8 waves x 16 rows x 64 fp32).
//   * a wave (wm 0..1, wn 0..3) owns rows {h*128 + wm*64 ..+64}, h = 0,1, and cols {wn*32 ..+32} of B0 plus
//     {wn*NB1 ..} of B1, so its four quadrants consume the half-tiles in the order they are staged.
//   * every phase: ds_read the fragments it needs, issue ONE half-tile of global_load_lds prefetch,
//     barrier, lgkmcnt(0), MFMAs, barrier.  vmcnt is only waited at the K-tile boundary phases, with a
//     COUNT (3 half-tiles stay in flight across the barriers), never 0 in steady state.
//   * a slot is restaged at least one full phase after its last ds_read; a staged tile is first read one
//     phase after the wait + barrier that retired it.
//   * persistent: a block walks tiles b, b+G, b+2G, ...; the NEXT tile's prologue loads are issued before
//     the current tile's epilogue, which runs out of its own LDS region, so loads and stores overlap.
#pragma once
#include "common.cuh"
#include "gemm.cuh"

namespace vtc {

constexpr int kHalfBytes = 128 * 128;  // one full half-tile: 128 rows x 64 elements x 2 B
constexpr int kEpiRegion = 8 * 4096;   // 8 waves x (16 rows x 64 fp32)

template <int BN>
struct Tile256 {
  static_assert(BN == 256 || BN == 192, "BN must be 256 or 192");
  static constexpr int NJ1 = BN == 256 ? 2 : 1;       // 16-col MFMA tiles per wave in the B1 half
  static constexpr int B1_ROWS = 64 * NJ1;            // rows of W in the B1 half-tile
  static constexpr int B1_INSTR = NJ1;                // glds instructions per wave for B1
  static constexpr int OFF_A0 = 0, OFF_B0 = kHalfBytes, OFF_B1 = 2 * kHalfBytes;
  static constexpr int OFF_A1 = OFF_B1 + B1_ROWS * 128;
  static constexpr int BUF = OFF_A1 + kHalfBytes;
  static constexpr int SMEM = 2 * BUF + kEpiRegion;
};

template <bool F16, int EPI, int BN>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmParams p) {
  using TC = Tile256<BN>;
  constexpr int NJ1 = TC::NJ1;
  __shared__ __attribute__((aligned(16))) char smem[TC::SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane >> 3, lc = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  const int z = blockIdx.y;
  const unsigned ntiles = (unsigned)(p.tiles_m * p.tiles_n);

  // per-lane global sources of the tile being staged: A0, A1, B0 (2 instructions each), B1 (NJ1)
  const char* src_a0[2];
  const char* src_a1[2];
  const char* src_b0[2];
  const char* src_b1[NJ1];

  auto set_sources = [&](int tile_m, int tile_n) {
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
      src_b0[i] = (const char*)(p.W + ((long long)z * p.w_z_stride + ((long long)tile_n * BN + r) * p.K + sw));
    }
#pragma unroll
    for (int i = 0; i < NJ1; ++i) {
      const int r = (wave * NJ1 + i) * 8 + lr;  // row inside the B1 half-tile
      const long long sw = (long long)((lc ^ lds_swz(r)) * 8);
      src_b1[i] = (const char*)(p.W + ((long long)z * p.w_z_stride + ((long long)tile_n * BN + 128 + r) * p.K + sw));
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

  auto read_a = [&](int buf, int off) {
    const char* base = smem + buf * TC::BUF + off;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i] = *(const u16x8*)(base + a_row[i] + (((ks * 4 + fq) ^ a_sw[i]) << 4));
  };
  auto read_b0 = [&](int buf) {
    const char* base = smem + buf * TC::BUF + TC::OFF_B0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) fb0[ks][j] = *(const u16x8*)(base + b0_row[j] + (((ks * 4 + fq) ^ b0_sw[j]) << 4));
  };
  auto read_b1 = [&](int buf) {
    const char* base = smem + buf * TC::BUF + TC::OFF_B1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < NJ1; ++j) fb1[ks][j] = *(const u16x8*)(base + b1_row[j] + (((ks * 4 + fq) ^ b1_sw[j]) << 4));
  };
  auto mma0 = [&](int mq) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc0[mq][i][j] = mfma16<F16>(fa[ks][i], fb0[ks][j], acc0[mq][i][j]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma1 = [&](int mq) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ1; ++j) acc1[mq][i][j] = mfma16<F16>(fa[ks][i], fb1[ks][j], acc1[mq][i][j]);
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
  auto prologue = [&]() {  // all of K-tile 0, A0 + B0 of K-tile 1 (4 glds stay in flight at the first wait)
    issue_a0(0, 0); issue_b0(0, 0); issue_b1(0, 0); issue_a1(0, 0);
    issue_a0(1, 1); issue_b0(1, 1);
  };

  const int nk = p.K >> 6;  // even, >= 2
  const unsigned G = gridDim.x;
  unsigned tl = blockIdx.x;
  {
    const unsigned t0 = xcd_remap(tl, ntiles);
    set_sources((int)(t0 / (unsigned)p.tiles_n), (int)(t0 % (unsigned)p.tiles_n));
  }
  prologue();

  // Staging order per K-tile T (one half-tile per phase): B1(T+1), A1(T+1), A0(T+2), B0(T+2).  Every slot is
  // restaged >= 2 phases after its last ds_read, which is what the half-phase stagger between the two wave
  // rows (below) needs; at the K-tile boundary wait A0(T+2) and B0(T+2) (4 glds) stay in flight.
#define VTC_KTILE(BUF, T)                                                                                  \
  {                                                                                                        \
    /* phase 1: quadrant (0, B0); reads A0,B0 */                                                           \
    read_b0(BUF);                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    read_a(BUF, TC::OFF_A0);                                                                               \
    if ((T) + 1 < nk) issue_b1((BUF) ^ 1, (T) + 1);                                                        \
    sync_reads();                                                                                          \
    mma0(0);                                                                                               \
    end_phase();                                                                                           \
    /* phase 2: quadrant (0, B1); reads B1 */                                                              \
    read_b1(BUF);                                                                                          \
    if ((T) + 1 < nk) issue_a1((BUF) ^ 1, (T) + 1);                                                        \
    sync_reads();                                                                                          \
    mma1(0);                                                                                               \
    end_phase();                                                                                           \
    /* phase 3: quadrant (1, B1); reads A1 */                                                              \
    read_a(BUF, TC::OFF_A1);                                                                               \
    if ((T) + 2 < nk) issue_a0(BUF, (T) + 2);                                                              \
    sync_reads();                                                                                          \
    mma1(1);                                                                                               \
    end_phase();                                                                                           \
    /* phase 4: quadrant (1, B0); registers only; K-tile boundary wait */                                  \
    if ((T) + 2 < nk) {                                                                                    \
      issue_b0(BUF, (T) + 2);                                                                              \
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                     \
    } else {                                                                                               \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    }                                                                                                      \
    __builtin_amdgcn_s_barrier();                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    mma0(1);                                                                                               \
    end_phase();                                                                                           \
  }

  float* epi = (float*)(smem + 2 * TC::BUF + wave * 4096);

  while (tl < ntiles) {
    const unsigned tcur = xcd_remap(tl, ntiles);
    const int tile_n = (int)(tcur % (unsigned)p.tiles_n);
    const int tile_m = (int)(tcur / (unsigned)p.tiles_n);

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc0[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ1; ++j) acc1[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }

    // this tile's prologue loads (and the previous tile's epilogue stores) must have landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // Stagger: wave row 1 (waves 4-7, the SIMD partners of waves 0-3) runs one barrier behind, so that on
    // every SIMD one wave is in its MFMA cluster while its partner reads LDS / issues loads.
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nk; t += 2) {
      VTC_KTILE(0, t)
      VTC_KTILE(1, t + 1)
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();

    // every wave is past its last ds_read (final barrier): stage the NEXT tile before this tile's epilogue
    tl += G;
    if (tl < ntiles) {
      const unsigned tn = xcd_remap(tl, ntiles);
      set_sources((int)(tn / (unsigned)p.tiles_n), (int)(tn % (unsigned)p.tiles_n));
      prologue();
    }

    // ---- epilogue: 8 passes of 16 rows x (32 + 16*NJ1) cols through the wave's private LDS region.
    // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg.  LDS image [16][64] fp32, the
    // 16-B chunk index XORed with (row>>2)<<2 so that the four row groups of a write hit different banks.
    const int ccol_z = z * p.c_col_z;
    // Global accesses go through buffer descriptors: lanes outside the matrix get the offset 0x80000000 and the
    // hardware drops the access (loads return 0), so the epilogue has no branches and vmcnt stays countable.
    constexpr int ESZ = (EPI & EPI_F32) ? 4 : 2;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)((long long)p.M * p.ldc * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.resid, 0, (EPI & (EPI_RESID | EPI_RESID16)) ? (int)((long long)p.M * p.ldc * ((EPI & EPI_RESID16) ? 2 : 4)) : 0,
        0x00020000);
    // per-lane output columns and bias (fp32 out: 4 columns per lane; 16-bit out: 8 columns per lane)
    constexpr int CPL = (EPI & EPI_F32) ? 4 : 8;
    constexpr int NIT = (EPI & EPI_F32) ? 4 : 2;          // row iterations per 16-row pass
    constexpr int RPI = 16 / NIT;                         // rows per iteration
    const int lrow = (EPI & EPI_F32) ? (lane >> 4) : (lane >> 3);
    const int lch = (EPI & EPI_F32) ? (lane & 15) : (lane & 7);
    const int c0 = lch * CPL;
    const int col = tile_n * BN + (c0 < 32 ? wn * 32 + c0 : 128 + wn * (16 * NJ1) + (c0 - 32));
    const bool col_ok = c0 < 32 + 16 * NJ1;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI & EPI_BIAS) {
      const float4 b0 = *(const float4*)(p.bias + ccol_z + (col_ok ? col : 0));
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
      if constexpr (CPL == 8) {
        const float4 b1 = *(const float4*)(p.bias + ccol_z + (col_ok ? col : 0) + 4);
        bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
      }
    }
    auto elem_index = [&](int pass, int it) -> unsigned {   // element index of this lane's first column, or OOB
      const int row = tile_m * 256 + (pass >> 2) * 128 + wm * 64 + (pass & 3) * 16 + it * RPI + lrow;
      return (row < p.M && col_ok) ? (unsigned)(row * p.ldc + ccol_z + col) : OOB;
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // residual of the NEXT pass is fetched before this pass's stores are issued (one pass of prefetch)
    u32x2 res16[2][4];
    u32x4 res32[2][4];
    auto fetch_resid = [&](int pass, int slot) {
      if constexpr (EPI & EPI_RESID16) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const unsigned e = elem_index(pass, it);
          res16[slot][it] = __builtin_amdgcn_raw_buffer_load_b64(rr, e == OOB ? OOB : e * 2u, 0, 0);
        }
      }
      if constexpr (EPI & EPI_RESID) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const unsigned e = elem_index(pass, it);
          res32[slot][it] = __builtin_amdgcn_raw_buffer_load_b128(rr, e == OOB ? OOB : e * 4u, 0, 0);
        }
      }
    };
    fetch_resid(0, 0);
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int mq = pass >> 2, i = pass & 3, slot = pass & 1;
      if (pass + 1 < 8) fetch_resid(pass + 1, slot ^ 1);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = fq * 4 + r;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = j * 16 + fr;
          epi[row * 64 + ((((c >> 2) ^ (fq << 2)) << 2) | (c & 3))] = acc0[mq][i][j][r];
        }
#pragma unroll
        for (int j = 0; j < NJ1; ++j) {
          const int c = 32 + j * 16 + fr;
          epi[row * 64 + ((((c >> 2) ^ (fq << 2)) << 2) | (c & 3))] = acc1[mq][i][j][r];
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int r = it * RPI + lrow;
        const int sw = (r >> 2) << 2;
        const unsigned e = elem_index(pass, it);
        float v[8];
        if constexpr (EPI & EPI_F32) {
          const float4 x = *(const float4*)(epi + r * 64 + ((lch ^ sw) << 2));
          v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        } else {
          const float4 x0 = *(const float4*)(epi + r * 64 + (((2 * lch) ^ sw) << 2));
          const float4 x1 = *(const float4*)(epi + r * 64 + (((2 * lch + 1) ^ sw) << 2));
          v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          v[k] += bv[k];
          if constexpr (EPI & EPI_GELU) v[k] = gelu_f(v[k]);
        }
        if constexpr (EPI & EPI_F32) {
          if constexpr (EPI & EPI_RESID16) {
            const u32x2 q = res16[slot][it];
            v[0] += from_h16<F16>((h16)(q[0] & 0xffffu)); v[1] += from_h16<F16>((h16)(q[0] >> 16));
            v[2] += from_h16<F16>((h16)(q[1] & 0xffffu)); v[3] += from_h16<F16>((h16)(q[1] >> 16));
          }
          if constexpr (EPI & EPI_RESID) {
            const u32x4 q = res32[slot][it];
            v[0] += __uint_as_float(q[0]); v[1] += __uint_as_float(q[1]);
            v[2] += __uint_as_float(q[2]); v[3] += __uint_as_float(q[3]);
          }
          const u32x4 o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
          __builtin_amdgcn_raw_buffer_store_b128(o, rc, e == OOB ? OOB : e * 4u, 0, 0);
        } else {
          const u32x4 o = {pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]), pack2<F16>(v[4], v[5]), pack2<F16>(v[6], v[7])};
          __builtin_amdgcn_raw_buffer_store_b128(o, rc, e == OOB ? OOB : e * 2u, 0, 0);
        }
      }
    }
  }
#undef VTC_KTILE
}

template <bool F16, int EPI, int BN>
inline void launch_gemm256(GemmParams p, int nz, hipStream_t s) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = p.N / BN;
  const int ntiles = p.tiles_m * p.tiles_n;
  dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256), (unsigned)nz);
  hipLaunchKernelGGL((gemm256_kernel<F16, EPI, BN>), grid, dim3(512), 0, s, p);
}

}  // namespace vtc
