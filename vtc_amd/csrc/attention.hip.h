// Multi-head self-attention for one window (12 heads x 64), flash-style, K/V streamed through LDS.
// Arithmetic: softmax(q k^T) v with q pre-scaled by 64^-0.5 (HF modeling_hubert.py:234-260, 318-341).  The QKV
// weights also carry log2(e), so the scores arrive in the log2 domain and the softmax is exp2(s - max): one
// subtract and one v_exp_f32 per score.
//
// Layout choices (v_mfma_f32_32x32x16, C/D: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)):
//   * S^T = K * Q^T, so the QUERY sits on the lane and a query's keys sit in the accumulator registers
//     (+ the other half-wave): row max / row sum need no LDS and one cross-half shuffle.
//   * O^T = V^T * P^T: the S^T accumulators, converted to 16 bit, ARE the B operand (no lane movement);
//     the matching V^T fragments come from the row-major V image by ds_read_b64_tr_b16.
//   * a block = 8 waves x 32 queries of one (window, head); K and V arrive in 64-key tiles by
//     global_load_lds into a 4-stage ring (64 KiB) with three tiles in flight and a counted vmcnt: a tile's
//     compute (~1 us) is much shorter than its load latency, so a one-deep prefetch ran at the load latency
//     (8 tiles x ~3 us per block).  K rows are XOR-swizzled for the ds_read_b128 fragment reads, V rows for
//     the transposed reads.
//   * online softmax over 64 keys per step; the running-max rescale of O is skipped when no query of the
//     wave raised its maximum (exact, wave-uniform branch).
#pragma once
#include "common.hip.h"

namespace vtc {

constexpr int kAttnWaves = 8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
constexpr int kAttnStage = 16384;  // K tile 64 x 128 B, then V tile 64 x 128 B
constexpr int kAttnStages = 4;     // ring depth: tiles kt+1 .. kt+3 are in flight while tile kt is consumed

#ifdef VTC_ATTN_STAMPS   // tools/attn_bench diagnostic build
__device__ long long g_attn_stamps[64 * 8 * 10 * 5];   // [block < 64][wave][tile < 10][point]
__device__ long long g_attn_blk[4096 * 4];             // [block][entry, first tile ready, loop end, exit] (wave 0)
#define ATTN_BLK(pt) do { if (threadIdx.x == 0) { const unsigned bi = blockIdx.y * gridDim.x + blockIdx.x; \
    if (bi < 4096) g_attn_blk[bi * 4 + (pt)] = (long long)__builtin_amdgcn_s_memrealtime(); } } while (0)
#define ATTN_STAMP(pt) do { if (blockIdx.x == 0 && blockIdx.y < 64 && kt < 10 && lane == 0) \
    g_attn_stamps[((blockIdx.y * 8 + wave) * 10 + kt) * 5 + (pt)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATTN_STAMP(pt) do { } while (0)
#define ATTN_BLK(pt) do { } while (0)
#endif

template <bool F16>
__global__ __launch_bounds__(512, 4) void attention_kernel(const h16* __restrict__ qkv, h16* __restrict__ ctx, int T, int nkt) {
  __shared__ __attribute__((aligned(16))) char smem[kAttnStages * kAttnStage];
  // XCD-aware block -> (query block, window x head) map.  Workgroups go to the 8 XCDs round-robin in launch order
  // (x fastest), so with the plain map the query blocks of one head sit on different XCDs and each fetches the head's K / V
  // through its own L2.  Here a run of 8 * nqb consecutive workgroups is 8 heads x nqb query blocks with the query block
  // changing every 8 workgroups: the blocks that share K / V share an XCD (and still start within a few microseconds of
  // each other).  Heads beyond the last full run of 8 keep the plain order.
  int qb_, bh_;
  {
    const int nqb = gridDim.x, nbh = gridDim.y;
    const int L = blockIdx.y * nqb + blockIdx.x;
    const int run = 8 * nqb;
#ifdef VTC_ATTN_PLAIN_MAP   // tools/attn_bench A/B: every head in the plain order
    const int full = 0, bh_base = 0;
#else
    const int full = (nbh / 8) * run, bh_base = (nbh / 8) * 8;
#endif
    if (L < full) {
      const int r_ = L % run;
      qb_ = r_ >> 3;
      bh_ = (L / run) * 8 + (r_ & 7);
    } else {
      qb_ = (L - full) % nqb;
      bh_ = bh_base + (L - full) / nqb;
    }
  }
  const int qblock = qb_, bh = bh_;
  const int b = bh / 12, h = bh - b * 12;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long rowbase = (long long)b * T;
  ATTN_BLK(0);

  // ---- staging: one 16-B piece of K and one of V per thread per 64-key tile
  const int srow = wave * 8 + (lane >> 3), sc = lane & 7;
  const int k_chunk = sc ^ ((srow >> 1) & 7);
  const int v_chunk = sc ^ (((srow >> 1) & 1) << 2);
  auto stage = [&](int buf, int kt) {
    int key = kt * 64 + srow;
    key = key < T ? key : T - 1;   // clamped rows are masked (K) or multiplied by P = 0 (V)
    const h16* src = qkv + (rowbase + key) * 2304 + h * 64;
    char* dst = smem + buf * kAttnStage + wave * 1024;
    __builtin_amdgcn_global_load_lds(GPTR(src + 768 + k_chunk * 8), LPTR(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GPTR(src + 1536 + v_chunk * 8), LPTR(dst + 8192), 16, 0, 0);
  };

  const int q0 = (qblock * kAttnWaves + wave) * 32;
  const bool active = q0 < T;   // wave-uniform; idle waves still stage and hit the barriers
  const int r = lane & 31, hh = lane >> 5;

  u16x8 qf[4];
  {
    const int qrow = min(q0 + r, T - 1);
    const h16* qsrc = qkv + (rowbase + qrow) * 2304 + h * 64 + hh * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const u16x8*)(qsrc + s * 16);
  }

  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 o0 = kZero16, o1 = kZero16;
  float m = -1e30f, l = 0.f;

  // K fragment: row = u*32 + r, 16-B chunk (2*st + hh) ^ swz(row);  swz(row) = (row>>1)&7 is the same for u = 0,1
  const int k_row_off = r * 128;
  const int k_sw = (r >> 1) & 7;
  // V^T fragment (4-key x 16-col block per 16-lane group): key row = 16*st2 + 4*hh + q (+8), col bytes 32*g1 + 8*p
  const int vq = (lane & 15) >> 2;
  const int v_row_off = (4 * hh + vq) * 128;
  const int v_col = (32 * ((lane >> 4) & 1) + 8 * (lane & 3)) ^ (((vq >> 1) & 1) << 6);   // dt = 0; dt = 1 is ^ 64

#pragma unroll
  for (int t = 0; t < kAttnStages - 1; ++t)
    if (t < nkt) stage(t, t);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & (kAttnStages - 1);
    // tile kt must have landed; each thread issued 2 loads per tile, in tile order, so the younger tiles may stay in flight
    ATTN_STAMP(0);
    const int ahead = nkt - 1 - kt;   // tiles issued after tile kt (at most kAttnStages - 2 here)
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tile kt visible to all waves; every wave is done with tile kt-1, whose stage is refilled next
    if (kt + kAttnStages - 1 < nkt) stage((kt + kAttnStages - 1) & (kAttnStages - 1), kt + kAttnStages - 1);
    if (kt == 0) ATTN_BLK(1);
    if (!active) continue;
    ATTN_STAMP(1);
    const char* Ks = smem + cur * kAttnStage;
    const char* Vs = Ks + 8192;

    f32x16 s0, s1;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int co = ((2 * st + hh) ^ k_sw) << 4;
      const u16x8 k0 = *(const u16x8*)(Ks + k_row_off + co);
      const u16x8 k1 = *(const u16x8*)(Ks + 32 * 128 + k_row_off + co);
      if (st == 0) {   // zero accumulator as an inline constant: no 32 v_mov per tile
        s0 = mfma32<F16>(k0, qf[0], kZero16);
        s1 = mfma32<F16>(k1, qf[0], kZero16);
      } else {
        s0 = mfma32<F16>(k0, qf[st], s0);
        s1 = mfma32<F16>(k1, qf[st], s1);
      }
    }
    if (kt == nkt - 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kt * 64 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (key >= T) s0[i] = -1e30f;
        if (key + 32 >= T) s1[i] = -1e30f;
      }
    }
    ATTN_STAMP(2);
    // row maximum: v_max3 chain (the scores are MFMA results, never signalling NaNs: no canonicalising v_max needed)
    float mx = max3f(s0[0], s0[1], s0[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) mx = max3f(mx, s0[i], s0[i + 1]);
    mx = max3f(mx, s0[15], s1[0]);
#pragma unroll
    for (int i = 1; i < 15; i += 2) mx = max3f(mx, s1[i], s1[i + 1]);
    mx = fmaxf(mx, s1[15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__any(mx > m)) {   // some query of this wave raised its running maximum: rescale (exact)
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
    }
    float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s0[i] = __builtin_amdgcn_exp2f(s0[i] - m);
      s1[i] = __builtin_amdgcn_exp2f(s1[i] - m);
      psum0 += s0[i];
      psum1 += s1[i];
    }
    l += psum0 + psum1;
    ATTN_STAMP(3);

#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        u32x4 pw;   // the S^T accumulators, 16 bit, ARE the B operand: pairs (0,1) (2,3) (4,5) (6,7) of this 8-key slice
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pw[j] = u ? pack2<F16>(s1[8 * st + 2 * j], s1[8 * st + 2 * j + 1]) : pack2<F16>(s0[8 * st + 2 * j], s0[8 * st + 2 * j + 1]);
        const u16x8 pf = __builtin_bit_cast(u16x8, pw);
        const char* va = Vs + (u * 32 + st * 16) * 128 + v_row_off;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int vc = v_col ^ (dt << 6);
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + vc));
          const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + 8 * 128 + vc));
          const s16x8 vv = __builtin_shufflevector(v1, v2, 0, 1, 2, 3, 4, 5, 6, 7);
          const u16x8 vf = __builtin_bit_cast(u16x8, vv);
          if (dt == 0) o0 = mfma32<F16>(vf, pf, o0);
          else         o1 = mfma32<F16>(vf, pf, o1);
        }
      }
    }
    ATTN_STAMP(4);
  }
  l += __shfl_xor(l, 32);
  const float inv = 1.f / l;
  // Epilogue through the (now idle) K/V ring: the wave parks its 32 x 64 context tile as 16-bit rows in a private
  // 4 KiB region (16-B chunk index XOR (row & 7): conflict-free for the b64 writes and the b128 reads), then every
  // lane stores 16 B with 8 lanes per 128-B row: 4 store instructions per wave instead of 16 row-scattered 8-B ones
  // (store issue, not bandwidth, is what the tail of this kernel costs).
  ATTN_BLK(2);
  __syncthreads();   // every wave is past its last K/V read
  if (!active) return;
  char* park = smem + wave * 4096;
  {
    const int sw = r & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // d = 8 g + 4 hh .. +4 (o0) and 32 + 8 g + 4 hh .. (o1): 8-B half (hh) of 16-B chunk g / 4 + g
      const uint2 a = make_uint2(pack2<F16>(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack2<F16>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
      const uint2 b = make_uint2(pack2<F16>(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack2<F16>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
      *(uint2*)(park + r * 128 + ((g ^ sw) << 4) + 8 * hh) = a;
      *(uint2*)(park + r * 128 + (((4 + g) ^ sw) << 4) + 8 * hh) = b;
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    const int c = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3);
      const uint4 v = *(const uint4*)(park + row * 128 + ((c ^ (row & 7)) << 4));
      const int q = q0 + row;
      if (q < T) *(uint4*)(ctx + (rowbase + q) * 768 + h * 64 + c * 8) = v;
    }
  }
#ifdef VTC_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  ATTN_BLK(3);
}

}  // namespace vtc
