// Experiment kept with its benchmark (tools/attn_bench.hip), NOT part of the library: the attention kernel with 64 queries
// per wave.  Measured on the MI355X (64 windows x 12 heads, T = 499): 85 - 88 us against 83 - 86 us for the shipped
// 32-queries-per-wave kernel, bit-identical context rows.  Halving the LDS reads per score buys nothing: at 2 waves per SIMD
// the dependent chain score MFMA -> max -> exp2 -> pack -> PV MFMA of a wave has one partner wave to hide under instead of
// three, and that costs what the LDS issue slots save.  -DVTC_ATTN64_PV_PER_HALF (P V of a half right behind its softmax,
// V fragments read per half) is no different.  DESIGN.md section 5.3.
#pragma once
#include "attention.hip.h"

namespace vtc {

// ---- 64 queries per wave (round 3).  Same arithmetic, layouts and K/V ring as attention_kernel above; what changes is the
// shape of the work: a block is 4 waves x 64 queries (two 32-query halves per wave) at 2 waves per SIMD (256 registers), so
// every K and V fragment a wave reads from LDS feeds two halves' MFMAs (half the LDS reads and half the LDS-issue slots per
// score), and one half's softmax has the other half's MFMAs to run under.  Two blocks share a CU as before (2 x 64 KiB ring).
constexpr int kAttn64Waves = 4;

template <bool F16>
__global__ __launch_bounds__(256, 2) void attention64_kernel(const h16* __restrict__ qkv, h16* __restrict__ ctx, int T, int nkt) {
  __shared__ __attribute__((aligned(16))) char smem[kAttnStages * kAttnStage];
  const int b = blockIdx.y / 12, h = blockIdx.y - b * 12;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long rowbase = (long long)b * T;

  // ---- staging: two 16-B pieces of K and two of V per thread per 64-key tile (piece p = rows 8 p .. 8 p + 7)
  int k_chunk[2], v_chunk[2], srow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    srow[i] = (wave * 2 + i) * 8 + (lane >> 3);
    const int sc = lane & 7;
    k_chunk[i] = sc ^ ((srow[i] >> 1) & 7);
    v_chunk[i] = sc ^ (((srow[i] >> 1) & 1) << 2);
  }
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = kt * 64 + srow[i];
      key = key < T ? key : T - 1;   // clamped rows are masked (K) or multiplied by P = 0 (V)
      const h16* src = qkv + (rowbase + key) * 2304 + h * 64;
      char* dst = smem + buf * kAttnStage + (wave * 2 + i) * 1024;
      __builtin_amdgcn_global_load_lds(GPTR(src + 768 + k_chunk[i] * 8), LPTR(dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(src + 1536 + v_chunk[i] * 8), LPTR(dst + 8192), 16, 0, 0);
    }
  };

  const int q0 = (blockIdx.x * kAttn64Waves + wave) * 64;
  const bool active = q0 < T;   // wave-uniform; idle waves still stage and hit the barriers
  const int r = lane & 31, hh = lane >> 5;

  u16x8 qf[2][4];
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    const int qrow = min(q0 + qh * 32 + r, T - 1);
    const h16* qsrc = qkv + (rowbase + qrow) * 2304 + h * 64 + hh * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qh][s] = *(const u16x8*)(qsrc + s * 16);
  }

  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 o0[2] = {kZero16, kZero16}, o1[2] = {kZero16, kZero16};
  float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};

  const int k_row_off = r * 128;
  const int k_sw = (r >> 1) & 7;
  const int vq = (lane & 15) >> 2;
  const int v_row_off = (4 * hh + vq) * 128;
  const int v_col = (32 * ((lane >> 4) & 1) + 8 * (lane & 3)) ^ (((vq >> 1) & 1) << 6);

#pragma unroll
  for (int t = 0; t < kAttnStages - 1; ++t)
    if (t < nkt) stage(t, t);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & (kAttnStages - 1);
    // tile kt must have landed; each thread issued 4 loads per tile, in tile order, so the younger tiles may stay in flight
    const int ahead = nkt - 1 - kt;
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + kAttnStages - 1 < nkt) stage((kt + kAttnStages - 1) & (kAttnStages - 1), kt + kAttnStages - 1);
    if (!active) continue;
    const char* Ks = smem + cur * kAttnStage;
    const char* Vs = Ks + 8192;

    f32x16 s0[2], s1[2];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int co = ((2 * st + hh) ^ k_sw) << 4;
      const u16x8 k0 = *(const u16x8*)(Ks + k_row_off + co);
      const u16x8 k1 = *(const u16x8*)(Ks + 32 * 128 + k_row_off + co);
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        if (st == 0) {
          s0[qh] = mfma32<F16>(k0, qf[qh][0], kZero16);
          s1[qh] = mfma32<F16>(k1, qf[qh][0], kZero16);
        } else {
          s0[qh] = mfma32<F16>(k0, qf[qh][st], s0[qh]);
          s1[qh] = mfma32<F16>(k1, qf[qh][st], s1[qh]);
        }
      }
    }
    if (kt == nkt - 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kt * 64 + (i & 3) + 8 * (i >> 2) + 4 * hh;
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
          if (key >= T) s0[qh][i] = -1e30f;
          if (key + 32 >= T) s1[qh][i] = -1e30f;
        }
      }
    }
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      float mx = max3f(s0[qh][0], s0[qh][1], s0[qh][2]);
#pragma unroll
      for (int i = 3; i < 15; i += 2) mx = max3f(mx, s0[qh][i], s0[qh][i + 1]);
      mx = max3f(mx, s0[qh][15], s1[qh][0]);
#pragma unroll
      for (int i = 1; i < 15; i += 2) mx = max3f(mx, s1[qh][i], s1[qh][i + 1]);
      mx = fmaxf(mx, s1[qh][15]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (__any(mx > m[qh])) {   // some query of this half raised its running maximum: rescale (exact)
        const float m_new = fmaxf(m[qh], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[qh] - m_new);
        m[qh] = m_new;
        l[qh] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[qh][i] *= alpha; o1[qh][i] *= alpha; }
      }
      float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s0[qh][i] = __builtin_amdgcn_exp2f(s0[qh][i] - m[qh]);
        s1[qh][i] = __builtin_amdgcn_exp2f(s1[qh][i] - m[qh]);
        psum0 += s0[qh][i];
        psum1 += s1[qh][i];
      }
      l[qh] += psum0 + psum1;
#ifdef VTC_ATTN64_PV_PER_HALF
      // PV of this half right behind its softmax (V fragments read once per half): the other half's softmax is independent
      // of these MFMAs and can issue under them
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          u32x4 pw;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            pw[j] = u ? pack2<F16>(s1[qh][8 * st + 2 * j], s1[qh][8 * st + 2 * j + 1])
                      : pack2<F16>(s0[qh][8 * st + 2 * j], s0[qh][8 * st + 2 * j + 1]);
          const u16x8 pf = __builtin_bit_cast(u16x8, pw);
          const char* va = Vs + (u * 32 + st * 16) * 128 + v_row_off;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int vc = v_col ^ (dt << 6);
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + vc));
            const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + 8 * 128 + vc));
            const s16x8 vv = __builtin_shufflevector(v1, v2, 0, 1, 2, 3, 4, 5, 6, 7);
            const u16x8 vf = __builtin_bit_cast(u16x8, vv);
            if (dt == 0) o0[qh] = mfma32<F16>(vf, pf, o0[qh]);
            else         o1[qh] = mfma32<F16>(vf, pf, o1[qh]);
          }
        }
      }
#endif
    }

#ifndef VTC_ATTN64_PV_PER_HALF
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        u16x8 pf[2];
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
          u32x4 pw;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            pw[j] = u ? pack2<F16>(s1[qh][8 * st + 2 * j], s1[qh][8 * st + 2 * j + 1])
                      : pack2<F16>(s0[qh][8 * st + 2 * j], s0[qh][8 * st + 2 * j + 1]);
          pf[qh] = __builtin_bit_cast(u16x8, pw);
        }
        const char* va = Vs + (u * 32 + st * 16) * 128 + v_row_off;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int vc = v_col ^ (dt << 6);
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + vc));
          const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + 8 * 128 + vc));
          const s16x8 vv = __builtin_shufflevector(v1, v2, 0, 1, 2, 3, 4, 5, 6, 7);
          const u16x8 vf = __builtin_bit_cast(u16x8, vv);
#pragma unroll
          for (int qh = 0; qh < 2; ++qh) {
            if (dt == 0) o0[qh] = mfma32<F16>(vf, pf[qh], o0[qh]);
            else         o1[qh] = mfma32<F16>(vf, pf[qh], o1[qh]);
          }
        }
      }
    }
#endif
  }
  // Epilogue through the (now idle) ring, as above: 4 KiB per 32-query half, 8 KiB per wave
  __syncthreads();
  if (!active) return;
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    float lt = l[qh];
    lt += __shfl_xor(lt, 32);
    const float inv = 1.f / lt;
    char* park = smem + (wave * 2 + qh) * 4096;
    const int sw = r & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint2 a = make_uint2(pack2<F16>(o0[qh][4 * g] * inv, o0[qh][4 * g + 1] * inv), pack2<F16>(o0[qh][4 * g + 2] * inv, o0[qh][4 * g + 3] * inv));
      const uint2 c = make_uint2(pack2<F16>(o1[qh][4 * g] * inv, o1[qh][4 * g + 1] * inv), pack2<F16>(o1[qh][4 * g + 2] * inv, o1[qh][4 * g + 3] * inv));
      *(uint2*)(park + r * 128 + ((g ^ sw) << 4) + 8 * hh) = a;
      *(uint2*)(park + r * 128 + (((4 + g) ^ sw) << 4) + 8 * hh) = c;
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    const char* park = smem + (wave * 2 + qh) * 4096;
    const int c = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3);
      const uint4 v = *(const uint4*)(park + row * 128 + ((c ^ (row & 7)) << 4));
      const int q = q0 + qh * 32 + row;
      if (q < T) *(uint4*)(ctx + (rowbase + q) * 768 + h * 64 + c * 8) = v;
    }
  }
}

}  // namespace vtc
