// Experiment (tools/attn_bench): the shipped attention kernel with ONE block barrier per TWO K/V tiles (tiles t+2, t+3 are
// staged behind the barrier in front of tile t), so that the eight waves may drift up to two tiles apart instead of meeting
// at every tile.  Same arithmetic per wave: bit-identical context rows.
#pragma once
#include "attention_pp.hip.h"

namespace vtc {

template <bool F16>
__global__ __launch_bounds__(512, 4) void attention_b2_kernel(const h16* __restrict__ qkv, h16* __restrict__ ctx, int T, int nkt) {
  __shared__ __attribute__((aligned(16))) char smem[kAttnStages * kAttnStage];
  const int b = blockIdx.y / 12, h = blockIdx.y - b * 12;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long rowbase = (long long)b * T;

  const int srow = wave * 8 + (lane >> 3), sc = lane & 7;
  const int k_chunk = sc ^ ((srow >> 1) & 7);
  const int v_chunk = sc ^ (((srow >> 1) & 1) << 2);
  auto stage = [&](int kt) {
    int key = kt * 64 + srow;
    key = key < T ? key : T - 1;
    const h16* src = qkv + (rowbase + key) * 2304 + h * 64;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LPTR(smem + (kt & (kAttnStages - 1)) * kAttnStage + wave * 1024));
    attn_pp_glds16(src + 768 + k_chunk * 8, dst);
    attn_pp_glds16(src + 1536 + v_chunk * 8, dst + 8192);
  };
  auto barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  const int q0 = (blockIdx.x * kAttnWaves + wave) * 32;
  const bool active = q0 < T;
  const int r = lane & 31, hh = lane >> 5;

#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < nkt) stage(t);
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


  f32x16 s0 = kZero16, s1 = kZero16;
  auto qk = [&](int kt) {
    const char* Ks = smem + (kt & (kAttnStages - 1)) * kAttnStage;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int co = ((2 * st + hh) ^ k_sw) << 4;
      const u16x8 k0 = *(const u16x8*)(Ks + k_row_off + co);
      const u16x8 k1 = *(const u16x8*)(Ks + 32 * 128 + k_row_off + co);
      if (st == 0) {
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
  };
  auto sm = [&]() {
    float mx = max3f(s0[0], s0[1], s0[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) mx = max3f(mx, s0[i], s0[i + 1]);
    mx = max3f(mx, s0[15], s1[0]);
#pragma unroll
    for (int i = 1; i < 15; i += 2) mx = max3f(mx, s1[i], s1[i + 1]);
    mx = fmaxf(mx, s1[15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__any(mx > m)) {
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
  };
  auto pv = [&](int kt) {
    const char* Vs = smem + (kt & (kAttnStages - 1)) * kAttnStage + 8192;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        u32x4 pw;
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
  };

  asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]));
  for (int t = 0; t < nkt; ++t) {
    if ((t & 1) == 0) {
      // tiles t and t+1 (staged two tiles ago) have landed; behind the barrier nobody reads tiles t-2, t-1 any more
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      barrier();
      if (t + 2 < nkt) stage(t + 2);
      if (t + 3 < nkt) stage(t + 3);
    }
    if (!active) continue;
    qk(t);
    sm();
    pv(t);
  }
  l += __shfl_xor(l, 32);
  const float inv = 1.f / l;
  // Epilogue through the (now idle) K/V ring: the wave parks its 32 x 64 context tile as 16-bit rows in a private
  // 4 KiB region (16-B chunk index XOR (row & 7): conflict-free for the b64 writes and the b128 reads), then every
  // lane stores 16 B with 8 lanes per 128-B row: 4 store instructions per wave instead of 16 row-scattered 8-B ones
  // (store issue, not bandwidth, is what the tail of this kernel costs).
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
}

}  // namespace vtc
