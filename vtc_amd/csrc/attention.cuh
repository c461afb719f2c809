// Multi-head self-attention for one window (T <= 512 frames, 12 heads x 64), flash-style.
// Arithmetic: softmax(q k^T) v with q pre-scaled by 64^-0.5 (HF modeling_hubert.py:234-260, 318-341).
//
// Layout choices (v_mfma_f32_32x32x16, C/D: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)):
//   * S^T = K * Q^T, so the QUERY sits on the lane and a query's keys sit in the 16 accumulator
//     registers (+ the other half-wave): row max / row sum need no LDS and one cross-half shuffle.
//   * O^T = V^T * P^T: the S^T accumulators, converted to 16 bit, ARE the B operand (no lane movement);
//     the matching V^T fragments come from the row-major V image by ds_read_b64_tr_b16.
// The whole K and V of one (window, head) live in LDS (2 x nkt*32 x 128 B <= 128 KiB).
#pragma once
#include "common.cuh"

namespace vtc {

constexpr int kAttnWaves = 8;

template <bool F16>
__global__ __launch_bounds__(512) void attention_kernel(const h16* __restrict__ qkv, h16* __restrict__ ctx, int T, int nkt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + nkt * 32 * 128;
  const int b = blockIdx.y / 12, h = blockIdx.y - b * 12;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long rowbase = (long long)b * T;

  for (int idx = tid; idx < nkt * 32 * 8; idx += 512) {
    const int key = idx >> 3, c = idx & 7;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
    if (key < T) {
      const h16* src = qkv + (rowbase + key) * 2304 + h * 64 + c * 8;
      kv = *(const uint4*)(src + 768);
      vv = *(const uint4*)(src + 1536);
    }
    *(uint4*)(Ks + key * 128 + ((c ^ ((key >> 1) & 7)) << 4)) = kv;
    *(uint4*)(Vs + key * 128 + (c << 4)) = vv;
  }
  __syncthreads();

  const int q0 = (blockIdx.x * kAttnWaves + wave) * 32;
  if (q0 >= T) return;  // whole wave leaves together (wave-uniform), after the only barrier
  const int r = lane & 31, hh = lane >> 5;

  u16x8 qf[4];
  {
    const int qrow = min(q0 + r, T - 1);
    const h16* qsrc = qkv + (rowbase + qrow) * 2304 + h * 64 + hh * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const u16x8*)(qsrc + s * 16);
  }

  f32x16 o0, o1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
  float m = -1e30f, l = 0.f;

  // V^T fragment address (bytes) of this lane inside a key tile: 4-key x 16-col block per 16-lane group
  const int v_lane_off = (4 * hh + ((lane & 15) >> 2)) * 128 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  for (int kt = 0; kt < nkt; ++kt) {
    f32x16 s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
    const int krow = kt * 32 + r;
    const char* kbase = Ks + krow * 128;
    const int ksw = (krow >> 1) & 7;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const u16x8 kf = *(const u16x8*)(kbase + (((2 * st + hh) ^ ksw) << 4));
      s = mfma32<F16>(kf, qf[st], s);
    }
    if (kt == nkt - 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (key >= T) s[i] = -1e30f;
      }
    }
    float mx = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    const float alpha = __expf(m - m_new);
    m = m_new;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = __expf(s[i] - m); psum += s[i]; }
    l = fmaf(l, alpha, psum);
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }

    const char* vt = Vs + kt * 32 * 128 + v_lane_off;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      u16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = to_h16<F16>(s[8 * st + j]);
      const char* va = vt + st * 16 * 128;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + dt * 64));
        const s16x4 v2 =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + dt * 64 + 8 * 128));
        u16x8 vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) { vf[j] = (unsigned short)v1[j]; vf[4 + j] = (unsigned short)v2[j]; }
        if (dt == 0) o0 = mfma32<F16>(vf, pf, o0);
        else         o1 = mfma32<F16>(vf, pf, o1);
      }
    }
  }

  l += __shfl_xor(l, 32);
  const float inv = 1.f / l;
  const int q = q0 + r;
  if (q < T) {
    h16* dst = ctx + (rowbase + q) * 768 + h * 64 + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *(uint2*)(dst + 8 * g) = make_uint2(pack2<F16>(o0[4 * g] * inv, o0[4 * g + 1] * inv),
                                          pack2<F16>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
      *(uint2*)(dst + 32 + 8 * g) = make_uint2(pack2<F16>(o1[4 * g] * inv, o1[4 * g + 1] * inv),
                                               pack2<F16>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
    }
  }
}

}  // namespace vtc
