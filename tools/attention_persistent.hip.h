// Experiment only (tools/attn_bench): a persistent form of vtc_amd/csrc/attention.hip.h's kernel.  Bit-identical to it, and
// SLOWER on MI355X (64 windows x T = 499: 91.3 us against 83.9 us; 128 x 199: 49.4 against 43.0; one item per block: 17.3
// against 13.7), so it is not used.  What it showed: the 23 % of a block's life spent in prologue and epilogue are hidden
// already, by the second block resident on the CU; the tile loop is bound by what a SIMD issues for its four waves (MFMA
// plus the softmax VALU and transcendental work), not by the K/V ring, so keeping the ring full across items buys nothing
// and the two extra barrier rounds per item (the Q rows travel through the ring) cost ~3 us per item.
#pragma once
#include "attention.hip.h"

namespace vtc {

// global_load_lds as inline assembly: the compiler then knows of no LDS-DMA in flight, and adds no vmcnt(0) of its own in
// front of the LDS accesses it cannot tell apart from the DMA's target (the ds_read_tr builtins, the park): with the builtin
// it did, before every first V read of a tile, which made the ring one tile deep whatever was staged ahead.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_wave_base) : "memory", "m0");
}

// ---- Persistent form (round 3).  Same arithmetic and tile loop as attention_kernel; what changes is what surrounds it.
// Measured on the one-shot kernel (tools/attn_bench with stamps, 64 windows x 12 heads x T = 499): a block lives 27.6 us, of
// which 4.6 us are its prologue (Q fragments and the first K/V tile from far memory, nothing to overlap with) and 1.7 us its
// epilogue, and a CU runs three rounds of such blocks back to back: 23 % of the kernel.  Here 2 blocks per CU stay resident
// and walk the (query block, window, head) items b, b + G, ...; the K/V ring never drains between items: an item is a
// stream of nkt + 2 pieces (two 16 KiB slots of Q rows, then the K/V tiles), piece p + 3 is staged while piece p is
// consumed, so the next item's Q and first tiles arrive under the current item's last tiles.  Q reaches the registers
// through LDS (a wave reads its 32 rows when its slot comes up), the context tile leaves through a wave-private 2 KiB
// park in two passes of 16 rows (no block barrier), and the counted vmcnt allows for the epilogue's four stores while they
// are among the eight youngest operations (vmcnt counts stores too; they are never waited for).
constexpr int kAttnPark = 8 * 2048;

template <bool F16>
__global__ __launch_bounds__(512, 4) void attention_persistent_kernel(const h16* __restrict__ qkv, h16* __restrict__ ctx, int T,
                                                                      int nkt, int nqb, int n_items) {
  __shared__ __attribute__((aligned(16))) char smem[kAttnStages * kAttnStage + kAttnPark];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int n_my = ((int)blockIdx.x < n_items) ? (n_items - 1 - (int)blockIdx.x) / G + 1 : 0;
  const int npi = nkt + 2;            // pieces per item
  const int P = n_my * npi;           // pieces of this block's stream

  const int srow = wave * 8 + (lane >> 3), sc = lane & 7;
  const int k_chunk = sc ^ ((srow >> 1) & 7);
  const int v_chunk = sc ^ (((srow >> 1) & 1) << 2);
  // piece k of item `id` into ring slot `buf`: k = 0, 1 the item's 256 Q rows (128 each, row-major, unswizzled), k >= 2 K/V tile k - 2
  auto stage = [&](int buf, int id, int k) {
    const int qb = id % nqb, bh = id / nqb;
    const int b = bh / 12, h = bh - b * 12;
    const long long rowbase = (long long)b * T;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LPTR(smem + buf * kAttnStage + wave * 1024));
    if (k < 2) {
      int q = qb * 256 + k * 128 + srow;
      int q2 = q + 64;
      q = q < T ? q : T - 1;
      q2 = q2 < T ? q2 : T - 1;
      glds16(qkv + (rowbase + q) * 2304 + h * 64 + sc * 8, dst);
      glds16(qkv + (rowbase + q2) * 2304 + h * 64 + sc * 8, dst + 8192);
    } else {
      int key = (k - 2) * 64 + srow;
      key = key < T ? key : T - 1;   // clamped rows are masked (K) or multiplied by P = 0 (V)
      const h16* src = qkv + (rowbase + key) * 2304 + h * 64;
      glds16(src + 768 + k_chunk * 8, dst);
      glds16(src + 1536 + v_chunk * 8, dst + 8192);
    }
  };

  const int r = lane & 31, hh = lane >> 5;
  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  u16x8 qf[4];
  const int k_row_off = r * 128;
  const int k_sw = (r >> 1) & 7;
  const int vq = (lane & 15) >> 2;
  const int v_row_off = (4 * hh + vq) * 128;
  const int v_col = (32 * ((lane >> 4) & 1) + 8 * (lane & 3)) ^ (((vq >> 1) & 1) << 6);

  // stream position of the next piece to stage: item pi (of this block's), piece pk
  int pi = 0, pk = 0;
  auto stage_next = [&](int p) {
    stage(p & (kAttnStages - 1), (int)blockIdx.x + pi * G, pk);
    if (++pk == npi) { pk = 0; ++pi; }
  };
#pragma unroll
  for (int t = 0; t < kAttnStages - 1; ++t)
    if (t < P) stage_next(t);
  int p = 0;
  // piece p must have landed: 2 loads per piece and thread, in order, so the (at most two) younger pieces and `extra`
  // younger stores may stay outstanding; then the block barrier makes it visible and frees piece p - 1's slot, refilled next
  auto advance = [&](int extra) -> const char* {
    const int ahead = P - 1 - p;
    const int n = 2 * (ahead < 2 ? ahead : 2) + extra;
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p + kAttnStages - 1 < P) stage_next(p + kAttnStages - 1);
    const char* slot = smem + (p & (kAttnStages - 1)) * kAttnStage;
    ++p;
    return slot;
  };
  int stores = 0;   // this wave's epilogue stores of the item before: younger than the next three pieces' loads
  for (int j = 0; j < n_my; ++j) {
    const int id = (int)blockIdx.x + j * G;
    const int qb = id % nqb, bh = id / nqb;
    const int b = bh / 12, h = bh - b * 12;
    const int q0 = (qb * kAttnWaves + wave) * 32;
    const bool active = q0 < T;   // wave-uniform
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const char* slot = advance(stores);
      if (active && (wave >> 2) == k) {   // this wave's 32 Q rows are rows (wave & 3) * 32 .. of this slot
        const char* qs = slot + ((wave & 3) * 32 + r) * 128 + hh * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const u16x8*)(qs + s * 32);
      }
    }
    f32x16 o0 = kZero16, o1 = kZero16;
    float m = -1e30f, l = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
    const char* slot = advance(kt == 0 ? stores : 0);
    if (!active) continue;
    const char* Ks = slot;
    const char* Vs = Ks + 8192;

    f32x16 s0, s1;
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
    }
    stores = 0;
    if (active) {
      // ---- the item's context rows: two passes of 16 query rows through this wave's 2 KiB park, whole 128-B rows out
      float lt = l + __shfl_xor(l, 32);
      const float inv = 1.f / lt;
      char* park = smem + kAttnStages * kAttnStage + wave * 2048;
      const long long rowbase = (long long)b * T;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if ((r >> 4) == pass) {
          const int rr = r & 15, sw = rr & 7;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint2 a = make_uint2(pack2<F16>(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack2<F16>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
            const uint2 c = make_uint2(pack2<F16>(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack2<F16>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
            *(uint2*)(park + rr * 128 + ((g ^ sw) << 4) + 8 * hh) = a;
            *(uint2*)(park + rr * 128 + (((4 + g) ^ sw) << 4) + 8 * hh) = c;
          }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int c8 = lane & 7;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3);
          const uint4 v = *(const uint4*)(park + row * 128 + ((c8 ^ (row & 7)) << 4));
          const int q = q0 + pass * 16 + row;
          if (q < T) *(uint4*)(ctx + (rowbase + q) * 768 + h * 64 + c8 * 8) = v;
        }
        __builtin_amdgcn_wave_barrier();   // the reads of this pass are issued before the next pass overwrites the park
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      stores = 4;
    }
  }
}

}  // namespace vtc
