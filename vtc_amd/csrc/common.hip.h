// Shared device helpers for the VTC hot path (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vtc {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8  = __attribute__((ext_vector_type(8))) _Float16;
using f32x4  = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u16x4  = __attribute__((ext_vector_type(4))) unsigned short;
using u16x8  = __attribute__((ext_vector_type(8))) unsigned short;
using s16x4  = __attribute__((ext_vector_type(4))) short;

typedef unsigned short h16;  // raw 16-bit storage; meaning (bf16 / f16) chosen by the F16 template flag

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int kRfStep = 320;   // product of conv strides (reference: src/analysis/chunk_artifact.py:51)
constexpr int kRfSize = 400;   // receptive field in samples (chunk_artifact.py:11-13)

template <bool F16> __device__ __forceinline__ h16 to_h16(float x) {
  if constexpr (F16) { _Float16 h = (_Float16)x; return __builtin_bit_cast(h16, h); }
  else               { __bf16 h = (__bf16)x;     return __builtin_bit_cast(h16, h); }
}
template <bool F16> __device__ __forceinline__ float from_h16(h16 v) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
  else               return __uint_as_float(((unsigned)v) << 16);
}
template <bool F16> __device__ __forceinline__ unsigned pack2(float a, float b) {
  // vector conversions so that one v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 (round to nearest even) makes the pair
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  if constexpr (F16) {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  } else {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  }
}

template <bool F16> __device__ __forceinline__ f32x4 mfma16(u16x8 a, u16x8 b, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ f32x16 mfma32(u16x8 a, u16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// GELU x * Phi(x) (HF "gelu" = 0.5*x*(1+erf(x/sqrt2))) as x * sigmoid(x * P(x^2)): logit(Phi(x)) is odd and smooth,
// and a degree-4 P (minimax fit of the GELU error on |x| <= 5.5, tools/fit_gelu.py) gives |err| <= 3.5e-6 for every
// float (Phi -> 0 / 1 correctly beyond the fit range; no NaN/inf for finite x).  That is 1/70 of one fp16 ulp at
// 1.0, the precision every GELU output is stored in.  8 VALU + v_exp_f32 + v_rcp_f32 per element, about half the
// cost of the erfc form it replaces; the coefficients carry the -log2(e) of the exp2.
__device__ __forceinline__ float gelu_f(float x) {
  const float s = x * x;
  float p = fmaf(-3.2290622584696393e-06f, s, 8.823935058899224e-05f);
  p = fmaf(p, s, 0.0003602687211241573f);
  p = fmaf(p, s, -0.10522668063640594f);
  p = fmaf(p, s, -2.3020453453063965f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
}

// Two elements at once with packed fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: the same IEEE operations
// in the same order, so the results are bit-identical to gelu_f).  tools/gelu_bench: 15.8 ns of SIMD time per 64 elements
// against 21.3; what is left is mostly the two quarter-rate transcendentals (8.9 ns).
__device__ __forceinline__ void gelu2(float& a, float& b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 x = {a, b};
  const f32x2 s = x * x;
  f32x2 p = __builtin_elementwise_fma(f32x2{-3.2290622584696393e-06f, -3.2290622584696393e-06f}, s,
                                      f32x2{8.823935058899224e-05f, 8.823935058899224e-05f});
  p = __builtin_elementwise_fma(p, s, f32x2{0.0003602687211241573f, 0.0003602687211241573f});
  p = __builtin_elementwise_fma(p, s, f32x2{-0.10522668063640594f, -0.10522668063640594f});
  p = __builtin_elementwise_fma(p, s, f32x2{-2.3020453453063965f, -2.3020453453063965f});
  const f32x2 z = x * p;
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + f32x2{1.f, 1.f};
  const f32x2 r = x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  a = r[0]; b = r[1];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Bijective XCD-aware remap of a linear workgroup id (cdna_hip_programming.md section 5): blocks that
// share an XCD (same id % 8) get a contiguous range of tiles, so neighbouring tiles share one L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned orig, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

}  // namespace vtc
