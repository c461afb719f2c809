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
  return (unsigned)to_h16<F16>(a) | ((unsigned)to_h16<F16>(b) << 16);
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

// Exact-form GELU 0.5*x*(1+erf(x/sqrt2)) via erfc(|z|) = poly(t)*exp(-z^2), t = 1/(1+p|z|)
// (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7); the erfc form keeps the negative tail relative-accurate.
__device__ __forceinline__ float gelu_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));  // v_rcp_f32 (1 ulp); a true divide costs 11 VALU
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = p * t * __expf(-z * z);      // erfc(z), z >= 0
  const float hx = 0.5f * x;
  return x > 0.f ? fmaf(-hx, e, x) : hx * e;
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
