// Bandwidth-bound kernels of the BabyHuBERT encoder: conv layer 0 + GroupNorm + GELU, LayerNorm
// (+ fused per-frame multi-label head), positional-conv input packing.
// Arithmetic follows HF modeling_hubert.py:154-175 (conv0/GroupNorm/GELU), :216-231 (projection LN),
// :439-441 (encoder LN), :387-397 (post-LN layers); the reference names this stack at
// src/analysis/chunk_artifact.py:40-48.
#pragma once
#include "common.hip.h"

namespace vtc {

struct WindowTable {          // device arrays, one entry per window of the batch
  const long long* start;     // first sample of the window in the waveform arena
  const int* valid;           // real samples available (the rest of the window is zero padding)
  const long long* dst_row;   // first output row (global frame index) of the window
  const int* keep;            // frames kept from the window (<= T)
};

// ---------------------------------------------------------------------------------------------
// conv0 statistics.  conv0 is linear without bias, y[c,t] = sum_j w[c,j] x[5t+j], so the GroupNorm
// statistics of every channel follow from 10 first moments S_j = sum_t x[5t+j] and 55 second moments
// R[j,j'] = sum_t x[5t+j] x[5t+j'] of the window -- 65 numbers instead of 512 x L0 conv outputs.
// Accumulated in fp64, per-block partials summed in a fixed order (bitwise reproducible).
// ---------------------------------------------------------------------------------------------
constexpr int kStatN = 65;
constexpr int kStatSlots = 66;    // 65 moments + max |x| of the chunk (scales the fp16 split of conv0, below)
constexpr int kStatChunk = 4096;  // conv0 frames per block (16 per thread: the 65 fp64 wave reductions at the end cost as much as 8 frames)

__global__ __launch_bounds__(256) void conv0_moments_kernel(const float* __restrict__ wav, WindowTable wt, int L0,
                                                            double* __restrict__ partial /*[B][nchunk][66]*/) {
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const long long s0 = wt.start[b];
  const int valid = wt.valid[b];
  double acc[kStatN];
#pragma unroll
  for (int i = 0; i < kStatN; ++i) acc[i] = 0.0;
  float amax = 0.f;
  const int t_end = min(L0, (chunk + 1) * kStatChunk);
  for (int t = chunk * kStatChunk + threadIdx.x; t < t_end; t += 256) {
    double x[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int i = 5 * t + j;
      const float xf = i < valid ? wav[s0 + i] : 0.f;
      amax = fmaxf(amax, fabsf(xf));
      x[j] = (double)xf;
    }
    int k = 10;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      acc[j] += x[j];
#pragma unroll
      for (int j2 = j; j2 < 10; ++j2) { acc[k] = fma(x[j], x[j2], acc[k]); ++k; }
    }
  }
  __shared__ double red[4][kStatSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kStatN; ++i) {
    const double v = wave_sum_d(acc[i]);
    if (lane == 0) red[wave][i] = v;
  }
  amax = wave_max(amax);
  if (lane == 0) red[wave][kStatN] = (double)amax;
  __syncthreads();
  double* dst = partial + ((long long)b * nchunk + chunk) * kStatSlots;
  if (threadIdx.x < kStatN) {
    const int i = threadIdx.x;
    dst[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  } else if (threadIdx.x == kStatN) {
    dst[kStatN] = fmax(fmax(red[0][kStatN], red[1][kStatN]), fmax(red[2][kStatN], red[3][kStatN]));
  }
}

// One block per window, one thread per channel: fold the moments with the channel's 10 taps.
//
// For the MFMA form of conv0 (conv0_mfma_kernel) the same thread also writes the channel's row of the
// A operand: with x' = x * 2^e (e such that max|x'| is in [8, 16)) and w' = w * scale * 2^-e,
//   y*scale + shift = sum_j (xh_j + xl_j)(wh_j + wl_j) + shift  ~  [xh | xl | xh | 1 1] . [wh | wh | wl | sh_h sh_l]
// where h / l are the fp16 head and tail of an fp32 number (the dropped xl*wl terms are 2^-22 relative).
// That is ONE 32-deep fp16 MFMA per 16 x 16 outputs with fp32-level accuracy.  A window whose operand
// would leave the fp16 range (|w'| or |shift| > 3e4: pathological weights) is flagged and takes the VALU path.
__global__ __launch_bounds__(512) void conv0_gn_finalize_kernel(const double* __restrict__ partial, int nchunk, int L0,
                                                                const float* __restrict__ w0 /*[512][10]*/,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float* __restrict__ scale /*[B][512]*/,
                                                                float* __restrict__ shift,
                                                                _Float16* __restrict__ aop /*[B][512][32]*/,
                                                                float* __restrict__ xscale /*[B]*/,
                                                                int* __restrict__ valu_flag /*[B]*/) {
  __shared__ double mom[kStatSlots];
  const int b = blockIdx.x;
  if (threadIdx.x < kStatN) {
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += partial[((long long)b * nchunk + c) * kStatSlots + threadIdx.x];
    mom[threadIdx.x] = s;
  } else if (threadIdx.x == kStatN) {
    double mx = 0.0;
    for (int c = 0; c < nchunk; ++c) mx = fmax(mx, partial[((long long)b * nchunk + c) * kStatSlots + kStatN]);
    mom[kStatN] = mx;
  }
  __syncthreads();
  const int c = threadIdx.x;
  double w[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) w[j] = (double)w0[c * 10 + j];
  double s1 = 0.0, s2 = 0.0;
  int k = 10;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    s1 += w[j] * mom[j];
#pragma unroll
    for (int j2 = j; j2 < 10; ++j2) {
      const double r = w[j] * w[j2] * mom[k++];
      s2 += (j2 == j) ? r : 2.0 * r;
    }
  }
  const double mean = s1 / L0;
  double var = s2 / L0 - mean * mean;   // biased variance, as GroupNorm
  var = var > 0.0 ? var : 0.0;
  const double rstd = 1.0 / sqrt(var + 1e-5);
  const double sc = (double)gamma[c] * rstd;
  const float scf = (float)sc;
  const float shf = (float)((double)beta[c] - mean * sc);
  scale[b * 512 + c] = scf;
  shift[b * 512 + c] = shf;

  // fp16-split operand row.  2^e: max|x| * 2^e in [8, 16); silence keeps e = 0.
  const float amax = (float)mom[kStatN];
  const float p2 = amax > 0.f ? exp2f(3.f - floorf(log2f(amax))) : 1.f;
  const float inv = 1.f / p2;   // exact, a power of two
  bool bad = !(fabsf(shf) < 3.0e4f) || !(p2 < 1.0e30f) || !(inv < 1.0e30f);
  _Float16 row[32];
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const float wp = (float)w[j] * scf * inv;   // the product the VALU path forms as fma(y, scale, shift), here pre-scaled
    bad = bad || !(fabsf(wp) < 3.0e4f);
    const _Float16 h = (_Float16)wp;
    const _Float16 l = (_Float16)(wp - (float)h);
    row[j] = h; row[10 + j] = h; row[20 + j] = l;
  }
  const _Float16 sh_h = (_Float16)shf;
  row[30] = sh_h;
  row[31] = (_Float16)(shf - (float)sh_h);
  _Float16* dst = aop + ((long long)b * 512 + c) * 32;
#pragma unroll
  for (int i = 0; i < 32; ++i) dst[i] = row[i];
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (threadIdx.x == 0) { xscale[b] = p2; valu_flag[b] = any_bad; }
}

// conv0 + GroupNorm affine + GELU -> channel-last (B, L0, 512) 16-bit, VALU form.  Waveform tile staged in
// LDS, each lane owns 8 adjacent channels (taps in registers) so a wave stores one 1 KiB frame row.
// Used for bf16 operands and for windows flagged by the finalize kernel.
constexpr int kC0Frames = 64;  // conv0 frames per VALU tile

template <bool F16>
__device__ __forceinline__ void conv0_valu_tile(float* xs /*[5*64+8]*/, const float* __restrict__ wav, long long s0, int valid,
                                                int b, int t0, int L0, const float* __restrict__ w0,
                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                h16* __restrict__ out) {
  __syncthreads();
  for (int i = threadIdx.x; i < 5 * kC0Frames + 5; i += 256) {
    const int g = 5 * t0 + i;
    xs[i] = g < valid ? wav[s0 + g] : 0.f;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * 8;
  float w[8][10], sc[8], sh[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
#pragma unroll
    for (int j = 0; j < 10; ++j) w[c][j] = w0[(c0 + c) * 10 + j];
    sc[c] = scale[b * 512 + c0 + c];
    sh[c] = shift[b * 512 + c0 + c];
  }
  __syncthreads();
  for (int f = wave; f < kC0Frames; f += 4) {
    const int t = t0 + f;
    if (t >= L0) break;
    float x[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) x[j] = xs[5 * f + j];
    unsigned pk[4];
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
      float y0 = 0.f, y1 = 0.f;
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        y0 = fmaf(w[c][j], x[j], y0);
        y1 = fmaf(w[c + 1][j], x[j], y1);
      }
      y0 = fmaf(y0, sc[c], sh[c]);
      y1 = fmaf(y1, sc[c + 1], sh[c + 1]);
      gelu2(y0, y1);
      pk[c >> 1] = pack2<F16>(y0, y1);
    }
    *(uint4*)(out + ((long long)b * L0 + t) * 512 + c0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void conv0_apply_kernel(const float* __restrict__ wav, WindowTable wt, int L0,
                                                          const float* __restrict__ w0, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, h16* __restrict__ out) {
  __shared__ float xs[5 * kC0Frames + 8];
  const int b = blockIdx.y;
  conv0_valu_tile<F16>(xs, wav, wt.start[b], wt.valid[b], b, blockIdx.x * kC0Frames, L0, w0, scale, shift, out);
}

// MFMA form (fp16 operands only; see conv0_gn_finalize_kernel for the operand split).  A block = 256 frames x
// 512 channels of one window: the frames' B operand [xh | xl | xh | 1 1] (64 B per frame) is built once in LDS;
// wave w owns channels [128 w, +128): 8 A fragments in registers, channels assigned to MFMA tiles so that a
// lane ends up with 16 consecutive channels of one frame per 64-channel group (two 16-B stores).
// Per output the VALU work left is the GELU and the fp16 pack.
constexpr int kC0mFrames = 256;

// VALU form for the windows the finalize kernel flagged (none with sane weights): 8 blocks per window walk its tiles.
__global__ __launch_bounds__(256) void conv0_valu_fallback_kernel(const float* __restrict__ wav, WindowTable wt, int L0,
                                                                  const int* __restrict__ valu_flag, const float* __restrict__ w0,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  h16* __restrict__ out) {
  __shared__ float xs[5 * kC0Frames + 8];
  const int b = blockIdx.y;
  if (!valu_flag[b]) return;
  for (int t0 = blockIdx.x * kC0Frames; t0 < L0; t0 += gridDim.x * kC0Frames)
    conv0_valu_tile<true>(xs, wav, wt.start[b], wt.valid[b], b, t0, L0, w0, scale, shift, out);
}

__global__ __launch_bounds__(256) void conv0_mfma_kernel(const float* __restrict__ wav, WindowTable wt, int L0,
                                                         const _Float16* __restrict__ aop, const float* __restrict__ xscale,
                                                         const int* __restrict__ valu_flag, h16* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) _Float16 xp[kC0mFrames * 32];   // 16 KiB
  __shared__ float xs[5 * kC0mFrames + 8];
  __shared__ __attribute__((aligned(16))) char outs[4 * 4096];             // per wave: 16 frames x 128 channels, 16 bit
  const int b = blockIdx.y;
  if (valu_flag[b]) return;   // block-uniform: operand outside the fp16 range, conv0_valu_fallback_kernel does this window
  const int t0 = blockIdx.x * kC0mFrames;
  const long long s0 = wt.start[b];
  const int valid = wt.valid[b];
  const float p2 = xscale[b];
#pragma unroll
  for (int i = 0; i < 6; ++i) {   // 5 * 256 + 5 samples, independent loads
    const int o = threadIdx.x + i * 256;
    const int g = 5 * t0 + o;
    if (o < 5 * kC0mFrames + 5) xs[o] = (g < valid ? wav[s0 + g] : 0.f) * p2;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < kC0mFrames * 10; idx += 256) {
    const int f = idx / 10, j = idx - f * 10;
    const float x = xs[5 * f + j];
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    _Float16* row = xp + f * 32;
    row[j] = h; row[10 + j] = l; row[20 + j] = h;
    if (j < 2) row[30 + j] = (_Float16)1.0f;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // A fragments: MFMA tile j of 64-channel group q holds channel 32*(j>>1) + 8*(i>>2) + 4*(j&1) + (i&3) in operand row i, so
  // that a lane's outputs are channels 8*fq .. +8 (tiles 0, 1) and 32 + 8*fq .. +8 (tiles 2, 3): each 16-B store instruction
  // then writes 64 contiguous bytes per frame and the pair one whole 128-B line (16-B pieces with gaps between them are
  // partial-line writes, which a CU retires at ~10 B/clk: tools/store_bench)
  u16x8 af[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = wave * 128 + q * 64 + 32 * (j >> 1) + 8 * (fr >> 2) + 4 * (j & 1) + (fr & 3);
      af[q][j] = *(const u16x8*)(aop + ((long long)b * 512 + c) * 32 + fq * 8);
    }
  __syncthreads();
  // Outputs leave through a wave-private LDS tile (16 frames x 256 B, 16-B chunks XOR-swizzled by the frame) and are stored
  // as whole rows: each store instruction writes 4 frames x 256 B = 8 whole lines.
  char* const ow = outs + wave * 4096;
#pragma unroll 2
  for (int ft = 0; ft < kC0mFrames / 16; ++ft) {
    const u16x8 bf = *(const u16x8*)(xp + (ft * 16 + fr) * 32 + fq * 8);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      unsigned pk[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma16<true>(af[q][j], bf, acc);   // rows: channels 32*(j>>1) + 8*fq + 4*(j&1) + r of the group, cols: frames
        float g0 = acc[0], g1 = acc[1], g2 = acc[2], g3 = acc[3];
        gelu2(g0, g1);
        gelu2(g2, g3);
        pk[2 * j] = pack2<true>(g0, g1);
        pk[2 * j + 1] = pack2<true>(g2, g3);
      }
      // chunk (of 16 in the frame's 256 B) = 8 q + fq and 8 q + 4 + fq
      *(uint4*)(ow + fr * 256 + (((8 * q + fq) ^ fr) << 4)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      *(uint4*)(ow + fr * 256 + (((8 * q + 4 + fq) ^ fr) << 4)) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = k * 4 + (lane >> 4), c = lane & 15;
      const uint4 v = *(const uint4*)(ow + f * 256 + ((c ^ f) << 4));
      const int t = t0 + ft * 16 + f;
      if (t < L0) *(uint4*)(out + ((long long)b * L0 + t) * 512 + wave * 128 + c * 8) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over rows; one wave per row, two-pass (mean, then centred variance) in registers.
//   D = 512: 16-bit input (conv6 output), 16-bit output.
//   D = 768: fp32 input, 16-bit output (the residual stream is kept 16-bit; fp32 copy only if out32 != null); HEAD fuses the per-frame multi-label head
//            (n_labels linear units on the normalised row, fp32) and writes stitched logits.
// ---------------------------------------------------------------------------------------------
struct HeadParams {
  const float* w;   // [n_labels][768]
  const float* b;   // [n_labels]
  float* logits;    // [total_rows][n_labels]
  int n_labels;
  int T;            // frames per window
  WindowTable wt;
};

template <bool F16>
__global__ __launch_bounds__(256) void layernorm512_kernel(const h16* __restrict__ in, const float* __restrict__ g,
                                                           const float* __restrict__ bt, h16* __restrict__ out, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const u16x8 raw = *(const u16x8*)(in + (long long)row * 512 + lane * 8);
  float v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = from_h16<F16>(raw[i]); s += v[i]; }
  const float mean = wave_sum(s) * (1.f / 512.f);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] -= mean; q = fmaf(v[i], v[i], q); }
  const float rstd = rsqrtf(wave_sum(q) * (1.f / 512.f) + 1e-5f);
  const float4 g0 = *(const float4*)(g + lane * 8), g1 = *(const float4*)(g + lane * 8 + 4);
  const float4 b0 = *(const float4*)(bt + lane * 8), b1 = *(const float4*)(bt + lane * 8 + 4);
  const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  unsigned pk[4];
#pragma unroll
  for (int i = 0; i < 8; i += 2)
    pk[i >> 1] = pack2<F16>(fmaf(v[i] * rstd, gg[i], bb[i]), fmaf(v[i + 1] * rstd, gg[i + 1], bb[i + 1]));
  *(uint4*)(out + (long long)row * 512 + lane * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
}

// IN16: the input rows are 16-bit (the GEMM epilogue that produced them rounded once more); else fp32.
// A wave handles kLnRows rows at once: all their loads are issued before the first reduction, which doubles the bytes
// in flight per wave (the kernel is a pure stream over rows that the producing GEMM left in L2 / Infinity Cache).
constexpr int kLnRows = 2;

// IN: 0 = fp32 rows, 1 = 16-bit rows.
template <bool F16, bool HEAD, int IN = 0>
__global__ __launch_bounds__(256) void layernorm768_kernel(const void* __restrict__ in,
                                                           const float* __restrict__ g, const float* __restrict__ bt,
                                                           float* __restrict__ out32, h16* __restrict__ out16, int rows,
                                                           HeadParams hp) {
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kLnRows;
  if (row0 >= rows) return;
  const int lane = threadIdx.x & 63;
  float v[kLnRows][12];
  float s[kLnRows];
#pragma unroll
  for (int r = 0; r < kLnRows; ++r) {
    const int row = min(row0 + r, rows - 1);   // a clamped duplicate row is computed and not stored
    s[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float4 x;
      if constexpr (IN == 1) {
        const u16x4 h = *(const u16x4*)((const h16*)in + (long long)row * 768 + i * 256 + lane * 4);
        x = make_float4(from_h16<F16>(h[0]), from_h16<F16>(h[1]), from_h16<F16>(h[2]), from_h16<F16>(h[3]));
      } else {
        x = *(const float4*)((const float*)in + (long long)row * 768 + i * 256 + lane * 4);
      }
      v[r][4 * i] = x.x; v[r][4 * i + 1] = x.y; v[r][4 * i + 2] = x.z; v[r][4 * i + 3] = x.w;
      s[r] += (x.x + x.y) + (x.z + x.w);
    }
  }
  float4 gg[3], bb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { gg[i] = *(const float4*)(g + i * 256 + lane * 4); bb[i] = *(const float4*)(bt + i * 256 + lane * 4); }
#pragma unroll
  for (int r = 0; r < kLnRows; ++r) {
    const int row = row0 + r;
    const float mean = wave_sum(s[r]) * (1.f / 768.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { v[r][i] -= mean; q = fmaf(v[r][i], v[r][i], q); }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / 768.f) + 1e-5f);
    if (row >= rows) break;   // wave-uniform
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = i * 256 + lane * 4;
      v[r][4 * i] = fmaf(v[r][4 * i] * rstd, gg[i].x, bb[i].x);
      v[r][4 * i + 1] = fmaf(v[r][4 * i + 1] * rstd, gg[i].y, bb[i].y);
      v[r][4 * i + 2] = fmaf(v[r][4 * i + 2] * rstd, gg[i].z, bb[i].z);
      v[r][4 * i + 3] = fmaf(v[r][4 * i + 3] * rstd, gg[i].w, bb[i].w);
      if (out32) *(float4*)(out32 + (long long)row * 768 + c) = make_float4(v[r][4 * i], v[r][4 * i + 1], v[r][4 * i + 2], v[r][4 * i + 3]);
      if (out16)
        *(uint2*)(out16 + (long long)row * 768 + c) =
            make_uint2(pack2<F16>(v[r][4 * i], v[r][4 * i + 1]), pack2<F16>(v[r][4 * i + 2], v[r][4 * i + 3]));
    }
    if constexpr (HEAD) {
      const int b = row / hp.T, t = row - b * hp.T;
      const bool keep = t < hp.wt.keep[b];
      const long long dst = hp.wt.dst_row[b] + t;
      for (int l = 0; l < hp.n_labels; ++l) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float4 ww = *(const float4*)(hp.w + l * 768 + i * 256 + lane * 4);
          d = fmaf(v[r][4 * i], ww.x, d); d = fmaf(v[r][4 * i + 1], ww.y, d);
          d = fmaf(v[r][4 * i + 2], ww.z, d); d = fmaf(v[r][4 * i + 3], ww.w, d);
        }
        d = wave_sum(d);
        if (lane == 0 && keep) hp.logits[dst * hp.n_labels + l] = d + hp.b[l];
      }
    }
  }
}

// Entry of the LayerNorm-free transformer layers (gemm256.hip.h): fp32 pre-LN rows (pos-conv output + residual) -> 16-bit
// rows and the row statistics (mean, rstd) of the encoder's input LayerNorm; one wave per row, two-pass.
template <bool F16>
__global__ __launch_bounds__(256) void ln_round_stats_kernel(const float* __restrict__ in, h16* __restrict__ out_h,
                                                             float* __restrict__ stats, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float v[12];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 x = *(const float4*)(in + (long long)row * 768 + i * 256 + lane * 4);
    v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
    s += (x.x + x.y) + (x.z + x.w);
  }
  const float mean = wave_sum(s) * (1.f / 768.f);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
  const float rstd = rsqrtf(wave_sum(q) * (1.f / 768.f) + 1e-5f);
  if (lane == 0) *(float2*)(stats + (long long)row * 2) = make_float2(mean, rstd);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const long long at = (long long)row * 768 + i * 256 + lane * 4;
    *(uint2*)(out_h + at) = make_uint2(pack2<F16>(v[4 * i], v[4 * i + 1]), pack2<F16>(v[4 * i + 2], v[4 * i + 3]));
  }
}

// Row statistics from the 16 partials the EPI_LNRES epilogue left (each the mean and centred sum of squares of 48 columns):
// equal group sizes, so mean = average of the group means and M2 = sum of the group M2 + 48 * sum (group mean - mean)^2
// (Chan et al.), always in the same order: bit-reproducible.  One thread per row.
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int rows,
                                                                int part_rows) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  float mp[16], qp[16];   // part is [16 groups][part_rows][2]
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float2 t = *(const float2*)(part + ((long long)i * part_rows + row) * 2);
    mp[i] = t.x; qp[i] = t.y;
  }
  float sm = 0.f, sq = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { sm += mp[i]; sq += qp[i]; }
  const float mean = sm * (1.f / 16.f);
  float dev = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { const float d = mp[i] - mean; dev = fmaf(d, d, dev); }
  const float var = fmaf(48.f, dev, sq) * (1.f / 768.f);
  *(float2*)(stats + (long long)row * 2) = make_float2(mean, rsqrtf(var + 1e-5f));
}

// Positional-conv input: fp32 (B*T, 768) -> zero-padded group-major 16-bit (B, 16, T+128, 48) so the
// k=128 window of output frame t is the contiguous slice rows [t, t+128) (pad 64 front, modeling_hubert.py:48-54).
template <bool F16>
__global__ __launch_bounds__(192) void posconv_pack_kernel(const float* __restrict__ h0, h16* __restrict__ out, int T) {
  // one thread per 4 channels (48 % 4 == 0: they share a group): a 16-B load and an 8-B store; 4 padded positions per block
  const int b = blockIdx.y;
  const int TP = T + 128;
  const int c = threadIdx.x * 4, g = c / 48, ci = c - g * 48;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = blockIdx.x * 4 + k;
    if (p >= TP) break;
    const int t = p - 64;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0 && t < T) v = *(const float4*)(h0 + ((long long)b * T + t) * 768 + c);
    *(uint2*)(out + (((long long)b * 16 + g) * TP + p) * 48 + ci) = make_uint2(pack2<F16>(v.x, v.y), pack2<F16>(v.z, v.w));
  }
}

}  // namespace vtc
