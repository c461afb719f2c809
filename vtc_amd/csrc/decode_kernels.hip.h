// Threshold -> run-length decoding of stitched logits, on the GPU.
// Replaces segma.inference.apply_thresholds + create_intervals (call sites
// /root/reference/src/pipeline/vtc.py:84-91) and the sigmoid summary (vtc.py:266-273).
// A frame of label j is active iff lo[j] <= logit <= hi[j] (bounds already mapped to logit space by
// the host, see vtc_amd/segma_compat/inference.py); a run of active frames i..j becomes
// (i*320, j*320+400) samples (src/analysis/chunk_artifact.py:11-13,75-83) on the host side.
// Three passes, deterministic: count run starts/ends per 256-frame block, exclusive scan, emit.
#pragma once
#include "common.hip.h"

namespace vtc {

constexpr int kMaxLabels = 8;

struct DecodeParams {
  const float* logits;  // [n][nl]
  long long n;
  int nl;
  int nblk;
  float lo[kMaxLabels];
  float hi[kMaxLabels];
  const unsigned char* file_start;   // optional [n + 1]: 1 where a new recording begins (runs never cross it); null = one recording
};

__device__ __forceinline__ bool frame_active(const DecodeParams& p, long long f, int l) {
  if (f < 0 || f >= p.n) return false;
  const float x = p.logits[f * p.nl + l];
  return x >= p.lo[l] && x <= p.hi[l];
}
// neighbours seen from frame f: the previous / next frame counts only if it belongs to the same recording
__device__ __forceinline__ bool prev_active(const DecodeParams& p, long long f, int l) {
  if (p.file_start && f >= 0 && f < p.n && p.file_start[f]) return false;
  return frame_active(p, f - 1, l);
}
__device__ __forceinline__ bool next_active(const DecodeParams& p, long long f, int l) {
  if (p.file_start && f + 1 < p.n && p.file_start[f + 1]) return false;
  return frame_active(p, f + 1, l);
}

// Maximum that remembers a non-finite input: the sigmoid summary's max logit is NaN iff some logit of the rows it covers
// is NaN or +-inf (an activation that left the 16-bit range upstream ends as NaN logits; a NaN compares false against every
// bound, so without this mark such a recording would decode as "no speech" and look valid).
__device__ __forceinline__ float max_keep_nan(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }
__device__ __forceinline__ float finite_or_nan(float x) { return fabsf(x) <= 3.4028234e38f ? x : __builtin_nanf(""); }
__device__ __forceinline__ float wave_max_keep_nan(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max_keep_nan(v, __shfl_xor(v, o));
  return v;
}

// counts: [2][nl][nblk] (0 = starts, 1 = ends); sig_sum[nblk] (fp64), max_logit[nblk]
__global__ __launch_bounds__(256) void decode_count_kernel(DecodeParams p, int* __restrict__ counts,
                                                           double* __restrict__ sig_sum, float* __restrict__ max_logit) {
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  double ssum = 0.0;
  float mx = -INFINITY;
  for (int l = 0; l < p.nl; ++l) {
    const bool a = frame_active(p, f, l);
    const bool st = a && !prev_active(p, f, l);
    const bool en = a && !next_active(p, f, l);
    const int cs = __syncthreads_count(st);
    const int ce = __syncthreads_count(en);
    if (threadIdx.x == 0) {
      counts[(0 * p.nl + l) * p.nblk + blockIdx.x] = cs;
      counts[(1 * p.nl + l) * p.nblk + blockIdx.x] = ce;
    }
    if (f < p.n) {
      const float x = p.logits[f * p.nl + l];
      ssum += (double)(1.f / (1.f + expf(-x)));
      mx = max_keep_nan(mx, finite_or_nan(x));
    }
  }
  __shared__ double rs[4];
  __shared__ float rm[4];
  ssum = wave_sum_d(ssum);
  mx = wave_max_keep_nan(mx);
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = ssum; rm[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sig_sum[blockIdx.x] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    max_logit[blockIdx.x] = max_keep_nan(max_keep_nan(rm[0], rm[1]), max_keep_nan(rm[2], rm[3]));
  }
}

// One block per (kind,label) row of counts: in-place exclusive scan, total -> totals[row].
// Block 0 additionally folds the sigmoid partials (fixed order) into summary[0..1].
__global__ __launch_bounds__(1024) void decode_scan_kernel(int* __restrict__ counts, int nblk, int* __restrict__ totals,
                                                           const double* __restrict__ sig_sum,
                                                           const float* __restrict__ max_logit,
                                                           double* __restrict__ summary) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  int* row = counts + (long long)blockIdx.x * nblk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblk ? row[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int carry = carry_s;
    if (i < nblk) row[i] = carry + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    float mx = -INFINITY;
    for (int i = 0; i < nblk; ++i) { s += sig_sum[i]; mx = max_keep_nan(mx, max_logit[i]); }
    summary[0] = s;
    summary[1] = (double)mx;
  }
}

__device__ __forceinline__ int block_rank(bool pred, int* wcnt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(pred);
  const int r = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wcnt[wave] = __popcll(bal);
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += wcnt[w];
  __syncthreads();
  return off + r;
}

// out_start / out_end: [nl][cap] frame indices, k-th start pairs with k-th end of the same label.
__global__ __launch_bounds__(256) void decode_emit_kernel(DecodeParams p, const int* __restrict__ offsets, long long cap,
                                                          int* __restrict__ out_start, int* __restrict__ out_end) {
  __shared__ int wcnt[4];
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  for (int l = 0; l < p.nl; ++l) {
    const bool a = frame_active(p, f, l);
    const bool st = a && !prev_active(p, f, l);
    const bool en = a && !next_active(p, f, l);
    const int rs = block_rank(st, wcnt);
    const int re = block_rank(en, wcnt);
    if (st) {
      const long long k = offsets[(0 * p.nl + l) * p.nblk + blockIdx.x] + rs;
      if (k < cap) out_start[l * cap + k] = (int)f;
    }
    if (en) {
      const long long k = offsets[(1 * p.nl + l) * p.nblk + blockIdx.x] + re;
      if (k < cap) out_end[l * cap + k] = (int)f;
    }
  }
}

// Per-recording sigmoid summary for vtc_decode_files: one block per recording, fixed summation order (thread-strided
// partials, then wave and block reduction): out[2 f] = sum of sigmoid over rows x labels (fp64), out[2 f + 1] = max logit.
__global__ __launch_bounds__(256) void file_summary_kernel(const float* __restrict__ logits, int nl, const long long* __restrict__ rows,
                                                           double* __restrict__ out) {
  const int f = blockIdx.x;
  const long long e0 = rows[f] * nl, e1 = rows[f + 1] * nl;
  double ssum = 0.0;
  float mx = -INFINITY;
  for (long long i = e0 + threadIdx.x; i < e1; i += 256) {
    const float x = logits[i];
    ssum += (double)(1.f / (1.f + expf(-x)));
    mx = max_keep_nan(mx, finite_or_nan(x));
  }
  __shared__ double rs[4];
  __shared__ float rm[4];
  ssum = wave_sum_d(ssum);
  mx = wave_max_keep_nan(mx);
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = ssum; rm[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * f] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    out[2 * f + 1] = (double)max_keep_nan(max_keep_nan(rm[0], rm[1]), max_keep_nan(rm[2], rm[3]));
  }
}

// apply_thresholds alone (the drop-in surface hands the 0/1 mask to the caller): mask[f][l] = lo[l] <= logit <= hi[l].
__global__ __launch_bounds__(256) void threshold_mask_kernel(DecodeParams p, unsigned char* __restrict__ mask) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n * p.nl) return;
  const int l = (int)(i % p.nl);
  const float x = p.logits[i];
  mask[i] = (x >= p.lo[l] && x <= p.hi[l]) ? 1 : 0;
}

// Last pass of the no-allocation decode (vtc_decode_static): per-label run lists (frame indices) -> the caller's device
// arrays in vtc_decode's order (label-major, time-ordered) and units (samples).  result[0] = number of runs (all of them,
// also when only `cap` fit), or -1 if a label's starts and ends disagree; summary_out = {max logit, mean sigmoid}.
__global__ __launch_bounds__(256) void decode_pack_kernel(const int* __restrict__ totals, const int* __restrict__ run_start,
                                                          const int* __restrict__ run_end, long long run_cap, int nl,
                                                          long long cap, long long* __restrict__ out_start,
                                                          long long* __restrict__ out_end, int* __restrict__ out_label,
                                                          long long* __restrict__ result, const double* __restrict__ summary,
                                                          double n_values, double* __restrict__ summary_out) {
  long long base = 0, total = 0;
  bool ok = true;
  const int l = blockIdx.y;
  for (int j = 0; j < nl; ++j) {
    const int c = totals[j];
    ok = ok && c == totals[nl + j] && c >= 0 && c <= run_cap;
    if (j < l) base += c;
    total += c;
  }
  if (blockIdx.x == 0 && l == 0 && threadIdx.x == 0) {
    result[0] = ok ? total : -1;
    summary_out[0] = summary[1];
    summary_out[1] = summary[0] / n_values;
  }
  if (!ok) return;
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= totals[l]) return;
  const long long at = base + k;
  if (at >= cap) return;
  out_start[at] = (long long)run_start[l * run_cap + k] * kRfStep;
  out_end[at] = (long long)run_end[l * run_cap + k] * kRfStep + kRfSize;
  out_label[at] = l;
}

// First frame of every recording of a multi-recording row layout (rows[f] = first row of recording f).
__global__ void mark_file_starts_kernel(const long long* __restrict__ rows, int n_files, long long n, unsigned char* __restrict__ mask) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < n_files) {
    const long long r = rows[f];
    if (r < n) mask[r] = 1;
  }
}

}  // namespace vtc
