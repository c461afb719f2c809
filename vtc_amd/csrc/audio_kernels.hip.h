// Audio front-end on the GPU: interleaved PCM in HBM -> mono float32 at 16 kHz ("decode file
// -> mono 16 kHz float", and section 8f rank 1).  The reference leaves this to segma / torchaudio (requirements.txt:393-394);
// the rule is unpinned, so this follows torchaudio.functional.resample's published algorithm (sinc_interp_hann,
// lowpass_filter_width 6, rolloff 0.99): a polyphase windowed-sinc filter, kernels computed in fp64 and applied in fp32.
// HBM-bound streaming work: one thread per output sample, ~12 * max(1, sr_in/16000) taps each.
#pragma once
#include "common.hip.h"

namespace vtc {

struct ResampleTable {
  int orig, new_;        // sr_in / g, 16000 / g
  int width;             // left padding of the conv (samples at the input rate)
  int kc;                // stored taps per phase (compact: only the support of the window)
  const float* w;        // [new_][kc]
  const int* first;      // [new_] index of the first stored tap inside the full (2*width + orig)-tap kernel
};

// FMT: 0 = int16, 1 = float32, 2 = packed little-endian int24 (3 bytes per sample, as a 24-bit PCM WAV stores them; value / 2^23,
// which float32 holds exactly)
template <int FMT>
__global__ __launch_bounds__(256) void prepare_audio_kernel(const void* __restrict__ pcm, long long n_in, int ch,
                                                            ResampleTable rt, float* __restrict__ out, long long n_out) {
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= n_out) return;
  auto mono = [&](long long i) -> float {   // mean over channels, summed in channel order
    float s = 0.f;
    for (int c = 0; c < ch; ++c) {
      if constexpr (FMT == 0) s += (float)((const short*)pcm)[i * ch + c] * (1.0f / 32768.0f);
      else if constexpr (FMT == 1) s += ((const float*)pcm)[i * ch + c];
      else {
        const unsigned char* b = (const unsigned char*)pcm + (i * ch + c) * 3;
        const int v = (int)((unsigned)b[0] | ((unsigned)b[1] << 8) | ((unsigned)(signed char)b[2] << 16));
        s += (float)v * (1.0f / 8388608.0f);
      }
    }
    return ch == 1 ? s : s / (float)ch;
  };
  if (rt.orig == rt.new_) { out[o] = mono(o); return; }
  const long long q = o / rt.new_;
  const int p = (int)(o - q * rt.new_);
  const long long base = q * rt.orig - rt.width + rt.first[p];
  const float* w = rt.w + (long long)p * rt.kc;
  float acc = 0.f;
  for (int j = 0; j < rt.kc; ++j) {
    const long long i = base + j;
    if (i >= 0 && i < n_in) acc = fmaf(w[j], mono(i), acc);
  }
  out[o] = acc;
}

}  // namespace vtc
