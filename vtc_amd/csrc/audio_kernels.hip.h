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

template <bool I16>
__global__ __launch_bounds__(256) void prepare_audio_kernel(const void* __restrict__ pcm, long long n_in, int ch,
                                                            ResampleTable rt, float* __restrict__ out, long long n_out) {
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= n_out) return;
  auto mono = [&](long long i) -> float {   // mean over channels, summed in channel order
    float s = 0.f;
    for (int c = 0; c < ch; ++c) {
      if constexpr (I16) s += (float)((const short*)pcm)[i * ch + c] * (1.0f / 32768.0f);
      else               s += ((const float*)pcm)[i * ch + c];
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
