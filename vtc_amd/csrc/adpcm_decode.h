// Host-side IMA / DVI ADPCM decoder for RIFF/WAVE format tag 0x0011 (no GPU work).  `.wav` is on the reference's manifest list
// (scripts/make_manifest.py:26-39) and the reference decodes whatever torchcodec/ffmpeg accept inside segma; 4-bit ADPCM is what
// voice recorders and telephony archives store in it.  The codec is the IMA recommendation's (1992): an 89-entry step table, an
// index adaptation table, the predictor clamped to 16 bits.  Block layout (Microsoft's "WAVE_FORMAT_DVI_ADPCM"): per channel a
// 4-byte header (int16 first sample, uint8 step index, one reserved byte), then 4-byte groups per channel in turn, each group
// eight 4-bit codes of that channel, LOW nibble first.  Pinned in tests/test_host.py against the standard library's audioop
// (same codec, high nibble first, no blocks).
#pragma once
#include <cstdint>

namespace adpcm {

static const int16_t kStep[89] = {
    7,     8,     9,     10,    11,    12,    13,    14,    16,    17,    19,    21,    23,    25,    28,    31,    34,    37,
    41,    45,    50,    55,    60,    66,    73,    80,    88,    97,    107,   118,   130,   143,   157,   173,   190,   209,
    230,   253,   279,   307,   337,   371,   408,   449,   494,   544,   598,   658,   724,   796,   876,   963,   1060,  1166,
    1282,  1411,  1552,  1707,  1878,  2066,  2272,  2499,  2749,  3024,  3327,  3660,  4026,  4428,  4871,  5358,  5894,  6484,
    7132,  7845,  8630,  9493,  10442, 11487, 12635, 13899, 15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};
static const int8_t kIndex[16] = {-1, -1, -1, -1, 2, 4, 6, 8, -1, -1, -1, -1, 2, 4, 6, 8};

static inline int16_t step_once(int code, int& pred, int& index) {
  const int step = kStep[index];
  int diff = step >> 3;
  if (code & 4) diff += step;
  if (code & 2) diff += step >> 1;
  if (code & 1) diff += step >> 2;
  pred += (code & 8) ? -diff : diff;
  pred = pred > 32767 ? 32767 : (pred < -32768 ? -32768 : pred);
  index += kIndex[code];
  index = index < 0 ? 0 : (index > 88 ? 88 : index);
  return (int16_t)pred;
}

// data: the 'data' chunk; returns inter-channel samples written (<= cap), or -1 with *why set.  A trailing partial block
// gives the samples it holds completely (whole groups of eight per channel).
static inline int64_t decode_ima(const uint8_t* data, int64_t n_bytes, int channels, int block_align, int samples_per_block,
                                 int16_t* out, int64_t cap, const char** why) {
  if (channels < 1 || channels > 8) { *why = "IMA ADPCM: 1..8 channels"; return -1; }
  if (block_align < 4 * channels || (block_align - 4 * channels) % (4 * channels)) {
    *why = "IMA ADPCM: block size is not a header plus whole 4-byte groups per channel";
    return -1;
  }
  const int full = 1 + (block_align - 4 * channels) / (4 * channels) * 8;   // samples per channel a whole block holds
  if (samples_per_block <= 0) samples_per_block = full;
  if (samples_per_block > full) { *why = "IMA ADPCM: samples per block exceed the block size"; return -1; }
  int64_t at = 0;   // inter-channel samples written
  for (int64_t off = 0; off + 4 * channels <= n_bytes && at < cap; off += block_align) {
    const uint8_t* b = data + off;
    const int64_t have = n_bytes - off < block_align ? n_bytes - off : block_align;
    int groups = (int)((have - 4 * channels) / (4 * channels));
    int n = 1 + groups * 8;
    if (n > samples_per_block) n = samples_per_block;
    if (n > cap - at) n = (int)(cap - at);
    int pred[8], index[8];
    for (int c = 0; c < channels; ++c) {
      pred[c] = (int16_t)(b[4 * c] | (b[4 * c + 1] << 8));
      index[c] = b[4 * c + 2];
      if (index[c] > 88) { *why = "IMA ADPCM: step index above 88 in a block header"; return -1; }
      out[at * channels + c] = (int16_t)pred[c];
    }
    const uint8_t* d = b + 4 * channels;
    for (int s = 1; s < n; ++s) {
      const int g = (s - 1) >> 3, k = (s - 1) & 7;        // group of eight, code inside it
      for (int c = 0; c < channels; ++c) {
        const uint8_t byte = d[(g * channels + c) * 4 + (k >> 1)];
        out[(at + s) * channels + c] = step_once((k & 1) ? byte >> 4 : byte & 15, pred[c], index[c]);
      }
    }
    at += n;
  }
  return at;
}

}  // namespace adpcm
