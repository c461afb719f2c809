// Host-side FLAC decoder (RFC 9639) for the audio front-end: the reference's clips are packaged as FLAC by default
// (src/packaging/writer.py:36,119; src/analysis/vtc_on_clips.py:163-166 prefers the .flac member) and its manifests
// may list .flac recordings (scripts/make_manifest.py:26-39); the reference decodes them through torchcodec/ffmpeg
// inside segma.  FLAC is lossless, so the contract is exact: the decoded integers are the encoder's input, and the
// stream carries their MD5.  Pure host code (no device work): it runs on the pipeline's reader threads, outside the
// GIL, and hands interleaved PCM to vtc_prepare_audio.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace flac {

struct StreamInfo {
  int32_t min_block = 0, max_block = 0, sample_rate = 0, channels = 0, bps = 0;
  int64_t total = 0;  // inter-channel samples; 0 = unknown
  uint8_t md5[16] = {0};
  int64_t audio_offset = 0;  // byte offset of the first frame
};

struct Error {
  std::string msg;
};

// CRC-8 (poly 0x07) of the frame header and CRC-16 (poly 0x8005) of the frame, both MSB-first, initial value 0.
// t16[k][v] is the CRC-16 of byte v followed by k zero bytes (slicing-by-8: eight bytes per step).
struct CrcTables {
  uint8_t t8[256];
  uint16_t t16[8][256];
  constexpr CrcTables() : t8(), t16() {
    for (int v = 0; v < 256; ++v) {
      uint8_t c = (uint8_t)v;
      for (int b = 0; b < 8; ++b) c = (uint8_t)((c << 1) ^ ((c & 0x80) ? 0x07 : 0));
      t8[v] = c;
      uint16_t d = (uint16_t)(v << 8);
      for (int b = 0; b < 8; ++b) d = (uint16_t)((d << 1) ^ ((d & 0x8000) ? 0x8005 : 0));
      t16[0][v] = d;
    }
    for (int k = 1; k < 8; ++k)
      for (int v = 0; v < 256; ++v) t16[k][v] = (uint16_t)((t16[k - 1][v] << 8) ^ t16[0][t16[k - 1][v] >> 8]);
  }
};
static constexpr CrcTables kCrc{};

inline uint8_t crc8(const uint8_t* p, const uint8_t* end) {
  uint8_t c = 0;
  for (; p < end; ++p) c = kCrc.t8[c ^ *p];
  return c;
}
inline uint16_t crc16(const uint8_t* p, const uint8_t* end) {
  uint16_t c = 0;
  for (; p + 8 <= end; p += 8)
    c = (uint16_t)(kCrc.t16[7][p[0] ^ (c >> 8)] ^ kCrc.t16[6][p[1] ^ (c & 0xff)] ^ kCrc.t16[5][p[2]] ^ kCrc.t16[4][p[3]] ^
                   kCrc.t16[3][p[4]] ^ kCrc.t16[2][p[5]] ^ kCrc.t16[1][p[6]] ^ kCrc.t16[0][p[7]]);
  for (; p < end; ++p) c = (uint16_t)((c << 8) ^ kCrc.t16[0][(c >> 8) ^ *p]);
  return c;
}

// MSB-first bit reader over a byte range.  Reads past the end return zeros and set `overrun`.
struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  const uint8_t* begin;
  uint64_t buf = 0;  // left-aligned: the next bit is bit 63
  int nbits = 0;
  bool overrun = false;
  BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e), begin(b) {}
  // position of the next unread bit, in bits from `begin`, and the way back from such a position (read_residual's fast loop
  // works on a plain bit offset with one unaligned 8-byte load per code)
  inline uint64_t bit_offset() const { return (uint64_t)(p - begin) * 8 - (uint64_t)nbits; }
  inline void seek_bits(uint64_t bp) {
    p = begin + (bp >> 3);
    buf = 0;
    nbits = 0;
    const int r = (int)(bp & 7);
    if (r) {
      refill();
      buf <<= r;
      nbits -= r;
    }
  }
  inline void refill() {
    if (p + 8 <= end) {  // fast path: one unaligned 8-byte load
      uint64_t w;
      memcpy(&w, p, 8);
      w = __builtin_bswap64(w);
      const int take = (64 - nbits) >> 3;  // whole bytes that fit
      if (take == 0) return;
      buf |= w >> nbits;  // nbits <= 56 here
      p += take;
      nbits += take * 8;
      if (nbits < 64) buf &= ~0ULL << (64 - nbits);  // keep the bits below the valid part zero
      return;
    }
    while (nbits <= 56) {
      uint64_t byte = 0;
      if (p < end) byte = *p;
      else if (p >= end + 8) overrun = true;  // more than the look-ahead slack consumed
      ++p;
      buf |= byte << (56 - nbits);
      nbits += 8;
    }
  }
  inline uint32_t bits(int n) {  // 0 <= n <= 32
    if (n == 0) return 0;
    if (nbits < n) refill();
    uint32_t v = (uint32_t)(buf >> (64 - n));
    buf <<= n;
    nbits -= n;
    return v;
  }
  inline int32_t sbits(int n) {  // signed, 1 <= n <= 32
    uint32_t v = bits(n);
    if (n == 32) return (int32_t)v;
    uint32_t sign = 1u << (n - 1);
    return (int32_t)((v ^ sign) - sign);
  }
  inline int64_t sbits_wide(int n) {  // up to 33 bits (side channel of 32-bit audio)
    if (n <= 32) return sbits(n);
    int64_t hi = sbits(n - 16);
    return hi * 65536 + (int64_t)bits(16);
  }
  inline uint32_t unary() {  // number of 0 bits before the next 1 bit
    uint32_t q = 0;
    for (;;) {
      if (nbits == 0) refill();
      if (buf == 0) {
        q += nbits;
        nbits = 0;
        if (p >= end + 8) {
          overrun = true;
          return q;
        }
        continue;
      }
      int z = __builtin_clzll(buf);
      if (z >= nbits) {  // only zeros in the valid part (buf != 0 cannot happen here, kept for safety)
        q += nbits;
        buf = 0;
        nbits = 0;
        continue;
      }
      q += z;
      buf = (buf << z) << 1;  // z + 1 can be 64
      nbits -= (z + 1);
      return q;
    }
  }
  inline void align() {
    int r = nbits & 7;
    buf <<= r;
    nbits -= r;
  }
  // byte position of the next unread bit (only meaningful when aligned)
  inline const uint8_t* pos() const { return p - nbits / 8; }
};

inline bool parse_header(const uint8_t* d, int64_t n, StreamInfo& si, Error& err) {
  int64_t o = 0;
  if (n >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3') {  // ID3v2 tag in front of the stream
    int64_t sz = ((int64_t)(d[6] & 0x7f) << 21) | ((int64_t)(d[7] & 0x7f) << 14) | ((int64_t)(d[8] & 0x7f) << 7) | (d[9] & 0x7f);
    o = 10 + sz + ((d[5] & 0x10) ? 10 : 0);
  }
  if (o + 4 > n || memcmp(d + o, "fLaC", 4) != 0) {
    err.msg = "not a FLAC stream (no fLaC marker)";
    return false;
  }
  o += 4;
  bool have_info = false;
  for (;;) {
    if (o + 4 > n) {
      err.msg = "truncated FLAC metadata";
      return false;
    }
    bool last = d[o] & 0x80;
    int type = d[o] & 0x7f;
    int64_t len = ((int64_t)d[o + 1] << 16) | ((int64_t)d[o + 2] << 8) | d[o + 3];
    o += 4;
    if (o + len > n) {
      err.msg = "truncated FLAC metadata block";
      return false;
    }
    if (type == 0) {
      if (len < 34) {
        err.msg = "short STREAMINFO";
        return false;
      }
      const uint8_t* s = d + o;
      si.min_block = (s[0] << 8) | s[1];
      si.max_block = (s[2] << 8) | s[3];
      si.sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      si.channels = ((s[12] >> 1) & 7) + 1;
      si.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      si.total = ((int64_t)(s[13] & 0xf) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
      memcpy(si.md5, s + 18, 16);
      have_info = true;
    } else if (type == 127) {
      err.msg = "invalid FLAC metadata block type 127";
      return false;
    }
    o += len;
    if (last) break;
  }
  if (!have_info) {
    err.msg = "FLAC stream without STREAMINFO";
    return false;
  }
  si.audio_offset = o;
  return true;
}

// Residual of one subframe into res[order .. blocksize).  Returns false on a malformed stream.
inline bool read_residual(BitReader& br, int64_t* out, int blocksize, int order, Error& err) {
  int method = br.bits(2);
  if (method > 1) {
    err.msg = "reserved residual coding method";
    return false;
  }
  int pbits = method ? 5 : 4;
  uint32_t esc = method ? 31 : 15;
  int porder = br.bits(4);
  int nparts = 1 << porder;
  if ((blocksize & (nparts - 1)) != 0 || (blocksize >> porder) < order) {
    err.msg = "invalid Rice partition order";
    return false;
  }
  int i = order;
  for (int part = 0; part < nparts; ++part) {
    int cnt = (blocksize >> porder) - (part == 0 ? order : 0);
    uint32_t k = br.bits(pbits);
    if (k == esc) {
      int nb = br.bits(5);
      for (int j = 0; j < cnt; ++j) out[i++] = nb ? br.sbits(nb) : 0;
    } else {
      int j = 0;
      if (br.end - br.begin >= 16) {
        // fast loop: one unaligned big-endian 8-byte load at bit offset bp, shifted by bp & 7, holds >= 57 valid bits = about five
        // codes of a 16-bit stream; they are taken out of the register one after the other (count leading zeros, shift), so the
        // chain load -> byte swap -> count -> next address is paid once per word, not once per code (19 -> 8 cycles per sample).
        // A code longer than the window (a unary run of 25+ zeros with a wide parameter) and the last bytes of the buffer leave
        // the loop for the general reader below.
        uint64_t bp = br.bit_offset();
        const uint8_t* lim = br.end - 8;
        while (j < cnt) {
          const uint8_t* q = br.begin + (bp >> 3);
          if (q > lim) break;
          uint64_t w;
          memcpy(&w, q, 8);
          const int r = (int)(bp & 7);
          w = __builtin_bswap64(w) << r;
          int avail = 64 - r, used = 0;
          while (w != 0) {
            const int z = __builtin_clzll(w);
            const int len = z + 1 + (int)k;
            if (len > avail) break;
            const uint64_t b = (w << z) << 1;
            const uint64_t u = ((uint64_t)z << k) | (k ? b >> (64 - k) : 0);
            w = b << k;
            avail -= len;
            used += len;
            out[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
            if (++j == cnt) break;
          }
          if (used == 0) break;
          bp += (uint64_t)used;
        }
        br.seek_bits(bp);
      }
      for (; j < cnt; ++j) {
        uint64_t u;
        if (br.nbits < 32 + (int)k) br.refill();
        const int z = br.buf ? __builtin_clzll(br.buf) : 64;
        if (z + 1 + (int)k <= br.nbits && z < 32) {  // stop bit and the k low bits are in the buffer: one shift each
          uint64_t b = (br.buf << z) << 1;
          u = ((uint64_t)z << k) | (k ? b >> (64 - k) : 0);
          br.buf = b << k;
          br.nbits -= z + 1 + (int)k;
        } else {
          uint32_t q = br.unary();
          u = ((uint64_t)q << k) | br.bits(k);
        }
        out[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
      }
    }
    if (br.overrun) {
      err.msg = "truncated FLAC frame (residual)";
      return false;
    }
  }
  return true;
}

// Prediction arithmetic wraps (unsigned): a valid stream stays far inside 64 bits (<= 33-bit samples x 15-bit coefficients x
// 32 taps), a damaged one whose frame CRC has not been reached yet may not, and signed overflow would be undefined behaviour
// (found by tools/fuzz_decoders.cpp under UBSan).  Same instructions as the signed form.
inline int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
inline int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
inline int64_t wmul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

// s[i] += (sum_j coef[j] * s[i-1-j]) >> shift, the order known at compile time so the taps unroll
template <int ORDER>
inline void lpc_restore_n(int64_t* s, int n, const int32_t* coef, int shift) {
  int64_t c[ORDER];
  for (int j = 0; j < ORDER; ++j) c[j] = coef[j];
  for (int i = ORDER; i < n; ++i) {
    int64_t acc = 0;
#pragma GCC unroll 32
    for (int j = 0; j < ORDER; ++j) acc = wadd(acc, wmul(c[j], s[i - 1 - j]));
    s[i] = wadd(s[i], acc >> shift);
  }
}
template <int... I>
inline void lpc_dispatch(int64_t* s, int n, int order, const int32_t* coef, int shift, std::integer_sequence<int, I...>) {
  using Fn = void (*)(int64_t*, int, const int32_t*, int);
  static constexpr Fn table[] = {&lpc_restore_n<I + 1>...};
  table[order - 1](s, n, coef, shift);
}
inline void lpc_restore(int64_t* s, int n, int order, const int32_t* coef, int shift) {
  lpc_dispatch(s, n, order, coef, shift, std::make_integer_sequence<int, 32>{});
}

inline bool read_subframe(BitReader& br, int64_t* s, int blocksize, int bps, Error& err) {
  if (br.bits(1)) {
    err.msg = "subframe padding bit set";
    return false;
  }
  int type = br.bits(6);
  int wasted = 0;
  if (br.bits(1)) wasted = (int)br.unary() + 1;
  bps -= wasted;
  if (bps <= 0) {
    err.msg = "wasted bits exceed the sample size";
    return false;
  }
  if (type == 0) {
    int64_t v = br.sbits_wide(bps);
    for (int i = 0; i < blocksize; ++i) s[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; ++i) s[i] = br.sbits_wide(bps);
  } else if (type >= 8 && type <= 12) {
    int order = type - 8;
    if (order > blocksize) {
      err.msg = "fixed predictor order exceeds the block size";
      return false;
    }
    for (int i = 0; i < order; ++i) s[i] = br.sbits_wide(bps);
    if (!read_residual(br, s, blocksize, order, err)) return false;
    switch (order) {
      case 1:
        for (int i = 1; i < blocksize; ++i) s[i] = wadd(s[i], s[i - 1]);
        break;
      case 2:
        for (int i = 2; i < blocksize; ++i) s[i] = wadd(s[i], wsub(wmul(2, s[i - 1]), s[i - 2]));
        break;
      case 3:
        for (int i = 3; i < blocksize; ++i) s[i] = wadd(s[i], wadd(wsub(wmul(3, s[i - 1]), wmul(3, s[i - 2])), s[i - 3]));
        break;
      case 4:
        for (int i = 4; i < blocksize; ++i)
          s[i] = wadd(s[i], wsub(wadd(wsub(wmul(4, s[i - 1]), wmul(6, s[i - 2])), wmul(4, s[i - 3])), s[i - 4]));
        break;
      default:
        break;
    }
  } else if (type >= 32) {
    int order = (type & 31) + 1;
    if (order > blocksize) {
      err.msg = "LPC order exceeds the block size";
      return false;
    }
    for (int i = 0; i < order; ++i) s[i] = br.sbits_wide(bps);
    int prec = br.bits(4) + 1;
    if (prec == 16) {
      err.msg = "invalid LPC precision";
      return false;
    }
    int shift = br.sbits(5);
    if (shift < 0) {
      err.msg = "negative LPC shift";
      return false;
    }
    int32_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.sbits(prec);
    if (!read_residual(br, s, blocksize, order, err)) return false;
    lpc_restore(s, blocksize, order, coef, shift);
  } else {
    err.msg = "reserved subframe type";
    return false;
  }
  if (wasted)
    for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
  if (br.overrun) {
    err.msg = "truncated FLAC frame";
    return false;
  }
  return true;
}

// One frame starting at byte `o`: parses and checks the header (CRC-8), decodes the subframes into `block`
// (channel c at block[c * 65536]), checks the frame CRC-16 and undoes the stereo decorrelation.  On success returns true
// with the block size, the coded frame / sample number and the byte offset of the next frame.
struct FrameInfo {
  int blocksize = 0;
  bool variable = false;   // blocking strategy: the coded number is a sample number, not a frame number
  uint64_t number = 0;
  int64_t next = 0;
};

inline bool decode_frame(const uint8_t* d, int64_t n, int64_t o, const StreamInfo& si, int64_t* block, FrameInfo& fi, Error& err) {
  char b[128];
  if (o + 2 > n || d[o] != 0xff || (d[o + 1] & 0xfe) != 0xf8) {
    snprintf(b, sizeof b, "lost FLAC frame sync at byte %lld", (long long)o);
    err.msg = b;
    return false;
  }
  BitReader br(d + o, d + n);
  br.bits(15);
  fi.variable = br.bits(1);
  int bs_code = br.bits(4), sr_code = br.bits(4), ch_code = br.bits(4), ss_code = br.bits(3);
  if (br.bits(1)) {
    err.msg = "reserved frame header bit set";
    return false;
  }
  // UTF-8 style coded frame / sample number (up to 36 bits)
  uint32_t b0 = br.bits(8);
  int extra = 0;
  uint64_t num = b0;
  if (b0 == 0xff) {
    err.msg = "invalid coded frame number";
    return false;
  }
  if (b0 >= 0xfe) extra = 6, num = 0;
  else if (b0 >= 0xfc) extra = 5, num = b0 & 0x01;
  else if (b0 >= 0xf8) extra = 4, num = b0 & 0x03;
  else if (b0 >= 0xf0) extra = 3, num = b0 & 0x07;
  else if (b0 >= 0xe0) extra = 2, num = b0 & 0x0f;
  else if (b0 >= 0xc0) extra = 1, num = b0 & 0x1f;
  else if (b0 >= 0x80) {
    err.msg = "invalid coded frame number";
    return false;
  }
  for (int i = 0; i < extra; ++i) {
    const uint32_t c = br.bits(8);
    if ((c & 0xc0) != 0x80) {
      err.msg = "invalid coded frame number";
      return false;
    }
    num = (num << 6) | (c & 0x3f);
  }
  fi.number = num;
  int blocksize;
  if (bs_code == 0) {
    err.msg = "reserved block size code";
    return false;
  } else if (bs_code == 1) blocksize = 192;
  else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
  else if (bs_code == 6) blocksize = br.bits(8) + 1;
  else if (bs_code == 7) blocksize = br.bits(16) + 1;
  else blocksize = 256 << (bs_code - 8);
  if (sr_code == 12) br.bits(8);
  else if (sr_code == 13 || sr_code == 14) br.bits(16);
  else if (sr_code == 15) {
    err.msg = "invalid sample rate code";
    return false;
  }
  const uint8_t* hdr_end = br.pos();
  if (hdr_end + 1 > d + n) {
    err.msg = "truncated FLAC frame header";
    return false;
  }
#ifdef VTC_FLAC_FUZZ_SKIP_CRC   // tools/fuzz_decoders.cpp only: lets damaged frames reach the subframe parsers
  if ((void)br.bits(8), false) {
#else
  if (crc8(d + o, hdr_end) != br.bits(8)) {
#endif
    snprintf(b, sizeof b, "FLAC frame header CRC mismatch at byte %lld", (long long)o);
    err.msg = b;
    return false;
  }
  int bps;
  switch (ss_code) {
    case 0: bps = si.bps; break;
    case 1: bps = 8; break;
    case 2: bps = 12; break;
    case 4: bps = 16; break;
    case 5: bps = 20; break;
    case 6: bps = 24; break;
    case 7: bps = 32; break;
    default:
      err.msg = "reserved sample size code";
      return false;
  }
  if (ch_code > 10) {
    err.msg = "reserved channel assignment";
    return false;
  }
  const int nch = si.channels, frame_ch = ch_code < 8 ? ch_code + 1 : 2;
  if (frame_ch != nch || bps != si.bps) {
    err.msg = "channel count or sample size changes inside the stream (not supported)";
    return false;
  }
  for (int c = 0; c < nch; ++c) {
    int side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
    if (!read_subframe(br, block + (size_t)c * 65536, blocksize, bps + side, err)) return false;
  }
  br.align();
  const uint8_t* body_end = br.pos();
  if (body_end + 2 > d + n) {
    err.msg = "truncated FLAC frame (no CRC)";
    return false;
  }
#ifdef VTC_FLAC_FUZZ_SKIP_CRC
  if (false) {
#else
  if (crc16(d + o, body_end) != (uint16_t)((body_end[0] << 8) | body_end[1])) {
#endif
    snprintf(b, sizeof b, "FLAC frame CRC mismatch at byte %lld", (long long)o);
    err.msg = b;
    return false;
  }
  int64_t* c0 = block;
  int64_t* c1 = block + 65536;
  if (ch_code == 8) {
    for (int i = 0; i < blocksize; ++i) c1[i] = wsub(c0[i], c1[i]);
  } else if (ch_code == 9) {
    for (int i = 0; i < blocksize; ++i) c0[i] = wadd(c0[i], c1[i]);
  } else if (ch_code == 10) {
    for (int i = 0; i < blocksize; ++i) {
      int64_t side = c1[i], mid = (int64_t)((uint64_t)c0[i] << 1) | (side & 1);
      c0[i] = wadd(mid, side) >> 1;
      c1[i] = wsub(mid, side) >> 1;
    }
  }
  fi.blocksize = blocksize;
  fi.next = (body_end + 2) - d;
  return true;
}

// Interleave `take` samples of a decoded block into the caller's buffer at sample index `at`.
inline void store_block(const int64_t* block, int nch, int64_t take, void* out, int out_format, int64_t at, float fscale) {
  for (int c = 0; c < nch; ++c) {
    const int64_t* src = block + (size_t)c * 65536;
    if (out_format == 0) {
      int16_t* dst = (int16_t*)out + at * nch + c;
      for (int64_t i = 0; i < take; ++i) dst[i * nch] = (int16_t)src[i];
    } else if (out_format == 1) {
      float* dst = (float*)out + at * nch + c;
      for (int64_t i = 0; i < take; ++i) dst[i * nch] = (float)src[i] * fscale;
    } else {
      int32_t* dst = (int32_t*)out + at * nch + c;
      for (int64_t i = 0; i < take; ++i) dst[i * nch] = (int32_t)src[i];
    }
  }
}

// out_format: 0 = int16 (bps <= 16, integers as stored), 1 = float32 (integer / 2^(bps-1)), 2 = int32 (integers as stored).
// out may be null (count only).  Returns the number of inter-channel samples decoded, -1 on a malformed stream, -2 when
// the buffer is too small.
inline int64_t decode(const uint8_t* d, int64_t n, void* out, int out_format, int64_t cap, StreamInfo& si, Error& err) {
  if (!parse_header(d, n, si, err)) return -1;
  if (out_format == 0 && si.bps > 16) {
    err.msg = "int16 output requested for a stream of more than 16 bits per sample";
    return -1;
  }
  const int nch = si.channels;
  std::vector<int64_t> block((size_t)nch * 65536);
  int64_t done = 0;
  int64_t o = si.audio_offset;
  const float fscale = 1.0f / (float)((int64_t)1 << (si.bps - 1));
  while (o < n) {
    // trailing padding / tags after the last frame: stop quietly once the declared total is reached
    if (si.total && done >= si.total) break;
    if (si.total == 0 && done > 0 && (o + 2 > n || d[o] != 0xff || (d[o + 1] & 0xfe) != 0xf8))
      break;  // unknown length: anything that is not a frame ends the stream
    FrameInfo fi;
    if (!decode_frame(d, n, o, si, block.data(), fi, err)) return -1;
    o = fi.next;
    int64_t take = fi.blocksize;
    if (si.total && done + take > si.total) take = si.total - done;
    if (out) {
      if (done + take > cap) {
        err.msg = "output buffer too small for the FLAC stream";
        return -2;
      }
      store_block(block.data(), nch, take, out, out_format, done, fscale);
    }
    done += take;
  }
  if (si.total && done < si.total) {
    char b[128];
    snprintf(b, sizeof b, "FLAC stream ends after %lld of %lld samples", (long long)done, (long long)si.total);
    err.msg = b;
    return -1;
  }
  return done;
}

// Frames are self-contained, so a long stream can be decoded by several threads: the byte range is cut into n_threads
// pieces, every thread finds the first frame that STARTS in its piece (a candidate must pass the header CRC-8 and decode
// to a matching frame CRC-16) and decodes frames until one starts in the next piece; a frame's place in the output comes
// from its coded frame / sample number.  Afterwards the pieces must join seamlessly (byte offsets and sample positions);
// if they do not (a false sync that survived both CRCs, a stream without a declared length or with a varying nominal
// block size), the stream is decoded again by the sequential decoder, whose result is the reference.
inline int64_t decode_parallel(const uint8_t* d, int64_t n, void* out, int out_format, int64_t cap, int n_threads, StreamInfo& si,
                               Error& err) {
  if (!parse_header(d, n, si, err)) return -1;
  const int64_t audio = n - si.audio_offset;
  if (n_threads > 64) n_threads = 64;
  while (n_threads > 1 && audio < (int64_t)n_threads * (1 << 20)) --n_threads;   // pieces of at least 1 MiB
  if (!out || n_threads < 2 || si.total == 0) return decode(d, n, out, out_format, cap, si, err);
  if (out_format == 0 && si.bps > 16) {
    err.msg = "int16 output requested for a stream of more than 16 bits per sample";
    return -1;
  }
  if (si.total > cap) {
    err.msg = "output buffer too small for the FLAC stream";
    return -2;
  }
  struct Piece {
    int64_t first_byte = -1, end_byte = -1, first_sample = -1, end_sample = -1;
    bool ok = false;
  };
  std::vector<Piece> pieces((size_t)n_threads);
  const int nch = si.channels;
  const float fscale = 1.0f / (float)((int64_t)1 << (si.bps - 1));
  const StreamInfo info = si;
  auto work = [&](int t) {
    Piece& pc = pieces[(size_t)t];
    const int64_t lo = info.audio_offset + audio * t / n_threads, hi = info.audio_offset + audio * (t + 1) / n_threads;
    std::vector<int64_t> block((size_t)nch * 65536);
    Error e;
    FrameInfo fi;
    int64_t o = lo;
    if (t > 0) {   // lock onto the first real frame of this piece
      for (;; ++o) {
        if (o + 2 > hi) return;   // no frame starts here (pieces are >= 1 MiB, frames <= ~0.5 MiB: does not happen)
        if (d[o] != 0xff || (d[o + 1] & 0xfe) != 0xf8) continue;
        if (decode_frame(d, n, o, info, block.data(), fi, e)) break;
      }
    } else if (!decode_frame(d, n, o, info, block.data(), fi, e)) {
      return;
    }
    pc.first_byte = o;
    for (;;) {
      const int64_t at = fi.variable ? (int64_t)fi.number : (int64_t)fi.number * info.min_block;
      if (pc.first_sample < 0) pc.first_sample = at;
      else if (at != pc.end_sample) return;       // numbering is not contiguous: leave it to the sequential decoder
      if (at >= info.total) return;
      int64_t take = fi.blocksize;
      if (at + take > info.total) take = info.total - at;
      store_block(block.data(), nch, take, out, out_format, at, fscale);
      pc.end_sample = at + take;
      o = fi.next;
      pc.end_byte = o;
      if (o >= hi || pc.end_sample >= info.total) break;
      if (!decode_frame(d, n, o, info, block.data(), fi, e)) return;
    }
    pc.ok = true;
  };
  std::vector<std::thread> threads;
  for (int t = 1; t < n_threads; ++t) threads.emplace_back(work, t);
  work(0);
  for (auto& th : threads) th.join();
  bool good = pieces[0].ok && pieces[0].first_sample == 0;
  int64_t last_end = pieces[0].end_sample;
  for (int t = 1; good && t < n_threads; ++t) {
    const Piece& a = pieces[(size_t)t - 1];
    const Piece& b = pieces[(size_t)t];
    if (a.end_sample >= si.total && !b.ok && b.first_byte < 0) continue;   // stream ended inside an earlier piece (trailing tags)
    good = b.ok && b.first_byte == a.end_byte && b.first_sample == a.end_sample;
    if (good) last_end = b.end_sample;
  }
  if (good && last_end == si.total) return si.total;
  return decode(d, n, out, out_format, cap, si, err);   // anything unusual: the sequential decoder decides
}

}  // namespace flac
