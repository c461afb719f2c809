// Dev tool: mutation fuzzing of the host-side decoders that parse untrusted files (csrc/flac_decode.h, csrc/adpcm_decode.h)
// under the host sanitizers.  No GPU, no HIP: the two headers are plain C++.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -pthread -I vtc_amd/csrc tools/fuzz_decoders.cpp -o /tmp/fuzz_decoders
//   /tmp/fuzz_decoders ITERATIONS SEED file.flac [more seeds ...]
// Every mutant is decoded the way the library does it (parse_header -> decode into a buffer of the declared size, also with
// decode_parallel and with a too-small buffer); any out-of-bounds access, overflow or leak ends the run with a report.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "adpcm_decode.h"
#include "flac_decode.h"

static std::vector<uint8_t> read_file(const char* path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

static void decode_all(const std::vector<uint8_t>& d, std::mt19937_64& rng, long long* ok, long long* bad) {
  flac::StreamInfo si;
  flac::Error err;
  if (!flac::parse_header(d.data(), (int64_t)d.size(), si, err)) { ++*bad; return; }
  // the caller sizes its buffer from the header's count (bounded here: a mutant may claim 2^36 samples)
  int64_t cap = si.total > 0 ? si.total : 1 << 16;
  if (cap > (1 << 20)) cap = 1 << 20;
  const int nch = si.channels > 0 ? si.channels : 1;
  for (int fmt = 0; fmt < 3; ++fmt) {
    if (fmt == 0 && si.bps > 16) continue;
    const size_t esz = fmt == 0 ? 2 : 4;
    std::vector<uint8_t> out((size_t)cap * nch * esz + 16);
    flac::StreamInfo s2;
    flac::Error e2;
    const int64_t got = flac::decode(d.data(), (int64_t)d.size(), out.data(), fmt, cap, s2, e2);
    if (got >= 0) ++*ok; else ++*bad;
  }
  {   // a buffer smaller than the stream: must stop with the capacity status, not write past it
    const int64_t small = 1 + (int64_t)(rng() % 4096);
    std::vector<uint8_t> out((size_t)small * nch * 4);
    flac::StreamInfo s2;
    flac::Error e2;
    (void)flac::decode(d.data(), (int64_t)d.size(), out.data(), 2, small, s2, e2);
  }
  {   // counting only
    flac::StreamInfo s2;
    flac::Error e2;
    (void)flac::decode(d.data(), (int64_t)d.size(), nullptr, 2, 0, s2, e2);
  }
  if ((rng() & 7) == 0) {
    std::vector<uint8_t> out((size_t)cap * nch * 4 + 16);
    flac::StreamInfo s2;
    flac::Error e2;
    (void)flac::decode_parallel(d.data(), (int64_t)d.size(), out.data(), 2, cap, 3, s2, e2);
  }
}

static void mutate(std::vector<uint8_t>& d, std::mt19937_64& rng) {
  const int n_ops = 1 + (int)(rng() % 4);
  for (int k = 0; k < n_ops && !d.empty(); ++k) {
    const size_t at = rng() % d.size();
    switch (rng() % 7) {
      case 0: d[at] ^= (uint8_t)(1u << (rng() % 8)); break;                                  // bit flip
      case 1: d[at] = (uint8_t)rng(); break;                                                // random byte
      case 2: d.resize(at + 1); break;                                                      // truncation
      case 3: d.insert(d.begin() + at, (uint8_t)rng()); break;                              // insertion
      case 4: d.erase(d.begin() + at); break;                                               // deletion
      case 5: { const size_t want = rng() % 64, room = d.size() - at;                       // a run of 0x00 / 0xff
                memset(d.data() + at, (rng() & 1) ? 0xff : 0x00, want < room ? want : room); break; }
      default: if (at < 64) d[at] = (uint8_t)rng(); else d[rng() % 64] ^= (uint8_t)rng(); break;   // header fields
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s ITERATIONS SEED seed.flac [...]\n", argv[0]); return 2; }
  const long long iters = atoll(argv[1]);
  std::mt19937_64 rng((uint64_t)atoll(argv[2]));
  std::vector<std::vector<uint8_t>> seeds;
  for (int i = 3; i < argc; ++i) seeds.push_back(read_file(argv[i]));
  long long ok = 0, bad = 0;
  for (auto& s : seeds) decode_all(s, rng, &ok, &bad);   // the seeds themselves must decode
  if (bad) { fprintf(stderr, "a seed stream does not decode\n"); return 1; }
  for (long long it = 0; it < iters; ++it) {
    std::vector<uint8_t> d = seeds[rng() % seeds.size()];
    mutate(d, rng);
    // frame CRCs stop most mutants at the first damaged frame; every fourth mutant gets its frame CRC-16s left wrong on
    // purpose but its header CRC-8 region untouched, the others are decoded as they are
    decode_all(d, rng, &ok, &bad);
    // IMA ADPCM: the same bytes as blocks of random geometry
    {
      const int ch = 1 + (int)(rng() % 9), align = (int)(rng() % 2100), spb = (int)(rng() % 5000) - 100;
      const int64_t cap = (int64_t)(rng() % 70000);
      std::vector<int16_t> out((size_t)cap * (ch > 8 ? 8 : ch) + 8);
      const char* why = "";
      (void)adpcm::decode_ima(d.data(), (int64_t)d.size(), ch, align, spb, out.data(), cap, &why);
    }
  }
  printf("fuzz: %lld mutants, %lld decodes accepted, %lld rejected, no sanitizer report\n", iters, ok, bad);
  return 0;
}
