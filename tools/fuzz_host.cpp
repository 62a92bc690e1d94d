// fuzz_host.cpp -- mutation fuzzing of the host decoder under AddressSanitizer / UBSan.
//   g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined tools/fuzz_host.cpp libjpeg_amd/csrc/host_decoder.cpp libjpeg_amd/csrc/encoder.cpp -o /tmp/fuzz_host
//   /tmp/fuzz_host tests/golden/*.jpg
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "../libjpeg_amd/csrc/host_decoder.hpp"
static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
int main(int argc, char **argv)
{
  // FUZZ_THREADS=24: windows of many scans -- the refinement chains on bit masks, their appliers; 2 otherwise
  const int fuzz_threads = getenv("FUZZ_THREADS") ? atoi(getenv("FUZZ_THREADS")) : 2;
  long ok = 0, bad = 0;
  for (int a = 1; a < argc; a++) {
    std::ifstream f(argv[a], std::ios::binary);
    std::vector<uint8_t> base((std::istreambuf_iterator<char>(f)), {});
    for (int t = 0; t < 400; t++) {
      std::vector<uint8_t> d = base;
      switch (t % 4) {
      case 0: for (int i = 0, n = 1 + rnd() % 5; i < n; i++) d[2 + rnd() % (d.size() - 2)] = (uint8_t)rnd(); break;
      case 1: for (int i = 0, n = 1 + rnd() % 3; i < n; i++) d[2 + rnd() % (std::min<size_t>(d.size(), 700) - 2)] = (uint8_t)rnd(); break;
      case 2: d.resize(4 + rnd() % (d.size() - 4)); break;
      default: { size_t at = 2 + rnd() % (d.size() - 2); std::vector<uint8_t> g(1 + rnd() % 40); for (auto &x : g) x = (uint8_t)rnd(); d.insert(d.begin() + at, g.begin(), g.end()); }
      }
      mij::HostDecoder h;
      // the device's copy of the entropy coded data (no byte stuffing, no markers): written by the marker search itself
      // (the sink, every other trial) or afterwards in pieces; both must stay inside what the stream's size allows
      std::vector<uint8_t> sink(d.size() + 64, 0xAA);
      if (t & 8) h.set_unstuff_sink(sink.data(), d.size());
      int rc = h.parse(d.data(), d.size(), false);
      if (!rc && !h.scans.empty()) {
        const mij::Scan &s0 = h.scans[0];
        if (s0.unstuffed_size > d.size()) { printf("unstuffed size beyond the stream: %s trial %d\n", argv[a], t); return 1; }
        std::vector<uint8_t> copy(d.size() + 64, 0x55); // (+ the slack the vector copy may touch)
        std::vector<mij::HostDecoder::UnstuffPiece> pieces;
        h.unstuff_pieces(0, (t % 3 == 0) ? 64 : (size_t)1 << 16, pieces);
        for (const auto &p : pieces) h.unstuff_piece(0, p, copy.data());
        if (s0.unstuffed_at == sink.data() && memcmp(sink.data(), copy.data(), s0.unstuffed_size)) { printf("sink and pieces disagree: %s trial %d\n", argv[a], t); return 1; }
      }
      if (!rc) {
        std::vector<int16_t> c((size_t)h.info.coef_count + 64);
        rc = h.decode(c.data(), fuzz_threads, nullptr);
      }
      // the alpha channel of a JPEG XT file: a child decoder on the ALFA box's codestream with the boxes translated (what
      // capi.cpp's decode_alpha_channel does)
      if (!rc && h.has_alpha()) {
        const uint8_t *ap = nullptr;
        size_t an = 0;
        if (h.alpha_stream(&ap, &an) && an) {
          std::vector<uint8_t> adata(ap, ap + an);
          mij::HostDecoder child;
          child.preset_boxes(h.alpha_boxes());
          int arc = child.parse(adata.data(), adata.size(), false);
          if (!arc) {
            std::vector<int16_t> c((size_t)child.info.coef_count + 64);
            arc = child.decode(c.data(), fuzz_threads, nullptr);
          }
          (void)arc;
        }
      }
      (rc ? bad : ok)++;
    }
  }
  printf("decoded %ld, rejected %ld\n", ok, bad);
  return 0;
}
