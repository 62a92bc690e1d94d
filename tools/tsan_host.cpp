// tsan_host.cpp -- the multi-threaded host decoder under ThreadSanitizer ("race detection").
//   g++ -O1 -g -std=c++17 -pthread -fsanitize=thread tools/tsan_host.cpp libjpeg_amd/csrc/host_decoder.cpp libjpeg_amd/csrc/encoder.cpp -o /tmp/tsan_host
//   /tmp/tsan_host tests/golden/*.jpg          (tools/tsan_host.sh does both)
// What runs concurrently in the product and therefore here: the restart-interval-parallel decode and the speculative
// (self-synchronising) decode of scans without restart markers on the library's worker pool (HostDecoder::decode with
// 8 threads), several decoder objects decoding different streams at the same time from different caller threads (the pool is
// shared process-wide), and damaged streams taking the sequential walk while other objects are in the parallel path.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <thread>
#include <vector>

#include "../libjpeg_amd/csrc/host_decoder.hpp"

static std::vector<uint8_t> slurp(const char *fn)
{
  std::ifstream f(fn, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {});
}

int main(int argc, char **argv)
{
  std::vector<std::vector<uint8_t>> files;
  for (int a = 1; a < argc; a++) files.push_back(slurp(argv[a]));
  if (files.empty()) return 2;
  long decoded = 0, rejected = 0, mismatched = 0;
  // 1. one object at a time, 8 pool threads, against the single-threaded decode of the same stream
  for (const auto &d : files) {
    mij::HostDecoder a, b;
    if (a.parse(d.data(), d.size(), false) || b.parse(d.data(), d.size(), false)) { rejected++; continue; }
    std::vector<int16_t> ca((size_t)a.info.coef_count + 64), cb((size_t)b.info.coef_count + 64);
    const int ra = a.decode(ca.data(), 8, nullptr), rb = b.decode(cb.data(), 1, nullptr);
    if (ra != rb || (!ra && memcmp(ca.data(), cb.data(), (size_t)a.info.coef_count * 2))) mismatched++;
    (ra ? rejected : decoded)++;
  }
  // 2. four caller threads, each with its own objects, all sharing the pool; every third stream is damaged
  std::vector<std::thread> ts;
  std::vector<long> ok(4, 0);
  for (int t = 0; t < 4; t++)
    ts.emplace_back([&, t] {
      for (size_t i = (size_t)t; i < files.size() * 3; i += 4) {
        std::vector<uint8_t> d = files[i % files.size()];
        if (i % 3 == 2 && d.size() > 600) { d[d.size() / 2] ^= 0x5a; d[d.size() / 2 + 1] = 0xff; d[d.size() / 2 + 2] = 0xd3; }
        mij::HostDecoder h;
        if (h.parse(d.data(), d.size(), false)) continue;
        std::vector<int16_t> c((size_t)h.info.coef_count + 64);
        if (!h.decode(c.data(), 4, nullptr)) ok[(size_t)t]++;
      }
    });
  for (auto &t : ts) t.join();
  printf("single: decoded %ld rejected %ld mismatched %ld; concurrent: %ld %ld %ld %ld decoded\n", decoded, rejected, mismatched, ok[0], ok[1], ok[2], ok[3]);
  return mismatched ? 1 : 0;
}
