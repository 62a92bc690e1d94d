// huff_bench.cpp -- host entropy decoder scaling: decode one JPEG repeatedly with 1..N pool workers.
//   g++ -O3 -std=c++17 -pthread tools/huff_bench.cpp libjpeg_amd/csrc/host_decoder.cpp -o /tmp/huff_bench
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iterator>

#include "../libjpeg_amd/csrc/host_decoder.hpp"
int main(int argc, char **argv)
{
  if (argc < 2) return 1;
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), {});
  mij::HostDecoder h;
  int rc = h.parse(d.data(), d.size(), false);
  printf("parse rc=%d %dx%d intervals=%zu\n", rc, h.info.width, h.info.height, h.scans.empty() ? 0 : h.scans[0].interval_begin.size());
  if (rc) return 1;
  {
    double best = 1e9;
    for (int rep = 0; rep < 20; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      rc = h.parse(d.data(), d.size(), false);
      double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (t < best) best = t;
    }
    printf("parse (headers + restart marker search) %.3f ms\n", best * 1e3);
  }
  std::vector<int16_t> c(h.info.coef_count);
  for (int th : {1, 2, 4, 8, 16, 32, 64, 128, 256}) {
    double best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      rc = h.parse(d.data(), d.size(), false);
      rc |= h.decode(c.data(), th, nullptr);
      double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (t < best) best = t;
    }
    printf("threads %3d rc=%d parse+decode %.2f ms  (%.0f Mpix/s)\n", th, rc, best * 1e3, h.info.width * (double)h.info.height / best / 1e6);
  }
  return 0;
}
