// pipe_bench.cpp -- end-to-end frame pipeline straight on the C ABI: D decoder objects, one thread each.
//   g++ -O2 -std=c++17 -pthread tools/pipe_bench.cpp -Iinclude -Llibjpeg_amd -lmijpeg -Wl,-rpath,$PWD/libjpeg_amd -o tools/pipe_bench
#include <atomic>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <thread>
#include <vector>

#include "mijpeg.h"
int main(int argc, char **argv)
{
  if (argc < 2) return 1;
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), {});
  for (int depth : {1, 2, 3, 4}) {
    const int frames = 32;
    std::atomic<int> next{0};
    std::vector<std::thread> ts;
    std::vector<mijpeg_decoder *> decs(depth);
    std::vector<void *> bufs(depth);
    mijpeg_info info;
    for (int i = 0; i < depth; i++) {
      if (mijpeg_create(&decs[i], 0)) return 2;
      mijpeg_set_input(decs[i], data.data(), data.size());
      mijpeg_read_header(decs[i], &info);
      bufs[i] = mijpeg_host_alloc((size_t)info.width * info.height * 3);
      mijpeg_decode_coefficients(decs[i], 0); // warm
      mijpeg_reconstruct_host(decs[i], bufs[i], (int64_t)info.width * 3, 0);
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < depth; i++)
      ts.emplace_back([&, i] {
        while (next.fetch_add(1) < frames) {
          mijpeg_set_input(decs[i], data.data(), data.size());
          if (mijpeg_decode_coefficients(decs[i], 0)) { fprintf(stderr, "decode failed\n"); return; }
          if (mijpeg_reconstruct_host(decs[i], bufs[i], (int64_t)info.width * 3, 0)) { fprintf(stderr, "reconstruct failed\n"); return; }
        }
      });
    for (auto &t : ts) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("depth %d: %.2f ms/frame = %.0f Mpix/s\n", depth, dt / frames * 1e3, (double)info.width * info.height * frames / dt / 1e6);
    for (int i = 0; i < depth; i++) { mijpeg_host_free(bufs[i]); mijpeg_destroy(decs[i]); }
  }
  return 0;
}
