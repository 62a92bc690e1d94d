// A C++ client of BASELINE config 4 built against include/mijpeg.h ONLY (plus the HIP runtime for its output buffer): the streams
// named on the command line -- one shape -- go through mijpeg_batch_pipeline_run into device memory and, full duplex, into pinned
// host memory; prints one line per frame with a 64-bit FNV-1a hash of its pixels (tests/test_batch4k.py compares them with the
// oracle's decode) and a summary line with the milliseconds of the timed passes.
//   batch_client <chunk> <objects> <ramp 0|1> <passes> a.jpg b.jpg ...
// Build: g++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include tests/cxx/batch_client.cpp -L libjpeg_amd -lmijpeg -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <string>
#include <vector>

#include "mijpeg.h"

static bool slurp(const char *path, std::string &out)
{
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
  fclose(f);
  return true;
}

int main(int argc, char **argv)
{
  if (argc < 6) return 2;
  const int chunk = atoi(argv[1]), objects = atoi(argv[2]), ramp = atoi(argv[3]), passes = atoi(argv[4]);
  std::vector<std::string> data((size_t)(argc - 5));
  std::vector<const uint8_t *> ptr;
  std::vector<size_t> len;
  for (int i = 5; i < argc; i++) {
    if (!slurp(argv[i], data[(size_t)(i - 5)])) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
    ptr.push_back((const uint8_t *)data[(size_t)(i - 5)].data());
    len.push_back(data[(size_t)(i - 5)].size());
  }
  const int n = (int)ptr.size();
  // the shape: from the first stream's header
  mijpeg_decoder *probe = nullptr;
  mijpeg_info f;
  if (mijpeg_create(&probe, -1) || mijpeg_set_input(probe, ptr[0], len[0]) || mijpeg_read_header(probe, &f)) { fprintf(stderr, "header\n"); return 3; }
  mijpeg_destroy(probe);
  const int64_t row = (int64_t)f.width * f.components * (f.sample_bytes > 1 ? f.sample_bytes : 1), frame = row * f.height;
  void *dev = nullptr, *host = nullptr;
  if (hipSetDevice(0) != hipSuccess || hipMalloc(&dev, (size_t)(frame * n)) != hipSuccess) { fprintf(stderr, "no device memory\n"); return 4; }
  host = mijpeg_host_alloc((size_t)(frame * n));
  if (!host) return 4;
  mijpeg_batch_pipeline *p = nullptr;
  int rc = mijpeg_batch_pipeline_create(&p, 0, chunk, objects, ramp);
  if (rc) { fprintf(stderr, "create %d\n", rc); return 5; }
  double best = 1e30;
  for (int pass = 0; pass < passes + 1; pass++) { // (the first pass allocates the objects' buffers: untimed)
    const auto t0 = std::chrono::steady_clock::now();
    rc = mijpeg_batch_pipeline_run(p, ptr.data(), len.data(), n, dev, frame, row, pass == passes ? host : nullptr);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc) {
      const char *m = nullptr;
      mijpeg_batch_pipeline_last_error(p, &m);
      fprintf(stderr, "run: %d %s\n", rc, m ? m : "");
      return 6;
    }
    if (pass > 0 && pass < passes && ms < best) best = ms;
  }
  int32_t chunks = 0, fallbacks = 0;
  mijpeg_batch_pipeline_stats(p, &chunks, &fallbacks, nullptr, 0);
  for (int i = 0; i < n; i++) {
    uint64_t h = 1469598103934665603ull;
    const uint8_t *px = (const uint8_t *)host + (int64_t)i * frame;
    for (int64_t k = 0; k < frame; k++) h = (h ^ px[k]) * 1099511628211ull;
    printf("frame %d %016llx\n", i, (unsigned long long)h);
  }
  printf("summary frames %d %dx%d chunks %d fallbacks %d best_ms %.3f\n", n, f.width, f.height, chunks, fallbacks, best < 1e29 ? best : -1.0);
  mijpeg_batch_pipeline_destroy(p);
  mijpeg_host_free(host);
  (void)hipFree(dev);
  return 0;
}
