// fuzz_encoder.cpp -- ASan/UBSan harness of the host entropy coder (tools only, not part of the library or the tests):
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 -pthread -Ilibjpeg_amd/csrc tools/fuzz_encoder.cpp libjpeg_amd/csrc/encoder.cpp libjpeg_amd/csrc/host_decoder.cpp -o /tmp/enc_fuzz && MIJPEG_THREADS=8 /tmp/enc_fuzz
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "host_decoder.hpp"
using namespace mij;
extern "C" int mijpeg_encode_coefficients(const mijpeg_info *, const int16_t *, int, int, int, uint8_t **, size_t *);
extern "C" void mijpeg_free(void *);
static uint64_t S = 88172645463325252ull;
static uint32_t rnd() { S ^= S << 13; S ^= S >> 7; S ^= S << 17; return (uint32_t)(S >> 11); }
int main()
{
  int fails = 0;
  for (int it = 0; it < 300; it++) {
    mijpeg_info f; memset(&f, 0, sizeof(f));
    f.width = 1 + rnd() % 700; f.height = 1 + rnd() % 500; f.components = (rnd() % 4) ? 3 : 1; f.precision = 8;
    const int lay = rnd() % 5;
    const int hs[5][3] = {{1,1,1},{2,1,1},{2,1,1},{4,1,1},{2,1,2}}, vs[5][3] = {{1,1,1},{2,1,1},{1,1,1},{2,1,1},{2,2,1}};
    for (int c = 0; c < f.components; c++) { f.hsamp[c] = f.components == 3 ? hs[lay][c] : 1; f.vsamp[c] = f.components == 3 ? vs[lay][c] : 1; f.quant_index[c] = c ? 1 : 0; }
    for (int t = 0; t < 2; t++) for (int i = 0; i < 64; i++) f.quant[t][i] = 1 + rnd() % 255;
    int hmax = 1, vmax = 1;
    for (int c = 0; c < f.components; c++) { hmax = std::max(hmax, f.hsamp[c]); vmax = std::max(vmax, f.vsamp[c]); }
    f.mcus_x = (f.width + 8 * hmax - 1) / (8 * hmax); f.mcus_y = (f.height + 8 * vmax - 1) / (8 * vmax);
    int64_t off = 0;
    for (int c = 0; c < f.components; c++) { f.subx[c] = hmax / f.hsamp[c]; f.suby[c] = vmax / f.vsamp[c]; f.blocks_w[c] = f.mcus_x * f.hsamp[c]; f.blocks_h[c] = f.mcus_y * f.vsamp[c]; f.coef_offset[c] = off; off += (int64_t)f.blocks_w[c] * f.blocks_h[c] * 64; }
    f.coef_count = off;
    std::vector<int16_t> coef((size_t)off);
    const int style = rnd() % 4;
    for (auto &v : coef) {
      const uint32_t r = rnd();
      if (style == 0) v = (r % 7 == 0) ? (int16_t)((int)(r >> 8) % 2047 - 1023) : 0;          // sparse, 10-bit AC range
      else if (style == 1) v = (int16_t)((int)(r >> 8) % 63 - 31);                           // dense small
      else if (style == 2) v = (r % 50 == 0) ? (int16_t)((int)(r >> 8) % 2047 - 1023) : 0;   // long zero runs (ZRL)
      else v = 0;
    }
    // DC values: keep the differences inside 11 bits
    for (int c = 0; c < f.components; c++)
      for (int64_t b = 0; b < (int64_t)f.blocks_w[c] * f.blocks_h[c]; b++) coef[(size_t)(f.coef_offset[c] + b * 64)] = (int16_t)((int)(rnd() % 1023) - 511);
    const int ri = (rnd() % 3 == 0) ? 0 : 1 + rnd() % 700;
    const int opt = rnd() & 1;
    uint8_t *ref = nullptr; size_t refn = 0;
    if (mijpeg_encode_coefficients(&f, coef.data(), ri, opt, 1, &ref, &refn)) { printf("encode failed it=%d\n", it); return 1; }
    for (int th : {2, 5, 16}) {
      uint8_t *s = nullptr; size_t n = 0;
      if (mijpeg_encode_coefficients(&f, coef.data(), ri, opt, th, &s, &n) || n != refn || memcmp(s, ref, n)) { printf("threads %d differ it=%d\n", th, it); fails++; }
      mijpeg_free(s);
    }
    HostDecoder d;
    if (d.parse(ref, refn, false)) { printf("parse failed it=%d: %s\n", it, d.error.message.c_str()); fails++; mijpeg_free(ref); continue; }
    std::vector<int16_t> back((size_t)d.info.coef_count);
    if (d.decode(back.data(), 4, nullptr)) { printf("decode failed it=%d: %s\n", it, d.error.message.c_str()); fails++; mijpeg_free(ref); continue; }
    for (int c = 0; c < f.components; c++) {
      const int nbx = ((f.width + f.subx[c] - 1) / f.subx[c] + 7) >> 3, nby = ((f.height + f.suby[c] - 1) / f.suby[c] + 7) >> 3;
      for (int by = 0; by < nby; by++)
        for (int bx = 0; bx < nbx; bx++)
          if (memcmp(&back[(size_t)(d.info.coef_offset[c] + ((int64_t)by * f.blocks_w[c] + bx) * 64)], &coef[(size_t)(f.coef_offset[c] + ((int64_t)by * f.blocks_w[c] + bx) * 64)], 128)) { printf("mismatch it=%d c=%d\n", it, c); fails++; by = nby; break; }
    }
    mijpeg_free(ref);
  }
  printf("done, %d failures\n", fails);
  return fails != 0;
}
