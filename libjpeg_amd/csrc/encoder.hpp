// encoder.hpp -- Huffman code tables and header writer shared by the host entropy coder (encoder.cpp) and the host side of
// the device entropy coder (capi.cpp / hencode.hip); internal to libmijpeg.so.
#ifndef MIJ_ENCODER_HPP
#define MIJ_ENCODER_HPP
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/mijpeg.h"

namespace mij {

// One Huffman table in encoder form: the DHT description (counts per length, symbols in code order) and, derived from
// it as Annex C prescribes, code word and length per symbol (length 0: the symbol has no code).
struct EncTable {
  uint8_t counts[16];
  uint8_t values[256];
  int nvalues;
  uint16_t code[256];
  uint8_t len[256];
  void derive()
  {
    memset(code, 0, sizeof(code));
    memset(len, 0, sizeof(len));
    unsigned c = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
      for (int i = 0; i < counts[l - 1]; i++, k++) {
        code[values[k]] = (uint16_t)c++;
        len[values[k]] = (uint8_t)l;
      }
      c <<= 1;
    }
    nvalues = k;
  }
};


struct EncTables {
  EncTable dc[2], ac[2]; // [0]: first component, [1]: the others
};

void enc_standard_tables(EncTables &t);                                                                   // Annex K.3
void enc_optimal_tables(EncTables &t, const uint32_t dcfreq[2][256], const uint32_t acfreq[2][256], int ntables); // Annex K.2
void enc_write_headers(std::vector<uint8_t> &out, const mijpeg_info &f, const EncTables &t, int restart_interval);

} // namespace mij
#endif
