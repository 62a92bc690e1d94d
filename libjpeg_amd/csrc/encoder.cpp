// encoder.cpp -- entropy coder and stream writer behind the encoder-direction kernel: quantised coefficient planes
// (host memory, the decoder's layout) -> baseline JPEG stream.
//
// What the reference's SequentialScan does when writing (codestream/sequentialscan.cpp:430-676: WriteMCU, EncodeBlock,
// Restart / Flush; marker/*.cpp for the segment syntax): one interleaved Huffman-sequential scan over all components,
// DC differences per component reset at every restart marker, AC run/size symbols with ZRL and EOB, byte stuffing,
// one-bits as padding in front of a marker (io/bitstream.hpp).  Restart intervals are independent, so they are coded
// in parallel into buffers of their own and concatenated with the RSTn markers in between.
// Huffman tables: the general purpose tables of ISO/IEC 10918-1 Annex K.3, or tables optimised for the picture
// (K.2: code lengths from the symbol statistics, limited to 16 bits).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <stdexcept>
#include <vector>

#include "encoder.hpp"
#include "host_decoder.hpp"

namespace mij {
namespace {

// Annex K.3 general purpose tables
const uint8_t K3_DC_L_COUNTS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t K3_DC_C_COUNTS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t K3_DC_VALUES[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t K3_AC_L_COUNTS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t K3_AC_L_VALUES[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
const uint8_t K3_AC_C_COUNTS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t K3_AC_C_VALUES[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};

void set_table(EncTable &t, const uint8_t counts[16], const uint8_t *values, int n)
{
  memcpy(t.counts, counts, 16);
  memset(t.values, 0, sizeof(t.values));
  memcpy(t.values, values, (size_t)n);
  t.derive();
}

// Annex K.2: optimal code lengths from frequencies, no code longer than 16 bits, no all-ones code
void optimal_table(EncTable &t, const uint32_t freq_in[256])
{
  long freq[257];
  int codesize[257], others[257];
  for (int i = 0; i < 256; i++) freq[i] = freq_in[i];
  freq[256] = 1; // reserves the all-ones code word
  memset(codesize, 0, sizeof(codesize));
  for (int i = 0; i < 257; i++) others[i] = -1;
  for (;;) {
    int c1 = -1, c2 = -1;
    long v = 1000000000L;
    for (int i = 0; i <= 256; i++)
      if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
    v = 1000000000L;
    for (int i = 0; i <= 256; i++)
      if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
    if (c2 < 0) break;
    freq[c1] += freq[c2];
    freq[c2] = 0;
    codesize[c1]++;
    while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
    others[c1] = c2;
    codesize[c2]++;
    while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
  }
  int bits[33];
  memset(bits, 0, sizeof(bits));
  for (int i = 0; i <= 256; i++)
    if (codesize[i]) bits[std::min(codesize[i], 32)]++;
  for (int i = 32; i > 16; i--)
    while (bits[i] > 0) {
      int j = i - 2;
      while (bits[j] == 0) j--;
      bits[i] -= 2;
      bits[i - 1]++;
      bits[j + 1] += 2;
      bits[j]--;
    }
  int i = 16;
  while (bits[i] == 0) i--;
  bits[i]--; // the reserved code point
  for (int l = 1; l <= 16; l++) t.counts[l - 1] = (uint8_t)bits[l];
  int k = 0;
  memset(t.values, 0, sizeof(t.values));
  for (int l = 1; l <= 32; l++)
    for (int s = 0; s < 256; s++)
      if (codesize[s] == l) t.values[k++] = (uint8_t)s;
  t.derive();
}

// Bit writer into a caller-provided buffer that is large enough for the worst case (see worst_case_bytes): 64-bit
// accumulator, four bytes at a time; a word that holds an 0xFF byte goes byte by byte through the stuffing.
struct BitWriter {
  uint8_t *p;           // next byte to write
  uint8_t *const begin;
  uint64_t acc = 0;
  int n = 0;
  const bool stuff;     // false: plain bits, for pieces that are merged (and stuffed) later
  uint64_t nbits = 0;   // bits written so far
  explicit BitWriter(uint8_t *buf, bool stuffing = true) : p(buf), begin(buf), stuff(stuffing) {}
  inline void byte(uint8_t b)
  {
    *p++ = b;
    if (b == 0xff && stuff) *p++ = 0; // byte stuffing
  }
  inline void put(unsigned bits, int len)
  {
    acc = (acc << len) | (bits & ((1u << len) - 1u));
    n += len;
    nbits += (uint64_t)len;
    if (n >= 32) {
      const uint32_t w = (uint32_t)(acc >> (n - 32));
      n -= 32;
      if (stuff && ((w & ~(w + 0x01010101u)) & 0x80808080u)) { // some byte is 0xFF
        byte((uint8_t)(w >> 24)); byte((uint8_t)(w >> 16)); byte((uint8_t)(w >> 8)); byte((uint8_t)w);
      } else {
        p[0] = (uint8_t)(w >> 24); p[1] = (uint8_t)(w >> 16); p[2] = (uint8_t)(w >> 8); p[3] = (uint8_t)w;
        p += 4;
      }
    }
  }
  void drain() { while (n >= 8) { byte((uint8_t)(acc >> (n - 8))); n -= 8; } }
  void flush() // one-bits up to the byte boundary
  {
    drain();
    if (n) { byte((uint8_t)(((acc << (8 - n)) | ((1u << (8 - n)) - 1u)) & 0xff)); n = 0; }
  }
  void flush_zero() // raw pieces: the rest of the last byte stays zero
  {
    drain();
    if (n) { byte((uint8_t)((acc << (8 - n)) & 0xff)); n = 0; }
  }
  size_t size() const { return (size_t)(p - begin); }
};

inline int category(int v)
{
  const unsigned a = (unsigned)(v < 0 ? -v : v);
  return a ? 32 - __builtin_clz(a) : 0;
}

// One block: sequentialscan.cpp EncodeBlock.  Either codes it (bw != null) or counts its symbols.
// Returns false when a value does not fit the 8-bit processes (DC differences of more than 11, AC coefficients of more
// than 10 bits: Tables F.1 / F.2), for which neither the Annex K tables nor SOF0 have room.
inline bool code_block(const int16_t *blk, int &pred, const EncTable &dc, const EncTable &ac, BitWriter *bw, uint32_t *dcfreq, uint32_t *acfreq,
                       const uint8_t *zz)
{
  const int diff = (blk ? blk[0] : pred) - pred;
  pred += diff;
  int s = category(diff);
  if (s > 11) return false;
  if (bw) {
    bw->put(dc.code[s], dc.len[s]);
    if (s) bw->put((unsigned)(diff < 0 ? diff - 1 : diff), s);
  } else dcfreq[s]++;
  int run = 0;
  if (blk)
    for (int k = 1; k < 64; k++) {
      const int v = blk[zz[k]];
      if (v == 0) { run++; continue; }
      while (run > 15) {
        if (bw) bw->put(ac.code[0xf0], ac.len[0xf0]);
        else acfreq[0xf0]++;
        run -= 16;
      }
      s = category(v);
      if (s > 10) return false;
      const int sym = (run << 4) | s;
      if (bw) {
        bw->put(ac.code[sym], ac.len[sym]);
        bw->put((unsigned)(v < 0 ? v - 1 : v), s);
      } else acfreq[sym]++;
      run = 0;
    }
  else run = 63;
  if (run > 0) {
    if (bw) bw->put(ac.code[0], ac.len[0]);
    else acfreq[0]++;
  }
  return true;
}

void put16(std::vector<uint8_t> &o, unsigned v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }

void write_dht(std::vector<uint8_t> &o, int tc, int th, const EncTable &t)
{
  o.push_back(0xff); o.push_back(0xc4);
  put16(o, (unsigned)(2 + 1 + 16 + t.nvalues));
  o.push_back((uint8_t)((tc << 4) | th));
  o.insert(o.end(), t.counts, t.counts + 16);
  o.insert(o.end(), t.values, t.values + t.nvalues);
}

} // namespace

void enc_standard_tables(EncTables &t)
{
  set_table(t.dc[0], K3_DC_L_COUNTS, K3_DC_VALUES, 12);
  set_table(t.dc[1], K3_DC_C_COUNTS, K3_DC_VALUES, 12);
  set_table(t.ac[0], K3_AC_L_COUNTS, K3_AC_L_VALUES, 162);
  set_table(t.ac[1], K3_AC_C_COUNTS, K3_AC_C_VALUES, 162);
}

void enc_optimal_tables(EncTables &t, const uint32_t dcfreq[2][256], const uint32_t acfreq[2][256], int ntables)
{
  for (int k = 0; k < ntables; k++) {
    optimal_table(t.dc[k], dcfreq[k]);
    optimal_table(t.ac[k], acfreq[k]);
  }
}

// SOI, DQT, SOF0, DHT, DRI, SOS: everything in front of the entropy coded data
void enc_write_headers(std::vector<uint8_t> &o, const mijpeg_info &f, const EncTables &tabs, int restart_interval)
{
  const int nc = f.components;
  const uint8_t *zz = scan_order();
  const EncTable *dct = tabs.dc, *act = tabs.ac;
  o.reserve(2048);
  o.push_back(0xff); o.push_back(0xd8);
  bool used[4] = {false, false, false, false};
  for (int c = 0; c < nc; c++) used[f.quant_index[c]] = true;
  for (int t = 0; t < 4; t++)
    if (used[t]) {
      bool wide = false;
      for (int i = 0; i < 64; i++) wide |= f.quant[t][i] > 255;
      o.push_back(0xff); o.push_back(0xdb);
      put16(o, (unsigned)(2 + 1 + (wide ? 128 : 64)));
      o.push_back((uint8_t)((wide ? 0x10 : 0) | t));
      for (int k = 0; k < 64; k++) {
        if (wide) o.push_back((uint8_t)(f.quant[t][zz[k]] >> 8));
        o.push_back((uint8_t)f.quant[t][zz[k]]);
      }
    }
  o.push_back(0xff); o.push_back(0xc0);
  put16(o, (unsigned)(8 + 3 * nc));
  o.push_back(8);
  put16(o, (unsigned)f.height);
  put16(o, (unsigned)f.width);
  o.push_back((uint8_t)nc);
  for (int c = 0; c < nc; c++) {
    o.push_back((uint8_t)(c + 1));
    o.push_back((uint8_t)((f.hsamp[c] << 4) | f.vsamp[c]));
    o.push_back((uint8_t)f.quant_index[c]);
  }
  for (int t = 0; t < (nc > 1 ? 2 : 1); t++) {
    write_dht(o, 0, t, dct[t]);
    write_dht(o, 1, t, act[t]);
  }
  if (restart_interval) {
    o.push_back(0xff); o.push_back(0xdd);
    put16(o, 4);
    put16(o, (unsigned)restart_interval);
  }
  o.push_back(0xff); o.push_back(0xda);
  put16(o, (unsigned)(6 + 2 * nc));
  o.push_back((uint8_t)nc);
  for (int c = 0; c < nc; c++) {
    o.push_back((uint8_t)(c + 1));
    o.push_back((uint8_t)(c ? 0x11 : 0x00));
  }
  o.push_back(0); o.push_back(63); o.push_back(0);
}
} // namespace mij

using namespace mij;

extern "C" void mijpeg_free(void *p) { free(p); }

extern "C" int mijpeg_encode_coefficients(const mijpeg_info *info, const int16_t *coef, int restart_interval, int optimize, int threads,
                                          uint8_t **stream, size_t *size)
try {
  if (!info || !coef || !stream || !size || restart_interval < 0 || restart_interval > 65535) return MIJPEG_ERR_INVALID_PARAMETER;
  const mijpeg_info &f = *info;
  if (f.precision != 8 || (f.components != 1 && f.components != 3)) return MIJPEG_ERR_OPERATION_UNIMPLEMENTED;
  const int nc = f.components;
  const uint8_t *zz = scan_order();
  const int64_t total_mcus = (int64_t)f.mcus_x * f.mcus_y;
  const int64_t ri = restart_interval ? restart_interval : total_mcus;
  const int64_t nint = (total_mcus + ri - 1) / ri;
  if (threads <= 0) threads = default_threads();
  // the blocks that cover samples; the MCU padding blocks are coded as "same DC, no AC"
  int nbx[4], nby[4], hs[4], vs[4];
  for (int c = 0; c < nc; c++) {
    nbx[c] = ((f.width + f.subx[c] - 1) / f.subx[c] + 7) >> 3;
    nby[c] = ((f.height + f.suby[c] - 1) / f.suby[c] + 7) >> 3;
    hs[c] = nc > 1 ? f.hsamp[c] : 1;
    vs[c] = nc > 1 ? f.vsamp[c] : 1;
  }
  std::atomic<bool> out_of_range{false};
  auto walk_mcus = [&](int64_t m0, int64_t m1, int (&pred)[4], const EncTable *dct, const EncTable *act, BitWriter *bw, uint32_t (*dcf)[256],
                       uint32_t (*acf)[256]) {
    for (int64_t m = m0; m < m1; m++) {
      const int my = (int)(m / f.mcus_x), mx = (int)(m - (int64_t)my * f.mcus_x);
      for (int c = 0; c < nc; c++) {
        const int t = c ? 1 : 0;
        for (int by = 0; by < vs[c]; by++)
          for (int bx = 0; bx < hs[c]; bx++) {
            const int gx = mx * hs[c] + bx, gy = my * vs[c] + by;
            const int16_t *blk = (gx < nbx[c] && gy < nby[c]) ? coef + f.coef_offset[c] + ((int64_t)gy * f.blocks_w[c] + gx) * 64 : nullptr;
            if (!code_block(blk, pred[c], dct[t], act[t], bw, dcf ? dcf[t] : nullptr, acf ? acf[t] : nullptr, zz)) out_of_range.store(true, std::memory_order_relaxed);
          }
      }
    }
  };
  int blocks_per_mcu = 0;
  for (int c = 0; c < nc; c++) blocks_per_mcu += hs[c] * vs[c];
  // bytes `mcus` MCUs can take at most: 63 AC coefficients of 16 + 10 bits and a DC difference of 16 + 11 bits per block,
  // every byte stuffed, plus slack for the word-wise writer
  auto worst_case_bytes = [&](int64_t mcus) -> size_t { return (size_t)mcus * (size_t)blocks_per_mcu * 420 + 64; };
  auto walk_interval = [&](int64_t i, const EncTable *dct, const EncTable *act, BitWriter *bw, uint32_t (*dcf)[256], uint32_t (*acf)[256]) {
    int pred[4] = {0, 0, 0, 0};
    walk_mcus(i * ri, std::min(total_mcus, i * ri + ri), pred, dct, act, bw, dcf, acf);
  };
  // DC predictor of component c in front of MCU m of an interval that starts at MCU m_first: the DC of the last block of c
  // that covers samples and is coded before m (padding blocks repeat the predictor)
  auto pred_before = [&](int64_t m, int64_t m_first, int c) -> int {
    for (int64_t k = m - 1; k >= m_first; k--) {
      const int my = (int)(k / f.mcus_x), mx = (int)(k - (int64_t)my * f.mcus_x);
      for (int by = vs[c] - 1; by >= 0; by--)
        for (int bx = hs[c] - 1; bx >= 0; bx--) {
          const int gx = mx * hs[c] + bx, gy = my * vs[c] + by;
          if (gx < nbx[c] && gy < nby[c]) return coef[f.coef_offset[c] + ((int64_t)gy * f.blocks_w[c] + gx) * 64];
        }
    }
    return 0;
  };
  // A long interval on several threads: pieces of whole MCUs are coded as plain bit strings from their predictors, put
  // together at their bit offsets (bytes that belong to one piece in parallel, the few bytes shared by two pieces
  // afterwards), padded with one-bits and byte-stuffed.
  auto code_interval_in_pieces = [&](int64_t i, int pieces, const EncTable *dct, const EncTable *act, std::vector<uint8_t> &out) {
    const int64_t m_first = i * ri, m_last = std::min(total_mcus, m_first + ri), count = m_last - m_first;
    pieces = (int)std::max<int64_t>(1, std::min<int64_t>(pieces, count / 64));
    std::vector<std::vector<uint8_t>> raw((size_t)pieces);
    std::vector<uint64_t> bits((size_t)pieces + 1, 0);
    parallel_for(std::min(pieces, threads), [&](int w) {
      for (int p = w; p < pieces; p += std::min(pieces, threads)) {
        const int64_t a0 = m_first + count * p / pieces, a1 = m_first + count * (p + 1) / pieces;
        int pred[4] = {0, 0, 0, 0};
        for (int c = 0; c < nc; c++) pred[c] = pred_before(a0, m_first, c);
        std::unique_ptr<uint8_t[]> scratch(new uint8_t[worst_case_bytes(a1 - a0)]); // not zeroed: only what is written gets touched
        BitWriter bw(scratch.get(), false);
        walk_mcus(a0, a1, pred, dct, act, &bw, nullptr, nullptr);
        bw.flush_zero();
        bits[(size_t)p + 1] = bw.nbits;
        raw[(size_t)p].assign(scratch.get(), scratch.get() + bw.size());
        raw[(size_t)p].push_back(0); // one byte of slack for the shifted reads below
        raw[(size_t)p].push_back(0);
      }
    });
    for (int p = 0; p < pieces; p++) bits[(size_t)p + 1] += bits[(size_t)p];
    const uint64_t total = bits[(size_t)pieces];
    std::vector<uint8_t> plain((size_t)((total + 7) >> 3), 0);
    parallel_for(std::min(pieces, threads), [&](int w) {
      for (int p = w; p < pieces; p += std::min(pieces, threads)) {
        const uint64_t b0 = bits[(size_t)p], b1 = bits[(size_t)p + 1];
        const uint8_t *src = raw[(size_t)p].data();
        const uint64_t j0 = (b0 + 7) >> 3, j1 = b1 >> 3; // output bytes made of this piece's bits only
        if (j1 <= j0) continue;
        const uint64_t q = 8 * j0 - b0; // piece-local bit position of output byte j0: 0..7
        const unsigned sh = (unsigned)(q & 7);
        const uint8_t *s0 = src + (q >> 3);
        if (sh == 0) memcpy(plain.data() + j0, s0, (size_t)(j1 - j0));
        else
          for (uint64_t j = j0; j < j1; j++, s0++) plain[(size_t)j] = (uint8_t)((s0[0] << sh) | (s0[1] >> (8 - sh)));
      }
    });
    auto bit_at = [&](uint64_t b) -> unsigned { // bit b of the merged string
      const size_t p = (size_t)(std::upper_bound(bits.begin(), bits.end(), b) - bits.begin()) - 1;
      const uint64_t l = b - bits[p];
      return (raw[p][(size_t)(l >> 3)] >> (7 - (l & 7))) & 1u;
    };
    for (int p = 0; p <= pieces; p++) { // the bytes around the piece boundaries (and the last, partial one)
      const uint64_t b = bits[(size_t)p];
      for (uint64_t j = (b >> 3); j <= (b >> 3) + 1 && j < plain.size(); j++) {
        if (p > 0 && p < pieces && (b & 7) == 0 && j != (b >> 3)) continue;
        unsigned v = 0;
        for (int k = 0; k < 8; k++) {
          const uint64_t gb = 8 * j + (uint64_t)k;
          v = (v << 1) | (gb < total ? bit_at(gb) : 1u); // one-bits pad the last byte
        }
        plain[(size_t)j] = (uint8_t)v;
      }
    }
    out.clear();
    out.reserve(plain.size() + plain.size() / 64 + 16);
    const uint8_t *q = plain.data(), *end = q + plain.size();
    while (q < end) { // byte stuffing: a zero byte behind every 0xFF
      const uint8_t *ff = (const uint8_t *)memchr(q, 0xff, (size_t)(end - q));
      if (!ff) { out.insert(out.end(), q, end); break; }
      out.insert(out.end(), q, ff + 1);
      out.push_back(0);
      q = ff + 1;
    }
  };
  EncTable dct[2], act[2];
  if (optimize) {
    const int parts = (int)std::min<int64_t>(threads, nint);
    std::vector<std::vector<uint32_t>> stats((size_t)parts, std::vector<uint32_t>(4 * 256, 0));
    parallel_for(parts, [&](int p) {
      uint32_t(*dcf)[256] = reinterpret_cast<uint32_t(*)[256]>(stats[(size_t)p].data());
      uint32_t(*acf)[256] = dcf + 2;
      for (int64_t i = nint * p / parts; i < nint * (p + 1) / parts; i++) walk_interval(i, dct, act, nullptr, dcf, acf);
    });
    uint32_t sum[4][256];
    memset(sum, 0, sizeof(sum));
    for (auto &s : stats)
      for (int k = 0; k < 4 * 256; k++) (&sum[0][0])[k] += s[(size_t)k];
    EncTables opt;
    enc_optimal_tables(opt, sum, sum + 2, nc > 1 ? 2 : 1);
    for (int t = 0; t < 2; t++) { dct[t] = opt.dc[t]; act[t] = opt.ac[t]; }
  } else {
    EncTables std_tabs;
    enc_standard_tables(std_tabs);
    for (int t = 0; t < 2; t++) { dct[t] = std_tabs.dc[t]; act[t] = std_tabs.ac[t]; }
  }
  // entropy coded segments, one buffer per restart interval
  // the entropy coded data: `parts` buffers that follow each other in the stream (RSTn markers included)
  std::vector<std::unique_ptr<uint8_t[]>> part_mem;
  std::vector<std::vector<uint8_t>> part_vec;
  std::vector<const uint8_t *> part_ptr;
  std::vector<size_t> part_len;
  if (nint < threads && threads > 1 && ri >= 256) { // few, long intervals: parallelism inside them
    const int pieces = (int)std::min<int64_t>(4096, ((int64_t)threads * 4 + nint - 1) / nint);
    part_vec.resize((size_t)nint);
    for (int64_t i = 0; i < nint; i++) {
      code_interval_in_pieces(i, pieces, dct, act, part_vec[(size_t)i]);
      if (i + 1 < nint) { part_vec[(size_t)i].push_back(0xff); part_vec[(size_t)i].push_back((uint8_t)(0xd0 + (i & 7))); }
      part_ptr.push_back(part_vec[(size_t)i].data());
      part_len.push_back(part_vec[(size_t)i].size());
    }
  } else { // every worker codes a run of whole intervals, one behind the other, into one buffer
    const int workers = (int)std::max<int64_t>(1, std::min<int64_t>(threads, nint));
    part_mem.resize((size_t)workers);
    part_ptr.resize((size_t)workers);
    part_len.resize((size_t)workers);
    parallel_for(workers, [&](int w) {
      const int64_t i0 = nint * w / workers, i1 = nint * (w + 1) / workers;
      const int64_t mcus = std::min(total_mcus, i1 * ri) - i0 * ri;
      part_mem[(size_t)w].reset(new uint8_t[worst_case_bytes(mcus) + (size_t)(i1 - i0) * 4]); // not zeroed: only what is written gets touched
      uint8_t *q = part_mem[(size_t)w].get();
      for (int64_t i = i0; i < i1; i++) {
        BitWriter bw(q);
        walk_interval(i, dct, act, &bw, nullptr, nullptr);
        bw.flush();
        q += bw.size();
        if (i + 1 < nint) { *q++ = 0xff; *q++ = (uint8_t)(0xd0 + (i & 7)); }
      }
      part_ptr[(size_t)w] = part_mem[(size_t)w].get();
      part_len[(size_t)w] = (size_t)(q - part_mem[(size_t)w].get());
    });
  }
  if (out_of_range.load()) return MIJPEG_ERR_OVERFLOW_PARAMETER; // coefficients outside what an 8-bit frame can hold
  // the stream
  std::vector<uint8_t> o;
  EncTables tabs;
  for (int t = 0; t < 2; t++) { tabs.dc[t] = dct[t]; tabs.ac[t] = act[t]; }
  enc_write_headers(o, f, tabs, restart_interval);
  // header so far, then the parts at their offsets (copied in parallel), EOI
  const size_t nparts = part_ptr.size();
  std::vector<size_t> at(nparts + 1);
  at[0] = o.size();
  for (size_t k = 0; k < nparts; k++) at[k + 1] = at[k] + part_len[k];
  const size_t total_size = at[nparts] + 2;
  uint8_t *p = (uint8_t *)malloc(total_size);
  if (!p) return MIJPEG_ERR_OUT_OF_MEMORY;
  memcpy(p, o.data(), o.size());
  parallel_for((int)std::min<size_t>((size_t)threads, nparts), [&](int w) {
    const size_t workers = std::min<size_t>((size_t)threads, nparts);
    for (size_t k = (size_t)w; k < nparts; k += workers) memcpy(p + at[k], part_ptr[k], part_len[k]);
  });
  p[total_size - 2] = 0xff;
  p[total_size - 1] = 0xd9;
  *stream = p;
  *size = total_size;
  return MIJPEG_OK;
} catch (const std::bad_alloc &) {
  return MIJPEG_ERR_OUT_OF_MEMORY; // (nothing crosses the C boundary as an exception)
} catch (...) {
  return MIJPEG_ERR_PHASE_ERROR;
}
