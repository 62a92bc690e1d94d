// huffman_dev.hpp -- argument blocks of the on-device entropy decoder (huffman.hip).
#ifndef MIJ_HUFFMAN_DEV_HPP
#define MIJ_HUFFMAN_DEV_HPP
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace mij {

constexpr int HUFF_DEV_LOOKAHEAD = 10;
constexpr int HUFF_ERR_MALFORMED = 1, HUFF_ERR_OVERFLOW = 2;
constexpr int HUFF_DEV_INVALID = 0x8000; // direct-table flag: AC symbol that does not exist in sequential scans
constexpr int HUFF_STREAM_PAD = 256; // bytes the device copy of the stream is padded with (the readers prefetch ahead)

// One Huffman table in device form: direct lookup for codes up to 10 bits ((length << 8) | symbol [| HUFF_DEV_INVALID
// in AC tables], 0 = longer code),
// canonical max-code / value-offset arrays for the rest (same layout as HuffTable in host_decoder.hpp).
struct HuffDevTable {
  uint16_t fast[1 << HUFF_DEV_LOOKAHEAD];
  int32_t maxcode[18];
  int32_t valoff[17];
  uint8_t values[256];
  uint8_t pad[4]; // size multiple of 16 (2448 bytes)
};
static_assert(sizeof(HuffDevTable) % 16 == 0, "tables are copied to LDS in dwords");

// Follows the tables in device memory (and in LDS).
struct HuffDevAux {
  // per scan component k and scan position i (padded with position 63 up to 63 + 15, where a corrupt run may point):
  // (delta << 16) | (2 * natural position): the byte offset into a coefficient block and the range-check weight
  uint32_t zq[4][80];
};
static_assert(sizeof(HuffDevAux) % 16 == 0, "copied in dwords");

struct HuffScanArgs {
  const uint8_t *data;           // device copy of the codestream
  const uint32_t *ibegin, *iend; // device: byte range of every restart interval
  int32_t n_intervals, restart_interval, total_mcus, mcus_x;
  int32_t ncomp;                 // components in the scan
  int32_t comp_of[4];            // frame component of scan component k
  int32_t hs[4], vs[4], bw[4];   // blocks per MCU and plane width in blocks, per scan component
  int64_t coef_off[4];
  int32_t dc_tab[4], ac_tab[4];  // indices into tables[]
  int32_t ntables;
  int32_t debug;                 // experiments only (MIJPEG_HUFF_DEBUG)
  int32_t lanes;                 // active lanes per wave (power of two, 1..64): fewer lanes = more waves, less divergence
  const HuffDevTable *tables;    // device: ntables tables followed by one HuffDevAux
  int16_t *coef;                 // frame base of the coefficient store (zeroed beforehand)
  uint32_t *status;              // device: [0] error (0 = ok), [1 + c] max over blocks of sum |c| q for frame component c
};

int launch_huffman_scan(const HuffScanArgs &a, hipStream_t stream);

} // namespace mij
#endif
