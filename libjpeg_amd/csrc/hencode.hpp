// hencode.hpp -- argument blocks of the on-device entropy coder (hencode.hip); internal to libmijpeg.so.
#ifndef MIJ_HENCODE_HPP
#define MIJ_HENCODE_HPP
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace mij {

// code word and length per symbol of the four tables of a scan (DC/AC for the first component, DC/AC for the others)
struct HencTables {
  uint16_t dc_code[2][16];
  uint16_t ac_code[2][256];
  uint8_t dc_len[2][16];
  uint8_t ac_len[2][256];
};
static_assert(sizeof(HencTables) % 16 == 0, "copied to LDS in dwords");

constexpr int HENC_STUFF_CHUNK = 64; // bytes of the plain stream one lane of the stuffing kernels handles

struct HencArgs {
  const int16_t *coef;          // coefficient planes of the frame (decoder layout)
  const HencTables *tables;     // device
  int32_t ncomp, mcus_x, total_mcus, ri, blocks_per_mcu; // ri: MCUs per restart interval (= total_mcus without restart markers)
  int32_t hs[4], vs[4], bw[4], nbx[4], nby[4];
  int64_t coef_off[4];
  uint8_t blk_comp[64], blk_bx[64], blk_by[64]; // block j of an MCU: component, position inside the MCU
  uint32_t total_blocks, n_intervals;
  uint32_t *bits;               // per block in scan order: length of its code in bits
  const uint64_t *bitpos;       // exclusive prefix sums of bits (total_blocks + 1 entries)
  uint32_t *ibytes;             // per interval: bytes of its entropy coded segment before stuffing
  const uint64_t *istart;       // exclusive prefix sums of ibytes (n_intervals + 1 entries): byte offsets in the plain stream
  uint32_t *plain;              // plain (unstuffed) stream as big-endian 32-bit words, zeroed
  uint64_t plain_bytes;
  uint32_t *ffcount;            // per HENC_STUFF_CHUNK bytes of the plain stream: 0xFF bytes in them
  const uint64_t *ffstart;      // exclusive prefix sums of ffcount (chunks + 1 entries)
  uint8_t *out;                 // entropy coded data with stuffing and RSTn markers
  uint32_t *hist;               // optional statistics: [2][256] DC symbol counts, [2][256] AC symbol counts
};

int henc_count(const HencArgs &a, bool statistics, hipStream_t stream);      // bits[] (and hist[])
int henc_interval_bytes(const HencArgs &a, hipStream_t stream);              // ibytes[] from bitpos[]
int henc_emit(const HencArgs &a, hipStream_t stream);                        // plain[]
int henc_count_ff(const HencArgs &a, hipStream_t stream);                    // ffcount[]
int henc_stuff(const HencArgs &a, hipStream_t stream);                       // out[]
// out[i] = sum of in[0..i) for i = 0..n (n + 1 entries); scratch: at least (n / 1024 + 2) * 2 uint64
int exclusive_scan_u32(const uint32_t *in, uint64_t *out, uint32_t n, uint64_t *scratch, hipStream_t stream);

} // namespace mij
#endif
