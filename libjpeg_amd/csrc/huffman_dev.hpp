// huffman_dev.hpp -- argument blocks of the on-device entropy decoder (huffman.hip).
#ifndef MIJ_HUFFMAN_DEV_HPP
#define MIJ_HUFFMAN_DEV_HPP
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace mij {

constexpr int HUFF_DEV_LOOKAHEAD = 10;
constexpr int HUFF_ERR_MALFORMED = 1, HUFF_ERR_OVERFLOW = 2;
constexpr int HUFF_ERR_DESYNC = 3; // a virtual restart interval did not end where the next one begins: the stream is damaged
constexpr int HUFF_DEV_INVALID = 0x8000; // direct-table flag: AC symbol that does not exist in sequential scans
constexpr int HUFF_DEV_SUB = 0x4000;     // direct-table flag: a longer code; the low bits name the second-level table of this prefix
constexpr int HUFF_DEV_SUBTABLES = 8;    // second-level tables per Huffman table (Annex K tables need five)
constexpr int HUFF_STREAM_PAD = 256; // bytes the device copy of the stream is padded with (the readers prefetch ahead)

// One Huffman table in device form: direct lookup for codes up to 10 bits, entry = (tot << 8) | symbol with tot = code length
// + value bits behind the code -- what the reader moves on by: 31 at most -- or HUFF_DEV_INVALID: an AC symbol that does not
// exist in sequential scans, a DC symbol beyond 15.  Longer codes: HUFF_DEV_SUB | t sends the lookup to second-level table t,
// indexed by the next six bits (same entry format) -- canonical codes longer than ten bits share very few ten-bit prefixes
// (all ones but the tail: five in the Annex K tables), and with 64 lanes per wave SOME lane meets a long code in every other
// symbol step, so what the wave pays for them is what it pays per step; t = HUFF_DEV_NO_SUB (and ten-bit prefixes of no code
// at all): canonical max-code / value-offset walk (same layout as HuffTable in host_decoder.hpp).
constexpr int HUFF_DEV_NO_SUB = 15;
// entry of a code of `length` bits for `symbol` (host: table upload; device: the canonical walk)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t huff_dev_entry(int length, uint32_t symbol, int ac)
{
  // ac: 0 = DC table, 1 = AC table of a sequential scan, 2 = AC table of a progressive / refinement scan (huffman_prog_kernel):
  // every run / size pair is legal there and the bits behind a code are the value bits of a coefficient (s of them), the low
  // bits of an EOB run (r of them for EOBr, r < 15; sequentialscan.cpp:722-750) or the one sign bit of a refinement scan's
  // new coefficient (s = 1; refinementscan.cpp:640-660): 16 + 15 = 31 bits at most, like everywhere
  if (ac == 2) {
    const uint32_t r = (symbol >> 4) & 15u, sz = symbol & 15u;
    const uint32_t extra = sz ? sz : (r < 15u ? r : 0u);
    return ((uint32_t)(length + (int)extra) << 8) | (symbol & 0xffu);
  }
  const uint32_t s = ac ? (symbol & 15u) : symbol;
  if (s > 15u) return (uint32_t)HUFF_DEV_INVALID;
  uint32_t e = ((uint32_t)(length + (int)s) << 8) | symbol;
  if (ac && s == 0 && symbol != 0 && symbol != 0xf0) e |= (uint32_t)HUFF_DEV_INVALID;
  return e;
}
struct HuffDevTable {
  uint16_t fast[1 << HUFF_DEV_LOOKAHEAD];
  uint16_t sub[HUFF_DEV_SUBTABLES][64];
  int32_t maxcode[18];
  int32_t valoff[17];
  uint8_t values[256];
  uint8_t pad[4]; // size multiple of 16 (3472 bytes)
};
static_assert(sizeof(HuffDevTable) % 16 == 0, "tables are copied to LDS in dwords");

// Follows the tables in device memory (and in LDS).
struct HuffDevAux {
  // per scan component k and scan position i (padded with position 63 up to 63 + 15, where a corrupt run may point):
  // (delta << 16) | (2 * natural position): the byte offset into a coefficient block and the range-check weight
  uint32_t zq[4][80];
};
static_assert(sizeof(HuffDevAux) % 16 == 0, "copied in dwords");

// One image of a batch (device memory).  All images of a launch share the frame geometry (components, sampling, plane
// sizes); stream bytes, restart interval, Huffman tables and deltas are per image.
struct HuffImage {
  uint32_t stream_off;       // byte offset of the image's codestream in `data` (16-byte aligned)
  uint32_t first_interval;   // its first entry in ibegin / iend (whose offsets are relative to stream_off)
  int32_t n_intervals, restart_interval, total_mcus, mcus_x;
  int64_t coef_base;         // int16 index of the image's coefficient store in `coef`
  uint32_t table_off;        // byte offset (from `tables`) of its ntables HuffDevTable + one HuffDevAux
  uint32_t status_off;       // dword index of its 8-dword status block in `status`
  uint32_t virt;             // 1: the intervals are virtual (no markers): start `iskip` bits into the byte, predictors from `ipred`
  uint32_t reserved;
};

// A workgroup works on ONE image (the tables in its LDS are that image's): first interval of the group inside the image.
struct HuffGroup {
  uint32_t image, first_interval;
};

struct HuffScanArgs {
  const uint8_t *data;           // device: codestreams of all images, each padded by HUFF_STREAM_PAD
  const uint32_t *ibegin, *iend; // device: byte range of every restart interval, relative to the image's stream_off
  const uint8_t *iskip;          // device, virtual intervals: bits to drop in front of the first block
  const int16_t *ipred;          // device, virtual intervals: DC predictors, four per interval
  const HuffImage *images;       // device
  const HuffGroup *groups;       // device: one per workgroup
  int32_t n_groups;
  int32_t ncomp;                 // components in the scan
  int32_t comp_of[4];            // frame component of scan component k
  int32_t hs[4], vs[4], bw[4];   // blocks per MCU and plane width in blocks, per scan component
  int64_t coef_off[4];           // plane offsets inside an image's coefficient store
  int32_t dc_tab[4], ac_tab[4];  // indices into the image's tables
  int32_t ntables;
  int32_t reserved;
  int32_t lanes;                 // active lanes per wave (power of two, 1..64): fewer lanes = more waves, less divergence
  int32_t waves_per_group;
  const uint8_t *tables;         // device: per image ntables tables followed by one HuffDevAux
  int16_t *coef;                 // coefficient stores
  uint32_t *status;              // device: per image [0] error (0 = ok), [1 + c] max over blocks of sum |c| q for frame component c
};

constexpr int HUFF_WALK_SUMS_BYTES = 40;
constexpr int HUFF_WALK_TILE = 1024;
// Self-synchronising walk of streams without restart markers (huffman_walk_kernel, see huffman.hip).
struct HuffWalkArgs {
  const uint8_t *data;           // device: the streams (as for HuffScanArgs)
  const HuffImage *images;       // device: stream_off, table_off of every image (n_intervals etc. unused here)
  const uint8_t *tables;
  const uint32_t *sub_image;     // device, per workgroup: image of the workgroup's subsequences ...
  const uint32_t *sub_first;     // ... and index of its first subsequence inside the image
  const uint32_t *img_sub0;      // device, per image: index of its first subsequence in the state arrays
  const uint32_t *img_nsub;      // per image: number of subsequences
  const uint32_t *img_e0, *img_e1; // per image: entropy coded segment [e0, e1) relative to stream_off
  // start state of every subsequence (byte | bits to skip << 32 | block inside the MCU << 40), updated in place: a
  // lane writes its successor's, stamps it with the round, and sets changed[round]
  uint64_t *state;
  uint32_t *stamp;               // round that last wrote the state (0 = the initial guess)
  uint32_t round;                // 1, 2, ...
  uint32_t *changed;             // one flag per round
  uint32_t *nblocks;             // out: blocks that start inside the subsequence
  int32_t *dcsum;                // out: four per subsequence
  uint32_t *walk_status;         // per image, from the prefix sums: 1 phase mismatch | 2 DC out of range | 4 too few blocks
  struct WalkSums *tile_sums;    // scratch of the prefix sums: tiles_per_image entries of HUFF_WALK_SUMS_BYTES per image
  int32_t tiles_per_image;       // ceil(max subsequences per image / 1024), at most 1024
  // EMIT only
  uint32_t *first_block;         // per subsequence: number of its first block in the image (huffman_walk_scan_kernel)
  int32_t *first_pred;           // four per subsequence
  const uint32_t *img_int0;      // per image: index of its first virtual interval
  uint32_t *ibegin;
  uint8_t *iskip;
  int16_t *ipred;
  uint32_t emit_every;           // blocks per virtual interval
  uint32_t total_blocks;         // per image (the images of a launch share their geometry)
  int32_t ncomp, nblk_mcu;       // scan components, blocks per MCU
  int32_t hs[4], vs[4];
  int32_t ntables, lanes, waves_per_group, n_groups;
  uint32_t sub_bytes;
};

// ---- progressive frames and hidden refinement scans (huffman_prog_kernel) -------------------------------------------------
// One scan of a launch (device memory).  A launch holds scans that do not depend on one another (different components, or
// disjoint spectral bands of one); scans that refine what earlier ones wrote go into later launches on the same stream.
struct ProgScanDev {
  uint32_t stream_off;      // the scan's entropy coded data (no stuffing, no markers) in `data`, 16-byte aligned
  uint32_t first_interval;  // its first entry in ibegin / iend (offsets relative to stream_off)
  int32_t n_intervals, restart_interval, total_mcus, mcus_x; // (a scan without restart markers: one interval of total_mcus MCUs)
  int32_t ncomp, ntables;
  int32_t comp[4], hs[4], vs[4], bw[4];
  int64_t coef_off[4];      // plane offsets inside the frame's coefficient store, in coefficients
  int32_t dc_tab[4], ac_tab[4];
  uint32_t table_off;       // byte offset (from `tables`) of its ntables HuffDevTable
  int32_t ss, se, ah, al;   // spectral selection, successive approximation (al includes the hidden bits below a visible scan)
  int32_t runs_legal;       // EOB runs are legal (the reference's parser is a progressive one, sequentialscan.cpp:84-87)
  int32_t reserved;
};
struct ProgGroup {
  uint32_t scan, first_interval;
};
struct ProgArgs {
  const uint8_t *data;
  const uint32_t *ibegin, *iend;
  const ProgScanDev *scans;
  const ProgGroup *groups;  // one per workgroup (already offset to this launch's)
  int32_t n_groups;
  int32_t lanes, waves_per_group;
  int32_t max_tables;       // LDS is sized for this many tables
  int32_t wide;             // 1: int32 coefficients (JPEG XT residual frames with hidden bits)
  int32_t reserved;
  const uint8_t *tables;
  void *coef;               // the frame's coefficient store
  uint32_t *status;         // [0] error (0 = ok)
};
// max over a component's blocks of sum |c| q, saturating at 2^31 - 1, of finished planes -> status[1 + c] (atomicMax)
struct CoefRangeArgs {
  const void *coef;
  int32_t wide, ncomp;
  int64_t coef_off[4];
  int64_t nblocks[4];
  uint16_t q[4][64];
  uint32_t *status;
};

int launch_huffman_prog(const ProgArgs &a, hipStream_t stream);
int launch_coef_range(const CoefRangeArgs &a, hipStream_t stream);
int launch_huffman_scan(const HuffScanArgs &a, hipStream_t stream);
int launch_huffman_walk(const HuffWalkArgs &a, bool emit, hipStream_t stream);
int launch_huffman_walk_scan(const HuffWalkArgs &a, int n_images, hipStream_t stream);

} // namespace mij
#endif
