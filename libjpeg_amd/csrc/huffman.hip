// huffman.hip -- on-device entropy decoding of Huffman sequential scans with restart markers.
//
// One THREAD decodes one restart interval: the intervals are independent by construction (DC predictors and the bit
// buffer are reset at every RSTn, codestream/sequentialscan.cpp:266-274), the host only pre-scans the entropy coded
// segment for the marker positions (~0.4 ms for an 8K frame) and uploads the compressed bytes (a few MB) instead of
// ~100 MB of coefficients.  Semantics are those of the host decoder (host_decoder.cpp), i.e. of
// SequentialScan::DecodeBlock (codestream/sequentialscan.cpp:678-773) and BitStream<false>::Fill
// (io/bitstream.cpp:56-118): FF00 -> FF -- which the host applies while it copies the data into the upload buffer
// (HostDecoder::unstuff_piece / the marker search's sink), so that the kernels address plain bits -- and zero bits once
// the reader stands at the marker that ends its interval.
//
// This is latency-bound pointer chasing, not bandwidth-bound work, so everything is arranged to keep the serial
// chain of one interval short and to run many chains side by side:
//   * All lanes of a wave walk the same MCU / component / block structure and only diverge inside the per-block
//     symbol loop, which is branch-free apart from its exit.
//   * A frame rarely has enough restart intervals to fill 1024 SIMDs with full waves, so only the first `lanes`
//     lanes of a wave decode: more waves in flight and less divergence for free.
//   * The compressed bytes of a lane are staged through a 64-byte ring in LDS.  The ring is topped up at the block
//     boundaries (a point all lanes pass together) from registers that were loaded one block earlier, so the
//     global-memory latency hides behind the decoding of a whole block; the symbol loop only touches LDS.
//   * Coefficients are collected in a 128-byte LDS slot per lane; after every block the whole wave writes the slots
//     out as full 128-byte lines (16 bytes per lane), which also provides the zeros: no memset of the store and no
//     partial-line writes.  They land de-zigzagged as int16 in the planar store the reconstruction kernels read.
//   * Decoder tables (10-bit direct table + canonical fallback per table), deltas and zigzag order live in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "huffman_dev.hpp"

namespace mij {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef MIJPEG_HUFF_RING
#define MIJPEG_HUFF_RING 64 // (128 measured 5 % slower on sparse and on dense content alike: a fourth workgroup per CU is worth more than the longer prefetch)
#endif
constexpr int RING = MIJPEG_HUFF_RING; // bytes of stream staged per lane (power of two, >= 64: open() commits four chunks)
constexpr int RING_PITCH = RING;       // the reader only ever loads aligned dwords: no mirror behind the end
constexpr int LANE_LDS = 128 + RING_PITCH + 16; // coefficient slot + ring + block index (padded: 16-byte granules)

// Bit reader over the device copy of the entropy coded data, through the lane's LDS ring.
//
// The host uploads the data WITHOUT its byte stuffing and without the markers (HostDecoder::unstuff_piece: FF 00 -> FF is
// BitStream<false>::Fill's rule, io/bitstream.cpp:56-118), one restart interval behind the other.  A position in it is a
// plain bit address, and the window of the next 32 bits comes out of three registers that hold the big-endian stream dwords
// d, d + 1, d + 2 with one 64-bit shift.  Per symbol: a shift and a compare tell whether the window moved on to the next
// dword (at most one: a code and its value bits are 31 bits at most); the dword that then becomes B2 is read from the ring
// every time (same address while the window stays) -- it is first needed a symbol later, so the LDS round trip is off the
// symbol-to-symbol chain.  Round 1 carried the stuffing to the device: 0xFF detection, a 64-bit accumulator refilled four
// bytes at a time and the parked / phantom-bit bookkeeping cost about 40 instructions per symbol; this reader costs 15.
//
// From `endbit` on (the end of the restart interval) the reader supplies zero bits, as the reference's does in front of a
// marker (io/bitstream.cpp:96-101).  ring holds the stream bytes [fill - RING, fill) at offset (address & (RING - 1)); fill is a
// multiple of 16.  The fast path needs bp + 32 <= lim = min(endbit, 8 * fill - 64): the window lies inside the interval and
// dword d + 2 is in the ring; everything else -- the last symbols of an interval, a block that outran the prefetch -- goes
// through slow().  The stream buffer is padded (HUFF_STREAM_PAD) so that topping up past the end of the last interval stays
// inside the allocation.
struct DevBits {
  const uint8_t *base; // device copy of the data (16-byte aligned)
  uint8_t *ring;       // LDS
  uint32_t bp;         // next unread bit: 8 * byte address + bit, most significant first
  uint32_t endbit;     // end of the interval
  uint32_t fill;       // stream bytes committed to the ring so far
  uint32_t lim;
  uint32_t d;          // stream dword in B0
  uint32_t B0, B1;     // stream dwords d, d + 1 (big-endian: bit 31 first)
  uint32_t R2;         // dword d + 2 as the ring holds it (bytes not swapped yet: nothing waits for this load before the next symbol)
  uint32_t win;

  __device__ __forceinline__ u32x4 fetch(uint32_t at) const { return *reinterpret_cast<const u32x4 *>(base + at); }
  __device__ __forceinline__ void set_lim() { lim = min(endbit, 8u * fill - 64u); }
  __device__ __forceinline__ void commit(u32x4 v)
  {
    *reinterpret_cast<u32x4 *>(ring + (fill & (RING - 1))) = v;
    fill += 16;
    set_lim();
  }
  // a chunk may go in when it does not overwrite the dword the window starts in (slow() loads from there on)
  __device__ __forceinline__ bool room() const { return (int)(fill - ((bp >> 5) << 2)) <= RING - 16; }
  __device__ __forceinline__ uint32_t ld_raw(uint32_t dword) const { return *reinterpret_cast<const uint32_t *>(ring + ((dword << 2) & (RING - 1))); }
  __device__ __forceinline__ uint32_t ld(uint32_t dword) const { return __builtin_bswap32(ld_raw(dword)); }
  __device__ __forceinline__ void idle() // a lane that does not decode: never touches a ring (its would alias a decoding lane's)
  {
    bp = endbit = fill = lim = d = 0;
    B0 = B1 = R2 = win = 0;
  }
  __device__ __forceinline__ void open(const uint8_t *stream, uint8_t *lds_ring, uint32_t begin, uint32_t stop, uint32_t skip_bits = 0)
  {
    base = stream;
    ring = lds_ring;
    bp = 8u * begin + skip_bits;
    endbit = 8u * stop;
    fill = begin & ~15u;
    const u32x4 c0 = fetch(fill), c1 = fetch(fill + 16), c2 = fetch(fill + 32), c3 = fetch(fill + 48);
    commit(c0);
    commit(c1);
    commit(c2);
    commit(c3);
    d = bp >> 5;
    B0 = ld(d);
    B1 = ld(d + 1);
    R2 = ld_raw(d + 2);
    win = 0;
  }
  __device__ __forceinline__ void refill()
  {
    if (__builtin_expect(bp + 32u > lim, 0)) { slow(); return; }
    const uint32_t nd = bp >> 5;
    const bool adv = nd != d;
    B0 = adv ? B1 : B0;
    B1 = adv ? __builtin_bswap32(R2) : B1; // (read when the window stood one dword earlier)
    R2 = ld_raw(nd + 2);
    d = nd;
    win = (uint32_t)(((((uint64_t)B0) << 32 | B1) << (bp & 31u)) >> 32);
  }
  __device__ __forceinline__ void slow()
  {
    const uint32_t nd = bp >> 5;
    while ((nd + 3u) * 4u > fill) commit(fetch(fill)); // a block that outran the prefetch (rare)
    B0 = ld(nd);
    B1 = ld(nd + 1);
    R2 = ld_raw(nd + 2);
    d = nd;
    uint32_t w = (uint32_t)(((((uint64_t)B0) << 32 | B1) << (bp & 31u)) >> 32);
    const int avail = (int)(endbit - bp); // bits of the interval that are left; zero bits behind them
    if (avail < 32) w = avail <= 0 ? 0u : w & (~0u << (32 - avail));
    win = w;
  }
  __device__ __forceinline__ uint32_t window() const { return win; }
  __device__ __forceinline__ void skip(int k) { bp += (uint32_t)k; }
  // refill() for a reader that may have moved on by more than one dword since the last one (refill() follows the window over
  // one at most; slow() takes the three registers from the ring again)
  __device__ __forceinline__ void refill_far()
  {
    const uint32_t nd = bp >> 5;
    if (__builtin_expect(bp + 32u > lim || nd - d > 1u, 0)) { slow(); return; }
    const bool adv = nd != d;
    B0 = adv ? B1 : B0;
    B1 = adv ? __builtin_bswap32(R2) : B1;
    R2 = ld_raw(nd + 2);
    d = nd;
    win = (uint32_t)(((((uint64_t)B0) << 32 | B1) << (bp & 31u)) >> 32);
  }
  // up to 64 bits from bit address `at` on, left-aligned, straight from the ring (which is topped up as far as they reach)
  __device__ __forceinline__ uint64_t peek64(uint32_t at, int n)
  {
    while (8u * fill < at + (uint32_t)n) commit(fetch(fill));
    const uint32_t i = at >> 5, sh = at & 31u;
    const uint32_t d0 = ld(i), d1 = ld(i + 1), d2 = ld(i + 2);
    const uint32_t hi = (uint32_t)(((((uint64_t)d0) << 32 | d1) << sh) >> 32), lo = (uint32_t)(((((uint64_t)d1) << 32 | d2) << sh) >> 32);
    return ((uint64_t)hi << 32) | lo;
  }
};

// Huffman code at the top of the 32-bit window -> (tot << 8) | symbol, tot = code length + value bits that follow (what the
// reader moves on by), or HUFF_DEV_INVALID: no code matches, a DC symbol beyond 15, an AC symbol that does not exist in
// sequential scans (s == 0 with a run other than 0 and 15, :747-750).
// a + b, saturating at 2^32 - 1 (one full-rate instruction)
__device__ __forceinline__ uint32_t add_sat_u32(uint32_t a, uint32_t b)
{
  uint32_t d;
  asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

template <int AC> __device__ __forceinline__ uint32_t dev_lookup(uint32_t win, const HuffDevTable *h)
{
  uint32_t e = h->fast[win >> (32 - HUFF_DEV_LOOKAHEAD)];
  if (__builtin_expect((e & HUFF_DEV_SUB) != 0, 0)) { // a code of 11..16 bits (or none at all): ONE rarely taken branch per symbol
    const uint32_t t = e & 15u;
    e = t < (uint32_t)HUFF_DEV_SUBTABLES ? h->sub[t][(win >> (32 - HUFF_DEV_LOOKAHEAD - 6)) & 63u] : 0u;
    if (e == 0) { // a prefix beyond the second-level tables: canonical walk
      e = HUFF_DEV_INVALID;
      const int code16 = (int)(win >> 16);
      for (int l = HUFF_DEV_LOOKAHEAD + 1; l <= 16; l++) {
        const int code = code16 >> (16 - l);
        if (code <= h->maxcode[l]) {
          const uint32_t sym = h->values[(code + h->valoff[l]) & 0xff];
          e = huff_dev_entry(l, sym, AC);
          break;
        }
      }
    }
  }
  return e;
}

// The s value bits that end `tot` bits into the window (code + value), sign-extended the JPEG way (F.2.2.1 EXTEND).
__device__ __forceinline__ int dev_value(uint32_t win, int tot, int s)
{
  const uint32_t v = __builtin_amdgcn_ubfe(win, (uint32_t)(32 - tot), (uint32_t)s); // s = 0 -> 0
  const uint32_t full = (1u << s) - 1u;
  return (int)v - (int)(v <= (full >> 1) ? full : 0u);
}

// One block (sequentialscan.cpp:678-773) into this lane's 128-byte LDS slot (16-byte chunks XOR-swizzled by the lane
// so that lanes writing the same coefficient position hit different banks).  zq[i] = (delta << 16) | (2 * natural
// position) of scan position i.  The symbol loop has no branch but its exit and two rarely taken ones (long codes,
// awkward refills); it is rotated: the table lookup of the next symbol is issued before the coefficient of the
// current one is stored (after the last symbol of a block that lookup is simply not used).  ZRL and EOB store a zero
// at a position that still holds zero.  Returns 0 or HUFF_ERR_*.
__device__ __forceinline__ int dev_block(DevBits &br, const HuffDevTable *dc, const HuffDevTable *ac, const uint32_t *zq,
                                         uint8_t *slot, int swz16, int &pred, uint32_t &qmax)
{
  br.refill();
  uint32_t win = br.window();
  uint32_t e = dev_lookup<0>(win, dc);
  int s = (int)(e & 0xff), tot = (int)((e >> 8) & 31u);
  if (e & HUFF_DEV_INVALID) return HUFF_ERR_MALFORMED; // ":686 DC coefficient decoding out of sync"
  pred += dev_value(win, tot, s);
  br.skip(tot);
  if (pred != (int16_t)pred) return HUFF_ERR_OVERFLOW;
  br.refill();
  win = br.window();
  e = dev_lookup<1>(win, ac);
  *reinterpret_cast<int16_t *>(slot + swz16) = (int16_t)pred;
  // range check sum |c| q: both factors fit 16 bits (24-bit multiply), the sum saturates -- the host cuts at 2^31 - 1 like its
  // own decoder does (evaluate_entropy_status), and anything beyond that has saturated or is beyond it for good
  uint32_t qsum = __umul24((uint32_t)abs(pred), zq[0] >> 16);
  const uint32_t *zqm1 = zq - 1;
  int kk = 1; // scan position the next symbol's run starts from
  bool bad = false;
  for (;;) {
    const int rs = (int)(e & 0xff);
    s = rs & 15;
    tot = (int)((e >> 8) & 31u);
    const int val = dev_value(win, tot, s);
    br.skip(tot);
    kk += (rs >> 4) + 1; // one behind the position of this coefficient
    // no code / a symbol of progressive scans only (:747-750) / ":763 AC coefficient decoding out of sync"
    bad |= (e >= (uint32_t)HUFF_DEV_INVALID) | ((s != 0) & (kk > 64));
    const bool last = (rs == 0) | bad | (kk >= 64); // EOB, error, or position 63 reached (by a coefficient or a ZRL)
    br.refill();
    win = br.window();
    const uint32_t z = zqm1[kk]; // position <= 63 + 15, the table is padded; in flight together with the lookup below
    e = dev_lookup<1>(win, ac);
    *reinterpret_cast<int16_t *>(slot + ((z & 0xffffu) ^ (uint32_t)swz16)) = (int16_t)val;
    qsum = add_sat_u32(qsum, __umul24((uint32_t)abs(val), z >> 16));
    if (last) break;
  }
  qmax = max(qmax, qsum);
  return bad ? HUFF_ERR_MALFORMED : 0;
}

__device__ __forceinline__ void wave_lds_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void dev_exact_position(const DevBits &br, uint32_t &byte, uint32_t &skip);

// LDS of a workgroup: [tables | aux][per wave: lanes x 128-byte block slot | lanes x ring | lanes x block number]
__global__ __launch_bounds__(512) void huffman_scan_kernel(const HuffScanArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const int table_bytes = a.ntables * (int)sizeof(HuffDevTable) + (int)sizeof(HuffDevAux);
  const HuffDevTable *tabs = reinterpret_cast<const HuffDevTable *>(lds_raw);
  const HuffDevAux *aux = reinterpret_cast<const HuffDevAux *>(lds_raw + a.ntables * sizeof(HuffDevTable));
  const int L = a.lanes, nwaves = blockDim.x >> 6;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const HuffGroup grp = a.groups[blockIdx.x];
  const HuffImage img = a.images[grp.image]; // uniform: scalar loads
  uint8_t *stage = lds_raw + table_bytes + wv * (L * LANE_LDS);                     // L x 128
  uint8_t *rings = stage + L * 128;                                                 // L x RING_PITCH
  uint32_t *blkno = reinterpret_cast<uint32_t *>(stage + L * (128 + RING_PITCH));   // L x 4 (of 16)
  {
    // cooperative copy of the image's tables, deltas and zigzag order (dwords); clear the slots
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.tables + img.table_off);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds_raw);
    const int words = table_bytes / 4, rest = nwaves * L * LANE_LDS / 4;
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i < rest; i += blockDim.x) dst[words + i] = 0;
  }
  __syncthreads();
  const int interval = (int)grp.first_interval + wv * L + lane; // inside the image
  const bool decoding = lane < L && interval < img.n_intervals;
  const int ln = lane & (L - 1);
  const uint8_t *stream = a.data + img.stream_off;
  int16_t *coef = a.coef + img.coef_base;
  uint32_t *status = a.status + img.status_off;

  DevBits br;
  br.base = stream;
  br.ring = rings + ln * RING_PITCH;
  br.idle();
  int pred[4] = {0, 0, 0, 0};
  if (decoding) {
    const uint32_t idx = img.first_interval + (uint32_t)interval;
    // (virtual intervals: a restart point found by a walk -- mid-byte, with the predictors accumulated so far)
    br.open(stream, rings + ln * RING_PITCH, a.ibegin[idx], a.iend[idx], img.virt ? (uint32_t)a.iskip[idx] : 0u);
    if (img.virt) {
#pragma unroll
      for (int k = 0; k < 4; k++) pred[k] = a.ipred[idx * 4 + k];
    }
  }
  // the next 32 bytes of the stream travel in registers for the duration of one block
  uint32_t pend_at = br.fill;
  u32x4 pend0 = br.fetch(pend_at), pend1 = br.fetch(pend_at + 16);
  uint32_t qmax[4] = {0, 0, 0, 0};
  int err = 0;
  const int m0 = interval * img.restart_interval;
  uint8_t *slot = stage + ln * 128;
  const int swz16 = (lane & 7) << 4;
  for (int mi = 0; mi < img.restart_interval; mi++) {
    const int m = m0 + mi;
    const bool live = decoding && m < img.total_mcus;
    if (__ballot(live && !err) == 0) break;
    const int my = m / img.mcus_x, mx = m - my * img.mcus_x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k >= a.ncomp) break;
      const HuffDevTable *dc = tabs + a.dc_tab[k], *ac = tabs + a.ac_tab[k];
      const int h = a.hs[k], v = a.vs[k];
      const uint32_t comp_base = (uint32_t)(a.coef_off[k] >> 6); // in blocks
      for (int by = 0; by < v; by++)
        for (int bx = 0; bx < h; bx++) {
          uint32_t blk = 0; // block number + 1
          if (live && !err) {
            err = dev_block(br, dc, ac, aux->zq[k], slot, swz16, pred[k], qmax[k]);
            if (!err) blk = comp_base + (uint32_t)(my * v + by) * (uint32_t)a.bw[k] + (uint32_t)(mx * h + bx) + 1u;
          }
          // top the ring up with what was requested a block ago
          if (decoding && pend_at == br.fill && br.room()) br.commit(pend0);
          if (decoding && pend_at + 16 == br.fill && br.room()) br.commit(pend1);
          if (lane < L) blkno[lane] = blk;
          wave_lds_sync();
          // the wave writes the L blocks out in 16-byte chunks (full 128-byte lines) and clears the slots
          for (int c = lane; c < L * 8; c += 64) {
            const int sl = c >> 3, ch = c & 7;
            const uint32_t b = blkno[sl];
            u32x4 *src = reinterpret_cast<u32x4 *>(stage + sl * 128 + ((ch ^ (sl & 7)) << 4));
            const u32x4 val = *src;
            *src = u32x4{0, 0, 0, 0};
            if (b) *reinterpret_cast<u32x4 *>(coef + ((size_t)(b - 1) << 6) + ch * 8) = val;
          }
          wave_lds_sync();
          // request the next 32 bytes; they are not looked at before the next block is done
          pend_at = br.fill;
          pend0 = br.fetch(pend_at);
          pend1 = br.fetch(pend_at + 16);
        }
    }
  }
  // Virtual restart intervals come from a self-synchronising walk: on a damaged stream the walk can heal what a sequential
  // decoder (the reference) trips over.  A lane that decoded its MCUs must stand exactly where its successor starts;
  // otherwise the stream goes to the host decoder, which does what the reference does.
  if (img.virt && decoding && !err && interval + 1 < img.n_intervals) {
    uint32_t byte, skip;
    dev_exact_position(br, byte, skip);
    const uint32_t nidx = img.first_interval + (uint32_t)interval + 1u;
    if (byte != a.ibegin[nidx] || skip != (uint32_t)a.iskip[nidx]) err = HUFF_ERR_DESYNC;
  }
  // The bit-addressed reader hands out zero bits behind the interval's data without complaint.  An interval whose codes
  // stayed valid while it consumed more bits than it holds is a damaged one: the host decoder calls it dirty and walks the
  // stream sequentially, the reference fails once a request needs more than eight such bits (io/bitstream.hpp Get / Fill).
  // Every kind of interval -- real, virtual, the last one -- reports it, and the stream goes to the host.
  if (decoding && !err && br.bp > br.endbit) err = HUFF_ERR_DESYNC;
  if (err) atomicMax(&status[0], (uint32_t)err);
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (k < a.ncomp && qmax[k]) atomicMax(&status[1 + a.comp_of[k]], qmax[k]);
}

// ==============================================================================================
// Streams WITHOUT restart markers, entirely on the device: self-synchronising walk
// ==============================================================================================
// The entropy coded segment of an image is cut into subsequences of `sub_bytes` bytes and one lane walks one
// subsequence (no coefficient stores): from a start state (byte, bits to skip, index of the block inside the MCU) to
// the first block start at or behind the end of its subsequence, which it hands to its successor as THAT lane's start
// state for the next round.  Round 0 starts every lane at its boundary with a guessed phase; since Huffman-coded JPEG
// data re-synchronises, the hand-over states stop changing after a few rounds (Weissenberger & Schmidt, "Accelerating
// JPEG Decompression on GPUs", 2021: the same fixed point), and what is then known per subsequence -- number of blocks,
// sum of the DC differences per component -- gives, by prefix sums on the host, the block number and the DC predictors
// at every subsequence start.  A last walk (EMIT) writes the "virtual restart intervals" huffman_scan_kernel decodes
// from: byte, bit offset and predictors at every block whose number is a multiple of `emit_every`.
// Start state of a subsequence in one 64-bit word (read and written as a unit): byte, bits to skip, block in the MCU.
__device__ __forceinline__ uint64_t walk_state(uint32_t byte, uint32_t skip, uint32_t phase) { return (uint64_t)byte | ((uint64_t)skip << 32) | ((uint64_t)phase << 40); }
__device__ __forceinline__ uint32_t walk_state_byte(uint64_t s) { return (uint32_t)s; }
__device__ __forceinline__ uint32_t walk_state_skip(uint64_t s) { return (uint32_t)(s >> 32) & 7u; }
__device__ __forceinline__ uint32_t walk_state_phase(uint64_t s) { return (uint32_t)(s >> 40) & 0xffu; }

// One block without stores.  0 = fine, 1 = cannot be a block.
__device__ __forceinline__ int dev_walk_block(DevBits &br, const HuffDevTable *dc, const HuffDevTable *ac, int &dcdiff)
{
  br.refill();
  uint32_t win = br.window();
  uint32_t e = dev_lookup<0>(win, dc);
  int s = (int)(e & 0xff), tot = (int)((e >> 8) & 31u);
  if (e & HUFF_DEV_INVALID) return 1;
  dcdiff = dev_value(win, tot, s);
  br.skip(tot);
  int kk = 1;
  bool bad = false;
  for (;;) {
    br.refill();
    win = br.window();
    e = dev_lookup<1>(win, ac);
    const int rs = (int)(e & 0xff);
    s = rs & 15;
    tot = (int)((e >> 8) & 31u);
    br.skip(tot);
    kk += (rs >> 4) + 1;
    bad |= (e >= (uint32_t)HUFF_DEV_INVALID) | ((s != 0) & (kk > 64));
    if ((rs == 0) | bad | (kk >= 64)) break;
  }
  return bad ? 1 : 0;
}

// Where the reader stands, as (byte that holds the next unread bit, bits of it already used).
__device__ __forceinline__ void dev_exact_position(const DevBits &br, uint32_t &byte, uint32_t &skip)
{
  if (br.bp >= br.endbit) { byte = br.endbit >> 3; skip = 0; return; } // (zero bits behind the data are no position)
  byte = br.bp >> 3;
  skip = br.bp & 7u;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void huffman_walk_kernel(const HuffWalkArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const int table_bytes = a.ntables * (int)sizeof(HuffDevTable) + (int)sizeof(HuffDevAux);
  const HuffDevTable *tabs = reinterpret_cast<const HuffDevTable *>(lds_raw);
  uint8_t *blk_comp = lds_raw + table_bytes; // 64 bytes: scan component of block j of an MCU
  const int L = a.lanes;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t img_i = a.sub_image[blockIdx.x];
  const HuffImage img = a.images[img_i];
  uint8_t *rings = lds_raw + table_bytes + 64 + wv * (L * RING_PITCH);
  const uint32_t nsub = a.img_nsub[img_i], sub = a.sub_first[blockIdx.x] + (uint32_t)(wv * L + lane);
  const uint32_t si = a.img_sub0[img_i] + sub;
  // a lane walks (again) when its start state was written in the previous round -- or in this one already: the
  // states are updated in place, a round may see some of its own hand-overs, which only gets it to the fixed point sooner
  bool go = lane < L && sub < nsub;
  if (!EMIT) go = go && a.stamp[si] + 1u >= a.round;
  if (!__syncthreads_or(go)) return; // late rounds: hardly any workgroup has work left
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.tables + img.table_off);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds_raw);
    for (int i = threadIdx.x; i < table_bytes / 4; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) {
      int j = 0;
      for (int k = 0; k < a.ncomp; k++)
        for (int i = 0; i < a.hs[k] * a.vs[k]; i++) blk_comp[j++] = (uint8_t)k;
    }
  }
  __syncthreads();
  if (!go) return;
  const uint32_t e0 = a.img_e0[img_i], e1 = a.img_e1[img_i];
  const uint32_t limit = sub + 1 < nsub ? e0 + (sub + 1) * a.sub_bytes : e1; // first byte of the successor's range
  const uint8_t *stream = a.data + img.stream_off;
  const uint64_t st = a.state[si];

  DevBits br;
  br.open(stream, rings + lane * RING_PITCH, walk_state_byte(st), e1, walk_state_skip(st));
  int j = (int)walk_state_phase(st);
  uint32_t pend_at = br.fill;
  u32x4 pend0 = br.fetch(pend_at), pend1 = br.fetch(pend_at + 16);
  uint32_t nb = 0;
  int dcs[4] = {0, 0, 0, 0};
  // EMIT: running block number and predictors
  uint32_t g = EMIT ? a.first_block[si] : 0;
  const uint32_t my_blocks = EMIT ? a.nblocks[si] : 0xffffffffu;
  int pred[4] = {0, 0, 0, 0};
  if (EMIT) {
#pragma unroll
    for (int k = 0; k < 4; k++) pred[k] = a.first_pred[si * 4 + k];
  }
  uint32_t end_byte = e1, end_skip = 0;
  for (;;) {
    if (EMIT) {
      if (nb >= my_blocks || g >= a.total_blocks) break;
      if (g % a.emit_every == 0) {
        uint32_t q, sk;
        dev_exact_position(br, q, sk);
        const uint32_t idx = a.img_int0[img_i] + g / a.emit_every;
        a.ibegin[idx] = q;
        a.iskip[idx] = (uint8_t)sk;
#pragma unroll
        for (int k = 0; k < 4; k++) a.ipred[idx * 4 + k] = (int16_t)pred[k];
      }
    } else {
      if (br.bp >= br.endbit) break; // the data end here
      if ((br.bp >> 3) >= limit) { // in the successor's range
        end_byte = br.bp >> 3;
        end_skip = br.bp & 7u;
        break;
      }
    }
    const int k = blk_comp[j];
    int dcdiff = 0;
    const int bad = dev_walk_block(br, tabs + 2 * k, tabs + 2 * k + 1, dcdiff);
    if (br.bp > br.endbit) break; // ran over the end of the data inside a block
    if (bad) {
      if (EMIT) break; // cannot happen on the path the rounds agreed on
      br.skip(1); // not a block: move on by one bit and guess again
      j = 0;
    } else {
      nb++;
      if (EMIT) { g++; pred[k] += dcdiff; }
      else dcs[k] += dcdiff;
      j = j + 1 == a.nblk_mcu ? 0 : j + 1;
    }
    // top the ring up with what was requested a block ago, request the next 32 bytes
    if (pend_at == br.fill && br.room()) br.commit(pend0);
    if (pend_at + 16 == br.fill && br.room()) br.commit(pend1);
    pend_at = br.fill;
    pend0 = br.fetch(pend_at);
    pend1 = br.fetch(pend_at + 16);
  }
  if (!EMIT) {
    a.nblocks[si] = nb;
#pragma unroll
    for (int k = 0; k < 4; k++) a.dcsum[si * 4 + k] = dcs[k];
    if (sub + 1 < nsub) {
      const uint32_t t = si + 1;
      const uint64_t handed = walk_state(end_byte, end_skip, (uint32_t)j);
      if (a.state[t] != handed) {
        a.state[t] = handed;
        a.stamp[t] = a.round;
        a.changed[a.round] = 1;
      }
    }
  }
}

// Between the rounds and the EMIT walk: exclusive prefix sums of the block counts and DC sums over the subsequences of
// every image, and the checks that tell a fixed point that is a decode of the image from one that is not: the phase
// every subsequence starts in must be its block number modulo the blocks per MCU, the DC predictors must fit the
// coefficient store, and the blocks must add up to the frame.  Three small kernels: sums per tile of 1024
// subsequences, scan of the tile sums (one workgroup per image), scan inside the tiles + checks.
constexpr int WALK_TILE = 1024;
struct WalkSums {
  long long v[5]; // blocks, DC sums
};
__device__ __forceinline__ WalkSums walk_sums_add(const WalkSums &x, const WalkSums &y)
{
  WalkSums r;
#pragma unroll
  for (int c = 0; c < 5; c++) r.v[c] = x.v[c] + y.v[c];
  return r;
}
// inclusive scan over the workgroup's threads (WALK_TILE of them); `wave_tot` = 16 WalkSums of LDS
__device__ __forceinline__ WalkSums walk_block_scan(WalkSums mine, WalkSums *wave_tot, WalkSums &block_total)
{
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    WalkSums o;
#pragma unroll
    for (int c = 0; c < 5; c++) o.v[c] = __shfl_up(mine.v[c], d, 64);
    if (lane >= d) mine = walk_sums_add(mine, o);
  }
  if (lane == 63) wave_tot[wv] = mine;
  __syncthreads();
  WalkSums before{{0, 0, 0, 0, 0}}, all{{0, 0, 0, 0, 0}};
  for (int k = 0; k < WALK_TILE / 64; k++) {
    const WalkSums t = wave_tot[k];
    if (k < wv) before = walk_sums_add(before, t);
    all = walk_sums_add(all, t);
  }
  __syncthreads();
  block_total = all;
  return walk_sums_add(mine, before);
}
__device__ __forceinline__ WalkSums walk_load(const HuffWalkArgs &a, uint32_t s0, uint32_t i, uint32_t nsub)
{
  WalkSums m{{0, 0, 0, 0, 0}};
  if (i < nsub) {
    m.v[0] = a.nblocks[s0 + i];
    const int4 dc = *reinterpret_cast<const int4 *>(a.dcsum + (size_t)(s0 + i) * 4);
    m.v[1] = dc.x; m.v[2] = dc.y; m.v[3] = dc.z; m.v[4] = dc.w;
  }
  return m;
}
// grid (tiles, images)
__global__ __launch_bounds__(WALK_TILE) void huffman_walk_tile_sums_kernel(const HuffWalkArgs a)
{
  __shared__ WalkSums wave_tot[WALK_TILE / 64];
  const uint32_t img_i = blockIdx.y, nsub = a.img_nsub[img_i], s0 = a.img_sub0[img_i];
  if (blockIdx.x * WALK_TILE >= nsub) return;
  WalkSums total;
  (void)walk_block_scan(walk_load(a, s0, blockIdx.x * WALK_TILE + threadIdx.x, nsub), wave_tot, total);
  if (threadIdx.x == 0) a.tile_sums[(size_t)img_i * a.tiles_per_image + blockIdx.x] = total;
}
// grid (images): exclusive scan of an image's tile sums, in place (at most WALK_TILE tiles per image)
__global__ __launch_bounds__(WALK_TILE) void huffman_walk_tile_scan_kernel(const HuffWalkArgs a)
{
  __shared__ WalkSums wave_tot[WALK_TILE / 64];
  const uint32_t img_i = blockIdx.x, nsub = a.img_nsub[img_i];
  if (nsub == 0) return;
  const uint32_t tiles = (nsub + WALK_TILE - 1) / WALK_TILE;
  WalkSums *ts = a.tile_sums + (size_t)img_i * a.tiles_per_image;
  WalkSums mine{{0, 0, 0, 0, 0}}, total;
  if (threadIdx.x < tiles) mine = ts[threadIdx.x];
  const WalkSums incl = walk_block_scan(mine, wave_tot, total);
  if (threadIdx.x < tiles) {
#pragma unroll
    for (int c = 0; c < 5; c++) ts[threadIdx.x].v[c] = incl.v[c] - mine.v[c];
  }
  if (threadIdx.x == 0 && total.v[0] < (long long)a.total_blocks) atomicOr(&a.walk_status[img_i], 4u);
}
// grid (tiles, images)
__global__ __launch_bounds__(WALK_TILE) void huffman_walk_prefix_kernel(const HuffWalkArgs a)
{
  __shared__ WalkSums wave_tot[WALK_TILE / 64];
  const uint32_t img_i = blockIdx.y, nsub = a.img_nsub[img_i], s0 = a.img_sub0[img_i];
  if (blockIdx.x * WALK_TILE >= nsub) return;
  const uint32_t i = blockIdx.x * WALK_TILE + threadIdx.x;
  const WalkSums mine = walk_load(a, s0, i, nsub);
  WalkSums total;
  WalkSums run = walk_block_scan(mine, wave_tot, total);
  const WalkSums off = a.tile_sums[(size_t)img_i * a.tiles_per_image + blockIdx.x];
#pragma unroll
  for (int c = 0; c < 5; c++) run.v[c] += off.v[c] - mine.v[c]; // exclusive
  if (i >= nsub) return;
  const long long total_blocks = a.total_blocks;
  const bool live = run.v[0] < total_blocks; // the subsequence starts inside the image, not in the padding behind it
  uint32_t bad = 0;
  if (live && mine.v[0] && walk_state_phase(a.state[s0 + i]) != (uint32_t)(run.v[0] % a.nblk_mcu)) bad |= 1;
  a.first_block[s0 + i] = (uint32_t)(live ? run.v[0] : total_blocks);
  int4 fp;
  fp.x = (int)run.v[1]; fp.y = (int)run.v[2]; fp.z = (int)run.v[3]; fp.w = (int)run.v[4];
#pragma unroll
  for (int c = 1; c < 5; c++)
    if (live && run.v[c] != (long long)(short)run.v[c]) bad |= 2;
  *reinterpret_cast<int4 *>(a.first_pred + (size_t)(s0 + i) * 4) = fp;
  if (bad) atomicOr(&a.walk_status[img_i], bad);
}


// ==============================================================================================
// Progressive frames (SOF2) and hidden refinement scans (JPEG XT) with restart markers: one lane per restart interval
// ==============================================================================================
// A progressive frame is a sequence of scans over the same coefficient store: DC first passes (interleaved), AC first passes
// over a spectral band of one component with EOB runs (codestream/sequentialscan.cpp:678-773 with m_bProgressive), and
// successive approximation refinement scans (codestream/refinementscan.cpp:584-700; T.81 G.1.2.3).  The hidden refinement
// scans of a JPEG XT frame (marker/scan.cpp:899-980) are refinement scans below a sequential frame's visible scan, which then
// carries a point transform.  Restart intervals are independent here as well: predictors, the bit reader AND the EOB run are
// reset at every RSTn (sequentialscan.cpp:266-274, refinementscan.cpp:223-232).
//
// What is NOT parallel in this coding: an AC refinement scan without restart markers.  How many bits a block takes there
// depends on the block's own non-zero pattern (one correction bit per coefficient the earlier scans left non-zero), so a
// decoder that starts somewhere in the middle of the data cannot know which block it is in and does not fall into step with
// the real one the way the decoders of first passes do: such scans stay on the host.
//
// Blocks of AC scans travel through a slot in LDS per lane: the wave loads its L blocks as full lines, every lane decodes
// into / refines its slot, the wave writes them back.  DC scans (band 0..0) touch one coefficient per block and go straight
// to memory.  Natural-order position of scan position k (padded like HuffDevAux::zq: a corrupt run may point behind 63).
__device__ const uint8_t __attribute__((aligned(16))) prog_zigzag[80] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                            6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                            39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

template <class T> struct ProgStore {
  uint8_t *slot;  // this lane's block in LDS, natural order, 16-byte chunks XOR-swizzled by the lane
  uint32_t swz;
  __device__ __forceinline__ T *at(int pos) const { return reinterpret_cast<T *>(slot + (((uint32_t)pos * (uint32_t)sizeof(T)) ^ swz)); }
  __device__ __forceinline__ int get(int pos) const { return (int)*at(pos); }
  __device__ __forceinline__ void put(int pos, int v) const { *at(pos) = (T)v; }
};

// First pass of a block (sequentialscan.cpp:678-773): DC difference when the band starts at 0, then run / size pairs with EOB
// runs over [max(ss, 1), se], everything shifted up by al.  Returns 0 or HUFF_ERR_*.
template <class T>
__device__ __forceinline__ int prog_block_first(DevBits &br, const HuffDevTable *dc, const HuffDevTable *ac, const uint8_t *zz, const ProgStore<T> &st,
                                                int &pred, int &skip, int ss, int se, int al, bool runs_legal)
{
  if (ss == 0) {
    br.refill();
    const uint32_t win = br.window();
    const uint32_t e = dev_lookup<0>(win, dc);
    if (e & HUFF_DEV_INVALID) return HUFF_ERR_MALFORMED;
    const int s = (int)(e & 0xff), tot = (int)((e >> 8) & 31u);
    pred += dev_value(win, tot, s);
    br.skip(tot);
    const int v = (int)((uint32_t)pred << al);
    if (v != (int)(T)v) return HUFF_ERR_OVERFLOW;
    st.put(0, v);
  }
  if (se) {
    if (skip > 0) { skip--; return 0; }
    int k = ss ? ss : 1;
    do {
      br.refill();
      const uint32_t win = br.window();
      const uint32_t e = dev_lookup<2>(win, ac);
      if (e >= (uint32_t)HUFF_DEV_INVALID) return HUFF_ERR_MALFORMED;
      const int rs = (int)(e & 0xff), r = rs >> 4, s = rs & 15, tot = (int)((e >> 8) & 31u);
      br.skip(tot);
      if (s == 0) {
        if (r == 15) { k += 16; continue; }
        skip = (int)((1u << r) | __builtin_amdgcn_ubfe(win, (uint32_t)(32 - tot), (uint32_t)r)) - 1;
        // an EOB run where the reference's parser is not a progressive one (sequentialscan.cpp:84-87, 722-750): its walk decides
        if (skip > 0 && !runs_legal) return HUFF_ERR_MALFORMED;
        break;
      }
      k += r;
      const int v = (int)((uint32_t)dev_value(win, tot, s) << al);
      if (k >= 64) return HUFF_ERR_MALFORMED;
      if (v != (int)(T)v) return HUFF_ERR_OVERFLOW;
      st.put(zz[k], v);
      k++;
    } while (k <= se);
  }
  return 0;
}

// scan position k[p] of natural position p and natural position zz[k] of scan position k, for straight-line code
struct ProgUnzz {
  uint8_t k[64], zz[64];
  constexpr ProgUnzz() : k{}, zz{}
  {
    constexpr uint8_t order[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    for (int i = 0; i < 64; i++) {
      zz[i] = order[i];
      k[order[i]] = (uint8_t)i;
    }
  }
};
constexpr ProgUnzz PROG_UNZZ;

// Refinement pass of a block (refinementscan.cpp:584-700; T.81 G.1.2.3), in three steps so that what is serial -- one
// Huffman symbol after the other, each of which says how far to go -- is a short loop per SYMBOL and everything per
// coefficient is the same straight code in every lane:
//   1. the block's non-zero pattern H in scan order (bit k = scan position k), from the slot, sixteen bytes at a time, and
//      the list of the band's FREE positions (scan position and natural position, two bytes each) in the lane's LDS;
//   2. per symbol: its target is the free position r entries on in that list; the history positions passed on the way
//      take one correction bit each -- target - cursor - r of them, and they follow the symbol in the data in position
//      order, so they are taken in one piece (from the reader's window when they fit) and appended to the block's string of
//      correction bits CC; a new coefficient goes straight into the slot;
//   3. every coefficient once: a history coefficient takes the bit of CC its rank in H names.
// Same statements as the host's decode_block_refine, same verdicts.
constexpr int PROG_ZLIST_PITCH = 132; // 64 entries of two bytes + a bank's worth, so that lanes at the same entry do not collide
template <class T>
__device__ __forceinline__ int prog_block_refine(DevBits &br, const HuffDevTable *ac, const ProgStore<T> &st, uint16_t *zl, int &skip, int ss, int se, int al)
{
  if (ss == 0) {
    br.refill_far(); // (the block before this one may have left the reader several dwords on)
    const int bit = (int)(br.window() >> 31);
    br.skip(1);
    st.put(0, st.get(0) | (bit << al));
  }
  if (!se) return 0;
  constexpr int PER = 16 / (int)sizeof(T), CHUNKS = 64 / PER;
  uint32_t hlo = 0, hhi = 0;
#pragma unroll
  for (int ch = 0; ch < CHUNKS; ch++) {
    const u32x4 w = *reinterpret_cast<const u32x4 *>(st.slot + (((uint32_t)ch << 4) ^ st.swz));
#pragma unroll
    for (int e = 0; e < PER; e++) {
      const int k = PROG_UNZZ.k[ch * PER + e];
      const int v = sizeof(T) == 4 ? (int)w[e] : (int)(short)(w[e >> 1] >> ((e & 1) * 16));
      if (k < 32) hlo |= v ? 1u << k : 0u;
      else hhi |= v ? 1u << (k - 32) : 0u;
    }
  }
  const uint64_t band = (~0ull << ss) & (~0ull >> (63 - se));
  const uint32_t Hlo = hlo & (uint32_t)band, Hhi = hhi & (uint32_t)(band >> 32);
  const uint32_t zlo = ~hlo & (uint32_t)band, zhi = ~hhi & (uint32_t)(band >> 32);
  int nz = 0;
#pragma unroll
  for (int k = 0; k < 64; k++) { // (written at every step, kept where the position is free)
    zl[nz] = (uint16_t)(k | (PROG_UNZZ.zz[k] << 8));
    nz += (int)((k < 32 ? zlo >> k : zhi >> (k - 32)) & 1u);
  }
  const int nHlo = __popc(Hlo), nH = nHlo + __popc(Hhi);
  const int one = (int)(1u << al);
  uint64_t CC = 0;
  int count = 0, passed = 0, zi = 0, k = ss;
  bool eob = false, bad = false;
  if (skip > 0) { skip--; eob = true; } // inside an EOB run: corrections only
  while (!eob && k <= se) {
    br.refill_far();
    const uint32_t win = br.window();
    const uint32_t e = dev_lookup<2>(win, ac);
    const int rs = (int)(e & 0xff), r = rs >> 4, s = rs & 15, tot = (int)((e >> 8) & 31u);
    if (e >= (uint32_t)HUFF_DEV_INVALID || s > 1) { // (a size beyond one: the reference warns and decodes on -- its walk on the host)
      bad = true;
      break;
    }
    if (s == 0 && r < 15) { // EOBr: the rest of the band, in this block and the next `skip`, only takes its corrections
      skip = (int)((1u << r) | __builtin_amdgcn_ubfe(win, (uint32_t)(32 - tot), (uint32_t)r)) - 1;
      br.skip(tot);
      eob = true;
      break;
    }
    const int j = zi + r; // (ZRL: the sixteenth free position from here, which stays zero)
    zi = j + 1;
    const bool found = j < nz;
    const uint32_t ent = found ? (uint32_t)zl[j] : 64u;
    const int target = (int)(ent & 0xffu);
    const int ncorr = found ? target - k - r : nH - passed; // the history positions between the cursor and the target
    if (ncorr) {
      uint64_t chunk;
      if (tot + ncorr <= 32) chunk = (uint64_t)((win << tot) & (~0u << (32 - ncorr))) << 32;
      else chunk = br.peek64(br.bp + (uint32_t)tot, ncorr) & (~0ull << (64 - ncorr));
      CC |= chunk >> count;
      count += ncorr;
      passed += ncorr;
    }
    br.bp += (uint32_t)(tot + ncorr);
    if (s && found) st.put((int)(ent >> 8), __builtin_amdgcn_ubfe(win, (uint32_t)(32 - tot), 1u) ? one : -one); // the sign bit behind the code
    k = target + 1;
  }
  if (eob && nH > passed) { // the history coefficients behind the last symbol
    const int ncorr = nH - passed;
    CC |= (br.peek64(br.bp, ncorr) & (~0ull << (64 - ncorr))) >> count;
    br.bp += (uint32_t)ncorr;
  }
  // every coefficient once
  bool overflow = false;
#pragma unroll
  for (int ch = 0; ch < CHUNKS; ch++) {
    u32x4 *slot = reinterpret_cast<u32x4 *>(st.slot + (((uint32_t)ch << 4) ^ st.swz));
    const u32x4 w = *slot;
    u32x4 o = w;
#pragma unroll
    for (int e = 0; e < PER; e++) {
      const int kp = PROG_UNZZ.k[ch * PER + e];
      const int v = sizeof(T) == 4 ? (int)w[e] : (int)(short)(w[e >> 1] >> ((e & 1) * 16));
      const bool hk = kp < 32 ? (Hlo >> kp) & 1u : (Hhi >> (kp - 32)) & 1u;
      const int rank = kp < 32 ? __popc(Hlo & ((1u << kp) - 1u)) : nHlo + __popc(Hhi & ((1u << (kp - 32)) - 1u));
      const bool cbit = (uint32_t)((CC << rank) >> 63) != 0;
      const int nv = v + (hk && cbit ? (v < 0 ? -one : one) : 0);
      overflow |= nv != (int)(T)nv;
      if (sizeof(T) == 4) o[e] = (uint32_t)nv;
      else o[e >> 1] = (e & 1) ? (o[e >> 1] & 0xffffu) | ((uint32_t)nv << 16) : (o[e >> 1] & 0xffff0000u) | ((uint32_t)nv & 0xffffu);
    }
    *slot = o;
  }
  return bad ? HUFF_ERR_MALFORMED : overflow ? HUFF_ERR_OVERFLOW : 0;
}

// LDS: [max_tables tables | zigzag order][per wave: L block slots | L rings | L block numbers | L lists of free positions]
constexpr int PROG_ZZ_BYTES = 80;
template <class T> __global__ __launch_bounds__(256) void huffman_prog_kernel(const ProgArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  constexpr int SLOT = 64 * (int)sizeof(T), CH = SLOT / 16;
  constexpr int LANE_BYTES = SLOT + RING_PITCH + 16 + PROG_ZLIST_PITCH;
  const int table_bytes = a.max_tables * (int)sizeof(HuffDevTable) + PROG_ZZ_BYTES;
  const uint8_t *zz = lds_raw + table_bytes - PROG_ZZ_BYTES; // (a look-up per coefficient, at a position that differs from lane to lane: LDS, not memory)
  const HuffDevTable *tabs = reinterpret_cast<const HuffDevTable *>(lds_raw);
  const int L = a.lanes, nwaves = blockDim.x >> 6;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const ProgGroup grp = a.groups[blockIdx.x];
  const ProgScanDev sc = a.scans[grp.scan]; // uniform
  uint8_t *stage = lds_raw + table_bytes + wv * (L * LANE_BYTES);
  uint8_t *rings = stage + L * SLOT;
  uint32_t *blkno = reinterpret_cast<uint32_t *>(stage + L * (SLOT + RING_PITCH));
  uint16_t *zlist = reinterpret_cast<uint16_t *>(stage + L * (SLOT + RING_PITCH + 16) + (lane & (L - 1)) * PROG_ZLIST_PITCH);
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.tables + sc.table_off);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds_raw);
    const int words = sc.ntables * (int)sizeof(HuffDevTable) / 4, first = table_bytes / 4, rest = nwaves * L * LANE_BYTES / 4;
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i < rest; i += blockDim.x) dst[first + i] = 0;
    if (threadIdx.x < PROG_ZZ_BYTES / 4) dst[first - PROG_ZZ_BYTES / 4 + threadIdx.x] = reinterpret_cast<const uint32_t *>(prog_zigzag)[threadIdx.x];
  }
  __syncthreads();
  const int interval = (int)grp.first_interval + wv * L + lane;
  const bool decoding = lane < L && interval < sc.n_intervals;
  const int ln = lane & (L - 1);
  const uint8_t *stream = a.data + sc.stream_off;
  T *coef = reinterpret_cast<T *>(a.coef);
  DevBits br;
  br.base = stream;
  br.ring = rings + ln * RING_PITCH;
  br.idle();
  if (decoding) {
    const uint32_t idx = sc.first_interval + (uint32_t)interval;
    br.open(stream, rings + ln * RING_PITCH, a.ibegin[idx], a.iend[idx]);
  }
  uint32_t pend_at = br.fill;
  u32x4 pend0 = br.fetch(pend_at), pend1 = br.fetch(pend_at + 16);
  int pred[4] = {0, 0, 0, 0}, skip[4] = {0, 0, 0, 0};
  int err = 0;
  const int m0 = interval * sc.restart_interval;
  ProgStore<T> st;
  st.slot = stage + ln * SLOT;
  st.swz = (uint32_t)(lane & (CH - 1)) << 4;
  const bool dc_only = sc.se == 0;
  const bool fresh = sc.ah == 0 && sc.ss == 0 && sc.se == 63;
  for (int mi = 0; mi < sc.restart_interval; mi++) {
    const int m = m0 + mi;
    const bool live = decoding && m < sc.total_mcus;
    // An interval that has read past its data is a damaged one (the check behind the loop): it stops HERE.  A scan without
    // restart markers is one interval of any number of blocks -- an EOB run covers 32 767 of them in three bytes -- and what a
    // lane reads behind its data is whatever lies there, a hundred bytes a block: it must not walk out of the buffer.
    if (live && !err && br.bp > br.endbit) err = HUFF_ERR_DESYNC;
    if (__ballot(live && !err) == 0) break;
    const int my = m / sc.mcus_x, mx = m - my * sc.mcus_x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k >= sc.ncomp) break;
      const HuffDevTable *dc = tabs + sc.dc_tab[k], *ac = tabs + sc.ac_tab[k];
      const int h = sc.hs[k], v = sc.vs[k];
      const uint32_t comp_base = (uint32_t)(sc.coef_off[k] >> 6);
      for (int by = 0; by < v; by++)
        for (int bx = 0; bx < h; bx++) {
          const uint32_t bidx = comp_base + (uint32_t)(my * v + by) * (uint32_t)sc.bw[k] + (uint32_t)(mx * h + bx);
          if (dc_only) {
            // one coefficient per block: straight to memory (DC first pass: prediction, point transform; DC refinement: one raw bit)
            if (live && !err) {
              T *p = coef + ((size_t)bidx << 6);
              br.refill();
              const uint32_t win = br.window();
              if (sc.ah == 0) {
                const uint32_t e = dev_lookup<0>(win, dc);
                if (e & HUFF_DEV_INVALID) err = HUFF_ERR_MALFORMED;
                else {
                  const int s = (int)(e & 0xff), tot = (int)((e >> 8) & 31u);
                  pred[k] += dev_value(win, tot, s);
                  br.skip(tot);
                  const int val = (int)((uint32_t)pred[k] << sc.al);
                  if (val != (int)(T)val) err = HUFF_ERR_OVERFLOW;
                  else *p = (T)val;
                }
              } else {
                br.skip(1);
                *p = (T)((int)*p | ((int)(win >> 31) << sc.al));
              }
            }
            if (decoding && pend_at == br.fill && br.room()) br.commit(pend0);
            if (decoding && pend_at + 16 == br.fill && br.room()) br.commit(pend1);
            pend_at = br.fill;
            pend0 = br.fetch(pend_at);
            pend1 = br.fetch(pend_at + 16);
            continue;
          }
          const bool work = live && !err;
          if (lane < L) blkno[lane] = work ? bidx + 1u : 0u;
          wave_lds_sync();
          for (int c = lane; c < L * CH; c += 64) { // the wave fetches its L blocks as whole lines
            const int sl = c / CH, ch = c % CH;
            const uint32_t b = blkno[sl];
            // (a first pass over the whole band meets the zeros the planes were cleared to: nothing to fetch)
            if (b) *reinterpret_cast<u32x4 *>(stage + sl * SLOT + ((ch ^ (sl & (CH - 1))) << 4)) =
                     fresh ? u32x4{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint8_t *>(coef + ((size_t)(b - 1) << 6)) + ch * 16);
          }
          wave_lds_sync();
          if (work)
            err = sc.ah == 0 ? prog_block_first<T>(br, dc, ac, zz, st, pred[k], skip[k], sc.ss, sc.se, sc.al, sc.runs_legal != 0)
                             : prog_block_refine<T>(br, ac, st, zlist, skip[k], sc.ss, sc.se, sc.al);
          if (decoding && pend_at == br.fill && br.room()) br.commit(pend0);
          if (decoding && pend_at + 16 == br.fill && br.room()) br.commit(pend1);
          wave_lds_sync();
          for (int c = lane; c < L * CH; c += 64) { // ... and writes them back (a block whose lane failed keeps what it held)
            const int sl = c / CH, ch = c % CH;
            const uint32_t b = blkno[sl];
            if (b) *reinterpret_cast<u32x4 *>(reinterpret_cast<uint8_t *>(coef + ((size_t)(b - 1) << 6)) + ch * 16) =
                     *reinterpret_cast<const u32x4 *>(stage + sl * SLOT + ((ch ^ (sl & (CH - 1))) << 4));
          }
          wave_lds_sync();
          pend_at = br.fill;
          pend0 = br.fetch(pend_at);
          pend1 = br.fetch(pend_at + 16);
        }
    }
  }
  // (as in huffman_scan_kernel: an interval that only decoded by reading the zero bits behind its data is a damaged one)
  if (decoding && !err && br.bp > br.endbit) err = HUFF_ERR_DESYNC;
  if (err) atomicMax(&a.status[0], (uint32_t)err);
}

int launch_huffman_prog(const ProgArgs &a, hipStream_t stream)
{
  if (a.n_groups <= 0) return 0;
  const size_t slot = a.wide ? 256 : 128;
  const size_t lds = (size_t)a.max_tables * sizeof(HuffDevTable) + PROG_ZZ_BYTES + (size_t)a.waves_per_group * a.lanes * (slot + RING_PITCH + 16 + PROG_ZLIST_PITCH);
  if (a.wide) hipLaunchKernelGGL(huffman_prog_kernel<int32_t>, dim3(a.n_groups), dim3(64 * a.waves_per_group), lds, stream, a);
  else hipLaunchKernelGGL(huffman_prog_kernel<int16_t>, dim3(a.n_groups), dim3(64 * a.waves_per_group), lds, stream, a);
  return (int)hipGetLastError();
}

// max over the blocks of a component of sum |c| q (saturating at 2^31 - 1) of finished planes: what selects the arithmetic of
// the reconstruction (the host decoder's range_pass).  grid (blocks / 256, components)
// (a lane takes sixteen bytes of a block, the lanes of a block add up what they found: the planes are read in whole lines)
constexpr int RANGE_ROUNDS = 8; // blocks per lane group and workgroup
template <class T> __global__ __launch_bounds__(256) void coef_range_kernel(const CoefRangeArgs a)
{
  constexpr int PER = 16 / (int)sizeof(T), LPB = 64 / PER, BPG = 256 / LPB; // coefficients per lane, lanes per block, blocks per round
  __shared__ uint16_t q[64];
  const int c = blockIdx.y;
  if (threadIdx.x < 64) q[threadIdx.x] = a.q[c][threadIdx.x];
  __syncthreads();
  const int j = threadIdx.x % LPB;
  uint32_t qs[PER];
#pragma unroll
  for (int e = 0; e < PER; e++) qs[e] = q[j * PER + e];
  const T *plane = reinterpret_cast<const T *>(a.coef) + a.coef_off[c];
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < RANGE_ROUNDS; r++) {
    const int64_t b = ((int64_t)blockIdx.x * RANGE_ROUNDS + r) * BPG + threadIdx.x / LPB;
    uint64_t sum = 0;
    if (b < a.nblocks[c]) {
      const u32x4 v = *reinterpret_cast<const u32x4 *>(plane + b * 64 + j * PER);
#pragma unroll
      for (int e = 0; e < PER; e++) {
        int x;
        if (sizeof(T) == 2) x = (int)(int16_t)(v[e >> 1] >> ((e & 1) * 16));
        else x = (int)v[e];
        const int64_t ax = x < 0 ? -(int64_t)x : (int64_t)x;
        sum += (uint64_t)ax * qs[e];
      }
    }
#pragma unroll
    for (int o = LPB / 2; o > 0; o >>= 1) {
      const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)sum, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(sum >> 32), o);
      sum += ((uint64_t)hi << 32) | lo;
    }
    m = max(m, (uint32_t)(sum < 0x7fffffffull ? sum : 0x7fffffffull));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  // (one word per component takes every group's maximum: asked only by groups that would raise it)
  if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(&a.status[1 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&a.status[1 + c], m);
}

int launch_coef_range(const CoefRangeArgs &a, hipStream_t stream)
{
  int64_t most = 0;
  for (int c = 0; c < a.ncomp; c++) most = a.nblocks[c] > most ? a.nblocks[c] : most;
  if (most <= 0 || a.ncomp <= 0) return 0;
  const int64_t per_group = (int64_t)RANGE_ROUNDS * (a.wide ? 16 : 32);
  const dim3 grid((unsigned)((most + per_group - 1) / per_group), (unsigned)a.ncomp);
  if (a.wide) hipLaunchKernelGGL(coef_range_kernel<int32_t>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(coef_range_kernel<int16_t>, grid, dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_huffman_walk_scan(const HuffWalkArgs &a, int n_images, hipStream_t stream)
{
  static_assert(sizeof(WalkSums) == HUFF_WALK_SUMS_BYTES, "host sizes the tile sums");
  if (a.tiles_per_image <= 0 || a.tiles_per_image > WALK_TILE) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(huffman_walk_tile_sums_kernel, dim3(a.tiles_per_image, n_images), dim3(WALK_TILE), 0, stream, a);
  hipLaunchKernelGGL(huffman_walk_tile_scan_kernel, dim3(n_images), dim3(WALK_TILE), 0, stream, a);
  hipLaunchKernelGGL(huffman_walk_prefix_kernel, dim3(a.tiles_per_image, n_images), dim3(WALK_TILE), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_huffman_walk(const HuffWalkArgs &a, bool emit, hipStream_t stream)
{
  if (a.n_groups <= 0) return 0;
  const size_t lds = (size_t)a.ntables * sizeof(HuffDevTable) + sizeof(HuffDevAux) + 64 + (size_t)a.waves_per_group * a.lanes * RING_PITCH;
  if (emit) hipLaunchKernelGGL(huffman_walk_kernel<true>, dim3(a.n_groups), dim3(64 * a.waves_per_group), lds, stream, a);
  else hipLaunchKernelGGL(huffman_walk_kernel<false>, dim3(a.n_groups), dim3(64 * a.waves_per_group), lds, stream, a);
  return (int)hipGetLastError();
}

int launch_huffman_scan(const HuffScanArgs &a, hipStream_t stream)
{
  if (a.n_groups <= 0) return 0;
  const size_t lds = (size_t)a.ntables * sizeof(HuffDevTable) + sizeof(HuffDevAux) + (size_t)a.waves_per_group * a.lanes * LANE_LDS;
  hipLaunchKernelGGL(huffman_scan_kernel, dim3(a.n_groups), dim3(64 * a.waves_per_group), lds, stream, a);
  return (int)hipGetLastError();
}

} // namespace mij
