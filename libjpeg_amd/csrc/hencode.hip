// hencode.hip -- baseline Huffman entropy coder on the device, so that the coefficients the forward kernels produce
// never leave HBM: what SequentialScan::WriteMCU / EncodeBlock (codestream/sequentialscan.cpp:430-676) and the byte
// stuffing bit writer (io/bitstream.hpp) do, restated as data-parallel passes.  Same stream as the host coder of
// encoder.cpp, byte for byte.
//
//   count     one lane per block in scan order: length of the block's code (DC difference against the previous block
//             of the component that covers samples -- MCU padding blocks are coded as "same DC, no AC" --, AC run/size
//             symbols with ZRL and EOB); optionally the symbol statistics for optimised tables
//   scan      exclusive prefix sums (bit position of every block), bytes per restart interval, their prefix sums
//   emit      one lane per block: the code words, OR-ed into the plain stream at the block's bit position (blocks share
//             words, hence atomicOr on a zeroed buffer); the last block of an interval pads with one-bits
//   stuff     0xFF bytes per 64-byte chunk, prefix sums, then every chunk is copied to its place with a zero byte behind
//             every 0xFF and the RSTn markers in front of the intervals
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hencode.hpp"

namespace mij {

namespace {

// zig-zag position -> natural index (dct/dct.cpp:57-74)
constexpr uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int category(int v)
{
  const unsigned a = (unsigned)(v < 0 ? -v : v);
  return a ? 32 - __builtin_clz(a) : 0;
}

struct BlockRef {
  const int16_t *blk; // null: MCU padding
  int pred, table;
};

// Block s of the scan: its coefficients (or null) and the DC predictor in front of it
__device__ __forceinline__ BlockRef locate(const HencArgs &a, uint32_t s)
{
  BlockRef r;
  const int B = a.blocks_per_mcu;
  int m = (int)(s / (uint32_t)B);
  const int j = (int)(s - (uint32_t)m * (uint32_t)B);
  const int c = a.blk_comp[j];
  int bx = a.blk_bx[j], by = a.blk_by[j];
  const int hs = a.hs[c], vs = a.vs[c];
  const int16_t *plane = a.coef + a.coef_off[c];
  auto at = [&](int mm, int yy, int xx) -> const int16_t * {
    const int my = mm / a.mcus_x, mx = mm - my * a.mcus_x;
    const int gx = mx * hs + xx, gy = my * vs + yy;
    return (gx < a.nbx[c] && gy < a.nby[c]) ? plane + ((int64_t)gy * a.bw[c] + gx) * 64 : nullptr;
  };
  r.table = c ? 1 : 0;
  r.blk = at(m, by, bx);
  // predictor: DC of the last block of the component that covers samples and is coded before this one in the interval
  const int first = (m / a.ri) * a.ri;
  r.pred = 0;
  for (;;) {
    if (bx > 0) bx--;
    else if (by > 0) { by--; bx = hs - 1; }
    else {
      if (m == first) break;
      m--;
      by = vs - 1;
      bx = hs - 1;
    }
    const int16_t *p = at(m, by, bx);
    if (p) { r.pred = p[0]; break; }
  }
  return r;
}

// Walks the symbols of a block in coding order: f_dc(category, difference), f_ac(symbol, value, category)
// (sequentialscan.cpp EncodeBlock: DC difference, then run/size symbols with ZRL for runs beyond 15 and EOB)
template <class FD, class FA>
__device__ __forceinline__ void walk_symbols(const BlockRef &r, FD f_dc, FA f_ac)
{
  const int diff = (r.blk ? (int)r.blk[0] : r.pred) - r.pred;
  f_dc(category(diff), diff);
  if (!r.blk) { f_ac(0, 0, 0); return; }
  unsigned wd[32]; // the block in registers
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u32x4 t = reinterpret_cast<const u32x4 *>(r.blk)[i];
    wd[4 * i] = t.x; wd[4 * i + 1] = t.y; wd[4 * i + 2] = t.z; wd[4 * i + 3] = t.w;
  }
  int run = 0;
#pragma unroll
  for (int k = 1; k < 64; k++) {
    const int nat = ZZ[k];
    const unsigned d = wd[nat >> 1];
    const int v = (nat & 1) ? (int)d >> 16 : (int)(short)(d & 0xffffu);
    if (v != 0) {
      while (run > 15) { f_ac(0xf0, 0, 0); run -= 16; }
      const int s = category(v);
      f_ac((run << 4) | s, v, s);
      run = 0;
    } else run++;
  }
  if (run > 0) f_ac(0, 0, 0);
}

__device__ __forceinline__ void load_tables(HencTables *lds, const HencTables *src)
{
  const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
  uint32_t *d = reinterpret_cast<uint32_t *>(lds);
  for (unsigned i = threadIdx.x; i < sizeof(HencTables) / 4; i += blockDim.x) d[i] = s[i];
  __syncthreads();
}

template <bool STATS>
__global__ __launch_bounds__(256) void henc_count_kernel(const HencArgs a)
{
  __shared__ HencTables tab;
  __shared__ uint32_t hist[STATS ? 4 * 256 : 1];
  load_tables(&tab, a.tables);
  if (STATS) {
    for (unsigned i = threadIdx.x; i < 4 * 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
  }
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < a.total_blocks) {
    const BlockRef r = locate(a, s);
    uint32_t bits = 0;
    const int t = r.table;
    walk_symbols(
        r,
        [&](int cat, int) {
          bits += tab.dc_len[t][cat] + (uint32_t)cat;
          if (STATS) atomicAdd(&hist[t * 256 + cat], 1u);
        },
        [&](int sym, int, int cat) {
          bits += tab.ac_len[t][sym] + (uint32_t)cat;
          if (STATS) atomicAdd(&hist[512 + t * 256 + sym], 1u);
        });
    a.bits[s] = bits;
  }
  if (STATS) {
    __syncthreads();
    for (unsigned i = threadIdx.x; i < 4 * 256; i += blockDim.x)
      if (hist[i]) atomicAdd(&a.hist[i], hist[i]);
  }
}

// bytes of every interval before stuffing: its bits rounded up
__global__ __launch_bounds__(256) void henc_interval_bytes_kernel(const HencArgs a)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_intervals) return;
  const uint64_t per = (uint64_t)a.ri * (uint64_t)a.blocks_per_mcu;
  const uint64_t b0 = (uint64_t)i * per, b1 = min((uint64_t)a.total_blocks, b0 + per);
  a.ibytes[i] = (uint32_t)((a.bitpos[b1] - a.bitpos[b0] + 7) >> 3);
}

// MSB-first bit writer into big-endian 32-bit words shared with the neighbouring blocks
struct WordWriter {
  uint32_t *out;
  uint64_t widx;
  uint32_t cur;
  int fill;
  __device__ __forceinline__ void open(uint32_t *words, uint64_t bitpos)
  {
    out = words;
    widx = bitpos >> 5;
    fill = (int)(bitpos & 31);
    cur = 0;
  }
  __device__ __forceinline__ void put(uint32_t bits, int len) // len <= 27
  {
    if (len == 0) return;
    bits &= (1u << len) - 1u;
    const int room = 32 - fill;
    if (len < room) {
      cur |= bits << (room - len);
      fill += len;
    } else {
      const int rest = len - room; // bits that go into the next word
      atomicOr(&out[widx], cur | (bits >> rest));
      widx++;
      cur = rest ? bits << (32 - rest) : 0u;
      fill = rest;
    }
  }
  __device__ __forceinline__ void close()
  {
    if (fill) atomicOr(&out[widx], cur);
  }
};

__global__ __launch_bounds__(256) void henc_emit_kernel(const HencArgs a)
{
  __shared__ HencTables tab;
  load_tables(&tab, a.tables);
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.total_blocks) return;
  const BlockRef r = locate(a, s);
  const uint64_t per = (uint64_t)a.ri * (uint64_t)a.blocks_per_mcu;
  const uint32_t interval = (uint32_t)(s / per);
  const uint64_t first = (uint64_t)interval * per;
  const uint64_t pos = a.istart[interval] * 8 + (a.bitpos[s] - a.bitpos[first]);
  WordWriter w;
  w.open(a.plain, pos);
  const int t = r.table;
  walk_symbols(
      r,
      [&](int cat, int diff) {
        w.put(tab.dc_code[t][cat], tab.dc_len[t][cat]);
        w.put((uint32_t)(diff < 0 ? diff - 1 : diff), cat);
      },
      [&](int sym, int v, int cat) {
        w.put(tab.ac_code[t][sym], tab.ac_len[t][sym]);
        w.put((uint32_t)(v < 0 ? v - 1 : v), cat);
      });
  // the last block of an interval pads its last byte with one-bits
  if (s + 1 == a.total_blocks || (uint64_t)(s + 1) == first + per) {
    const int used = (int)((a.bitpos[s + 1] - a.bitpos[first]) & 7);
    if (used) w.put((1u << (8 - used)) - 1u, 8 - used);
  }
  w.close();
}

__device__ __forceinline__ uint32_t plain_byte(const uint32_t *plain, uint64_t u) { return (plain[u >> 2] >> (24 - 8 * (int)(u & 3))) & 0xffu; }

constexpr int STUFF_CHUNK = HENC_STUFF_CHUNK;

__global__ __launch_bounds__(256) void henc_count_ff_kernel(const HencArgs a)
{
  const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t u0 = chunk * STUFF_CHUNK;
  if (u0 >= a.plain_bytes) return;
  const uint64_t u1 = min(a.plain_bytes, u0 + STUFF_CHUNK);
  uint32_t n = 0;
  for (uint64_t wi = u0 >> 2; wi < (u1 + 3) >> 2; wi++) { // whole words: the tail behind plain_bytes is zero
    const uint32_t x = a.plain[wi];
    if (x & ~(x + 0x01010101u) & 0x80808080u) // some byte is 0xFF
      n += ((x >> 24) == 0xffu) + (((x >> 16) & 0xffu) == 0xffu) + (((x >> 8) & 0xffu) == 0xffu) + ((x & 0xffu) == 0xffu);
  }
  a.ffcount[chunk] = n;
}

// Copies chunk by chunk: a zero byte behind every 0xFF, RSTn in front of every interval but the first
__global__ __launch_bounds__(256) void henc_stuff_kernel(const HencArgs a)
{
  const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t u0 = chunk * STUFF_CHUNK;
  if (u0 >= a.plain_bytes) return;
  const uint64_t u1 = min(a.plain_bytes, u0 + STUFF_CHUNK);
  // first interval whose start lies at or behind u0 (istart is ascending, istart[0] = 0)
  uint32_t lo = 0, hi = a.n_intervals;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.istart[mid] < u0) lo = mid + 1;
    else hi = mid;
  }
  uint32_t next = lo; // intervals [1, next) have put their markers in front of earlier bytes
  uint8_t *q = a.out + u0 + a.ffstart[chunk] + 2ull * (uint64_t)(next > 0 ? next - 1 : 0);
  for (uint64_t u = u0; u < u1; u++) {
    while (next < a.n_intervals && a.istart[next] == u) {
      if (next > 0) { *q++ = 0xff; *q++ = (uint8_t)(0xd0 + ((next - 1) & 7)); }
      next++;
    }
    const uint32_t b = plain_byte(a.plain, u);
    *q++ = (uint8_t)b;
    if (b == 0xff) *q++ = 0;
  }
}

// ---- exclusive prefix sums: tiles of 1024, recursively ----------------------------------------------------------
constexpr int SCAN_TILE = 1024;
template <class T>
__global__ __launch_bounds__(SCAN_TILE) void scan_tile_sums_kernel(const T *in, uint64_t *sums, uint32_t n)
{
  __shared__ uint64_t wave_tot[SCAN_TILE / 64];
  const uint32_t i = blockIdx.x * SCAN_TILE + threadIdx.x;
  uint64_t v = i < n ? (uint64_t)in[i] : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t t = 0;
    for (int k = 0; k < SCAN_TILE / 64; k++) t += wave_tot[k];
    sums[blockIdx.x] = t;
  }
}
// out[i] = offsets[tile] + sum of in[tile start .. i) ; out[n] = total (written by the thread behind the last element)
template <class T>
__global__ __launch_bounds__(SCAN_TILE) void scan_apply_kernel(const T *in, const uint64_t *offsets, uint64_t *out, uint32_t n)
{
  __shared__ uint64_t wave_tot[SCAN_TILE / 64];
  const uint32_t i = blockIdx.x * SCAN_TILE + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t mine = i < n ? (uint64_t)in[i] : 0;
  uint64_t v = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  if (lane == 63) wave_tot[wv] = v;
  __syncthreads();
  uint64_t before = offsets ? offsets[blockIdx.x] : 0;
  for (int k = 0; k < wv; k++) before += wave_tot[k];
  if (i <= n) out[i] = before + v - mine;
}

} // namespace

int exclusive_scan_u32(const uint32_t *in, uint64_t *out, uint32_t n, uint64_t *scratch, hipStream_t stream)
{
  // level 0: n elements (u32); level 1: tile sums (u64), scanned in place of `scratch`; level 2 if needed
  const uint32_t t1 = n / SCAN_TILE + 1; // tiles incl. the one that holds out[n]
  if (t1 == 1) {
    hipLaunchKernelGGL((scan_apply_kernel<uint32_t>), dim3(1), dim3(SCAN_TILE), 0, stream, in, (const uint64_t *)nullptr, out, n);
    return (int)hipGetLastError();
  }
  uint64_t *sums1 = scratch, *off1 = scratch + t1 + 1; // off1: t1 + 1 entries
  hipLaunchKernelGGL((scan_tile_sums_kernel<uint32_t>), dim3(t1), dim3(SCAN_TILE), 0, stream, in, sums1, n);
  const uint32_t t2 = t1 / SCAN_TILE + 1;
  if (t2 == 1) {
    hipLaunchKernelGGL((scan_apply_kernel<uint64_t>), dim3(1), dim3(SCAN_TILE), 0, stream, (const uint64_t *)sums1, (const uint64_t *)nullptr, off1, t1);
  } else {
    uint64_t *sums2 = off1 + t1 + 1, *off2 = sums2 + t2 + 1;
    if (t2 > SCAN_TILE) return (int)hipErrorInvalidValue; // more than 2^30 elements
    hipLaunchKernelGGL((scan_tile_sums_kernel<uint64_t>), dim3(t2), dim3(SCAN_TILE), 0, stream, (const uint64_t *)sums1, sums2, t1);
    hipLaunchKernelGGL((scan_apply_kernel<uint64_t>), dim3(1), dim3(SCAN_TILE), 0, stream, (const uint64_t *)sums2, (const uint64_t *)nullptr, off2, t2);
    hipLaunchKernelGGL((scan_apply_kernel<uint64_t>), dim3(t2), dim3(SCAN_TILE), 0, stream, (const uint64_t *)sums1, (const uint64_t *)off2, off1, t1);
  }
  hipLaunchKernelGGL((scan_apply_kernel<uint32_t>), dim3(t1), dim3(SCAN_TILE), 0, stream, in, (const uint64_t *)off1, out, n);
  return (int)hipGetLastError();
}

int henc_count(const HencArgs &a, bool statistics, hipStream_t stream)
{
  const dim3 grid((a.total_blocks + 255) / 256);
  if (statistics) hipLaunchKernelGGL(henc_count_kernel<true>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(henc_count_kernel<false>, grid, dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}
int henc_interval_bytes(const HencArgs &a, hipStream_t stream)
{
  hipLaunchKernelGGL(henc_interval_bytes_kernel, dim3((a.n_intervals + 255) / 256), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}
int henc_emit(const HencArgs &a, hipStream_t stream)
{
  hipLaunchKernelGGL(henc_emit_kernel, dim3((a.total_blocks + 255) / 256), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}
static inline uint32_t stuff_chunks(const HencArgs &a) { return (uint32_t)((a.plain_bytes + STUFF_CHUNK - 1) / STUFF_CHUNK); }
int henc_count_ff(const HencArgs &a, hipStream_t stream)
{
  hipLaunchKernelGGL(henc_count_ff_kernel, dim3((stuff_chunks(a) + 255) / 256), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}
int henc_stuff(const HencArgs &a, hipStream_t stream)
{
  hipLaunchKernelGGL(henc_stuff_kernel, dim3((stuff_chunks(a) + 255) / 256), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

} // namespace mij
