// forward.hip -- encoder direction of the block pipeline: pixels -> quantised coefficient planes.
//
// What the reference does per 8x8 block in front of its entropy coder (control/blockbitmaprequester.cpp:505-576,
// 708-846): forward L transformation RGB -> YCbCr at FIX_BITS 13 into samples with COLOR_BITS = 4 fractional bits
// (colortrafo/ycbcrtrafo.cpp:85-242, matrix colortransformerfactory.cpp:177-183), box downsampling of subsampled
// components (upsampling/downsampler.cpp:70-139; right edge mirrored, lines below the image missing:
// downsamplerbase.cpp:124-155), forward DCT and quantisation (dct/idct.cpp:114-222, dct/idct.hpp:90-111).
// Everything is integer arithmetic; the results are the reference's bits (oracle/jpeg_oracle.c: oj_forward, pinned
// against the coefficients the reference encoder writes).
//
// One lane, one coefficient block; no intermediate planes: the only HBM traffic is the image and the coefficients
// (128-byte stores in the decoder's plane layout).  Three kernels share the blocks of a frame:
//   fdct420_tile_kernel    4:2:0, 128 x 128 tiles wholly inside the picture: luma lanes read their pixels once and leave the
//                          box-filtered chroma samples in LDS for the lanes that transform the chroma blocks
//   fdct_interior_kernel   other layouts with subsampling factors 1 or 2: whole blocks inside the picture, pixels read as
//                          dwords in batches, only the block's own component converted
//   fdct_blocks_kernel     everything else (edges with pre-fill / mirror / missing lines, 3x and 4x factors, grey, identity
//                          transformation): per-pixel gather
// The integer work is in 32-bit wrapping arithmetic like the reference's LONG; the quantiser is its 64-bit
// multiply-and-shift.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "forward.hpp"

namespace mij {

#define F9(x) ((int)((x) * 512.0 + 0.5)) // TO_FIX, dct/idct.cpp:65

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int wmul(int a, int c) { return (int)((unsigned)a * (unsigned)c); }

// Quantize, dct/idct.hpp:100-103 (no dead zone): (n * q + (n > 0) + 2^45) >> 46
__device__ __forceinline__ int quantize(int n, int q)
{
  const long long p = (long long)n * (long long)q + (long long)((unsigned)(-n) >> 31) + (1ll << 45);
  return (int)(p >> 46);
}

// One 8-point forward transform, dct/idct.cpp:126-169; outputs before any shift: o[0], o[4] plain sums, the others
// scaled by 2^9 (FIX_BITS)
__device__ __forceinline__ void fdct_1d(const int (&s)[8], int (&o)[8])
{
  int tmp0 = wadd(s[0], s[7]), tmp1 = wadd(s[1], s[6]), tmp2 = wadd(s[2], s[5]), tmp3 = wadd(s[3], s[4]);
  int tmp10 = wadd(tmp0, tmp3), tmp12 = wsub(tmp0, tmp3), tmp11 = wadd(tmp1, tmp2), tmp13 = wsub(tmp1, tmp2);
  tmp0 = wsub(s[0], s[7]); tmp1 = wsub(s[1], s[6]); tmp2 = wsub(s[2], s[5]); tmp3 = wsub(s[3], s[4]);
  o[0] = wadd(tmp10, tmp11);
  o[4] = wsub(tmp10, tmp11);
  int z1 = wmul(wadd(tmp12, tmp13), F9(0.541196100));
  o[2] = wadd(z1, wmul(tmp12, F9(0.765366865)));
  o[6] = wadd(z1, wmul(tmp13, -F9(1.847759065)));
  tmp10 = wadd(tmp0, tmp3); tmp11 = wadd(tmp1, tmp2); tmp12 = wadd(tmp0, tmp2); tmp13 = wadd(tmp1, tmp3);
  z1 = wmul(wadd(tmp12, tmp13), F9(1.175875602));
  const int tt0 = wmul(tmp0, F9(1.501321110)), tt1 = wmul(tmp1, F9(3.072711026)), tt2 = wmul(tmp2, F9(2.053119869)), tt3 = wmul(tmp3, F9(0.298631336));
  const int tt10 = wmul(tmp10, -F9(0.899976223)), tt11 = wmul(tmp11, -F9(2.562915447));
  const int tt12 = wadd(wmul(tmp12, -F9(0.390180644)), z1), tt13 = wadd(wmul(tmp13, -F9(1.961570560)), z1);
  o[1] = wadd(wadd(tt0, tt10), tt12);
  o[3] = wadd(wadd(tt1, tt11), tt13);
  o[5] = wadd(wadd(tt2, tt11), tt12);
  o[7] = wadd(wadd(tt3, tt10), tt13);
}

// component `c` of the forward L transformation of one pixel (ycbcrtrafo.cpp:176-199): FIX_TO_COLOR, clamp
__device__ __forceinline__ int ycc_component(int c, int r, int g, int b)
{
  const int dc = (128 << 13) + 256;
  int v;
  if (c == 0) v = (r * 2449 + g * 4809 + b * 934 + 256) >> 9;
  else if (c == 1) v = (r * -1382 + g * -2714 + b * 4096 + dc) >> 9;
  else v = (r * 4096 + g * -3430 + b * -666 + dc) >> 9;
  return min(max(v, 0), (256 << 4) - 1);
}

// forward transform of one block of samples, quantisation, 128-byte store (idct.cpp:125-170 columns, :174-218 rows)
__device__ __forceinline__ void transform_and_store(const int (&blk)[64], const int *__restrict__ invq, int16_t *dst)
{
  // pass over columns (idct.cpp:125-170), then rows with quantisation (:174-218)
  int t[64];
#pragma unroll
  for (int col = 0; col < 8; col++) {
    const int s[8] = {blk[col], blk[8 + col], blk[16 + col], blk[24 + col], blk[32 + col], blk[40 + col], blk[48 + col], blk[56 + col]};
    int o[8];
    fdct_1d(s, o);
    t[col] = o[0];
    t[32 + col] = o[4];
#pragma unroll
    for (int k = 1; k < 8; k++)
      if (k != 4) t[k * 8 + col] = wadd(o[k], 256) >> 9; // FIXED_TO_INTERMEDIATE
  }
  const int dcoffset = 128 << 10; // 2^(P-1) << (preshift + 3 + 3)
  unsigned packed[32];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int s[8] = {t[r * 8], t[r * 8 + 1], t[r * 8 + 2], t[r * 8 + 3], t[r * 8 + 4], t[r * 8 + 5], t[r * 8 + 6], t[r * 8 + 7]};
    int o[8];
    fdct_1d(s, o);
    o[0] = (int)((unsigned)wsub(o[0], r == 0 ? dcoffset : 0) << 9);
    o[4] = (int)((unsigned)o[4] << 9);
    int qv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) qv[k] = quantize(o[k], invq[r * 8 + k]);
#pragma unroll
    for (int k = 0; k < 4; k++) packed[r * 4 + k] = ((unsigned)qv[2 * k] & 0xffffu) | ((unsigned)qv[2 * k + 1] << 16);
  }
  u32x4 *d4 = reinterpret_cast<u32x4 *>(dst);
#pragma unroll
  for (int i = 0; i < 8; i++) d4[i] = u32x4{packed[4 * i], packed[4 * i + 1], packed[4 * i + 2], packed[4 * i + 3]};
}

// Interior blocks of RGB -> YCbCr frames with subsampling factors 1 or 2: the block's SX*8 x SY*8 pixels are read as
// dwords (rows of 24 * SX bytes; the host checks that lines start dword-aligned), the bytes picked apart in registers,
// only the block's own component computed, the box filter's division a shift (the sums are not negative).
template <int SX, int SY>
__device__ __forceinline__ void gather_block_fast(const uint8_t *img, int64_t row_stride, int x0, int y0, int c, int (&blk)[64])
{
  constexpr int ND = 6 * SX;                 // dwords per line of the block
  constexpr int RB = 8 / (SX * SY);          // output rows per batch: 48 dwords in flight at a time, one memory round trip each
#pragma unroll
  for (int r0 = 0; r0 < 8; r0 += RB) {
    unsigned dw[RB * SY][ND];
#pragma unroll
    for (int l = 0; l < RB * SY; l++) {
      const unsigned *line = reinterpret_cast<const unsigned *>(img + (int64_t)(y0 + r0 * SY + l) * row_stride + (int64_t)x0 * 3);
#pragma unroll
      for (int i = 0; i < ND; i++) dw[l][i] = line[i];
    }
    __builtin_amdgcn_sched_barrier(0); // the loads of one batch together, those of the next not before this one is used up
#pragma unroll
    for (int rr = 0; rr < RB; rr++) {
      int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int ly = 0; ly < SY; ly++)
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
          for (int k = 0; k < SX; k++) {
            const int j = 3 * (i * SX + k); // byte of the pixel's R inside the line
            const unsigned *d = dw[rr * SY + ly];
            const int r8 = (int)((d[j >> 2] >> (8 * (j & 3))) & 0xffu), g8 = (int)((d[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xffu),
                      b8 = (int)((d[(j + 2) >> 2] >> (8 * ((j + 2) & 3))) & 0xffu);
            acc[i] += ycc_component(c, r8, g8, b8);
          }
#pragma unroll
      for (int i = 0; i < 8; i++) blk[(r0 + rr) * 8 + i] = acc[i] >> (SX * SY == 4 ? 2 : SX * SY == 2 ? 1 : 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256) void fdct_blocks_kernel(const ForwardArgs a)
{
  const unsigned per_frame = a.first_block[a.ncomp];
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned frame = blockIdx.y;
  if (gid >= per_frame) return;
  int c = 0;
  while (c + 1 < a.ncomp && gid >= a.first_block[c + 1]) c++;
  const unsigned bi = gid - a.first_block[c];
  const int by = (int)(bi / (unsigned)a.bw[c]), bx = (int)(bi - (unsigned)by * (unsigned)a.bw[c]);
  int16_t *dst = a.coef + (int64_t)frame * a.coef_frame_stride + a.coef_off[c] + (int64_t)bi * 64;
  if (bx >= a.nbx[c] || by >= a.nby[c]) { // MCU padding: no samples; left zero for the entropy coder to fill
    u32x4 *d4 = reinterpret_cast<u32x4 *>(dst);
#pragma unroll
    for (int i = 0; i < 8; i++) d4[i] = u32x4{0, 0, 0, 0};
    return;
  }
  const int W = a.width, H = a.height, nc = a.ncomp, sx = a.subx[c], sy = a.suby[c];
  const uint8_t *img = a.pixels + (int64_t)frame * a.pixel_frame_stride;
  const bool ycc = nc == 3 && a.ycbcr;
  auto sample = [&](int x, int y) -> int { // component c of pixel (x, y), x < W, y < H, with COLOR_BITS fractional bits
    const uint8_t *p = img + (int64_t)y * a.pixel_row_stride + (int64_t)x * nc;
    if (ycc) return ycc_component(c, p[0], p[1], p[2]);
    return (int)p[c] << 4;
  };
  int blk[64];
  // interior blocks of frames the fast kernels cover are theirs
  if (a.fast[c] && bx < a.fast_nbx[c] && by < a.fast_nby[c]) return;
  if (sx == 1 && sy == 1) {
    // partial blocks are pre-filled with the level shift (ycbcrtrafo.cpp:100-113)
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int y = by * 8 + r;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int x = bx * 8 + i;
        blk[r * 8 + i] = (x < W && y < H) ? sample(x, y) : (128 << 4);
      }
    }
  } else {
    // box filter over the lines that exist; beyond the right edge the line is the mirror image of its end
    // (downsamplerbase.cpp:141-145), a row of the block without any line stays zero (downsampler.cpp:92-95)
    const int ofs = (bx * sx) << 3;
    int y = (by * sy) << 3;
#pragma unroll
    for (int r = 0; r < 8; r++) { // unrolled: blk stays in registers
      int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int lines = 0;
      while (lines < sy && y < H) {
        for (int i = 0; i < 8; i++)
          for (int k = 0; k < sx; k++) {
            int x = ofs + i * sx + k;
            if (x >= W) { const int m = x - W; x = W > m ? W - 1 - m : 0; }
            acc[i] += sample(x, y);
          }
        lines++;
        y++;
      }
      const int norm = lines * sx;
#pragma unroll
      for (int i = 0; i < 8; i++) blk[r * 8 + i] = norm > 1 ? acc[i] / norm : acc[i];
    }
  }
  transform_and_store(blk, a.invq[c], dst);
}

// 4:2:0, tiles of 128 x 128 pixels that lie wholly inside the picture: one workgroup of 256 lanes per tile.  Every lane reads
// the 8 x 8 pixels of ONE luma block once (48 dwords, one memory round trip), computes Y, Cb and Cr of each, transforms the
// luma block, and leaves the 4 x 4 box-filtered chroma samples of its pixels (sums of 2 x 2, >> 2) in LDS; after a
// barrier 128 lanes pick up the 64 + 64 chroma blocks of the tile and transform them.  Compared with the per-component
// kernels no pixel is fetched or unpacked twice.  grid (tiles_x * tiles_y, frames)
__global__ __launch_bounds__(256, 2) void fdct420_tile_kernel(const ForwardArgs a)
{
  __shared__ short chroma[2][64 * 64]; // [Cb, Cr][64 lines of 64 samples]
  const int tiles_x = a.width >> 7;
  const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
  const unsigned frame = blockIdx.y;
  const uint8_t *img = a.pixels + (int64_t)frame * a.pixel_frame_stride;
  int16_t *coef = a.coef + (int64_t)frame * a.coef_frame_stride;
  const int lane = threadIdx.x;
  const int lbx = lane & 15, lby = lane >> 4; // luma block inside the tile
  const int x0 = tx * 128 + lbx * 8, y0 = ty * 128 + lby * 8;
  {
    unsigned dw[8][6];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const unsigned *line = reinterpret_cast<const unsigned *>(img + (int64_t)(y0 + r) * a.pixel_row_stride + (int64_t)x0 * 3);
#pragma unroll
      for (int i = 0; i < 6; i++) dw[r][i] = line[i];
    }
    int blk[64];
    short *cb = chroma[0] + (lby * 4) * 64 + lbx * 4, *cr = chroma[1] + (lby * 4) * 64 + lbx * 4;
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
      int sb[4] = {0, 0, 0, 0}, sr[4] = {0, 0, 0, 0};
#pragma unroll
      for (int rr = 0; rr < 2; rr++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int j = 3 * i;
          const unsigned *d = dw[r + rr];
          const int r8 = (int)((d[j >> 2] >> (8 * (j & 3))) & 0xffu), g8 = (int)((d[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xffu),
                    b8 = (int)((d[(j + 2) >> 2] >> (8 * ((j + 2) & 3))) & 0xffu);
          blk[(r + rr) * 8 + i] = ycc_component(0, r8, g8, b8);
          sb[i >> 1] += ycc_component(1, r8, g8, b8);
          sr[i >> 1] += ycc_component(2, r8, g8, b8);
        }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        cb[(r >> 1) * 64 + i] = (short)(sb[i] >> 2);
        cr[(r >> 1) * 64 + i] = (short)(sr[i] >> 2);
      }
    }
    transform_and_store(blk, a.invq[0], coef + a.coef_off[0] + ((int64_t)(y0 >> 3) * a.bw[0] + (x0 >> 3)) * 64);
  }
  __syncthreads();
  if (lane < 128) {
    const int c = 1 + (lane >> 6), n = lane & 63, cbx = n & 7, cby = n >> 3;
    const short *src = chroma[c - 1] + (cby * 8) * 64 + cbx * 8;
    int blk[64];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) blk[r * 8 + i] = src[r * 64 + i];
    transform_and_store(blk, a.invq[c], coef + a.coef_off[c] + ((int64_t)(ty * 8 + cby) * a.bw[c] + (tx * 8 + cbx)) * 64);
  }
}

// the interior blocks of component c: grid (blocks of 256 lanes over fast_nbx * fast_nby, frames)
template <int SX, int SY>
__global__ __launch_bounds__(256, SX * SY == 4 ? 2 : 3) void fdct_interior_kernel(const ForwardArgs a, int c)
{
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, frame = blockIdx.y;
  const unsigned n = (unsigned)a.fast_nbx[c] * (unsigned)a.fast_nby[c];
  if (gid >= n) return;
  const int by = (int)(gid / (unsigned)a.fast_nbx[c]), bx = (int)(gid - (unsigned)by * (unsigned)a.fast_nbx[c]);
  if (a.tiled420 && bx < a.tile_nbx[c] && by < a.tile_nby[c]) return; // the tile kernel's
  int16_t *dst = a.coef + (int64_t)frame * a.coef_frame_stride + a.coef_off[c] + ((int64_t)by * a.bw[c] + bx) * 64;
  const uint8_t *img = a.pixels + (int64_t)frame * a.pixel_frame_stride;
  int blk[64];
  gather_block_fast<SX, SY>(img, a.pixel_row_stride, (bx * SX) << 3, (by * SY) << 3, c, blk);
  transform_and_store(blk, a.invq[c], dst);
}

int launch_forward(const ForwardArgs &a, hipStream_t stream)
{
  const unsigned per_frame = a.first_block[a.ncomp];
  if (per_frame == 0 || a.frames < 1) return 0;
  if (a.tiled420) hipLaunchKernelGGL(fdct420_tile_kernel, dim3((unsigned)(a.width >> 7) * (unsigned)(a.height >> 7), a.frames), dim3(256), 0, stream, a);
  for (int c = 0; c < a.ncomp; c++) {
    if (!a.fast[c]) continue;
    const unsigned n = (unsigned)a.fast_nbx[c] * (unsigned)a.fast_nby[c];
    if (n == 0) continue;
    const dim3 grid((n + 255) / 256, a.frames);
    const int key = a.subx[c] * 4 + a.suby[c];
    if (key == 5) hipLaunchKernelGGL((fdct_interior_kernel<1, 1>), grid, dim3(256), 0, stream, a, c);
    else if (key == 10) hipLaunchKernelGGL((fdct_interior_kernel<2, 2>), grid, dim3(256), 0, stream, a, c);
    else if (key == 9) hipLaunchKernelGGL((fdct_interior_kernel<2, 1>), grid, dim3(256), 0, stream, a, c);
    else hipLaunchKernelGGL((fdct_interior_kernel<1, 2>), grid, dim3(256), 0, stream, a, c);
  }
  hipLaunchKernelGGL(fdct_blocks_kernel, dim3((per_frame + 255) / 256, a.frames), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

} // namespace mij
