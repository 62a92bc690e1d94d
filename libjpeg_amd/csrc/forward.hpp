// forward.hpp -- argument block of the encoder-direction kernel (forward.hip); internal to libmijpeg.so.
#ifndef MIJ_FORWARD_HPP
#define MIJ_FORWARD_HPP
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace mij {

struct ForwardArgs {
  const uint8_t *pixels;          // interleaved 8-bit samples, ncomp per pixel
  int64_t pixel_frame_stride;     // bytes
  int64_t pixel_row_stride;       // bytes
  int16_t *coef;                  // coefficient store, the decoder's layout
  int64_t coef_frame_stride;      // int16 units
  int32_t width, height, ncomp, ycbcr, frames;
  // interior blocks -- whole blocks inside the picture: columns < fast_nbx, rows < fast_nby -- of components with subsampling
  // factors 1 or 2 go through fdct_interior_kernel when the lines can be read as dwords (RGB -> YCbCr frames only)
  int32_t fast[4], fast_nbx[4], fast_nby[4];
  // 4:2:0 frames: the whole 128 x 128 tiles go through fdct420_tile_kernel (blocks with column < tile_nbx and row < tile_nby);
  // the interior kernels then only take the whole blocks to the right of and below the tiles
  int32_t tiled420, tile_nbx[4], tile_nby[4];
  int32_t subx[4], suby[4];       // subsampling factors per component
  int32_t bw[4], bh[4];           // plane size in blocks (MCU padded)
  int32_t nbx[4], nby[4];         // blocks that cover samples: ceil(ceil(W / subx) / 8), ...
  int64_t coef_off[4];            // plane offsets (int16 units)
  uint32_t first_block[5];        // prefix sums of bw * bh over the components: block index -> component
  int32_t invq[4][64];            // quantiser multipliers LONG(FLOAT(1 << 30) / delta + 0.5), per component, natural order
};

int launch_forward(const ForwardArgs &a, hipStream_t stream);

} // namespace mij
#endif
