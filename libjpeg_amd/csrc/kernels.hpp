// kernels.hpp -- argument blocks and launch entry points of kernels.hip (internal to libmijpeg.so).
#ifndef MIJ_KERNELS_HPP
#define MIJ_KERNELS_HPP

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace mij {

constexpr int MAXC = 4;

// A component's dequantisation operands as the kernels take them: 64 deltas << 4 (natural order, dct/idct.cpp:98-109) and,
// behind them, the same deltas (not shifted) packed in pairs of 16 bits for the transforms' first pass in 16-bit arithmetic
// (dequant_idct16 in kernels.hip): per coefficient row k the pairs (0,4) (1,5) (2,6) (3,7), then for rows 0..3 the pairs
// (0,2) (1,3) of the pruned pass.
constexpr int QROW_PACKED = 64, QROW = 64 + 32 + 8;
inline void fill_deltas(int32_t (&dst)[QROW], const uint16_t *delta)
{
  for (int i = 0; i < 64; i++) dst[i] = (int32_t)delta[i] << 4;
  for (int k = 0; k < 8; k++)
    for (int j = 0; j < 4; j++) dst[QROW_PACKED + k * 4 + j] = (int32_t)((uint32_t)delta[k * 8 + j] | ((uint32_t)delta[k * 8 + j + 4] << 16));
  for (int k = 0; k < 4; k++)
    for (int j = 0; j < 2; j++) dst[QROW_PACKED + 32 + k * 2 + j] = (int32_t)((uint32_t)delta[k * 8 + j] | ((uint32_t)delta[k * 8 + j + 2] << 16));
}

// fused 4:2:0 (Y 1x1, Cb/Cr 2x2 subsampled).  Passed by value: the quantiser tables travel in the
// kernarg segment and are read with scalar loads.
struct Fused420Args {
  const int16_t *coef;        // frame 0; planes at off_y / off_cb / off_cr (int16 units)
  int64_t coef_frame_stride;  // int16 units
  int64_t off_y, off_cb, off_cr;
  uint8_t *out;               // interleaved RGB
  int64_t out_frame_stride;   // bytes
  int64_t row_stride;         // bytes
  int32_t width, height;
  int32_t bw_y, bh_y, bw_c, bh_c; // coefficient plane sizes in blocks (MCU padded)
  int32_t cw, ch;                 // valid chroma samples: ceil(W/2), ceil(H/2)
  int32_t tiles_x, tiles_y, frames;
  int32_t q[3][QROW];             // per component (Y, Cb, Cr): fill_deltas
  const int32_t *qdev;            // or null: per-frame tables in device memory, [frames][4][64] deltas << 4 (replace q)
  uint32_t magic_tx, magic_ty;    // set by the launchers: floor(2^32 / tiles_x) + 1 and the same for tiles_y, or 0 (kernels.hip tile_position)
};

// fused JPEG XT profile C (8-bit 4:2:0 legacy frame + 12-bit 4:4:4 residual frame, see fusedxt420_kernel)
struct FusedXtExtra {
  int64_t off_r[3];           // residual planes (int16 units from the frame's coefficient base)
  int32_t bw_r, bh_r;         // residual planes in blocks (4:4:4, padded to 8 only: may be narrower than the luma plane)
  int32_t rq[3][64];          // residual deltas << 4
  const int32_t *ltable;      // device: [3][256]
  int32_t rtrafo_ycbcr, is_float, out_max, out_shift;
  int32_t rprecision;         // residual precision incl. hidden bits: 12 -> fusedxt420_kernel (int16 residual coefficients);
                              // 13..16 -> fusedxtw420_kernel (int32 coefficients, two int16 slots each; bw_r in blocks of 256 bytes)
};
struct FusedXtArgs {
  Fused420Args base;          // legacy frame, geometry; out = 16-bit samples, strides in bytes
  FusedXtExtra ext;
  int luma_fits16;            // host: the legacy luma plane's range check admits int16 samples (fusedxtw420_kernel<true>)
};

// generic path: any sampling / component count / precision, two kernels with int32 sample planes in between.
// "Planes" are the component planes of the legacy codestream, followed -- for JPEG XT -- by those of the
// residual codestream (3 + 3).
constexpr int MAXP = 6;
struct GenericArgs {
  const int16_t *coef;
  int64_t coef_frame_stride;
  int64_t coef_off[MAXP];
  int32_t *samples;            // workspace: per frame, per plane (bw*8) x (bh*8) int32
  int64_t sample_frame_stride; // int32 units
  int64_t sample_off[MAXP];
  uint8_t *out;
  int64_t out_frame_stride, row_stride;
  int32_t width, height, ncomp, ycbcr, frames, nplanes;
  int32_t bw[MAXP], bh[MAXP], cw[MAXP], ch[MAXP], subx[MAXP], suby[MAXP];
  int32_t dcoff[MAXP];         // level shift of the plane's transform: 2^(P-1) << 7 (dct/idct.cpp:231)
  int32_t q[MAXP][QROW];       // fill_deltas
  // output stage
  int32_t sample_bytes;        // 1: 8-bit samples, 2: 16-bit samples
  int32_t maxval;              // 2^P - 1: clamp of the integer output
  int32_t dcshift;             // 2^(P-1) << 4: chroma level shift seen by the colour transformation
  // JPEG XT profile C merge (colortrafo/ycbcrtrafo.cpp:750-955)
  int32_t xt, rtrafo_ycbcr, out_shift, out_max, is_float, rprecision;
  int32_t xt_no_residual;      // mijpeg_xt_params::no_residual: the residual chain's result is the output shift
  int32_t xt_rct;              // R transformation = RCT (lossless coding): Q tables, reversible transformation with wrap-around
  int32_t xt_noclamp;          // output without clamping: wrap-around (integers) or sign conversion alone (half float codes)
  int32_t xt_rbits;            // fractional bits of the residual path (4, 1 with the RCT, 0 identity + lossless)
  int32_t legacy32;            // legacy colour stage may run in 32 bits (8-bit frame that passed the range check)
  int32_t ltable_entries;      // entries per L table: 256 << hidden bits of the legacy frame
  int32_t wide_first, wide_count; // planes [wide_first, wide_first + wide_count) hold int32 coefficients at coef_off (in
                                  // int16 units) and are transformed by idct_planes_wide_kernel (IDCT<4,QUAD>) ...
  int32_t narrow;                 // the sample planes may hold int16 (8-bit frame, every sample times 16 incl. its level shift inside
                                  // 16 bits: range_max < 7600): half the bytes between the two kernels
  int32_t wide_long;              // ... or, set, by idct_planes_long_kernel (IDCT<0,LONG>): frames with info.coef_wide
  const int32_t *ltable;       // device: L lookup tables [3][ltable_entries]
  const int32_t *qdev;         // or null: per-frame tables in device memory, [frames][4][64] deltas << 4 (replace q; not for JPEG XT)
  // JPEG XT beyond the default subset (mijpeg_xt_params.general): literal 64-bit merge with matrices and table gathers
  int32_t xt_general;
  int32_t rbypass, rnoise;     // RDCT box: residual planes are dequantised without a DCT (bypass_planes_kernel)
  int32_t rquant63[3];         // ... with delta[63] << 4 of the residual component
  int32_t rdcshift;            // ... and the level shift 2^(Pr-1), NOT scaled (control/residualblockhelper.cpp:196, 225)
  int32_t lmat[9], rmat[9], cmat[9];
  const int32_t *qlut[3];      // device, 2^(Pr + 4) entries each, or null = identity
  const int32_t *r2lut[3];     // device, 2^20 entries each, or null = identity
  // Rectangle requests whose result is not the plain picture (request_model.hpp): the first kernel transforms, for block
  // row g of plane p, the coefficient row rowmap[p * rowmap_stride + g] (-1: the samples are 0, dct/idct.cpp:336-338); the
  // second kernel works on lines [y_base, y_base + grid.y), reads subsampled planes inside their upsampler's window
  // [wstart, wlimit) and displaces them in the first row / column of blocks of the request (corner req_x0, req_y0).
  const int32_t *rowmap;       // device, or null
  int32_t rowmap_stride;
  int32_t request;             // 1: window / displacement semantics below are in force (LAYOUT_ANY instances only)
  int32_t req_x0, req_y0, y_base, y_count; // y_count: lines the second kernel works on (0: the whole frame)
  int32_t wstart[MAXP], wlimit[MAXP];
  // DNL frames (mijpeg_info::dnl): block rows of plane p from zero_from[p] on were never created -- the reference reads NULL
  // there and transforms it to sample value 0 (dct/idct.cpp:336-338); 0 = every row is there
  int32_t zero_from[MAXP];
  // fused tile kernel (launch_fused_tile fills these): tile size in pixels (whole MCUs) and tile grid
  int32_t tile_w, tile_h, tiles_x, tiles_y;
};

int launch_fused420(const Fused420Args &a, bool fast, hipStream_t stream);
int launch_fused1_12(const Fused420Args &a, hipStream_t stream);   // 12-bit single component frames inside the range gate
int launch_fused420_12(const Fused420Args &a, bool narrow, hipStream_t stream); // 12-bit frames inside the range gates, 16-bit samples out; narrow: the colour sums fit 32 bits (colour12)
int launch_fused420p(const Fused420Args &a, bool dot2, hipStream_t stream); // FAST only, chroma samples within int16 filter range; dot2: the second pass in 16 bits too (range_max <= 1476)
int launch_fused444(const Fused420Args &a, hipStream_t stream); // same argument block; all planes bw_y x bh_y
int launch_fused422_12(const Fused420Args &a, bool narrow, hipStream_t stream); // 12-bit 4:2:2 frames inside the range gates
int launch_fused444_12(const Fused420Args &a, bool narrow, hipStream_t stream); // 12-bit 4:4:4 frames inside the range gates, 16-bit samples out
int launch_fused1(const Fused420Args &a, hipStream_t stream);   // single component: plane off_y, bw_y x bh_y blocks, one byte per pixel
int launch_fused440(const Fused420Args &a, bool wide, hipStream_t stream);
int launch_fused411(const Fused420Args &a, hipStream_t stream); // same argument block; chroma planes bw_c x bh_y, cw = ceil(W/4), ch = H // same argument block; chroma planes bw_y x bh_c, cw = W, ch = ceil(H/2)
int launch_fused422(const Fused420Args &a, bool wide, hipStream_t stream); // wide: 32-bit filters for chroma ranges between the packed gate and 8190 // same argument block; chroma planes bw_c x bh_y, cw = ceil(W/2), ch = H
int launch_fusedxt420(const FusedXtArgs &x, hipStream_t stream);
int launch_generic(const GenericArgs &a, bool fast, hipStream_t stream);
// The same frames in one pass through LDS (plain JPEG: no residual planes, int16 coefficients, tables by value): any sampling
// layout, 1..4 components, 8 or 12 bit.  Uses the plane description of GenericArgs; no workspace.
int launch_fused_tile(const GenericArgs &a, bool fast, hipStream_t stream);
// three or four components, all 1 x 1, 8 bit, fast arithmetic, no colour transformation (CMYK, RGB stored as such): one lane per
// block position, no LDS round trip.  Uses the plane description of GenericArgs; no workspace.
int launch_fused_flat(const GenericArgs &a, hipStream_t stream);
int launch_expand_deltas(const uint16_t *in, int32_t *out, int frames, hipStream_t stream); // u16 [frames][4][64] -> int32 << 4

// Rectangle of the reconstructed interleaved frame -> bitmaps in DEVICE memory described like the reference's
// ImageBitMap (interface/imagebitmap.hpp): per component the address of canvas pixel (0,0) and the two strides.
struct ScatterArgs {
  const uint8_t *src;         // interleaved frame, ncomp * sample_bytes bytes per pixel
  int64_t src_row;            // bytes per line of src
  int32_t ncomp, sample_bytes;
  int32_t x0, y0, w, h;       // rectangle (already clipped to the frame)
  int32_t c0, c1;             // component range, inclusive
  uint8_t *dst[4];
  int32_t bytes_per_pixel[4];
  int64_t bytes_per_row[4];
};
int launch_scatter_rect(const ScatterArgs &a, hipStream_t stream);

} // namespace mij
#endif
