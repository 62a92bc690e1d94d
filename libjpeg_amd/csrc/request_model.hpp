// request_model.hpp -- the state JPEG::DisplayRectangle keeps between calls, as a planner for whole-frame hardware.
//
// The reference reconstructs a requested rectangle block by block while it walks per-component cursors over its lists
// of coefficient rows and feeds the line buffers of its upsamplers (BlockBitmapRequester::ReconstructRegion,
// control/blockbitmaprequester.cpp:1249-1272 with PullQData :1079-1112, PushReconstructedData :1151-1224,
// ReconstructUnsampled :1013-1074; UpsamplerBase::SetBufferedImageRegion / SetBufferedRegion / ExtendBufferedRegion,
// upsampling/upsamplerbase.cpp:138-259).  Nothing ever rewinds the cursors, so WHICH coefficient row ends up in which
// block row of the output depends on the calls made before:
//   * on the upsampling path every call advances the cursor of every component that has no upsampler, requested or not
//     (:1214-1223) -- the reference's own component-by-component loop (cmd/reconstruct.cpp:272-303) therefore gets planes of
//     zeros for all but the first unsubsampled component of a frame that also has a subsampled one: a cursor behind the last
//     row reads NULL and IDCT::InverseTransformBlock turns NULL into samples of value 0 (dct/idct.cpp:336-338);
//   * PullQData advances only the requested subsampled components, by the block rows their upsampler did not hold yet;
//   * ReconstructUnsampled advances only the requested components (:1066-1071);
//   * components outside [first, last] enter the colour transformation as zeros (:1190-1193);
//   * the colour transformer is built by the first call that reconstructs anything and kept
//     (colortrafo/colortransformerfactory.cpp:220-221): JPGTAG_MATRIX_LTRAFO of later calls changes nothing;
//   * Upsampler::UpsampleRegion produces 8x8 samples starting AT the corner of the (sub)rectangle (upsampler.cpp:85-86)
//     but the colour transformer reads its source blocks at (x & 7, y & 7) (colortrafo/ycbcrtrafo.cpp:683-686):
//     a request whose corner is off the 8-pixel grid sees subsampled components displaced in its first row / column of
//     blocks, with the vertical filter clamped at the end of the upsampler's buffered window instead of the image.
// This class keeps that state and turns one request into a PLAN: per component and block row the coefficient row to
// transform (or "zeros"), the displacement and the filter window -- which capi.cpp either recognises as the plain picture
// (served from the cached whole-frame reconstruction) or hands to the generic kernels (kernels.hip, GenericArgs::rowmap).
// Pure host integer logic, no device code.  tests/test_rect_calls.py drives it against the real reference library.
#ifndef MIJ_REQUEST_MODEL_HPP
#define MIJ_REQUEST_MODEL_HPP

#include <stdint.h>

#include <algorithm>
#include <vector>

namespace mij {

struct RequestPlan {
  bool nothing = true;        // nothing is reconstructed (empty region, bitmap too low)
  bool upsampling_path = true; // PullQData + PushReconstructedData; false: ReconstructUnsampled
  bool ycc = false;           // the colour transformer in force: YCbCr -> RGB on three components
  int view = -1;              // ReconstructUnsampled without upsampling: the component on whose grid the region lives
  int min_x = 0, min_y = 0, max_x = -1, max_y = -1; // region that is reconstructed (canvas, or the view's grid)
  int corner_x = 0, corner_y = 0; // the request's corner: subsampled components are displaced by (corner & 7)
  bool requested[4] = {false, false, false, false};
  bool upsampler[4] = {false, false, false, false}; // the component goes through an upsampler in this request
  // rowmap[c][g]: coefficient row whose transform fills block row g of component c's (virtual) sample plane, -1 = samples
  // of value 0; defined for g in [g0[c], g1[c]] (what the request can read), identity elsewhere
  std::vector<int32_t> rowmap[4];
  int g0[4] = {0, 0, 0, 0}, g1[4] = {-1, -1, -1, -1};
  int wstart[4] = {0, 0, 0, 0}, wlimit[4] = {0, 0, 0, 0}; // upsampler window in lines of the component: [wstart, wlimit)
  bool plain = false;         // the request shows the plain picture: identity row maps, nothing displaced, every component
                              // that the colour transformation mixes is there
};

class RequestModel {
public:
  // dnl: the frame's height arrived in a DNL marker, i.e. after the reference built its upsamplers -- they took 0 for "unknown"
  // and made it 2^31 - 1 lines (upsampling/upsamplerbase.cpp:61-75), so nothing clips their buffered region at the bottom of
  // the picture (:138-156, :218-228) -- and after its first scan created block rows without knowing where to stop
  // (control/blockbuffer.cpp:212-265): rows[c] of them exist (the store keeps store_rows[c]).
  // other_image_subsampled: JPEG XT -- the legacy and the residual image of one frame share m_bSubsampling
  // (control/blockbitmaprequester.cpp:318, 370): a subsampled component in EITHER image puts both on the upsampling path, where
  // every component without an upsampler advances with every block row shown (:1214-1222).  One model per image, same requests.
  void reset(int ncomp, int width, int height, const int32_t *subx, const int32_t *suby, bool frame_ycbcr, bool dnl = false,
             const int32_t *rows = nullptr, const int32_t *store_rows = nullptr, bool other_image_subsampled = false)
  {
    nc_ = ncomp; w_ = width; h_ = height; frame_ycbcr_ = frame_ycbcr;
    dnl_ = dnl && rows && store_rows;
    subsampling_ = other_image_subsampled;
    trafo_built_ = false;
    ycc_ = false;
    for (int c = 0; c < 4; c++) {
      sx_[c] = c < ncomp ? subx[c] : 1;
      sy_[c] = c < ncomp ? suby[c] : 1;
      cur_[c] = 0;
      rows_[c] = c < ncomp ? (((height + sy_[c] - 1) / sy_[c]) + 7) >> 3 : 0; // control/blockbuffer.cpp:212-265
      if (dnl_ && c < ncomp) rows_[c] = std::min(rows[c], store_rows[c]);
      up_[c] = c < ncomp && (sx_[c] > 1 || sy_[c] > 1);                       // blockbitmaprequester.cpp:310-318
      subsampling_ = subsampling_ || up_[c];
      uy_[c] = uh_[c] = 0;
      tags_[c].clear();
    }
  }
  int cursor(int c) const { return cur_[c]; }
  bool subsampled() const { for (int c = 0; c < nc_; c++) if (up_[c]) return true; return false; }
  int rows(int c) const { return rows_[c]; }
  bool transformer_built() const { return trafo_built_; }

  // One DisplayRectangle call.  Rectangle and component range as the tags give them (inclusive; clipped here like
  // codestream/rectanglerequest.cpp:62-190 does), bm_height[c]: BIO_HEIGHT the hook reported for the requested components.
  // Advances the state and returns what the call shows.
  RequestPlan request(int min_x, int min_y, int max_x, int max_y, int c0, int c1, bool upsample, bool ctrafo, const uint32_t bm_height[4])
  {
    RequestPlan p;
    min_x = std::max(min_x, 0);
    min_y = std::max(min_y, 0);
    max_x = std::min(max_x, w_ - 1);
    max_y = std::min(max_y, h_ - 1);
    c0 = std::max(c0, 0);
    c1 = std::min(c1, nc_ - 1);
    if (!upsample) ctrafo = false; // rectanglerequest.cpp:157-159
    uint32_t maxmcu = 0xffffffffu; // blockbitmaprequester.cpp:1229-1244, ULONG arithmetic: heights below 8 wrap to "no bound"
    for (int c = c0; c <= c1; c++) maxmcu = std::min(maxmcu, (bm_height[c] >> 3) - 1u);
    if (min_x > max_x || min_y > max_y || c0 > c1) return p; // codestream/image.cpp:1115
    if (!trafo_built_) {
      trafo_built_ = true;
      ycc_ = ctrafo && frame_ycbcr_ && nc_ == 3;
    }
    p.ycc = ycc_;
    for (int c = c0; c <= c1; c++) p.requested[c] = true;
    p.corner_x = min_x;
    p.corner_y = min_y;
    if (subsampling_ && upsample) {
      p.upsampling_path = true;
      // PullQData: the upsamplers of the requested components take in the block rows they do not hold yet
      for (int c = c0; c <= c1; c++) {
        if (!up_[c]) continue;
        const int sx = sx_[c], sy = sy_[c];
        const int total = dnl_ ? (int)((0x7fffffffu + (uint32_t)sy - 1) / (uint32_t)sy) : (h_ + sy - 1) / sy;
        const int bheight = (int)(((uint32_t)total + 7u) >> 3);
        int gmin = (min_y / sy - (sy > 1 ? 1 : 0)) >> 3, gmax = (max_y / sy + (sy > 1 ? 1 : 0)) >> 3;
        gmin = std::max(gmin, 0);
        gmax = std::min(gmax, bheight - 1);
        (void)sx;
        // SetBufferedRegion: lines above the region leave the buffer one by one ...
        const int target = gmin << 3;
        if (uy_[c] < target) {
          const int gone = target - uy_[c];
          uh_[c] = std::max(0, uh_[c] - gone);
          const size_t groups = std::min<size_t>(tags_[c].size(), (size_t)(gone >> 3));
          tags_[c].erase(tags_[c].begin(), tags_[c].begin() + (ptrdiff_t)groups);
          if (uh_[c] == 0) tags_[c].clear();
          uy_[c] = target;
        } else if (uy_[c] > target) { // ... and a buffer that starts below the region's top is disposed of
          uh_[c] = 0;
          tags_[c].clear();
          uy_[c] = target;
        }
        int first_new = (uy_[c] + uh_[c] + 7) >> 3;
        const int maxy = std::min((gmax + 1) << 3, total); // ExtendBufferedRegion
        if (uy_[c] + uh_[c] < maxy) uh_[c] = maxy - uy_[c];
        for (int g = first_new; g <= gmax; g++) { // PullQData's row loop: one cursor row per new block row
          const size_t k = (size_t)(g - (uy_[c] >> 3));
          if (tags_[c].size() <= k) tags_[c].resize(k + 1, -1);
          tags_[c][k] = cur_[c] < rows_[c] ? cur_[c] : -1;
          if (cur_[c] < rows_[c]) cur_[c]++;
        }
      }
      // PushReconstructedData
      uint32_t by0 = (uint32_t)min_y >> 3, by1 = (uint32_t)max_y >> 3;
      if (by1 > maxmcu) by1 = maxmcu;
      if (by0 > by1) { // nothing is shown, nothing advances
        return p;
      }
      p.nothing = false;
      p.min_x = min_x; p.max_x = max_x; p.min_y = min_y;
      p.max_y = std::min<int64_t>(max_y, (int64_t)by1 * 8 + 7);
      for (int c = 0; c < nc_; c++) {
        p.upsampler[c] = up_[c];
        if (up_[c]) {
          if (!p.requested[c]) continue;
          p.g0[c] = uy_[c] >> 3;
          p.g1[c] = p.g0[c] + (int)tags_[c].size() - 1;
          p.rowmap[c].assign((size_t)std::max(rows_[c], dnl_ ? p.g1[c] + 1 : 0), -1);
          for (int g = p.g0[c]; g <= p.g1[c] && g < (int)p.rowmap[c].size(); g++) p.rowmap[c][(size_t)g] = tags_[c][(size_t)(g - p.g0[c])];
          p.wstart[c] = uy_[c];
          p.wlimit[c] = uy_[c] + uh_[c];
        } else {
          // rows of the components without an upsampler: the cursor row for every block row of the stripe; ALL of them
          // advance, requested or not (:1214-1223)
          if (p.requested[c]) {
            p.g0[c] = (int)by0;
            p.g1[c] = (int)by1;
            p.rowmap[c].assign((size_t)std::max<int>(rows_[c], (int)by1 + 1), -1);
          }
          for (uint32_t by = by0; by <= by1; by++) {
            if (p.requested[c]) p.rowmap[c][by] = cur_[c] < rows_[c] ? cur_[c] : -1;
            if (cur_[c] < rows_[c]) cur_[c]++;
          }
        }
      }
    } else {
      p.upsampling_path = false;
      if (!upsample) { // control/bitmapctrl.cpp:273-294 (c0 == c1 is the caller's check: INVALID_PARAMETER otherwise)
        const int sx = sx_[c0], sy = sy_[c0];
        p.view = c0;
        min_x = (min_x + sx - 1) / sx;
        max_x = (max_x + sx) / sx - 1;
        min_y = (min_y + sy - 1) / sy;
        max_y = (max_y + sy) / sy - 1;
      }
      uint32_t by0 = (uint32_t)min_y >> 3, by1 = (uint32_t)max_y >> 3;
      if (by1 > maxmcu) by1 = maxmcu;
      if (by0 > by1) return p;
      // (a region that the division by the subsampling factors turned inside out still walks its block rows -- the loops
      // run on block indices -- and moves the cursors; it just shows nothing)
      p.nothing = min_x > max_x || min_y > max_y;
      p.min_x = min_x; p.max_x = max_x; p.min_y = min_y;
      p.max_y = std::min<int64_t>(max_y, (int64_t)by1 * 8 + 7);
      for (int c = c0; c <= c1; c++) { // only the requested components advance (:1066-1071)
        p.g0[c] = (int)by0;
        p.g1[c] = (int)by1;
        p.rowmap[c].assign((size_t)std::max<int>(rows_[c], (int)by1 + 1), -1);
        for (uint32_t by = by0; by <= by1; by++) {
          p.rowmap[c][by] = cur_[c] < rows_[c] ? cur_[c] : -1;
          if (cur_[c] < rows_[c]) cur_[c]++;
        }
      }
    }
    // Is this the plain picture?  Identity maps over what is read, no displaced upsampler output, and no colour
    // transformation that would mix in the zeros of components that were not asked for.
    bool plain = true;
    for (int c = 0; c < nc_ && plain; c++) {
      if (!p.requested[c]) {
        if (p.ycc) plain = false;
        continue;
      }
      for (int g = p.g0[c]; g <= p.g1[c] && plain; g++)
        if (g < (int)p.rowmap[c].size() && p.rowmap[c][(size_t)g] != g) plain = false;
      if (p.upsampling_path && up_[c]) {
        if ((p.corner_x & 7) || (p.corner_y & 7)) plain = false;
        // the filter window must reach as far as the image-bound filter would read: the last line shown plus one
        // (plus / minus one line where there is a vertical filter)
        const int halo = sy_[c] > 1 ? 1 : 0;
        const int need = dnl_ ? (p.max_y / sy_[c]) + halo : std::min((p.max_y / sy_[c]) + halo, (h_ + sy_[c] - 1) / sy_[c] - 1);
        if (p.wlimit[c] <= need || p.wstart[c] > std::max(p.min_y / sy_[c] - halo, 0)) plain = false;
      }
    }
    p.plain = plain;
    return p;
  }

private:
  int nc_ = 0, w_ = 0, h_ = 0;
  bool frame_ycbcr_ = false, subsampling_ = false, trafo_built_ = false, ycc_ = false, dnl_ = false;
  int sx_[4] = {1, 1, 1, 1}, sy_[4] = {1, 1, 1, 1};
  int cur_[4] = {0, 0, 0, 0}, rows_[4] = {0, 0, 0, 0};
  bool up_[4] = {false, false, false, false};
  int uy_[4] = {0, 0, 0, 0}, uh_[4] = {0, 0, 0, 0}; // m_lY, m_lHeight of the component's upsampler
  std::vector<int32_t> tags_[4];                     // coefficient row behind each buffered block row, from uy_ >> 3 on
};

} // namespace mij
#endif
