// jpeg.cpp -- class JPEG (decode half) on top of the C ABI.  What each member replaces in the reference:
//   Read             interface/jpeg.cpp:205-354   (stream pulled through the I/O hook, io/iostream.cpp;
//                                                  whole entropy decode happens here, as in the reference)
//   GetInformation   interface/jpeg.cpp:822-957
//   DisplayRectangle interface/jpeg.cpp:694-722 -> codestream/rectanglerequest.cpp:62-190 (tag parsing),
//                    control/bitmapctrl.cpp:142-176 + interface/bitmaphook.cpp:130-248 (REQUEST / RELEASE
//                    protocol per component), control/blockbitmaprequester.cpp:1229-1244 (the height the hook
//                    reports bounds the block rows that are reconstructed)
//   LastError        interface/jpeg.cpp:959-979
//   ProvideImage     interface/jpeg.cpp:461-597 (image parameters from the tags, pixel data pulled stripe by stripe through
//                    the bitmap hook with the same REQUEST / RELEASE protocol) -- baseline / sequential 8-bit frames
//   Write            interface/jpeg.cpp:356-459 (the finished stream is pushed through the I/O hook, JPGFLAG_ACTION_WRITE)
// Error convention: every call returns JPG_TRUE / JPG_FALSE, nothing is thrown across the boundary.
#include "jpeg.hpp"

#include <stdlib.h>
#include <string.h>

#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mijpeg.h"
#include "hooks.hpp"
#include "parameters.hpp"

struct JPEG::Impl {
  mijpeg_decoder *dec = nullptr;
  std::vector<uint8_t> stream; // the codestream, pulled through the I/O hook
  mijpeg_info info;
  bool loaded = false;
  // incremental reading (JPGTAG_DECODER_STOP) and the marker calls: the position the reference's IOStream would stand at
  // while the headers are walked marker by marker, and the byte ranges the client took out of the stream itself
  enum Phase { P_NONE, P_SOI, P_TABLES, P_FRAME, P_PRESCAN_INIT, P_PRESCAN, P_DECODE, P_SCANS, P_DONE } phase = P_NONE;
  // JPGFLAG_DECODER_STOP_SCAN: where the reference stands when it returns with a scan header parsed (the first entropy coded
  // byte of every scan), in the order of the scans; the scans are decoded -- in parallel -- by the call that meets the first one
  std::vector<size_t> scan_stops, scan_ends; // first entropy coded byte / first byte behind the data of every scan of the codestream
  // JPGFLAG_DECODER_STOP_ROW / _MCU (interface/jpeg.cpp:326-350): behind a scan header the reference returns at the start of every
  // MCU row and behind every MCU of a row but its last.  The scans are decoded already; what is walked is the grid of each.
  std::vector<int32_t> grid_x, grid_y; // per scan, the boxed ones behind the codestream's (mijpeg_scan_grids)
  bool in_scan = false, in_row = false;
  int32_t row = 0, mcu = 0;
  size_t walking = 0; // the scan whose grid is being walked
  size_t boxed_stops = 0;                   // JPEG XT: further scans that live in boxes
  size_t next_stop = 0;
  bool between = false;                     // walking the marker segments between two scans
  bool pulled = false;
  int read_err = 0;            // what a failed Read reported (repeated by further Read calls)
  std::string read_errmsg;
  size_t cursor = 0;
  std::vector<std::pair<size_t, size_t>> taken; // [from, to) ranges removed by ReadMarker / SkipMarker, ascending
  std::vector<uint8_t> effective;               // the stream without them (only built when something was taken)
  long peekword() const { return cursor + 2 > stream.size() ? -1L : ((long)stream[cursor] << 8) | stream[cursor + 1]; }
  void take(size_t n)
  {
    if (n == 0) return;
    if (!taken.empty() && taken.back().second == cursor) taken.back().second = cursor + n;
    else taken.push_back(std::make_pair(cursor, cursor + n));
    cursor += n;
  }
  // Positions only, no validation (the full parse of the final Read validates): where Tables::ParseTablesIncremental
  // (codestream/tables.cpp:1003-1418) stands after one call.  false: the marker here does not belong to the tables /
  // miscellaneous section (frame header, scan header, EOI, DHP) or the data ran out.
  bool tables_step()
  {
    const long m = peekword();
    if (m < 0) return false;
    switch (m) {
    case 0xffc0: case 0xffc1: case 0xffc2: case 0xffc3: case 0xffc5: case 0xffc6: case 0xffc7: case 0xffc9: case 0xffca: case 0xffcb:
    case 0xffcd: case 0xffce: case 0xffcf: case 0xffb1: case 0xffb2: case 0xffb3: case 0xffb9: case 0xffba: case 0xffbb:
    case 0xffd9: case 0xffda: case 0xffde: case 0xfff7:
      return false;
    case 0xffff: cursor += 1; return true;                // a fill byte in front of a marker
    case 0xffd0: case 0xffd1: case 0xffd2: case 0xffd3: case 0xffd4: case 0xffd5: case 0xffd6: case 0xffd7:
      cursor += 2; return true;                           // stray restart marker: warned about and ignored
    default: break;
    }
    if (m >= 0xffc0 && m < 0xfff0) { // length-prefixed segment (tables, APPn, COM, EXP, ...): LSE (fff8) is an error anyway
      if (cursor + 4 > stream.size()) { cursor = stream.size(); return true; }
      const size_t len = ((size_t)stream[cursor + 2] << 8) | stream[cursor + 3];
      cursor = cursor + 2 + len > stream.size() ? stream.size() : cursor + 2 + (len < 2 ? 2 : len);
      return true;
    }
    // "found invalid marker": advance to the next 0xff by hand (tables.cpp:1399-1413)
    cursor += 1;
    while (cursor < stream.size()) {
      if (stream[cursor++] == 0xff) { cursor--; return true; }
    }
    return false;
  }
  // encoder direction: the picture as ProvideImage collects it, and its parameters
  std::vector<uint8_t> picture;
  int enc_width = 0, enc_height = 0, enc_depth = 0, enc_quality = 75, enc_restart = 0, enc_lines = 0;
  bool enc_optimize = false, enc_ycbcr = true;
  int32_t enc_hsamp[4] = {1, 1, 1, 1}, enc_vsamp[4] = {1, 1, 1, 1};
  int err = 0;
  std::string errmsg;
  int fail(int code, const char *msg)
  {
    err = code;
    errmsg = msg ? msg : "";
    return JPG_FALSE;
  }
  // Nothing is thrown across the boundary (the reference: JPG_TRY / JPG_CATCH around every call, interface/jpeg.cpp:205-220): what
  // the methods' function-try-blocks catch is out of memory in the vectors this object keeps, or a defect.
  void caught(const char *where) noexcept
  {
    try {
      throw;
    } catch (const std::bad_alloc &) {
      err = JPGERR_OUT_OF_MEMORY;
    } catch (const std::length_error &) {
      err = JPGERR_OUT_OF_MEMORY;
    } catch (...) {
      err = JPGERR_PHASE_ERROR;
    }
    try {
      errmsg = std::string(where) + (err == JPGERR_OUT_OF_MEMORY ? ": out of memory" : ": unexpected exception");
    } catch (...) {
      errmsg.clear();
    }
  }
  int fail_from_decoder(int code)
  {
    const char *m = nullptr;
    mijpeg_last_error(dec, &m);
    return fail(code, m ? m : "decoder error");
  }
};

JPEG::JPEG() : m_pImpl(nullptr) {}
JPEG::~JPEG() {}

class JPEG *JPEG::Construct(struct JPG_TagItem *tags)
{
  JPEG *o = new (std::nothrow) JPEG();
  if (!o) return nullptr;
  o->m_pImpl = new (std::nothrow) Impl();
  if (!o->m_pImpl) {
    delete o;
    return nullptr;
  }
  memset(&o->m_pImpl->info, 0, sizeof(mijpeg_info));
  int device = getenv("MIJPEG_DEVICE") ? atoi(getenv("MIJPEG_DEVICE")) : 0;
  if (tags) device = tags->GetTagData(JPGTAG_MIJPEG_DEVICE, device);
  if (mijpeg_create(&o->m_pImpl->dec, device) != MIJPEG_OK) {
    delete o->m_pImpl;
    delete o;
    return nullptr;
  }
  return o;
}

void JPEG::Destruct(class JPEG *o)
{
  if (!o) return;
  if (o->m_pImpl) {
    mijpeg_destroy(o->m_pImpl->dec);
    delete o->m_pImpl;
  }
  delete o;
}

JPG_LONG JPEG::Read(struct JPG_TagItem *tags)
try {
  Impl *p = m_pImpl;
  p->err = 0;
  if (!tags) return p->fail(JPGERR_MISSING_PARAMETER, "JPEG::Read requires a tag list with an I/O hook");
  const JPG_LONG stopflags = tags->GetTagData(JPGTAG_DECODER_STOP, 0);
  if (p->loaded && p->phase != Impl::P_SCANS) return JPG_TRUE; // "if (!m_bDecoding) return" (interface/jpeg.cpp:262-263)
  if (p->phase == Impl::P_DONE) // the decode failed before: reading on cannot succeed either
    return p->fail(p->read_err ? p->read_err : JPGERR_MALFORMED_STREAM, p->read_errmsg.empty() ? "the stream could not be decoded" : p->read_errmsg.c_str());
  if (!p->pulled) {
    struct JPG_Hook *io = (struct JPG_Hook *)tags->GetTagPtr(JPGTAG_HOOK_IOHOOK);
    if (!io) return p->fail(JPGERR_OBJECT_DOESNT_EXIST, "no IOHook defined to read the data from");
    // pull the whole codestream: the entropy decoder works on an in-memory stream.  As io/iostream.cpp:173-197:
    // only a return of 0 is the end of the file (short reads from pipes and sockets are normal), the hook may
    // hand back a buffer of its own and a changed user data word, both are re-fetched after every call.
    p->stream.clear();
    const size_t chunk = 1 << 20;
    JPG_LONG userdata = tags->GetTagData(JPGTAG_FIO_USERDATA, 0); // io/iostream.cpp:89-91
    size_t have = 0;
    for (;;) {
      if (p->stream.size() - have < chunk / 2) p->stream.resize(have + chunk);
      uint8_t *buf = p->stream.data() + have;
      const size_t room = p->stream.size() - have;
      struct JPG_TagItem iotags[] = {
          JPG_PointerTag(JPGTAG_FIO_BUFFER, buf),
          JPG_ValueTag(JPGTAG_FIO_SIZE, (JPG_LONG)room),
          JPG_PointerTag(JPGTAG_FIO_HANDLE, tags->GetTagPtr(JPGTAG_HOOK_IOSTREAM)),
          JPG_ValueTag(JPGTAG_FIO_ACTION, JPGFLAG_ACTION_READ),
          JPG_ValueTag(JPGTAG_FIO_USERDATA, userdata),
          JPG_ValueTag(JPGTAG_FIO_SEEKMODE, JPGFLAG_OFFSET_CURRENT),
          JPG_ValueTag(JPGTAG_FIO_OFFSET, 0),
          JPG_EndTag};
      const JPG_LONG got = io->CallLong(iotags);
      if (got < 0) {
        p->stream.clear();
        return p->fail(got, "Client signalled an error on reading from the file hook");
      }
      if (got == 0) break;
      const uint8_t *from = (const uint8_t *)iotags[0].ti_Data.ti_pPtr;
      userdata = iotags[4].ti_Data.ti_lData;
      if (from != buf) { // the hook filled a buffer of its own
        if ((size_t)got > room) p->stream.resize(have + (size_t)got);
        memcpy(p->stream.data() + have, from, (size_t)got);
      } else if ((size_t)got > room) {
        p->stream.clear();
        return p->fail(JPGERR_OVERFLOW_PARAMETER, "the I/O hook reports more bytes than the buffer holds");
      }
      have += (size_t)got;
    }
    p->stream.resize(have);
    // (codestream/decoder.cpp:92-96: an empty stream fails the SOI test like any other non-JPEG)
    if (p->stream.empty()) return p->fail(JPGERR_MALFORMED_STREAM, "stream does not contain a JPEG file, SOI marker missing");
    p->pulled = true;
    p->phase = Impl::P_SOI;
    p->cursor = 0;
  }
  // The header walk of JPEG::ReadInternal (interface/jpeg.cpp:276-318) over the in-memory stream, one marker segment per
  // step, returning where the stop flags ask for it.  Without stop flags it falls straight through to the decode.
  for (;;) {
    switch (p->phase) {
    case Impl::P_SOI: // Decoder::ParseHeaderIncremental, first call (codestream/decoder.cpp:94-104)
      if (p->peekword() != 0xffd8) {
        p->phase = Impl::P_DONE;
        p->read_err = JPGERR_MALFORMED_STREAM;
        p->read_errmsg = "stream does not contain a JPEG file, SOI marker missing";
        return p->fail(JPGERR_MALFORMED_STREAM, "stream does not contain a JPEG file, SOI marker missing");
      }
      p->cursor = 2;
      p->phase = Impl::P_TABLES;
      if (stopflags & JPGFLAG_DECODER_STOP_IMAGE) return JPG_TRUE;
      break;
    case Impl::P_TABLES: // ... later calls: one marker of the tables / misc section each
      if (p->tables_step()) {
        if (stopflags & JPGFLAG_DECODER_STOP_IMAGE) return JPG_TRUE;
      } else {
        p->phase = Impl::P_FRAME;
        if (stopflags & JPGFLAG_DECODER_STOP_IMAGE) return JPG_TRUE;
      }
      break;
    case Impl::P_FRAME: { // Image::StartParseFrame -> ParseFrameHeader (codestream/image.cpp:616-683): the frame header
      const long m = p->peekword();
      if (m >= 0 && m != 0xffd9 && p->cursor + 4 <= p->stream.size()) {
        const size_t len = ((size_t)p->stream[p->cursor + 2] << 8) | p->stream[p->cursor + 3];
        p->cursor = p->cursor + 2 + len > p->stream.size() ? p->stream.size() : p->cursor + 2 + (len < 2 ? 2 : len);
        p->phase = Impl::P_PRESCAN_INIT;
        if (stopflags & JPGFLAG_DECODER_STOP_FRAME) return JPG_TRUE;
      } else {
        p->phase = Impl::P_DECODE; // EOF / EOI here: the full parse reports it
      }
      break;
    }
    case Impl::P_PRESCAN_INIT: // Frame::StartParseScan (marker/frame.cpp:821-831): its first call only arms the table parser
      p->phase = Impl::P_PRESCAN;
      if (stopflags & JPGFLAG_DECODER_STOP_FRAME) return JPG_TRUE;
      break;
    case Impl::P_PRESCAN: // ... the later ones take the tables between the frame header and the scan, one per call
      if (p->tables_step()) {
        if (stopflags & JPGFLAG_DECODER_STOP_FRAME) return JPG_TRUE;
      } else {
        p->phase = Impl::P_DECODE;
      }
      break;
    case Impl::P_DECODE: {
      // From the first scan header on everything happens in this call: the stop flags for scans, rows and MCUs have
      // nothing finer to stop at (the scans are decoded in parallel, not MCU by MCU); INTEGRATION.md says so.
      const uint8_t *data = p->stream.data();
      size_t size = p->stream.size();
      if (!p->taken.empty()) { // what the client consumed through ReadMarker / SkipMarker never reaches the parser
        p->effective.clear();
        size_t at = 0;
        for (const auto &r : p->taken) {
          p->effective.insert(p->effective.end(), p->stream.begin() + (ptrdiff_t)at, p->stream.begin() + (ptrdiff_t)r.first);
          at = r.second;
        }
        p->effective.insert(p->effective.end(), p->stream.begin() + (ptrdiff_t)at, p->stream.end());
        data = p->effective.data();
        size = p->effective.size();
      }
      p->phase = Impl::P_DONE;
      auto failed = [&](int code) { // remember it for later Read calls
        const JPG_LONG r = p->fail_from_decoder(code);
        p->read_err = p->err;
        p->read_errmsg = p->errmsg;
        return r;
      };
      int rc = mijpeg_set_input(p->dec, data, size);
      if (rc) return failed(rc);
      const int threads = tags->GetTagData(JPGTAG_MIJPEG_THREADS, getenv("MIJPEG_THREADS") ? atoi(getenv("MIJPEG_THREADS")) : 0);
      // streams with enough restart intervals are entropy-decoded on the device (JPGTAG_MIJPEG_ENTROPY /
      // MIJPEG_ENTROPY: 0 = automatic, 1 = always on the host)
      const int entropy = tags->GetTagData(JPGTAG_MIJPEG_ENTROPY, getenv("MIJPEG_ENTROPY") ? atoi(getenv("MIJPEG_ENTROPY")) : 0);
      rc = entropy == 1 ? MIJPEG_ERR_NOT_AVAILABLE : mijpeg_decode_coefficients_device(p->dec, 0);
      if (rc == MIJPEG_ERR_NOT_AVAILABLE) rc = mijpeg_decode_coefficients(p->dec, threads);
      if (rc) return failed(rc);
      mijpeg_get_info(p->dec, &p->info);
      p->loaded = true;
      // interface/jpeg.cpp:310-353: with JPGFLAG_DECODER_STOP_SCAN the reference returns every time Frame::StartParseScan has
      // parsed a scan header -- the client sees the stream positioned at that scan's entropy coded data -- and decodes the scan
      // in the next call.  Here all scans are decoded by now (in parallel); the calls that follow walk the same positions.
      p->scan_stops.clear();
      p->scan_ends.clear();
      p->next_stop = 0;
      p->between = false;
      {
        // (a progressive frame may bring hundreds of scans: ask for the count first)
        const int ns = mijpeg_scan_offsets(p->dec, nullptr, nullptr, 0);
        std::vector<uint64_t> at((size_t)std::max(ns, 1)), end((size_t)std::max(ns, 1));
        mijpeg_scan_offsets(p->dec, at.data(), end.data(), ns);
        auto in_stream = [&](uint64_t off) { // offsets count the bytes the parser saw: what the client took out lies in front of them in `stream`
          size_t pos = (size_t)off;
          for (const auto &r : p->taken)
            if (r.first <= pos) pos += r.second - r.first;
          return pos < p->stream.size() ? pos : p->stream.size();
        };
        p->grid_x.assign((size_t)std::max(ns, 1), 1);
        p->grid_y.assign((size_t)std::max(ns, 1), 0);
        mijpeg_scan_grids(p->dec, p->grid_x.data(), p->grid_y.data(), ns);
        p->in_scan = p->in_row = false;
        p->boxed_stops = 0;
        for (int k = 0; k < ns; k++) {
          if (end[(size_t)k] == 0) { p->boxed_stops++; continue; }
          p->scan_stops.push_back(in_stream(at[(size_t)k]));
          p->scan_ends.push_back(in_stream(end[(size_t)k]));
        }
      }
      p->phase = Impl::P_SCANS;
      break;
    }
    case Impl::P_SCANS: {
      // The reference's walk from scan to scan (interface/jpeg.cpp:296-353, marker/frame.cpp:794-862), positions only: behind
      // a scan's data Frame::StartParseScan takes the marker segments in front of the next scan header one per call (each a
      // return under JPGFLAG_DECODER_STOP_FRAME), then parses the header (a return under JPGFLAG_DECODER_STOP_SCAN, with the
      // stream at the scan's entropy coded data).  The scans themselves are decoded already.
      if (p->in_scan) { // the rows and MCUs of the scan whose header was the last stop
        const int32_t rows = p->walking < p->grid_y.size() ? p->grid_y[p->walking] : 0, across = p->walking < p->grid_x.size() ? p->grid_x[p->walking] : 1;
        if (!p->in_row) {
          if (p->row < rows) { // StartMCURow: another row
            p->in_row = true;
            p->mcu = 0;
            if (stopflags & JPGFLAG_DECODER_STOP_ROW) return JPG_TRUE;
          } else {
            p->in_scan = false;
            break;
          }
        }
        // ParseMCU says "more in this row" behind every MCU but the last
        if (!(stopflags & JPGFLAG_DECODER_STOP_MCU)) p->mcu = across > 0 ? across - 1 : 0;
        if (p->mcu + 1 < across) {
          p->mcu++;
          return JPG_TRUE;
        }
        p->in_row = false;
        p->row++;
        break;
      }
      const size_t k = p->next_stop;
      auto enter_scan = [&](size_t which) {
        p->in_scan = true;
        p->in_row = false;
        p->row = p->mcu = 0;
        p->walking = which;
      };
      if (k < p->scan_stops.size()) {
        if (k > 0) {
          if (!p->between) {
            p->between = true;
            p->cursor = p->scan_ends[k - 1];
          }
          if (p->cursor < p->scan_stops[k] && p->tables_step() && p->cursor <= p->scan_stops[k]) {
            if (stopflags & JPGFLAG_DECODER_STOP_FRAME) return JPG_TRUE;
            break;
          }
        }
        p->between = false;
        p->cursor = p->scan_stops[k];
        p->next_stop++;
        enter_scan(k);
        if (stopflags & JPGFLAG_DECODER_STOP_SCAN) return JPG_TRUE;
        break;
      }
      if (k < p->scan_stops.size() + p->boxed_stops) { // scans from boxes: the input stands behind the last scan of the codestream
        p->cursor = p->scan_ends.empty() ? p->stream.size() : p->scan_ends.back();
        if (!p->between) {
          // JPEG XT: the frame of the residual image starts here (Image::StartParseFrame on its box, a return under
          // JPGFLAG_DECODER_STOP_FRAME; the returns its own table walk adds stand at the same byte of the input and are not
          // told apart from it)
          p->between = true;
          if (stopflags & JPGFLAG_DECODER_STOP_FRAME) return JPG_TRUE;
        }
        p->next_stop++;
        enter_scan(k);
        if (stopflags & JPGFLAG_DECODER_STOP_SCAN) return JPG_TRUE;
        break;
      }
      p->phase = Impl::P_DONE;
      p->cursor = p->stream.size(); // the reference stands behind the EOI now
      return JPG_TRUE;
    }
    default: return JPG_TRUE;
    }
  }
} catch (...) { m_pImpl->caught("JPEG::Read"); return JPG_FALSE; }

JPG_LONG JPEG::GetInformation(struct JPG_TagItem *tags)
try {
  Impl *p = m_pImpl;
  if (!p->loaded) return p->fail(JPGERR_OBJECT_DOESNT_EXIST, "no image loaded to request information from");
  if (!tags) return JPG_TRUE;
  const mijpeg_info &f = p->info;
  tags->SetTagData(JPGTAG_IMAGE_WIDTH, f.width);
  tags->SetTagData(JPGTAG_IMAGE_HEIGHT, f.height);
  tags->SetTagData(JPGTAG_IMAGE_DEPTH, f.components);
  // JPEG XT: the image precision includes the extra range bits of the output conversion (8 + 8 for the encoder's HDR files,
  // 8 + 0 for its integer ones)
  auto precision_of = [](mijpeg_decoder *dec, const mijpeg_info &g) {
    int prec = g.precision;
    if (g.xt) {
      mijpeg_xt_params x;
      prec = 16;
      if (mijpeg_get_xt_params(dec, &x) == MIJPEG_OK)
        for (prec = 1; (1 << prec) <= x.out_max; prec++) {}
    }
    return prec;
  };
  tags->SetTagData(JPGTAG_IMAGE_PRECISION, precision_of(p->dec, f));
  const JPG_LONG n = tags->GetTagData(JPGTAG_IMAGE_SUBLENGTH, 0);
  if (n > 0) {
    uint8_t *sx = (uint8_t *)tags->GetTagPtr(JPGTAG_IMAGE_SUBX), *sy = (uint8_t *)tags->GetTagPtr(JPGTAG_IMAGE_SUBY);
    if (sx) memset(sx, 0, (size_t)n);
    if (sy) memset(sy, 0, (size_t)n);
    for (int c = 0; c < f.components && c < n; c++) {
      if (sx) sx[c] = (uint8_t)f.subx[c];
      if (sy) sy[c] = (uint8_t)f.suby[c];
    }
  }
  // jpeg.cpp:836-862: with a merging specification that casts to float the samples are half-float codes
  // (IS_FLOAT) which the client expands itself (OUTPUT_CONVERSION); plain JPEG: integer output
  tags->SetTagData(JPGTAG_IMAGE_IS_FLOAT, f.xt && f.is_float ? 1 : 0);
  tags->SetTagData(JPGTAG_IMAGE_OUTPUT_CONVERSION, f.xt && f.is_float ? 1 : 0);
  // The alpha channel (interface/jpeg.cpp:919-951): with an alpha merging specification that carries a compositing box and an
  // alpha image that has been read, the method, the matte colour and -- in the client's JPGTAG_ALPHA_TAGLIST -- precision and
  // output conversion of the alpha image; without, the reference neutralises the two tags
  struct JPG_TagItem *alphatag = tags->FindTagItem(JPGTAG_ALPHA_MODE), *alphalist = tags->FindTagItem(JPGTAG_ALPHA_TAGLIST);
  int32_t mode = -1, matte[3] = {0, 0, 0};
  mijpeg_decoder *adec = mijpeg_has_alpha(p->dec) ? mijpeg_alpha_channel(p->dec) : nullptr;
  mijpeg_info a;
  if (adec && mijpeg_alpha_info(p->dec, &mode, matte) == MIJPEG_OK && mode >= 0 && mijpeg_get_info(adec, &a) == MIJPEG_OK) {
    if (alphatag) alphatag->ti_Data.ti_lData = mode;
    for (int k = 0; k < 3; k++) tags->SetTagData(JPGTAG_ALPHA_MATTE(k), matte[k]);
    if (alphalist) {
      struct JPG_TagItem *al = (struct JPG_TagItem *)alphalist->ti_Data.ti_pPtr;
      if (al) {
        // (Image::PrecisionOf of the alpha image: its frame's precision plus the extra range bits of ITS specification)
        al->SetTagData(JPGTAG_IMAGE_PRECISION, precision_of(adec, a));
        al->SetTagData(JPGTAG_IMAGE_IS_FLOAT, a.xt && a.is_float ? 1 : 0);
        al->SetTagData(JPGTAG_IMAGE_OUTPUT_CONVERSION, a.xt && a.is_float ? 1 : 0);
      }
    }
  } else {
    if (alphatag) alphatag->ti_Tag = JPGTAG_TAG_IGNORE;
    if (alphalist) alphalist->ti_Tag = JPGTAG_TAG_IGNORE;
  }
  return JPG_TRUE;
} catch (...) { m_pImpl->caught("JPEG::GetInformation"); return JPG_FALSE; }

JPG_LONG JPEG::DisplayRectangle(struct JPG_TagItem *tags)
try {
  Impl *p = m_pImpl;
  p->err = 0;
  if (!p->loaded) return p->fail(JPGERR_OBJECT_DOESNT_EXIST, "no image loaded that could be reconstructed");
  const mijpeg_info &f = p->info;
  // codestream/rectanglerequest.cpp:62-190: defaults = whole canvas, requests are clipped, negatives are errors
  JPG_LONG minx = 0, miny = 0, maxx = f.width - 1, maxy = f.height - 1, c0 = 0, c1 = f.components - 1;
  bool upsample = true, ctrafo = true, device_bitmaps = false, include_alpha = false;
  struct JPG_Hook *bmh = nullptr, *alphahook = nullptr;
  for (const struct JPG_TagItem *t = tags ? tags->FirstTagItem() : nullptr; t; t = t->NextTagItem()) {
    const JPG_LONG v = t->ti_Data.ti_lData;
    switch (t->ti_Tag) {
    case JPGTAG_DECODER_MINX: if (v < 0) return p->fail(JPGERR_OVERFLOW_PARAMETER, "Rectangle MinX underflow, must be >= 0"); if (v > minx) minx = v; break;
    case JPGTAG_DECODER_MINY: if (v < 0) return p->fail(JPGERR_OVERFLOW_PARAMETER, "Rectangle MinY underflow, must be >= 0"); if (v > miny) miny = v; break;
    case JPGTAG_DECODER_MAXX: if (v < 0) return p->fail(JPGERR_OVERFLOW_PARAMETER, "Rectangle MaxX underflow, must be >= 0"); if (v < maxx) maxx = v; break;
    case JPGTAG_DECODER_MAXY: if (v < 0) return p->fail(JPGERR_OVERFLOW_PARAMETER, "Rectangle MaxY underflow, must be >= 0"); if (v < maxy) maxy = v; break;
    case JPGTAG_DECODER_MINCOMPONENT: if (v < 0 || v > 65535) return p->fail(JPGERR_OVERFLOW_PARAMETER, "MinComponent overflow, must be >= 0 && < 65536"); if (v > c0) c0 = v; break;
    case JPGTAG_DECODER_MAXCOMPONENT: if (v < 0 || v > 65535) return p->fail(JPGERR_OVERFLOW_PARAMETER, "MaxComponent overflow, must be >= 0 && < 65536"); if (v < c1) c1 = v; break;
    case JPGTAG_DECODER_UPSAMPLE: upsample = v != 0; break;
    case JPGTAG_MATRIX_LTRAFO: ctrafo = v != JPGFLAG_MATRIX_COLORTRANSFORMATION_NONE; break;
    case JPGTAG_BIH_HOOK: bmh = (struct JPG_Hook *)t->ti_Data.ti_pPtr; break;
    case JPGTAG_BIH_ALPHAHOOK: alphahook = (struct JPG_Hook *)t->ti_Data.ti_pPtr; break;
    case JPGTAG_DECODER_INCLUDE_ALPHA: include_alpha = v != 0; break;
    case JPGTAG_MIJPEG_DEVICE_BITMAPS: device_bitmaps = v != 0; break;
    default: break;
    }
  }
  if (minx > maxx || miny > maxy || c0 > c1) return JPG_TRUE; // empty request: nothing to do
  if (!upsample) {
    // codestream/rectanglerequest.cpp:157-159, control/bitmapctrl.cpp:273-294
    ctrafo = false;
    if (c0 != c1)
      return p->fail(JPGERR_INVALID_PARAMETER, "if upsampling is disabled, components can only be reconstructed one by one");
  }
  if (!bmh) return p->fail(JPGERR_OBJECT_DOESNT_EXIST, "no bitmap hook (JPGTAG_BIH_HOOK) specified");

  // REQUEST: one hook call per component with the tag layout of interface/bitmaphook.cpp:130-161
  struct Bitmap { void *mem; JPG_LONG width, height, bpr, bpp, type; void *user; } bm[MIJPEG_MAX_COMPONENTS];
  auto call_hook = [&](struct JPG_Hook *hook, const mijpeg_info &g, bool alpha, int c, int action, Bitmap &b) -> JPG_LONG {
    const int sx = g.subx[c], sy = g.suby[c];
    struct JPG_TagItem ht[] = {
        JPG_ValueTag(JPGTAG_BIO_ACTION, action),
        JPG_PointerTag(JPGTAG_BIO_MEMORY, b.mem),
        JPG_ValueTag(JPGTAG_BIO_WIDTH, b.width),
        JPG_ValueTag(JPGTAG_BIO_HEIGHT, b.height),
        JPG_ValueTag(JPGTAG_BIO_BYTESPERROW, b.bpr),
        JPG_ValueTag(JPGTAG_BIO_BYTESPERPIXEL, b.bpp),
        JPG_ValueTag(JPGTAG_BIO_PIXELTYPE, b.type),
        JPG_ValueTag(JPGTAG_BIO_ROI, 0),
        JPG_ValueTag(JPGTAG_BIO_COMPONENT, c),
        JPG_PointerTag(JPGTAG_BIO_USERDATA, b.user),
        JPG_ValueTag(JPGTAG_BIO_MINX, minx),
        JPG_ValueTag(JPGTAG_BIO_MINY, miny),
        JPG_ValueTag(JPGTAG_BIO_MAXX, maxx),
        JPG_ValueTag(JPGTAG_BIO_MAXY, maxy),
        JPG_ValueTag(JPGTAG_BIO_ALPHA, alpha ? 1 : 0),
        JPG_ValueTag(JPGTAG_BIO_PIXEL_MINX, (minx + sx - 1) / sx),
        JPG_ValueTag(JPGTAG_BIO_PIXEL_MINY, (miny + sy - 1) / sy),
        JPG_ValueTag(JPGTAG_BIO_PIXEL_MAXX, (maxx + sx) / sx - 1),
        JPG_ValueTag(JPGTAG_BIO_PIXEL_MAXY, (maxy + sy) / sy - 1),
        JPG_ValueTag(JPGTAG_BIO_PIXEL_XORG, 0),
        JPG_ValueTag(JPGTAG_BIO_PIXEL_YORG, 0),
        JPG_EndTag};
    const JPG_LONG r = hook ? hook->CallLong(ht) : 0; // (no hook: the bitmap stays blank, interface/bitmaphook.cpp:186-190)
    if (action == JPGFLAG_BIO_REQUEST) {
      b.mem = ht[1].ti_Data.ti_pPtr;
      b.width = ht[2].ti_Data.ti_lData;
      b.height = ht[3].ti_Data.ti_lData;
      b.bpr = ht[4].ti_Data.ti_lData;
      b.bpp = ht[5].ti_Data.ti_lData;
      b.type = ht[6].ti_Data.ti_lData;
      b.user = ht[9].ti_Data.ti_pPtr;
    }
    return r;
  };
  mijpeg_bitmap maps[MIJPEG_MAX_COMPONENTS];
  memset(maps, 0, sizeof(maps));
  for (int c = c0; c <= c1; c++) {
    bm[c] = Bitmap{nullptr, 0, 0, 0, 0, f.sample_bytes == 2 ? CTYP_UWORD : CTYP_UBYTE, nullptr};
    const JPG_LONG r = call_hook(bmh, f, false, c, JPGFLAG_BIO_REQUEST, bm[c]);
    if (r < 0) return p->fail(r, "BitMapHook signalled an error");
    const JPG_LONG want = f.sample_bytes == 2 ? CTYP_UWORD : CTYP_UBYTE;
    if (bm[c].type != want && bm[c].type != 0) // control/bitmapctrl.cpp:152-158: types must fit the data
      return p->fail(JPGERR_INVALID_PARAMETER, "pixel type of the user bitmap does not fit the sample precision of the image");
    maps[c].data = bm[c].type ? bm[c].mem : nullptr; // pixel type 0 = "no memory for this component"
    maps[c].bytes_per_pixel = bm[c].bpp;
    maps[c].bytes_per_row = bm[c].bpr;
    // blockbitmaprequester.cpp:1229-1244: the reported height bounds the reconstructed block rows; a blank bitmap puts no
    // constraint on the dimensions (interface/imagebitmap.cpp:118-121)
    maps[c].width = bm[c].type ? (uint32_t)bm[c].width : 0x7fffffffu;
    maps[c].height = (uint32_t)bm[c].height;
  }
  // The alpha channel beside the picture (Image::ReconstructRegion, codestream/image.cpp:1087-1123): its one component is
  // requested from the alpha hook behind the picture's components, reconstructed behind the picture, released in front of it
  mijpeg_decoder *adec = include_alpha ? mijpeg_alpha_channel(p->dec) : nullptr;
  mijpeg_info ainfo;
  Bitmap abm{nullptr, 0, 0, 0, 0, 0, nullptr};
  mijpeg_bitmap amaps[MIJPEG_MAX_COMPONENTS];
  memset(amaps, 0, sizeof(amaps));
  if (adec && mijpeg_get_info(adec, &ainfo) != MIJPEG_OK) adec = nullptr;
  if (adec) {
    const JPG_LONG want = ainfo.sample_bytes == 2 ? CTYP_UWORD : CTYP_UBYTE;
    abm.type = want;
    const JPG_LONG r = call_hook(alphahook, ainfo, true, 0, JPGFLAG_BIO_REQUEST, abm);
    if (r < 0) return p->fail(r, "BitMapHook signalled an error");
    if (abm.type != want && abm.type != 0)
      return p->fail(JPGERR_INVALID_PARAMETER, "pixel type of the user bitmap does not fit the sample precision of the image");
    amaps[0].data = abm.type ? abm.mem : nullptr;
    amaps[0].bytes_per_pixel = abm.bpp;
    amaps[0].bytes_per_row = abm.bpr;
    amaps[0].width = abm.type ? (uint32_t)abm.width : 0x7fffffffu;
    amaps[0].height = (uint32_t)abm.height;
  }
  // the request joins the sequence of requests this object has seen: row cursors and upsampler buffers of the reference's
  // BlockBitmapRequester carry over from call to call, see mijpeg_display_rect
  int arc = 0;
  const int rc = mijpeg_display_rect(p->dec, minx, miny, maxx, maxy, c0, c1,
                                     (ctrafo ? 0 : MIJPEG_FLAG_NO_COLOR_TRANSFORM) | (device_bitmaps ? MIJPEG_FLAG_DEVICE_OUTPUT : 0) |
                                         (upsample ? 0 : MIJPEG_FLAG_NO_UPSAMPLING),
                                     maps);
  if (adec && !rc)
    arc = mijpeg_display_rect(adec, minx, miny, maxx, maxy, 0, 0,
                              (ctrafo ? 0 : MIJPEG_FLAG_NO_COLOR_TRANSFORM) | (device_bitmaps ? MIJPEG_FLAG_DEVICE_OUTPUT : 0) |
                                  (upsample ? 0 : MIJPEG_FLAG_NO_UPSAMPLING),
                              amaps);
  // RELEASE is always delivered, also after a failure, so the client can let go of its buffers
  JPG_LONG hookerr = 0;
  if (adec) {
    const JPG_LONG r = call_hook(alphahook, ainfo, true, 0, JPGFLAG_BIO_RELEASE, abm);
    if (r < 0) hookerr = r;
  }
  for (int c = c0; c <= c1; c++) {
    const JPG_LONG r = call_hook(bmh, f, false, c, JPGFLAG_BIO_RELEASE, bm[c]);
    if (r < 0 && !hookerr) hookerr = r;
  }
  if (rc) return p->fail_from_decoder(rc);
  if (arc) {
    const char *m = nullptr;
    mijpeg_last_error(adec, &m);
    return p->fail(arc, m ? m : "the alpha channel does not reconstruct");
  }
  if (hookerr) return p->fail(hookerr, "BitMapHook signalled an error");
  return JPG_TRUE;
} catch (...) { m_pImpl->caught("JPEG::DisplayRectangle"); return JPG_FALSE; }

JPG_LONG JPEG::LastError(const char *&error)
{
  error = m_pImpl->err ? m_pImpl->errmsg.c_str() : nullptr;
  return m_pImpl->err;
}

JPG_LONG JPEG::LastWarning(const char *&warning)
{
  const char *m = nullptr;
  const int code = m_pImpl->loaded ? mijpeg_last_warning(m_pImpl->dec, &m) : 0;
  warning = m;
  return code;
}

JPG_LONG JPEG::ProvideImage(struct JPG_TagItem *tags)
try {
  Impl *p = m_pImpl;
  p->err = 0;
  if (!tags) return p->fail(JPGERR_MISSING_PARAMETER, "JPEG::ProvideImage requires a tag list");
  if (p->picture.empty()) { // first call: the image parameters (interface/jpeg.cpp:461-560, codestream/encoder.cpp)
    const JPG_LONG w = tags->GetTagData(JPGTAG_IMAGE_WIDTH, 0), h = tags->GetTagData(JPGTAG_IMAGE_HEIGHT, 0), depth = tags->GetTagData(JPGTAG_IMAGE_DEPTH, 3);
    const JPG_LONG prec = tags->GetTagData(JPGTAG_IMAGE_PRECISION, 8), type = tags->GetTagData(JPGTAG_IMAGE_FRAMETYPE, JPGFLAG_BASELINE);
    if (w < 1 || h < 1 || w > 65535 || h > 65535) return p->fail(JPGERR_OVERFLOW_PARAMETER, "image dimensions must be between 1 and 65535");
    if ((depth != 1 && depth != 3) || prec != 8) return p->fail(JPGERR_NOT_IMPLEMENTED, "the accelerated encoder handles 8-bit images of one or three components");
    if ((type & ~JPGFLAG_OPTIMIZE_HUFFMAN) != JPGFLAG_BASELINE && (type & ~JPGFLAG_OPTIMIZE_HUFFMAN) != JPGFLAG_SEQUENTIAL)
      return p->fail(JPGERR_NOT_IMPLEMENTED, "the accelerated encoder writes baseline / sequential Huffman frames only");
    p->enc_width = w; p->enc_height = h; p->enc_depth = depth;
    p->enc_optimize = (type & JPGFLAG_OPTIMIZE_HUFFMAN) != 0;
    p->enc_quality = tags->GetTagData(JPGTAG_IMAGE_QUALITY, 75);
    p->enc_restart = tags->GetTagData(JPGTAG_IMAGE_RESTART_INTERVAL, 0);
    p->enc_ycbcr = tags->GetTagData(JPGTAG_MATRIX_LTRAFO, JPGFLAG_MATRIX_COLORTRANSFORMATION_YCBCR) != JPGFLAG_MATRIX_COLORTRANSFORMATION_NONE;
    if (depth == 3 && !p->enc_ycbcr) return p->fail(JPGERR_NOT_IMPLEMENTED, "the accelerated encoder codes three components as YCbCr");
    const unsigned char *subx = (const unsigned char *)tags->GetTagPtr(JPGTAG_IMAGE_SUBX, nullptr), *suby = (const unsigned char *)tags->GetTagPtr(JPGTAG_IMAGE_SUBY, nullptr);
    int mx = 1, my = 1;
    for (int c = 0; c < depth; c++) {
      const int sx = subx ? subx[c] : 1, sy = suby ? suby[c] : 1;
      if (sx < 1 || sx > 4 || sy < 1 || sy > 4) return p->fail(JPGERR_OVERFLOW_PARAMETER, "subsampling factors must be between 1 and 4");
      mx = sx > mx ? sx : mx;
      my = sy > my ? sy : my;
    }
    for (int c = 0; c < depth; c++) { // subsampling factors -> sampling factors (marker/frame.cpp)
      const int sx = subx ? subx[c] : 1, sy = suby ? suby[c] : 1;
      if (mx % sx || my % sy) return p->fail(JPGERR_INVALID_PARAMETER, "unsupported combination of subsampling factors");
      p->enc_hsamp[c] = mx / sx;
      p->enc_vsamp[c] = my / sy;
    }
    p->picture.assign((size_t)w * (size_t)h * (size_t)depth, 0);
    p->enc_lines = 0;
  }
  struct JPG_Hook *bmh = (struct JPG_Hook *)tags->GetTagPtr(JPGTAG_BIH_HOOK, nullptr);
  if (!bmh) return p->fail(JPGERR_OBJECT_DOESNT_EXIST, "no bitmap hook (JPGTAG_BIH_HOOK) specified");
  const bool loop = tags->GetTagData(JPGTAG_ENCODER_LOOP_ON_INCOMPLETE, 0) != 0;
  const int w = p->enc_width, h = p->enc_height, nc = p->enc_depth;
  // eight lines per request, component by component (control/blockbitmaprequester.cpp:968-1011, interface/bitmaphook.cpp:130-248)
  do {
    if (p->enc_lines >= h) break;
    const JPG_LONG miny = p->enc_lines, maxy = (miny + 7 < h ? miny + 7 : h - 1);
    for (int c = 0; c < nc; c++) {
      struct JPG_TagItem ht[] = {
          JPG_ValueTag(JPGTAG_BIO_ACTION, JPGFLAG_BIO_REQUEST),
          JPG_PointerTag(JPGTAG_BIO_MEMORY, nullptr),
          JPG_ValueTag(JPGTAG_BIO_WIDTH, 0),
          JPG_ValueTag(JPGTAG_BIO_HEIGHT, 0),
          JPG_ValueTag(JPGTAG_BIO_BYTESPERROW, 0),
          JPG_ValueTag(JPGTAG_BIO_BYTESPERPIXEL, 0),
          JPG_ValueTag(JPGTAG_BIO_PIXELTYPE, CTYP_UBYTE),
          JPG_ValueTag(JPGTAG_BIO_ROI, 0),
          JPG_ValueTag(JPGTAG_BIO_COMPONENT, c),
          JPG_PointerTag(JPGTAG_BIO_USERDATA, nullptr),
          JPG_ValueTag(JPGTAG_BIO_MINX, 0),
          JPG_ValueTag(JPGTAG_BIO_MINY, miny),
          JPG_ValueTag(JPGTAG_BIO_MAXX, w - 1),
          JPG_ValueTag(JPGTAG_BIO_MAXY, maxy),
          JPG_ValueTag(JPGTAG_BIO_ALPHA, 0),
          JPG_ValueTag(JPGTAG_BIO_PIXEL_MINX, 0),
          JPG_ValueTag(JPGTAG_BIO_PIXEL_MINY, miny),
          JPG_ValueTag(JPGTAG_BIO_PIXEL_MAXX, w - 1),
          JPG_ValueTag(JPGTAG_BIO_PIXEL_MAXY, maxy),
          JPG_ValueTag(JPGTAG_BIO_PIXEL_XORG, 0),
          JPG_ValueTag(JPGTAG_BIO_PIXEL_YORG, 0),
          JPG_EndTag};
      JPG_LONG r = bmh->CallLong(ht);
      if (r < 0) return p->fail(r, "BitMapHook signalled an error");
      const unsigned char *mem = (const unsigned char *)ht[1].ti_Data.ti_pPtr;
      const JPG_LONG bpr = ht[4].ti_Data.ti_lData, bpp = ht[5].ti_Data.ti_lData, type = ht[6].ti_Data.ti_lData;
      if (!mem || type != CTYP_UBYTE) return p->fail(JPGERR_INVALID_PARAMETER, "the accelerated encoder expects CTYP_UBYTE pixel data from the bitmap hook");
      for (JPG_LONG y = miny; y <= maxy; y++) { // mem is the address of canvas pixel (0,0)
        const unsigned char *src = mem + (ptrdiff_t)y * bpr;
        unsigned char *dst = p->picture.data() + ((size_t)y * (size_t)w) * (size_t)nc + (size_t)c;
        for (int x = 0; x < w; x++) dst[(size_t)x * (size_t)nc] = src[(ptrdiff_t)x * bpp];
      }
      ht[0].ti_Data.ti_lData = JPGFLAG_BIO_RELEASE;
      r = bmh->CallLong(ht);
      if (r < 0) return p->fail(r, "BitMapHook signalled an error");
    }
    p->enc_lines = maxy + 1;
  } while (loop);
  tags->SetTagData(JPGTAG_ENCODER_IMAGE_COMPLETE, p->enc_lines >= h);
  return JPG_TRUE;
} catch (...) { m_pImpl->caught("JPEG::ProvideImage"); return JPG_FALSE; }

JPG_LONG JPEG::Write(struct JPG_TagItem *tags)
try {
  Impl *p = m_pImpl;
  p->err = 0;
  if (!tags) return p->fail(JPGERR_MISSING_PARAMETER, "JPEG::Write requires a tag list with an I/O hook");
  if (p->picture.empty() || p->enc_lines < p->enc_height) return p->fail(JPGERR_OBJECT_DOESNT_EXIST, "no complete image has been provided that could be written");
  struct JPG_Hook *io = (struct JPG_Hook *)tags->GetTagPtr(JPGTAG_HOOK_IOHOOK);
  if (!io) return p->fail(JPGERR_MISSING_PARAMETER, "no I/O hook (JPGTAG_HOOK_IOHOOK) specified");
  uint8_t *stream = nullptr;
  size_t size = 0;
  const int rc = mijpeg_encode_image(p->dec, p->picture.data(), p->enc_width, p->enc_height, p->enc_depth, (int64_t)p->enc_width * p->enc_depth, p->enc_quality,
                                     p->enc_hsamp, p->enc_vsamp, p->enc_restart, p->enc_optimize ? 1 : 0, &stream, &size);
  if (rc) return p->fail_from_decoder(rc);
  for (size_t at = 0; at < size;) { // io/iostream.cpp: the stream goes out through the hook in pieces
    const size_t n = size - at < ((size_t)1 << 20) ? size - at : ((size_t)1 << 20);
    struct JPG_TagItem iotags[] = {
        JPG_PointerTag(JPGTAG_FIO_HANDLE, tags->GetTagPtr(JPGTAG_HOOK_IOSTREAM)),
        JPG_PointerTag(JPGTAG_FIO_BUFFER, stream + at),
        JPG_ValueTag(JPGTAG_FIO_SIZE, (JPG_LONG)n),
        JPG_ValueTag(JPGTAG_FIO_ACTION, JPGFLAG_ACTION_WRITE),
        JPG_ValueTag(JPGTAG_FIO_SEEKMODE, JPGFLAG_OFFSET_CURRENT),
        JPG_ValueTag(JPGTAG_FIO_OFFSET, 0),
        JPG_PointerTag(JPGTAG_FIO_USERDATA, io->hk_pData),
        JPG_EndTag};
    const JPG_LONG put = io->CallLong(iotags);
    if (put != (JPG_LONG)n) {
      mijpeg_free(stream);
      return p->fail(put < 0 ? put : JPGERR_INVALID_PARAMETER, "the I/O hook did not take the data");
    }
    at += n;
  }
  mijpeg_free(stream);
  return JPG_TRUE;
} catch (...) { m_pImpl->caught("JPEG::Write"); return JPG_FALSE; }

// interface/jpeg.cpp:505-575: the 16 bits at the position reading stopped at, 0 for the markers that can only be handled by
// the library (frame and scan headers, EOI, DHP), -1 at the end of the data or when no decoding is in progress.
JPG_LONG JPEG::PeekMarker(struct JPG_TagItem *)
{
  Impl *p = m_pImpl;
  if (!p->pulled) { p->fail(JPGERR_OBJECT_DOESNT_EXIST, "decoding not in progress"); return -1; }
  const long m = p->peekword();
  switch (m) {
  case 0xffc0: case 0xffc1: case 0xffc2: case 0xffc3: case 0xffc5: case 0xffc6: case 0xffc7: case 0xffc8: case 0xffc9: case 0xffca:
  case 0xffcb: case 0xffcd: case 0xffce: case 0xffcf: case 0xffb1: case 0xffb2: case 0xffb3: case 0xffb9: case 0xffba: case 0xffbb:
  case 0xffd9: case 0xffda: case 0xffde: case 0xfff7:
    return 0;
  default: return (JPG_LONG)m;
  }
}

// interface/jpeg.cpp:577-611: the client takes bytes out of the stream itself; the library never sees them.
JPG_LONG JPEG::ReadMarker(void *buffer, JPG_LONG bufsize, struct JPG_TagItem *)
try {
  Impl *p = m_pImpl;
  if (!p->pulled) { p->fail(JPGERR_OBJECT_DOESNT_EXIST, "decoding not in progress"); return -1; }
  if (p->phase == Impl::P_DONE || bufsize < 0 || !buffer) return bufsize == 0 ? 0 : -1;
  const size_t left = p->stream.size() - p->cursor;
  const size_t n = (size_t)bufsize < left ? (size_t)bufsize : left;
  memcpy(buffer, p->stream.data() + p->cursor, n);
  p->take(n);
  return (JPG_LONG)n;
} catch (...) { m_pImpl->caught("JPEG::ReadMarker"); return -1; }

// interface/jpeg.cpp:613-645
JPG_LONG JPEG::SkipMarker(JPG_LONG bytes, struct JPG_TagItem *)
try {
  Impl *p = m_pImpl;
  if (!p->pulled) { p->fail(JPGERR_OBJECT_DOESNT_EXIST, "decoding not in progress"); return -1; }
  if (p->phase == Impl::P_DONE) return 0;
  if (bytes > 0) {
    const size_t left = p->stream.size() - p->cursor;
    p->take((size_t)bytes < left ? (size_t)bytes : left);
  }
  return 0;
} catch (...) { m_pImpl->caught("JPEG::SkipMarker"); return -1; }
JPG_LONG JPEG::WriteMarker(void *, JPG_LONG, struct JPG_TagItem *) { m_pImpl->fail(JPGERR_NOT_IMPLEMENTED, "the encoder is not part of the accelerated path"); return -1; }
