// host_decoder.hpp -- host side of the path: marker parsing and restart-interval-parallel Huffman
// decoding into planar int16 coefficient planes.  No HIP in here: this part also runs (and is
// tested) on a box without a GPU.
//
// Reference behaviour reproduced, damaged streams included (the state machine of JPEG::ReadInternal with its
// warn-and-continue paths and the restart marker resynchronisation of codestream/entropyparser.cpp:117-201; see the
// RefWalker class in host_decoder.cpp):
//   codestream/tables.cpp:1003-...      marker dispatch (DQT, DHT, DRI, APP14)
//   marker/quantization.cpp:474-537     DQT, stored de-zigzagged
//   coding/huffmantemplate.cpp:802-905  DHT -> decoder tables
//   marker/frame.cpp:111-..             SOF0/SOF1
//   marker/scan.cpp:163-..              SOS
//   codestream/sequentialscan.cpp:381-428, 678-773   ParseMCU / DecodeBlock
//   codestream/entropyparser.cpp:117-135             restart markers
//   io/bitstream.cpp:56-118                          byte stuffing, zero bits at a marker
#ifndef MIJ_HOST_DECODER_HPP
#define MIJ_HOST_DECODER_HPP

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <functional>
#include <string>
#include <memory>
#include <vector>

#include "../../include/mijpeg.h"

namespace mij {

struct HuffTable {
  bool defined = false;
  bool built = false; // decoder tables below are valid (built when a scan first uses the table, as the reference does)
  bool oversize = false; // the DHT segment listed more than 256 values
  uint8_t counts[16] = {0};
  uint8_t values[256] = {0};
  // decoder: LOOKAHEAD-bit direct table, entry = (length << 8) | symbol, 0 = longer code
  static constexpr int LOOKAHEAD = 10;
  uint16_t fast[1 << LOOKAHEAD];
  int32_t maxcode[18]; // maxcode[l] for codes of length l (left-aligned compare uses plain codes)
  int32_t valoff[17];  // values index = code + valoff[l]
  bool build(); // false: the code lengths over-subscribe the code space
  bool same_code(const HuffTable &o) const { return !memcmp(counts, o.counts, sizeof(counts)) && !memcmp(values, o.values, sizeof(values)); }
};

struct ScanComponent {
  int comp = 0; // index into the frame's components
  int td = 0, ta = 0;
};

struct Scan {
  int ncomp = 0;
  ScanComponent sc[MIJPEG_MAX_COMPONENTS];
  HuffTable dc[MIJPEG_MAX_COMPONENTS], ac[MIJPEG_MAX_COMPONENTS]; // snapshot per scan component
  int restart_interval = 0;
  int ss = 0, se = 63, ah = 0, al = 0; // spectral selection and successive approximation (progressive frames)
  int mcus_x = 0, mcus_y = 0; // MCU grid of THIS scan
  size_t ecs_begin = 0, ecs_end = 0; // entropy coded data [begin, end) in the input
  std::vector<size_t> interval_begin; // byte offset of every restart interval (first = ecs_begin)
  // What the device decoder reads is the entropy coded data WITHOUT the byte stuffing (unstuff_scan below): per interval the
  // number of bytes that leaves -- up to the first FF that is not followed by 00 (the marker, or a fill byte in front of it:
  // the reference's bit reader stops there, io/bitstream.cpp:96-101), FF 00 counted once -- and, for intervals that span
  // several search chunks, positions inside them with the number of stuffed pairs in front (so that pieces of a long
  // interval can be copied in parallel)
  std::vector<uint32_t> interval_ulen;
  std::vector<uint32_t> interval_ubegin, interval_uend; // ... and where that puts every interval in the copy (back to back)
  size_t unstuffed_size = 0;
  const uint8_t *unstuffed_at = nullptr; // the copy already exists here: the marker search wrote it as it went (set_unstuff_sink)
  struct StuffCheckpoint { size_t pos; uint32_t interval; uint32_t pairs; };
  std::vector<StuffCheckpoint> stuff_ckpt;
  const uint8_t *base = nullptr;      // stream the offsets refer to; null = the decoder's input (hidden scans live in boxes)
  // what the reference's parser for this scan is (marker/scan.cpp:355-470, codestream/sequentialscan.cpp:72-94)
  bool refinement = false;      // RefinementScan instead of SequentialScan
  bool progressive_run = false; // EOB runs are legal (m_bProgressive)
  bool residual = false;        // the residual scan type of part 8 (SequentialScan(.., true, true), marker/scan.cpp:483-489)
  int lowbit = 0;               // point transform incl. hidden bits
};

struct StreamError {
  int code = 0;
  std::string message;
};

// One JPEG XT box reassembled from its APP11 segments (boxes/box.cpp:88-200)
struct XtBox {
  uint32_t type = 0;
  uint16_t en = 0;
  std::vector<uint8_t> data;
  uint64_t boxsize = 0;  // payload bytes the box header announces
  uint64_t parsed = 0;   // payload bytes its segments announced so far (m_uqParsedBytes: counted even where the file ends inside one)
  bool complete = false; // all of them announced: only then the reference's tables know the box (codestream/tables.cpp:1191-1283)
  // A codestream box of many segments (RESI, FINE, RFIN, the alpha kinds) is not copied segment by segment while the walk goes
  // over the file: the walk notes where its pieces lie, and HostDecoder::materialize_boxes copies them -- all boxes' pieces, on
  // the pool -- when the walk is through.  Nothing reads such a box before that.
  struct Piece { const uint8_t *src; size_t len, pad; }; // (pad: bytes the file no longer had: zeros, io/decoderstream.cpp:136-158)
  std::vector<Piece> pieces;
  size_t pending = 0; // bytes of the pieces, padding included
  bool deferred = false;
};

// "Virtual restart intervals" of a scan without restart markers: exact restart points (byte, bits to skip, DC
// predictors) every mcus_per_interval MCUs, found by the self-synchronising walk of the host decoder; the device
// kernel then decodes the scan as if it had restart markers there.
struct VirtualIntervals {
  int mcus_per_interval = 0;
  std::vector<uint32_t> byte_off; // offset into the stream of the first byte to load
  std::vector<uint8_t> bit_skip;  // bits of that byte that belong to the previous block (0..7)
  std::vector<int16_t> pred;      // 4 per interval: DC predictors of the scan components
};

class HostDecoder {
public:
  HostDecoder();
  ~HostDecoder();
  // Parse SOI .. EOI structure: fills info, scans (with their restart-interval offsets).
  // header_only stops after the first SOS header has been seen.
  int parse(const uint8_t *data, size_t size, bool header_only);

  // Entropy-decode every scan into `coef` (info.coef_count int16, layout info.coef_offset).
  // on_rows_done(first_mcu_row, end_mcu_row) is called from the calling thread, in order, as
  // bands of frame MCU rows become final (while later bands are still being decoded when the
  // last scan is interleaved; otherwise once at the end).  Used for streaming uploads; may be empty.
  int decode(int16_t *coef, int threads, const std::function<void(int, int)> &on_rows_done);
  // The same into int32 planes (same plane offsets, counted in int32 elements), for the streams decode() turns down with
  // MIJPEG_ERR_OVERFLOW_PARAMETER: damaged ones whose DC prediction (or point transform) leaves the 16-bit range.  The
  // reference keeps LONG coefficients and reconstructs whatever they hold.  Plain JPEG only.  On success info.coef_wide
  // is set and info.coef_offset[] / coef_count count the two int16 slots of every coefficient.
  int decode_wide(int32_t *coef, int threads);
  // The stream is damaged in a way the parallel decoders (host and device) must not touch: restart markers missing,
  // out of sequence or in excess, or a sequential scan with a point transform.  decode() then walks the stream
  // sequentially the way the reference does (resynchronisation, grey intervals; RefWalker in host_decoder.cpp).
  bool needs_sequential() const { return needs_sequential_; }
  // The scans a frame's SEQUENTIAL walk met (decode_sequential: damaged streams, DNL frames, the residual scan types of part 8):
  // where their entropy coded data begins and ends and their MCU grid -- what the stop loops of class JPEG walk for frames that
  // planned no scans.  walked(): the last decode went that way.
  struct WalkedScan { size_t begin, end; int mcus_x, mcus_y; bool boxed; };
  bool walked() const { return walked_; }
  const std::vector<WalkedScan> &walked_scans() const { return walked_scans_; }
  // conditions the reference only warns about that the last parse / decode passed (stray markers, resynchronisation ...)
  int warnings() const { return warnings_; }
  // JPEG::LastWarning: 0, or the code of what the reference warns about at this stream (message in *msg): the LCHK checksum of
  // a JPEG XT file that does not fit its legacy codestream (interface/jpeg.cpp:222-238), or a damaged codestream the decoder
  // resynchronised in.  The checksum is computed when this is first asked for.
  int last_warning(const char **msg);

  // Walk scan `scan` (Huffman sequential, no restart markers needed) speculatively in parallel and return its virtual
  // restart intervals; nonzero if the scan does not lend itself to it (the caller then decodes on the host).
  int plan_virtual_intervals(size_t scan, int mcus_per_interval, int threads, VirtualIntervals &out);

  // The entropy coded data of scan `scan` without its byte stuffing and without the markers: interval k's bytes at
  // Scan::interval_ubegin[k] (relative to the start of the copy), interval_ulen[k] of them, back to back, Scan::unstuffed_size
  // in all.  unstuff_pieces lists the copy as independent pieces (each at most ~piece_bytes of source) that unstuff_piece
  // carries out -- the callers spread them over their workers.
  struct UnstuffPiece { uint32_t k0, k1; size_t src0, src1; size_t dst; };
  // Batches (one stream per worker anyway): let the marker search of the FIRST scan of the next parse() write the copy while
  // it walks the segment -- one pass over the stream instead of two.  `capacity` bytes at dst (the stream's size is enough).
  // Scan::unstuffed_at tells whether it did (it does not when the search runs in parallel chunks).
  void set_unstuff_sink(uint8_t *dst, size_t capacity) { sink_ = dst; sink_cap_ = capacity; }
  void unstuff_pieces(size_t scan, size_t piece_bytes, std::vector<UnstuffPiece> &out) const;
  void unstuff_piece(size_t scan, const UnstuffPiece &p, uint8_t *dst) const;

  const uint8_t *stream_base() const { return data_; } // the parsed input
  size_t stream_size() const { return size_; }
  HostDecoder *residual() const { return residual_; } // JPEG XT: decoder of the residual codestream
  // JPEG XT, for one decode(): block rows [y0, y1) of residual component c hold their final coefficients -- called from a
  // worker thread while the decode goes on (the last refinement window's appliers, decode_t), so that the owner can send them
  // on their way; residual_rows_reported(c) = rows it has been told about when decode() returns (0: none, or told and taken
  // back -- a decode that fell back to the sequential walk)
  void set_residual_rows_callback(std::function<void(int, int, int)> cb) { final_rows_cb_ = std::move(cb); }
  int residual_rows_reported(int c) const { return residual_ ? residual_->rows_reported_[c].load() : 0; }
  int hidden_bits() const { return hidden_; }
  // hidden refinement scans follow the visible ones (with a merging specification that never arrived they refine zero hidden bits)
  bool has_hidden_scans() const { return hidden_ > 0 || !hidden_src_.empty(); }
  // JPEG XT: the legacy codestream came to its EOI, i.e. the reference merges the residual codestream (else nothing: see decode_t)
  bool residual_merged() const { return eoi_image_; }
  // what decode() would still have to say about this file beside decoding it: a residual codestream whose header is refused, a
  // quantiser table of the residual image that is looked up at the first request (the device decoders leave such files to it)
  bool verdict_pending() const { return residual_error_.code != 0 || late_error_.code != 0 || (residual_ && residual_->late_quant_missing_ != nullptr); }
  // a component that appears in no scan reconstructs from a stand-in block (fill_unseen_components): the host decoder's business
  bool every_component_seen() const
  {
    for (int c = 0; c < info.components; c++)
      if (!comp_seen_[c]) return false;
    return true;
  }

  mijpeg_info info{};
  // restart-interval byte ranges of scan i (valid after parse(..., false)): [interval_begin[k], interval_ends(i)[k])
  const std::vector<size_t> &interval_ends(size_t scan) const { return scan_interval_end_[scan]; }
  const std::vector<uint8_t> &restart_codes(size_t scan) const { return scan_rst_code_[scan]; }
  // JPEG XT profile C: parameters and the decoder of the residual codestream (RESI box); null for plain JPEG
  // (... or a merging specification that sends the legacy picture alone through the L chain: xt.no_residual, no residual())
  bool is_xt() const { return residual_ != nullptr || lonly_; }
  // JPEG XT alpha channel (after a parse of the whole file): the file has a complete ALFA box and the legacy codestream came to
  // the EOI behind which the reference turns to it (codestream/image.cpp:1430-1460); its codestream; the boxes its decoder sees;
  // compositing method (-1: no AMUL box) and matte colour of the alpha merging specification (interface/jpeg.cpp:919-945)
  bool has_alpha() const { return have_alpha_ && eoi_image_; }
  bool alpha_stream(const uint8_t **data, size_t *size) const;
  std::vector<XtBox> alpha_boxes() const;
  int alpha_mode() const { return alpha_mode_; }
  const uint32_t *alpha_matte() const { return alpha_matte_; }
  // this object decodes an alpha channel's codestream: `boxes` are the file's, translated (alpha_boxes of the file's decoder)
  void preset_boxes(std::vector<XtBox> boxes) { preset_boxes_ = std::move(boxes); alpha_child_ = true; }
  // the last decode() stopped at a coefficient beyond the 16-bit store: decode_wide() is the next step (plain JPEG), or a refusal
  bool left_16bit_store() const { return left_16bit_store_ || (residual_ && residual_->left_16bit_store_); } // (either codestream's walk)
  // the last parse failed with what the colour transformer refuses (a table or transformation that does not exist or does not fit):
  // the reference reads such a file without complaint and fails at the first request for pixels
  bool transformer_refused() const { return transformer_refused_; }
  void materialize_boxes();
  size_t segment_end(const uint8_t *base, size_t size, size_t from);
  const uint8_t *term_base_ = nullptr; // segment_end's list of segment-ending markers: of which bytes, from where on
  size_t term_size_ = 0, term_from_ = 0;
  std::vector<size_t> term_pos_;
  int declined_verdict(); // after a parse / decode that ended with -1034: the codestreams' own verdict, else -1034 again
  mijpeg_xt_params xt{};
  std::vector<Scan> scans;
  StreamError error;
  double huffman_seconds = 0;

private:
  friend class RefWalker;
  const uint8_t *data_ = nullptr;
  size_t size_ = 0;
  HuffTable dc_[4], ac_[4];
  uint16_t quant_[4][64];
  bool quant_defined_[4] = {false, false, false, false};
  bool have_quant_ = false, have_huff_ = false; // a DQT / DHT marker was seen at all
  int frame_type_ = 0;                          // 0 baseline (SOF0), 1 extended sequential (SOF1), 2 progressive (SOF2), 3 / 4 residual sequential / progressive (FFB1 / FFB2, inside a RESI box)
  // the transform of a component uses the quantiser table that was in force when the component first appeared in a
  // scan; a component that appears in no scan reconstructs as sample value 0 (control/blockbuffer.cpp:177-208,
  // control/blockbitmaprequester.cpp:1047-1054)
  bool comp_seen_[MIJPEG_MAX_COMPONENTS] = {false, false, false, false};
  uint16_t comp_quant_[MIJPEG_MAX_COMPONENTS][64];
  uint8_t *sink_ = nullptr; // set_unstuff_sink: armed for the NEXT parse() only ...
  size_t sink_cap_ = 0;
  uint8_t *active_sink_ = nullptr; // ... which takes it over (and forgets it whatever becomes of the parse)
  size_t active_cap_ = 0;
  bool needs_sequential_ = false;
  bool walked_ = false;
  std::vector<WalkedScan> walked_scans_;
  bool parsed_ = false;
  int warnings_ = 0;
  bool have_lchk_ = false;
  uint32_t lchk_value_ = 0;
  int checksum_state_ = -1; // -1: not computed yet, 0: fits, 1: mismatch
  void reset_stream_state();
  void publish_tables(bool header_only);
  template <class T> int decode_sequential(T *coef, int threads);
  template <class T> void fill_unseen_components(T *coef);
  template <class T> void range_pass(const T *coef, int threads, std::atomic<uint32_t> (&qmax_all)[MIJPEG_MAX_COMPONENTS], unsigned done = 0);
  int restart_interval_ = 0;
  int adobe_transform_ = -1;
  bool have_frame_ = false;
  bool need_dnl_ = false; // SOF carried zero lines
  // ... then the height is what the first scan's parser comes to while it decodes (EntropyParser::ParseDNLMarker looks for the
  // marker at every MCU, codestream/entropyparser.hpp:147-152): parse() decodes that scan without keeping anything to learn
  // the height -- and the block rows the scan makes (control/blockbuffer.cpp:212-265 has no bound while the height is 0) --
  // the way the reference will; decode() then knows both
  int dnl_height_ = 0;
  int dnl_rows_made_[MIJPEG_MAX_COMPONENTS] = {0, 0, 0, 0};
  bool progressive_ = false; // SOF2
  int comp_id_[MIJPEG_MAX_COMPONENTS] = {0, 0, 0, 0};
  // per scan: end offset of every restart interval and the RSTn code that terminated it
  std::vector<size_t> interval_end_;
  std::vector<uint8_t> rst_code_;
  std::vector<std::vector<size_t>> scan_interval_end_;
  std::vector<std::vector<uint8_t>> scan_rst_code_;
  // the mask planes of deferred AC refinement scans (decode_t; kept between reads, never initialised)
  std::unique_ptr<uint64_t[]> refine_scratch_;
  size_t refine_scratch_cap_ = 0;
  // ... and the residual decoder's, which is a new object with every parse, while there is none
  std::unique_ptr<uint64_t[]> spare_scratch_;
  size_t spare_scratch_cap_ = 0;
  void drop_residual();
  std::function<void(int, int, int)> final_rows_cb_;
  std::atomic<int> rows_reported_[MIJPEG_MAX_COMPONENTS] = {};
  // the byte stores of the last parse's big boxes (the residual codestream, refinement scans): the next file's boxes take them
  // over instead of growing fresh vectors segment by segment
  std::vector<std::vector<uint8_t>> box_spares_;
  size_t box_bytes_promised_ = 0; // what the boxes of this parse reserved up front: bounded by the length of the file
  std::vector<XtBox> boxes_;
  std::vector<int32_t> xt_q_[3], xt_r2_[3]; // Q / R2 tables of a JPEG XT stream when they are not the identities (xt.qtable / r2table point here)
  HostDecoder *residual_ = nullptr;
  bool lonly_ = false; // JPEG XT without a residual codestream: the L chain alone (finish_xt)
  bool have_alpha_ = false, alpha_child_ = false;
  int alpha_mode_ = -1;
  uint32_t alpha_matte_[3] = {0, 0, 0};
  std::vector<XtBox> preset_boxes_;
  bool ignore_residual_ = false; // late_verdict parses again: the legacy codestream has no EOI, the residual codestream is never looked at
  bool transformer_refused_ = false;
  int verdict_hidden_l_ = 0, verdict_hidden_r_ = 0; // RSPC's counts as finish_xt read them: for late_verdict's decoders
  const char *late_quant_missing_ = nullptr; // nested: a quantiser table of the residual image does not exist -- the first request's finding (RefWalker::scan)
  bool left_16bit_store_ = false; // the last decode stopped at a coefficient beyond the 16-bit store (OVERFLOW_PARAMETER): int32 planes next
  bool nested_ = false; // this object decodes a residual codestream
  int hidden_ = 0;      // JPEG XT: low bits of every coefficient that arrive in hidden refinement scans
  bool parsing_hidden_ = false;
  int64_t plane_offset_[MIJPEG_MAX_COMPONENTS] = {0, 0, 0, 0}; // component planes inside this frame's own store
  int finish_xt(bool header_only);
  int spec_without_residual(const XtBox &spec);
  int spec_ltrafo_ = 255; // L transformation a merging specification WITHOUT a residual names (255: none; codestream/tables.cpp:1994-2021)
  int add_hidden_scans(uint32_t type, const std::vector<XtBox> &boxes, int hidden);
  std::vector<std::pair<size_t, size_t>> seq_spans_; // what the scans of a sequentially walked stream read (RefWalker::spans)
  StreamError residual_error_; // JPEG XT: what is wrong with the residual codestream's header, reported behind the legacy frame's decode
  // JPEG XT: what the colour transformer refuses when the first request builds it (tables and transformations that do not exist
  // or do not fit).  The reference has read the whole file by then: what stops either codestream comes first (late_verdict)
  StreamError late_error_;
  bool late_residual_only_ = false; // ... about a table of the residual's side: not looked up unless there is a residual frame to merge
  bool plain_only_ = false;         // (the verdict's decoder of the legacy codestream alone: the boxes are not interpreted)
  int late_verdict();
  bool residual_unspecified_ = false; // a residual codestream and no merging specification: residual_error_ is the transformer's refusal
  bool eoi_frame_ = true, eoi_image_ = true; // the walk's frame / image trailer stood at an EOI (RefWalker::frame_eoi / image_eoi)
  std::vector<const XtBox *> hidden_src_; // this frame's hidden refinement scans in box order (elements of the legacy decoder's boxes_)
  template <class T> int decode_t(T *coef, int threads, const std::function<void(int, int)> &on_rows_done);
  template <class T> int decode_scan_speculative(T *coef, const Scan &s, int threads, uint32_t (&qmax_out)[MIJPEG_MAX_COMPONENTS],
                                                 VirtualIntervals *plan_only = nullptr, bool first_pass = false);
  int fail(int code, const char *msg);
  int frame_geometry();
  void find_intervals(Scan &s);
  void find_intervals_in(Scan &s, const uint8_t *data, size_t size, std::vector<size_t> &interval_end, std::vector<uint8_t> &rst_code, bool may_sink);
  // the marker searches of a frame's hidden refinement scans, put off until all their boxes have been walked (add_hidden_scans)
  struct DeferredSearch { size_t scan; const uint8_t *data; size_t size; int type; bool hidden; };
  std::vector<DeferredSearch> *deferred_hidden_ = nullptr;
  std::vector<DeferredSearch> deferred_scans_; // ... and of a progressive frame's scans: run when parse()'s walk is through
  void run_deferred_searches(std::vector<DeferredSearch> &searches);
  static bool defer_progressive_scans(); // (MIJPEG_NO_DEFERRED_SEARCH: A-B)
};

int default_threads();

// diagnostics: scans decoded by the self-synchronising parallel path since the library was loaded, and their pieces
extern std::atomic<int64_t> g_speculative_scans, g_speculative_pieces;

// Run fn(i) for i in [0, n) on the entropy-decoder worker pool (used for large host-side pixel copies).
void parallel_for(int n, const std::function<void(int)> &fn);

// zig-zag position -> natural index (dct/dct.cpp:57-74), generated at start-up
extern const uint8_t *scan_order();

} // namespace mij
#endif
