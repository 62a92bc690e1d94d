/*
 * jpeg_oracle.c -- plain-C restatement of the thorfdbg/libjpeg block-decode path
 * (Huffman sequential scan -> dequant + 8x8 IDCT -> centred chroma upsampling -> YCbCr->RGB).
 *
 * TEST INFRASTRUCTURE ONLY (see jpeg_oracle.h).  Written to be read next to the reference:
 * it favours literal emulation of the reference's buffer manipulations (including the in-place
 * aliasing of the horizontal upsampling filter) over speed.  Scalar, single-threaded.
 *
 * Parity status: PINNED against the reference binary (tests/test_oracle.py, tests/golden/).
 */
#include "jpeg_oracle.h"

#include <math.h>
#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Zig-zag scan order: position k of the scan -> natural index x + 8y.
 * Restates dct/dct.cpp:57-74 (DCT::ScanOrder), regenerated here by walking the diagonals.
 * ---------------------------------------------------------------------------------------- */
static int g_scan_order[64];
static int g_scan_order_ready = 0;

static void build_scan_order(void)
{
  int k = 0, d, i;
  for (d = 0; d < 15; d++) {
    /* diagonal d holds the positions with x + y == d; even diagonals run bottom-left -> top-right */
    for (i = 0; i <= d; i++) {
      int x = (d & 1) ? (d - i) : i;
      int y = d - x;
      if (x < 8 && y < 8) {
        /* the walk direction alternates: odd diagonals go from (d,0) down-left, even ones up-right */
        g_scan_order[k++] = x + (y << 3);
      }
    }
  }
  g_scan_order_ready = 1;
}

/* ------------------------------------------------------------------------------------------
 * The codestream state machine, restated from the reference *including its behaviour on damaged
 * streams*: which conditions it throws on (JPG_THROW -> the whole Read fails, no picture), which it only
 * warns about and decodes through (JPG_WARN), how the entropy parser resynchronises at restart
 * markers and how the bit reader behaves at markers and at the end of the data.
 *
 *   interface/jpeg.cpp:244-354          JPEG::ReadInternal (the driver loop)
 *   codestream/decoder.cpp:77-108       SOI
 *   codestream/tables.cpp:1003-1418     Tables::ParseTablesIncremental (one marker segment per call)
 *   codestream/image.cpp:616-650, 480-600   frame header, "found a double frame header"
 *   marker/frame.cpp:111-208, 796-899, 1016-1123   SOF, StartParseScan / ScanForScanHeader, ParseTrailer
 *   marker/scan.cpp:163-315, 985-994    SOS, parser selection
 *   codestream/entropyparser.{hpp:147-160, cpp:117-201}   BeginReadMCU / ParseRestartMarker
 *   codestream/sequentialscan.cpp:112-141, 266-274, 381-428, 678-773
 *   codestream/refinementscan.cpp:225-233, 305-345, 584-700
 *   io/bitstream.{hpp:106-210, cpp:56-137}, io/bytestream.hpp:176-244, io/iostream.cpp:538-618
 *   coding/huffmandecoder.hpp:103-124, coding/huffmantemplate.cpp:802-905, marker/huffmantable.cpp:127-169
 *   marker/quantization.cpp:474-537, marker/restartintervalmarker.cpp:80-102, marker/adobemarker.cpp:96-117,
 *   marker/jfifmarker.cpp:103-129, marker/exifmarker.cpp:117-128, boxes/box.cpp:93-200
 * Errors leave through longjmp like the reference's JPG_THROW (std/setjmp based, tools/environment.hpp); the
 * reference's error code (JPGERR_*, interface/parameters.hpp:1156-1228) is kept in oj_info.ref_error.
 * ---------------------------------------------------------------------------------------- */
#define RS_INVALID_PARAMETER (-1024)
#define RS_UNEXPECTED_EOF (-1025)
#define RS_OVERFLOW_PARAMETER (-1028)
#define RS_OBJECT_DOESNT_EXIST (-1031)
#define RS_NOT_IMPLEMENTED (-1034)
#define RS_MALFORMED_STREAM (-1038)
#define RS_OUT_OF_MEMORY (-2048)

/* io/bytestream.hpp over a memory buffer.  Get() past the end returns EOF; PeekWord() needs two bytes and
 * leaves the position alone; LastUnDo() takes back the last byte unless the last Get() hit the end
 * (the refill then left an empty buffer: bytestream.hpp:229-236, iostream.cpp:132-197); SkipBytes() over the
 * end does not fail with the reference's seekable file hook (iostream.cpp:327-365 caches the seek), the
 * following reads return EOF. */
#define BS_EOF (-1L)
typedef struct {
  const uint8_t *d;
  size_t n, pos;
  int at_eof; /* the last Get() failed */
  int in_memory; /* the memory stream of a box (bs_skip) */
} oj_bs;

static void bs_open(oj_bs *s, const uint8_t *d, size_t n) { s->d = d; s->n = n; s->pos = 0; s->at_eof = 0; s->in_memory = 0; }
static long bs_get(oj_bs *s)
{
  if (s->pos >= s->n) { s->at_eof = 1; return BS_EOF; }
  s->at_eof = 0;
  return s->d[s->pos++];
}
static long bs_peekword(oj_bs *s)
{
  if (s->pos + 2 > s->n) return BS_EOF;
  return ((long)s->d[s->pos] << 8) | s->d[s->pos + 1];
}
static long bs_getword(oj_bs *s)
{
  long a = bs_get(s), b;
  if (a == BS_EOF) return BS_EOF;
  b = bs_get(s);
  if (b == BS_EOF) return BS_EOF;
  return (a << 8) | b;
}
static void bs_lastundo(oj_bs *s) { if (!s->at_eof && s->pos > 0) s->pos--; }
/* ByteStream::SkipBytes.  The file itself (IOStream over a seekable hook, io/iostream.cpp:327-365) caches a seek beyond its end
 * and the next read finds EOF; the memory stream of a box -- the residual codestream, a refinement scan -- throws
 * UNEXPECTED_EOF "unexpectedly hit the end of the stream while skipping bytes" (io/bytestream.cpp:231-276,
 * io/decoderstream.hpp:212-216): returns 1 for that. */
static int bs_skip(oj_bs *s, long n)
{
  if (n <= 0) return 0;
  if ((size_t)n > s->n - s->pos) { s->pos = s->n; return s->in_memory; }
  s->pos += (size_t)n;
  return 0;
}

/* One Huffman table as the DHT marker delivered it (coding/huffmantemplate.cpp:878-905: sixteen counts, then
 * as many values as they add up to -- up to 4080, nothing checks 256) and its decoder, built on first use. */
typedef struct {
  int defined;
  uint8_t counts[16];
  uint8_t values[16 * 255];
  int nvalues;
  int built;
  uint32_t first[17];  /* left-aligned 16-bit code of the first code word of length l */
  int32_t valptr[17];
} oj_huff;

/* One JPEG XT box, reassembled from its APP11 segments (boxes/box.cpp:88-200). */
typedef struct {
  uint32_t type;
  uint16_t en;
  uint8_t *data;
  size_t len, cap;
  uint64_t boxsize; /* payload bytes the box header announces (LBox - 8, XLBox - 16) */
  uint64_t parsed;  /* payload bytes its APP11 segments announced so far (m_uqParsedBytes: counted even where the file ends inside) */
  int complete;     /* all of them arrived and the content was parsed: only then the tables know the box (tables.cpp:1191-1283) */
} oj_box;

#define OJ_MAX_BOXES 64
enum { FT_BASELINE = 0, FT_SEQUENTIAL = 1, FT_PROGRESSIVE = 2, FT_RESIDUAL = 3 /* SOF 0xffb1: the residual scan type of part 8 (lossless / near-lossless coding) */,
       FT_RESIDUAL_PROGRESSIVE = 4 /* SOF 0xffb2: the same with spectral bands, successive approximation and EOB runs (`-rv`) */ };

typedef struct {
  jmp_buf jb;
  int err;          /* reference error code of the throw */
  int unsupported;  /* the throw is ours: a coding process outside the accelerated path */
  int warnings;
  const uint8_t *data;
  size_t len;
  oj_info *info;
  /* tables (codestream/tables.hpp) */
  int have_quant, have_huff;
  int late_quant_missing; /* nested: a component's quantiser table does not exist -- found out at the first request, see rs_scan */
  oj_huff huff[8]; /* 0..3 DC, 4..7 AC (marker/huffmantable.cpp:153) */
  uint32_t restart_interval;
  /* frame */
  int have_frame, frame_type;
  int need_dnl;
  int eoi_frame, eoi_image; /* the frame trailer / the image trailer stood at an EOI marker: only there does the reference turn to
                               the hidden refinement scans (marker/frame.cpp:1063-1070) and to the residual codestream
                               (codestream/image.cpp:1416-1431) -- a file whose codestream runs out without one has neither */
  int known_height; /* decode pass of a DNL frame: the height the header pass (which decodes the first scan to find it) came to */
  int known_bh[OJ_MAX_COMP]; /* ... and the block rows it made room for */
  int progressive; /* SOF2 */
  int hidden;      /* JPEG XT: bits of every coefficient that travel in hidden refinement scans (RSPC box) */
  oj_box *boxes;   /* where APP11 boxes are collected (OJ_MAX_BOXES entries); the caller's array or walk()'s own */
  int nboxes;
  int walk_all;    /* the caller wants the boxes: walk all scans even without planes */
  int xt_legacy;   /* the legacy codestream of a JPEG XT decode: the caller follows the merging specification itself */
  int residual_ok; /* the header of a RESI box's codestream is being read: SOF 0xffb1 is a frame type here */
  int in_memory;   /* the codestream lives in a box (the alpha channel's): a memory stream like a nested one's, an image of its own otherwise */
  int nested;      /* this is the residual codestream of a RESI box */
  int legacy_eoi_gone; /* nested: the residual codestream ran dry in front of a scan header and the search for one took the legacy
                        * stream's EOI (see rs_run) */
  int dri_seen, dri_extended; /* the first DRI marker stood in the header part: lengths 5 and 6 are legal from then on */
  int header_part;   /* the tables in front of the frame header of the file itself are being read: an LSE marker may stand there */
  int ls_trafo_seen; /* ... and one held a JPEG LS colour transformation: Tables::LTrafoTypeOf (codestream/tables.cpp:2024-2028)
                      * makes that the transformation of a three component frame without a merging specification --
                      * outside the accelerated subset (declined) */
  int32_t *const *planes; /* NULL: headers only */
} oj_parser;

static void rs_throw(oj_parser *ps, int code)
{
  ps->err = code;
  longjmp(ps->jb, 1);
}
static void rs_unsupported(oj_parser *ps)
{
  ps->unsupported = 1;
  rs_throw(ps, RS_NOT_IMPLEMENTED);
}
#define RS_WARN(ps) ((ps)->warnings++)

static int rd16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

/* HuffmanTemplate::BuildDecoder, coding/huffmantemplate.cpp:802-874.  The reference fills a two-level 8+8 bit
 * table; code words are handed out in increasing order without gaps, so "the code word whose range holds the next
 * sixteen bits" is the same function.  Throws where the reference throws. */
static void huff_build(oj_parser *ps, oj_huff *h)
{
  uint32_t code = 0;
  int i, j, k = 0;
  for (i = 0; i < 16; i++) {
    h->first[i + 1] = code;
    h->valptr[i + 1] = k;
    for (j = 0; j < h->counts[i]; j++) {
      const uint32_t last = code + (1u << (15 - i));
      if (last > 0x10000u) rs_throw(ps, RS_MALFORMED_STREAM); /* "entry depends on more bits than available" */
      if ((code >> (15 - i)) >= (1u << (i + 1)) - 1) RS_WARN(ps); /* all-1 code */
      code = last;
      k++;
    }
  }
  h->built = 1;
}

/* Annex K.3.3 tables: what HuffmanTable::DCTemplateOf / ACTemplateOf (marker/huffmantable.cpp:186-228) install when
 * a scan selects a table no DHT segment defined (as long as one DHT segment exists at all).  The reference picks its
 * defaults by frame type, precision and scan index (coding/huffmantemplate.cpp:140-790); for baseline and sequential
 * 8-bit frames they are the tables of the standard, regenerated here from the standard's description.  Other frame
 * types: not restated (reported as unsupported). */
static void huff_default(oj_parser *ps, oj_huff *h, int ac, int chroma)
{
  static const uint8_t dc_l_bits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
  static const uint8_t dc_c_bits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
  static const uint8_t ac_l_bits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
  static const uint8_t ac_c_bits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
  static const uint8_t ac_l_head[] = {0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61,
                                      0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52,
                                      0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82};
  static const uint8_t ac_c_head[] = {0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61,
                                      0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33,
                                      0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1};
  int i, n = 0;
  if (ps->frame_type == FT_PROGRESSIVE || ps->info->precision != 8) rs_unsupported(ps);
  memset(h, 0, sizeof(*h));
  if (!ac) {
    memcpy(h->counts, chroma ? dc_c_bits : dc_l_bits, 16);
    for (i = 0; i < 12; i++) h->values[n++] = (uint8_t)i;
  } else {
    /* K.5 / K.6: the code words of up to 15 bits in the order the standard lists them, then all the remaining
     * run/size pairs (sizes 1..10) in numerical order as 16-bit codes */
    const uint8_t *head = chroma ? ac_c_head : ac_l_head;
    const int nhead = chroma ? (int)sizeof(ac_c_head) : (int)sizeof(ac_l_head);
    uint8_t used[256];
    int r, s;
    memcpy(h->counts, chroma ? ac_c_bits : ac_l_bits, 16);
    memset(used, 0, sizeof(used));
    for (i = 0; i < nhead; i++) { h->values[n++] = head[i]; used[head[i]] = 1; }
    for (r = 0; r < 16; r++)
      for (s = 1; s <= 10; s++) {
        const int v = (r << 4) | s;
        if (!used[v]) h->values[n++] = (uint8_t)v;
      }
  }
  h->nvalues = n;
  h->defined = 1;
}

/* Quantization::ParseMarker, marker/quantization.cpp:474-537 */
static void rs_parse_dqt(oj_parser *ps, oj_bs *io)
{
  long len = bs_getword(io);
  if (len < 2) rs_throw(ps, RS_MALFORMED_STREAM);
  len -= 2;
  while (len > 2) {
    uint16_t deltas[64];
    long type = bs_get(io), target;
    int i;
    if (type == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
    target = type & 15;
    type >>= 4;
    if (type != 0 && type != 1) rs_throw(ps, RS_MALFORMED_STREAM);
    if (target > 3) rs_throw(ps, RS_MALFORMED_STREAM);
    len -= 1;
    if (type == 0) {
      if (len < 64) rs_throw(ps, RS_MALFORMED_STREAM);
      for (i = 0; i < 64; i++) {
        long v = bs_get(io);
        if (v == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
        deltas[g_scan_order[i]] = (uint16_t)v;
      }
      len -= 64;
    } else {
      if (len < 128) rs_throw(ps, RS_MALFORMED_STREAM);
      for (i = 0; i < 64; i++) {
        long v = bs_getword(io);
        if (v == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
        deltas[g_scan_order[i]] = (uint16_t)v;
      }
      len -= 128;
    }
    memcpy(ps->info->quant[target], deltas, sizeof(deltas));
    ps->info->quant_defined[target] = 1;
  }
  ps->have_quant = 1; /* m_pQuant exists as soon as a DQT marker was seen, tables.cpp:1007-1009 */
  if (len != 0) rs_throw(ps, RS_MALFORMED_STREAM);
}

/* HuffmanTable::ParseMarker, marker/huffmantable.cpp:127-169 + HuffmanTemplate::ParseMarker */
static void rs_parse_dht(oj_parser *ps, oj_bs *io)
{
  long len = bs_getword(io);
  if (len < 2) rs_throw(ps, RS_MALFORMED_STREAM);
  len -= 2;
  while (len > 0) {
    long t = bs_get(io);
    size_t p = io->pos, q;
    oj_huff *h;
    int i, total = 0;
    if (t == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
    len--;
    if ((t >> 4) > 1) rs_throw(ps, RS_MALFORMED_STREAM);
    if ((t & 15) > 3) rs_throw(ps, RS_MALFORMED_STREAM);
    h = &ps->huff[(t & 3) | ((t & 0xf0) >> 2)];
    memset(h, 0, sizeof(*h));
    for (i = 0; i < 16; i++) {
      long cnt = bs_get(io);
      if (cnt == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
      h->counts[i] = (uint8_t)cnt;
      total += (int)cnt;
    }
    for (i = 0; i < total; i++) {
      long v = bs_get(io);
      if (v == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
      h->values[i] = (uint8_t)v;
    }
    h->nvalues = total;
    h->defined = 1;
    q = io->pos - p;
    if (q > (size_t)len) rs_throw(ps, RS_MALFORMED_STREAM);
    len -= (long)q;
  }
  ps->have_huff = 1;
}

/* Box::ParseBoxMarker, boxes/box.cpp:93-200: the segment framing only (en, z, LBox, TBox [, XLBox], payload); what the
 * boxes mean is checked where they are used.  `length` is the marker length, CI already removed from the stream. */
/* The FORM of a merging specification, checked where its box completes: SuperBox::ParseBoxContent (boxes/superbox.cpp:93-206)
 * frames the sub-boxes, MergingSpecBox::CreateBox / AcknowledgeBox (boxes/mergingspecbox.cpp:108-256) allow one box of each
 * kind and one curve / matrix per index, each sub-box parses its own payload (outputconversionbox.cpp:93-127,
 * colortrafobox.cpp:62-79, nonlineartrafobox.cpp:62-80, dctbox.cpp:61-90, refinementspecbox.cpp:58-82,
 * parametrictonemappingbox.cpp:85-149, lineartransformationbox.cpp:62-99).  announced: the box length; have: the bytes that
 * arrived (fewer when the file ends inside the box: reads beyond them find EOF). */
#define SPEC_ID(a, b, c, d) (((uint32_t)(a) << 24) | ((uint32_t)(b) << 16) | ((uint32_t)(c) << 8) | (uint32_t)(d))
/* is_alpha: the box is the ALPHA merging specification (ASPC), where the compositing box AMUL belongs (boxes/alphabox.cpp:60-90) */
static void rs_check_merging_spec(oj_parser *ps, const uint8_t *d, size_t have, uint64_t announced, int is_alpha)
{
  static const uint32_t once[16] = {
    SPEC_ID('R', 'S', 'P', 'C'), SPEC_ID('O', 'C', 'O', 'N'), SPEC_ID('L', 'D', 'C', 'T'), SPEC_ID('R', 'D', 'C', 'T'),
    SPEC_ID('L', 'T', 'R', 'F'), SPEC_ID('C', 'T', 'R', 'F'), SPEC_ID('R', 'T', 'R', 'F'), SPEC_ID('D', 'T', 'R', 'F'),
    SPEC_ID('S', 'T', 'R', 'F'), SPEC_ID('L', 'P', 'T', 'S'), SPEC_ID('Q', 'P', 'T', 'S'), SPEC_ID('C', 'P', 'T', 'S'),
    SPEC_ID('R', 'P', 'T', 'S'), SPEC_ID('S', 'P', 'T', 'S'), SPEC_ID('P', 'P', 'T', 'S'), SPEC_ID('D', 'P', 'T', 'S')};
  int seen[16] = {0}, curve[16] = {0}, matrix[16] = {0}, amul = 0;
  uint64_t j = 0;
  while (j < announced) {
    const uint64_t left = announced - j;
    uint64_t xl, overhead = 8, len;
    uint32_t lbox, tbox;
    const uint8_t *pl;
    int k;
#define SPEC_BYTE(n) ((j + overhead + (uint64_t)(n)) < have ? (long)pl[n] : -1L) /* ByteStream::Get at the end: EOF */
    if (left < 8) rs_throw(ps, RS_MALFORMED_STREAM);
    if (j + 8 > have) rs_throw(ps, RS_UNEXPECTED_EOF);
    lbox = ((uint32_t)d[j] << 24) | ((uint32_t)d[j + 1] << 16) | ((uint32_t)d[j + 2] << 8) | d[j + 3];
    tbox = ((uint32_t)d[j + 4] << 24) | ((uint32_t)d[j + 5] << 16) | ((uint32_t)d[j + 6] << 8) | d[j + 7];
    xl = lbox;
    if (lbox == 1) {
      if (left < 16) rs_throw(ps, RS_MALFORMED_STREAM);
      if (j + 16 > have) rs_throw(ps, RS_UNEXPECTED_EOF);
      for (xl = 0, k = 0; k < 8; k++) xl = (xl << 8) | d[j + 8 + k];
      if (xl < 16) rs_throw(ps, RS_MALFORMED_STREAM);
      overhead = 16;
    } else if (lbox < 8) rs_throw(ps, RS_MALFORMED_STREAM); /* zero: "found a box size of zero within a superbox" */
    if (left < xl) rs_throw(ps, RS_MALFORMED_STREAM);
    len = xl - overhead;
    pl = d + j + overhead;
    for (k = 0; k < 16; k++)
      if (tbox == once[k]) { if (seen[k]) rs_throw(ps, RS_MALFORMED_STREAM); seen[k] = 1; } /* "found a double ... box" */
    if (tbox == SPEC_ID('R', 'S', 'P', 'C')) {
      const long v = (len == 1) ? SPEC_BYTE(0) : 0;
      if (len != 1 || (v >> 4) > 4 || (v & 15) > 4) rs_throw(ps, RS_MALFORMED_STREAM);
    } else if (tbox == SPEC_ID('O', 'C', 'O', 'N')) {
      long v;
      if (len != 3) rs_throw(ps, RS_MALFORMED_STREAM);
      v = SPEC_BYTE(0) & 0xff;
      if ((v >> 4) > 8) rs_throw(ps, RS_MALFORMED_STREAM); /* "bit depths cannot be larger than 16" */
      if (!(v & 1) && (SPEC_BYTE(1) != 0 || SPEC_BYTE(2) != 0)) rs_throw(ps, RS_MALFORMED_STREAM); /* "output conversion is disabled, but lookup information is not zero" */
    } else if (tbox == SPEC_ID('L', 'D', 'C', 'T') || tbox == SPEC_ID('R', 'D', 'C', 'T')) {
      long v;
      int t, ns;
      if (len != 1) rs_throw(ps, RS_MALFORMED_STREAM);
      v = SPEC_BYTE(0);
      t = (int)((v >> 4) & 0xff); ns = (int)(v & 15);
      if ((t != 0 && t != 2 && t != 3) || ns > 1 || (ns && t != 3)) rs_throw(ps, RS_MALFORMED_STREAM);
    } else if (tbox == SPEC_ID('L', 'T', 'R', 'F') || tbox == SPEC_ID('C', 'T', 'R', 'F') || tbox == SPEC_ID('R', 'T', 'R', 'F') ||
               tbox == SPEC_ID('D', 'T', 'R', 'F') || tbox == SPEC_ID('S', 'T', 'R', 'F')) {
      if (len != 1 || (SPEC_BYTE(0) & 15)) rs_throw(ps, RS_MALFORMED_STREAM); /* size; "the reserved field is not zero" */
    } else if (tbox == SPEC_ID('L', 'P', 'T', 'S') || tbox == SPEC_ID('Q', 'P', 'T', 'S') || tbox == SPEC_ID('C', 'P', 'T', 'S') ||
               tbox == SPEC_ID('R', 'P', 'T', 'S') || tbox == SPEC_ID('S', 'P', 'T', 'S') || tbox == SPEC_ID('P', 'P', 'T', 'S') ||
               tbox == SPEC_ID('D', 'P', 'T', 'S')) {
      if (len != 2) rs_throw(ps, RS_MALFORMED_STREAM);
    } else if (tbox == SPEC_ID('C', 'U', 'R', 'V')) {
      int m, e;
      if (len != 18) rs_throw(ps, RS_MALFORMED_STREAM);
      m = (int)(SPEC_BYTE(0) & 0xff); e = (int)(SPEC_BYTE(1) & 0xff);
      if ((m & 15) == 3 || (m & 15) > 8 || (e & 15) || (e >> 4) > 1) rs_throw(ps, RS_MALFORMED_STREAM);
      if (curve[m >> 4]) rs_throw(ps, RS_MALFORMED_STREAM); /* "found an double parametric curve box for the same index" */
      curve[m >> 4] = 1;
    } else if (tbox == SPEC_ID('M', 'T', 'R', 'X')) {
      long b;
      if (len != 19) rs_throw(ps, RS_MALFORMED_STREAM);
      b = SPEC_BYTE(0);
      if (b < 0 || (b >> 4) < 5 || (b & 15) != 13 || j + overhead + len > have) rs_throw(ps, RS_MALFORMED_STREAM);
      if (matrix[b >> 4]) rs_throw(ps, RS_MALFORMED_STREAM); /* "found an double linear transformation for the same index" */
      matrix[b >> 4] = 1;
    } else if (tbox == SPEC_ID('A', 'M', 'U', 'L')) {
      if (!is_alpha || amul) rs_throw(ps, RS_MALFORMED_STREAM); /* outside the alpha specification, or twice */
      amul = 1;
      if (len != 10 || (SPEC_BYTE(0) >> 4) > 3 || (SPEC_BYTE(0) & 15) || SPEC_BYTE(1) != 0) rs_throw(ps, RS_MALFORMED_STREAM);
    }
#undef SPEC_BYTE
    j += xl;
  }
}

static void rs_parse_box_marker(oj_parser *ps, oj_bs *io, long length)
{
  long overhead = 2 + 2 + 2 + 4 + 4 + 4, blen, dt;
  uint16_t en;
  uint64_t lbox;
  uint32_t tbox;
  int b;
  if (length <= overhead) rs_throw(ps, RS_MALFORMED_STREAM);
  en = (uint16_t)bs_getword(io);
  bs_getword(io); bs_getword(io); /* sequence number: segments arrive in order */
  lbox = (uint64_t)(uint32_t)(bs_getword(io) << 16);
  lbox |= (uint64_t)(bs_getword(io) & 0xffff);
  blen = length - overhead;
  if (lbox != 1 && lbox < 8) rs_throw(ps, RS_MALFORMED_STREAM);
  tbox = (uint32_t)(bs_getword(io) << 16);
  dt = bs_getword(io);
  if (dt == BS_EOF) rs_throw(ps, RS_UNEXPECTED_EOF);
  tbox |= (uint32_t)dt;
  if (lbox == 1) {
    overhead += 8;
    if (length <= overhead) rs_throw(ps, RS_MALFORMED_STREAM);
    lbox = (uint64_t)(bs_getword(io) & 0xffff) << 48;
    lbox |= (uint64_t)(bs_getword(io) & 0xffff) << 32;
    lbox |= (uint64_t)(bs_getword(io) & 0xffff) << 16;
    dt = bs_getword(io);
    if (dt == BS_EOF) rs_throw(ps, RS_UNEXPECTED_EOF);
    if (lbox < 8 + 8) rs_throw(ps, RS_MALFORMED_STREAM);
    lbox |= (uint64_t)dt;
    blen -= 8;
    lbox -= 8;
  }
  lbox -= 8; /* the box length and type are not payload */
  switch (tbox) { /* Box::CreateBox, boxes/box.cpp:391-428: boxes of unknown type are skipped, nothing about them is checked */
  case 0x52455349u: case 0x46494e45u: case 0x5246494eu: case 0x414c4641u: case 0x4146494eu: case 0x41524553u:
  case 0x41525246u:                                     /* RESI FINE RFIN ALFA AFIN ARES ARRF */
  case 0x53504543u: case 0x41535043u:                   /* SPEC ASPC */
  case 0x544f4e45u: case 0x46544f4eu: case 0x43555256u: /* TONE FTON CURV */
  case 0x4d545258u: case 0x4c43484bu: case 0x66747970u: /* MTRX LCHK ftyp */
    break;
  default:
    { if (bs_skip(io, blen)) rs_throw(ps, RS_UNEXPECTED_EOF); }
    return;
  }
  for (b = 0; b < ps->nboxes; b++)
    if (ps->boxes[b].type == tbox && ps->boxes[b].en == en) break;
  if (b < ps->nboxes) {
    if (ps->boxes[b].boxsize != lbox) rs_throw(ps, RS_MALFORMED_STREAM); /* "box size is not consistent across APP11 markers" */
    if (ps->boxes[b].complete) rs_throw(ps, RS_MALFORMED_STREAM);        /* "received box data beyond box length" */
  } else {
    if (ps->nboxes == OJ_MAX_BOXES) rs_unsupported(ps);
    memset(&ps->boxes[b], 0, sizeof(oj_box));
    ps->boxes[b].type = tbox; ps->boxes[b].en = en; ps->boxes[b].boxsize = lbox;
    ps->nboxes++;
  }
  ps->boxes[b].parsed += (uint64_t)blen; /* boxes/box.cpp:183-184: what the segment announces, not what the stream still had */
  {
    /* DecoderStream::Append (io/decoderstream.cpp:136-158): the buffer has the segment's announced size, and what the stream
     * no longer had -- the file ends inside the segment -- is FILLED WITH ZEROS (and warned about): the box holds every byte it
     * was promised.  (Rounds 4-5 let reads beyond the bytes that arrived find EOF: "found a box size of zero within a superbox",
     * -1038, is what the reference says of a merging specification cut short, tools/box_campaign.py r5.) */
    const size_t want = (size_t)blen;
    if ((size_t)blen > io->n - io->pos) { blen = (long)(io->n - io->pos); RS_WARN(ps); }
    if (ps->boxes[b].len + want > ps->boxes[b].cap) {
      size_t cap = (ps->boxes[b].len + want) * 2 + 64;
      uint8_t *nd = (uint8_t *)realloc(ps->boxes[b].data, cap);
      if (!nd) rs_throw(ps, RS_OUT_OF_MEMORY);
      ps->boxes[b].data = nd; ps->boxes[b].cap = cap;
    }
    memcpy(ps->boxes[b].data + ps->boxes[b].len, io->d + io->pos, (size_t)blen);
    memset(ps->boxes[b].data + ps->boxes[b].len + (size_t)blen, 0, want - (size_t)blen);
    ps->boxes[b].len += want;
    io->pos += (size_t)blen;
  }
  if (ps->boxes[b].parsed > ps->boxes[b].boxsize) rs_throw(ps, RS_MALFORMED_STREAM); /* "more data in the application marker than indicated" */
  if (ps->boxes[b].parsed == ps->boxes[b].boxsize) {
    const oj_box *bx = &ps->boxes[b];
    ps->boxes[b].complete = 1;
    if (tbox == 0x53504543u) rs_check_merging_spec(ps, bx->data, bx->len, bx->boxsize, 0);
    if (tbox == 0x41535043u) rs_check_merging_spec(ps, bx->data, bx->len, bx->boxsize, 1); /* ASPC */
    /* tables and matrices of the file's own list parse where they complete (inversetonemappingbox.cpp:72-118,
     * parametrictonemappingbox.cpp:85-149, lineartransformationbox.cpp:62-99); a table index / matrix id may be taken once
     * (codestream/tables.cpp:1247-1266; NameSpace::isUniqueNonlinearity counts TONE, FTON and CURV boxes) */
    if (tbox == 0x544f4e45u || tbox == 0x43555256u || tbox == 0x4d545258u) {
      const uint64_t n = bx->boxsize;
      if (tbox == 0x544f4e45u) {
        const uint64_t entries = (n - 1) >> 1;
        if (n > 65536u * 2 + 1 || !(n & 1) || n < 512 || (entries & (entries - 1))) rs_throw(ps, RS_MALFORMED_STREAM);
      } else if (tbox == 0x43555256u) {
        int ty, e;
        if (n != 18) rs_throw(ps, RS_MALFORMED_STREAM);
        ty = bx->len > 0 ? (bx->data[0] & 15) : 15; e = bx->len > 1 ? bx->data[1] : 0xff;
        if (ty == 3 || ty > 8 || (e & 15) || (e >> 4) > 1) rs_throw(ps, RS_MALFORMED_STREAM);
      } else {
        if (n != 19 || bx->len < n || (bx->data[0] >> 4) < 5 || (bx->data[0] & 15) != 13) rs_throw(ps, RS_MALFORMED_STREAM);
      }
      if (tbox != 0x43555256u && bx->len > 0) {
        int same = 0, o;
        for (o = 0; o < ps->nboxes; o++) {
          const oj_box *ob = &ps->boxes[o];
          if (!ob->complete || ob->len == 0 || (ob->data[0] >> 4) != (bx->data[0] >> 4)) continue;
          if (tbox == 0x4d545258u ? ob->type == 0x4d545258u : (ob->type == 0x544f4e45u || ob->type == 0x43555256u)) same++;
        }
        if (same > 1) rs_throw(ps, RS_MALFORMED_STREAM); /* "found a doubly used table destination ..." */
      }
    }
    if (tbox == 0x66747970u) { /* 'ftyp': FileTypeBox::ParseBoxContent, boxes/filetypebox.cpp:71-120 */
      if (bx->boxsize < 8) rs_throw(ps, RS_MALFORMED_STREAM);
      if (bx->len < 4 || memcmp(bx->data, "jpxt", 4) != 0) rs_throw(ps, RS_MALFORMED_STREAM); /* "file is not compatible to JPEG XT" */
      if ((bx->boxsize - 8) & 3) rs_throw(ps, RS_MALFORMED_STREAM);
    }
  }
}

/* Tables::ParseTablesIncremental, codestream/tables.cpp:1003-1418: one marker segment of the tables/misc section.
 * Returns 0 when the next marker does not belong to it (SOFn, SOS, EOI, DHP) or the stream ended. */
static int rs_tables_incremental(oj_parser *ps, oj_bs *io)
{
  long marker = bs_peekword(io);
  switch (marker) {
  case 0xffdb: bs_getword(io); rs_parse_dqt(ps, io); break;
  case 0xffc4: bs_getword(io); rs_parse_dht(ps, io); break;
  case 0xffcc: { /* DAC, ACTable::ParseMarker (marker/actable.cpp:119-149) with ACTemplate::ParseDCMarker / ParseACMarker
                  * (coding/actemplate.cpp:71-107): parsed and checked; the conditioning never matters to Huffman scans */
    long len;
    bs_getword(io);
    len = bs_getword(io);
    if (len < 2) rs_throw(ps, RS_MALFORMED_STREAM);
    len -= 2;
    while (len > 0) {
      long t = bs_get(io), v;
      if (t == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
      len--;
      if ((t >> 4) > 1) rs_throw(ps, RS_MALFORMED_STREAM); /* "undefined conditioning table type" */
      v = bs_get(io);
      if (v == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
      if ((t >> 4) == 1) {
        if (v < 1 || v > 63) rs_throw(ps, RS_MALFORMED_STREAM);
      } else if ((v >> 4) < (v & 0x0f)) rs_throw(ps, RS_MALFORMED_STREAM);
      len--;
    }
    break;
  }
  case 0xffdd: { /* RestartIntervalMarker::ParseMarker, marker/restartintervalmarker.cpp:80-102.  The marker object is made where the
                  * first DRI stands (codestream/tables.cpp:1045-1047), with the JPEG LS lengths (5 and 6 bytes: 24 and 32 bit
                  * intervals) when that is the header part of the file: see the LSE marker below */
    long len, upper = 0;
    bs_getword(io);
    if (!ps->dri_seen) { ps->dri_seen = 1; ps->dri_extended = ps->header_part; }
    len = bs_getword(io);
    if (len < 4 || len > (ps->dri_extended ? 6 : 4)) rs_throw(ps, RS_MALFORMED_STREAM);
    if (len == 6) upper = bs_getword(io);
    else if (len == 5) upper = bs_get(io);
    len = bs_getword(io);
    if (len == BS_EOF) rs_throw(ps, RS_UNEXPECTED_EOF);
    if (upper != 0) rs_unsupported(ps); /* intervals beyond 16 bits: outside the accelerated subset (declined) */
    ps->restart_interval = (uint32_t)(len & 0xffff);
    break;
  }
  case 0xfffe: { /* COM */
    long size;
    bs_getword(io);
    size = bs_getword(io);
    if (size == BS_EOF) rs_throw(ps, RS_UNEXPECTED_EOF);
    if (size <= 2) rs_throw(ps, RS_MALFORMED_STREAM);
    { if (bs_skip(io, size - 2)) rs_throw(ps, RS_UNEXPECTED_EOF); }
    break;
  }
  case 0xfff8: { /* LSE, codestream/tables.cpp:1073-1110: read in the header part of the file itself -- Decoder::ParseHeaderIncremental,
                  * codestream/decoder.cpp:85, passes isls = true as the frame type is not known yet --, "outside of a JPEG LS
                  * stream" behind the frame header (marker/frame.cpp:812-822) and in the codestreams of boxes (image.cpp:1285, 1362) */
    long len;
    if (!ps->header_part) rs_throw(ps, RS_MALFORMED_STREAM);
    bs_getword(io);
    len = bs_getword(io);
    if (len > 3) {
      const int id = (int)(bs_get(io) & 0xff);
      if (id == 1) { /* Thresholds::ParseMarker, marker/thresholds.cpp:84-94 */
        int k;
        if (len != 13) rs_throw(ps, RS_MALFORMED_STREAM);
        for (k = 0; k < 5; k++) bs_getword(io);
        break;
      } else if (id == 2 || id == 3 || id == 4) {
        rs_throw(ps, RS_NOT_IMPLEMENTED); /* the reference's own refusal: mapping tables, size extensions */
      } else if (id == 0x0d) { /* LSColorTrafo::ParseMarker, marker/lscolortrafo.cpp:120-169 */
        int depth, k, j;
        if (ps->ls_trafo_seen) rs_throw(ps, RS_MALFORMED_STREAM);
        ps->ls_trafo_seen = 1;
        if (len < 6) rs_throw(ps, RS_MALFORMED_STREAM);
        bs_getword(io);
        depth = (int)(bs_get(io) & 0xff);
        len -= 6;
        if (len != 2 * depth * depth) rs_throw(ps, RS_MALFORMED_STREAM);
        if (depth == 0) rs_throw(ps, RS_MALFORMED_STREAM);
        for (k = 0; k < depth; k++) bs_get(io);
        for (k = 0; k < depth; k++) {
          const int v = (int)(bs_get(io) & 0xff);
          if ((v & 0x7f) > 32) rs_throw(ps, RS_OVERFLOW_PARAMETER);
          for (j = 0; j + 1 < depth; j++) bs_getword(io);
        }
        break;
      } else {
        RS_WARN(ps);
        len--;
      }
    }
    if (len <= 2) rs_throw(ps, RS_MALFORMED_STREAM);
    bs_skip(io, len - 2);
    break;
  }
  case 0xffe0: { /* APP0: JFIF is parsed (marker/jfifmarker.cpp:103-129), anything else skipped */
    long len;
    bs_getword(io);
    len = bs_getword(io);
    if (len >= 2 + 5 + 2 + 1 + 2 + 2 + 1 + 1) {
      const char *id = "JFIF";
      while (*id) { len--; if (bs_get(io) != *id) break; id++; }
      if (*id == 0) {
        len--;
        if (bs_get(io) == 0) {
          long unit, l = (len + 5) & 0xffff;
          if (l < 2 + 5 + 2 + 1 + 2 + 2 + 1 + 1) rs_throw(ps, RS_MALFORMED_STREAM);
          bs_get(io); bs_get(io);
          unit = bs_get(io);
          if ((unit & 0xff) > 2) rs_throw(ps, RS_MALFORMED_STREAM); /* UBYTE unit > Centimeter; EOF reads as 0xff */
          bs_getword(io); bs_getword(io);
          l -= 2 + 5 + 2 + 1 + 2 + 2;
          if (l > 0) { if (bs_skip(io, l)) rs_throw(ps, RS_UNEXPECTED_EOF); }
          break;
        }
      }
    }
    if (len <= 2) rs_throw(ps, RS_MALFORMED_STREAM);
    { if (bs_skip(io, len - 2)) rs_throw(ps, RS_UNEXPECTED_EOF); }
    break;
  }
  case 0xffe1: { /* APP1: Exif header checked (marker/exifmarker.cpp:117-128) */
    long len;
    bs_getword(io);
    len = bs_getword(io);
    if (len >= 2 + 4 + 2 + 2 + 2 + 4 + 2) {
      const char *id = "Exif";
      while (*id) { len--; if (bs_get(io) != *id) break; id++; }
      if (*id == 0) {
        len -= 2;
        if (bs_getword(io) == 0) {
          long l = (len + 4 + 2) & 0xffff;
          if (l < 2 + 4 + 2 + 2 + 2 + 4 + 2 + 4) rs_throw(ps, RS_MALFORMED_STREAM);
          l -= 2 + 4 + 2;
          if (l > 0) { if (bs_skip(io, l)) rs_throw(ps, RS_UNEXPECTED_EOF); }
          break;
        }
      }
    }
    if (len < 2) rs_throw(ps, RS_MALFORMED_STREAM);
    { if (bs_skip(io, len - 2)) rs_throw(ps, RS_UNEXPECTED_EOF); }
    break;
  }
  case 0xffeb: { /* APP11: JPEG XT boxes */
    long len;
    bs_getword(io);
    len = bs_getword(io);
    if (len >= 2 + 2 + 2 + 4 + 4 + 4) {
      if (bs_peekword(io) == 0x4a50) {
        bs_getword(io);
        if (ps->nested) rs_throw(ps, RS_MALFORMED_STREAM); /* "Found a box in the residual codestream." */
        rs_parse_box_marker(ps, io, len & 0xffff);
        break;
      }
    }
    if (len < 2) rs_throw(ps, RS_MALFORMED_STREAM);
    { if (bs_skip(io, len - 2)) rs_throw(ps, RS_UNEXPECTED_EOF); }
    break;
  }
  case 0xffee: { /* APP14: Adobe (marker/adobemarker.cpp:96-117), only at its exact size */
    long len;
    bs_getword(io);
    len = bs_getword(io);
    if (len == 2 + 5 + 2 + 2 + 2 + 1) {
      const char *id = "Adobe";
      while (*id) { len--; if (bs_get(io) != *id) break; id++; }
      if (*id == 0) {
        long version, color;
        if (((len + 5) & 0xffff) != 2 + 5 + 2 + 2 + 2 + 1) rs_throw(ps, RS_MALFORMED_STREAM);
        version = bs_getword(io) & 0xffff;
        if (version != 100 && version != 101) rs_throw(ps, RS_MALFORMED_STREAM);
        bs_getword(io); bs_getword(io);
        color = bs_get(io);
        if (color < 0 || color > 2) rs_throw(ps, RS_MALFORMED_STREAM);
        ps->info->adobe_transform = (int)color;
        break;
      }
    }
    if (len < 2) rs_throw(ps, RS_MALFORMED_STREAM);
    { if (bs_skip(io, len - 2)) rs_throw(ps, RS_UNEXPECTED_EOF); }
    break;
  }
  case 0xffdf: rs_throw(ps, RS_MALFORMED_STREAM); break; /* EXP outside a hierarchical process (size / content errors are MALFORMED as well) */
  case 0xffc8: { /* JPG extensions */
    long len;
    bs_getword(io);
    len = bs_getword(io);
    if (len < 2) rs_throw(ps, RS_MALFORMED_STREAM);
    { if (bs_skip(io, len - 2)) rs_throw(ps, RS_UNEXPECTED_EOF); }
    break;
  }
  case 0xffc0: case 0xffc1: case 0xffc2: case 0xffc3: case 0xffc5: case 0xffc6: case 0xffc7: case 0xffc9:
  case 0xffca: case 0xffcb: case 0xffcd: case 0xffce: case 0xffcf: case 0xffb1: case 0xffb2: case 0xffb3:
  case 0xffb9: case 0xffba: case 0xffbb: case 0xffd9: case 0xffda: case 0xffde: case 0xfff7:
    return 0;
  case 0xffff: bs_get(io); break; /* a filler byte */
  case 0xffd0: case 0xffd1: case 0xffd2: case 0xffd3: case 0xffd4: case 0xffd5: case 0xffd6: case 0xffd7:
    bs_getword(io);
    RS_WARN(ps); /* stray restart marker */
    break;
  default:
    if (marker >= 0xffc0 && (marker < 0xffd0 || marker >= 0xffd8) && marker < 0xfff0) {
      long size;
      bs_getword(io);
      size = bs_getword(io);
      if (size == BS_EOF) rs_throw(ps, RS_UNEXPECTED_EOF);
      if (size <= 2) rs_throw(ps, RS_MALFORMED_STREAM);
      { if (bs_skip(io, size - 2)) rs_throw(ps, RS_UNEXPECTED_EOF); }
    } else {
      long dt;
      RS_WARN(ps); /* "found invalid marker, probably a marker size is out of range" (covers EOF = -1 as well) */
      bs_get(io);
      do { dt = bs_get(io); } while (dt != 0xff && dt != BS_EOF);
      if (dt == 0xff) bs_lastundo(io);
      else return 0;
    }
  }
  return 1;
}

static void frame_geometry(oj_info *f)
{
  int c;
  f->mcus_x = (f->width + 8 * f->hmax - 1) / (8 * f->hmax);
  f->mcus_y = (f->height + 8 * f->vmax - 1) / (8 * f->vmax);
  for (c = 0; c < f->ncomp; c++) {
    f->subx[c] = f->hmax / f->hs[c];
    f->suby[c] = f->vmax / f->vs[c];
    f->bw[c] = f->mcus_x * f->hs[c];
    f->bh[c] = f->mcus_y * f->vs[c];
    f->cw[c] = (f->width + f->subx[c] - 1) / f->subx[c];
    f->ch[c] = (f->height + f->suby[c] - 1) / f->suby[c];
    f->rows[c] = (f->ch[c] + 7) >> 3; /* rows a scan creates when the height is known (control/blockbuffer.cpp:212-265) */
    if (f->dnl) f->bh[c] += f->vs[c]; /* one MCU row more: what the first scan may create below the picture, see jpeg_oracle.h */
  }
}

/* Image::ParseFrameHeader + Frame::ParseMarker + Component::ParseMarker: codestream/image.cpp:616-650,
 * marker/frame.cpp:111-208, marker/component.cpp:86-111, marker/component.hpp:99-107 */
static void rs_parse_frame_header(oj_parser *ps, oj_bs *io)
{
  oj_info *f = ps->info;
  long marker = bs_peekword(io), len, data;
  int c, type = -1, other_process, lossless_kind, residual_kind;
  if (marker == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
  if (marker == 0xffd9) rs_throw(ps, RS_MALFORMED_STREAM);
  marker = bs_getword(io);
  switch (marker) {
  case 0xffc0: type = FT_BASELINE; break;
  case 0xffc1: type = FT_SEQUENTIAL; break;
  case 0xffc2: type = FT_PROGRESSIVE; break;
  case 0xffb1: if (ps->nested || ps->residual_ok) type = FT_RESIDUAL; break; /* residual sequential: what `-ro` / `-Q 100` put into the RESI box */
  case 0xffb2: if (ps->nested || ps->residual_ok) type = FT_RESIDUAL_PROGRESSIVE; break; /* ... with `-rv` */
  case 0xffc5: case 0xffc6: case 0xffc7: case 0xffcd: case 0xffce: case 0xffcf:
    /* Image::CreateFrameBuffer, codestream/image.cpp:487-500: "found a differential frame outside a hierarchical image process"
     * -- in front of everything else, the header is not even read (a DHP marker, the only way into such a process, is declined) */
    rs_throw(ps, RS_MALFORMED_STREAM);
    break;
  case 0xffc3: case 0xffc9: case 0xffca: case 0xffcb:
  case 0xffb3: case 0xffb9: case 0xffba: case 0xffbb: case 0xfff7: case 0xffde:
    break; /* lossless, arithmetic, hierarchical, the other residual types, JPEG LS: other coding processes */
  default: rs_throw(ps, RS_MALFORMED_STREAM); /* "unexpected marker while parsing the image, decoder out of sync" */
  }
  if (ps->have_frame) rs_throw(ps, RS_MALFORMED_STREAM); /* "found a double frame header" */
  /* A coding process outside this restatement is "unsupported" -- behind the checks Frame::ParseMarker makes on every header
   * (marker/frame.cpp:111-208): a damaged byte that spells such a marker in front of garbage is MALFORMED_STREAM there. */
  other_process = type < 0;
  lossless_kind = marker == 0xffc3 || marker == 0xffc7 || marker == 0xffcb || marker == 0xffcf || marker == 0xfff7;
  residual_kind = marker == 0xffb1 || marker == 0xffb2 || marker == 0xffb3 || marker == 0xffb9 || marker == 0xffba || marker == 0xffbb;
  if (other_process) type = (marker == 0xffca || marker == 0xffce) ? FT_PROGRESSIVE : FT_SEQUENTIAL; /* (:163-170: four components at most for these two) */
  ps->frame_type = type;
  ps->progressive = ps->frame_type == FT_PROGRESSIVE;
  f->residual_type = type == FT_RESIDUAL || type == FT_RESIDUAL_PROGRESSIVE;
  len = bs_getword(io);
  if (len < 8) rs_throw(ps, RS_MALFORMED_STREAM);
  f->precision = (int)(bs_get(io) & 0xff);
  /* marker/frame.cpp:121-149: lossless types 2..16 bits, residual types 2..17, baseline 8, the rest 8 or 12 */
  if (other_process && lossless_kind ? (f->precision < 2 || f->precision > 16)
      : (f->residual_type || (other_process && residual_kind)) ? (f->precision < 2 || f->precision > 17)
      : ps->frame_type == FT_BASELINE ? f->precision != 8 : (f->precision != 8 && f->precision != 12)) rs_throw(ps, RS_MALFORMED_STREAM);
  data = bs_getword(io);
  if (data == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
  f->height = (int)data;
  data = bs_getword(io);
  if (data == BS_EOF || data == 0) rs_throw(ps, RS_MALFORMED_STREAM);
  f->width = (int)data;
  data = bs_get(io);
  if (data == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
  if (data <= 0 || data > (ps->frame_type == FT_PROGRESSIVE ? 4 : 255)) rs_throw(ps, RS_MALFORMED_STREAM);
  len -= 8;
  if (len != 3 * data) rs_throw(ps, RS_MALFORMED_STREAM);
  if (other_process) rs_unsupported(ps);
  if (ps->ls_trafo_seen && data == 3) rs_unsupported(ps); /* the JPEG LS colour transformation on a DCT frame: declined */
  if (data > OJ_MAX_COMP) rs_unsupported(ps); /* more than four components: not on the accelerated path */
  f->ncomp = (int)data;
  f->hmax = f->vmax = 0;
  for (c = 0; c < f->ncomp; c++) {
    data = bs_get(io);
    if (data == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
    f->comp_id[c] = (int)data;
    data = bs_get(io);
    if (data == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
    f->hs[c] = (int)(data >> 4);
    f->vs[c] = (int)(data & 15);
    if (f->hs[c] == 0 || f->vs[c] == 0) rs_throw(ps, RS_MALFORMED_STREAM);
    data = bs_get(io);
    if (data < 0 || data > 3) rs_throw(ps, RS_MALFORMED_STREAM);
    f->tq[c] = (int)data;
    if (f->hs[c] > f->hmax) f->hmax = f->hs[c];
    if (f->vs[c] > f->vmax) f->vmax = f->vs[c];
  }
  for (c = 0; c < f->ncomp; c++)
    if (f->hmax % f->hs[c] || f->vmax % f->vs[c]) rs_throw(ps, RS_NOT_IMPLEMENTED); /* non-integer subsampling factors */
  ps->need_dnl = f->height == 0;
  frame_geometry(f);
  /* Image::CreateFrameBuffer ends with m_pImageBuffer->PrepareForDecoding() (codestream/image.cpp:604-607):
   * upsamplers exist for factors up to 4 (upsampling/upsamplerbase.cpp:330-476) */
  for (c = 0; c < f->ncomp; c++)
    if (f->subx[c] > 4 || f->suby[c] > 4) rs_throw(ps, RS_NOT_IMPLEMENTED);
  ps->have_frame = 1;
  f->scan_state_valid = ps->planes != NULL || ps->walk_all; /* a header-only walk does not see all scans */
}

/* Frame::PostImageHeight (marker/frame.cpp:1241-1258) for a frame whose header carried zero lines: the DNL marker the
 * first scan ran into delivered them (EntropyParser::ParseDNLMarker, codestream/entropyparser.cpp:204-249).  The reference
 * discovers the marker WHILE it decodes -- no look-ahead can tell what it makes of a damaged segment -- so the header pass
 * (oj_read_info) decodes the first scan of such a frame as well, without keeping the coefficients, and the decode pass is
 * told the height it came to. */
static void post_image_height(oj_parser *ps, int h)
{
  ps->info->height = h;
  ps->info->dnl = 1;
  frame_geometry(ps->info);
  ps->need_dnl = 0;
}

/* ------------------------------------------------------------------------------------------
 * Bit reader: io/bitstream.hpp:106-210 + io/bitstream.cpp:56-137, byte-stuffing flavour, stated literally.
 * FF 00 -> FF.  In front of any other FF xx (a marker) Fill() stops and adds EIGHT zero bits per call; at the end of
 * the data it adds zero bytes until the window is full.  A read that still finds too few bits throws -- which is how
 * an unassigned Huffman code (length 0xff in the reference's table) and a truncated restart interval fail.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  oj_parser *ps;
  oj_bs *io;
  uint32_t b;
  int bits;
  int marker, eof;
} oj_bits;

static void bits_open(oj_bits *s, oj_parser *ps, oj_bs *io) { s->ps = ps; s->io = io; s->b = 0; s->bits = 0; s->marker = 0; s->eof = 0; }

static void bits_fill(oj_bits *s)
{
  do {
    long dt = bs_get(s->io);
    if (dt == 0xff) {
      bs_lastundo(s->io);
      if (bs_peekword(s->io) == 0xff00) {
        bs_getword(s->io);
        s->b |= (uint32_t)0xff << (24 - s->bits);
        s->bits += 8;
      } else {
        s->marker = 1;
        s->bits += 8;
        break;
      }
    } else if (dt == BS_EOF) {
      s->eof = 1;
      s->bits += 8;
    } else {
      s->b |= (uint32_t)dt << (24 - s->bits);
      s->bits += 8;
    }
  } while (s->bits <= 24);
}

static void bits_report_error(oj_bits *s)
{
  if (s->eof) rs_throw(s->ps, RS_UNEXPECTED_EOF);
  if (s->marker) rs_throw(s->ps, RS_UNEXPECTED_EOF);
  rs_throw(s->ps, RS_MALFORMED_STREAM);
}

static uint32_t bits_get(oj_bits *s, int n)
{
  uint32_t v;
  if (n > s->bits) {
    bits_fill(s);
    if (n > s->bits) bits_report_error(s);
  }
  v = s->b >> (32 - n);
  s->b <<= n;
  s->bits -= n;
  return v;
}

static uint32_t bits_peekword(oj_bits *s)
{
  if (s->bits < 16) bits_fill(s);
  return s->b >> 16;
}

static void bits_skip(oj_bits *s, int size)
{
  if (size > s->bits) bits_report_error(s);
  s->b <<= size;
  s->bits -= size;
}

/* HuffmanDecoder::Get, coding/huffmandecoder.hpp:103-124 */
static int huff_get(oj_bits *s, const oj_huff *h)
{
  const uint32_t data = bits_peekword(s);
  int l;
  for (l = 1; l <= 16; l++) {
    if (h->counts[l - 1]) {
      const uint32_t lo = h->first[l], hi = lo + ((uint32_t)h->counts[l - 1] << (16 - l));
      if (data >= lo && data < hi) {
        bits_skip(s, l);
        return h->values[h->valptr[l] + (int)((data - lo) >> (16 - l))];
      }
    }
  }
  bits_skip(s, 0xff); /* unassigned code: the table holds length 0xff there -> ReportError */
  return 0;
}

/* One entropy coded scan and its parser state (EntropyParser + SequentialScan / RefinementScan members) */
typedef struct {
  oj_parser *ps;
  oj_bs *io;
  oj_bits bits;
  int ns, ci[OJ_MAX_COMP];
  const oj_huff *dc[OJ_MAX_COMP], *ac[OJ_MAX_COMP];
  int ss, se, lowbit;
  int refinement;      /* RefinementScan instead of SequentialScan */
  int progressive_run; /* m_bProgressive: EOB runs are legal (sequentialscan.cpp:84-87) */
  int residual;        /* m_bResidual: no DC coding, the band starts at position 0, symbol 0x10 = -0x8000 (sequentialscan.cpp:682, 709, 727-736) */
  int32_t pred[OJ_MAX_COMP];
  int skip[OJ_MAX_COMP];
  uint32_t ri, togo;
  long next_rst;
  int valid;
  int scan_for_dnl, dnl_found; /* m_bScanForDNL, m_bDNLFound (codestream/entropyparser.cpp:79-80) */
} oj_scan;

/* SequentialScan::DecodeBlock, codestream/sequentialscan.cpp:678-773 (not large range, not differential; the residual flavour
 * of `SequentialScan(.., true, true)`, marker/scan.cpp:483-489: no DC part, the AC part from position 0 on, symbol 0x10) */
static void decode_block(oj_scan *sc, int32_t *block, const oj_huff *dc, const oj_huff *ac, int32_t *prevdc, int *skip)
{
  oj_bits *b = &sc->bits;
  if (sc->ss == 0 && !sc->residual) {
    int32_t diff = 0;
    const int value = huff_get(b, dc);
    if (value > 0) {
      const int32_t v = 1 << ((value - 1) & 31);
      if (value > 15) rs_throw(sc->ps, RS_MALFORMED_STREAM);
      diff = (int32_t)bits_get(b, value);
      if (diff < v) diff += (int32_t)((-1L) * (1L << value)) + 1;
    }
    *prevdc += diff;
    block[0] = (int32_t)((uint32_t)*prevdc << sc->lowbit);
  }
  if (sc->se) {
    if (*skip > 0) {
      (*skip)--;
    } else {
      int k = sc->ss ? sc->ss : (sc->residual ? 0 : 1);
      do {
        const int rs = huff_get(b, ac);
        int r = rs >> 4;
        const int s = rs & 15;
        int32_t diff;
        if (s == 0) {
          if (r == 15) { k += 16; continue; }
          if (r == 0 || sc->progressive_run) {
            *skip = 1 << r;
            if (r) *skip |= (int)bits_get(b, r);
            *skip = (*skip - 1) & 0xffff; /* UWORD */
            break;
          }
          if (sc->residual && rs == 0x10) { /* the value -0x8000, which has no magnitude category: four bits of run follow */
            r = (int)bits_get(b, 4);
            k += r;
            if (k >= 64) rs_throw(sc->ps, RS_MALFORMED_STREAM);
            block[g_scan_order[k]] = (int32_t)((uint32_t)(-0x8000) << sc->lowbit);
            k++;
            continue;
          }
          rs_throw(sc->ps, RS_MALFORMED_STREAM);
        }
        k += r;
        diff = (int32_t)bits_get(b, s);
        if (diff < (1 << (s - 1))) diff += (int32_t)((-1L) * (1L << s)) + 1;
        if (k >= 64) rs_throw(sc->ps, RS_MALFORMED_STREAM);
        block[g_scan_order[k]] = (int32_t)((uint32_t)diff << sc->lowbit);
        k++;
      } while (k <= sc->se);
    }
  }
}

/* RefinementScan::DecodeBlock, codestream/refinementscan.cpp:584-700 */
static void decode_block_refine(oj_scan *sc, int32_t *block, const oj_huff *ac, int *skip)
{
  oj_bits *b = &sc->bits;
  const int al = sc->lowbit;
  if (sc->ss == 0 && !sc->residual) block[0] |= (int32_t)(bits_get(b, 1) << al);
  if (sc->se || sc->residual) { /* :594 */
    int k = sc->ss, run = 0;
    int32_t s = 0;
    int enter_at_start = 0;
    if (*skip > 0) { run = (sc->se - sc->ss + 1) & 0xff; (*skip)--; }
    else { k--; enter_at_start = 1; }
    do {
      int32_t data;
      if (!enter_at_start) {
        data = block[g_scan_order[k]];
        if (data) {
          if (bits_get(b, 1)) block[g_scan_order[k]] += data > 0 ? (1 << al) : -(1 << al);
          continue;
        } else if (run) {
          run--;
          continue;
        }
        block[g_scan_order[k]] = (int32_t)((uint32_t)s << al);
        if (k == sc->se) break;
      }
      enter_at_start = 0;
      {
        const int rs = huff_get(b, ac), r = rs >> 4;
        s = rs & 15;
        if (s == 0) {
          if (r == 15) run = r;
          else {
            *skip = 1 << r;
            if (r) *skip |= (int)bits_get(b, r);
            *skip = (*skip - 1) & 0xffff;
            run = (sc->se - k + 1) & 0xff;
          }
        } else if (s != 1) {
          RS_WARN(sc->ps); /* :667 "unexpected Huffman symbol in refinement coding": leave the block unrefined, go on */
          run = 0;
          s = 0;
        } else {
          if (bits_get(b, 1) == 0) s = -s;
          run = r;
        }
      }
    } while (++k <= sc->se);
  }
}

/* SequentialScan::Restart / RefinementScan::Restart, sequentialscan.cpp:266-274, refinementscan.cpp:225-233 */
static void scan_restart(oj_scan *sc)
{
  int i;
  for (i = 0; i < sc->ns; i++) { sc->pred[i] = 0; sc->skip[i] = 0; }
  bits_open(&sc->bits, sc->ps, sc->io);
}

/* EntropyParser::ParseDNLMarker, codestream/entropyparser.cpp:204-249.  It looks at the BYTE stream, whose position is
 * where the bit reader's prefetch stopped (up to four bytes ahead of the bits in use, io/bitstream.cpp:56-118): the
 * marker is "found" -- and every later MCU of the scan skipped -- while the last MCUs' bits still wait in the window. */
static int parse_dnl_marker(oj_scan *sc)
{
  oj_bs *io = sc->io;
  long dt;
  if (sc->dnl_found) return 1;
  dt = bs_peekword(io);
  while (dt == 0xffff) { bs_get(io); dt = bs_peekword(io); }
  if (dt != 0xffdc) return 0;
  bs_getword(io);
  dt = bs_getword(io);
  if (dt != 4) rs_throw(sc->ps, RS_MALFORMED_STREAM);
  dt = bs_getword(io);
  if (dt == BS_EOF) rs_throw(sc->ps, RS_UNEXPECTED_EOF);
  if (dt == 0) rs_throw(sc->ps, RS_MALFORMED_STREAM);
  if (sc->ps->need_dnl) post_image_height(sc->ps, (int)dt);
  else if (dt != sc->ps->info->height) rs_unsupported(sc->ps); /* (the decode pass reads the marker the header pass read) */
  sc->dnl_found = 1;
  return 1;
}

/* EntropyParser::ParseRestartMarker, codestream/entropyparser.cpp:117-201 */
static void parse_restart_marker(oj_scan *sc)
{
  oj_bs *io = sc->io;
  long dt = bs_peekword(io);
  while (dt == 0xffff) { bs_get(io); dt = bs_peekword(io); }
  if (dt == 0xffdc && sc->scan_for_dnl) { parse_dnl_marker(sc); return; } /* :127-128: no restart, the counter stays at 0 */
  if (dt == sc->next_rst) {
    bs_getword(io);
    scan_restart(sc);
    sc->next_rst = (sc->next_rst + 1) & 0xfff7;
    sc->togo = sc->ri;
    sc->valid = 1;
    return;
  }
  RS_WARN(sc->ps); /* "entropy coder is out of sync, trying to advance to the next marker" */
  for (;;) {
    dt = bs_get(io);
    if (dt == BS_EOF) rs_throw(sc->ps, RS_UNEXPECTED_EOF);
    if (dt != 0xff) continue;
    bs_lastundo(io);
    dt = bs_peekword(io);
    if (dt >= 0xffd0 && dt < 0xffd8) {
      if (dt == sc->next_rst) { /* the decoder was behind and is back in step */
        bs_getword(io);
        scan_restart(sc);
        sc->next_rst = (sc->next_rst + 1) & 0xfff7;
        sc->togo = sc->ri;
        sc->valid = 1;
        return;
      } else if (((dt - sc->next_rst) & 7) >= 4) {
        bs_getword(io); /* likely a marker the decoder already passed: drop it, keep looking */
      } else {
        /* the marker is ahead: this segment is lost, leave the marker where it is and look at it again
         * when the next interval starts */
        sc->valid = 0;
        sc->next_rst = (sc->next_rst + 1) & 0xfff7;
        sc->togo = sc->ri;
        return;
      }
    } else if (dt >= 0xffc0 && dt < 0xfff0) { /* some other marker: the scan is over, the rest is lost */
      sc->valid = 0;
      sc->next_rst = (sc->next_rst + 1) & 0xfff7;
      sc->togo = sc->ri;
      return;
    } else {
      bs_get(io); /* garbage, FF00, or a lone FF in front of the end: eat one byte */
    }
  }
}

/* Scan::ParseMarker (marker/scan.cpp:163-315) for `type`, then Scan::CreateParser / StartParseScan (:355-470, 985-994),
 * SequentialScan::StartParseScan (sequentialscan.cpp:112-141), then the MCU loop of JPEG::ReadInternal
 * (interface/jpeg.cpp:318-348) with SequentialScan::ParseMCU / RefinementScan::ParseMCU.
 * hidden_scan: the scan comes from a FINE / RFIN box (Scan::StartParseHiddenRefinementScan, marker/scan.cpp:899-980):
 * parsed as a progressive scan that must refine by one bit, its Al counts from the true LSB.  Visible scans of a frame
 * with hidden bits address the bits above them: Al + hidden. */
static void rs_scan(oj_parser *ps, oj_bs *io, int hidden_scan)
{
  oj_info *f = ps->info;
  oj_scan sc;
  long len, data;
  int id[OJ_MAX_COMP], td[OJ_MAX_COMP], ta[OJ_MAX_COMP], i, j, c, ah, al, type;
  int mx, my, mcus_x, mcus_y, rows_made[OJ_MAX_COMP] = {0, 0, 0, 0};
  memset(&sc, 0, sizeof(sc));
  sc.ps = ps; sc.io = io;
  /* (the hidden scans of a frame of the residual kind are parsed as residual progressive scans and refine from position 0 on:
   * RefinementScan(.., residual), marker/scan.cpp:941-953) */
  type = hidden_scan ? (f->residual_type ? FT_RESIDUAL_PROGRESSIVE : FT_PROGRESSIVE) : ps->frame_type;
  len = bs_getword(io);
  if (len < 8) rs_throw(ps, RS_MALFORMED_STREAM);
  data = bs_get(io);
  if (data < 1 || data > 4) rs_throw(ps, RS_MALFORMED_STREAM);
  sc.ns = (int)data;
  if (len != sc.ns * 2 + 6) rs_throw(ps, RS_MALFORMED_STREAM);
  for (i = 0; i < sc.ns; i++) {
    data = bs_get(io);
    if (data == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
    id[i] = (int)data;
    for (j = 0; j < i; j++) if (id[j] == id[i]) rs_throw(ps, RS_MALFORMED_STREAM);
    data = bs_get(io);
    if (data == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
    td[i] = (int)(data >> 4); ta[i] = (int)(data & 15);
    if (td[i] > 3 || ta[i] > 3) rs_throw(ps, RS_MALFORMED_STREAM);
  }
  data = bs_get(io);
  if (data == BS_EOF || data > 63) rs_throw(ps, RS_MALFORMED_STREAM);
  sc.ss = (int)data;
  data = bs_get(io);
  if (data == BS_EOF || data > 63) rs_throw(ps, RS_MALFORMED_STREAM);
  sc.se = (int)data;
  data = bs_get(io);
  if (data == BS_EOF) rs_throw(ps, RS_MALFORMED_STREAM);
  ah = (int)(data >> 4); al = (int)(data & 15);
  if (ah > 13) rs_throw(ps, RS_MALFORMED_STREAM);
  if (type == FT_PROGRESSIVE) {
    if (ah && ah != al + 1) rs_throw(ps, RS_MALFORMED_STREAM);
    if (sc.se < sc.ss) rs_throw(ps, RS_MALFORMED_STREAM);
    if (sc.ss == 0 && sc.se != 0) rs_throw(ps, RS_MALFORMED_STREAM);
    if (sc.ss && sc.ns != 1) rs_throw(ps, RS_MALFORMED_STREAM);
  } else if (type == FT_RESIDUAL || type == FT_RESIDUAL_PROGRESSIVE) {
    /* marker/scan.cpp:262-272; Scan::CreateParser: Residual -> SequentialScan(.., differential, residual) whatever Ah says (:483-489),
     * ResidualProgressive -> that for Ah = 0, RefinementScan(.., residual) behind it (:411-424) */
    if (ah && ah != al + 1) rs_throw(ps, RS_MALFORMED_STREAM);
    if (sc.se < sc.ss) rs_throw(ps, RS_MALFORMED_STREAM);
    sc.residual = 1;
  } else {
    if (sc.se != 63 || sc.ss != 0) rs_throw(ps, RS_MALFORMED_STREAM);
    if (ah != 0) rs_throw(ps, RS_MALFORMED_STREAM);
  }
  if (hidden_scan) {
    if (ah != al + 1) rs_throw(ps, RS_MALFORMED_STREAM); /* "hidden refinement must refine by one bit per scan" */
    sc.refinement = 1;
    sc.lowbit = al;
  } else {
    sc.refinement = (type == FT_PROGRESSIVE || type == FT_RESIDUAL_PROGRESSIVE) && ah != 0;
    sc.lowbit = al + ps->hidden;
  }
  sc.progressive_run = sc.ss > 0 || sc.se < 63 || sc.lowbit > ps->hidden; /* sequentialscan.cpp:84-87 */
  /* Scan::CreateParser: all components must exist (Frame::FindComponent throws OBJECT_DOESNT_EXIST) */
  for (i = 0; i < sc.ns; i++) {
    for (c = 0; c < f->ncomp; c++) if (f->comp_id[c] == id[i]) break;
    if (c == f->ncomp) rs_throw(ps, RS_OBJECT_DOESNT_EXIST);
    sc.ci[i] = c;
  }
  /* EntropyParser::EntropyParser, entropyparser.cpp:60-82 */
  sc.ri = ps->restart_interval;
  sc.next_rst = 0xffd0;
  sc.togo = sc.ri;
  sc.valid = 1;
  f->restart_interval = (int)ps->restart_interval;
  /* Huffman decoders: Tables::FindDC/ACHuffmanTable (tables.cpp:1423-1452) */
  for (i = 0; i < sc.ns; i++) {
    if (sc.ss == 0 && !sc.refinement && !sc.residual) {
      oj_huff *h = &ps->huff[td[i]];
      if (!ps->have_huff) rs_throw(ps, RS_OBJECT_DOESNT_EXIST);
      if (!h->defined) huff_default(ps, h, 0, td[i] != 0);
      if (!h->built) huff_build(ps, h);
      sc.dc[i] = h;
    }
    if (sc.se || (sc.residual && sc.refinement)) { /* (refinementscan.cpp:104) */
      oj_huff *h = &ps->huff[4 + ta[i]];
      if (!ps->have_huff) rs_throw(ps, RS_OBJECT_DOESNT_EXIST);
      if (!h->defined) huff_default(ps, h, 1, ta[i] != 0);
      if (!h->built) huff_build(ps, h);
      sc.ac[i] = h;
    }
  }
  /* BlockBuffer::ResetToStartOfScan (control/blockbuffer.cpp:177-208): the transform of a component is built, with
   * the quantiser table in force NOW, when the component first appears in a scan (codestream/tables.cpp:1741-1750) */
  for (i = 0; i < sc.ns; i++) {
    c = sc.ci[i];
    if (!f->comp_seen[c]) {
      if ((ps->nested || ps->residual_ok) && (!ps->have_quant || !f->quant_defined[f->tq[c]])) {
        /* The RESIDUAL image's transforms are not built at the start of a scan: ResidualBlockHelper::AllocateBuffers builds them
         * (and looks the quantiser tables up, codestream/tables.cpp:1481-1494, 1750) when the first block is dequantised, i.e. at
         * the first request for pixels -- whatever stops either codestream, and what the colour transformer refuses, comes first
         * (a residual codestream whose DQT marker is gone AND whose entropy coded data is damaged: -1038, not -1031). */
        ps->late_quant_missing = 1;
        memset(f->cquant[c], 0, sizeof(f->cquant[c]));
        f->comp_seen[c] = 1;
        continue;
      }
      if (!ps->have_quant) rs_throw(ps, RS_OBJECT_DOESNT_EXIST);
      if (!f->quant_defined[f->tq[c]]) rs_throw(ps, RS_OBJECT_DOESNT_EXIST);
      memcpy(f->cquant[c], f->quant[f->tq[c]], sizeof(f->cquant[c]));
      f->comp_seen[c] = 1;
    }
  }
  sc.scan_for_dnl = ps->need_dnl; /* EntropyParser::EntropyParser: the frame height is still 0 */
  if (ps->need_dnl && ps->known_height > 0) {
    post_image_height(ps, ps->known_height);
    for (c = 0; c < f->ncomp; c++) f->bh[c] = ps->known_bh[c];
  }
  if (!ps->planes && !sc.scan_for_dnl) { /* headers only: skip the entropy coded data */
    const uint8_t *q = io->d + io->pos, *end = io->d + io->n;
    while (q + 1 < end && !(q[0] == 0xff && q[1] != 0x00 && q[1] != 0xff && !(q[1] >= 0xd0 && q[1] <= 0xd7))) q++;
    if (q + 1 >= end) q = end;
    io->pos = (size_t)(q - io->d);
    return;
  }
  if (sc.ns > 1) { mcus_x = f->mcus_x; mcus_y = f->mcus_y; }
  else {
    /* single-component scan: 1x1 MCUs over ceil(cw/8) x ceil(ch/8) blocks (sequentialscan.cpp:396-397) */
    mcus_x = (f->cw[sc.ci[0]] + 7) >> 3; mcus_y = (f->ch[sc.ci[0]] + 7) >> 3;
  }
  /* BlockBuffer::ResetToStartOfScan / StartMCUQuantizerRow (control/blockbuffer.cpp:177-265) with m_ulPixelHeight == 0:
   * every MCU row gets all its block rows, nothing bounds their number until the DNL marker delivered the height;
   * from then on rows end at ceil(ch / 8) as usual.  rows[] counts what exists afterwards. */
  bits_open(&sc.bits, ps, io);
  for (my = 0; sc.scan_for_dnl || my < mcus_y; my++) {
    if (sc.scan_for_dnl) {
      int more = 1;
      for (i = 0; i < sc.ns; i++) {
        const int h = (sc.ns > 1) ? f->vs[sc.ci[i]] : 1;
        int ymin, ymax;
        c = sc.ci[i];
        ymin = my * h * 8;
        ymax = ymin + h * 8;
        if (sc.dnl_found) {
          if (ymin > f->ch[c]) ymin = f->ch[c]; /* m_pulY stopped at the clipped end of the previous row */
          if (ymax > f->ch[c]) ymax = f->ch[c];
        }
        if (ymin < ymax) { if (my * h + ((ymax - ymin + 7) >> 3) > rows_made[c]) rows_made[c] = my * h + ((ymax - ymin + 7) >> 3); }
        else more = 0;
      }
      if (!more) break;
      /* a scan that never comes to a DNL marker goes on until the reference runs out of memory: nothing to compare with */
      if (my > 65536 / 8 + 8) rs_unsupported(ps);
    }
    for (mx = 0; mx < mcus_x; mx++) {
      int valid;
      /* EntropyParser::BeginReadMCU, entropyparser.hpp:147-160 */
      if (sc.scan_for_dnl && parse_dnl_marker(&sc)) valid = 0;
      else {
        if (sc.ri) {
          if (sc.togo == 0) parse_restart_marker(&sc);
          sc.togo--;
        }
        valid = sc.valid;
      }
      for (i = 0; i < sc.ns; i++) {
        int bx, by, w = (sc.ns > 1) ? f->hs[sc.ci[i]] : 1, h = (sc.ns > 1) ? f->vs[sc.ci[i]] : 1;
        c = sc.ci[i];
        for (by = 0; by < h; by++)
          for (bx = 0; bx < w; bx++) {
            int32_t dummy[64];
            int X = mx * w + bx, Y = my * h + by, k;
            /* DNL frames: a row below ceil(ch / 8) is there iff the first scan created it -- later scans reach it through
             * the list (`if (q) q = q->NextOf()`, sequentialscan.cpp:423).  (While the first scan of such a frame runs
             * every row it is at exists; the header pass keeps no coefficients.) */
            int32_t *blk = (ps->planes && X < f->bw[c] && Y < f->bh[c] && (!f->dnl || sc.scan_for_dnl || Y < f->rows[c]))
                               ? ps->planes[c] + ((size_t)Y * f->bw[c] + X) * 64 : dummy;
            if (blk == dummy) memset(dummy, 0, sizeof(dummy));
            if (valid) {
              if (sc.refinement) decode_block_refine(&sc, blk, sc.ac[i], &sc.skip[i]);
              else decode_block(&sc, blk, sc.dc[i], sc.ac[i], &sc.pred[i], &sc.skip[i]);
            } else if (!sc.refinement) {
              /* sequentialscan.cpp:416-420: block[i] = 0 for i = ScanStart..ScanStop -- natural positions, not zigzag */
              for (k = sc.ss; k <= sc.se; k++) blk[k] = 0;
            }
          }
      }
    }
  }
  if (sc.scan_for_dnl) {
    if (ps->need_dnl) rs_throw(ps, RS_MALFORMED_STREAM); /* (unreachable: the loop ends behind the marker or by a throw) */
    for (i = 0; i < sc.ns; i++) {
      c = sc.ci[i];
      f->rows[c] = rows_made[c];
      /* a damaged segment (restart markers that make the parser skip intervals ...) can leave more rows than one MCU row
       * behind the picture, and the command line's component-by-component requests walk through all of them */
      if (!ps->planes && rows_made[c] > f->bh[c]) f->bh[c] = rows_made[c];
    }
  }
}

/* Frame::ScanForScanHeader, marker/frame.cpp:863-899 */
static int rs_scan_for_scan_header(oj_parser *ps, oj_bs *io)
{
  long data = bs_getword(io);
  if (data != 0xffda) {
    RS_WARN(ps);
    if (data == BS_EOF) return 0;
    do {
      bs_lastundo(io);
      do { data = bs_get(io); } while (data != 0xff && data != BS_EOF);
      if (data == BS_EOF) break;
      bs_lastundo(io);
      data = bs_getword(io);
      if (data == BS_EOF) break;
    } while (data != 0xffda);
  }
  return data == 0xffda;
}

/* Frame::ParseTrailer, marker/frame.cpp:1016-1123 (no hidden refinement boxes here: those are driven by the XT code).
 * 1: another scan follows, 0: the frame is over. */
static int rs_frame_trailer(oj_parser *ps, oj_bs *io)
{
  /* The residual codestream at its end: Image::InputStreamOf (codestream/image.cpp:978-996) hands out the LEGACY stream
   * instead, and that one stands at its EOI (it stays in the buffer while the residual codestream is read, image.cpp:1416-1431).
   * A residual frame whose data simply ends is therefore at "its" EOI, and its hidden refinement scans are read. */
  if (ps->nested && !ps->legacy_eoi_gone && bs_peekword(io) == BS_EOF) { ps->eoi_frame = 1; return 0; }
  for (;;) {
    long marker = bs_peekword(io);
    switch (marker) {
    case 0xffb1: case 0xffb2: case 0xffb3: case 0xffb9: case 0xffba: case 0xffbb: case 0xffc0: case 0xffc1: case 0xffc2:
    case 0xffc3: case 0xffc9: case 0xffca: case 0xffcb: case 0xfff7:
      RS_WARN(ps); return 0;
    case 0xffde: RS_WARN(ps); return 0;
    case 0xffc5: case 0xffc6: case 0xffc7: case 0xffcd: case 0xffce: case 0xffcf:
      RS_WARN(ps); return 0;
    case 0xffda: return 1;
    case 0xffd9: ps->eoi_frame = 1; return 0;
    case 0xffff: bs_get(io); break;
    case 0xffd0: case 0xffd1: case 0xffd2: case 0xffd3: case 0xffd4: case 0xffd5: case 0xffd6: case 0xffd7:
      bs_getword(io); RS_WARN(ps); break;
    case BS_EOF: RS_WARN(ps); return 0;
    default:
      if (marker < 0xff00) {
        RS_WARN(ps);
        bs_get(io);
        do { marker = bs_get(io); } while (marker != 0xff && marker != BS_EOF);
        if (marker == BS_EOF) { RS_WARN(ps); return 0; }
        bs_lastundo(io);
      } else {
        while (rs_tables_incremental(ps, io)) {} /* Tables::ParseTables, tables.cpp:968-980 */
      }
    }
  }
}

/* Image::ParseTrailer, codestream/image.cpp:1408-1497. 1: something that is not the end follows. */
static int rs_image_trailer(oj_parser *ps, oj_bs *io)
{
  int first;
  for (first = 1;; first = 0) {
    long marker = bs_peekword(io);
    if (marker == 0xffd9) { ps->eoi_image = 1; bs_getword(io); return 0; }
    else if (marker == 0xffff) bs_get(io);
    else if (marker == BS_EOF) { if (!(ps->nested && first && !ps->legacy_eoi_gone)) RS_WARN(ps); return 0; } /* (image.cpp:1320-1323: the residual codestream may simply end) */
    else if (marker < 0xff00) {
      RS_WARN(ps);
      bs_get(io);
      do { marker = bs_get(io); } while (marker != 0xff && marker != BS_EOF);
      if (marker == BS_EOF) { RS_WARN(ps); return 0; }
      bs_lastundo(io);
    } else return 1;
  }
}

/* JPEG::ReadInternal, interface/jpeg.cpp:244-354, for one codestream. */
static void rs_run(oj_parser *ps, oj_bs *io)
{
  oj_info *f = ps->info;
  if (!g_scan_order_ready) build_scan_order();
  f->adobe_transform = -1;
  /* Decoder::ParseHeaderIncremental, codestream/decoder.cpp:77-108 */
  if (bs_getword(io) != 0xffd8) rs_throw(ps, RS_MALFORMED_STREAM);
  ps->header_part = !ps->nested && !ps->in_memory;
  while (rs_tables_incremental(ps, io)) {}
  ps->header_part = 0;
  for (;;) { /* frames */
    rs_parse_frame_header(ps, io);
    for (;;) { /* scans: Frame::StartParseScan, marker/frame.cpp:796-861 */
      while (rs_tables_incremental(ps, io)) {}
      /* (a residual codestream that is through here: the search for a scan header works on the legacy stream -- InputStreamOf,
       * see rs_frame_trailer -- and takes its EOI away: one warning like at the end of this stream, but the trailers that
       * follow find the end of the file, not an EOI) */
      if (ps->nested && bs_peekword(io) == BS_EOF) ps->legacy_eoi_gone = 1;
      if (rs_scan_for_scan_header(ps, io)) {
        rs_scan(ps, io, 0);
        if (!ps->planes && !ps->walk_all) return; /* header-only walk stops behind the first scan header */
        if (rs_frame_trailer(ps, io)) continue;
        if (!rs_image_trailer(ps, io)) return;
        /* The residual codestream: Image::ParseResidualStream (codestream/image.cpp:1318-1331) hands the SAME frame back when
         * its image trailer finds a marker (the parent's m_pCurrent, m_bReceivedFrameHeader set): no frame header is read,
         * the frame goes on looking for scans. */
        if (ps->nested || ps->in_memory) continue; /* (... and Image::ParseAlphaChannel, image.cpp:1386-1397, does the same for the alpha codestream) */
        break; /* next frame: a second frame header throws */
      } else {
        /* no scan: end of frame (interface/jpeg.cpp:305-317) */
        if (rs_frame_trailer(ps, io)) continue;
        if (!rs_image_trailer(ps, io)) return;
        if (ps->nested || ps->in_memory) continue; /* (... and Image::ParseAlphaChannel, image.cpp:1386-1397, does the same for the alpha codestream) */
        rs_throw(ps, RS_INVALID_PARAMETER); /* the reference dereferences a NULL frame here */
      }
    }
  }
}

/* Maps the outcome of the state machine to the oracle's return codes. */
static int rs_result(const oj_parser *ps, int thrown)
{
  if (!thrown) return OJ_OK;
  if (ps->unsupported) return OJ_ERR_UNSUPPORTED;
  if (ps->err == RS_UNEXPECTED_EOF) return OJ_ERR_EOF;
  if (ps->err == RS_OUT_OF_MEMORY) return OJ_ERR_NOMEM;
  return OJ_ERR_MALFORMED;
}

#define BOXID_(a, b, c, d) (((uint32_t)(a) << 24) | ((uint32_t)(b) << 16) | ((uint32_t)(c) << 8) | (uint32_t)(d))
/* A merging specification box in a file that has no residual codestream -- what the reference's encoder writes for `-c`
 * (SPEC{OCON, LTRF = identity}, codestream/tables.cpp:625-632) and for grey scale pictures.  Tables::LTrafoTypeOf
 * (codestream/tables.cpp:1994-2021) takes the L transformation from the box before it looks at the Adobe marker or the
 * component count; the transformer is the Extended flavour (colortransformerfactory.cpp:232-236): identity L tables, identity
 * C transformation, nothing to merge, clamp to 2^P - 1 (colortrafo/ycbcrtrafo.cpp:861-878, 921-936) -- the plain picture.
 * Returns 0 and sets f->ycbcr, or the reference's error code, or 1: a specification this restatement does not follow. */
static int spec_without_residual(const oj_box *spec, const oj_box *boxes, int nboxes, oj_info *f)
{
  int ltrafo = 255, ctrafo = 255, ocon = -1, tables = 0, hidden = 0, have_mtx[16] = {0}, b;
  size_t j;
  for (j = 0; j + 8 <= spec->len;) {
    const uint32_t l = ((uint32_t)rd16(spec->data + j) << 16) | (uint32_t)rd16(spec->data + j + 2);
    const uint32_t t = ((uint32_t)rd16(spec->data + j + 4) << 16) | (uint32_t)rd16(spec->data + j + 6);
    const uint8_t *pl = spec->data + j + 8;
    if (l < 8 || j + l > spec->len) return 1;
    if (t == BOXID_('L', 'T', 'R', 'F')) ltrafo = pl[0] >> 4;
    else if (t == BOXID_('C', 'T', 'R', 'F')) ctrafo = pl[0] >> 4;
    else if (t == BOXID_('O', 'C', 'O', 'N')) ocon = pl[0];
    else if (t == BOXID_('R', 'S', 'P', 'C')) hidden = pl[0];
    else if (t == BOXID_('R', 'T', 'R', 'F') || t == BOXID_('R', 'D', 'C', 'T') || t == BOXID_('L', 'D', 'C', 'T')) { if (t == BOXID_('L', 'D', 'C', 'T') && pl[0]) return 1; }
    else if (t == BOXID_('M', 'T', 'R', 'X')) { if (l >= 9) have_mtx[pl[0] >> 4] = 1; } /* (a matrix or a curve nobody names changes nothing) */
    else if (t == BOXID_('F', 'T', 'R', 'X') || (t & 0xffffff) == (BOXID_(0, 'P', 'T', 'S')) || t == BOXID_('D', 'T', 'R', 'F') || t == BOXID_('S', 'T', 'R', 'F'))
      tables = 1; /* point transformations, float matrices, transformations of the other profiles */
    /* (a type MergingSpecBox::CreateBox does not know is skipped, boxes/superbox.cpp:161-176) */
    j += l;
  }
  for (b = 0; b < nboxes; b++)
    if (boxes[b].type == BOXID_('M', 'T', 'R', 'X') && boxes[b].complete && boxes[b].len >= 1) have_mtx[boxes[b].data[0] >> 4] = 1;
  if (f->ncomp == 1 && ltrafo != 255) return RS_MALFORMED_STREAM; /* "Base transformation box exists even though the number of components is one" */
  if (ltrafo == 0 || ltrafo == 3 || ltrafo == 4) return RS_MALFORMED_STREAM; /* Zero, JPEG_LS, RCT: "Found an invalid base transformation" */
  /* a free-form transformation nobody defined: "the base transformation specified in the codestream does not exist"
   * (colortransformerfactory.cpp:379-383) */
  if (ltrafo != 255 && ltrafo >= 5 && !have_mtx[ltrafo]) return RS_OBJECT_DOESNT_EXIST;
  if (tables || hidden || (ctrafo != 255 && ctrafo != 1) || (ltrafo != 255 && ltrafo >= 5)) return 1;
  /* (the lossless flag changes the residual's side only; the lookup indices are read and never used: xt_decode_common) */
  if (ocon >= 0 && ((ocon >> 4) != 0 || (ocon & 0x04) || !(ocon & 0x02))) return 1; /* more bits, float, or wrap-around */
  if (ltrafo == 2 && f->ncomp != 3) return 1;
  if (ltrafo == 2) f->ycbcr = 1;
  else if (ltrafo == 1) f->ycbcr = 0;
  return 0;
}

static int walk(oj_parser *ps, int32_t *const planes[OJ_MAX_COMP])
{
  oj_bs io;
  oj_box *volatile own = NULL;
  volatile int thrown = 0;
  ps->planes = planes;
  ps->err = 0; ps->unsupported = 0; ps->warnings = 0;
  if (!ps->boxes) { /* the boxes are followed even when nobody asks for them: their framing is checked */
    own = (oj_box *)calloc(OJ_MAX_BOXES, sizeof(oj_box));
    if (!own) return OJ_ERR_NOMEM;
    ps->boxes = own; ps->nboxes = 0;
  }
  bs_open(&io, ps->data, ps->len);
  io.in_memory = ps->nested || ps->in_memory;
  if (setjmp(ps->jb) == 0) rs_run(ps, &io);
  else thrown = 1;
  if (!thrown) {
    oj_info *f = ps->info;
    if (!ps->have_frame) { thrown = 1; ps->err = RS_MALFORMED_STREAM; }
    /* a frame that announced a DNL marker and never met one keeps zero lines; reading it succeeds, and the first request
     * for pixels fails: "Rectangle MaxY underflow, must be >= 0" (codestream/rectanglerequest.cpp:107-115, cmd/reconstruct.cpp:334-342) */
    else if (f->height == 0) { thrown = 1; ps->err = RS_OVERFLOW_PARAMETER; }
    /* codestream/tables.cpp:2021-2030: three components and no Adobe "None" -> YCbCr, else identity */
    f->ycbcr = (f->ncomp == 3 && f->adobe_transform != 0) ? 1 : 0;
    if (!thrown && !ps->nested && !ps->xt_legacy) {
      const oj_box *spec = NULL, *resi = NULL;
      int b;
      for (b = 0; b < ps->nboxes; b++) {
        if (!ps->boxes[b].complete) continue; /* a box that never filled up stays in the list unparsed: nobody looks at it */
        if (ps->boxes[b].type == BOXID_('S', 'P', 'E', 'C')) spec = &ps->boxes[b];
        if (ps->boxes[b].type == BOXID_('R', 'E', 'S', 'I')) resi = &ps->boxes[b];
      }
      if (spec && !resi) {
        const int v = spec_without_residual(spec, ps->boxes, ps->nboxes, f);
        /* (a header-only walk has not seen the whole file: it takes the transformation, the verdict waits for the full walk) */
        if (v == 1) { if (planes) { thrown = 1; ps->unsupported = 1; ps->err = RS_NOT_IMPLEMENTED; } }
        else if (v) { thrown = 1; ps->err = v; }
      }
    }
  }
  if (own) {
    int b;
    for (b = 0; b < ps->nboxes; b++) free(own[b].data);
    free(own);
    ps->boxes = NULL; ps->nboxes = 0;
  }
  ps->info->ref_error = thrown ? ps->err : 0;
  ps->info->warnings = ps->warnings;
  return rs_result(ps, thrown);
}

int oj_read_info(const uint8_t *data, size_t len, oj_info *info)
{
  oj_parser ps;
  memset(&ps, 0, sizeof(ps));
  memset(info, 0, sizeof(*info));
  ps.data = data; ps.len = len; ps.info = info;
  return walk(&ps, NULL);
}

/* ... of the codestream in a RESI box (its frame may be of the residual type) */
int oj_read_info_residual(const uint8_t *data, size_t len, oj_info *info);
static int read_residual_info(const uint8_t *data, size_t len, oj_info *info)
{
  oj_parser ps;
  memset(&ps, 0, sizeof(ps));
  memset(info, 0, sizeof(*info));
  ps.data = data; ps.len = len; ps.info = info; ps.residual_ok = 1;
  ps.in_memory = 1; /* (the payload of a box: a memory stream, whose skips beyond the end throw -- see rs_run) */
  return walk(&ps, NULL);
}

static int decode_hidden_scans(oj_parser *ps, const oj_box *boxes, int nboxes, uint32_t type, int32_t *const planes[OJ_MAX_COMP]);

static int decode_coefficients_as(const uint8_t *data, size_t len, const oj_info *info,
                                  int32_t *const planes[OJ_MAX_COMP], int nested)
{
  oj_parser ps;
  oj_info tmp;
  oj_box *boxes = NULL;
  oj_info *out = (oj_info *)info; /* the scan state (per-component tables, components seen, error code) is reported back */
  int c, rc;
  memset(&ps, 0, sizeof(ps));
  memset(&tmp, 0, sizeof(tmp));
  ps.data = data; ps.len = len; ps.info = &tmp; ps.nested = nested;
  ps.known_height = info->dnl ? info->height : 0;
  for (c = 0; c < OJ_MAX_COMP; c++) ps.known_bh[c] = info->bh[c];
  for (c = 0; c < info->ncomp; c++)
    memset(planes[c], 0, (size_t)info->bw[c] * info->bh[c] * 64 * sizeof(int32_t));
  if (!nested) { /* the boxes outlive the walk: refinement boxes are read behind the frame's last visible scan */
    boxes = (oj_box *)calloc(OJ_MAX_BOXES, sizeof(oj_box));
    if (!boxes) return OJ_ERR_NOMEM;
    ps.boxes = boxes;
  }
  rc = walk(&ps, planes);
  /* Frame::ParseTrailer, marker/frame.cpp:1063-1070: at the EOI the frame turns to its FINE boxes whether or not a merging
   * specification says how many bits hide in them -- with none (a plain JPEG, or a specification that never arrived) they
   * refine the bits the visible scans left as they are */
  if (!rc && boxes && ps.eoi_frame) {
    rc = decode_hidden_scans(&ps, boxes, ps.nboxes, BOXID_('F', 'I', 'N', 'E'), planes);
    if (rc) tmp.ref_error = ps.err;
  }
  if (boxes) { int b; for (b = 0; b < ps.nboxes; b++) free(boxes[b].data); free(boxes); }
  memcpy(out->cquant, tmp.cquant, sizeof(tmp.cquant));
  memcpy(out->comp_seen, tmp.comp_seen, sizeof(tmp.comp_seen));
  out->scan_state_valid = tmp.scan_state_valid;
  out->ref_error = tmp.ref_error;
  out->warnings = tmp.warnings;
  if (tmp.dnl) memcpy(out->rows, tmp.rows, sizeof(tmp.rows));
  if (!rc) out->ycbcr = tmp.ycbcr; /* the full walk has seen every box */
  return rc;
}

int oj_read_info_residual(const uint8_t *data, size_t len, oj_info *info) { return read_residual_info(data, len, info); }

int oj_decode_coefficients(const uint8_t *data, size_t len, const oj_info *info,
                           int32_t *const planes[OJ_MAX_COMP])
{
  return decode_coefficients_as(data, len, info, planes, 0);
}

/* ... of a JPEG XT residual codestream (the payload of the RESI box), walked the way the reference walks it from inside the
 * legacy image's trailer: Image::ParseResidualStream, codestream/image.cpp:1264-1334 (see rs_run). */
int oj_decode_coefficients_residual(const uint8_t *data, size_t len, const oj_info *info,
                                    int32_t *const planes[OJ_MAX_COMP])
{
  return decode_coefficients_as(data, len, info, planes, 1);
}

/* ------------------------------------------------------------------------------------------
 * Dequantisation + inverse DCT: dct/idct.cpp:226-339 (IDCT<4,LONG,false,false>), constants
 * dct/idct.hpp:70-77 (FIX_BITS = 9, INTERMEDIATE_BITS = 0) and idct.cpp:65-78 (TO_FIX).
 * All arithmetic wraps modulo 2^32 like the reference's LONG does on this platform; the
 * rounding additions are carried out in 64 bits because the reference adds `1L << n` (a 64-bit
 * long on LP64) before shifting.
 * ---------------------------------------------------------------------------------------- */
#define FIX9(x) ((int32_t)((x) * 512.0 + 0.5))
static int32_t w32(int64_t v) { return (int32_t)(uint32_t)(uint64_t)v; }
static int32_t mul32(int32_t a, int32_t b) { return w32((int64_t)a * (int64_t)b); }
static int32_t add32(int32_t a, int32_t b) { return w32((int64_t)a + (int64_t)b); }
static int32_t sub32(int32_t a, int32_t b) { return w32((int64_t)a - (int64_t)b); }
static int32_t shl32(int32_t a, int n) { return w32((int64_t)((uint64_t)(int64_t)a << n)); }

static void idct_1d(const int32_t s[8], int32_t o[8])
{
  /* even part */
  int32_t z1 = mul32(add32(s[2], s[6]), FIX9(0.541196100));
  int32_t tmp2 = add32(z1, mul32(s[6], -FIX9(1.847759065)));
  int32_t tmp3 = add32(z1, mul32(s[2], FIX9(0.765366865)));
  int32_t tmp0 = shl32(add32(s[0], s[4]), 9);
  int32_t tmp1 = shl32(sub32(s[0], s[4]), 9);
  int32_t tmp10 = add32(tmp0, tmp3), tmp13 = sub32(tmp0, tmp3);
  int32_t tmp11 = add32(tmp1, tmp2), tmp12 = sub32(tmp1, tmp2);
  /* odd part */
  int32_t t0 = s[7], t1 = s[5], t2 = s[3], t3 = s[1];
  int32_t tz1 = add32(t0, t3), tz2 = add32(t1, t2), tz3 = add32(t0, t2), tz4 = add32(t1, t3);
  int32_t z5 = mul32(add32(tz3, tz4), FIX9(1.175875602));
  int32_t z2, z3, z4;
  tmp0 = mul32(t0, FIX9(0.298631336));
  tmp1 = mul32(t1, FIX9(2.053119869));
  tmp2 = mul32(t2, FIX9(3.072711026));
  tmp3 = mul32(t3, FIX9(1.501321110));
  z1 = mul32(tz1, -FIX9(0.899976223));
  z2 = mul32(tz2, -FIX9(2.562915447));
  z3 = add32(mul32(tz3, -FIX9(1.961570560)), z5);
  z4 = add32(mul32(tz4, -FIX9(0.390180644)), z5);
  tmp0 = add32(tmp0, add32(z1, z3));
  tmp1 = add32(tmp1, add32(z2, z4));
  tmp2 = add32(tmp2, add32(z2, z3));
  tmp3 = add32(tmp3, add32(z1, z4));
  o[0] = add32(tmp10, tmp3); o[7] = sub32(tmp10, tmp3);
  o[1] = add32(tmp11, tmp2); o[6] = sub32(tmp11, tmp2);
  o[2] = add32(tmp12, tmp1); o[5] = sub32(tmp12, tmp1);
  o[3] = add32(tmp13, tmp0); o[4] = sub32(tmp13, tmp0);
}

/* The same butterfly for IDCT<4,QUAD>: T = 64 bits (dct/idct.cpp with FIXED = QUAD).  Nothing wraps, with one
 * exception: the second pass forms `(dptr[0 << 3] + dptr[4 << 3]) << FIX_BITS` (idct.cpp:297-298) from LONG operands,
 * so this sum and shift are 32-bit before they are widened (second_pass). */
static void idct_1d_quad(const int64_t s[8], int64_t o[8], int second_pass)
{
  int64_t z1 = (s[2] + s[6]) * FIX9(0.541196100);
  int64_t tmp2 = z1 + s[6] * -FIX9(1.847759065);
  int64_t tmp3 = z1 + s[2] * FIX9(0.765366865);
  int64_t tmp0 = second_pass ? (int64_t)shl32(add32((int32_t)s[0], (int32_t)s[4]), 9) : (s[0] + s[4]) * 512;
  int64_t tmp1 = second_pass ? (int64_t)shl32(sub32((int32_t)s[0], (int32_t)s[4]), 9) : (s[0] - s[4]) * 512;
  int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  int64_t t0 = s[7], t1 = s[5], t2 = s[3], t3 = s[1];
  int64_t tz1 = t0 + t3, tz2 = t1 + t2, tz3 = t0 + t2, tz4 = t1 + t3;
  int64_t z5 = (tz3 + tz4) * FIX9(1.175875602), z2, z3, z4;
  tmp0 = t0 * FIX9(0.298631336);
  tmp1 = t1 * FIX9(2.053119869);
  tmp2 = t2 * FIX9(3.072711026);
  tmp3 = t3 * FIX9(1.501321110);
  z1 = tz1 * -FIX9(0.899976223);
  z2 = tz2 * -FIX9(2.562915447);
  z3 = tz3 * -FIX9(1.961570560) + z5;
  z4 = tz4 * -FIX9(0.390180644) + z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3;
  o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
  o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1;
  o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}

void oj_idct_block(int32_t out[64], const int32_t coef[64], const uint16_t quant[64], int precision)
{
  int32_t tmp[64];
  int r, c, k;
  int32_t dcoffset;
  if (!coef) { memset(out, 0, 64 * sizeof(int32_t)); return; }
  if (precision > 12) {
    /* codestream/tables.cpp:1876-1891: IDCT<4,QUAD>.  The dequantising products and the DC offset are still LONG
     * expressions (`source[k] * qnt[k] + dcoffset`, idct.cpp:238-259) and the pass results are stored as LONG. */
    dcoffset = (int32_t)(1L << (precision - 1)) << (4 + 3);
    for (r = 0; r < 8; r++) {
      int64_t s[8], o[8];
      for (k = 0; k < 8; k++) s[k] = mul32(coef[r * 8 + k], (int32_t)quant[r * 8 + k] << 4);
      if (r == 0) s[0] = add32((int32_t)s[0], dcoffset);
      idct_1d_quad(s, o, 0);
      for (k = 0; k < 8; k++) tmp[r * 8 + k] = w32((o[k] + 256) >> 9);
    }
    for (c = 0; c < 8; c++) {
      int64_t s[8], o[8];
      for (k = 0; k < 8; k++) s[k] = tmp[k * 8 + c];
      idct_1d_quad(s, o, 1);
      for (k = 0; k < 8; k++) out[k * 8 + c] = w32((o[k] + 2048) >> 12);
    }
    return;
  }
  /* caller passes dcoffset = 1 << (P-1) (blockbitmaprequester.cpp:1048); shifted by preshift + 3 */
  dcoffset = (int32_t)(1L << (precision - 1)) << (4 + 3);
  for (r = 0; r < 8; r++) {
    int32_t s[8], o[8];
    for (k = 0; k < 8; k++) s[k] = mul32(coef[r * 8 + k], (int32_t)quant[r * 8 + k] << 4);
    if (r == 0) s[0] = add32(s[0], dcoffset);
    idct_1d(s, o);
    for (k = 0; k < 8; k++) tmp[r * 8 + k] = w32(((int64_t)o[k] + 256) >> 9);
  }
  for (c = 0; c < 8; c++) {
    int32_t s[8], o[8];
    for (k = 0; k < 8; k++) s[k] = tmp[k * 8 + c];
    idct_1d(s, o);
    for (k = 0; k < 8; k++) out[k * 8 + c] = w32(((int64_t)o[k] + 2048) >> 12);
  }
}

void oj_idct_plane(int32_t *samples, const int32_t *coef, int bw, int bh, const uint16_t quant[64],
                   int precision)
{
  int bx, by, y;
  for (by = 0; by < bh; by++)
    for (bx = 0; bx < bw; bx++) {
      int32_t o[64];
      oj_idct_block(o, coef + ((size_t)by * bw + bx) * 64, quant, precision);
      for (y = 0; y < 8; y++)
        memcpy(samples + ((size_t)(by * 8 + y) * bw + bx) * 8, o + y * 8, 8 * sizeof(int32_t));
    }
}

/* ------------------------------------------------------------------------------------------
 * Upsampling: upsampling/upsampler.cpp.  The 8x8 buffer is filled by the vertical core from
 * three line pointers and then filtered IN PLACE by the horizontal core, exactly like the
 * reference, so its aliasing behaviour (a freshly written output feeding a later one) is kept.
 * ---------------------------------------------------------------------------------------- */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* value at line `row`, line-buffer index `idx` (reference: line->m_pData[idx], data starts at +1,
 * m_pData[0] = sample 0 and m_pData[cw+1] = sample cw-1: upsamplerbase.cpp:322-323) */
static int32_t line_at(const int32_t *plane, int pitch, int cw, int row, int idx)
{
  return plane[(size_t)row * pitch + clampi(idx - 1, 0, cw - 1)];
}

/* The two filter cores over lines addressed by number: `at(ctx, row, idx)` is line `row`'s m_pData[idx]; rows advance
 * like the reference's list walk (`if (bot->m_pNext) bot = bot->m_pNext`) with `limit` = one past the last line there is. */
typedef int32_t (*oj_line_fn)(const void *ctx, int row, int idx);
static void upsample_core(int32_t out[64], oj_line_fn at, const void *ctx, int top, int cur, int bot, int limit, int x,
                          int sx, int sy, int xmod, int ymod)
{
  int l, j;
  int32_t *target = out;
#define T(j) at(ctx, top, x + (j))
#define C(j) at(ctx, cur, x + (j))
#define B(j) at(ctx, bot, x + (j))
#define ADVANCE() do { top = cur; cur = bot; if (bot + 1 < limit) bot++; } while (0)
  for (l = 0; l < 8; l++, target += 8) {
    switch (sy) {
    case 1: /* :118-131 */
      for (j = 0; j < 8; j++) target[j] = C(j);
      if (cur + 1 < limit) cur++;
      break;
    case 2: /* :136-168 */
      if (ymod == 0) {
        for (j = 0; j < 8; j++) target[j] = w32((int64_t)T(j) + 3 * (int64_t)C(j) + ((j & 1) ? 1 : 2)) >> 2;
        ymod = 1;
      } else {
        for (j = 0; j < 8; j++) target[j] = w32((int64_t)B(j) + 3 * (int64_t)C(j) + ((j & 1) ? 2 : 1)) >> 2;
        ymod = 0; ADVANCE();
      }
      break;
    case 3: /* :174-215 */
      if (ymod == 0) {
        for (j = 0; j < 8; j++) target[j] = w32((int64_t)T(j) + 3 * (int64_t)C(j) + ((j & 1) ? 1 : 2)) >> 2;
        ymod = 1;
      } else if (ymod == 1) {
        for (j = 0; j < 8; j++) target[j] = C(j);
        ymod = 2;
      } else {
        for (j = 0; j < 8; j++) target[j] = w32((int64_t)B(j) + 3 * (int64_t)C(j) + ((j & 1) ? 2 : 1)) >> 2;
        ymod = 0; ADVANCE();
      }
      break;
    case 4: /* :221-271 */
      if (ymod == 0) {
        for (j = 0; j < 8; j++) target[j] = w32(3 * (int64_t)T(j) + 5 * (int64_t)C(j) + ((j & 1) ? 3 : 4)) >> 3;
        ymod = 1;
      } else if (ymod == 1) {
        for (j = 0; j < 8; j++) target[j] = w32((int64_t)T(j) + 7 * (int64_t)C(j) + ((j & 1) ? 4 : 3)) >> 3;
        ymod = 2;
      } else if (ymod == 2) {
        for (j = 0; j < 8; j++) target[j] = w32((int64_t)B(j) + 7 * (int64_t)C(j) + ((j & 1) ? 3 : 4)) >> 3;
        ymod = 3;
      } else {
        for (j = 0; j < 8; j++) target[j] = w32(3 * (int64_t)B(j) + 5 * (int64_t)C(j) + ((j & 1) ? 3 : 4)) >> 3;
        ymod = 0; ADVANCE();
      }
      break;
    }
  }
#undef T
#undef C
#undef B
#undef ADVANCE
  /* horizontal, in place on each buffer line; src = target + 1 */
#define F2(a, b, r) (w32((int64_t)(a) + 3 * (int64_t)(b) + (r)) >> 2)
#define F8(wa, a, wb, b, r) (w32((wa) * (int64_t)(a) + (wb) * (int64_t)(b) + (r)) >> 3)
  for (l = 0, target = out; l < 8; l++, target += 8) {
    int32_t *src = target + 1, *o = target, t;
    switch (sx) {
    case 1: break;
    case 2: /* :283-307 */
      o[7] = F2(src[4], src[3], 1);
      o[6] = F2(src[2], src[3], 2);
      o[5] = F2(src[3], src[2], 1);
      o[4] = F2(src[1], src[2], 2);
      o[3] = F2(src[2], src[1], 1);
      o[2] = F2(src[0], src[1], 2); t = src[0];
      o[1] = F2(src[1], t, 1); /* src[1] is o[2] by now */
      o[0] = F2(src[-1], t, 2);
      break;
    case 3: /* :313-361 */
      if (xmod == 0) {
        o[7] = src[2];
        o[6] = F2(src[1], src[2], 2);
        o[5] = F2(src[2], src[1], 1);
        o[4] = src[1];
        o[3] = F2(src[0], src[1], 2);
        o[2] = F2(src[1], src[0], 1);
        o[0] = F2(src[-1], src[0], 2);
        o[1] = src[0];
      } else if (xmod == 1) {
        o[7] = F2(src[3], src[2], 1);
        o[6] = src[2];
        o[5] = F2(src[1], src[2], 2);
        o[4] = F2(src[2], src[1], 1);
        o[3] = src[1]; t = src[0];
        o[2] = F2(t, src[1], 2);
        o[1] = F2(src[1], t, 1);
        o[0] = t;
      } else {
        o[7] = F2(src[2], src[3], 2);
        o[6] = F2(src[3], src[2], 1);
        o[5] = src[2];
        o[4] = F2(src[1], src[2], 2);
        o[3] = F2(src[2], src[1], 1);
        o[2] = src[1]; t = src[0];
        o[1] = F2(t, src[1], 2);
        o[0] = F2(src[1], t, 1);
      }
      break;
    case 4: /* :367-387 */
      o[7] = F8(3, src[2], 5, src[1], 1);
      o[6] = F8(1, src[2], 7, src[1], 2);
      o[5] = F8(1, src[0], 7, src[1], 1);
      o[4] = F8(3, src[0], 5, src[1], 2); t = src[0];
      o[3] = F8(3, src[1], 5, t, 1);
      o[2] = F8(1, src[1], 7, t, 2);
      o[1] = F8(1, src[-1], 7, t, 1);
      o[0] = F8(3, src[-1], 5, t, 2);
      break;
    }
  }
#undef F2
#undef F8
}

typedef struct { const int32_t *plane; int pitch, cw; } oj_plane_ctx;
static int32_t plane_line_at(const void *ctx, int row, int idx)
{
  const oj_plane_ctx *p = (const oj_plane_ctx *)ctx;
  return line_at(p->plane, p->pitch, p->cw, row, idx);
}

void oj_upsample_block(int32_t out[64], const int32_t *plane, int pitch, int cw, int ch, int sx,
                       int sy, int X0, int Y0)
{
  /* upsampler.cpp:83-117 */
  oj_plane_ctx ctx;
  int y = Y0 / sy, x = X0 / sx + 1;
  int top = y > 0 ? y - 1 : 0, cur = y, bot;
  ctx.plane = plane; ctx.pitch = pitch; ctx.cw = cw;
  if (cur > ch - 1) cur = ch - 1; /* blocks entirely below the last stored line are never output */
  if (top > ch - 1) top = ch - 1;
  bot = cur + 1 < ch ? cur + 1 : cur;
  if (sx > 1) x--;
  upsample_core(out, plane_line_at, &ctx, top, cur, bot, ch, x, sx, sy, X0 % sx, Y0 % sy);
}

/* ------------------------------------------------------------------------------------------
 * Colour transformation + clamp + store: colortrafo/ycbcrtrafo.cpp:842-850 (YCbCr), :852-856
 * (identity), :921-936 (clamp), tools/numerics.hpp:57-71 (FIX_BITS = 13, COLOR_BITS = 4),
 * colortrafo/colortransformerfactory.cpp:136-138 (matrix).
 * ---------------------------------------------------------------------------------------- */
#define FIX13(x) ((int64_t)((x) * 8192.0 + 0.5))
static int64_t clampmax(int64_t v, int64_t max) { return v < 0 ? 0 : (v > max ? max : v); }
/* INVERT_NEGS, colortrafo/ycbcrtrafo.cpp:66: two's complement -> sign-magnitude half-float bit pattern */
static int16_t invert_negs(int16_t w) { return (int16_t)(((w >> 15) & 0x7fff) ^ w); }

/* XT profile C parameters (NULL for a plain JPEG): colortrafo/colortransformerfactory.cpp:206-594 */
typedef struct {
  const oj_info *rinfo;             /* residual frame geometry */
  int32_t *const *rplanes;          /* residual coefficient planes */
  const int32_t *ltable[3];         /* L lookup tables (2^8 entries) or NULL = none (identity) */
  int ltrafo_ycbcr, rtrafo_ycbcr;   /* 1: matrix branch (standard YCbCr or free-form, lmat / rmat), 0: identity branch */
  int64_t lmat[9], rmat[9], cmat[9]; /* L, R and C transformations, 13 fractional bits (DefineLTransformation etc.) */
  const int32_t *qlut[3];           /* Q tables, 2^(Pr + 4) entries, NULL = the identity (a shift) */
  const int32_t *r2lut[3];          /* R2 tables, 2^(16 + 4) entries, NULL = the identity (x + 8) >> 4 */
  int rbypass, rnoise;              /* RDCT box: residual DCT bypassed (control/residualblockhelper.cpp:203-231), noise shaping */
  int rct;                          /* R transformation = RCT (lossless / near-lossless coding: colortrafo/ycbcrtrafo.cpp:752-766) */
  int rbits;                        /* fractional bits of the residual path: 4, 1 (RCT: a precision bit really) or 0 (identity,
                                       lossless): Tables::FractionalColorBitsOf, codestream/tables.cpp:1621-1660 */
  int64_t outmax, outshift;         /* 2^(8 + extra bits) - 1 and its half */
  int is_float, clamp;              /* OCON: cast to float (half codes), clamping */
  int nc;                           /* components: 3, or 1 (grey scale; every transformation is the identity then) */
  int no_residual;                  /* the legacy codestream never came to its EOI: the reference has not parsed the residual
                                       codestream and merges nothing (rr = m_lOutDCShift, colortrafo/ycbcrtrafo.cpp:744-746) */
} oj_xt;

/* ResidualBlockHelper::DequantizeResidual without a DCT (control/residualblockhelper.cpp:203-231, quantiser of
 * AllocateBuffers :351-364): one block, residual coefficients in their stored (natural) positions -> samples * 16. */
static void xt_bypass_block(int32_t *dst, const int32_t *res, int32_t quant, int noise, int32_t dcshift)
{
  int x, y, dx, dy;
  for (y = 0; y < 64; y += 16)
    for (x = 0; x < 8; x += 2) {
      int32_t avg = 0;
      if (noise)
        for (dy = 0; dy < 16; dy += 8)
          for (dx = 0; dx < 2; dx++) avg += res[x + dx + y + dy] * quant;
      avg = (avg + 2) >> 2;
      for (dy = 0; dy < 16; dy += 8)
        for (dx = 0; dx < 2; dx++) {
          int32_t v = res[x + dx + y + dy] * quant;
          if (noise && v > avg - quant && v < avg + quant) v = avg;
          dst[x + dx + y + dy] = v + dcshift;
        }
    }
}

/* One pixel of the JPEG XT merge: v = the legacy sample after the L transformation (integer), rk = the residual samples * 16
 * as the residual transform (or the bypass) left them.  colortrafo/ycbcrtrafo.cpp:750-829 (residual), :861-878 (L-LUT,
 * C transformation, merge), :897-955 (half clamp). */
#define W32(x) ((int64_t)(int32_t)(uint32_t)(uint64_t)(x)) /* what a LONG keeps of it */
static void xt_merge_pixel(const oj_xt *xt, int64_t maxval, const int64_t vin[3], const int32_t rk[3], uint16_t out[3])
{
  const oj_info *r = xt->rinfo; /* (NULL where nothing is merged) */
  const int64_t rmax16 = r ? ((((int64_t)1 << r->precision)) << 4) - 1 : 0; /* ((m_lRMax + 1) << COLOR_BITS) - 1 */
  const int64_t rmax = r ? ((int64_t)1 << r->precision) - 1 : 0;            /* m_lRMax */
  const int64_t omax16 = ((xt->outmax + 1) << 4) - 1;
  int64_t rr[3], q3[3], lv[3], v[3];
  int c;
  const int nc = xt->nc == 1 ? 1 : 3;
  rr[0] = rr[1] = rr[2] = 0; q3[1] = q3[2] = 0; lv[1] = lv[2] = 0;
  if (xt->no_residual) { rr[0] = rr[1] = rr[2] = xt->outshift; goto merge; }
  if (xt->rct) {
    /* colortrafo/ycbcrtrafo.cpp:752-766: the Q tables on the samples as they are (one extra bit, no fractional ones), then the
     * reversible transformation with wrap-around (all LONG) */
    int64_t y = xt->qlut[0][clampmax(rk[0], rmax)], cb = xt->qlut[1][clampmax(rk[1], rmax)], cr = xt->qlut[2][clampmax(rk[2], rmax)];
    y = W32(y) >> 1;
    cb = W32(cb - (xt->outshift << 1));
    cr = W32(cr - (xt->outshift << 1));
    rr[1] = W32(y - (W32(cb + cr) >> 2)) & xt->outmax;
    rr[0] = W32(cr + rr[1]) & xt->outmax;
    rr[2] = W32(cb + rr[1]) & xt->outmax;
    goto merge;
  }
  if (!xt->rtrafo_ycbcr && !xt->clamp) { /* identity without clamping (:797-801, :820-822): the Q table alone, no fractional bits */
    for (c = 0; c < nc; c++) rr[c] = xt->qlut[c][clampmax(rk[c], rmax)];
    goto merge;
  }
  /* Q tables (APPLY_LUT: index clamped to the table); the identity, 2^(Pr + 4) -> 2^(16 + 4), scales by 2^(16 - Pr)
   * (parametrictonemappingbox.cpp:387-430) */
  for (c = 0; c < nc; c++) {
    const int64_t idx = clampmax(rk[c], rmax16);
    q3[c] = xt->qlut[c] ? xt->qlut[c][idx] : idx << (16 - r->precision);
  }
  if (xt->rtrafo_ycbcr) {
    const int64_t *M = xt->rmat;
    /* (LONG variables, QUAD products: a table entry beyond the range -- a curve with parameters no encoder writes -- wraps
     * where the reference narrows, colortrafo/ycbcrtrafo.cpp:776-789, 868-879) */
    const int64_t ry = q3[0], rcb = W32(q3[1] - (xt->outshift << 4)), rcr = W32(q3[2] - (xt->outshift << 4));
    rr[0] = W32((ry * M[0] + rcb * M[1] + rcr * M[2] + 4096) >> 13); /* FIX_COLOR_TO_INTCOLOR */
    rr[1] = W32((ry * M[3] + rcb * M[4] + rcr * M[5] + 4096) >> 13);
    rr[2] = W32((ry * M[6] + rcb * M[7] + rcr * M[8] + 4096) >> 13);
  } else {
    rr[0] = q3[0]; rr[1] = q3[1]; rr[2] = q3[2];
  }
  /* R2 tables; the identity 2^(16 + 4) -> 2^16 is floor(x / 16 + 0.5) */
  for (c = 0; c < nc; c++) {
    const int64_t idx = clampmax(rr[c], omax16);
    rr[c] = xt->r2lut[c] ? xt->r2lut[c][idx] : (idx + 8) >> 4;
  }
merge:
  for (c = 0; c < nc; c++) lv[c] = xt->ltable[c] ? xt->ltable[c][clampmax(vin[c], maxval)] : vin[c];
  /* C transformation, FIX_TO_INT (the identity leaves the values alone: (x * 8192 + 4096) >> 13 == x) */
  for (c = 0; c < 3; c++)
    v[c] = W32(((lv[0] * xt->cmat[3 * c] + lv[1] * xt->cmat[3 * c + 1] + lv[2] * xt->cmat[3 * c + 2] + 4096) >> 13) + rr[c] - xt->outshift);
  if (xt->is_float && xt->clamp) {
    const int64_t pinf = (xt->outmax >> 1) - (xt->outmax >> 6) - 1;
    const int64_t minf = invert_negs((int16_t)(uint16_t)(pinf | 0x8000));
    for (c = 0; c < nc; c++) {
      int64_t t = v[c] > pinf ? pinf : (v[c] < minf ? minf : v[c]);
      out[c] = (uint16_t)invert_negs((int16_t)t);
    }
  } else if (xt->clamp) {
    for (c = 0; c < nc; c++) out[c] = (uint16_t)clampmax(v[c], xt->outmax);
  } else if (xt->is_float) { /* :940-955: complement -> sign-magnitude, nothing else */
    for (c = 0; c < nc; c++) out[c] = (uint16_t)invert_negs((int16_t)(uint16_t)v[c]);
  } else { /* :957-972: WRAP */
    for (c = 0; c < nc; c++) out[c] = (uint16_t)(v[c] & xt->outmax);
  }
}

/* Output: pixels8 (precision 8, no XT) or pixels16 (precision 12, or XT: 16 bit codes). */
static int reconstruct_ex(const oj_info *f, int32_t *const planes[OJ_MAX_COMP], uint8_t *pixels8, uint16_t *pixels16,
                          int use_ycbcr, const oj_xt *xt)
{
  int32_t *samp[OJ_MAX_COMP] = {0, 0, 0, 0}, *rsamp[OJ_MAX_COMP] = {0, 0, 0, 0};
  int c, X0, Y0, x, y, rc = OJ_OK;
  int ycc = use_ycbcr < 0 ? f->ycbcr : use_ycbcr;
  const int64_t L[9] = {FIX13(1.0), FIX13(0.0), FIX13(1.40200),
                        FIX13(1.0), -FIX13(0.3441362861), -FIX13(0.7141362859),
                        FIX13(1.0), FIX13(1.772), FIX13(0.0)};
  const int32_t dcshift = (int32_t)(1 << (f->precision - 1)) << 4;
  const int64_t maxval = ((int64_t)1 << f->precision) - 1;
  for (c = 0; c < f->ncomp; c++) {
    /* the component's transform carries the table that was in force at its first scan (control/blockbuffer.cpp:177-208) */
    const uint16_t *q = f->scan_state_valid ? f->cquant[c] : f->quant[f->tq[c]];
    if (!f->scan_state_valid && !f->quant_defined[f->tq[c]]) { rc = OJ_ERR_MALFORMED; goto out; }
    samp[c] = (int32_t *)malloc((size_t)f->bw[c] * f->bh[c] * 64 * sizeof(int32_t));
    if (!samp[c]) { rc = OJ_ERR_NOMEM; goto out; }
    if (f->scan_state_valid && !f->comp_seen[c]) /* no scan, no transform: samples are 0 (blockbitmaprequester.cpp:1047-1054, 1100-1104) */
      memset(samp[c], 0, (size_t)f->bw[c] * f->bh[c] * 64 * sizeof(int32_t));
    else
      oj_idct_plane(samp[c], planes[c], f->bw[c], f->bh[c], q, f->precision);
    if (f->dnl && f->rows[c] < f->bh[c]) /* rows nobody created: NULL -> sample value 0 (blockbitmaprequester.cpp:1097-1108, idct.cpp:336-338) */
      memset(samp[c] + (size_t)f->rows[c] * 8 * f->bw[c] * 8, 0, (size_t)(f->bh[c] - f->rows[c]) * f->bw[c] * 64 * sizeof(int32_t));
    if (xt && !xt->no_residual) { /* residual: same transform, level shift 2^(Pr-1) (control/residualblockhelper.cpp:191-202) */
      const oj_info *r = xt->rinfo;
      if (!r->quant_defined[r->tq[c]]) { rc = OJ_ERR_MALFORMED; goto out; }
      rsamp[c] = (int32_t *)malloc((size_t)r->bw[c] * r->bh[c] * 64 * sizeof(int32_t));
      if (!rsamp[c]) { rc = OJ_ERR_NOMEM; goto out; }
      if (xt->rbypass) {
        /* only the highest-frequency delta is used, times 2^COLOR_BITS; the level shift is NOT scaled (:196, :225) */
        const uint16_t *rq = r->scan_state_valid ? r->cquant[c] : r->quant[r->tq[c]];
        /* UWORD m_usQuantization, shifted by the fractional bits where there is more than one (residualblockhelper.cpp:351-364) */
        const int32_t quant = xt->rbits > 1 ? ((int32_t)rq[63] << xt->rbits) & 0xffff : (int32_t)rq[63], dcs = (int32_t)(1 << r->precision) >> 1;
        int bx, by, i;
        for (by = 0; by < r->bh[c]; by++)
          for (bx = 0; bx < r->bw[c]; bx++) {
            int32_t blk[64];
            xt_bypass_block(blk, xt->rplanes[c] + ((size_t)by * r->bw[c] + bx) * 64, quant, xt->rnoise, dcs);
            for (i = 0; i < 64; i++) rsamp[c][((size_t)by * 8 + (i >> 3)) * ((size_t)r->bw[c] * 8) + (size_t)bx * 8 + (i & 7)] = blk[i];
          }
      } else
        oj_idct_plane(rsamp[c], xt->rplanes[c], r->bw[c], r->bh[c], r->scan_state_valid ? r->cquant[c] : r->quant[r->tq[c]], r->precision);
    }
  }
  for (Y0 = 0; Y0 < f->height; Y0 += 8)
    for (X0 = 0; X0 < f->width; X0 += 8) {
      int32_t blk[OJ_MAX_COMP][64], rblk[OJ_MAX_COMP][64];
      for (c = 0; c < f->ncomp; c++) {
        /* DNL frames: the upsampler was built with the height unknown (upsamplerbase.cpp:61-75) and its buffer has no
         * bottom edge -- the line below the picture is whatever the next block row holds (:138-156, :218-228) */
        oj_upsample_block(blk[c], samp[c], f->bw[c] * 8, f->cw[c], f->dnl ? f->bh[c] * 8 : f->ch[c], f->subx[c], f->suby[c], X0, Y0);
        if (xt && !xt->no_residual) {
          const oj_info *r = xt->rinfo;
          oj_upsample_block(rblk[c], rsamp[c], r->bw[c] * 8, r->cw[c], r->ch[c], r->subx[c], r->suby[c], X0, Y0);
        }
      }
      for (y = 0; y < 8 && Y0 + y < f->height; y++)
        for (x = 0; x < 8 && X0 + x < f->width; x++) {
          const size_t pix = ((size_t)(Y0 + y) * f->width + (X0 + x)) * f->ncomp;
          int k = y * 8 + x;
          int64_t v[OJ_MAX_COMP];
          if (ycc && f->ncomp == 3) {
            const int64_t *M = xt ? xt->lmat : L;
            int64_t yy = blk[0][k], cb = (int64_t)blk[1][k] - dcshift, cr = (int64_t)blk[2][k] - dcshift;
            v[0] = (yy * M[0] + cb * M[1] + cr * M[2] + 65536) >> 17;
            v[1] = (yy * M[3] + cb * M[4] + cr * M[5] + 65536) >> 17;
            v[2] = (yy * M[6] + cb * M[7] + cr * M[8] + 65536) >> 17;
          } else {
            for (c = 0; c < f->ncomp; c++) v[c] = ((int64_t)blk[c][k] + 8) >> 4;
          }
          if (xt) {
            int32_t rk[3] = {0, 0, 0};
            if (!xt->no_residual)
              for (c = 0; c < f->ncomp; c++) rk[c] = rblk[c][k];
            xt_merge_pixel(xt, maxval, v, rk, pixels16 + pix);
          } else if (pixels8) {
            for (c = 0; c < f->ncomp; c++) pixels8[pix + c] = (uint8_t)clampmax(v[c], maxval);
          } else {
            for (c = 0; c < f->ncomp; c++) pixels16[pix + c] = (uint16_t)clampmax(v[c], maxval);
          }
        }
    }
out:
  for (c = 0; c < OJ_MAX_COMP; c++) { free(samp[c]); free(rsamp[c]); }
  return rc;
}

int oj_reconstruct(const oj_info *f, int32_t *const planes[OJ_MAX_COMP], uint8_t *pixels, int use_ycbcr)
{
  if (f->precision != 8) return OJ_ERR_UNSUPPORTED;
  return reconstruct_ex(f, planes, pixels, NULL, use_ycbcr, NULL);
}

int oj_reconstruct16(const oj_info *f, int32_t *const planes[OJ_MAX_COMP], uint16_t *pixels, int use_ycbcr)
{
  return reconstruct_ex(f, planes, NULL, pixels, use_ycbcr, NULL);
}


/* ------------------------------------------------------------------------------------------
 * The rectangle service as a sequence of requests: JPEG::DisplayRectangle (interface/jpeg.cpp:694-722) ->
 * Image::ReconstructRegion (codestream/image.cpp:1087-1123) -> BlockBitmapRequester::RequestUserDataForDecoding /
 * ReconstructRegion (control/blockbitmaprequester.cpp:1229-1272), restated LITERALLY with the state the reference
 * keeps between calls: one row cursor per component into its list of coefficient rows (m_pppQImage, only ever advanced:
 * nothing resets it while decoding) and the line buffer of every subsampled component's upsampler
 * (upsampling/upsamplerbase.cpp).  A cursor behind the last row reads NULL and a NULL row transforms to SAMPLE value 0
 * (dct/idct.cpp:336-338).  What follows from it -- and what whole-frame reconstruction cannot show:
 *  - on the upsampling path every call advances the cursor of EVERY component without an upsampler, requested or not
 *    (:1214-1223), so the reference's own component-by-component PGX loop (cmd/reconstruct.cpp:272-303) delivers zero
 *    planes for the second, third ... unsubsampled component of a frame that also has a subsampled one;
 *  - components outside the requested range enter the colour transformation as 0 (:1190-1193, :1047-1054);
 *  - the upsampler hands out 8x8 samples starting AT the rectangle's corner (upsampler.cpp:85-86) while the colour
 *    transformer reads its source block at (x & 7, y & 7) (colortrafo/ycbcrtrafo.cpp:683-686): rectangles that do not
 *    start on the block grid see subsampled components displaced in their first row / column of blocks;
 *  - requests that skip or repeat stripes read the rows the cursors happen to stand at;
 *  - the colour transformer is chosen by the first request and kept (colortransformerfactory.cpp:220-221).
 * JPEG XT (oj_xt_requester_new): the residual image's cursors and upsamplers run beside the legacy image's, for requests over all
 * three components with upsampling and colour transformation on; a request that reads a residual row behind the last one is
 * where the reference dereferences NULL: OJ_ERR_UNSUPPORTED at that call.
 * Not modelled: component subsets / no transformation on XT frames, alpha channels, and reads of line-buffer memory the reference
 * never initialised (rectangles narrower than the frame that move sideways between calls): OJ_ERR_UNSUPPORTED.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int sx, sy, width, total; /* m_ucSubX/Y, m_ulWidth, m_lTotalLines (upsamplerbase.cpp:61-77) */
  int pw, ph;               /* m_ulPixelWidth / Height */
  int y, h;                 /* m_lY, m_lHeight: lines [y, y + h) are buffered */
  int32_t **line;           /* line[k] = m_pData of line y + k, width + 2 + 8 LONGs */
  int cap;
} oj_up;

struct oj_requester {
  oj_info f;
  const int32_t *planes[OJ_MAX_COMP];
  int cur[OJ_MAX_COMP];   /* m_pppQImage[c]: index of the row it stands at; >= rows[c]: NULL */
  int rows[OJ_MAX_COMP];  /* rows the scans created: ceil(ch / 8) (control/blockbuffer.cpp:212-265) */
  oj_up *up[OJ_MAX_COMP]; /* m_ppUpsampler (blockbitmaprequester.cpp:298-322) */
  int subsampling;        /* m_bSubsampling */
  int trafo_built, ycc;   /* the colour transformer is built by the first request that reconstructs something and kept
                             (colortrafo/colortransformerfactory.cpp:220-221): later JPGTAG_MATRIX_LTRAFO values change nothing */
  /* JPEG XT: the residual image's side of the same state (m_pppRImage, m_ppResidualUpsampler; blockbitmaprequester.cpp:
   * 228-232, 356-372, 1118-1146, 1197-1222) and the merge the colour transformer performs (oj_xt) */
  const oj_xt *xt;
  const oj_info *rf;
  const int32_t *rplanes[OJ_MAX_COMP];
  int rcur[OJ_MAX_COMP], rrows[OJ_MAX_COMP];
  oj_up *rup[OJ_MAX_COMP];
  struct oj_xt_ctx *owner; /* what oj_xt_requester_new decoded and built; freed with the requester */
};
struct oj_xt_ctx;
static void xt_ctx_free(struct oj_xt_ctx *ctx);

static void up_free(oj_up *u)
{
  int k;
  if (!u) return;
  for (k = 0; k < u->h; k++) free(u->line[k]);
  free(u->line);
  free(u);
}

/* upsamplerbase.cpp:163-212 SetBufferedRegion + :218-259 ExtendBufferedRegion; min_y / max_y in blocks; returns the first
 * block row the caller has to define (the rectangle the reference hands back), or a negative error */
static int up_set_buffered_region(oj_up *u, int min_y, int max_y)
{
  int maxy;
  while (u->y < (min_y << 3)) {
    if (u->h > 0) {
      free(u->line[0]);
      memmove(u->line, u->line + 1, (size_t)(u->h - 1) * sizeof(*u->line));
      u->h--;
    }
    u->y++;
  }
  if (u->y > (min_y << 3)) { /* part of the buffer lies below the top line: dispose of it */
    int k;
    for (k = 0; k < u->h; k++) free(u->line[k]);
    u->h = 0;
    u->y = min_y << 3;
  }
  min_y = (u->y + u->h + 7) >> 3;
  maxy = (1 + max_y) << 3;
  if (maxy > u->total) maxy = u->total;
  while (u->y + u->h < maxy) {
    if (u->h >= u->cap) {
      int32_t **nl = (int32_t **)realloc(u->line, (size_t)(u->cap + 64) * sizeof(*u->line));
      if (!nl) return OJ_ERR_NOMEM;
      u->line = nl;
      u->cap += 64;
    }
    u->line[u->h] = (int32_t *)calloc((size_t)u->width + 2 + 8, sizeof(int32_t));
    if (!u->line[u->h]) return OJ_ERR_NOMEM;
    u->h++;
  }
  return min_y;
}

/* upsamplerbase.cpp:300-327 */
static int up_define_region(oj_up *u, int bx, int by, const int32_t *data)
{
  int k = (by << 3) - u->y, cnt = 8;
  if (k < 0 || k >= u->h) return OJ_ERR_UNSUPPORTED; /* the reference asserts */
  do {
    int32_t *dest = u->line[k] + 1;
    memcpy(dest + (bx << 3), data, 8 * sizeof(int32_t));
    dest[-1] = dest[0];
    dest[u->width] = dest[u->width - 1];
    k++;
    data += 8;
  } while (--cnt && k < u->h);
  return OJ_OK;
}

static int32_t up_line_at(const void *ctx, int row, int idx)
{
  const oj_up *u = (const oj_up *)ctx;
  return u->line[row - u->y][idx];
}

/* upsampler.cpp:83-117 UpsampleRegion for the rectangle whose corner is (min_x, min_y) */
static int up_upsample_region(const oj_up *u, int min_x, int min_y, int32_t out[64])
{
  int y = min_y / u->sy, x = min_x / u->sx + 1;
  int top, cur, bot;
  if (y < u->y || y >= u->y + u->h) return OJ_ERR_UNSUPPORTED; /* "must be in the buffer" */
  top = u->y;
  if (y - 1 > top) top = y - 1;
  cur = top;
  if (y > u->y) cur++;
  if (cur >= u->y + u->h) return OJ_ERR_UNSUPPORTED;
  bot = cur + 1 < u->y + u->h ? cur + 1 : cur;
  if (u->sx > 1) x--;
  upsample_core(out, up_line_at, u, top, cur, bot, u->y + u->h, x, u->sx, u->sy, min_x % u->sx, min_y % u->sy);
  return OJ_OK;
}

oj_requester *oj_requester_new(const oj_info *f, int32_t *const planes[OJ_MAX_COMP])
{
  oj_requester *rq = (oj_requester *)calloc(1, sizeof(*rq));
  int c;
  if (!rq) return NULL;
  rq->f = *f;
  for (c = 0; c < f->ncomp; c++) {
    rq->planes[c] = planes[c];
    rq->rows[c] = f->dnl ? f->rows[c] : (f->ch[c] + 7) >> 3;
    if (f->subx[c] > 1 || f->suby[c] > 1) { /* blockbitmaprequester.cpp:310-318 */
      oj_up *u = (oj_up *)calloc(1, sizeof(*u));
      if (!u) { oj_requester_free(rq); return NULL; }
      u->sx = f->subx[c]; u->sy = f->suby[c];
      u->pw = f->width; u->ph = f->dnl ? 0x7fffffff : f->height; /* `if (pixelheight == 0) pixelheight = ~0U >> 1`, upsamplerbase.cpp:65-67 */
      u->width = (f->width + u->sx - 1) / u->sx;
      u->total = (int)(((uint32_t)u->ph + (uint32_t)u->sy - 1) / (uint32_t)u->sy);
      rq->up[c] = u;
      rq->subsampling = 1;
    }
  }
  return rq;
}

int oj_requester_cursor(const oj_requester *rq, int c) { return c >= OJ_MAX_COMP ? rq->rcur[c - OJ_MAX_COMP] : rq->cur[c]; } /* (4 + c: the residual image's) */

void oj_requester_free(oj_requester *rq)
{
  int c;
  if (!rq) return;
  for (c = 0; c < OJ_MAX_COMP; c++) { up_free(rq->up[c]); up_free(rq->rup[c]); }
  if (rq->owner) xt_ctx_free(rq->owner);
  free(rq);
}

static const int32_t *rq_row_block(const oj_requester *rq, int c, int bx)
{
  if (rq->cur[c] >= rq->rows[c] || rq->cur[c] >= rq->f.bh[c]) return NULL; /* *m_pppQImage[c] == NULL (or a row the oracle does not keep) */
  return rq->planes[c] + ((size_t)rq->cur[c] * rq->f.bw[c] + bx) * 64;
}

static void rq_idct(const oj_requester *rq, int c, const int32_t *src, int32_t dst[64])
{
  const oj_info *f = &rq->f;
  if (f->scan_state_valid && !f->comp_seen[c]) { memset(dst, 0, 64 * sizeof(int32_t)); return; } /* m_ppDCT[c] == NULL */
  oj_idct_block(dst, src, f->scan_state_valid ? f->cquant[c] : f->quant[f->tq[c]], f->precision);
}

/* colortrafo/ycbcrtrafo.cpp:679-1009 for the rectangle r = [x0, x1] x [y0, y1] inside one block: reads the sources at
 * (x & 7, y & 7), writes through the bitmaps that are there */
static void rq_color(const oj_requester *rq, int ycc, int x0, int y0, int x1, int y1, int32_t src[OJ_MAX_COMP][64],
                     void *const dst[OJ_MAX_COMP], const int bpp[OJ_MAX_COMP], const int bpr[OJ_MAX_COMP],
                     const int bm_width[OJ_MAX_COMP], const int bm_height[OJ_MAX_COMP], int sample_bytes)
{
  const oj_info *f = &rq->f;
  const int64_t L[9] = {FIX13(1.0), FIX13(0.0), FIX13(1.40200), FIX13(1.0), -FIX13(0.3441362861), -FIX13(0.7141362859),
                        FIX13(1.0), FIX13(1.772), FIX13(0.0)};
  const int32_t dcshift = (int32_t)(1 << (f->precision - 1)) << 4;
  const int64_t maxval = ((int64_t)1 << f->precision) - 1;
  int x, y, c;
  for (y = y0; y <= y1; y++)
    for (x = x0; x <= x1; x++) {
      const int k = (y & 7) * 8 + (x & 7);
      int64_t v[OJ_MAX_COMP];
      if (ycc) {
        int64_t yy = src[0][k], cb = (int64_t)src[1][k] - dcshift, cr = (int64_t)src[2][k] - dcshift;
        v[0] = (yy * L[0] + cb * L[1] + cr * L[2] + 65536) >> 17;
        v[1] = (yy * L[3] + cb * L[4] + cr * L[5] + 65536) >> 17;
        v[2] = (yy * L[6] + cb * L[7] + cr * L[8] + 65536) >> 17;
      } else
        for (c = 0; c < f->ncomp; c++) v[c] = ((int64_t)src[c][k] + 8) >> 4;
      for (c = 0; c < f->ncomp; c++) {
        uint8_t *p;
        if (!dst[c]) continue;
        /* BitmapCtrl::ExtractBitmap -> interface/imagebitmap.cpp:58-129: a block whose corner lies outside the bitmap the
         * hook described is blank, nothing of it is written (a block that starts inside is written in full) */
        if ((uint32_t)bm_width[c] <= (uint32_t)x0 || (uint32_t)bm_height[c] <= (uint32_t)y0) continue;
        p = (uint8_t *)dst[c] + (ptrdiff_t)y * bpr[c] + (ptrdiff_t)x * bpp[c];
        if (sample_bytes == 2) { uint16_t w = (uint16_t)clampmax(v[c], maxval); memcpy(p, &w, 2); }
        else *p = (uint8_t)clampmax(v[c], maxval);
      }
    }
}

/* ResidualBlockHelper::DequantizeResidual (control/residualblockhelper.cpp:150-231) of one residual block: the residual
 * frame's own transform with its level shift 2^(Pr-1), or -- RDCT box -- the bypass.  src may be NULL (a row nobody created:
 * the transform's NULL branch, dct/idct.cpp:336-338). */
static void rq_residual_block(const oj_requester *rq, int c, const int32_t *src, int32_t dst[64])
{
  const oj_info *r = rq->rf;
  const uint16_t *q = r->scan_state_valid ? r->cquant[c] : r->quant[r->tq[c]];
  if (rq->xt->rbypass && src) {
    const int32_t quant = ((int32_t)q[63] << 4) & 0xffff, dcs = (int32_t)(1 << r->precision) >> 1;
    xt_bypass_block(dst, src, quant, rq->xt->rnoise, dcs);
  } else
    oj_idct_block(dst, src, q, r->precision);
}
static const int32_t *rq_rrow_block(const oj_requester *rq, int c, int bx)
{
  if (rq->rcur[c] >= rq->rrows[c] || rq->rcur[c] >= rq->rf->bh[c]) return NULL;
  return rq->rplanes[c] + ((size_t)rq->rcur[c] * rq->rf->bw[c] + bx) * 64;
}

/* colortrafo/ycbcrtrafo.cpp:679-955 with a residual, for the rectangle [x0, x1] x [y0, y1] inside one block: L transformation,
 * then the merge of xt_merge_pixel; 16-bit codes (or 8-bit samples when the output has no extra range bits) */
static void rq_color_xt(const oj_requester *rq, int x0, int y0, int x1, int y1, int32_t src[OJ_MAX_COMP][64], int32_t rsrc[OJ_MAX_COMP][64],
                        void *const dst[OJ_MAX_COMP], const int bpp[OJ_MAX_COMP], const int bpr[OJ_MAX_COMP],
                        const int bm_width[OJ_MAX_COMP], const int bm_height[OJ_MAX_COMP], int sample_bytes)
{
  const oj_info *f = &rq->f;
  const oj_xt *xt = rq->xt;
  const int32_t dcshift = (int32_t)(1 << (f->precision - 1)) << 4;
  const int64_t maxval = ((int64_t)1 << f->precision) - 1;
  int x, y, c;
  for (y = y0; y <= y1; y++)
    for (x = x0; x <= x1; x++) {
      const int k = (y & 7) * 8 + (x & 7);
      int64_t v[3];
      int32_t rk[3];
      uint16_t o[3];
      if (xt->ltrafo_ycbcr) {
        const int64_t *M = xt->lmat;
        const int64_t yy = src[0][k], cb = (int64_t)src[1][k] - dcshift, cr = (int64_t)src[2][k] - dcshift;
        v[0] = (yy * M[0] + cb * M[1] + cr * M[2] + 65536) >> 17;
        v[1] = (yy * M[3] + cb * M[4] + cr * M[5] + 65536) >> 17;
        v[2] = (yy * M[6] + cb * M[7] + cr * M[8] + 65536) >> 17;
      } else
        for (c = 0; c < 3; c++) v[c] = ((int64_t)src[c][k] + 8) >> 4;
      for (c = 0; c < 3; c++) rk[c] = rsrc[c][k];
      xt_merge_pixel(xt, maxval, v, rk, o);
      for (c = 0; c < 3; c++) {
        uint8_t *p;
        if (!dst[c]) continue;
        if ((uint32_t)bm_width[c] <= (uint32_t)x0 || (uint32_t)bm_height[c] <= (uint32_t)y0) continue;
        p = (uint8_t *)dst[c] + (ptrdiff_t)y * bpr[c] + (ptrdiff_t)x * bpp[c];
        if (sample_bytes == 2) memcpy(p, &o[c], 2);
        else *p = (uint8_t)o[c];
      }
    }
}

int oj_requester_display(oj_requester *rq, int min_x, int min_y, int max_x, int max_y, int c0, int c1, int upsample, int ctrafo,
                         void *const dst[OJ_MAX_COMP], const int bpp[OJ_MAX_COMP], const int bpr[OJ_MAX_COMP],
                         const int bm_width[OJ_MAX_COMP], const int bm_height[OJ_MAX_COMP], int sample_bytes)
{
  const oj_info *f = &rq->f;
  void *bm[OJ_MAX_COMP] = {0, 0, 0, 0};
  uint32_t maxmcu = 0xffffffffu;
  int c, ycc, rc;
  if (rq->xt) {
    /* JPEG XT: what is restated is what the reference's command line asks for -- all three components, upsampling and colour
     * transformation on -- in any order and size of rectangles.  (A component subset merges with what m_ppDTemp still holds from
     * the block before; without the transformation another transformer is built: neither is restated.) */
    if (sample_bytes != (rq->xt->outmax > 255 ? 2 : 1)) return OJ_ERR_UNSUPPORTED;
    if (!upsample || !ctrafo || c0 > 0 || c1 < 2) return OJ_ERR_UNSUPPORTED;
  } else if (sample_bytes != (f->precision > 8 ? 2 : 1)) return OJ_ERR_UNSUPPORTED;
  /* codestream/rectanglerequest.cpp:62-190: clipped to the canvas; without upsampling no colour transformation */
  if (min_x < 0) min_x = 0;
  if (min_y < 0) min_y = 0;
  if (max_x > f->width - 1) max_x = f->width - 1;
  if (max_y > f->height - 1) max_y = f->height - 1;
  if (c0 < 0) c0 = 0;
  if (c1 > f->ncomp - 1) c1 = f->ncomp - 1;
  if (!upsample) ctrafo = 0;
  /* blockbitmaprequester.cpp:1229-1244: only the requested components have bitmaps; their heights bound the block rows
   * (ULONG arithmetic: a height below 8 wraps to "no bound") */
  for (c = c0; c <= c1; c++) {
    const uint32_t m = ((uint32_t)bm_height[c] >> 3) - 1u;
    bm[c] = dst[c];
    if (m < maxmcu) maxmcu = m;
  }
  if (min_x > max_x || min_y > max_y) return OJ_OK; /* codestream/image.cpp:1115: empty regions are not reconstructed */
  if (c0 > c1) return OJ_OK;
  /* BlockBitmapRequester::ColorTrafoOf -> colortransformerfactory.cpp: YCbCr for three components unless switched off */
  if (!rq->trafo_built) {
    rq->trafo_built = 1;
    rq->ycc = ctrafo && f->ycbcr && f->ncomp == 3;
  }
  ycc = rq->ycc;
  if (rq->subsampling && upsample) {
    uint32_t minx = (uint32_t)min_x >> 3, maxx = (uint32_t)max_x >> 3, miny = (uint32_t)min_y >> 3, maxy = (uint32_t)max_y >> 3, bx, by;
    int r_min_y, r_max_y;
    /* PullQData, :1079-1112 */
    for (c = c0; c <= c1; c++) {
      oj_up *u = rq->up[c];
      int bwidth, bheight, rx, ry, b_min_x, b_max_x, b_min_y, b_max_y, yy, xx;
      if (!u) continue;
      /* upsamplerbase.cpp:138-156 SetBufferedImageRegion */
      bwidth = ((u->pw + u->sx - 1) / u->sx + 7) >> 3;
      bheight = (int)((((uint32_t)u->ph + (uint32_t)u->sy - 1) / (uint32_t)u->sy + 7) >> 3);
      rx = u->sx > 1; ry = u->sy > 1;
      b_min_x = (min_x / u->sx - rx) >> 3;
      b_max_x = (max_x / u->sx + rx) >> 3;
      b_min_y = (min_y / u->sy - ry) >> 3;
      b_max_y = (max_y / u->sy + ry) >> 3;
      if (b_min_x < 0) b_min_x = 0;
      if (b_max_x >= bwidth) b_max_x = bwidth - 1;
      if (b_min_y < 0) b_min_y = 0;
      if (b_max_y >= bheight) b_max_y = bheight - 1;
      b_min_y = up_set_buffered_region(u, b_min_y, b_max_y);
      if (b_min_y < 0) return b_min_y;
      for (yy = b_min_y; yy <= b_max_y; yy++) {
        for (xx = b_min_x; xx <= b_max_x; xx++) {
          int32_t blk[64];
          rq_idct(rq, c, rq_row_block(rq, c, xx), blk);
          if ((rc = up_define_region(u, xx, yy, blk)) != OJ_OK) return rc;
        }
        if (rq->cur[c] < rq->rows[c]) rq->cur[c]++;
      }
    }
    /* PullRData, :1118-1146: the same for the residual image's upsamplers */
    for (c = c0; rq->xt && c <= c1; c++) {
      oj_up *u = rq->rup[c];
      int bwidth, bheight, rx, ry, b_min_x, b_max_x, b_min_y, b_max_y, yy, xx;
      if (!u) continue;
      bwidth = ((u->pw + u->sx - 1) / u->sx + 7) >> 3;
      bheight = (int)((((uint32_t)u->ph + (uint32_t)u->sy - 1) / (uint32_t)u->sy + 7) >> 3);
      rx = u->sx > 1; ry = u->sy > 1;
      b_min_x = (min_x / u->sx - rx) >> 3;
      b_max_x = (max_x / u->sx + rx) >> 3;
      b_min_y = (min_y / u->sy - ry) >> 3;
      b_max_y = (max_y / u->sy + ry) >> 3;
      if (b_min_x < 0) b_min_x = 0;
      if (b_max_x >= bwidth) b_max_x = bwidth - 1;
      if (b_min_y < 0) b_min_y = 0;
      if (b_max_y >= bheight) b_max_y = bheight - 1;
      b_min_y = up_set_buffered_region(u, b_min_y, b_max_y);
      if (b_min_y < 0) return b_min_y;
      for (yy = b_min_y; yy <= b_max_y; yy++) {
        for (xx = b_min_x; xx <= b_max_x; xx++) {
          int32_t blk[64];
          rq_residual_block(rq, c, rq_rrow_block(rq, c, xx), blk);
          if ((rc = up_define_region(u, xx, yy, blk)) != OJ_OK) return rc;
        }
        if (rq->rcur[c] < rq->rrows[c]) rq->rcur[c]++;
      }
    }
    /* PushReconstructedData, :1151-1224 */
    if (maxy > maxmcu) maxy = maxmcu;
    for (by = miny, r_min_y = min_y; by <= maxy; by++, r_min_y = r_max_y + 1) {
      int r_min_x, r_max_x;
      r_max_y = (r_min_y & -8) + 7;
      if (r_max_y > max_y) r_max_y = max_y;
      for (bx = minx, r_min_x = min_x; bx <= maxx; bx++, r_min_x = r_max_x + 1) {
        int32_t src[OJ_MAX_COMP][64];
        r_max_x = (r_min_x & -8) + 7;
        if (r_max_x > max_x) r_max_x = max_x;
        for (c = 0; c < f->ncomp; c++) {
          if (c >= c0 && c <= c1) {
            if (rq->up[c]) {
              if ((rc = up_upsample_region(rq->up[c], r_min_x, r_min_y, src[c])) != OJ_OK) return rc;
            } else
              rq_idct(rq, c, rq_row_block(rq, c, (int)bx), src[c]);
          } else
            memset(src[c], 0, sizeof(src[c]));
        }
        if (rq->xt) {
          int32_t rsrc[OJ_MAX_COMP][64];
          memset(rsrc, 0, sizeof(rsrc));
          for (c = 0; c < 3 && !rq->xt->no_residual; c++) {
            if (rq->rup[c]) {
              if ((rc = up_upsample_region(rq->rup[c], r_min_x, r_min_y, rsrc[c])) != OJ_OK) return rc;
            } else {
              const int32_t *rb = rq_rrow_block(rq, c, (int)bx);
              if (!rb) return OJ_ERR_UNSUPPORTED; /* `rrow->BlockAt(x)` on a NULL row (:1201-1202): the reference crashes here */
              rq_residual_block(rq, c, rb, rsrc[c]);
            }
          }
          rq_color_xt(rq, r_min_x, r_min_y, r_max_x, r_max_y, src, rsrc, bm, bpp, bpr, bm_width, bm_height, sample_bytes);
        } else
          rq_color(rq, ycc, r_min_x, r_min_y, r_max_x, r_max_y, src, bm, bpp, bpr, bm_width, bm_height, sample_bytes);
      }
      for (c = 0; c < f->ncomp; c++) { /* every component without an upsampler, requested or not (:1214-1222) */
        if (!rq->up[c] && rq->cur[c] < rq->rows[c]) rq->cur[c]++;
        if (rq->xt && !rq->rup[c] && rq->rcur[c] < rq->rrows[c]) rq->rcur[c]++;
      }
    }
  } else {
    /* ReconstructUnsampled, :1013-1074, on the region of control/bitmapctrl.cpp:273-294 */
    uint32_t minx, maxx, miny, maxy, bx, by;
    int r_min_y, r_max_y;
    if (!upsample) {
      int sx, sy;
      if (c0 != c1) return OJ_ERR_MALFORMED; /* JPGERR_INVALID_PARAMETER in the reference */
      sx = f->subx[c0]; sy = f->suby[c0];
      min_x = (min_x + sx - 1) / sx;
      max_x = (max_x + sx) / sx - 1;
      min_y = (min_y + sy - 1) / sy;
      max_y = (max_y + sy) / sy - 1;
    }
    minx = (uint32_t)min_x >> 3; maxx = (uint32_t)max_x >> 3; miny = (uint32_t)min_y >> 3; maxy = (uint32_t)max_y >> 3;
    if (maxy > maxmcu) maxy = maxmcu;
    for (by = miny, r_min_y = min_y; by <= maxy; by++, r_min_y = r_max_y + 1) {
      int r_min_x, r_max_x;
      r_max_y = (r_min_y & -8) + 7;
      if (r_max_y > max_y) r_max_y = max_y;
      for (bx = minx, r_min_x = min_x; bx <= maxx; bx++, r_min_x = r_max_x + 1) {
        int32_t src[OJ_MAX_COMP][64];
        r_max_x = (r_min_x & -8) + 7;
        if (r_max_x > max_x) r_max_x = max_x;
        for (c = 0; c < f->ncomp; c++) {
          if (c >= c0 && c <= c1) rq_idct(rq, c, rq_row_block(rq, c, (int)bx), src[c]);
          else memset(src[c], 0, sizeof(src[c]));
        }
        if (rq->xt) {
          int32_t rsrc[OJ_MAX_COMP][64];
          memset(rsrc, 0, sizeof(rsrc));
          for (c = c0; c <= c1 && !rq->xt->no_residual; c++) {
            const int32_t *rb = rq_rrow_block(rq, c, (int)bx);
            if (!rb) return OJ_ERR_UNSUPPORTED; /* `rrow->BlockAt(x)` on a NULL row (:1057-1058) */
            rq_residual_block(rq, c, rb, rsrc[c]);
          }
          rq_color_xt(rq, r_min_x, r_min_y, r_max_x, r_max_y, src, rsrc, bm, bpp, bpr, bm_width, bm_height, sample_bytes);
        } else
          rq_color(rq, ycc, r_min_x, r_min_y, r_max_x, r_max_y, src, bm, bpp, bpr, bm_width, bm_height, sample_bytes);
      }
      for (c = c0; c <= c1; c++) { /* only the requested ones (:1066-1071) */
        if (rq->cur[c] < rq->rows[c]) rq->cur[c]++;
        if (rq->xt && rq->rcur[c] < rq->rrows[c]) rq->rcur[c]++;
      }
    }
  }
  return OJ_OK;
}

/* ------------------------------------------------------------------------------------------
 * JPEG XT (ISO/IEC 18477) profile C, the subset the reference's encoder writes for
 * `-r -h -profile c -r12`: legacy 8-bit codestream + APP11 boxes SPEC{LTRF,RTRF,LPTS,OCON}, TONE (explicit
 * inverse tone mapping table) and RESI (a second, 12-bit Huffman sequential codestream).
 * Box plumbing: codestream/tables.cpp:1172-1277, boxes/box.cpp:88-200, boxes/mergingspecbox.cpp.
 * ---------------------------------------------------------------------------------------- */
#define BOXID(a, b, c, d) (((uint32_t)(a) << 24) | ((uint32_t)(b) << 16) | ((uint32_t)(c) << 8) | (uint32_t)(d))

static void free_boxes(oj_box *boxes, int n) { int i; for (i = 0; i < n; i++) free(boxes[i].data); }

/* Hidden refinement scans: box number `en` = 0, 1, ... of `type` (FINE for the legacy frame, RFIN for the residual
 * one) holds tables and one scan; they are read in this order once the visible scans are through
 * (marker/frame.cpp:805-818, 1063-1071). */
static int decode_hidden_scans(oj_parser *ps, const oj_box *boxes, int nboxes, uint32_t type, int32_t *const planes[OJ_MAX_COMP])
{
  volatile int en;
  ps->planes = planes;
  for (en = 0;; en++) {
    const oj_box *bx = NULL;
    oj_bs bio;
    int b;
    for (b = 0; b < nboxes; b++) if (boxes[b].type == type && boxes[b].en == en && boxes[b].complete) bx = &boxes[b]; /* Tables::RefinementDataOf, tables.cpp:872-891 */
    if (!bx) return OJ_OK;
    /* Frame::StartParseScan, marker/frame.cpp:805-822: tables, then the scan header, from the box's own stream */
    bs_open(&bio, bx->data, bx->len);
    bio.in_memory = 1;
    if (setjmp(ps->jb)) return rs_result(ps, 1);
    while (rs_tables_incremental(ps, &bio)) {}
    if (rs_scan_for_scan_header(ps, &bio)) rs_scan(ps, &bio, 1);
  }
}


/* Non-linear point transformations a merging specification can name (boxes/namespace.cpp:60-91): explicit tables (TONE,
 * boxes/inversetonemappingbox.cpp) and parametric curves (CURV, boxes/parametrictonemappingbox.cpp). */
typedef struct {
  int kind;          /* 0: none, 1: TONE, 2: CURV */
  int entries, resbits;
  int32_t *lut;      /* TONE */
  int type, e;       /* CURV */
  float p[4];
} oj_nlt;

/* IEEEDecode, tools/numerics.cpp:56-89 */
static float ieee_decode(uint32_t bits)
{
  float f;
  if (((bits >> 23) & 0xff) == 0xff) return (bits >> 31) ? -HUGE_VALF : HUGE_VALF;
  memcpy(&f, &bits, 4);
  return f;
}

/* ParametricToneMappingBox::TableValue, boxes/parametrictonemappingbox.cpp:199-272.  *err: the curve is invalid (INVALID_PARAMETER).
 * The reference is built for the x87 unit with -ffast-math (Makefile_Settings.gcc:13): sums and products of an expression stay in
 * 80-bit registers, only the arguments and results of pow / exp / log pass through 64 bits.  long double arithmetic around
 * double library calls reproduces that (pinned by tests/golden/xt_craft_* against the reference binary). */
static long double curve_value(const oj_nlt *t, double v, int *err)
{
  const long double p1 = t->p[0], p2 = t->p[1], p3 = t->p[2], p4 = t->p[3];
  switch (t->type) {
  case 0: return 0.0L;
  case 1: return 1.0L;
  case 2: return v;
  case 4: return v >= p1 ? (long double)pow((double)((v + p3) / (1.0L + p3)), (double)p2)
                         : (long double)pow((double)((p1 + p3) / (1.0L + p3)), (double)p2) * v / p1;
  case 5: if (t->p[1] >= t->p[0]) return v * (p2 - p1) + p1; *err = 1; return 0.0L;
  case 6: if (t->p[1] > t->p[0]) return p3 * (long double)exp((double)(v * (p2 - p1) + p1)) + p4; *err = 1; return 0.0L;
  case 7:
    if (p1 > 0.0L) return (v > 0.0 || (p3 > 0.0L && v >= 0.0)) ? (long double)log((double)((long double)pow((double)(p1 * v), (double)p2) + p3)) + p4 : -HUGE_VALL;
    return (v > 0.0 || (p3 > 0.0L && v >= 0.0)) ? -(long double)log((double)((long double)pow((double)(-p1 * v), (double)p2) + p3)) + p4 : HUGE_VALL;
  case 8: return v > 0.0 ? (p2 - p1) * (long double)pow(v, (double)p3) + p1 : p1;
  }
  return 0.0L;
}

/* ToneMapperBox::ScaledTableOf for an integer path: -> malloc'ed table of 2^(inbits + infract) entries, *rc = reference error.
 * TONE: inversetonemappingbox.cpp:192-212 (the table itself, if it fits); CURV: parametrictonemappingbox.cpp:387-430. */
static int32_t *scaled_table(const oj_nlt *t, int inbits, int outbits, int infract, int outfract, int *rc)
{
  int32_t *tab;
  uint32_t i, max;
  *rc = 0;
  if (t->kind == 1) {
    if (outbits + outfract != 8 + t->resbits || inbits > 16 || (1u << inbits) != (uint32_t)t->entries || infract != 0) { *rc = -1024; return NULL; }
    tab = (int32_t *)malloc((size_t)t->entries * sizeof(int32_t));
    if (tab) memcpy(tab, t->lut, (size_t)t->entries * sizeof(int32_t));
    return tab;
  }
  max = 1u << (inbits + infract);
  tab = (int32_t *)malloc((size_t)max * sizeof(int32_t));
  if (!tab) return NULL;
  {
    const double inscale = inbits > 1 ? 1.0 / (double)((((uint32_t)1 << inbits) - (uint32_t)t->e) << infract) : 1.0 / (double)(1 << infract);
    const double outscale = outbits > 1 ? 1.0 * (double)((((uint32_t)1 << outbits) - (uint32_t)t->e) << outfract) : 1.0 * (double)(1 << outfract);
    int err = 0;
    for (i = 0; i < max; i++) {
      const double w = floor((double)((long double)outscale * curve_value(t, i * inscale, &err) + 0.5L));
      /* LONG(double): out-of-range conversions yield the x86 "integer indefinite" value */
      tab[i] = (w >= -2147483648.0 && w < 2147483648.0) ? (int32_t)w : INT32_MIN;
    }
    if (err) { free(tab); *rc = -1024; return NULL; }
  }
  return tab;
}

/* One TONE / CURV / MTRX box into the registries (first definition of an index wins: the lists are searched front to back). */
static int register_box(uint32_t type, const uint8_t *d, size_t len, oj_nlt *nlt, int64_t (*mtx)[9], int *have_mtx)
{
  if (type == BOXID('T', 'O', 'N', 'E')) {
    /* boxes/inversetonemappingbox.cpp:72-118: index/residual-bits byte, then 2^n entries of 16 (or 32) bits */
    int idx, n, i, wide;
    if (len < 1 + 512 || !(len & 1)) return OJ_ERR_MALFORMED;
    idx = d[0] >> 4; wide = (d[0] & 15) > 8;
    n = (int)((len - 1) >> (wide ? 2 : 1));
    if (wide && ((len - 1) & 3)) return OJ_ERR_MALFORMED;
    if (n & (n - 1)) return OJ_ERR_MALFORMED;
    if (nlt[idx].kind) return OJ_OK;
    nlt[idx].kind = 1; nlt[idx].entries = n; nlt[idx].resbits = d[0] & 15;
    nlt[idx].lut = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    if (!nlt[idx].lut) return OJ_ERR_NOMEM;
    for (i = 0; i < n; i++) nlt[idx].lut[i] = wide ? (int32_t)(((uint32_t)rd16(d + 1 + 4 * i) << 16) | rd16(d + 3 + 4 * i)) : rd16(d + 1 + 2 * i);
  } else if (type == BOXID('C', 'U', 'R', 'V')) {
    /* boxes/parametrictonemappingbox.cpp:84-149 */
    int idx, ty, i;
    if (len != 2 + 16) return OJ_ERR_MALFORMED;
    idx = d[0] >> 4; ty = d[0] & 15;
    if (ty == 3 || ty > 8) return OJ_ERR_MALFORMED;
    if ((d[1] & 15) || (d[1] >> 4) > 1) return OJ_ERR_MALFORMED;
    if (nlt[idx].kind) return OJ_OK;
    nlt[idx].kind = 2; nlt[idx].type = ty; nlt[idx].e = d[1] >> 4;
    for (i = 0; i < 4; i++) nlt[idx].p[i] = ieee_decode(((uint32_t)rd16(d + 2 + 4 * i) << 16) | rd16(d + 4 + 4 * i));
  } else if (type == BOXID('M', 'T', 'R', 'X')) {
    /* boxes/lineartransformationbox.cpp:62-99: id (5..15) and fractional bits (13), nine 16-bit entries */
    int id, i;
    if (len != 1 + 18) return OJ_ERR_MALFORMED;
    id = d[0] >> 4;
    if (id < 5 || (d[0] & 15) != 13) return OJ_ERR_MALFORMED;
    if (have_mtx[id]) return OJ_OK;
    have_mtx[id] = 1;
    for (i = 0; i < 9; i++) mtx[id][i] = (int16_t)rd16(d + 1 + 2 * i);
  }
  return OJ_OK;
}

/* what a requester on a JPEG XT stream owns: both frames' coefficient planes, the tables of the merge, the residual frame's
 * description */
struct oj_xt_ctx {
  oj_xt xt;
  oj_info rinfo;
  int32_t *planes[OJ_MAX_COMP], *rplanes[OJ_MAX_COMP];
  int32_t *owned[9];
};
static void xt_ctx_free(struct oj_xt_ctx *ctx)
{
  int c;
  if (!ctx) return;
  for (c = 0; c < OJ_MAX_COMP; c++) { free(ctx->planes[c]); free(ctx->rplanes[c]); }
  for (c = 0; c < 9; c++) free(ctx->owned[c]);
  free(ctx);
}

/* The reference's command line reads the whole file before it asks for pixels (cmd/reconstruct.cpp:119-121): whatever stops the
 * legacy codestream, its hidden scans, the residual codestream's header (looked at behind the legacy EOI, image.cpp:1416-1431)
 * or the residual scans is reported before anything the colour transformer finds when the first request builds it
 * (Tables::ColorTrafoOf, codestream/tables.cpp:1517-1555).  -> 0, or the error with the reference's code in *ref_error;
 * *eoi_image: the legacy codestream came to its EOI, i.e. there is a residual frame for the transformer to merge. */
static int xt_codestreams_verdict(const uint8_t *data, size_t len, const oj_info *info, const oj_box *boxes, int nboxes,
                                  const oj_box *resi, int hidden_l, int hidden_r, int *ref_error, int *eoi_image, int in_memory)
{
  oj_parser ls, rs;
  oj_info ltmp, rtmp, rinfo;
  int32_t *planes[OJ_MAX_COMP] = {0, 0, 0, 0}, *rplanes[OJ_MAX_COMP] = {0, 0, 0, 0};
  int c, rc;
  memset(&ls, 0, sizeof(ls)); memset(&rs, 0, sizeof(rs)); memset(&ltmp, 0, sizeof(ltmp)); memset(&rtmp, 0, sizeof(rtmp));
  *ref_error = 0; *eoi_image = 0;
  ls.data = data; ls.len = len; ls.info = &ltmp; ls.hidden = hidden_l; ls.xt_legacy = 1; ls.in_memory = in_memory;
  ls.known_height = info->dnl ? info->height : 0; /* (a DNL legacy frame: the header pass's height and rows, as in xt_decode_common) */
  for (c = 0; c < OJ_MAX_COMP; c++) ls.known_bh[c] = info->bh[c];
  for (c = 0; c < info->ncomp; c++) {
    planes[c] = (int32_t *)calloc((size_t)info->bw[c] * info->bh[c] * 64, sizeof(int32_t));
    if (!planes[c]) { rc = OJ_ERR_NOMEM; goto done; }
  }
  rc = walk(&ls, planes);
  if (rc) { *ref_error = ltmp.ref_error; goto done; }
  if (ls.eoi_frame) { rc = decode_hidden_scans(&ls, boxes, nboxes, BOXID('F', 'I', 'N', 'E'), planes); if (rc) { *ref_error = ls.err; goto done; } }
  /* (the ALPHA image's residual codestream is read whichever way its own codestream ended: the outer image's trailer turns to it
   * when Image::ParseAlphaChannel has no more scans to give, codestream/image.cpp:1440-1462 -- EOI, end of the box or anything else) */
  if (in_memory) ls.eoi_image = 1;
  *eoi_image = ls.eoi_image;
  if (!ls.eoi_image || !resi) goto done;
  rc = read_residual_info(resi->data, resi->len, &rinfo);
  if (!rc && (rinfo.dnl || rinfo.width != info->width || rinfo.height != info->height || rinfo.ncomp != info->ncomp)) { rinfo.ref_error = RS_MALFORMED_STREAM; rc = OJ_ERR_MALFORMED; }
  if (rc) { *ref_error = rinfo.ref_error; goto done; }
  rs.data = resi->data; rs.len = resi->len; rs.info = &rtmp; rs.hidden = hidden_r; rs.nested = 1;
  for (c = 0; c < rinfo.ncomp; c++) {
    rplanes[c] = (int32_t *)calloc((size_t)rinfo.bw[c] * rinfo.bh[c] * 64, sizeof(int32_t));
    if (!rplanes[c]) { rc = OJ_ERR_NOMEM; goto done; }
  }
  rc = walk(&rs, rplanes);
  if (rc) { *ref_error = rtmp.ref_error; goto done; }
  if (rs.eoi_frame) { rc = decode_hidden_scans(&rs, boxes, nboxes, BOXID('R', 'F', 'I', 'N'), rplanes); if (rc) *ref_error = rs.err; }
done:
  for (c = 0; c < OJ_MAX_COMP; c++) { free(planes[c]); free(rplanes[c]); }
  return rc;
}

/* pixels != NULL: the whole picture (oj_decode_xt); rq_out != NULL: a requester that keeps what was decoded (oj_xt_requester_new) */
/* disable_to_rgb: a request without colour transformation (cmd/reconstruct.cpp -c -> rr_bColorTrafo false ->
 * ColorTransformerFactory::BuildColorTransformer(.., disabletorgb)): the standard YCbCr L transformation becomes the identity,
 * nothing else changes (colortrafo/colortransformerfactory.cpp:231-232) */
/* flags: XT_DISABLE_TO_RGB; XT_IGNORE_RESIDUAL (internal): the legacy codestream has no EOI, the reference never gets to the residual
 * codestream (codestream/image.cpp:1416-1431) -- whatever is wrong with it or with the tables of its side -- and shows the legacy
 * picture through the L chain alone */
#define XT_DISABLE_TO_RGB 1
#define XT_IGNORE_RESIDUAL 2
/* given / ngiven: the codestream is an alpha channel's (the payload of the ALFA box): its boxes are these -- the file's, the
 * alpha kinds under the names of their image counterparts (oj_decode_alpha) -- instead of what the walk over it collects */
static int32_t **g_rplanes_out; /* oj_decode_xt_planes2: where the residual planes go */
static oj_info *g_rinfo_out;
static int xt_decode_common(const uint8_t *data, size_t len, oj_info *info, uint16_t **pixels, int *is_float, oj_requester **rq_out, int flags,
                            int32_t **lplanes_out, const oj_box *given, int ngiven)
{
  const int disable_to_rgb = flags & XT_DISABLE_TO_RGB;
  int retry_lonly = 0;
  int header_error_rc = 0, header_error = 0; /* what the header walk over all scans met behind the first scan header (see below) */
  oj_box boxes[OJ_MAX_BOXES];
  oj_parser ps;
  oj_info rinfo;
  oj_xt xt;
  int32_t *planes[OJ_MAX_COMP] = {0, 0, 0, 0}, *rplanes[OJ_MAX_COMP] = {0, 0, 0, 0};
  oj_nlt nlt[16];
  int64_t mtx[16][9];
  int have_mtx[16] = {0};
  int32_t *owned[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* tables built here: L, Q, R2 per component */
  int hidden_l = 0, hidden_r = 0; /* RSPC: bits of the legacy / residual coefficients in hidden refinement scans */
  const oj_box *spec = NULL, *resi = NULL;
  int ltrafo = 255, rtrafo = 255, ctrafo = 255, lidx[4] = {255, 255, 255, 255}, qidx[4] = {255, 255, 255, 255}, r2idx[4] = {255, 255, 255, 255};
  int ocon = -1, rdct = 0, b, c, rc, nc = 3, late_residual_only = 0, lonly = 0;
  size_t j;
  static const int64_t std_ycc[9] = {FIX13(1.0), FIX13(0.0), FIX13(1.40200), FIX13(1.0), -FIX13(0.3441362861), -FIX13(0.7141362859),
                                     FIX13(1.0), FIX13(1.772), FIX13(0.0)};
  static const int64_t std_id[9] = {8192, 0, 0, 0, 8192, 0, 0, 0, 8192};
  if (pixels) *pixels = NULL;
  if (rq_out) *rq_out = NULL;
  memset(&ps, 0, sizeof(ps)); memset(info, 0, sizeof(*info)); memset(nlt, 0, sizeof(nlt)); memset(&xt, 0, sizeof(xt)); memset(&rinfo, 0, sizeof(rinfo));
  ps.data = data; ps.len = len; ps.info = info; ps.boxes = boxes; ps.walk_all = 1; ps.in_memory = given != NULL; ps.xt_legacy = given != NULL;
  rc = walk(&ps, NULL);
  if (rc) {
    /* The header walk over ALL scans (it is after the boxes) met an error.  Behind the first scan header that need not be the
     * reference's verdict: it reads the stream in order, and the entropy coded data of an earlier scan may stop it first (a DHT
     * value changed into a symbol that runs into the next marker, -1025, in front of a scan header whose length is wrong, -1038:
     * tools/box_campaign.py r5).  Walk again up to the first scan header: an error there stands; otherwise go on with the boxes
     * in front of it (where every encoder puts them) and let the decoding walks below meet the errors in stream order. */
    const int first_rc = rc, first_err = info->ref_error;
    free_boxes(boxes, ps.nboxes);
    memset(&ps, 0, sizeof(ps)); memset(info, 0, sizeof(*info));
    ps.data = data; ps.len = len; ps.info = info; ps.boxes = boxes; ps.in_memory = given != NULL; ps.xt_legacy = given != NULL;
    rc = walk(&ps, NULL);
    if (rc) { free_boxes(boxes, ps.nboxes); info->ref_error = first_err; return first_rc; }
    header_error_rc = first_rc; header_error = first_err;
  }
  if (given) {
    free_boxes(boxes, ps.nboxes);
    memset(boxes, 0, sizeof(boxes));
    for (b = 0; b < ngiven && b < OJ_MAX_BOXES; b++) {
      boxes[b] = given[b];
      boxes[b].data = (uint8_t *)malloc(given[b].len ? given[b].len : 1);
      if (!boxes[b].data) { free_boxes(boxes, b); return OJ_ERR_NOMEM; }
      memcpy(boxes[b].data, given[b].data, given[b].len);
      boxes[b].cap = given[b].len;
    }
    ps.nboxes = b;
  }
  for (b = 0; b < ps.nboxes; b++) {
    if (!boxes[b].complete) continue; /* (tables.cpp:1191-1225: the tables learn of a box when its last byte was announced) */
    if (boxes[b].type == BOXID('S', 'P', 'E', 'C')) spec = &boxes[b];
    if (boxes[b].type == BOXID('R', 'E', 'S', 'I')) resi = &boxes[b];
  }
  if ((flags & XT_IGNORE_RESIDUAL) && spec) resi = NULL;
  if (resi && !spec && info->ncomp <= 4) {
    /* A residual codestream and no merging specification (its APP11 segment damaged, or its box never complete): the command
     * line reads the whole file first (cmd/reconstruct.cpp:119-121) -- whatever stops either codestream is reported -- and the
     * first request for pixels builds the colour transformer: no specification, so the R transformation is "zero"
     * (tables.cpp:2070-2071), and with a residual frame no transformer exists for that (colortransformerfactory.cpp:277-291):
     * INVALID_PARAMETER "The combination of L and R transformation is non-standard and not supported".  Without an EOI behind
     * the legacy codestream the residual frame never comes to be (image.cpp:1416-1431): a plain picture, not this function's. */
    int verr = 0, eoi = 0;
    rc = xt_codestreams_verdict(data, len, info, boxes, ps.nboxes, resi, 0, 0, &verr, &eoi, given != NULL);
    if (rc) { info->ref_error = verr; goto out; }
    if (!eoi) { rc = OJ_ERR_UNSUPPORTED; goto out; }
    info->ref_error = RS_INVALID_PARAMETER;
    info->transformer_refused = 1;
    rc = OJ_ERR_MALFORMED;
    goto out;
  }
  if (!spec || (info->ncomp != 3 && info->ncomp != 1) || info->precision != 8) { rc = OJ_ERR_UNSUPPORTED; goto out; }
  /* A merging specification and no residual codestream -- what the reference's encoder writes for `-R n` without `-r` from a
   * picture of more than eight bits (HDR or 16-bit integer): hidden refinement scans, an L table that expands 8 + n bits to
   * the output's depth, an output conversion.  ColorTransformerFactory::BuildColorTransformer (colortransformerfactory.cpp:
   * 262-283) builds the Extended transformer with R transformation "zero": the whole legacy chain -- L transformation, L tables,
   * C transformation, clamp or cast to half float -- with nothing merged (rr = m_lOutDCShift, colortrafo/ycbcrtrafo.cpp:744-746,
   * 861-878); InstallIntegerParameters (:300-594) looks at the L tables and the L and C transformations only (`residual` false). */
  lonly = resi == NULL;
  nc = info->ncomp; /* three components, or one: a grey scale picture with a residual (`jpeg -r ... in.pgm`) */
  xt.nc = nc;
  /* the specification's own boxes are searched first (primary list), then the file's (boxes/namespace.cpp:60-125) */
  for (j = 0; j + 8 <= spec->len;) { /* superbox: LBox(4) TBox(4) payload (boxes/superbox.cpp) */
    uint32_t l = ((uint32_t)rd16(spec->data + j) << 16) | (uint32_t)rd16(spec->data + j + 2);
    uint32_t t = ((uint32_t)rd16(spec->data + j + 4) << 16) | (uint32_t)rd16(spec->data + j + 6);
    const uint8_t *pl = spec->data + j + 8;
    if (l < 8 || j + l > spec->len) { rc = OJ_ERR_MALFORMED; goto out; }
    if (t == BOXID('L', 'T', 'R', 'F')) ltrafo = pl[0] >> 4;
    else if (t == BOXID('R', 'T', 'R', 'F')) rtrafo = pl[0] >> 4;
    else if (t == BOXID('C', 'T', 'R', 'F')) ctrafo = pl[0] >> 4;
    else if (t == BOXID('L', 'P', 'T', 'S')) { lidx[0] = pl[0] >> 4; lidx[1] = pl[0] & 15; lidx[2] = pl[1] >> 4; lidx[3] = pl[1] & 15; }
    else if (t == BOXID('Q', 'P', 'T', 'S')) { qidx[0] = pl[0] >> 4; qidx[1] = pl[0] & 15; qidx[2] = pl[1] >> 4; qidx[3] = pl[1] & 15; }
    else if (t == BOXID('R', 'P', 'T', 'S')) { r2idx[0] = pl[0] >> 4; r2idx[1] = pl[0] & 15; r2idx[2] = pl[1] >> 4; r2idx[3] = pl[1] & 15; }
    else if (t == BOXID('O', 'C', 'O', 'N')) ocon = pl[0];
    else if (t == BOXID('R', 'S', 'P', 'C')) { /* boxes/refinementspecbox.cpp:57-83 */
      hidden_l = pl[0] >> 4; hidden_r = pl[0] & 15;
      if (hidden_l > 4 || hidden_r > 4) { rc = OJ_ERR_MALFORMED; goto out; }
    }
    else if (t == BOXID('L', 'D', 'C', 'T')) { if ((pl[0] >> 4) != 0 || (pl[0] & 15)) { rc = OJ_ERR_UNSUPPORTED; goto out; } } /* only the fixpoint DCT in the base */
    else if (t == BOXID('R', 'D', 'C', 'T')) { /* boxes/dctbox.cpp:60-92: 0 = fixpoint DCT, 2 = integer DCT (outside the subset), 3 = bypass (+ noise shaping) */
      rdct = pl[0];
      if ((rdct >> 4) != 0 && (rdct >> 4) != 3) { rc = OJ_ERR_UNSUPPORTED; goto out; }
      if ((rdct & 15) > 1 || ((rdct & 15) && (rdct >> 4) != 3)) { rc = OJ_ERR_MALFORMED; goto out; }
    }
    else if (t == BOXID('T', 'O', 'N', 'E') || t == BOXID('C', 'U', 'R', 'V') || t == BOXID('M', 'T', 'R', 'X')) {
      rc = register_box(t, pl, l - 8, nlt, mtx, have_mtx);
      if (rc) goto out;
    }
    else if (t == BOXID('C', 'P', 'T', 'S') || t == BOXID('D', 'P', 'T', 'S') || t == BOXID('S', 'P', 'T', 'S') || t == BOXID('P', 'P', 'T', 'S') ||
             t == BOXID('D', 'T', 'R', 'F') || t == BOXID('S', 'T', 'R', 'F') || t == BOXID('F', 'T', 'R', 'X')) {
      rc = OJ_ERR_UNSUPPORTED; goto out; /* L2 / R / S / P tables, D and S transformations, float matrices: profiles A and B */
    }
    /* (any other type: MergingSpecBox::CreateBox knows no such box and the superbox skips it, boxes/superbox.cpp:161-176) */
    j += l;
  }
  for (b = 0; b < ps.nboxes; b++) {
    if (!boxes[b].complete) continue;
    rc = register_box(boxes[b].type, boxes[b].data, boxes[b].len, nlt, mtx, have_mtx);
    if (rc) goto out;
  }
  /* A TONE box that never completes stays in the reference's list as an object that was constructed and never parsed: no entries,
   * and a table index nobody initialised (boxes/tonemapperbox.hpp:64-76) -- in practice the zero of fresh heap memory.  A
   * specification that names table 0 finds it (namespace.cpp:60-91) and its ScaledTableOf refuses, -1024
   * (inversetonemappingbox.cpp:192-212) -- not "does not exist", -1031.  Restated as the table with no entries. */
  if (!nlt[0].kind)
    for (b = 0; b < ps.nboxes; b++)
      if (!boxes[b].complete && boxes[b].type == BOXID('T', 'O', 'N', 'E')) { nlt[0].kind = 1; nlt[0].entries = 0; nlt[0].resbits = 0; }
  /* codestream/tables.cpp:1994-2031 and the R analogue: undefined -> YCbCr for three components */
  if (nc == 1) {
    /* one component: the L and C transformation boxes must not exist (tables.cpp:2003-2005, 2079-2081), everything is the identity;
     * an R transformation other than the identity has no transformer (BuildIntegerTransformationSimple, colortransformerfactory.cpp:681-757) */
    if (ltrafo != 255 || ctrafo != 255) { info->ref_error = -1038; rc = OJ_ERR_MALFORMED; goto late; }
    if (!lonly && rtrafo != 255 && rtrafo != 1) { rc = OJ_ERR_UNSUPPORTED; goto out; }
    ltrafo = 1;
    if (!lonly) rtrafo = 1;
  }
  /* undefined: the rule of plain JPEG -- three components and no Adobe marker that says "none" (Tables::LTrafoTypeOf, tables.cpp:2021-2030) */
  if (ltrafo == 255) ltrafo = (nc == 3 && info->adobe_transform != 0) ? 2 : 1;
  /* Tables::RTrafoTypeOf (codestream/tables.cpp:2040-2075) is asked whether there is a residual or not: "Found an invalid
   * residual transformation" for zero and JPEG_LS; any other value is never used without one */
  if (lonly && (rtrafo == 0 || rtrafo == 3)) { info->ref_error = RS_MALFORMED_STREAM; rc = OJ_ERR_MALFORMED; goto late; }
  if (rtrafo == 255 || lonly) rtrafo = 2;
  /* (what the colour transformer finds: behind both codestreams' verdicts, see `late`) */
  if (ltrafo == 0 || ltrafo == 3 || ltrafo == 4) { info->ref_error = RS_MALFORMED_STREAM; rc = OJ_ERR_MALFORMED; goto late; } /* "the base transformation ... is invalid" */
  if (rtrafo == 3) { info->ref_error = RS_MALFORMED_STREAM; rc = OJ_ERR_MALFORMED; late_residual_only = 1; goto late; }
  if (rtrafo == 0) { rc = OJ_ERR_UNSUPPORTED; goto out; } /* zero */
  if (rtrafo == 4 && nc != 3) { rc = OJ_ERR_UNSUPPORTED; goto out; }
  if (ctrafo != 255 && ctrafo != 1 && ctrafo < 5) { info->ref_error = RS_MALFORMED_STREAM; rc = OJ_ERR_MALFORMED; goto late; }
  if (ocon < 0 && lonly) ocon = 0x02; /* no output conversion box: no extra bits, clipping (boxes/mergingspecbox.cpp:323-331, 648-656) */
  /* the lossless flag only changes the residual's side (codestream/tables.cpp:1643, 1687; marker/frame.cpp:595), the output
   * lookup indices are read and never used by the decoder: without a residual neither matters */
  if (lonly) ocon &= ~0x09;
  /* (the output lookup flag and its table indices are read -- boxes/outputconversionbox.cpp:91-127 -- and never used by the
   * decoder: MergingSpecBox::OutputConversionLookupOf has no caller; the encoder sets them for alpha channels with a residual) */
  if (ocon >= 0) ocon &= ~0x01;
  if (ocon < 0) { rc = OJ_ERR_UNSUPPORTED; goto out; }
  xt.outmax = ((int64_t)1 << (8 + (ocon >> 4))) - 1;
  xt.outshift = (xt.outmax + 1) >> 1;
  xt.is_float = (ocon & 0x04) ? 1 : 0;
  xt.clamp = (ocon & 0x02) ? 1 : 0;
  /* without a residual only the clamping flavours of the transformer exist (colortransformerfactory.cpp:698-725, 850-885):
   * INVALID_PARAMETER "The combination of L and R transformation is non-standard and not supported" */
  if (!xt.clamp && lonly) { info->ref_error = RS_INVALID_PARAMETER; rc = OJ_ERR_MALFORMED; goto late; }
  /* Which transformers exist beside a residual (colortransformerfactory.cpp:681-757, 793-995): R = YCbCr / free-form needs the
   * clamping flavours, R = RCT the ones without, R = identity has all four; anything else: INVALID_PARAMETER "The combination of L
   * and R transformation is non-standard and not supported" */
  if (!lonly && ((rtrafo == 4 && xt.clamp) || (rtrafo != 4 && rtrafo != 1 && !xt.clamp))) { info->ref_error = RS_INVALID_PARAMETER; rc = OJ_ERR_MALFORMED; goto late; }
  /* ... and only a transformer that exists gets its parameters (InstallIntegerParameters, colortransformerfactory.cpp:283-286): */
  /* OBJECT_DOESNT_EXIST "the base / color / residual transformation specified in the codestream does not exist" (colortransformerfactory.cpp:355-400, 528-566) */
  if ((ltrafo >= 5 && !have_mtx[ltrafo]) || (ctrafo != 255 && ctrafo >= 5 && !have_mtx[ctrafo])) { info->ref_error = RS_OBJECT_DOESNT_EXIST; rc = OJ_ERR_MALFORMED; goto late; }
  if (rtrafo >= 5 && !have_mtx[rtrafo]) { info->ref_error = RS_OBJECT_DOESNT_EXIST; rc = OJ_ERR_MALFORMED; late_residual_only = 1; goto late; } /* (looked up beside a residual frame only) */
  /* fractional bits of the residual path (Tables::FractionalColorBitsOf, tables.cpp:1621-1660): RCT one, the identity none when
   * the lossless flag is set, four otherwise */
  xt.rct = rtrafo == 4;
  xt.rbits = rtrafo == 4 ? 1 : (rtrafo == 1 && (ocon & 0x08)) ? 0 : 4;
  if (!lonly && xt.rbits == 0 && xt.clamp) { rc = OJ_ERR_UNSUPPORTED; goto out; } /* (the reference indexes a table of 2^Pr entries with 2^(Pr+4): not followed) */
  if (xt.is_float && xt.outmax != 65535) { rc = OJ_ERR_UNSUPPORTED; goto out; }
  /* free-form matrices run through the YCbCr branches of the transformer (colortransformerfactory.cpp:1036-1058) */
  xt.ltrafo_ycbcr = ltrafo != 1; xt.rtrafo_ycbcr = rtrafo != 1 && rtrafo != 4;
  if (disable_to_rgb && ltrafo == 2) xt.ltrafo_ycbcr = 0; /* MergingSpecBox::YCbCr only: a free-form matrix stays */
  memcpy(xt.lmat, ltrafo >= 5 ? mtx[ltrafo] : ltrafo == 2 ? std_ycc : std_id, sizeof(xt.lmat));
  memcpy(xt.rmat, rtrafo >= 5 ? mtx[rtrafo] : (rtrafo == 2 || rtrafo == 4) ? std_ycc : std_id, sizeof(xt.rmat));
  memcpy(xt.cmat, (ctrafo != 255 && ctrafo >= 5) ? mtx[ctrafo] : std_id, sizeof(xt.cmat));
  xt.rbypass = (rdct >> 4) == 3; xt.rnoise = rdct & 1;
  if (!lonly && xt.rbits != 4 && !xt.rbypass) { rc = OJ_ERR_UNSUPPORTED; goto out; } /* (a DCT with other preshifts: IDCT<1> / IDCT<0>, not restated) */
  /* the residual codestream is needed for the table dimensions */
  rc = lonly ? OJ_OK : read_residual_info(resi->data, resi->len, &rinfo);
  /* Image::ParseResidualStream (codestream/image.cpp:1289-1299) compares right behind the residual frame header, where a
   * residual codestream with a DNL marker still has zero lines: "residual image dimensions do not match ..." */
  if (!rc && !lonly && (rinfo.dnl || rinfo.width != info->width || rinfo.height != info->height)) { rinfo.ref_error = RS_MALFORMED_STREAM; rc = OJ_ERR_MALFORMED; }
  if (rc) {
    /* ... but the reference only gets there behind the legacy codestream's EOI: whatever stops the legacy codestream first
     * is what it reports (and without an EOI it never looks at the residual: not followed here) */
    const int rrc = rc, rerr = rinfo.ref_error;
    oj_parser ls;
    oj_info ltmp;
    memset(&ls, 0, sizeof(ls)); memset(&ltmp, 0, sizeof(ltmp));
    ls.data = data; ls.len = len; ls.info = &ltmp; ls.hidden = hidden_l; ls.xt_legacy = 1; ls.in_memory = given != NULL;
    for (c = 0; c < nc; c++) {
      planes[c] = (int32_t *)calloc((size_t)info->bw[c] * info->bh[c] * 64, sizeof(int32_t));
      if (!planes[c]) { rc = OJ_ERR_NOMEM; goto out; }
    }
    rc = walk(&ls, planes);
    if (rc) info->ref_error = ltmp.ref_error;
    if (!rc && ls.eoi_frame) { rc = decode_hidden_scans(&ls, boxes, ps.nboxes, BOXID('F', 'I', 'N', 'E'), planes); if (rc) info->ref_error = ls.err; }
    if (!rc) {
      if (ls.eoi_image || given) { rc = rrc; info->ref_error = rerr; } /* (the alpha image's residual: always reached) */
      else retry_lonly = 1; /* the residual codestream that does not parse is never looked at */
    }
    goto out;
  }
  for (c = 0; c < nc; c++) {
    /* L: ScaledTableOf(8 + hidden bits, 16, 0, 0), default = identity with e = 1; Q: (Pr + hidden bits, 16, 4, 4) and
     * R2: (16, 16, 4, 0), defaults = identities with e = 0 (colortransformerfactory.cpp:312-345, 435-474, 486-520) */
    oj_nlt id1, id0;
    const oj_nlt *t;
    const int pr = rinfo.precision + hidden_r;
    const int outbits = 8 + (ocon >> 4); /* JPGTAG_IMAGE_PRECISION of the output: 8 + extra range bits (integer JPEG XT of an 8-bit picture: 8) */
    memset(&id1, 0, sizeof(id1)); id1.kind = 2; id1.type = 2; id1.e = 1;
    memset(&id0, 0, sizeof(id0)); id0.kind = 2; id0.type = 2; id0.e = 0;
    t = lidx[c] == 255 ? &id1 : &nlt[lidx[c]];
    if (!t->kind) { info->ref_error = -1031; rc = OJ_ERR_MALFORMED; goto late; } /* OBJECT_DOESNT_EXIST "the L lookup table specified in the codestream does not exist" */
    owned[c] = scaled_table(t, 8 + hidden_l, outbits, 0, 0, &rc);
    if (!owned[c]) { info->ref_error = rc; rc = rc ? OJ_ERR_MALFORMED : OJ_ERR_NOMEM; if (info->ref_error) goto late; goto out; }
    xt.ltable[c] = owned[c];
    if (lonly) continue; /* (Q and R2 tables are looked up beside a residual frame only, colortransformerfactory.cpp:452, 496) */
    if (pr - (xt.rbits == 1) > 16) { rc = OJ_ERR_UNSUPPORTED; goto out; }
    t = qidx[c] == 255 ? &id0 : &nlt[qidx[c]];
    if (!t->kind) { info->ref_error = -1031; rc = OJ_ERR_MALFORMED; late_residual_only = 1; goto late; }
    /* (the RCT's extra bit is a precision bit, not a fractional one: colortransformerfactory.cpp:441-446) */
    owned[3 + c] = scaled_table(t, pr - (xt.rbits == 1), outbits, xt.rbits, xt.rbits, &rc);
    if (!owned[3 + c]) { info->ref_error = rc; rc = rc ? OJ_ERR_MALFORMED : OJ_ERR_NOMEM; late_residual_only = 1; if (info->ref_error) goto late; goto out; }
    xt.qlut[c] = owned[3 + c];
    if (!xt.clamp) continue; /* (R2 tables exist with clipping only, colortransformerfactory.cpp:481) */
    t = r2idx[c] == 255 ? &id0 : &nlt[r2idx[c]];
    if (!t->kind) { info->ref_error = -1031; rc = OJ_ERR_MALFORMED; late_residual_only = 1; goto late; }
    owned[6 + c] = scaled_table(t, outbits, outbits, 4, 0, &rc);
    if (!owned[6 + c]) { info->ref_error = rc; rc = rc ? OJ_ERR_MALFORMED : OJ_ERR_NOMEM; late_residual_only = 1; if (info->ref_error) goto late; goto out; }
    xt.r2lut[c] = owned[6 + c];
  }
  /* the residual codestream is an ordinary codestream of its own (codestream/image.cpp:1264-1300) */
  if (!lonly && (rinfo.width != info->width || rinfo.height != info->height || rinfo.ncomp != info->ncomp)) { rc = OJ_ERR_MALFORMED; goto out; }
  if (rinfo.precision + hidden_r - (xt.rbits == 1) > 16) { rc = OJ_ERR_UNSUPPORTED; goto out; }
  info->ycbcr = xt.ltrafo_ycbcr;
  for (c = 0; c < nc; c++) {
    planes[c] = (int32_t *)malloc((size_t)info->bw[c] * info->bh[c] * 64 * sizeof(int32_t));
    rplanes[c] = lonly ? NULL : (int32_t *)malloc((size_t)rinfo.bw[c] * rinfo.bh[c] * 64 * sizeof(int32_t));
    if (!planes[c] || (!lonly && !rplanes[c])) { rc = OJ_ERR_NOMEM; goto out; }
  }
  {
    /* visible scans with their bits moved up by the hidden ones, then the hidden refinement scans; from here on both
     * frames simply have precision + hidden bits (Frame::HiddenPrecisionOf, marker/frame.cpp:368-373) */
    oj_parser ls, rs;
    oj_info ltmp, rtmp;
    memset(&ls, 0, sizeof(ls)); memset(&rs, 0, sizeof(rs)); memset(&ltmp, 0, sizeof(ltmp)); memset(&rtmp, 0, sizeof(rtmp));
    ls.data = data; ls.len = len; ls.info = &ltmp; ls.hidden = hidden_l; ls.xt_legacy = 1; ls.in_memory = given != NULL;
    /* (a legacy frame whose height arrives in a DNL marker -- no residual codestream beside it, the reference refuses those: the
     * header pass has decoded the first scan for the height and the block rows, as for a plain frame) */
    ls.known_height = info->dnl ? info->height : 0;
    for (c = 0; c < OJ_MAX_COMP; c++) ls.known_bh[c] = info->bh[c];
    if (!lonly) { rs.data = resi->data; rs.len = resi->len; rs.info = &rtmp; rs.hidden = hidden_r; rs.nested = 1; }
    for (c = 0; c < nc; c++) {
      memset(planes[c], 0, (size_t)info->bw[c] * info->bh[c] * 64 * sizeof(int32_t));
      if (!lonly) memset(rplanes[c], 0, (size_t)rinfo.bw[c] * rinfo.bh[c] * 64 * sizeof(int32_t));
    }
    rc = walk(&ls, planes);
    if (rc) info->ref_error = ltmp.ref_error;
    if (ltmp.dnl) memcpy(info->rows, ltmp.rows, sizeof(ltmp.rows));
    /* (hidden scans and residual only behind an EOI, see eoi_frame / eoi_image -- the alpha image's residual always: xt_codestreams_verdict) */
    if (given) ls.eoi_image = 1;
    if (!rc && ls.eoi_frame) { rc = decode_hidden_scans(&ls, boxes, ps.nboxes, BOXID('F', 'I', 'N', 'E'), planes); if (rc) info->ref_error = ls.err; }
    if (!rc && ls.eoi_image && !lonly) {
      rc = walk(&rs, rplanes);
      if (rc) info->ref_error = rtmp.ref_error;
      if (!rc && rs.eoi_frame) { rc = decode_hidden_scans(&rs, boxes, ps.nboxes, BOXID('R', 'F', 'I', 'N'), rplanes); if (rc) info->ref_error = rs.err; }
    } else if (!rc) {
      xt.no_residual = 1;
      /* ... and the transformer is built without a residual frame: only the clamping flavours exist then
       * (colortransformerfactory.cpp:262-283 with R transformation "zero", :698-725, 850-885) */
      if (!xt.clamp) { info->ref_error = RS_INVALID_PARAMETER; info->transformer_refused = 1; rc = OJ_ERR_MALFORMED; }
    }
    if (rc) goto out; /* (the reference's error code travels in info->ref_error) */
    /* (header_error_rc and no error here: the header walk skips entropy coded data by looking for the next marker and took one
     * INSIDE a damaged scan for a header -- the decoding walk resynchronised over it like the reference does: its verdict counts) */
    (void)header_error_rc; (void)header_error;
    if (!lonly && !xt.no_residual && rs.late_quant_missing) { /* (see rs_scan: the first request finds the residual's quantiser table missing) */
      info->ref_error = RS_OBJECT_DOESNT_EXIST; info->transformer_refused = 1; rc = OJ_ERR_MALFORMED; goto out;
    }
    memcpy(info->cquant, ltmp.cquant, sizeof(ltmp.cquant)); memcpy(info->comp_seen, ltmp.comp_seen, sizeof(ltmp.comp_seen));
    if (!lonly) { memcpy(rinfo.cquant, rtmp.cquant, sizeof(rtmp.cquant)); memcpy(rinfo.comp_seen, rtmp.comp_seen, sizeof(rtmp.comp_seen)); }
    info->scan_state_valid = ltmp.scan_state_valid; rinfo.scan_state_valid = rtmp.scan_state_valid;
    info->precision += hidden_l;
    rinfo.precision += hidden_r;
  }
  xt.rinfo = lonly ? NULL : &rinfo; xt.rplanes = rplanes;
  if (is_float) *is_float = xt.is_float;
  if (lplanes_out) { /* the legacy frame's coefficients as the merge sees them (hidden bits included): the caller's now */
    for (c = 0; c < nc; c++) { lplanes_out[c] = planes[c]; planes[c] = NULL; }
    if (g_rplanes_out && !lonly) { /* ... and the residual frame's (oj_decode_xt_planes2) */
      for (c = 0; c < nc; c++) { g_rplanes_out[c] = rplanes[c]; rplanes[c] = NULL; }
      *g_rinfo_out = rinfo;
    }
    goto out;
  }
  if (rq_out) {
    /* BlockBitmapRequester::PrepareForDecoding, control/blockbitmaprequester.cpp:298-372: cursors at the first rows of both
     * images, upsamplers for the subsampled components of either */
    struct oj_xt_ctx *ctx = (struct oj_xt_ctx *)calloc(1, sizeof(*ctx));
    oj_requester *rq;
    if (!ctx) { rc = OJ_ERR_NOMEM; goto out; }
    ctx->xt = xt;
    ctx->rinfo = rinfo;
    ctx->xt.rinfo = lonly ? NULL : &ctx->rinfo;
    ctx->xt.rplanes = ctx->rplanes;
    rq = oj_requester_new(info, planes);
    if (!rq) { free(ctx); rc = OJ_ERR_NOMEM; goto out; }
    rq->xt = &ctx->xt;
    rq->rf = &ctx->rinfo;
    rq->owner = ctx;
    for (c = 0; c < nc; c++) {
      ctx->planes[c] = planes[c]; planes[c] = NULL; /* (the requester reads them through rq->planes) */
      if (lonly) continue;
      ctx->rplanes[c] = rplanes[c]; rplanes[c] = NULL;
      rq->rplanes[c] = ctx->rplanes[c];
      rq->rrows[c] = (rinfo.ch[c] + 7) >> 3;
      if (rinfo.subx[c] > 1 || rinfo.suby[c] > 1) {
        oj_up *u = (oj_up *)calloc(1, sizeof(*u));
        if (!u) { oj_requester_free(rq); rc = OJ_ERR_NOMEM; goto out; }
        u->sx = rinfo.subx[c]; u->sy = rinfo.suby[c];
        u->pw = info->width; u->ph = info->height;
        u->width = (info->width + u->sx - 1) / u->sx;
        u->total = (info->height + u->sy - 1) / u->sy;
        rq->rup[c] = u;
        rq->subsampling = 1;
      }
    }
    for (c = 0; c < 9; c++) { ctx->owned[c] = owned[c]; owned[c] = NULL; }
    *rq_out = rq;
    goto out;
  }
  *pixels = (uint16_t *)malloc((size_t)info->width * info->height * (size_t)nc * sizeof(uint16_t));
  if (!*pixels) { rc = OJ_ERR_NOMEM; goto out; }
  rc = reconstruct_ex(info, planes, NULL, *pixels, xt.ltrafo_ycbcr, &xt);
  if (rc) { free(*pixels); *pixels = NULL; }
  goto out;
late:
  /* What the colour transformer refuses when the first request builds it: the whole file has been read by then, and what
   * stopped either codestream was reported instead (xt_codestreams_verdict).  A table of the residual's side is looked up only
   * when there is a residual frame to merge -- without an EOI behind the legacy codestream the picture comes out without
   * (not followed here). */
  {
    const int lrc = rc, lerr = info->ref_error;
    int verr = 0, eoi = 0;
    rc = xt_codestreams_verdict(data, len, info, boxes, ps.nboxes, resi, hidden_l, hidden_r, &verr, &eoi, given != NULL);
    if (rc) info->ref_error = verr;
    else if (!eoi && late_residual_only) { retry_lonly = 1; info->ref_error = 0; }
    else { rc = lrc; info->ref_error = lerr; info->transformer_refused = 1; }
  }
out:
  for (c = 0; c < OJ_MAX_COMP; c++) { free(planes[c]); free(rplanes[c]); }
  for (c = 0; c < 16; c++) free(nlt[c].lut);
  for (c = 0; c < 9; c++) free(owned[c]);
  free_boxes(boxes, ps.nboxes);
  if (retry_lonly && !(flags & XT_IGNORE_RESIDUAL)) return xt_decode_common(data, len, info, pixels, is_float, rq_out, flags | XT_IGNORE_RESIDUAL, lplanes_out, given, ngiven);
  return rc;
}

int oj_decode_xt(const uint8_t *data, size_t len, oj_info *info, uint16_t **pixels, int *is_float)
{
  return xt_decode_common(data, len, info, pixels, is_float, NULL, 0, NULL, NULL, 0);
}

/* The legacy frame's coefficient planes as the JPEG XT merge sees them -- visible scans moved up by the hidden bits, hidden
 * refinement scans applied: planes[c] is malloc'ed (bw[c] * bh[c] * 64 int32, free with oj_free); info->precision includes the
 * hidden bits.  For the tests of the product's host decoder. */
int oj_decode_xt_planes(const uint8_t *data, size_t len, oj_info *info, int32_t **planes)
{
  return xt_decode_common(data, len, info, NULL, NULL, NULL, 0, planes, NULL, 0);
}

/* ... and the residual frame's beside them (rplanes[c]: malloc'ed like planes[c], NULL where the file has no residual codestream;
 * rinfo->precision includes the residual's hidden bits).  Test infrastructure, one caller at a time. */
int oj_decode_xt_planes2(const uint8_t *data, size_t len, oj_info *info, int32_t **planes, oj_info *rinfo, int32_t **rplanes)
{
  int rc, c;
  for (c = 0; c < OJ_MAX_COMP; c++) rplanes[c] = NULL;
  memset(rinfo, 0, sizeof(*rinfo));
  g_rplanes_out = rplanes; g_rinfo_out = rinfo;
  rc = xt_decode_common(data, len, info, NULL, NULL, NULL, 0, planes, NULL, 0);
  g_rplanes_out = NULL; g_rinfo_out = NULL;
  return rc;
}

/* ... as the reference's command line decodes it with -c (no colour transformation) */
int oj_decode_xt_ex(const uint8_t *data, size_t len, oj_info *info, uint16_t **pixels, int *is_float, int disable_to_rgb)
{
  return xt_decode_common(data, len, info, pixels, is_float, NULL, disable_to_rgb ? XT_DISABLE_TO_RGB : 0, NULL, NULL, 0);
}

/* A requester (oj_requester_display / _cursor / _free) on a JPEG XT stream: both codestreams decoded, the residual image's
 * cursors and upsamplers beside the legacy image's.  *out_max = 2^(8 + extra range bits) - 1: samples of 2 bytes above 255. */
int oj_xt_requester_new(const uint8_t *data, size_t len, oj_info *info, oj_requester **rq, int *is_float, int *out_max)
{
  const int rc = xt_decode_common(data, len, info, NULL, is_float, rq, 0, NULL, NULL, 0);
  if (!rc && out_max) *out_max = (int)(*rq)->xt->outmax;
  return rc;
}

/* Exact half -> float expansion the reference CLI applies before writing PFM (cmd/iohelpers.hpp:60-77). */
float oj_half_to_float(uint16_t h)
{
  const int sign = h >> 15, exp = (h >> 10) & 31, man = h & 1023;
  double v;
  if (exp == 0) v = ldexp((double)man, -14 - 10);
  else if (exp == 31) v = HUGE_VAL;
  else v = ldexp((double)(man | 1024), -15 - 10 + exp);
  return (float)(sign ? -v : v);
}

int oj_decode(const uint8_t *data, size_t len, oj_info *info, uint8_t **pixels)
{
  int32_t *planes[OJ_MAX_COMP] = {0, 0, 0, 0};
  int rc = oj_read_info(data, len, info), c;
  *pixels = NULL;
  if (rc) return rc;
  for (c = 0; c < info->ncomp; c++) {
    planes[c] = (int32_t *)malloc((size_t)info->bw[c] * info->bh[c] * 64 * sizeof(int32_t));
    if (!planes[c]) { rc = OJ_ERR_NOMEM; goto out; }
  }
  rc = oj_decode_coefficients(data, len, info, planes);
  if (rc) goto out;
  *pixels = (uint8_t *)malloc((size_t)info->width * info->height * info->ncomp);
  if (!*pixels) { rc = OJ_ERR_NOMEM; goto out; }
  rc = oj_reconstruct(info, planes, *pixels, -1);
  if (rc) { free(*pixels); *pixels = NULL; }
out:
  for (c = 0; c < OJ_MAX_COMP; c++) free(planes[c]);
  return rc;
}

/* The alpha channel of a JPEG XT file (part 9 of the standard as the reference implements it): a one-component image of its own
 * in the ALFA box -- SOI, tables, frame, scans (Image::ParseAlphaChannel, codestream/image.cpp:1337-1404, entered from the legacy
 * image's trailer, :1430-1460) -- with the alpha merging specification ASPC in the role of SPEC (Tables::ResidualSpecsOf,
 * codestream/tables.hpp:451-460), the boxes ARES / AFIN / ARRF in the roles of RESI / FINE / RFIN (boxes/databox.hpp:90-96,
 * codestream/tables.cpp:752-775, 850-903) and the file's TONE / CURV / MTRX boxes behind the specification's own
 * (Tables::AlphaNamespace, codestream/tables.cpp:109, 2115-2125).  The decoder hands the plane out beside the picture, it does
 * not composite; mode and matte colour of the AMUL box (boxes/alphabox.cpp:60-90) are what JPEG::GetInformation reports
 * (interface/jpeg.cpp:919-945).
 * -> 16-bit codes, one per pixel (*out_max = 2^bits - 1; half-float codes when *is_float); *mode: AMUL's compositing method, -1
 * without that box.  OJ_ERR_UNSUPPORTED: the file has no (complete) ALFA box. */
int oj_decode_alpha(const uint8_t *data, size_t len, oj_info *info, uint16_t **pixels, int *is_float, int *out_max, int *mode, uint32_t matte[3])
{
  oj_box boxes[OJ_MAX_BOXES], given[OJ_MAX_BOXES];
  oj_parser ps;
  oj_info main_info;
  const oj_box *alfa = NULL, *aspc = NULL;
  int b, n = 0, rc, isf = 0;
  *pixels = NULL;
  if (mode) *mode = -1;
  if (matte) matte[0] = matte[1] = matte[2] = 0;
  memset(&ps, 0, sizeof(ps)); memset(&main_info, 0, sizeof(main_info));
  ps.data = data; ps.len = len; ps.info = &main_info; ps.boxes = boxes; ps.walk_all = 1;
  rc = walk(&ps, NULL);
  if (rc) { free_boxes(boxes, ps.nboxes); return rc; }
  for (b = 0; b < ps.nboxes; b++) {
    const oj_box *bx = &boxes[b];
    uint32_t t = bx->type;
    if (!bx->complete) continue;
    if (t == BOXID('A', 'L', 'F', 'A')) { alfa = bx; continue; }
    if (t == BOXID('A', 'S', 'P', 'C')) { aspc = bx; t = BOXID('S', 'P', 'E', 'C'); }
    else if (t == BOXID('A', 'R', 'E', 'S')) t = BOXID('R', 'E', 'S', 'I');
    else if (t == BOXID('A', 'F', 'I', 'N')) t = BOXID('F', 'I', 'N', 'E');
    else if (t == BOXID('A', 'R', 'R', 'F')) t = BOXID('R', 'F', 'I', 'N');
    else if (t != BOXID('T', 'O', 'N', 'E') && t != BOXID('C', 'U', 'R', 'V') && t != BOXID('M', 'T', 'R', 'X')) continue;
    given[n] = *bx;
    given[n].type = t;
    n++;
  }
  /* (the reference turns to the ALFA box behind the legacy codestream's EOI, codestream/image.cpp:1430-1460: without one the file
   * has no alpha channel) */
  if (!alfa || !ps.eoi_image) { free_boxes(boxes, ps.nboxes); return OJ_ERR_UNSUPPORTED; }
  {
    /* Image::ParseAlphaChannel, codestream/image.cpp:1366-1380: right behind the alpha image's frame header -- whatever its scans
     * would do to a frame of that size -- its dimensions are compared with the image's ("residual image dimensions do not match
     * the dimensions of the legacy image", the residual's message), and it may have one component only.  What stops the SOI, the
     * tables or the frame header itself comes first: the walk below reports it. */
    oj_info ai;
    if (oj_read_info(alfa->data, alfa->len, &ai) == OJ_OK && (ai.width != main_info.width || ai.height != main_info.height || ai.ncomp != 1)) {
      memset(info, 0, sizeof(*info));
      info->ref_error = RS_MALFORMED_STREAM;
      free_boxes(boxes, ps.nboxes);
      return OJ_ERR_MALFORMED;
    }
  }
  if (aspc) { /* AMUL inside the alpha merging specification */
    size_t j;
    for (j = 0; j + 8 <= aspc->len;) {
      const uint32_t l = ((uint32_t)rd16(aspc->data + j) << 16) | (uint32_t)rd16(aspc->data + j + 2);
      const uint32_t t = ((uint32_t)rd16(aspc->data + j + 4) << 16) | (uint32_t)rd16(aspc->data + j + 6);
      if (l < 8 || j + l > aspc->len) break;
      if (t == BOXID('A', 'M', 'U', 'L') && l == 8 + 10) {
        const uint8_t *pl = aspc->data + j + 8;
        if (mode) *mode = pl[0] >> 4;
        if (matte) { matte[0] = rd16(pl + 2); matte[1] = rd16(pl + 4); matte[2] = rd16(pl + 6); }
      }
      j += l;
    }
  }
  rc = xt_decode_common(alfa->data, alfa->len, info, pixels, &isf, NULL, 0, NULL, given, n);
  if (rc == OJ_ERR_UNSUPPORTED && aspc) {
    /* a specification outside this restatement: the reference has read the alpha image's codestreams all the same, and what stops
     * one of them fails the read (the rest waits for a request for alpha pixels: not followed) */
    oj_info ai;
    const oj_box *ares = NULL;
    int verr = 0, eoi = 0;
    for (b = 0; b < n; b++)
      if (given[b].type == BOXID('R', 'E', 'S', 'I') && given[b].complete) ares = &given[b];
    if (oj_read_info(alfa->data, alfa->len, &ai) == OJ_OK) {
      const int vrc = xt_codestreams_verdict(alfa->data, alfa->len, &ai, given, n, ares, 0, 0, &verr, &eoi, 1);
      if (vrc && vrc != OJ_ERR_UNSUPPORTED && verr) { info->ref_error = verr; info->transformer_refused = 0; rc = OJ_ERR_MALFORMED; }
    }
  }
  if (!rc) {
    /* (the depth of the output: the OCON box of the specification) */
    int bits = 8;
    size_t j;
    for (j = 0; aspc && j + 8 <= aspc->len;) {
      const uint32_t l = ((uint32_t)rd16(aspc->data + j) << 16) | (uint32_t)rd16(aspc->data + j + 2);
      const uint32_t t = ((uint32_t)rd16(aspc->data + j + 4) << 16) | (uint32_t)rd16(aspc->data + j + 6);
      if (l < 8 || j + l > aspc->len) break;
      if (t == BOXID('O', 'C', 'O', 'N') && l >= 9) bits = 8 + (aspc->data[j + 8] >> 4);
      j += l;
    }
    if (out_max) *out_max = (1 << bits) - 1;
  } else if (rc == OJ_ERR_UNSUPPORTED && !aspc) {
    /* no alpha merging specification: a plain one-component picture of 8 or 12 bits */
    int32_t *planes[OJ_MAX_COMP] = {0, 0, 0, 0};
    int c;
    rc = oj_read_info(alfa->data, alfa->len, info);
    for (c = 0; !rc && c < info->ncomp; c++) {
      planes[c] = (int32_t *)malloc((size_t)info->bw[c] * info->bh[c] * 64 * sizeof(int32_t));
      if (!planes[c]) rc = OJ_ERR_NOMEM;
    }
    if (!rc) rc = oj_decode_coefficients(alfa->data, alfa->len, info, planes);
    if (!rc) {
      *pixels = (uint16_t *)malloc((size_t)info->width * info->height * info->ncomp * sizeof(uint16_t));
      if (!*pixels) rc = OJ_ERR_NOMEM;
    }
    if (!rc) {
      if (info->precision > 8) rc = oj_reconstruct16(info, planes, *pixels, -1);
      else {
        uint8_t *p8 = (uint8_t *)malloc((size_t)info->width * info->height * info->ncomp);
        size_t i, cnt = (size_t)info->width * info->height * info->ncomp;
        if (!p8) rc = OJ_ERR_NOMEM;
        else {
          rc = oj_reconstruct(info, planes, p8, -1);
          for (i = 0; !rc && i < cnt; i++) (*pixels)[i] = p8[i];
          free(p8);
        }
      }
      if (rc) { free(*pixels); *pixels = NULL; }
    }
    for (c = 0; c < OJ_MAX_COMP; c++) free(planes[c]);
    if (out_max) *out_max = (1 << info->precision) - 1;
    isf = 0;
  }
  if (!rc && (info->width != main_info.width || info->height != main_info.height || info->ncomp != 1)) {
    info->ref_error = RS_MALFORMED_STREAM; rc = OJ_ERR_MALFORMED; free(*pixels); *pixels = NULL; /* codestream/image.cpp:1370-1380 */
  }
  if (is_float) *is_float = isf;
  free_boxes(boxes, ps.nboxes);
  return rc;
}

void oj_free(void *p) { free(p); }

/* ==================================================================================================================
 * Encoder direction: forward colour transformation, box downsampling, forward DCT + quantisation
 * ================================================================================================================== */
#define F9(x) ((int32_t)((x) * 512.0 + 0.5)) /* TO_FIX of dct/idct.cpp:65 */

/* The quantiser multiplier LONG(FLOAT(1L << 30) / delta + 0.5) of dct/idct.cpp:106: a SINGLE precision quotient (FLOAT
 * is float), widened for the addition.  For deltas that do not divide 2^30 it differs from the exactly rounded value
 * by up to ~3e-8 relative, which flips about one coefficient in several thousand -- the reference binary this oracle
 * is pinned to shows exactly these flips (tests/test_oracle.py::test_forward_matches_the_reference_encoder). */
static int64_t inv_quant(uint16_t delta)
{
  volatile float q = (float)(1L << 30) / (float)delta; /* volatile: keep it a float whatever the FP unit */
  return (int64_t)((double)q + 0.5);
}

/* Quantize of dct/idct.hpp:90-111 without dead zone: preshift 4, FIX_BITS 9, INTERMEDIATE_BITS 0, QUANTIZER_BITS 30 */
static int32_t quantize(int32_t n, int64_t qnt)
{
  const int sh = 9 + 0 + 30 + 4 + 3;
  return (int32_t)(((int64_t)n * qnt + (int64_t)(((uint32_t)(-n)) >> 31) + (((int64_t)1) << (sh - 1))) >> sh);
}


void oj_fdct_block(int32_t out[64], const int32_t in[64], const uint16_t quant[64], int precision)
{
  int32_t t[64];
  int32_t dcoffset = w32((int64_t)(1 << (precision - 1)) << (4 + 3 + 3));
  /* pass over columns (idct.cpp:125-170): t[k*8 + c] = frequency k of column c */
  for (int c = 0; c < 8; c++) {
    const int32_t *s = in + c;
    int32_t tmp0 = w32((int64_t)s[0] + s[56]), tmp1 = w32((int64_t)s[8] + s[48]), tmp2 = w32((int64_t)s[16] + s[40]), tmp3 = w32((int64_t)s[24] + s[32]);
    int32_t tmp10 = w32((int64_t)tmp0 + tmp3), tmp12 = w32((int64_t)tmp0 - tmp3), tmp11 = w32((int64_t)tmp1 + tmp2), tmp13 = w32((int64_t)tmp1 - tmp2);
    tmp0 = w32((int64_t)s[0] - s[56]); tmp1 = w32((int64_t)s[8] - s[48]); tmp2 = w32((int64_t)s[16] - s[40]); tmp3 = w32((int64_t)s[24] - s[32]);
    t[0 + c] = w32((int64_t)tmp10 + tmp11);
    t[32 + c] = w32((int64_t)tmp10 - tmp11);
    int32_t z1 = w32(((int64_t)tmp12 + tmp13) * F9(0.541196100));
    t[16 + c] = w32(((int64_t)z1 + (int64_t)tmp12 * F9(0.765366865) + 256)) >> 9;
    t[48 + c] = w32(((int64_t)z1 + (int64_t)tmp13 * -F9(1.847759065) + 256)) >> 9;
    tmp10 = w32((int64_t)tmp0 + tmp3); tmp11 = w32((int64_t)tmp1 + tmp2); tmp12 = w32((int64_t)tmp0 + tmp2); tmp13 = w32((int64_t)tmp1 + tmp3);
    z1 = w32(((int64_t)tmp12 + tmp13) * F9(1.175875602));
    const int32_t tt0 = w32((int64_t)tmp0 * F9(1.501321110)), tt1 = w32((int64_t)tmp1 * F9(3.072711026)), tt2 = w32((int64_t)tmp2 * F9(2.053119869)),
                  tt3 = w32((int64_t)tmp3 * F9(0.298631336)), tt10 = w32((int64_t)tmp10 * -F9(0.899976223)), tt11 = w32((int64_t)tmp11 * -F9(2.562915447)),
                  tt12 = w32((int64_t)tmp12 * -F9(0.390180644) + z1), tt13 = w32((int64_t)tmp13 * -F9(1.961570560) + z1);
    t[8 + c] = w32((int64_t)tt0 + tt10 + tt12 + 256) >> 9;
    t[24 + c] = w32((int64_t)tt1 + tt11 + tt13 + 256) >> 9;
    t[40 + c] = w32((int64_t)tt2 + tt11 + tt12 + 256) >> 9;
    t[56 + c] = w32((int64_t)tt3 + tt10 + tt13 + 256) >> 9;
  }
  /* pass over rows and quantise (idct.cpp:174-218) */
  for (int r = 0; r < 8; r++) {
    const int32_t *d = t + r * 8;
    int64_t q[8];
    for (int k = 0; k < 8; k++) q[k] = inv_quant(quant[r * 8 + k]);
    int32_t tmp0 = w32((int64_t)d[0] + d[7]), tmp1 = w32((int64_t)d[1] + d[6]), tmp2 = w32((int64_t)d[2] + d[5]), tmp3 = w32((int64_t)d[3] + d[4]);
    int32_t tmp10 = w32((int64_t)tmp0 + tmp3), tmp12 = w32((int64_t)tmp0 - tmp3), tmp11 = w32((int64_t)tmp1 + tmp2), tmp13 = w32((int64_t)tmp1 - tmp2);
    tmp0 = w32((int64_t)d[0] - d[7]); tmp1 = w32((int64_t)d[1] - d[6]); tmp2 = w32((int64_t)d[2] - d[5]); tmp3 = w32((int64_t)d[3] - d[4]);
    out[r * 8 + 0] = quantize(w32(((int64_t)tmp10 + tmp11 - dcoffset) << 9), q[0]);
    out[r * 8 + 4] = quantize(w32(((int64_t)tmp10 - tmp11) << 9), q[4]);
    int32_t z1 = w32(((int64_t)tmp12 + tmp13) * F9(0.541196100));
    out[r * 8 + 2] = quantize(w32((int64_t)z1 + (int64_t)tmp12 * F9(0.765366865)), q[2]);
    out[r * 8 + 6] = quantize(w32((int64_t)z1 + (int64_t)tmp13 * -F9(1.847759065)), q[6]);
    tmp10 = w32((int64_t)tmp0 + tmp3); tmp11 = w32((int64_t)tmp1 + tmp2); tmp12 = w32((int64_t)tmp0 + tmp2); tmp13 = w32((int64_t)tmp1 + tmp3);
    z1 = w32(((int64_t)tmp12 + tmp13) * F9(1.175875602));
    const int32_t tt0 = w32((int64_t)tmp0 * F9(1.501321110)), tt1 = w32((int64_t)tmp1 * F9(3.072711026)), tt2 = w32((int64_t)tmp2 * F9(2.053119869)),
                  tt3 = w32((int64_t)tmp3 * F9(0.298631336)), tt10 = w32((int64_t)tmp10 * -F9(0.899976223)), tt11 = w32((int64_t)tmp11 * -F9(2.562915447)),
                  tt12 = w32((int64_t)tmp12 * -F9(0.390180644) + z1), tt13 = w32((int64_t)tmp13 * -F9(1.961570560) + z1);
    out[r * 8 + 1] = quantize(w32((int64_t)tt0 + tt10 + tt12), q[1]);
    out[r * 8 + 3] = quantize(w32((int64_t)tt1 + tt11 + tt13), q[3]);
    out[r * 8 + 5] = quantize(w32((int64_t)tt2 + tt11 + tt12), q[5]);
    out[r * 8 + 7] = quantize(w32((int64_t)tt3 + tt10 + tt13), q[7]);
    dcoffset = 0;
  }
}

/* forward L transformation of one pixel, FIX_BITS 13 -> COLOR_BITS 4 (ycbcrtrafo.cpp:176-199; numerics.hpp:61) */
static void rgb_to_ycc(int r, int g, int b, int32_t *y, int32_t *cb, int32_t *cr)
{
  const int64_t dc = ((int64_t)128) << 13, half = 256;
  int64_t yy = ((int64_t)r * 2449 + (int64_t)g * 4809 + (int64_t)b * 934 + half) >> 9;
  int64_t bb = ((int64_t)r * -1382 + (int64_t)g * -2714 + (int64_t)b * 4096 + dc + half) >> 9;
  int64_t rr = ((int64_t)r * 4096 + (int64_t)g * -3430 + (int64_t)b * -666 + dc + half) >> 9;
  const int64_t hi = (256 << 4) - 1;
  *y = (int32_t)(yy < 0 ? 0 : yy > hi ? hi : yy);
  *cb = (int32_t)(bb < 0 ? 0 : bb > hi ? hi : bb);
  *cr = (int32_t)(rr < 0 ? 0 : rr > hi ? hi : rr);
}

int oj_forward(const oj_info *info, const uint8_t *rgb, int use_ycbcr, int32_t *const planes[OJ_MAX_COMP])
{
  const int W = info->width, H = info->height, nc = info->ncomp;
  if ((nc != 1 && nc != 3) || info->precision != 8) return OJ_ERR_UNSUPPORTED;
  /* colour-transformed samples of the whole image at COLOR_BITS precision, one plane per component; line buffers of
   * subsampled components are width + 8 * subx long: beyond the image the last block column is mirrored
   * (downsamplerbase.cpp:141-145), lines beyond the image do not exist */
  int32_t *full[OJ_MAX_COMP] = {0, 0, 0, 0};
  int pitch[OJ_MAX_COMP];
  for (int c = 0; c < nc; c++) {
    pitch[c] = W + 8 * info->subx[c] + 8;
    full[c] = (int32_t *)calloc((size_t)pitch[c] * (size_t)(H + 8), sizeof(int32_t));
    if (!full[c]) return OJ_ERR_NOMEM;
  }
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const uint8_t *p = rgb + ((size_t)y * W + x) * nc;
      if (nc == 3 && use_ycbcr) rgb_to_ycc(p[0], p[1], p[2], &full[0][(size_t)y * pitch[0] + x], &full[1][(size_t)y * pitch[1] + x], &full[2][(size_t)y * pitch[2] + x]);
      else
        for (int c = 0; c < nc; c++) full[c][(size_t)y * pitch[c] + x] = (int32_t)p[c] << 4; /* INT_TO_COLOR; LUTs are identities */
    }
  for (int c = 0; c < nc; c++) {
    const int sx = info->subx[c], sy = info->suby[c];
    const int nbx = (info->cw[c] + 7) >> 3, nby = (info->ch[c] + 7) >> 3;
    memset(planes[c], 0, (size_t)info->bw[c] * info->bh[c] * 64 * sizeof(int32_t));
    if (sx == 1 && sy == 1) {
      /* straight into the transform; partial blocks are pre-filled with the level shift (ycbcrtrafo.cpp:100-113) */
      for (int by = 0; by < nby; by++)
        for (int bx = 0; bx < nbx; bx++) {
          int32_t blk[64];
          for (int i = 0; i < 64; i++) {
            const int x = bx * 8 + (i & 7), y = by * 8 + (i >> 3);
            blk[i] = (x < W && y < H) ? full[c][(size_t)y * pitch[c] + x] : (128 << 4);
          }
          oj_fdct_block(planes[c] + ((size_t)by * info->bw[c] + bx) * 64, blk, info->quant[info->tq[c]], 8);
        }
      continue;
    }
    /* the line buffers: DefineRegion copies whole block rows (the pre-fill included), then mirrors the right edge */
    const int ovl = (sx << 3) - 1;
    for (int y = 0; y < H; y++) {
      int32_t *line = full[c] + (size_t)y * pitch[c];
      for (int x = W; x < ((W + 7) & ~7); x++) line[x] = 128 << 4;
      for (int i = 0; i < ovl; i++) line[W + i] = line[W > i ? W - 1 - i : 0];
    }
    for (int by = 0; by < nby; by++)
      for (int bx = 0; bx < nbx; bx++) {
        int32_t blk[64];
        const int ofs = (bx * sx) << 3;
        int y = (by * sy) << 3;
        for (int r = 0; r < 8; r++) {
          int32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          int lines = 0;
          while (lines < sy && y < H) {
            const int32_t *src = full[c] + (size_t)y * pitch[c] + ofs;
            for (int i = 0; i < 8; i++)
              for (int k = 0; k < sx; k++) acc[i] += src[i * sx + k];
            lines++;
            y++;
          }
          const int norm = lines * sx;
          for (int i = 0; i < 8; i++) blk[r * 8 + i] = norm > 1 ? acc[i] / norm : acc[i];
        }
        oj_fdct_block(planes[c] + ((size_t)by * info->bw[c] + bx) * 64, blk, info->quant[info->tq[c]], 8);
      }
  }
  for (int c = 0; c < nc; c++) free(full[c]);
  return OJ_OK;
}

