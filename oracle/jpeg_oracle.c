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
 * Header parsing (marker/frame.cpp:111-..., marker/scan.cpp:163-..., marker/quantization.cpp:474-537,
 * coding/huffmantemplate.cpp:878-905, codestream/tables.cpp:1003-...).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int defined;
  uint8_t counts[16];
  uint8_t values[256];
  int nvalues;
  /* canonical decode tables, T.81 F.2.2.3 (equivalent to the two-level LUT the reference builds
   * in coding/huffmantemplate.cpp:802-874) */
  int32_t mincode[17], maxcode[18], valptr[17];
} oj_huff;

/* One JPEG XT box, reassembled from its APP11 segments (boxes/box.cpp:88-200). */
typedef struct {
  uint32_t type;
  uint16_t en;
  uint8_t *data;
  size_t len, cap;
} oj_box;

#define OJ_MAX_BOXES 64
typedef struct {
  const uint8_t *data;
  size_t len;
  oj_info *info;
  oj_huff dc[4], ac[4];
  int restart_interval;
  int have_frame;
  int need_dnl; /* SOF carried zero lines */
  int progressive; /* SOF2 */
  int hidden;      /* JPEG XT: bits of every coefficient that travel in hidden refinement scans (RSPC box) */
  oj_box *boxes; /* optional: where APP11 boxes are collected (OJ_MAX_BOXES entries) */
  int nboxes;
} oj_parser;

static int rd16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

static void huff_build(oj_huff *h)
{
  int l, code = 0, k = 0;
  for (l = 1; l <= 16; l++) {
    h->valptr[l] = k;
    h->mincode[l] = code;
    code += h->counts[l - 1];
    k += h->counts[l - 1];
    h->maxcode[l] = h->counts[l - 1] ? code - 1 : -1;
    code <<= 1;
  }
  h->maxcode[17] = 0x7fffffff;
}

static int parse_dqt(oj_parser *ps, const uint8_t *p, int n)
{
  /* marker/quantization.cpp:474-537: Pq/Tq byte, then 64 entries in zig-zag order, stored
   * de-zigzagged (:502-527). */
  while (n > 0) {
    int pq = p[0] >> 4, tq = p[0] & 15, i;
    if (tq > 3 || pq > 1) return OJ_ERR_MALFORMED;
    if (n < 1 + 64 * (pq + 1)) return OJ_ERR_MALFORMED;
    for (i = 0; i < 64; i++) {
      int v = pq ? rd16(p + 1 + 2 * i) : p[1 + i];
      ps->info->quant[tq][g_scan_order[i]] = (uint16_t)v;
    }
    ps->info->quant_defined[tq] = 1;
    p += 1 + 64 * (pq + 1);
    n -= 1 + 64 * (pq + 1);
  }
  return OJ_OK;
}

static int parse_dht(oj_parser *ps, const uint8_t *p, int n)
{
  while (n > 0) {
    int tc = p[0] >> 4, th = p[0] & 15, i, total = 0;
    oj_huff *h;
    if (tc > 1 || th > 3 || n < 17) return OJ_ERR_MALFORMED;
    h = tc ? &ps->ac[th] : &ps->dc[th];
    for (i = 0; i < 16; i++) { h->counts[i] = p[1 + i]; total += p[1 + i]; }
    if (total > 256 || n < 17 + total) return OJ_ERR_MALFORMED;
    memcpy(h->values, p + 17, (size_t)total);
    h->nvalues = total;
    h->defined = 1;
    huff_build(h);
    p += 17 + total;
    n -= 17 + total;
  }
  return OJ_OK;
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
  }
}

static int parse_sof(oj_parser *ps, const uint8_t *p, int n)
{
  oj_info *f = ps->info;
  int c;
  if (n < 6) return OJ_ERR_MALFORMED;
  f->precision = p[0];
  f->height = rd16(p + 1);
  f->width = rd16(p + 3);
  f->ncomp = p[5];
  if (f->precision != 8 && f->precision != 12) return OJ_ERR_UNSUPPORTED;
  if (f->ncomp < 1 || f->ncomp > OJ_MAX_COMP || n < 6 + 3 * f->ncomp) return OJ_ERR_MALFORMED;
  if (f->width == 0) return OJ_ERR_MALFORMED;
  ps->need_dnl = f->height == 0; /* height arrives in a DNL marker behind the first scan (entropyparser.cpp:204-249) */
  f->hmax = f->vmax = 1;
  for (c = 0; c < f->ncomp; c++) {
    f->comp_id[c] = p[6 + 3 * c];
    f->hs[c] = p[7 + 3 * c] >> 4;
    f->vs[c] = p[7 + 3 * c] & 15;
    f->tq[c] = p[8 + 3 * c];
    if (f->hs[c] < 1 || f->hs[c] > 4 || f->vs[c] < 1 || f->vs[c] > 4 || f->tq[c] > 3)
      return OJ_ERR_MALFORMED;
    if (f->hs[c] > f->hmax) f->hmax = f->hs[c];
    if (f->vs[c] > f->vmax) f->vmax = f->vs[c];
  }
  for (c = 0; c < f->ncomp; c++) /* marker/component.cpp: subsampling = max / own; must divide evenly */
    if (f->hmax % f->hs[c] || f->vmax % f->vs[c]) return OJ_ERR_UNSUPPORTED;
  frame_geometry(f);
  ps->have_frame = 1;
  return OJ_OK;
}

/* Height from the DNL marker that must follow the entropy coded data of the first scan
 * (EntropyParser::ParseDNLMarker, codestream/entropyparser.cpp:204-249): FFDC, length 4, number of lines > 0. */
static int resolve_dnl(oj_parser *ps, const uint8_t *ecs, const uint8_t *end)
{
  const uint8_t *q = ecs;
  int h;
  while (q + 1 < end && !(q[0] == 0xff && q[1] != 0x00 && q[1] != 0xff && !(q[1] >= 0xd0 && q[1] <= 0xd7))) q++;
  if (q + 6 > end || q[1] != 0xdc) return OJ_ERR_MALFORMED; /* the reference then fails on the frame height as well */
  if (rd16(q + 2) != 4) return OJ_ERR_MALFORMED;
  h = rd16(q + 4);
  if (h == 0) return OJ_ERR_MALFORMED;
  ps->info->height = h;
  frame_geometry(ps->info);
  ps->need_dnl = 0;
  return OJ_OK;
}

/* ------------------------------------------------------------------------------------------
 * Bit reader: io/bitstream.cpp:56-118 (byte-stuffing flavour) + io/bitstream.hpp:106-210.
 * FF 00 -> FF; any other FF xx is a marker: the reader stays in front of it and hands out
 * zero bits from then on.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t *p, *end;
  uint32_t acc; /* MSB-first window */
  int nbits;
  int marker; /* sitting at a marker / end of data */
} oj_bits;

static void bits_init(oj_bits *b, const uint8_t *p, const uint8_t *end)
{
  b->p = p; b->end = end; b->acc = 0; b->nbits = 0; b->marker = 0;
}

static void bits_fill(oj_bits *b)
{
  while (b->nbits <= 24) {
    uint32_t c = 0;
    if (!b->marker) {
      if (b->p >= b->end) {
        b->marker = 1;
      } else if (b->p[0] == 0xff) {
        if (b->p + 1 < b->end && b->p[1] == 0x00) { c = 0xff; b->p += 2; }
        else b->marker = 1;
      } else {
        c = *b->p++;
      }
    }
    b->acc |= c << (24 - b->nbits);
    b->nbits += 8;
  }
}

static uint32_t bits_get(oj_bits *b, int n)
{
  uint32_t v;
  if (n == 0) return 0;
  if (b->nbits < n) bits_fill(b);
  v = b->acc >> (32 - n);
  b->acc <<= n;
  b->nbits -= n;
  return v;
}

/* coding/huffmandecoder.hpp:103-124 in its T.81 F.2.2.3 form. Returns -1 for an unassigned code. */
static int huff_get(oj_bits *b, const oj_huff *h)
{
  int32_t code = 0;
  int l;
  for (l = 1; l <= 16; l++) {
    code = (code << 1) | (int32_t)bits_get(b, 1);
    if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l])
      return h->values[h->valptr[l] + (code - h->mincode[l])];
  }
  return -1;
}

/* codestream/sequentialscan.cpp:678-773 for ScanStart=0, ScanStop=63, lowbit=0, non-residual,
 * non-progressive. */
static int decode_block(oj_bits *b, const oj_huff *dc, const oj_huff *ac, int32_t *prevdc,
                        int32_t *block)
{
  int s = huff_get(b, dc), k;
  int32_t diff = 0;
  if (s < 0) return OJ_ERR_MALFORMED;
  if (s > 0) {
    if (s > 15) return OJ_ERR_MALFORMED; /* :686-688 */
    diff = (int32_t)bits_get(b, s);
    if (diff < (1 << (s - 1))) diff += (int32_t)((-1L) * (1L << s)) + 1; /* :690-692 */
  }
  *prevdc += diff;
  block[0] = *prevdc;
  k = 1;
  do {
    int rs = huff_get(b, ac), r, ss;
    if (rs < 0) return OJ_ERR_MALFORMED;
    r = rs >> 4; ss = rs & 15;
    if (ss == 0) {
      if (r == 15) { k += 16; continue; } /* ZRL, :713-715 */
      if (r == 0) break;                  /* EOB, :718-722 */
      return OJ_ERR_MALFORMED;            /* :747-750 (EOB runs exist only in progressive mode) */
    }
    k += r;
    diff = (int32_t)bits_get(b, ss);
    if (diff < (1 << (ss - 1))) diff += (int32_t)((-1L) * (1L << ss)) + 1;
    if (k >= 64) return OJ_ERR_MALFORMED; /* :763-765 */
    block[g_scan_order[k]] = diff;
    k++;
  } while (k <= 63);
  return OJ_OK;
}

/* Progressive first passes: codestream/sequentialscan.cpp:678-773 with spectral selection [ss, se], point transform
 * al and EOB runs (`skip`, :716-722). */
static int decode_block_first(oj_bits *b, const oj_huff *dc, const oj_huff *ac, int32_t *prevdc, int32_t *block,
                              int ss, int se, int al, int *skip)
{
  if (ss == 0) {
    int s = huff_get(b, dc);
    int32_t diff = 0;
    if (s < 0 || s > 15) return OJ_ERR_MALFORMED;
    if (s) { diff = (int32_t)bits_get(b, s); if (diff < (1 << (s - 1))) diff += (int32_t)((-1L) * (1L << s)) + 1; }
    *prevdc += diff;
    block[0] = (int32_t)((uint32_t)*prevdc << al);
  }
  if (se) {
    if (*skip > 0) { (*skip)--; return OJ_OK; }
    {
      int k = ss ? ss : 1;
      do {
        int rs = huff_get(b, ac), r, s;
        int32_t diff;
        if (rs < 0) return OJ_ERR_MALFORMED;
        r = rs >> 4; s = rs & 15;
        if (s == 0) {
          if (r == 15) { k += 16; continue; }
          *skip = 1 << r;
          if (r) *skip |= (int)bits_get(b, r);
          (*skip)--;
          break;
        }
        k += r;
        diff = (int32_t)bits_get(b, s);
        if (diff < (1 << (s - 1))) diff += (int32_t)((-1L) * (1L << s)) + 1;
        if (k >= 64) return OJ_ERR_MALFORMED;
        block[g_scan_order[k]] = (int32_t)((uint32_t)diff << al);
        k++;
      } while (k <= se);
    }
  }
  return OJ_OK;
}

/* Successive approximation refinement: codestream/refinementscan.cpp:584-700 */
static int decode_block_refine(oj_bits *b, const oj_huff *ac, int32_t *block, int ss, int se, int al, int *skip)
{
  if (ss == 0) block[0] |= (int32_t)(bits_get(b, 1) << al);
  if (se) {
    int k = ss, run = 0;
    int32_t s = 0;
    int enter_at_start = 0;
    if (*skip > 0) { run = se - ss + 1; (*skip)--; }
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
        if (k == se) break;
      }
      enter_at_start = 0;
      {
        int rs = huff_get(b, ac), r;
        if (rs < 0) return OJ_ERR_MALFORMED;
        r = rs >> 4; s = rs & 15;
        if (s == 0) {
          if (r == 15) run = r;
          else {
            *skip = 1 << r;
            if (r) *skip |= (int)bits_get(b, r);
            (*skip)--;
            run = se - k + 1;
          }
        } else {
          if (s != 1) return OJ_ERR_MALFORMED; /* the reference warns and leaves the block unrefined */
          if (bits_get(b, 1) == 0) s = -s;
          run = r;
        }
      }
    } while (++k <= se);
  }
  return OJ_OK;
}

/* One scan: codestream/sequentialscan.cpp:381-428 (ParseMCU) driven row by row, restart handling as
 * in codestream/entropyparser.hpp:147-160 / entropyparser.cpp:117-135 (happy path only: a missing or
 * wrong RSTn is reported as OJ_ERR_MALFORMED instead of being resynchronised). */
static int decode_scan_ex(oj_parser *ps, const uint8_t *sos, int n, const uint8_t *ecs,
                          const uint8_t *end, int32_t *const planes[OJ_MAX_COMP],
                          const uint8_t **next, int hidden_scan);

static int decode_scan(oj_parser *ps, const uint8_t *sos, int n, const uint8_t *ecs,
                       const uint8_t *end, int32_t *const planes[OJ_MAX_COMP],
                       const uint8_t **next)
{
  return decode_scan_ex(ps, sos, n, ecs, end, planes, next, 0);
}

/* hidden_scan: the scan comes from a FINE / RFIN box (marker/scan.cpp:899-980): always a successive approximation
 * refinement of one bit, its Al counts from the true LSB.  Visible scans of a frame with hidden bits address the bits
 * above them: Al + hidden (marker/scan.cpp:353-437). */
static int decode_scan_ex(oj_parser *ps, const uint8_t *sos, int n, const uint8_t *ecs,
                          const uint8_t *end, int32_t *const planes[OJ_MAX_COMP],
                          const uint8_t **next, int hidden_scan)
{
  const int progressive = ps->progressive || hidden_scan || ps->hidden > 0;
  const oj_info *f = ps->info;
  int ns = sos[0], ci[OJ_MAX_COMP], td[OJ_MAX_COMP], ta[OJ_MAX_COMP], i, c;
  int32_t pred[OJ_MAX_COMP] = {0, 0, 0, 0};
  int mx, my, mcus_x, mcus_y, togo, rstn = 0;
  int ss, se, ah, al, skip[OJ_MAX_COMP] = {0, 0, 0, 0};
  oj_bits b;
  if (ns < 1 || ns > f->ncomp || n < 1 + 2 * ns + 3) return OJ_ERR_MALFORMED;
  for (i = 0; i < ns; i++) {
    int id = sos[1 + 2 * i];
    for (c = 0; c < f->ncomp; c++) if (f->comp_id[c] == id) break;
    if (c == f->ncomp) return OJ_ERR_MALFORMED;
    ci[i] = c; td[i] = sos[2 + 2 * i] >> 4; ta[i] = sos[2 + 2 * i] & 15;
    if (td[i] > 3 || ta[i] > 3) return OJ_ERR_MALFORMED;
  }
  ss = sos[1 + 2 * ns]; se = sos[2 + 2 * ns]; ah = sos[3 + 2 * ns] >> 4; al = sos[3 + 2 * ns] & 15;
  if (!ps->progressive && !hidden_scan) {
    if (ss != 0 || se != 63 || ah != 0 || al != 0) return OJ_ERR_MALFORMED;
  } else if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) {
    return OJ_ERR_MALFORMED; /* T.81 G.1.1.1.1 */
  }
  if (hidden_scan) { if (ah != al + 1) return OJ_ERR_MALFORMED; } /* "hidden refinement must refine by one bit per scan" */
  else al += ps->hidden;
  for (i = 0; i < ns; i++) {
    if ((ss == 0 && ah == 0 && !ps->dc[td[i]].defined) || (se > 0 && !ps->ac[ta[i]].defined)) return OJ_ERR_MALFORMED;
  }
  if (ns > 1) { mcus_x = f->mcus_x; mcus_y = f->mcus_y; }
  else {
    /* single-component scan: 1x1 MCUs over ceil(cw/8) x ceil(ch/8) blocks
     * (sequentialscan.cpp:396-397; marker/component.cpp) */
    mcus_x = (f->cw[ci[0]] + 7) >> 3; mcus_y = (f->ch[ci[0]] + 7) >> 3;
  }
  bits_init(&b, ecs, end);
  togo = ps->restart_interval;
  for (my = 0; my < mcus_y; my++) {
    for (mx = 0; mx < mcus_x; mx++) {
      if (ps->restart_interval) {
        if (togo == 0) {
          /* entropyparser.cpp:117-135: skip FF fillers, require RSTn, reset predictors and the bit
           * buffer (sequentialscan.cpp:266-274) */
          const uint8_t *p = b.p;
          while (p + 1 < end && p[0] == 0xff && p[1] == 0xff) p++;
          if (p + 1 >= end || p[0] != 0xff || p[1] != 0xd0 + rstn) return OJ_ERR_MALFORMED;
          rstn = (rstn + 1) & 7;
          bits_init(&b, p + 2, end);
          for (i = 0; i < OJ_MAX_COMP; i++) { pred[i] = 0; skip[i] = 0; }
          togo = ps->restart_interval;
        }
        togo--;
      }
      for (i = 0; i < ns; i++) {
        int bx, by, w = (ns > 1) ? f->hs[ci[i]] : 1, h = (ns > 1) ? f->vs[ci[i]] : 1;
        c = ci[i];
        for (by = 0; by < h; by++)
          for (bx = 0; bx < w; bx++) {
            int32_t dummy[64];
            int X = mx * w + bx, Y = my * h + by, rc;
            int32_t *blk = (X < f->bw[c] && Y < f->bh[c]) ? planes[c] + ((size_t)Y * f->bw[c] + X) * 64
                                                          : dummy;
            if (!progressive) rc = decode_block(&b, &ps->dc[td[i]], &ps->ac[ta[i]], &pred[i], blk);
            else if (ah == 0) rc = decode_block_first(&b, &ps->dc[td[i]], &ps->ac[ta[i]], &pred[i], blk, ss, se, al, &skip[i]);
            else rc = decode_block_refine(&b, &ps->ac[ta[i]], blk, ss, se, al, &skip[i]);
            if (rc) return rc;
          }
      }
    }
  }
  /* advance to the next marker */
  {
    const uint8_t *p = b.p;
    while (p + 1 < end && !(p[0] == 0xff && p[1] != 0x00 && p[1] != 0xff)) p++;
    *next = p;
  }
  return OJ_OK;
}

static int walk(oj_parser *ps, int32_t *const planes[OJ_MAX_COMP])
{
  const uint8_t *p = ps->data, *end = ps->data + ps->len;
  oj_info *f = ps->info;
  int rc;
  if (!g_scan_order_ready) build_scan_order();
  if (ps->len < 4 || p[0] != 0xff || p[1] != 0xd8) return OJ_ERR_MALFORMED;
  p += 2;
  f->adobe_transform = -1;
  for (;;) {
    int m, n;
    if (p + 2 > end) return OJ_ERR_EOF;
    if (p[0] != 0xff) return OJ_ERR_MALFORMED;
    while (p + 1 < end && p[1] == 0xff) p++; /* fill bytes */
    m = p[1];
    p += 2;
    if (m == 0xd9) break; /* EOI */
    if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
    if (p + 2 > end) return OJ_ERR_EOF;
    n = rd16(p);
    if (n < 2 || p + n > end) return OJ_ERR_EOF;
    switch (m) {
    case 0xdb: rc = parse_dqt(ps, p + 2, n - 2); if (rc) return rc; break;
    case 0xc4: rc = parse_dht(ps, p + 2, n - 2); if (rc) return rc; break;
    case 0xdd: if (n < 4) return OJ_ERR_MALFORMED; ps->restart_interval = rd16(p + 2); break;
    case 0xc0: case 0xc1: case 0xc2: /* baseline, extended sequential, progressive (Huffman) */
      if (ps->have_frame) return OJ_ERR_MALFORMED;
      ps->progressive = m == 0xc2;
      rc = parse_sof(ps, p + 2, n - 2); if (rc) return rc; break;
    case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb:
    case 0xcd: case 0xce: case 0xcf:
      return OJ_ERR_UNSUPPORTED;
    case 0xeb: /* APP11 "JP": one segment of a box: en(2) z(4) lbox(4) tbox(4) [xlbox(8)] payload; boxes/box.cpp:88-150 */
      if (ps->boxes && n >= 2 + 2 + 2 + 4 + 4 + 4 && p[2] == 0x4a && p[3] == 0x50) {
        const uint8_t *q = p + 4;
        uint16_t en = (uint16_t)rd16(q);
        uint32_t lbox = ((uint32_t)rd16(q + 6) << 16) | (uint32_t)rd16(q + 8);
        uint32_t tbox = ((uint32_t)rd16(q + 10) << 16) | (uint32_t)rd16(q + 12);
        const uint8_t *pay = q + 14;
        int blen = n - 2 - 2 - 2 - 4 - 4 - 4, b;
        if (lbox == 1) { pay += 8; blen -= 8; }
        if (blen < 0) return OJ_ERR_MALFORMED;
        for (b = 0; b < ps->nboxes; b++)
          if (ps->boxes[b].type == tbox && ps->boxes[b].en == en) break;
        if (b == ps->nboxes) {
          if (ps->nboxes == OJ_MAX_BOXES) return OJ_ERR_UNSUPPORTED;
          memset(&ps->boxes[b], 0, sizeof(oj_box));
          ps->boxes[b].type = tbox; ps->boxes[b].en = en;
          ps->nboxes++;
        }
        if (ps->boxes[b].len + (size_t)blen > ps->boxes[b].cap) {
          size_t cap = (ps->boxes[b].len + (size_t)blen) * 2 + 64;
          uint8_t *nd = (uint8_t *)realloc(ps->boxes[b].data, cap);
          if (!nd) return OJ_ERR_NOMEM;
          ps->boxes[b].data = nd; ps->boxes[b].cap = cap;
        }
        memcpy(ps->boxes[b].data + ps->boxes[b].len, pay, (size_t)blen); /* segments arrive in sequence order */
        ps->boxes[b].len += (size_t)blen;
      }
      break;
    case 0xee: /* APP14 Adobe: marker/adobemarker.cpp; "Adobe" + version(2) flags0(2) flags1(2) transform(1) */
      if (n >= 14 && memcmp(p + 2, "Adobe", 5) == 0) f->adobe_transform = p[13];
      break;
    case 0xda: {
      const uint8_t *next = NULL;
      if (!ps->have_frame) return OJ_ERR_MALFORMED;
      f->restart_interval = ps->restart_interval;
      if (ps->need_dnl) { rc = resolve_dnl(ps, p + n, end); if (rc) return rc; }
      if (!planes && !ps->boxes) goto done; /* header-only walk stops at the first scan (unless boxes are wanted) */
      if (!planes) { /* skip the entropy coded data */
        const uint8_t *q = p + n;
        while (q + 1 < end && !(q[0] == 0xff && q[1] != 0x00 && q[1] != 0xff && !(q[1] >= 0xd0 && q[1] <= 0xd7))) q++;
        p = q;
        continue;
      }
      rc = decode_scan(ps, p + 2, n - 2, p + n, end, planes, &next);
      if (rc) return rc;
      p = next;
      continue;
    }
    default: break; /* APPn, COM, ...: skipped */
    }
    p += n;
  }
done:
  if (!ps->have_frame) return OJ_ERR_MALFORMED;
  /* codestream/tables.cpp:2021-2030: three components and no Adobe "None" -> YCbCr, else identity */
  f->ycbcr = (f->ncomp == 3 && f->adobe_transform != 0) ? 1 : 0;
  return OJ_OK;
}

int oj_read_info(const uint8_t *data, size_t len, oj_info *info)
{
  oj_parser ps;
  memset(&ps, 0, sizeof(ps));
  memset(info, 0, sizeof(*info));
  ps.data = data; ps.len = len; ps.info = info;
  return walk(&ps, NULL);
}

int oj_decode_coefficients(const uint8_t *data, size_t len, const oj_info *info,
                           int32_t *const planes[OJ_MAX_COMP])
{
  oj_parser ps;
  oj_info tmp;
  int c;
  memset(&ps, 0, sizeof(ps));
  memset(&tmp, 0, sizeof(tmp));
  ps.data = data; ps.len = len; ps.info = &tmp;
  for (c = 0; c < info->ncomp; c++)
    memset(planes[c], 0, (size_t)info->bw[c] * info->bh[c] * 64 * sizeof(int32_t));
  return walk(&ps, planes);
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

void oj_upsample_block(int32_t out[64], const int32_t *plane, int pitch, int cw, int ch, int sx,
                       int sy, int X0, int Y0)
{
  /* upsampler.cpp:83-117 */
  int y = Y0 / sy, x = X0 / sx + 1, l, j;
  int top = y > 0 ? y - 1 : 0, cur = y, bot;
  int ymod = Y0 % sy, xmod = X0 % sx;
  int32_t *target = out;
  if (cur > ch - 1) cur = ch - 1; /* blocks entirely below the last stored line are never output */
  if (top > ch - 1) top = ch - 1;
  bot = cur + 1 < ch ? cur + 1 : cur;
  if (sx > 1) x--;
#define T(j) line_at(plane, pitch, cw, top, x + (j))
#define C(j) line_at(plane, pitch, cw, cur, x + (j))
#define B(j) line_at(plane, pitch, cw, bot, x + (j))
#define ADVANCE() do { top = cur; cur = bot; if (bot + 1 < ch) bot++; } while (0)
  for (l = 0; l < 8; l++, target += 8) {
    switch (sy) {
    case 1: /* :118-131 */
      for (j = 0; j < 8; j++) target[j] = C(j);
      if (cur + 1 < ch) cur++;
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
  int ltrafo_ycbcr, rtrafo_ycbcr;   /* 1: YCbCr matrix, 0: identity */
  int64_t outmax, outshift;         /* 2^(8 + extra bits) - 1 and its half */
  int is_float, clamp;              /* OCON: cast to float (half codes), clamping */
} oj_xt;

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
    if (!f->quant_defined[f->tq[c]]) { rc = OJ_ERR_MALFORMED; goto out; }
    samp[c] = (int32_t *)malloc((size_t)f->bw[c] * f->bh[c] * 64 * sizeof(int32_t));
    if (!samp[c]) { rc = OJ_ERR_NOMEM; goto out; }
    oj_idct_plane(samp[c], planes[c], f->bw[c], f->bh[c], f->quant[f->tq[c]], f->precision);
    if (xt) { /* residual: same transform, level shift 2^(Pr-1) (control/residualblockhelper.cpp:191-202) */
      const oj_info *r = xt->rinfo;
      if (!r->quant_defined[r->tq[c]]) { rc = OJ_ERR_MALFORMED; goto out; }
      rsamp[c] = (int32_t *)malloc((size_t)r->bw[c] * r->bh[c] * 64 * sizeof(int32_t));
      if (!rsamp[c]) { rc = OJ_ERR_NOMEM; goto out; }
      oj_idct_plane(rsamp[c], xt->rplanes[c], r->bw[c], r->bh[c], r->quant[r->tq[c]], r->precision);
    }
  }
  for (Y0 = 0; Y0 < f->height; Y0 += 8)
    for (X0 = 0; X0 < f->width; X0 += 8) {
      int32_t blk[OJ_MAX_COMP][64], rblk[OJ_MAX_COMP][64];
      for (c = 0; c < f->ncomp; c++) {
        oj_upsample_block(blk[c], samp[c], f->bw[c] * 8, f->cw[c], f->ch[c], f->subx[c], f->suby[c], X0, Y0);
        if (xt) {
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
            int64_t yy = blk[0][k], cb = (int64_t)blk[1][k] - dcshift, cr = (int64_t)blk[2][k] - dcshift;
            v[0] = (yy * L[0] + cb * L[1] + cr * L[2] + 65536) >> 17;
            v[1] = (yy * L[3] + cb * L[4] + cr * L[5] + 65536) >> 17;
            v[2] = (yy * L[6] + cb * L[7] + cr * L[8] + 65536) >> 17;
          } else {
            for (c = 0; c < f->ncomp; c++) v[c] = ((int64_t)blk[c][k] + 8) >> 4;
          }
          if (xt) {
            /* colortrafo/ycbcrtrafo.cpp:750-829 (residual), :861-878 (L-LUT, C = identity, merge), :897-955 (half clamp) */
            const oj_info *r = xt->rinfo;
            const int64_t rmax16 = ((((int64_t)1 << r->precision)) << 4) - 1; /* ((m_lRMax + 1) << COLOR_BITS) - 1 */
            const int64_t omax16 = ((xt->outmax + 1) << 4) - 1;
            int64_t rr[3];
            /* Q tables: identity, 2^(Pr + 4) -> 2^(16 + 4): scales by 2^(16 - Pr)  (parametrictonemappingbox.cpp:387-430) */
            int64_t ry = clampmax(rblk[0][k], rmax16) << (16 - r->precision);
            int64_t rcb = clampmax(rblk[1][k], rmax16) << (16 - r->precision);
            int64_t rcr = clampmax(rblk[2][k], rmax16) << (16 - r->precision);
            if (xt->rtrafo_ycbcr) {
              rcb -= xt->outshift << 4; rcr -= xt->outshift << 4;
              rr[0] = (ry * L[0] + rcb * L[1] + rcr * L[2] + 4096) >> 13; /* FIX_COLOR_TO_INTCOLOR */
              rr[1] = (ry * L[3] + rcb * L[4] + rcr * L[5] + 4096) >> 13;
              rr[2] = (ry * L[6] + rcb * L[7] + rcr * L[8] + 4096) >> 13;
            } else {
              rr[0] = ry; rr[1] = rcb; rr[2] = rcr;
            }
            /* R2 tables: identity 2^(16 + 4) -> 2^16: floor(x / 16 + 0.5) */
            for (c = 0; c < 3; c++) rr[c] = (clampmax(rr[c], omax16) + 8) >> 4;
            for (c = 0; c < 3; c++) {
              int64_t lv = xt->ltable[c] ? xt->ltable[c][clampmax(v[c], maxval)] : v[c];
              v[c] = lv + rr[c] - xt->outshift; /* C transformation = identity: FIX_TO_INT(x * 8192) == x */
            }
            if (xt->is_float && xt->clamp) {
              const int64_t pinf = (xt->outmax >> 1) - (xt->outmax >> 6) - 1;
              const int64_t minf = invert_negs((int16_t)(uint16_t)(pinf | 0x8000));
              for (c = 0; c < 3; c++) {
                int64_t t = v[c] > pinf ? pinf : (v[c] < minf ? minf : v[c]);
                pixels16[pix + c] = (uint16_t)invert_negs((int16_t)t);
              }
            } else {
              for (c = 0; c < 3; c++) pixels16[pix + c] = (uint16_t)clampmax(v[c], xt->outmax);
            }
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
  int en;
  for (en = 0;; en++) {
    const oj_box *bx = NULL;
    const uint8_t *p, *end, *next;
    int b, rc, done = 0;
    for (b = 0; b < nboxes; b++) if (boxes[b].type == type && boxes[b].en == en) bx = &boxes[b];
    if (!bx) return OJ_OK;
    p = bx->data; end = bx->data + bx->len;
    while (!done) {
      int m, n;
      if (p + 4 > end || p[0] != 0xff) return OJ_ERR_MALFORMED;
      m = p[1]; n = rd16(p + 2);
      if (n < 2 || p + 2 + n > end) return OJ_ERR_MALFORMED;
      switch (m) {
      case 0xc4: rc = parse_dht(ps, p + 4, n - 2); if (rc) return rc; break;
      case 0xdb: rc = parse_dqt(ps, p + 4, n - 2); if (rc) return rc; break;
      case 0xdd: if (n < 4) return OJ_ERR_MALFORMED; ps->restart_interval = rd16(p + 4); break;
      case 0xda:
        rc = decode_scan_ex(ps, p + 4, n - 2, p + 2 + n, end, planes, &next, 1);
        if (rc) return rc;
        done = 1;
        break;
      default: break;
      }
      p += 2 + n;
    }
  }
}

int oj_decode_xt(const uint8_t *data, size_t len, oj_info *info, uint16_t **pixels, int *is_float)
{
  oj_box boxes[OJ_MAX_BOXES];
  oj_parser ps;
  oj_info rinfo;
  oj_xt xt;
  int32_t *planes[OJ_MAX_COMP] = {0, 0, 0, 0}, *rplanes[OJ_MAX_COMP] = {0, 0, 0, 0};
  int32_t *tables[16];
  int tabsize[16] = {0};
  int hidden_l = 0, hidden_r = 0; /* RSPC: bits of the legacy / residual coefficients in hidden refinement scans */
  int32_t *identity = NULL;
  const oj_box *spec = NULL, *resi = NULL;
  int ltrafo = 255, rtrafo = 255, ctrafo = 255, lidx[4] = {255, 255, 255, 255}, have_lpts = 0;
  int ocon = -1, b, c, rc;
  size_t j;
  *pixels = NULL;
  memset(&ps, 0, sizeof(ps)); memset(info, 0, sizeof(*info)); memset(tables, 0, sizeof(tables));
  ps.data = data; ps.len = len; ps.info = info; ps.boxes = boxes;
  rc = walk(&ps, NULL);
  if (rc) { free_boxes(boxes, ps.nboxes); return rc; }
  for (b = 0; b < ps.nboxes; b++) {
    if (boxes[b].type == BOXID('S', 'P', 'E', 'C')) spec = &boxes[b];
    if (boxes[b].type == BOXID('R', 'E', 'S', 'I')) resi = &boxes[b];
    if (boxes[b].type == BOXID('T', 'O', 'N', 'E')) {
      /* boxes/inversetonemappingbox.cpp: index/residual-bits byte, then 2^n 16-bit entries (n = 8 + hidden bits) */
      const oj_box *t = &boxes[b];
      int idx, n, i;
      if (t->len < 1 + 512 || !(t->len & 1)) { rc = OJ_ERR_MALFORMED; goto out; }
      idx = t->data[0] >> 4; n = (int)((t->len - 1) >> 1);
      if ((t->data[0] & 15) > 8 || n < 256 || n > 4096 || (n & (n - 1))) { rc = OJ_ERR_UNSUPPORTED; goto out; }
      tabsize[idx] = n;
      tables[idx] = (int32_t *)malloc((size_t)n * sizeof(int32_t));
      if (!tables[idx]) { rc = OJ_ERR_NOMEM; goto out; }
      for (i = 0; i < n; i++) tables[idx][i] = rd16(t->data + 1 + 2 * i);
    }
  }
  if (!spec || !resi || info->ncomp != 3 || info->precision != 8) { rc = OJ_ERR_UNSUPPORTED; goto out; }
  for (j = 0; j + 8 <= spec->len;) { /* superbox: LBox(4) TBox(4) payload (boxes/superbox.cpp) */
    uint32_t l = ((uint32_t)rd16(spec->data + j) << 16) | (uint32_t)rd16(spec->data + j + 2);
    uint32_t t = ((uint32_t)rd16(spec->data + j + 4) << 16) | (uint32_t)rd16(spec->data + j + 6);
    const uint8_t *pl = spec->data + j + 8;
    if (l < 8 || j + l > spec->len) { rc = OJ_ERR_MALFORMED; goto out; }
    if (t == BOXID('L', 'T', 'R', 'F')) ltrafo = pl[0] >> 4;
    else if (t == BOXID('R', 'T', 'R', 'F')) rtrafo = pl[0] >> 4;
    else if (t == BOXID('C', 'T', 'R', 'F')) ctrafo = pl[0] >> 4;
    else if (t == BOXID('L', 'P', 'T', 'S')) { have_lpts = 1; lidx[0] = pl[0] >> 4; lidx[1] = pl[0] & 15; lidx[2] = pl[1] >> 4; lidx[3] = pl[1] & 15; }
    else if (t == BOXID('O', 'C', 'O', 'N')) ocon = pl[0];
    else if (t == BOXID('R', 'S', 'P', 'C')) { /* boxes/refinementspecbox.cpp:57-83 */
      hidden_l = pl[0] >> 4; hidden_r = pl[0] & 15;
      if (hidden_l > 4 || hidden_r > 4) { rc = OJ_ERR_MALFORMED; goto out; }
    }
    else if (t == BOXID('L', 'D', 'C', 'T') || t == BOXID('R', 'D', 'C', 'T')) { if ((pl[0] >> 4) != 0 && (pl[0] >> 4) != 2) { rc = OJ_ERR_UNSUPPORTED; goto out; } }
    else { rc = OJ_ERR_UNSUPPORTED; goto out; } /* Q/R/R2/S tables, D/S transformations, ...: outside the subset */
    j += l;
  }
  /* codestream/tables.cpp:1994-2031 and the R analogue: undefined -> YCbCr for three components */
  if (ltrafo == 255) ltrafo = 2;
  if (rtrafo == 255) rtrafo = 2;
  if ((ltrafo != 1 && ltrafo != 2) || (rtrafo != 1 && rtrafo != 2) || (ctrafo != 255 && ctrafo != 1)) { rc = OJ_ERR_UNSUPPORTED; goto out; }
  if (ocon < 0 || (ocon & 0x08) || (ocon & 0x01)) { rc = OJ_ERR_UNSUPPORTED; goto out; } /* lossless / output lookup */
  memset(&xt, 0, sizeof(xt));
  xt.outmax = ((int64_t)1 << (8 + (ocon >> 4))) - 1;
  xt.outshift = (xt.outmax + 1) >> 1;
  xt.is_float = (ocon & 0x04) ? 1 : 0;
  xt.clamp = (ocon & 0x02) ? 1 : 0;
  if (!xt.clamp || xt.outmax != 65535) { rc = OJ_ERR_UNSUPPORTED; goto out; }
  xt.ltrafo_ycbcr = ltrafo == 2; xt.rtrafo_ycbcr = rtrafo == 2;
  for (c = 0; c < 3; c++) {
    const int entries = 256 << hidden_l; /* the L table is indexed with 8 + hidden bits (codestream/tables.cpp:549) */
    if (have_lpts) {
      if (!tables[lidx[c]]) { rc = OJ_ERR_MALFORMED; goto out; } /* "the L lookup table specified in the codestream does not exist" */
      if (tabsize[lidx[c]] != entries) { rc = OJ_ERR_MALFORMED; goto out; }
      xt.ltable[c] = tables[lidx[c]];
    } else { /* identity, e = 1: floor((2^16 - 1) * (i / (2^n - 1)) + 0.5), = 257 i for n = 8 */
      if (!identity) {
        int i;
        identity = (int32_t *)malloc((size_t)entries * sizeof(int32_t));
        if (!identity) { rc = OJ_ERR_NOMEM; goto out; }
        for (i = 0; i < entries; i++) identity[i] = (int32_t)floor(65535.0 * ((double)i / (double)(entries - 1)) + 0.5);
      }
      xt.ltable[c] = identity;
    }
  }
  /* the residual codestream is an ordinary codestream of its own (codestream/image.cpp:1264-1300) */
  rc = oj_read_info(resi->data, resi->len, &rinfo);
  if (rc) goto out;
  if (rinfo.width != info->width || rinfo.height != info->height || rinfo.ncomp != info->ncomp) { rc = OJ_ERR_MALFORMED; goto out; }
  if (rinfo.precision + hidden_r > 16) { rc = OJ_ERR_UNSUPPORTED; goto out; }
  info->ycbcr = xt.ltrafo_ycbcr;
  for (c = 0; c < 3; c++) {
    planes[c] = (int32_t *)malloc((size_t)info->bw[c] * info->bh[c] * 64 * sizeof(int32_t));
    rplanes[c] = (int32_t *)malloc((size_t)rinfo.bw[c] * rinfo.bh[c] * 64 * sizeof(int32_t));
    if (!planes[c] || !rplanes[c]) { rc = OJ_ERR_NOMEM; goto out; }
  }
  {
    /* visible scans with their bits moved up by the hidden ones, then the hidden refinement scans; from here on both
     * frames simply have precision + hidden bits (Frame::HiddenPrecisionOf, marker/frame.cpp:368-373) */
    oj_parser ls, rs;
    oj_info ltmp, rtmp;
    memset(&ls, 0, sizeof(ls)); memset(&rs, 0, sizeof(rs)); memset(&ltmp, 0, sizeof(ltmp)); memset(&rtmp, 0, sizeof(rtmp));
    ls.data = data; ls.len = len; ls.info = &ltmp; ls.hidden = hidden_l;
    rs.data = resi->data; rs.len = resi->len; rs.info = &rtmp; rs.hidden = hidden_r;
    for (c = 0; c < 3; c++) {
      memset(planes[c], 0, (size_t)info->bw[c] * info->bh[c] * 64 * sizeof(int32_t));
      memset(rplanes[c], 0, (size_t)rinfo.bw[c] * rinfo.bh[c] * 64 * sizeof(int32_t));
    }
    rc = walk(&ls, planes);
    if (!rc) rc = decode_hidden_scans(&ls, boxes, ps.nboxes, BOXID('F', 'I', 'N', 'E'), planes);
    if (!rc) rc = walk(&rs, rplanes);
    if (!rc) rc = decode_hidden_scans(&rs, boxes, ps.nboxes, BOXID('R', 'F', 'I', 'N'), rplanes);
    if (rc) goto out;
    info->precision += hidden_l;
    rinfo.precision += hidden_r;
  }
  xt.rinfo = &rinfo; xt.rplanes = rplanes;
  *pixels = (uint16_t *)malloc((size_t)info->width * info->height * 3 * sizeof(uint16_t));
  if (!*pixels) { rc = OJ_ERR_NOMEM; goto out; }
  rc = reconstruct_ex(info, planes, NULL, *pixels, xt.ltrafo_ycbcr, &xt);
  if (rc) { free(*pixels); *pixels = NULL; }
  if (is_float) *is_float = xt.is_float;
out:
  for (c = 0; c < OJ_MAX_COMP; c++) { free(planes[c]); free(rplanes[c]); }
  for (c = 0; c < 16; c++) free(tables[c]);
  free(identity);
  free_boxes(boxes, ps.nboxes);
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

