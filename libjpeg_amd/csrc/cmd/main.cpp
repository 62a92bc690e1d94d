// main.cpp -- `jpeg` command line front end of the MI355X path: the decode half of the reference CLI.
//   jpeg [-c] [-U] [-al alpha.pgm] [-t threads] [-d device] in.jpg out.ppm
// reproduces cmd/main.cpp:746-747 -> cmd/reconstruct.cpp:68-376 for the streams this path handles: the
// image (8 bit -> PNM, 12 bit -> 16-bit PNM, JPEG XT profile C -> PFM) is reconstructed stripe by stripe (eight lines per JPEG::DisplayRectangle call) through a file I/O
// hook and a bitmap hook, and written as binary PNM (P6 for three components, P5 for one) -- byte for
// byte what the reference binary writes.  -c disables the colour transformation (reference: -c); -U disables the
// upsampling (reference: -U): like images with neither one nor three components, the components then go one by one
// into PGX files (out_N.h + out_N.raw, listed in `out`), cmd/reconstruct.cpp:208-306.  -al names the file the alpha channel of a
// JPEG XT file goes to (PGM / 16-bit PGM / one-channel PFM), cmd/reconstruct.cpp:154-217, 319-342: no compositing, like the reference.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/mijpeg.h"
#include "../interface/hooks.hpp"
#include "../interface/jpeg.hpp"
#include "../interface/parameters.hpp"
#include "../interface/tagitem.hpp"

// I/O hook: serves JPGFLAG_ACTION_READ/SEEK/QUERY from a FILE (cmd/filehook.cpp:59-99)
static JPG_LONG FileHook(struct JPG_Hook *hook, struct JPG_TagItem *tags)
{
  FILE *fp = (FILE *)hook->hk_pData;
  switch (tags->GetTagData(JPGTAG_FIO_ACTION)) {
  case JPGFLAG_ACTION_READ:
    return (JPG_LONG)fread(tags->GetTagPtr(JPGTAG_FIO_BUFFER), 1, (size_t)tags->GetTagData(JPGTAG_FIO_SIZE), fp);
  case JPGFLAG_ACTION_WRITE:
    return (JPG_LONG)fwrite(tags->GetTagPtr(JPGTAG_FIO_BUFFER), 1, (size_t)tags->GetTagData(JPGTAG_FIO_SIZE), fp);
  case JPGFLAG_ACTION_SEEK: {
    const JPG_LONG off = tags->GetTagData(JPGTAG_FIO_OFFSET);
    const JPG_LONG mode = tags->GetTagData(JPGTAG_FIO_SEEKMODE);
    return fseek(fp, off, mode == JPGFLAG_OFFSET_BEGINNING ? SEEK_SET : mode == JPGFLAG_OFFSET_END ? SEEK_END : SEEK_CUR);
  }
  case JPGFLAG_ACTION_QUERY: return 0;
  }
  return -1;
}

// Bitmap hook state: one interleaved stripe of eight lines (cmd/bitmaphook.cpp:102-340)
struct StripeBuffer {
  unsigned char *mem;
  unsigned width, height, depth;
  unsigned bytes; // per sample: 1 (8 bit) or 2 (12 bit, JPEG XT half-float codes)
  bool halffloat; // 16-bit samples are half-float codes: expanded to big-endian float32 when written (PFM)
  bool upsampling; // false: the hook works on the component's own sample grid (JPGTAG_BIO_PIXEL_*)
  bool pgx;        // plane by plane into pgxfiles[] instead of stripe by stripe into target
  FILE *target;
  FILE *pgxfiles[4];
};

// cmd/iohelpers.hpp:60-77: exact expansion of a half-float bit pattern
static float HalfToFloat(unsigned short h)
{
  const int exponent = (h >> 10) & 31, mantissa = h & 1023;
  double v;
  if (exponent == 0) v = ldexp((double)mantissa, -14 - 10);
  else if (exponent == 31) v = HUGE_VAL;
  else v = ldexp((double)(mantissa | 1024), -15 - 10 + exponent);
  return (float)((h & 0x8000) ? -v : v);
}

static JPG_LONG BitmapHook(struct JPG_Hook *hook, struct JPG_TagItem *tags)
{
  StripeBuffer *sb = (StripeBuffer *)hook->hk_pData;
  const unsigned comp = (unsigned)tags->GetTagData(JPGTAG_BIO_COMPONENT);
  // cmd/bitmaphook.cpp:106-108
  const unsigned miny = (unsigned)tags->GetTagData(sb->upsampling ? JPGTAG_BIO_MINY : JPGTAG_BIO_PIXEL_MINY);
  const unsigned maxy = (unsigned)tags->GetTagData(sb->upsampling ? JPGTAG_BIO_MAXY : JPGTAG_BIO_PIXEL_MAXY);
  const unsigned width = 1 + (unsigned)tags->GetTagData(sb->upsampling ? JPGTAG_BIO_MAXX : JPGTAG_BIO_PIXEL_MAXX);
  switch (tags->GetTagData(JPGTAG_BIO_ACTION)) {
  case JPGFLAG_BIO_REQUEST:
    // address of canvas pixel (0,0) of this component: the stripe buffer starts at line miny
    tags->SetTagPtr(JPGTAG_BIO_MEMORY, sb->mem + ((size_t)comp - (size_t)miny * sb->depth * width) * sb->bytes);
    tags->SetTagData(JPGTAG_BIO_WIDTH, width);
    tags->SetTagData(JPGTAG_BIO_HEIGHT, 8 + miny);
    tags->SetTagData(JPGTAG_BIO_BYTESPERROW, sb->depth * width * sb->bytes);
    tags->SetTagData(JPGTAG_BIO_BYTESPERPIXEL, sb->depth * sb->bytes);
    tags->SetTagData(JPGTAG_BIO_PIXELTYPE, sb->bytes == 2 ? CTYP_UWORD : CTYP_UBYTE);
    break;
  case JPGFLAG_BIO_RELEASE:
    if (sb->pgx) { // cmd/bitmaphook.cpp:254, 308-326: PGX is plane-interleaved, samples big-endian
      const size_t n = (size_t)width * (maxy + 1 - miny);
      FILE *out = sb->pgxfiles[comp];
      for (size_t i = 0; i < n; i++) {
        const unsigned char *px = sb->mem + (i * sb->depth + comp) * sb->bytes;
        if (sb->bytes == 2) {
          unsigned short v;
          memcpy(&v, px, 2);
          if (fputc(v >> 8, out) == EOF || fputc(v & 0xff, out) == EOF) return JPGERR_UNEXPECTED_EOF;
        } else if (fputc(px[0], out) == EOF) return JPGERR_UNEXPECTED_EOF;
      }
    } else if (comp == sb->depth - 1) { // all components of the stripe are in: write it
      const size_t n = (size_t)width * (maxy + 1 - miny) * sb->depth;
      if (sb->bytes == 1) {
        if (fwrite(sb->mem, 1, n, sb->target) != n) return JPGERR_UNEXPECTED_EOF;
      } else {
        const unsigned short *px = (const unsigned short *)sb->mem;
        for (size_t i = 0; i < n; i++) {
          if (sb->halffloat) { // cmd/bitmaphook.cpp:282-305: big-endian IEEE single
            const float v = HalfToFloat(px[i]);
            unsigned int bits;
            memcpy(&bits, &v, 4);
            const unsigned char be[4] = {(unsigned char)(bits >> 24), (unsigned char)(bits >> 16), (unsigned char)(bits >> 8), (unsigned char)bits};
            if (fwrite(be, 1, 4, sb->target) != 4) return JPGERR_UNEXPECTED_EOF;
          } else { // PNM is big-endian
            const unsigned char be[2] = {(unsigned char)(px[i] >> 8), (unsigned char)px[i]};
            if (fwrite(be, 1, 2, sb->target) != 2) return JPGERR_UNEXPECTED_EOF;
          }
        }
      }
    }
    break;
  }
  return 0;
}

static int Reconstruct(const char *infile, const char *outfile, bool colortrafo, const char *alpha, bool upsample, int threads, int device)
{
  FILE *in = fopen(infile, "rb");
  if (!in) { perror("failed to open the input file"); return 10; }
  int rc = 0;
  struct JPG_Hook filehook(FileHook, in);
  struct JPG_TagItem ctags[] = {JPG_ValueTag(device >= 0 ? JPGTAG_MIJPEG_DEVICE : JPGTAG_TAG_IGNORE, device), JPG_EndTag};
  class JPEG *jpeg = JPEG::Construct(ctags);
  if (!jpeg) { fprintf(stderr, "failed to construct the JPEG object (no usable MI355X device?)\n"); fclose(in); return 20; }
  struct JPG_TagItem rtags[] = {JPG_PointerTag(JPGTAG_HOOK_IOHOOK, &filehook), JPG_PointerTag(JPGTAG_HOOK_IOSTREAM, in),
                                JPG_ValueTag(threads > 0 ? JPGTAG_MIJPEG_THREADS : JPGTAG_TAG_IGNORE, threads), JPG_EndTag};
  int ok = jpeg->Read(rtags);
  if (ok) {
    unsigned char subx[4], suby[4];
    // cmd/reconstruct.cpp:126-146: the alpha channel's description arrives in a tag list of its own
    struct JPG_TagItem atags[] = {JPG_ValueTag(JPGTAG_IMAGE_PRECISION, 0), JPG_ValueTag(JPGTAG_IMAGE_IS_FLOAT, 0),
                                  JPG_ValueTag(JPGTAG_IMAGE_OUTPUT_CONVERSION, 1), JPG_EndTag};
    struct JPG_TagItem itags[] = {JPG_ValueTag(JPGTAG_IMAGE_WIDTH, 0), JPG_ValueTag(JPGTAG_IMAGE_HEIGHT, 0), JPG_ValueTag(JPGTAG_IMAGE_DEPTH, 0),
                                  JPG_ValueTag(JPGTAG_IMAGE_PRECISION, 0), JPG_ValueTag(JPGTAG_IMAGE_IS_FLOAT, 0),
                                  JPG_ValueTag(JPGTAG_IMAGE_OUTPUT_CONVERSION, 0), JPG_ValueTag(JPGTAG_ALPHA_MODE, JPGFLAG_ALPHA_OPAQUE),
                                  JPG_PointerTag(JPGTAG_ALPHA_TAGLIST, atags), JPG_PointerTag(JPGTAG_IMAGE_SUBX, subx),
                                  JPG_PointerTag(JPGTAG_IMAGE_SUBY, suby), JPG_ValueTag(JPGTAG_IMAGE_SUBLENGTH, 4), JPG_EndTag};
    ok = jpeg->GetInformation(itags);
    if (ok) {
      const unsigned width = itags->GetTagData(JPGTAG_IMAGE_WIDTH), height = itags->GetTagData(JPGTAG_IMAGE_HEIGHT);
      const unsigned depth = itags->GetTagData(JPGTAG_IMAGE_DEPTH), prec = itags->GetTagData(JPGTAG_IMAGE_PRECISION);
      const bool pfm = itags->GetTagData(JPGTAG_IMAGE_IS_FLOAT) != 0, convert = itags->GetTagData(JPGTAG_IMAGE_OUTPUT_CONVERSION) != 0;
      const bool writepgx = (depth != 1 && depth != 3) || !upsample; // cmd/reconstruct.cpp:218-222
      // cmd/reconstruct.cpp:154-166: an alpha file name and a compositing method other than "opaque" -- else alpha is ignored
      const bool doalpha = alpha && itags->GetTagData(JPGTAG_ALPHA_MODE, JPGFLAG_ALPHA_OPAQUE) != 0 && !writepgx;
      const unsigned aprec = doalpha ? atags->GetTagData(JPGTAG_IMAGE_PRECISION) : 0;
      const bool apfm = doalpha && atags->GetTagData(JPGTAG_IMAGE_IS_FLOAT) != 0, aconvert = doalpha && atags->GetTagData(JPGTAG_IMAGE_OUTPUT_CONVERSION) != 0;
      if (depth > 4 || prec > 16 || (pfm && (!convert || writepgx)) || aprec > 16 || (apfm && !aconvert)) {
        fprintf(stderr, "only images of up to four components of up to 16 bits (PNM/PGX), or JPEG XT profile C with output conversion (PFM), are written by this front end\n");
        ok = 0; rc = 5;
      } else {
        StripeBuffer sb;
        sb.bytes = prec > 8 ? 2 : 1; // cmd/reconstruct.cpp:169-180
        sb.halffloat = pfm;
        sb.mem = (unsigned char *)malloc((size_t)width * 8 * depth * sb.bytes);
        sb.width = width; sb.height = height; sb.depth = depth;
        sb.upsampling = upsample;
        sb.pgx = writepgx;
        memset(sb.pgxfiles, 0, sizeof(sb.pgxfiles));
        sb.target = fopen(outfile, "wb");
        StripeBuffer asb; // the alpha channel's stripe: one component (cmd/reconstruct.cpp:182-196)
        memset(&asb, 0, sizeof(asb));
        if (doalpha) {
          asb.bytes = aprec > 8 ? 2 : 1;
          asb.halffloat = apfm;
          asb.mem = (unsigned char *)malloc((size_t)width * 8 * asb.bytes);
          asb.width = width; asb.height = height; asb.depth = 1;
          asb.upsampling = true;
          asb.target = fopen(alpha, "wb");
          if (!asb.mem || !asb.target) { perror("failed to open the alpha output file"); ok = 0; rc = 10; }
        }
        if (upsample) { // cmd/reconstruct.cpp:227-232: the subsampling factors are all implicitly 1 then
          memset(subx, 1, sizeof(subx));
          memset(suby, 1, sizeof(suby));
        }
        if (!sb.mem || !sb.target) { perror("failed to open the output file"); ok = 0; rc = 10; }
        else if (writepgx) { // cmd/reconstruct.cpp:236-306
          for (unsigned i = 0; i < depth && ok; i++) {
            char headername[512], rawname[512];
            snprintf(headername, sizeof(headername), "%s_%u.h", outfile, i);
            snprintf(rawname, sizeof(rawname), "%s_%u.raw", outfile, i);
            fprintf(sb.target, "%s\n", rawname);
            FILE *hdr = fopen(headername, "wb");
            if (hdr) {
              fprintf(hdr, "PG ML +%u %u %u\n", prec, (width + subx[i] - 1) / subx[i], (height + suby[i] - 1) / suby[i]);
              fclose(hdr);
            }
            sb.pgxfiles[i] = fopen(rawname, "wb");
            if (!hdr || !sb.pgxfiles[i]) { perror("cannot create output file"); ok = 0; rc = 10; }
          }
          struct JPG_Hook bmhook(BitmapHook, &sb);
          for (unsigned comp = 0; comp < depth && ok; comp++) {
            const unsigned step = (unsigned)suby[comp] << 3;
            for (unsigned y = 0; y < height && ok; y += step) {
              const unsigned last = y + step < height ? y + step : height;
              struct JPG_TagItem dtags[] = {JPG_PointerTag(JPGTAG_BIH_HOOK, &bmhook), JPG_ValueTag(JPGTAG_DECODER_MINY, y),
                                            JPG_ValueTag(JPGTAG_DECODER_MAXY, last - 1), JPG_ValueTag(JPGTAG_DECODER_UPSAMPLE, upsample),
                                            JPG_ValueTag(JPGTAG_MATRIX_LTRAFO, colortrafo ? JPGFLAG_MATRIX_COLORTRANSFORMATION_YCBCR
                                                                                           : JPGFLAG_MATRIX_COLORTRANSFORMATION_NONE),
                                            JPG_ValueTag(JPGTAG_DECODER_MINCOMPONENT, comp), JPG_ValueTag(JPGTAG_DECODER_MAXCOMPONENT, comp),
                                            JPG_EndTag};
              ok = jpeg->DisplayRectangle(dtags);
            }
          }
          for (unsigned i = 0; i < depth; i++)
            if (sb.pgxfiles[i]) fclose(sb.pgxfiles[i]);
        } else {
          struct JPG_Hook bmhook(BitmapHook, &sb), alphahook(BitmapHook, &asb);
          // cmd/reconstruct.cpp:321-323
          fprintf(sb.target, "P%c\n%u %u\n%u\n", pfm ? (depth > 1 ? 'F' : 'f') : (depth > 1 ? '6' : '5'), width, height,
                  pfm ? 1u : (1u << prec) - 1);
          if (doalpha && asb.target) // cmd/reconstruct.cpp:325-328
            fprintf(asb.target, "P%c\n%u %u\n%u\n", apfm ? 'f' : '5', width, height, apfm ? 1u : (1u << aprec) - 1);
          for (unsigned y = 0; y < height && ok; y += 8) { // cmd/reconstruct.cpp:334-342
            const unsigned last = y + 8 < height ? y + 8 : height;
            struct JPG_TagItem dtags[] = {JPG_PointerTag(JPGTAG_BIH_HOOK, &bmhook), JPG_PointerTag(JPGTAG_BIH_ALPHAHOOK, &alphahook),
                                          JPG_ValueTag(JPGTAG_DECODER_MINY, y),
                                          JPG_ValueTag(JPGTAG_DECODER_MAXY, last - 1), JPG_ValueTag(JPGTAG_DECODER_UPSAMPLE, 1),
                                          JPG_ValueTag(JPGTAG_MATRIX_LTRAFO, colortrafo ? JPGFLAG_MATRIX_COLORTRANSFORMATION_YCBCR
                                                                                         : JPGFLAG_MATRIX_COLORTRANSFORMATION_NONE),
                                          JPG_ValueTag(JPGTAG_DECODER_INCLUDE_ALPHA, doalpha && asb.target ? 1 : 0),
                                          JPG_EndTag};
            ok = jpeg->DisplayRectangle(dtags);
          }
        }
        if (sb.target) fclose(sb.target);
        if (asb.target) fclose(asb.target);
        free(asb.mem);
        free(sb.mem);
      }
    }
  }
  if (!ok && rc == 0) {
    const char *msg;
    const int code = jpeg->LastError(msg);
    fprintf(stderr, "reading a JPEG file failed - error %d - %s\n", code, msg ? msg : "");
    rc = 5;
  }
  JPEG::Destruct(jpeg);
  fclose(in);
  return rc;
}

// Encoder direction (the reference CLI's `jpeg -bl -q n [-s HxV,HxV,HxV] [-z n] [-h] source.ppm target.jpg`, cmd/main.cpp ->
// cmd/encodec.cpp): binary PNM (P5 / P6, maxval 255) in, baseline JPEG out, as a tag/hook client of class JPEG:
// JPEG::ProvideImage pulls the picture eight lines at a time through a bitmap hook that reads them from the file
// (cmd/bitmaphook.cpp:102-340 in its encoder role), JPEG::Write pushes the stream through the file I/O hook.
// -s takes the reference's SUBSAMPLING factors per component (1x1,2x2,2x2 = 4:2:0).
struct SourceStripe {
  FILE *in;
  unsigned char *mem; // eight lines, interleaved
  int width, depth, next_line;
};

static JPG_LONG SourceBitmapHook(struct JPG_Hook *hook, struct JPG_TagItem *tags)
{
  SourceStripe *s = (SourceStripe *)hook->hk_pData;
  if (tags->GetTagData(JPGTAG_BIO_ACTION) != JPGFLAG_BIO_REQUEST) return 0;
  const JPG_LONG comp = tags->GetTagData(JPGTAG_BIO_COMPONENT), miny = tags->GetTagData(JPGTAG_BIO_MINY), maxy = tags->GetTagData(JPGTAG_BIO_MAXY);
  if (comp == 0) { // a new stripe: read its lines (the requests arrive top-down, component by component)
    if (miny != s->next_line) return -1;
    const size_t bytes = (size_t)(maxy - miny + 1) * (size_t)s->width * (size_t)s->depth;
    if (fread(s->mem, 1, bytes, s->in) != bytes) return -1;
    s->next_line = maxy + 1;
  }
  const JPG_LONG bpr = s->width * s->depth;
  tags->SetTagPtr(JPGTAG_BIO_MEMORY, s->mem + comp - (ptrdiff_t)miny * bpr); // address of canvas pixel (0,0)
  tags->SetTagData(JPGTAG_BIO_WIDTH, s->width);
  tags->SetTagData(JPGTAG_BIO_HEIGHT, 8 + miny);
  tags->SetTagData(JPGTAG_BIO_BYTESPERROW, bpr);
  tags->SetTagData(JPGTAG_BIO_BYTESPERPIXEL, s->depth);
  tags->SetTagData(JPGTAG_BIO_PIXELTYPE, CTYP_UBYTE);
  return 0;
}

static int Encode(const char *src, const char *dst, int quality, const char *sub, int restart, bool optimize, int device)
{
  FILE *in = fopen(src, "rb");
  if (!in) { perror(src); return 10; }
  int w = 0, h = 0, maxval = 0;
  char magic[3] = {0, 0, 0};
  if (fscanf(in, "%2s %d %d %d", magic, &w, &h, &maxval) != 4 || magic[0] != 'P' || (magic[1] != '5' && magic[1] != '6') || maxval != 255 || w < 1 || h < 1) {
    fprintf(stderr, "%s: only binary PGM / PPM files with 8 bits per sample are supported as encoder input\n", src);
    fclose(in);
    return 10;
  }
  fgetc(in); // the single white space behind the header
  const int nc = magic[1] == '6' ? 3 : 1;
  unsigned char subx[4] = {1, 1, 1, 1}, suby[4] = {1, 1, 1, 1};
  if (sub && nc == 3) {
    int sx[3], sy[3];
    if (sscanf(sub, "%dx%d,%dx%d,%dx%d", &sx[0], &sy[0], &sx[1], &sy[1], &sx[2], &sy[2]) != 6) { fprintf(stderr, "-s expects e.g. 1x1,2x2,2x2\n"); fclose(in); return 5; }
    for (int c = 0; c < 3; c++) { subx[c] = (unsigned char)sx[c]; suby[c] = (unsigned char)sy[c]; }
  }
  SourceStripe stripe = {in, (unsigned char *)malloc((size_t)w * (size_t)nc * 8), w, nc, 0};
  FILE *out = fopen(dst, "wb");
  if (!stripe.mem || !out) { perror(dst); fclose(in); if (out) fclose(out); free(stripe.mem); return 10; }
  struct JPG_Hook bmhook(SourceBitmapHook, &stripe), filehook(FileHook, out);
  struct JPG_TagItem ctags[] = {JPG_ValueTag(device >= 0 ? JPGTAG_MIJPEG_DEVICE : JPGTAG_TAG_IGNORE, device), JPG_EndTag};
  class JPEG *jpeg = JPEG::Construct(ctags);
  int rc = 0;
  if (!jpeg) {
    fprintf(stderr, "failed to construct the JPEG object (no MI355X device?)\n");
    rc = 10;
  } else {
    struct JPG_TagItem itags[] = {JPG_PointerTag(JPGTAG_BIH_HOOK, &bmhook),
                                  JPG_ValueTag(JPGTAG_ENCODER_LOOP_ON_INCOMPLETE, true),
                                  JPG_ValueTag(JPGTAG_ENCODER_IMAGE_COMPLETE, false),
                                  JPG_ValueTag(JPGTAG_IMAGE_WIDTH, w),
                                  JPG_ValueTag(JPGTAG_IMAGE_HEIGHT, h),
                                  JPG_ValueTag(JPGTAG_IMAGE_DEPTH, nc),
                                  JPG_ValueTag(JPGTAG_IMAGE_PRECISION, 8),
                                  JPG_ValueTag(JPGTAG_IMAGE_FRAMETYPE, JPGFLAG_BASELINE | (optimize ? JPGFLAG_OPTIMIZE_HUFFMAN : 0)),
                                  JPG_ValueTag(JPGTAG_IMAGE_QUALITY, quality),
                                  JPG_ValueTag(JPGTAG_IMAGE_RESTART_INTERVAL, restart),
                                  JPG_PointerTag(JPGTAG_IMAGE_SUBX, subx),
                                  JPG_PointerTag(JPGTAG_IMAGE_SUBY, suby),
                                  JPG_ValueTag(JPGTAG_MATRIX_LTRAFO, JPGFLAG_MATRIX_COLORTRANSFORMATION_YCBCR),
                                  JPG_EndTag};
    struct JPG_TagItem iotags[] = {JPG_PointerTag(JPGTAG_HOOK_IOHOOK, &filehook), JPG_PointerTag(JPGTAG_HOOK_IOSTREAM, out), JPG_EndTag};
    if (!jpeg->ProvideImage(itags) || !jpeg->Write(iotags)) {
      const char *msg = NULL;
      const JPG_LONG code = jpeg->LastError(msg);
      fprintf(stderr, "encoding failed: error %ld %s\n", (long)code, msg ? msg : "");
      rc = 10;
    }
    JPEG::Destruct(jpeg);
  }
  fclose(in);
  fclose(out);
  free(stripe.mem);
  return rc;
}

int main(int argc, char **argv)
{
  bool colortrafo = true, upsample = true;
  int threads = 0, device = -1;
  int quality = -1, restart = 0;
  const char *sub = NULL, *alpha = NULL;
  bool optimize = false;
  while (argc > 3) {
    if (!strcmp(argv[1], "-q") && argc > 4) { quality = atoi(argv[2]); argv += 2; argc -= 2; continue; }
    if (!strcmp(argv[1], "-s") && argc > 4) { sub = argv[2]; argv += 2; argc -= 2; continue; }
    if (!strcmp(argv[1], "-z") && argc > 4) { restart = atoi(argv[2]); argv += 2; argc -= 2; continue; }
    if (!strcmp(argv[1], "-h")) { optimize = true; argv++; argc--; continue; }
    if (!strcmp(argv[1], "-bl")) { argv++; argc--; continue; } // baseline is what the encoder writes anyway
    if (!strcmp(argv[1], "-c")) { colortrafo = false; argv++; argc--; }
    else if (!strcmp(argv[1], "-U")) { upsample = false; argv++; argc--; }
    else if (!strcmp(argv[1], "-al") && argc > 4) { alpha = argv[2]; argv += 2; argc -= 2; }
    else if (!strcmp(argv[1], "-t") && argc > 4) { threads = atoi(argv[2]); argv += 2; argc -= 2; }
    else if (!strcmp(argv[1], "-d") && argc > 4) { device = atoi(argv[2]); argv += 2; argc -= 2; }
    else break;
  }
  if (argc != 3) {
    fprintf(stderr, "usage: %s [-c] [-U] [-al alpha.pgm] [-t threads] [-d device] source.jpg target.ppm\n"
                    "  reconstructs a Huffman sequential JPEG on an MI355X and writes a binary PNM,\n"
                    "  byte-identical to the output of the reference `jpeg source.jpg target.ppm`\n"
                    "       %s -q quality [-bl] [-s 1x1,2x2,2x2] [-z restart-interval] [-h] [-d device] source.ppm target.jpg\n"
                    "  encodes a binary PNM as baseline JPEG: the coefficients the reference encoder computes\n", argv[0], argv[0]);
    return 5;
  }
  if (quality >= 0) return Encode(argv[1], argv[2], quality, sub, restart, optimize, device);
  return Reconstruct(argv[1], argv[2], colortrafo, alpha, upsample, threads, device);
}
