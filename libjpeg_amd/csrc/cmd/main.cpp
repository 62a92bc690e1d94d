// main.cpp -- `jpeg` command line front end of the MI355X path: the decode half of the reference CLI.
//   jpeg [-c] [-t threads] [-d device] in.jpg out.ppm
// reproduces cmd/main.cpp:746-747 -> cmd/reconstruct.cpp:68-376 for the streams this path handles: the
// image (8 bit -> PNM, 12 bit -> 16-bit PNM, JPEG XT profile C -> PFM) is reconstructed stripe by stripe (eight lines per JPEG::DisplayRectangle call) through a file I/O
// hook and a bitmap hook, and written as binary PNM (P6 for three components, P5 for one) -- byte for
// byte what the reference binary writes.  -c disables the colour transformation (reference: -c).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
  FILE *target;
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
  const unsigned miny = (unsigned)tags->GetTagData(JPGTAG_BIO_MINY), maxy = (unsigned)tags->GetTagData(JPGTAG_BIO_MAXY);
  const unsigned width = 1 + (unsigned)tags->GetTagData(JPGTAG_BIO_MAXX);
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
    if (comp == sb->depth - 1) { // all components of the stripe are in: write it
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

static int Reconstruct(const char *infile, const char *outfile, bool colortrafo, int threads, int device)
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
    struct JPG_TagItem itags[] = {JPG_ValueTag(JPGTAG_IMAGE_WIDTH, 0), JPG_ValueTag(JPGTAG_IMAGE_HEIGHT, 0), JPG_ValueTag(JPGTAG_IMAGE_DEPTH, 0),
                                  JPG_ValueTag(JPGTAG_IMAGE_PRECISION, 0), JPG_ValueTag(JPGTAG_IMAGE_IS_FLOAT, 0),
                                  JPG_ValueTag(JPGTAG_IMAGE_OUTPUT_CONVERSION, 0), JPG_PointerTag(JPGTAG_IMAGE_SUBX, subx),
                                  JPG_PointerTag(JPGTAG_IMAGE_SUBY, suby), JPG_ValueTag(JPGTAG_IMAGE_SUBLENGTH, 4), JPG_EndTag};
    ok = jpeg->GetInformation(itags);
    if (ok) {
      const unsigned width = itags->GetTagData(JPGTAG_IMAGE_WIDTH), height = itags->GetTagData(JPGTAG_IMAGE_HEIGHT);
      const unsigned depth = itags->GetTagData(JPGTAG_IMAGE_DEPTH), prec = itags->GetTagData(JPGTAG_IMAGE_PRECISION);
      const bool pfm = itags->GetTagData(JPGTAG_IMAGE_IS_FLOAT) != 0, convert = itags->GetTagData(JPGTAG_IMAGE_OUTPUT_CONVERSION) != 0;
      if ((depth != 1 && depth != 3) || prec > 16 || (pfm && !convert)) {
        fprintf(stderr, "only images with one or three components of up to 16 bits can be written as PNM/PFM by this front end\n");
        ok = 0; rc = 5;
      } else {
        StripeBuffer sb;
        sb.bytes = prec > 8 ? 2 : 1; // cmd/reconstruct.cpp:169-180
        sb.halffloat = pfm;
        sb.mem = (unsigned char *)malloc((size_t)width * 8 * depth * sb.bytes);
        sb.width = width; sb.height = height; sb.depth = depth;
        sb.target = fopen(outfile, "wb");
        if (!sb.mem || !sb.target) { perror("failed to open the output file"); ok = 0; rc = 10; }
        else {
          struct JPG_Hook bmhook(BitmapHook, &sb);
          // cmd/reconstruct.cpp:321-323
          fprintf(sb.target, "P%c\n%u %u\n%u\n", pfm ? (depth > 1 ? 'F' : 'f') : (depth > 1 ? '6' : '5'), width, height,
                  pfm ? 1u : (1u << prec) - 1);
          for (unsigned y = 0; y < height && ok; y += 8) { // cmd/reconstruct.cpp:334-342
            const unsigned last = y + 8 < height ? y + 8 : height;
            struct JPG_TagItem dtags[] = {JPG_PointerTag(JPGTAG_BIH_HOOK, &bmhook), JPG_ValueTag(JPGTAG_DECODER_MINY, y),
                                          JPG_ValueTag(JPGTAG_DECODER_MAXY, last - 1), JPG_ValueTag(JPGTAG_DECODER_UPSAMPLE, 1),
                                          JPG_ValueTag(JPGTAG_MATRIX_LTRAFO, colortrafo ? JPGFLAG_MATRIX_COLORTRANSFORMATION_YCBCR
                                                                                         : JPGFLAG_MATRIX_COLORTRANSFORMATION_NONE),
                                          JPG_EndTag};
            ok = jpeg->DisplayRectangle(dtags);
          }
        }
        if (sb.target) fclose(sb.target);
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

int main(int argc, char **argv)
{
  bool colortrafo = true;
  int threads = 0, device = -1;
  while (argc > 3) {
    if (!strcmp(argv[1], "-c")) { colortrafo = false; argv++; argc--; }
    else if (!strcmp(argv[1], "-t") && argc > 4) { threads = atoi(argv[2]); argv += 2; argc -= 2; }
    else if (!strcmp(argv[1], "-d") && argc > 4) { device = atoi(argv[2]); argv += 2; argc -= 2; }
    else break;
  }
  if (argc != 3) {
    fprintf(stderr, "usage: %s [-c] [-t threads] [-d device] source.jpg target.ppm\n"
                    "  reconstructs a Huffman sequential JPEG on an MI355X and writes a binary PNM,\n"
                    "  byte-identical to the output of the reference `jpeg source.jpg target.ppm`\n", argv[0]);
    return 5;
  }
  return Reconstruct(argv[1], argv[2], colortrafo, threads, device);
}
