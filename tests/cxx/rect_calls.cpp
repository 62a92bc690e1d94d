// Test client for the rectangle service of class JPEG (interface/jpeg.hpp): a scripted SEQUENCE of JPEG::DisplayRectangle
// calls against one decoded image -- what cmd/reconstruct.cpp:272-342 does in two fixed patterns, here with any order of
// stripes, component ranges and bitmap heights, so the state the reference keeps between calls (row cursors per component,
// control/blockbitmaprequester.cpp:1013-1224; upsampler line buffers, upsampling/upsamplerbase.cpp) shows in the output.
//   rect_calls <in.jpg> <script> <out.bin>
// script: one request per line: minx miny maxx maxy c0 c1 upsample ctrafo hmode   (maxx / maxy < 0: the canvas edge)
//   hmode 0: the hook reports the canvas height; k > 0: BIO_HEIGHT = miny + k (cmd/bitmaphook.cpp:122 has k = 8)
// out.bin: per component one canvas-sized plane (W x H samples of 1 or 2 bytes, initialised to 0xAA) as the calls left it.
// stdout: "info W H depth precision", then one line "call <index> <JPG_TRUE/FALSE> <LastError code>" per request.
// The same source is linked against the real reference objects (oracle/Makefile: _ref/rect_calls_ref) and against
// libmijpeg.so (_ref/rect_calls_ours); tests/test_rect_calls.py compares the two and the oracle's oj_requester.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "interface/hooks.hpp"
#include "interface/jpeg.hpp"
#include "interface/parameters.hpp"
#include "interface/tagitem.hpp"

#ifndef CTYP_UBYTE // the reference keeps the pixel type ids in tools/traits.hpp:65-85 (= the sample size for integer types)
#define CTYP_UBYTE 1
#define CTYP_UWORD 2
#endif

static JPG_LONG FileHook(struct JPG_Hook *hook, struct JPG_TagItem *tags)
{
  FILE *in = (FILE *)(hook->hk_pData);
  switch (tags->GetTagData(JPGTAG_FIO_ACTION)) {
  case JPGFLAG_ACTION_READ: return (JPG_LONG)fread(tags->GetTagPtr(JPGTAG_FIO_BUFFER), 1, (size_t)tags->GetTagData(JPGTAG_FIO_SIZE), in);
  default: return -1;
  }
}

struct Canvas {
  unsigned char *plane[4];
  long width, height, sample_bytes;
  long hmode;
};

static JPG_LONG BitmapHook(struct JPG_Hook *hook, struct JPG_TagItem *tags)
{
  Canvas *cv = (Canvas *)(hook->hk_pData);
  const long comp = tags->GetTagData(JPGTAG_BIO_COMPONENT);
  if (tags->GetTagData(JPGTAG_BIO_ACTION) == JPGFLAG_BIO_REQUEST) {
    const long miny = tags->GetTagData(JPGTAG_BIO_MINY);
    tags->SetTagPtr(JPGTAG_BIO_MEMORY, cv->plane[comp]);
    tags->SetTagData(JPGTAG_BIO_WIDTH, cv->width);
    tags->SetTagData(JPGTAG_BIO_HEIGHT, cv->hmode ? miny + cv->hmode : cv->height);
    tags->SetTagData(JPGTAG_BIO_BYTESPERROW, cv->width * cv->sample_bytes);
    tags->SetTagData(JPGTAG_BIO_BYTESPERPIXEL, cv->sample_bytes);
    tags->SetTagData(JPGTAG_BIO_PIXELTYPE, cv->sample_bytes == 2 ? CTYP_UWORD : CTYP_UBYTE);
  }
  return 0;
}

int main(int argc, char **argv)
{
  if (argc < 4) return 2;
  FILE *in = fopen(argv[1], "rb");
  FILE *script = fopen(argv[2], "r");
  if (!in || !script) return 2;
  struct JPG_Hook filehook(FileHook, in);
  class JPEG *jpeg = JPEG::Construct(NULL);
  if (!jpeg) return 3;
  struct JPG_TagItem tags[] = {JPG_PointerTag(JPGTAG_HOOK_IOHOOK, &filehook), JPG_PointerTag(JPGTAG_HOOK_IOSTREAM, in), JPG_EndTag};
  if (!jpeg->Read(tags)) {
    const char *msg = NULL;
    printf("error %ld\n", (long)jpeg->LastError(msg));
    JPEG::Destruct(jpeg);
    return 1;
  }
  struct JPG_TagItem itags[] = {JPG_ValueTag(JPGTAG_IMAGE_WIDTH, 0), JPG_ValueTag(JPGTAG_IMAGE_HEIGHT, 0), JPG_ValueTag(JPGTAG_IMAGE_DEPTH, 0),
                                JPG_ValueTag(JPGTAG_IMAGE_PRECISION, 0), JPG_EndTag};
  jpeg->GetInformation(itags);
  Canvas cv;
  memset(&cv, 0, sizeof(cv));
  cv.width = itags->GetTagData(JPGTAG_IMAGE_WIDTH);
  cv.height = itags->GetTagData(JPGTAG_IMAGE_HEIGHT);
  const long depth = itags->GetTagData(JPGTAG_IMAGE_DEPTH), prec = itags->GetTagData(JPGTAG_IMAGE_PRECISION);
  cv.sample_bytes = prec > 8 ? 2 : 1;
  printf("info %ld %ld %ld %ld\n", cv.width, cv.height, depth, prec);
  if (depth < 1 || depth > 4) return 4;
  const size_t plane_bytes = (size_t)cv.width * (size_t)cv.height * (size_t)cv.sample_bytes;
  for (long c = 0; c < depth; c++) {
    cv.plane[c] = (unsigned char *)malloc(plane_bytes);
    memset(cv.plane[c], 0xAA, plane_bytes);
  }
  struct JPG_Hook bmhook(BitmapHook, &cv);
  long minx, miny, maxx, maxy, c0, c1, ups, ctrafo, hmode;
  int index = 0;
  while (fscanf(script, "%ld %ld %ld %ld %ld %ld %ld %ld %ld", &minx, &miny, &maxx, &maxy, &c0, &c1, &ups, &ctrafo, &hmode) == 9) {
    cv.hmode = hmode;
    struct JPG_TagItem rtags[] = {JPG_PointerTag(JPGTAG_BIH_HOOK, &bmhook),
                                  JPG_ValueTag(JPGTAG_DECODER_MINX, minx),
                                  JPG_ValueTag(JPGTAG_DECODER_MINY, miny),
                                  JPG_ValueTag(JPGTAG_DECODER_MAXX, maxx < 0 ? cv.width - 1 : maxx),
                                  JPG_ValueTag(JPGTAG_DECODER_MAXY, maxy < 0 ? cv.height - 1 : maxy),
                                  JPG_ValueTag(JPGTAG_DECODER_MINCOMPONENT, c0),
                                  JPG_ValueTag(JPGTAG_DECODER_MAXCOMPONENT, c1),
                                  JPG_ValueTag(JPGTAG_DECODER_UPSAMPLE, ups),
                                  JPG_ValueTag(JPGTAG_MATRIX_LTRAFO, ctrafo ? JPGFLAG_MATRIX_COLORTRANSFORMATION_YCBCR : JPGFLAG_MATRIX_COLORTRANSFORMATION_NONE),
                                  JPG_EndTag};
    const JPG_LONG ok = jpeg->DisplayRectangle(rtags);
    const char *msg = NULL;
    printf("call %d %ld %ld\n", index++, (long)ok, ok ? 0L : (long)jpeg->LastError(msg));
  }
  FILE *out = fopen(argv[3], "wb");
  if (!out) return 2;
  for (long c = 0; c < depth; c++) fwrite(cv.plane[c], 1, plane_bytes, out);
  fclose(out);
  JPEG::Destruct(jpeg);
  fclose(in);
  fclose(script);
  return 0;
}
