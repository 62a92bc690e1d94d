// Measurement client for the drop-in boundary (not a test): class JPEG decodes one stream from memory N times, (a) with ONE
// DisplayRectangle request for the whole frame, (b) with the loop of cmd/reconstruct.cpp:308-342 -- stripes of eight lines, the
// bitmap hook reporting BIO_HEIGHT = miny + 8 as cmd/bitmaphook.cpp:122 does -- into an interleaved 8-bit bitmap in host memory.
//   stripe_loop <in.jpg> [repetitions]
// Prints, in ms (best of the repetitions): Read, whole-frame request, stripe loop, first stripe.
// The same source builds against the real reference library (oracle/Makefile: _ref/stripe_loop_ref, one thread, no GPU) and
// against libmijpeg.so (_ref/stripe_loop_ours): what a client of the tag/hook API sees when it swaps the library.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "interface/hooks.hpp"
#include "interface/jpeg.hpp"
#include "interface/parameters.hpp"
#include "interface/tagitem.hpp"

#ifndef CTYP_UBYTE
#define CTYP_UBYTE 1
#endif

struct Memory {
  const unsigned char *data;
  size_t size, at;
};
static JPG_LONG MemoryHook(struct JPG_Hook *hook, struct JPG_TagItem *tags)
{
  Memory *m = (Memory *)(hook->hk_pData);
  if (tags->GetTagData(JPGTAG_FIO_ACTION) != JPGFLAG_ACTION_READ) return -1;
  const size_t want = (size_t)tags->GetTagData(JPGTAG_FIO_SIZE), n = std::min(want, m->size - m->at);
  memcpy(tags->GetTagPtr(JPGTAG_FIO_BUFFER), m->data + m->at, n);
  m->at += n;
  return (JPG_LONG)n;
}

struct Canvas {
  unsigned char *pixels;
  long width, height, depth, stripe;
};
static JPG_LONG BitmapHook(struct JPG_Hook *hook, struct JPG_TagItem *tags)
{
  Canvas *cv = (Canvas *)(hook->hk_pData);
  if (tags->GetTagData(JPGTAG_BIO_ACTION) == JPGFLAG_BIO_REQUEST) {
    const long comp = tags->GetTagData(JPGTAG_BIO_COMPONENT), miny = tags->GetTagData(JPGTAG_BIO_MINY);
    tags->SetTagPtr(JPGTAG_BIO_MEMORY, cv->pixels + comp);
    tags->SetTagData(JPGTAG_BIO_WIDTH, cv->width);
    tags->SetTagData(JPGTAG_BIO_HEIGHT, cv->stripe ? miny + cv->stripe : cv->height);
    tags->SetTagData(JPGTAG_BIO_BYTESPERROW, cv->width * cv->depth);
    tags->SetTagData(JPGTAG_BIO_BYTESPERPIXEL, cv->depth);
    tags->SetTagData(JPGTAG_BIO_PIXELTYPE, CTYP_UBYTE);
  }
  return 0;
}

typedef std::chrono::steady_clock clk;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

int main(int argc, char **argv)
{
  if (argc < 2) return 2;
  FILE *in = fopen(argv[1], "rb");
  if (!in) return 2;
  std::vector<unsigned char> file;
  unsigned char buf[65536];
  for (size_t n; (n = fread(buf, 1, sizeof(buf), in)) > 0;) file.insert(file.end(), buf, buf + n);
  fclose(in);
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  Canvas cv;
  memset(&cv, 0, sizeof(cv));
  double best_read[2] = {1e30, 1e30}, best_total[2] = {1e30, 1e30}, best_first = 1e30;
  unsigned long sums[2] = {0, 0};
  for (int mode = 0; mode < 2; mode++) { // 0: one request, 1: stripes of eight lines
    for (int r = 0; r < reps; r++) {
      Memory mem = {file.data(), file.size(), 0};
      struct JPG_Hook filehook(MemoryHook, &mem);
      const clk::time_point t0 = clk::now();
      class JPEG *jpeg = JPEG::Construct(NULL);
      if (!jpeg) return 3;
      struct JPG_TagItem tags[] = {JPG_PointerTag(JPGTAG_HOOK_IOHOOK, &filehook), JPG_PointerTag(JPGTAG_HOOK_IOSTREAM, &mem), JPG_EndTag};
      if (!jpeg->Read(tags)) {
        const char *msg = NULL;
        printf("error %ld %s\n", (long)jpeg->LastError(msg), msg ? msg : "");
        return 1;
      }
      const clk::time_point t1 = clk::now();
      struct JPG_TagItem itags[] = {JPG_ValueTag(JPGTAG_IMAGE_WIDTH, 0), JPG_ValueTag(JPGTAG_IMAGE_HEIGHT, 0), JPG_ValueTag(JPGTAG_IMAGE_DEPTH, 0), JPG_EndTag};
      jpeg->GetInformation(itags);
      cv.width = itags->GetTagData(JPGTAG_IMAGE_WIDTH);
      cv.height = itags->GetTagData(JPGTAG_IMAGE_HEIGHT);
      cv.depth = itags->GetTagData(JPGTAG_IMAGE_DEPTH);
      if (!cv.pixels) cv.pixels = (unsigned char *)calloc((size_t)cv.width * cv.height * cv.depth, 1);
      cv.stripe = mode ? 8 : 0;
      struct JPG_Hook bmhook(BitmapHook, &cv);
      clk::time_point tf = t1;
      for (long y = 0; y < cv.height; y += mode ? 8 : cv.height) {
        const long last = mode ? std::min(cv.height, y + 8) - 1 : cv.height - 1;
        struct JPG_TagItem rtags[] = {JPG_PointerTag(JPGTAG_BIH_HOOK, &bmhook),
                                      JPG_ValueTag(JPGTAG_DECODER_MINY, y),
                                      JPG_ValueTag(JPGTAG_DECODER_MAXY, last),
                                      JPG_ValueTag(JPGTAG_DECODER_MINCOMPONENT, 0),
                                      JPG_ValueTag(JPGTAG_DECODER_MAXCOMPONENT, cv.depth - 1),
                                      JPG_ValueTag(JPGTAG_DECODER_UPSAMPLE, 1),
                                      JPG_EndTag};
        if (!jpeg->DisplayRectangle(rtags)) {
          const char *msg = NULL;
          printf("error %ld %s\n", (long)jpeg->LastError(msg), msg ? msg : "");
          return 1;
        }
        if (y == 0) tf = clk::now();
      }
      const clk::time_point t2 = clk::now();
      JPEG::Destruct(jpeg);
      best_read[mode] = std::min(best_read[mode], ms(t0, t1));
      best_total[mode] = std::min(best_total[mode], ms(t0, t2));
      if (mode) best_first = std::min(best_first, ms(t1, tf));
      unsigned long s = 0;
      for (size_t i = 0; i < (size_t)cv.width * cv.height * cv.depth; i += 4099) s += cv.pixels[i];
      sums[mode] = s;
      memset(cv.pixels, 0, (size_t)cv.width * cv.height * cv.depth);
    }
  }
  printf("%ldx%ldx%ld, best of %d: Read %.2f ms; whole frame in one request %.2f ms (Read included); stripe loop %.2f ms (Read included), "
         "first stripe %.2f ms after Read; sample sums %lu %lu (%s)\n",
         cv.width, cv.height, cv.depth, reps, std::min(best_read[0], best_read[1]), best_total[0], best_total[1], best_first, sums[0], sums[1],
         sums[0] == sums[1] ? "equal" : "DIFFERENT");
  return sums[0] == sums[1] ? 0 : 1;
}
