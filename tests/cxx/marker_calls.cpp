// Test client for the incremental half of class JPEG (interface/jpeg.hpp): JPEG::Read with JPGTAG_DECODER_STOP,
// PeekMarker / ReadMarker / SkipMarker -- the protocol of the reference's own marker injection test
// (cmd/reconstruct.cpp:80-119), here with STOP_IMAGE or STOP_FRAME chosen on the command line and every step printed.
//   marker_calls <in.jpg> <image|frame|scan> [skip-garbage-bytes]    (scan: Read with JPGFLAG_DECODER_STOP_SCAN until the data ends)
// Prints one line per Read: "peek <hex>" and, for APP9 (ffe9) segments, "took <n>"; finally "info <w> <h> <depth>".
// Build: g++ -I libjpeg_amd/csrc tests/cxx/marker_calls.cpp -L libjpeg_amd -lmijpeg
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "interface/hooks.hpp"
#include "interface/jpeg.hpp"
#include "interface/parameters.hpp"
#include "interface/tagitem.hpp"

static JPG_LONG FileHook(struct JPG_Hook *hook, struct JPG_TagItem *tags)
{
  FILE *in = (FILE *)(hook->hk_pData);
  switch (tags->GetTagData(JPGTAG_FIO_ACTION)) {
  case JPGFLAG_ACTION_READ: {
    void *buffer = tags->GetTagPtr(JPGTAG_FIO_BUFFER);
    JPG_LONG size = tags->GetTagData(JPGTAG_FIO_SIZE);
    if (size > 777) size = 777; // short reads are normal (io/iostream.cpp)
    return (JPG_LONG)fread(buffer, 1, (size_t)size, in);
  }
  default: return -1;
  }
}

int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  FILE *in = fopen(argv[1], "rb");
  if (!in) return 2;
  // row / mcu / scanrow: JPGFLAG_DECODER_STOP_ROW, _MCU, _SCAN | _ROW until the data ends.  Inside a scan the stream stands wherever
  // the bit reader's buffer got to, so what is compared there is the NUMBER of returns ("reads N"), not what a peek shows
  const bool rows = !strcmp(argv[2], "row") || !strcmp(argv[2], "mcu") || !strcmp(argv[2], "scanrow");
  const bool scans = !strcmp(argv[2], "scan") || rows; // JPGFLAG_DECODER_STOP_SCAN: one Read per scan header, until the data ends
  const JPG_LONG stop = !strcmp(argv[2], "image") ? JPGFLAG_DECODER_STOP_IMAGE : !strcmp(argv[2], "row") ? JPGFLAG_DECODER_STOP_ROW :
                        !strcmp(argv[2], "mcu") ? JPGFLAG_DECODER_STOP_MCU : !strcmp(argv[2], "scanrow") ? (JPGFLAG_DECODER_STOP_SCAN | JPGFLAG_DECODER_STOP_ROW) :
                        scans ? JPGFLAG_DECODER_STOP_SCAN : JPGFLAG_DECODER_STOP_FRAME;
  const long extra = argc > 3 ? atol(argv[3]) : 0; // raw bytes behind every APP9 segment that the client removes as well
  struct JPG_Hook filehook(FileHook, in);
  class JPEG *jpeg = JPEG::Construct(NULL);
  if (!jpeg) { fprintf(stderr, "no JPEG object\n"); return 3; }
  if (jpeg->PeekMarker(NULL) != -1) { fprintf(stderr, "PeekMarker before Read must fail\n"); return 4; }
  struct JPG_TagItem tags[] = {JPG_PointerTag(JPGTAG_HOOK_IOHOOK, &filehook), JPG_PointerTag(JPGTAG_HOOK_IOSTREAM, in),
                               JPG_ValueTag(JPGTAG_DECODER_STOP, stop), JPG_EndTag};
  JPG_LONG marker;
  int ok;
  int guard = 0;
  do {
    ok = jpeg->Read(tags);
    marker = jpeg->PeekMarker(NULL);
    if (!rows) printf("peek %lx\n", (unsigned long)(marker & 0xffffffffUL));
    if (rows && marker == 0xffe9) marker = 1; // (entropy coded bytes that look like a marker are no marker)
    if (marker == 0xffe9) {
      unsigned char buffer[4];
      ok = jpeg->ReadMarker(buffer, sizeof(buffer), NULL) == (JPG_LONG)sizeof(buffer);
      if (ok) {
        const int size = (buffer[2] << 8) + buffer[3];
        ok = size >= 2 && jpeg->SkipMarker(size - 2 + extra, NULL) != -1;
        printf("took %ld\n", (long)(2 + size + extra));
      }
    }
  } while ((scans ? marker != -1L : (marker && marker != -1L)) && ok && ++guard < (rows ? 50000000 : 1000));
  if (rows) printf("reads %d\n", guard + 1);
  tags->SetTagData(JPGTAG_DECODER_STOP, 0);
  if (!ok || !jpeg->Read(tags)) {
    const char *msg = NULL;
    const JPG_LONG code = jpeg->LastError(msg);
    printf("error %ld %s\n", (long)code, msg ? msg : "");
    JPEG::Destruct(jpeg);
    return 1;
  }
  struct JPG_TagItem itags[] = {JPG_ValueTag(JPGTAG_IMAGE_WIDTH, 0), JPG_ValueTag(JPGTAG_IMAGE_HEIGHT, 0), JPG_ValueTag(JPGTAG_IMAGE_DEPTH, 0), JPG_EndTag};
  jpeg->GetInformation(itags);
  printf("info %ld %ld %ld\n", (long)itags->GetTagData(JPGTAG_IMAGE_WIDTH), (long)itags->GetTagData(JPGTAG_IMAGE_HEIGHT),
         (long)itags->GetTagData(JPGTAG_IMAGE_DEPTH));
  printf("peek-after %ld\n", (long)jpeg->PeekMarker(NULL));
  JPEG::Destruct(jpeg);
  fclose(in);
  return 0;
}
