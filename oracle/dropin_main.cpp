// dropin_main.cpp -- TEST INFRASTRUCTURE: the smallest possible driver around the reference's own decoder client,
// cmd/reconstruct.cpp (compiled unchanged from the reference tree, see Makefile target "dropin"), to show that it runs
// on top of libmijpeg.so.  What the reference's cmd/main.cpp does for `jpeg in.jpg out.ppm` (cmd/main.cpp: the branch
// that ends in Reconstruct(), default arguments: YCbCr colour transformation, no alpha file, upsampling on;
// `-c`, `-U` and `-al file` map to the switches below).
#include "cmd/reconstruct.hpp"
#include "interface/parameters.hpp"
#include <stdio.h>
#include <string.h>

int main(int argc, char **argv)
{
  int colortrafo = JPGFLAG_MATRIX_COLORTRANSFORMATION_YCBCR;
  bool upsample = true;
  const char *alpha = NULL;
  int i = 1;
  for (; i < argc && argv[i][0] == '-'; i++) {
    if (!strcmp(argv[i], "-c")) colortrafo = JPGFLAG_MATRIX_COLORTRANSFORMATION_NONE;
    else if (!strcmp(argv[i], "-U")) upsample = false;
    else if (!strcmp(argv[i], "-al") && i + 1 < argc) alpha = argv[++i];
    else { fprintf(stderr, "unknown switch %s\n", argv[i]); return 2; }
  }
  if (argc - i != 2) { fprintf(stderr, "usage: %s [-c] [-U] [-al alpha.pgm] in.jpg out.ppm\n", argv[0]); return 2; }
  Reconstruct(argv[i], argv[i + 1], colortrafo, alpha, upsample);
  return 0;
}
