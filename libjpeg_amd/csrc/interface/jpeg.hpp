// jpeg.hpp -- class JPEG, the library object of the tag/hook API, decode half.
// Mirrors the public surface of the reference's interface/jpeg.hpp:185-252 (same member names, argument
// meaning, return conventions: JPG_TRUE / JPG_FALSE, error via LastError) on top of the C ABI of
// include/mijpeg.h.  Encoder half: ProvideImage / Write for baseline frames; WriteMarker answers JPGERR_NOT_IMPLEMENTED.
#ifndef MIJ_INTERFACE_JPEG_HPP
#define MIJ_INTERFACE_JPEG_HPP
#include "jpgtypes.hpp"
#include "tagitem.hpp"

class JPG_EXPORT JPEG {
  struct Impl;
  Impl *m_pImpl;
  JPEG();
  ~JPEG();
  JPEG(const JPEG &);
  const JPEG &operator=(const JPEG &);

public:
  // interface/jpeg.cpp:142 / :183 -- the only way to create and delete the object
  static class JPEG *Construct(struct JPG_TagItem *tags);
  static void Destruct(class JPEG *o);
  // interface/jpeg.cpp:205 -- read the codestream through JPGTAG_HOOK_IOHOOK and entropy-decode it
  JPG_LONG Read(struct JPG_TagItem *tags);
  // interface/jpeg.cpp:694 -- reconstruct a rectangle into the memory the JPGTAG_BIH_HOOK hands out
  JPG_LONG DisplayRectangle(struct JPG_TagItem *tags);
  // interface/jpeg.cpp:822 -- dimensions, depth, precision, subsampling
  JPG_LONG GetInformation(struct JPG_TagItem *tags);
  // interface/jpeg.cpp:959 / :981
  JPG_LONG LastError(const char *&error);
  JPG_LONG LastWarning(const char *&warning);
  // encoder half and marker injection: not on this path
  JPG_LONG Write(struct JPG_TagItem *tags);
  JPG_LONG ProvideImage(struct JPG_TagItem *tags);
  JPG_LONG PeekMarker(struct JPG_TagItem *tags);
  JPG_LONG ReadMarker(void *buffer, JPG_LONG bufsize, struct JPG_TagItem *tags);
  JPG_LONG SkipMarker(JPG_LONG bytes, struct JPG_TagItem *tags);
  JPG_LONG WriteMarker(void *buffer, JPG_LONG bufsize, struct JPG_TagItem *tags);
};
#endif
