// jpgtypes.hpp -- elementary types of the tag/hook API.  Source compatible with the identifiers of the
// reference's interface/jpgtypes.hpp:63-110 (JPG_LONG is a 32-bit int, JPG_APTR a void pointer).
#ifndef MIJ_INTERFACE_JPGTYPES_HPP
#define MIJ_INTERFACE_JPGTYPES_HPP
#include <stdint.h>

typedef int32_t JPG_LONG;
typedef uint32_t JPG_ULONG;
typedef float JPG_FLOAT;
typedef void *JPG_APTR;
typedef const void *JPG_CPTR;

#define JPG_TRUE (1)
#define JPG_FALSE (0)

#ifndef JPG_EXPORT
#define JPG_EXPORT __attribute__((visibility("default")))
#endif
#endif
