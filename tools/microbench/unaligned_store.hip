// Do dwordx2 / dwordx4 global stores work at any byte address on gfx950 (unaligned access mode)?  24 bytes per thread at out + off + 24 t.
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench/unaligned_store.hip -o tools/microbench/unaligned_store
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(1))) U2 { u32x2 v; };
__global__ void k(unsigned char *out, int off)
{
  const int t = threadIdx.x + blockIdx.x * 64;
  unsigned char *p = out + off + 24 * t;
  for (int i = 0; i < 3; i++) {
    U2 u; u.v = u32x2{0x03020100u + 0x04040404u * (2 * i) + (t & 127) * 0x01010101u, 0x07060504u + 0x04040404u * (2 * i) + (t & 127) * 0x01010101u};
    *reinterpret_cast<U2 *>(p + 8 * i) = u;
  }
}
int main()
{
  unsigned char *d; unsigned char h[64 * 4 * 24 + 64];
  hipMalloc(&d, sizeof(h));
  int bad = 0;
  for (int off = 0; off < 8; off++) {
    hipMemset(d, 0xEE, sizeof(h));
    hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, 0, d, off);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int here = 0, shown = 0;
    for (int t = 0; t < 256; t++) for (int b = 0; b < 24; b++) if (h[off + 24 * t + b] != (unsigned char)(b + (t & 127))) { here++; if (shown++ < 6) printf("  off %d: byte %d (t %d b %d) is %02x, expected %02x\n", off, off + 24 * t + b, t, b, h[off + 24 * t + b], (unsigned char)(b + (t & 127))); }
    printf("off %d: %d wrong\n", off, here); bad += here;
    for (int b = 0; b < off; b++) bad += h[b] != 0xEE;
    bad += h[off + 24 * 256] != 0xEE;
  }
  printf("unaligned dwordx2 stores: %d wrong bytes\n", bad);
  return 0;
}
