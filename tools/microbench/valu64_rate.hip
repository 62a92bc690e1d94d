// valu64_rate.hip -- issue rate of the 64-bit integer VALU instructions the device Huffman decoder uses (bit window shifts,
// range accumulation) next to their 32-bit replacements.  Same method as valu_rate.hip.
// Build: hipcc --offload-arch=gfx950 -O3 valu64_rate.hip -o valu64_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITER 2048
typedef unsigned long long u64;

#define KERNEL64(NAME, ASM)                                                                          \
  __global__ __launch_bounds__(256) void k_##NAME(u64 *out, int a, int b)                            \
  {                                                                                                  \
    u64 r[8];                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 8; i++) r[i] = (u64)threadIdx.x * (i + 1) + a;             \
    int x = (b + threadIdx.x) & 31, y = a ^ 0x55;                                                    \
    for (int it = 0; it < ITER; it++) {                                                              \
      _Pragma("unroll") for (int k = 0; k < 4; k++) {                                                \
        _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y) : "vcc"); \
      }                                                                                              \
    }                                                                                                \
    u64 s = 0;                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; i++) s ^= r[i];                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                  \
  }

KERNEL64(lshlrev_b64, "v_lshlrev_b64 %0, %1, %0")
KERNEL64(lshrrev_b64, "v_lshrrev_b64 %0, %1, %0")
KERNEL64(mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0")

typedef void (*kfn)(u64 *, int, int);
struct Entry { const char *name; kfn fn; int per; };

int main()
{
  Entry tab[] = {{"v_lshlrev_b64", k_lshlrev_b64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_mad_u64_u32", k_mad_u64_u32, 1},
                 {"v_lshl_add_u64", k_lshl_add_u64, 1}};
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  u64 *out;
  const int blocks = cus * 8;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  for (auto &e : tab) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    e.fn<<<blocks, 256>>>(out, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    e.fn<<<blocks, 256>>>(out, 1, 2);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double insts = (double)blocks * 4 /*waves*/ * ITER * 32.0 * e.per;  // wave-instructions
    const double per_simd = insts / (cus * 4);
    printf("%-34s %8.3f ms  %6.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", e.name, ms, ms * 1e-3 * 2.4e9 / per_simd);
  }
  return 0;
}
