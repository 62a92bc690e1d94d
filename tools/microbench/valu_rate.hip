// valu_rate.hip -- measures the issue rate of the VALU instructions the JPEG kernels are made of on gfx950.
// Each kernel runs ITER x 32 independent instructions of one kind per lane (16 accumulators, 2 rounds).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define ITER 2048

#define R16(OP)  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

#define KERNEL(NAME, ASM)                                                                            \
  __global__ __launch_bounds__(256) void k_##NAME(int *out, int a, int b)                            \
  {                                                                                                  \
    int r[16];                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 16; i++) r[i] = threadIdx.x * (i + 1) + a;                 \
    int x = b + threadIdx.x, y = a ^ 0x55;                                                           \
    for (int it = 0; it < ITER; it++) {                                                              \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y)); \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y)); \
    }                                                                                                \
    int s = 0;                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= r[i];                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                  \
  }

KERNEL(add_u32, "v_add_u32 %0, %0, %1")
KERNEL(sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(ashr, "v_ashrrev_i32 %0, 3, %0")
KERNEL(mul_i24, "v_mul_i32_i24 %0, %0, %1")
KERNEL(mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(mad_i32_i16, "v_mad_i32_i16 %0, %0, %1, %2")
KERNEL(med3, "v_med3_i32 %0, %0, %1, %2")
KERNEL(bfe, "v_bfe_i32 %0, %0, 0, 16")
KERNEL(perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(lshl_or, "v_lshl_or_b32 %0, %0, 8, %1")
KERNEL(fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL(pk_add_i16, "v_pk_add_i16 %0, %0, %1")
KERNEL(pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2")
KERNEL(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(pk_ashr_i16, "v_pk_ashrrev_i16 %0, 2, %0")
KERNEL(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL(dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
KERNEL(dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
KERNEL(cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(mov, "v_mov_b32 %0, %1")
KERNEL(sat_pk_u8_i16, "v_sat_pk_u8_i16 %0, %0")
KERNEL(cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %0, %1")
KERNEL(max_i32, "v_max_i32 %0, %0, %1")
KERNEL(min_i32, "v_min_i32 %0, %0, %1")
KERNEL(lshlrev, "v_lshlrev_b32 %0, 1, %0")
KERNEL(and_b32, "v_and_b32 %0, %0, %1")
KERNEL(or_b32, "v_or_b32 %0, %0, %1")
KERNEL(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL(floor_f32, "v_floor_f32 %0, %0")
KERNEL(add_sdwa, "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
KERNEL(ashr_sdwa, "v_ashrrev_i32_sdwa %0, %1, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
KERNEL(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(alignbit, "v_alignbit_b32 %0, %0, %1, 8")
KERNEL(mul_i24_sdwa, "v_mul_i32_i24_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")

typedef void (*kfn)(int *, int, int);
struct Entry { const char *name; kfn fn; };
#define E(NAME) {#NAME, k_##NAME}

int main()
{
  Entry tab[] = {E(add_u32), E(sub_u32), E(lshl_add), E(add3), E(ashr), E(mul_i24), E(mad_i24), E(mul_lo), E(mad_i32_i16),
                 E(med3), E(bfe), E(perm), E(lshl_or), E(fma_f32), E(mul_f32), E(pk_add_i16), E(pk_mad_i16),
                 E(pk_mul_lo_u16), E(pk_ashr_i16), E(pk_max_i16), E(dot2_i32_i16), E(dot4_i32_i8), E(cvt_pk_u8_f32),
                 E(cndmask), E(mov), E(mul_i24_sdwa), E(sat_pk_u8_i16), E(cvt_pk_i16_i32), E(max_i32), E(min_i32), E(lshlrev), E(and_b32), E(or_b32), E(cvt_f32_i32), E(cvt_i32_f32), E(floor_f32), E(add_sdwa), E(ashr_sdwa), E(mad_u32_u24), E(alignbit)};
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8; // 8 x 256 threads = 32 waves per CU
  int *out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(int));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
  for (auto &t : tab) {
    hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double waveinst = (double)blocks * 4 * ITER * 32; // wave-instructions
    const double per_simd = waveinst / (cus * 4.0);
    // cycles per wave-instruction per SIMD at an assumed 2.4 GHz
    printf("%-16s %8.3f ms  %7.2f Gwaveinst/s  -> %.2f cycles/wave-inst/SIMD @2.4GHz (%.2f @2.1GHz)\n", t.name, ms,
           waveinst / ms / 1e6, ms * 1e-3 * 2.4e9 / per_simd, ms * 1e-3 * 2.1e9 / per_simd);
  }
  return 0;
}
