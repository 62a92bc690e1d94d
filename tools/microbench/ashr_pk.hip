// What does v_ashr_pk_u8_i32 (gfx950) write?  D[15:0] = {sat_u8(S1 >> S2), sat_u8(S0 >> S2)}; this prints what happens to D[31:16]
// (LLVM 19 of ROCm 7.2 assumes zeros when it matches the instruction from C code; the hardware keeps the old contents), and what
// op_sel:[0,0,0,1] does (writes D[31:16], keeps D[15:0]).
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench/ashr_pk.hip -o gpurun_out/ashr_pk && gpurun_out/ashr_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const int *in, unsigned *out)
{
  const int t = threadIdx.x;
  int a = in[t], b = in[t + 64], c = in[t + 128], d = in[t + 192];
  unsigned r = 0xdeadbeefu;
  asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 4" : "+v"(r) : "v"(a), "v"(b));
  out[t] = r;
  unsigned q = 0x12345678u;
  asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 4 op_sel:[0,0,0,1]" : "+v"(q) : "v"(c), "v"(d));
  out[t + 64] = q;
  unsigned w = 0xffffffffu;
  asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 17\n\tv_ashr_pk_u8_i32 %0, %3, %4, 17 op_sel:[0,0,0,1]" : "+v"(w) : "v"(a), "v"(b), "v"(c), "v"(d));
  out[t + 128] = w;
}
int main()
{
  int h[256];
  const int vals[16] = {0, 15, 16, 255 * 16, 255 * 16 + 15, 256 * 16, -1, -16, -17, 0x7fffffff, (int)0x80000000, 100 << 17, (255 << 17) + 131071, 256 << 17, -(1 << 17), 12345678};
  for (int i = 0; i < 256; i++) h[i] = vals[(i + i / 64 * 3) % 16];
  int *din; unsigned *dout; unsigned o[192];
  hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  auto sat = [](long long v) { return (unsigned)(v < 0 ? 0 : v > 255 ? 255 : v); };
  for (int t = 0; t < 16; t++) {
    const int a = h[t], b = h[t + 64], c = h[t + 128], d = h[t + 192];
    const unsigned e0 = 0xdead0000u | sat(a >> 4) | (sat(b >> 4) << 8);
    const unsigned e1 = 0x00005678u | (sat(c >> 4) << 16) | (sat(d >> 4) << 24);
    const unsigned e2 = sat(a >> 17) | (sat(b >> 17) << 8) | (sat(c >> 17) << 16) | (sat(d >> 17) << 24);
    printf("a %11d b %11d c %11d d %11d : lo %08x (keep-high model %08x) hi %08x (model %08x) both %08x (model %08x)\n", a, b, c, d, o[t], e0, o[t + 64], e1, o[t + 128], e2);
    bad += o[t] != e0 || o[t + 64] != e1 || o[t + 128] != e2;
  }
  printf("model mismatches: %d\n", bad);
  return 0;
}
