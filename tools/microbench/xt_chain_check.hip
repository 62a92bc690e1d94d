// xt_chain_check.hip -- checks the rearranged per-pixel chain of fusedxt420_kernel against the plain one of
// xt_merge_kernel on random inputs, and prints what v_cvt_pk_i16_i32 does outside the int16 range.
// hipcc --offload-arch=gfx950 -O2 -o xt_chain_check xt_chain_check.hip && ./xt_chain_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define L_CR_R 11485
#define L_CB_G 2819
#define L_CR_G 5850
#define L_CB_B 14516
__device__ int mad16_lo(unsigned pk, int k, int c) { int d; asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(pk), "s"(k), "v"(c)); return d; }
__device__ int mad16_hi(unsigned pk, int k, int c) { int d; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(pk), "s"(k), "v"(c)); return d; }
__global__ void check(const int *in, int n, int *out, unsigned *cvt)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    int a = 40000, b = -40000, c = 123, d = -7;
    asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(cvt[0]) : "v"(a), "v"(b));
    asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(cvt[1]) : "v"(c), "v"(d));
  }
  if (t >= n) return;
  // inputs: residual samples before the clamp (v + 32768 domain), legacy table values
  const int vy = in[8 * t], vb = in[8 * t + 1], vr = in[8 * t + 2];
  const int vy1 = in[8 * t + 3], vb1 = in[8 * t + 4], vr1 = in[8 * t + 5];
  const int lv = in[8 * t + 6], hi = in[8 * t + 7] & 1;
  const int out_shift = 32768, out_max = 65535, omax16 = ((out_max + 1) << 4) - 1;
  const int pinf = (out_max >> 1) - (out_max >> 6) - 1, minf = -pinf - 1;
  // plain
  int ref[3];
  {
    const int sy = hi ? vy1 : vy, sb = hi ? vb1 : vb, sr = hi ? vr1 : vr;
    const int qy = min(max(sy + 32768, 0), 65535) << 4, qb = min(max(sb + 32768, 0), 65535) << 4, qr = min(max(sr + 32768, 0), 65535) << 4;
    const int db = (qb >> 4) - out_shift, dr = (qr >> 4) - out_shift;
    int rr[3];
    rr[0] = qy + ((dr * L_CR_R + 256) >> 9);
    rr[1] = qy + ((-db * L_CB_G - dr * L_CR_G + 256) >> 9);
    rr[2] = qy + ((db * L_CB_B + 256) >> 9);
    for (int c = 0; c < 3; c++) {
      rr[c] = (min(max(rr[c], 0), omax16) + 8) >> 4;
      int m = lv + rr[c] - out_shift;
      m = min(max(m, minf), pinf);
      const short w = (short)m;
      ref[c] = (unsigned short)(short)(((w >> 15) & 0x7fff) ^ w);
    }
  }
  // rearranged
  int got[3];
  {
    unsigned rp0, rp1, rp2;
    asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(rp0) : "v"(vy), "v"(vy1));
    asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(rp1) : "v"(vb), "v"(vb1));
    asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(rp2) : "v"(vr), "v"(vr1));
    const int C = 256 + (1 << 28) + 4096;
    const int t0 = hi ? mad16_hi(rp2, L_CR_R, C) : mad16_lo(rp2, L_CR_R, C);
    const int t1 = hi ? mad16_hi(rp1, -L_CB_G, mad16_hi(rp2, -L_CR_G, C)) : mad16_lo(rp1, -L_CB_G, mad16_lo(rp2, -L_CR_G, C));
    const int t2 = hi ? mad16_hi(rp1, L_CB_B, C) : mad16_lo(rp1, L_CB_B, C);
    const int u0 = hi ? mad16_hi(rp0, 16, t0 >> 9) : mad16_lo(rp0, 16, t0 >> 9);
    const int u1 = hi ? mad16_hi(rp0, 16, t1 >> 9) : mad16_lo(rp0, 16, t1 >> 9);
    const int u2 = hi ? mad16_hi(rp0, 16, t2 >> 9) : mad16_lo(rp0, 16, t2 >> 9);
    int r2[3] = {min(max(u0 >> 4, 0), out_max + 1), min(max(u1 >> 4, 0), out_max + 1), min(max(u2 >> 4, 0), out_max + 1)};
    const unsigned pinf2 = (unsigned)pinf * 0x10001u, minf2 = ((unsigned)minf & 0xffffu) * 0x10001u;
    for (int c = 0; c < 3; c++) {
      int m0 = (lv - out_shift) + r2[c], m1 = m0;
      if (hi) m0 = 12345; // the sample under test travels in the half the input selects
      unsigned pk, sg;
      asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(pk) : "v"(m0), "v"(m1));
      asm("v_pk_max_i16 %0, %1, %2" : "=v"(pk) : "v"(pk), "v"(minf2));
      asm("v_pk_min_i16 %0, %1, %2" : "=v"(pk) : "v"(pk), "v"(pinf2));
      asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(sg) : "v"(pk)); // the constant has no high half of its own
      got[c] = ((pk ^ (sg & 0x7fff7fffu)) >> (hi ? 16 : 0)) & 0xffffu;
    }
  }
  for (int c = 0; c < 3; c++) { out[6 * t + c] = ref[c]; out[6 * t + 3 + c] = got[c]; }
}
int main()
{
  const int n = 1 << 20;
  std::vector<int> in(8 * n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
  for (int i = 0; i < n; i++) {
    const int spread = (i & 3) == 0 ? 80000 : (i & 3) == 1 ? 4000 : 66000;
    for (int k = 0; k < 6; k++) in[8 * i + k] = (int)(rnd() % (2 * spread)) - spread;
    in[8 * i + 6] = (int)(rnd() % 65536);
    in[8 * i + 7] = (int)rnd();
  }
  int *din, *dout; unsigned *dc;
  hipMalloc(&din, in.size() * 4); hipMalloc(&dout, 6 * n * 4); hipMalloc(&dc, 16);
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, din, n, dout, dc);
  std::vector<int> out(6 * n); unsigned cvt[2];
  hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(cvt, dc, 8, hipMemcpyDeviceToHost);
  printf("v_cvt_pk_i16_i32(40000, -40000) = %08x, (123, -7) = %08x\n", cvt[0], cvt[1]);
  long bad = 0;
  for (int i = 0; i < n; i++)
    for (int c = 0; c < 3; c++)
      if (out[6 * i + c] != out[6 * i + 3 + c]) {
        if (bad < 8) printf("mismatch at %d c=%d: ref %04x got %04x; in vy %d/%d vb %d/%d vr %d/%d lv %d hi %d\n", i, c, out[6 * i + c], out[6 * i + 3 + c],
                            in[8 * i], in[8 * i + 3], in[8 * i + 1], in[8 * i + 4], in[8 * i + 2], in[8 * i + 5], in[8 * i + 6], in[8 * i + 7] & 1);
        bad++;
      }
  printf("%ld mismatches of %d\n", bad, 3 * n);
  return bad != 0;
}
