// kernels.hip -- CDNA4 (gfx950) kernels of the JPEG block-reconstruction path.
//
// What they replace in the reference (thorfdbg/libjpeg), all integer arithmetic, bit-exact:
//   dequant + 8x8 inverse DCT   dct/idct.cpp:226-339            (IDCT<4,LONG,false,false>)
//   centred chroma upsampling   upsampling/upsampler.cpp:83-117, 136-168, 283-307 (+ 3x/4x cores)
//                               upsampling/upsamplerbase.cpp:300-327 (edge replication)
//   YCbCr->RGB / identity       colortrafo/ycbcrtrafo.cpp:842-856, 921-936, 974-1006
//   the block loop around them  control/blockbitmaprequester.cpp:1013-1224
//
// Two families:
//   fused420_kernel   the hot path (4:2:0, three components): one launch turns int16 coefficient planes
//                     into interleaved RGB; chroma never leaves the CU (LDS), luma never leaves registers.
//   generic kernels   any sampling 1..4 x 1..4, 1..4 components: IDCT to int32 planes, then a per-line
//                     upsample+colour kernel.  Correct everywhere, used for everything that is not 4:2:0.
//
// Mapping (MI355X-first): ONE LANE OWNS ONE 8x8 BLOCK.  Both IDCT passes run out of the
// lane's registers (64 VGPRs), so there is no transposition and no cross-lane traffic in the transform;
// the coefficient rows reach the owning lane through a 2 KB-per-wave LDS staging buffer that is fed by
// fully coalesced 16-byte loads (64 lanes x 16 B = 1 KiB contiguous per instruction).
//
// Arithmetic flavours: FAST uses 24-bit multiplies (v_mul_i32_i24 / v_mad_i32_i24, full rate) and 32-bit
// colour accumulation; it is selected by the host only when every block of the frame satisfies
// sum_k |c_k| * q_k < 16384, which bounds every intermediate below 2^31 and every multiplicand below 2^23
// (DESIGN.md "range check"); every 8-bit image produced by a forward DCT satisfies it.  SAFE reproduces
// the reference's wrap-around LONG arithmetic and its 64-bit colour accumulation for any input.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.hpp"

namespace mij {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// the same at any byte address: lines of the caller's bitmap start wherever its row stride puts them, and gfx950 stores
// dwordx2 / dwordx4 at any address (tools/microbench/unaligned_store.hip); the only thing the fast stores need is a full group
typedef u32x2 u32x2_any __attribute__((aligned(1)));
typedef u32x4 u32x4_any __attribute__((aligned(1)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------------
template <bool FAST>
__device__ __forceinline__ int mulc(int a, int c)
{
  if (FAST) return __mul24(a, c);
  return (int)((unsigned)a * (unsigned)c);
}
__device__ __forceinline__ int addw(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int subw(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int shlw(int a, int n) { return (int)((unsigned)a << n); }

// Hand-picked instructions (rates measured with tools/microbench/valu_rate.hip: every VOP3 integer op and
// every multiply issues once per ~4 cycles per SIMD, so the instruction COUNT is what the kernel pays for):
//   v_mad_i32_i24  d = a(24 bit) * k + c            multiply-accumulate in one issue slot
//   v_mad_i32_i16  d = half(w) * q + 0              unpacks a 16-bit coefficient and dequantises it in one
//   v_cvt_pk_i16_i32 + v_sat_pk_u8_i16              clamp to [0,255] and pack two samples per instruction
__device__ __forceinline__ int mad24(int a, int k, int c)
{
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c));
  return d;
}
// (the same with a per-lane factor)
__device__ __forceinline__ int mad24v(int a, int b, int c)
{
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int mul16_lo(unsigned w, int q)
{
  int d;
  asm("v_mad_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(w), "s"(q));
  return d;
}
__device__ __forceinline__ int mul16_hi(unsigned w, int q)
{
  int d;
  asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[1,0,0,0]" : "=v"(d) : "v"(w), "s"(q));
  return d;
}
// d = half(pk) * k + c with k an SGPR constant that fits 16 bits: consumes a packed int16 sample without unpacking
__device__ __forceinline__ int mad16_lo(unsigned pk, int k, int c)
{
  int d;
  asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(pk), "s"(k), "v"(c));
  return d;
}
__device__ __forceinline__ int mad16_hi(unsigned pk, int k, int c)
{
  int d;
  asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(pk), "s"(k), "v"(c));
  return d;
}

// d = lo(pk) * klo + hi(pk) * khi + c in ONE issue slot (v_dot2_i32_i16): the two chroma terms of the green channel when a
// position's (Cb, Cr) travel as the halves of one register
__device__ __forceinline__ int dot2_16(unsigned pk, int klo, int khi, int c)
{
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 k = {(short)klo, (short)khi};
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pk), k, c, false);
}

// {lo16(hi), lo16(lo)} -> one register.  volatile: the packing must happen where it is written (right after the
// transform), otherwise the compiler keeps all 64 unpacked samples alive and packs lazily at the use.
__device__ __forceinline__ unsigned pack_lo16_now(int hi, int lo)
{
  unsigned d;
  asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(hi), "v"(lo), "s"(0x05040100u));
  return d;
}

// (a >> SH, b >> SH, c >> SH, d >> SH), each clamped to [0,255], as the four bytes of a dword (a lowest): gfx950's
// v_ashr_pk_u8_i32 shifts, saturates and packs TWO samples into 16 bits of its destination and leaves the other 16 alone
// (tools/microbench/ashr_pk.hip prints what the hardware does), op_sel:[0,0,0,1] names the upper half: half an instruction
// per sample.  (Always through this helper: matched from C code by the compiler of ROCm 7.2, the instruction's result is
// taken to have zeros in its upper half, which the hardware does not write.)
template <int SH>
__device__ __forceinline__ unsigned ashr_sat_pack4(int a, int b, int c, int d)
{
  unsigned w;
  asm("v_ashr_pk_u8_i32 %0, %1, %2, %5\n\tv_ashr_pk_u8_i32 %0, %3, %4, %5 op_sel:[0,0,0,1]" : "=&v"(w) : "v"(a), "v"(b), "v"(c), "v"(d), "n"(SH));
  return w;
}
// the 24 bytes r0 g0 b0 r1 ... b7 of eight pixels from their colour sums (17 fraction bits, rounding inside)
__device__ __forceinline__ void rgb_shift17_sat_pack(const int (&rr)[8], const int (&gg)[8], const int (&bb)[8], unsigned (&w)[6])
{
#pragma unroll
  for (int x = 0; x < 8; x += 4) {
    w[3 * (x / 4) + 0] = ashr_sat_pack4<17>(rr[x], gg[x], bb[x], rr[x + 1]);
    w[3 * (x / 4) + 1] = ashr_sat_pack4<17>(gg[x + 1], bb[x + 1], rr[x + 2], gg[x + 2]);
    w[3 * (x / 4) + 2] = ashr_sat_pack4<17>(bb[x + 2], rr[x + 3], gg[x + 3], bb[x + 3]);
  }
}

// ... and their way out: 24 bytes at base + off, base uniform (SGPR pair), off per lane (32 bits, the fused kernels' frames
// fit: fits32 in capi.cpp).  The store instructions take the pair and the offset as they are (`saddr` form); written as a
// pointer the compiler widens the offset and adds in 64 bits first, three VALU instructions per line of a block.
// NT = false: ordinary stores.  A frame whose lines are not multiples of 128 bytes long has tile rows that share their first and
// last cache line with the neighbouring tiles' rows; streaming stores push those half-written lines out one half at a time,
// ordinary ones let the L2 put the halves together first (the neighbouring tiles run on the same XCD, one after the other).
template <bool NT = true>
__device__ __forceinline__ void store24_nt(uint8_t *__restrict__ base, unsigned off, const unsigned (&w)[6])
{
  const u32x4 lo = {w[0], w[1], w[2], w[3]};
  const u32x2 hi = {w[4], w[5]};
  if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\tglobal_store_dwordx2 %0, %3, %2 offset:16 nt" : : "v"(off), "v"(lo), "s"(base), "v"(hi) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2\n\tglobal_store_dwordx2 %0, %3, %2 offset:16" : : "v"(off), "v"(lo), "s"(base), "v"(hi) : "memory");
}
// 32 bytes at base + off (uniform base, 32-bit lane offset; any alignment), non-temporal.  One asm statement for both stores and
// a wait state behind them: the compiler does not see stores in inline asm, so it does not keep the next VALU instruction from
// overwriting the registers a store of more than 64 bits has yet to read (the first version of this, one statement per store,
// got the second store's offset as its first pixel).  store24_nt: the 8-byte store behind the 16-byte one is that wait state.
__device__ __forceinline__ void store32_nt(uint8_t *__restrict__ base, unsigned off, const unsigned (&w)[8])
{
  const u32x4 lo = {w[0], w[1], w[2], w[3]}, hi = {w[4], w[5], w[6], w[7]};
  asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\tglobal_store_dwordx4 %0, %3, %2 offset:16 nt\n\ts_nop 0" : : "v"(off), "v"(lo), "s"(base), "v"(hi) : "memory");
}

// 16 bytes at base + off, non-temporal (one asm statement with its wait state, as above)
__device__ __forceinline__ void store16_nt(uint8_t *__restrict__ base, unsigned off, const u32x4 v)
{
  asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : : "v"(off), "v"(v), "s"(base) : "memory");
}

// The same 24 bytes when the line's address is NOT a multiple of four (m = address & 3: a packed frame whose width is not a
// multiple of four pixels has such lines): gfx950 stores dwords at any address, but at a third more time per launch
// (`profiles/r03/layouts.txt`, W = 7678).  The sixteen lanes of a row own 384 consecutive bytes; each lane takes the last m bytes
// of its left neighbour (one DPP move, row_shr:1), shifts its six dwords by m bytes (v_alignbyte_b32) and stores 24 bytes at
// the dword boundary m bytes in front of its own position.  The row's first lane has no neighbour inside the wave: it stores
// its first 4 - m bytes as bytes and the rest aligned; the row's last lane adds the m bytes that are left over.  bx = lane & 15;
// every lane of the wave must be active (the caller checks), m is wave-uniform and 1..3.
__device__ __forceinline__ void store24_nt_shifted(uint8_t *__restrict__ base, unsigned off, const unsigned (&w)[6], int bx, int m)
{
  const unsigned left = (unsigned)__builtin_amdgcn_update_dpp((int)w[5], (int)w[5], 0x111, 0xF, 0xF, false); // row_shr:1
  const unsigned sh = (unsigned)(4 - m);
  unsigned d[6];
  d[0] = __builtin_amdgcn_alignbyte(w[0], left, sh);
#pragma unroll
  for (int k = 1; k < 6; k++) d[k] = __builtin_amdgcn_alignbyte(w[k], w[k - 1], sh);
  const unsigned at = off - (unsigned)m; // a multiple of four
  if (bx != 0) store24_nt<false>(base, at, d);
  else {
    // bytes m..3 of the first dword are this lane's own first 4 - m bytes
    uint8_t *p = base + off;
    if (m == 1) { p[0] = (uint8_t)w[0]; *reinterpret_cast<uint16_t *>(p + 1) = (uint16_t)(w[0] >> 8); }
    else if (m == 2) *reinterpret_cast<uint16_t *>(p) = (uint16_t)w[0];
    else p[0] = (uint8_t)w[0];
    u32x4 *q = reinterpret_cast<u32x4 *>(base + (at + 4)); // (dword aligned; 16-byte alignment is not needed)
    *reinterpret_cast<u32x4_any *>(q) = u32x4{d[1], d[2], d[3], d[4]};
    *reinterpret_cast<unsigned *>(base + (at + 20)) = d[5];
  }
  if (bx == 15) { // the last m bytes of the row
    uint8_t *p = base + (at + 24);
    const unsigned t = w[5] >> (8 * sh);
    if (m == 1) p[0] = (uint8_t)t;
    else if (m == 2) *reinterpret_cast<uint16_t *>(p) = (uint16_t)t;
    else { *reinterpret_cast<uint16_t *>(p) = (uint16_t)t; p[2] = (uint8_t)(t >> 16); }
  }
}

// floor((x + 2^(n-1)) / 2^n) with the addition carried out beyond 32 bits, as the reference's
// `(x + (1L << (n-1))) >> n` does on LP64 (dct/idct.cpp:70-78).
template <bool FAST, int N>
__device__ __forceinline__ int round_shift(int x)
{
  if (FAST) return (x + (1 << (N - 1))) >> N;
  return (x >> N) + (((x & ((1 << N) - 1)) + (1 << (N - 1))) >> N);
}

// TO_FIX(x) = WORD(x * 512 + 0.5), dct/idct.cpp:65
#define FIX9(x) ((int)((x) * 512.0 + 0.5))

// One 8-point inverse transform (dct/idct.cpp:233-283 / :287-332), SHIFT = 9 (rows) or 12 (columns).
// Before the final shift everything is exact integer arithmetic modulo 2^32, so any algebraically equal
// arrangement gives the reference's bits.  The FAST flavour is arranged for the fewest issue slots
// (44 per transform): products are folded into nested v_mad_i32_i24 chains, e.g. the reference's
//   tmp0 = ttmp0 * c5 + z1 + z3  with  z1 = (ttmp0 + ttmp3) * -c9,  z3 = (ttmp0 + ttmp2) * -c11 + z5
// becomes mad(tz1, -c9, mad(ttmp0, c5, z3)); the rounding constant is folded into the even part.
// NZ = 4 (FAST only): s4..s7 are known to be zero and are not read; every term they feed is dropped and constants that
// multiply the same input are added up (still exact: the ring Z / 2^32 is distributive), 31 slots instead of 44.
// R: what the even part starts from -- the rounding constant 2^(SHIFT-1), plus whatever multiple of 2^SHIFT the caller wants
// added to every output for free (FAST flavours only; LUMA_FOLD_R below).
// X2 (FAST only): every constant of the pass doubled and no final shift -- the outputs are TWICE the sums in front of the
// rounding shift, exactly (the pass is linear over Z / 2^32, and v_mad_i32_i24 sees the same 24-bit inputs and constants below
// 2^13).  What the luma of the fast 8-bit YCbCr kernels wants: the colour stage needs y << 13 = (sum >> 12) << 13, which is the
// doubled sum with its low 13 bits cleared -- one v_and_b32 per sample where the single sum takes a shift and the mask (R then
// holds twice the constants, see LUMA_FOLD_R2).
template <bool FAST, int SHIFT, int NZ = 8, bool X2 = false>
__device__ __forceinline__ void idct_1d(int &s0, int &s1, int &s2, int &s3, int &s4, int &s5, int &s6, int &s7, const int R = 1 << (SHIFT - 1))
{
  constexpr int M = X2 ? 2 : 1, OSH = X2 ? 0 : SHIFT;
  static_assert(!X2 || FAST, "the doubled pass is a FAST flavour");
  if (FAST && NZ == 4) {
    // even part (7): tmp2 = z1, tmp3 = s2 * (c0.541 + c0.765)
    const int t0 = (s0 << (X2 ? 10 : 9)) + R;
    const int tmp2 = __mul24(s2, M * FIX9(0.541196100));
    const int tmp3 = __mul24(s2, M * (FIX9(0.541196100) + FIX9(0.765366865)));
    const int t10 = t0 + tmp3, t13 = t0 - tmp3, t11 = t0 + tmp2, t12 = t0 - tmp2;
    // odd part (8): tz1 = tz4 = s1, tz2 = tz3 = s3
    const int z5 = __mul24(s3 + s1, M * FIX9(1.175875602));
    const int z3 = mad24(s3, M * -FIX9(1.961570560), z5);
    const int z4 = mad24(s1, M * -FIX9(0.390180644), z5);
    const int o0 = mad24(s1, M * -FIX9(0.899976223), z3);
    const int o1 = mad24(s3, M * -FIX9(2.562915447), z4);
    const int o2 = mad24(s3, M * (FIX9(3.072711026) - FIX9(2.562915447)), z3);
    const int o3 = mad24(s1, M * (FIX9(1.501321110) - FIX9(0.899976223)), z4);
    s0 = (t10 + o3) >> OSH; s7 = (t10 - o3) >> OSH;
    s1 = (t11 + o2) >> OSH; s6 = (t11 - o2) >> OSH;
    s2 = (t12 + o1) >> OSH; s5 = (t12 - o1) >> OSH;
    s3 = (t13 + o0) >> OSH; s4 = (t13 - o0) >> OSH;
    return;
  }
  if (FAST) {
    // even part (12)
    const int t0 = ((s0 + s4) << (X2 ? 10 : 9)) + R;
    const int t1 = ((s0 - s4) << (X2 ? 10 : 9)) + R;
    const int z1 = __mul24(s2 + s6, M * FIX9(0.541196100));
    const int tmp2 = mad24(s6, M * -FIX9(1.847759065), z1);
    const int tmp3 = mad24(s2, M * FIX9(0.765366865), z1);
    const int t10 = t0 + tmp3, t13 = t0 - tmp3, t11 = t1 + tmp2, t12 = t1 - tmp2;
    // odd part (16)
    const int tz1 = s7 + s1, tz2 = s5 + s3, tz3 = s7 + s3, tz4 = s5 + s1;
    const int z5 = __mul24(tz3 + tz4, M * FIX9(1.175875602));
    const int z3 = mad24(tz3, M * -FIX9(1.961570560), z5);
    const int z4 = mad24(tz4, M * -FIX9(0.390180644), z5);
    const int o0 = mad24(tz1, M * -FIX9(0.899976223), mad24(s7, M * FIX9(0.298631336), z3));
    const int o1 = mad24(tz2, M * -FIX9(2.562915447), mad24(s5, M * FIX9(2.053119869), z4));
    const int o2 = mad24(tz2, M * -FIX9(2.562915447), mad24(s3, M * FIX9(3.072711026), z3));
    const int o3 = mad24(tz1, M * -FIX9(0.899976223), mad24(s1, M * FIX9(1.501321110), z4));
    // outputs (16)
    s0 = (t10 + o3) >> OSH; s7 = (t10 - o3) >> OSH;
    s1 = (t11 + o2) >> OSH; s6 = (t11 - o2) >> OSH;
    s2 = (t12 + o1) >> OSH; s5 = (t12 - o1) >> OSH;
    s3 = (t13 + o0) >> OSH; s4 = (t13 - o0) >> OSH;
    return;
  }
  // SAFE: the reference's statement order in wrapping 32-bit arithmetic
  int z1 = mulc<false>(addw(s2, s6), FIX9(0.541196100));
  int tmp2 = addw(z1, mulc<false>(s6, -FIX9(1.847759065)));
  int tmp3 = addw(z1, mulc<false>(s2, FIX9(0.765366865)));
  int tmp0 = shlw(addw(s0, s4), 9);
  int tmp1 = shlw(subw(s0, s4), 9);
  int tmp10 = addw(tmp0, tmp3), tmp13 = subw(tmp0, tmp3);
  int tmp11 = addw(tmp1, tmp2), tmp12 = subw(tmp1, tmp2);
  int tz1 = addw(s7, s1), tz2 = addw(s5, s3), tz3 = addw(s7, s3), tz4 = addw(s5, s1);
  int z5 = mulc<false>(addw(tz3, tz4), FIX9(1.175875602));
  int o0 = mulc<false>(s7, FIX9(0.298631336));
  int o1 = mulc<false>(s5, FIX9(2.053119869));
  int o2 = mulc<false>(s3, FIX9(3.072711026));
  int o3 = mulc<false>(s1, FIX9(1.501321110));
  int y1 = mulc<false>(tz1, -FIX9(0.899976223));
  int y2 = mulc<false>(tz2, -FIX9(2.562915447));
  int y3 = addw(mulc<false>(tz3, -FIX9(1.961570560)), z5);
  int y4 = addw(mulc<false>(tz4, -FIX9(0.390180644)), z5);
  o0 = addw(o0, addw(y1, y3));
  o1 = addw(o1, addw(y2, y4));
  o2 = addw(o2, addw(y2, y3));
  o3 = addw(o3, addw(y1, y4));
  s0 = round_shift<false, SHIFT>(addw(tmp10, o3)); s7 = round_shift<false, SHIFT>(subw(tmp10, o3));
  s1 = round_shift<false, SHIFT>(addw(tmp11, o2)); s6 = round_shift<false, SHIFT>(subw(tmp11, o2));
  s2 = round_shift<false, SHIFT>(addw(tmp12, o1)); s5 = round_shift<false, SHIFT>(subw(tmp12, o1));
  s3 = round_shift<false, SHIFT>(addw(tmp13, o0)); s4 = round_shift<false, SHIFT>(subw(tmp13, o0));
}

// Dequantise the eight packed rows of a block and run both passes in registers.
// rows[k] holds coefficient row k (8 x int16, natural order), q the component's 64 deltas << 4 (idct.cpp:98-109).
// Result v[y*8+x] = sample * 16 (COLOR_BITS = 4 fractional bits), not clamped.
// dcoff = 0 leaves out the level shift dcoffset = 2^(P-1) << 7 (idct.cpp:231, :246).  That constant
// passes through both rounding shifts exactly ((x + 2^14 * 2^9 + 2^8) >> 9 = ((x + 2^8) >> 9) + 2^14, and
// (x + 2^14 * 2^9 + 2^11) >> 12 = ((x + 2^11) >> 12) + 2^11), so the result is exactly 2048 lower.
// NR = 4 (FAST only): coefficient rows 4..7 are known to be zero (see rows_4_to_7_zero): they are neither dequantised nor
// transformed, and the second pass runs the pruned butterfly.
// NC = 4 (with NR = 4): coefficient columns 4..7 are zero too; the first pass is pruned the same way.
// Luma of the fast 8-bit YCbCr kernels: the colour stage wants y * 8192 + (2048 << 13) + 65536 = (y + 2048 + 8) << 13 -- level
// shift and the rounding of its >> 17 -- and a multiple of 4096 added in front of the second pass's >> 12 comes out as that
// multiple / 4096 on every sample, exactly: the two constants ride in the pass's rounding constant and the colour stage is left
// with the shift.
constexpr int LUMA_FOLD_R = 2048 + ((2048 + 8) << 12);
// ... and with the second pass doubled (idct_1d X2) the outputs are 2 * (sum + LUMA_FOLD_R): luma13() of one is the y << 13 above
constexpr int LUMA_FOLD_R2 = 2 * LUMA_FOLD_R;
__device__ __forceinline__ int luma13(int doubled_sum) { return doubled_sum & (int)0xffffe000; }
template <int NR, bool X2> __device__ __forceinline__ void idct_columns_dot2(int (&v)[64], const int R);
template <bool FAST, int NR = 8, int NC = 8, bool X2 = false, bool D2 = false>
__device__ __forceinline__ void dequant_idct(const u32x4 (&rows)[8], const int *__restrict__ q, int (&v)[64], int dcoff, const int colr = 2048)
{
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const unsigned w[4] = {rows[k].x, rows[k].y, rows[k].z, rows[k].w};
#pragma unroll
    for (int i = 0; i < NC / 2; i++) {
      if (FAST) { // deltas <= 2047 (checked by the host): q fits a signed 16-bit operand
        v[k * 8 + 2 * i] = mul16_lo(w[i], q[k * 8 + 2 * i]);
        v[k * 8 + 2 * i + 1] = mul16_hi(w[i], q[k * 8 + 2 * i + 1]);
      } else {
        v[k * 8 + 2 * i] = mulc<false>((int)(short)(w[i] & 0xffffu), q[k * 8 + 2 * i]);
        v[k * 8 + 2 * i + 1] = mulc<false>(((int)w[i]) >> 16, q[k * 8 + 2 * i + 1]);
      }
    }
  }
  v[0] = addw(v[0], dcoff); // level shift 2^(P-1) << (preshift + 3); the fused fast kernels pass 0
#pragma unroll
  for (int r = 0; r < NR; r++)
    idct_1d<FAST, 9, NC>(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5],
                         v[r * 8 + 6], v[r * 8 + 7]);
  if constexpr (D2) { idct_columns_dot2<NR, X2>(v, colr); return; }
#pragma unroll
  for (int c = 0; c < 8; c++)
    idct_1d<FAST, 12, NR, X2>(v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c], colr);
}

// The same with the FIRST pass in 16-bit arithmetic (FAST, 8-bit frames: the range check bounds sum |c| delta of a block by
// 16384, so every dequantised coefficient -- without the << 4 the reference's deltas carry -- and the level shift 1024 on top
// of the DC term fit 16 bits).  Before its rounding shift the pass is a product of the eight inputs with a matrix of integers
// (sums of the butterfly's 9-bit constants, |entry| <= 710): written out as v_dot2_i32_i16 inner products over the pairs
// (s0,s4) (s2,s6) for the even half and (s1,s5) (s3,s7) for the odd half, the accumulator of one feeding the next --
// exact, the ring Z / 2^32 is distributive -- and the factor 16 that left the deltas taken out of the rounding:
// (16 x + 256) >> 9 = (x + 16) >> 5.  v_perm_b32 pairs the coefficients as the products want them, v_pk_mul_lo_u16
// dequantises two at a time with the packed deltas behind q[64] (fill_deltas): 38 issue slots per row where
// dequantisation + butterfly took 52; with columns 4..7 zero 28 for 35.  The second pass sees 18-bit values and stays.
__device__ __forceinline__ unsigned pk_mul_lo16(unsigned a, unsigned q)
{
  unsigned d;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(q));
  return d;
}
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b)
{
  unsigned d;
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b));
  return d;
}
#define PK16(lo, hi) ((int)(((unsigned)(lo) & 0xffffu) | ((unsigned)(hi) << 16)))
__device__ __forceinline__ int sdot2(unsigned pk, int k, int c)
{
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pk), __builtin_bit_cast(s16x2, k), c, false);
}
// The SECOND pass as v_dot2_i32_i16 inner products (D2 flavours; 8-bit FAST frames whose first-pass results fit 16 bits: the host
// admits them by sum |c| q <= 1476 per block -- a first-pass output is (sum_x s_x M_xj + 16) >> 5 with |M_xj| <= 710, the largest
// entry of the butterfly's integer matrix, and (710 * 1476 + 16) >> 5 = 32749; the level shift is not in it, dcoff = 0).  The pass
// is the same product of its eight inputs with that matrix as the first one (dequant_idct16 has the derivation): the rows of a
// column are paired (0,4) (2,6) (1,5) (3,7) by v_perm_b32 -- (0,2) (1,3) when rows 4..7 are zero -- and every butterfly term is
// one dot product that takes the previous one as its accumulator; 22 issue slots per column where the mad24 butterfly takes 36
// (X2) / 44, plus four packings.  Exact: the ring Z / 2^32 is distributive, the operands are the same integers.
// R, X2 as in idct_1d (the doubled pass: every constant twice, no final shift).
template <int NR, bool X2>
__device__ __forceinline__ void idct_columns_dot2(int (&v)[64], const int R)
{
  constexpr int M = X2 ? 2 : 1, OSH = X2 ? 0 : 12;
  constexpr int C0541 = M * FIX9(0.541196100), C0765 = M * FIX9(0.765366865), C1847 = M * FIX9(1.847759065), C1175 = M * FIX9(1.175875602),
                C1961 = M * FIX9(1.961570560), C0390 = M * FIX9(0.390180644), C0899 = M * FIX9(0.899976223), C0298 = M * FIX9(0.298631336),
                C2562 = M * FIX9(2.562915447), C2053 = M * FIX9(2.053119869), C3072 = M * FIX9(3.072711026), C1501 = M * FIX9(1.501321110),
                ONE = M * 512;
#pragma unroll
  for (int c = 0; c < 8; c++) {
    int t10, t11, t12, t13, o0, o1, o2, o3;
    if (NR == 8) {
      const unsigned s04 = pack_lo16_now(v[32 + c], v[c]), s15 = pack_lo16_now(v[40 + c], v[8 + c]);
      const unsigned s26 = pack_lo16_now(v[48 + c], v[16 + c]), s37 = pack_lo16_now(v[56 + c], v[24 + c]);
      const int A = sdot2(s04, PK16(ONE, ONE), R), B = sdot2(s04, PK16(ONE, -ONE), R);
      t10 = sdot2(s26, PK16(C0541 + C0765, C0541), A);
      t13 = sdot2(s26, PK16(-(C0541 + C0765), -C0541), A);
      t11 = sdot2(s26, PK16(C0541, C0541 - C1847), B);
      t12 = sdot2(s26, PK16(-C0541, C1847 - C0541), B);
      o0 = sdot2(s37, PK16(C1175 - C1961, -C0899 + C0298 + C1175 - C1961), sdot2(s15, PK16(-C0899 + C1175, C1175), 0));
      o1 = sdot2(s37, PK16(-C2562 + C1175, C1175), sdot2(s15, PK16(C1175 - C0390, -C2562 + C2053 + C1175 - C0390), 0));
      o2 = sdot2(s37, PK16(-C2562 + C3072 + C1175 - C1961, C1175 - C1961), sdot2(s15, PK16(C1175, -C2562 + C1175), 0));
      o3 = sdot2(s37, PK16(C1175, -C0899 + C1175), sdot2(s15, PK16(-C0899 + C1501 + C1175 - C0390, C1175 - C0390), 0));
    } else {
      const unsigned s02 = pack_lo16_now(v[16 + c], v[c]), s13 = pack_lo16_now(v[24 + c], v[8 + c]);
      t10 = sdot2(s02, PK16(ONE, C0541 + C0765), R);
      t13 = sdot2(s02, PK16(ONE, -(C0541 + C0765)), R);
      t11 = sdot2(s02, PK16(ONE, C0541), R);
      t12 = sdot2(s02, PK16(ONE, -C0541), R);
      o0 = sdot2(s13, PK16(-C0899 + C1175, C1175 - C1961), 0);
      o1 = sdot2(s13, PK16(C1175 - C0390, -C2562 + C1175), 0);
      o2 = sdot2(s13, PK16(C1175, -C2562 + C3072 + C1175 - C1961), 0);
      o3 = sdot2(s13, PK16(-C0899 + C1501 + C1175 - C0390, C1175), 0);
    }
    v[c] = (t10 + o3) >> OSH; v[56 + c] = (t10 - o3) >> OSH;
    v[8 + c] = (t11 + o2) >> OSH; v[48 + c] = (t11 - o2) >> OSH;
    v[16 + c] = (t12 + o1) >> OSH; v[40 + c] = (t12 - o1) >> OSH;
    v[24 + c] = (t13 + o0) >> OSH; v[32 + c] = (t13 - o0) >> OSH;
  }
}

template <int NR, int NC, bool X2 = false, bool D2 = false>
__device__ __forceinline__ void dequant_idct16(const u32x4 (&rows)[8], const int *__restrict__ q, int (&v)[64], int dcoff, const int colr = 2048)
{
  constexpr int C0541 = FIX9(0.541196100), C0765 = FIX9(0.765366865), C1847 = FIX9(1.847759065), C1175 = FIX9(1.175875602),
                C1961 = FIX9(1.961570560), C0390 = FIX9(0.390180644), C0899 = FIX9(0.899976223), C0298 = FIX9(0.298631336),
                C2562 = FIX9(2.562915447), C2053 = FIX9(2.053119869), C3072 = FIX9(3.072711026), C1501 = FIX9(1.501321110);
  const unsigned *__restrict__ qp = reinterpret_cast<const unsigned *>(q + QROW_PACKED);
  const unsigned dc16 = (unsigned)(dcoff >> 4) & 0xffffu; // the level shift, added to the DC term (low half of the first pair of row 0)
#pragma unroll
  for (int k = 0; k < NR; k++) {
    int t10, t11, t12, t13, o0, o1, o2, o3;
    if (NC == 8) {
      // pairs (c0,c4) (c1,c5) (c2,c6) (c3,c7) from the row's dwords (c0,c1) (c2,c3) (c4,c5) (c6,c7)
      unsigned s04 = pk_mul_lo16(__builtin_amdgcn_perm(rows[k].z, rows[k].x, 0x05040100u), qp[k * 4 + 0]);
      const unsigned s15 = pk_mul_lo16(__builtin_amdgcn_perm(rows[k].z, rows[k].x, 0x07060302u), qp[k * 4 + 1]);
      const unsigned s26 = pk_mul_lo16(__builtin_amdgcn_perm(rows[k].w, rows[k].y, 0x05040100u), qp[k * 4 + 2]);
      const unsigned s37 = pk_mul_lo16(__builtin_amdgcn_perm(rows[k].w, rows[k].y, 0x07060302u), qp[k * 4 + 3]);
      if (k == 0) s04 = pk_add16(s04, dc16);
      const int A = sdot2(s04, PK16(512, 512), 16), B = sdot2(s04, PK16(512, -512), 16); // (s0 +- s4) << 9, rounding inside
      t10 = sdot2(s26, PK16(C0541 + C0765, C0541), A);
      t13 = sdot2(s26, PK16(-(C0541 + C0765), -C0541), A);
      t11 = sdot2(s26, PK16(C0541, C0541 - C1847), B);
      t12 = sdot2(s26, PK16(-C0541, C1847 - C0541), B);
      o0 = sdot2(s37, PK16(C1175 - C1961, -C0899 + C0298 + C1175 - C1961), sdot2(s15, PK16(-C0899 + C1175, C1175), 0));
      o1 = sdot2(s37, PK16(-C2562 + C1175, C1175), sdot2(s15, PK16(C1175 - C0390, -C2562 + C2053 + C1175 - C0390), 0));
      o2 = sdot2(s37, PK16(-C2562 + C3072 + C1175 - C1961, C1175 - C1961), sdot2(s15, PK16(C1175, -C2562 + C1175), 0));
      o3 = sdot2(s37, PK16(C1175, -C0899 + C1175), sdot2(s15, PK16(-C0899 + C1501 + C1175 - C0390, C1175 - C0390), 0));
    } else {
      // columns 4..7 are zero: pairs (c0,c2) (c1,c3)
      unsigned s02 = pk_mul_lo16(__builtin_amdgcn_perm(rows[k].y, rows[k].x, 0x05040100u), qp[32 + k * 2 + 0]);
      const unsigned s13 = pk_mul_lo16(__builtin_amdgcn_perm(rows[k].y, rows[k].x, 0x07060302u), qp[32 + k * 2 + 1]);
      if (k == 0) s02 = pk_add16(s02, dc16);
      t10 = sdot2(s02, PK16(512, C0541 + C0765), 16);
      t13 = sdot2(s02, PK16(512, -(C0541 + C0765)), 16);
      t11 = sdot2(s02, PK16(512, C0541), 16);
      t12 = sdot2(s02, PK16(512, -C0541), 16);
      o0 = sdot2(s13, PK16(-C0899 + C1175, C1175 - C1961), 0);
      o1 = sdot2(s13, PK16(C1175 - C0390, -C2562 + C1175), 0);
      o2 = sdot2(s13, PK16(C1175, -C2562 + C3072 + C1175 - C1961), 0);
      o3 = sdot2(s13, PK16(-C0899 + C1501 + C1175 - C0390, C1175), 0);
    }
    v[k * 8 + 0] = (t10 + o3) >> 5; v[k * 8 + 7] = (t10 - o3) >> 5;
    v[k * 8 + 1] = (t11 + o2) >> 5; v[k * 8 + 6] = (t11 - o2) >> 5;
    v[k * 8 + 2] = (t12 + o1) >> 5; v[k * 8 + 5] = (t12 - o1) >> 5;
    v[k * 8 + 3] = (t13 + o0) >> 5; v[k * 8 + 4] = (t13 - o0) >> 5;
  }
  if (D2) { idct_columns_dot2<NR, X2>(v, colr); return; }
#pragma unroll
  for (int c = 0; c < 8; c++)
    idct_1d<true, 12, NR, X2>(v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c], colr);
}

// True if coefficient rows 4..7 (the upper half of the vertical frequencies) are zero in every block the wave holds:
// the usual case for chroma and for smooth luma.  Wave-uniform, so the caller branches without divergence.
__device__ __forceinline__ bool rows_4_to_7_zero(const u32x4 (&rows)[8])
{
  unsigned o = 0;
#pragma unroll
  for (int r = 4; r < 8; r++) o |= rows[r].x | rows[r].y | rows[r].z | rows[r].w;
  return __builtin_amdgcn_ballot_w64(o != 0) == 0;
}

// ... and the horizontal frequencies 4..7 of the remaining rows (dwords z, w of rows 0..3)
__device__ __forceinline__ bool cols_4_to_7_zero(const u32x4 (&rows)[8])
{
  unsigned o = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) o |= rows[r].z | rows[r].w;
  return __builtin_amdgcn_ballot_w64(o != 0) == 0;
}

// dequant_idct<true> with the pruned paths where the data allow it.  PK16ROW: q is a row of the kernel's argument block (the
// packed deltas follow the 64) and the frame has 8-bit samples -- the first pass runs in 16 bits (dequant_idct16).
// D2 (with PK16ROW): the second pass on v_dot2 as well (idct_columns_dot2: the host's gate, dcoff = 0).
template <bool PK16ROW = false, bool X2 = false, bool D2 = false>
__device__ __forceinline__ void dequant_idct_sparse(const u32x4 (&rows)[8], const int *__restrict__ q, int (&v)[64], int dcoff = 0, const int colr = 2048)
{
  static_assert(!D2 || PK16ROW, "the 16-bit second pass belongs to the 8-bit FAST flavours");
  if (PK16ROW) {
    if (rows_4_to_7_zero(rows)) {
      if (cols_4_to_7_zero(rows)) dequant_idct<true, 4, 4, X2, D2>(rows, q, v, dcoff, colr); // (the pruned butterfly is as short as the products)
      else dequant_idct16<4, 8, X2, D2>(rows, q, v, dcoff, colr);
    } else dequant_idct16<8, 8, X2, D2>(rows, q, v, dcoff, colr);
    return;
  }
  if (rows_4_to_7_zero(rows)) {
    if (cols_4_to_7_zero(rows)) dequant_idct<true, 4, 4, X2>(rows, q, v, dcoff, colr);
    else dequant_idct<true, 4, 8, X2>(rows, q, v, dcoff, colr);
  } else dequant_idct<true, 8, 8, X2>(rows, q, v, dcoff, colr);
}

// ----------------------------------------------------------------------------------------------
// Tile order of the fused kernels.  Consecutive workgroup ids land on different XCDs (id % 8).  XCD x takes every eighth ROW of
// tiles, over the frames of the launch: neighbouring tiles -- which read each other's halo blocks -- share an L2, and the eight
// XCDs work on eight adjacent tile rows of the same frame at any time.  Rounds 1-3 gave every XCD one contiguous eighth of the
// launch (its own frame, with 8 frames per launch): same L2 sharing, but the traffic-only copy of the headline kernel
// (tools/microbench/stream_ceiling) showed HBM delivering 3-4 % less for eight far-apart write fronts than for one, and the
// kernel itself runs at that copy's speed: 0.644-0.648 -> 0.667-0.674 of 8 TB/s on one box (profiles/r04/headline_variants.txt).
// The 12-bit 4:2:0 kernel keeps the old order (13 % slower with the new one), and so does the single-component kernel (no
// difference beyond the noise; profiles/r04/layouts_tile_order.txt); runs of 8, 15 or 30 tiles instead of a tile row measured the same;
// MIJ_TILE_ORDER 1 gives it to all kernels for A-B builds.  ~0u: a padding workgroup (the launch has whole groups of 8 tile rows).
// ----------------------------------------------------------------------------------------------
#ifndef MIJ_TILE_ORDER
#define MIJ_TILE_ORDER 2
#endif
template <int ORDER = MIJ_TILE_ORDER> __device__ __forceinline__ unsigned tile_of_workgroup(unsigned b, unsigned tiles_x, unsigned tile_rows)
{
  if (ORDER == 2) {
    const unsigned x = b & 7, i = b >> 3; // i-th workgroup of XCD x
    const unsigned g = i / tiles_x, row = g * 8 + x;
    return row < tile_rows ? row * tiles_x + (i - g * tiles_x) : ~0u;
  }
  const unsigned total = tiles_x * tile_rows, q = total >> 3, r = total & 7, x = b & 7, i = b >> 3;
  return x * q + min(x, r) + i;
}
// The same for the kernels on Fused420Args, with the two divisions by launch constants as multiplications (Fused420Args::magic_*,
// filled by the launchers: floor(2^32 / d) + 1, exact while dividend * d < 2^32 -- 0 where that cannot be promised, and the
// kernel divides): the workgroup's position costs scalar instructions only.  frame < 0: a padding workgroup.
struct TilePos { int frame, tx, ty; };
__device__ __forceinline__ unsigned div_by(unsigned x, unsigned d, unsigned magic) { return magic ? __umulhi(x, magic) : x / d; }
template <int ORDER = MIJ_TILE_ORDER> __device__ __forceinline__ TilePos tile_position(unsigned b, const Fused420Args &a)
{
  const unsigned tiles_x = (unsigned)a.tiles_x, tiles_y = (unsigned)a.tiles_y;
  if (ORDER == 2) {
    const unsigned x = b & 7, i = b >> 3;                                    // i-th workgroup of XCD x
    const unsigned g = div_by(i, tiles_x, a.magic_tx), col = i - g * tiles_x; // its g-th tile row, column col
    const unsigned row = g * 8 + x, frame = div_by(row, tiles_y, a.magic_ty);
    if (row >= tiles_y * (unsigned)a.frames) return TilePos{-1, 0, 0};
    return TilePos{(int)frame, (int)col, (int)(row - frame * tiles_y)};
  }
  const unsigned logical = tile_of_workgroup<1>(b, tiles_x, tiles_y * (unsigned)a.frames);
  const unsigned row = div_by(logical, tiles_x, a.magic_tx), col = logical - row * tiles_x, frame = div_by(row, tiles_y, a.magic_ty);
  return TilePos{(int)frame, (int)col, (int)(row - frame * tiles_y)};
}
#ifndef XT_TILE_ORDER
#define XT_TILE_ORDER 2 // the JPEG XT kernels: 0.339 -> 0.332 ms per 8 x 4K frames with the tile-row order (profiles/r04/layouts_tile_order.txt)
#endif
template <int ORDER = MIJ_TILE_ORDER> static unsigned workgroups_for_tiles(unsigned tiles_x, unsigned tile_rows)
{
  return ORDER == 2 ? ((tile_rows + 7u) / 8u) * 8u * tiles_x : tiles_x * tile_rows;
}

// ----------------------------------------------------------------------------------------------
// Coalesced block fetch: the wave reads 64 blocks x 128 B with eight 1-KiB-per-instruction loads
// (lane i takes the i-th 16-byte chunk), then hands every lane the eight rows of ITS block through a
// 2 KB LDS staging buffer, 16 blocks at a time.  The chunk position inside a block is XOR-swizzled
// with the block index so that both the 8-lane-group writes and the 16-lane-group reads are free of
// bank conflicts (MI355X_MICROARCH.md, LDS: ds_write_b128 / ds_read_b128 lane groups).
// chunkptr(m) returns the address of the lane's m-th 16-byte chunk; blocks outside the plane are redirected to a
// clamped (valid) block whose result is simply never used, so the loads need no predication.
// ----------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void load_blocks(u32x4 (&raw)[8], F chunkptr)
{
#pragma unroll
  for (int m = 0; m < 8; m++) raw[m] = *chunkptr(m); // m-th load of this lane: chunk (lane & 7) of local block (lane >> 3) + 8 m
}
// (the two halves of fetch_blocks: a kernel that has other work between issuing the loads and needing the rows calls them apart)
__device__ __forceinline__ void transpose_blocks(u32x4 (&rows)[8], u32x4 *stage, int lane, const u32x4 (&raw)[8])
{
  const int k = lane & 7;
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int nb = (lane >> 3) + 8 * t;
      stage[nb * 8 + (k ^ (nb & 7))] = raw[2 * j + t];
    }
    __builtin_amdgcn_wave_barrier();
    if ((lane >> 4) == j) {
      const int nb = lane & 15;
#pragma unroll
      for (int r = 0; r < 8; r++) rows[r] = stage[nb * 8 + (r ^ (nb & 7))];
    }
    __builtin_amdgcn_wave_barrier();
  }
}
template <class F>
__device__ __forceinline__ void fetch_blocks(u32x4 (&rows)[8], u32x4 *stage, int lane, F chunkptr)
{
  u32x4 raw[8];
#pragma unroll
  for (int m = 0; m < 8; m++) raw[m] = *chunkptr(m); // m-th load of this lane: chunk (lane & 7) of local block (lane >> 3) + 8 m
  const int k = lane & 7;
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int nb = (lane >> 3) + 8 * t;
      stage[nb * 8 + (k ^ (nb & 7))] = raw[2 * j + t];
    }
    __builtin_amdgcn_wave_barrier();
    if ((lane >> 4) == j) {
      const int nb = lane & 15;
#pragma unroll
      for (int r = 0; r < 8; r++) rows[r] = stage[nb * 8 + (r ^ (nb & 7))];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// The same for 16 blocks (two loads per lane, lanes 0..15 receive their rows); chunkptr(m), m = 0, 1, as above.
template <class F>
__device__ __forceinline__ void fetch_blocks16(u32x4 (&rows)[8], u32x4 *stage, int lane, F chunkptr)
{
  const u32x4 raw0 = *chunkptr(0), raw1 = *chunkptr(1);
  const int k = lane & 7, nb0 = lane >> 3, nb1 = nb0 + 8;
  stage[nb0 * 8 + (k ^ (nb0 & 7))] = raw0;
  stage[nb1 * 8 + (k ^ (nb1 & 7))] = raw1;
  __builtin_amdgcn_wave_barrier();
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < 8; r++) rows[r] = stage[lane * 8 + (r ^ (lane & 7))];
  }
  __builtin_amdgcn_wave_barrier();
}

// One column (x = 0 or x = 7) of a block's samples: the first pass only has to produce that one output per row, which
// is the inner product of the row with the corresponding row of the transform's integer matrix -- written out of the
// butterfly of idct_1d<true, 9>: out0/out7 = (t10 +- o3) with every product expanded (exact: the ring Z / 2^32 is
// distributive) -- then the ordinary second pass on the eight results.  FAST arithmetic, no level shift.
__device__ __forceinline__ void dequant_idct_column(const u32x4 (&rows)[8], const int *__restrict__ q, bool last, int (&col)[8])
{
  constexpr int E2 = FIX9(0.541196100) + FIX9(0.765366865), E4 = 512, E6 = FIX9(0.541196100); // (s0 enters shifted by 9)
  constexpr int O1 = FIX9(1.501321110) - FIX9(0.899976223) - FIX9(0.390180644) + FIX9(1.175875602), O3 = FIX9(1.175875602),
                O5 = FIX9(1.175875602) - FIX9(0.390180644), O7 = FIX9(1.175875602) - FIX9(0.899976223);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const unsigned w[4] = {rows[r].x, rows[r].y, rows[r].z, rows[r].w};
    int s[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      s[2 * i] = mul16_lo(w[i], q[r * 8 + 2 * i]);
      s[2 * i + 1] = mul16_hi(w[i], q[r * 8 + 2 * i + 1]);
    }
    int even = (s[0] << 9) + (1 << 8), odd = __mul24(s[1], O1);
    even = mad24(s[2], E2, even);
    odd = mad24(s[3], O3, odd);
    even = mad24(s[4], E4, even);
    odd = mad24(s[5], O5, odd);
    even = mad24(s[6], E6, even);
    odd = mad24(s[7], O7, odd);
    const int acc = last ? even - odd : even + odd; // x = 7 : x = 0
    col[r] = acc >> 9;
  }
  idct_1d<true, 12>(col[0], col[1], col[2], col[3], col[4], col[5], col[6], col[7]);
}

// One LINE (y = 0 or y = 7) of a block's samples, computed by the EIGHT lanes that fetched the block's eight 16-byte chunks
// (chunk k = coefficient row k): every lane runs the first pass on its own row, weights the eight results with its entry of
// the corresponding row of the transform's integer matrix (see dequant_idct_column: even - odd for y = 7, even + odd for
// y = 0), and three DPP steps add the eight lanes' terms -- exact, the ring Z / 2^32 is associative.  qrow = the eight
// (pre-shifted) deltas of row k; all eight lanes return all eight samples.  FAST arithmetic, no level shift.
__device__ __forceinline__ void dequant_idct_line8(const u32x4 row, const int (&qrow)[8], int weight, int (&line)[8])
{
  const unsigned w[4] = {row.x, row.y, row.z, row.w};
  int s[8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s[2 * i] = mul16_lo(w[i], qrow[2 * i]);
    s[2 * i + 1] = mul16_hi(w[i], qrow[2 * i + 1]);
  }
  idct_1d<true, 9>(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
#pragma unroll
  for (int x = 0; x < 8; x++) {
    int p = __mul24(s[x], weight);
    p += __builtin_amdgcn_update_dpp(0, p, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    p += __builtin_amdgcn_update_dpp(0, p, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    p += __builtin_amdgcn_update_dpp(0, p, 0x141, 0xF, 0xF, false); // row_half_mirror: the other quad of the eight
    line[x] = (p + (1 << 11)) >> 12;
  }
}

// ----------------------------------------------------------------------------------------------
// colour transform of one pixel (colortrafo/ycbcrtrafo.cpp:842-850 + clamp :921-936)
// FIX_BITS = 13, matrix = TO_FIX of {1,0,1.402; 1,-0.3441362861,-0.7141362859; 1,1.772,0}
// ----------------------------------------------------------------------------------------------
#define L_CR_R 11485
#define L_CB_G 2819
#define L_CR_G 5850
#define L_CB_B 14516

__device__ __forceinline__ int clamp255(int v) { return min(max(v, 0), 255); }

template <bool FAST>
__device__ __forceinline__ void ycc_to_rgb(int y, int cb, int cr, int &r, int &g, int &b)
{
  if (FAST) {
    // dc shift (128 << 4 = 2048) and rounding folded into the constants
    const int KR = 65536 - 2048 * L_CR_R;
    const int KG = 65536 + 2048 * (L_CB_G + L_CR_G);
    const int KB = 65536 - 2048 * L_CB_B;
    const int y13 = y << 13;
    r = clamp255((__mul24(cr, L_CR_R) + (y13 + KR)) >> 17);
    g = clamp255((__mul24(cb, -L_CB_G) + __mul24(cr, -L_CR_G) + (y13 + KG)) >> 17);
    b = clamp255((__mul24(cb, L_CB_B) + (y13 + KB)) >> 17);
  } else {
    const long long yy = (long long)y * 8192 + 65536;
    const long long cbl = (long long)cb - 2048, crl = (long long)cr - 2048;
    long long rr = (yy + crl * L_CR_R) >> 17;
    long long gg = (yy - cbl * L_CB_G - crl * L_CR_G) >> 17;
    long long bb = (yy + cbl * L_CB_B) >> 17;
    r = (int)(rr < 0 ? 0 : (rr > 255 ? 255 : rr));
    g = (int)(gg < 0 ? 0 : (gg > 255 ? 255 : gg));
    b = (int)(bb < 0 ? 0 : (bb > 255 ? 255 : bb));
  }
}

// COLOR_TO_INT + clamp (identity transformation, tools/numerics.hpp:69)
template <bool FAST>
__device__ __forceinline__ int color_to_int(int x)
{
  if (FAST) return clamp255((x + 8) >> 4);
  long long v = ((long long)x + 8) >> 4;
  return (int)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// (a + 3 b + r) >> 2 in wrapping 32-bit arithmetic (the reference's LONG), upsampler.cpp filter taps
__device__ __forceinline__ int tap13(int a, int b, int r) { return (int)((unsigned)a + 3u * (unsigned)b + (unsigned)r) >> 2; }

// The colour stage of the 12-bit kernels, one pixel: (y * 8192 + cb' * Lb + cr' * Lr + 65536) >> 17 per channel (ycbcrtrafo.cpp:842-856,
// :921-936; 64-bit sums in the reference) with y = y' + 32768 (the level shift 2048, times 16) and cb' = cb - 32768: the transforms leave
// the level shift out, so y', cb', cr' are what arrives here -- samples times 16.
//   NARROW: the whole sum in 32 bits -- one multiply-add per product and one shift per channel.  Exact where
//           (|y'| + 32776) * 8192 + 14516 |c| < 2^31; the host admits it by the range check (narrow12_colour, capi.cpp).  Partial
//           sums of the green channel may wrap, the complete one does not.
//   else:   c L = q 2^13 + r, 0 <= r < 2^13, gives ((y' + 32776 + q) 2^13 + r) >> 17 = (y' + 32776 + q) >> 4: products alone must fit
//           32 bits (fused420_kernel<.., 12> has the derivation and the bounds).
template <bool NARROW>
__device__ __forceinline__ void colour12(int y, int cb, int cr, int &r, int &g, int &b)
{
  static_assert(L_CB_B % 4 == 0, "the blue product is taken at a quarter of the constant");
  if (NARROW) {
    const int yk = shlw(y, 13) + ((32768 + 8) << 13);
    r = mad24(cr, L_CR_R, yk) >> 17;
    g = mad24(cr, -L_CR_G, mad24(cb, -L_CB_G, yk)) >> 17;
    b = mad24(cb, L_CB_B, yk) >> 17;
  } else {
    const int yk = y + (32768 + 8);
    r = (yk + (__mul24(cr, L_CR_R) >> 13)) >> 4;
    g = (yk + (mad24(cr, -L_CR_G, __mul24(cb, -L_CB_G)) >> 13)) >> 4;
    b = (yk + (__mul24(cb, L_CB_B / 4) >> 11)) >> 4;
  }
}

// ==============================================================================================
// fused 4:2:0 kernel
// ==============================================================================================
// Work decomposition: one workgroup (256 threads = 4 waves) reconstructs a 128x128-pixel tile =
// 8x8 MCUs = 16x16 luma blocks + (8+2)x(8+2) chroma blocks per chroma component (one block of halo
// all around, recomputed from the coefficients instead of exchanged between workgroups).
//   phase A  waves 0,1 transform the 100 Cb blocks, waves 2,3 the 100 Cr blocks, and store the
//            66x66 samples the tile needs into LDS (int32, pitch 72 -> 16-byte aligned block rows)
//   fix-up   (tiles on an image edge only) replicate the last valid chroma column/row outwards,
//            upsamplerbase.cpp:322-323 and the top/bottom line duplication of upsampler.cpp:100-112
//   phase B  every lane transforms one luma block, keeps it in registers, upsamples its 8x8 chroma
//            neighbourhood from LDS (vertical then horizontal filter, including the in-place aliasing
//            of output column 1), converts to RGB and stores 8 lines x 24 bytes.
constexpr int F420_TILE_BLOCKS = 16;      // luma blocks per tile side
constexpr int F420_CGRID = 10;            // chroma blocks per tile side incl. halo
constexpr int F420_CROWS = 66;            // chroma lines kept in LDS (64 + 2 halo)
constexpr int F420_CPITCH = 72;           // dwords per LDS chroma line; column pc <-> chroma x_rel = pc - 4
constexpr int F420_THREADS = 256;

// The deltas of component c of a frame: from the kernel arguments (scalar loads out of the kernarg segment), or -- QDEV, batches
// whose frames bring their own tables -- from the per-frame tables in device memory (scalar loads all the same: the address is
// wave-uniform).  One of the two survives constant folding, so the address space is known where the loads are emitted.
template <bool QDEV, class Args>
__device__ __forceinline__ const int *frame_deltas(const Args &a, int frame, int c)
{
  if (QDEV) return a.qdev + ((int64_t)frame * 4 + c) * 64;
  return a.q[c];
}

// phase A of the 4:2:0 kernels with 32-bit chroma samples: the (8+2) x (8+2) chroma blocks of tile (tx, ty) -> LDS
// F420P_PREFETCH: the luma blocks of phase B are requested in front of phase A's transform (32 more registers across it; with
// F420P_MINW 4 the allocation still leaves four workgroups per CU); 0: at the start of phase B, for A-B builds.
// Measured on one box (profiles/r04/headline_variants.txt): reference-encoded frames 0.617 -> 0.635, dense blocks 0.583 -> 0.595.
#ifndef F420P_PREFETCH
#define F420P_PREFETCH 1
#endif
#ifndef F420_12_PREFETCH
#define F420_12_PREFETCH 0 // the same for the 12-bit flavour of fused420_kernel (A-B builds)
#endif
// F420P_STAGED: whole waves of whole blocks send a line's pixels through the wave's (idle) fetch staging buffer -- sixteen 24-byte
// pieces per block row in, 96 chunks of 16 contiguous bytes out -- so that every store instruction writes whole aligned runs of the
// four 384-byte line segments instead of 16 and then 8 bytes of every lane's 24: tools/microbench/stream_ceiling --staged puts the
// kernel's access pattern at 0.74-0.755 of 8 TB/s with such stores, 0.71-0.72 with the pieces (profiles/r06/staged_stores.txt).
#ifndef F420P_STAGED
#define F420P_STAGED 1 // (0: the 24-byte pieces everywhere, for A-B builds)
#endif
#ifndef F420P_TEMPORAL
#define F420P_TEMPORAL 0 // A-B builds: 1 = the pixel stores of aligned frames without the nt hint as well
#endif
#ifndef F420P_MINW
#define F420P_MINW 4 // workgroups per CU the register allocation must leave room for (the per-frame-table build keeps 3: it spills at 4)
#endif
struct NoAfterFetch { __device__ __forceinline__ void operator()() const {} };
// after_fetch: called between the chroma blocks' arrival and their transform -- where a caller requests the blocks it needs next
// (fused420_kernel: the luma blocks of phase B; their latency then hides behind this transform)
template <bool FAST, bool QDEV = false, bool PK = false, class AfterFetch = NoAfterFetch>
__device__ __forceinline__ void f420_chroma_to_lds(const Fused420Args &a, const int16_t *__restrict__ coef, int (*cplane)[F420_CROWS * F420_CPITCH],
                                                   u32x4 *stage, int lane, int wave, int tx, int ty, int frame = 0, AfterFetch after_fetch = AfterFetch())
{
  const int comp = wave >> 1; // 0 = Cb, 1 = Cr (wave-uniform)
  const int16_t *__restrict__ plane = coef + (comp ? a.off_cr : a.off_cb);
  const int gx0 = tx * 8 - 1, gy0 = ty * 8 - 1;
  const int base = (wave & 1) * 64;
  // tiles whose whole 10 x 10 block window lies inside the plane need no clamping (wave-uniform)
  const bool inside = gx0 >= 0 && gy0 >= 0 && gx0 + F420_CGRID <= a.bw_c && gy0 + F420_CGRID <= a.bh_c;
  u32x4 rows[8];
  // local block n = (lane >> 3) + 8 m is grid index base + n (clamped to the grid: waves 1 and 3 only hold 36 blocks)
  const int idx0 = base + (lane >> 3);
  const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
  if (inside) {
    const unsigned off0 = (unsigned)((gy0 * a.bw_c + gx0) * 128), rowb = (unsigned)a.bw_c * 128u;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const unsigned i = (unsigned)min(idx0 + 8 * m, F420_CGRID * F420_CGRID - 1);
      const unsigned y = (i * 205u) >> 11, x = i - y * F420_CGRID; // i / 10, i % 10 for i < 1029
      return reinterpret_cast<const u32x4 *>(pbase + (off0 + y * rowb + x * 128u));
    });
  } else {
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int i = min(idx0 + 8 * m, F420_CGRID * F420_CGRID - 1);
      const int y = (i * 205) >> 11, x = i - y * F420_CGRID;
      const int gx = min(max(gx0 + x, 0), a.bw_c - 1), gy = min(max(gy0 + y, 0), a.bh_c - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((gy * a.bw_c + gx) * 128));
    });
  }
  after_fetch();
  if constexpr (!std::is_same<AfterFetch, NoAfterFetch>::value) __builtin_amdgcn_sched_barrier(0); // (what it issued stays in front of the transform)
  const int idx = base + lane;
  const int cby = idx / F420_CGRID, cbx = idx - cby * F420_CGRID;
  const int gx = gx0 + cbx, gy = gy0 + cby;
  if (idx < F420_CGRID * F420_CGRID && gx >= 0 && gy >= 0 && gx < a.bw_c && gy < a.bh_c) {
    int v[64];
    const int *q = frame_deltas<QDEV>(a, frame, 1 + comp);
    if (FAST) dequant_idct_sparse<PK>(rows, q, v);
    else dequant_idct<false>(rows, q, v, 128 << 7);
    int *cp = cplane[comp];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int pr = 8 * cby + r - 7;
      if (pr >= 0 && pr < F420_CROWS) {
        if (cbx == 0) {
          cp[pr * F420_CPITCH + 3] = v[r * 8 + 7];
        } else if (cbx == F420_CGRID - 1) {
          cp[pr * F420_CPITCH + 68] = v[r * 8 + 0];
        } else {
          i32x4 *dst = reinterpret_cast<i32x4 *>(cp + pr * F420_CPITCH + 8 * cbx - 4);
          dst[0] = i32x4{v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3]};
          dst[1] = i32x4{v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]};
        }
      }
    }
  }
}

// image-edge tiles: replicate the last valid chroma column / line outwards (upsamplerbase.cpp:322-323, upsampler.cpp:100-112)
__device__ __forceinline__ void f420_chroma_edges(const Fused420Args &a, int (*cplane)[F420_CROWS * F420_CPITCH], int tid, int tx, int ty)
{
  const int last_col = a.cw - 1 - tx * 64; // last valid chroma column, tile-relative
  const int last_row = a.ch - 1 - ty * 64;
  const bool edge = (tx == 0) | (ty == 0) | (last_col < 64) | (last_row < 64);
  if (edge) {
    if (tid < 2 * F420_CROWS) { // one thread per stored line: replicate columns
      int *p = cplane[tid / F420_CROWS] + (tid % F420_CROWS) * F420_CPITCH;
      if (tx == 0) p[3] = p[4];
      if (last_col < 64) {
        const int v = p[last_col + 4];
        for (int pc = last_col + 5; pc <= 68; pc++) p[pc] = v;
      }
    }
    __syncthreads();
    if (tid < 2 * F420_CROWS) { // one thread per stored column: replicate lines
      int *p = cplane[tid / F420_CROWS] + 3 + (tid % F420_CROWS);
      if (ty == 0) p[0] = p[F420_CPITCH];
      if (last_row < 64) {
        const int v = p[(last_row + 1) * F420_CPITCH];
        for (int pr = last_row + 2; pr < F420_CROWS; pr++) p[pr * F420_CPITCH] = v;
      }
    }
    __syncthreads();
  }
}

// P = 12 (FAST only): 12-bit frames (SOF1, P = 12) -- the same transforms and filters on samples sixteen times as large, the
// colour stage rearranged so that it stays inside 32 bits (see there), 16-bit samples out (clamp 4095).
// 24-byte line pieces as three 8-byte stores (the kernels that do not use store24_nt's 16 + 8): A-B macros for the hint
#ifndef F444_TEMPORAL
#define F444_TEMPORAL 0
#endif
#ifndef F420U_TEMPORAL
#define F420U_TEMPORAL 0
#endif
template <bool TEMPORAL>
__device__ __forceinline__ void store24_3x8(uint8_t *dst, const unsigned (&w)[6])
{
  u32x2_any *d2 = reinterpret_cast<u32x2_any *>(dst);
  if (TEMPORAL) {
    d2[0] = u32x2{w[0], w[1]}; d2[1] = u32x2{w[2], w[3]}; d2[2] = u32x2{w[4], w[5]};
  } else {
    __builtin_nontemporal_store(u32x2{w[0], w[1]}, d2);
    __builtin_nontemporal_store(u32x2{w[2], w[3]}, d2 + 1);
    __builtin_nontemporal_store(u32x2{w[4], w[5]}, d2 + 2);
  }
}

// The 48-byte line pieces of the 12-bit kernels (8 pixels x 3 x 16 bit) leave as three 16-byte stores per lane, i.e. every store
// instruction writes 16 bytes out of every 48: with the non-temporal hint the write counter showed 1.17 x the bytes written
// (profiles/r05/summary_12bit_r05.txt: partial lines leave the cache before their neighbours arrive), without it the lines are
// completed in L2 -- 12-bit 4:4:4 369 -> 429 Gpixel/s, 4:2:2 425 -> 477 (profiles/r05/f12_stores.txt).  0 = the hint, for A-B builds.
#ifndef F12_TEMPORAL
#define F12_TEMPORAL 1
#endif
#ifndef F420_12_TEMPORAL
#define F420_12_TEMPORAL 1 // the 12-bit 4:2:0 kernel: 418-438 -> 491 Gpixel/s
#endif
#ifndef FXT_TEMPORAL
#define FXT_TEMPORAL 0 // fusedxt420_kernel / fusedxtw420_kernel (config 5: 8 pixels x 3 half-float codes per line piece)
#endif
template <bool TEMPORAL = F12_TEMPORAL != 0>
__device__ __forceinline__ void store48(uint8_t *dst, const unsigned (&w)[12])
{
  u32x4_any *d4 = reinterpret_cast<u32x4_any *>(dst);
  if (TEMPORAL) {
    d4[0] = u32x4{w[0], w[1], w[2], w[3]}; d4[1] = u32x4{w[4], w[5], w[6], w[7]}; d4[2] = u32x4{w[8], w[9], w[10], w[11]};
  } else {
    __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, d4);
    __builtin_nontemporal_store(u32x4{w[4], w[5], w[6], w[7]}, d4 + 1);
    __builtin_nontemporal_store(u32x4{w[8], w[9], w[10], w[11]}, d4 + 2);
  }
}

template <bool FAST, int MINW, bool QDEV, int P = 8, bool N12 = false>
__global__ __launch_bounds__(F420_THREADS, MINW) void fused420_kernel(const Fused420Args a)
{
  static_assert(P == 8 || (P == 12 && FAST), "12-bit frames: FAST flavour only (the host checks the ranges)");
  __shared__ __attribute__((aligned(16))) int cplane[2][F420_CROWS * F420_CPITCH];
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position<P == 8 ? MIJ_TILE_ORDER : 1>(blockIdx.x, a); // (tile order: see there)
  if (tp.frame < 0) return; // the launch is padded to whole groups of eight tile rows
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;

  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  // the luma blocks of phase B are requested in front of phase A's transform (see fused420p_kernel; two workgroups per CU leave
  // the 32 registers)
  u32x4 yraw[8];
  auto luma_loads = [&]() {
    const int16_t *__restrict__ plane = coef + a.off_y;
    const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
    // local block n = (lane >> 3) + 8 m sits at column n & 15 = (lane >> 3) + 8 (m & 1), row n >> 4 = m >> 1 of the wave's 16 x 4 blocks
    const int x0 = gbx0 + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    load_blocks(yraw, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  };
  // (8-bit frames only: the 12-bit flavour measured 30 % slower with it, profiles/r04/headline_variants.txt visit x)
  constexpr bool PREFETCH = F420P_PREFETCH && (P == 8 || F420_12_PREFETCH);
  if (PREFETCH) f420_chroma_to_lds<FAST, QDEV, !QDEV && P == 8>(a, coef, cplane, stage, lane, wave, tx, ty, frame, luma_loads);
  else f420_chroma_to_lds<FAST, QDEV, !QDEV && P == 8>(a, coef, cplane, stage, lane, wave, tx, ty, frame);
  __syncthreads();
  f420_chroma_edges(a, cplane, tid, tx, ty);

  // ------------------------------------------------------------------ phase B: luma + colour
  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  u32x4 rows[8];
  if (!PREFETCH) luma_loads();
  transpose_blocks(rows, stage, lane, yraw);
  const int X0 = gbx * 8, Y0 = gby * 8;
  if (X0 >= a.width || Y0 >= a.height) return; // no barrier below this point
  int yv[64];
  if (FAST) dequant_idct_sparse<!QDEV && P == 8, P == 8>(rows, frame_deltas<QDEV>(a, frame, 0), yv, 0, P == 8 ? LUMA_FOLD_R2 : 2048); // (8 bit: doubled sums, luma13)
  else dequant_idct<false>(rows, frame_deltas<QDEV>(a, frame, 0), yv, 128 << 7);

  // uniform frame base + 32-bit lane offsets (a frame of pixels is far below 4 GB)
  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * (P == 12 ? 6u : 3u);
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
  const bool fast_store = npx == 8;

  // chroma window of this block: lines pr = 4 by + m (+0 top, +1 cur, +2 bot), columns pc = 4 bx + 3 + j
  const int *cb_base = cplane[0] + (4 * by) * F420_CPITCH + 4 * bx;
  const int *cr_base = cplane[1] + (4 * by) * F420_CPITCH + 4 * bx;

  auto load6 = [](const int *p, int (&d)[6]) {
    // p is 16-byte aligned; wanted: p[3..8]
    const i32x4 mid = *reinterpret_cast<const i32x4 *>(p + 4);
    d[0] = p[3]; d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w; d[5] = p[8];
  };

  int cbT[6], cbC[6], cbB[6], crT[6], crC[6], crB[6];
  load6(cb_base, cbT); load6(cb_base + F420_CPITCH, cbC);
  load6(cr_base, crT); load6(cr_base + F420_CPITCH, crC);

#pragma unroll
  for (int m = 0; m < 4; m++) {
    load6(cb_base + (m + 2) * F420_CPITCH, cbB);
    load6(cr_base + (m + 2) * F420_CPITCH, crB);
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int l = 2 * m + half;
      // vertical filter (upsampler.cpp:149-165): even output lines look up, odd ones down; the
      // rounding constant alternates with the buffer column parity
      int vb[6], vr[6];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const int rnd = ((j & 1) ^ half) ? 1 : 2;
        vb[j] = tap13(half ? cbB[j] : cbT[j], cbC[j], rnd);
        vr[j] = tap13(half ? crB[j] : crT[j], crC[j], rnd);
      }
      // horizontal filter in place (upsampler.cpp:291-303); src[k] = v[k + 1]
      int ub[8], ur[8];
      auto hfilt = [](const int (&v)[6], int (&o)[8]) {
        o[7] = tap13(v[5], v[4], 1);
        o[6] = tap13(v[3], v[4], 2);
        o[5] = tap13(v[4], v[3], 1);
        o[4] = tap13(v[2], v[3], 2);
        o[3] = tap13(v[3], v[2], 1);
        o[2] = tap13(v[1], v[2], 2);
        o[1] = tap13(o[2], v[1], 1); // src[1] has already been overwritten by out[2]
        o[0] = tap13(v[0], v[1], 2);
      };
      hfilt(vb, ub);
      hfilt(vr, ur);
      if (l < nln) {
        uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
        if (FAST && P == 12) {
          // (y * 8192 + (cb - 32768) * Lb + (cr - 32768) * Lr + 65536) >> 17, clamped to [0, 4095] (ycbcrtrafo.cpp:842-856,
          // :921-936; the reference accumulates in 64 bits).  With y = y' + 32768, cb - 32768 = cb' (no level shift in the
          // transforms) and c L = q 2^13 + r, 0 <= r < 2^13:  ((y' + 32776 + q) 2^13 + r) >> 17 = (y' + 32776 + q) >> 4
          // exactly.  The products must fit 32 bits: |c| <= 4.02 * 45056 + 2 < 181 200 by the host's range check (an IDCT
          // output times 16 is at most 4 sum |c_k| q_k; the 9-bit constants and the two roundings add < 0.5 %), times 11485
          // < 2^31; Lb = 14516 = 4 * 3629 is applied as (c * 3629) >> 11, which is the same number.
          int rr[8], gg[8], bb[8];
#pragma unroll
          for (int x = 0; x < 8; x++) colour12<N12>(yv[l * 8 + x], ub[x], ur[x], rr[x], gg[x], bb[x]);
          auto c12 = [](int v) { return (unsigned)min(max(v, 0), 4095); };
          if (fast_store) {
            unsigned w[12]; // 48 bytes r0 g0 b0 r1 ... b7, 16-bit samples
#pragma unroll
            for (int x = 0; x < 8; x += 2) {
              w[3 * (x / 2) + 0] = c12(rr[x]) | (c12(gg[x]) << 16);
              w[3 * (x / 2) + 1] = c12(bb[x]) | (c12(rr[x + 1]) << 16);
              w[3 * (x / 2) + 2] = c12(gg[x + 1]) | (c12(bb[x + 1]) << 16);
            }
            store48<F420_12_TEMPORAL != 0>(dst, w);
          } else {
            uint16_t *d16 = reinterpret_cast<uint16_t *>(dst);
#pragma unroll
            for (int x = 0; x < 8; x++)
              if (x < npx) {
                d16[3 * x] = (uint16_t)c12(rr[x]); d16[3 * x + 1] = (uint16_t)c12(gg[x]); d16[3 * x + 2] = (uint16_t)c12(bb[x]);
              }
          }
        } else if (FAST) {
          // y, cb, cr arrive WITHOUT the level shift (DCOFF = false): with y = y' + 2048 and cb - 2048 = cb' the
          // reference's (y * 8192 + (cb - 2048) * Lb + (cr - 2048) * Lr + 65536) >> 17 becomes
          // (y' * 8192 + K + cb' * Lb + cr' * Lr) >> 17 with one constant K for all three channels
          int rr[8], gg[8], bb[8];
#pragma unroll
          for (int x = 0; x < 8; x++) {
            const int yk = luma13(yv[l * 8 + x]); // (level shift and rounding are inside: LUMA_FOLD_R2)
            rr[x] = mad24(ur[x], L_CR_R, yk); // still scaled by 2^17
            gg[x] = mad24(ur[x], -L_CR_G, mad24(ub[x], -L_CB_G, yk));
            bb[x] = mad24(ub[x], L_CB_B, yk);
          }
          if (fast_store) {
            // 24 bytes r0 g0 b0 r1 ... b7: clamp + pack two samples per instruction pair
            unsigned w[6];
            rgb_shift17_sat_pack(rr, gg, bb, w);
            store24_3x8<F420U_TEMPORAL != 0>(dst, w);
          } else {
#pragma unroll
            for (int x = 0; x < 8; x++)
              if (x < npx) {
                dst[3 * x] = (uint8_t)clamp255(rr[x] >> 17); dst[3 * x + 1] = (uint8_t)clamp255(gg[x] >> 17); dst[3 * x + 2] = (uint8_t)clamp255(bb[x] >> 17);
              }
          }
        } else {
          unsigned px[24];
#pragma unroll
          for (int x = 0; x < 8; x++) {
            int r, g, b;
            ycc_to_rgb<false>(yv[l * 8 + x], ub[x], ur[x], r, g, b);
            px[3 * x] = r; px[3 * x + 1] = g; px[3 * x + 2] = b;
          }
          if (fast_store) {
            unsigned w[6];
#pragma unroll
            for (int i = 0; i < 6; i++) w[i] = px[4 * i] | (px[4 * i + 1] << 8) | (px[4 * i + 2] << 16) | (px[4 * i + 3] << 24);
            store24_3x8<F420U_TEMPORAL != 0>(dst, w);
          } else {
#pragma unroll
            for (int x = 0; x < 8; x++)
              if (x < npx) {
                dst[3 * x] = (uint8_t)px[3 * x]; dst[3 * x + 1] = (uint8_t)px[3 * x + 1]; dst[3 * x + 2] = (uint8_t)px[3 * x + 2];
              }
          }
        }
      }
    }
    // slide the three-line window
#pragma unroll
    for (int j = 0; j < 6; j++) { cbT[j] = cbC[j]; cbC[j] = cbB[j]; crT[j] = crC[j]; crC[j] = crB[j]; }
  }
}

// ==============================================================================================
// fused 4:2:0 kernel, packed chroma flavour
// ==============================================================================================
// Same decomposition as fused420_kernel<true>, for frames whose chroma samples are small enough for 16-bit filter
// arithmetic: 4 * range_max < 8190 bounds every chroma sample (times 16), so the filter sums a + 3 b + r of
// upsampler.cpp stay inside int16 and (Cb, Cr) of one position travel as ONE register (Cb low, Cr high half):
//   * phase A is unchanged (waves 0, 1: Cb, waves 2, 3: Cr -- every wave, i.e. every SIMD, carries the same load)
//     except that a sample is stored as the 16-bit half of its position's dword: one LDS plane instead of two;
//   * both filter passes run as v_pk_mad_i16 / v_pk_add_i16 / v_pk_ashrrev_i16 on the pairs: half the instructions
//     and half the LDS reads of the 32-bit flavour; the colour multiply-adds read the halves in place
//     (v_mad_i32_i16 op_sel).
// Everything else (tile shape, halo, edge replication, in-place aliasing of output column 1, store path) is
// identical, and so are the results: wherever nothing overflows, int16 and int32 arithmetic agree.
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned tap13_pk(unsigned a, unsigned b, short r)
{
  const s16x2 va = __builtin_bit_cast(s16x2, a), vb = __builtin_bit_cast(s16x2, b);
  const s16x2 t = (va + vb * (short)3 + r) >> (short)2;
  return __builtin_bit_cast(unsigned, t);
}
// the same in two steps for a centre sample that two output lines share: 3 b + r once per rounding (v_pk_mad_u16), then
// (a + that) >> 2 per line -- 3 instructions per tap where a shared 3 b costs 3.5
__device__ __forceinline__ unsigned centre3_pk(unsigned b, short r)
{
  unsigned d;
  asm("v_pk_mad_u16 %0, %1, 3, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(b), "n"(r));
  return d;
}
__device__ __forceinline__ unsigned tap_sum_pk(unsigned a, unsigned w)
{
  const s16x2 t = (__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, w)) >> (short)2;
  return __builtin_bit_cast(unsigned, t);
}

// D2: the second pass of every transform on v_dot2 (idct_columns_dot2; admitted by the host where sum |c| q <= 1476 in all three
// components: use_dot2_pass in capi.cpp)
template <int MINW, bool QDEV, bool D2 = false>
__global__ __launch_bounds__(F420_THREADS, MINW) void fused420p_kernel(const Fused420Args a)
{
  __shared__ __attribute__((aligned(16))) unsigned cpair[F420_CROWS * F420_CPITCH];
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position(blockIdx.x, a); // (tile order: see there)
  if (tp.frame < 0) return; // the launch is padded to whole groups of eight tile rows
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;
#if F420P_PREFETCH
  // the luma blocks of phase B are requested early: their latency hides behind phase A's transform
  u32x4 yraw[8];
  auto luma_loads = [&]() {
    const int16_t *__restrict__ plane = coef + a.off_y;
    const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
    const int x0 = gbx0 + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    load_blocks(yraw, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  };
#endif

  // ------------------------------------------------------------------ phase A: chroma -> LDS halves
  {
    const int comp = wave >> 1; // 0 = Cb (low halves), 1 = Cr (high halves); wave-uniform
    const int16_t *__restrict__ plane = coef + (comp ? a.off_cr : a.off_cb);
    const int gx0 = tx * 8 - 1, gy0 = ty * 8 - 1;
    const int base = (wave & 1) * 64;
    const bool inside = gx0 >= 0 && gy0 >= 0 && gx0 + F420_CGRID <= a.bw_c && gy0 + F420_CGRID <= a.bh_c;
    u32x4 rows[8];
    const int idx0 = base + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    if (inside) {
      const unsigned off0 = (unsigned)((gy0 * a.bw_c + gx0) * 128), rowb = (unsigned)a.bw_c * 128u;
      fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
        const unsigned i = (unsigned)min(idx0 + 8 * m, F420_CGRID * F420_CGRID - 1);
        const unsigned y = (i * 205u) >> 11, x = i - y * F420_CGRID; // i / 10, i % 10 for i < 1029
        return reinterpret_cast<const u32x4 *>(pbase + (off0 + y * rowb + x * 128u));
      });
    } else {
      fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
        const int i = min(idx0 + 8 * m, F420_CGRID * F420_CGRID - 1);
        const int y = (i * 205) >> 11, x = i - y * F420_CGRID;
        const int gx = min(max(gx0 + x, 0), a.bw_c - 1), gy = min(max(gy0 + y, 0), a.bh_c - 1);
        return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((gy * a.bw_c + gx) * 128));
      });
    }
    const int idx = base + lane;
    const int cby = idx / F420_CGRID, cbx = idx - cby * F420_CGRID;
    const int gx = gx0 + cbx, gy = gy0 + cby;
#if F420P_PREFETCH
    luma_loads();
#endif
    if (idx < F420_CGRID * F420_CGRID && gx >= 0 && gy >= 0 && gx < a.bw_c && gy < a.bh_c) {
      int v[64];
      dequant_idct_sparse<!QDEV, false, D2>(rows, frame_deltas<QDEV>(a, frame, 1 + comp), v);
      short *cp = reinterpret_cast<short *>(cpair) + comp; // this component's half of every dword
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int pr = 8 * cby + r - 7;
        const int line = pr * F420_CPITCH;
        if (pr >= 0 && pr < F420_CROWS) {
          if (cbx == 0) {
            cp[2 * (line + 3)] = (short)v[r * 8 + 7];
          } else if (cbx == F420_CGRID - 1) {
            cp[2 * (line + 68)] = (short)v[r * 8 + 0];
          } else {
            short *dst = cp + 2 * (line + 8 * cbx - 4);
#pragma unroll
            for (int x = 0; x < 8; x++) dst[2 * x] = (short)v[r * 8 + x];
          }
        }
      }
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ edge fix-up (uniform branch)
  {
    const int last_col = a.cw - 1 - tx * 64; // last valid chroma column, tile-relative
    const int last_row = a.ch - 1 - ty * 64;
    const bool edge = (tx == 0) | (ty == 0) | (last_col < 64) | (last_row < 64);
    if (edge) {
      if (tid < F420_CROWS) { // one thread per stored line: replicate columns
        unsigned *p = cpair + tid * F420_CPITCH;
        if (tx == 0) p[3] = p[4];
        if (last_col < 64) {
          const unsigned v = p[last_col + 4];
          for (int pc = last_col + 5; pc <= 68; pc++) p[pc] = v;
        }
      }
      __syncthreads();
      if (tid < F420_CROWS) { // one thread per stored column: replicate lines
        auto at = [&](int pr) -> unsigned & { return cpair[pr * F420_CPITCH + 3 + tid]; };
        if (ty == 0) at(0) = at(1);
        if (last_row < 64) {
          const unsigned v = at(last_row + 1);
          for (int pr = last_row + 2; pr < F420_CROWS; pr++) at(pr) = v;
        }
      }
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ phase B: luma, upsampling, colour
  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  u32x4 rows[8];
#if F420P_PREFETCH
  transpose_blocks(rows, stage, lane, yraw);
#else
  {
    const int16_t *__restrict__ plane = coef + a.off_y;
    const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
    const int x0 = gbx0 + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  }
#endif
  const int X0 = gbx * 8, Y0 = gby * 8;
  if (X0 >= a.width || Y0 >= a.height) return; // no barrier below this point
  int yv[64];
  dequant_idct_sparse<!QDEV, true, D2>(rows, frame_deltas<QDEV>(a, frame, 0), yv, 0, LUMA_FOLD_R2);

  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * 3u;
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
  const bool fast_store = npx == 8;
  // chroma window of this block: lines pr = 4 by + m (+0 top, +1 cur, +2 bot), columns pc = 4 bx + 3 + j
  const unsigned *c_base = cpair + (4 * by) * F420_CPITCH + 4 * bx;
  auto load6 = [](const unsigned *p, unsigned (&d)[6]) { // p is 16-byte aligned; wanted: p[3..8]
    const u32x4 mid = *reinterpret_cast<const u32x4 *>(p + 4);
    d[0] = p[3]; d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w; d[5] = p[8];
  };

  // Blocks that lie wholly inside the picture -- all of them, for every wave but those on the right and bottom edges -- take a
  // copy of the loop without the per-line exec masks (wave-uniform choice: one ballot)
  // (a third copy, FULL with every lane active, for frames whose lines do not all start on a dword: store24_nt_shifted)
  const unsigned base_lo = (unsigned)(uintptr_t)out_frame;
  auto lines = [&](auto full_tag, auto shift_tag, auto staged_tag) {
    constexpr bool FULL = decltype(full_tag)::value, SHIFTED = decltype(shift_tag)::value, STAGED = decltype(staged_tag)::value;
    // STAGED: chunk c of the wave's 96 per line is bytes 16 (c % 24) .. of block row c / 24's segment; lanes 0..31 own a second one
    unsigned so0 = 0, so1 = 0;
    if (STAGED) {
      const unsigned wave_off = (unsigned)((ty * F420_TILE_BLOCKS + wave * 4) * 8) * (unsigned)a.row_stride + (unsigned)(tx * F420_TILE_BLOCKS * 8) * 3u;
      const unsigned c1 = 64u + (unsigned)lane;
      so0 = wave_off + ((unsigned)lane / 24u) * 8u * (unsigned)a.row_stride + ((unsigned)lane % 24u) * 16u;
      so1 = wave_off + (c1 / 24u) * 8u * (unsigned)a.row_stride + (c1 % 24u) * 16u;
    }
    unsigned cT[6], cC[6], cB[6];
    load6(c_base, cT);
    load6(c_base + F420_CPITCH, cC);
#pragma unroll
    for (int m = 0; m < 4; m++) {
      load6(c_base + (m + 2) * F420_CPITCH, cB);
      // 3 * centre + rounding, both roundings, for the two lines that share the centre line
      unsigned w1[6], w2[6];
#pragma unroll
      for (int j = 0; j < 6; j++) { w1[j] = centre3_pk(cC[j], 1); w2[j] = centre3_pk(cC[j], 2); }
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int l = 2 * m + half;
        // vertical filter (upsampler.cpp:149-165), both components at once
        unsigned v[6];
#pragma unroll
        for (int j = 0; j < 6; j++) v[j] = tap_sum_pk(half ? cB[j] : cT[j], ((j & 1) ^ half) ? w1[j] : w2[j]);
        // horizontal filter in place (upsampler.cpp:291-303); src[k] = v[k + 1]
        unsigned u[8];
        u[7] = tap13_pk(v[5], v[4], 1);
        u[6] = tap13_pk(v[3], v[4], 2);
        u[5] = tap13_pk(v[4], v[3], 1);
        u[4] = tap13_pk(v[2], v[3], 2);
        u[3] = tap13_pk(v[3], v[2], 1);
        u[2] = tap13_pk(v[1], v[2], 2);
        u[1] = tap13_pk(u[2], v[1], 1); // src[1] has already been overwritten by out[2]
        u[0] = tap13_pk(v[0], v[1], 2);
        if (FULL || l < nln) {
          int rr[8], gg[8], bb[8];
#pragma unroll
          for (int x = 0; x < 8; x++) {
            const int yk = luma13(yv[l * 8 + x]); // (level shift and rounding are inside: LUMA_FOLD_R2)
            rr[x] = mad16_hi(u[x], L_CR_R, yk); // still scaled by 2^17
            bb[x] = mad16_lo(u[x], L_CB_B, yk);
            gg[x] = dot2_16(u[x], -L_CB_G, -L_CR_G, yk);
          }
          if (FULL || fast_store) {
            unsigned w[6];
            rgb_shift17_sat_pack(rr, gg, bb, w);
            const unsigned off = out_off + (unsigned)l * (unsigned)a.row_stride;
            const int m = SHIFTED ? (int)__builtin_amdgcn_readfirstlane((base_lo + (unsigned)l * (unsigned)a.row_stride) & 3u) : 0;
            if (STAGED) {
              unsigned *sw = reinterpret_cast<unsigned *>(stage);
              u32x2 *pw = reinterpret_cast<u32x2 *>(sw + lane * 6);
              pw[0] = u32x2{w[0], w[1]}; pw[1] = u32x2{w[2], w[3]}; pw[2] = u32x2{w[4], w[5]};
              __builtin_amdgcn_wave_barrier();
              const u32x4 v0 = *reinterpret_cast<const u32x4 *>(sw + lane * 4);
              const u32x4 v1 = *reinterpret_cast<const u32x4 *>(sw + ((lane & 31) + 64) * 4);
              __builtin_amdgcn_wave_barrier();
              store16_nt(out_frame, so0 + (unsigned)l * (unsigned)a.row_stride, v0);
              if (lane < 32) store16_nt(out_frame, so1 + (unsigned)l * (unsigned)a.row_stride, v1);
            }
            else if (SHIFTED && m) store24_nt_shifted(out_frame, off, w, bx, m);
            else if (SHIFTED) store24_nt<false>(out_frame, off, w);
            else store24_nt<!F420P_TEMPORAL>(out_frame, off, w);
          } else {
            uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
#pragma unroll
            for (int x = 0; x < 8; x++)
              if (x < npx) {
                dst[3 * x] = (uint8_t)clamp255(rr[x] >> 17); dst[3 * x + 1] = (uint8_t)clamp255(gg[x] >> 17); dst[3 * x + 2] = (uint8_t)clamp255(bb[x] >> 17);
              }
          }
        }
      }
      // slide the three-line window
#pragma unroll
      for (int j = 0; j < 6; j++) { cT[j] = cC[j]; cC[j] = cB[j]; }
    }
  };
  const bool whole = __builtin_amdgcn_ballot_w64(npx != 8 || nln != 8) == 0;
  if (whole && ((base_lo | (unsigned)a.row_stride) & 3u) && __builtin_amdgcn_ballot_w64(true) == ~0ull) lines(std::true_type{}, std::true_type{}, std::false_type{});
#if F420P_STAGED
  else if (whole && __builtin_amdgcn_ballot_w64(true) == ~0ull) lines(std::true_type{}, std::false_type{}, std::true_type{});
#endif
  else if (whole) lines(std::true_type{}, std::false_type{}, std::false_type{});
  else lines(std::false_type{}, std::false_type{}, std::false_type{});
}

// ==============================================================================================
// fused 4:2:2 kernel (Y 1x1, Cb/Cr subsampled 2x1), packed chroma
// ==============================================================================================
// fused420p_kernel without the vertical direction: the chroma planes have the luma plane's height, so a 128x128 tile
// needs (8 + 2) x 16 chroma blocks per component -- 1.25 rounds of transforms for the four waves where 4:2:0 needs
// one -- and no halo lines; the samples of a line go through the horizontal filter only.  Same gate as the packed
// 4:2:0 flavour (FAST arithmetic, chroma range_max < 2047), same store path.  Algorithmic bytes: 4 B in + 3 B out per pixel.
constexpr int F422_CROWS = 128;

// WIDE: the samples still travel through LDS as int16 pairs (|sample * 16| <= 4 * range_max < 2^15 for range_max < 8190,
// the fused 4:4:4 kernel's bound, which legitimate 8-bit content cannot exceed: sum |c| q <= 8 sqrt(64 * 128^2) by
// Cauchy-Schwarz), but the filter runs on unpacked 32-bit values: frames between the packed gate (2047) and 8190 --
// saturated colours with hard edges -- stay on a fused kernel instead of falling to the generic pair.
template <int MINW, bool QDEV, bool WIDE>
__global__ __launch_bounds__(F420_THREADS, MINW) void fused422_kernel(const Fused420Args a)
{
  __shared__ __attribute__((aligned(16))) unsigned cpair[F422_CROWS * F420_CPITCH];
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position(blockIdx.x, a); // (tile order: see there)
  if (tp.frame < 0) return; // the launch is padded to whole groups of eight tile rows
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  // ------------------------------------------------------------------ phase A: chroma -> LDS halves
  // 8 x 16 blocks per component, 64 per wave (waves 0, 1: Cb, waves 2, 3: Cr; upper / lower half of the tile), plus one
  // COLUMN of the 16 blocks left and right of them: the horizontal filter needs nothing else of the neighbours
  {
    const int comp = wave >> 1; // 0 = Cb (low halves), 1 = Cr (high halves); wave-uniform
    const int16_t *__restrict__ plane = coef + (comp ? a.off_cr : a.off_cb);
    const int gx0 = tx * 8, gy0 = ty * 16 + (wave & 1) * 8;
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    short *cp = reinterpret_cast<short *>(cpair) + comp; // this component's half of every dword
    u32x4 rows[8];
    { // the wave's 8 x 8 blocks: local block n = (lane >> 3) + 8 m is column n & 7 = lane >> 3, row n >> 3 = m
      const int xx = min(gx0 + (lane >> 3), a.bw_c - 1);
      fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
        const int yy = min(gy0 + m, a.bh_c - 1);
        return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * a.bw_c + xx) * 128));
      });
      const int cbx = lane & 7, cby = lane >> 3;
      if (gx0 + cbx < a.bw_c && gy0 + cby < a.bh_c) {
        int v[64];
        dequant_idct_sparse<!QDEV>(rows, frame_deltas<QDEV>(a, frame, 1 + comp), v);
#pragma unroll
        for (int r = 0; r < 8; r++) {
          short *dst = cp + 2 * ((8 * ((wave & 1) * 8 + cby) + r) * F420_CPITCH + 8 * cbx + 4);
#pragma unroll
          for (int x = 0; x < 8; x++) dst[2 * x] = (short)v[r * 8 + x];
        }
      }
    }
    { // halo columns: local block n = lane >> 3 (+ 8) is side n & 1 (0: left neighbour, 1: right neighbour), row n >> 1
      fetch_blocks16(rows, stage, lane, [&](int m) -> const u32x4 * {
        const int n = (lane >> 3) + 8 * m;
        const int xx = min(max((n & 1) ? gx0 + 8 : gx0 - 1, 0), a.bw_c - 1), yy = min(gy0 + (n >> 1), a.bh_c - 1);
        return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * a.bw_c + xx) * 128));
      });
      const int side = lane & 1, cby = lane >> 1, gx = side ? gx0 + 8 : gx0 - 1;
      if (lane < 16 && gx >= 0 && gx < a.bw_c && gy0 + cby < a.bh_c) {
        int col[8];
        dequant_idct_column(rows, frame_deltas<QDEV>(a, frame, 1 + comp), side == 0, col); // left neighbour: its last column, right one: its first
#pragma unroll
        for (int r = 0; r < 8; r++) cp[2 * ((8 * ((wave & 1) * 8 + cby) + r) * F420_CPITCH + (side ? 68 : 3))] = (short)col[r];
      }
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ edge fix-up (uniform branch): columns only
  {
    const int last_col = a.cw - 1 - tx * 64; // last valid chroma column, tile-relative
    if ((tx == 0) | (last_col < 64)) {
      if (tid < F422_CROWS) { // one thread per stored line
        unsigned *p = cpair + tid * F420_CPITCH;
        if (tx == 0) p[3] = p[4];
        if (last_col < 64) {
          const unsigned v = p[last_col + 4];
          for (int pc = last_col + 5; pc <= 68; pc++) p[pc] = v;
        }
      }
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ phase B: luma, upsampling, colour
  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  u32x4 rows[8];
  {
    const int16_t *__restrict__ plane = coef + a.off_y;
    const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
    const int x0 = gbx0 + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  }
  const int X0 = gbx * 8, Y0 = gby * 8;
  if (X0 >= a.width || Y0 >= a.height) return; // no barrier below this point
  int yv[64];
  dequant_idct_sparse<!QDEV, true>(rows, frame_deltas<QDEV>(a, frame, 0), yv, 0, LUMA_FOLD_R2);

  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * 3u;
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
  const bool fast_store = npx == 8;
  // chroma window of this block: lines pr = 8 by + l, columns pc = 4 bx + 3 + j
  const unsigned *c_base = cpair + (8 * by) * F420_CPITCH + 4 * bx;
  auto load6 = [](const unsigned *p, unsigned (&d)[6]) { // p is 16-byte aligned; wanted: p[3..8]
    const u32x4 mid = *reinterpret_cast<const u32x4 *>(p + 4);
    d[0] = p[3]; d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w; d[5] = p[8];
  };
#pragma unroll
  for (int l = 0; l < 8; l++) {
    {
      // no vertical filter (VerticalFilterCore<1>, upsampler.cpp:118-131: a copy); horizontal filter in place
      // (upsampler.cpp:291-303); src[k] = v[k + 1]
      unsigned v[6];
      load6(c_base + l * F420_CPITCH, v);
      unsigned u[8];
      int ub[8], ur[8]; // WIDE: the two components on their own, 32 bits
      if (WIDE) {
        int vb[6], vr[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { vb[j] = (int)(short)(v[j] & 0xffffu); vr[j] = (int)v[j] >> 16; }
#define MIJ_HFILTER(o, s)                                                                                        \
        o[7] = tap13(s[5], s[4], 1); o[6] = tap13(s[3], s[4], 2); o[5] = tap13(s[4], s[3], 1); o[4] = tap13(s[2], s[3], 2); \
        o[3] = tap13(s[3], s[2], 1); o[2] = tap13(s[1], s[2], 2); o[1] = tap13(o[2], s[1], 1); o[0] = tap13(s[0], s[1], 2);
        MIJ_HFILTER(ub, vb)
        MIJ_HFILTER(ur, vr)
#undef MIJ_HFILTER
      } else {
        u[7] = tap13_pk(v[5], v[4], 1);
        u[6] = tap13_pk(v[3], v[4], 2);
        u[5] = tap13_pk(v[4], v[3], 1);
        u[4] = tap13_pk(v[2], v[3], 2);
        u[3] = tap13_pk(v[3], v[2], 1);
        u[2] = tap13_pk(v[1], v[2], 2);
        u[1] = tap13_pk(u[2], v[1], 1); // src[1] has already been overwritten by out[2]
        u[0] = tap13_pk(v[0], v[1], 2);
      }
      if (l < nln) {
        int rr[8], gg[8], bb[8];
#pragma unroll
        for (int x = 0; x < 8; x++) {
          const int yk = luma13(yv[l * 8 + x]); // (level shift and rounding are inside: LUMA_FOLD_R2)
          if (WIDE) {
            rr[x] = mad24(ur[x], L_CR_R, yk); // still scaled by 2^17
            bb[x] = mad24(ub[x], L_CB_B, yk);
            gg[x] = mad24(ur[x], -L_CR_G, mad24(ub[x], -L_CB_G, yk));
          } else {
            rr[x] = mad16_hi(u[x], L_CR_R, yk); // still scaled by 2^17
            bb[x] = mad16_lo(u[x], L_CB_B, yk);
            gg[x] = dot2_16(u[x], -L_CB_G, -L_CR_G, yk);
          }
        }
        if (fast_store) {
          unsigned w[6];
          rgb_shift17_sat_pack(rr, gg, bb, w);
          store24_nt(out_frame, out_off + (unsigned)l * (unsigned)a.row_stride, w);
        } else {
          uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
#pragma unroll
          for (int x = 0; x < 8; x++)
            if (x < npx) {
              dst[3 * x] = (uint8_t)clamp255(rr[x] >> 17); dst[3 * x + 1] = (uint8_t)clamp255(gg[x] >> 17); dst[3 * x + 2] = (uint8_t)clamp255(bb[x] >> 17);
            }
        }
      }
    }
  }
}

// 12-bit flavour of the 4:2:2 kernel (SOF1, P = 12; 16-bit samples out, 4 B in + 6 B out per pixel): the same tile, the chroma
// samples as 32-bit values in two LDS planes (a 12-bit chroma sample times 16 does not fit 16 bits: 72 KB + the fetch staging
// = half a CU's LDS, two workgroups per CU), fused420_kernel<.., 12>'s colour stage behind the horizontal filter.  Gates:
// use_fused422_12 (capi.cpp), the 12-bit 4:2:0 kernel's.
template <bool QDEV, bool N12>
__global__ __launch_bounds__(F420_THREADS, 2) void fused422_12_kernel(const Fused420Args a)
{
  __shared__ __attribute__((aligned(16))) int cplane[2][F422_CROWS * F420_CPITCH];
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position(blockIdx.x, a);
  if (tp.frame < 0) return;
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  // ------------------------------------------------------------------ phase A: chroma -> LDS (see fused422_kernel)
  {
    const int comp = wave >> 1;
    const int16_t *__restrict__ plane = coef + (comp ? a.off_cr : a.off_cb);
    const int gx0 = tx * 8, gy0 = ty * 16 + (wave & 1) * 8;
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    int *cp = cplane[comp];
    u32x4 rows[8];
    {
      const int xx = min(gx0 + (lane >> 3), a.bw_c - 1);
      fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
        const int yy = min(gy0 + m, a.bh_c - 1);
        return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * a.bw_c + xx) * 128));
      });
      const int cbx = lane & 7, cby = lane >> 3;
      if (gx0 + cbx < a.bw_c && gy0 + cby < a.bh_c) {
        int v[64];
        dequant_idct_sparse<false>(rows, frame_deltas<QDEV>(a, frame, 1 + comp), v);
#pragma unroll
        for (int r = 0; r < 8; r++) {
          i32x4 *dst = reinterpret_cast<i32x4 *>(cp + (8 * ((wave & 1) * 8 + cby) + r) * F420_CPITCH + 8 * cbx + 4);
          dst[0] = i32x4{v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3]};
          dst[1] = i32x4{v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]};
        }
      }
    }
    {
      fetch_blocks16(rows, stage, lane, [&](int m) -> const u32x4 * {
        const int n = (lane >> 3) + 8 * m;
        const int xx = min(max((n & 1) ? gx0 + 8 : gx0 - 1, 0), a.bw_c - 1), yy = min(gy0 + (n >> 1), a.bh_c - 1);
        return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * a.bw_c + xx) * 128));
      });
      const int side = lane & 1, cby = lane >> 1, gx = side ? gx0 + 8 : gx0 - 1;
      if (lane < 16 && gx >= 0 && gx < a.bw_c && gy0 + cby < a.bh_c) {
        int col[8];
        dequant_idct_column(rows, frame_deltas<QDEV>(a, frame, 1 + comp), side == 0, col);
#pragma unroll
        for (int r = 0; r < 8; r++) cp[(8 * ((wave & 1) * 8 + cby) + r) * F420_CPITCH + (side ? 68 : 3)] = col[r];
      }
    }
  }
  __syncthreads();
  {
    const int last_col = a.cw - 1 - tx * 64; // last valid chroma column, tile-relative
    if ((tx == 0) | (last_col < 64)) {
      if (tid < 2 * F422_CROWS) { // one thread per stored line and component
        int *p = cplane[tid >> 7] + (tid & (F422_CROWS - 1)) * F420_CPITCH;
        if (tx == 0) p[3] = p[4];
        if (last_col < 64) {
          const int v = p[last_col + 4];
          for (int pc = last_col + 5; pc <= 68; pc++) p[pc] = v;
        }
      }
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ phase B: luma, upsampling, colour
  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  u32x4 rows[8];
  {
    const int16_t *__restrict__ plane = coef + a.off_y;
    const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
    const int x0 = gbx0 + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  }
  const int X0 = gbx * 8, Y0 = gby * 8;
  if (X0 >= a.width || Y0 >= a.height) return; // no barrier below this point
  int yv[64];
  dequant_idct_sparse<false>(rows, frame_deltas<QDEV>(a, frame, 0), yv);

  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * 6u;
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
  const bool fast_store = npx == 8;
  const int *cb_base = cplane[0] + (8 * by) * F420_CPITCH + 4 * bx;
  const int *cr_base = cplane[1] + (8 * by) * F420_CPITCH + 4 * bx;
  auto load6 = [](const int *p, int (&d)[6]) { // p is 16-byte aligned; wanted: p[3..8]
    const i32x4 mid = *reinterpret_cast<const i32x4 *>(p + 4);
    d[0] = p[3]; d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w; d[5] = p[8];
  };
  static_assert(L_CB_B % 4 == 0, "the blue product is taken at a quarter of the constant");
  auto c12 = [](int v) { return (unsigned)min(max(v, 0), 4095); };
#pragma unroll
  for (int l = 0; l < 8; l++) {
    int vb[6], vr[6], ub[8], ur[8];
    load6(cb_base + l * F420_CPITCH, vb);
    load6(cr_base + l * F420_CPITCH, vr);
    // horizontal filter in place (upsampler.cpp:291-303); src[k] = v[k + 1]
#define MIJ_HFILTER(o, s)                                                                                        \
    o[7] = tap13(s[5], s[4], 1); o[6] = tap13(s[3], s[4], 2); o[5] = tap13(s[4], s[3], 1); o[4] = tap13(s[2], s[3], 2); \
    o[3] = tap13(s[3], s[2], 1); o[2] = tap13(s[1], s[2], 2); o[1] = tap13(o[2], s[1], 1); o[0] = tap13(s[0], s[1], 2);
    MIJ_HFILTER(ub, vb)
    MIJ_HFILTER(ur, vr)
#undef MIJ_HFILTER
    if (l < nln) {
      int rr[8], gg[8], bb[8];
#pragma unroll
      for (int x = 0; x < 8; x++) colour12<N12>(yv[l * 8 + x], ub[x], ur[x], rr[x], gg[x], bb[x]);
      uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
      if (fast_store) {
        unsigned w[12];
#pragma unroll
        for (int x = 0; x < 8; x += 2) {
          w[3 * (x / 2) + 0] = c12(rr[x]) | (c12(gg[x]) << 16);
          w[3 * (x / 2) + 1] = c12(bb[x]) | (c12(rr[x + 1]) << 16);
          w[3 * (x / 2) + 2] = c12(gg[x + 1]) | (c12(bb[x + 1]) << 16);
        }
        store48(dst, w);
      } else {
        uint16_t *d16 = reinterpret_cast<uint16_t *>(dst);
#pragma unroll
        for (int x = 0; x < 8; x++)
          if (x < npx) {
            d16[3 * x] = (uint16_t)c12(rr[x]); d16[3 * x + 1] = (uint16_t)c12(gg[x]); d16[3 * x + 2] = (uint16_t)c12(bb[x]);
          }
      }
    }
  }
}

// ==============================================================================================
// fused 4:1:1 kernel (Y 1x1, Cb/Cr subsampled 4x1: DV-style sampling, `jpeg -s 1x1,4x1,4x1`)
// ==============================================================================================
// The 4:2:2 kernel with the four-fold horizontal core (HorizontalFilterCore<4>, upsampling/upsampler.cpp:367-387:
// out[0..7] of a block from the chroma samples c[-1], c[0], c[1], c[2] with the weights (3,5) (1,7) (1,7) (3,5) and their
// mirror images; the in-place order of the reference touches no sample it still needs, so the taps see the original four).
// A 128x128 tile holds 4 x 16 chroma blocks per component -- one round for waves 0 (Cb) and 2 (Cr) -- and the last / first
// COLUMN of the 16 blocks left / right of them, transformed column-only by waves 1 and 3.  Samples travel through LDS as
// (Cb, Cr) int16 pairs (chroma range_max < 8190), the filter and the colour stage run on unpacked 32-bit values (the WIDE
// arithmetic of fused422_kernel: the weights up to 7 leave no room for 16-bit sums).  Algorithmic bytes: 3 B in + 3 B out.
__device__ __forceinline__ int f8(int wa, int x, int wb, int y, int r);
constexpr int F411_CPITCH = 40; // dwords per LDS chroma line; column pc <-> chroma x_rel = pc - 4 (3 and 36: the halo columns)

template <int MINW, bool QDEV>
__global__ __launch_bounds__(F420_THREADS, MINW) void fused411_kernel(const Fused420Args a)
{
  __shared__ __attribute__((aligned(16))) unsigned cpair[F422_CROWS * F411_CPITCH];
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position(blockIdx.x, a); // (tile order: see there)
  if (tp.frame < 0) return; // the launch is padded to whole groups of eight tile rows
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  // ------------------------------------------------------------------ phase A: chroma -> LDS halves
  {
    const int comp = wave >> 1; // 0 = Cb (low halves), 1 = Cr (high halves); wave-uniform
    const int16_t *__restrict__ plane = coef + (comp ? a.off_cr : a.off_cb);
    const int gx0 = tx * 4, gy0 = ty * 16;
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    short *cp = reinterpret_cast<short *>(cpair) + comp; // this component's half of every dword
    // (two static indices and a select: a run-time index into the by-value argument block would send the block to scratch)
    const int *qc = QDEV ? frame_deltas<QDEV>(a, frame, 1 + comp) : (comp ? a.q[2] : a.q[1]);
    u32x4 rows[8];
    if ((wave & 1) == 0) { // the 4 x 16 blocks of the tile: local block n is column n & 3, row n >> 2
      fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
        const int n = (lane >> 3) + 8 * m;
        const int xx = min(gx0 + (n & 3), a.bw_c - 1), yy = min(gy0 + (n >> 2), a.bh_c - 1);
        return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * a.bw_c + xx) * 128));
      });
      const int cbx = lane & 3, cby = lane >> 2;
      if (gx0 + cbx < a.bw_c && gy0 + cby < a.bh_c) {
        int v[64];
        dequant_idct_sparse<!QDEV>(rows, qc, v);
#pragma unroll
        for (int r = 0; r < 8; r++) {
          short *dst = cp + 2 * ((8 * cby + r) * F411_CPITCH + 8 * cbx + 4);
#pragma unroll
          for (int x = 0; x < 8; x++) dst[2 * x] = (short)v[r * 8 + x];
        }
      }
    } else { // halo columns: local block n (0..31) is side n & 1 (0: left neighbour, 1: right neighbour), row n >> 1
      fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
        const int n = min((lane >> 3) + 8 * m, 31);
        const int xx = min(max((n & 1) ? gx0 + 4 : gx0 - 1, 0), a.bw_c - 1), yy = min(gy0 + (n >> 1), a.bh_c - 1);
        return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * a.bw_c + xx) * 128));
      });
      const int side = lane & 1, cby = lane >> 1, gx = side ? gx0 + 4 : gx0 - 1;
      if (lane < 32 && gx >= 0 && gx < a.bw_c && gy0 + cby < a.bh_c) {
        int col[8];
        dequant_idct_column(rows, qc, side == 0, col); // left neighbour: its last column, right one: its first
#pragma unroll
        for (int r = 0; r < 8; r++) cp[2 * ((8 * cby + r) * F411_CPITCH + (side ? 36 : 3))] = (short)col[r];
      }
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ edge fix-up (uniform branch): columns only
  {
    const int last_col = a.cw - 1 - tx * 32; // last valid chroma column, tile-relative
    if ((tx == 0) | (last_col < 32)) {
      if (tid < F422_CROWS) { // one thread per stored line
        unsigned *p = cpair + tid * F411_CPITCH;
        if (tx == 0) p[3] = p[4];
        if (last_col < 32) {
          const unsigned v = p[last_col + 4];
          for (int pc = last_col + 5; pc <= 36; pc++) p[pc] = v;
        }
      }
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ phase B: luma, upsampling, colour
  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  u32x4 rows[8];
  {
    const int16_t *__restrict__ plane = coef + a.off_y;
    const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
    const int x0 = gbx0 + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  }
  const int X0 = gbx * 8, Y0 = gby * 8;
  if (X0 >= a.width || Y0 >= a.height) return; // no barrier below this point
  int yv[64];
  dequant_idct_sparse<!QDEV, true>(rows, frame_deltas<QDEV>(a, frame, 0), yv, 0, LUMA_FOLD_R2);

  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * 3u;
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
  const bool fast_store = npx == 8;
  // chroma window of this block: lines 8 by + l, columns x_rel = 2 bx - 1 .. 2 bx + 2, i.e. pc = 2 bx + 3 .. 2 bx + 6
  const unsigned *c_base = cpair + (8 * by) * F411_CPITCH + 2 * bx + 3;
#pragma unroll
  for (int l = 0; l < 8; l++) {
    const unsigned *p = c_base + l * F411_CPITCH;
    const unsigned w[4] = {p[0], p[1], p[2], p[3]};
    int cb[4], cr[4], ub[8], ur[8];
#pragma unroll
    for (int j = 0; j < 4; j++) { cb[j] = (int)(short)(w[j] & 0xffffu); cr[j] = (int)w[j] >> 16; }
    // no vertical filter; the four-fold horizontal core on c[-1], c[0], c[1], c[2] = s[0..3]
#define MIJ_HFILTER4(o, s)                                                                                   \
    o[0] = f8(3, s[0], 5, s[1], 2); o[1] = f8(1, s[0], 7, s[1], 1); o[2] = f8(1, s[2], 7, s[1], 2); o[3] = f8(3, s[2], 5, s[1], 1); \
    o[4] = f8(3, s[1], 5, s[2], 2); o[5] = f8(1, s[1], 7, s[2], 1); o[6] = f8(1, s[3], 7, s[2], 2); o[7] = f8(3, s[3], 5, s[2], 1);
    MIJ_HFILTER4(ub, cb)
    MIJ_HFILTER4(ur, cr)
#undef MIJ_HFILTER4
    if (l < nln) {
      uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
      int rr[8], gg[8], bb[8];
#pragma unroll
      for (int x = 0; x < 8; x++) {
        const int yk = luma13(yv[l * 8 + x]); // (level shift and rounding are inside: LUMA_FOLD_R2)
        rr[x] = mad24(ur[x], L_CR_R, yk); // still scaled by 2^17
        bb[x] = mad24(ub[x], L_CB_B, yk);
        gg[x] = mad24(ur[x], -L_CR_G, mad24(ub[x], -L_CB_G, yk));
      }
      if (fast_store) {
        unsigned wd[6];
        rgb_shift17_sat_pack(rr, gg, bb, wd);
        u32x2_any *d2 = reinterpret_cast<u32x2_any *>(dst);
        __builtin_nontemporal_store(u32x2{wd[0], wd[1]}, d2);
        __builtin_nontemporal_store(u32x2{wd[2], wd[3]}, d2 + 1);
        __builtin_nontemporal_store(u32x2{wd[4], wd[5]}, d2 + 2);
      } else {
#pragma unroll
        for (int x = 0; x < 8; x++)
          if (x < npx) {
            dst[3 * x] = (uint8_t)clamp255(rr[x] >> 17); dst[3 * x + 1] = (uint8_t)clamp255(gg[x] >> 17); dst[3 * x + 2] = (uint8_t)clamp255(bb[x] >> 17);
          }
      }
    }
  }
}

// ==============================================================================================
// fused 4:4:0 kernel (Y 1x1, Cb/Cr subsampled 1x2: what a losslessly rotated 4:2:2 picture is), packed chroma
// ==============================================================================================
// The 4:2:2 kernel turned by ninety degrees: chroma planes of full width and half height, so a 128x128 tile holds 16 x 8
// chroma blocks per component = one transform round for the four waves; of the 16 blocks above and below them the
// vertical filter needs a single LINE each (dequant_idct_line: the first pass in full, one inner product per column).
// Vertical filter only (Upsampler<1,2>: VerticalFilterCore<2>, HorizontalFilterCore<1> = copy), same 16-bit gate.
constexpr int F440_CROWS = 66, F440_CPITCH = 128;
template <int MINW, bool QDEV, bool WIDE> // WIDE: see fused422_kernel
__global__ __launch_bounds__(F420_THREADS, MINW) void fused440_kernel(const Fused420Args a)
{
  __shared__ __attribute__((aligned(16))) unsigned cpair[F440_CROWS * F440_CPITCH];
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position(blockIdx.x, a); // (tile order: see there)
  if (tp.frame < 0) return; // the launch is padded to whole groups of eight tile rows
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  // ------------------------------------------------------------------ phase A: chroma -> LDS halves
  // line pr of the LDS plane is chroma line ty * 64 + pr - 1 (pr = 0 and 65: the lines above and below the tile)
  {
    const int comp = wave >> 1; // 0 = Cb (low halves), 1 = Cr (high halves); wave-uniform
    const int16_t *__restrict__ plane = coef + (comp ? a.off_cr : a.off_cb);
    const int gx0 = tx * 16 + (wave & 1) * 8, gy0 = ty * 8; // this wave: eight block columns, the tile's eight block rows
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    short *cp = reinterpret_cast<short *>(cpair) + comp; // this component's half of every dword
    u32x4 rows[8];
    { // local block n = (lane >> 3) + 8 m: column n & 7 = lane >> 3, row n >> 3 = m
      const int xx = min(gx0 + (lane >> 3), a.bw_c - 1);
      fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
        const int yy = min(gy0 + m, a.bh_c - 1);
        return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * a.bw_c + xx) * 128));
      });
      const int cbx = lane & 7, cby = lane >> 3;
      if (gx0 + cbx < a.bw_c && gy0 + cby < a.bh_c) {
        int v[64];
        dequant_idct_sparse<!QDEV>(rows, frame_deltas<QDEV>(a, frame, 1 + comp), v);
#pragma unroll
        for (int r = 0; r < 8; r++) {
          short *dst = cp + 2 * ((8 * cby + r + 1) * F440_CPITCH + 8 * ((wave & 1) * 8 + cbx));
#pragma unroll
          for (int x = 0; x < 8; x++) dst[2 * x] = (short)v[r * 8 + x];
        }
      }
    }
    { // halo lines: the lane's two chunks are row k = lane & 7 of the block above (m = 0) and below (m = 1) column lane >> 3
      constexpr int W[8] = {512, FIX9(1.501321110) - FIX9(0.899976223) - FIX9(0.390180644) + FIX9(1.175875602), FIX9(0.541196100) + FIX9(0.765366865),
                            FIX9(1.175875602), 512, FIX9(1.175875602) - FIX9(0.390180644), FIX9(0.541196100), FIX9(1.175875602) - FIX9(0.899976223)};
      const int k = lane & 7, nb = lane >> 3;
      const int xx = min(gx0 + nb, a.bw_c - 1);
      const u32x4 above = *reinterpret_cast<const u32x4 *>(pbase + (unsigned)((max(gy0 - 1, 0) * a.bw_c + xx) * 128));
      const u32x4 below = *reinterpret_cast<const u32x4 *>(pbase + (unsigned)((min(gy0 + 8, a.bh_c - 1) * a.bw_c + xx) * 128));
      // the deltas of row k, per lane: read from the kernel argument segment itself (indexing the by-value struct with a lane
      // dependent index would make the compiler copy it to scratch)
      typedef const __attribute__((address_space(4))) int kernarg_int;
      int qrow[8];
      if (QDEV) {
        const int *qk = a.qdev + ((int64_t)frame * 4 + 1 + comp) * 64 + k * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) qrow[i] = qk[i];
      } else {
        kernarg_int *qk = (kernarg_int *)__builtin_amdgcn_kernarg_segment_ptr() + (offsetof(Fused420Args, q) / sizeof(int) + (1 + comp) * QROW + k * 8);
#pragma unroll
        for (int i = 0; i < 8; i++) qrow[i] = qk[i];
      }
      const int weight = W[k];
      int line[8];
      short *dst = cp + 2 * (8 * ((wave & 1) * 8 + nb) + k);
      const bool col_ok = gx0 + nb < a.bw_c;
      auto mine = [&](const int (&l)[8]) { // line[k] without a register-indexed access
        const int a01 = (k & 1) ? l[1] : l[0], a23 = (k & 1) ? l[3] : l[2], a45 = (k & 1) ? l[5] : l[4], a67 = (k & 1) ? l[7] : l[6];
        const int lo = (k & 2) ? a23 : a01, hi = (k & 2) ? a67 : a45;
        return (short)((k & 4) ? hi : lo);
      };
      dequant_idct_line8(above, qrow, (k & 1) ? -weight : weight, line); // the block above: its last line (even - odd)
      if (col_ok && gy0 > 0) dst[0] = mine(line);
      dequant_idct_line8(below, qrow, weight, line); // the block below: its first line (even + odd)
      if (col_ok && gy0 + 8 < a.bh_c) dst[2 * (F440_CROWS - 1) * F440_CPITCH] = mine(line);
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ edge fix-up (uniform branch): lines only
  // (upsampler.cpp:100-112: the line above the first and below the last chroma line is that line again)
  {
    const int last_row = a.ch - 1 - ty * 64; // last valid chroma line, tile-relative
    if ((ty == 0) | (last_row < 64)) {
      if (tid < F440_CPITCH) {
        unsigned *p = cpair + tid;
        if (ty == 0) p[0] = p[F440_CPITCH];
        if (last_row < 64) {
          const unsigned v = p[(last_row + 1) * F440_CPITCH];
          for (int pr = last_row + 2; pr < F440_CROWS; pr++) p[pr * F440_CPITCH] = v;
        }
      }
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ phase B: luma, upsampling, colour
  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  u32x4 rows[8];
  {
    const int16_t *__restrict__ plane = coef + a.off_y;
    const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
    const int x0 = gbx0 + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  }
  const int X0 = gbx * 8, Y0 = gby * 8;
  if (X0 >= a.width || Y0 >= a.height) return; // no barrier below this point
  int yv[64];
  dequant_idct_sparse<!QDEV, true>(rows, frame_deltas<QDEV>(a, frame, 0), yv, 0, LUMA_FOLD_R2);

  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * 3u;
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
  const bool fast_store = npx == 8;
  // chroma window of this block: lines pr = 4 by + m (+0 top, +1 cur, +2 bot), columns 8 bx + x
  const unsigned *c_base = cpair + (4 * by) * F440_CPITCH + 8 * bx;
  auto load8 = [](const unsigned *p, unsigned (&d)[8]) {
    const u32x4 lo = *reinterpret_cast<const u32x4 *>(p), hi = *reinterpret_cast<const u32x4 *>(p + 4);
    d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w; d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
  };
  unsigned cT[8], cC[8], cB[8];
  load8(c_base, cT);
  load8(c_base + F440_CPITCH, cC);
#pragma unroll
  for (int m = 0; m < 4; m++) {
    load8(c_base + (m + 2) * F440_CPITCH, cB);
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int l = 2 * m + half;
      // vertical filter (upsampler.cpp:149-165), both components at once; no horizontal filter (HorizontalFilterCore<1>: a copy)
      unsigned u[8];
      int ub[8], ur[8]; // WIDE: the two components on their own, 32 bits
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const unsigned o = half ? cB[j] : cT[j];
        const int r = ((j & 1) ^ half) ? 1 : 2;
        if (WIDE) {
          ub[j] = tap13((int)(short)(o & 0xffffu), (int)(short)(cC[j] & 0xffffu), r);
          ur[j] = tap13((int)o >> 16, (int)cC[j] >> 16, r);
        } else {
          u[j] = tap13_pk(o, cC[j], (short)r);
        }
      }
      if (l < nln) {
        int rr[8], gg[8], bb[8];
#pragma unroll
        for (int x = 0; x < 8; x++) {
          const int yk = luma13(yv[l * 8 + x]); // (level shift and rounding are inside: LUMA_FOLD_R2)
          if (WIDE) {
            rr[x] = mad24(ur[x], L_CR_R, yk); // still scaled by 2^17
            bb[x] = mad24(ub[x], L_CB_B, yk);
            gg[x] = mad24(ur[x], -L_CR_G, mad24(ub[x], -L_CB_G, yk));
          } else {
            rr[x] = mad16_hi(u[x], L_CR_R, yk); // still scaled by 2^17
            bb[x] = mad16_lo(u[x], L_CB_B, yk);
            gg[x] = dot2_16(u[x], -L_CB_G, -L_CR_G, yk);
          }
        }
        if (fast_store) {
          unsigned w[6];
          rgb_shift17_sat_pack(rr, gg, bb, w);
          store24_nt(out_frame, out_off + (unsigned)l * (unsigned)a.row_stride, w);
        } else {
          uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
#pragma unroll
          for (int x = 0; x < 8; x++)
            if (x < npx) {
              dst[3 * x] = (uint8_t)clamp255(rr[x] >> 17); dst[3 * x + 1] = (uint8_t)clamp255(gg[x] >> 17); dst[3 * x + 2] = (uint8_t)clamp255(bb[x] >> 17);
            }
        }
      }
    }
    // slide the three-line window
#pragma unroll
    for (int j = 0; j < 8; j++) { cT[j] = cC[j]; cC[j] = cB[j]; }
  }
}

// ==============================================================================================
// fused JPEG XT profile C kernel: 8-bit 4:2:0 legacy frame + 12-bit 4:4:4 residual frame -> 16-bit codes
// ==============================================================================================
// The shape BASELINE config 5 names.  Decomposition of fused420_kernel<true>: phase A puts the tile's chroma samples of
// the legacy frame into LDS; in phase B every lane owns one 8x8 pixel block and transforms FOUR coefficient blocks for
// it: the three residual blocks first -- their samples are clamped to [0, 2^16) right away, which is the first thing
// the merge does with them (Q table of the subset: clamp, scale by 16), so two of them share a register: 96 registers
// for the block's residual -- then the legacy luma block.  Per line it upsamples chroma from LDS exactly as
// fused420_kernel does, runs the legacy colour stage, looks the three 8-bit results up in the L tables (LDS), runs
// the residual chain of xt_merge_kernel on the packed samples and stores eight pixels = 48 bytes.
// All transforms are the FAST flavour without level shift; for the residual frame that is exact when
// max_block sum |c| q < 2^16: the multiplier inputs then stay below 2^23 and the sums in front of the rounding shifts
// below 2^31 (|row output| <= 16 M_r 725 / 512 with M_r the row's share of the sum, 725 = largest entry of the scaled
// transform matrix; the column pass sees 22.7 M in total), so wrapping 32-bit and 24-bit-operand arithmetic agree.
// The level shift 2^11 << 7 passes through both rounding shifts exactly and comes out as 2^15 (see dequant_idct).
#ifndef FXT_PREFETCH
#define FXT_PREFETCH 0
#endif
constexpr int FXT_MINW = 2;
__global__ __launch_bounds__(F420_THREADS, FXT_MINW) void fusedxt420_kernel(const Fused420Args a, const FusedXtExtra x)
{
  __shared__ __attribute__((aligned(16))) int cplane[2][F420_CROWS * F420_CPITCH];
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];
  __shared__ int ltab[3 * 256];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position<XT_TILE_ORDER>(blockIdx.x, a); // (tile order: see there)
  if (tp.frame < 0) return; // the launch is padded to whole groups of eight tile rows
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
  const int X0 = (gbx0 + bx) * 8, Y0 = (ty * F420_TILE_BLOCKS + by) * 8;
  u32x4 rows[8];
  // the wave's 16 x 4 blocks of a plane of bw x bh blocks: local block n = (lane >> 3) + 8 m sits at column
  // (lane >> 3) + 8 (m & 1), row m >> 1; blocks outside the plane are redirected to a valid one and never used
  auto load_plane = [&](u32x4 (&raw)[8], const int16_t *__restrict__ plane, int bw, int bh) {
    const int x0 = gbx0 + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(plane) + (lane & 7) * 16;
    load_blocks(raw, [&](int m) -> const u32x4 * {
      const int xx = min(x0 + 8 * (m & 1), bw - 1), yy = min(gby0 + (m >> 1), bh - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * bw + xx) * 128));
    });
  };
  // FXT_PREFETCH (A-B builds): a plane's blocks are requested while the plane before them is transformed (two waves per SIMD hide
  // little of a fetch by themselves), the first residual plane's in front of phase A
#if FXT_PREFETCH
  u32x4 rawA[8], rawB[8];
  load_plane(rawA, coef + x.off_r[0], x.bw_r, x.bh_r);
#endif

  for (int i = tid; i < 3 * 256; i += F420_THREADS) ltab[i] = x.ltable[i] - x.out_shift; // the merge subtracts it anyway
  f420_chroma_to_lds<true, false, true>(a, coef, cplane, stage, lane, wave, tx, ty); // (the legacy frame passed the 16384 range check: use_fusedxt)
  __syncthreads();
  f420_chroma_edges(a, cplane, tid, tx, ty);

  // ------------------------------------------------------------------ residual blocks -> packed, clamped samples
  unsigned rp0[32], rp1[32], rp2[32];
  auto residual_rows = [&](const int *__restrict__ q, unsigned (&rp)[32]) {
    int v[64];
    dequant_idct_sparse(rows, q, v);
#pragma unroll
    for (int i = 0; i < 32; i++) {
      // clamp(v + 2^15, 0, 2^16 - 1) - 2^15 as int16: the saturating conversion does clamp and pack for two samples
      unsigned d;
      asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(d) : "v"(v[2 * i]), "v"(v[2 * i + 1])); // volatile: pack here, see pack_lo16_now
      rp[i] = d;
    }
  };
  int yv[64];
#if FXT_PREFETCH
  load_plane(rawB, coef + x.off_r[1], x.bw_r, x.bh_r);
  __builtin_amdgcn_sched_barrier(0);
  transpose_blocks(rows, stage, lane, rawA);
  residual_rows(x.rq[0], rp0);
  load_plane(rawA, coef + x.off_r[2], x.bw_r, x.bh_r);
  __builtin_amdgcn_sched_barrier(0);
  transpose_blocks(rows, stage, lane, rawB);
  residual_rows(x.rq[1], rp1);
  load_plane(rawB, coef + a.off_y, a.bw_y, a.bh_y);
  __builtin_amdgcn_sched_barrier(0);
  transpose_blocks(rows, stage, lane, rawA);
  residual_rows(x.rq[2], rp2);
  // ------------------------------------------------------------------ legacy luma
  transpose_blocks(rows, stage, lane, rawB);
  dequant_idct_sparse<true, true>(rows, a.q[0], yv, 0, LUMA_FOLD_R2);
#else
  auto fetch_plane = [&](const int16_t *__restrict__ plane, int bw, int bh) {
    u32x4 raw[8];
    load_plane(raw, plane, bw, bh);
    transpose_blocks(rows, stage, lane, raw);
  };
  fetch_plane(coef + x.off_r[0], x.bw_r, x.bh_r);
  residual_rows(x.rq[0], rp0);
  fetch_plane(coef + x.off_r[1], x.bw_r, x.bh_r);
  residual_rows(x.rq[1], rp1);
  fetch_plane(coef + x.off_r[2], x.bw_r, x.bh_r);
  residual_rows(x.rq[2], rp2);
  // ------------------------------------------------------------------ legacy luma
  fetch_plane(coef + a.off_y, a.bw_y, a.bh_y);
  dequant_idct_sparse<true, true>(rows, a.q[0], yv, 0, LUMA_FOLD_R2);
#endif

  const bool active = X0 < a.width && Y0 < a.height;
  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * 6u;
  const int npx = min(8, a.width - X0);
  const int nln = active ? min(8, a.height - Y0) : 0;
  const bool fast_store = npx == 8;

  const int *cb_base = cplane[0] + (4 * by) * F420_CPITCH + 4 * bx;
  const int *cr_base = cplane[1] + (4 * by) * F420_CPITCH + 4 * bx;
  auto load6 = [](const int *p, int (&d)[6]) {
    const i32x4 mid = *reinterpret_cast<const i32x4 *>(p + 4);
    d[0] = p[3]; d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w; d[5] = p[8];
  };
  // constants of the merge (see xt_merge_kernel)
  const int pinf = (x.out_max >> 1) - (x.out_max >> 6) - 1, minf = -pinf - 1; // largest finite half 0x7bff, and its mirror
  const unsigned pinf2 = (unsigned)pinf * 0x10001u, minf2 = ((unsigned)minf & 0xffffu) * 0x10001u;

  int cbT[6], cbC[6], cbB[6], crT[6], crC[6], crB[6];
  load6(cb_base, cbT); load6(cb_base + F420_CPITCH, cbC);
  load6(cr_base, crT); load6(cr_base + F420_CPITCH, crC);
#pragma unroll
  for (int m = 0; m < 4; m++) {
    load6(cb_base + (m + 2) * F420_CPITCH, cbB);
    load6(cr_base + (m + 2) * F420_CPITCH, crB);
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int l = 2 * m + half;
      int vb[6], vr[6];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const int rnd = ((j & 1) ^ half) ? 1 : 2;
        vb[j] = tap13(half ? cbB[j] : cbT[j], cbC[j], rnd);
        vr[j] = tap13(half ? crB[j] : crT[j], crC[j], rnd);
      }
      int ub[8], ur[8];
      auto hfilt = [](const int (&v)[6], int (&o)[8]) {
        o[7] = tap13(v[5], v[4], 1);
        o[6] = tap13(v[3], v[4], 2);
        o[5] = tap13(v[4], v[3], 1);
        o[4] = tap13(v[2], v[3], 2);
        o[3] = tap13(v[3], v[2], 1);
        o[2] = tap13(v[1], v[2], 2);
        o[1] = tap13(o[2], v[1], 1); // src[1] has already been overwritten by out[2]
        o[0] = tap13(v[0], v[1], 2);
      };
      hfilt(vb, ub);
      hfilt(vr, ur);
      int mm[24]; // legacy + residual - output shift, R G B of the eight pixels
#pragma unroll
      for (int xx = 0; xx < 8; xx++) {
        // legacy chain: L transformation, clamp to 8 bits, L table (output shift already subtracted)
        const int yk = luma13(yv[l * 8 + xx]); // (level shift and rounding are inside: LUMA_FOLD_R2)
        const int lr = clamp255(mad24(ur[xx], L_CR_R, yk) >> 17);
        const int lg = clamp255(mad24(ur[xx], -L_CR_G, mad24(ub[xx], -L_CB_G, yk)) >> 17);
        const int lb = clamp255(mad24(ub[xx], L_CB_B, yk) >> 17);
        const int lv[3] = {ltab[lr], ltab[256 + lg], ltab[512 + lb]};
        // residual chain on the packed samples s = q - 2^15 (q = Q table output / 16): R transformation
        //   rr = 16 q_y + ((d L + 256) >> 9), R2 table r2 = (clamp(rr, 0, 16 (out_max + 1) - 1) + 8) >> 4
        // = clamp((rr + 8) >> 4, 0, out_max + 1); with 16 q_y = 16 s_y + 2^19 and 2^19 = 2^28 >> 9, 8 = 4096 >> 9 the
        // constants move into the multiply-add: all exact (|d L| < 2^29, so nothing wraps)
        const int i = (l * 8 + xx) >> 1;
        const bool hi = (xx & 1) != 0;
        int r2[3];
        if (x.rtrafo_ycbcr) {
          const int C = 256 + (1 << 28) + 4096;
          const int t0 = hi ? mad16_hi(rp2[i], L_CR_R, C) : mad16_lo(rp2[i], L_CR_R, C);
          const int t1 = hi ? mad16_hi(rp1[i], -L_CB_G, mad16_hi(rp2[i], -L_CR_G, C)) : mad16_lo(rp1[i], -L_CB_G, mad16_lo(rp2[i], -L_CR_G, C));
          const int t2 = hi ? mad16_hi(rp1[i], L_CB_B, C) : mad16_lo(rp1[i], L_CB_B, C);
          const int u0 = hi ? mad16_hi(rp0[i], 16, t0 >> 9) : mad16_lo(rp0[i], 16, t0 >> 9);
          const int u1 = hi ? mad16_hi(rp0[i], 16, t1 >> 9) : mad16_lo(rp0[i], 16, t1 >> 9);
          const int u2 = hi ? mad16_hi(rp0[i], 16, t2 >> 9) : mad16_lo(rp0[i], 16, t2 >> 9);
          r2[0] = min(max(u0 >> 4, 0), x.out_max + 1);
          r2[1] = min(max(u1 >> 4, 0), x.out_max + 1);
          r2[2] = min(max(u2 >> 4, 0), x.out_max + 1);
        } else { // identity: (16 q + 8) >> 4 = q
          r2[0] = (hi ? (int)rp0[i] >> 16 : (int)(short)rp0[i]) + 32768;
          r2[1] = (hi ? (int)rp1[i] >> 16 : (int)(short)rp1[i]) + 32768;
          r2[2] = (hi ? (int)rp2[i] >> 16 : (int)(short)rp2[i]) + 32768;
        }
#pragma unroll
        for (int c = 0; c < 3; c++) mm[3 * xx + c] = lv[c] + r2[c];
      }
      // clamp and, for float output, INVERT_NEGS (ycbcrtrafo.cpp:66) -- two samples per instruction: the saturating
      // conversion to int16 cannot cut into the half-float range [minf, pinf], which lies inside int16
      unsigned w[12];
#pragma unroll
      for (int i = 0; i < 12; i++) {
        if (x.is_float) {
          unsigned pk, sg;
          asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(pk) : "v"(mm[2 * i]), "v"(mm[2 * i + 1]));
          asm("v_pk_max_i16 %0, %1, %2" : "=v"(pk) : "v"(pk), "v"(minf2));
          asm("v_pk_min_i16 %0, %1, %2" : "=v"(pk) : "v"(pk), "v"(pinf2));
          asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(sg) : "v"(pk)); // the constant has no high half of its own
          w[i] = pk ^ (sg & 0x7fff7fffu);
        } else {
          w[i] = (unsigned)min(max(mm[2 * i], 0), x.out_max) | ((unsigned)min(max(mm[2 * i + 1], 0), x.out_max) << 16);
        }
      }
      if (l < nln) {
        uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
        if (fast_store) {
          store48<FXT_TEMPORAL != 0>(dst, w);
        } else {
          unsigned short *d16 = reinterpret_cast<unsigned short *>(dst);
#pragma unroll
          for (int k = 0; k < 24; k++)
            if (k < 3 * npx) d16[k] = (unsigned short)(w[k >> 1] >> ((k & 1) * 16));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) { cbT[j] = cbC[j]; cbC[j] = cbB[j]; crT[j] = crC[j]; crC[j] = crB[j]; }
  }
}

// ==============================================================================================
// fused JPEG XT profile C kernel for residual frames with HIDDEN BITS (the reference encoder's -rR n, RFIN boxes):
// 8-bit 4:2:0 legacy frame + 4:4:4 residual frame of 13..16 bits whose coefficients are int32
// ==============================================================================================
// fusedxt420_kernel's decomposition with three differences.
// (1) The residual coefficients are int32 (two int16 slots each in the frame's buffer: 256-byte blocks, fetched in two halves).
// (2) A residual frame of more than 12 bits is transformed by the reference with IDCT<4,QUAD> (codestream/tables.cpp:1876-1891):
//     64-bit butterflies, but `(dptr[0 << 3] + dptr[4 << 3]) << FIX_BITS` of the second pass is a LONG expression
//     (dct/idct.cpp:297-298) and WRAPS: with the level shift 2^(P-1) << 7 sitting in row 0 the sum times 512 leaves 32 bits
//     whenever it reaches 2^22.  Everything else is exact, so the result is the exact transform minus k 2^32 >> 12 = k 2^20 in the
//     four outputs that the wrapped term feeds (rows 0, 7, 3, 4 for s0 + s4; 1, 6, 2, 5 for s0 - s4), k = the number of times
//     (s0 +- s4 + level) 512 wraps: floor((t + 2^22) / 2^23).  The exact transform is the FAST 32-bit one without level shift
//     (exact for sum |c| q < 2^16, the host's gate) plus the level shift, which passes both rounding shifts as 2^(P+3).
//     Checked against the oracle's literal 64-bit restatement on random blocks for P = 13..16 before it was written down here.
// (3) The residual samples do not fit 16 bits: after the Q table of the subset (clamp to [0, 2^(P+4)), scale to 2^20) each is
//     a 20-bit number; minus 2^19 -- what the R transformation subtracts from the chroma ones anyway -- the three of a pixel
//     are kept in two registers (20 + 12 | 20 + 12 bits).  The R transformation runs on them in 64-bit multiply-adds
//     (|d| < 2^19 times a 14-bit constant), the rest of the merge is fusedxt420_kernel's.
// 128 registers for the block's residual instead of 96: one wave per SIMD less than the 12-bit kernel.
__device__ __forceinline__ void idct_column_quadwrap(int &s0, int &s1, int &s2, int &s3, int &s4, int &s5, int &s6, int &s7, int level7)
{
  // where the reference's LONG expression wraps (see above): t = s0 +- s4 + (level shift << 7), k = floor((t + 2^22) / 2^23)
  const int k0 = (s0 + s4 + level7 + (1 << 22)) >> 23, k1 = (s0 - s4 + level7 + (1 << 22)) >> 23;
  idct_1d<true, 12>(s0, s1, s2, s3, s4, s5, s6, s7);
  const int f0 = k0 << 20, f1 = k1 << 20;
  s0 -= f0; s7 -= f0; s3 -= f0; s4 -= f0;
  s1 -= f1; s6 -= f1; s2 -= f1; s5 -= f1;
}

// Round 6: two waves per SIMD.  The block's legacy luma samples wait in LDS for the merge (eight int16 per line and lane:
// (sample * 16 + 2056) fits sixteen bits on the 16384 gate) instead of in 64 registers beside the 128 of packed residual samples,
// and the L tables shrink to int16 (entries minus the output shift lie in [-2^15, 2^15)): 80 512 bytes per workgroup, two of
// them per CU, and a register budget of 256 (profiles/r05/xt_kernels.txt has round 5's counters; profiles/r06/xt_kernels.txt these).
// LUMA_LDS = false is round 5's kernel (one wave per SIMD, the luma block in registers): legacy frames whose luma leaves the
// int16 line (sum |c| q >= 7600 -- 4 * that + 2056 must stay below 2^15: no photograph, but nothing forbids it) keep it.
#ifndef XTW_HALF_BARRIER
#define XTW_HALF_BARRIER 1
#endif
template <bool LUMA_LDS>
__global__ __launch_bounds__(F420_THREADS, LUMA_LDS ? 2 : 1) void fusedxtw420_kernel(const Fused420Args a, const FusedXtExtra x)
{
  __shared__ __attribute__((aligned(16))) int cplane[2][F420_CROWS * F420_CPITCH];
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];
  typedef typename std::conditional<LUMA_LDS, short, int>::type ltab_t;
  __shared__ ltab_t ltab[3 * 256];
  __shared__ __attribute__((aligned(16))) u32x4 luma_lines[LUMA_LDS ? 8 * F420_THREADS : 1]; // [line][thread]: 8 x int16, no two lanes on one bank

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position<XT_TILE_ORDER>(blockIdx.x, a); // (tile order: see there)
  if (tp.frame < 0) return; // the launch is padded to whole groups of eight tile rows
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  for (int i = tid; i < 3 * 256; i += F420_THREADS) ltab[i] = (ltab_t)(x.ltable[i] - x.out_shift);
  f420_chroma_to_lds<true, false, true>(a, coef, cplane, stage, lane, wave, tx, ty); // (the legacy frame passed the 16384 range check: use_fusedxt)
  __syncthreads();
  f420_chroma_edges(a, cplane, tid, tx, ty);

  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
  const int X0 = (gbx0 + bx) * 8, Y0 = (ty * F420_TILE_BLOCKS + by) * 8;
  u32x4 rows[8];

  // ------------------------------------------------------------------ legacy luma
  {
    const int x0 = gbx0 + (lane >> 3);
    const char *pbase = reinterpret_cast<const char *>(coef + a.off_y) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int xx = min(x0 + 8 * (m & 1), a.bw_y - 1), yy = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * a.bw_y + xx) * 128));
    });
  }
  int yv[LUMA_LDS ? 1 : 64];
  if constexpr (LUMA_LDS) {
    int yw[64];
    dequant_idct_sparse<true, true>(rows, a.q[0], yw, 0, LUMA_FOLD_R2);
#pragma unroll
    for (int l = 0; l < 8; l++) {
      u32x4 w;
      w.x = pack_lo16_now(yw[l * 8 + 1] >> 13, yw[l * 8 + 0] >> 13);
      w.y = pack_lo16_now(yw[l * 8 + 3] >> 13, yw[l * 8 + 2] >> 13);
      w.z = pack_lo16_now(yw[l * 8 + 5] >> 13, yw[l * 8 + 4] >> 13);
      w.w = pack_lo16_now(yw[l * 8 + 7] >> 13, yw[l * 8 + 6] >> 13);
      luma_lines[l * F420_THREADS + tid] = w;
    }
  } else {
    int yw[64];
    dequant_idct_sparse<true, true>(rows, a.q[0], yw, 0, LUMA_FOLD_R2);
#pragma unroll
    for (int i = 0; i < 64; i++) yv[LUMA_LDS ? 0 : i] = yw[i];
  }

  __builtin_amdgcn_sched_barrier(0); // (the residual blocks' fetches stay behind the luma block's transform: register pressure)
  // ------------------------------------------------------------------ residual blocks -> 20-bit samples minus 2^19, packed
  const int rprec = x.rprecision;                 // 13..16
  const int level7 = 1 << (rprec + 6);            // (2^(P-1)) << 7: the level shift as the column pass sees it in row 0
  const int level_out = 1 << (rprec + 3);         // ... and as it leaves the transform
  const int rmax = (1 << (rprec + 4)) - 1, qshift = 16 - rprec;
  // two registers per pixel: rA = Y sample (20 bits) | low 12 bits of the Cb sample, rB = Cr sample (20 bits) | high 8 bits of
  // the Cb sample.  Those eight bits wait in hB, four to a register, while the Cr block is transformed: 80 registers of samples
  // beside that transform instead of 128 (two waves per SIMD leave 256 in all)
  unsigned rA[64], rB[LUMA_LDS ? 32 : 64], hB[16];
  const int omax16 = ((x.out_max + 1) << 4) - 1;
  auto residual_block = [&](int64_t off, const int *__restrict__ q, int which) {
    const char *plane = reinterpret_cast<const char *>(coef + off);
    int v[64];
#pragma unroll
    for (int h = 0; h < 2; h++) { // coefficients 32 h .. 32 h + 31 of every block: rows 4 h .. 4 h + 3
      const int x0 = gbx0 + (lane >> 3);
      const char *pbase = plane + (lane & 7) * 16 + h * 128;
      fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
        const int xx = min(x0 + 8 * (m & 1), x.bw_r - 1), yy = min(gby0 + (m >> 1), x.bh_r - 1);
        return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((yy * x.bw_r + xx) * 256));
      });
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int base = 32 * h + 4 * k;
        v[base + 0] = __mul24((int)rows[k].x, q[base + 0]);
        v[base + 1] = __mul24((int)rows[k].y, q[base + 1]);
        v[base + 2] = __mul24((int)rows[k].z, q[base + 2]);
        v[base + 3] = __mul24((int)rows[k].w, q[base + 3]);
      }
      if (LUMA_LDS && XTW_HALF_BARRIER) __builtin_amdgcn_sched_barrier(0); // (the second half's rows are not asked for before the first half's are spent)
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
      idct_1d<true, 9>(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]);
    // column by column: transform, then clamp / scale / pack the column's eight samples at once -- the transform's 64 registers
    // drain into the packed ones as the pass goes (what is live beside it decides whether two waves fit a SIMD)
#pragma unroll
    for (int c = 0; c < 8; c++) {
      idct_column_quadwrap(v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c], level7);
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int i = r * 8 + c;
        // Q table of the subset: clamp to [0, 2^(P+4)), scale to 2^20; kept minus 2^19
        const int d = (min(max(v[i] + level_out, 0), rmax) << qshift) - (1 << 19);
        // (LUMA_LDS: the empty asm statements make the packed word exist HERE -- left to itself the compiler keeps the raw sample
        // of every block and packs where the words are used, a hundred registers later; profiles/r06/xt_kernels.txt)
        if (which == 0) {
          rA[i] = (unsigned)d & 0xfffffu;
          if (LUMA_LDS) asm volatile("" : "+v"(rA[i]));
        } else if (which == 1) {
          rA[i] |= ((unsigned)d & 0xfffu) << 20;
          const unsigned top = ((unsigned)(d >> 12) & 0xffu) << (8 * (i & 3)); // eight bits with the sign
          if ((i & 3) == 0) hB[i >> 2] = top;
          else hB[i >> 2] |= top;
          if (LUMA_LDS) asm volatile("" : "+v"(rA[i]), "+v"(hB[i >> 2]));
        } else if constexpr (!LUMA_LDS) {
          rB[i] = ((unsigned)d & 0xfffffu) | (((hB[i >> 2] >> (8 * (i & 3))) & 0xffu) << 20);
        } else {
          // The two-wave flavour (half-float output) finishes the residual chain HERE -- R transformation and the R2 table of the
          // subset, (clamp(rr, 0, 2^20 - 1) + 8) >> 4 (colortrafo/ycbcrtrafo.cpp:776-800) -- and keeps three 16-bit results per
          // pixel, 96 registers for the block instead of 128.  A result of 65536 (rr at the very top) is kept as 65535: with
          // an L table entry of at least -32768 the sum is then 32767 or more either way, beyond the largest half-float code
          // the clamp below lets through (31743) -- the integer flavour of the output would see the difference and keeps the
          // other kernel.
          const int d0 = ((int)(rA[i] << 12)) >> 12;
          const int d1 = (((int)(hB[i >> 2] << (24 - 8 * (i & 3))) >> 24) << 12) | (int)(rA[i] >> 20);
          int rr0, rr1, rr2;
          if (x.rtrafo_ycbcr) {
            const int qy = d0 + (1 << 19);
            rr0 = qy + (int)(((long long)d * L_CR_R + 4096) >> 13);
            rr1 = qy + (int)(((long long)d1 * -L_CB_G + (long long)d * -L_CR_G + 4096) >> 13);
            rr2 = qy + (int)(((long long)d1 * L_CB_B + 4096) >> 13);
          } else {
            rr0 = d0 + (1 << 19); rr1 = d1 + (1 << 19); rr2 = d + (1 << 19);
          }
          const unsigned v0 = (unsigned)min((min(max(rr0, 0), omax16) + 8) >> 4, 65535), v1 = (unsigned)min((min(max(rr1, 0), omax16) + 8) >> 4, 65535),
                         v2 = (unsigned)min((min(max(rr2, 0), omax16) + 8) >> 4, 65535);
          rA[i] = v0 | (v1 << 16);
          if ((i & 1) == 0) rB[i >> 1] = v2;
          else rB[i >> 1] |= v2 << 16;
          asm volatile("" : "+v"(rA[i]), "+v"(rB[i >> 1]));
        }
      }
      if (LUMA_LDS) __builtin_amdgcn_sched_barrier(0);
    }
  };
  residual_block(x.off_r[0], x.rq[0], 0);
  __builtin_amdgcn_sched_barrier(0);
  residual_block(x.off_r[1], x.rq[1], 1);
  __builtin_amdgcn_sched_barrier(0);
  residual_block(x.off_r[2], x.rq[2], 2);
  __builtin_amdgcn_sched_barrier(0);

  const bool active = X0 < a.width && Y0 < a.height;
  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * 6u;
  const int npx = min(8, a.width - X0);
  const int nln = active ? min(8, a.height - Y0) : 0;
  const bool fast_store = npx == 8;

  const int *cb_base = cplane[0] + (4 * by) * F420_CPITCH + 4 * bx;
  const int *cr_base = cplane[1] + (4 * by) * F420_CPITCH + 4 * bx;
  auto load6 = [](const int *p, int (&d)[6]) {
    const i32x4 mid = *reinterpret_cast<const i32x4 *>(p + 4);
    d[0] = p[3]; d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w; d[5] = p[8];
  };
  const int pinf = (x.out_max >> 1) - (x.out_max >> 6) - 1, minf = -pinf - 1;
  const unsigned pinf2 = (unsigned)pinf * 0x10001u, minf2 = ((unsigned)minf & 0xffffu) * 0x10001u;

  int cbT[6], cbC[6], cbB[6], crT[6], crC[6], crB[6];
  load6(cb_base, cbT); load6(cb_base + F420_CPITCH, cbC);
  load6(cr_base, crT); load6(cr_base + F420_CPITCH, crC);
#pragma unroll
  for (int m = 0; m < 4; m++) {
    load6(cb_base + (m + 2) * F420_CPITCH, cbB);
    load6(cr_base + (m + 2) * F420_CPITCH, crB);
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int l = 2 * m + half;
      int vb[6], vr[6];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const int rnd = ((j & 1) ^ half) ? 1 : 2;
        vb[j] = tap13(half ? cbB[j] : cbT[j], cbC[j], rnd);
        vr[j] = tap13(half ? crB[j] : crT[j], crC[j], rnd);
      }
      int ub[8], ur[8];
      auto hfilt = [](const int (&v)[6], int (&o)[8]) {
        o[7] = tap13(v[5], v[4], 1);
        o[6] = tap13(v[3], v[4], 2);
        o[5] = tap13(v[4], v[3], 1);
        o[4] = tap13(v[2], v[3], 2);
        o[3] = tap13(v[3], v[2], 1);
        o[2] = tap13(v[1], v[2], 2);
        o[1] = tap13(o[2], v[1], 1); // src[1] has already been overwritten by out[2]
        o[0] = tap13(v[0], v[1], 2);
      };
      hfilt(vb, ub);
      hfilt(vr, ur);
      int mm[24];
      u32x4 yl = u32x4{0, 0, 0, 0};
      if constexpr (LUMA_LDS) yl = luma_lines[l * F420_THREADS + tid]; // (written by this lane: no barrier)
#pragma unroll
      for (int xx = 0; xx < 8; xx++) {
        int yk;
        if constexpr (LUMA_LDS) {
          const unsigned ypair = xx < 2 ? yl.x : xx < 4 ? yl.y : xx < 6 ? yl.z : yl.w;
          yk = (int)((xx & 1) ? (ypair & 0xffff0000u) : (ypair << 16)) >> 3; // (sample * 16 + 2056) << 13, sign included
        } else
          yk = luma13(yv[LUMA_LDS ? 0 : l * 8 + xx]); // (level shift and rounding are inside: LUMA_FOLD_R2)
        const int lr = clamp255(mad24(ur[xx], L_CR_R, yk) >> 17);
        const int lg = clamp255(mad24(ur[xx], -L_CR_G, mad24(ub[xx], -L_CB_G, yk)) >> 17);
        const int lb = clamp255(mad24(ub[xx], L_CB_B, yk) >> 17);
        const int lv[3] = {ltab[lr], ltab[256 + lg], ltab[512 + lb]};
        // residual chain (colortrafo/ycbcrtrafo.cpp:750-829): the three 20-bit samples minus 2^19
        const int i = l * 8 + xx;
        if constexpr (LUMA_LDS) { // (done where the samples were packed: three 16-bit results)
          const unsigned p2 = rB[LUMA_LDS ? i >> 1 : 0];
          mm[3 * xx + 0] = lv[0] + (int)(rA[i] & 0xffffu);
          mm[3 * xx + 1] = lv[1] + (int)(rA[i] >> 16);
          mm[3 * xx + 2] = lv[2] + (int)((i & 1) ? p2 >> 16 : p2 & 0xffffu);
          continue;
        }
        const int d0 = ((int)(rA[i] << 12)) >> 12, d2 = ((int)(rB[LUMA_LDS ? 0 : i] << 12)) >> 12;
        const int d1 = (((int)(rB[LUMA_LDS ? 0 : i] << 4) >> 24) << 12) | (int)(rA[i] >> 20);
        int rr[3];
        if (x.rtrafo_ycbcr) {
          // (ry 8192 + rcb Lb + rcr Lr + 4096) >> 13 with ry = d0 + 2^19 a whole number of 8192ths: ry + ((d L + 4096) >> 13)
          const int qy = d0 + (1 << 19);
          rr[0] = qy + (int)(((long long)d2 * L_CR_R + 4096) >> 13);
          rr[1] = qy + (int)(((long long)d1 * -L_CB_G + (long long)d2 * -L_CR_G + 4096) >> 13);
          rr[2] = qy + (int)(((long long)d1 * L_CB_B + 4096) >> 13);
        } else {
          rr[0] = d0 + (1 << 19); rr[1] = d1 + (1 << 19); rr[2] = d2 + (1 << 19);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) mm[3 * xx + c] = lv[c] + ((min(max(rr[c], 0), omax16) + 8) >> 4); // R2 table of the subset
      }
      unsigned w[12];
#pragma unroll
      for (int i = 0; i < 12; i++) {
        if (x.is_float) {
          unsigned pk, sg;
          asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(pk) : "v"(mm[2 * i]), "v"(mm[2 * i + 1]));
          asm("v_pk_max_i16 %0, %1, %2" : "=v"(pk) : "v"(pk), "v"(minf2));
          asm("v_pk_min_i16 %0, %1, %2" : "=v"(pk) : "v"(pk), "v"(pinf2));
          asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(sg) : "v"(pk));
          w[i] = pk ^ (sg & 0x7fff7fffu);
        } else {
          w[i] = (unsigned)min(max(mm[2 * i], 0), x.out_max) | ((unsigned)min(max(mm[2 * i + 1], 0), x.out_max) << 16);
        }
      }
      if (l < nln) {
        uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
        if (fast_store) {
          store48<FXT_TEMPORAL != 0>(dst, w);
        } else {
          unsigned short *d16 = reinterpret_cast<unsigned short *>(dst);
#pragma unroll
          for (int k = 0; k < 24; k++)
            if (k < 3 * npx) d16[k] = (unsigned short)(w[k >> 1] >> ((k & 1) * 16));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) { cbT[j] = cbC[j]; cbC[j] = cbB[j]; crT[j] = crC[j]; crC[j] = crB[j]; }
  }
}

// ==============================================================================================
// fused 4:4:4 kernel (three components, no subsampling, YCbCr): ReconstructUnsampled,
// control/blockbitmaprequester.cpp:1013-1074
// ==============================================================================================
// One lane owns one block POSITION and transforms its Cb, Cr and Y blocks one after the other; the two
// chroma results are kept as packed int16 pairs (32 + 32 VGPRs) that the colour multiply-adds read half by half
// (v_mad_i32_i16 op_sel), the luma result stays in 64 VGPRs.  No barrier, no halo, LDS only for the coalesced
// block fetch.  FAST arithmetic only, and only when the host's range check bounds every chroma sample by
// 4 * range_max < 32768 (so the packing is exact); everything else takes the generic two-kernel path.
// Algorithmic bytes: 3 x 2 B in + 3 B out = 9 B/pixel.
template <int MINW, bool QDEV>
__global__ __launch_bounds__(F420_THREADS, MINW) void fused444_kernel(const Fused420Args a)
{
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position(blockIdx.x, a); // (tile order: see there)
  if (tp.frame < 0) return; // the launch is padded to whole groups of eight tile rows
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
  const int x0 = gbx0 + (lane >> 3);

  // all three planes have the same geometry (bw_y x bh_y blocks)
  auto fetch = [&](u32x4 (&rows)[8], int64_t plane_off) {
    const char *pbase = reinterpret_cast<const char *>(coef + plane_off) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  };

  unsigned cbp[32], crp[32];
  {
    u32x4 rows[8];
    int v[64];
    fetch(rows, a.off_cb);
    dequant_idct_sparse<!QDEV>(rows, frame_deltas<QDEV>(a, frame, 1), v);
#pragma unroll
    for (int i = 0; i < 32; i++) cbp[i] = pack_lo16_now(v[2 * i + 1], v[2 * i]);
    __builtin_amdgcn_sched_barrier(0); // keep the next component's loads from being hoisted above this transform (register pressure)
    fetch(rows, a.off_cr);
    dequant_idct_sparse<!QDEV>(rows, frame_deltas<QDEV>(a, frame, 2), v);
#pragma unroll
    for (int i = 0; i < 32; i++) crp[i] = pack_lo16_now(v[2 * i + 1], v[2 * i]);
    __builtin_amdgcn_sched_barrier(0);
  }
  int yv[64];
  {
    u32x4 rows[8];
    fetch(rows, a.off_y);
    const int X0 = gbx * 8, Y0 = gby * 8;
    if (X0 >= a.width || Y0 >= a.height) return;
    dequant_idct_sparse<!QDEV, true>(rows, frame_deltas<QDEV>(a, frame, 0), yv, 0, LUMA_FOLD_R2);
  }
  const int X0 = gbx * 8, Y0 = gby * 8;
  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * 3u;
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
  const bool fast_store = npx == 8;
#pragma unroll
  for (int l = 0; l < 8; l++) {
    if (l < nln) {
      int rr[8], gg[8], bb[8];
#pragma unroll
      for (int x = 0; x < 8; x++) {
        const int yk = luma13(yv[l * 8 + x]); // (level shift and rounding are inside: LUMA_FOLD_R2)
        const unsigned cb2 = cbp[(l * 8 + x) >> 1], cr2 = crp[(l * 8 + x) >> 1];
        if (x & 1) {
          rr[x] = mad16_hi(cr2, L_CR_R, yk);
          gg[x] = mad16_hi(cr2, -L_CR_G, mad16_hi(cb2, -L_CB_G, yk));
          bb[x] = mad16_hi(cb2, L_CB_B, yk);
        } else {
          rr[x] = mad16_lo(cr2, L_CR_R, yk);
          gg[x] = mad16_lo(cr2, -L_CR_G, mad16_lo(cb2, -L_CB_G, yk));
          bb[x] = mad16_lo(cb2, L_CB_B, yk);
        }
      }
      uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
      if (fast_store) {
        unsigned w[6];
        rgb_shift17_sat_pack(rr, gg, bb, w);
        store24_3x8<F444_TEMPORAL != 0>(dst, w);
        __builtin_amdgcn_sched_barrier(0); // one line at a time: do not interleave the lines' temporaries
      } else {
#pragma unroll
        for (int x = 0; x < 8; x++)
          if (x < npx) {
            dst[3 * x] = (uint8_t)clamp255(rr[x] >> 17); dst[3 * x + 1] = (uint8_t)clamp255(gg[x] >> 17); dst[3 * x + 2] = (uint8_t)clamp255(bb[x] >> 17);
          }
      }
    }
  }
}

// 12-bit flavour (SOF1, P = 12; 16-bit samples out, 6 B in + 6 B out per pixel): the same decomposition with the chroma blocks kept as
// 32-bit values (64 + 64 VGPRs: a 12-bit chroma sample times 16 does not fit 16 bits), two waves per SIMD.  The colour stage is
// fused420_kernel<.., 12>'s: (y' + 32776 + (c L >> 13)) >> 4 clamped to [0, 4095], exact under the host's range gates
// (use_fused444_12: the 12-bit 4:2:0 kernel's bounds; there is no filter between transform and colour stage here).
template <bool QDEV, bool N12>
__global__ __launch_bounds__(F420_THREADS, 2) void fused444_12_kernel(const Fused420Args a)
{
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position(blockIdx.x, a);
  if (tp.frame < 0) return;
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
  const int x0 = gbx0 + (lane >> 3);
  auto fetch = [&](u32x4 (&rows)[8], int64_t plane_off) {
    const char *pbase = reinterpret_cast<const char *>(coef + plane_off) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  };

  int cb[64], cr[64], yv[64];
  {
    u32x4 rows[8];
    fetch(rows, a.off_cb);
    dequant_idct_sparse<false>(rows, frame_deltas<QDEV>(a, frame, 1), cb);
    __builtin_amdgcn_sched_barrier(0); // (register pressure: one component at a time)
    fetch(rows, a.off_cr);
    dequant_idct_sparse<false>(rows, frame_deltas<QDEV>(a, frame, 2), cr);
    __builtin_amdgcn_sched_barrier(0);
    fetch(rows, a.off_y);
  
    if (gbx * 8 >= a.width || gby * 8 >= a.height) return;
    dequant_idct_sparse<false>(rows, frame_deltas<QDEV>(a, frame, 0), yv);
  }
  const int X0 = gbx * 8, Y0 = gby * 8;
  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * 6u;
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
  const bool fast_store = npx == 8;
  static_assert(L_CB_B % 4 == 0, "the blue product is taken at a quarter of the constant");
  auto c12 = [](int v) { return (unsigned)min(max(v, 0), 4095); };
#pragma unroll
  for (int l = 0; l < 8; l++) {
    if (l < nln) {
      int rr[8], gg[8], bb[8];
#pragma unroll
      for (int x = 0; x < 8; x++) colour12<N12>(yv[l * 8 + x], cb[l * 8 + x], cr[l * 8 + x], rr[x], gg[x], bb[x]);
      uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
      if (fast_store) {
        unsigned w[12]; // 48 bytes r0 g0 b0 r1 ... b7, 16-bit samples
#pragma unroll
        for (int x = 0; x < 8; x += 2) {
          w[3 * (x / 2) + 0] = c12(rr[x]) | (c12(gg[x]) << 16);
          w[3 * (x / 2) + 1] = c12(bb[x]) | (c12(rr[x + 1]) << 16);
          w[3 * (x / 2) + 2] = c12(gg[x + 1]) | (c12(bb[x + 1]) << 16);
        }
        store48(dst, w);
        __builtin_amdgcn_sched_barrier(0); // one line at a time
      } else {
        uint16_t *d16 = reinterpret_cast<uint16_t *>(dst);
#pragma unroll
        for (int x = 0; x < 8; x++)
          if (x < npx) {
            d16[3 * x] = (uint16_t)c12(rr[x]); d16[3 * x + 1] = (uint16_t)c12(gg[x]); d16[3 * x + 2] = (uint16_t)c12(bb[x]);
          }
      }
    }
  }
}

// ==============================================================================================
// single-component kernel (grey scale frames, and one component of any frame reconstructed without upsampling):
// ReconstructUnsampled with the identity transformation, control/blockbitmaprequester.cpp:1013-1074
// ==============================================================================================
// One lane, one block: fetch, dequantise, transform, COLOR_TO_INT (x + 8) >> 4 with the level shift the fast transform
// leaves out folded into the rounding constant, clamp, eight bytes per line.  Samples travel as packed int16 from the
// addition on (range check: |sample * 16| <= 4 * range_max < 2^15).  Algorithmic bytes: 2 B in + 1 B out per pixel.
// P = 12: 12-bit frames -- the same transform, 16-bit samples out (clamp 4095), samples as 32-bit values throughout.
template <bool QDEV, int P = 8>
__global__ __launch_bounds__(F420_THREADS, 4) void fused1_kernel(const Fused420Args a)
{
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const TilePos tp = tile_position<1>(blockIdx.x, a); // (tile order: see there)
  if (tp.frame < 0) return; // the launch is padded to whole groups of eight tile rows
  const int frame = tp.frame, ty = tp.ty, tx = tp.tx;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
  const int x0 = gbx0 + (lane >> 3);
  u32x4 rows[8];
  {
    const char *pbase = reinterpret_cast<const char *>(coef + a.off_y) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), a.bw_y - 1), y = min(gby0 + (m >> 1), a.bh_y - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * a.bw_y + x) * 128));
    });
  }
  const int X0 = gbx * 8, Y0 = gby * 8;
  if (X0 >= a.width || Y0 >= a.height) return;
  int v[64];
  dequant_idct_sparse<!QDEV && P == 8>(rows, frame_deltas<QDEV>(a, frame, 0), v);
  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * (P == 12 ? 2u : 1u);
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
  const bool fast_store = npx == 8;
  if (P == 12) {
    // COLOR_TO_INT with the level shift 2^11 << 4 the transform left out: (x + 32768 + 8) >> 4, clamped to [0, 4095]
#pragma unroll
    for (int l = 0; l < 8; l++) {
      if (l < nln) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const unsigned lo = (unsigned)min(max((v[l * 8 + 2 * i] + (32768 + 8)) >> 4, 0), 4095);
          const unsigned hi = (unsigned)min(max((v[l * 8 + 2 * i + 1] + (32768 + 8)) >> 4, 0), 4095);
          w[i] = lo | (hi << 16);
        }
        uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
        if (fast_store) {
          __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4_any *>(dst));
        } else {
          uint16_t *d16 = reinterpret_cast<uint16_t *>(dst);
#pragma unroll
          for (int x = 0; x < 8; x++)
            if (x < npx) d16[x] = (uint16_t)(w[x >> 1] >> (16 * (x & 1)));
        }
      }
    }
    return;
  }
#pragma unroll
  for (int l = 0; l < 8; l++) {
    if (l < nln) {
      unsigned b4[2]; // four clamped samples each: level shift, COLOR_TO_INT's rounding, shift, clamp
#pragma unroll
      for (int h = 0; h < 2; h++)
        b4[h] = ashr_sat_pack4<4>(v[l * 8 + 4 * h] + (2048 + 8), v[l * 8 + 4 * h + 1] + (2048 + 8), v[l * 8 + 4 * h + 2] + (2048 + 8), v[l * 8 + 4 * h + 3] + (2048 + 8));
      uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
      if (fast_store) {
        __builtin_nontemporal_store(u32x2{b4[0], b4[1]}, reinterpret_cast<u32x2_any *>(dst));
      } else {
#pragma unroll
        for (int x = 0; x < 8; x++)
          if (x < npx) dst[x] = (uint8_t)(b4[x >> 2] >> (8 * (x & 3)));
      }
    }
  }
}

// ==============================================================================================
// generic path, kernel 1: dequant + IDCT of every block of every component into int32 sample planes
// ==============================================================================================
// NARROW (FAST only, 8-bit frames whose samples times 16 -- level shift included -- fit 16 bits: the host checks): the
// sample planes hold int16, half the bytes written here and read back by the second kernel.
typedef short i16x8 __attribute__((ext_vector_type(8)));
template <bool FAST, bool QDEV, bool NARROW = false>
__global__ __launch_bounds__(256) void idct_planes_kernel(const GenericArgs a)
{
  static_assert(FAST || !NARROW, "int16 sample planes need the range check of the fast flavour");
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // grid.x covers 64-block groups of one (frame, plane): blockIdx.y = frame * nplanes + plane
  const int comp = blockIdx.y % a.nplanes, frame = blockIdx.y / a.nplanes;
  const int nblocks = a.bw[comp] * a.bh[comp];
  const int first = (blockIdx.x * 4 + wave) * 64;
  if (first >= nblocks) return;
  if (comp >= a.wide_first && comp < a.wide_first + a.wide_count) return; // idct_planes_wide_kernel's
  const int16_t *__restrict__ plane = a.coef + (int64_t)frame * a.coef_frame_stride + a.coef_off[comp];
  u32x4 rows[8];
  const int32_t *__restrict__ rowmap = a.rowmap ? a.rowmap + comp * a.rowmap_stride : nullptr;
  fetch_blocks(rows, stage_all[wave], lane, [&](int m) -> const u32x4 * {
    int n = min(first + (lane >> 3) + 8 * m, nblocks - 1);
    if (rowmap) { // the block row's coefficients come from another row (rows of zeros read row 0 and are zeroed below)
      const int gy = n / a.bw[comp];
      n += (max(rowmap[gy], 0) - gy) * a.bw[comp];
    }
    return reinterpret_cast<const u32x4 *>(plane + (int64_t)n * 64) + (lane & 7);
  });
  const int blk = first + lane;
  if (blk >= nblocks) return;
  int v[64];
  const int *q = frame_deltas<QDEV>(a, frame, comp);
  if (FAST) dequant_idct_sparse(rows, q, v, a.dcoff[comp]);
  else dequant_idct<false>(rows, q, v, a.dcoff[comp]);
  const int by = blk / a.bw[comp], bx = blk - by * a.bw[comp];
  if ((rowmap && rowmap[by] < 0) || (a.zero_from[comp] > 0 && by >= a.zero_from[comp])) {
#pragma unroll
    for (int i = 0; i < 64; i++) v[i] = 0;
  }
  const int pitch = a.bw[comp] * 8;
  if (NARROW) {
    short *dst = reinterpret_cast<short *>(a.samples) + (int64_t)frame * a.sample_frame_stride + a.sample_off[comp] + ((int64_t)by * 8) * pitch + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++)
      *reinterpret_cast<i16x8 *>(dst + (int64_t)r * pitch) = i16x8{(short)v[r * 8 + 0], (short)v[r * 8 + 1], (short)v[r * 8 + 2], (short)v[r * 8 + 3],
                                                                  (short)v[r * 8 + 4], (short)v[r * 8 + 5], (short)v[r * 8 + 6], (short)v[r * 8 + 7]};
    return;
  }
  int *dst = a.samples + (int64_t)frame * a.sample_frame_stride + a.sample_off[comp] + ((int64_t)by * 8) * pitch + bx * 8;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    i32x4 *d = reinterpret_cast<i32x4 *>(dst + (int64_t)r * pitch);
    d[0] = i32x4{v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3]};
    d[1] = i32x4{v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]};
  }
}

// ==============================================================================================
// generic path, kernel 1w: planes of frames whose precision (with hidden bits) exceeds 12 -- the residual frame of
// JPEG XT streams written with -rR n.  The reference transforms them with IDCT<4,QUAD> (codestream/tables.cpp:1876-
// 1891): 64-bit butterflies, while the dequantising products, the DC offset and the stored pass results stay LONG
// (dct/idct.cpp:238-259) -- and so does `(dptr[0 << 3] + dptr[4 << 3]) << FIX_BITS` in the second pass (:297-298),
// which wraps at 12 + 4 bits.  Coefficients are int32 (12 + 4 bits do not fit the 16-bit store).  One lane per block;
// this path is about correctness, not speed.
// ==============================================================================================
__device__ __forceinline__ void idct_1d_quad(const long long (&s)[8], long long (&o)[8], bool second_pass)
{
  const long long z1 = (s[2] + s[6]) * FIX9(0.541196100);
  const long long tmp2 = z1 + s[6] * -FIX9(1.847759065);
  const long long tmp3 = z1 + s[2] * FIX9(0.765366865);
  const long long tmp0 = second_pass ? (long long)shlw(addw((int)s[0], (int)s[4]), 9) : (s[0] + s[4]) * 512;
  const long long tmp1 = second_pass ? (long long)shlw(subw((int)s[0], (int)s[4]), 9) : (s[0] - s[4]) * 512;
  const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  const long long tz1 = s[7] + s[1], tz2 = s[5] + s[3], tz3 = s[7] + s[3], tz4 = s[5] + s[1];
  const long long z5 = (tz3 + tz4) * FIX9(1.175875602);
  const long long y1 = tz1 * -FIX9(0.899976223), y2 = tz2 * -FIX9(2.562915447);
  const long long y3 = tz3 * -FIX9(1.961570560) + z5, y4 = tz4 * -FIX9(0.390180644) + z5;
  const long long o0 = s[7] * FIX9(0.298631336) + y1 + y3;
  const long long o1 = s[5] * FIX9(2.053119869) + y2 + y4;
  const long long o2 = s[3] * FIX9(3.072711026) + y2 + y3;
  const long long o3 = s[1] * FIX9(1.501321110) + y1 + y4;
  o[0] = tmp10 + o3; o[7] = tmp10 - o3;
  o[1] = tmp11 + o2; o[6] = tmp11 - o2;
  o[2] = tmp12 + o1; o[5] = tmp12 - o1;
  o[3] = tmp13 + o0; o[4] = tmp13 - o0;
}

__global__ __launch_bounds__(256) void idct_planes_wide_kernel(const GenericArgs a)
{
  const int comp = a.wide_first + blockIdx.y % a.wide_count, frame = blockIdx.y / a.wide_count;
  const int nblocks = a.bw[comp] * a.bh[comp];
  const int blk = blockIdx.x * blockDim.x + threadIdx.x;
  if (blk >= nblocks) return;
  const int32_t *__restrict__ rowmap = a.rowmap ? a.rowmap + comp * a.rowmap_stride : nullptr;
  const int sblk = rowmap ? blk + (max(rowmap[blk / a.bw[comp]], 0) - blk / a.bw[comp]) * a.bw[comp] : blk;
  const bool zeros = (rowmap && rowmap[blk / a.bw[comp]] < 0) || (a.zero_from[comp] > 0 && blk / a.bw[comp] >= a.zero_from[comp]);
  const int32_t *__restrict__ src =
      reinterpret_cast<const int32_t *>(a.coef + (int64_t)frame * a.coef_frame_stride + a.coef_off[comp]) + (int64_t)sblk * 64;
  int tmp[64];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const i32x4 c0 = reinterpret_cast<const i32x4 *>(src + r * 8)[0], c1 = reinterpret_cast<const i32x4 *>(src + r * 8)[1];
    const int c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    long long s[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = (long long)(int)((unsigned)c[k] * (unsigned)a.q[comp][r * 8 + k]); // LONG product
    if (r == 0) s[0] = (long long)addw((int)s[0], a.dcoff[comp]);
    idct_1d_quad(s, o, false);
#pragma unroll
    for (int k = 0; k < 8; k++) tmp[r * 8 + k] = (int)((o[k] + 256) >> 9);
  }
  const int by = blk / a.bw[comp], bx = blk - by * a.bw[comp];
  const int pitch = a.bw[comp] * 8;
  int *dst = a.samples + (int64_t)frame * a.sample_frame_stride + a.sample_off[comp] + ((int64_t)by * 8) * pitch + bx * 8;
#pragma unroll
  for (int x = 0; x < 8; x++) {
    long long s[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = tmp[k * 8 + x];
    idct_1d_quad(s, o, true);
#pragma unroll
    for (int k = 0; k < 8; k++) dst[(int64_t)k * pitch + x] = zeros ? 0 : (int)((o[k] + 2048) >> 12);
  }
}

// ==============================================================================================
// generic path, kernel 1l: planes of a frame with info.coef_wide -- a damaged stream whose DC prediction (or point
// transform) left the 16-bit range.  The reference keeps LONG coefficients and transforms them with the frame's ordinary
// IDCT<0,LONG> (dct/idct.cpp:225-335): the statements of idct_planes_kernel's SAFE flavour in wrapping 32-bit
// arithmetic, on int32 coefficients.  One lane per block; rare, and about correctness only.
// ==============================================================================================
__global__ __launch_bounds__(256) void idct_planes_long_kernel(const GenericArgs a)
{
  const int comp = a.wide_first + blockIdx.y % a.wide_count, frame = blockIdx.y / a.wide_count;
  const int nblocks = a.bw[comp] * a.bh[comp];
  const int blk = blockIdx.x * blockDim.x + threadIdx.x;
  if (blk >= nblocks) return;
  const int32_t *__restrict__ rowmap = a.rowmap ? a.rowmap + comp * a.rowmap_stride : nullptr;
  const int sblk = rowmap ? blk + (max(rowmap[blk / a.bw[comp]], 0) - blk / a.bw[comp]) * a.bw[comp] : blk;
  const bool zeros = (rowmap && rowmap[blk / a.bw[comp]] < 0) || (a.zero_from[comp] > 0 && blk / a.bw[comp] >= a.zero_from[comp]);
  const int32_t *__restrict__ src =
      reinterpret_cast<const int32_t *>(a.coef + (int64_t)frame * a.coef_frame_stride + a.coef_off[comp]) + (int64_t)sblk * 64;
  int v[64];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const i32x4 c = reinterpret_cast<const i32x4 *>(src)[r];
    v[r * 4 + 0] = mulc<false>(c.x, a.q[comp][r * 4 + 0]);
    v[r * 4 + 1] = mulc<false>(c.y, a.q[comp][r * 4 + 1]);
    v[r * 4 + 2] = mulc<false>(c.z, a.q[comp][r * 4 + 2]);
    v[r * 4 + 3] = mulc<false>(c.w, a.q[comp][r * 4 + 3]);
  }
  v[0] = addw(v[0], a.dcoff[comp]);
#pragma unroll
  for (int r = 0; r < 8; r++)
    idct_1d<false, 9>(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]);
#pragma unroll
  for (int c = 0; c < 8; c++) idct_1d<false, 12>(v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c]);
  if (zeros) {
#pragma unroll
    for (int i = 0; i < 64; i++) v[i] = 0;
  }
  const int by = blk / a.bw[comp], bx = blk - by * a.bw[comp];
  const int pitch = a.bw[comp] * 8;
  int *dst = a.samples + (int64_t)frame * a.sample_frame_stride + a.sample_off[comp] + ((int64_t)by * 8) * pitch + bx * 8;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    i32x4 *d = reinterpret_cast<i32x4 *>(dst + (int64_t)r * pitch);
    d[0] = i32x4{v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3]};
    d[1] = i32x4{v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]};
  }
}

// ==============================================================================================
// generic path, kernel 2: one thread = one line of one 8-pixel output group; upsample every
// component (any factor 1..4) exactly like the reference's buffer code, transform, store.
// ==============================================================================================
// Line buffer of the reference for output line Y of the 8x8 block at X0 (upsampler.cpp:83-117):
// vertical core output for buffer columns j = 0..7, then the in-place horizontal core.
__device__ __forceinline__ int f8(int wa, int x, int wb, int y, int r)
{
  return (int)((unsigned)wa * (unsigned)x + (unsigned)wb * (unsigned)y + (unsigned)r) >> 3;
}

// (lo: first line the filter may read -- 0, or the start of the upsampler's buffered window for rectangle requests, where
// ch is the window's end)
template <class T>
__device__ __forceinline__ void upsample_line_any(const T *__restrict__ plane, int pitch, int cw, int ch, int sx, int sy, int X0, int Y,
                                               int (&o)[8], int lo = 0, int defcols = 0)
{
  const int y = Y / sy, ymod = Y - y * sy;
  const int cur = min(y, ch - 1), top = min(max(y - 1, lo), ch - 1), bot = min(cur + 1, ch - 1);
  const int x = (sx > 1) ? X0 / sx - 1 : X0; // chroma column of buffer entry 0
  const T *pc = plane + (int64_t)cur * pitch, *pt = plane + (int64_t)top * pitch, *pb = plane + (int64_t)bot * pitch;
  int v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    // line buffer of the reference: entry -1 and entry cw replicate the edge samples (upsamplerbase.cpp:322-323); behind
    // that the buffer still holds what the transform of the last block left there, which only a displaced read sees
    // (defcols: samples per line that the blocks of the plane cover)
    int col = min(max(x + j, 0), cw - 1);
    if (defcols && x + j > cw) col = min(x + j, defcols - 1);
    const int c = pc[col];
    const int odd = j & 1;
    int val = c;
    if (sy == 2) {
      val = ymod == 0 ? tap13(pt[col], c, odd ? 1 : 2) : tap13(pb[col], c, odd ? 2 : 1);
    } else if (sy == 3) {
      if (ymod == 0) val = tap13(pt[col], c, odd ? 1 : 2);
      else if (ymod == 2) val = tap13(pb[col], c, odd ? 2 : 1);
    } else if (sy == 4) {
      if (ymod == 0) val = f8(3, pt[col], 5, c, odd ? 3 : 4);
      else if (ymod == 1) val = f8(1, pt[col], 7, c, odd ? 4 : 3);
      else if (ymod == 2) val = f8(1, pb[col], 7, c, odd ? 3 : 4);
      else val = f8(3, pb[col], 5, c, odd ? 3 : 4);
    }
    v[j] = val;
  }
  // horizontal core, literally in place like the reference (target == v, src == v + 1):
  // the statement order below is the reference's, so an entry overwritten by an earlier output is
  // seen by later ones exactly as there
  if (sx == 2) { // upsampler.cpp:283-307
    v[7] = tap13(v[5], v[4], 1);
    v[6] = tap13(v[3], v[4], 2);
    v[5] = tap13(v[4], v[3], 1);
    v[4] = tap13(v[2], v[3], 2);
    v[3] = tap13(v[3], v[2], 1);
    const int s0 = v[1];
    v[2] = tap13(s0, v[2], 2);
    v[1] = tap13(v[2], s0, 1);
    v[0] = tap13(v[0], s0, 2);
  } else if (sx == 3) { // upsampler.cpp:313-361
    const int xmod = X0 % 3;
    if (xmod == 0) {
      v[7] = v[3];
      v[6] = tap13(v[2], v[3], 2);
      v[5] = tap13(v[3], v[2], 1);
      v[4] = v[2];
      v[3] = tap13(v[1], v[2], 2);
      v[2] = tap13(v[2], v[1], 1);
      v[0] = tap13(v[0], v[1], 2);
      // out[1] = src[0]: v[1] stays
    } else if (xmod == 1) {
      v[7] = tap13(v[4], v[3], 1);
      v[6] = v[3];
      v[5] = tap13(v[2], v[3], 2);
      v[4] = tap13(v[3], v[2], 1);
      v[3] = v[2];
      const int s0 = v[1];
      v[2] = tap13(s0, v[2], 2);
      v[1] = tap13(v[2], s0, 1);
      v[0] = s0;
    } else {
      v[7] = tap13(v[3], v[4], 2);
      v[6] = tap13(v[4], v[3], 1);
      v[5] = v[3];
      v[4] = tap13(v[2], v[3], 2);
      v[3] = tap13(v[3], v[2], 1);
      const int s0 = v[1]; // out[2] = src[1]: v[2] stays
      v[1] = tap13(s0, v[2], 2);
      v[0] = tap13(v[2], s0, 1);
    }
  } else if (sx == 4) { // upsampler.cpp:367-387
    v[7] = f8(3, v[3], 5, v[2], 1);
    v[6] = f8(1, v[3], 7, v[2], 2);
    v[5] = f8(1, v[1], 7, v[2], 1);
    v[4] = f8(3, v[1], 5, v[2], 2);
    const int s0 = v[1];
    v[3] = f8(3, v[2], 5, s0, 1);
    v[2] = f8(1, v[2], 7, s0, 2);
    v[1] = f8(1, v[0], 7, s0, 1);
    v[0] = f8(3, v[0], 5, s0, 2);
  }
#pragma unroll
  for (int j = 0; j < 8; j++) o[j] = v[j];
}

// Compile-time specialisations of the same arithmetic for the layouts that matter (1x1, 2x1, 1x2, 2x2); groups that lie
// inside the plane skip the index clamps, and 1x1 lines are two 16-byte loads.
template <int SX, int SY, class T>
__device__ __forceinline__ void upsample_line_t(const T *__restrict__ plane, int pitch, int cw, int ch, int X0, int Y, int (&o)[8])
{
  const int y = Y / SY, ymod = Y - y * SY;
  const int cur = min(y, ch - 1), top = min(max(y - 1, 0), ch - 1), bot = min(cur + 1, ch - 1);
  const int x = (SX > 1) ? X0 / SX - 1 : X0;
  const T *pc = plane + (int64_t)cur * pitch, *pv = plane + (int64_t)(ymod == 0 ? top : bot) * pitch;
  int v[8];
  if (SX == 1 && SY == 1) {
    if (x + 7 < cw) { // X0 is a multiple of 8 and the pitch a multiple of 8 samples: aligned to the eight samples
      if constexpr (sizeof(T) == 2) {
        const i16x8 a = *reinterpret_cast<const i16x8 *>(pc + x);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = a[j];
      } else {
        const i32x4 a = *reinterpret_cast<const i32x4 *>(pc + x), b = *reinterpret_cast<const i32x4 *>(pc + x + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = pc[min(x + j, cw - 1)];
    }
    return;
  }
  const bool inside = x >= 0 && x + 7 < cw;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int col = inside ? x + j : min(max(x + j, 0), cw - 1);
    const int c = pc[col];
    if (SY == 2) v[j] = tap13(pv[col], c, ((j & 1) != 0) == (ymod == 0) ? 1 : 2);
    else v[j] = c;
  }
  if (SX == 2) {
    o[7] = tap13(v[5], v[4], 1);
    o[6] = tap13(v[3], v[4], 2);
    o[5] = tap13(v[4], v[3], 1);
    o[4] = tap13(v[2], v[3], 2);
    o[3] = tap13(v[3], v[2], 1);
    o[2] = tap13(v[1], v[2], 2);
    o[1] = tap13(o[2], v[1], 1); // in-place aliasing: src[1] already holds out[2]
    o[0] = tap13(v[0], v[1], 2);
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = v[j];
  }
}

// Layout of a three-component frame as a template parameter of the generic kernels: luma 1x1 and both chroma
// components subsampled by (CSX, CSY) in {1,2} x {1,2}; LAYOUT_ANY = runtime factors (everything else).
constexpr int LAYOUT_ANY = 0;
constexpr int layout_id(int csx, int csy) { return csx * 4 + csy; }
template <int LAYOUT, bool NARROW = false>
__device__ __forceinline__ void upsample_plane_line(const GenericArgs &a, int p, int comp, int frame, int X0, int Y, int (&o)[8])
{
  using T = typename std::conditional<NARROW, short, int>::type; // (NARROW: int16 sample planes, same offsets in samples)
  const T *plane = reinterpret_cast<const T *>(a.samples) + (int64_t)frame * a.sample_frame_stride + a.sample_off[p];
  if constexpr (LAYOUT == LAYOUT_ANY) {
    if (a.request && (a.subx[p] > 1 || a.suby[p] > 1)) {
      // Upsampler::UpsampleRegion starts at the corner of the rectangle (upsampler.cpp:85-86), the colour transformer reads
      // the result at (x & 7, y & 7): displaced in the request's first row / column of blocks; the vertical filter stops
      // at the buffered window
      const int Xd = (X0 == (a.req_x0 & ~7)) ? a.req_x0 : X0;
      const int Yd = ((Y & ~7) == (a.req_y0 & ~7)) ? a.req_y0 + (Y & 7) : Y;
      upsample_line_any<T>(plane, a.bw[p] * 8, a.cw[p], a.wlimit[p], a.subx[p], a.suby[p], Xd, Yd, o, a.wstart[p], ((a.cw[p] + 7) >> 3) * 8);
    } else
      upsample_line_any<T>(plane, a.bw[p] * 8, a.cw[p], a.ch[p], a.subx[p], a.suby[p], X0, Y, o);
  } else {
    if (comp == 0) upsample_line_t<1, 1, T>(plane, a.bw[p] * 8, a.cw[p], a.ch[p], X0, Y, o);
    else upsample_line_t<LAYOUT / 4, LAYOUT % 4, T>(plane, a.bw[p] * 8, a.cw[p], a.ch[p], X0, Y, o);
  }
}

// Exact (reference LONG / QUAD) colour stage for any precision: L matrix at FIX_BITS = 13 on samples with
// COLOR_BITS = 4 fractional bits, ycbcrtrafo.cpp:842-856; not clamped here.
__device__ __forceinline__ void ycc_to_rgb_wide(int y, int cb, int cr, int dcshift, long long &r, long long &g, long long &b)
{
  const long long yy = (long long)y * 8192 + 65536;
  const long long cbl = (long long)cb - dcshift, crl = (long long)cr - dcshift;
  r = (yy + crl * L_CR_R) >> 17;
  g = (yy - cbl * L_CB_G - crl * L_CR_G) >> 17;
  b = (yy + cbl * L_CB_B) >> 17;
}
__device__ __forceinline__ long long clampll(long long v, long long hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

template <bool FAST, int LAYOUT, bool NARROW = false>
__global__ __launch_bounds__(256) void upsample_color_kernel(const GenericArgs a)
{
  const int groups = (a.width + 7) >> 3;
  const int gxi = blockIdx.x * blockDim.x + threadIdx.x;
  const int Y = blockIdx.y + a.y_base;
  const int frame = blockIdx.z;
  if (gxi >= groups) return;
  const int X0 = gxi * 8;
  int s[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; c++) {
    if (c < a.ncomp) upsample_plane_line<LAYOUT, NARROW>(a, c, c, frame, X0, Y, s[c]);
  }
  uint8_t *dst = a.out + (int64_t)frame * a.out_frame_stride + (int64_t)Y * a.row_stride + (int64_t)X0 * a.ncomp * a.sample_bytes;
  const int npx = min(8, a.width - X0);
#pragma unroll
  for (int x = 0; x < 8; x++) {
    if (x >= npx) break;
    if (a.sample_bytes == 1) {
      if (a.ycbcr && a.ncomp == 3) {
        int r, g, b;
        ycc_to_rgb<FAST>(s[0][x], s[1][x], s[2][x], r, g, b);
        dst[3 * x] = (uint8_t)r; dst[3 * x + 1] = (uint8_t)g; dst[3 * x + 2] = (uint8_t)b;
      } else {
#pragma unroll
        for (int c = 0; c < MAXC; c++)
          if (c < a.ncomp) dst[a.ncomp * x + c] = (uint8_t)color_to_int<FAST>(s[c][x]);
      }
    } else { // precision 12: 16-bit samples, clamp to 2^P - 1 (ycbcrtrafo.cpp:921-936 with m_lOutMax = 4095)
      uint16_t *d16 = reinterpret_cast<uint16_t *>(dst);
      if (a.ycbcr && a.ncomp == 3) {
        long long r, g, b;
        ycc_to_rgb_wide(s[0][x], s[1][x], s[2][x], a.dcshift, r, g, b);
        d16[3 * x] = (uint16_t)clampll(r, a.maxval); d16[3 * x + 1] = (uint16_t)clampll(g, a.maxval); d16[3 * x + 2] = (uint16_t)clampll(b, a.maxval);
      } else {
#pragma unroll
        for (int c = 0; c < MAXC; c++)
          if (c < a.ncomp) d16[a.ncomp * x + c] = (uint16_t)clampll(((long long)s[c][x] + 8) >> 4, a.maxval);
      }
    }
  }
}

// ==============================================================================================
// fused tile kernel: ANY sampling layout (factors 1..4 per component and direction, one to four components, 8 or 12 bit)
// in one pass -- what the generic pair does through HBM, through LDS.
// ==============================================================================================
// A workgroup owns a tile of tile_w x tile_h pixels (whole MCUs).  Phase A: for every component the blocks that cover the
// tile's samples plus, for subsampled components, one sample of halo on each side (the filters of upsampling/upsampler.cpp
// reach one sample to the left / right and one line up / down: a block row or column of neighbours is transformed again by
// the neighbouring tile -- recomputed, not exchanged, like fused411_kernel's column halo) are fetched with the coalesced
// 64-block fetch, transformed by one lane each and written to the component's sample plane in LDS (int16 when the host's
// range check allows, NARROW).  Phase B: one thread per line of one 8-pixel group runs the reference's buffer arithmetic
// (upsample_line_any, literally the generic second kernel's) on the LDS planes, the colour transformation, and stores the
// group's 8 x ncomp samples as whole dwords.  Nothing but coefficients in and samples out touches HBM: traffic = algorithmic
// bytes + the halo blocks.  Samples carry their level shift (dcoff), as in the generic pair, so SAFE arithmetic and 12-bit
// frames share the code.
// n / s for the subsampling factors 1..4 (uniform s): shifts and one constant division instead of the ~35 instructions of a
// division by a run-time value
__device__ __forceinline__ int div_small(int n, int s) { return s == 1 ? n : s == 2 ? n >> 1 : s == 4 ? n >> 2 : n / 3; }
// n / w for 0 <= n < 2^15 and a small uniform w (blocks per line of a tile plane, groups per line of a tile), rw = 1.0f / w:
// (n + 0.5) * rw lies at least 0.5 / w away from every integer, far more than the product's rounding error
__device__ __forceinline__ int div_recip(int n, float rw) { return (int)(((float)n + 0.5f) * rw); }
// one of four uniform values by a uniform index: scalar selects, no register indexing
__device__ __forceinline__ int pick4(const int (&v)[MAXC], int c) { return c == 0 ? v[0] : c == 1 ? v[1] : c == 2 ? v[2] : v[3]; }

// A plane line in LDS is preceded by TILE_PAD samples (and one such piece follows the plane): the column left of a line's first
// sample and the column right of its last one exist in memory, image-edge tiles fill them with the replicated edge sample
// (upsamplerbase.cpp:322-323), and phase B reads its columns without clamping any of them.
constexpr int TILE_PAD = 8;
#ifndef TILE_MINW
#define TILE_MINW 3
#endif

// (wn n + wc c + r) >> sh: vertical filter with the phase's weights as data
template <bool FAST>
__device__ __forceinline__ int tile_vmix(int n, int c, int wn, int wc, int r, int sh)
{
  if (FAST) return mad24v(c, wc, mad24v(n, wn, r)) >> sh; // |sample * 16| < 2^20 (range check): nothing wraps, 24-bit operands
  return (int)((unsigned)wn * (unsigned)n + (unsigned)wc * (unsigned)c + (unsigned)r) >> sh;
}
template <bool FAST>
__device__ __forceinline__ int tile_f8(int wa, int x, int wb, int y, int r)
{
  if (FAST) return mad24(x, wa, mad24(y, wb, r)) >> 3;
  return f8(wa, x, wb, y, r);
}

template <class T>
__device__ __forceinline__ void tile_load8(const T *p, int (&o)[8])
{
  if constexpr (sizeof(T) == 2) {
    const i16x8 v = *reinterpret_cast<const i16x8 *>(p);
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = v[j];
  } else {
    const i32x4 v0 = *reinterpret_cast<const i32x4 *>(p), v1 = *reinterpret_cast<const i32x4 *>(p + 4);
    o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
  }
}

// What phase B needs to know about a component: uniform values, computed once per workgroup.
struct TileComp {
  int sx, sy;             // subsampling factors
  int rx, ry;             // 2^16 / s + 1: n / s = (n * r) >> 16 for the n of a tile (< 2^12)
  int pitch;              // samples per plane line in LDS
  int lines;              // lines of the image the plane holds: the vertical filter clamps to lines - 1
  int base;               // index of sample (0, 0) of the plane
  int xoff, yoff;         // pixel coordinates of that sample
  int sh;                 // shift of the vertical filter: 2, or 3 for sy = 4
  unsigned long long vt;  // the vertical filter's four phases, 16 bits each (tile_vphases)
};
// The vertical cores of upsampling/upsampler.cpp:136-271 as data: in this kernel the lanes of a wave sit on different lines, so
// the phase (Y mod sy) differs between them -- a branch per phase would make the wave walk every one of them.  Each phase
// mixes the current line c with ONE neighbour n (above for the upper phases, below for the lower ones) as
// (wn n + wc c + r) >> sh, wc = 2^sh - wn, r alternating between even and odd columns:
//   sy 2: (n + 3 c + r) >> 2, r = 2,1 (even, odd column) above / 1,2 below
//   sy 3: the same for phases 0 and 2, phase 1 is the line itself (weights 0 and 4, r = 0)
//   sy 4: phases 0,3: (3 n + 5 c + r) >> 3; 1,2: (n + 7 c + r) >> 3; r = 4,3 except phase 1: 3,4
// An entry: wn | r_even << 2 | r_odd << 5 | d << 8, the neighbour is line y - 1 + d (0: above, 2: below, 1: the line itself).
__device__ __forceinline__ constexpr unsigned long long tile_vphase(int wn, int re, int ro, int d) { return (unsigned long long)(wn | re << 2 | ro << 5 | d << 8); }
__device__ __forceinline__ unsigned long long tile_vphases(int sy)
{
  constexpr unsigned long long up = tile_vphase(1, 2, 1, 0), down = tile_vphase(1, 1, 2, 2), self = tile_vphase(0, 0, 0, 1);
  constexpr unsigned long long v2 = up | down << 16, v3 = up | self << 16 | down << 32;
  constexpr unsigned long long v4 = tile_vphase(3, 4, 3, 0) | tile_vphase(1, 3, 4, 0) << 16 | tile_vphase(1, 4, 3, 2) << 32 | tile_vphase(3, 4, 3, 2) << 48;
  return sy == 2 ? v2 : sy == 3 ? v3 : sy == 4 ? v4 : self;
}

// Eight output samples of one component on pixel line Y from pixel column X0 on (X0 a multiple of 8): upsample_line_any's
// arithmetic on the LDS plane, without a branch that differs between lanes.
template <bool FAST, class T>
__device__ __forceinline__ void tile_plane_line(const T *planes, const TileComp &k, int X0, int Y, int (&o)[8])
{
  const int Xl = X0 - k.xoff, Yl = Y - k.yoff;
  if (k.sx == 1 && k.sy == 1) { // the full-resolution components: eight samples, one load
    tile_load8(planes + k.base + mad24(min(Yl, k.lines - 1), k.pitch, Xl), o);
    return;
  }
  const int y = k.sy == 1 ? Yl : (int)((unsigned)__mul24(Yl, k.ry) >> 16);
  const int cur = min(y, k.lines - 1);
  const T *pc = planes + k.base + __mul24(cur, k.pitch), *pn = pc;
  int wn = 0, wc = 0, r_even = 0, r_odd = 0;
  if (k.sy > 1) { // (uniform)
    const int ymod = mad24(y, -k.sy, Yl);
    const unsigned e = (unsigned)(k.vt >> (ymod << 4));
    wn = e & 3; wc = (1 << k.sh) - wn; r_even = (e >> 2) & 7; r_odd = (e >> 5) & 7;
    const int other = min(max(y - 1 + (int)((e >> 8) & 3), 0), k.lines - 1);
    pn = planes + k.base + __mul24(other, k.pitch);
  }
  if (k.sx == 1) { // vertical filter only: two aligned loads
    int c[8], n[8];
    tile_load8(pc + Xl, c);
    tile_load8(pn + Xl, n);
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = tile_vmix<FAST>(n[j], c[j], wn, wc, (j & 1) ? r_odd : r_even, k.sh);
    return;
  }
  const int xq = (int)((unsigned)__mul24(Xl, k.rx) >> 16);
  pc += xq - 1; pn += xq - 1; // buffer entry 0 of the reference's line buffer: the column left of the group's first one
  auto in = [&](int j) -> int { // buffer entry j after the vertical core
    const int c = pc[j];
    return k.sy > 1 ? tile_vmix<FAST>(pn[j], c, wn, wc, (j & 1) ? r_odd : r_even, k.sh) : c;
  };
  // horizontal cores: upsample_line_any's statements (the in-place order of the reference), on the entries each one reads
  if (k.sx == 2) {
    const int v0 = in(0), v1 = in(1), v2 = in(2), v3 = in(3), v4 = in(4), v5 = in(5);
    o[7] = tap13(v5, v4, 1);
    o[6] = tap13(v3, v4, 2);
    o[5] = tap13(v4, v3, 1);
    o[4] = tap13(v2, v3, 2);
    o[3] = tap13(v3, v2, 1);
    o[2] = tap13(v1, v2, 2);
    o[1] = tap13(o[2], v1, 1); // (in-place aliasing of the reference: src[1] already holds out[2])
    o[0] = tap13(v0, v1, 2);
  } else if (k.sx == 3) {
    // the group starts in one of three column phases (X0 mod 3), which differ between the lanes: the ten output samples from
    // pixel 3 xq on -- u[t] = a tap of two neighbours or, in the middle of a column, the sample itself -- and every lane
    // takes eight of them from its phase on
    const int xmod = Xl - 3 * xq;
    const int v0 = in(0), v1 = in(1), v2 = in(2), v3 = in(3), v4 = in(4);
    const int u0 = tap13(v0, v1, 2), u2 = tap13(v2, v1, 1), u3 = tap13(v1, v2, 2), u5 = tap13(v3, v2, 1), u6 = tap13(v2, v3, 2), u8 = tap13(v4, v3, 1),
              u9 = tap13(v3, v4, 2);
    const int u[10] = {u0, v1, u2, u3, v2, u5, u6, v3, u8, u9};
    const bool m1 = xmod == 1, m2 = xmod == 2;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = m2 ? u[j + 2] : m1 ? u[j + 1] : u[j];
    // (in-place aliasing of the reference in phase 1: its second output reads src[1] when that already holds out[2])
    o[1] = m1 ? tap13(u3, v1, 1) : o[1];
  } else {
    const int v0 = in(0), v1 = in(1), v2 = in(2), v3 = in(3);
    o[7] = tile_f8<FAST>(3, v3, 5, v2, 1);
    o[6] = tile_f8<FAST>(1, v3, 7, v2, 2);
    o[5] = tile_f8<FAST>(1, v1, 7, v2, 1);
    o[4] = tile_f8<FAST>(3, v1, 5, v2, 2);
    o[3] = tile_f8<FAST>(3, v2, 5, v1, 1);
    o[2] = tile_f8<FAST>(1, v2, 7, v1, 2);
    o[1] = tile_f8<FAST>(1, v0, 7, v1, 1);
    o[0] = tile_f8<FAST>(3, v0, 5, v1, 2);
  }
}

#ifndef TILE_STRIP
#define TILE_STRIP 1
#endif
template <bool FAST, bool NARROW>
__global__ __launch_bounds__(256, TILE_MINW) void fused_tile_kernel(const GenericArgs a)
{
  using T = typename std::conditional<NARROW, short, int>::type;
  extern __shared__ __attribute__((aligned(16))) uint8_t tile_lds[];
  u32x4 *stage_all = reinterpret_cast<u32x4 *>(tile_lds); // 4 waves x 128 x 16 bytes
  T *planes = reinterpret_cast<T *>(tile_lds + 4 * 128 * sizeof(u32x4));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all + wave * 128;

  const unsigned logical = tile_of_workgroup(blockIdx.x, (unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames); // (tile order: see there)
  if (logical == ~0u) return; // the launch is padded to whole groups of eight tile rows
  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int frame = logical / tiles_per_frame;
  const int tile = logical - frame * tiles_per_frame;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int px0 = tx * a.tile_w, py0 = ty * a.tile_h;
  const int px1 = min(px0 + a.tile_w, a.width) - 1, py1 = min(py0 + a.tile_h, a.height) - 1;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  // geometry of the component planes in LDS: first block column / row, blocks held; base = index of sample (0, 0), which is
  // sample (bx0 * 8, by0 * 8) of the component; a line takes nbx * 8 samples and TILE_PAD more in front of it (needed by the
  // horizontally subsampled components only, but a pitch of 64 or 128 samples would send the block rows of phase A's
  // stores to the same banks: CMYK 335 -> 296 Gpixel/s without the pad)
  int bx0[MAXC], by0[MAXC], nbx[MAXC], nby[MAXC], base[MAXC], nchunk[MAXC], ppitch[MAXC];
  int off = 0, chunks = 0;
  bool edge = false;
#pragma unroll
  for (int c = 0; c < MAXC; c++) {
    bx0[c] = by0[c] = nbx[c] = nby[c] = base[c] = nchunk[c] = ppitch[c] = 0;
    if (c < a.ncomp) {
      const int sx = a.subx[c], sy = a.suby[c];
      const int cx0 = max(div_small(px0, sx) - (sx > 1 ? 1 : 0), 0), cx1 = min(div_small(px1, sx) + (sx > 1 ? 1 : 0), a.cw[c] - 1);
      const int cy0 = max(div_small(py0, sy) - (sy > 1 ? 1 : 0), 0), cy1 = min(div_small(py1, sy) + (sy > 1 ? 1 : 0), a.ch[c] - 1);
      bx0[c] = cx0 >> 3; by0[c] = cy0 >> 3;
      nbx[c] = (cx1 >> 3) - bx0[c] + 1; nby[c] = (cy1 >> 3) - by0[c] + 1;
      const int pad = TILE_PAD;
      ppitch[c] = nbx[c] * 8 + pad;
      base[c] = off + pad;
      off += nby[c] * 8 * ppitch[c] + pad;
      nchunk[c] = (nbx[c] * nby[c] + 63) >> 6;
      chunks += nchunk[c];
      if (sx > 1) edge |= (bx0[c] == 0) | (a.cw[c] - bx0[c] * 8 <= nbx[c] * 8);
    }
  }
  // ------------------------------------------------------------------ phase A: blocks -> sample planes in LDS
  // the (component, 64 blocks) chunks of the tile go to the four waves in turn: every wave transforms a quarter of the
  // tile's blocks whatever the components' sizes are.  One copy of the code for all components (the component is a uniform
  // run-time value here: phase A alone would otherwise be four times three transforms long).
#ifndef TILE_SKIP_A // (A-B measurements: phase B alone)
  for (int chunk = wave; chunk < chunks; chunk += 4) {
    int c = 0, k = chunk;
#pragma unroll
    for (int i = 0; i + 1 < MAXC; i++)
      if (c == i && k >= nchunk[i]) { k -= nchunk[i]; c = i + 1; }
    const int w = pick4(nbx, c), nblk = w * pick4(nby, c), pitch = pick4(ppitch, c), b0 = k * 64;
    const float rw = 1.0f / (float)w;
    const int bw = c == 0 ? a.bw[0] : c == 1 ? a.bw[1] : c == 2 ? a.bw[2] : a.bw[3];
    const int64_t coff = c == 0 ? a.coef_off[0] : c == 1 ? a.coef_off[1] : c == 2 ? a.coef_off[2] : a.coef_off[3];
    const int dcoff = c == 0 ? a.dcoff[0] : c == 1 ? a.dcoff[1] : c == 2 ? a.dcoff[2] : a.dcoff[3];
    const int *__restrict__ q = a.q[c];
    const char *__restrict__ first = reinterpret_cast<const char *>(coef + coff) + (int64_t)(pick4(by0, c) * bw + pick4(bx0, c)) * 128;
    u32x4 rows[8];
    auto chunkptr = [&](int m) -> const u32x4 * {
      const int n = min(b0 + (lane >> 3) + 8 * m, nblk - 1);
      const int y = div_recip(n, rw), x = mad24(y, -w, n);
      return reinterpret_cast<const u32x4 *>(first + (((unsigned)mad24(y, bw, x) << 7) | ((unsigned)(lane & 7) << 4)));
    };
    fetch_blocks(rows, stage, lane, chunkptr);
    const int blk = b0 + lane;
    if (blk < nblk) {
      int v[64];
      if (FAST) dequant_idct_sparse<NARROW>(rows, q, v, dcoff);
      else dequant_idct<false>(rows, q, v, dcoff);
      const int y = div_recip(blk, rw), x = mad24(y, -w, blk);
      T *dst = planes + pick4(base, c) + (y * 8) * pitch + x * 8;
#pragma unroll
      for (int r = 0; r < 8; r++) {
        if constexpr (NARROW) {
          *reinterpret_cast<i16x8 *>(dst + r * pitch) = i16x8{(short)v[r * 8 + 0], (short)v[r * 8 + 1], (short)v[r * 8 + 2], (short)v[r * 8 + 3],
                                                              (short)v[r * 8 + 4], (short)v[r * 8 + 5], (short)v[r * 8 + 6], (short)v[r * 8 + 7]};
        } else {
          i32x4 *d = reinterpret_cast<i32x4 *>(dst + r * pitch);
          d[0] = i32x4{v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3]};
          d[1] = i32x4{v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]};
        }
      }
    }
  }
#endif
  __syncthreads();
  if (edge) { // (uniform) tiles on the left / right image edge: the replicated columns of the horizontally subsampled planes
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      if (c < a.ncomp && a.subx[c] > 1) {
        const int pitch = ppitch[c], last = a.cw[c] - bx0[c] * 8 - 1; // last column of the image, plane-relative
        for (int r = tid; r < nby[c] * 8; r += 256) {
          T *line = planes + base[c] + r * pitch;
          if (bx0[c] == 0) line[-1] = line[0];
          if (last < nbx[c] * 8) {
            const T v = line[last];
            for (int k = last + 1; k <= nbx[c] * 8; k++) line[k] = v;
          }
        }
      }
    }
    __syncthreads();
  }
  // ------------------------------------------------------------------ phase B: lines of 8-pixel groups
#ifdef TILE_SKIP_B // (A-B measurements: phase A alone)
  if (tid == 0) a.out[(int64_t)frame * a.out_frame_stride + (int64_t)py0 * a.row_stride + px0] = (uint8_t)planes[base[0]];
  return;
#endif
  const int groups = (px1 - px0 + 8) >> 3, lines = py1 - py0 + 1;
  const int sb = a.sample_bytes;
  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const float rgroups = 1.0f / (float)groups;
  TileComp comp[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; c++) {
    const int sx = c < a.ncomp ? a.subx[c] : 1, sy = c < a.ncomp ? a.suby[c] : 1;
    comp[c].sx = sx; comp[c].sy = sy;
    comp[c].rx = 65536 / sx + 1; comp[c].ry = 65536 / sy + 1;
    comp[c].pitch = ppitch[c];
    comp[c].lines = min(a.ch[c] - by0[c] * 8, nby[c] * 8);
    comp[c].base = base[c];
    comp[c].xoff = bx0[c] * 8 * sx; comp[c].yoff = by0[c] * 8 * sy;
    comp[c].sh = sy == 4 ? 3 : 2;
    comp[c].vt = tile_vphases(sy);
  }
  // one instance per component count: the loops over components and the sample packing have static shapes
  auto phase_b = [&](auto NCc) {
    constexpr int NC = decltype(NCc)::value;
    // A lane owns TILE_STRIP consecutive lines of one 8-pixel group (what depends on the column alone -- the group's place in every
    // plane, the horizontal phase -- is then worked out once per strip).  The product is 1: strips of 2 and 4 lines measured 2-6 %
    // SLOWER on every layout in round 6 (the stores of a wave spread over more lines; profiles/r06/tile_strips.txt); the switch
    // stays for A-B builds.
    const int strips = (lines + TILE_STRIP - 1) / TILE_STRIP;
    for (int it = tid; it < groups * strips; it += 256) {
      const int st = div_recip(it, rgroups), g = mad24(st, -groups, it);
      const int X0 = px0 + 8 * g;
#pragma unroll
     for (int sub = 0; sub < TILE_STRIP; sub++) {
      const int ly = st * TILE_STRIP + sub, Y = py0 + ly;
      if (TILE_STRIP > 1 && ly >= lines) break;
      int s[NC][8];
#pragma unroll
      for (int c = 0; c < NC; c++) tile_plane_line<FAST, T>(planes, comp[c], X0, Y, s[c]);
      // the group's samples as the dwords of the output line: 2 NC of them (8-bit samples) or 4 NC (16-bit samples)
      unsigned w[4 * NC];
      if (sb == 1) {
        if (FAST && NC == 3 && a.ycbcr) {
          // the colour stage of the fused 4:2:0 kernel: constants carry level shift and rounding, two samples per clamp
          const int KR = 65536 - 2048 * L_CR_R, KG = 65536 + 2048 * (L_CB_G + L_CR_G), KB = 65536 - 2048 * L_CB_B;
          int rr[8], gg[8], bb[8];
#pragma unroll
          for (int x = 0; x < 8; x++) {
            const int y13 = s[0][x] << 13, cb = s[NC > 1 ? 1 : 0][x], cr = s[NC > 2 ? 2 : 0][x];
            rr[x] = mad24(cr, L_CR_R, y13 + KR);
            gg[x] = mad24(cb, -L_CB_G, mad24(cr, -L_CR_G, y13 + KG));
            bb[x] = mad24(cb, L_CB_B, y13 + KB);
          }
          unsigned w6[6];
          rgb_shift17_sat_pack(rr, gg, bb, w6);
#pragma unroll
          for (int i = 0; i < 6; i++) w[i % (4 * NC)] = w6[i];
        } else {
          int px[8 * NC]; // sample e of the group: pixel e / NC, component e % NC
#pragma unroll
          for (int x = 0; x < 8; x++) {
            if (NC == 3 && a.ycbcr) { // (SAFE arithmetic here)
              int r, gg, b;
              ycc_to_rgb<FAST>(s[0][x], s[NC > 1 ? 1 : 0][x], s[NC > 2 ? 2 : 0][x], r, gg, b);
              px[x * NC] = r; px[x * NC + (NC > 1 ? 1 : 0)] = gg; px[x * NC + (NC > 2 ? 2 : 0)] = b;
            } else {
#pragma unroll
              for (int c = 0; c < NC; c++) px[x * NC + c] = FAST ? s[c][x] + 8 : color_to_int<false>(s[c][x]);
            }
          }
#pragma unroll
          for (int k = 0; k < 2 * NC; k++) {
            if (FAST) w[k] = ashr_sat_pack4<4>(px[4 * k], px[4 * k + 1], px[4 * k + 2], px[4 * k + 3]); // COLOR_TO_INT + clamp (identity)
            else w[k] = ashr_sat_pack4<0>(px[4 * k], px[4 * k + 1], px[4 * k + 2], px[4 * k + 3]);
          }
        }
      } else {
        unsigned px[8 * NC];
#pragma unroll
        for (int x = 0; x < 8; x++) {
          if (NC == 3 && a.ycbcr) {
            if (FAST) {
              // the reference's 64-bit sum (y 8192 + c' L + 65536) >> 17 with c' = c - level shift in 32 bits: c' L = q 2^13 + r,
              // 0 <= r < 2^13, gives ((y + 8 + q) 2^13 + r) >> 17 = (y + 8 + q) >> 4 exactly (fused420_kernel<12> has the proof;
              // FAST bounds the chroma samples times 16 by 181 200, see use_fused420_12 in capi.cpp: the products stay inside
              // 32 bits, the operands inside 24)
              const int yk = s[0][x] + 8, cb = s[NC > 1 ? 1 : 0][x] - a.dcshift, cr = s[NC > 2 ? 2 : 0][x] - a.dcshift;
              const int r = (yk + (__mul24(cr, L_CR_R) >> 13)) >> 4;
              const int gg = (yk + (mad24(cr, -L_CR_G, __mul24(cb, -L_CB_G)) >> 13)) >> 4;
              const int b = (yk + (__mul24(cb, L_CB_B / 4) >> 11)) >> 4;
              px[x * NC] = (unsigned)min(max(r, 0), a.maxval); px[x * NC + (NC > 1 ? 1 : 0)] = (unsigned)min(max(gg, 0), a.maxval);
              px[x * NC + (NC > 2 ? 2 : 0)] = (unsigned)min(max(b, 0), a.maxval);
            } else {
              long long r, gg, b;
              ycc_to_rgb_wide(s[0][x], s[NC > 1 ? 1 : 0][x], s[NC > 2 ? 2 : 0][x], a.dcshift, r, gg, b);
              px[x * NC] = (unsigned)clampll(r, a.maxval); px[x * NC + (NC > 1 ? 1 : 0)] = (unsigned)clampll(gg, a.maxval);
              px[x * NC + (NC > 2 ? 2 : 0)] = (unsigned)clampll(b, a.maxval);
            }
          } else {
#pragma unroll
            for (int c = 0; c < NC; c++)
              px[x * NC + c] = FAST ? (unsigned)min(max((s[c][x] + 8) >> 4, 0), a.maxval) : (unsigned)clampll(((long long)s[c][x] + 8) >> 4, a.maxval);
          }
        }
#pragma unroll
        for (int k = 0; k < 4 * NC; k++) w[k] = px[2 * k] | (px[2 * k + 1] << 16);
      }
      uint8_t *dst = out_frame + (int64_t)Y * a.row_stride + (int64_t)X0 * (NC * sb);
      const int npx = min(8, a.width - X0);
      if (npx == 8) { // 8 * NC * sb bytes = a whole number of 8-byte pieces (at any address: u32x2_any)
        u32x2_any *d2 = reinterpret_cast<u32x2_any *>(dst);
        if (sb == 1) {
#pragma unroll
          for (int k = 0; k < NC; k++) d2[k] = u32x2{w[2 * k], w[2 * k + 1]};
        } else {
#pragma unroll
          for (int k = 0; k < 2 * NC; k++) d2[k] = u32x2{w[2 * k], w[2 * k + 1]};
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8 * NC; e++) { // sample e of the group: pixel e / NC, component e % NC
          if (e < npx * NC) {
            if (sb == 1) dst[e] = (uint8_t)(w[e >> 2] >> (8 * (e & 3)));
            else reinterpret_cast<uint16_t *>(dst)[e] = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
          }
        }
      }
     }
    }
  };
  if (a.ncomp == 1) phase_b(std::integral_constant<int, 1>{});
  else if (a.ncomp == 2) phase_b(std::integral_constant<int, 2>{});
  else if (a.ncomp == 3) phase_b(std::integral_constant<int, 3>{});
  else phase_b(std::integral_constant<int, 4>{});
}

// ==============================================================================================
// JPEG XT profile C: legacy samples (planes 0..2) + residual samples (planes 3..5) -> 16-bit codes.
// colortrafo/ycbcrtrafo.cpp:750-829 (residual chain: Q table, R transformation, R2 table), :842-878 (legacy chain:
// L transformation, L table, C transformation = identity, merge), :897-955 (half-float clamp, INVERT_NEGS).
// The Q and R2 tables of the supported subset are identities whose scaling is a shift
// (boxes/parametrictonemappingbox.cpp:387-430 with e = 0), so they are evaluated arithmetically.
// ==============================================================================================
template <int LAYOUT, int RLAYOUT>
__global__ __launch_bounds__(256) void xt_merge_kernel(const GenericArgs a)
{
  const int groups = (a.width + 7) >> 3;
  const int gxi = blockIdx.x * blockDim.x + threadIdx.x;
  const int Y = a.y_base + blockIdx.y; // (rectangle requests: the lines of the request only)
  const int frame = blockIdx.z;
  if (gxi >= groups) return;
  const int X0 = gxi * 8;
  int s[3][8], rs[3][8];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    upsample_plane_line<LAYOUT>(a, c, c, frame, X0, Y, s[c]);
    if (a.xt_no_residual) { // nothing to merge: there may be no residual planes at all (a specification without a residual codestream)
#pragma unroll
      for (int x = 0; x < 8; x++) rs[c][x] = 0;
    } else
      upsample_plane_line<RLAYOUT>(a, 3 + c, c, frame, X0, Y, rs[c]);
  }
  uint16_t *dst = reinterpret_cast<uint16_t *>(a.out + (int64_t)frame * a.out_frame_stride + (int64_t)Y * a.row_stride) + (int64_t)X0 * 3;
  const int npx = min(8, a.width - X0);
  // All of this fits 32 bits for the subset (12-bit residual, 16-bit output), and is arranged to stay exact:
  //   Q table: clamp to [0, 2^(Pr+4) - 1], scale by 2^(16-Pr)  ->  ry = 16 * qy with qy in [0, 65535]
  //   R transformation: (ry * 8192 + rcb * Lb + rcr * Lr + 4096) >> 13 with rcb = 16 * db, db = qcb - 32768:
  //     = ry + ((db * Lb + dr * Lr + 256) >> 9)   (ry * 8192 is a multiple of 8192, 16 d L + 4096 = 16 (d L + 256))
  //   R2 table: clamp to [0, 2^20 - 1], (x + 8) >> 4
  const int rmax16 = ((1 << a.rprecision) << 4) - 1; // ((m_lRMax + 1) << COLOR_BITS) - 1
  const int omax16 = ((a.out_max + 1) << 4) - 1;
  const int qshift = 16 - a.rprecision;
  const int pinf = (a.out_max >> 1) - (a.out_max >> 6) - 1; // largest finite half: 0x7bff
  const int minf = -pinf - 1;                               // INVERT_NEGS(pinf | 0x8000) = -31744
  const bool narrow = a.legacy32 != 0; // 8-bit legacy frame that passed the range check: the 32-bit colour stage is exact
#pragma unroll
  for (int x = 0; x < 8; x++) {
    if (x >= npx) break;
    // residual chain
    const int qy = min(max(rs[0][x], 0), rmax16) << qshift, qb = min(max(rs[1][x], 0), rmax16) << qshift,
              qr = min(max(rs[2][x], 0), rmax16) << qshift; // ry, rcb, rcr before the level shift (multiples of 16 when Pr = 12)
    int rr[3];
    if (a.rtrafo_ycbcr && qshift >= 4) {
      const int db = (qb >> 4) - a.out_shift, dr = (qr >> 4) - a.out_shift; // exact: qshift >= 4
      rr[0] = qy + ((dr * L_CR_R + 256) >> 9);
      rr[1] = qy + ((-db * L_CB_G - dr * L_CR_G + 256) >> 9);
      rr[2] = qy + ((db * L_CB_B + 256) >> 9);
    } else if (a.rtrafo_ycbcr) {
      // hidden residual bits: the inputs are no multiples of 16 any more -- FIX_COLOR_TO_INTCOLOR of the 64-bit sums
      const long long y13 = ((long long)qy << 13) + 4096, cb = qb - (a.out_shift << 4), cr = qr - (a.out_shift << 4);
      rr[0] = (int)((y13 + cr * L_CR_R) >> 13);
      rr[1] = (int)((y13 - cb * L_CB_G - cr * L_CR_G) >> 13);
      rr[2] = (int)((y13 + cb * L_CB_B) >> 13);
    } else {
      rr[0] = qy; rr[1] = qb; rr[2] = qr;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) rr[c] = (min(max(rr[c], 0), omax16) + 8) >> 4;
    if (a.xt_no_residual) rr[0] = rr[1] = rr[2] = a.out_shift; // nothing to merge (colortrafo/ycbcrtrafo.cpp:744-746)
    // legacy chain
    int v[3];
    if (a.ycbcr) {
      if (narrow) {
        const int cb = s[1][x] - a.dcshift, cr = s[2][x] - a.dcshift, y13 = (s[0][x] << 13) + 65536;
        v[0] = (y13 + cr * L_CR_R) >> 17;
        v[1] = (y13 - cb * L_CB_G - cr * L_CR_G) >> 17;
        v[2] = (y13 + cb * L_CB_B) >> 17;
      } else {
        long long r, g, b;
        ycc_to_rgb_wide(s[0][x], s[1][x], s[2][x], a.dcshift, r, g, b);
        v[0] = (int)clampll(r, a.maxval); v[1] = (int)clampll(g, a.maxval); v[2] = (int)clampll(b, a.maxval);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; c++) v[c] = (int)((((long long)s[c][x]) + 8) >> 4);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int lv = a.ltable[c * a.ltable_entries + min(max(v[c], 0), a.maxval)];
      int m = lv + rr[c] - a.out_shift;
      if (a.is_float) {
        m = min(max(m, minf), pinf);
        const short w = (short)m;
        dst[3 * x + c] = (uint16_t)(short)(((w >> 15) & 0x7fff) ^ w); // INVERT_NEGS
      } else {
        dst[3 * x + c] = (uint16_t)min(max(m, 0), a.out_max);
      }
    }
  }
}

// ==============================================================================================
// JPEG XT beyond the encoder's default subset: free-form L / R / C transformations (MTRX boxes), Q and R2 tables that are
// real tables (parametric curves), residual planes that bypassed the DCT.  The literal chain of YCbCrTrafo::YCbCr2RGB
// (colortrafo/ycbcrtrafo.cpp:750-955) in 64-bit sums, table gathers from HBM (up to 2^20 entries each: not LDS material).
// ==============================================================================================
template <int LAYOUT, int RLAYOUT>
__global__ __launch_bounds__(256) void xt_merge_general_kernel(const GenericArgs a)
{
  const int groups = (a.width + 7) >> 3;
  const int gxi = blockIdx.x * blockDim.x + threadIdx.x;
  const int Y = a.y_base + blockIdx.y; // (rectangle requests: the lines of the request only)
  const int frame = blockIdx.z;
  if (gxi >= groups) return;
  const int X0 = gxi * 8;
  int s[3][8], rs[3][8];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    upsample_plane_line<LAYOUT>(a, c, c, frame, X0, Y, s[c]);
    if (a.xt_no_residual) { // nothing to merge: there may be no residual planes at all (a specification without a residual codestream)
#pragma unroll
      for (int x = 0; x < 8; x++) rs[c][x] = 0;
    } else
      upsample_plane_line<RLAYOUT>(a, 3 + c, c, frame, X0, Y, rs[c]);
  }
  uint16_t *dst = reinterpret_cast<uint16_t *>(a.out + (int64_t)frame * a.out_frame_stride + (int64_t)Y * a.row_stride) + (int64_t)X0 * 3;
  const int npx = min(8, a.width - X0);
  const int rmax16 = ((1 << a.rprecision) << 4) - 1; // ((m_lRMax + 1) << COLOR_BITS) - 1
  const int omax16 = ((a.out_max + 1) << 4) - 1;
  const int qshift = 16 - a.rprecision;
  const int pinf = (a.out_max >> 1) - (a.out_max >> 6) - 1;
  const int minf = -pinf - 1;
  for (int x = 0; x < npx; x++) {
    // residual chain: Q table, R transformation (FIX_COLOR_TO_INTCOLOR), R2 table
    long long q3[3] = {0, 0, 0};
    if (!a.xt_rct && !(a.xt_noclamp && !a.rtrafo_ycbcr)) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int idx = min(max(rs[c][x], 0), rmax16);
        q3[c] = a.qlut[c] ? (long long)a.qlut[c][idx] : (long long)idx << qshift;
      }
    }
    long long rr[3];
    if (a.xt_rct) {
      // lossless coding (colortrafo/ycbcrtrafo.cpp:752-766): the Q tables on the samples as they are -- the RCT's extra bit is a
      // precision bit, no fractional ones -- then the reversible transformation with wrap-around, all in LONGs
      const int rmax = (1 << a.rprecision) - 1;
      // (a missing table is APPLY_LUT's pass-through, :59; the host materialises them for these flavours and the launch refuses otherwise)
      int y = a.qlut[0] ? a.qlut[0][min(max(rs[0][x], 0), rmax)] : rs[0][x];
      int cb = a.qlut[1] ? a.qlut[1][min(max(rs[1][x], 0), rmax)] : rs[1][x];
      int cr = a.qlut[2] ? a.qlut[2][min(max(rs[2][x], 0), rmax)] : rs[2][x];
      y >>= 1;
      cb = (int)((unsigned)cb - ((unsigned)a.out_shift << 1));
      cr = (int)((unsigned)cr - ((unsigned)a.out_shift << 1));
      const int rg = (int)((unsigned)y - (unsigned)((int)((unsigned)cb + (unsigned)cr) >> 2)) & a.out_max;
      rr[0] = (int)((unsigned)cr + (unsigned)rg) & a.out_max;
      rr[1] = rg;
      rr[2] = (int)((unsigned)cb + (unsigned)rg) & a.out_max;
    } else if (a.xt_noclamp && !a.rtrafo_ycbcr) {
      // identity without clamping (:797-801): the Q table alone, no fractional bits, no R2 table
      const int rmax = (1 << a.rprecision) - 1;
#pragma unroll
      for (int c = 0; c < 3; c++) rr[c] = a.qlut[c] ? a.qlut[c][min(max(rs[c][x], 0), rmax)] : rs[c][x];
    } else {
    if (a.rtrafo_ycbcr) {
      // (LONG variables around QUAD products in the reference: a table entry beyond the range -- curves with parameters no
      // encoder writes -- wraps where it narrows, colortrafo/ycbcrtrafo.cpp:776-789)
      const long long ry = q3[0], rcb = (int)(q3[1] - ((long long)a.out_shift << 4)), rcr = (int)(q3[2] - ((long long)a.out_shift << 4));
#pragma unroll
      for (int c = 0; c < 3; c++) rr[c] = (int)((ry * a.rmat[3 * c] + rcb * a.rmat[3 * c + 1] + rcr * a.rmat[3 * c + 2] + 4096) >> 13);
    } else {
      rr[0] = q3[0]; rr[1] = q3[1]; rr[2] = q3[2];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int idx = (int)min(max(rr[c], 0ll), (long long)omax16);
      rr[c] = a.r2lut[c] ? (long long)a.r2lut[c][idx] : (long long)((idx + 8) >> 4);
    }
    }
    if (a.xt_no_residual) rr[0] = rr[1] = rr[2] = a.out_shift; // nothing to merge (colortrafo/ycbcrtrafo.cpp:744-746)
    // legacy chain: L transformation (FIX_COLOR_TO_INT), L table, C transformation (FIX_TO_INT)
    long long v[3];
    if (a.ycbcr) {
      const long long yy = s[0][x], cb = (long long)s[1][x] - a.dcshift, cr = (long long)s[2][x] - a.dcshift;
#pragma unroll
      for (int c = 0; c < 3; c++) v[c] = (yy * a.lmat[3 * c] + cb * a.lmat[3 * c + 1] + cr * a.lmat[3 * c + 2] + 65536) >> 17;
    } else {
#pragma unroll
      for (int c = 0; c < 3; c++) v[c] = ((long long)s[c][x] + 8) >> 4;
    }
    long long lv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) lv[c] = a.ltable[c * a.ltable_entries + (int)min(max(v[c], 0ll), (long long)a.maxval)];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      long long m = (int)(((lv[0] * a.cmat[3 * c] + lv[1] * a.cmat[3 * c + 1] + lv[2] * a.cmat[3 * c + 2] + 4096) >> 13) + rr[c] - a.out_shift); // (:868-879)
      if (a.xt_noclamp) { // :940-972: the sign conversion of half float codes alone, or wrap-around
        const short w = (short)m;
        const unsigned v = a.is_float ? (unsigned)(uint16_t)(short)(((w >> 15) & 0x7fff) ^ w) : (unsigned)m & (unsigned)a.out_max;
        if (a.sample_bytes == 1) reinterpret_cast<uint8_t *>(a.out + (int64_t)frame * a.out_frame_stride + (int64_t)Y * a.row_stride)[3 * (X0 + x) + c] = (uint8_t)v;
        else dst[3 * x + c] = (uint16_t)v;
      } else if (a.is_float) {
        m = min(max(m, (long long)minf), (long long)pinf);
        const short w = (short)m;
        dst[3 * x + c] = (uint16_t)(short)(((w >> 15) & 0x7fff) ^ w); // INVERT_NEGS
      } else if (a.sample_bytes == 1) { // an output of at most eight bits (OCON without extra range bits: jpeg -r on 8-bit input)
        reinterpret_cast<uint8_t *>(a.out + (int64_t)frame * a.out_frame_stride + (int64_t)Y * a.row_stride)[3 * (X0 + x) + c] =
            (uint8_t)min(max(m, 0ll), (long long)a.out_max);
      } else {
        dst[3 * x + c] = (uint16_t)min(max(m, 0ll), (long long)a.out_max);
      }
    }
  }
}

// The merge for ONE component: a grey scale picture with a residual (`jpeg -r .. in.pgm`).  Every transformation is the identity
// (codestream/tables.cpp:2003-2005, 2055-2060, 2079-2081; YCbCrTrafo<.., 1, .., Identity, Identity>, colortransformerfactory.cpp:
// 681-757): out = clamp(L[(y + 8) >> 4] + R2[Q[residual]] - 2^(bits - 1)) with the tables (or their arithmetic identities at 16
// bits) of component 0, as half-float code or integer, one or two bytes per sample.  Planes 0 (legacy) and 3 (residual).
__global__ __launch_bounds__(256) void xt_merge1_kernel(const GenericArgs a)
{
  const int groups = (a.width + 7) >> 3;
  const int gxi = blockIdx.x * blockDim.x + threadIdx.x;
  const int Y = a.y_base + blockIdx.y;
  const int frame = blockIdx.z;
  if (gxi >= groups) return;
  const int X0 = gxi * 8;
  int s[8], rs[8];
  upsample_plane_line<LAYOUT_ANY>(a, 0, 0, frame, X0, Y, s);
  if (a.xt_no_residual) { // (no residual plane to read)
#pragma unroll
    for (int x = 0; x < 8; x++) rs[x] = 0;
  } else
    upsample_plane_line<LAYOUT_ANY>(a, 3, 0, frame, X0, Y, rs);
  uint8_t *line = a.out + (int64_t)frame * a.out_frame_stride + (int64_t)Y * a.row_stride;
  const int npx = min(8, a.width - X0);
  const int rmax16 = ((1 << a.rprecision) << 4) - 1; // ((m_lRMax + 1) << COLOR_BITS) - 1
  const int omax16 = ((a.out_max + 1) << 4) - 1;
  const int qshift = 16 - a.rprecision;
  const int pinf = (a.out_max >> 1) - (a.out_max >> 6) - 1;
  const int minf = -pinf - 1;
  for (int x = 0; x < npx; x++) {
    long long rr;
    if (a.xt_noclamp) { // identity without clamping (colortrafo/ycbcrtrafo.cpp:820-822): the Q table alone, no fractional bits
      rr = a.qlut[0] ? (long long)a.qlut[0][min(max(rs[x], 0), (1 << a.rprecision) - 1)] : (long long)rs[x];
    } else {
      const int idx = min(max(rs[x], 0), rmax16);
      const long long q = a.qlut[0] ? (long long)a.qlut[0][idx] : (long long)idx << qshift;
      const int idx2 = (int)min(max(q, 0ll), (long long)omax16);
      rr = a.r2lut[0] ? (long long)a.r2lut[0][idx2] : (long long)((idx2 + 8) >> 4);
    }
    if (a.xt_no_residual) rr = a.out_shift; // nothing to merge (colortrafo/ycbcrtrafo.cpp:744-746)
    const long long v = ((long long)s[x] + 8) >> 4;
    const long long lv = a.ltable[(int)min(max(v, 0ll), (long long)a.maxval)];
    long long m = (int)(lv + rr - a.out_shift); // (a LONG: colortrafo/ycbcrtrafo.cpp:886-890)
    if (a.xt_noclamp) {
      const short w = (short)m;
      const unsigned v = a.is_float ? (unsigned)(uint16_t)(short)(((w >> 15) & 0x7fff) ^ w) : (unsigned)m & (unsigned)a.out_max;
      if (a.sample_bytes == 1) line[X0 + x] = (uint8_t)v;
      else reinterpret_cast<uint16_t *>(line)[X0 + x] = (uint16_t)v;
    } else if (a.is_float) {
      m = min(max(m, (long long)minf), (long long)pinf);
      const short w = (short)m;
      reinterpret_cast<uint16_t *>(line)[X0 + x] = (uint16_t)(short)(((w >> 15) & 0x7fff) ^ w); // INVERT_NEGS
    } else if (a.sample_bytes == 1) {
      line[X0 + x] = (uint8_t)min(max(m, 0ll), (long long)a.out_max);
    } else {
      reinterpret_cast<uint16_t *>(line)[X0 + x] = (uint16_t)min(max(m, 0ll), (long long)a.out_max);
    }
  }
}

// Residual planes whose DCT was bypassed (RDCT box): ResidualBlockHelper::DequantizeResidual without a transform,
// control/residualblockhelper.cpp:203-231 -- sample = coefficient * (delta[63] << 4) + 2^(Pr-1), optionally the 2 x 2 noise
// shaping average.  One thread per block; planes 3..5 of the sample workspace (they overwrite what idct_planes_kernel put there).
__global__ __launch_bounds__(256) void bypass_planes_kernel(const GenericArgs a)
{
  const int c = blockIdx.y % 3, frame = blockIdx.y / 3, p = 3 + c;
  const int nblocks = a.bw[p] * a.bh[p];
  const int blk = blockIdx.x * blockDim.x + threadIdx.x;
  if (blk >= nblocks) return;
  const bool wide = p >= a.wide_first && p < a.wide_first + a.wide_count; // int32 coefficients (two int16 slots each)
  const int16_t *plane = a.coef + (int64_t)frame * a.coef_frame_stride + a.coef_off[p];
  const int by = blk / a.bw[p], bx = blk - by * a.bw[p];
  // rectangle requests: the block row's coefficients come from the row the residual image's cursor stood at (GenericArgs::rowmap);
  // a row that does not exist keeps what idct_planes_kernel made of it
  int sblk = blk;
  if (a.rowmap) {
    const int srow = a.rowmap[p * a.rowmap_stride + by];
    if (srow < 0) return;
    sblk = srow * a.bw[p] + bx;
  }
  int res[64];
  if (wide) {
    const int *src = reinterpret_cast<const int *>(plane) + (int64_t)sblk * 64;
#pragma unroll
    for (int i = 0; i < 64; i++) res[i] = src[i];
  } else {
    const int16_t *src = plane + (int64_t)sblk * 64;
#pragma unroll
    for (int i = 0; i < 64; i++) res[i] = src[i];
  }
  const int quant = a.rquant63[c], dcs = a.rdcshift;
  const int pitch = a.bw[p] * 8;
  int *dst = a.samples + (int64_t)frame * a.sample_frame_stride + a.sample_off[p] + ((int64_t)by * 8) * pitch + bx * 8;
  for (int y = 0; y < 64; y += 16)
    for (int x = 0; x < 8; x += 2) {
      int avg = 0;
      if (a.rnoise)
        for (int dy = 0; dy < 16; dy += 8)
          for (int dx = 0; dx < 2; dx++) avg += res[x + dx + y + dy] * quant;
      avg = (avg + 2) >> 2;
      for (int dy = 0; dy < 16; dy += 8)
        for (int dx = 0; dx < 2; dx++) {
          const int i = x + dx + y + dy;
          int v = res[i] * quant;
          if (a.rnoise && v > avg - quant && v < avg + quant) v = avg;
          dst[(int64_t)(i >> 3) * pitch + (i & 7)] = v + dcs;
        }
    }
}

// ==============================================================================================
// fused kernel for frames whose components all have the sampling factors 1 x 1 and leave without a colour transformation:
// CMYK (four components: Adobe files), RGB stored as such (three: Adobe transform 0, the reference's `jpeg -c`, a merging
// specification with the identity L transformation)
// ==============================================================================================
// ReconstructUnsampled (control/blockbitmaprequester.cpp:1013-1074) with the identity transformation
// (colortrafo/ycbcrtrafo.cpp:852-856, 921-936): sample = clamp((IDCT + 8) >> 4).  fused444_kernel's decomposition -- one lane
// owns one block POSITION of a 128 x 128 tile and transforms its NC blocks in turn, nothing goes through LDS but the block
// fetch -- with the samples of every component packed to bytes right behind its transform (v_ashr_pk_u8_i32: shift, clamp
// and pack; the + 8 rides in the second pass's rounding constant), 16 dwords per component.  The interleaved line is a byte
// permutation of those: a 4 x 4 byte transpose for four components (8 v_perm_b32 per 4 pixels), 6 per 4 pixels for three.
// Algorithmic bytes: 2 NC in + NC out per pixel (CMYK 12, RGB 9).  The tile kernel took these layouts through LDS before
// (CMYK 0.49 of 8 TB/s).
#ifndef FLAT4_MINW
#define FLAT4_MINW 2 // four components: 64 packed dwords + a transform's 96 do not fit the 170 registers of three workgroups per CU
#endif
template <int NC>
__global__ __launch_bounds__(F420_THREADS, NC == 4 ? FLAT4_MINW : 3) void fused_flat_kernel(const GenericArgs a)
{
  static_assert(NC == 3 || NC == 4, "three or four components");
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4][128];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 *stage = stage_all[wave];

  const unsigned logical = tile_of_workgroup(blockIdx.x, (unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames); // (tile order: see there)
  if (logical == ~0u) return; // the launch is padded to whole groups of eight tile rows
  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int frame = logical / tiles_per_frame;
  const int tile = logical - frame * tiles_per_frame;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int16_t *__restrict__ coef = a.coef + (int64_t)frame * a.coef_frame_stride;

  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int gbx = tx * F420_TILE_BLOCKS + bx, gby = ty * F420_TILE_BLOCKS + by;
  const int gbx0 = tx * F420_TILE_BLOCKS, gby0 = ty * F420_TILE_BLOCKS + wave * 4;
  const int x0 = gbx0 + (lane >> 3);
  const int bw = a.bw[0], bh = a.bh[0]; // every plane has the same geometry
  const int X0 = gbx * 8, Y0 = gby * 8;

  unsigned pk[NC][16]; // component c, pixels 4 k .. 4 k + 3 of the block (row k / 2, half k % 2)
#pragma unroll
  for (int c = 0; c < NC; c++) {
    u32x4 rows[8];
    int v[64];
    const char *pbase = reinterpret_cast<const char *>(coef + a.coef_off[c]) + (lane & 7) * 16;
    fetch_blocks(rows, stage, lane, [&](int m) -> const u32x4 * {
      const int x = min(x0 + 8 * (m & 1), bw - 1), y = min(gby0 + (m >> 1), bh - 1);
      return reinterpret_cast<const u32x4 *>(pbase + (unsigned)((y * bw + x) * 128));
    });
    if (c == NC - 1 && (X0 >= a.width || Y0 >= a.height)) return; // (no barrier in this kernel; the last fetch needed every lane)
    // level shift inside, + 8 of COLOR_TO_INT in the second pass's rounding constant: v = sample * 16 + 8
    dequant_idct_sparse<true>(rows, a.q[c], v, a.dcoff[c], 2048 + (8 << 12));
#pragma unroll
    for (int k = 0; k < 16; k++) {
      pk[c][k] = ashr_sat_pack4<4>(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
      asm volatile("" : "+v"(pk[c][k])); // (the packed word exists from here on: see fusedxtw420_kernel, profiles/r06/xt_kernels.txt)
    }
    __builtin_amdgcn_sched_barrier(0); // keep the next component's loads from being hoisted above this transform (register pressure)
  }
  uint8_t *__restrict__ out_frame = a.out + (int64_t)frame * a.out_frame_stride;
  const unsigned out_off = (unsigned)Y0 * (unsigned)a.row_stride + (unsigned)X0 * (unsigned)NC;
  const int npx = min(8, a.width - X0);
  const int nln = min(8, a.height - Y0);
#pragma unroll
  for (int l = 0; l < 8; l++) {
    if (l < nln) {
      unsigned w[2 * NC]; // the line's 8 * NC bytes
#pragma unroll
      for (int h = 0; h < 2; h++) { // pixels 4 h .. 4 h + 3
        const unsigned A = pk[0][2 * l + h], B = pk[1][2 * l + h], C = pk[2][2 * l + h];
        if (NC == 4) {
          const unsigned D = pk[NC - 1][2 * l + h];
          const unsigned ab_lo = __builtin_amdgcn_perm(B, A, 0x05010400u), ab_hi = __builtin_amdgcn_perm(B, A, 0x07030602u);
          const unsigned cd_lo = __builtin_amdgcn_perm(D, C, 0x05010400u), cd_hi = __builtin_amdgcn_perm(D, C, 0x07030602u);
          w[4 * h + 0] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u);
          w[4 * h + 1] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
          w[4 * h + 2] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u);
          w[4 * h + 3] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
        } else {
          // a0 b0 c0 a1 | b1 c1 a2 b2 | c2 a3 b3 c3
          const unsigned t0 = __builtin_amdgcn_perm(B, A, 0x01000400u); // a0 b0 .. a1
          const unsigned t1 = __builtin_amdgcn_perm(B, A, 0x06020005u); // b1 .. a2 b2
          const unsigned t2 = __builtin_amdgcn_perm(B, A, 0x00070300u); // .. a3 b3 ..
          w[3 * h + 0] = __builtin_amdgcn_perm(C, t0, 0x03040100u);    // c0 into byte 2
          w[3 * h + 1] = __builtin_amdgcn_perm(C, t1, 0x03020500u);    // c1 into byte 1
          w[3 * h + 2] = __builtin_amdgcn_perm(C, t2, 0x07020106u);    // c2 into byte 0, c3 into byte 3
        }
      }
      if (npx == 8) {
        const unsigned off = out_off + (unsigned)l * (unsigned)a.row_stride;
        if (NC == 4) {
          const unsigned w8[8] = {w[0], w[1], w[2], w[3], w[4 % (2 * NC)], w[5 % (2 * NC)], w[6 % (2 * NC)], w[7 % (2 * NC)]};
          store32_nt(out_frame, off, w8);
        } else {
          const unsigned w6[6] = {w[0], w[1], w[2], w[3], w[4], w[5]};
          store24_nt(out_frame, off, w6);
        }
      } else {
        uint8_t *dst = out_frame + (out_off + (unsigned)l * (unsigned)a.row_stride);
#pragma unroll
        for (int e = 0; e < 8 * NC; e++)
          if (e < npx * NC) dst[e] = (uint8_t)(w[e >> 2] >> (8 * (e & 3)));
      }
    }
  }
}

// ==============================================================================================
// launchers
// ==============================================================================================
// floor(2^32 / d) + 1 for the two divisions of tile_position, when every dividend the launch can produce keeps x * d < 2^32
static Fused420Args with_tile_magic(const Fused420Args &a0)
{
  Fused420Args a = a0;
  const uint64_t tiles_x = (uint64_t)a.tiles_x, tiles_y = (uint64_t)a.tiles_y, rows = tiles_y * (uint64_t)a.frames + 8;
  const auto magic = [](uint64_t d, uint64_t max_dividend) -> uint32_t {
    return d > 1 && max_dividend * d < (1ull << 32) ? (uint32_t)((1ull << 32) / d) + 1u : 0u; // (d = 1: 2^32 + 1 does not fit)
  };
  a.magic_tx = magic(tiles_x, rows * tiles_x); // dividends: a workgroup's index inside its XCD, or a tile number
  a.magic_ty = magic(tiles_y, rows);
  return a;
}

#ifndef F420_FAST_MINW
#define F420_FAST_MINW 2 // workgroups per CU of the unpacked 4:2:0 kernel, fast flavour (A-B builds)
#endif
#ifndef F420_12_MINW
#define F420_12_MINW 2
#endif
int launch_fused420(const Fused420Args &a0, bool fast, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (total == 0) return 0;
  // two workgroups per CU for both flavours (132 / 194 VGPRs)
  if (a.qdev) {
    if (!fast) hipLaunchKernelGGL((fused420_kernel<false, 2, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((fused420_kernel<true, F420_FAST_MINW, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  } else if (!fast) hipLaunchKernelGGL((fused420_kernel<false, 2, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((fused420_kernel<true, F420_FAST_MINW, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_fused420_12(const Fused420Args &a0, bool narrow, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles<1>((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (total == 0) return 0;
  if (a.qdev) {
    if (narrow) hipLaunchKernelGGL((fused420_kernel<true, F420_12_MINW, true, 12, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((fused420_kernel<true, F420_12_MINW, true, 12, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  } else if (narrow) hipLaunchKernelGGL((fused420_kernel<true, F420_12_MINW, false, 12, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((fused420_kernel<true, F420_12_MINW, false, 12, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_fused420p(const Fused420Args &a0, bool dot2, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (total == 0) return 0;
  // four workgroups per CU (127 VGPRs with the luma prefetch, 27 KB LDS); the per-frame-table build three (133 VGPRs)
  if (a.qdev) hipLaunchKernelGGL((fused420p_kernel<3, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else if (dot2) hipLaunchKernelGGL((fused420p_kernel<F420P_MINW, false, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((fused420p_kernel<F420P_MINW, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_fusedxt420(const FusedXtArgs &x0, hipStream_t stream)
{
  FusedXtArgs x = x0;
  x.base = with_tile_magic(x0.base);
  const unsigned total = workgroups_for_tiles<XT_TILE_ORDER>((unsigned)x.base.tiles_x, (unsigned)x.base.tiles_y * (unsigned)x.base.frames);
  if (x.ext.rprecision > 12) {
    static const bool one_wave = getenv("MIJPEG_XTW_ONE_WAVE") != nullptr; // A-B comparisons
    if (x.luma_fits16 && x.ext.is_float && !one_wave) hipLaunchKernelGGL(fusedxtw420_kernel<true>, dim3(total), dim3(F420_THREADS), 0, stream, x.base, x.ext);
    else hipLaunchKernelGGL(fusedxtw420_kernel<false>, dim3(total), dim3(F420_THREADS), 0, stream, x.base, x.ext);
  }
  else hipLaunchKernelGGL(fusedxt420_kernel, dim3(total), dim3(F420_THREADS), 0, stream, x.base, x.ext);
  return (int)hipGetLastError();
}

int launch_fused422(const Fused420Args &a0, bool wide, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (wide) {
    if (a.qdev) hipLaunchKernelGGL((fused422_kernel<3, true, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((fused422_kernel<3, false, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  } else {
    if (a.qdev) hipLaunchKernelGGL((fused422_kernel<3, true, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((fused422_kernel<3, false, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  }
  return (int)hipGetLastError();
}

int launch_fused411(const Fused420Args &a0, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (a.qdev) hipLaunchKernelGGL((fused411_kernel<3, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((fused411_kernel<3, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_fused1(const Fused420Args &a0, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles<1>((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (a.qdev) hipLaunchKernelGGL((fused1_kernel<true, 8>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((fused1_kernel<false, 8>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_fused1_12(const Fused420Args &a0, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles<1>((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (a.qdev) hipLaunchKernelGGL((fused1_kernel<true, 12>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((fused1_kernel<false, 12>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_fused440(const Fused420Args &a0, bool wide, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (wide) {
    if (a.qdev) hipLaunchKernelGGL((fused440_kernel<3, true, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((fused440_kernel<3, false, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  } else {
    if (a.qdev) hipLaunchKernelGGL((fused440_kernel<3, true, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((fused440_kernel<3, false, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  }
  return (int)hipGetLastError();
}

int launch_fused444(const Fused420Args &a0, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (total == 0) return 0;
  // 168 VGPRs -> three waves per SIMD: 5 % faster than the unconstrained 171-register build
  if (a.qdev) hipLaunchKernelGGL((fused444_kernel<2, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((fused444_kernel<3, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_fused422_12(const Fused420Args &a0, bool narrow, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (total == 0) return 0;
  if (a.qdev) {
    if (narrow) hipLaunchKernelGGL((fused422_12_kernel<true, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((fused422_12_kernel<true, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  } else if (narrow) hipLaunchKernelGGL((fused422_12_kernel<false, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((fused422_12_kernel<false, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  return (int)hipGetLastError();
}

int launch_fused444_12(const Fused420Args &a0, bool narrow, hipStream_t stream)
{
  const Fused420Args a = with_tile_magic(a0);
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (total == 0) return 0;
  if (a.qdev) {
    if (narrow) hipLaunchKernelGGL((fused444_12_kernel<true, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((fused444_12_kernel<true, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  } else if (narrow) hipLaunchKernelGGL((fused444_12_kernel<false, true>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((fused444_12_kernel<false, false>), dim3(total), dim3(F420_THREADS), 0, stream, a);
  return (int)hipGetLastError();
}

// per-frame tables as the client hands them over (u16 deltas, [frames][4][64]) -> the operands of the transforms (<< 4, int32)
__global__ __launch_bounds__(256) void expand_deltas_kernel(const uint16_t *__restrict__ in, int32_t *__restrict__ out, int n)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i] << 4;
}

int launch_expand_deltas(const uint16_t *in, int32_t *out, int frames, hipStream_t stream)
{
  const int n = frames * 4 * 64;
  hipLaunchKernelGGL(expand_deltas_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, out, n);
  return (int)hipGetLastError();
}

// Tile of the fused tile kernel: whole MCUs, about 64 x 64 pixels -- wider for horizontally subsampled frames, where a
// wider tile halves the share of halo blocks -- shrunk until the sample planes fit 64 KB of LDS together with the fetch
// staging (two workgroups per CU at least).  Returns the dynamic LDS the launch needs (0: nothing fits).
static size_t fused_tile_geometry(GenericArgs &a, bool narrow)
{
  int hmax = 1, vmax = 1;
  for (int c = 0; c < a.ncomp; c++) { hmax = max(hmax, a.subx[c]); vmax = max(vmax, a.suby[c]); }
  const int mw = 8 * hmax, mh = 8 * vmax;
  static const int env_w = getenv("MIJPEG_TILE_W") ? atoi(getenv("MIJPEG_TILE_W")) : 0, env_h = getenv("MIJPEG_TILE_H") ? atoi(getenv("MIJPEG_TILE_H")) : 0; // A-B measurements
  // a subsampled direction pays a block row / column of halo on each side of the tile: twice the extent there halves its share
  // (1x4 frames: 327 -> 479 Gpixel/s with 128 lines instead of 64)
  const int want_w[3] = {env_w ? env_w : hmax > 1 ? 128 : 64, 64, 32};
  int want_h[4] = {env_h ? env_h : vmax > 1 ? 128 : 64, 64, 48, 32};
  if (want_h[0] == 64) { want_h[1] = 48; want_h[2] = 32; want_h[3] = 16; }
  auto plane_bytes = [&](int tw, int th) {
    size_t samples = 0;
    for (int c = 0; c < a.ncomp; c++) {
      // (a tile of whole MCUs starts on a block boundary of every component; the halo sample on each side costs one more block)
      const int wx = tw / (8 * a.subx[c]) + (a.subx[c] > 1 ? 2 : 0), wy = th / (8 * a.suby[c]) + (a.suby[c] > 1 ? 2 : 0);
      samples += (size_t)wy * 8 * (wx * 8 + TILE_PAD) + TILE_PAD;
    }
    return samples * (narrow ? 2 : 4);
  };
  // LDS per workgroup: 64 KB at most (two workgroups per CU); a tile that needs more than 160 / 3 KB gives way to the next
  // smaller one if that has at least three quarters of its lines and fits three times (12-bit 4:4:4: 64 x 48 pixels, 244 -> 270
  // Gpixel/s).  (Halving the fetch staging instead was measured: the second fetch routine alone, never called, cost every
  // layout 3 to 8 %.)
  const size_t two = 64 * 1024, three = (160 * 1024 / 3) & ~(size_t)255, staging = 4 * 128 * 16;
  for (int iw = 0; iw < 3; iw++)
    for (int ih = 0; ih < 4; ih++) {
      const int tw = mw * max(1, want_w[iw] / mw);
      int th = mh * max(1, want_h[ih] / mh);
      if (plane_bytes(tw, th) + staging > two) continue;
      if (plane_bytes(tw, th) + staging > three && ih + 1 < 4 && !env_h) {
        const int th2 = mh * max(1, want_h[ih + 1] / mh);
        if (th2 * 4 >= th * 3 && plane_bytes(tw, th2) + staging <= three) th = th2;
      }
      a.tile_w = tw; a.tile_h = th;
      a.tiles_x = (a.width + a.tile_w - 1) / a.tile_w;
      a.tiles_y = (a.height + a.tile_h - 1) / a.tile_h;
      return plane_bytes(tw, th) + staging;
    }
  return 0;
}

int launch_fused_tile(const GenericArgs &a0, bool fast, hipStream_t stream)
{
  GenericArgs a = a0;
  const bool narrow = fast && a.narrow;
  const size_t lds = fused_tile_geometry(a, narrow);
  if (!lds) return -1;
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (narrow) hipLaunchKernelGGL((fused_tile_kernel<true, true>), dim3(total), dim3(256), lds, stream, a);
  else if (fast) hipLaunchKernelGGL((fused_tile_kernel<true, false>), dim3(total), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL((fused_tile_kernel<false, false>), dim3(total), dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}

int launch_fused_flat(const GenericArgs &a0, hipStream_t stream)
{
  GenericArgs a = a0;
  a.tiles_x = (a.width + 127) / 128;
  a.tiles_y = (a.height + 127) / 128;
  const unsigned total = workgroups_for_tiles((unsigned)a.tiles_x, (unsigned)a.tiles_y * (unsigned)a.frames);
  if (a.ncomp == 4) hipLaunchKernelGGL(fused_flat_kernel<4>, dim3(total), dim3(F420_THREADS), 0, stream, a);
  else if (a.ncomp == 3) hipLaunchKernelGGL(fused_flat_kernel<3>, dim3(total), dim3(F420_THREADS), 0, stream, a);
  else return -1;
  return (int)hipGetLastError();
}

int launch_generic(const GenericArgs &a, bool fast, hipStream_t stream)
{
  int maxblocks = 0;
  for (int c = 0; c < a.nplanes; c++) maxblocks = max(maxblocks, a.bw[c] * a.bh[c]);
  dim3 g1((maxblocks + 255) / 256, a.nplanes * a.frames);
  const bool narrow = fast && a.narrow && !a.xt && a.wide_count == 0;
  if (narrow)
    if (a.qdev) hipLaunchKernelGGL((idct_planes_kernel<true, true, true>), g1, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((idct_planes_kernel<true, false, true>), g1, dim3(256), 0, stream, a);
  else if (fast)
    if (a.qdev) hipLaunchKernelGGL((idct_planes_kernel<true, true>), g1, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((idct_planes_kernel<true, false>), g1, dim3(256), 0, stream, a);
  else
    if (a.qdev) hipLaunchKernelGGL((idct_planes_kernel<false, true>), g1, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((idct_planes_kernel<false, false>), g1, dim3(256), 0, stream, a);
  if (a.wide_count > 0) {
    int wb = 0;
    for (int c = a.wide_first; c < a.wide_first + a.wide_count; c++) wb = max(wb, a.bw[c] * a.bh[c]);
    // (workgroups of four waves: one per SIMD)
    if (a.wide_long) hipLaunchKernelGGL(idct_planes_long_kernel, dim3((wb + 255) / 256, a.wide_count * a.frames), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(idct_planes_wide_kernel, dim3((wb + 255) / 256, a.wide_count * a.frames), dim3(256), 0, stream, a);
  }
  const int groups = (a.width + 7) >> 3;
  const int bs = groups >= 256 ? 256 : 64;
  dim3 g2((groups + bs - 1) / bs, a.y_count > 0 ? a.y_count : a.height, a.frames);
  // layout of planes [first, first + 3): luma 1x1 and equal chroma factors in {1,2}^2 get a specialised instance
  auto layout_of = [&](int first, int n) {
    if (n != 3 || a.subx[first] != 1 || a.suby[first] != 1 || a.subx[first + 1] != a.subx[first + 2] || a.suby[first + 1] != a.suby[first + 2] ||
        a.subx[first + 1] > 2 || a.suby[first + 1] > 2)
      return LAYOUT_ANY;
    return layout_id(a.subx[first + 1], a.suby[first + 1]);
  };
  const int lay = a.request ? LAYOUT_ANY : layout_of(0, a.ncomp); // requests: window and displacement live in the runtime-factor instance
#define LAUNCH_COLOR(F, L) hipLaunchKernelGGL((upsample_color_kernel<F, L>), g2, dim3(bs), 0, stream, a)
#define LAUNCH_XT(L, R)                                                                       \
  do {                                                                                         \
    if (a.xt_general) hipLaunchKernelGGL((xt_merge_general_kernel<L, R>), g2, dim3(bs), 0, stream, a); \
    else hipLaunchKernelGGL((xt_merge_kernel<L, R>), g2, dim3(bs), 0, stream, a);              \
  } while (0)
  if (a.xt && a.rbypass) {
    int rb = 0;
    for (int c = 3; c < 6; c++) rb = max(rb, a.bw[c] * a.bh[c]);
    hipLaunchKernelGGL(bypass_planes_kernel, dim3((rb + 255) / 256, 3 * a.frames), dim3(256), 0, stream, a);
  }
  if (a.xt && a.ncomp == 1) {
    hipLaunchKernelGGL(xt_merge1_kernel, g2, dim3(bs), 0, stream, a);
  } else if (a.xt) {
    const int rlay = a.request ? LAYOUT_ANY : layout_of(3, 3);
    if (rlay == layout_id(1, 1) && lay == layout_id(2, 2)) LAUNCH_XT(layout_id(2, 2), layout_id(1, 1));
    else if (rlay == layout_id(1, 1) && lay == layout_id(1, 1)) LAUNCH_XT(layout_id(1, 1), layout_id(1, 1));
    else if (rlay == layout_id(1, 1) && lay == layout_id(2, 1)) LAUNCH_XT(layout_id(2, 1), layout_id(1, 1));
    else if (rlay == layout_id(2, 2) && lay == layout_id(2, 2)) LAUNCH_XT(layout_id(2, 2), layout_id(2, 2));
    else LAUNCH_XT(LAYOUT_ANY, LAYOUT_ANY);
  } else if (narrow) {
#define LAUNCH_COLOR_NARROW(L) hipLaunchKernelGGL((upsample_color_kernel<true, L, true>), g2, dim3(bs), 0, stream, a)
    if (lay == layout_id(1, 1)) LAUNCH_COLOR_NARROW(layout_id(1, 1));
    else if (lay == layout_id(2, 2)) LAUNCH_COLOR_NARROW(layout_id(2, 2));
    else if (lay == layout_id(2, 1)) LAUNCH_COLOR_NARROW(layout_id(2, 1));
    else if (lay == layout_id(1, 2)) LAUNCH_COLOR_NARROW(layout_id(1, 2));
    else LAUNCH_COLOR_NARROW(LAYOUT_ANY);
#undef LAUNCH_COLOR_NARROW
  } else if (fast) {
    if (lay == layout_id(1, 1)) LAUNCH_COLOR(true, layout_id(1, 1));
    else if (lay == layout_id(2, 2)) LAUNCH_COLOR(true, layout_id(2, 2));
    else if (lay == layout_id(2, 1)) LAUNCH_COLOR(true, layout_id(2, 1));
    else if (lay == layout_id(1, 2)) LAUNCH_COLOR(true, layout_id(1, 2));
    else LAUNCH_COLOR(true, LAYOUT_ANY);
  } else {
    if (lay == layout_id(1, 1)) LAUNCH_COLOR(false, layout_id(1, 1));
    else if (lay == layout_id(2, 2)) LAUNCH_COLOR(false, layout_id(2, 2));
    else if (lay == layout_id(2, 1)) LAUNCH_COLOR(false, layout_id(2, 1));
    else LAUNCH_COLOR(false, LAYOUT_ANY);
  }
#undef LAUNCH_COLOR
#undef LAUNCH_XT
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// device-side bitmap hand-off: what PushReconstructedData's stores through ImageBitMap{ptr, BytesPerPixel,
// BytesPerRow} do in the reference (colortrafo/ycbcrtrafo.cpp:974-1006), for bitmaps that live in HBM
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scatter_rect_kernel(const ScatterArgs a)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= a.w) return;
  const uint8_t *src = a.src + (int64_t)(a.y0 + y) * a.src_row + (int64_t)(a.x0 + x) * a.ncomp * a.sample_bytes;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    if (c < a.c0 || c > a.c1 || !a.dst[c]) continue;
    uint8_t *out = a.dst[c] + (int64_t)(a.y0 + y) * a.bytes_per_row[c] + (int64_t)(a.x0 + x) * a.bytes_per_pixel[c];
    if (a.sample_bytes == 1) out[0] = src[c];
    else {
      out[0] = src[2 * c];
      out[1] = src[2 * c + 1];
    }
  }
}

int launch_scatter_rect(const ScatterArgs &a, hipStream_t stream)
{
  if (a.w <= 0 || a.h <= 0) return 0;
  hipLaunchKernelGGL(scatter_rect_kernel, dim3((a.w + 255) / 256, a.h), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

} // namespace mij
