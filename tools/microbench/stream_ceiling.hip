// What does HBM deliver for the headline launch's traffic, without the arithmetic?  Three kernels over the buffers of 8 x 8K 4:2:0
// frames (coefficients int16 planar: 3 B/pixel in, interleaved RGB: 3 B/pixel out):
//   copy     : the ideal -- every lane reads 16 contiguous bytes and writes 16 contiguous bytes, grid-stride, dwordx4 both ways;
//   pattern  : the fused kernel's access pattern -- one workgroup per 128 x 128 tile (XCD-aware order), phase A reads the
//              10 x 10 chroma blocks of both planes (1 KB per wave-instruction), phase B the 16 x 16 luma blocks, then every lane
//              writes the 8 lines of its 8 x 8 block as 24-byte pieces (dwordx4 + dwordx2, non-temporal), lines 23 040 B apart;
//   pattern-t: the same with temporal stores.
// The stored values are a cheap mix of what was loaded, so nothing is optimised away.  Prints ms per launch and TB/s of the
// 6 B/pixel both ways.  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_ceiling.hip -o tools/microbench/stream_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int W = 7680, H = 4320, FRAMES = 8;
constexpr int BWY = W / 8, BHY = H / 8, BWC = W / 16, BHC = H / 16;
constexpr int64_t COEF_FRAME = (int64_t)BWY * BHY * 64 + 2ll * BWC * BHC * 64; // int16 units
constexpr int64_t OUT_FRAME = (int64_t)W * H * 3;
constexpr int TX = W / 128, TY = (H + 127) / 128;

__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, int64_t n_in, int64_t n_out)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  // the two streams are equally long here (3 B/pixel each way): one load and one store per iteration
  for (; i < n_in && i < n_out; i += stride) {
    const u32x4 v = in[i];
    acc ^= v;
    __builtin_nontemporal_store(v + acc, out + i);
  }
}

template <bool NT>
__global__ __launch_bounds__(256, 4) void pattern_kernel(const int16_t *__restrict__ coef_all, uint8_t *__restrict__ out_all)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned total = TX * TY * FRAMES;
  unsigned logical;
  {
    const unsigned b = blockIdx.x, q = total >> 3, r = total & 7, x = b & 7, i = b >> 3;
    logical = x * q + min(x, r) + i;
  }
  const int frame = logical / (TX * TY), tile = logical - frame * (TX * TY);
  const int ty = tile / TX, tx = tile - ty * TX;
  const int16_t *coef = coef_all + frame * COEF_FRAME;
  const int64_t off_cb = (int64_t)BWY * BHY * 64, off_cr = off_cb + (int64_t)BWC * BHC * 64;
  u32x4 acc = {0, 0, 0, 0};
  { // phase A: waves 0, 1 Cb, waves 2, 3 Cr; 100 blocks each over two waves
    const int16_t *plane = coef + (wave >> 1 ? off_cr : off_cb);
    const int gx0 = tx * 8 - 1, gy0 = ty * 8 - 1, base = (wave & 1) * 64;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int i = min(base + (lane >> 3) + 8 * m, 99), y = i / 10, x = i - y * 10;
      const int gx = min(max(gx0 + x, 0), BWC - 1), gy = min(max(gy0 + y, 0), BHC - 1);
      acc ^= *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(plane) + (lane & 7) * 16 + (unsigned)((gy * BWC + gx) * 128));
    }
  }
  __syncthreads();
  { // phase B: 16 x 4 luma blocks per wave
    const int gbx0 = tx * 16, gby0 = ty * 16 + wave * 4;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int x = min(gbx0 + (lane >> 3) + 8 * (m & 1), BWY - 1), y = min(gby0 + (m >> 1), BHY - 1);
      acc ^= *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(coef) + (lane & 7) * 16 + (unsigned)((y * BWY + x) * 128));
    }
  }
  const int bx = lane & 15, by = wave * 4 + (lane >> 4);
  const int X0 = (tx * 16 + bx) * 8, Y0 = (ty * 16 + by) * 8;
  if (Y0 >= H) return;
  uint8_t *dst = out_all + frame * OUT_FRAME + (int64_t)Y0 * (W * 3) + X0 * 3;
#pragma unroll
  for (int l = 0; l < 8; l++) {
    const u32x4 a = acc + (uint32_t)l;
    const u32x2 b = {acc.x ^ (uint32_t)l, acc.w + (uint32_t)l};
    if (NT) {
      __builtin_nontemporal_store(a, reinterpret_cast<u32x4 *>(dst + l * (W * 3)));
      __builtin_nontemporal_store(b, reinterpret_cast<u32x2 *>(dst + l * (W * 3) + 16));
    } else {
      *reinterpret_cast<u32x4 *>(dst + l * (W * 3)) = a;
      *reinterpret_cast<u32x2 *>(dst + l * (W * 3) + 16) = b;
    }
  }
}


// The same with the tile shape, the order of the tiles and the two directions as parameters: TBX x TBY luma blocks per workgroup
// (TBX * TBY = 256: one block per lane; a line of a wave's stores is min(TBX, 64) * 24 contiguous bytes), MODE 0 = reads and
// writes, 1 = reads only (one dword per lane written), 2 = writes only; ORDER 0 = blockIdx is the tile, 1 = XCD-aware.
template <int TBX, int TBY, int MODE, bool NT>
__global__ __launch_bounds__(256, 4) void shape_kernel(const int16_t *__restrict__ coef_all, uint8_t *__restrict__ out_all, int order)
{
  static_assert(TBX * TBY == 256, "one luma block per lane");
  constexpr int TXN = W / (8 * TBX), TYN = (H / 8 + TBY - 1) / TBY, CGX = TBX / 2 + 2, CGY = TBY / 2 + 2, NC = CGX * CGY;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned total = TXN * TYN * FRAMES;
  unsigned logical = blockIdx.x;
  const int order_in = order;
  order = order < 0 ? -order : order;
  if (order == 1) {
    const unsigned b = blockIdx.x, q = total >> 3, r = total & 7, x = b & 7, i = b >> 3;
    logical = x * q + min(x, r) + i;
  } else if (order >= 2) { // XCD x takes runs of `order` consecutive tiles, every eighth run (60 = a tile row of the 16 x 16 shape)
    const unsigned b = blockIdx.x, x = b & 7, i = b >> 3, rl = (unsigned)order;
    logical = ((i / rl) * 8 + x) * rl + i % rl;
    if (logical >= total) return;
  }
  const int frame = logical / (TXN * TYN), tile = logical - frame * (TXN * TYN);
  const int ty = tile / TXN, tx = tile - ty * TXN;
  const int16_t *coef = coef_all + frame * COEF_FRAME;
  const int64_t off_cb = (int64_t)BWY * BHY * 64, off_cr = off_cb + (int64_t)BWC * BHC * 64;
  u32x4 acc = {(uint32_t)tid, 0, 0, 0};
  if (MODE != 2) {
    { // chroma grid incl. halo, both planes: 2 * NC blocks over 4 waves, 8 blocks per wave-instruction
      const int gx0 = tx * (TBX / 2) - 1, gy0 = ty * (TBY / 2) - 1;
      for (int i0 = wave * 8 + (lane >> 3); i0 < 2 * NC; i0 += 32) {
        const int pl = i0 >= NC, i = i0 - pl * NC, y = i / CGX, x = i - y * CGX;
        const int gx = min(max(gx0 + x, 0), BWC - 1), gy = min(max(gy0 + y, 0), BHC - 1);
        acc ^= *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(coef + (pl ? off_cr : off_cb)) + (lane & 7) * 16 + (unsigned)((gy * BWC + gx) * 128));
      }
    }
    __syncthreads();
    { // luma: 256 blocks, 8 adjacent blocks per wave-instruction
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int b = wave * 64 + m * 8 + (lane >> 3), bx = b % TBX, by = b / TBX;
        const int x = min(tx * TBX + bx, BWY - 1), y = min(ty * TBY + by, BHY - 1);
        acc ^= *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(coef) + (lane & 7) * 16 + (unsigned)((y * BWY + x) * 128));
      }
    }
  }
  const int bx = tid % TBX, by = tid / TBX;
  const int X0 = (tx * TBX + bx) * 8, Y0 = (ty * TBY + by) * 8;
  if (MODE == 3) {
    // staged stores (TBX = 16): a wave's four block rows of sixteen 24-byte pieces are four 384-byte line segments per line number;
    // they go through 1.5 KB of LDS and leave as 96 chunks of 16 contiguous bytes -- every store instruction writes whole
    // 16-byte-aligned runs (lanes 0..23 the first segment, 24..47 the second, ...), none of them a piece of a lane's 24 bytes
    static_assert(MODE != 3 || TBX == 16, "staged stores: sixteen blocks per row");
    __shared__ __attribute__((aligned(16))) uint32_t stg[4][384];
    uint32_t *sw = stg[wave];
    const int c0 = lane, c1 = 64 + lane;
    const int Yw = (ty * TBY + wave * 4) * 8, Xw = tx * TBX * 8;
    uint8_t *wbase = out_all + frame * OUT_FRAME + (int64_t)Yw * (W * 3) + Xw * 3;
    const unsigned o0 = (unsigned)(c0 / 24) * 8u * (W * 3) + (unsigned)(c0 % 24) * 16u, o1 = (unsigned)(c1 / 24) * 8u * (W * 3) + (unsigned)(c1 % 24) * 16u;
    const bool ok0 = Yw + (c0 / 24) * 8 < H, ok1 = lane < 32 && Yw + (c1 / 24) * 8 < H;
#pragma unroll
    for (int l = 0; l < 8; l++) {
      const u32x4 a = acc + (uint32_t)l;
      const u32x2 b = {acc.x ^ (uint32_t)l, acc.w + (uint32_t)l};
      u32x2 *pw = reinterpret_cast<u32x2 *>(sw + lane * 6);
      pw[0] = u32x2{a.x, a.y}; pw[1] = u32x2{a.z, a.w}; pw[2] = b;
      __builtin_amdgcn_wave_barrier();
      const u32x4 v0 = *reinterpret_cast<const u32x4 *>(sw + c0 * 4);
      const u32x4 v1 = *reinterpret_cast<const u32x4 *>(sw + (lane < 32 ? c1 : c0) * 4);
      __builtin_amdgcn_wave_barrier();
      if (NT) {
        if (ok0) __builtin_nontemporal_store(v0, reinterpret_cast<u32x4 *>(wbase + o0 + (unsigned)l * (W * 3)));
        if (ok1) __builtin_nontemporal_store(v1, reinterpret_cast<u32x4 *>(wbase + o1 + (unsigned)l * (W * 3)));
      } else {
        if (ok0) *reinterpret_cast<u32x4 *>(wbase + o0 + (unsigned)l * (W * 3)) = v0;
        if (ok1) *reinterpret_cast<u32x4 *>(wbase + o1 + (unsigned)l * (W * 3)) = v1;
      }
    }
    return;
  }
  if (Y0 >= H) return;
  uint8_t *dst = out_all + frame * OUT_FRAME + (int64_t)Y0 * (W * 3) + X0 * 3;
  if (MODE == 1) {
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *reinterpret_cast<uint32_t *>(dst) = acc.x; // (never: keeps the loads alive)
    return;
  }
  if (order_in < 0) {
    // paired stores (order < 0 selects them; the tile order is -order): the 48 bytes of two neighbouring lanes leave as three
    // 16-byte-aligned dwordx4 stores -- every lane one (even lane: bytes 0-15 of the pair, odd lane: bytes 32-47), the even
    // lane a second one for bytes 16-31 (its own last 8 and the odd lane's first 8, which two v_mov_dpp would fetch)
    uint8_t *pair = dst - (tid & 1) * 24; // start of the pair's 48 bytes
#pragma unroll
    for (int l = 0; l < 8; l++) {
      const u32x4 a = acc + (uint32_t)l;
      const u32x4 b = {acc.x ^ (uint32_t)l, acc.w + (uint32_t)l, (uint32_t)__shfl_xor((int)acc.y, 1), (uint32_t)__shfl_xor((int)acc.z, 1)};
      __builtin_nontemporal_store(a, reinterpret_cast<u32x4 *>(pair + l * (W * 3) + (tid & 1) * 32));
      if (!(tid & 1)) __builtin_nontemporal_store(b, reinterpret_cast<u32x4 *>(pair + l * (W * 3) + 16));
    }
    return;
  }
#pragma unroll
  for (int l = 0; l < 8; l++) {
    const u32x4 a = acc + (uint32_t)l;
    const u32x2 b = {acc.x ^ (uint32_t)l, acc.w + (uint32_t)l};
    if (NT) {
      __builtin_nontemporal_store(a, reinterpret_cast<u32x4 *>(dst + l * (W * 3)));
      __builtin_nontemporal_store(b, reinterpret_cast<u32x2 *>(dst + l * (W * 3) + 16));
    } else {
      *reinterpret_cast<u32x4 *>(dst + l * (W * 3)) = a;
      *reinterpret_cast<u32x2 *>(dst + l * (W * 3) + 16) = b;
    }
  }
}

template <class F> static float time_ms(F launch, int reps)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 60; i++) launch(); // clock settling
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char **argv)
{
  const bool brief = argc > 1 && !strcmp(argv[1], "--brief"); // bench.py: one JSON line -- the ideal copy and the kernel's own pattern
  const int64_t coef_bytes = COEF_FRAME * 2 * FRAMES, out_bytes = OUT_FRAME * FRAMES;
  int16_t *coef;
  uint8_t *out;
  if (hipMalloc(&coef, coef_bytes + 4096) != hipSuccess || hipMalloc(&out, out_bytes + 4096) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(coef, 1, coef_bytes);
  hipMemset(out, 0, out_bytes);
  const double gb = (double)(coef_bytes + out_bytes) * 1e-9;
  const int tiles = TX * TY * FRAMES;
  const int LDS = 40 * 1024; // dynamic LDS nobody touches: four workgroups per CU, the fused kernel's occupancy
  if (argc > 1 && !strcmp(argv[1], "--staged")) { // the fused kernel's pattern against the same with LDS-staged, fully coalesced stores
    const int order = TX, n = (tiles + 8 * order - 1) / (8 * order) * (8 * order);
    for (int round = 0; round < 3; round++) {
      const float p = time_ms([&] { hipLaunchKernelGGL((shape_kernel<16, 16, 0, true>), dim3(n), dim3(256), LDS, 0, coef, out, order); }, 100);
      const float sn = time_ms([&] { hipLaunchKernelGGL((shape_kernel<16, 16, 3, true>), dim3(n), dim3(256), LDS - 6144, 0, coef, out, order); }, 100);
      const float st = time_ms([&] { hipLaunchKernelGGL((shape_kernel<16, 16, 3, false>), dim3(n), dim3(256), LDS - 6144, 0, coef, out, order); }, 100);
      const float pt = time_ms([&] { hipLaunchKernelGGL((shape_kernel<16, 16, 0, false>), dim3(n), dim3(256), LDS, 0, coef, out, order); }, 100);
      printf("round %d: 24-byte pieces nt %.4f ms (%.3f of 8)  temporal %.4f (%.3f) | staged 16-byte chunks nt %.4f ms (%.3f)  temporal %.4f (%.3f)\n", round, p, gb / p / 8.0, pt,
             gb / pt / 8.0, sn, gb / sn / 8.0, st, gb / st / 8.0);
    }
    return hipDeviceSynchronize() != hipSuccess;
  }
  if (brief) {
    float copy_ms = 1e9f;
    for (int blocks : {32768, 131072, 388800}) {
      const float ms = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, (const u32x4 *)coef, (u32x4 *)out, coef_bytes / 16, out_bytes / 16); }, 100);
      if (ms < copy_ms) copy_ms = ms;
    }
    const int order = TX; // the fused kernels' tile order: runs of one tile row per XCD
    const int n = (tiles + 8 * order - 1) / (8 * order) * (8 * order);
    // (the kernel's stores since round 6's end: 16 contiguous bytes per lane through LDS, MODE 3; "pieces" = the 24-byte pieces before that)
    const float pieces_ms = time_ms([&] { hipLaunchKernelGGL((shape_kernel<16, 16, 0, true>), dim3(n), dim3(256), LDS, 0, coef, out, order); }, 100);
    const float pat_ms = time_ms([&] { hipLaunchKernelGGL((shape_kernel<16, 16, 3, true>), dim3(n), dim3(256), LDS - 6144, 0, coef, out, order); }, 100);
    if (hipDeviceSynchronize() != hipSuccess) { printf("{\"error\": \"device\"}\n"); return 1; }
    printf("{\"frames\": %d, \"bytes_per_launch\": %lld, \"copy_ms\": %.4f, \"copy_frac\": %.4f, \"pattern_ms\": %.4f, \"pattern_frac\": %.4f, \"pieces_ms\": %.4f, \"pieces_frac\": %.4f}\n",
           FRAMES, (long long)(coef_bytes + out_bytes), copy_ms, gb / copy_ms / 8.0, pat_ms, gb / pat_ms / 8.0, pieces_ms, gb / pieces_ms / 8.0);
    return 0;
  }
  printf("8 x 8K 4:2:0 frames: %.1f MB in + %.1f MB out per launch\n", coef_bytes * 1e-6, out_bytes * 1e-6);
  for (int blocks : {2048, 8192, 32768, 131072, 388800}) {
    const float ms = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, (const u32x4 *)coef, (u32x4 *)out, coef_bytes / 16, out_bytes / 16); }, 200);
    printf("copy      %6d workgroups: %.4f ms/launch  %.2f TB/s  (%.3f of 8)\n", blocks, ms, gb / ms, gb / ms / 8.0);
  }
  {
    const float ms = time_ms([&] { hipLaunchKernelGGL((pattern_kernel<true>), dim3(tiles), dim3(256), LDS, 0, coef, out); }, 200);
    printf("pattern   %6d workgroups: %.4f ms/launch  %.2f TB/s  (%.3f of 8)  non-temporal stores\n", tiles, ms, gb / ms, gb / ms / 8.0);
  }
  {
    const float ms = time_ms([&] { hipLaunchKernelGGL((pattern_kernel<false>), dim3(tiles), dim3(256), LDS, 0, coef, out); }, 200);
    printf("pattern-t %6d workgroups: %.4f ms/launch  %.2f TB/s  (%.3f of 8)  temporal stores\n", tiles, ms, gb / ms, gb / ms / 8.0);
  }

  // shapes, orders, directions
  auto shape = [&](const char *name, auto kernel, int tbx, int tby, int order, double bytes) {
    int n = (W / (8 * tbx)) * ((H / 8 + tby - 1) / tby) * FRAMES;
    const int ao = order < 0 ? -order : order;
    if (ao >= 2) n = (n + 8 * ao - 1) / (8 * ao) * (8 * ao);
    const float ms = time_ms([&] { hipLaunchKernelGGL(kernel, dim3(n), dim3(256), LDS, 0, coef, out, order); }, 200);
    printf("%-28s %2d x %2d blocks, order %d: %.4f ms/launch  %.2f TB/s  (%.3f of 8)\n", name, tbx, tby, order, ms, bytes * 1e-9 / ms, bytes * 1e-9 / ms / 8.0);
  };
  const double both = (double)(coef_bytes + out_bytes), rd = (double)coef_bytes, wr = (double)out_bytes;
  for (int order : {4, 15, 30, 60, 120, 240, 480}) { // run lengths (16 x 16: 60 tiles per row, 34 rows per frame)
    shape("read+write nt", shape_kernel<16, 16, 0, true>, 16, 16, order, both);
    shape("read+write temporal", shape_kernel<16, 16, 0, false>, 16, 16, order, both);
  }
  for (int rep = 0; rep < 3; rep++) { // the kernel's stores against paired 16-byte-aligned ones (negative order), tile-row order
    shape("read+write nt, 24-byte pieces", shape_kernel<16, 16, 0, true>, 16, 16, 60, both);
    shape("read+write nt, paired 16 B", shape_kernel<16, 16, 0, true>, 16, 16, -60, both);
    shape("write only nt, 24-byte pieces", shape_kernel<16, 16, 2, true>, 16, 16, 60, wr);
    shape("write only nt, paired 16 B", shape_kernel<16, 16, 2, true>, 16, 16, -60, wr);
  }
  for (int order : {30, 60, 120}) { // 32 x 8: 30 tiles per row, 68 rows per frame
    shape("read+write nt", shape_kernel<32, 8, 0, true>, 32, 8, order, both);
    shape("read only", shape_kernel<16, 16, 1, true>, 16, 16, order, rd);
    shape("write only nt", shape_kernel<16, 16, 2, true>, 16, 16, order, wr);
  }
  for (int order = 0; order < 2; order++) {
    shape("read+write nt", shape_kernel<16, 16, 0, true>, 16, 16, order, both);
    shape("read+write temporal", shape_kernel<16, 16, 0, false>, 16, 16, order, both);
    shape("read+write nt", shape_kernel<32, 8, 0, true>, 32, 8, order, both);
    shape("read+write temporal", shape_kernel<32, 8, 0, false>, 32, 8, order, both);
    shape("read+write nt", shape_kernel<64, 4, 0, true>, 64, 4, order, both);
    shape("read+write temporal", shape_kernel<64, 4, 0, false>, 64, 4, order, both);
    shape("read only", shape_kernel<16, 16, 1, true>, 16, 16, order, rd);
    shape("read only", shape_kernel<64, 4, 1, true>, 64, 4, order, rd);
    shape("write only nt", shape_kernel<16, 16, 2, true>, 16, 16, order, wr);
    shape("write only temporal", shape_kernel<16, 16, 2, false>, 16, 16, order, wr);
    shape("write only nt", shape_kernel<64, 4, 2, true>, 64, 4, order, wr);
    shape("write only temporal", shape_kernel<64, 4, 2, false>, 64, 4, order, wr);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
  return 0;
}
