// Read-pattern microbenchmark: how fast does this memory system serve the SOURCE WINDOWS of a remap when the source region
// (863 x 730 bytes of a 1280 x 1024 u8 frame, the bench camera's bounding box) is cut into windows of different shapes?
// No stores (one dword per workgroup at the end), no arithmetic.  A workgroup reads its window of every frame of its frame
// group: piece = `wb` bytes (multiple of 16) per source row, `rows` rows, 16 bytes per lane, lane = chunk (row-major), plain
// (L2-allocating) loads like the kernels' LDS-DMA.  Windows are laid on a grid with steps (sx, sy) -- neighbouring windows
// overlap like the real ones (sx < wb: shared bytes; unaligned to 128-byte lines unless sx % 128 == 0).
//   tools/bin/rpat [frames]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int IW = 1280, IH = 1024;
constexpr int BX0 = 208, BY0 = 144, BW = 864, BH = 736;  // the source bounding box (rounded)

__global__ void r_lin(const u32x4* __restrict__ p, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n; i += s) { u32x4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
// only the bounding box of every frame, row by row (what an ideal reader of the box would do)
__global__ void r_box(const uint8_t* __restrict__ in, uint32_t* out, int nframes) {
  const int cpr = BW / 16, total = cpr * BH;
  uint32_t acc = 0;
  for (int f = blockIdx.y; f < nframes; f += gridDim.y)
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
      const int r = c / cpr, k = c - r * cpr;
      u32x4 v = *reinterpret_cast<const u32x4*>(in + (size_t)f * IW * IH + (size_t)(BY0 + r) * IW + BX0 + k * 16);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void r_win(const uint8_t* __restrict__ in, uint32_t* out, int wb, int rows, int sx, int sy, int tiles_x, int ntiles, int nframes, int fpb, int xcd) {
  int tile = blockIdx.x;
  if (xcd) {
    const int per = (gridDim.x + 7) / 8;
    tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
  }
  if (tile >= ntiles) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int cpr = wb / 16, total = cpr * rows;
  const int x0 = (BX0 + tx * sx) & ~15, y0 = BY0 + ty * sy;
  const int f0 = blockIdx.y * fpb, f1 = min(nframes, f0 + fpb);
  uint32_t acc = 0;
  for (int f = f0; f < f1; f++) {
    const uint8_t* base = in + (size_t)f * IW * IH;
    for (int c = threadIdx.x; c < total; c += blockDim.x) {
      const int r = c / cpr, k = c - r * cpr;
      const int y = min(y0 + r, IH - 1), x = min(x0 + k * 16, IW - 16);
      u32x4 v = *reinterpret_cast<const u32x4*>(base + (size_t)y * IW + x);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// The same windows through the kernels' actual load path: buffer_load_dwordx4 ... lds (LDS-DMA, 16 bytes per lane landing in
// LDS), DEPTH frames in flight per wave (s_waitcnt vmcnt(DEPTH-1) after each issue), or through VGPRs (MODE 1).
typedef __attribute__((address_space(3))) void* lds_void_ptr;
template <int MODE, int DEPTH>
__global__ void r_win_dma(const uint8_t* __restrict__ in, uint32_t* out, int wb, int rows, int sx, int sy, int tiles_x, int ntiles, int nframes, int fpb) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int per = (gridDim.x + 7) / 8;
  const int tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if (tile >= ntiles) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int cpr = wb / 16, total = cpr * rows;
  const int x0 = (BX0 + tx * sx) & ~15, y0 = BY0 + ty * sy;
  const int f0 = blockIdx.y * fpb, f1 = min(nframes, f0 + fpb);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t acc = 0;
  const int c = threadIdx.x;
  const int r = c / cpr, k = c - r * cpr;
  const int y = min(y0 + r, IH - 1), x = min(x0 + k * 16, IW - 16);
  const uint32_t off = c < total ? (uint32_t)(y * IW + x) : 0xfffffff0u;
  for (int f = f0; f < f1; f++) {
    const uint8_t* base = in + (size_t)f * IW * IH;
    if (MODE == 0) {
      const unsigned long long v = (unsigned long long)base;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, IW * IH, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(smem + ((f - f0) % DEPTH) * 8192 + wave * 1024), 16, off, 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    } else {
      if (c < total) {
        u32x4 v = *reinterpret_cast<const u32x4*>(base + off);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (MODE == 0) acc = *reinterpret_cast<uint32_t*>(smem + threadIdx.x * 4);
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const int NF = argc > 1 ? atoi(argv[1]) : 1024;
  const size_t n = (size_t)IW * IH * NF;
  uint8_t* d;
  uint32_t* o;
  CK(hipMalloc(&d, n));
  CK(hipMalloc(&o, 64));
  CK(hipMemset(d, 1, n));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](auto launch) {
    for (int i = 0; i < 3; i++) launch();
    std::vector<float> ms;
    for (int rep = 0; rep < 7; rep++) {
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[3];
  };
  for (int i = 0; i < 20; i++) r_lin<<<8192, 256>>>((const u32x4*)d, o, n / 16);
  CK(hipDeviceSynchronize());
  printf("read-only patterns, %d frames of 1280 x 1024 u8 (%.2f GB); bounding box %d x %d = %.3f MB per frame\n", NF, n / 1e9, BW, BH, BW * BH / 1e6);
  { const float t = timeit([&] { r_lin<<<8192, 256>>>((const u32x4*)d, o, n / 16); });
    printf("%-66s: %.4f ms  %.2f TB/s\n", "linear, whole frames", t, n / t / 1e9); }
  { const float t = timeit([&] { r_box<<<dim3(16, 256), 256>>>(d, o, NF); });
    printf("%-66s: %.4f ms  %.2f TB/s of box bytes\n", "bounding box only, row by row", t, (double)BW * BH * NF / t / 1e9); }
  struct Cfg { const char* what; int wb, rows, sx, sy, nt; };
  const Cfg cfgs[] = {
      {"strip-like   112 B x  8 rows, step  86 x  6 (1 wave)", 112, 8, 86, 6, 64},
      {"strip-like   112 B x  8 rows, step  86 x  6 (4 windows per WG: 256 thr reading 4 adjacent)", 112 * 4 - 3 * 16, 8, 86 * 4, 6, 256},
      {"aligned      128 B x  8 rows, step 128 x  6", 128, 8, 128, 6, 64},
      {"aligned      128 B x  8 rows, step 128 x  8 (no overlap)", 128, 8, 128, 8, 64},
      {"direct-like  112 B x 14 rows, step  86 x 12 (128x16 outputs at scale 1)", 112, 14, 86, 12, 128},
      {"             112 B x 26 rows, step  86 x 24", 112, 26, 86, 24, 192},
      {"headline     192 B x 26 rows, step 172 x 24 (128x16 outputs at 1.5x)", 192, 26, 172, 24, 320},
      {"             256 B x 26 rows, step 256 x 24 (aligned)", 256, 26, 256, 24, 448},
      {"             192 B x  8 rows, step 172 x  6", 192, 8, 172, 6, 128},
      {"             368 B x  8 rows, step 344 x  6", 368, 8, 344, 6, 192},
      {"full width   864 B x  8 rows, step 864 x  6", 864, 8, 864, 6, 448},
      {"full width   864 B x 14 rows, step 864 x 12", 864, 14, 864, 12, 768},
      {"full width   864 B x  2 rows, step 864 x  2 (no overlap)", 864, 2, 864, 2, 128},
  };
  for (const Cfg& c : cfgs) {
    const int tx = (BW + c.sx - 1) / c.sx, ty = (BH + c.sy - 1) / c.sy, nt = tx * ty;
    const double req = (double)c.wb * c.rows * nt * NF;
    for (int fpb : {8, 64})
      for (int xcd : {0, 1}) {
        dim3 grid((nt + 7) / 8 * 8, (NF + fpb - 1) / fpb);
        const float t = timeit([&] { r_win<<<grid, c.nt>>>(d, o, c.wb, c.rows, c.sx, c.sy, tx, nt, NF, fpb, xcd); });
        printf("%-66s fpb %2d %s: %.4f ms  requested %.2f MB/frame = %.2f TB/s  (%d windows, %d threads)\n", c.what, fpb, xcd ? "bands" : "plain", t,
               req / NF / 1e6, req / t / 1e9, nt, c.nt);
      }
  }
  printf("---- load path and depth (XCD bands, fpb 32): LDS-DMA dwordx4 with D frames in flight per wave vs loads into VGPRs\n");
  for (const Cfg& c : cfgs) {
    if (c.nt > 512) continue;
    const int tx = (BW + c.sx - 1) / c.sx, ty = (BH + c.sy - 1) / c.sy, nt = tx * ty;
    const double req = (double)c.wb * c.rows * nt * NF;
    const int fpb = 32;
    const int thr = (c.wb / 16 * c.rows + 63) / 64 * 64;
    dim3 grid((nt + 7) / 8 * 8, (NF + fpb - 1) / fpb);
    const size_t lds = 8 * 8192;
    float t[6];
    t[0] = timeit([&] { r_win_dma<0, 1><<<grid, thr, lds>>>(d, o, c.wb, c.rows, c.sx, c.sy, tx, nt, NF, fpb); });
    t[1] = timeit([&] { r_win_dma<0, 2><<<grid, thr, lds>>>(d, o, c.wb, c.rows, c.sx, c.sy, tx, nt, NF, fpb); });
    t[2] = timeit([&] { r_win_dma<0, 4><<<grid, thr, lds>>>(d, o, c.wb, c.rows, c.sx, c.sy, tx, nt, NF, fpb); });
    t[3] = timeit([&] { r_win_dma<0, 8><<<grid, thr, lds>>>(d, o, c.wb, c.rows, c.sx, c.sy, tx, nt, NF, fpb); });
    t[4] = timeit([&] { r_win_dma<1, 1><<<grid, thr, 0>>>(d, o, c.wb, c.rows, c.sx, c.sy, tx, nt, NF, fpb); });
    t[5] = timeit([&] { r_win_dma<1, 8><<<grid, thr, 0>>>(d, o, c.wb, c.rows, c.sx, c.sy, tx, nt, NF, fpb); });
    printf("%-66s (%3d thr): LDS-DMA D=1 %.4f  D=2 %.4f  D=4 %.4f  D=8 %.4f ms | VGPR loads waited %.4f  free-running %.4f ms | requested %.2f MB/frame\n", c.what, thr, t[0], t[1],
           t[2], t[3], t[4], t[5], req / NF / 1e6);
  }
  return 0;
}
