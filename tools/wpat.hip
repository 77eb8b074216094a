// Write-pattern microbenchmark: how fast does this memory system take the f32 output of a remap when the frame is cut
// into workgroup tiles of different shapes?  No reads, no arithmetic: every workgroup writes its TW x TH patch of every
// frame of its frame group with wave-contiguous nontemporal dword stores (256 bytes per wave instruction), exactly like
// the remap kernels do.  Geometry OW x OH = the rectified frame (default 1280 x 1024, config 5), frames = 1024.
//   hipcc --offload-arch=gfx950 -O3 tools/wpat.hip -o tools/bin/wpat && tools/bin/wpat [OW OH frames]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void w_lin(float* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += s) __builtin_nontemporal_store((float)i, p + i);
}

// tile (tx, ty) of frames [g*fpb, (g+1)*fpb): element e = k*NT + t -> (row e / TW, col e % TW)
__global__ void w_tile(float* __restrict__ out, int OW, int OH, int TW, int TH, int tiles_x, int nframes, int fpb, int order_xcd) {
  int tile = blockIdx.x;
  if (order_xcd) {  // contiguous band of tiles per XCD (block b runs on XCD b % 8)
    const int per = (gridDim.x + 7) / 8;
    tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (tile >= (int)gridDim.x) return;
  }
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int f0 = blockIdx.y * fpb, f1 = min(nframes, f0 + fpb);
  const int elems = TW * TH;
  const size_t frame = (size_t)OW * OH;
  float* base = out + (size_t)f0 * frame + (size_t)(ty * TH) * OW + tx * TW;
  for (int f = f0; f < f1; f++, base += frame)
    for (int e = threadIdx.x; e < elems; e += blockDim.x) {
      const int r = e / TW, c = e - r * TW;
      if (ty * TH + r < OH && tx * TW + c < OW) __builtin_nontemporal_store((float)e, base + (size_t)r * OW + c);
    }
}

int main(int argc, char** argv) {
  const int OW = argc > 1 ? atoi(argv[1]) : 1280, OH = argc > 2 ? atoi(argv[2]) : 1024, NF = argc > 3 ? atoi(argv[3]) : 1024;
  const size_t n = (size_t)OW * OH * NF;
  float* d;
  CK(hipMalloc(&d, n * 4));
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
  for (int i = 0; i < 20; i++) w_lin<<<8192, 256>>>(d, n);  // clock ramp
  CK(hipDeviceSynchronize());
  printf("write-only patterns, %d x %d f32, %d frames (%.2f GB)\n", OW, OH, NF, n * 4 / 1e9);
  for (int blocks : {2048, 8192, 65536}) {
    const float t = timeit([&] { w_lin<<<blocks, 256>>>(d, n); });
    printf("linear grid-stride, %6d workgroups of 256              : %.4f ms  %.2f TB/s\n", blocks, t, n * 4 / t / 1e9);
  }
  struct Cfg { int tw, th, nt; };
  const Cfg cfgs[] = {{128, 16, 512}, {128, 32, 512}, {128, 32, 1024}, {64, 32, 512}, {512, 8, 256}, {256, 16, 512}, {640, 8, 320}, {OW, 8, 640},
                      {OW, 8, 256}, {OW, 4, 320}, {OW, 16, 1024}, {OW, 16, 512}, {OW, 2, 256}, {OW, 32, 1024}, {OW, 1, 256}};
  for (const Cfg& c : cfgs) {
    if (OW % c.tw) continue;
    const int tx = OW / c.tw, ty = (OH + c.th - 1) / c.th;
    for (int fpb : {8, 32, 64})
      for (int xcd : {0, 1}) {
        dim3 grid(tx * ty, (NF + fpb - 1) / fpb);
        const float t = timeit([&] { w_tile<<<grid, c.nt>>>(d, OW, OH, c.tw, c.th, tx, NF, fpb, xcd); });
        printf("tile %4d x %2d, %4d threads (%2d elems/thread), fpb %2d, %s: %.4f ms  %.2f TB/s\n", c.tw, c.th, c.nt, c.tw * c.th / c.nt, fpb,
               xcd ? "XCD bands" : "plain    ", t, n * 4 / t / 1e9);
      }
  }
  return 0;
}
