// Write-pattern microbenchmark, second take: the tile patterns of wpat.hip with compile-time tile shapes (no integer
// division in the store loop -- wpat's tile kernel computes e / TW per store and may be VALU-bound rather than
// memory-bound), plain and nontemporal stores, and the frame loop innermost or outermost.
//   hipcc --offload-arch=gfx950 -O3 tools/wpat2.hip -Iinclude -Lmono_dataset_code_amd -lmdc_hip -Wl,-rpath,'$ORIGIN/../../mono_dataset_code_amd' -o tools/bin/wpat2
//   tools/bin/wpat2 [OW OH frames]          WPAT_PLACED=1: the buffer comes from the product's allocator (striped over the device's memory classes)
#include <hip/hip_runtime.h>

#include "mdc_hip.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ void st(float* p, float v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <bool NT>
__global__ void w_lin(float* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += s) st<NT>(p + i, (float)i);
}

// tile (tx, ty), frames [g*fpb, (g+1)*fpb); thread t: column t % TW, rows t / TW + k * (NTH / TW)
template <int TW, int TH, int NTH, bool NT>
__global__ __launch_bounds__(NTH) void w_tile(float* __restrict__ out, int OW, int OH, int tiles_x, int nframes, int fpb, int order_xcd, int ntiles) {
  int tile = blockIdx.x;
  if (order_xcd) {
    const int per = (ntiles + 7) / 8;
    tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (tile >= ntiles) return;
  }
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  constexpr int K = TW * TH / NTH;  // elements per thread: e = k * NTH + t -> row e / TW, column e % TW (TW a power of two)
  const int f0 = blockIdx.y * fpb, f1 = min(nframes, f0 + fpb);
  const size_t frame = (size_t)OW * OH;
  float* base = out + (size_t)f0 * frame + (size_t)(ty * TH) * OW + tx * TW;
  int off[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int e = k * NTH + threadIdx.x, r = e / TW, c = e % TW;
    off[k] = (ty * TH + r < OH && tx * TW + c < OW) ? r * OW + c : -1;
  }
  for (int f = f0; f < f1; f++, base += frame) {
#pragma unroll
    for (int k = 0; k < K; k++)
      if (off[k] >= 0) st<NT>(base + off[k], (float)k);
  }
}

int main(int argc, char** argv) {
  const int OW = argc > 1 ? atoi(argv[1]) : 640, OH = argc > 2 ? atoi(argv[2]) : 480, NF = argc > 3 ? atoi(argv[3]) : 4096;
  const size_t n = (size_t)OW * OH * NF;
  float* d;
  if (getenv("WPAT_PLACED")) {
    mdc_ctx* ctx = nullptr;
    if (mdc_create(0, &ctx) != MDC_OK) { printf("mdc_create: %s\n", mdc_last_error(nullptr)); return 1; }
    size_t bytes[1] = {n * 4};
    mdc_striped_set set;
    if (mdc_alloc_striped_set_device(ctx, 1, bytes, nullptr, &set) != MDC_OK) { printf("allocator: %s\n", mdc_last_error(ctx)); return 1; }
    d = (float*)set.d_ptr[0];
    printf("buffer from the product's allocator: %s\n", set.note);
  } else {
    CK(hipMalloc(&d, n * 4));
  }
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
  for (int i = 0; i < 20; i++) w_lin<true><<<8192, 256>>>(d, n);
  CK(hipDeviceSynchronize());
  printf("write-only patterns (compile-time tile shapes), %d x %d f32, %d frames (%.2f GB)\n", OW, OH, NF, n * 4 / 1e9);
  for (int blocks : {2048, 8192, 65536}) {
    float t = timeit([&] { w_lin<true><<<blocks, 256>>>(d, n); });
    printf("linear grid-stride nt,    %6d workgroups of 256 : %.4f ms  %.2f TB/s\n", blocks, t, n * 4 / t / 1e9);
    t = timeit([&] { w_lin<false><<<blocks, 256>>>(d, n); });
    printf("linear grid-stride plain, %6d workgroups of 256 : %.4f ms  %.2f TB/s\n", blocks, t, n * 4 / t / 1e9);
  }
#define RUN(TW_, TH_, NT_)                                                                                                          \
  do {                                                                                                                              \
    const int tx = (OW + TW_ - 1) / TW_, ty = (OH + TH_ - 1) / TH_;                                                                 \
    for (int fpb : {8, 32, 64})                                                                                                     \
      for (int xcd : {0, 1})                                                                                                        \
        for (int nt : {0, 1}) {                                                                                                     \
          dim3 grid(xcd ? ((tx * ty + 7) / 8) * 8 : tx * ty, (NF + fpb - 1) / fpb);                                                 \
          const float t = nt ? timeit([&] { w_tile<TW_, TH_, NT_, true><<<grid, NT_>>>(d, OW, OH, tx, NF, fpb, xcd, tx * ty); })   \
                             : timeit([&] { w_tile<TW_, TH_, NT_, false><<<grid, NT_>>>(d, OW, OH, tx, NF, fpb, xcd, tx * ty); }); \
          printf("tile %4d x %2d, %4d threads, fpb %2d, %s, %s: %.4f ms  %.2f TB/s\n", TW_, TH_, NT_, fpb, xcd ? "XCD bands" : "plain    ", \
                 nt ? "nt   " : "plain", t, n * 4 / t / 1e9);                                                                       \
        }                                                                                                                           \
  } while (0)
  RUN(128, 32, 1024);
  RUN(128, 16, 512);
  RUN(128, 8, 256);
  RUN(128, 8, 64);
  RUN(640, 4, 640);
  RUN(640, 2, 640);
  RUN(640, 8, 640);
  return 0;
}
