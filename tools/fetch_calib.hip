// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// widths this repo uses (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide
// coalesced reads by 2x; "calibrate on a known byte count in your own access
// pattern").  Each kernel moves a KNOWN number of bytes of a 1 GiB buffer
// (> the 256 MiB Infinity Cache); run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- ./fetch_calib
// and compare the counter with the printed byte counts.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void read16_nt(const u32x4* __restrict__ p, uint32_t* out, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n16; i += stride) { u32x4 v = __builtin_nontemporal_load(p + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read16_plain(const u32x4* __restrict__ p, uint32_t* out, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n16; i += stride) { u32x4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read4_nt(const uint32_t* __restrict__ p, uint32_t* out, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n4; i += stride) acc ^= __builtin_nontemporal_load(p + i);
  if (acc == 0x12345678u) out[0] = acc;
}
// window-like: each group of 8 lanes reads one 128-byte piece; pieces are 1280 bytes apart
// (one image row), i.e. only 10% of the bytes of the buffer are requested.
__global__ void read_rows128_nt(const u32x4* __restrict__ p, uint32_t* out, size_t n16) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; ; t += stride) {
    size_t i = (t >> 3) * 80 + (t & 7);  // 80 x 16 B = 1280 B per row
    if (i >= n16) break;
    u32x4 v = __builtin_nontemporal_load(p + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// same but the 128-byte piece starts 64 bytes into a 128-byte line (straddles two lines)
__global__ void read_rows128_straddle_nt(const u32x4* __restrict__ p, uint32_t* out, size_t n16) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; ; t += stride) {
    size_t i = (t >> 3) * 80 + (t & 7) + 4;
    if (i >= n16) break;
    u32x4 v = __builtin_nontemporal_load(p + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void write4_nt(float* __restrict__ p, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) __builtin_nontemporal_store((float)i, p + i);
}
__global__ void write16_nt(u32x4* __restrict__ p, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) { u32x4 v = {(uint32_t)i, 1, 2, 3}; __builtin_nontemporal_store(v, p + i); }
}
__global__ void copy16(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) b[i] = a[i];
}

#define T(name, bytes, ...)                                                          \
  do {                                                                               \
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                     \
    __VA_ARGS__; hipDeviceSynchronize();                                             \
    hipEventRecord(e0); __VA_ARGS__; hipEventRecord(e1); hipEventSynchronize(e1);    \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                      \
    printf("%-28s requested %12zu bytes  %8.3f ms  %8.1f GB/s\n", name, (size_t)(bytes), ms, (bytes) / ms / 1e6); \
  } while (0)

int main() {
  const size_t N = (size_t)1 << 30;
  void *a, *b; uint32_t* out;
  hipMalloc(&a, N); hipMalloc(&b, N); hipMalloc(&out, 4);
  hipMemset(a, 1, N); hipMemset(b, 2, N);
  const int G = 256 * 8, B = 256;
  T("read16_nt", N, read16_nt<<<G, B>>>((const u32x4*)a, out, N / 16));
  T("read16_plain", N, read16_plain<<<G, B>>>((const u32x4*)a, out, N / 16));
  T("read4_nt", N, read4_nt<<<G, B>>>((const uint32_t*)a, out, N / 4));
  T("read_rows128_nt", N / 10, read_rows128_nt<<<G, B>>>((const u32x4*)a, out, N / 16));
  T("read_rows128_straddle_nt", N / 10, read_rows128_straddle_nt<<<G, B>>>((const u32x4*)a, out, N / 16 - 8));
  T("write4_nt", N, write4_nt<<<G, B>>>((float*)b, N / 4));
  T("write16_nt", N, write16_nt<<<G, B>>>((u32x4*)b, N / 16));
  T("copy16 (r+w)", 2 * N, copy16<<<G, B>>>((const u32x4*)a, (u32x4*)b, N / 16));
  return 0;
}
