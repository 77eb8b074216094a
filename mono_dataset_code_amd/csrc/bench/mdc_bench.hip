// libmdc_bench.so -- measurement and test utilities (include/mdc_bench.h).  NOT part of the product ABI: nothing of the
// reference corresponds to these, the product libraries do not link or load this one.
//   mdcb_synth_frames_device : the synthetic sequence generator of SURVEY.md 8(d)
//   mdcb_ceiling_mix_device  : a linear read + write stream of given byte counts (the same-box yardstick of bench.py)
//   mdcb_alias_alloc / _free : a device range whose virtual pages map ONE physical chunk over and over (tools/mall_bracket.py)
#include "../../../include/mdc_bench.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
__global__ __launch_bounds__(256) void synth_kernel(uint8_t* __restrict__ out, long long first_pix, long long n,
                                                    uint32_t seed) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  uint32_t word = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t b = fmix32(seed + (uint32_t)(first_pix + i + k)) >> 24;
    word |= b << (8 * k);
  }
  if (i + 3 < n && ((reinterpret_cast<uintptr_t>(out + i) & 3) == 0)) *reinterpret_cast<uint32_t*>(out + i) = word;
  else
    for (int k = 0; k < 4 && i + k < n; k++) out[i + k] = (uint8_t)(word >> (8 * k));
}

// Bench utility (no arithmetic of the path): a LINEAR stream that reads n_r 16-byte chunks and writes n_w
// dwords, interleaved at that ratio -- the memory system's rate for the traffic MIX of a kernel without
// its access pattern.  bench.py runs it with the algorithmic byte counts of the benchmarked launch, in
// the same process on the same box, and reports the kernel's rate as a fraction of it.
// Stores are wave-contiguous dwords with the nontemporal hint (the fastest store form measured on this
// memory system, tools/hbm_mix.hip), loads wave-contiguous 16-byte.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// SPAN = false: grid-stride (thread t touches element t, t+T, ...); SPAN = true: every workgroup walks its own
// contiguous span of the write range (and of the read range), as a tiled kernel's workgroups do.
template <bool SPAN>
__global__ __launch_bounds__(256) void mix_ceiling_kernel(const u32x4* __restrict__ a, float* __restrict__ b,
                                                          unsigned long long n_r, unsigned long long n_w) {
  const unsigned long long G = gridDim.x, T = G * 256;
  const unsigned long long iters = SPAN ? (n_w + T - 1) / T : (n_w + T - 1) / T;
  // SPAN: block g owns writes [g*iters*256, (g+1)*iters*256) and reads [g*riters*256, ...)
  const unsigned long long riters = (n_r + T - 1) / T;
  const unsigned long long w0 = SPAN ? (unsigned long long)blockIdx.x * iters * 256 + threadIdx.x : (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long r0 = SPAN ? (unsigned long long)blockIdx.x * riters * 256 + threadIdx.x : (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long step = SPAN ? 256 : T;
  unsigned long long racc = 0, rk = r0, rdone = 0;
  uint32_t x = 0;
  for (unsigned long long it = 0; it < iters; it++) {
    racc += n_r;  // one chunk is read every n_w / n_r stores, the same iteration for every thread
    if (racc >= n_w) {
      racc -= n_w;
      if (rk < n_r && rdone < riters) {
        const u32x4 v = __builtin_nontemporal_load(a + rk);  // streamed once: the faster load form (tools/hbm_mix.hip)
        x ^= v.x ^ v.y ^ v.z ^ v.w;
      }
      rk += step;
      rdone++;
    }
    const unsigned long long i = w0 + it * step;
    if (i < n_w) __builtin_nontemporal_store(__uint_as_float(x & 0x3fffffffu), b + i);
  }
}

// A kernel that does nothing, under a name of its own: bench.py --markers launches it right before and right after its timed region, on
// the stream of the timed launches, so that a rocprofv3 kernel trace / counter collection of the run can be cut to the timed launches.
__global__ void mdcb_marker_kernel(int id, int* sink) {
  if (sink && id == -12345) *sink = id;
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (dev < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

extern "C" {

int mdcb_synth_frames_device(int device, uint8_t* d_out, int64_t first_frame, int64_t nframes, int npix, uint32_t seed, void* stream) {
  if (!d_out || nframes < 0 || npix <= 0) return -1;
  DeviceGuard dg(device);
  const long long n = (long long)nframes * npix;
  if (n <= 0) return 0;
  // a launch is limited to 2^32 threads in total: sequences beyond 2^34 pixels (a 50,000-frame one is 6.5e10) go in pieces
  const long long piece = 1ll << 32;  // pixels per launch = 2^30 threads
  for (long long at = 0; at < n; at += piece) {
    const long long m = n - at < piece ? n - at : piece;
    synth_kernel<<<ceil_div(m, 1024), 256, 0, (hipStream_t)stream>>>(d_out + at, first_frame * (long long)npix + at, m, seed);
    if (hipGetLastError() != hipSuccess) return -4;
  }
  return 0;
}

int mdcb_marker_device(int device, int id, void* stream) {
  DeviceGuard dg(device);
  mdcb_marker_kernel<<<1, 64, 0, (hipStream_t)stream>>>(id, nullptr);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int mdcb_ceiling_mix_device(int device, const void* d_read, int64_t read_bytes, float* d_write, int64_t write_bytes, int blocks, int span,
                            void* stream) {
  if (read_bytes < 0 || write_bytes < 0 || (read_bytes > 0 && !d_read) || (write_bytes > 0 && !d_write) || blocks <= 0 ||
      (reinterpret_cast<uintptr_t>(d_read) & 15) != 0)
    return -1;
  if (write_bytes <= 0) return 0;
  DeviceGuard dg(device);
  const unsigned long long n_r = (unsigned long long)(read_bytes / 16), n_w = (unsigned long long)(write_bytes / 4);
  if (span) mix_ceiling_kernel<true><<<blocks, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const u32x4*>(d_read), d_write, n_r, n_w);
  else mix_ceiling_kernel<false><<<blocks, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const u32x4*>(d_read), d_write, n_r, n_w);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

// One physical allocation of chunk_bytes mapped `repeats` times back to back into a fresh virtual range: a kernel walking
// repeats * chunk_bytes of addresses touches only chunk_bytes of memory.  With the chunk below the 256-MiB Infinity Cache
// every read of the range is served on-die -- the SAME launch (same addresses per workgroup, same instruction stream, same
// fabric request counts) with and without its reads reaching HBM.
int mdcb_alias_alloc(int device, int64_t chunk_bytes, int repeats, void** out_ptr, int64_t* out_granularity) {
  if (chunk_bytes <= 0 || repeats <= 0 || !out_ptr) return -1;
  DeviceGuard dg(device);
  int dev = device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return -4;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return -4;
  if (out_granularity) *out_granularity = (int64_t)gran;
  if ((size_t)chunk_bytes % gran != 0) return -3;  // the caller picks a chunk that is a whole number of pages
  hipMemGenericAllocationHandle_t h;
  if (hipMemCreate(&h, (size_t)chunk_bytes, &prop, 0) != hipSuccess) return -4;
  void* va = nullptr;
  const size_t total = (size_t)chunk_bytes * (size_t)repeats;
  if (hipMemAddressReserve(&va, total, gran, nullptr, 0) != hipSuccess) {
    (void)hipMemRelease(h);
    return -4;
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int r = 0; r < repeats; r++) {
    char* at = static_cast<char*>(va) + (size_t)r * (size_t)chunk_bytes;
    if (hipMemMap(at, (size_t)chunk_bytes, 0, h, 0) != hipSuccess || hipMemSetAccess(at, (size_t)chunk_bytes, &acc, 1) != hipSuccess) {
      for (int k = 0; k <= r; k++) (void)hipMemUnmap(static_cast<char*>(va) + (size_t)k * (size_t)chunk_bytes, (size_t)chunk_bytes);
      (void)hipMemAddressFree(va, total);
      (void)hipMemRelease(h);
      return -4;
    }
  }
  (void)hipMemRelease(h);  // the mappings keep the physical memory alive
  *out_ptr = va;
  return 0;
}

int mdcb_alias_free(int device, void* ptr, int64_t chunk_bytes, int repeats) {
  if (!ptr) return 0;
  DeviceGuard dg(device);
  int rc = 0;
  for (int r = 0; r < repeats; r++)
    if (hipMemUnmap(static_cast<char*>(ptr) + (size_t)r * (size_t)chunk_bytes, (size_t)chunk_bytes) != hipSuccess) rc = -4;
  if (hipMemAddressFree(ptr, (size_t)chunk_bytes * (size_t)repeats) != hipSuccess) rc = -4;
  return rc;
}

// Experiment (tools/alloc_probe.py): a device range of n * chunk_bytes built from n separately created physical chunks (hipMemCreate),
// mapped in the order of creation (stride = 1) or so that chunk i of the range is the (i * stride mod n)-th chunk created: neighbouring
// pieces of the range then lie far apart physically.  What does the placement of a buffer's pieces do to a linear stream's rate?
int mdcb_chunked_alloc(int device, int64_t chunk_bytes, int n, int stride, void** out_ptr) {
  if (chunk_bytes <= 0 || n <= 0 || stride <= 0 || !out_ptr) return -1;
  DeviceGuard dg(device);
  int dev = device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return -4;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return -4;
  if ((size_t)chunk_bytes % gran != 0) return -3;
  void* va = nullptr;
  const size_t total = (size_t)chunk_bytes * (size_t)n;
  if (hipMemAddressReserve(&va, total, gran, nullptr, 0) != hipSuccess) return -4;
  hipMemGenericAllocationHandle_t* h = new hipMemGenericAllocationHandle_t[n];
  int made = 0;
  for (; made < n; made++)
    if (hipMemCreate(&h[made], (size_t)chunk_bytes, &prop, 0) != hipSuccess) break;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  int rc = made == n ? 0 : -4;
  int mapped = 0;
  for (; mapped < n && rc == 0; mapped++) {
    char* at = static_cast<char*>(va) + (size_t)mapped * (size_t)chunk_bytes;
    const int src = (int)(((long long)mapped * stride) % n);  // a permutation when gcd(stride, n) == 1
    if (hipMemMap(at, (size_t)chunk_bytes, 0, h[src], 0) != hipSuccess) rc = -4;
  }
  if (rc == 0 && hipMemSetAccess(va, total, &acc, 1) != hipSuccess) rc = -4;
  for (int k = 0; k < made; k++) (void)hipMemRelease(h[k]);  // the mappings keep the memory
  delete[] h;
  if (rc != 0) {
    for (int k = 0; k < mapped; k++) (void)hipMemUnmap(static_cast<char*>(va) + (size_t)k * (size_t)chunk_bytes, (size_t)chunk_bytes);
    (void)hipMemAddressFree(va, total);
    return rc;
  }
  *out_ptr = va;
  return 0;
}

}  // extern "C"
