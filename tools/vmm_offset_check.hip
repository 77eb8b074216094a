// Does hipMemMap(ptr, size, OFFSET, handle) map the sub-range [offset, offset + size) of the handle?  (mdc_placement.hip maps a range in
// stripes: stripe t = sub-range (t / n) of piece (t mod n).)  Every piece is also mapped whole in a probe range; a kernel tags every 2-MiB
// page through the striped range, the tags are read back through the probe range and compared with the mapping function.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_offset_check tools/vmm_offset_check.hip && /tmp/vmm_offset_check [pieces] [piece MiB] [stripe MiB] [rounds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)

__global__ void tag_pages(unsigned* base, size_t pages, unsigned salt) {  // page p: every word of it = salt + p
  const size_t words_per_page = (2u << 20) / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pages * words_per_page; i += (size_t)gridDim.x * blockDim.x)
    base[i] = salt + (unsigned)(i / words_per_page);
}
__global__ void read_pages(const unsigned* base, size_t pages, unsigned* out) {  // first and last word of every page
  const size_t words_per_page = (2u << 20) / 4;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < pages) {
    out[2 * p] = base[p * words_per_page];
    out[2 * p + 1] = base[(p + 1) * words_per_page - 1];
  }
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? std::atoi(argv[1]) : 3, piece = (size_t)(argc > 2 ? std::atoi(argv[2]) : 512) << 20, stripe = (size_t)(argc > 3 ? std::atoi(argv[3]) : 64) << 20;
  const int rounds = argc > 4 ? std::atoi(argv[4]) : 3;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t pages_per_piece = piece >> 21, pages = n * pages_per_piece, per = piece / stripe;
  unsigned* d_out = nullptr;
  CK(hipMalloc(&d_out, pages * 8));
  std::vector<unsigned> h(pages * 2);
  int bad_total = 0;
  for (int r = 0; r < rounds; r++) {
    std::vector<hipMemGenericAllocationHandle_t> hs(n);
    for (size_t k = 0; k < n; k++) CK(hipMemCreate(&hs[k], piece, &prop, 0));
    void *probe = nullptr, *range = nullptr;
    CK(hipMemAddressReserve(&probe, n * piece, gran, nullptr, 0));
    for (size_t k = 0; k < n; k++) CK(hipMemMap((char*)probe + k * piece, piece, 0, hs[k], 0));
    CK(hipMemSetAccess(probe, n * piece, &acc, 1));
    hipLaunchKernelGGL(tag_pages, dim3(4096), dim3(256), 0, 0, (unsigned*)probe, pages, 0x70000000u);  // what the probe mapping sees before
    CK(hipDeviceSynchronize());
    CK(hipMemAddressReserve(&range, n * piece, gran, nullptr, 0));
    for (size_t t = 0; t < n * per; t++) {
      const hipError_t e = hipMemMap((char*)range + t * stripe, stripe, (t / n) * stripe, hs[t % n], 0);
      if (e != hipSuccess) {
        std::printf("hipMemMap of stripe %zu (offset %zu MiB into its handle) refused: %s\n", t, ((t / n) * stripe) >> 20, hipGetErrorString(e));
        return 3;
      }
    }
    CK(hipMemSetAccess(range, n * piece, &acc, 1));
    const unsigned salt = 0x10000000u * (unsigned)(r + 1);
    hipLaunchKernelGGL(tag_pages, dim3(4096), dim3(256), 0, 0, (unsigned*)range, pages, salt);
    CK(hipDeviceSynchronize());
    // through the probe range: page q of piece k must carry the tag of range page  (stripe t = (q_stripe) * n + k) ...
    hipLaunchKernelGGL(read_pages, dim3((unsigned)((pages + 255) / 256)), dim3(256), 0, 0, (const unsigned*)probe, pages, d_out);
    CK(hipMemcpy(h.data(), d_out, pages * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    const size_t pages_per_stripe = stripe >> 21;
    for (size_t k = 0; k < n; k++)
      for (size_t q = 0; q < pages_per_piece; q++) {
        const size_t sub = q / pages_per_stripe, within = q % pages_per_stripe;  // sub-range of piece k
        const size_t t = sub * n + k;                                              // the stripe that maps it
        const unsigned want = salt + (unsigned)(t * pages_per_stripe + within);
        const unsigned a = h[2 * (k * pages_per_piece + q)], b = h[2 * (k * pages_per_piece + q) + 1];
        if (a != want || b != want) {
          if (bad < 6) std::printf("  round %d piece %zu page %zu (offset %zu MiB): expected tag %08x, found %08x / %08x\n", r, k, q, q * 2, want, a, b);
          bad++;
        }
      }
    // and through the range itself
    hipLaunchKernelGGL(read_pages, dim3((unsigned)((pages + 255) / 256)), dim3(256), 0, 0, (const unsigned*)range, pages, d_out);
    CK(hipMemcpy(h.data(), d_out, pages * 8, hipMemcpyDeviceToHost));
    int bad2 = 0;
    for (size_t p = 0; p < pages; p++)
      if (h[2 * p] != salt + (unsigned)p || h[2 * p + 1] != salt + (unsigned)p) bad2++;
    // hipMemcpy of the range (the copy engine's view)
    std::vector<unsigned> first(pages);
    for (size_t p = 0; p < pages; p += 37) CK(hipMemcpy(&first[p], (char*)range + (p << 21), 4, hipMemcpyDeviceToHost));
    int bad3 = 0;
    for (size_t p = 0; p < pages; p += 37)
      if (first[p] != salt + (unsigned)p) bad3++;
    std::printf("round %d: %zu pieces of %zu MiB, stripes of %zu MiB, range %p probe %p: pages wrong through the probe range %d, through the range %d, by hipMemcpy %d (of %zu)\n", r, n,
                piece >> 20, stripe >> 20, range, probe, bad, bad2, bad3, pages);
    bad_total += bad + bad2 + bad3;
    CK(hipDeviceSynchronize());
    for (size_t t = 0; t < n * per; t++) CK(hipMemUnmap((char*)range + t * stripe, stripe));
    for (size_t k = 0; k < n; k++) CK(hipMemUnmap((char*)probe + k * piece, piece));
    if (!std::getenv("VMM_KEEP_VA")) {  // (VMM_KEEP_VA=1: the reservations are never given back, no address is ever mapped twice)
      CK(hipMemAddressFree(range, n * piece));
      CK(hipMemAddressFree(probe, n * piece));
    }
    if (std::getenv("VMM_CHURN")) {  // an ordinary allocation between the rounds (does its unmap flush what the vmem calls left behind?)
      void* x = nullptr;
      CK(hipMalloc(&x, (size_t)1 << 30));
      CK(hipMemset(x, 1, (size_t)1 << 30));
      CK(hipDeviceSynchronize());
      CK(hipFree(x));
    }
    for (size_t k = 0; k < n; k++) CK(hipMemRelease(hs[k]));
  }
  std::printf("%s\n", bad_total ? "MISMATCHES" : "all pages where the mapping function says");
  return bad_total ? 1 : 0;
}
