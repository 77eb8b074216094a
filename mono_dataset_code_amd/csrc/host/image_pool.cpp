// Recycling pool behind ExposureImage::image (include/mono_dataset_code/ExposureImage.h).
//
// The reference allocates `new float[w*h]` per frame and the caller deletes it
// (src/ExposureImage.h:45,49): 1.2 - 5.2 MB of fresh, pageable memory per getImage().  Here the blocks are
// page-locked (mdc_host_alloc = hipHostMalloc) so that the GPU writes a result into it at PCIe rate, and freed
// blocks are kept for the next image of the same size.
//
// Blocks are carved out of SLABS -- one page-locked allocation holding up to 64 images back to back -- and the
// lowest free address is handed out first: the images a getImages call makes one after the other lie back to
// back in memory, which is what lets the pipelined GPU call (csrc/mdc_pipeline.hip) move a whole
// chunk of results with ONE copy -- or one kernel launch writing them in place -- instead of one copy per image
// (64 copies of 1.2 MB: 25 GB/s; one of 79 MB: 54).  It also page-locks once per slab instead of once per image
// (~0.4 ms each).
//
// Opt-out: MDC_IMAGE_POOL=0 in the environment makes the block what the reference's is -- `new float[w*h]`, released with
// `delete[]` -- for downstream code that frees or swaps `image` itself (legal against src/ExposureImage.h:42-50).  The GPU then
// reaches the images through staging copies instead of writing them in place; same values.
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstdint>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <vector>

#include "mdc_hip.h"
#include "mdc_host.h"  // the MDC_API (exported) declarations of the mdch_image_* functions defined below

namespace {

struct Slab {
  size_t nfloats = 0;  // per image
  size_t stride = 0;   // floats from one image to the next: nfloats, or -- sizes that are not whole 64-byte lines -- rounded up to one
  int count = 0;       // images in the slab
  int live = 0;        // of them handed out
  bool pinned = false;
};

struct Pool {
  std::mutex mu;
  std::map<float*, Slab> slabs;                 // base address -> slab
  std::map<size_t, std::set<float*>> idle;     // floats per image -> free blocks, by address
  size_t idle_bytes = 0;
  // beyond this, slabs whose images have all come back go to the system.  Page-locking is slow (a fresh 1.2-MB block: ~0.4 ms,
  // twenty times the GPU's time for the frame it will hold): a caller that keeps 768 images of a getImages call alive and then
  // deletes them ran at 4.3 k frames/s with a 512-MiB stock and at 20 k with one that holds them all (mdch_image_pool_trim
  // releases it).  MDC_IMAGE_POOL_MAX_MB in the environment moves the limit (a caller that holds 2048 results of 640x480 at a time and
  // deletes them together needs 2.5 GB of stock to get the same blocks back: 13 k frames/s with the default, 35 k with 4096).
  static size_t max_idle_bytes() {
    static const size_t v = [] {
      const char* e = std::getenv("MDC_IMAGE_POOL_MAX_MB");
      const long mb = e ? std::atol(e) : 2048;
      return (size_t)std::max(64l, std::min(mb, 1l << 20)) << 20;
    }();
    return v;
  }
  static constexpr size_t kSlabBytes = (size_t)96 << 20;
  static constexpr int kSlabImages = 64;
  static constexpr size_t kPageableSlabBytes = (size_t)16 << 20;  // without a GPU / page-locked memory
  ~Pool() {
    // process exit: the HIP runtime may already be gone -- leave page-locked blocks to the OS
  }

  // the slab a block belongs to (nullptr: not one of ours, or not a block boundary)
  std::map<float*, Slab>::iterator slab_of(float* b) {
    auto it = slabs.upper_bound(b);
    if (it == slabs.begin()) return slabs.end();
    --it;
    const Slab& s = it->second;
    const size_t off = (size_t)(b - it->first);
    if (off >= s.stride * (size_t)s.count || off % s.stride != 0) return slabs.end();
    return it;
  }
  // takes a slab whose images are all idle out of the books; the caller frees the memory outside the lock
  void retire(std::map<float*, Slab>::iterator it, std::vector<std::pair<float*, bool>>* drop) {
    const Slab s = it->second;
    std::set<float*>& free_blocks = idle[s.nfloats];
    for (int i = 0; i < s.count; i++) free_blocks.erase(it->first + (size_t)i * s.stride);
    if (free_blocks.empty()) idle.erase(s.nfloats);
    idle_bytes -= (size_t)s.count * s.nfloats * sizeof(float);
    drop->push_back(std::make_pair(it->first, s.pinned));
    slabs.erase(it);
  }
};
Pool& pool() {
  static Pool* p = new Pool();  // never destroyed: images may outlive static destruction order
  return *p;
}

bool pool_off() {
  static const bool off = [] {
    const char* e = std::getenv("MDC_IMAGE_POOL");
    return e && e[0] == '0';
  }();
  return off;
}

void release(const std::vector<std::pair<float*, bool>>& drop) {
  for (const auto& d : drop) {
    if (d.second) mdc_host_free(d.first);
    else delete[] d.first;
  }
}

}  // namespace

extern "C" float* mdch_image_alloc(unsigned long nfloats) {
  if (nfloats == 0) nfloats = 1;
  if (pool_off()) return new (std::nothrow) float[nfloats];
  Pool& P = pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.idle.find(nfloats);
    if (it != P.idle.end() && !it->second.empty()) {
      float* b = *it->second.begin();  // lowest address first: consecutive images lie back to back
      it->second.erase(it->second.begin());
      P.idle_bytes -= nfloats * sizeof(float);
      P.slab_of(b)->second.live++;
      return b;
    }
  }
  // a new slab (outside the lock: page-locking takes milliseconds); if that much page-locked memory is not to be had, one image;
  // without a GPU ordinary memory (a container, not a compute fallback)
  const size_t stride = (nfloats + 15) & ~(size_t)15, bytes = nfloats * sizeof(float);
  const int want = (int)std::max<size_t>(1, std::min<size_t>(Pool::kSlabImages, Pool::kSlabBytes / (stride * sizeof(float))));
  int count = want;
  bool pinned = true;
  float* base = static_cast<float*>(mdc_host_alloc(stride * sizeof(float) * (size_t)count));
  if (!base && count > 1) {
    count = 1;
    base = static_cast<float*>(mdc_host_alloc(bytes));
  }
  if (!base) {  // the same bookkeeping over ordinary memory, in smaller slabs
    pinned = false;
    count = (int)std::max<size_t>(1, std::min<size_t>((size_t)want, Pool::kPageableSlabBytes / (stride * sizeof(float))));
    base = new float[stride * (size_t)count];
  }
  std::lock_guard<std::mutex> lk(P.mu);
  Slab s;
  s.nfloats = nfloats;
  s.stride = stride;
  s.count = count;
  s.live = 1;
  s.pinned = pinned;
  P.slabs[base] = s;
  if (count > 1) {
    std::set<float*>& free_blocks = P.idle[nfloats];
    for (int i = 1; i < count; i++) free_blocks.insert(base + (size_t)i * stride);
    P.idle_bytes += (size_t)(count - 1) * bytes;
  }
  return base;
}

extern "C" void mdch_image_free(float* b) {
  if (!b) return;
  if (pool_off()) {  // whatever the pointer holds now is a `new float[]` block (ours, or one the caller swapped in)
    delete[] b;
    return;
  }
  Pool& P = pool();
  std::vector<std::pair<float*, bool>> drop;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.slab_of(b);
    if (it == P.slabs.end()) return;  // not ours (a foreign pointer): leave it alone
    Slab& s = it->second;
    std::set<float*>& free_blocks = P.idle[s.nfloats];
    if (!free_blocks.insert(b).second) return;  // a second free of the same block is ignored
    s.live--;
    P.idle_bytes += s.nfloats * sizeof(float);
    if (P.idle_bytes > Pool::max_idle_bytes()) {
      // over the cap: slabs without a live image go back (this one first; one with live images cannot)
      if (s.live == 0) P.retire(it, &drop);
      for (auto jt = P.slabs.begin(); jt != P.slabs.end() && P.idle_bytes > Pool::max_idle_bytes();) {
        auto cur = jt++;
        if (cur->second.live == 0) P.retire(cur, &drop);
      }
    }
  }
  release(drop);
}

// Releases every slab without a live image (tests; long-running hosts that switch sequence geometry).
extern "C" void mdch_image_pool_trim() {
  Pool& P = pool();
  std::vector<std::pair<float*, bool>> drop;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto jt = P.slabs.begin(); jt != P.slabs.end();) {
      auto cur = jt++;
      if (cur->second.live == 0) P.retire(cur, &drop);
    }
  }
  release(drop);
}

// Bytes of idle blocks in slabs that could be released (slabs without a live image).
extern "C" unsigned long mdch_image_pool_idle_bytes() {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  size_t n = 0;
  for (const auto& kv : P.slabs)
    if (kv.second.live == 0) n += (size_t)kv.second.count * kv.second.nfloats * sizeof(float);
  return (unsigned long)n;
}
