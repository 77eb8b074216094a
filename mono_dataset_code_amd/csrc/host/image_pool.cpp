// Recycling pool behind ExposureImage::image (include/mono_dataset_code/ExposureImage.h).
//
// The reference allocates `new float[w*h]` per frame and the caller deletes it
// (src/ExposureImage.h:45,49): 1.2 - 5.2 MB of fresh, pageable memory per getImage().  Here the blocks are
// page-locked (mdc_host_alloc = hipHostMalloc) so that the device-to-host copy of a result runs at PCIe
// rate, and freed blocks are kept for the next image of the same size.
#include <cstddef>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "mdc_hip.h"

namespace {

struct Pool {
  std::mutex mu;
  std::map<float*, std::pair<size_t, bool>> live;           // block -> (floats, page-locked)
  std::map<size_t, std::vector<float*>> idle_pinned;         // floats -> blocks ready for reuse
  size_t idle_bytes = 0;
  // beyond this, freed blocks go back to the system.  Page-locking a fresh 1.2-MB block costs ~0.4 ms (hipHostMalloc), twenty
  // times the GPU's time for the frame it will hold: a caller that keeps 768 images of a getImages call alive and then deletes
  // them ran at 4.3 k frames/s with a 512-MiB stock and at 20 k with one that holds them all (mdch_image_pool_trim releases it)
  static constexpr size_t kMaxIdleBytes = (size_t)2048 << 20;
  ~Pool() {
    // process exit: the HIP runtime may already be gone -- leave page-locked blocks to the OS
  }
};
Pool& pool() {
  static Pool* p = new Pool();  // never destroyed: images may outlive static destruction order
  return *p;
}

}  // namespace

extern "C" float* mdch_image_alloc(unsigned long nfloats) {
  if (nfloats == 0) nfloats = 1;
  Pool& P = pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.idle_pinned.find(nfloats);
    if (it != P.idle_pinned.end() && !it->second.empty()) {
      float* b = it->second.back();
      it->second.pop_back();
      P.idle_bytes -= nfloats * sizeof(float);
      P.live[b] = std::make_pair((size_t)nfloats, true);
      return b;
    }
  }
  bool pinned = true;
  float* b = static_cast<float*>(mdc_host_alloc(nfloats * sizeof(float)));
  if (!b) {  // no GPU / no page-locked memory left: an ordinary block (a container, not a compute fallback)
    pinned = false;
    b = new float[nfloats];
  }
  std::lock_guard<std::mutex> lk(P.mu);
  P.live[b] = std::make_pair((size_t)nfloats, pinned);
  return b;
}

extern "C" void mdch_image_free(float* b) {
  if (!b) return;
  Pool& P = pool();
  size_t n = 0;
  bool pinned = false, keep = false;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.live.find(b);
    if (it == P.live.end()) return;  // not ours (double free of a foreign pointer): leave it alone
    n = it->second.first;
    pinned = it->second.second;
    P.live.erase(it);
    if (pinned && P.idle_bytes + n * sizeof(float) <= Pool::kMaxIdleBytes) {
      P.idle_pinned[n].push_back(b);
      P.idle_bytes += n * sizeof(float);
      keep = true;
    }
  }
  if (keep) return;
  if (pinned) mdc_host_free(b);
  else delete[] b;
}

// Releases every idle block (tests; long-running hosts that switch sequence geometry).
extern "C" void mdch_image_pool_trim() {
  Pool& P = pool();
  std::vector<float*> drop;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto& kv : P.idle_pinned)
      for (float* b : kv.second) drop.push_back(b);
    P.idle_pinned.clear();
    P.idle_bytes = 0;
  }
  for (float* b : drop) mdc_host_free(b);
}

extern "C" unsigned long mdch_image_pool_idle_bytes() {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  return (unsigned long)P.idle_bytes;
}
