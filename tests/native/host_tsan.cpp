// ThreadSanitizer harness for the host layer's threads (tests/test_sanitize.py::test_host_threads_under_tsan):
//   * the reader's decode pool: getImageRaw in order (frames prefetched by the pool while the caller consumes), jumps, a change
//     of the pool size and of the prefetch depth between calls, a damaged frame met by a pool thread, the reader destroyed
//     while prefetches are in flight;
//   * several readers on one folder, each on its own thread;
//   * the ExposureImage pool hammered from several threads (slabs created, blocks reused, trimmed concurrently);
//   * the frame decoders and mdch_jpeg_stream called concurrently on shared input bytes.
// No GPU work: this runs on the CPU-only test host; the HIP side has its own thread tests (tests/native/thread_soak.cpp).
// usage: host_tsan <fixture root made by the test>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "BenchmarkDatasetReader.h"
#include "image_codecs.h"
#include "image_codecs_internal.h"
#include "mdc_hip.h"
#include "mdc_host.h"

static long walk(const std::string& dir, int rounds, int threads, int prefetch) {
  DatasetReader reader(dir);
  reader.setDecodeThreads(threads);
  reader.setPrefetch(prefetch);
  const int n = reader.getNumImages();
  long sum = 0;
  for (int r = 0; r < rounds; r++) {
    for (int i = 0; i < n; i++) {
      int w = 0, h = 0;
      const unsigned char* p = reader.getImageRaw(i, &w, &h);
      if (p && w > 0 && h > 0) sum += p[0] + p[(size_t)w * h - 1];
    }
    for (int i = n - 1; i >= 0; i -= 2) {
      int w = 0, h = 0;
      const unsigned char* p = reader.getImageRaw(i, &w, &h);
      if (p && w > 0) sum += p[w / 2];
    }
    reader.setDecodeThreads(1 + (r % 3));
    reader.setPrefetch(r % 2 ? 3 : prefetch);
  }
  (void)reader.getImageRaw(0, 0, 0);  // leaves prefetches in flight for the destructor
  return sum;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string root = argv[1];
  long total = 0;
  for (const char* seq : {"seq_png", "seq_zip_jpg", "seq_zip_badsize"}) total += walk(root + "/sequences/" + seq + "/", 3, 4, 8);
  {
    std::vector<std::thread> th;
    std::atomic<long> sum(0);
    for (int t = 0; t < 4; t++)
      th.emplace_back([&, t] { sum += walk(root + "/sequences/" + (t % 2 ? "seq_zip_jpg" : "seq_png") + "/", 2, 2 + t % 2, 4); });
    for (auto& x : th) x.join();
    total += sum.load();
  }
  {
    std::vector<std::thread> th;
    std::atomic<int> bad(0);
    for (int t = 0; t < 8; t++)
      th.emplace_back([&, t] {
        std::vector<float*> mine;
        for (int r = 0; r < 200; r++) {
          float* b = mdch_image_alloc(561 + 16 * (t % 2));
          b[0] = (float)t;
          b[560] = (float)r;
          mine.push_back(b);
          if (r % 3 == 2) {
            float* f = mine[(size_t)r / 3];
            if (f && f[0] != (float)t) bad++;
            mdch_image_free(f);
            mine[(size_t)r / 3] = nullptr;
          }
          if (r % 50 == 49 && t == 0) mdch_image_pool_trim();
        }
        for (float* b : mine)
          if (b) {
            if (b[0] != (float)t) bad++;
            mdch_image_free(b);
          }
      });
    for (auto& x : th) x.join();
    mdch_image_pool_trim();
    if (bad.load() || mdch_image_pool_idle_bytes() != 0) {
      std::printf("image pool: %d blocks were handed out twice, %lu idle bytes after the trim\n", bad.load(), mdch_image_pool_idle_bytes());
      return 1;
    }
  }
  {
    std::vector<unsigned char> jpg, png;
    if (!mdc_host::read_file(root + "/images_any/b.jpg", jpg) || !mdc_host::read_file(root + "/images_any/a.png", png)) return 2;
    std::vector<std::thread> th;
    std::atomic<long> sum(0);
    for (int t = 0; t < 6; t++)
      th.emplace_back([&] {
        std::vector<unsigned char> px(1 << 16), st(sizeof(mdc_jpeg_stream_header) + jpg.size() + 64);
        for (int r = 0; r < 30; r++) {
          int w = 0, h = 0;
          std::string err;
          if (mdc_host::decode_gray8(jpg.data(), jpg.size(), px.data(), px.size(), &w, &h, &err)) sum += px[0];
          if (mdc_host::decode_gray8(png.data(), png.size(), px.data(), px.size(), &w, &h, &err)) sum += px[1];
          size_t used = 0;
          if (mdc_host::jpeg_stream(jpg.data(), jpg.size(), st.data(), st.size(), &used, &w, &h, &err)) sum += (long)used;
        }
      });
    for (auto& x : th) x.join();
    total += sum.load();
  }
  std::printf("HOST_TSAN_OK %ld\n", total);
  return 0;
}
