// End-to-end rate of a sequence reader: decode + photometric + rectify, frames/s.  ONE source, three builds
// (oracle/Makefile):
//   reader_rate_ref  : the reference's reader + the reference's classes           (CPU, single-threaded as shipped)
//   reader_rate_mdc  : the reference's UNMODIFIED reader + this repo's drop-in classes (two GPU calls per frame)
//   reader_rate_fast : this repo's reader (fused getImage; with "batch" as 4th argument: getImages)
//
//   reader_rate_X <sequence folder> <rgvo flags> <passes> [batch | device | device_dso]
//     device     : getImagesDevice -- results left in HBM (base image only)
//     device_dso : ... + box levels 1-3 + gradient images of every level (the DSO hand-off)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "BenchmarkDatasetReader.h"
#ifdef MDC_OWN_READER
#include "mdc_hip.h"
#endif

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::string folder = argv[1];
  if (folder.empty() || folder[folder.size() - 1] != '/') folder += "/";
  const char* f = argv[2];
  const int passes = std::atoi(argv[3]);
  const bool batch = argc > 4 && !std::strcmp(argv[4], "batch");
  const bool device = argc > 4 && !std::strncmp(argv[4], "device", 6), device_dso = argc > 4 && !std::strcmp(argv[4], "device_dso");
  DatasetReader* reader = new DatasetReader(folder);
  const int n = reader->getNumImages();
#ifdef MDC_OWN_READER
  if (const char* t = std::getenv("MDC_READER_THREADS")) reader->setDecodeThreads(std::atoi(t));
#endif
  double checksum = 0;
#ifdef MDC_OWN_READER
  // device-resident outputs: one set of arrays for the whole sequence, made once (a GPU consumer's buffers)
  mdc_device_outputs dout;
  std::memset(&dout, 0, sizeof dout);
  std::vector<void*> dmem;
  const bool rect = f[0] == '1';
  if (device) {
    mdc_ctx* ctx = reader->getContext();
    if (!ctx) return 3;
    const int wh = rect ? reader->getUndistorter()->getOutputDims()[0] * reader->getUndistorter()->getOutputDims()[1]
                        : reader->getUndistorter()->getInputDims()[0] * reader->getUndistorter()->getInputDims()[1];
    const int w0 = rect ? reader->getUndistorter()->getOutputDims()[0] : reader->getUndistorter()->getInputDims()[0];
    const int h0 = wh / w0;
    auto dev_floats = [&](size_t count) {
      void* p = 0;
      if (mdc_device_alloc(ctx, count * sizeof(float), &p) != MDC_OK) std::exit(4);
      dmem.push_back(p);
      return (float*)p;
    };
    dout.base = dev_floats((size_t)n * wh);
    dout.levels = device_dso ? 4 : 1;
    for (int l = 0; l < dout.levels; l++) {
      const size_t npl = (size_t)(w0 >> l) * (h0 >> l);
      if (l) dout.level[l - 1] = dev_floats((size_t)n * npl);
      if (device_dso) {
        dout.dI[l] = dev_floats((size_t)n * npl * 3);
        dout.abs_squared_grad[l] = dev_floats((size_t)n * npl);
      }
    }
  }
#endif
  auto run = [&](int reps) {
    for (int p = 0; p < reps; p++) {
#ifdef MDC_OWN_READER
      if (device) {
        std::vector<unsigned char> valid((size_t)n);
        const int got = reader->getImagesDevice(0, n, rect, f[1] == '1', f[2] == '1', f[3] == '1', &dout, valid.data());
        checksum += got;
        continue;
      }
      if (batch) {
        std::vector<ExposureImage*> imgs((size_t)n);
        reader->getImages(0, n, f[0] == '1', f[1] == '1', f[2] == '1', f[3] == '1', imgs.data());
        for (int i = 0; i < n; i++)
          if (imgs[(size_t)i]) {
            const float x = imgs[(size_t)i]->image[(size_t)(i * 7919) % ((size_t)imgs[(size_t)i]->w * imgs[(size_t)i]->h)];
            if (x == x) checksum += x;
            delete imgs[(size_t)i];
          }
        continue;
      }
#endif
      for (int i = 0; i < n; i++) {
        ExposureImage* img = reader->getImage(i, f[0] == '1', f[1] == '1', f[2] == '1', f[3] == '1');
        if (!img) continue;
        const float x = img->image[(size_t)(i * 7919) % ((size_t)img->w * img->h)];
        if (x == x) checksum += x;
        delete img;
      }
    }
  };
  run(1);  // warm: page cache, GPU context, pools
  const auto t0 = std::chrono::steady_clock::now();
  run(passes);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("READER_RATE %s%s flags %s: %d frames x %d passes in %.3f s = %.1f frames/s (checksum %.6g)\n", argv[0],
              batch ? " batch" : device_dso ? " device_dso (results + levels + gradients left in HBM)" : device ? " device (results left in HBM)" : "", f, n, passes, dt, n * passes / dt, checksum);
#ifdef MDC_OWN_READER
  long hits = 0, misses = 0;
  reader->getPrefetchStats(&hits, &misses);
  std::printf("READER_RATE prefetch: %ld frames found decoded ahead, %ld decoded by the calling thread\n", hits, misses);
  // per device (MDC_DEVICES): frames it produced over the run (warm-up pass included) and the rate inside its own GPU calls
  for (int l = 0; l < reader->getDeviceCount(); l++) {
    int dev = -1;
    long frames = 0;
    double wait_s = 0, gpu_s = 0;
    reader->getDeviceStats(l, &dev, &frames, &wait_s, &gpu_s);
    std::printf("READER_RATE device %d (lane %d of %d): %ld frames, %.3f s in GPU calls (%.1f frames/s there), %.3f s waiting for the decoders\n", dev, l,
                reader->getDeviceCount(), frames, gpu_s, gpu_s > 0 ? frames / gpu_s : 0.0, wait_s);
  }
#endif
#ifdef MDC_OWN_READER
  if (device) {
    float probe[4] = {0, 0, 0, 0};  // the first pixels of the last pass's first result, copied back: the arrays really hold results
    mdc_copy_to_host(reader->getContext(), probe, dout.base, sizeof probe);
    std::printf("READER_RATE device outputs: base[0..3] = %g %g %g %g\n", probe[0], probe[1], probe[2], probe[3]);
    for (void* p : dmem) mdc_device_free(reader->getContext(), p);
  }
#endif
  delete reader;
  return 0;
}
