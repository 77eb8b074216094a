// Drop-in for the reference's src/ExposureImage.h:33-51 -- the frame container
// DatasetReader::getImage returns (src/BenchmarkDatasetReader.h:221).  Field
// names, order and the constructor signature are the public API that callers
// (main_playbackDataset.cpp:82,116, DSO-style consumers) rely on: `new ExposureImage(...)`
// by the reader, `delete img` by the caller.
//
// The pixel block comes from a recycling pool of PAGE-LOCKED host memory (libmdc_host.so,
// csrc/host/image_pool.cpp) instead of a fresh `new float[w*h]` per frame: the GPU writes results
// into it at PCIe rate, and a caller that deletes each image after use (as playDataset does) gets
// the same block back for the next frame -- no page faults, no re-pinning.  Without a GPU the pool
// hands out ordinary heap memory.  Code that frees or replaces `image` itself (`delete[] img->image`,
// legal against the reference's type) sets MDC_IMAGE_POOL=0 in the environment: the block is then a
// plain `new float[w*h]` released with `delete[]`, as in the reference.
#pragma once

#ifndef MDC_API  /* the libraries are built with -fvisibility=hidden: this marks what they export */
#if defined(__GNUC__) || defined(__clang__)
#define MDC_API __attribute__((visibility("default")))
#else
#define MDC_API
#endif
#endif
extern "C" {
MDC_API float* mdch_image_alloc(unsigned long nfloats);
MDC_API void mdch_image_free(float* block);
}

class MDC_API ExposureImage {
 public:
  float* image;         // w*h irradiance / intensity values, row-major, owned
  double timestamp;     // seconds, from times.txt
  int w, h;
  float exposure_time;  // milliseconds, from times.txt (0 if unknown)
  int id;               // frame index in the sequence

  ExposureImage(int width, int height, double stamp, float exposure, int frame_id)
      : image(mdch_image_alloc(static_cast<unsigned long>(width) * static_cast<unsigned long>(height))),
        timestamp(stamp), w(width), h(height), exposure_time(exposure), id(frame_id) {}
  ~ExposureImage() { mdch_image_free(image); }

  // the reference type is used through pointers only; copying would double-free
  ExposureImage(const ExposureImage&) = delete;
  ExposureImage& operator=(const ExposureImage&) = delete;
};
