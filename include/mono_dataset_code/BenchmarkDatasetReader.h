// class DatasetReader with the public interface of the reference's reader
// (src/BenchmarkDatasetReader.h:83-345): same constructor, same getters, same
// getImage(id, rectify, removeGamma, removeVignette, nanOverexposed) returning a caller-owned
// ExposureImage*.  playDataset (src/main_playbackDataset.cpp:56-116) and DSO-style front ends
// compile against it unchanged; what changes is behind the interface:
//
//   * getImage is ONE fused GPU pass (mdc_process_host: response LUT x vignette + bilinear remap,
//     include/mdc_hip.h) from a page-locked decode buffer into a pooled page-locked ExposureImage --
//     no W*H float intermediate (the reference's internalTempBuffer, :145,:222), no per-frame
//     `new float[]`;
//   * frames are decoded by the library's own PNG / PGM / baseline-JPEG decoders, from the images/
//     folder or straight out of images.zip (own zip reader) -- no OpenCV, no libzip;
//   * the multi-threaded loader the reference's comment announces (:81) exists: a pool of decode
//     threads prefetches the frames after the one just asked for, and getImages() runs a whole range
//     through decode pool -> ring of page-locked chunks -> pipelined GPU chunks (mdc_process_frames_host).
//
// Results are the reference's, bit for bit (tests/test_reader.py against the reference's own reader).
// Like the reference's, an object is NOT re-entrant: one thread calls into it at a time.
#pragma once
#include <string>

// The reference's header pulls OpenCV in (src/BenchmarkDatasetReader.h:33-37) and hands out cv::Mat from
// getImageRaw_internal (:247, used by src/main_responseCalib.cpp:194).  This library itself needs no OpenCV, so the
// include is taken only where the headers exist; wherever they do, the accessor below is a regular public member.
#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#endif
#endif

#include "ExposureImage.h"
#include "FOVUndistorter.h"
#include "PhotometricUndistorter.h"

#ifndef MDC_API  /* the libraries are built with -fvisibility=hidden: this marks what they export */
#if defined(__GNUC__) || defined(__clang__)
#define MDC_API __attribute__((visibility("default")))
#else
#define MDC_API
#endif
#endif
class MDC_API DatasetReader {
 public:
  // `folder` with trailing slash, holding camera.txt, pcalib.txt, vignette.png, times.txt and either
  // images/ or images.zip (reference :86-148).  Prints the reference's log lines.
  DatasetReader(std::string folder);
  ~DatasetReader();

  UndistorterFOV* getUndistorter();
  PhotometricUndistorter* getPhotoUndistorter();
  int getNumImages();
  double getTimestamp(int id);  // 0 outside the sequence
  float getExposure(int id);    // 0 outside the sequence

  // The frame `id`, processed as the four switches say (reference :188-243).  Caller owns the result.
  // 0 if the frame has the wrong size / cannot be decoded / the GPU pass fails (message on stdout /
  // stderr); there is no CPU fallback.
  ExposureImage* getImage(int id, bool rectify, bool removeGamma, bool removeVignette, bool nanOverexposed);

  // ---- beyond the reference's interface ----------------------------------------------------------
  // Frames first .. first+count-1 in one go: out[i] = what getImage(first+i, ...) returns (0 for a
  // frame that fails).  Decoding runs on the thread pool while earlier chunks are on the GPU.
  // Returns the number of images produced.
  int getImages(int first, int count, bool rectify, bool removeGamma, bool removeVignette, bool nanOverexposed,
                ExposureImage** out);

  // The same frames with a DEVICE-RESIDENT end (include/mdc_hip.h: mdc_device_outputs): files are parsed / decoded by the thread
  // pool, go up, and the processed frame -- plus, on request, box-pyramid levels 1..3 and DSO-style gradient images of every
  // level -- is left in the caller's device arrays, frame first+i at position i of each; nothing comes back over PCIe (getImages
  // tops out at ~31 k frames/s of 640x480 float results; this path is bound by the JPEG decode).  valid[i] (optional, `count`
  // bytes) = 1 where position i holds a result (0: wrong size / undecodable, as getImage's 0).  The arrays live on the device
  // of getDevice(): any device allocation there (hipMalloc, a framework tensor), or mdc_device_alloc(getContext(), ...).  Returns the number of frames
  // produced.  Same bytes as getImages followed by a copy to the device.  With several devices (MDC_DEVICES) the call runs on
  // the first one: shard with one reader per device, frame f on device f % N.
  int getImagesDevice(int first, int count, bool rectify, bool removeGamma, bool removeVignette, bool nanOverexposed,
                      const struct mdc_device_outputs* out, unsigned char* valid);
  struct mdc_ctx* getContext();  // the GPU context behind getImage / getImages / getImagesDevice (0 without a GPU)
  int getDevice() const;         // its HIP device ordinal (-1 without a GPU)

  // The decoded 8-bit frame (what cv::imread / cv::imdecode give the reference, :247-276); the
  // pointer stays valid until the next call on this object.  0 on failure.
  const unsigned char* getImageRaw(int id, int* width, int* height);
#ifdef CV_8U  // OpenCV (or the test shim) is on the include path: the reference's accessor (:247-276), same signature.
              // The Mat wraps the reader's own buffer: valid until the next call on this object (clone() to keep it).
  cv::Mat getImageRaw_internal(int id) {
    int w = 0, h = 0;
    const unsigned char* p = getImageRaw(id, &w, &h);
    return p ? cv::Mat(h, w, CV_8U, const_cast<unsigned char*>(p)) : cv::Mat();
  }
#endif

  void setDecodeThreads(int n);  // worker threads of the decode pool; 0 = automatic (default)
  void setPrefetch(int frames);  // frames decoded ahead after a getImage (default 16, 0 = off)
  // getImages on JPEG sequences: stage 2 (default): the decode threads only parse the file's markers and strip the byte
  // stuffing, Huffman decoding, inverse DCT and the fused pass run on the GPU (grayscale baseline files without restart
  // markers; others take stage 1); stage 1: the host Huffman-decodes, the inverse DCT runs on the GPU; stage 0: JPEG is decoded
  // on the host.  Same bytes in every stage.  setGpuJpeg(on) = stage 2 / 0; MDC_GPU_JPEG=0|1|2 in the environment.
  void setGpuJpeg(bool on);
  void setGpuJpegStage(int stage);
  // getImage on a JPEG sequence read in order (stage 2): from the third consecutive id on, the next results are made ahead in
  // one pass of the getImages pipeline with the caller's switches and handed out by the following calls: 64, then 128, then
  // 256 at a time while the caller keeps reading in order, never more than `frames` (default 256, 0 = off, also
  // MDC_READER_LOOKAHEAD in the environment).  Another id or other switches drop what was made ahead.  (256 results of 640x480
  // are 315 MB of page-locked images held by the reader until they are handed out.)
  void setResultLookahead(int frames);
  const char* lastError() const; // why the last getImage / getImages / getImageRaw returned 0 / fewer images
  void getPrefetchStats(long* hits, long* misses) const;  // frames found decoded ahead / decoded by the calling thread
  // Several GPUs (MDC_DEVICES=all | 0,1,... in the environment, read by the constructor): getImages -- and with it the results
  // getImage makes ahead -- deals its range to the devices in chunks of >= 64 frames, round-robin; every device holds the same
  // calibration tables (one RCCL broadcast over xGMI when libmdc_multi.so is there, else uploaded to each), has its own decode
  // ring and runs its GPU calls from its own host thread; results land in the caller's order, bit-identical to one device.
  int getDeviceCount() const;  // devices in use (1 without MDC_DEVICES)
  void getDeviceStats(int lane, int* device, long* frames, double* decoder_wait_s, double* gpu_call_s) const;  // over the reader's life

 private:
  DatasetReader(const DatasetReader&);
  DatasetReader& operator=(const DatasetReader&);
  int run_batch(int first, int count, bool rectify, bool removeGamma, bool removeVignette, bool nanOverexposed, ExposureImage** out,
                const struct mdc_device_outputs* dev, unsigned char* valid);
  struct State;
  State* s_;
};
