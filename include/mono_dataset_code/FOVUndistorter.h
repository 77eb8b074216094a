// Drop-in for the reference's src/FOVUndistorter.h:36-96 (class UndistorterFOV).
//
// Same public surface -- constructor from camera.txt, undistort<T> for float and
// unsigned char, distortCoordinates, the K / calibration / dimension getters,
// isValid -- so src/BenchmarkDatasetReader.h, main_playbackDataset.cpp and
// main_vignetteCalib.cpp compile against it unchanged.  Differences are inside:
// the remap tables are built on the host with the reference's exact float
// operation sequence (they must be bit-identical) and uploaded
// once to the GPU; undistort<T> runs as a HIP kernel on gfx950 through the C ABI
// in include/mdc_hip.h.  There is no CPU fallback for the per-frame work.
#pragma once
#include "ExposureImage.h"
#include "Eigen/Core"

struct mdc_ctx;
struct MdcHostAccess;

#ifndef MDC_API  /* the libraries are built with -fvisibility=hidden: this marks what they export */
#if defined(__GNUC__) || defined(__clang__)
#define MDC_API __attribute__((visibility("default")))
#else
#define MDC_API
#endif
#endif
class MDC_API UndistorterFOV {
 public:
  UndistorterFOV(const char* configFileName);           // parses camera.txt, builds + uploads the remap
  UndistorterFOV();                                     // invalid object (reference :39-44)
  ~UndistorterFOV();

  // Bilinear warp of a in_w*in_h image into out_w*out_h floats (host pointers,
  // blocking).  Silent no-op on an invalid object; prints ERROR and returns on a
  // pixel-count mismatch -- as the reference (src/FOVUndistorter.cpp:325-338).
  template <typename T>
  void undistort(const T* input, float* output, int nPixIn, int nPixOut) const;

  // In place: rectified pixel coordinates -> raw (distorted) pixel coordinates.
  void distortCoordinates(float* in_x, float* in_y, int n);

  Eigen::Matrix3f getK_rect() const { return k_rect_; }
  Eigen::Matrix3f getK_org() const { return k_org_; }
  float getOmega() const { return calib_in_[4]; }
  const Eigen::VectorXf getOriginalCalibration() const {
    Eigen::VectorXf c(5);
    c[0] = calib_in_[0] * in_w_;
    c[1] = calib_in_[1] * in_h_;
    c[2] = calib_in_[2] * in_w_ - 0.5;
    c[3] = calib_in_[3] * in_h_ - 0.5;
    c[4] = calib_in_[4];
    return c;
  }
  const Eigen::Vector2i getInputDims() const { return Eigen::Vector2i(in_w_, in_h_); }
  const Eigen::Vector2i getOutputDims() const { return Eigen::Vector2i(out_w_, out_h_); }
  bool isValid() const { return valid_; }

 private:
  UndistorterFOV(const UndistorterFOV&);             // owning raw tables: not copyable
  UndistorterFOV& operator=(const UndistorterFOV&);
  friend struct MdcHostAccess;

  Eigen::Matrix3f k_rect_, k_org_;
  float calib_in_[5];   // camera.txt line 1: fx fy cx cy omega, relative to the input size
  float calib_out_[5];  // normalised output intrinsics after construction
  int in_w_, in_h_, out_w_, out_h_;
  float* remap_x_;      // out_w*out_h source x coordinates, -1 = black
  float* remap_y_;
  bool valid_;
  mdc_ctx* gpu_;        // device context holding the uploaded remap (0 if no GPU)
};

// The task description calls this class "Undistorter"; the reference names it
// UndistorterFOV.  Both spellings work.
typedef UndistorterFOV Undistorter;
