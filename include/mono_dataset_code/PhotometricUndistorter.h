// Drop-in for the reference's src/PhotometricUndistorter.h:37-54.
//
// Same public surface (constructor from pcalib.txt + vignette image + frame size,
// unMapImage, getGInv, getG).  The constructor builds GInv / G / vignetteMapInv on
// the host exactly as the reference does and uploads them once; unMapImage runs
// as a HIP kernel on gfx950 through the C ABI in include/mdc_hip.h.  There is no
// CPU fallback for the per-frame work.
#pragma once
#include <string>
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
class MDC_API PhotometricUndistorter {
 public:
  PhotometricUndistorter(std::string file, std::string vignetteImage, int w, int h);
  ~PhotometricUndistorter();

  // image_out[i] = GInv[image_in[i]] * vignetteMapInv[i] (per the three flags),
  // raw 255 -> NaN with killOverexposed.  Host pointers, blocking.  Prints the
  // reference's notices when a requested table is missing.
  void unMapImage(unsigned char* image_in, float* image_out, int n, bool undoGamma, bool undoVignette,
                  bool killOverexposed);

  float* getGInv() { return valid_gamma_ ? ginv_ : 0; }
  float* getG() { return valid_gamma_ ? g_ : 0; }

 private:
  PhotometricUndistorter(const PhotometricUndistorter&);
  PhotometricUndistorter& operator=(const PhotometricUndistorter&);
  friend struct MdcHostAccess;
  void read_calibration(const std::string& file, const std::string& vignetteImage);

  float g_[256];     // forward response (intensity -> raw), informational
  float ginv_[256];  // inverse response, rescaled to 0..255
  float* vignette_;      // w*h, normalised to max 1
  float* vignette_inv_;  // w*h, reciprocal
  int w_, h_;
  bool valid_vignette_, valid_gamma_;
  mdc_ctx* gpu_;
};
