// TEST INFRASTRUCTURE ONLY -- extern "C" handles onto the *reference's own*
// classes, compiled from /root/reference/src/{FOVUndistorter,PhotometricUndistorter}.cpp
// where they lie (see oracle/Makefile).  Loaded with ctypes by tests/ and by
// bench.py's cpu_baseline leg; the product never links or loads this.
//
// Class names are renamed on the compile line (-DUndistorterFOV=RefUndistorterFOV
// -DPhotometricUndistorter=RefPhotometricUndistorter) so this object can sit in
// one process with the product's drop-in classes of the same name.
#include <string>
#include <vector>
#include <thread>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <sstream>
#include <fstream>
#include <opencv2/core/core.hpp>
#include "Eigen/Core"

// Read-only access to the reference's tables (remapX/remapY, vignetteMapInv)
// for bitwise table-parity tests.  Access control only; layout is unchanged.
#define private public
#include "FOVUndistorter.h"
#include "PhotometricUndistorter.h"
#undef private

extern "C" {

void* ref_fov_create(const char* camera_txt) { return new UndistorterFOV(camera_txt); }
void ref_fov_destroy(void* p) { delete (UndistorterFOV*)p; }
int ref_fov_valid(void* p) { return ((UndistorterFOV*)p)->isValid() ? 1 : 0; }

void ref_fov_dims(void* p, int* d4) {
  UndistorterFOV* u = (UndistorterFOV*)p;
  d4[0] = u->getInputDims()[0]; d4[1] = u->getInputDims()[1];
  d4[2] = u->getOutputDims()[0]; d4[3] = u->getOutputDims()[1];
}

// Krect(9) Korg(9) originalCalibration(5) omega(1) outputCalibration(5) = 29 floats
void ref_fov_intrinsics(void* p, float* o) {
  UndistorterFOV* u = (UndistorterFOV*)p;
  Eigen::Matrix3f a = u->getK_rect(), b = u->getK_org();
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { o[r * 3 + c] = a(r, c); o[9 + r * 3 + c] = b(r, c); }
  Eigen::VectorXf v = u->getOriginalCalibration();
  for (int i = 0; i < 5; i++) o[18 + i] = v[i];
  o[23] = u->getOmega();
  for (int i = 0; i < 5; i++) o[24 + i] = u->outputCalibration[i];
}

// copies remapX/remapY (out_w*out_h each); returns 0 if the object holds none
int ref_fov_remap(void* p, float* rx, float* ry) {
  UndistorterFOV* u = (UndistorterFOV*)p;
  if (!u->valid || !u->remapX) return 0;
  size_t n = (size_t)u->out_width * u->out_height;
  memcpy(rx, u->remapX, n * 4); memcpy(ry, u->remapY, n * 4);
  return 1;
}

void ref_fov_distort(void* p, float* x, float* y, int n) { ((UndistorterFOV*)p)->distortCoordinates(x, y, n); }
void ref_fov_undistort_f32(void* p, const float* in, float* out, int nin, int nout) {
  ((UndistorterFOV*)p)->undistort<float>(in, out, nin, nout);
}
void ref_fov_undistort_u8(void* p, const unsigned char* in, float* out, int nin, int nout) {
  ((UndistorterFOV*)p)->undistort<unsigned char>(in, out, nin, nout);
}

void* ref_photo_create(const char* pcalib, const char* vignette, int w, int h) {
  return new PhotometricUndistorter(pcalib, vignette, w, h);
}
void ref_photo_destroy(void* p) { delete (PhotometricUndistorter*)p; }
// bit0 = validGamma, bit1 = validVignette
int ref_photo_valid(void* p) {
  PhotometricUndistorter* u = (PhotometricUndistorter*)p;
  return (u->validGamma ? 1 : 0) | (u->validVignette ? 2 : 0);
}
int ref_photo_ginv(void* p, float* o) {
  float* g = ((PhotometricUndistorter*)p)->getGInv();
  if (!g) return 0;
  memcpy(o, g, 256 * 4);
  return 1;
}
int ref_photo_g(void* p, float* o) {
  float* g = ((PhotometricUndistorter*)p)->getG();
  if (!g) return 0;
  memcpy(o, g, 256 * 4);
  return 1;
}
int ref_photo_vignette(void* p, float* map, float* inv) {
  PhotometricUndistorter* u = (PhotometricUndistorter*)p;
  if (!u->validVignette) return 0;
  size_t n = (size_t)u->w * u->h;
  if (map) memcpy(map, u->vignetteMap, n * 4);
  if (inv) memcpy(inv, u->vignetteMapInv, n * 4);
  return 1;
}
void ref_photo_unmap(void* p, unsigned char* in, float* out, int n, int g, int v, int o) {
  ((PhotometricUndistorter*)p)->unMapImage(in, out, n, g != 0, v != 0, o != 0);
}

// The composition site DatasetReader::getImage (BenchmarkDatasetReader.h:207-241)
// driven on a raw u8 frame instead of a decoded JPEG.  `tmp` is the reader's
// internalTempBuffer (in_w*in_h floats).  Calls only reference code.
void ref_get_image(void* fov, void* photo, unsigned char* raw, float* out, float* tmp,
                   int rectify, int g, int v, int o) {
  UndistorterFOV* u = (UndistorterFOV*)fov;
  PhotometricUndistorter* ph = (PhotometricUndistorter*)photo;
  int W = u->getInputDims()[0], H = u->getInputDims()[1];
  int w = u->getOutputDims()[0], h = u->getOutputDims()[1];
  if (g || v || o) {
    if (!rectify) ph->unMapImage(raw, out, W * H, g != 0, v != 0, o != 0);
    else {
      ph->unMapImage(raw, tmp, W * H, g != 0, v != 0, o != 0);
      u->undistort<float>(tmp, out, W * H, w * h);
    }
  } else {
    if (rectify) u->undistort<unsigned char>(raw, out, W * H, w * h);
    else for (int i = 0; i < W * H; i++) out[i] = raw[i];
  }
}

// CPU baseline: the reference path timed as getImage drives it
// (unMapImage -> temp -> undistort<float>), `nthreads` host threads sharing the
// const tables with per-thread temp/out buffers, frames sharded round-robin.
// `frames` holds nframes raw frames back to back; each thread walks its share
// `passes` times.  Returns wall seconds; *checksum defeats dead-code removal.
double ref_time_path(void* fov, void* photo, unsigned char* frames, int nframes, int nthreads,
                     int passes, int rectify, int g, int v, int o, double* checksum) {
  UndistorterFOV* u = (UndistorterFOV*)fov;
  int W = u->getInputDims()[0], H = u->getInputDims()[1];
  int w = u->getOutputDims()[0], h = u->getOutputDims()[1];
  size_t nout = rectify ? (size_t)w * h : (size_t)W * H;
  std::vector<double> sums((size_t)nthreads, 0.0);
  auto work = [&](int t) {
    std::vector<float> tmp((size_t)W * H), out(nout);
    double s = 0;
    for (int p = 0; p < passes; p++)
      for (int f = t; f < nframes; f += nthreads) {
        ref_get_image(fov, photo, frames + (size_t)f * W * H, out.data(), tmp.data(), rectify, g, v, o);
        float x = out[(size_t)(f * 7919) % nout];
        if (x == x) s += x;
      }
    sums[(size_t)t] = s;
  };
  auto t0 = std::chrono::steady_clock::now();
  if (nthreads <= 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  auto t1 = std::chrono::steady_clock::now();
  double s = 0;
  for (double x : sums) s += x;
  if (checksum) *checksum = s;
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
