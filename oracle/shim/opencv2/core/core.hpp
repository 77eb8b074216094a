// TEST INFRASTRUCTURE ONLY -- stand-in for <opencv2/core/core.hpp>.
//
// OpenCV is not installed in this image.  The two hot-path translation units
// of the reference (src/PhotometricUndistorter.cpp, src/FOVUndistorter.cpp)
// touch OpenCV only to decode the vignette PNG (cv::imread,
// src/PhotometricUndistorter.cpp:120) and to index the decoded pixels
// (cv::Mat::at<T>(int), :134,:137,:143,:146).  This header declares just that
// much so the reference sources compile *unmodified, where they lie* into
// oracle/_ref/.  Nothing in the product includes this file.
#pragma once
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cassert>
#include <algorithm>
#include <memory>

typedef unsigned short ushort;
typedef unsigned char uchar;

#define CV_8U 0
#define CV_16U 2
#define CV_8UC3 16
#define CV_LOAD_IMAGE_UNCHANGED -1
#define CV_LOAD_IMAGE_GRAYSCALE 0

namespace cv {

// Continuous, single-channel, row-major matrix; at<T>(i) is a flat index,
// which is how OpenCV resolves Mat::at<T>(int) on a continuous matrix.
struct Mat {
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  int type_ = CV_8U;
  std::shared_ptr<std::vector<uchar>> store;  // owning storage (imread)

  Mat() {}
  // non-owning header over caller memory (BenchmarkDatasetReader.h:274)
  Mat(long r, int c, int t, void* p) : rows((int)r), cols(c), data((uchar*)p), type_(t) {}

  int type() const { return type_; }
  template <class T> T& at(int i) { return ((T*)data)[i]; }
  template <class T> const T& at(int i) const { return ((const T*)data)[i]; }
};

// Implemented in oracle/shim_imread.cpp (libpng; also reads binary PGM).
Mat imread(const std::string& file, int flags);
Mat imdecode(const Mat& buf, int flags);

}  // namespace cv
