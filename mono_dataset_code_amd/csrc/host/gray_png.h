// Minimal grayscale PNG / PGM reader for the vignette image (host side).
// Stands in for cv::imread(vignetteImage, CV_LOAD_IMAGE_UNCHANGED) at
// src/PhotometricUndistorter.cpp:120 of the reference, so the library does not
// drag OpenCV in for one file read.
#pragma once
#include <string>
#include <vector>

namespace mdc_host {

struct GrayImage {
  int width = 0, height = 0;
  int channels = 0;                  // channels of the file as cv::imread(..., UNCHANGED) would deliver them (0 = unreadable)
  int bits = 0;                      // 8 or 16; 0 = could not be read / not single-channel
  std::vector<unsigned short> px;    // width*height samples, host endian (8-bit values as-is)
};

// Any PNG flavour (every bit depth, interlacing) and PGM.  bits == 0: the file is missing or corrupt (width == height ==
// 0) or has more than one channel (size and `channels` set: the caller reports it -- the reference's behaviour for such a
// vignette is undefined, src/PhotometricUndistorter.cpp:148 asserts and NDEBUG compiles that out).
GrayImage read_gray_image(const std::string& path);

}  // namespace mdc_host
