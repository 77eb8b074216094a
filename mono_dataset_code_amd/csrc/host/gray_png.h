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
  int bits = 0;                      // 8 or 16; 0 = could not be read / not grayscale
  std::vector<unsigned short> px;    // width*height samples, host endian (8-bit values as-is)
};

// Returns an image with bits == 0 and width == height == 0 when the file is
// missing, corrupt, interlaced or not 8/16-bit single-channel grayscale.
GrayImage read_gray_image(const std::string& path);

}  // namespace mdc_host
