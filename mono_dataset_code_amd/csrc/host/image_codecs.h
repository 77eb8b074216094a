// Frame decoders of the dataset reader (host side): 8-bit grayscale out of PNG, PGM (P5) and
// baseline JPEG byte streams -- what cv::imread / cv::imdecode(..., CV_LOAD_IMAGE_GRAYSCALE)
// deliver to DatasetReader::getImageRaw_internal in the reference
// (src/BenchmarkDatasetReader.h:247-276), without OpenCV.
//
// All decoders are re-entrant (no global state): the reader's decode pool calls them from many
// threads at once.  The JPEG decoder is a from-scratch baseline (sequential DCT, Huffman, 8-bit)
// decoder with libjpeg's "islow" integer inverse DCT, so a grayscale JPEG decodes to the same
// bytes as libjpeg / libjpeg-turbo (what OpenCV links); of a YCbCr JPEG only the luma plane is
// reconstructed, which is libjpeg's own JCS_GRAYSCALE output.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

namespace mdc_host {

// Decodes `n` bytes at `data` into `out` (row-major, *w x *h bytes, at most `cap` bytes).
// Returns false and sets *err for unsupported / corrupt input or when cap is too small (then *w
// and *h still hold the image size if the header could be read, else 0).
bool decode_gray8(const unsigned char* data, size_t n, unsigned char* out, size_t cap, int* w, int* h, std::string* err);

// Whole file into memory; false if it cannot be read.
bool read_file(const std::string& path, std::vector<unsigned char>& buf);

}  // namespace mdc_host
