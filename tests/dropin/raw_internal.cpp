// responseCalib-style use of the reader (reference src/main_responseCalib.cpp:183-200): only the decoded raw frames are
// wanted, through getImageRaw_internal(int) -> cv::Mat.  This translation unit includes ONLY this repo's reader header --
// no OpenCV header before it -- and is compiled with the OpenCV stand-in on the include path: the accessor must be there.
//   raw_internal <sequence folder>   prints one line per frame: index rows cols type sum-of-bytes
#include <cstdio>
#include <string>

#include "BenchmarkDatasetReader.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::string folder = argv[1];
  if (folder[folder.size() - 1] != '/') folder += "/";
  DatasetReader* reader = new DatasetReader(folder);
  for (int i = 0; i < reader->getNumImages(); i++) {
    cv::Mat img = reader->getImageRaw_internal(i);
    if (img.rows == 0 || img.cols == 0 || img.type() != CV_8U) {
      std::printf("RAW %d failed\n", i);
      continue;
    }
    unsigned long sum = 0;
    for (long k = 0; k < (long)img.rows * img.cols; k++) sum += img.at<unsigned char>(k);
    std::printf("RAW %d %d %d %d %lu\n", i, img.rows, img.cols, img.type(), sum);
  }
  delete reader;
  return 0;
}
