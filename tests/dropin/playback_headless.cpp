// Headless stand-in for the reference's playDataset tool (src/main_playbackDataset.cpp):
// same DatasetReader calls, no GUI.  It includes the reference's UNMODIFIED
// "BenchmarkDatasetReader.h" and is built twice by oracle/Makefile:
//   playback_ref : against the reference's own FOVUndistorter / PhotometricUndistorter
//   playback_mdc : against this repo's drop-in headers + libmdc_host.so (GPU path)
// Both write every requested getImage() result as raw floats; tests/test_dropin.py
// compares the two files byte for byte.  (BASELINE.json configs[0] plumbing.)
//
//   playback_X <sequence folder> <output file> <flags> [<flags> ...]
// where <flags> is a 4-character string of 0/1: rectify, removeGamma, removeVignette,
// nanOverexposed -- the keys r/g/v/o of the interactive viewer (:117-127).
#include <cstdio>
#include <cstring>
#include <string>

#include "BenchmarkDatasetReader.h"

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <sequence folder> <out file> <rgvo flags>...\n", argv[0]);
    return 2;
  }
  std::string folder = argv[1];
  if (folder.empty() || folder[folder.size() - 1] != '/') folder += "/";
  DatasetReader* reader = new DatasetReader(folder);  // as main_playbackDataset.cpp:56

  // the getters the viewer prints (:59-67)
  Eigen::Matrix3f K_rect = reader->getUndistorter()->getK_rect();
  Eigen::Vector2i dim_rect = reader->getUndistorter()->getOutputDims();
  Eigen::Vector2i dim_org = reader->getUndistorter()->getInputDims();
  float omega = reader->getUndistorter()->getOmega();
  std::printf("PLAYBACK %d images, %dx%d -> %dx%d, fx=%.6f omega=%.6f\n", reader->getNumImages(), dim_org[0],
              dim_org[1], dim_rect[0], dim_rect[1], K_rect(0, 0), omega);

  FILE* out = std::fopen(argv[2], "wb");
  if (!out) return 3;
  for (int a = 3; a < argc; a++) {
    const char* f = argv[a];
    if (std::strlen(f) != 4) return 2;
    for (int i = 0; i < reader->getNumImages(); i++) {
      ExposureImage* img = reader->getImage(i, f[0] == '1', f[1] == '1', f[2] == '1', f[3] == '1');
      if (!img) return 4;
      const int hdr[4] = {img->w, img->h, img->id, a};
      std::fwrite(hdr, sizeof hdr, 1, out);
      std::fwrite(&img->timestamp, sizeof(double), 1, out);
      std::fwrite(&img->exposure_time, sizeof(float), 1, out);
      std::fwrite(img->image, sizeof(float), (size_t)img->w * img->h, out);
      delete img;  // caller owns the ExposureImage (:82,:116)
    }
  }
  std::fclose(out);
  delete reader;
  return 0;
}
