// playback_headless.cpp's twin for the batched entry point of this repo's reader
// (DatasetReader::getImages, include/mono_dataset_code/BenchmarkDatasetReader.h): the whole sequence per flag set
// in ONE call -- decode pool, page-locked ring, pipelined GPU chunks -- written in the same file format, so that
// tests/test_reader.py can compare it byte for byte with the reference reader's frame-by-frame getImage().
//
//   playback_batch <sequence folder> <output file> <rgvo flags>...
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "BenchmarkDatasetReader.h"

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <sequence folder> <out file> <rgvo flags>...\n", argv[0]);
    return 2;
  }
  std::string folder = argv[1];
  if (folder.empty() || folder[folder.size() - 1] != '/') folder += "/";
  DatasetReader reader(folder);
  const int n = reader.getNumImages();
  std::printf("PLAYBACK_BATCH %d images\n", n);
  FILE* out = std::fopen(argv[2], "wb");
  if (!out) return 3;
  std::vector<ExposureImage*> imgs((size_t)n);
  for (int a = 3; a < argc; a++) {
    const char* f = argv[a];
    if (std::strlen(f) != 4) return 2;
    if (reader.getImages(0, n, f[0] == '1', f[1] == '1', f[2] == '1', f[3] == '1', imgs.data()) != n) {
      std::fprintf(stderr, "getImages: %s\n", reader.lastError());
      return 4;
    }
    for (int i = 0; i < n; i++) {
      ExposureImage* img = imgs[(size_t)i];
      const int hdr[4] = {img->w, img->h, img->id, a};
      std::fwrite(hdr, sizeof hdr, 1, out);
      std::fwrite(&img->timestamp, sizeof(double), 1, out);
      std::fwrite(&img->exposure_time, sizeof(float), 1, out);
      std::fwrite(img->image, sizeof(float), (size_t)img->w * img->h, out);
      delete img;
    }
  }
  std::fclose(out);
  return 0;
}
