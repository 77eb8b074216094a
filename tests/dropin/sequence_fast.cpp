// The fast path of INTEGRATION.md section B as a program: the reference's UNMODIFIED reader
// (BenchmarkDatasetReader.h) still lists and decodes the frames and owns the two calibration
// objects, but instead of calling getImage() frame by frame (unMapImage -> W*H floats -> undistort)
// the whole sequence goes through ONE mdc_process_frames_host call per flag set: frames and
// results live in page-locked memory (mdc_host_alloc), uploads, kernels and downloads overlap.
// Output file format = playback_headless.cpp's, so tests/test_dropin.py can compare it byte for
// byte with the reference's own getImage() results.
//
//   sequence_fast <sequence folder> <output file> <rgvo flags>...
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "BenchmarkDatasetReader.h"
#include "MdcBind.h"

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <sequence folder> <out file> <rgvo flags>...\n", argv[0]);
    return 2;
  }
  std::string folder = argv[1];
  if (folder.empty() || folder[folder.size() - 1] != '/') folder += "/";
  DatasetReader* reader = new DatasetReader(folder);
  const Eigen::Vector2i dim_in = reader->getUndistorter()->getInputDims();
  const Eigen::Vector2i dim_out = reader->getUndistorter()->getOutputDims();
  const int n = reader->getNumImages();
  const size_t n_in = (size_t)dim_in[0] * dim_in[1], n_rect = (size_t)dim_out[0] * dim_out[1];

  mdc_ctx* gpu = 0;
  if (mdc_create(-1, &gpu) != MDC_OK) {
    std::fprintf(stderr, "mdc_create: %s\n", mdc_last_error(0));
    return 5;
  }
  if (mdc_bind_objects(gpu, reader->getUndistorter(), reader->getPhotoUndistorter()) != MDC_OK) {
    std::fprintf(stderr, "bind: %s\n", mdc_last_error(gpu));
    return 5;
  }

  // decode once (the reader's own decoder), into one page-locked block
  unsigned char* raw = (unsigned char*)mdc_host_alloc(n_in * n);
  float* res = (float*)mdc_host_alloc(std::max(n_in, n_rect) * sizeof(float) * n);
  if (!raw || !res) return 6;
  std::vector<const uint8_t*> raw_ptr(n);
  for (int i = 0; i < n; i++) {
    cv::Mat m = reader->getImageRaw_internal(i);
    if (m.rows * m.cols != (int)n_in || m.type() != CV_8U) return 4;
    std::memcpy(raw + (size_t)i * n_in, m.data, n_in);
    raw_ptr[i] = raw + (size_t)i * n_in;
  }
  std::printf("SEQUENCE_FAST %d images, %dx%d -> %dx%d\n", n, dim_in[0], dim_in[1], dim_out[0], dim_out[1]);

  FILE* out = std::fopen(argv[2], "wb");
  if (!out) return 3;
  for (int a = 3; a < argc; a++) {
    const char* f = argv[a];
    if (std::strlen(f) != 4) return 2;
    const bool rect = f[0] == '1';
    const unsigned flags = (rect ? MDC_RECTIFY : 0) | (f[1] == '1' ? MDC_GAMMA : 0) | (f[2] == '1' ? MDC_VIGNETTE : 0) |
                           (f[3] == '1' ? MDC_KILL_OVEREXPOSED : 0);
    const size_t n_res = rect ? n_rect : n_in;
    std::vector<float*> res_ptr(n);
    for (int i = 0; i < n; i++) res_ptr[i] = res + (size_t)i * n_res;
    if (mdc_process_frames_host(gpu, raw_ptr.data(), res_ptr.data(), n, flags) != MDC_OK) {
      std::fprintf(stderr, "mdc_process_frames_host: %s\n", mdc_last_error(gpu));
      return 7;
    }
    for (int i = 0; i < n; i++) {
      const int hdr[4] = {rect ? dim_out[0] : dim_in[0], rect ? dim_out[1] : dim_in[1], i, a};
      const double ts = reader->getTimestamp(i);
      const float ex = reader->getExposure(i);
      std::fwrite(hdr, sizeof hdr, 1, out);
      std::fwrite(&ts, sizeof(double), 1, out);
      std::fwrite(&ex, sizeof(float), 1, out);
      std::fwrite(res_ptr[i], sizeof(float), n_res, out);
    }
  }
  std::fclose(out);
  mdc_host_free(raw);
  mdc_host_free(res);
  mdc_destroy(gpu);
  delete reader;
  return 0;
}
