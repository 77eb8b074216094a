// Host side of PhotometricUndistorter (drop-in for the reference's
// src/PhotometricUndistorter.cpp).
//
// Runs once per sequence on the CPU: pcalib.txt -> inverse response GInv (and the
// informational forward response G), vignette image -> vignetteMapInv, with the
// reference's arithmetic (file:line cited; compiled with -ffp-contract=off) so the
// tables are bit-identical (tests/test_tables_vs_ref.py).  unMapImage itself is a
// gfx950 kernel behind mdc_unmap_host (include/mdc_hip.h).
#include "PhotometricUndistorter.h"

#include <cstdio>
#include <fstream>
#include <sstream>
#include <vector>

#include "gray_png.h"
#include "host_device.h"
#include "mdc_hip.h"

namespace {

// First line of pcalib.txt as floats; formatted extraction stops at the first
// token that is not a number, like the istream_iterator the reference uses (:70-73).
std::vector<float> first_line_floats(std::ifstream& f) {
  std::string line;
  std::getline(f, line);
  std::istringstream ss(line);
  std::vector<float> v;
  float x;
  while (ss >> x) v.push_back(x);
  return v;
}

}  // namespace

void PhotometricUndistorter::read_calibration(const std::string& file, const std::string& vignetteImage) {
  if (file == "" || vignetteImage == "") return;

  // ---- response function ------------------------------------------------------
  std::ifstream f(file.c_str());
  std::printf("Reading Photometric Calibration from file %s\n", file.c_str());
  if (!f.good()) {
    std::printf("PhotometricUndistorter: Could not open file!\n");
    return;
  }
  const std::vector<float> raw = first_line_floats(f);
  if (raw.size() != 256) {
    std::printf("PhotometricUndistorter: invalid format! got %d entries in first line, expected 256!\n", (int)raw.size());
    return;
  }
  for (int i = 0; i < 255; i++)
    if (raw[i + 1] <= raw[i]) {
      std::printf("PhotometricUndistorter: G invalid! it has to be strictly increasing, but it isnt!\n");
      // the reference has already copied the raw values into GInv at this point (:79)
      for (int k = 0; k < 256; k++) ginv_[k] = raw[k];
      return;
    }
  // rescale so that 0..255 maps onto 0..255 (:89-91): double arithmetic, stored as float
  const float lo = raw[0], hi = raw[255];
  for (int i = 0; i < 256; i++) ginv_[i] = 255.0 * (raw[i] - lo) / (hi - lo);

  // forward response by bracketing search (:94-108); entries without a bracket
  // keep the 0 the constructor put there -- the reference leaves them uninitialised
  for (int i = 1; i < 255; i++)
    for (int s = 1; s < 255; s++)
      if (ginv_[s] <= i && ginv_[s + 1] >= i) {
        g_[i] = s + (i - ginv_[s]) / (ginv_[s + 1] - ginv_[s]);
        break;
      }
  g_[0] = 0;
  g_[255] = 255;
  valid_gamma_ = true;

  // ---- vignette -------------------------------------------------------------------
  std::printf("Reading Vignette Image from %s\n", vignetteImage.c_str());
  const mdc_host::GrayImage img = mdc_host::read_gray_image(vignetteImage);
  const int n = w_ * h_;
  vignette_ = new float[n];
  vignette_inv_ = new float[n];
  bool have_vignette = false;
  if (img.height != h_ || img.width != w_) {
    std::printf("PhotometricUndistorter: Invalid vignette image size! got %d x %d, expected %d x %d. Set vignette to 1.\n",
                img.width, img.height, w_, h_);
  } else if (img.bits != 8 && img.bits != 16) {
    // the reference asserts here (compiled out under NDEBUG) and carries on with
    // uninitialised maps; we refuse the image instead
    // (a colour, alpha or palette PNG: OpenCV would hand the reference a multi-channel Mat)
    std::printf("PhotometricUndistorter: ERROR: vignette image has %d channels, need 8- or 16-bit single-channel grayscale "
                "(the reference's behaviour for such a file is undefined). Set vignette to 1.\n", img.channels);
  } else {
    float peak = 0;  // (:130-147) same loop for 8- and 16-bit samples
    for (int i = 0; i < n; i++)
      if (img.px[i] > peak) peak = img.px[i];
    for (int i = 0; i < n; i++) vignette_[i] = img.px[i] / peak;
    for (int i = 0; i < n; i++) vignette_inv_[i] = 1.0f / vignette_[i];  // (:151-152)
    have_vignette = true;
  }
  if (!have_vignette)
    for (int i = 0; i < n; i++) vignette_[i] = vignette_inv_[i] = 1.0f;  // reference: uninitialised
  else {
    std::printf("Successfully read photometric calibration!\n");
    valid_vignette_ = true;
  }
}

PhotometricUndistorter::PhotometricUndistorter(std::string file, std::string vignetteImage, int w, int h)
    : vignette_(0), vignette_inv_(0), w_(w), h_(h), valid_vignette_(false), valid_gamma_(false), gpu_(0) {
  for (int i = 0; i < 256; i++) g_[i] = ginv_[i] = 0;
  read_calibration(file, vignetteImage);

  // one-time upload (also for an object without usable calibration: unMapImage then
  // degrades to the plain u8 -> float conversion, as the reference does)
  gpu_ = mdc_host::open_device_context("PhotometricUndistorter");
  if (gpu_ && mdc_set_photometric(gpu_, valid_gamma_ ? ginv_ : 0, valid_vignette_ ? vignette_inv_ : 0, w_, h_) != MDC_OK) {
    std::printf("PhotometricUndistorter: uploading the tables failed: %s\n", mdc_last_error(gpu_));
    mdc_destroy(gpu_);
    gpu_ = 0;
  }
}

PhotometricUndistorter::~PhotometricUndistorter() {
  if (gpu_) mdc_destroy(gpu_);
  delete[] vignette_;
  delete[] vignette_inv_;
}

void PhotometricUndistorter::unMapImage(unsigned char* image_in, float* image_out, int n, bool undoGamma,
                                        bool undoVignette, bool killOverexposed) {
  // the reference's notices, printed on every call (:173-189); the library applies
  // the same degradation to the flag word
  if (!valid_gamma_ && undoGamma)
    std::printf("Photometric Undistorter did not load Gamma correctly. correctly. Not undoing gamma!\n");
  if (!valid_vignette_ && undoVignette)
    std::printf("Photometric Undistorter did not load Vignette correctly. correctly. Not undoing Vignette!\n");
  if (!(valid_gamma_ && undoGamma) && (valid_vignette_ && undoVignette))
    std::printf("it doesn't make sense to undo vignette without undoing gamma! not doing neither.\n");

  const unsigned flags = (undoGamma ? MDC_GAMMA : 0u) | (undoVignette ? MDC_VIGNETTE : 0u) |
                         (killOverexposed ? MDC_KILL_OVEREXPOSED : 0u);
  if (!gpu_) {
    std::fprintf(stderr, "ERROR: PhotometricUndistorter::unMapImage needs a gfx950 GPU (no HIP device context); output not written\n");
    return;
  }
  if (mdc_unmap_host(gpu_, image_in, image_out, n, flags) != MDC_OK)
    std::fprintf(stderr, "ERROR: PhotometricUndistorter::unMapImage failed on the GPU: %s\n", mdc_last_error(gpu_));
}
