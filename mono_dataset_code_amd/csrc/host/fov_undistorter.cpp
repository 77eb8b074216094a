// Host side of UndistorterFOV (drop-in for the reference's src/FOVUndistorter.cpp).
//
// Everything here runs once per sequence on the CPU: camera.txt parsing, choice
// of the rectified intrinsics, and the remapX/remapY tables.  The tables are the
// sensitive part of the whole path -- a 1-ulp difference in a remap entry moves
// the output by up to 4e-2 relative on a noisy frame -- so they are computed with
// the host libm in single precision with exactly the operation sequence of the
// reference (file:line cited at each step; this translation unit is compiled
// with -ffp-contract=off) and compared bit-for-bit against the reference build
// in tests/test_tables_vs_ref.py.  The per-frame warp itself is NOT here: it is a
// gfx950 kernel behind mdc_undistort_host_* (include/mdc_hip.h).
#include "FOVUndistorter.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "mdc_hip.h"
#include "host_device.h"
#include "../fov_point_model.h"  // atanf_host_libm: the restatement the device runs, checked against THIS process's libm below

namespace {

// The four text lines of camera.txt, decoded (reference :63-123).
struct CameraFile {
  float in[5];   // fx fy cx cy omega (relative to the input size)
  int in_w, in_h;
  int mode;      // kCrop, kFull or kExplicit
  float out[5];  // only for kExplicit
  int out_w, out_h;
};
enum { kExplicit = 0, kCrop = -1, kFull = -2 };
enum ParseResult { kParsed, kUnreadable, kBadHeader, kNoRectification, kBadOutputPars, kBadOutputSize };

ParseResult parse_camera_file(const char* path, CameraFile& c) {
  std::ifstream f(path);
  if (!f.good()) return kUnreadable;
  std::string line[4];
  for (int i = 0; i < 4; i++) std::getline(f, line[i]);

  const bool head = std::sscanf(line[0].c_str(), "%f %f %f %f %f", &c.in[0], &c.in[1], &c.in[2], &c.in[3], &c.in[4]) == 5 &&
                    std::sscanf(line[1].c_str(), "%d %d", &c.in_w, &c.in_h) == 2;
  if (!head) return kBadHeader;
  std::printf("Input resolution: %d %d\n", c.in_w, c.in_h);
  std::printf("Input Calibration (fx fy cx cy): %f %f %f %f %f\n", c.in_w * c.in[0], c.in_h * c.in[1], c.in_w * c.in[2],
              c.in_h * c.in[3], c.in[4]);

  if (line[2] == "crop") {
    c.mode = kCrop;
    std::printf("Out: Crop\n");
  } else if (line[2] == "full") {
    c.mode = kFull;
    std::printf("Out: Full\n");
  } else if (line[2] == "none") {
    std::printf("NO RECTIFICATION\n");
    return kNoRectification;
  } else if (std::sscanf(line[2].c_str(), "%f %f %f %f %f", &c.out[0], &c.out[1], &c.out[2], &c.out[3], &c.out[4]) == 5) {
    c.mode = kExplicit;
    std::printf("Out: %f %f %f %f %f\n", c.out[0], c.out[1], c.out[2], c.out[3], c.out[4]);
  } else {
    std::printf("Out: Failed to Read Output pars... not rectifying.\n");
    return kBadOutputPars;
  }

  if (std::sscanf(line[3].c_str(), "%d %d", &c.out_w, &c.out_h) != 2) {
    std::printf("Out: Failed to Read Output resolution... not rectifying.\n");
    return kBadOutputSize;
  }
  std::printf("Output resolution: %d %d\n", c.out_w, c.out_h);
  return kParsed;
}

// Input pinhole parameters in pixels + the FOV-model constant (reference :131-138 / :289-296).
struct InputModel {
  float omega, d2t, fx, fy, cx, cy;
  InputModel(const float calib[5], int w, int h) {
    omega = calib[4];
    d2t = 2.0f * ::tan((double)(omega / 2.0f));  // double-precision tan (as the reference build binds it), narrowed
    fx = calib[0] * w;
    fy = calib[1] * h;
    cx = calib[2] * w - 0.5;         // double subtraction, narrowed
    cy = calib[3] * h - 0.5;
  }
};

inline float fmax_std(float a, float b) { return a < b ? b : a; }  // std::max

// Undistorted radius of a distorted radius (tan in double, quotient narrowed).
inline float undistorted_radius(float r, const InputModel& m) { return ::tan((double)(r * m.omega)) / m.d2t; }

// Rectified intrinsics (pixels) for the three output modes + the omega == 0 case
// (reference :141-212), returned normalised by the output size (:214-218).
void pick_output_intrinsics(const CameraFile& c, float norm[5]) {
  const InputModel m(c.in, c.in_w, c.in_h);
  float ofx, ofy, ocx, ocy;
  if (c.in[4] == 0) {  // pinhole input: same relative intrinsics at the new size
    ofx = c.in[0] * c.out_w;
    ofy = c.in[1] * c.out_h;
    ocx = (c.in[2] * c.out_w) - 0.5;
    ocy = (c.in[3] * c.out_h) - 0.5;
  } else if (c.mode == kCrop || c.mode == kFull) {
    const float left = m.cx / m.fx;
    const float right = (c.in_w - 1 - m.cx) / m.fx;
    const float top = m.cy / m.fy;
    const float bottom = (c.in_h - 1 - m.cy) / m.fy;
    const float sx = (float)c.out_w / (float)c.in_w, sy = (float)c.out_h / (float)c.in_h;
    if (c.mode == kCrop) {  // largest rectangle with no black pixels: use the edge mid-points
      const float t_left = undistorted_radius(left, m), t_right = undistorted_radius(right, m);
      const float t_top = undistorted_radius(top, m), t_bottom = undistorted_radius(bottom, m);
      ofy = m.fy * ((top + bottom) / (t_top + t_bottom)) * sy;
      ocy = (t_top / top) * ofy * m.cy / m.fy;
      ofx = m.fx * ((left + right) / (t_left + t_right)) * sx;
      ocx = (t_left / left) * ofx * m.cx / m.fx;
    } else {  // every input pixel visible: use the corners
      const float tl = ::sqrt((double)(left * left + top * top)), tr = ::sqrt((double)(right * right + top * top));
      const float bl = ::sqrt((double)(left * left + bottom * bottom)), br = ::sqrt((double)(right * right + bottom * bottom));
      const float t_tl = undistorted_radius(tl, m), t_tr = undistorted_radius(tr, m);
      const float t_bl = undistorted_radius(bl, m), t_br = undistorted_radius(br, m);
      const float hor = fmax_std(br, tr) + fmax_std(bl, tl);
      const float vert = fmax_std(tr, tl) + fmax_std(bl, br);
      const float t_hor = fmax_std(t_br, t_tr) + fmax_std(t_bl, t_tl);
      const float t_vert = fmax_std(t_tr, t_tl) + fmax_std(t_bl, t_br);
      ofy = m.fy * (vert / t_vert) * sy;
      ocy = fmax_std(t_tl / tl, t_tr / tr) * ofy * m.cy / m.fy;
      ofx = m.fx * (hor / t_hor) * sx;
      ocx = fmax_std(t_bl / bl, t_tl / tl) * ofx * m.cx / m.fx;
    }
    std::printf("new K: %f %f %f %f\n", ofx, ofy, ocx, ocy);
    std::printf("old K: %f %f %f %f\n", m.fx, m.fy, m.cx, m.cy);
  } else {
    ofx = c.out[0] * c.out_w;
    ofy = c.out[1] * c.out_h;
    ocx = c.out[2] * c.out_w - 0.5;
    ocy = c.out[3] * c.out_h - 0.5;
  }
  norm[0] = ofx / c.out_w;
  norm[1] = ofy / c.out_h;
  norm[2] = (ocx + 0.5) / c.out_w;
  norm[3] = (ocy + 0.5) / c.out_h;
  norm[4] = 0;
}

// Rectified pixel -> raw pixel through the FOV (atan) lens model, in place
// (reference :289-318).  Single precision, host libm sqrtf/atanf.
void warp_points(const float calib_in[5], int in_w, int in_h, const float calib_out[5], int out_w, int out_h, float* xs,
                 float* ys, int n) {
  const InputModel m(calib_in, in_w, in_h);
  const float ofx = calib_out[0] * out_w, ofy = calib_out[1] * out_h;
  const float ocx = calib_out[2] * out_w - 0.5f, ocy = calib_out[3] * out_h - 0.5f;
  for (int i = 0; i < n; i++) {
    float ix = (xs[i] - ocx) / ofx;
    float iy = (ys[i] - ocy) / ofy;
    const float r = sqrtf(ix * ix + iy * iy);
    const float fac = (r == 0 || m.omega == 0) ? 1 : atanf(r * m.d2t) / (m.omega * r);
    ix = m.fx * fac * ix + m.cx;
    iy = m.fy * fac * iy + m.cy;
    xs[i] = ix;
    ys[i] = iy;
  }
}

void set_pinhole(Eigen::Matrix3f& k, const float rel[5], int w, int h) {
  k.setIdentity();
  k(0, 0) = rel[0] * w;
  k(1, 1) = rel[1] * h;
  k(0, 2) = rel[2] * w - 0.5;
  k(1, 2) = rel[3] * h - 0.5;
}

}  // namespace

UndistorterFOV::UndistorterFOV()
    : in_w_(0), in_h_(0), out_w_(0), out_h_(0), remap_x_(0), remap_y_(0), valid_(false), gpu_(0) {
  for (int i = 0; i < 5; i++) calib_in_[i] = calib_out_[i] = 0;
}

UndistorterFOV::UndistorterFOV(const char* configFileName)
    : in_w_(0), in_h_(0), out_w_(0), out_h_(0), remap_x_(0), remap_y_(0), valid_(false), gpu_(0) {
  for (int i = 0; i < 5; i++) calib_in_[i] = calib_out_[i] = 0;

  CameraFile cam = CameraFile();
  const ParseResult pr = parse_camera_file(configFileName, cam);
  if (pr == kUnreadable || pr == kBadHeader) {
    std::printf("Failed to read camera calibration (invalid format?)\nCalibration file: %s\n", configFileName);
    return;
  }
  // the reference keeps whatever it parsed before bailing out (getInputDims() is
  // used by DatasetReader even for an invalid undistorter); its output size is
  // uninitialised in that case, ours is 0.
  for (int i = 0; i < 5; i++) calib_in_[i] = cam.in[i];
  in_w_ = cam.in_w;
  in_h_ = cam.in_h;
  if (pr != kParsed) return;
  out_w_ = cam.out_w;
  out_h_ = cam.out_h;
  valid_ = true;

  pick_output_intrinsics(cam, calib_out_);

  // identity grid pushed through the lens model (:223-232)
  const int n = out_w_ * out_h_;
  remap_x_ = new float[n];
  remap_y_ = new float[n];
  for (int y = 0; y < out_h_; y++)
    for (int x = 0; x < out_w_; x++) {
      remap_x_[x + y * out_w_] = x;
      remap_y_[x + y * out_w_] = y;
    }
  warp_points(calib_in_, in_w_, in_h_, calib_out_, out_w_, out_h_, remap_x_, remap_y_, n);

  // border rules (:235-251): exact hits on the first/last row or column are
  // nudged inside; everything not strictly inside becomes the black sentinel.
  bool black = false;
  for (int i = 0; i < n; i++) {
    float& x = remap_x_[i];
    float& y = remap_y_[i];
    if (x == 0) x = 0.01;
    if (y == 0) y = 0.01;
    if (x == in_w_ - 1) x = in_w_ - 1.01;
    if (y == in_h_ - 1) y = in_h_ - 1.01;
    if (!(x > 0 && y > 0 && x < in_w_ - 1 && y < in_h_ - 1)) {
      black = true;
      x = -1;
      y = -1;
    }
  }
  if (black) std::printf("\n\nFOV Undistorter: Warning! Image has black pixels.\n\n\n");

  set_pinhole(k_rect_, calib_out_, out_w_, out_h_);
  set_pinhole(k_org_, calib_in_, in_w_, in_h_);

  // one-time upload; a missing GPU is reported here and again on every undistort()
  gpu_ = mdc_host::open_device_context("UndistorterFOV");
  if (gpu_ && mdc_set_remap(gpu_, remap_x_, remap_y_, in_w_, in_h_, out_w_, out_h_) != MDC_OK) {
    std::printf("UndistorterFOV: uploading the remap failed: %s\n", mdc_last_error(gpu_));
    mdc_destroy(gpu_);
    gpu_ = 0;
  }
}

UndistorterFOV::~UndistorterFOV() {
  if (gpu_) mdc_destroy(gpu_);
  delete[] remap_x_;
  delete[] remap_y_;
}

void UndistorterFOV::distortCoordinates(float* in_x, float* in_y, int n) {
  if (!valid_) {
    std::printf("ERROR: invalid UndistorterFOV!\n");
    return;
  }
  // Bulk callers (vignetteCalib: 10^6 plane points per image, src/main_vignetteCalib.cpp:284) go to the device: its kernel
  // restates the host libm's atanf and gives the same bits (csrc/fov_point_model.h; pinned on the CPU against this very
  // loop and on the GPU against the host).  Small counts stay here -- a launch and two copies cost more than the loop --,
  // and so does the constructor's table build, which calls warp_points directly.  MDC_DISTORT_GPU_MIN overrides the
  // threshold (0 = never on the device).
  static const long gpu_min = [] {
    const char* e = std::getenv("MDC_DISTORT_GPU_MIN");
    return e ? std::atol(e) : 65536l;
  }();
  // The device's atanf is a restatement of ONE libm's algorithm (glibc's fdlibm atanf).  A process linked against another libm
  // -- a newer glibc whose atanf is correctly rounded, musl, a vendor libm -- would get last-bit differences between small
  // calls (host loop, ::atanf) and large ones (device): checked once against the libm this process really runs, on a spread of
  // arguments through every interval of the algorithm; on any difference bulk calls stay on the host, like the small ones.
  static const bool libm_is_the_restated_one = [] {
    uint32_t bad = 0;
    for (uint32_t u = 0x30000000u; u < 0x4d800000u && !bad; u += 9973u) {  // 2^-31 .. 2^28, ~49,000 arguments, both signs
      const float x = mdc::fov_float(u);
      if (mdc::fov_bits(mdc::atanf_host_libm(x)) != mdc::fov_bits(::atanf(x)) ||
          mdc::fov_bits(mdc::atanf_host_libm(-x)) != mdc::fov_bits(::atanf(-x)))
        bad = u;
    }
    if (bad)
      std::fprintf(stderr, "UndistorterFOV: this process's atanf differs from the algorithm the GPU kernel restates (first at %a): "
                           "distortCoordinates stays on the host for every point count\n", mdc::fov_float(bad));
    return bad == 0;
  }();
  if (gpu_ && gpu_min > 0 && n >= gpu_min && libm_is_the_restated_one) {
    mdc_fov_model m;
    for (int i = 0; i < 5; i++) {
      m.in_calib[i] = calib_in_[i];
      m.out_calib[i] = calib_out_[i];
    }
    m.in_w = in_w_;
    m.in_h = in_h_;
    m.out_w = out_w_;
    m.out_h = out_h_;
    if (mdc_distort_points_host(gpu_, &m, in_x, in_y, n) == MDC_OK) return;
    // the call leaves the coordinates untouched when it fails (they are copied back last): say so and do the work here
    std::fprintf(stderr, "UndistorterFOV::distortCoordinates: the GPU call failed (%s); computing %d points on the host\n", mdc_last_error(gpu_), n);
  }
  warp_points(calib_in_, in_w_, in_h_, calib_out_, out_w_, out_h_, in_x, in_y, n);
}

namespace {
int run_undistort(mdc_ctx* g, const float* in, float* out, int n_in, int n_out) { return mdc_undistort_host_f32(g, in, out, n_in, n_out); }
int run_undistort(mdc_ctx* g, const unsigned char* in, float* out, int n_in, int n_out) { return mdc_undistort_host_u8(g, in, out, n_in, n_out); }
}  // namespace

template <typename T>
void UndistorterFOV::undistort(const T* input, float* output, int nPixIn, int nPixOut) const {
  if (!valid_) return;
  if (nPixIn != in_w_ * in_h_) {
    std::printf("ERROR: undistort called with wrong input image dismesions (expected %d pixel, got %d pixel)\n", in_w_ * in_h_, nPixIn);
    return;
  }
  if (nPixOut != out_w_ * out_h_) {
    std::printf("ERROR: undistort called with wrong output image dismesions (expected %d pixel, got %d pixel)\n", out_w_ * out_h_, nPixOut);
    return;
  }
  if (!gpu_) {
    std::fprintf(stderr, "ERROR: UndistorterFOV::undistort needs a gfx950 GPU (no HIP device context); output not written\n");
    return;
  }
  if (run_undistort(gpu_, input, output, nPixIn, nPixOut) != MDC_OK)
    std::fprintf(stderr, "ERROR: UndistorterFOV::undistort failed on the GPU: %s\n", mdc_last_error(gpu_));
}
template void UndistorterFOV::undistort<float>(const float*, float*, int, int) const;
template void UndistorterFOV::undistort<unsigned char>(const unsigned char*, float*, int, int) const;
