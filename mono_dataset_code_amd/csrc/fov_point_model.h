// The FOV lens model applied to one point, written once for the device kernel (distort_points_kernel, mdc_kernels.hip) and
// for the host compiler (tests/native/distort_points_cpu.cpp): reference src/FOVUndistorter.cpp:303-318.
//
// Bit-exactness with the reference needs the HOST libm's atanf, not the GPU math library's (last-bit differences):
// atanf_host_libm restates the fdlibm single-precision algorithm glibc ships (sysdeps/ieee754/flt-32/s_atanf.c; no FMA
// variant on x86-64) -- argument reduction to one of four intervals, odd/even split of an 11-term polynomial in x^2, hi/lo
// table of atan(0.5), atan(1), atan(1.5), pi/2.  Every operation is an IEEE single-precision add / multiply / divide
// (both compilers run with -ffp-contract=off), sqrtf and '/' are correctly rounded on both sides, so the device's results
// equal the host's bit for bit.  Pinned twice: on the CPU (this header under g++ against the build box's atanf and against
// UndistorterFOV::distortCoordinates' host path, 10^7 arguments) and on the GPU (kernel against the host, tests/test_gpu_parity.py).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MDC_HD __host__ __device__ __forceinline__
#else
#define MDC_HD inline
#endif

namespace mdc {

// FOV lens model in pixel units, derived on the host exactly as src/FOVUndistorter.cpp:289-301 does.
struct DistortModel {
  float fx, fy, cx, cy, omega, d2t;  // input camera; d2t = 2 tan(omega / 2)
  float ofx, ofy, ocx, ocy;          // rectified (output) camera
};

// camera.txt line 1 (relative to the input size) + the normalised output calibration -> pixel units, with the reference's
// own float / double steps: d2t = 2.0f * tan(dist / 2.0f) binds to DOUBLE tan in the reference build (nm: U tan), the
// "- 0.5" of the input centre is a double subtraction narrowed on assignment (:293-301), the output centre's is float (:303).
inline DistortModel make_distort_model(const float in_calib[5], int in_w, int in_h, const float out_calib[5], int out_w, int out_h) {
  DistortModel m;
  const float dist = in_calib[4];
  m.omega = dist;
  m.d2t = 2.0f * ::tan((double)(dist / 2.0f));
  m.fx = in_calib[0] * in_w;
  m.fy = in_calib[1] * in_h;
  m.cx = in_calib[2] * in_w - 0.5;
  m.cy = in_calib[3] * in_h - 0.5;
  m.ofx = out_calib[0] * out_w;
  m.ofy = out_calib[1] * out_h;
  m.ocx = out_calib[2] * out_w - 0.5f;
  m.ocy = out_calib[3] * out_h - 0.5f;
  return m;
}

MDC_HD uint32_t fov_bits(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
}
MDC_HD float fov_float(uint32_t u) {
  float x;
  memcpy(&x, &u, 4);
  return x;
}

MDC_HD float atanf_host_libm(float x) {
  const float hi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float lo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f, a3 = -1.1111110449e-01f,
              a4 = 9.0908870101e-02f, a5 = -7.6918758452e-02f, a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f,
              a8 = 4.9768779427e-02f, a9 = -3.6531571299e-02f, a10 = 1.6285819933e-02f;
  const uint32_t hx = fov_bits(x), ix = hx & 0x7fffffffu;
  if (ix >= 0x4c000000u) {  // |x| >= 2^25, inf, NaN (glibc's float threshold; other fdlibm descendants use 2^26 or 2^34 and
                            // round pi/2 the other way in between -- found by tests/native/distort_points_cpu.cpp)
    if (ix > 0x7f800000u) return x + x;
    return (hx >> 31) ? -hi[3] - lo[3] : hi[3] + lo[3];
  }
  int id;
  if (ix < 0x3ee00000u) {              // |x| < 0.4375
    if (ix < 0x31000000u) return x;    // |x| < 2^-29
    id = -1;
  } else {
    x = fov_float(ix);                 // fabsf
    if (ix < 0x3f980000u) {            // |x| < 1.1875
      if (ix < 0x3f300000u) {          // 7/16 <= |x| < 11/16
        id = 0;
        x = (2.0f * x - 1.0f) / (2.0f + x);
      } else {                         // 11/16 <= |x| < 19/16
        id = 1;
        x = (x - 1.0f) / (x + 1.0f);
      }
    } else if (ix < 0x401c0000u) {     // |x| < 2.4375
      id = 2;
      x = (x - 1.5f) / (1.0f + 1.5f * x);
    } else {                           // 2.4375 <= |x| < 2^25
      id = 3;
      x = -1.0f / x;
    }
  }
  const float z = x * x, w = z * z;
  const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = hi[id] - ((x * (s1 + s2) - lo[id]) - x);
  return (hx >> 31) ? -r : r;
}

// (x, y) rectified pixel -> raw pixel, in place
MDC_HD void fov_distort_point(const DistortModel& m, float& x, float& y) {
  float ix = (x - m.ocx) / m.ofx;  // :306-307
  float iy = (y - m.ocy) / m.ofy;
  const float r = sqrtf(ix * ix + iy * iy);  // correctly rounded (no fast-math)
  const float fac = (r == 0 || m.omega == 0) ? 1 : atanf_host_libm(r * m.d2t) / (m.omega * r);  // :310-311
  x = m.fx * fac * ix + m.cx;  // :313-314
  y = m.fy * fac * iy + m.cy;
}

}  // namespace mdc
