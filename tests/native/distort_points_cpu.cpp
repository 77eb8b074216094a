// CPU pin of the device's distortCoordinates arithmetic (csrc/fov_point_model.h, the header distort_points_kernel is made
// of), compiled by the HOST compiler with the product's flags (-ffp-contract=off):
//   1. atanf_host_libm(x) == this box's libm atanf(x), bit for bit, over a dense sweep of the float line + random bits;
//   2. fov_distort_point == UndistorterFOV::distortCoordinates' host loop (libmdc_host.so, no GPU: the class computes on
//      the host here) on a grid + random points of every camera given on the command line.
// usage: distort_points_cpu camera.txt [camera.txt ...]     prints "ok <checked>" or the first mismatches, exit 1
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "FOVUndistorter.h"
#include "MdcBind.h"
#include "fov_point_model.h"

static uint32_t bits(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
}
static bool same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }

int main(int argc, char** argv) {
  long checked = 0, bad = 0;
  // 1. atanf: every 97th float of the whole line (44 M values: all exponents, both signs, NaN / inf included)
  for (uint64_t u = 0; u <= 0xffffffffull; u += 97) {
    float x;
    const uint32_t v = (uint32_t)u;
    memcpy(&x, &v, 4);
    const float a = mdc::atanf_host_libm(x), b = atanf(x);
    checked++;
    if (!same(a, b) && bad++ < 10) std::printf("atanf(%a): restatement %a, libm %a\n", x, a, b);
  }
  // ... and densely where the lens model lives: r * d2t in (0, 8)
  for (uint32_t v = bits(1e-6f); v < bits(8.0f); v += 13) {
    float x;
    memcpy(&x, &v, 4);
    checked++;
    if (!same(mdc::atanf_host_libm(x), atanf(x)) && bad++ < 10) std::printf("atanf(%a) differs\n", x);
  }
  // 2. the whole point model against the class's host loop
  uint64_t rng = 0x9e3779b97f4a7c15ull;
  for (int c = 1; c < argc; c++) {
    UndistorterFOV u(argv[c]);
    if (!u.isValid()) continue;
    mdc_fov_model fm;
    mdc_fov_model_of(u, &fm);
    const mdc::DistortModel m = mdc::make_distort_model(fm.in_calib, fm.in_w, fm.in_h, fm.out_calib, fm.out_w, fm.out_h);
    const int n = 1 << 16;  // below the class's GPU threshold as well (there is no GPU here anyway)
    std::vector<float> x(n), y(n), hx(n), hy(n);
    for (int round = 0; round < 16; round++) {
      for (int i = 0; i < n; i++) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        const float fx = (float)((rng >> 20) & 0xfffff) / 1048576.f, fy = (float)((rng >> 40) & 0xfffff) / 1048576.f;
        // inside, around and far outside the rectified image; exact centre and tiny radii now and then
        const float span = round < 8 ? 1.2f : (round < 12 ? 40.f : 1e-3f);
        x[i] = m.ocx + (fx - 0.5f) * span * fm.out_w;
        y[i] = m.ocy + (fy - 0.5f) * span * fm.out_h;
        if (i % 1000 == 0) x[i] = m.ocx, y[i] = m.ocy;
      }
      hx = x;
      hy = y;
      u.distortCoordinates(hx.data(), hy.data(), n);
      for (int i = 0; i < n; i++) {
        float px = x[i], py = y[i];
        mdc::fov_distort_point(m, px, py);
        checked++;
        if ((!same(px, hx[i]) || !same(py, hy[i])) && bad++ < 10)
          std::printf("%s point (%a, %a): header (%a, %a), class (%a, %a)\n", argv[c], x[i], y[i], px, py, hx[i], hy[i]);
      }
    }
  }
  if (bad) {
    std::printf("MISMATCHES %ld of %ld\n", bad, checked);
    return 1;
  }
  std::printf("ok %ld\n", checked);
  return 0;
}
