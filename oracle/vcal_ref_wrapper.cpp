// TEST INFRASTRUCTURE.  extern "C" doors onto the reference's own vignetteCalib solver loops: the .inc files are cut
// out of /root/reference/src/main_vignetteCalib.cpp at build time by oracle/vcal_extract.py (into the git-ignored
// oracle/_ref/) and compiled here with the names the surrounding main() gives them (:214-216, :380-398).
// Standard headers as the reference's translation unit sees them (its own :44-47 plus what OpenCV / Eigen pull in),
// so that overload resolution -- notably abs(double) at :424,:481 -- is the reference build's.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#define EIGEN_ALWAYS_INLINE inline
#include "vcal_interp.inc"

static void displayImage(float*, int, int, std::string) {}  // the GUI call inside the plane step (:446)

extern "C" {

// one "optimize planeColor" half-iteration (:400-448): planeColor is read and rewritten, FF/FC are its scratch
void ref_vcal_plane_step(int n, float** p2x, float** p2y, float** imgs, int gw, int gh, int wI, int hI, float* planeColor,
                         float* planeColorFF, float* planeColorFC, float* vignetteFactor, int oth2, double* E_out, double* R_out) {
  std::vector<float*> images(imgs, imgs + n), p2imgX(p2x, p2x + n), p2imgY(p2y, p2y + n);
  double E = 0, R = 0;
  (void)hI;
#include "vcal_body_plane.inc"
  *E_out = E;
  *R_out = R;
}

// one "optimize vignette" half-iteration (:455-527): vignetteFactor is read and rewritten (normalised to max 1)
void ref_vcal_vignette_step(int n, float** p2x, float** p2y, float** imgs, int gw, int gh, int wI, int hI, float* planeColor,
                            float* vignetteFactor, float* vignetteFactorTT, float* vignetteFactorCT, int oth2, double* E_out,
                            double* R_out) {
  std::vector<float*> images(imgs, imgs + n), p2imgX(p2x, p2x + n), p2imgY(p2y, p2y + n);
  double E = 0, R = 0;
#include "vcal_body_vignette.inc"
  *E_out = E;
  *R_out = R;
}

// "dilate & smoothe vignette by 4 pixel for output" (:541-566): TT = the smoothed factors, CT = its scratch copy
void ref_vcal_smooth(int wI, int hI, float* vignetteFactor, float* vignetteFactorTT, float* vignetteFactorCT) {
#include "vcal_body_smooth.inc"
}

// plane points whose rounded image position is not strictly inside [2, w-3] x [2, h-3] get NaN coordinates (:345-357)
void ref_vcal_mask_coords(float* plane2imgX, float* plane2imgY, int gw, int gh, int wI, int hI) {
#include "vcal_body_mask.inc"
}

// pixels of a calibration image that differ from a 5 x 5 neighbour by more than maxAbsGrad -> NaN, both of them, in place
// and in raster order (:293-301; maxAbsGrad is the reference's int, :130)
void ref_vcal_gradient_mask(float* image, int wI, int hI, int maxAbsGrad) {
#include "vcal_body_gradmask.inc"
}

}  // extern "C"
