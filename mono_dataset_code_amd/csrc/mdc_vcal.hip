// gfx950 kernels for the vignetteCalib solver's accumulate loops (reference src/main_vignetteCalib.cpp:395-527):
// the alternating least-squares iteration that estimates the calibration plane's colour (one value per plane
// point, gw*gh = 10^6 of them) and the per-pixel vignette factor from n images (up to ~1000) of the plane.
// Per half-iteration the reference makes n*gw*gh bilinear gathers from the image stack and from the current
// factor map on ONE core; here a thread owns a plane point (plane step) or a (plane point, image) pair
// (vignette step).  Same arithmetic family as the hot path: FOV-warped coordinates + bilinear taps, f32, no FMA.
//
// Numerics (this file is built with -ffp-contract=off, division correctly rounded):
//   plane step    : every plane point sums over the images in the reference's order -> FF, FC and the new
//                   planeColor are BIT-IDENTICAL to the reference;
//   vignette step : a scatter-add into the image grid.  As written (vcal_vignette_accumulate_kernel, concurrent float
//                   atomics) the reference's sequential order cannot be kept: TT / CT / vignetteFactor agree to ~1e-6
//                   relative (tests/test_vcal.py: 1e-5).  Inverted into an ordered gather over a prebuilt index
//                   (VcalIndex, vcal_vignette_gather_kernel) it is BIT-IDENTICAL and has no atomics on the data path;
//   E (printed only, :449,:523): double sums in tree order instead of sequential order; R is an exact count.
#include <vector>

#include "mdc_internal.h"

namespace mdc {
namespace {

// getInterpolatedElement, src/main_vignetteCalib.cpp:52-70
__device__ __forceinline__ float interp(const float* __restrict__ mat, float x, float y, int width) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float* bp = mat + ix + iy * width;
  return dxdy * bp[1 + width] + (dy - dxdy) * bp[width] + (dx - dxdy) * bp[1] + (1 - dx - dy + dxdy) * bp[0];
}

__device__ __forceinline__ void block_add_er(double e, double r, double* er) {
  __shared__ double s_e[256], s_r[256];
  s_e[threadIdx.x] = e;
  s_r[threadIdx.x] = r;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      s_e[threadIdx.x] += s_e[threadIdx.x + k];
      s_r[threadIdx.x] += s_r[threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && (s_e[0] != 0 || s_r[0] != 0)) {
    atomicAdd(er, s_e[0]);
    atomicAdd(er + 1, s_r[0]);
  }
}

// the 2x2 footprint of a sample lies inside the image (also false for NaN coordinates)
__device__ __forceinline__ bool vcal_inside(float x, float y, int wI, int hI) {
  return x >= 0.f && y >= 0.f && x < (float)(wI - 1) && y < (float)(hI - 1);
}

// "optimize planeColor" (:400-448): one thread per plane point, images in order
__global__ __launch_bounds__(256) void vcal_plane_kernel(const float* __restrict__ images, const float* __restrict__ p2x,
                                                         const float* __restrict__ p2y, int n, int wI, int hI, int np,
                                                         float* __restrict__ plane_color, const float* __restrict__ vig,
                                                         double oth2, float* __restrict__ FF, float* __restrict__ FC,
                                                         double* __restrict__ er) {
  const int pi = blockIdx.x * 256 + threadIdx.x;
  double E = 0, R = 0;
  if (pi < np) {
    const float pc = plane_color[pi];
    float ff = 0.f, fc = 0.f;
    const size_t img_px = (size_t)wI * hI;
    for (int img = 0; img < n; img++) {
      const float x = p2x[(size_t)img * np + pi];
      if (isnan(x)) continue;
      const float y = p2y[(size_t)img * np + pi];
      // a sample whose 2x2 footprint leaves the image: the reference relies on its caller's mask (:345-357) and would read
      // out of bounds; dropped here, exactly as the contribution index drops it (vcal_sample), so both half-iterations
      // always see the same sample set
      if (!vcal_inside(x, y, wI, hI)) continue;
      const float color = interp(images + img * img_px, x, y, wI);
      const float fac = interp(vig, x, y, wI);
      if (isnan(fac)) continue;
      if (isnan(color)) continue;
      const double residual = (double)((color - pc * fac) * (color - pc * fac));
      if (fabs(residual) > oth2) {
        E += oth2;
        R += 1;
        continue;
      }
      ff += fac * fac;
      fc += color * fac;
      if (isnan(pc)) continue;
      E += residual;
      R += 1;
    }
    FF[pi] = ff;
    FC[pi] = fc;
    plane_color[pi] = ff < 1 ? __builtin_nanf("") : fc / ff;  // :441-447
  }
  block_add_er(E, R, er);
}

// "optimize vignette", accumulation (:461-509): one thread per (plane point, image), bilinear scatter by float atomics
__global__ __launch_bounds__(256) void vcal_vignette_accumulate_kernel(const float* __restrict__ images,
                                                                       const float* __restrict__ p2x,
                                                                       const float* __restrict__ p2y, int wI, int hI, int np,
                                                                       const float* __restrict__ plane_color,
                                                                       const float* __restrict__ vig, double oth2,
                                                                       float* __restrict__ TT, float* __restrict__ CT,
                                                                       double* __restrict__ er) {
  const int pi = blockIdx.x * 256 + threadIdx.x;
  const int img = blockIdx.y;
  double E = 0, R = 0;
  if (pi < np) {
    const float x = p2x[(size_t)img * np + pi];
    const float colorPlane = plane_color[pi];
    const float y = isnan(x) ? 0.f : p2y[(size_t)img * np + pi];
    if (!isnan(x) && !isnan(colorPlane) && vcal_inside(x, y, wI, hI)) {  // (footprint outside the image: dropped, see vcal_plane_kernel)
      const float colorImage = interp(images + (size_t)img * wI * hI, x, y, wI);
      if (!isnan(colorImage)) {
        const float fac = interp(vig, x, y, wI);
        const double residual = (double)((colorImage - colorPlane * fac) * (colorImage - colorPlane * fac));
        if (fabs(residual) > oth2) {
          E = oth2;
          R = 1;
        } else {
          const int ix = (int)x, iy = (int)y;
          const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
          float* tt = TT + ix + iy * wI;
          float* ct = CT + ix + iy * wI;
          unsafeAtomicAdd(tt + 0, (1 - dx - dy + dxdy) * colorPlane * colorPlane);  // :495-498
          unsafeAtomicAdd(tt + 1, (dx - dxdy) * colorPlane * colorPlane);
          unsafeAtomicAdd(tt + wI, (dy - dxdy) * colorPlane * colorPlane);
          unsafeAtomicAdd(tt + 1 + wI, dxdy * colorPlane * colorPlane);
          unsafeAtomicAdd(ct + 0, (1 - dx - dy + dxdy) * colorImage * colorPlane);  // :500-503
          unsafeAtomicAdd(ct + 1, (dx - dxdy) * colorImage * colorPlane);
          unsafeAtomicAdd(ct + wI, (dy - dxdy) * colorImage * colorPlane);
          unsafeAtomicAdd(ct + 1 + wI, dxdy * colorImage * colorPlane);
          if (!isnan(fac)) {
            E = residual;
            R = 1;
          }
        }
      }
    }
  }
  block_add_er(E, R, er);
}

// ---------------------------------------------------------------------------------------------------------
// "optimize vignette" WITHOUT atomics, bit-identical to the reference: the scatter turned into a gather.
// plane2img coordinates and image colours do not change between iterations (:395 ff. only updates planeColor and
// vignetteFactor), so which (image, plane point, corner) lands in which image pixel is fixed.  Built once
// (VcalIndex, below): for every image pixel ("bin") the list of its contributions in the REFERENCE'S ORDER
// (image-major, plane point ascending = the order of the sequential loop :461-509).  Per iteration one lane walks
// one bin's list front to back and adds exactly the terms the reference adds, in the same order, with the same f32
// expressions -> TT, CT and the new vignetteFactor equal the reference's bit for bit, and nothing is atomic.
// Memory layout sized for HBM, not for a cache: 4 entries of 16 bytes per valid (image, point) sample (200 images x
// 10^6 points -> 12.8 GB), ELL-packed per group of 64 consecutive bins (entry k of the 64 bins is one 1-KB line
// group, so a wave's k-th load is one coalesced dwordx4 per lane); a group is as long as its longest bin.
// ---------------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) VcalEntry {
  float x, y;     // plane2img coordinates of the sample
  float color;    // interpolated image colour (:473), constant over the iterations
  unsigned pc;    // plane point | corner << 30 (corner 0..3 = +0, +1, +w, +1+w of :495-503)
};
constexpr unsigned kCornerShift = 30;
constexpr int kVcalSlackRows = 8;  // rows a walking wave may read past a group's last row (= its loads in flight)

// validity that does not depend on the iteration: coordinate present (:468), taps inside the image (the reference
// relies on its caller for that, :283-300 -- a sample whose 2x2 footprint leaves the image is dropped here instead
// of writing out of bounds), colour not NaN (:478)
__device__ __forceinline__ bool vcal_sample(const float* __restrict__ images, const float* __restrict__ p2x,
                                            const float* __restrict__ p2y, int img, int pi, int wI, int hI, int np, float* x,
                                            float* y, float* color) {
  *x = p2x[(size_t)img * np + pi];
  if (isnan(*x)) return false;
  *y = p2y[(size_t)img * np + pi];
  if (!vcal_inside(*x, *y, wI, hI)) return false;
  *color = interp(images + (size_t)img * wI * hI, *x, *y, wI);
  return !isnan(*color);
}

__global__ __launch_bounds__(256) void vcal_index_count_kernel(const float* __restrict__ images, const float* __restrict__ p2x,
                                                               const float* __restrict__ p2y, int wI, int hI, int np,
                                                               unsigned* __restrict__ counts) {
  const int pi = blockIdx.x * 256 + threadIdx.x;
  if (pi >= np) return;
  float x, y, c;
  if (!vcal_sample(images, p2x, p2y, blockIdx.y, pi, wI, hI, np, &x, &y, &c)) return;
  unsigned* b = counts + (int)x + (int)y * wI;
  atomicAdd(b, 1u);  // integer counts: the result does not depend on the order
  atomicAdd(b + 1, 1u);
  atomicAdd(b + wI, 1u);
  atomicAdd(b + 1 + wI, 1u);
}

// group g = bins 64g .. 64g+63: length = longest list; exclusive scan of the lengths by one workgroup
__global__ __launch_bounds__(1024) void vcal_index_scan_kernel(const unsigned* __restrict__ counts, int nbins, int ngroups,
                                                               unsigned* __restrict__ glen, unsigned long long* __restrict__ gbase) {
  __shared__ unsigned long long s_sum[1024];
  __shared__ unsigned long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int g0 = 0; g0 < ngroups; g0 += 1024) {
    const int g = g0 + threadIdx.x;
    unsigned m = 0;
    if (g < ngroups)
      for (int k = 0; k < 64; k++) {
        const int b = g * 64 + k;
        if (b < nbins) m = max(m, counts[b]);
      }
    if (g < ngroups) glen[g] = m;
    s_sum[threadIdx.x] = m;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
      const unsigned long long v = (int)threadIdx.x >= d ? s_sum[threadIdx.x - d] : 0;
      __syncthreads();
      s_sum[threadIdx.x] += v;
      __syncthreads();
    }
    if (g < ngroups) gbase[g] = s_carry + s_sum[threadIdx.x] - m;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry += s_sum[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) gbase[ngroups] = s_carry;  // total number of 64-entry rows
}

__device__ __forceinline__ size_t vcal_slot(const unsigned long long* __restrict__ gbase, int bin, unsigned k) {
  return ((size_t)gbase[bin >> 6] + k) * 64 + (bin & 63);
}

// entries of ONE image appended to the lists (launched image by image, so lists stay image-major); within the
// image's segment the order is whatever the slot atomics gave -- vcal_index_sort_kernel puts it right
__global__ __launch_bounds__(256) void vcal_index_fill_kernel(const float* __restrict__ images, const float* __restrict__ p2x,
                                                              const float* __restrict__ p2y, int img, int wI, int hI, int np,
                                                              const unsigned long long* __restrict__ gbase,
                                                              unsigned* __restrict__ cursor, VcalEntry* __restrict__ entries) {
  const int pi = blockIdx.x * 256 + threadIdx.x;
  if (pi >= np) return;
  VcalEntry e;
  if (!vcal_sample(images, p2x, p2y, img, pi, wI, hI, np, &e.x, &e.y, &e.color)) return;
  const int bin0 = (int)e.x + (int)e.y * wI;
#pragma unroll
  for (unsigned c = 0; c < 4; c++) {
    const int bin = bin0 + (c & 1) + (c >> 1) * wI;
    e.pc = (unsigned)pi | c << kCornerShift;
    entries[vcal_slot(gbase, bin, atomicAdd(cursor + bin, 1u))] = e;
  }
}

// one lane per bin: the segment [seg_begin, cursor) this image appended, ordered by plane point (insertion sort;
// a pixel receives a handful of samples per image)
__global__ __launch_bounds__(256) void vcal_index_sort_kernel(const unsigned long long* __restrict__ gbase,
                                                              const unsigned* __restrict__ seg_begin,
                                                              const unsigned* __restrict__ cursor, int nbins,
                                                              VcalEntry* __restrict__ entries) {
  const int bin = blockIdx.x * 256 + threadIdx.x;
  if (bin >= nbins) return;
  const unsigned a = seg_begin[bin], b = cursor[bin];
  for (unsigned i = a + 1; i < b; i++) {
    const VcalEntry e = entries[vcal_slot(gbase, bin, i)];
    const unsigned key = e.pc & ((1u << kCornerShift) - 1);
    unsigned j = i;
    while (j > a) {
      const VcalEntry p = entries[vcal_slot(gbase, bin, j - 1)];
      if ((p.pc & ((1u << kCornerShift) - 1)) <= key) break;
      entries[vcal_slot(gbase, bin, j)] = p;
      j--;
    }
    if (j != i) entries[vcal_slot(gbase, bin, j)] = e;
  }
}

// the accumulation (:461-509) as a gather: a wave owns a group of 64 bins, lane = bin, lists walked front to back
__global__ __launch_bounds__(256) void vcal_vignette_gather_kernel(const VcalEntry* __restrict__ entries,
                                                                   const unsigned long long* __restrict__ gbase,
                                                                   const unsigned* __restrict__ glen,
                                                                   const unsigned* __restrict__ counts, int nbins, int wI,
                                                                   const float* __restrict__ plane_color,
                                                                   const float* __restrict__ vig, double oth2,
                                                                   float* __restrict__ TT, float* __restrict__ CT,
                                                                   double* __restrict__ er) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int bin = g * 64 + lane;
  double E = 0, R = 0;
  if (g * 64 < nbins) {  // wave-uniform
    const unsigned len = bin < nbins ? counts[bin] : 0;
    const unsigned rows = __builtin_amdgcn_readfirstlane(glen[g]);  // wave-uniform trip count
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u* row = reinterpret_cast<const v4u*>(entries) + (size_t)gbase[g] * 64 + lane;
    float tt = 0.f, ct = 0.f;  // :457-458
    // Every sample in this bin's list has its 2x2 footprint (:52-70) inside the 3x3 pixels around the bin: the
    // current factors of that neighbourhood live in registers, the taps of interp(vignetteFactor, x, y) are picked
    // from them by the entry's corner.  (Neighbours outside the image are never picked: clamped addresses.)
    float V[3][3];
    {
      const int bx = bin % wI, by = bin / wI, hI = nbins / wI;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const int yy = min(max(by + r - 1, 0), hI - 1), xx = min(max(bx + q - 1, 0), wI - 1);
          V[r][q] = vig[bin < nbins ? yy * wI + xx : 0];
        }
    }
    // Branch-free body: an entry that the reference skips adds +0.0f, which leaves a sum that started at +0 unchanged
    // bit for bit (x + 0 == x for every x but -0, and a sum that starts at +0 never becomes -0).  Loads are
    // unconditional too: the lists are allocated with kVcalSlackRows rows of slack, rows beyond a bin's own length hold
    // padding that `live` masks.
    unsigned Rn = 0;
    for (unsigned k0 = 0; k0 < rows; k0 += kVcalSlackRows) {
      v4u raw[kVcalSlackRows];
#pragma unroll
      for (int u = 0; u < kVcalSlackRows; u++) raw[u] = __builtin_nontemporal_load(row + (size_t)(k0 + u) * 64);
#pragma unroll
      for (int u = 0; u < kVcalSlackRows; u++) {
        const bool live = k0 + u < len;
        const float x = __uint_as_float(raw[u].x), y = __uint_as_float(raw[u].y), colorImage = __uint_as_float(raw[u].z);
        const unsigned corner = raw[u].w >> kCornerShift;
        const float colorPlane = plane_color[live ? raw[u].w & ((1u << kCornerShift) - 1) : 0u];
        const int ix = (int)x, iy = (int)y;
        const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
        // the sample's pixel (ix, iy) is the bin minus the corner offset: rows (1 - cy, 2 - cy), columns (1 - cx, 2 - cx)
        const bool cx = corner & 1, cy = corner >> 1;
        const float a0 = cy ? V[0][0] : V[1][0], a1 = cy ? V[0][1] : V[1][1], a2 = cy ? V[0][2] : V[1][2];
        const float b0 = cy ? V[1][0] : V[2][0], b1 = cy ? V[1][1] : V[2][1], b2 = cy ? V[1][2] : V[2][2];
        const float t00 = cx ? a0 : a1, t01 = cx ? a1 : a2, t10 = cx ? b0 : b1, t11 = cx ? b1 : b2;
        const float w0 = 1 - dx - dy + dxdy, w1 = dx - dxdy, w2 = dy - dxdy;
        const float fac = dxdy * t11 + w2 * t10 + w1 * t01 + w0 * t00;  // getInterpolatedElement, :52-70
        const float diff = colorImage - colorPlane * fac;
        const double residual = (double)(diff * diff);  // :480
        const bool seen = live && !isnan(colorPlane);   // :468 and :478 were settled when the index was built, :477 here
        const bool outlier = fabs(residual) > oth2;     // :481-486 (false for a NaN residual, as in the reference)
        const bool adds = seen && !outlier;
        const float w = cy ? (cx ? dxdy : w2) : (cx ? w1 : w0);
        tt += adds ? w * colorPlane * colorPlane : 0.f;  // :495-498
        ct += adds ? w * colorImage * colorPlane : 0.f;  // :500-503
        // E and R once per sample (its corner-0 entry): oth2 for an outlier, else the residual unless fac is NaN (:505-507)
        const bool counted = seen && corner == 0 && (outlier || !isnan(fac));
        E += counted ? (outlier ? oth2 : residual) : 0.0;
        Rn += counted;
      }
    }
    R = (double)Rn;
    if (bin < nbins) {
      TT[bin] = tt;
      CT[bin] = ct;
    }
  }
  block_add_er(E, R, er);
}

// :511-521: the new factor and its maximum (NaN where fewer than 1 unit of weight arrived)
__global__ __launch_bounds__(256) void vcal_vignette_update_kernel(const float* __restrict__ TT, const float* __restrict__ CT,
                                                                   float* __restrict__ vig, int npix, unsigned* max_bits) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (i < npix) {
    if (TT[i] < 1) vig[i] = __builtin_nanf("");
    else {
      v = CT[i] / TT[i];
      vig[i] = v;
    }
  }
  if (!(v > 0.f)) v = 0.f;  // maxFac starts at 0 and only larger values replace it; NaN never does
  // positive floats order like their bit patterns
  __shared__ unsigned s_m[256];
  s_m[threadIdx.x] = __float_as_uint(v);
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) s_m[threadIdx.x] = max(s_m[threadIdx.x], s_m[threadIdx.x + k]);
    __syncthreads();
  }
  if (threadIdx.x == 0 && s_m[0]) atomicMax(max_bits, s_m[0]);
}
// :526-527
__global__ __launch_bounds__(256) void vcal_vignette_normalise_kernel(float* __restrict__ vig, int npix, const unsigned* max_bits) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < npix) vig[i] = vig[i] / __uint_as_float(*max_bits);
}

// :286-291: image = meanExposure * getImage(i, false, true, false, false) / exposure_time (0 counts as 1), float
__global__ __launch_bounds__(256) void vcal_scale_images_kernel(float* __restrict__ images, long long npix, float mean_exposure,
                                                                const float* __restrict__ exposure) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  float e = exposure[blockIdx.y];
  if (e == 0) e = 1;
  float* p = images + (size_t)blockIdx.y * npix + i;
  *p = mean_exposure * *p / e;
}

// :293-301, the gradient mask of a calibration image: a pixel and a 5 x 5 neighbour that differ by more than maxAbsGrad
// both become NaN -- IN PLACE and in raster order, so a pixel an earlier one masked no longer takes part.  The result
// depends on that order, but only between pixels whose 5 x 5 neighbourhoods intersect (|dx| <= 4, |dy| <= 4), and for
// every such pair the later one in raster order also has the larger t = x + 5*y.  Pixels of equal t are 5 columns apart
// per row -- their neighbourhoods are disjoint -- so sweeping t upwards with all pixels of one t in parallel replays the
// sequential loop exactly.  One workgroup per image (the images of a stack run side by side), a barrier per step.
__global__ __launch_bounds__(256) void vcal_gradient_mask_kernel(float* __restrict__ images, int wI, int hI, float max_abs_grad) {
  volatile float* img = images + (size_t)blockIdx.x * wI * hI;  // volatile: every step sees the NaNs of the steps before
  const int t_last = (wI - 3) + 5 * (hI - 3);
  for (int t = 2 + 5 * 2; t <= t_last; t++) {
    // rows with a pixel on this front: 2 <= x = t - 5y <= wI-3
    const int y_lo = max(2, (t - (wI - 3) + 4) / 5), y_hi = min(hI - 3, (t - 2) / 5);
    for (int y = y_lo + (int)threadIdx.x; y <= y_hi; y += 256) {
      const int x = t - 5 * y;
      const int p = x + y * wI;
      float vp = img[p];
      if (!isnan(vp)) {
        for (int deltax = -2; deltax < 3 && !isnan(vp); deltax++)
          for (int deltay = -2; deltay < 3; deltay++) {
            const int q = p + deltax + deltay * wI;
            if (fabsf(vp - img[q]) > max_abs_grad) {  // false for a NaN neighbour
              vp = __builtin_nanf("");
              img[p] = vp;
              img[q] = vp;
              break;  // image[x+y*wI] is NaN from here on: no later comparison of this pixel can be true
            }
          }
      }
    }
    __syncthreads();
  }
}

// :345-357: plane points whose image position, rounded as (int)(v + 0.5) (float + double), is not strictly inside
// (1, w-2) x (1, h-2) lose both coordinates.  (NaN / out-of-range conversions saturate here and give INT_MIN on the
// reference's x86 -- either way the test fails and the point is masked.)
__global__ __launch_bounds__(256) void vcal_mask_coords_kernel(float* __restrict__ x, float* __restrict__ y, long long n, int wI, int hI) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int u_d = (int)(x[i] + 0.5);
  const int v_d = (int)(y[i] + 0.5);
  if (!(u_d > 1 && v_d > 1 && u_d < wI - 2 && v_d < hI - 2)) {
    x[i] = __builtin_nanf("");
    y[i] = __builtin_nanf("");
  }
}

// "dilate & smoothe vignette by 4 pixel for output" (:541-566): one pass of the NaN-aware 3 x 3 mean, src -> dst; the nine
// conditional adds in the reference's order, float sum / float count; a pixel without a finite neighbour keeps its value
__global__ __launch_bounds__(256) void vcal_smooth_pass_kernel(const float* __restrict__ src, float* __restrict__ dst, int wI, int hI) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= wI * hI) return;
  const int y = idx / wI, x = idx - y * wI;
  float sum = 0, num = 0;
  const bool xr = x < wI - 1, xl = x > 0, yd = y < hI - 1, yu = y > 0;
  float v;
  if (xr && yd && !isnan(v = src[idx + 1 + wI])) { sum += v; num++; }  // :551-562
  if (xr && !isnan(v = src[idx + 1])) { sum += v; num++; }
  if (xr && yu && !isnan(v = src[idx + 1 - wI])) { sum += v; num++; }
  if (yd && !isnan(v = src[idx + wI])) { sum += v; num++; }
  const float own = src[idx];
  if (!isnan(own)) { sum += own; num++; }
  if (yu && !isnan(v = src[idx - wI])) { sum += v; num++; }
  if (yd && xl && !isnan(v = src[idx - 1 + wI])) { sum += v; num++; }
  if (xl && !isnan(v = src[idx - 1])) { sum += v; num++; }
  if (yu && xl && !isnan(v = src[idx - 1 - wI])) { sum += v; num++; }
  dst[idx] = num > 0 ? sum / num : own;  // :563
}

// ---------------------------------------------------------------------------------------------------------
// DSO hand-off (NOT in the reference -- DSO's FrameHessian::makeImages, definition in
// DESIGN.md section 5.5): for one pyramid level, per pixel the triple (I, dx, dy) with central differences
// dx = 0.5f*(I[idx+1] - I[idx-1]), dy = 0.5f*(I[idx+w] - I[idx-w]) over the LINEAR index range [w, w*(h-1)) -- rows
// 1 .. h-2, all columns, so the first and last column difference across the row boundary exactly as DSO's loop
// does --, non-finite differences replaced by 0, and absSquaredGrad = dx*dx + dy*dy.  First and last row: 0.
// A wave handles 64 consecutive columns of kGradRows rows: all of its loads (rows y0-1 .. y0+R of the column, the left and
// right neighbours of the R rows) are issued before the first result is needed -- the first version (one pixel per thread, one
// 256-pixel workgroup per 4 KB of output) ran at the latency of its five loads, 3.5 TB/s of output; per row the wave's 192
// floats (I, dx, dy interleaved) leave as three wave-contiguous dword stores through a wave-private LDS transpose instead
// of three stores at a 12-byte stride.  Up to four pyramid levels of a chunk of frames in ONE launch
// (mdc_process_pyramid_gradients_batch_device): a workgroup finds its level from the first-block table.
// ---------------------------------------------------------------------------------------------------------
constexpr int kGradRows = MDC_EXP_GRAD_ROWS, kGradThreads = 128;
struct GradLevels {
  const float* src[4];
  float* dI[4];
  float* abs2[4];
  int w[4], h[4];
  unsigned bx[4], bands[4];  // workgroups per row band (128 columns each), row bands (kGradRows rows each)
  unsigned first_block[5];   // blocks [first_block[l], first_block[l+1]) belong to level l
  int n;
};
__global__ __launch_bounds__(kGradThreads) void gradients_levels_kernel(GradLevels g) {
  constexpr int R = kGradRows;
  __shared__ float s_t[kGradThreads / 64][2][192];
  int l = 0;
#pragma unroll
  for (int k = 1; k < 4; k++)
    if (k < g.n && blockIdx.x >= g.first_block[k]) l = k;
  const int w = g.w[l], h = g.h[l];
  const unsigned b = blockIdx.x - g.first_block[l], per_frame = g.bx[l] * g.bands[l];
  const unsigned f = b / per_frame, rb = b - f * per_frame, band = rb / g.bx[l], bx = rb - band * g.bx[l];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = (int)bx * kGradThreads + wave * 64;  // first column of this wave
  const int nvalid = min(64, w - x0);                  // wave-uniform
  if (nvalid <= 0) return;
  const int x = min(x0 + lane, w - 1);  // lanes past the row end repeat its last column (loads stay in bounds; nothing stored)
  const int y0 = (int)band * R;
  const int npx = w * h;
  const float* p = g.src[l] + (long long)f * npx;
  float c[R + 2], lf[R], rt[R];
#ifdef MDC_EXP_GRAD_FAKE_READ  // diagnosis (wrong results): every read of the levels comes from the frame's first 16 KB -- what do these reads cost?
#define MDC_GRAD_IDX(i) ((i) & 4095)
#else
#define MDC_GRAD_IDX(i) (i)
#endif
#pragma unroll
  for (int r = -1; r <= R; r++) c[r + 1] = p[MDC_GRAD_IDX(min(max(y0 + r, 0), h - 1) * w + x)];
  // left / right neighbours (linear index +-1): the neighbouring lanes' centre values; only the wave's first lane and its last valid
  // one load theirs (one instruction with two active lanes per row instead of two unaligned row loads)
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int idx = min(y0 + r, h - 1) * w + x;
    float le = 0.f, re = 0.f;
    if (lane == 0) le = p[MDC_GRAD_IDX(max(idx - 1, 0))];
    if (lane == nvalid - 1) re = p[MDC_GRAD_IDX(min(idx + 1, npx - 1))];
    const float up = __shfl_up(c[r + 1], 1), dn = __shfl_down(c[r + 1], 1);
    lf[r] = lane == 0 ? le : up;
    rt[r] = lane == nvalid - 1 ? re : dn;
  }
#undef MDC_GRAD_IDX
  float* dI = g.dI[l] + ((long long)f * npx + (long long)y0 * w + x0) * 3;
  float* abs2 = g.abs2[l] + (long long)f * npx + (long long)y0 * w + x0;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int y = y0 + r;
    if (y >= h) break;  // wave-uniform
    float dx = 0.f, dy = 0.f;
    if (y >= 1 && y <= h - 2) {  // the linear index range [w, w*(h-1))
      dx = 0.5f * (rt[r] - lf[r]);
      dy = 0.5f * (c[r + 2] - c[r]);
      if (!isfinite(dx)) dx = 0.f;
      if (!isfinite(dy)) dy = 0.f;
    }
    if (lane < nvalid) __builtin_nontemporal_store(dx * dx + dy * dy, abs2 + (long long)r * w + lane);
    float* t = s_t[wave][r & 1];
    t[3 * lane + 0] = c[r + 1];
    t[3 * lane + 1] = dx;
    t[3 * lane + 2] = dy;
    __builtin_amdgcn_wave_barrier();  // wave-private transpose: a wave's LDS operations execute in order
#pragma unroll
    for (int k = 0; k < 3; k++)
      if (k * 64 + lane < 3 * nvalid) __builtin_nontemporal_store(t[k * 64 + lane], dI + (long long)r * w * 3 + k * 64 + lane);
    __builtin_amdgcn_wave_barrier();
  }
}

inline int blocks(long long n) { return (int)((n + 255) / 256); }

}  // namespace

hipError_t launch_gradients_levels(int n_levels, const float* const* d_src, float* const* d_dI, float* const* d_abs2, const int* w,
                                   const int* h, int64_t nframes, hipStream_t s) {
  if (n_levels <= 0 || nframes <= 0) return hipSuccess;
  if (n_levels > 4) return hipErrorInvalidValue;
  GradLevels g;
  uint64_t nb = 0;
  for (int l = 0; l < 4; l++) {
    const bool on = l < n_levels && w[l] > 0 && h[l] > 0;
    g.src[l] = on ? d_src[l] : nullptr;
    g.dI[l] = on ? d_dI[l] : nullptr;
    g.abs2[l] = on ? d_abs2[l] : nullptr;
    g.w[l] = on ? w[l] : 1;
    g.h[l] = on ? h[l] : 1;
    g.bx[l] = (unsigned)((g.w[l] + kGradThreads - 1) / kGradThreads);
    g.bands[l] = (unsigned)((g.h[l] + kGradRows - 1) / kGradRows);
    g.first_block[l] = (unsigned)nb;
    if (on) nb += (uint64_t)g.bx[l] * g.bands[l] * (uint64_t)nframes;
    if ((int64_t)g.w[l] * g.h[l] >= (1ll << 30)) return hipErrorInvalidValue;  // 32-bit pixel indices inside a frame
  }
  if (nb >= (1ull << 31)) return hipErrorInvalidValue;  // callers split such batches (mdc_capi.hip)
  g.first_block[4] = (unsigned)nb;
  g.n = n_levels;
  if (nb == 0) return hipSuccess;
  gradients_levels_kernel<<<(unsigned)nb, kGradThreads, 0, s>>>(g);
  return hipGetLastError();
}

hipError_t launch_vcal_plane_step(const float* d_images, const float* d_p2x, const float* d_p2y, int n, int wI, int hI, int np,
                                  float* d_plane_color, const float* d_vig, int oth2, float* d_ff, float* d_fc, double* d_er,
                                  hipStream_t s) {
  hipError_t e = hipMemsetAsync(d_er, 0, 2 * sizeof(double), s);
  if (e != hipSuccess) return e;
  if (np <= 0) return hipSuccess;
  vcal_plane_kernel<<<blocks(np), 256, 0, s>>>(d_images, d_p2x, d_p2y, n, wI, hI, np, d_plane_color, d_vig, (double)oth2, d_ff, d_fc,
                                               d_er);
  return hipGetLastError();
}

hipError_t launch_vcal_vignette_step(const float* d_images, const float* d_p2x, const float* d_p2y, int n, int wI, int hI, int np,
                                     const float* d_plane_color, float* d_vig, int oth2, float* d_tt, float* d_ct, double* d_er,
                                     unsigned* d_max_bits, hipStream_t s) {
  const size_t img_bytes = (size_t)wI * hI * sizeof(float);
  hipError_t e;
  if ((e = hipMemsetAsync(d_er, 0, 2 * sizeof(double), s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(d_tt, 0, img_bytes, s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(d_ct, 0, img_bytes, s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(d_max_bits, 0, sizeof(unsigned), s)) != hipSuccess) return e;
  if (np > 0 && n > 0)
    vcal_vignette_accumulate_kernel<<<dim3(blocks(np), n), 256, 0, s>>>(d_images, d_p2x, d_p2y, wI, hI, np, d_plane_color, d_vig,
                                                                        (double)oth2, d_tt, d_ct, d_er);
  vcal_vignette_update_kernel<<<blocks((long long)wI * hI), 256, 0, s>>>(d_tt, d_ct, d_vig, wI * hI, d_max_bits);
  vcal_vignette_normalise_kernel<<<blocks((long long)wI * hI), 256, 0, s>>>(d_vig, wI * hI, d_max_bits);
  return hipGetLastError();
}

struct VcalIndex {
  int n = 0, wI = 0, hI = 0, np = 0, nbins = 0, ngroups = 0;
  unsigned* d_counts = nullptr;           // entries per bin
  unsigned* d_glen = nullptr;             // rows (of 64 entries) per group of 64 bins
  unsigned long long* d_gbase = nullptr;  // first row of each group; [ngroups] = total rows
  VcalEntry* d_entries = nullptr;
  unsigned long long rows = 0, samples4 = 0;
};

void vcal_index_free(VcalIndex* ix) {
  if (!ix) return;
  (void)hipFree(ix->d_counts);
  (void)hipFree(ix->d_glen);
  (void)hipFree(ix->d_gbase);
  (void)hipFree(ix->d_entries);
  delete ix;
}

long long vcal_index_bytes(const VcalIndex* ix) { return ix ? (long long)((ix->rows + kVcalSlackRows) * 64 * sizeof(VcalEntry)) : 0; }
long long vcal_index_entries(const VcalIndex* ix) { return ix ? (long long)ix->samples4 : 0; }

// Synchronises the stream twice (list sizes come back to the host before the entries can be allocated): set-up work,
// once per calibration run.
hipError_t vcal_index_build(const float* d_images, const float* d_p2x, const float* d_p2y, int n, int wI, int hI, int np,
                            hipStream_t s, VcalIndex** out) {
  *out = nullptr;
  VcalIndex* ix = new VcalIndex;
  ix->n = n;
  ix->wI = wI;
  ix->hI = hI;
  ix->np = np;
  ix->nbins = wI * hI;
  ix->ngroups = (ix->nbins + 63) / 64;
  unsigned *d_cursor = nullptr, *d_seg = nullptr;
  const size_t bin_bytes = (size_t)ix->nbins * sizeof(unsigned);
  hipError_t e;
  auto bail = [&](hipError_t err) {
    (void)hipFree(d_cursor);
    (void)hipFree(d_seg);
    vcal_index_free(ix);
    return err;
  };
  if ((e = hipMalloc(&ix->d_counts, bin_bytes)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ix->d_glen, (size_t)ix->ngroups * sizeof(unsigned))) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ix->d_gbase, ((size_t)ix->ngroups + 1) * sizeof(unsigned long long))) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&d_cursor, bin_bytes)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&d_seg, bin_bytes)) != hipSuccess) return bail(e);
  if ((e = hipMemsetAsync(ix->d_counts, 0, bin_bytes, s)) != hipSuccess) return bail(e);
  if ((e = hipMemsetAsync(d_cursor, 0, bin_bytes, s)) != hipSuccess) return bail(e);
  if (np > 0 && n > 0)  // an empty problem has empty lists: every bin keeps the reference's memset value
    vcal_index_count_kernel<<<dim3(blocks(np), n), 256, 0, s>>>(d_images, d_p2x, d_p2y, wI, hI, np, ix->d_counts);
  vcal_index_scan_kernel<<<1, 1024, 0, s>>>(ix->d_counts, ix->nbins, ix->ngroups, ix->d_glen, ix->d_gbase);
  if ((e = hipGetLastError()) != hipSuccess) return bail(e);
  if ((e = hipMemcpyAsync(&ix->rows, ix->d_gbase + ix->ngroups, sizeof(unsigned long long), hipMemcpyDeviceToHost, s)) != hipSuccess)
    return bail(e);
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return bail(e);
  if (ix->rows) {
    if ((e = hipMalloc(&ix->d_entries, ((size_t)ix->rows + kVcalSlackRows) * 64 * sizeof(VcalEntry))) != hipSuccess) return bail(e);
    for (int img = 0; img < n; img++) {
      if ((e = hipMemcpyAsync(d_seg, d_cursor, bin_bytes, hipMemcpyDeviceToDevice, s)) != hipSuccess) return bail(e);
      vcal_index_fill_kernel<<<blocks(np), 256, 0, s>>>(d_images, d_p2x, d_p2y, img, wI, hI, np, ix->d_gbase, d_cursor, ix->d_entries);
      vcal_index_sort_kernel<<<blocks(ix->nbins), 256, 0, s>>>(ix->d_gbase, d_seg, d_cursor, ix->nbins, ix->d_entries);
    }
    if ((e = hipGetLastError()) != hipSuccess) return bail(e);
  }
  // number of list entries (4 per valid sample): sum of the counts, on the host (set-up)
  {
    std::vector<unsigned> h((size_t)ix->nbins);
    if ((e = hipMemcpyAsync(h.data(), ix->d_counts, bin_bytes, hipMemcpyDeviceToHost, s)) != hipSuccess) return bail(e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return bail(e);
    for (unsigned v : h) ix->samples4 += v;
  }
  (void)hipFree(d_cursor);
  (void)hipFree(d_seg);
  *out = ix;
  return hipSuccess;
}

hipError_t launch_vcal_vignette_step_indexed(const VcalIndex* ix, const float* d_plane_color, float* d_vig, int oth2, float* d_tt,
                                             float* d_ct, double* d_er, unsigned* d_max_bits, hipStream_t s) {
  hipError_t e;
  if ((e = hipMemsetAsync(d_er, 0, 2 * sizeof(double), s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(d_max_bits, 0, sizeof(unsigned), s)) != hipSuccess) return e;
  // every bin is written by its lane (empty lists write the reference's memset value 0)
  vcal_vignette_gather_kernel<<<(ix->ngroups + 3) / 4, 256, 0, s>>>(ix->d_entries, ix->d_gbase, ix->d_glen, ix->d_counts, ix->nbins,
                                                                    ix->wI, d_plane_color, d_vig, (double)oth2, d_tt, d_ct, d_er);
  vcal_vignette_update_kernel<<<blocks(ix->nbins), 256, 0, s>>>(d_tt, d_ct, d_vig, ix->nbins, d_max_bits);
  vcal_vignette_normalise_kernel<<<blocks(ix->nbins), 256, 0, s>>>(d_vig, ix->nbins, d_max_bits);
  return hipGetLastError();
}

hipError_t launch_vcal_scale_images(float* d_images, int n, int64_t npix, float mean_exposure, const float* d_exposure,
                                    hipStream_t s) {
  if (n <= 0 || npix <= 0) return hipSuccess;
  vcal_scale_images_kernel<<<dim3(blocks(npix), n), 256, 0, s>>>(d_images, npix, mean_exposure, d_exposure);
  return hipGetLastError();
}

hipError_t launch_vcal_gradient_mask(float* d_images, int n, int wI, int hI, int max_abs_grad, hipStream_t s) {
  if (n <= 0 || wI < 5 || hI < 5) return hipSuccess;  // the loops :293-294 are empty
  vcal_gradient_mask_kernel<<<n, 256, 0, s>>>(d_images, wI, hI, (float)max_abs_grad);
  return hipGetLastError();
}

hipError_t launch_vcal_mask_coords(float* d_x, float* d_y, int64_t n, int wI, int hI, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  vcal_mask_coords_kernel<<<blocks(n), 256, 0, s>>>(d_x, d_y, n, wI, hI);
  return hipGetLastError();
}

// four passes, d_vig -> d_ct -> d_tt -> d_ct -> d_tt: d_tt ends as the smoothed map, d_ct as the input of the last pass
// (the reference's TT and CT after :541-566)
hipError_t launch_vcal_smooth(const float* d_vig, int wI, int hI, float* d_tt, float* d_ct, hipStream_t s) {
  const int n = wI * hI;
  vcal_smooth_pass_kernel<<<blocks(n), 256, 0, s>>>(d_vig, d_ct, wI, hI);
  vcal_smooth_pass_kernel<<<blocks(n), 256, 0, s>>>(d_ct, d_tt, wI, hI);
  vcal_smooth_pass_kernel<<<blocks(n), 256, 0, s>>>(d_tt, d_ct, wI, hI);
  vcal_smooth_pass_kernel<<<blocks(n), 256, 0, s>>>(d_ct, d_tt, wI, hI);
  return hipGetLastError();
}

hipError_t launch_gradients(const float* d_level, float* d_dI, float* d_abs2, int w, int h, int64_t nframes, hipStream_t s) {
  if (w <= 0 || h <= 0) return hipSuccess;
  // launches of at most 2^30 workgroups
  const int64_t per_frame = (int64_t)((w + kGradThreads - 1) / kGradThreads) * ((h + kGradRows - 1) / kGradRows);
  const int64_t step = std::max<int64_t>(1, (1ll << 30) / per_frame);
  const int64_t npx = (int64_t)w * h;
  for (int64_t f0 = 0; f0 < nframes; f0 += step) {
    const float* src = d_level + f0 * npx;
    float* dI = d_dI + f0 * npx * 3;
    float* a2 = d_abs2 + f0 * npx;
    hipError_t e = launch_gradients_levels(1, &src, &dI, &a2, &w, &h, std::min(step, nframes - f0), s);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace mdc
