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
//   vignette step : a scatter-add into the image grid.  The reference's sequential order cannot be kept by
//                   concurrent float atomics, so TT / CT / vignetteFactor agree to ~1e-6 relative, not bitwise
//                   (tests/test_vcal.py: 1e-5);
//   E (printed only, :449,:523): double sums in tree order instead of sequential order; R is an exact count.
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
    if (!isnan(x) && !isnan(colorPlane)) {
      const float y = p2y[(size_t)img * np + pi];
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

// ---------------------------------------------------------------------------------------------------------
// DSO hand-off (NOT in the reference -- DSO's FrameHessian::makeImages, definition in
// DESIGN.md section 5.5): for one pyramid level, per pixel the triple (I, dx, dy) with central differences
// dx = 0.5f*(I[idx+1] - I[idx-1]), dy = 0.5f*(I[idx+w] - I[idx-w]) over the LINEAR index range [w, w*(h-1)) -- rows
// 1 .. h-2, all columns, so the first and last column difference across the row boundary exactly as DSO's loop
// does --, non-finite differences replaced by 0, and absSquaredGrad = dx*dx + dy*dy.  First and last row: 0.
// A wave handles 64 consecutive pixels and writes their 192 floats as three wave-contiguous dword stores (through a
// wave-private LDS transpose) instead of three stores at a 12-byte stride.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gradients_kernel(const float* __restrict__ lvl, float* __restrict__ dI,
                                                        float* __restrict__ abs2, int w, int h, long long nframes) {
  __shared__ float s_t[4][192];
  const long long npx = (long long)w * h;
  const long long base = (long long)blockIdx.x * 256;  // first pixel (over all frames) of this workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long i = base + threadIdx.x;
  float I = 0.f, dx = 0.f, dy = 0.f;
  const bool in = i < npx * nframes;
  if (in) {
    const long long f = i / npx;
    const int idx = (int)(i - f * npx);
    const float* p = lvl + f * npx;
    I = p[idx];
    if (idx >= w && idx < w * (h - 1)) {
      dx = 0.5f * (p[idx + 1] - p[idx - 1]);
      dy = 0.5f * (p[idx + w] - p[idx - w]);
      if (!isfinite(dx)) dx = 0.f;
      if (!isfinite(dy)) dy = 0.f;
    }
    abs2[i] = dx * dx + dy * dy;
  }
  s_t[wave][3 * lane + 0] = I;
  s_t[wave][3 * lane + 1] = dx;
  s_t[wave][3 * lane + 2] = dy;
  // wave-private: no workgroup barrier needed, only the LDS writes of this wave have to land
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  const long long out0 = (base + wave * 64) * 3;
  const long long total = npx * nframes * 3;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const long long o = out0 + k * 64 + lane;
    if (o < total) __builtin_nontemporal_store(s_t[wave][k * 64 + lane], dI + o);
  }
}

inline int blocks(long long n) { return (int)((n + 255) / 256); }

}  // namespace

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

hipError_t launch_gradients(const float* d_level, float* d_dI, float* d_abs2, int w, int h, int64_t nframes, hipStream_t s) {
  const long long n = (long long)w * h * nframes;
  if (n <= 0) return hipSuccess;
  gradients_kernel<<<blocks(n), 256, 0, s>>>(d_level, d_dI, d_abs2, w, h, nframes);
  return hipGetLastError();
}

}  // namespace mdc
