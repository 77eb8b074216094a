// gfx950 (MI355X, CDNA4) kernels for the photometric + FOV undistortion hot path.
//
// What they compute (reference = tum-vision/mono_dataset_code, paths relative to it):
//   unmap   : PhotometricUndistorter::unMapImage      src/PhotometricUndistorter.cpp:193-211
//   remap_* : UndistorterFOV::undistort<T>            src/FOVUndistorter.cpp:341-367
//   fused   : the two composed as DatasetReader::getImage does
//             (src/BenchmarkDatasetReader.h:222-223) without the W*H float intermediate.
//
// Numerics.  The reference is built without FMA (CMakeLists.txt:16-18), so this
// file is compiled with -ffp-contract=off and every expression keeps the
// reference's evaluation order; results are bit-identical, not merely within
// the 1e-4 gate.  The three photometric modes and the overexposure kill are
// folded into ONE 256-entry table chosen by the host:
//   identity : lut[b] = (float)b        gamma : lut[b] = GInv[b]
//   kill     : lut[255] = NaN  (src/PhotometricUndistorter.cpp:208-211 tests the RAW byte)
// and an optional per-pixel factor vinv[i] (gamma+vignette mode, :205).  NaN taps
// propagate through the bilinear sum even under a zero weight, as in the reference.
//
// All kernels are HBM-bound byte/float streaming; there is no contraction, so
// no MFMA.  Frames are batched: a workgroup owns a fixed set of pixels (or one
// output tile) and loops over `fpb` frames, so calibration tables are read once
// per workgroup, not once per frame.
#include "mdc_internal.h"

namespace mdc {

namespace {

constexpr int kLutBytes = 256 * kLutRep * 4;
// Build-time experiment switches (tools/sweep.py --lib ...; defaults are the shipped configuration).
#ifndef MDC_EXP_LOAD_NT
#define MDC_EXP_LOAD_NT 0   // staging loads: plain (L2-allocating) -- neighbouring tiles re-use halo lines; nt measured slower
#endif
#ifndef MDC_EXP_STORE_NT
#define MDC_EXP_STORE_NT 1  // output stores carry the nontemporal hint
#endif
#ifndef MDC_EXP_BATCHED
#define MDC_EXP_BATCHED 0   // issue all 16 tap reads, then all 16 LUT reads, then the arithmetic
#endif
#ifndef MDC_EXP_SKIP_STORE
#define MDC_EXP_SKIP_STORE 0  // diagnosis: outputs are computed but (practically) never stored -> read side alone
#endif
#ifndef MDC_EXP_SKIP_LOAD
#define MDC_EXP_SKIP_LOAD 0   // diagnosis: every frame re-stages frame 0 (L2 hits) -> write side alone
#endif
#ifndef MDC_EXP_PF2
#define MDC_EXP_PF2 0         // staging loads run TWO frames ahead (second stage in registers)
#endif
#ifndef MDC_EXP_WAVES
#define MDC_EXP_WAVES 5     // __launch_bounds__ min waves per SIMD of the 256-thread tiled kernel
#endif
#ifndef MDC_EXP_WAVES_512
#define MDC_EXP_WAVES_512 6 // same for the 512-thread (64x32 tile) kernel: 3 workgroups per CU
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Replicate the 256-entry LUT kLutRep times so that lane l reads replica l%32:
// word address b*32 + (l&31) lies in bank l&31 for every byte value b, i.e. a
// data-dependent table lookup with zero LDS bank conflicts.
template <int NT>
__device__ __forceinline__ void fill_lut(float* s_lut, const float* __restrict__ lut, int tid) {
#pragma unroll 4
  for (int i = tid; i < 256 * kLutRep; i += NT) s_lut[i] = lut[i / kLutRep];
}

// ----------------------------------------------------------------------------
// unMapImage, vector path: npix % 4 == 0, 16-byte aligned bases.
// A workgroup owns 4096 consecutive pixels and loops over its frames.  Access
// k of thread t touches pixels 4*(k*256+t) .. +3: each wave-instruction loads
// 256 contiguous bytes of the raw frame and stores 1 KiB contiguous floats.
// ----------------------------------------------------------------------------
template <bool VIG>
__global__ __launch_bounds__(256) void unmap_vec_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ lut,
                                                        const float* __restrict__ vinv, long long npix, int nframes,
                                                        int fpb) {
  __shared__ float s_lut[256 * kLutRep];
  const int tid = threadIdx.x;
  fill_lut<256>(s_lut, lut, tid);
  __syncthreads();
  const float* my_lut = s_lut + (tid & (kLutRep - 1));

  const long long base = (long long)blockIdx.x * 4096;
  long long p[4];
  bool ok[4];
  f32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    p[k] = base + (long long)(k * 256 + tid) * 4;
    ok[k] = p[k] < npix;
    v[k] = (f32x4)(1.f);
    if (VIG && ok[k]) v[k] = *reinterpret_cast<const f32x4*>(vinv + p[k]);
  }
  const int f0 = blockIdx.y * fpb;
  const int f1 = min(nframes, f0 + fpb);
  const uint8_t* src = in + (long long)f0 * npix;
  float* dst = out + (long long)f0 * npix;
  for (int f = f0; f < f1; f++, src += npix, dst += npix) {
    uint32_t raw[4];
#pragma unroll
    for (int k = 0; k < 4; k++) raw[k] = ok[k] ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(src + p[k])) : 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      f32x4 r;
      r.x = my_lut[((raw[k]) & 255u) * kLutRep];
      r.y = my_lut[((raw[k] >> 8) & 255u) * kLutRep];
      r.z = my_lut[((raw[k] >> 16) & 255u) * kLutRep];
      r.w = my_lut[(raw[k] >> 24) * kLutRep];
      if (VIG) {
        r.x = r.x * v[k].x;
        r.y = r.y * v[k].y;
        r.z = r.z * v[k].z;
        r.w = r.w * v[k].w;
      }
      if (ok[k]) __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(dst + p[k]));
    }
  }
}

// unMapImage, scalar path for pixel counts / bases the vector path cannot take.
template <bool VIG>
__global__ __launch_bounds__(256) void unmap_scalar_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                           const float* __restrict__ lut,
                                                           const float* __restrict__ vinv, long long npix,
                                                           int nframes, int fpb) {
  __shared__ float s_lut[256];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const float v = VIG ? vinv[i] : 1.f;
  const int f0 = blockIdx.y * fpb;
  const int f1 = min(nframes, f0 + fpb);
  for (int f = f0; f < f1; f++) {
    float r = s_lut[in[(long long)f * npix + i]];
    if (VIG) r = r * v;
    out[(long long)f * npix + i] = r;
  }
}

// ----------------------------------------------------------------------------
// Bilinear coefficients of one output pixel, in the reference's operation order
// (src/FOVUndistorter.cpp:352-365).
// ----------------------------------------------------------------------------
struct Bilin {
  int xi, yi;
  float w11, w01, w10, w00;
};
__device__ __forceinline__ Bilin bilin_of(float xx, float yy) {
  Bilin b;
  b.xi = (int)xx;
  b.yi = (int)yy;
  xx -= (float)b.xi;
  yy -= (float)b.yi;
  const float xxyy = xx * yy;
  b.w11 = xxyy;
  b.w01 = yy - xxyy;
  b.w10 = xx - xxyy;
  b.w00 = ((1.f - xx) - yy) + xxyy;
  return b;
}
__device__ __forceinline__ float bilin_sum(const Bilin& b, float t00, float t10, float t01, float t11) {
  // xxyy*src[1+W] + (yy-xxyy)*src[W] + (xx-xxyy)*src[1] + (1-xx-yy+xxyy)*src[0], left to right
  return ((b.w11 * t11 + b.w01 * t01) + b.w10 * t10) + b.w00 * t00;
}

// ----------------------------------------------------------------------------
// Generic fused kernel: one output pixel per thread, taps gathered straight from
// global memory.  Legal for every remap and every alignment; used when the tiled
// kernel cannot be planned, and as an in-library cross-check.
// ----------------------------------------------------------------------------
template <bool VIG>
__global__ __launch_bounds__(256) void remap_gather_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                              RemapArgs a, int nframes, int fpb) {
  __shared__ float s_lut[256];
  s_lut[threadIdx.x] = a.lut[threadIdx.x];
  __syncthreads();
  const int n_out = a.out_w * a.out_h;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_out) return;
  const long long n_in = (long long)a.in_w * a.in_h;
  const float xx = a.rx[idx], yy = a.ry[idx];
  const bool black = xx < 0;
  Bilin b = bilin_of(black ? 0.f : xx, black ? 0.f : yy);
  const int s = b.xi + b.yi * a.in_w;
  float v00 = 1.f, v10 = 1.f, v01 = 1.f, v11 = 1.f;
  if (VIG && !black) {
    v00 = a.vinv[s];
    v10 = a.vinv[s + 1];
    v01 = a.vinv[s + a.in_w];
    v11 = a.vinv[s + a.in_w + 1];
  }
  const int f0 = blockIdx.y * fpb;
  const int f1 = min(nframes, f0 + fpb);
  for (int f = f0; f < f1; f++) {
    float r = 0.f;
    if (!black) {
      const uint8_t* src = in + (long long)f * n_in + s;
      float t00 = s_lut[src[0]], t10 = s_lut[src[1]], t01 = s_lut[src[a.in_w]], t11 = s_lut[src[a.in_w + 1]];
      if (VIG) {
        t00 = t00 * v00;
        t10 = t10 * v10;
        t01 = t01 * v01;
        t11 = t11 * v11;
      }
      r = bilin_sum(b, t00, t10, t01, t11);
    }
    out[(long long)f * n_out + idx] = r;
  }
}

__global__ __launch_bounds__(256) void remap_gather_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               RemapArgs a, int nframes, int fpb) {
  const int n_out = a.out_w * a.out_h;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_out) return;
  const long long n_in = (long long)a.in_w * a.in_h;
  const float xx = a.rx[idx], yy = a.ry[idx];
  const bool black = xx < 0;
  Bilin b = bilin_of(black ? 0.f : xx, black ? 0.f : yy);
  const int s = b.xi + b.yi * a.in_w;
  const int f0 = blockIdx.y * fpb;
  const int f1 = min(nframes, f0 + fpb);
  for (int f = f0; f < f1; f++) {
    float r = 0.f;
    if (!black) {
      const float* src = in + (long long)f * n_in + s;
      r = bilin_sum(b, src[0], src[1], src[a.in_w], src[a.in_w + 1]);
    }
    out[(long long)f * n_out + idx] = r;
  }
}

// ----------------------------------------------------------------------------
// Tiled fused kernel (the headline path).
//
// Workgroup = one kTileW x kTileH output tile, looping over `fpb` frames.
//   once per workgroup : remap -> LDS byte offset + 4 weights per output, the 4
//                        vignette factors of its taps (registers), LUT replicas (LDS);
//   once per frame     : the tile's source window of the raw u8 frame is copied
//                        HBM -> registers -> LDS in 16-byte chunks (each window row
//                        is a run of aligned 16-byte pieces: coalesced, every byte
//                        of the frame fetched by the workgroup at most once);
//                        then per output 4 byte taps, 4 conflict-free LUT reads,
//                        4 (+4) multiplies, 3 adds, one coalesced store.
// Lane = output column, so the 32 lanes of an LDS access group read ~43
// consecutive source bytes of one row: broadcast within a dword, distinct banks
// across dwords.  Two LDS window buffers: the loads of frame f+1 are issued before
// frame f is computed and land in the other buffer afterwards; one barrier per frame.
// The staging loads are unconditional (chunk index clamped to the window, so a
// surplus lane re-reads the last chunk) and the frame loop is instantiated per
// number of staging rounds R: straight-line code keeps the loads in flight across
// the compute phase instead of waiting at a divergent merge.
//
// XCD placement: the dispatcher deals workgroups round-robin over the 8 XCDs
// (block b -> XCD b%8).  The host hands over a table block -> tile (TilePlan::d_order)
// built so that each XCD owns a contiguous band of tiles; neighbouring tiles share
// source-window halo lines, which then hit in the same L2.  Speed only --
// correctness does not depend on placement.
// ----------------------------------------------------------------------------
struct TileThread {  // per-thread, frame-invariant
  Bilin bl[4];
  int off[4];         // LDS byte offset of tap (0,0) inside the window
  uint32_t obyte[4];  // byte offset of the output inside a frame (unsigned: scalar base + 32-bit lane offset addressing)
  bool inside[4], black[4];
  float v00[4], v10[4], v01[4], v11[4];
};

// EDGE = the tile sticks out of the output image: stores are predicated per output.
// Interior tiles store unconditionally, which keeps the store count of a frame
// known at compile time (exact s_waitcnt vmcnt(N) for the prefetched loads that
// were issued before them -- vmcnt retires in order on gfx9).
// TAPS selects how the two horizontally adjacent byte taps of a row are fetched from LDS:
//   0  two byte loads as written (hipcc fuses them into one ds_read_u16 at an arbitrary,
//      often odd, address)
//   1  two separate ds_read_u8
//   2  the two aligned dwords around the pair (ds_read2_b32) + v_alignbyte
typedef const __attribute__((address_space(3))) unsigned char* lds_u8_ptr;
typedef const volatile __attribute__((address_space(3))) unsigned char* lds_vu8_ptr;
typedef const __attribute__((address_space(3))) uint32_t* lds_u32_ptr;

template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
#if MDC_EXP_LOAD_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename T>
__device__ __forceinline__ void stream_store(T v, T* p) {
#if MDC_EXP_SKIP_STORE
  if (v != (T)-1.2345e30f) return;
#endif
#if MDC_EXP_STORE_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

template <int TAPS>
__device__ __forceinline__ void tap_pair(const unsigned char* p, int& a, int& b) {
  if (TAPS == 0) {
    lds_u8_ptr q = (lds_u8_ptr)p;
    a = q[0];
    b = q[1];
  } else if (TAPS == 1) {
    lds_vu8_ptr q = (lds_vu8_ptr)p;
    a = q[0];
    b = q[1];
  } else {
    lds_u8_ptr q8 = (lds_u8_ptr)p;
    const uint32_t addr = (uint32_t)(uintptr_t)q8;  // LDS byte address (32-bit in address space 3)
    lds_u32_ptr q = (lds_u32_ptr)(uintptr_t)(addr & ~3u);
    const uint32_t lo = q[0], hi = q[1];
    const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, addr & 3u);
    a = w & 255u;
    b = (w >> 8) & 255u;
  }
}

template <bool VIG, int LUTREP, bool EDGE, int TAPS>
__device__ __forceinline__ void tile_compute(const TileThread& t, const unsigned char* __restrict__ w, int pitch,
                                             const float* __restrict__ my_lut, float* __restrict__ dst) {
#if MDC_EXP_BATCHED
  int b00[4], b10[4], b01[4], b11[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned char* p = w + t.off[j];
    tap_pair<TAPS>(p, b00[j], b10[j]);
    tap_pair<TAPS>(p + pitch, b01[j], b11[j]);
  }
  float t00[4], t10[4], t01[4], t11[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    t00[j] = my_lut[b00[j] * LUTREP];
    t10[j] = my_lut[b10[j] * LUTREP];
    t01[j] = my_lut[b01[j] * LUTREP];
    t11[j] = my_lut[b11[j] * LUTREP];
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (VIG) {
      t00[j] = t00[j] * t.v00[j];
      t10[j] = t10[j] * t.v10[j];
      t01[j] = t01[j] * t.v01[j];
      t11[j] = t11[j] * t.v11[j];
    }
    float r = bilin_sum(t.bl[j], t00[j], t10[j], t01[j], t11[j]);
    if (t.black[j]) r = 0.f;
    if (!EDGE || t.inside[j]) stream_store(r, reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + t.obyte[j]));
  }
#else
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned char* p = w + t.off[j];
    int b00, b10, b01, b11;
    tap_pair<TAPS>(p, b00, b10);
    tap_pair<TAPS>(p + pitch, b01, b11);
    float t00 = my_lut[b00 * LUTREP];
    float t10 = my_lut[b10 * LUTREP];
    float t01 = my_lut[b01 * LUTREP];
    float t11 = my_lut[b11 * LUTREP];
    if (VIG) {
      t00 = t00 * t.v00[j];
      t10 = t10 * t.v10[j];
      t01 = t01 * t.v01[j];
      t11 = t11 * t.v11[j];
    }
    float r = bilin_sum(t.bl[j], t00, t10, t01, t11);
    if (t.black[j]) r = 0.f;
    if (!EDGE || t.inside[j]) stream_store(r, reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + t.obyte[j]));
  }
#endif
}

template <bool VIG, int LUTREP, int R, bool EDGE, int TAPS, int NT>
__device__ __forceinline__ void tile_frames(const TileThread& t, const uint8_t* __restrict__ src,
                                            float* __restrict__ dst, long long n_in, long long n_out, int nframes,
                                            int nch, const TileDesc& td, int in_w, unsigned char* s_win,
                                            int win_bytes, const float* my_lut, int tid) {
  const int pitch = td.cpr * 16;
  uint32_t goff[R];
  int loff[R];
#pragma unroll
  for (int k = 0; k < R; k++) {
    const int c = min(tid + k * NT, nch - 1);
    const int r = c / td.cpr;
    goff[k] = (uint32_t)((td.y0 + r) * in_w + td.x0 + (c - r * td.cpr) * 16);
    loff[k] = c * 16;
  }
  u32x4 stage[R];
#pragma unroll
  for (int k = 0; k < R; k++) stage[k] = stream_load(reinterpret_cast<const u32x4*>(src + goff[k]));
#pragma unroll
  for (int k = 0; k < R; k++) *reinterpret_cast<u32x4*>(s_win + loff[k]) = stage[k];
  __syncthreads();
  int cur = 0;
  for (int f = 0; f < nframes - 1; f++) {
#if !MDC_EXP_SKIP_LOAD
    src += n_in;
#endif
#pragma unroll
    for (int k = 0; k < R; k++) stage[k] = stream_load(reinterpret_cast<const u32x4*>(src + goff[k]));
    tile_compute<VIG, LUTREP, EDGE, TAPS>(t, s_win + cur * win_bytes, pitch, my_lut, dst);
    dst += n_out;
    unsigned char* wn = s_win + (cur ^ 1) * win_bytes;
#pragma unroll
    for (int k = 0; k < R; k++) *reinterpret_cast<u32x4*>(wn + loff[k]) = stage[k];
    __syncthreads();
    cur ^= 1;
  }
  tile_compute<VIG, LUTREP, EDGE, TAPS>(t, s_win + cur * win_bytes, pitch, my_lut, dst);
}

// Same, with the staging loads running two frames ahead: the chunks of frame f+2 are requested
// before frame f is computed, those of frame f+1 (requested one iteration earlier) are written to
// the other LDS buffer afterwards.  Unrolled by two so the two register stages need no moves.
template <bool VIG, int LUTREP, int R, bool EDGE, int TAPS, int NT>
__device__ __forceinline__ void tile_frames_pf2(const TileThread& t, const uint8_t* __restrict__ src,
                                                float* __restrict__ dst, long long n_in, long long n_out, int nframes,
                                                int nch, const TileDesc& td, int in_w, unsigned char* s_win,
                                                int win_bytes, const float* my_lut, int tid) {
  const int pitch = td.cpr * 16;
  uint32_t goff[R];
  int loff[R];
#pragma unroll
  for (int k = 0; k < R; k++) {
    const int c = min(tid + k * NT, nch - 1);
    const int r = c / td.cpr;
    goff[k] = (uint32_t)((td.y0 + r) * in_w + td.x0 + (c - r * td.cpr) * 16);
    loff[k] = c * 16;
  }
  u32x4 sa[R], sb[R];
  const int last = nframes - 1;
#pragma unroll
  for (int k = 0; k < R; k++) sa[k] = stream_load(reinterpret_cast<const u32x4*>(src + goff[k]));
#pragma unroll
  for (int k = 0; k < R; k++) *reinterpret_cast<u32x4*>(s_win + loff[k]) = sa[k];
  {
    const uint8_t* p1 = src + (long long)min(1, last) * n_in;
#pragma unroll
    for (int k = 0; k < R; k++) sa[k] = stream_load(reinterpret_cast<const u32x4*>(p1 + goff[k]));
  }
  __syncthreads();
  unsigned char* w0 = s_win;
  unsigned char* w1 = s_win + win_bytes;
  int f = 0;
  for (; f + 2 <= last; f += 2) {
    {  // frame f from w0; request f+2 -> sb; land f+1 (sa) in w1
      const uint8_t* p = src + (long long)(f + 2) * n_in;
#pragma unroll
      for (int k = 0; k < R; k++) sb[k] = stream_load(reinterpret_cast<const u32x4*>(p + goff[k]));
      tile_compute<VIG, LUTREP, EDGE, TAPS>(t, w0, pitch, my_lut, dst);
      dst += n_out;
#pragma unroll
      for (int k = 0; k < R; k++) *reinterpret_cast<u32x4*>(w1 + loff[k]) = sa[k];
      __syncthreads();
    }
    {  // frame f+1 from w1; request f+3 -> sa; land f+2 (sb) in w0
      const uint8_t* p = src + (long long)min(f + 3, last) * n_in;
#pragma unroll
      for (int k = 0; k < R; k++) sa[k] = stream_load(reinterpret_cast<const u32x4*>(p + goff[k]));
      tile_compute<VIG, LUTREP, EDGE, TAPS>(t, w1, pitch, my_lut, dst);
      dst += n_out;
#pragma unroll
      for (int k = 0; k < R; k++) *reinterpret_cast<u32x4*>(w0 + loff[k]) = sb[k];
      __syncthreads();
    }
  }
  // here: w0 holds frame f, sa holds frame min(f+1, last); f == last or f == last-1
  tile_compute<VIG, LUTREP, EDGE, TAPS>(t, w0, pitch, my_lut, dst);
  if (f < last) {
    dst += n_out;
#pragma unroll
    for (int k = 0; k < R; k++) *reinterpret_cast<u32x4*>(w1 + loff[k]) = sa[k];
    __syncthreads();
    tile_compute<VIG, LUTREP, EDGE, TAPS>(t, w1, pitch, my_lut, dst);
  }
}

template <bool VIG, int LUTREP, int TAPS, int NT>
__global__ __launch_bounds__(NT, (NT == 512 ? MDC_EXP_WAVES_512 : MDC_EXP_WAVES)) void remap_tiled_u8_kernel(const uint8_t* __restrict__ in,
                                                                      float* __restrict__ out, RemapArgs a,
                                                                      const TileDesc* __restrict__ tiles,
                                                                      const int* __restrict__ order, int tiles_x,
                                                                      int win_bytes, int nframes, int fpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_lut = reinterpret_cast<float*>(smem);
  unsigned char* s_win = smem + 256 * LUTREP * 4;

  const int tile = order[blockIdx.x];  // host-made placement table (plan_tiles); -1 = padding slot
  if (tile < 0) return;                // whole workgroup leaves before any barrier
  const int f0 = blockIdx.y * fpb;
  const int nf = min(nframes, f0 + fpb) - f0;
  if (nf <= 0) return;

  const int tid = threadIdx.x;
  const int lane_x = tid % kTileW;
  const int row0 = (tid / kTileW) * 4;
  const TileDesc td = tiles[tile];
  const int pitch = td.cpr * 16;
  const int ox = (tile % tiles_x) * kTileW + lane_x;
  constexpr int kTileRows = NT / 16;  // 4 output rows per thread, kTileW lanes per row
  const int oy0 = (tile / tiles_x) * kTileRows + row0;

#pragma unroll 4
  for (int i = tid; i < 256 * LUTREP; i += NT) s_lut[i] = a.lut[i / LUTREP];
  const float* my_lut = s_lut + (tid & (LUTREP - 1));

  TileThread t;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int oy = oy0 + j;
    t.inside[j] = (ox < a.out_w) && (oy < a.out_h);
    const int oidx = oy * a.out_w + ox;
    t.obyte[j] = (uint32_t)oidx * 4u;
    float xx = -1.f, yy = -1.f;
    if (t.inside[j]) {
      xx = a.rx[oidx];
      yy = a.ry[oidx];
    }
    t.black[j] = xx < 0;
    t.bl[j] = bilin_of(t.black[j] ? 0.f : xx, t.black[j] ? 0.f : yy);
    t.off[j] = t.black[j] ? 0 : (t.bl[j].yi - td.y0) * pitch + (t.bl[j].xi - td.x0);
    t.v00[j] = t.v10[j] = t.v01[j] = t.v11[j] = 1.f;
    if (VIG && !t.black[j]) {
      const int s = t.bl[j].xi + t.bl[j].yi * a.in_w;
      t.v00[j] = a.vinv[s];
      t.v10[j] = a.vinv[s + 1];
      t.v01[j] = a.vinv[s + a.in_w];
      t.v11[j] = a.vinv[s + a.in_w + 1];
    }
  }

  const long long n_in = (long long)a.in_w * a.in_h;
  const long long n_out = (long long)a.out_w * a.out_h;
  const uint8_t* src = in + (long long)f0 * n_in;
  float* dst = out + (long long)f0 * n_out;
  const int nch = td.rows * td.cpr;
  if (nch == 0) {  // every output of the tile is black (or outside): zeros, no staging
    for (int f = 0; f < nf; f++, dst += n_out)
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (t.inside[j]) *reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + t.obyte[j]) = 0.f;
    return;
  }
  const int rounds = (nch + NT - 1) / NT;  // workgroup-uniform
  const bool edge = ((tile % tiles_x) + 1) * kTileW > a.out_w || ((tile / tiles_x) + 1) * kTileRows > a.out_h;
#if MDC_EXP_PF2
#define MDC_TILE_FRAMES tile_frames_pf2
#else
#define MDC_TILE_FRAMES tile_frames
#endif
#define MDC_TILE_RUN(R_, E_) \
  MDC_TILE_FRAMES<VIG, LUTREP, R_, E_, TAPS, NT>(t, src, dst, n_in, n_out, nf, nch, td, a.in_w, s_win, win_bytes, my_lut, tid)
  if (!edge) {
    if (rounds == 1) MDC_TILE_RUN(1, false);
    else if (rounds == 2) MDC_TILE_RUN(2, false);
    else MDC_TILE_RUN(kTileMaxChunks, false);
  } else {
    if (rounds == 1) MDC_TILE_RUN(1, true);
    else if (rounds == 2) MDC_TILE_RUN(2, true);
    else MDC_TILE_RUN(kTileMaxChunks, true);
  }
#undef MDC_TILE_RUN
}

// ----------------------------------------------------------------------------
// 2x2 box pyramid level (config 5; not in the reference): 0.25f*(((a+b)+c)+d).
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pyramid_level_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int w, int h, long long nframes) {
  const int w2 = w >> 1, h2 = h >> 1;
  const long long n2 = (long long)w2 * h2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n2 * nframes) return;
  const long long f = i / n2;
  const int r = (int)(i - f * n2);
  const int y = r / w2, x = r - y * w2;
  const float* p = src + f * (long long)w * h + (long long)(2 * y) * w + 2 * x;
  dst[i] = 0.25f * (((p[0] + p[1]) + p[w]) + p[w + 1]);
}

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
__global__ __launch_bounds__(256) void synth_kernel(uint8_t* __restrict__ out, long long first_pix, long long n,
                                                    uint32_t seed) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  uint32_t word = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t b = fmix32(seed + (uint32_t)(first_pix + i + k)) >> 24;
    word |= b << (8 * k);
  }
  if (i + 3 < n && ((reinterpret_cast<uintptr_t>(out + i) & 3) == 0)) *reinterpret_cast<uint32_t*>(out + i) = word;
  else
    for (int k = 0; k < 4 && i + k < n; k++) out[i + k] = (uint8_t)(word >> (8 * k));
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace

size_t tiled_lds_bytes(int win_bytes, int lut_rep) { return (size_t)256 * lut_rep * 4 + 2 * (size_t)win_bytes; }

hipError_t launch_unmap(const uint8_t* d_in, float* d_out, const float* d_lut, const float* d_vinv, int64_t npix,
                        int64_t nframes, int fpb, hipStream_t s) {
  if (nframes <= 0 || npix <= 0) return hipSuccess;
  const int groups = ceil_div(nframes, fpb);
  const bool vec = (npix % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_in) & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(d_out) & 15) == 0) &&
                   (!d_vinv || (reinterpret_cast<uintptr_t>(d_vinv) & 15) == 0);
  if (vec) {
    dim3 grid(ceil_div(npix, 4096), groups);
    if (d_vinv) unmap_vec_kernel<true><<<grid, 256, 0, s>>>(d_in, d_out, d_lut, d_vinv, npix, (int)nframes, fpb);
    else unmap_vec_kernel<false><<<grid, 256, 0, s>>>(d_in, d_out, d_lut, d_vinv, npix, (int)nframes, fpb);
  } else {
    dim3 grid(ceil_div(npix, 256), groups);
    if (d_vinv) unmap_scalar_kernel<true><<<grid, 256, 0, s>>>(d_in, d_out, d_lut, d_vinv, npix, (int)nframes, fpb);
    else unmap_scalar_kernel<false><<<grid, 256, 0, s>>>(d_in, d_out, d_lut, d_vinv, npix, (int)nframes, fpb);
  }
  return hipGetLastError();
}

hipError_t launch_remap_gather_u8(const uint8_t* d_in, float* d_out, const RemapArgs& a, int64_t nframes, int fpb,
                                  hipStream_t s) {
  if (nframes <= 0) return hipSuccess;
  dim3 grid(ceil_div((long long)a.out_w * a.out_h, 256), ceil_div(nframes, fpb));
  if (a.vinv) remap_gather_u8_kernel<true><<<grid, 256, 0, s>>>(d_in, d_out, a, (int)nframes, fpb);
  else remap_gather_u8_kernel<false><<<grid, 256, 0, s>>>(d_in, d_out, a, (int)nframes, fpb);
  return hipGetLastError();
}

hipError_t launch_remap_gather_f32(const float* d_in, float* d_out, const RemapArgs& a, int64_t nframes, int fpb,
                                   hipStream_t s) {
  if (nframes <= 0) return hipSuccess;
  dim3 grid(ceil_div((long long)a.out_w * a.out_h, 256), ceil_div(nframes, fpb));
  remap_gather_f32_kernel<<<grid, 256, 0, s>>>(d_in, d_out, a, (int)nframes, fpb);
  return hipGetLastError();
}

template <bool VIG, int LUTREP, int TAPS, int NT>
static hipError_t launch_tiled_variant(const uint8_t* d_in, float* d_out, const RemapArgs& a, const TilePlan& p,
                                       int64_t nframes, int fpb, hipStream_t s) {
  dim3 grid(p.n_blocks, ceil_div(nframes, fpb));
  const size_t lds = tiled_lds_bytes(p.win_bytes, LUTREP);
  remap_tiled_u8_kernel<VIG, LUTREP, TAPS, NT><<<grid, NT, lds, s>>>(d_in, d_out, a, p.d_tiles, p.d_order, p.tiles_x,
                                                                      p.win_bytes, (int)nframes, fpb);
  return hipGetLastError();
}

template <bool VIG, int LUTREP, int TAPS>
static hipError_t launch_tiled_nt(const uint8_t* d_in, float* d_out, const RemapArgs& a, const TilePlan& p,
                                  int64_t nframes, int fpb, hipStream_t s) {
  return p.tile_h == 32 ? launch_tiled_variant<VIG, LUTREP, TAPS, 512>(d_in, d_out, a, p, nframes, fpb, s)
                        : launch_tiled_variant<VIG, LUTREP, TAPS, 256>(d_in, d_out, a, p, nframes, fpb, s);
}

template <bool VIG>
static hipError_t launch_tiled_vig(const uint8_t* d_in, float* d_out, const RemapArgs& a, const TilePlan& p,
                                   int64_t nframes, int fpb, int lut_rep, int taps, hipStream_t s) {
  if (lut_rep == 16)
    return taps == 2 ? launch_tiled_nt<VIG, 16, 2>(d_in, d_out, a, p, nframes, fpb, s)
                     : launch_tiled_nt<VIG, 16, 1>(d_in, d_out, a, p, nframes, fpb, s);
  return taps == 2 ? launch_tiled_nt<VIG, 32, 2>(d_in, d_out, a, p, nframes, fpb, s)
                   : launch_tiled_nt<VIG, 32, 1>(d_in, d_out, a, p, nframes, fpb, s);
}

hipError_t launch_remap_tiled_u8(const uint8_t* d_in, float* d_out, const RemapArgs& a, const TilePlan& p,
                                 int64_t nframes, int fpb, int lut_rep, int taps, hipStream_t s) {
  if (nframes <= 0) return hipSuccess;
  return a.vinv ? launch_tiled_vig<true>(d_in, d_out, a, p, nframes, fpb, lut_rep, taps, s)
                : launch_tiled_vig<false>(d_in, d_out, a, p, nframes, fpb, lut_rep, taps, s);
}

hipError_t launch_pyramid_level(const float* d_src, float* d_dst, int w, int h, int64_t nframes, hipStream_t s) {
  const long long n = (long long)(w >> 1) * (h >> 1) * nframes;
  if (n <= 0) return hipSuccess;
  pyramid_level_kernel<<<ceil_div(n, 256), 256, 0, s>>>(d_src, d_dst, w, h, nframes);
  return hipGetLastError();
}

hipError_t launch_synth(uint8_t* d_out, int64_t first_frame, int64_t nframes, int npix, uint32_t seed,
                        hipStream_t s) {
  const long long n = (long long)nframes * npix;
  if (n <= 0) return hipSuccess;
  synth_kernel<<<ceil_div(n, 1024), 256, 0, s>>>(d_out, first_frame * (long long)npix, n, seed);
  return hipGetLastError();
}

}  // namespace mdc
